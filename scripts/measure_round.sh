#!/bin/bash
# Round measurement on the GPU box: default bench (JSON), rocprofv3 kernel-trace stats of the SAME command, PMC passes.
# usage: scripts/measure_round.sh <tag>     outputs: gpurun_out/<tag>_*
TAG=${1:-r01}
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; tail -c 1500 $OUT/${TAG}_bench.json
rm -rf /tmp/prof_$TAG
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline) > /tmp/prof_$TAG.log 2>&1
cp $(find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_bench_kernel_stats.csv
head -8 $OUT/${TAG}_bench_kernel_stats.csv | cut -c1-160
rm -f $OUT/pmc_$TAG.txt
scripts/pmc_kstep.sh $TAG 2>&1 | tail -25
