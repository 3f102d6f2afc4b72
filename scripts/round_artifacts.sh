#!/bin/bash
# Run on the GPU box: every artifact a round commits under profiles/ (default-bench stats + PMC via profile_round.sh, the other
# configs, config 3 / 4 / APF kernel stats, env-only points).  usage: scripts/round_artifacts.sh <tag>   -> gpurun_out/<tag>_*
TAG=${1:-round}
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out
bash scripts/profile_round.sh ${TAG} > $O/${TAG}_profile.log 2>&1
python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
python bench.py --config 3 --no-cpu-baseline > $O/${TAG}_config3_bench.json 2> $O/${TAG}_config3.err
python bench.py --config 4 --steps 4 --warmup 1 --no-cpu-baseline > $O/${TAG}_config4_bench.json 2> $O/${TAG}_config4.err
python bench.py --config 5 --no-cpu-baseline > $O/${TAG}_config5_bench.json 2> $O/${TAG}_config5.err
rm -rf /tmp/prof_c4; (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c4 -o bench -- python $GRAFT_REPO_ROOT/bench.py --config 4 --steps 2 --warmup 1 --no-cpu-baseline) > /tmp/c4.log 2>&1
find /tmp/prof_c4 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${TAG}_config4_kernel_stats.csv
rm -rf /tmp/prof_c3; (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c3 -o bench -- python $GRAFT_REPO_ROOT/bench.py --config 3 --steps 6 --warmup 2 --no-cpu-baseline) > /tmp/c3.log 2>&1
find /tmp/prof_c3 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${TAG}_config3_kernel_stats.csv
rm -rf /tmp/prof_apf; (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_apf -o bench -- python $GRAFT_REPO_ROOT/bench.py --env-only --envs 32768 --uav-per-env 4 --apf --no-cpu-baseline --env-only-iters 50) > /tmp/apf.log 2>&1
find /tmp/prof_apf -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${TAG}_apf_envonly_kernel_stats.csv
tail -1 /tmp/apf.log > $O/${TAG}_apf_envonly.json
for n in 65536 262144; do python bench.py --env-only --envs $n --no-cpu-baseline 2>/dev/null | tail -1 > $O/${TAG}_envonly_$n.json; done
for f in ${TAG}_bench ${TAG}_config3_bench ${TAG}_config4_bench ${TAG}_config5_bench; do python -c "import json,sys; d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d.get('ms_per_pass'))"; done
head -8 $O/${TAG}_kernel_stats.csv | cut -c1-150
