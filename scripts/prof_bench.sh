#!/bin/bash
# Run on the GPU box: kernel-trace stats of the default bench + PMC passes for k_step.  Outputs under gpurun_out/prof_*.
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
ARGS="--steps 100 --warmup 10 --no-cpu-baseline"
(cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/prof_stats -o bench -- python $GRAFT_REPO_ROOT/bench.py $ARGS) > $OUT/prof_stats.log 2>&1
find $OUT/prof_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
head -30 $OUT/kernel_stats.csv
for PMC in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"; do
  TAG=$(echo $PMC | cut -d' ' -f1)
  (cd /tmp && rocprofv3 --kernel-trace --pmc $PMC -d $OUT/prof_pmc_$TAG -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --env-only-iters 20) > $OUT/prof_pmc_$TAG.log 2>&1
  F=$(find $OUT/prof_pmc_$TAG -name "*counter_collection.csv" | head -1)
  python - "$F" <<'PY'
import csv, sys, collections
f=sys.argv[1]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
try:
    for r in csv.DictReader(open(f)):
        k=r.get("Kernel_Name","")
        if "k_step" in k or "k_sample" in k:
            agg[k[:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,d in agg.items():
        for c,v in d.items(): print(k, c, "n=%d mean=%.1f"%(len(v), sum(v)/len(v)))
except Exception as e: print("pmc parse failed", e, f)
PY
done
