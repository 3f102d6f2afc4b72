#!/bin/bash
# PMC counters for k_step (separate passes, --kernel-trace only).  usage: scripts/pmc_kstep.sh <tag> [bench args...]
TAG=$1; shift
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
for PMC in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  N=$(echo $PMC | cut -d' ' -f1)
  rm -rf /tmp/pmc_$N
  (cd /tmp && rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d /tmp/pmc_$N -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --env-only-iters 20 "$@") > /tmp/pmc_$N.log 2>&1
  F=$(find /tmp/pmc_$N -name "*counter_collection.csv" | head -1)
  python - "$F" "$TAG" <<'PY' | tee -a $OUT/pmc_$TAG.txt
import csv, sys, collections
f=sys.argv[1]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
try:
    for r in csv.DictReader(open(f)):
        k=r.get("Kernel_Name","")
        if "k_step" in k:
            agg["k_step"][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,d in agg.items():
        for c,v in d.items(): print(sys.argv[2], k, c, "n=%d mean=%.1f"%(len(v), sum(v)/len(v)))
except Exception as e: print("pmc parse failed", e, f)
PY
done
