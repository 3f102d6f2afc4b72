#!/bin/bash
# PMC counters for k_step in env-only mode. usage: scripts/pmc_envonly.sh <tag> <envs> [extra bench args]
TAG=$1; N=$2; shift; shift
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT; rm -f $OUT/pmc_$TAG.txt
for PMC in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS" "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "GRBM_GUI_ACTIVE"; do
  NME=$(echo $PMC | cut -d' ' -f1)
  rm -rf /tmp/pmc_$NME
  (cd /tmp && rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d /tmp/pmc_$NME -o pmc -- python $GRAFT_REPO_ROOT/bench.py --env-only --envs $N --replay $((N*4)) --steps 20 "$@") > /tmp/pmc_$NME.log 2>&1
  F=$(find /tmp/pmc_$NME -name "*counter_collection.csv" | head -1)
  python - "$F" "$TAG" <<'PY' | tee -a $OUT/pmc_$TAG.txt
import csv, sys, collections
f=sys.argv[1]
agg=collections.defaultdict(list)
try:
    rows=[r for r in csv.DictReader(open(f)) if "k_step" in r.get("Kernel_Name","")]
    # keep the last 20 launches per counter (steady state)
    by=collections.defaultdict(list)
    for r in rows: by[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c,v in by.items():
        v=v[-20:]; print(sys.argv[2], "k_step", c, "n=%d mean=%.1f"%(len(v), sum(v)/len(v)))
except Exception as e: print("pmc parse failed", e, f)
PY
done
