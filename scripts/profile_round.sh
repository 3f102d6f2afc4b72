#!/bin/bash
# Run on the GPU box: rocprofv3 kernel-trace stats of a bench command + PMC passes (separate runs, --kernel-trace only)
# for k_step and k_dqn_grad.  usage: scripts/profile_round.sh <tag> [bench args...]   -> gpurun_out/<tag>_*
TAG=$1; shift
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
ARGS="--steps ${PROF_STEPS:-2} --warmup 1 --no-cpu-baseline --no-other-configs --no-reset-count --env-only-iters 50 $*"
rm -rf /tmp/prof_$TAG
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o bench -- python $GRAFT_REPO_ROOT/bench.py $ARGS) > $OUT/${TAG}_stats.log 2>&1
find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/${TAG}_kernel_stats.csv
tail -1 $OUT/${TAG}_stats.log > $OUT/${TAG}_bench_under_rocprof.json
head -12 $OUT/${TAG}_kernel_stats.csv
PARGS="--steps 1 --warmup 1 --no-cpu-baseline --no-other-configs --no-reset-count --env-only-iters 20 $*"
: > $OUT/${TAG}_pmc.txt
# PMC_SETS=traffic: only the two HBM-traffic passes (FETCH_SIZE, WRITE_SIZE: separate passes, MI355X_MICROARCH.md)
if [ "$PMC_SETS" = "traffic" ]; then
  SETS=("FETCH_SIZE" "WRITE_SIZE")
else
  SETS=("SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
        "SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_VALU_MFMA_F16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" \
        "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE GRBM_COUNT")
fi
for PMC in "${SETS[@]}"; do
  N=$(echo $PMC | cut -d' ' -f1)
  rm -rf /tmp/pmc_$N
  (cd /tmp && rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d /tmp/pmc_$N -o pmc -- python $GRAFT_REPO_ROOT/bench.py $PARGS) > /tmp/pmc_$N.log 2>&1
  F=$(find /tmp/pmc_$N -name "*counter_collection.csv" | head -1)
  python - "$F" "$TAG" <<'PY' | tee -a $OUT/${TAG}_pmc.txt
import csv, sys, collections, re
f = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
try:
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        for short in ("k_step", "k_dqn_grad", "k_dqn_act", "k_dqn_reduce_adam", "k_apf_adjust", "k_sac_critic_grad", "k_sac_actor_grad", "k_sac_reduce_adam", "k_sac"):
            if short in k:
                m = re.search(r"k_step_coop<([^>]*)>", k)
                pol = bool(m) and len(m.group(1).split(",")) >= 4 and m.group(1).split(",")[3].strip() == "true"
                if short == "k_step" and (pol or "k_step_polh" in k):
                    short = "k_step_policy"          # the step kernel with the policy in its prologue
                agg[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
                break
    for k, d in agg.items():
        for c, v in d.items():
            print(sys.argv[2], k, c, "n=%d mean=%.1f" % (len(v), sum(v) / len(v)))
except Exception as e:
    print("pmc parse failed", e, f)
PY
done
