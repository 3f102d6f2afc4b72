#!/bin/bash
# A/B under rocprofv3 --kernel-trace --stats: the step / grad / reduce averages of the configs[1] loop for four builds
# (wave priority on / off x the placement of step_pre's distances).  Round 6: profiles/r06_ab_prio_and_step_pre_under_rocprof.txt
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
one() { rm -rf /tmp/prof_ab; (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ab -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --no-reset-count --env-only-iters 50) > /tmp/ab.log 2>&1; f=$(find /tmp/prof_ab -name "*kernel_stats.csv" | head -1); python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-other-configs --no-reset-count --full-line 2>/dev/null | tail -1 > /tmp/ab_un.json; python - "$f" "$1" <<'PY'
import csv, sys, json
rows = {}
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    for k in ("k_step_coop<unsigned int, false, 2, true", "k_dqn_grad_packed8", "k_dqn_reduce_adam"):
        if k in n:
            rows[k] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3)
u = json.loads(open("/tmp/ab_un.json").read())
pr = [l for l in open("/tmp/ab.log") if l.startswith("{")]
p = json.loads(pr[-1]) if pr else {}
print("%-40s rocprof avg us: step %.2f grad %.2f reduce %.2f (sum %.2f) | pass under rocprof %.2f | unprofiled pass %.2f, step b2b %.2f" % (
    sys.argv[2], rows["k_step_coop<unsigned int, false, 2, true"][1], rows["k_dqn_grad_packed8"][1], rows["k_dqn_reduce_adam"][1],
    sum(v[1] for v in rows.values()), 1e3 * p.get("ms_per_pass", float("nan")), 1e3 * u["ms_per_pass"], 1e3 * u["roofline"]["kernel_ms_back_to_back"]))
PY
}
one "current (prio 3, dists early)"
UAVENV_EXTRA_FLAGS=-DUAVENV_HOT_PRIO=0 python -c "from dqn_based_uav_3d_path_planer_amd import _build; _build.build(force=True)" > /dev/null 2>&1
one "prio 0, dists early"
UAVENV_EXTRA_FLAGS=-DUAVENV_PRE_DISTS_LATE python -c "from dqn_based_uav_3d_path_planer_amd import _build; _build.build(force=True)" > /dev/null 2>&1
one "prio 3, dists late"
UAVENV_EXTRA_FLAGS="-DUAVENV_PRE_DISTS_LATE -DUAVENV_HOT_PRIO=0" python -c "from dqn_based_uav_3d_path_planer_amd import _build; _build.build(force=True)" > /dev/null 2>&1
one "prio 0, dists late (= round 5)"
one "prio 0, dists late (= round 5), again"
