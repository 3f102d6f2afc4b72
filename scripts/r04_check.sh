#!/bin/bash
# Round-4 mid-round check on the GPU box: full -m gpu suite, then the default bench line, then the rolling-refresh A/B.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out
python -m pytest tests -m gpu -q --durations=8 > $O/t2_all.log 2>&1; tail -15 $O/t2_all.log
python bench.py --no-other-configs --steps 20 --warmup 4 > $O/t2_bench.json 2> $O/t2_bench.err
for RE in 16 64 256; do
  python bench.py --no-other-configs --no-cpu-baseline --steps 20 --warmup 4 --replan-every $RE --replan-count 256 > $O/t2_bench_replan$RE.json 2>> $O/t2_bench.err
done
UAVENV_REPLAN_WGS=512 python bench.py --no-other-configs --no-cpu-baseline --steps 20 --warmup 4 --replan-every 16 --replan-count 1024 > $O/t2_bench_replan16_w512.json 2>> $O/t2_bench.err
python - <<'PY'
import json
for f in ("t2_bench", "t2_bench_replan16", "t2_bench_replan64", "t2_bench_replan256", "t2_bench_replan16_w512"):
    try:
        d = json.loads(open("gpurun_out/%s.json" % f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f, "value %.4g ms/pass %.5f frac %.4f phys_stored %.4f phys_cnt %s grad_ms %.5f" % (d["value"], d["ms_per_pass"], r["frac"], r["frac_physical_stored"], r["frac_physical_counters"], d["roofline_learner"]["kernel_ms"]))
        print("   resets", json.dumps(d["config"]["resets"])[:600])
    except Exception as ex:
        print(f, "FAILED", ex)
PY
tail -5 $O/t2_bench.err
