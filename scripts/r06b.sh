#!/bin/bash
# round 6, GPU call b: the new parity tests, the replan sweep, the k_step_coop<policy> phase stamps
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
python -m pytest tests/test_checkpoint_interchange.py tests/test_env_parity_gpu.py -m gpu -q -k "checkpoint or apf" -s 2>&1 | tail -15 > $O/r06b_tests.log
: > $O/r06b_replan_sweep.jsonl
for c in 4096 8192 16384; do for e in 64 256; do
  python bench.py --no-cpu-baseline --no-other-configs --full-line --replan-every $e --replan-count $c --steps 20 --warmup 4 2>/dev/null | tail -1 >> $O/r06b_replan_sweep.jsonl
done; done
UAVENV_REPLAN_WGS=256 python bench.py --no-cpu-baseline --no-other-configs --full-line --replan-every 256 --replan-count 8192 --steps 20 --warmup 4 2>/dev/null | tail -1 >> $O/r06b_replan_sweep.jsonl
python - <<'PY' > $O/r06b_replan_sweep.txt
import json
for l in open('gpurun_out/r06b_replan_sweep.jsonl'):
    try: d=json.loads(l)
    except Exception as e: print('bad line', e); continue
    r=d['config']['resets']; f=r.get('refresh') or {}
    print(f.get('every_passes'), f.get('rows_per_slice'), 'ms/pass %.5f'%d['ms_per_pass'], 'value %.4g'%d['value'], 'consumed/s %.0f'%r['consumed_per_s'], 'committed/s %.0f'%f.get('rows_committed_per_s',0), {k:f.get(k) for k in ('refreshes','rows_planned','rows_committed','rows_in_use')})
PY
cat $O/r06b_replan_sweep.txt
UAVENV_PHASE_PROFILE=1 python -c "from dqn_based_uav_3d_path_planer_amd import _build; _build.build(force=True)" 2>&1 | tail -2
POLICY=1 python scripts/phase_profile_coop.py 16384 > $O/r06b_phase_coop_policy.txt 2>&1
python scripts/phase_profile_coop.py 16384 > $O/r06b_phase_coop.txt 2>&1
cat $O/r06b_phase_coop_policy.txt; cat $O/r06b_tests.log
