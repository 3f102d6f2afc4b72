#!/bin/bash
# On the GPU box: the counter calibration, the reset-bank refresh sweep (slice size x planner wavefronts: fresh plans per second
# against the pass time) and the env-only sweep over agents per launch.  Round 6: profiles/r06_replan_sweep.txt, r06_envonly_sweep.txt.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
bash scripts/calib/run_calibration.sh r06 > /dev/null 2>&1
: > $O/r06c_sweep.jsonl
python bench.py --no-cpu-baseline --no-other-configs --full-line --steps 20 --warmup 4 2>/dev/null | tail -1 >> $O/r06c_sweep.jsonl
for c in 8192 16384; do
  python bench.py --no-cpu-baseline --no-other-configs --full-line --replan-every 256 --replan-count $c --steps 20 --warmup 4 2>/dev/null | tail -1 >> $O/r06c_sweep.jsonl
done
UAVENV_REPLAN_WGS=256 python bench.py --no-cpu-baseline --no-other-configs --full-line --replan-every 256 --replan-count 16384 --steps 20 --warmup 4 2>/dev/null | tail -1 >> $O/r06c_sweep.jsonl
UAVENV_REPLAN_WGS=512 python bench.py --no-cpu-baseline --no-other-configs --full-line --replan-every 256 --replan-count 32768 --bank-size 131072 --steps 20 --warmup 4 2>/dev/null | tail -1 >> $O/r06c_sweep.jsonl
python - <<'PY' > $O/r06c_sweep.txt
import json
for l in open('gpurun_out/r06c_sweep.jsonl'):
    try: d=json.loads(l)
    except Exception as e: print('bad line', e); continue
    r=d['config']['resets']; f=r.get('refresh') or {}
    print(f.get('every_passes'), f.get('rows_per_slice'), 'ms/pass %.5f'%d['ms_per_pass'], 'value %.4g'%d['value'], 'consumed/s %.0f'%r['consumed_per_s'], 'committed/s %.0f'%f.get('rows_committed_per_s',0), {k:f.get(k) for k in ('refreshes','rows_planned','rows_committed','rows_in_use')}, 'kstep', d['roofline']['kernel_ms_back_to_back'])
PY
: > $O/r06c_envonly.txt
for n in 98304 131072 163840 196608 262144; do
  python bench.py --env-only --envs $n --steps 20 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(d['envs'], 'k_step us %.2f'%(1e3*d['k_step_ms_back_to_back']), 'steps/s %.4g'%d['env_steps_per_s'], 'frac_algorithmic %.3f'%r['frac_algorithmic'], 'frac_stored %.3f'%r['frac_physical_stored'], 'copy GB/s %.0f'%d['measured_copy_GBs'])" >> $O/r06c_envonly.txt
done
cat $O/r06_counter_calibration.txt $O/r06c_sweep.txt $O/r06c_envonly.txt
