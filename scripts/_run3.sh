python -m pytest tests/test_packed_obs_gpu.py tests/test_learner_fused_gpu.py tests/test_hotloop_gpu.py tests/test_replay_sample_gpu.py -x -q 2>&1 | tail -15
for d in packed f32; do python bench.py --no-cpu-baseline --obs-dtype $d > gpurun_out/r02c_bench_$d.json 2> gpurun_out/r02c_bench.err; python -c "
import json,sys; d=json.load(open('gpurun_out/r02c_bench_$d.json')); print('$d', 'ms/pass', d['ms_per_pass'], 'value', d['value'], 'host', d['host_enqueue_ms_per_pass'], 'kstep', d['roofline']['kernel_ms_back_to_back'], 'grad', d['roofline_learner']['kernel_ms_back_to_back'])"; done; tail -3 gpurun_out/r02c_bench.err
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --env-only-iters 50 > /dev/null 2>&1; head -7 $(find /tmp/p1 -name "*kernel_stats.csv" | head -1) | cut -c1-150
cd $GRAFT_REPO_ROOT && UAVENV_PHASE_PROFILE=1 python -c "
from dqn_based_uav_3d_path_planer_amd import _build; _build.build(force=True)" && for o in f32 packed; do OBS=$o UAVENV_PHASE_PROFILE=1 python scripts/phase_profile_learner.py 2>&1 | tail -8; done
