python -m pytest tests/test_hotloop_gpu.py tests/test_packed_obs_gpu.py tests/test_env_parity_gpu.py tests/test_plugins_gpu.py -x -q 2>&1 | tail -8
python bench.py --no-cpu-baseline > gpurun_out/r02h_bench.json 2> gpurun_out/r02h.err; python -c "
import json; d=json.load(open('gpurun_out/r02h_bench.json')); print('fused ms/pass', d['ms_per_pass'], 'value', d['value'], 'host', d['host_enqueue_ms_per_pass'], 'kstep b2b', d['roofline']['kernel_ms_back_to_back'], 'pair', d['roofline']['kernel_ms_event_pair_in_loop'])"; tail -2 gpurun_out/r02h.err
UAVENV_NO_FUSED_ACT=1 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('unfused ms/pass', d['ms_per_pass'], 'value', d['value'])"
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --env-only-iters 50 > /dev/null 2>&1; head -7 $(find /tmp/p1 -name "*kernel_stats.csv" | head -1) | cut -c1-150
