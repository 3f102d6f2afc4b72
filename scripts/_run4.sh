python -m pytest tests -m gpu -x -q 2>&1 | tail -8
python bench.py > gpurun_out/r02d_bench.json 2> gpurun_out/r02d_bench.err; python -c "
import json; d=json.load(open('gpurun_out/r02d_bench.json')); print('ms/pass', d['ms_per_pass'], 'value', d['value'], 'host', d['host_enqueue_ms_per_pass'], 'kstep', d['roofline']['kernel_ms_back_to_back'], 'grad', d['roofline_learner']['kernel_ms_back_to_back'], 'cpu', d['cpu_baseline']['value'], d['cpu_baseline']['one_core_value'], d['cpu_baseline']['cgroup_cpu_quota_cores'])"; tail -2 gpurun_out/r02d_bench.err
scripts/profile_round.sh r02d 2>&1 | grep -v "^r02d" | tail -12
