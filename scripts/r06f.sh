#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
python -m pytest tests/test_env_parity_gpu.py tests/test_hotloop_gpu.py tests/test_packed_obs_gpu.py tests/test_meta_records_gpu.py -m gpu -q -x --durations=12 2>&1 | tail -30 > $O/r06f_tests.log
python bench.py --no-cpu-baseline --no-other-configs --full-line --steps 20 --warmup 4 2>/dev/null | tail -1 > $O/r06f_bench.json
python -c "
import json; d=json.load(open('gpurun_out/r06f_bench.json')); r=d['roofline']
print('value %.4g ms/pass %.5f kstep_policy b2b %.5f k_step alone %.5f grad %.5f resets' % (d['value'], d['ms_per_pass'], r['kernel_ms_back_to_back'], r['k_step_alone']['kernel_ms_back_to_back'], d['roofline_learner']['kernel_ms_back_to_back']), d['config']['resets'])"
UAVENV_PHASE_PROFILE=1 UAVENV_EXTRA_FLAGS=-DUAVENV_PHASE_POLICY python -c "from dqn_based_uav_3d_path_planer_amd import _build; _build.build(force=True)" 2>&1 | tail -1
POLICY=2 python scripts/phase_profile_coop.py 16384 2>&1 | grep -v amdgpu.ids > $O/r06f_phase_policy_prologue.txt
cat $O/r06f_phase_policy_prologue.txt $O/r06f_tests.log
