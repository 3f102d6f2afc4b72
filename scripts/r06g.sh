#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
for i in 1 2; do
python bench.py --no-cpu-baseline --no-other-configs --full-line --steps 20 --warmup 4 2>/dev/null | tail -1 > $O/r06g_bench.json
python -c "
import json; d=json.load(open('gpurun_out/r06g_bench.json')); r=d['roofline']
print('value %.4g ms/pass %.5f kstep_policy b2b %.5f k_step alone %.5f grad %.5f resets' % (d['value'], d['ms_per_pass'], r['kernel_ms_back_to_back'], r['k_step_alone']['kernel_ms_back_to_back'], d['roofline_learner']['kernel_ms_back_to_back']), d['config']['resets']['consumed_per_s'])"
done
