#!/bin/bash
# A/B helper on the GPU box: rebuild libuavenv.so with extra hipcc flags, then the default bench line's pass and kernel times.
# usage: scripts/ab_learner.sh "<flags of variant 1>" "<flags of variant 2>" ...   ("" = the tree as it is)
cd $GRAFT_REPO_ROOT
for FLAGS in "$@"; do
  UAVENV_EXTRA_FLAGS="$FLAGS" python -c "from dqn_based_uav_3d_path_planer_amd import _build; _build.build(force=True)" || exit 1
  for rep in 1 2; do
  python bench.py --no-other-configs --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('flags=[$FLAGS] us/pass %.3f  grad %.3f  step_policy %.3f  M steps/s %.1f' % (d['ms_per_pass'] * 1e3, d['roofline_learner']['kernel_ms_back_to_back'] * 1e3, d['roofline']['kernel_ms_back_to_back'] * 1e3, d['value'] / 1e6))"
  done
done
