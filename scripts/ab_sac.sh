#!/bin/bash
# A/B helper on the GPU box: rebuild libuavenv.so with extra hipcc flags, then BASELINE configs[3]'s pass (bench.py --config 4).
# usage: scripts/ab_sac.sh "<flags of variant 1>" "<flags of variant 2>" ...   ("" = the tree as it is)
cd $GRAFT_REPO_ROOT
for FLAGS in "$@"; do
  UAVENV_EXTRA_FLAGS="$FLAGS" python -c "from dqn_based_uav_3d_path_planer_amd import _build; _build.build(force=True)" || exit 1
  python bench.py --config 4 --no-other-configs --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('flags=[$FLAGS] config 4: ms/pass %.4f  M agent-steps/s %.1f' % (d['ms_per_pass'], d['value'] / 1e6))"
done
