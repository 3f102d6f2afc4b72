#!/bin/bash
# A/B helper on the GPU box: rebuild libuavenv.so with extra flags, then run env-only benches.
# usage: scripts/ab_build.sh "<extra hipcc flags>" <envs...>
FLAGS="$1"; shift
cd $GRAFT_REPO_ROOT
UAVENV_EXTRA_FLAGS="$FLAGS" python -c "from dqn_based_uav_3d_path_planer_amd import _build; _build.build(force=True)" || exit 1
for n in "$@"; do
  python bench.py --env-only --envs $n --replay $((n*4)) --steps 100 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('flags=[$FLAGS] envs', d['envs'], 'k_step_us %.2f' % (d['k_step_ms_back_to_back']*1e3), 'Gsteps/s %.3f' % (d['env_steps_per_s']/1e9), 'frac %.3f' % d['frac_of_8TBs'])"
done
