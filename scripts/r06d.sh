#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out; mkdir -p $O
UAVENV_PHASE_PROFILE=1 UAVENV_EXTRA_FLAGS=-DUAVENV_PHASE_POLICY python -c "from dqn_based_uav_3d_path_planer_amd import _build; _build.build(force=True)" 2>&1 | tail -2
POLICY=2 python scripts/phase_profile_coop.py 16384 > $O/r06d_phase_policy_prologue.txt 2>&1
cat $O/r06d_phase_policy_prologue.txt
