"""Where the plugins get their vectorised env from.  Production: the HIP library (no fallback).  The CPU test
suite swaps `make_backend` for an oracle-backed fake to exercise the host logic without a GPU."""


def make_backend(n_envs, buildings, **kw):
    from dqn_based_uav_3d_path_planer_amd.env import VecPathPlanEnv
    return VecPathPlanEnv(n_envs, buildings, **kw)
