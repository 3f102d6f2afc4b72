"""MI355X-native PathPlan_City hot path (env-step kernels + device replay + DQN-family learner).

Import name ``dqn_based_uav_3d_path_planer_amd`` (the hyphenated directory name
``dqn-based-uav-3d_path_planer_amd`` is a symlink: Python cannot import a hyphen).
"""
from . import _lib  # noqa: F401
from ._build import build as build_native  # noqa: F401

__all__ = ["build_native"]
