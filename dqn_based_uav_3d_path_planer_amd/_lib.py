"""ctypes binding of include/uavenv.h (libuavenv.so).  Fails loudly: no fallback of any kind."""
from __future__ import annotations

import ctypes as C
import os

from . import _build

OBS_DIM = 100
MAX_BUILDINGS = 64
ABI_VERSION = 5

OK, EINVAL, ENOMEM, EHIP, ENODEV, EP2P = 0, -22, -12, -5, -19, -70
P2P_ERR_TIMEOUT, P2P_ERR_DIVERGED = 1, 2
COLL_ID_BYTES = 128
INFO_NORMAL, INFO_SUCCESS, INFO_LOSE, INFO_SKIPPED = 0, 1, 2, 3
INFO_NAMES = ("normal", "success", "lose", "skipped")
ACT_STEER_F32, ACT_STEER_F64, ACT_INDEX_I32 = 0, 1, 2
OBS_F32, OBS_F16, OBS_PACKED = 0, 1, 2
PACKED_DWORDS = 20
META_BYTES = 16             # one transition record of a ring's `meta` plane (include/uavenv.h: UAVENV_META_BYTES)
MFMA_F32, MFMA_F16 = 0, 1
P2P_HANDLE_BYTES = 64
P2P_CHECK_MAX_BLOCKS = 40
STEP_AUTO_RESET, STEP_SKIP_DONE, STEP_NO_OBS, STEP_ONE_WAVE, STEP_APF_LANE = 1, 2, 4, 8, 16

# every symbol include/uavenv.h declares (tests/test_abi.py checks the .so exports all of them)
SYMBOLS = (
    "uavenv_abi_version", "uavenv_last_error", "uavenv_create", "uavenv_destroy", "uavenv_num_agents",
    "uavenv_set_buildings", "uavenv_load_scenarios", "uavenv_plan_scenarios", "uavenv_bank_stats", "uavenv_replan_begin", "uavenv_replan_ready", "uavenv_replan_commit", "uavenv_replan_stats", "uavenv_bank_read", "uavenv_set_moved_word", "uavenv_set_step_meta", "uavenv_tick", "uavenv_dqn_reduce_adam_gated", "uavenv_per_set_f32_gated", "uavenv_rrt_plan", "uavenv_reset_all", "uavenv_set_state", "uavenv_get_state",
    "uavenv_step", "uavenv_step_policy", "uavenv_set_debug_buffer", "uavenv_observe", "uavenv_threaten_rate", "uavenv_threaten_rate_allpairs", "uavenv_geometry",
    "uavenv_replay_sample", "uavenv_obs_unpack", "uavenv_replay_draw", "uavenv_replay_draw_valid", "uavenv_select_actions",
    "uavenv_dqn_num_params", "uavenv_dqn_partial_stride", "uavenv_dqn_partial_rows", "uavenv_dqn_set_debug_buffer", "uavenv_dqn_grad", "uavenv_dqn_grad_w", "uavenv_dqn_reduce", "uavenv_dqn_adam", "uavenv_dqn_reduce_adam", "uavenv_dqn_act",
    "uavenv_p2p_create", "uavenv_p2p_handle", "uavenv_p2p_connect", "uavenv_p2p_destroy", "uavenv_p2p_errors",
    "uavenv_p2p_configure", "uavenv_p2p_status", "uavenv_p2p_error_word", "uavenv_p2p_check_blocks", "uavenv_p2p_inject_fault", "uavenv_p2p_can_reach",
    "uavenv_coll_last_error", "uavenv_coll_unique_id", "uavenv_coll_create", "uavenv_coll_destroy", "uavenv_coll_allreduce_sum",
    "uavenv_dqn_reduce_p2p", "uavenv_dqn_adam_p2p",
    "uavenv_loop_create", "uavenv_loop_destroy", "uavenv_loop_set_eps", "uavenv_loop_run", "uavenv_loop_get", "uavenv_loop_get_per", "uavenv_loop_step_times",
    "uavenv_randn", "uavenv_sac_loop_noise_floats", "uavenv_sac_loop_create", "uavenv_sac_loop_destroy", "uavenv_sac_loop_run",
    "uavenv_sac_loop_get", "uavenv_sac_loop_get_per", "uavenv_per_fill_frame_strided", "uavenv_sac_act_multi", "uavenv_sac_critic_grad_multi", "uavenv_sac_actor_grad_multi",
    "uavenv_sac_critic_adam_multi", "uavenv_sac_actor_adam_multi",
    "uavenv_per_num_chunks", "uavenv_per_rotation", "uavenv_per_rebuild", "uavenv_per_sample", "uavenv_per_set", "uavenv_per_fill", "uavenv_per_set_f32", "uavenv_per_weights", "uavenv_per_fill_frame", "uavenv_per_rebuild_frame", "uavenv_p2p_allreduce", "uavenv_sac_partial_rows_n",
    "uavenv_sac_act", "uavenv_sac_reduce", "uavenv_sac_partial_rows", "uavenv_sac_last_error", "uavenv_sac_set_debug_buffer", "uavenv_sac_critic_grad", "uavenv_sac_critic_adam", "uavenv_sac_actor_grad",
    "uavenv_sac_actor_adam", "uavenv_fed_aggregate",
)
SAC_CRITIC_IN, SAC_ACTOR_PARAMS, SAC_CRITIC_PARAMS, SAC_ACTOR_STRIDE, SAC_CRITIC_STRIDE = 102, 6724, 10882, 6728, 21768


class UavEnvConfig(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("device", C.c_int32), ("n_envs", C.c_int32), ("uav_per_env", C.c_int32),
        ("max_subgoals", C.c_int32), ("max_step", C.c_int32), ("apf_enabled", C.c_int32), ("obs_dtype", C.c_int32),
        ("n_actions", C.c_int32), ("reserved0", C.c_int32),
        ("len", C.c_double), ("width", C.c_double), ("h", C.c_double),
        ("max_v", C.c_double), ("steering_angle", C.c_double), ("power", C.c_double * 8), ("cell_size", C.c_double),
    ]


class UavReplayRing(C.Structure):
    _fields_ = [
        ("obs", C.c_void_p), ("action", C.c_void_p), ("reward", C.c_void_p), ("done", C.c_void_p),
        ("valid", C.c_void_p), ("frames", C.c_int32), ("n_agents", C.c_int32), ("obs_dtype", C.c_int32),
        ("action_is_index", C.c_int32), ("meta", C.c_void_p),
    ]


class UavDqnNet(C.Structure):
    _fields_ = [("local", C.c_void_p), ("target", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p),
                ("w", C.c_int32), ("hid", C.c_int32), ("n_actions", C.c_int32), ("dueling", C.c_int32),
                ("mfma_dtype", C.c_int32), ("reserved0", C.c_int32)]


class UavPer(C.Structure):
    _fields_ = [("prio", C.c_void_p), ("chunk_sum", C.c_void_p), ("chunk_prefix", C.c_void_p),
                ("capacity", C.c_int64), ("rot", C.c_int64), ("group_sum", C.c_void_p)]


class UavLoopConfig(C.Structure):
    _fields_ = [("env", C.c_void_p), ("ring", UavReplayRing), ("net", UavDqnNet),
                ("head", C.c_int32), ("filled", C.c_int32), ("batch", C.c_int32), ("kind", C.c_int32),
                ("huber", C.c_int32), ("update_loop", C.c_int32), ("epoch", C.c_int32), ("learn_start", C.c_int32),
                ("seed", C.c_uint64), ("counter", C.c_uint64),
                ("eps", C.c_float), ("gamma", C.c_float), ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float),
                ("adam_eps", C.c_float), ("step_flags", C.c_uint32), ("partials_dev", C.c_void_p),
                ("loss_dev", C.c_void_p), ("info_dev", C.c_void_p), ("p2p", C.c_void_p), ("time_every", C.c_int32),
                ("sample_lag", C.c_int32), ("coll", C.c_void_p), ("raw_dev", C.c_void_p),
                ("per", UavPer), ("per_alpha", C.c_double), ("per_beta", C.c_double), ("per_beta_inc", C.c_double),
                ("per_eps", C.c_double), ("per_clip", C.c_double),
                ("per_slots_dev", C.c_void_p), ("per_prio_dev", C.c_void_p), ("per_w_dev", C.c_void_p),
                ("per_abs_dev", C.c_void_p), ("per_idx_dev", C.c_void_p),
                ("replan_every", C.c_int32), ("replan_count", C.c_int32), ("replan_max_iter", C.c_int32), ("reserved1", C.c_int32),
                ("moved_dev", C.c_void_p)]


class UavLoopCursor(C.Structure):
    _fields_ = [("head", C.c_int32), ("filled", C.c_int32), ("epoch", C.c_int32), ("reserved0", C.c_int32),
                ("counter", C.c_uint64)]


class UavSacNets(C.Structure):
    _fields_ = [("actor", C.c_void_p), ("critic1", C.c_void_p), ("critic2", C.c_void_p), ("target1", C.c_void_p),
                ("target2", C.c_void_p), ("log_alpha", C.c_void_p)]


class UavSacBatch(C.Structure):
    _fields_ = [("obs_packed", C.c_void_p), ("idx_s", C.c_void_p), ("idx_n", C.c_void_p), ("draws", C.c_void_p),
                ("n_agents", C.c_int32), ("uav_per_env", C.c_int32), ("slot", C.c_int32), ("frames", C.c_int32),
                ("act0", C.c_void_p), ("act1", C.c_void_p), ("reward", C.c_void_p), ("done", C.c_void_p), ("valid", C.c_void_p),
                ("eps", C.c_void_p), ("batch", C.c_int32), ("tiles_per_wg", C.c_int32),
                ("is_weights", C.c_void_p), ("abs_td_out", C.c_void_p), ("meta", C.c_void_p), ("td_scratch", C.c_void_p)]


class UavSacAdam(C.Structure):
    _fields_ = [("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("bias_correction1", C.c_float), ("bias_correction2_sqrt", C.c_float), ("tau", C.c_float), ("grad_scale", C.c_float),
                ("skip_word", C.c_void_p), ("go_word", C.c_void_p), ("go_value", C.c_uint32), ("reserved0", C.c_uint32)]


SAC_LOOP_MAX_SLOTS = 8
FED_MAX_BLOCKS = 8
DRAW_MAX_TRIES = 8
DQN_IMAGE_FLOATS = 6912      # csrc/dqn_internal.hpp


class UavSacLoopSlot(C.Structure):
    _fields_ = [("nets", UavSacNets), ("m_actor", C.c_void_p), ("v_actor", C.c_void_p), ("alpha_mv", C.c_void_p),
                ("m1", C.c_void_p), ("v1", C.c_void_p), ("m2", C.c_void_p), ("v2", C.c_void_p), ("scalars", C.c_void_p),
                ("partials_critic", C.c_void_p), ("partials_actor", C.c_void_p),
                ("epoch", C.c_int32), ("adam_steps", C.c_int32),
                ("per", UavPer), ("per_slots_dev", C.c_void_p), ("per_prio_dev", C.c_void_p), ("per_w_dev", C.c_void_p),
                ("per_abs_dev", C.c_void_p), ("per_beta", C.c_double), ("td_dev", C.c_void_p)]


class UavSacLoopConfig(C.Structure):
    _fields_ = [("env", C.c_void_p), ("ring", UavReplayRing), ("act1_plane", C.c_void_p), ("info_dev", C.c_void_p),
                ("n_slots", C.c_int32), ("batch", C.c_int32), ("head", C.c_int32), ("filled", C.c_int32),
                ("is_train", C.c_int32), ("valid_draws", C.c_int32), ("seed", C.c_uint64), ("counter", C.c_uint64),
                ("beta1", C.c_double), ("beta2", C.c_double), ("adam_eps", C.c_double),
                ("gamma", C.c_float), ("tau", C.c_float), ("action_bound", C.c_float), ("actor_lr", C.c_float),
                ("critic_lr", C.c_float), ("alpha_lr", C.c_float), ("target_entropy", C.c_float), ("reserved1", C.c_float),
                ("step_flags", C.c_uint32), ("reserved2", C.c_uint32),
                ("draws_dev", C.c_void_p), ("noise_dev", C.c_void_p), ("slot", UavSacLoopSlot * SAC_LOOP_MAX_SLOTS),
                ("p2p", C.c_void_p), ("coll", C.c_void_p), ("xbuf_dev", C.c_void_p),
                ("per_alpha", C.c_double), ("per_beta_inc", C.c_double), ("per_eps", C.c_double), ("per_clip", C.c_double),
                ("moved_dev", C.c_void_p), ("check_every", C.c_int32), ("reserved3", C.c_int32)]


class UavSacLoopCursor(C.Structure):
    _fields_ = [("head", C.c_int32), ("filled", C.c_int32), ("counter", C.c_uint64),
                ("epoch", C.c_int32 * SAC_LOOP_MAX_SLOTS), ("adam_steps", C.c_int32 * SAC_LOOP_MAX_SLOTS)]


class UavEnvError(RuntimeError):
    pass


_LIB = None


def load() -> C.CDLL:
    """Load (building first if the sources are newer) libuavenv.so.  Raises if that is impossible."""
    global _LIB
    if _LIB is not None:
        return _LIB
    # torch first: it bundles its own libamdhip64/libhsa-runtime64; if libuavenv.so pulled /opt/rocm's copies in
    # before torch loads, the process ends up with two HSA runtimes and the second sees no device.
    import torch  # noqa: F401
    path = _build.LIB_PATH
    if _build.needs_build():
        _build.build()
    if not os.path.exists(path):
        raise UavEnvError(f"{path} is missing and could not be built; the env hot path has no CPU fallback")
    lib = C.CDLL(path)
    vp, i32, u32, u64, i64, f32 = C.c_void_p, C.c_int32, C.c_uint32, C.c_uint64, C.c_int64, C.c_float
    lib.uavenv_abi_version.restype = C.c_int
    lib.uavenv_last_error.restype = C.c_char_p
    lib.uavenv_create.restype = C.c_int
    lib.uavenv_create.argtypes = [C.POINTER(UavEnvConfig), C.POINTER(vp)]
    lib.uavenv_destroy.restype = C.c_int
    lib.uavenv_destroy.argtypes = [vp]
    lib.uavenv_num_agents.restype = C.c_int
    lib.uavenv_num_agents.argtypes = [vp]
    lib.uavenv_set_buildings.restype = C.c_int
    lib.uavenv_set_buildings.argtypes = [vp, vp, vp, i32]
    lib.uavenv_load_scenarios.restype = C.c_int
    lib.uavenv_load_scenarios.argtypes = [vp, vp, vp, vp, i32]
    lib.uavenv_plan_scenarios.restype = C.c_int
    lib.uavenv_plan_scenarios.argtypes = [vp, i32, u64, i32, vp]
    lib.uavenv_bank_stats.restype = C.c_int
    lib.uavenv_bank_stats.argtypes = [vp, vp, vp]
    lib.uavenv_replan_begin.restype = C.c_int
    lib.uavenv_replan_begin.argtypes = [vp, i32, i32, u64, i32, vp]
    lib.uavenv_replan_ready.restype = C.c_int
    lib.uavenv_replan_ready.argtypes = [vp]
    lib.uavenv_replan_commit.restype = C.c_int
    lib.uavenv_replan_commit.argtypes = [vp, i32, vp]
    lib.uavenv_replan_stats.restype = C.c_int
    lib.uavenv_replan_stats.argtypes = [vp, vp]
    lib.uavenv_set_moved_word.restype = C.c_int
    lib.uavenv_set_moved_word.argtypes = [vp, vp]
    lib.uavenv_set_step_meta.restype = C.c_int
    lib.uavenv_set_step_meta.argtypes = [vp, vp, vp]
    lib.uavenv_tick.restype = C.c_uint64
    lib.uavenv_tick.argtypes = [vp]
    lib.uavenv_bank_read.restype = C.c_int
    lib.uavenv_bank_read.argtypes = [vp, i32, i32, vp, vp, vp]
    lib.uavenv_rrt_plan.restype = C.c_int
    lib.uavenv_rrt_plan.argtypes = [vp, i32, vp, vp, i32, u64, i32, C.c_double, C.c_double, vp, vp, vp, vp, vp]
    lib.uavenv_reset_all.restype = C.c_int
    lib.uavenv_reset_all.argtypes = [vp, u64, vp]
    lib.uavenv_set_state.restype = C.c_int
    lib.uavenv_set_state.argtypes = [vp, i32, i32, vp, vp, vp, vp, vp]
    lib.uavenv_get_state.restype = C.c_int
    lib.uavenv_get_state.argtypes = [vp, i32, i32, vp, vp, vp]
    lib.uavenv_step.restype = C.c_int
    lib.uavenv_step.argtypes = [vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, u32, vp]
    lib.uavenv_step_policy.restype = C.c_int
    lib.uavenv_step_policy.argtypes = [vp, C.POINTER(UavDqnNet), vp, f32, u64, u64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, u32, vp]
    lib.uavenv_set_debug_buffer.restype = C.c_int
    lib.uavenv_set_debug_buffer.argtypes = [vp, vp]
    lib.uavenv_observe.restype = C.c_int
    lib.uavenv_observe.argtypes = [vp, vp, vp]
    lib.uavenv_threaten_rate.restype = C.c_int
    lib.uavenv_threaten_rate.argtypes = [vp, vp, vp, i64, vp]
    lib.uavenv_geometry.restype = C.c_int
    lib.uavenv_geometry.argtypes = [vp, vp, vp, i64, vp]
    lib.uavenv_threaten_rate_allpairs.restype = C.c_int
    lib.uavenv_threaten_rate_allpairs.argtypes = [vp, vp, vp, i64, vp]
    lib.uavenv_replay_sample.restype = C.c_int
    lib.uavenv_replay_sample.argtypes = [C.POINTER(UavReplayRing), i32, i32, i32, u64, u64, vp, vp, vp, vp, vp, vp, vp]
    lib.uavenv_obs_unpack.restype = C.c_int
    lib.uavenv_obs_unpack.argtypes = [vp, i64, vp, i32, vp]
    lib.uavenv_replay_draw.restype = C.c_int
    lib.uavenv_replay_draw.argtypes = [i32, i32, i32, i32, i32, u64, u64, vp, vp]
    lib.uavenv_replay_draw_valid.restype = C.c_int
    lib.uavenv_replay_draw_valid.argtypes = [i32, i32, i32, i32, i32, i32, i32, i32, vp, i32, u64, u64, vp, vp]
    lib.uavenv_p2p_create.restype = C.c_int
    lib.uavenv_p2p_create.argtypes = [i32, i32, i32, C.POINTER(vp)]
    lib.uavenv_p2p_handle.restype = C.c_int
    lib.uavenv_p2p_handle.argtypes = [vp, vp]
    lib.uavenv_p2p_connect.restype = C.c_int
    lib.uavenv_p2p_connect.argtypes = [vp, vp]
    lib.uavenv_p2p_destroy.restype = C.c_int
    lib.uavenv_p2p_destroy.argtypes = [vp]
    lib.uavenv_p2p_errors.restype = C.c_int
    lib.uavenv_p2p_errors.argtypes = [vp, C.POINTER(i32)]
    lib.uavenv_p2p_configure.restype = C.c_int
    lib.uavenv_p2p_configure.argtypes = [vp, i32, i32]
    lib.uavenv_p2p_status.restype = C.c_int
    lib.uavenv_p2p_status.argtypes = [vp, i32, C.POINTER(i32)]
    lib.uavenv_p2p_can_reach.restype = C.c_int
    lib.uavenv_p2p_can_reach.argtypes = [i32, i32]
    lib.uavenv_p2p_inject_fault.restype = C.c_int
    lib.uavenv_p2p_inject_fault.argtypes = [vp, i32]
    lib.uavenv_coll_last_error.restype = C.c_char_p
    lib.uavenv_coll_unique_id.restype = C.c_int
    lib.uavenv_coll_unique_id.argtypes = [C.c_char_p, vp]
    lib.uavenv_coll_create.restype = C.c_int
    lib.uavenv_coll_create.argtypes = [C.c_char_p, i32, i32, vp, C.POINTER(vp)]
    lib.uavenv_coll_destroy.restype = C.c_int
    lib.uavenv_coll_destroy.argtypes = [vp]
    lib.uavenv_coll_allreduce_sum.restype = C.c_int
    lib.uavenv_coll_allreduce_sum.argtypes = [vp, vp, i64, vp]
    lib.uavenv_dqn_reduce_p2p.restype = C.c_int
    lib.uavenv_dqn_reduce_p2p.argtypes = [C.POINTER(UavDqnNet), vp, i32, vp, vp]
    lib.uavenv_dqn_adam_p2p.restype = C.c_int
    lib.uavenv_dqn_adam_p2p.argtypes = [C.POINTER(UavDqnNet), vp, f32, f32, f32, f32, i32, i32, vp, vp, vp]
    lib.uavenv_loop_create.restype = C.c_int
    lib.uavenv_loop_create.argtypes = [C.POINTER(UavLoopConfig), C.POINTER(vp)]
    lib.uavenv_loop_destroy.restype = C.c_int
    lib.uavenv_loop_destroy.argtypes = [vp]
    lib.uavenv_loop_set_eps.restype = C.c_int
    lib.uavenv_loop_set_eps.argtypes = [vp, f32]
    lib.uavenv_loop_run.restype = C.c_int
    lib.uavenv_loop_run.argtypes = [vp, i32, vp]
    lib.uavenv_loop_get.restype = C.c_int
    lib.uavenv_loop_get.argtypes = [vp, C.POINTER(UavLoopCursor)]
    lib.uavenv_loop_step_times.restype = C.c_int
    lib.uavenv_loop_step_times.argtypes = [vp, vp, i32, C.POINTER(i32)]
    for _n in ("uavenv_sac_act_multi", "uavenv_sac_critic_grad_multi", "uavenv_sac_actor_grad_multi",
               "uavenv_sac_critic_adam_multi", "uavenv_sac_actor_adam_multi"):       # (called from csrc/loop.hip; no Python caller)
        getattr(lib, _n).restype = C.c_int
    lib.uavenv_randn.restype = C.c_int
    lib.uavenv_randn.argtypes = [u64, u64, i64, vp, vp]
    lib.uavenv_sac_loop_noise_floats.restype = C.c_int64
    lib.uavenv_sac_loop_noise_floats.argtypes = [i32, i32, i32]
    lib.uavenv_sac_loop_create.restype = C.c_int
    lib.uavenv_sac_loop_create.argtypes = [C.POINTER(UavSacLoopConfig), C.POINTER(vp)]
    lib.uavenv_sac_loop_destroy.restype = C.c_int
    lib.uavenv_sac_loop_destroy.argtypes = [vp]
    lib.uavenv_sac_loop_run.restype = C.c_int
    lib.uavenv_sac_loop_run.argtypes = [vp, i32, vp]
    lib.uavenv_sac_loop_get.restype = C.c_int
    lib.uavenv_sac_loop_get.argtypes = [vp, C.POINTER(UavSacLoopCursor)]
    lib.uavenv_sac_act.restype = C.c_int
    lib.uavenv_sac_act.argtypes = [vp, vp, i32, i32, i32, vp, f32, vp, vp, vp]
    lib.uavenv_sac_reduce.restype = C.c_int
    lib.uavenv_sac_reduce.argtypes = [vp, i32, i32, vp, vp]
    lib.uavenv_sac_partial_rows.restype = C.c_int
    lib.uavenv_sac_partial_rows.argtypes = [i32]
    lib.uavenv_sac_partial_rows_n.restype = C.c_int
    lib.uavenv_sac_partial_rows_n.argtypes = [i32, i32, i32]
    lib.uavenv_sac_last_error.restype = C.c_char_p
    lib.uavenv_sac_set_debug_buffer.restype = C.c_int
    lib.uavenv_sac_set_debug_buffer.argtypes = [vp]
    lib.uavenv_sac_critic_grad.restype = C.c_int
    lib.uavenv_sac_critic_grad.argtypes = [C.POINTER(UavSacNets), C.POINTER(UavSacBatch), f32, f32, vp, vp]
    lib.uavenv_sac_critic_adam.restype = C.c_int
    lib.uavenv_sac_critic_adam.argtypes = [C.POINTER(UavSacNets), vp, i32, vp, vp, vp, vp, C.POINTER(UavSacAdam), vp, vp]
    lib.uavenv_sac_actor_grad.restype = C.c_int
    lib.uavenv_sac_actor_grad.argtypes = [C.POINTER(UavSacNets), C.POINTER(UavSacBatch), f32, vp, vp]
    lib.uavenv_sac_actor_adam.restype = C.c_int
    lib.uavenv_sac_actor_adam.argtypes = [C.POINTER(UavSacNets), vp, i32, i32, vp, vp, vp, C.POINTER(UavSacAdam), f32, f32, vp, vp]
    lib.uavenv_select_actions.restype = C.c_int
    lib.uavenv_select_actions.argtypes = [vp, i32, i32, f32, u64, u64, vp, vp, vp]
    net = C.POINTER(UavDqnNet)
    lib.uavenv_dqn_num_params.restype = C.c_int
    lib.uavenv_dqn_num_params.argtypes = [net]
    lib.uavenv_dqn_partial_stride.restype = C.c_int
    lib.uavenv_dqn_partial_stride.argtypes = [net]
    lib.uavenv_dqn_partial_rows.restype = C.c_int
    lib.uavenv_dqn_partial_rows.argtypes = [i32]
    lib.uavenv_dqn_set_debug_buffer.restype = C.c_int
    lib.uavenv_dqn_set_debug_buffer.argtypes = [vp]
    lib.uavenv_dqn_grad.restype = C.c_int
    lib.uavenv_dqn_grad.argtypes = [C.POINTER(UavReplayRing), i32, i32, i32, u64, u64, vp, net, i32, f32, i32, vp, vp]
    lib.uavenv_dqn_grad_w.restype = C.c_int
    lib.uavenv_dqn_grad_w.argtypes = [C.POINTER(UavReplayRing), i32, i32, i32, u64, u64, vp, net, i32, f32, i32, vp, vp, vp, vp]
    # not part of the C ABI (csrc/dqn_internal.hpp): the layer-1 image the C loop keeps for its DQN launches, for measurements
    # of exactly those launches (bench.py's back-to-back legs)
    lib.uavenv_dqn_split_image.restype = C.c_int
    lib.uavenv_dqn_split_image.argtypes = [net, vp, vp]
    lib.uavenv_dqn_grad_img.restype = C.c_int
    lib.uavenv_dqn_grad_img.argtypes = [C.POINTER(UavReplayRing), i32, i32, i32, u64, u64, vp, net, i32, f32, i32, vp, vp, vp, vp, vp]
    lib.uavenv_step_policy_img.restype = C.c_int
    lib.uavenv_step_policy_img.argtypes = [vp, C.POINTER(UavDqnNet), vp, f32, u64, u64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, u32, vp, vp]
    lib.uavenv_dqn_reduce.restype = C.c_int
    lib.uavenv_dqn_reduce.argtypes = [net, vp, i32, vp, vp]
    lib.uavenv_dqn_adam.restype = C.c_int
    lib.uavenv_dqn_adam.argtypes = [net, vp, f32, f32, f32, f32, i32, i32, vp, vp]
    lib.uavenv_dqn_reduce_adam.restype = C.c_int
    lib.uavenv_dqn_reduce_adam.argtypes = [net, vp, i32, f32, f32, f32, f32, i32, i32, vp, vp, vp]
    lib.uavenv_dqn_reduce_adam_gated.restype = C.c_int
    lib.uavenv_dqn_reduce_adam_gated.argtypes = [net, vp, i32, f32, f32, f32, f32, i32, i32, vp, vp, vp, C.c_uint32, vp]
    lib.uavenv_dqn_act.restype = C.c_int
    lib.uavenv_dqn_act.argtypes = [net, vp, i32, i32, f32, u64, u64, vp, vp, vp, vp]
    per, f64 = C.POINTER(UavPer), C.c_double
    lib.uavenv_per_num_chunks.restype = C.c_int
    lib.uavenv_per_num_chunks.argtypes = [i64]
    lib.uavenv_per_rotation.restype = C.c_int
    lib.uavenv_per_rotation.argtypes = [i64]
    lib.uavenv_per_rebuild.restype = C.c_int
    lib.uavenv_per_rebuild.argtypes = [per, vp]
    lib.uavenv_per_sample.restype = C.c_int
    lib.uavenv_per_sample.argtypes = [per, i32, vp, u64, u64, vp, vp, vp]
    lib.uavenv_per_set.restype = C.c_int
    lib.uavenv_per_set.argtypes = [per, vp, vp, i32, f64, f64, f64, vp]
    lib.uavenv_per_set_f32.restype = C.c_int
    lib.uavenv_per_set_f32.argtypes = [per, vp, vp, i32, f64, f64, f64, vp]
    lib.uavenv_per_set_f32_gated.restype = C.c_int
    lib.uavenv_per_set_f32_gated.argtypes = [per, vp, vp, i32, f64, f64, f64, vp, C.c_uint32, vp]
    lib.uavenv_p2p_allreduce.restype = C.c_int
    lib.uavenv_p2p_allreduce.argtypes = [vp, vp, i64, vp]
    lib.uavenv_per_rebuild_frame.restype = C.c_int
    lib.uavenv_per_rebuild_frame.argtypes = [per, i64, i64, f64, vp, i64, vp]
    lib.uavenv_per_fill_frame.restype = C.c_int
    lib.uavenv_per_fill_frame.argtypes = [per, i64, i64, f64, vp, i64, vp]
    lib.uavenv_per_weights.restype = C.c_int
    lib.uavenv_per_weights.argtypes = [per, vp, vp, i32, i64, f64, i32, vp, vp, vp]
    lib.uavenv_loop_get_per.restype = C.c_int
    lib.uavenv_loop_get_per.argtypes = [vp, C.POINTER(C.c_double)]
    lib.uavenv_per_fill.restype = C.c_int
    lib.uavenv_per_fill.argtypes = [per, i64, i64, f64, vp, vp]
    lib.uavenv_p2p_check_blocks.restype = C.c_int
    lib.uavenv_p2p_check_blocks.argtypes = [vp, vp, vp, i32, vp]
    lib.uavenv_p2p_error_word.restype = C.c_void_p
    lib.uavenv_p2p_error_word.argtypes = [vp]
    lib.uavenv_sac_loop_get_per.restype = C.c_int
    lib.uavenv_sac_loop_get_per.argtypes = [vp, C.POINTER(C.c_double)]
    lib.uavenv_per_fill_frame_strided.restype = C.c_int
    lib.uavenv_per_fill_frame_strided.argtypes = [per, i64, i64, f64, vp, i64, i64, vp]
    lib.uavenv_fed_aggregate.restype = C.c_int
    lib.uavenv_fed_aggregate.argtypes = [vp, i32, i32, f32, vp]
    if lib.uavenv_abi_version() != ABI_VERSION:
        raise UavEnvError(f"libuavenv ABI {lib.uavenv_abi_version()} != binding {ABI_VERSION}")
    _LIB = lib
    return lib


def rccl_path() -> bytes:
    """PyTorch's own librccl.so (so that csrc/coll.hip shares the instance torch.distributed already mapped)."""
    import torch
    p = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    return p.encode() if os.path.exists(p) else b""


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().uavenv_last_error().decode("utf-8", "replace")
        raise UavEnvError(f"{what or 'uavenv call'} failed with code {rc}: {msg}")
