"""VecPathPlanEnv -- torch-tensor front end of the C ABI (include/uavenv.h).

N = n_envs * uav_per_env agents live in HBM inside libuavenv; this class only owns the
I/O tensors and forwards device pointers + the current torch stream.  It mirrors the
reference's per-agent calls (Agents/UAV.py reset/update/state, BaseEnv.Move_Agent) in
batched form; the reflection-compatible plugin classes are in plugins/.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

from . import _lib

# config/UAV.xml Power_param defaults (P_i v_0 d_0 rho s A P_b F_b), Agents/UAV.py:50-57
DEFAULT_POWER = (89.0, 4.05, 0.6, 1.225, 0.05, 0.5, 79.0, 120.0)


@dataclass
class StepOut:
    obs: Optional[torch.Tensor]          # [N,100] state after the step (after the reset when auto-reset fired)
    reward: torch.Tensor                 # [N] f64  global_r
    reward32: torch.Tensor               # [N] f32
    ret_done: torch.Tensor               # [N] u8   the `done` update_PathPlan RETURNS (stored in replay)
    agent_done: torch.Tensor             # [N] u8   self.done (ends the episode)
    info: torch.Tensor                   # [N] u8   0 normal / 1 success / 2 lose / 3 skipped
    valid: torch.Tensor                  # [N] u8
    energy: Optional[torch.Tensor]       # [N] f64  Calc_Fly_Power at the new speed


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _host(a, dtype) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a), dtype=dtype)


class VecPathPlanEnv:
    def __init__(self, n_envs: int, buildings, *, uav_per_env: int = 1, device="cuda:0", max_subgoals: int = 48,
                 max_step: int = 150, apf_enabled: int = 0, obs_dtype=torch.float32, n_actions: int = 3,
                 length: float = 500.0, width: float = 500.0, h: float = 100.0, max_v: float = 1.0,
                 steering_angle: float = 30.0 / 180.0 * math.pi, power=DEFAULT_POWER, cell_size: float = 0.0,
                 velocities=None):
        if not torch.cuda.is_available():
            raise _lib.UavEnvError("VecPathPlanEnv needs an MI355X (torch.cuda unavailable); there is no CPU path")
        self.lib = _lib.load()
        self.device = torch.device(device)
        self.n_envs, self.uav_per_env = int(n_envs), int(uav_per_env)
        self.K = int(max_subgoals)
        # observation storage: torch.float32 / torch.float16 rows of 100, or "packed" = 20 int32 per row (15 scalars +
        # 80 flag bits, lossless; include/uavenv.h UAVENV_OBS_PACKED) -- what the fused act / learner kernels read
        self.packed = isinstance(obs_dtype, str) and obs_dtype == "packed"
        self.obs_dtype = torch.int32 if self.packed else obs_dtype
        self.obs_width = _lib.PACKED_DWORDS if self.packed else _lib.OBS_DIM
        self.obs_code = _lib.OBS_PACKED if self.packed else (_lib.OBS_F16 if obs_dtype == torch.float16 else _lib.OBS_F32)
        cfg = _lib.UavEnvConfig()
        cfg.abi_version = _lib.ABI_VERSION
        cfg.device = self.device.index or 0
        cfg.n_envs, cfg.uav_per_env = self.n_envs, self.uav_per_env
        cfg.max_subgoals, cfg.max_step, cfg.apf_enabled = self.K, int(max_step), int(apf_enabled)
        cfg.obs_dtype = self.obs_code
        cfg.n_actions = int(n_actions)
        cfg.len, cfg.width, cfg.h = float(length), float(width), float(h)
        cfg.max_v, cfg.steering_angle = float(max_v), float(steering_angle)
        for k in range(8):
            cfg.power[k] = float(power[k])
        cfg.cell_size = float(cell_size)
        self.cfg = cfg
        self._h = C.c_void_p()
        torch.cuda.set_device(self.device)
        _lib.check(self.lib.uavenv_create(C.byref(cfg), C.byref(self._h)), "uavenv_create")
        self.N = self.lib.uavenv_num_agents(self._h)
        self.set_buildings(buildings, velocities)
        self._out = None

    # ------------------------------------------------------------------ lifetime
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            torch.cuda.synchronize(self.device)
            self.lib.uavenv_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    # ------------------------------------------------------------------ world / bank
    def set_buildings(self, buildings, velocities=None):
        b = _host(buildings, np.float64).reshape(-1, 5)
        v = None if velocities is None else _host(velocities, np.float64).reshape(-1, 3)
        self.buildings = b
        _lib.check(self.lib.uavenv_set_buildings(self._h, b.ctypes.data, None if v is None else v.ctypes.data,
                                                 len(b)), "uavenv_set_buildings")

    def load_scenarios(self, start_goal, sub_goals, n_sub):
        """start_goal [M,6]; sub_goals [M,k,3] (k <= K, padded here); n_sub [M]."""
        sg = _host(start_goal, np.float64).reshape(-1, 6)
        ns = _host(n_sub, np.int32).reshape(-1)
        sub = np.asarray(sub_goals, dtype=np.float64)
        m = len(sg)
        if int(ns.max()) > self.K:
            raise ValueError(f"scenario with {int(ns.max())} sub-goals exceeds max_subgoals={self.K}")
        padded = np.zeros((m, self.K, 3), dtype=np.float64)
        k = min(sub.shape[1], self.K)
        padded[:, :k] = sub[:, :k]
        _lib.check(self.lib.uavenv_load_scenarios(self._h, sg.ctypes.data, padded.ctypes.data, ns.ctypes.data, m),
                   "uavenv_load_scenarios")
        self.n_scenarios = m

    def plan_scenarios(self, m: int, seed: int = 0, max_iter: int = 10000):
        """Plan a fresh bank of m reset scenarios on the GPU (start/goal draws + RRT, one wavefront each)."""
        _lib.check(self.lib.uavenv_plan_scenarios(self._h, int(m), int(seed), int(max_iter), self._stream()),
                   "uavenv_plan_scenarios")
        self.n_scenarios = int(m)

    def bank_stats(self):
        """-> (scenarios in the bank, how many of them are copies of a neighbour because their own plan failed)."""
        m, r = C.c_int32(0), C.c_int32(0)
        _lib.check(self.lib.uavenv_bank_stats(self._h, C.byref(m), C.byref(r)), "uavenv_bank_stats")
        return m.value, r.value

    # -- rolling refresh of the bank (the reference plans at every reset; include/uavenv.h: uavenv_replan_*) --------------
    def replan_begin(self, first: int, count: int, seed: int, max_iter: int = 10000, stream: Optional[torch.cuda.Stream] = None):
        """Plan `count` fresh scenarios for bank rows [first, first + count) in the background (no LDS: fits beside any
        kernel); stream: where the planner runs (default: the current stream)."""
        s = self._stream() if stream is None else stream.cuda_stream
        _lib.check(self.lib.uavenv_replan_begin(self._h, int(first), int(count), int(seed), int(max_iter), s), "uavenv_replan_begin")

    def replan_ready(self) -> int:
        """1 = planned, 0 = still planning, -1 = nothing pending."""
        return int(self.lib.uavenv_replan_ready(self._h))

    def replan_commit(self, force: bool = False):
        """Hand the planned slice over on the current stream: rows no agent is flying take their new plan (force: every row
        -- the caller resets all agents next)."""
        _lib.check(self.lib.uavenv_replan_commit(self._h, 1 if force else 0, self._stream()), "uavenv_replan_commit")

    def replan_stats(self) -> dict:
        out = (C.c_int64 * 5)()
        _lib.check(self.lib.uavenv_replan_stats(self._h, out), "uavenv_replan_stats")
        return dict(zip(("refreshes", "rows_planned", "rows_committed", "rows_in_use", "rows_without_plan"), (int(x) for x in out)))

    def bank_read(self, first: int = 0, count: Optional[int] = None):
        """-> (start_goal [count, 6], sub_goals [count, K, 3], n_sub [count]) of the bank rows, as numpy arrays."""
        count = self.n_scenarios - first if count is None else int(count)
        sg = np.zeros((count, 6))
        sub = np.zeros((count, self.K, 3))
        ns = np.zeros(count, dtype=np.int32)
        _lib.check(self.lib.uavenv_bank_read(self._h, int(first), count, sg.ctypes.data, sub.ctypes.data, ns.ctypes.data), "uavenv_bank_read")
        return sg, sub, ns

    def rrt_plan(self, m: int, *, start_goal=None, uniforms=None, seed: int = 0, max_iter: int = 10000,
                 step_size: float = 30.0, obstacle_step: float = 5.0):
        """The GPU planner alone -> (start_goal [m,6], sub_goals [m,K,3], n_sub [m], iters [m]) device tensors."""
        d = self.device
        sg_in = None if start_goal is None else torch.as_tensor(start_goal, dtype=torch.float64, device=d).contiguous()
        u = None if uniforms is None else torch.as_tensor(uniforms, dtype=torch.float64, device=d).contiguous()
        sg = torch.zeros((m, 6), dtype=torch.float64, device=d)
        sub = torch.zeros((m, self.K, 3), dtype=torch.float64, device=d)
        ns = torch.zeros(m, dtype=torch.int32, device=d)
        it = torch.zeros(m, dtype=torch.int32, device=d)
        _lib.check(self.lib.uavenv_rrt_plan(self._h, int(m), _ptr(sg_in), _ptr(u), 0 if u is None else u.shape[1], int(seed),
                                            int(max_iter), float(step_size), float(obstacle_step), sg.data_ptr(),
                                            sub.data_ptr(), ns.data_ptr(), it.data_ptr(), self._stream()), "uavenv_rrt_plan")
        return sg, sub, ns, it

    def reset(self, seed: int = 0, obs: Optional[torch.Tensor] = None) -> torch.Tensor:
        _lib.check(self.lib.uavenv_reset_all(self._h, int(seed), self._stream()), "uavenv_reset_all")
        return self.observe(obs)

    # ------------------------------------------------------------------ parity injection
    def set_state(self, first: int, kin, step, n_sub, sub, alias=None):
        kin = _host(kin, np.float64).reshape(-1, 8)
        count = len(kin)
        step = _host(step, np.int32).reshape(-1)
        n_sub = _host(n_sub, np.int32).reshape(-1)
        sub = np.asarray(sub, dtype=np.float64).reshape(count, -1, 3)
        padded = np.zeros((count, self.K, 3), dtype=np.float64)
        k = min(sub.shape[1], self.K)
        padded[:, :k] = sub[:, :k]
        al = None if alias is None else _host(alias, np.int32).reshape(-1)
        torch.cuda.synchronize(self.device)
        _lib.check(self.lib.uavenv_set_state(self._h, int(first), count, kin.ctypes.data, step.ctypes.data,
                                             n_sub.ctypes.data, None if al is None else al.ctypes.data,
                                             padded.ctypes.data), "uavenv_set_state")

    def get_state(self, first: int = 0, count: Optional[int] = None, want_sub: bool = False):
        """-> state16 [count,16] (see include/uavenv.h), or (state16, sub [count,K,3], alias [count]) if want_sub."""
        count = self.N - first if count is None else count
        out = np.zeros((count, 16), dtype=np.float64)
        sub = np.zeros((count, self.K, 3), dtype=np.float64) if want_sub else None
        alias = np.zeros(count, dtype=np.int32) if want_sub else None
        torch.cuda.synchronize(self.device)
        _lib.check(self.lib.uavenv_get_state(self._h, int(first), int(count), out.ctypes.data,
                                             None if sub is None else sub.ctypes.data,
                                             None if alias is None else alias.ctypes.data), "uavenv_get_state")
        return (out, sub, alias) if want_sub else out

    # ------------------------------------------------------------------ hot path
    def new_obs(self) -> torch.Tensor:
        return torch.empty((self.N, self.obs_width), dtype=self.obs_dtype, device=self.device)

    def unpack(self, obs: torch.Tensor, dtype=torch.float32) -> torch.Tensor:
        """[..., 100] rows of `dtype` from observations as this env stores them (a no-op cast unless packed)."""
        if not self.packed:
            return obs.to(dtype)
        flat = obs.contiguous().view(-1, _lib.PACKED_DWORDS)
        out = torch.empty((flat.shape[0], _lib.OBS_DIM), dtype=dtype, device=obs.device)
        code = _lib.OBS_F16 if dtype == torch.float16 else _lib.OBS_F32
        if dtype not in (torch.float16, torch.float32):
            raise TypeError("unpack to float32 or float16")
        _lib.check(self.lib.uavenv_obs_unpack(flat.data_ptr(), flat.shape[0], out.data_ptr(), code, self._stream()),
                   "uavenv_obs_unpack")
        return out.view(*obs.shape[:-1], _lib.OBS_DIM)

    def observe(self, obs: Optional[torch.Tensor] = None) -> torch.Tensor:
        obs = self.new_obs() if obs is None else obs
        _lib.check(self.lib.uavenv_observe(self._h, obs.data_ptr(), self._stream()), "uavenv_observe")
        return obs

    def alloc_out(self, want_energy: bool = False) -> StepOut:
        d, n = self.device, self.N
        u8 = lambda: torch.empty(n, dtype=torch.uint8, device=d)   # noqa: E731
        return StepOut(obs=self.new_obs(), reward=torch.empty(n, dtype=torch.float64, device=d),
                       reward32=torch.empty(n, dtype=torch.float32, device=d), ret_done=u8(), agent_done=u8(),
                       info=u8(), valid=u8(),
                       energy=torch.empty(n, dtype=torch.float64, device=d) if want_energy else None)

    def step(self, actions: torch.Tensor, out: Optional[StepOut] = None, *, auto_reset: bool = False,
             skip_done: bool = False, want_energy: bool = False, active: Optional[torch.Tensor] = None,
             one_wave: bool = False, apf_lane: bool = False) -> StepOut:
        """One update_PathPlan + state_PathPlan for every agent.  actions: [N] float32/float64 steer or int32 index.
        one_wave (diagnostics) forces the one-wavefront-per-64-agents kernel on small launches, apf_lane (diagnostics) the
        in-kernel per-lane Adjust_subgoal instead of the k_apf_adjust launch; results are identical either way."""
        if out is None:
            out = self.alloc_out(want_energy)
        if actions.dtype == torch.float32:
            kind = _lib.ACT_STEER_F32
        elif actions.dtype == torch.float64:
            kind = _lib.ACT_STEER_F64
        elif actions.dtype == torch.int32:
            kind = _lib.ACT_INDEX_I32
        else:
            raise TypeError(f"actions dtype {actions.dtype}: expected float32 / float64 steer or int32 index")
        if actions.numel() != self.N or not actions.is_contiguous() or actions.device != self.device:
            raise ValueError("actions must be a contiguous [N] tensor on the env device")
        flags = (_lib.STEP_AUTO_RESET if auto_reset else 0) | (_lib.STEP_SKIP_DONE if skip_done else 0) | \
            (_lib.STEP_ONE_WAVE if one_wave else 0) | (_lib.STEP_APF_LANE if apf_lane else 0)
        if out.obs is None:
            flags |= _lib.STEP_NO_OBS
        _lib.check(self.lib.uavenv_step(self._h, actions.data_ptr(), kind, _ptr(out.obs), _ptr(out.reward),
                                        _ptr(out.reward32), _ptr(out.ret_done), _ptr(out.agent_done), _ptr(out.info),
                                        _ptr(out.valid), _ptr(out.energy), _ptr(active), flags, self._stream()),
                   "uavenv_step")
        return out

    def step_raw(self, actions_ptr: int, kind: int, obs_ptr, reward32_ptr, ret_done_ptr, valid_ptr, flags: int,
                 agent_done_ptr=None, info_ptr=None, reward64_ptr=None):
        """Pointer-level step used by the replay-fused rollout (no tensor bookkeeping on the hot loop)."""
        _lib.check(self.lib.uavenv_step(self._h, actions_ptr, kind, obs_ptr, reward64_ptr, reward32_ptr, ret_done_ptr,
                                        agent_done_ptr, info_ptr, valid_ptr, None, None, flags, self._stream()),
                   "uavenv_step")

    def threaten_rate(self, points: torch.Tensor, allpairs: bool = False) -> torch.Tensor:
        """PathPlan_City.Threaten_rate for [n,3] float64 device points -> uint8[n]."""
        pts = points.to(device=self.device, dtype=torch.float64).contiguous().reshape(-1, 3)
        out = torch.empty(len(pts), dtype=torch.uint8, device=self.device)
        fn = self.lib.uavenv_threaten_rate_allpairs if allpairs else self.lib.uavenv_threaten_rate
        _lib.check(fn(self._h, pts.data_ptr(), out.data_ptr(), len(pts), self._stream()), "uavenv_threaten_rate")
        return out
