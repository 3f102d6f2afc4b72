"""Device-resident replay ring (BaseClass/replay_buffer.py:28-54, ReplayMemory) fused with the env kernels.

Frame-major ring in HBM:  frame t = { obs[t] (state BEFORE action t), action[t], reward[t], done[t], valid[t] }.
The env kernel writes obs[t+1] / reward[t] / done[t] / valid[t] directly (uavenv_step's output pointers aim
into the ring), so ReplayMemory.add costs no extra traffic and next_state of (t, i) is simply obs[t+1][i]
(for terminal transitions it is the post-reset observation, which the TD target multiplies by (1-done)=0,
Trainer/DQN_Trainer.py:114).  Capacity in transitions = (frames - 1) * N.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from .env import VecPathPlanEnv


class DeviceReplayRing:
    def __init__(self, env: VecPathPlanEnv, capacity_transitions: int, discrete: bool = True):
        self.env = env
        n = env.N
        self.frames = max(3, -(-int(capacity_transitions) // n) + 1)
        d = env.device
        self.discrete = discrete
        self.obs = torch.zeros((self.frames, n, _lib.OBS_DIM), dtype=env.obs_dtype, device=d)
        self.action = torch.zeros((self.frames, n), dtype=torch.int32 if discrete else torch.float32, device=d)
        self.reward = torch.zeros((self.frames, n), dtype=torch.float32, device=d)
        self.done = torch.zeros((self.frames, n), dtype=torch.uint8, device=d)
        self.valid = torch.zeros((self.frames, n), dtype=torch.uint8, device=d)
        self.head = 0          # frame whose obs is the current state (its action/reward are not written yet)
        self.filled = 0        # complete transitions frames behind head
        self._c = _lib.UavReplayRing(self.obs.data_ptr(), self.action.data_ptr(), self.reward.data_ptr(),
                                     self.done.data_ptr(), self.valid.data_ptr(), self.frames, n,
                                     _lib.OBS_F16 if env.obs_dtype == torch.float16 else _lib.OBS_F32,
                                     1 if discrete else 0)
        self._obs_stride = n * _lib.OBS_DIM * self.obs.element_size()
        self._batch_bufs = {}
        self.extra_flags = 0           # diagnostics (e.g. _lib.STEP_NO_OBS to time the step without the observation)

    @property
    def capacity(self) -> int:
        return (self.frames - 1) * self.env.N

    def __len__(self) -> int:
        return self.filled * self.env.N

    def reset(self, seed: int = 0):
        self.head, self.filled = 0, 0
        self.env.reset(seed, obs=self.obs[0])

    def current_obs(self) -> torch.Tensor:
        return self.obs[self.head]

    def current_action(self) -> torch.Tensor:
        """The slot the policy writes its action for the current frame into."""
        return self.action[self.head]

    def step_env(self, auto_reset: bool = True, skip_done: bool = False):
        """Apply action[head] with the fused kernel; the transition lands in the ring in the same launch."""
        t, nxt = self.head, (self.head + 1) % self.frames
        n = self.env.N
        flags = (_lib.STEP_AUTO_RESET if auto_reset else 0) | (_lib.STEP_SKIP_DONE if skip_done else 0) | self.extra_flags
        kind = _lib.ACT_INDEX_I32 if self.discrete else _lib.ACT_STEER_F32
        self.env.step_raw(self.action.data_ptr() + t * n * 4, kind, self.obs.data_ptr() + nxt * self._obs_stride,
                          self.reward.data_ptr() + t * n * 4, self.done.data_ptr() + t * n,
                          self.valid.data_ptr() + t * n, flags)
        self.head = nxt
        self.filled = min(self.filled + 1, self.frames - 1)

    def _bufs(self, batch: int):
        b = self._batch_bufs.get(batch)
        if b is None:
            d = self.env.device
            b = dict(states=torch.empty((batch, _lib.OBS_DIM), dtype=self.obs.dtype, device=d),
                     next_states=torch.empty((batch, _lib.OBS_DIM), dtype=self.obs.dtype, device=d),
                     actions=torch.empty(batch, dtype=self.action.dtype, device=d),
                     rewards=torch.empty(batch, dtype=torch.float32, device=d),
                     dones=torch.empty(batch, dtype=torch.float32, device=d),
                     valid=torch.empty(batch, dtype=torch.float32, device=d))
            self._batch_bufs[batch] = b
        return b

    def sample(self, batch: int, seed: int, counter: int) -> dict:
        """ReplayMemory.sample2 (replay_buffer.py:48-51): uniform (frame, agent) draws, gathered on device."""
        if self.filled <= 0:
            raise RuntimeError("replay ring is empty")
        b = self._bufs(batch)
        lib = self.env.lib
        rc = lib.uavenv_replay_sample(C.byref(self._c), self.head, self.filled, batch, int(seed), int(counter),
                                      b["states"].data_ptr(), b["next_states"].data_ptr(), b["actions"].data_ptr(),
                                      b["rewards"].data_ptr(), b["dones"].data_ptr(), b["valid"].data_ptr(),
                                      self.env._stream())
        _lib.check(rc, "uavenv_replay_sample")
        return b


def select_actions(env: VecPathPlanEnv, q: torch.Tensor, eps: float, seed: int, counter: int,
                   index_out: torch.Tensor = None, steer_out: torch.Tensor = None):
    """Fused epsilon-greedy over [N, A] Q-values (Trainer/DuelingDQN_Trainer.py:86-97)."""
    q = q.contiguous()
    assert q.dtype == torch.float32 and q.dim() == 2
    rc = env.lib.uavenv_select_actions(q.data_ptr(), q.shape[0], q.shape[1], float(eps), int(seed), int(counter),
                                       None if index_out is None else index_out.data_ptr(),
                                       None if steer_out is None else steer_out.data_ptr(), env._stream())
    _lib.check(rc, "uavenv_select_actions")
