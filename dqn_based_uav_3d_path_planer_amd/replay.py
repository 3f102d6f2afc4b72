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
    def __init__(self, env: VecPathPlanEnv, capacity_transitions: int, discrete: bool = True, records: bool = True):
        """records: keep the 16-byte transition records next to the planes (what the fused learners gather).  A ring nothing
        learns from (a rollout-only measurement) can do without them: the step kernels then skip that store."""
        self.env = env
        n = env.N
        self.frames = max(3, -(-int(capacity_transitions) // n) + 1)
        d = env.device
        self.discrete = discrete
        self.obs = torch.zeros((self.frames, n, env.obs_width), dtype=env.obs_dtype, device=d)
        self.action = torch.zeros((self.frames, n), dtype=torch.int32 if discrete else torch.float32, device=d)
        self.reward = torch.zeros((self.frames, n), dtype=torch.float32, device=d)
        self.done = torch.zeros((self.frames, n), dtype=torch.uint8, device=d)
        self.valid = torch.zeros((self.frames, n), dtype=torch.uint8, device=d)
        # transition records (ABI 5: UavReplayRing.meta): {a1, a0, reward, done | valid << 8 | info << 16} per (frame, agent), written by
        # the step kernels next to the planes; the fused learners gather ONE 16-byte record per sample instead of a line from each plane
        self.meta = torch.zeros((self.frames, n, 4), dtype=torch.int32, device=d) if records else None
        self.action1 = None    # [frames, N] f32, optional: the second action component (SAC) for the records (attach_action1)
        self.head = 0          # frame whose obs is the current state (its action/reward are not written yet)
        self.filled = 0        # complete transitions frames behind head
        self._c = _lib.UavReplayRing(self.obs.data_ptr(), self.action.data_ptr(), self.reward.data_ptr(),
                                     self.done.data_ptr(), self.valid.data_ptr(), self.frames, n,
                                     env.obs_code, 1 if discrete else 0, None if self.meta is None else self.meta.data_ptr())
        self._obs_stride = n * env.obs_width * self.obs.element_size()
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

    def attach_action1(self, plane: torch.Tensor):
        """The plane that holds the second action component of every frame (SAC_Trainer.get_action's action[1], which the env never
        reads): from now on the step launches copy it into the transition records, so that a learner given `meta` finds both."""
        assert plane.dtype == torch.float32 and tuple(plane.shape) == (self.frames, self.env.N) and plane.is_contiguous()
        self.action1 = plane

    def rewrite_records(self, frames=None):
        """INVARIANT: the action / reward / done / valid planes (and action1) are written by step launches only, which write the
        transition record of the same (frame, agent) in the same launch -- the fused learners read the RECORDS.  Code that edits a plane
        by hand (a test injecting rewards, a tool replaying a log) calls this afterwards: it rebuilds the records of the given frames
        (default: all) from the planes; the info byte, which has no plane in the ring, is kept."""
        if self.meta is None:
            return
        idx = slice(None) if frames is None else torch.as_tensor(frames, device=self.meta.device, dtype=torch.long).reshape(-1)
        m = self.meta[idx]
        a0 = self.action[idx]
        m[..., 1] = a0 if self.discrete else a0.contiguous().view(torch.int32)
        m[..., 0] = 0 if self.action1 is None else self.action1[idx].contiguous().view(torch.int32)
        m[..., 2] = self.reward[idx].contiguous().view(torch.int32)
        m[..., 3] = (m[..., 3] & 0x00FF0000) | self.done[idx].int() | (self.valid[idx].int() << 8)
        self.meta[idx] = m

    def _set_step_meta(self, t: int):
        if self.meta is None:
            return
        n = self.env.N
        a1 = None if self.action1 is None else self.action1.data_ptr() + t * n * 4
        _lib.check(self.env.lib.uavenv_set_step_meta(self.env._h, self.meta.data_ptr() + t * n * _lib.META_BYTES, a1),
                   "uavenv_set_step_meta")

    def current_obs(self) -> torch.Tensor:
        return self.obs[self.head]

    def current_action(self) -> torch.Tensor:
        """The slot the policy writes its action for the current frame into."""
        return self.action[self.head]

    def step_env(self, auto_reset: bool = True, skip_done: bool = None, info: torch.Tensor = None):
        """Apply action[head] with the fused kernel; the transition lands in the ring in the same launch.
        skip_done defaults to True when an env holds several UAVs: an env only auto-resets once ALL its agents are
        done, and the reference never steps a finished agent while it waits (PathPlan_City.py:365-366) -- its rows
        land in the ring with valid = 0.  info (optional, uint8 [frames, N]): receives the step's info codes, frame-major
        like reward."""
        if skip_done is None:
            skip_done = self.env.uav_per_env > 1
        t, nxt = self.head, (self.head + 1) % self.frames
        n = self.env.N
        flags = (_lib.STEP_AUTO_RESET if auto_reset else 0) | (_lib.STEP_SKIP_DONE if skip_done else 0) | self.extra_flags
        kind = _lib.ACT_INDEX_I32 if self.discrete else _lib.ACT_STEER_F32
        self._set_step_meta(t)
        self.env.step_raw(self.action.data_ptr() + t * n * 4, kind, self.obs.data_ptr() + nxt * self._obs_stride,
                          self.reward.data_ptr() + t * n * 4, self.done.data_ptr() + t * n,
                          self.valid.data_ptr() + t * n, flags,
                          info_ptr=None if info is None else info.data_ptr() + t * n)
        self.head = nxt
        self.filled = min(self.filled + 1, self.frames - 1)

    def step_policy(self, learner, eps: float, seed: int, counter: int, auto_reset: bool = True, skip_done: bool = None,
                    image: torch.Tensor = None, agent_done: torch.Tensor = None) -> bool:
        """get_action + step in ONE launch (uavenv_step_policy: Q(s) + epsilon-greedy in the step kernel's prologue, the
        launch csrc/loop.hip issues per pass).  Returns False -- nothing enqueued -- when this env / net cannot take it
        (the callers then issue learner.act + step_env)."""
        if skip_done is None:
            skip_done = self.env.uav_per_env > 1
        t, nxt = self.head, (self.head + 1) % self.frames
        n = self.env.N
        flags = (_lib.STEP_AUTO_RESET if auto_reset else 0) | (_lib.STEP_SKIP_DONE if skip_done else 0) | self.extra_flags
        # (image: learner.split_image() -- the launch as the C loop issues it, csrc/dqn_internal.hpp; same actions)
        fn = self.env.lib.uavenv_step_policy if image is None else self.env.lib.uavenv_step_policy_img
        tail = (self.env._stream(),) if image is None else (image.data_ptr(), self.env._stream())
        self._set_step_meta(t)
        rc = fn(self.env._h, C.byref(learner.net), self.obs.data_ptr() + t * self._obs_stride,
                float(eps), int(seed), int(counter), self.action.data_ptr() + t * n * 4,
                self.obs.data_ptr() + nxt * self._obs_stride, None,
                self.reward.data_ptr() + t * n * 4, self.done.data_ptr() + t * n,
                None if agent_done is None else agent_done.data_ptr(), None,      # (agent_done: uint8 [N], which agents ended an episode)
                self.valid.data_ptr() + t * n, None, None, flags, *tail)
        if rc == _lib.EINVAL:
            return False
        _lib.check(rc, "uavenv_step_policy")
        self.head = nxt
        self.filled = min(self.filled + 1, self.frames - 1)
        return True

    def _bufs(self, batch: int):
        b = self._batch_bufs.get(batch)
        if b is None:
            d = self.env.device
            w = self.env.obs_width
            b = dict(states=torch.empty((batch, w), dtype=self.obs.dtype, device=d),
                     next_states=torch.empty((batch, w), dtype=self.obs.dtype, device=d),
                     actions=torch.empty(batch, dtype=self.action.dtype, device=d),
                     rewards=torch.empty(batch, dtype=torch.float32, device=d),
                     dones=torch.empty(batch, dtype=torch.float32, device=d),
                     valid=torch.empty(batch, dtype=torch.float32, device=d))
            self._batch_bufs[batch] = b
        return b

    def gather(self, slots: torch.Tensor) -> dict:
        """The transitions at data slots frame * N + agent (what a prioritised sampler returns), as a batch dict."""
        n = self.env.N
        f = torch.div(slots, n, rounding_mode="floor")
        nxt = ((f + 1) % self.frames) * n + (slots - f * n)
        flat = self.obs.view(-1, self.env.obs_width)
        return dict(states=self.env.unpack(flat[slots]) if self.env.packed else flat[slots],
                    next_states=self.env.unpack(flat[nxt]) if self.env.packed else flat[nxt],
                    actions=self.action.view(-1)[slots],
                    rewards=self.reward.view(-1)[slots], dones=self.done.view(-1)[slots].float(),
                    valid=self.valid.view(-1)[slots].float())

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
        if self.env.packed:         # torch-side consumers get ordinary f32 rows; the packed batch stays available
            return dict(b, states=self.env.unpack(b["states"]), next_states=self.env.unpack(b["next_states"]),
                        packed_states=b["states"], packed_next_states=b["next_states"])
        return b


class DevicePER:
    """ReplayTree (BaseClass/replay_buffer.py:121-223) on the device: one f64 priority per data slot in HBM, selection
    by the two-level prefix search of csrc/per.hip.  Hyper-parameters and update rules are the reference's:
    push -> (|error| + epsilon) ** alpha (:143), batch_update -> min(|error| + epsilon, clip) ** alpha (:215-222),
    beta += beta_inc per sample() (:155), weights (n_entries * p / int(total)) ** -beta / max (:175-178).

    With a replay ring the data slot is frame * N + agent: `on_frame` gives the N transitions k_step has just written
    the "new transition" priority and clears the frame that became the ring's head (its rows are being reused)."""

    def __init__(self, capacity: int, device="cuda:0", alpha: float = 0.6, beta: float = 0.4, beta_inc: float = 0.001,
                 epsilon: float = 0.01, clip: float = 1.0, tree_order: bool = True):
        self.lib = _lib.load()
        self.device = torch.device(device)
        self.capacity = int(capacity)
        self.alpha, self.beta, self.beta_inc, self.epsilon, self.clip = alpha, beta, beta_inc, epsilon, clip
        nc = self.lib.uavenv_per_num_chunks(self.capacity)
        self.prio = torch.zeros(self.capacity, dtype=torch.float64, device=self.device)
        self._chunk_sum = torch.zeros(nc, dtype=torch.float64, device=self.device)
        self._chunk_prefix = torch.zeros(nc + 1, dtype=torch.float64, device=self.device)
        self._group_sum = torch.zeros(nc * 64, dtype=torch.float64, device=self.device)     # 16-leaf sums: the sampler's fast path
        rot = self.lib.uavenv_per_rotation(self.capacity) if tree_order else 0
        self._c = _lib.UavPer(self.prio.data_ptr(), self._chunk_sum.data_ptr(), self._chunk_prefix.data_ptr(),
                              self.capacity, rot, self._group_sum.data_ptr())
        self.n_entries = 0
        self._dirty = True

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def __len__(self) -> int:
        return int(self.total())                                                   # ReplayTree.__len__ :133-134

    def total(self) -> float:
        self._rebuild()
        return float(self._chunk_prefix[-1].item())

    def _rebuild(self):
        if self._dirty:
            _lib.check(self.lib.uavenv_per_rebuild(C.byref(self._c), self._stream()), "uavenv_per_rebuild")
            self._dirty = False

    def set_priorities(self, prio: torch.Tensor, n_entries: int = None):
        self.prio.copy_(prio.to(self.device, torch.float64))
        self.n_entries = self.capacity if n_entries is None else int(n_entries)
        self._dirty = True

    def fill(self, first: int, count: int, error: float = 0.0, valid: torch.Tensor = None, zero: bool = False):
        """ReplayTree.push for `count` consecutive slots with the same error (valid[i] == 0 -> priority 0)."""
        p = 0.0 if zero else (abs(float(error)) + self.epsilon) ** self.alpha
        rc = self.lib.uavenv_per_fill(C.byref(self._c), int(first), int(count), p,
                                      None if valid is None else valid.data_ptr(), self._stream())
        _lib.check(rc, "uavenv_per_fill")
        self._dirty = True

    def on_frame(self, ring: "DeviceReplayRing"):
        """Call after ring.step_env(): prioritise the transitions of the frame just completed, retire the new head."""
        n = ring.env.N
        t = (ring.head - 1) % ring.frames
        self.fill(t * n, n, 0.0, valid=ring.valid[t])
        self.fill(ring.head * n, n, zero=True)
        self.n_entries = ring.filled * n

    def update(self, slots: torch.Tensor, abs_errors: torch.Tensor, assume_sorted: bool = False):
        """ReplayTree.batch_update (:215-222).  assume_sorted: the list comes from uavenv_per_sample (prefix order: equal slots are
        adjacent already) -- nothing to order.  Otherwise the list is stably sorted ON THE DEVICE, unconditionally: asking whether it
        needs sorting would be a device-to-host round trip per priority update."""
        slots = slots.to(self.device, torch.int64).reshape(-1).contiguous()
        e = abs_errors.detach().to(self.device, torch.float64).reshape(-1).contiguous()
        # uavenv_per_set resolves a slot listed several times only when the equal entries are ADJACENT (include/uavenv.h); this
        # entry point takes arbitrary lists, so order them first -- a stable sort keeps the batch order among equal slots, i.e.
        # the LAST error of a slot still wins, as in the reference's sequential loop
        if slots.numel() > 1 and not assume_sorted:
            slots, order = torch.sort(slots, stable=True)
            e = e[order].contiguous()
        rc = self.lib.uavenv_per_set(C.byref(self._c), slots.data_ptr(), e.data_ptr(), slots.numel(), self.epsilon,
                                     self.alpha, self.clip, self._stream())
        _lib.check(rc, "uavenv_per_set")
        self._dirty = True

    def sample_into(self, batch: int, seed: int, counter: int, bufs: dict, n_agents: int):
        """ReplayTree.sample without a host round trip (the fused learners' form): slots -> bufs['slots'] (int64 [batch]),
        priorities -> bufs['prio'] (f64 [batch + (batch + 255) // 256]), importance weights -> bufs['w'] (f32 [batch]) and
        the (frame, agent-of-the-frame) split of every slot -> bufs['pairs'] (int32 [batch, 2]; slot = frame * n_agents +
        agent).  Four launches on the current stream."""
        self._rebuild()
        self.beta = min(1.0, self.beta + self.beta_inc)
        s = self._stream()
        _lib.check(self.lib.uavenv_per_sample(C.byref(self._c), int(batch), None, int(seed), int(counter), bufs["slots"].data_ptr(),
                                              bufs["prio"].data_ptr(), s), "uavenv_per_sample")
        _lib.check(self.lib.uavenv_per_weights(C.byref(self._c), bufs["slots"].data_ptr(), bufs["prio"].data_ptr(), int(batch),
                                               int(self.n_entries), float(self.beta), int(n_agents), bufs["w"].data_ptr(),
                                               bufs["pairs"].data_ptr(), s), "uavenv_per_weights")

    def make_bufs(self, batch: int, pairs: torch.Tensor = None) -> dict:
        d = self.device
        return dict(slots=torch.zeros(batch, dtype=torch.int64, device=d),
                    prio=torch.zeros(batch + (batch + 255) // 256, dtype=torch.float64, device=d),
                    w=torch.zeros(batch, dtype=torch.float32, device=d), abs=torch.zeros(batch, dtype=torch.float32, device=d),
                    pairs=pairs if pairs is not None else torch.zeros((batch, 2), dtype=torch.int32, device=d))

    def update_f32(self, slots: torch.Tensor, abs_errors: torch.Tensor, go=None):
        """ReplayTree.batch_update (:215-222) from f32 |TD errors| already on the device.  go = (device word, value): nothing
        changes unless the word holds the value (uavenv_set_moved_word)."""
        gw, gv = go if go is not None else (None, 0)
        _lib.check(self.lib.uavenv_per_set_f32_gated(C.byref(self._c), slots.data_ptr(), abs_errors.data_ptr(), slots.numel(),
                                                     self.epsilon, self.alpha, self.clip, gw, int(gv) & 0xffffffff, self._stream()),
                   "uavenv_per_set_f32")
        self._dirty = True

    def sample(self, batch: int, seed: int = 0, counter: int = 0, draws: torch.Tensor = None):
        """ReplayTree.sample (:146-180) -> (slots int64 [batch], is_weights float64 [batch], priorities)."""
        self._rebuild()
        self.beta = min(1.0, self.beta + self.beta_inc)
        slots = torch.empty(batch, dtype=torch.int64, device=self.device)
        p = torch.empty(batch, dtype=torch.float64, device=self.device)
        if draws is not None:
            draws = draws.to(self.device, torch.float64).contiguous()
        rc = self.lib.uavenv_per_sample(C.byref(self._c), batch, None if draws is None else draws.data_ptr(), int(seed),
                                        int(counter), slots.data_ptr(), p.data_ptr(), self._stream())
        _lib.check(rc, "uavenv_per_sample")
        # SumTree.total() :117-118 is int(total): 0 while the priorities sum to less than 1, and the reference then
        # divides by zero (:176).  Here the divisor is clamped to 1, and a zero-priority pick (only an all-zero chunk can
        # yield one, csrc/per.hip) gets weight 0 instead of pow(0, -beta) = inf turning the whole batch into NaN.
        total_int = torch.floor(self._chunk_prefix[-1]).clamp_min(1.0)
        live = p > 0
        w = torch.pow(self.n_entries * (torch.where(live, p, torch.ones_like(p)) / total_int), -self.beta)
        w = torch.where(live, w, torch.zeros_like(w))
        w = w / w.max().clamp_min(torch.finfo(torch.float64).tiny)
        return slots, w, p


def select_actions(env: VecPathPlanEnv, q: torch.Tensor, eps: float, seed: int, counter: int,
                   index_out: torch.Tensor = None, steer_out: torch.Tensor = None):
    """Fused epsilon-greedy over [N, A] Q-values (Trainer/DuelingDQN_Trainer.py:86-97)."""
    q = q.contiguous()
    assert q.dtype == torch.float32 and q.dim() == 2
    rc = env.lib.uavenv_select_actions(q.data_ptr(), q.shape[0], q.shape[1], float(eps), int(seed), int(counter),
                                       None if index_out is None else index_out.data_ptr(),
                                       None if steer_out is None else steer_out.data_ptr(), env._stream())
    _lib.check(rc, "uavenv_select_actions")
