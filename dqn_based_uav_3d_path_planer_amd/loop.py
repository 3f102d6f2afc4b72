"""HotLoop -- PathPlan_City.run_thread_OffPolicy (Envs/PathPlan_City.py:364-385) for a whole env shard, enqueued by
csrc/loop.hip: act -> step (+ replay write) -> learn, K steps per call, four launches per step issued from C.

The ring cursor and the learner's update count live in the C object while a HotLoop exists; `run` writes them back to
the DeviceReplayRing / FusedDQNLearner it was built from, so the two stay usable between calls."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from .learner import FusedDQNLearner
from .replay import DeviceReplayRing


class P2PExchangeError(_lib.UavEnvError):
    """uavenv_loop_run returned UAVENV_EP2P (csrc/p2p.hip: timeout or diverged ranks)."""


class HotLoop:
    def __init__(self, ring: DeviceReplayRing, learner: FusedDQNLearner, batch: int, seed: int, eps: float = 0.1,
                 counter: int = 0, learn_start: int = 0, auto_reset: bool = True, skip_done: bool = None,
                 time_every: int = 0, info: torch.Tensor = None, per=None, sample_lag: int = 0,
                 replan_every: int = 0, replan_count: int = 0, replan_max_iter: int = 10000, gate_updates: bool = False,
                 valid_draws: bool = None):
        """per: a replay.DevicePER over the ring's frames * N slots -- prioritised replay (IsPriority_Replay = 1) inside the C
        loop: new-frame priorities, rebuild, ReplayTree.sample, importance weights, the weighted update and batch_update are
        enqueued per pass (csrc/loop.hip); per.beta / per.n_entries are kept in step."""
        if not ring.discrete:
            raise ValueError("HotLoop drives the discrete (DQN-family) path")
        if batch % 64:
            raise ValueError("batch must be a multiple of 64")
        self.lib = _lib.load()
        self.ring, self.learner = ring, learner
        env = ring.env
        if skip_done is None:
            skip_done = env.uav_per_env > 1
        self._partials = learner.new_partials(max(batch, 64))
        cfg = _lib.UavLoopConfig()
        cfg.env = env._h
        cfg.ring = ring._c
        cfg.net = learner.net
        cfg.head, cfg.filled = ring.head, ring.filled
        cfg.batch = int(batch)
        cfg.kind = 0 if learner.kind == "dqn" else 1
        cfg.huber = learner.huber
        cfg.update_loop = learner.update_loop
        cfg.epoch = learner.epoch
        cfg.learn_start = int(learn_start)
        cfg.seed, cfg.counter = int(seed), int(counter)
        cfg.eps, cfg.gamma, cfg.lr = float(eps), learner.gamma, learner.lr
        cfg.beta1, cfg.beta2, cfg.adam_eps = learner.betas[0], learner.betas[1], learner.eps
        cfg.step_flags = (_lib.STEP_AUTO_RESET if auto_reset else 0) | (_lib.STEP_SKIP_DONE if skip_done else 0) | \
            ring.extra_flags
        cfg.partials_dev = self._partials.data_ptr()
        cfg.loss_dev = learner.loss.data_ptr()
        cfg.time_every = int(time_every)
        cfg.sample_lag = int(sample_lag)     # 1: experiment -- update t samples transitions <= t - 1, gradient beside the step
        # rolling refresh of the reset bank: every replan_every passes the loop commits the planned slice and starts planning the
        # next replan_count bank rows on a low-priority stream beside the passes (the reference plans at EVERY reset)
        cfg.replan_every, cfg.replan_count, cfg.replan_max_iter = int(replan_every), int(replan_count), int(replan_max_iter)
        # gate_updates: no learner update behind a step that moved nobody (every agent had already finished: the reference has left
        # its episode loop by then, Envs/PathPlan_City.py:456-459; the fused plugin path only looks every <done_check> steps)
        self._moved = torch.zeros(1, dtype=torch.int32, device=env.device) if gate_updates else None
        if self._moved is not None:
            cfg.moved_dev = self._moved.data_ptr()
        if info is not None:        # [frames, N] uint8: the info code of every transition (episode statistics)
            assert info.dtype == torch.uint8 and tuple(info.shape) == (ring.frames, env.N) and info.is_contiguous()
            cfg.info_dev = info.data_ptr()
        self._info = info
        if getattr(learner, "_p2p", None) is not None:      # multi-GPU: the gradient sum goes through csrc/p2p.hip
            cfg.p2p = learner._p2p
        elif getattr(learner, "_coll", None) is not None:   # ... or through an RCCL all-reduce enqueued from C
            cfg.coll = learner._coll
            cfg.raw_dev = learner.raw.data_ptr()
        self._per = per
        if per is not None:
            if per.capacity != ring.frames * env.N or per._c.rot != 0:
                raise ValueError("the DevicePER must cover the ring's frames * N slots in slot order (tree_order=False)")
            d, b = env.device, max(int(batch), 64)
            self._per_bufs = (torch.zeros(b, dtype=torch.int64, device=d), torch.zeros(b + (b + 255) // 256, dtype=torch.float64, device=d),
                              torch.zeros(b, dtype=torch.float32, device=d), torch.zeros(b, dtype=torch.float32, device=d),
                              torch.zeros((b, 2), dtype=torch.int32, device=d))
            cfg.per = per._c
            cfg.per_alpha, cfg.per_beta, cfg.per_beta_inc = per.alpha, per.beta, per.beta_inc
            cfg.per_eps, cfg.per_clip = per.epsilon, per.clip
            cfg.per_slots_dev, cfg.per_prio_dev, cfg.per_w_dev, cfg.per_abs_dev, cfg.per_idx_dev = \
                (t.data_ptr() for t in self._per_bufs)
        # valid_draws (default: on when finished agents are skipped and not restarted): the rows such agents leave in the ring
        # (valid = 0) are not drawn -- the reference never stores them (Envs/PathPlan_City.py:456-459) -- instead of drawn with weight 0
        if valid_draws is None:
            valid_draws = bool(skip_done) and not auto_reset
        self._draw_idx = None
        if valid_draws and per is None and batch > 0:
            if not skip_done:
                raise ValueError("valid_draws needs skip_done (with auto-reset every stored row is valid)")
            self._draw_idx = torch.zeros((int(batch), 2), dtype=torch.int32, device=env.device)
            cfg.per_idx_dev = self._draw_idx.data_ptr()
        self._h = C.c_void_p()
        _lib.check(self.lib.uavenv_loop_create(C.byref(cfg), C.byref(self._h)), "uavenv_loop_create")
        self.counter = int(counter)
        self.batch = int(batch)
        self._seen = (ring.head, ring.filled, learner.epoch)

    def in_sync(self, batch: int = None) -> bool:
        """The cursor and the update count live in the C object while the loop exists: False when the Python side moved on
        without it (a step or update issued around the loop, Load_Mod setting the epoch, another batch size) -- the owner
        then closes this loop and creates a new one instead of letting run() overwrite the newer values."""
        return (self._seen == (self.ring.head, self.ring.filled, self.learner.epoch) and
                (batch is None or int(batch) == self.batch))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.uavenv_loop_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_eps(self, eps: float):
        _lib.check(self.lib.uavenv_loop_set_eps(self._h, float(eps)), "uavenv_loop_set_eps")

    def run(self, n_steps: int):
        """Enqueue n_steps of act -> step -> learn on the current torch stream (asynchronous)."""
        s = torch.cuda.current_stream(self.ring.env.device).cuda_stream
        rc = self.lib.uavenv_loop_run(self._h, int(n_steps), s)
        if rc == _lib.EP2P:
            self._sync_cursor()
            raise P2PExchangeError("the peer-to-peer gradient exchange raised its sticky error "
                                   f"({self.learner.p2p_status()}): this rank's weights are frozen; fall back to the "
                                   "collective (FusedDQNLearner.enable_coll) and re-broadcast the weights")
        _lib.check(rc, "uavenv_loop_run")
        self._sync_cursor()

    def _sync_cursor(self):
        cur = _lib.UavLoopCursor()
        _lib.check(self.lib.uavenv_loop_get(self._h, C.byref(cur)), "uavenv_loop_get")
        self.ring.head, self.ring.filled = cur.head, cur.filled
        self.learner.epoch = cur.epoch
        self.counter = int(cur.counter)
        self._seen = (cur.head, cur.filled, cur.epoch)
        if self._per is not None:
            beta = C.c_double(0.0)
            _lib.check(self.lib.uavenv_loop_get_per(self._h, C.byref(beta)), "uavenv_loop_get_per")
            self._per.beta = float(beta.value)
            self._per.n_entries = self.ring.filled * self.ring.env.N
            self._per._dirty = True

    def step_times_ms(self, max_n: int = 4096) -> np.ndarray:
        """Durations of the event-bracketed step kernels since the last call (synchronises on them)."""
        buf = np.zeros(max_n, dtype=np.float32)
        n = C.c_int32(0)
        _lib.check(self.lib.uavenv_loop_step_times(self._h, buf.ctypes.data, max_n, C.byref(n)), "uavenv_loop_step_times")
        return buf[:n.value].copy()



class SACHotLoop:
    """The off-policy loop with one fused SAC trainer per UAV slot (BASELINE configs[3]'s shape), enqueued by csrc/loop.hip:
    per step one launch of N(0,1) draws, U x get_action, the env step (replay write included), one draw of (frame, env) pairs
    and U x the four launches of SAC_Trainer.update -- what PathPlan_City._run_eposide_fused_sac issues from Python, bit for
    bit, without the interpreter between the launches.  With the peer exchange (N > 1) the ranks' parameter blocks are hashed and
    compared on the device every check_every updates (uavenv_p2p_check_blocks): a difference raises the exchange's sticky error.  The ring cursor and the learners' update counts live in the C object
    while the loop exists; `run` writes them back.
    exchange (one process per GPU, torch.distributed initialised): "p2p" (csrc/p2p.hip: every phase's column sums of all
    slots summed over peer-mapped HBM on the stream), "coll" (RCCL from C, csrc/coll.hip), "auto" (p2p, else coll) or None;
    `self.exchange` says what is in use (None: one GPU, or neither could be set up -- the caller keeps the Python loop, whose
    learners exchange through torch.distributed).  Every rank must start from the same weights."""

    def __init__(self, ring: DeviceReplayRing, learners, batch: int, seed: int, act1_plane: torch.Tensor, counter: int = 0,
                 info: torch.Tensor = None, is_train: bool = True, auto_reset: bool = True, skip_done: bool = True,
                 exchange: str = None, spin_limit: int = 0, pers=None, check_every: int = 64, gate_updates: bool = False,
                 valid_draws: bool = None):
        """valid_draws (default: on when finished agents are skipped and not restarted): the uniform draws go over the valid rows
        only (uavenv_replay_draw_valid) -- the reference never stores a row for a finished agent (Envs/PathPlan_City.py:456-459).
        pers: one replay.DevicePER per UAV slot (capacity ring.frames * n_envs, tree_order=False) -- prioritised replay, the
        reference's own use of ReplayTree (Trainer/SAC_Trainer.py:336-352), inside the C loop: per step and slot the new frame's
        priorities, rebuild, ReplayTree.sample, importance weights, the four update phases (weights in, |TD| out), batch_update."""
        if ring.discrete or not ring.env.packed:
            raise ValueError("SACHotLoop drives the continuous-action path on a packed ring")
        env = ring.env
        U = env.uav_per_env
        if len(learners) != U or U > _lib.SAC_LOOP_MAX_SLOTS:
            raise ValueError("one FusedSACLearner per UAV slot (at most %d)" % _lib.SAC_LOOP_MAX_SLOTS)
        if batch % 64:
            raise ValueError("batch must be a multiple of 64")
        self.lib = _lib.load()
        self.ring, self.learners = ring, list(learners)
        L0 = self.learners[0]
        for L in self.learners:      # one Trainer.xml: the slots share their hyper-parameters
            if (L.gamma, L.tau, L.action_bound, L.actor_lr, L.critic_lr, L.alpha_lr, L.target_entropy) != \
               (L0.gamma, L0.tau, L0.action_bound, L0.actor_lr, L0.critic_lr, L0.alpha_lr, L0.target_entropy):
                raise ValueError("the SAC slots must share their hyper-parameters")
        d = env.device
        n_envs = env.N // U
        self._draws = torch.zeros((U * batch, 2), dtype=torch.int32, device=d)
        self._noise = torch.zeros(int(self.lib.uavenv_sac_loop_noise_floats(U, n_envs, int(batch))), dtype=torch.float32, device=d)
        cfg = _lib.UavSacLoopConfig()
        cfg.env = env._h
        cfg.ring = ring._c
        assert act1_plane.dtype == torch.float32 and tuple(act1_plane.shape) == (ring.frames, env.N) and act1_plane.is_contiguous()
        cfg.act1_plane = act1_plane.data_ptr()
        ring.attach_action1(act1_plane)      # (Python-issued steps around the loop record the second action component too)
        if info is not None:
            assert info.dtype == torch.uint8 and tuple(info.shape) == (ring.frames, env.N) and info.is_contiguous()
            cfg.info_dev = info.data_ptr()
        cfg.n_slots, cfg.batch = U, int(batch)
        cfg.head, cfg.filled = ring.head, ring.filled
        cfg.is_train = 1 if is_train else 0
        cfg.valid_draws = int(bool(skip_done) and not auto_reset) if valid_draws is None else int(bool(valid_draws))
        cfg.seed, cfg.counter = int(seed), int(counter)
        cfg.beta1, cfg.beta2, cfg.adam_eps = L0.beta1, L0.beta2, L0.adam_eps
        cfg.gamma, cfg.tau, cfg.action_bound = L0.gamma, L0.tau, L0.action_bound
        cfg.actor_lr, cfg.critic_lr, cfg.alpha_lr, cfg.target_entropy = L0.actor_lr, L0.critic_lr, L0.alpha_lr, L0.target_entropy
        cfg.step_flags = (_lib.STEP_AUTO_RESET if auto_reset else 0) | (_lib.STEP_SKIP_DONE if skip_done else 0) | ring.extra_flags
        cfg.draws_dev, cfg.noise_dev = self._draws.data_ptr(), self._noise.data_ptr()
        self._td = []
        for j, L in enumerate(self.learners):
            sl = cfg.slot[j]
            _, pc, pa = L._scratch(int(batch))           # every slot's own partial rows: the phases run for all slots at once
            sl.partials_critic, sl.partials_actor = pc.data_ptr(), pa.data_ptr()
            sl.nets = L._nets
            sl.m_actor, sl.v_actor, sl.alpha_mv = L._blocks[1].data_ptr(), L._blocks[2].data_ptr(), L._alpha_mv.data_ptr()
            sl.m1, sl.v1, sl.m2, sl.v2 = (L._cblocks[k].data_ptr() for k in (4, 5, 6, 7))
            sl.scalars = L._scalars.data_ptr()
            sl.epoch, sl.adam_steps = L.epoch, L.adam_steps
            self._td.append(torch.empty(2 * int(batch), dtype=torch.float32, device=d))     # td targets of the slot's update
            sl.td_dev = self._td[-1].data_ptr()
        self._pers = list(pers) if pers is not None and any(p is not None for p in pers) else None
        if self._pers is not None:
            if len(self._pers) != U or any(p is None for p in self._pers):
                raise ValueError("prioritised replay in the SAC loop: one DevicePER per UAV slot, all slots or none")
            p0 = self._pers[0]
            cfg.per_alpha, cfg.per_beta_inc, cfg.per_eps, cfg.per_clip = p0.alpha, p0.beta_inc, p0.epsilon, p0.clip
            self._per_bufs = []
            for j, p in enumerate(self._pers):
                if p.capacity != ring.frames * n_envs or p._c.rot != 0:
                    raise ValueError("each slot's DevicePER must cover ring.frames * n_envs slots in slot order (tree_order=False)")
                if (p.alpha, p.beta_inc, p.epsilon, p.clip) != (p0.alpha, p0.beta_inc, p0.epsilon, p0.clip):
                    raise ValueError("the slots' prioritised replays must share their hyper-parameters")
                b = int(batch)
                bufs = (torch.zeros(b, dtype=torch.int64, device=d), torch.zeros(b + (b + 255) // 256, dtype=torch.float64, device=d),
                        torch.zeros(b, dtype=torch.float32, device=d), torch.zeros(b, dtype=torch.float32, device=d))
                self._per_bufs.append(bufs)
                sl = cfg.slot[j]
                sl.per = p._c
                sl.per_slots_dev, sl.per_prio_dev, sl.per_w_dev, sl.per_abs_dev = (t.data_ptr() for t in bufs)
                sl.per_beta = p.beta
        self._moved = torch.zeros(1, dtype=torch.int32, device=d) if gate_updates else None      # as HotLoop(gate_updates)
        if self._moved is not None:
            cfg.moved_dev = self._moved.data_ptr()
        self._keep = (act1_plane, info)
        self.exchange, self._p2p, self._coll = None, None, None
        if exchange is not None:
            from . import exchange as ex
            n = U * _lib.SAC_CRITIC_STRIDE
            stream = torch.cuda.current_stream(d).cuda_stream
            if exchange in ("p2p", "auto"):
                h = ex.open_p2p(self.lib, d, n, check_every=0, spin_limit=spin_limit)
                if h is not None and not ex.verify_p2p_allreduce(self.lib, h, d, n, stream):
                    self.lib.uavenv_p2p_destroy(h)
                    h = None
                if h is not None:
                    self._p2p, self.exchange = h, "p2p"
            if self.exchange is None and exchange in ("coll", "auto"):
                h = ex.open_coll(self.lib, d, n, stream)
                if h is not None:
                    self._coll, self.exchange = h, "coll"
            if self.exchange is not None:
                self._xbuf = torch.zeros(n, dtype=torch.float32, device=d)
                cfg.p2p = self._p2p
                cfg.coll = self._coll
                cfg.check_every = int(check_every) if self._p2p is not None else 0
                cfg.xbuf_dev = self._xbuf.data_ptr()
        self._h = C.c_void_p()
        _lib.check(self.lib.uavenv_sac_loop_create(C.byref(cfg), C.byref(self._h)), "uavenv_sac_loop_create")
        self.counter = int(counter)
        self.batch, self.is_train = int(batch), bool(is_train)
        self._seen = (ring.head, ring.filled, tuple((L.epoch, L.adam_steps) for L in self.learners))

    def in_sync(self, batch: int = None, is_train: bool = None) -> bool:
        """As HotLoop.in_sync: the C object captured Is_Train, the batch size, the ring cursor and every learner's epoch /
        adam_steps at creation; False when any of them changed on the Python side since the last run()."""
        return (self._seen == (self.ring.head, self.ring.filled, tuple((L.epoch, L.adam_steps) for L in self.learners)) and
                (batch is None or int(batch) == self.batch) and (is_train is None or bool(is_train) == self.is_train))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.uavenv_sac_loop_destroy(self._h)
            self._h = C.c_void_p()
        if getattr(self, "_p2p", None) is not None:
            self.lib.uavenv_p2p_destroy(self._p2p)
            self._p2p = None
        if getattr(self, "_coll", None) is not None:
            self.lib.uavenv_coll_destroy(self._coll)
            self._coll = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def p2p_status(self, synchronise: bool = True) -> dict:
        """{code, timeouts, mismatches, checks} of the peer exchange (code 1 = timeout, 2 = the ranks' weights diverged)."""
        if self._p2p is None:
            return {"code": 0, "timeouts": 0, "mismatches": 0, "checks": 0}
        out = (C.c_int32 * 4)()
        _lib.check(self.lib.uavenv_p2p_status(self._p2p, 1 if synchronise else 0, out), "uavenv_p2p_status")
        return {"code": int(out[0]), "timeouts": int(out[1]), "mismatches": int(out[2]), "checks": int(out[3])}

    def run(self, n_steps: int):
        s = torch.cuda.current_stream(self.ring.env.device).cuda_stream
        rc = self.lib.uavenv_sac_loop_run(self._h, int(n_steps), s)
        if rc == _lib.EP2P:
            self._sync_cursor()          # the steps before the failure are in the ring and in the learners' counters
            raise P2PExchangeError(f"the peer exchange of the SAC loop raised its sticky error ({self.p2p_status(False)}: code 1 = "
                                   "a peer's block did not arrive in time, 2 = the ranks' weights diverged): stop stepping and "
                                   "re-synchronise parameters and moments from one rank")
        if rc != 0:
            raise _lib.UavEnvError(f"uavenv_sac_loop_run failed with code {rc}: {self.lib.uavenv_sac_last_error().decode()} / "
                                   f"{self.lib.uavenv_last_error().decode()}")
        self._sync_cursor()

    def _sync_cursor(self):
        cur = _lib.UavSacLoopCursor()
        _lib.check(self.lib.uavenv_sac_loop_get(self._h, C.byref(cur)), "uavenv_sac_loop_get")
        self.ring.head, self.ring.filled = cur.head, cur.filled
        self.counter = int(cur.counter)
        for j, L in enumerate(self.learners):
            L.epoch, L.adam_steps = int(cur.epoch[j]), int(cur.adam_steps[j])
        self._seen = (cur.head, cur.filled, tuple((L.epoch, L.adam_steps) for L in self.learners))
        if self._pers is not None:
            beta = (C.c_double * len(self._pers))()
            _lib.check(self.lib.uavenv_sac_loop_get_per(self._h, beta), "uavenv_sac_loop_get_per")
            n_envs = self.ring.env.N // len(self._pers)
            for j, p in enumerate(self._pers):
                p.beta = float(beta[j])
                p.n_entries = self.ring.filled * n_envs
                p._dirty = True
