"""SAC with continuous actions on PyTorch-ROCm -- Trainer/SAC_Trainer.py:325-379 (update), :122-131 (calc_target),
:145-147 (soft_update), :444-448 (get_action); BASELINE config 4's trainer (the reference's shipped default).
Same tensor shapes and broadcasting as the reference (critics emit action_dim=2 values, td_target is [B,2])."""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist
import torch.nn.functional as F

from .nets import create_network


class SACLearner:
    def __init__(self, param: dict, device="cuda:0", capturable: bool = False):
        """capturable: build the Adam optimizers with capturable=True so that a whole update can be recorded into a HIP
        graph (torch.cuda.graph) and replayed with one launch -- the update is ~150 small kernels, launch-bound in eager
        mode."""
        ap, cp, sp = param.get("actor"), param.get("critic"), param.get("SAC_param")
        self.device = torch.device(device)
        kw = dict(capturable=True) if capturable and self.device.type == "cuda" else {}
        mk = lambda p: create_network(p).to(self.device)   # noqa: E731
        self.actor = mk(ap)
        self.critic_1, self.critic_2 = mk(cp), mk(cp)
        self.target_critic_1, self.target_critic_2 = mk(cp), mk(cp)
        self.target_critic_1.load_state_dict(self.critic_1.state_dict())
        self.target_critic_2.load_state_dict(self.critic_2.state_dict())
        self.actor_optimizer = torch.optim.Adam(self.actor.parameters(), lr=float(ap.get("lr")), **kw)
        self.critic_1_optimizer = torch.optim.Adam(self.critic_1.parameters(), lr=float(cp.get("lr")), **kw)
        self.critic_2_optimizer = torch.optim.Adam(self.critic_2.parameters(), lr=float(cp.get("lr")), **kw)
        self.log_alpha = torch.tensor(np.log(0.01), dtype=torch.float, device=self.device, requires_grad=True)
        self.log_alpha_optimizer = torch.optim.Adam([self.log_alpha], lr=float(sp.get("alpha_lr")), **kw)
        self.target_entropy = float(sp.get("target_entropy"))
        self.gamma, self.tau = float(sp.get("gamma")), float(sp.get("tau"))
        self.epoch = 0
        self.loss = torch.zeros((), device=self.device)

    def calc_target(self, rewards, next_states, dones, eps=None):
        next_actions, log_prob = self.actor(next_states, eps)
        entropy = -log_prob
        q1 = self.target_critic_1(next_states, next_actions)
        q2 = self.target_critic_2(next_states, next_actions)
        next_value = torch.min(q1, q2) + self.log_alpha.exp() * entropy
        return rewards + self.gamma * next_value * (1 - dones)

    def soft_update(self, net, target_net):
        with torch.no_grad():
            for pt, p in zip(target_net.parameters(), net.parameters()):
                pt.copy_(pt * (1.0 - self.tau) + p * self.tau)

    def _sync_grads(self, params):
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            flat = torch.cat([p.grad.reshape(-1) for p in params])
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            flat /= dist.get_world_size()
            off = 0
            for p in params:
                n = p.numel()
                p.grad.copy_(flat[off:off + n].view_as(p))
                off += n

    def learn(self, batch: dict, noise=None, is_weights=None, valid=None):
        """One SAC_Trainer.update (continuous branch).  batch: states [B,100], actions [B,2], rewards, next_states,
        dones.  noise = (eps_next, eps_cur) optionally pins the two rsample() draws.  is_weights [B] (prioritised
        replay, IsPriority_Replay == 1): the critic losses become mean_i w_i * err_i^2 and self.abs_errors holds
        |min(Q1, Q2) - td_target| for ReplayTree.batch_update (SAC_Trainer.py:346-352; the reference's own
        `is_weights * critic_loss` there is a vector and its backward() raises).  valid [B] (0 / 1): rows that are not
        replay memory in the reference (a finished agent waiting for its team-mates, PathPlan_City.py:365-366, re-emitted
        by the vectorised step) -- they carry weight 0 in EVERY loss (critics, actor, log_alpha) and every mean is over the
        valid rows, so the update equals the reference's on the valid rows alone."""
        self.epoch += 1
        states, next_states = batch["states"].float(), batch["next_states"].float()
        actions = batch["actions"].float().reshape(len(states), -1)
        rewards, dones = batch["rewards"].float().view(-1, 1), batch["dones"].float().view(-1, 1)
        e_next, e_cur = noise if noise is not None else (None, None)
        if valid is not None:
            vm = valid.to(states.dtype).view(-1, 1)
            inv_n = 1.0 / vm.sum().clamp_min(1.0)

            def mean(x):                                       # mean over the valid rows (and the 2 output columns)
                return (vm * x).sum() * inv_n / x.shape[1]
        else:
            mean = torch.mean
        td_target = self.calc_target(rewards, next_states, dones, e_next)
        q1, q2 = self.critic_1(states, actions), self.critic_2(states, actions)
        if is_weights is None:
            critic_1_loss = mean((q1 - td_target.detach()) ** 2)
            critic_2_loss = mean((q2 - td_target.detach()) ** 2)
        else:
            w = is_weights.to(q1.dtype).view(-1, 1)
            critic_1_loss = mean(w * (q1 - td_target.detach()) ** 2)
            critic_2_loss = mean(w * (q2 - td_target.detach()) ** 2)
            self.abs_errors = torch.abs(torch.min(q1, q2) - td_target).detach()[:, 0]      # :351 .squeeze() of [B,2] rows
        self.critic_losses = (critic_1_loss.detach(), critic_2_loss.detach())
        for opt, loss, net in ((self.critic_1_optimizer, critic_1_loss, self.critic_1),
                               (self.critic_2_optimizer, critic_2_loss, self.critic_2)):
            opt.zero_grad()
            loss.backward()
            self._sync_grads(list(net.parameters()))
            opt.step()
        new_actions, log_prob = self.actor(states, e_cur)
        entropy = -log_prob
        actor_loss = mean(-self.log_alpha.exp() * entropy -
                          torch.min(self.critic_1(states, new_actions), self.critic_2(states, new_actions)))
        self.actor_optimizer.zero_grad()
        actor_loss.backward()
        self._sync_grads(list(self.actor.parameters()))
        self.actor_optimizer.step()
        alpha_loss = mean((entropy - self.target_entropy).detach() * self.log_alpha.exp())
        self.log_alpha_optimizer.zero_grad()
        alpha_loss.backward()
        self._sync_grads([self.log_alpha])
        self.log_alpha_optimizer.step()
        self.soft_update(self.critic_1, self.target_critic_1)
        self.soft_update(self.critic_2, self.target_critic_2)
        self.loss = actor_loss.detach()
        return self.loss

    def act(self, states: torch.Tensor, eps=None) -> torch.Tensor:
        """[n,100] -> actions [n,2] (only [:,0] steers: Agents/UAV.py:414)."""
        with torch.no_grad():
            return self.actor(states.to(self.device).float(), eps)[0]


class FusedSACLearner:
    """The same update as SACLearner.learn (SAC_Trainer.py:325-379, continuous branch) as four hand-written HIP launches
    (csrc/sac.hip: critic_grad -> critic_adam -> actor_grad -> actor_adam) on packed observation rows in place.
    `actor`, `critic_1`, ... are ordinary modules with the reference's parameter names whose parameters are VIEWS of the
    kernels' flat blocks, so acting, state_dict() and load_state_dict() work as on SACLearner.  Shapes are the
    reference's shipped config (w 100, hiden_dim 64, two action components); anything else raises."""

    def __init__(self, param: dict, device="cuda:0"):
        import ctypes as C
        from . import _lib
        self._C, self._lib = C, _lib
        self.lib = _lib.load()
        ap, cp, sp = param.get("actor"), param.get("critic"), param.get("SAC_param")
        if (int(ap.get("w")), int(ap.get("hiden_dim")), int(ap.get("output"))) != (100, 64, 2) or \
           (int(cp.get("w")), int(cp.get("hiden_dim")), int(cp.get("action_dim"))) != (100, 64, 2):
            raise ValueError("FusedSACLearner handles the reference's 100-64-(2+2) actor and 102-64-64-2 critics only")
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.UavEnvError("FusedSACLearner needs a GPU (no CPU fallback); use SACLearner")
        d = self.device
        pa, pc = _lib.SAC_ACTOR_PARAMS, _lib.SAC_CRITIC_PARAMS
        pad = lambda n: (n + 3) & ~3      # noqa: E731
        self._blocks = torch.zeros((3, pad(pa)), dtype=torch.float32, device=d)            # actor: params, exp_avg, exp_avg_sq
        self._cblocks = torch.zeros((8, pad(pc)), dtype=torch.float32, device=d)           # c1 c2 t1 t2 | m1 v1 m2 v2
        mk = lambda p: create_network(p).to(d)   # noqa: E731
        self.actor = mk(ap)
        self.critic_1, self.critic_2 = mk(cp), mk(cp)
        self.target_critic_1, self.target_critic_2 = mk(cp), mk(cp)
        self.target_critic_1.load_state_dict(self.critic_1.state_dict())
        self.target_critic_2.load_state_dict(self.critic_2.state_dict())
        self._bind(self.actor, self._blocks[0], ("fc1.weight", "fc1.bias", "fc_mu.weight", "fc_std.weight", "fc_mu.bias", "fc_std.bias"))
        for k, net in enumerate((self.critic_1, self.critic_2, self.target_critic_1, self.target_critic_2)):
            self._bind(net, self._cblocks[k], ("fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias", "fc_out.weight", "fc_out.bias"))
        self.log_alpha = torch.tensor(np.log(0.01), dtype=torch.float32, device=d)
        self._alpha_mv = torch.zeros(2, dtype=torch.float32, device=d)
        self.actor_lr, self.critic_lr, self.alpha_lr = float(ap.get("lr")), float(cp.get("lr")), float(sp.get("alpha_lr"))
        self.target_entropy = float(sp.get("target_entropy"))
        self.gamma, self.tau = float(sp.get("gamma")), float(sp.get("tau"))
        self.action_bound = float(ap.get("action_bound"))
        self.beta1, self.beta2, self.adam_eps = 0.9, 0.999, 1e-8                           # torch.optim.Adam defaults
        self.epoch = 0                 # SAC_Trainer.update calls (incl. the warm-up / Is_Train = 0 no-ops, as the reference counts)
        self.adam_steps = 0            # updates actually taken: torch.optim.Adam's own `step` (bias correction)
        self._scalars = torch.zeros(8, dtype=torch.float32, device=d)                      # critic losses [0:4], actor [4:8]
        self._partials = {}
        self._raw = {}
        self._nets = _lib.UavSacNets(self._blocks[0].data_ptr(), self._cblocks[0].data_ptr(), self._cblocks[1].data_ptr(),
                                     self._cblocks[2].data_ptr(), self._cblocks[3].data_ptr(), self.log_alpha.data_ptr())

    @staticmethod
    def _bind(net, flat, names):
        params = dict(net.named_parameters())
        off = 0
        with torch.no_grad():
            for nm in names:
                p = params[nm]
                n = p.numel()
                flat[off:off + n].copy_(p.reshape(-1))
                p.data = flat[off:off + n].view_as(p)
                off += n

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def _scratch(self, batch: int):
        """(rows held, critic rows buffer, actor rows buffer): one row per 64-sample tile -- the most any partition writes.  A
        launch writes one row per WORKGROUP: uavenv_sac_partial_rows_n(batch, slots in the launch, UavSacBatch.tiles_per_wg)."""
        s = self._partials.get(batch)
        if s is None:
            if self.lib.uavenv_sac_partial_rows(int(batch)) <= 0:
                raise ValueError("batch must be a positive multiple of 64")
            cap = int(batch) // 64
            s = (cap, torch.empty((cap, self._lib.SAC_CRITIC_STRIDE), dtype=torch.float32, device=self.device),
                 torch.empty((cap, self._lib.SAC_ACTOR_STRIDE), dtype=torch.float32, device=self.device))
            self._partials[batch] = s
        return s

    def _check(self, rc, what):
        if rc != 0:
            raise self._lib.UavEnvError(f"{what} failed with code {rc}: {self.lib.uavenv_sac_last_error().decode()}")

    def make_batch(self, obs_packed: torch.Tensor, act0, act1, reward, done, *, valid=None, idx_s=None, idx_n=None,
                   draws=None, n_agents=0, uav_per_env=1, slot=0, frames=0, is_weights=None, abs_td_out=None, tiles_per_wg=0,
                   meta=None):
        """Describe the sampled transitions in place (see UavSacBatch in include/uavenv.h).  Keeps the tensors alive.
        is_weights / abs_td_out (f32 [batch], prioritised replay): importance weights into the critic losses, and where
        uavenv_sac_critic_grad leaves |min(Q1, Q2) - td_target| per sample.  tiles_per_wg: 0 = chosen per launch; > 0 pins the
        partition of the partial rows (and with it the summation order) whatever shares the launch.
        meta: the ring's transition records (DeviceReplayRing.meta, flattened) -- the kernels then gather one 16-byte record per
        sample instead of a line from each of act0 / act1 / reward / done / valid.  Only for a ring whose steps recorded the second
        action component (DeviceReplayRing.attach_action1)."""
        keep = (obs_packed, act0, act1, reward, done, valid, idx_s, idx_n, draws, is_weights, abs_td_out, meta)
        assert meta is None or (meta.dtype == torch.int32 and meta.is_contiguous())
        for t in (is_weights, abs_td_out):
            assert t is None or (t.dtype == torch.float32 and t.is_contiguous())
        n = int(draws.shape[0]) if draws is not None else int(idx_s.numel())
        ptr = lambda t: None if t is None else t.data_ptr()      # noqa: E731
        for t, dt in ((act0, torch.float32), (act1, torch.float32), (reward, torch.float32), (done, torch.uint8)):
            assert t.dtype == dt and t.is_contiguous()
        assert valid is None or valid.dtype == torch.uint8
        assert draws is None or (draws.dtype == torch.int32 and draws.is_contiguous())
        assert idx_s is None or (idx_s.dtype == torch.int32 and idx_n.dtype == torch.int32)
        td = torch.empty(2 * n, dtype=torch.float32, device=self.device)      # the td targets of an update (k_sac_td -> k_sac_critic_grad)
        b = self._lib.UavSacBatch(obs_packed.data_ptr(), ptr(idx_s), ptr(idx_n), ptr(draws), int(n_agents), int(uav_per_env),
                                  int(slot), int(frames), act0.data_ptr(), act1.data_ptr(), reward.data_ptr(), done.data_ptr(),
                                  ptr(valid), None, n, int(tiles_per_wg), ptr(is_weights), ptr(abs_td_out), ptr(meta), td.data_ptr())
        b._keep = keep + (td,)
        return b

    def _adam(self, lr, tau=0.0, scale=0.0):
        t = max(self.adam_steps, 1)
        # behind a peer exchange that raised its sticky error the Adam launch must change nothing (csrc/p2p.hip)
        skip = self.lib.uavenv_p2p_error_word(self._p2p) if getattr(self, "_p2p", None) is not None else None
        # go = (device word, value): the update changes nothing unless the word holds the value (uavenv_set_moved_word: the step
        # of this pass moved at least one agent); None = always
        go = getattr(self, "go", None) or (None, 0)
        return self._lib.UavSacAdam(lr, self.beta1, self.beta2, self.adam_eps, 1.0 - self.beta1 ** t,
                                    float(np.sqrt(1.0 - self.beta2 ** t)), tau, scale, skip, go[0], int(go[1]) & 0xffffffff, 0)

    def enable_exchange(self, kind: str = "auto", spin_limit: int = 0):
        """The on-stream form of the N > 1 exchange (no host round trip, no torch.distributed call per phase): "p2p" =
        uavenv_p2p_allreduce over peer-mapped HBM (csrc/p2p.hip), "coll" = the RCCL communicator driven from C
        (csrc/coll.hip), "auto" = p2p, else coll.  Verified against torch.distributed on set-up; returns what is in use on
        ALL ranks ("p2p" / "coll") or None (the torch.distributed path stays)."""
        from . import exchange as ex
        self.disable_exchange()
        n = self._lib.SAC_CRITIC_STRIDE
        if kind in ("p2p", "auto"):
            h = ex.open_p2p(self.lib, self.device, n, check_every=0, spin_limit=spin_limit)
            if h is not None and not ex.verify_p2p_allreduce(self.lib, h, self.device, n, self._stream()):
                self.lib.uavenv_p2p_destroy(h)
                h = None
            if h is not None:
                self._p2p = h
                return "p2p"
        if kind in ("coll", "auto"):
            h = ex.open_coll(self.lib, self.device, n, self._stream())
            if h is not None:
                self._coll = h
                return "coll"
        return None

    def disable_exchange(self):
        if getattr(self, "_p2p", None) is not None:
            self.lib.uavenv_p2p_destroy(self._p2p)
        if getattr(self, "_coll", None) is not None:
            self.lib.uavenv_coll_destroy(self._coll)
        self._p2p, self._coll = None, None

    def _exchange(self, partials: torch.Tensor, rows: int):
        """Multi-GPU (one process per GPU, torch.distributed initialised): this rank's column sums, summed over the ranks
        -- the means of SACLearner._sync_grads once the Adam kernel scales by 1 / world.  -> (row tensor, 1, scale)"""
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        if world == 1:
            return partials, rows, 0.0
        stride = partials.shape[1]
        raw = self._raw.setdefault(stride, torch.empty(stride, dtype=torch.float32, device=self.device))
        self._check(self.lib.uavenv_sac_reduce(partials.data_ptr(), rows, stride, raw.data_ptr(), self._stream()), "uavenv_sac_reduce")
        if getattr(self, "_p2p", None) is not None:
            rc = self.lib.uavenv_p2p_allreduce(self._p2p, raw.data_ptr(), stride, self._stream())
            if rc == self._lib.EP2P:
                raise RuntimeError("the peer exchange raised its sticky error (timeout): stop stepping, re-synchronise from one rank")
            self._check(rc, "uavenv_p2p_allreduce")
        elif getattr(self, "_coll", None) is not None:
            self._check(self.lib.uavenv_coll_allreduce_sum(self._coll, raw.data_ptr(), stride, self._stream()), "uavenv_coll_allreduce_sum")
        else:
            dist.all_reduce(raw, op=dist.ReduceOp.SUM)
        return raw, 1, 0.0        # (the valid-fraction column, summed over the ranks, normalises: 1 / world when all valid)

    # the four launches, separately (tests drive them one by one)
    def critic_grad(self, batch, eps_next: torch.Tensor):
        C = self._C
        _, pc, _ = self._scratch(batch.batch)
        self._rows_launched = self.lib.uavenv_sac_partial_rows_n(int(batch.batch), 1, int(batch.tiles_per_wg))   # what this launch writes
        batch.eps = eps_next.data_ptr()
        self._check(self.lib.uavenv_sac_critic_grad(C.byref(self._nets), C.byref(batch), self.gamma, self.action_bound,
                                                    pc.data_ptr(), self._stream()), "uavenv_sac_critic_grad")
        return pc

    def critic_step(self, n: int):
        C = self._C
        _, pc, _ = self._scratch(n)
        pc, rows, scale = self._exchange(pc, self._rows_launched)
        h = self._adam(self.critic_lr, self.tau, scale)
        cb = self._cblocks
        self._check(self.lib.uavenv_sac_critic_adam(C.byref(self._nets), pc.data_ptr(), rows, cb[4].data_ptr(), cb[5].data_ptr(),
                                                    cb[6].data_ptr(), cb[7].data_ptr(), C.byref(h), self._scalars.data_ptr(),
                                                    self._stream()), "uavenv_sac_critic_adam")

    def actor_grad(self, batch, eps_cur: torch.Tensor):
        C = self._C
        _, _, pa = self._scratch(batch.batch)
        self._rows_launched = self.lib.uavenv_sac_partial_rows_n(int(batch.batch), 1, int(batch.tiles_per_wg))
        batch.eps = eps_cur.data_ptr()
        self._check(self.lib.uavenv_sac_actor_grad(C.byref(self._nets), C.byref(batch), self.action_bound, pa.data_ptr(),
                                                   self._stream()), "uavenv_sac_actor_grad")
        return pa

    def actor_step(self, n: int):
        C = self._C
        _, _, pa = self._scratch(n)
        pa, rows, scale = self._exchange(pa, self._rows_launched)
        h = self._adam(self.actor_lr, 0.0, scale)
        self._check(self.lib.uavenv_sac_actor_adam(C.byref(self._nets), pa.data_ptr(), rows, int(n), self._blocks[1].data_ptr(),
                                                   self._blocks[2].data_ptr(), self._alpha_mv.data_ptr(), C.byref(h),
                                                   self.alpha_lr, self.target_entropy, self._scalars[4:].data_ptr(),
                                                   self._stream()), "uavenv_sac_actor_adam")

    def learn(self, batch, noise=None):
        """One SAC_Trainer.update on `batch` (make_batch).  noise = (eps_next, eps_cur), [B, 2] f32 each, pins the two
        rsample() draws; default: fresh torch.randn."""
        n = batch.batch
        if noise is None:
            z = torch.randn((2, n, 2), dtype=torch.float32, device=self.device)
            noise = (z[0], z[1])
        e_next, e_cur = (t.contiguous() for t in noise)
        assert e_next.shape == (n, 2) and e_next.dtype == torch.float32
        self.epoch += 1
        self.adam_steps += 1
        self.critic_grad(batch, e_next)
        self.critic_step(n)
        self.actor_grad(batch, e_cur)
        self.actor_step(n)
        batch._noise = (e_next, e_cur)
        return self._scalars[4]

    # -- checkpoints: torch.optim.Adam's state-dict format around the kernels' flat moment blocks (as FusedDQNLearner) ----
    def _opt_slots(self):
        return ((self.actor, self._blocks[0], self._blocks[1], self._blocks[2], self.actor_lr),
                (self.critic_1, self._cblocks[0], self._cblocks[4], self._cblocks[5], self.critic_lr),
                (self.critic_2, self._cblocks[1], self._cblocks[6], self._cblocks[7], self.critic_lr))

    def _opt_view(self, net, pblock, m, v, lr):
        opt = torch.optim.Adam(net.parameters(), lr=lr, betas=(self.beta1, self.beta2), eps=self.adam_eps)
        base = pblock.data_ptr()
        for p in net.parameters():
            off, n = (p.data_ptr() - base) // 4, p.numel()
            opt.state[p] = {"step": torch.tensor(float(self.adam_steps)), "exp_avg": m[off:off + n].view_as(p).clone(),
                            "exp_avg_sq": v[off:off + n].view_as(p).clone()}
        return opt

    def optimizer_state_dicts(self):
        """(actor, critic_1, critic_2) optimizer state dicts in torch.optim.Adam's own format: interchangeable with
        SACLearner / the reference's {'model', 'optimizer', 'epoch'} checkpoints (Trainer/SAC_Trainer.py:98-119)."""
        return tuple(self._opt_view(*slot).state_dict() for slot in self._opt_slots())

    def load_optimizer_state_dicts(self, sds):
        """The inverse; validates all three before touching any moment block."""
        views = []
        for slot, sd in zip(self._opt_slots(), sds):
            opt = self._opt_view(*slot)
            opt.load_state_dict(sd)
            views.append(opt)
        steps = []
        with torch.no_grad():
            for (net, pblock, m, v, _), opt in zip(self._opt_slots(), views):
                base = pblock.data_ptr()
                for p in net.parameters():
                    st = opt.state.get(p)
                    if not st:
                        continue
                    off, n = (p.data_ptr() - base) // 4, p.numel()
                    m[off:off + n].copy_(st["exp_avg"].reshape(-1))
                    v[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
                    steps.append(int(float(st["step"])))
        if steps:
            self.adam_steps = max(steps)

    @property
    def loss(self):
        return self._scalars[4]

    @property
    def critic_losses(self):
        return self._scalars[0:2]

    def act(self, states: torch.Tensor, eps=None) -> torch.Tensor:
        with torch.no_grad():
            return self.actor(states.to(self.device).float(), eps)[0]

    def act_rows(self, obs_packed: torch.Tensor, first_row: int, row_stride: int, count: int, act0: torch.Tensor,
                 act1: torch.Tensor, eps: torch.Tensor = None):
        """get_action for the agents whose packed rows are first_row + i * row_stride of `obs_packed` (a flat view of the
        ring), one launch; the two action components are written to act0[row] / act1[row] (flat planes)."""
        if eps is None:
            eps = torch.randn((count, 2), dtype=torch.float32, device=self.device)
        assert eps.shape == (count, 2) and eps.dtype == torch.float32 and eps.is_contiguous()
        self._check(self.lib.uavenv_sac_act(self._blocks[0].data_ptr(), obs_packed.data_ptr(), int(first_row), int(row_stride),
                                            int(count), eps.data_ptr(), self.action_bound, act0.data_ptr(), act1.data_ptr(),
                                            self._stream()), "uavenv_sac_act")
