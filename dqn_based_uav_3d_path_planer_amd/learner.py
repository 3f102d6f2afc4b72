"""DQN-family learner on PyTorch-ROCm (Trainer/DQN_Trainer.py:85-136, DDQN_Trainer.py:72-117,
DuelingDQN_Trainer.py:99-190): TD target -> MSE loss (BaseTrainer.py:40) -> Adam -> hard target copy
every Update_loop updates.  The MLP GEMMs run on MFMA through rocBLAS/hipBLASLt; for N>1 GPUs the flat
gradient bucket is all-reduced (RCCL) before Adam.step, so every rank applies the same update.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist

from .nets import create_network

KINDS = ("dqn", "ddqn", "dueling")   # dueling = VAnet2 + double-DQN target, as DuelingDQN_Trainer does


class DQNLearner:
    def __init__(self, param: dict, kind: str = "dqn", device="cuda:0", lr: Optional[float] = None,
                 gamma: Optional[float] = None, update_loop: Optional[int] = None, amp_dtype=None,
                 loss: str = "mse"):
        assert kind in KINDS
        self.kind = kind
        self.device = torch.device(device)
        self.q_local = create_network(param).to(self.device)
        self.q_target = create_network(param).to(self.device)
        self.lr = float(lr if lr is not None else (param.get("LEARNING_RATE") or 0.001))
        self.gamma = float(gamma if gamma is not None else (param.get("gamma") or 0.99))
        self.update_loop = int(update_loop if update_loop is not None else (param.get("Update_loop") or 3))
        self.optim = torch.optim.Adam(self.q_local.parameters(), lr=self.lr)
        self.epoch = 0
        self.amp_dtype = amp_dtype           # torch.float16 / torch.bfloat16 for BASELINE config 3
        self.loss_kind = loss                # "mse" (reference) or "huber" (north_star option)
        self.loss = torch.zeros((), device=self.device)
        self._flat = None
        # multi-rank exchange: "grad" = all-reduce the gradient bucket before every Adam step (the ranks
        # stay bit-identical); "fedavg" = every rank trains alone and the WEIGHTS are averaged every fl_loop updates
        # (the reference's federated idea, PathPlan_City.py:590-603 / FL_Loop, as one all-reduce(avg)).
        self.sync = "grad"
        self.fl_loop = int(param.get("FL_Loop") or 3)

    # -- reference API -------------------------------------------------------------------------
    def hard_update(self):
        with torch.no_grad():
            for t, p in zip(self.q_target.parameters(), self.q_local.parameters()):
                t.copy_(p)

    def q_values(self, states: torch.Tensor) -> torch.Tensor:
        with torch.no_grad():
            if self.amp_dtype is not None:
                with torch.autocast("cuda", dtype=self.amp_dtype):
                    return self.q_local(states).float()
            return self.q_local(states.float())

    def td_loss(self, states, actions, rewards, next_states, dones, valid=None):
        states, next_states = states.float(), next_states.float()
        actions = actions.long().view(-1, 1)
        rewards, dones = rewards.view(-1, 1), dones.view(-1, 1)
        q_expected = self.q_local(states).gather(1, actions)                          # DQN_Trainer.py:107
        with torch.no_grad():
            if self.kind == "dqn":
                q_next = self.q_target(next_states).max(1)[0].view(-1, 1)             # DQN_Trainer.py:109
            else:
                max_action = self.q_local(next_states).max(1)[1].view(-1, 1)          # DDQN_Trainer.py:94
                q_next = self.q_target(next_states).gather(1, max_action)             # DDQN_Trainer.py:95
            q_targets = rewards + (self.gamma * q_next * (1 - dones))                 # DQN_Trainer.py:114
        if self.loss_kind == "huber":
            per = torch.nn.functional.smooth_l1_loss(q_expected, q_targets, reduction="none")
        else:
            per = (q_expected - q_targets) ** 2                                        # MSELoss, BaseTrainer.py:40
        if valid is None:
            return per.mean()
        v = valid.view(-1, 1)
        return (per * v).sum() / v.sum().clamp_min(1.0)

    def _allreduce_grads(self):
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return
        params = [p for p in self.q_local.parameters() if p.grad is not None]
        flat = torch.cat([p.grad.reshape(-1) for p in params])       # one ~26 KB bucket: latency-bound
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat /= dist.get_world_size()
        off = 0
        for p in params:
            n = p.numel()
            p.grad.copy_(flat[off:off + n].view_as(p))
            off += n

    def learn_weighted(self, batch: dict, is_weights: torch.Tensor):
        """One update with importance-sampling weights (prioritised replay): loss = mean_i w_i * per_i, and the
        |TD error| per sample comes back for ReplayTree.batch_update.  (The reference multiplies the already
        reduced loss by the weight VECTOR and calls backward on the result -- SAC_Trainer.py:349-350, a RuntimeError
        in torch; this is the per-sample weighting that line is reaching for.)"""
        self.epoch += 1
        states, next_states = batch["states"].float(), batch["next_states"].float()
        actions = batch["actions"].long().view(-1, 1)
        rewards, dones = batch["rewards"].view(-1, 1), batch["dones"].view(-1, 1)
        q_expected = self.q_local(states).gather(1, actions)
        with torch.no_grad():
            if self.kind == "dqn":
                q_next = self.q_target(next_states).max(1)[0].view(-1, 1)
            else:
                q_next = self.q_target(next_states).gather(1, self.q_local(next_states).max(1)[1].view(-1, 1))
            q_targets = rewards + (self.gamma * q_next * (1 - dones))
        delta = q_expected - q_targets
        per = torch.nn.functional.smooth_l1_loss(q_expected, q_targets, reduction="none") if self.loss_kind == "huber" \
            else delta ** 2
        loss = (per * is_weights.view(-1, 1).to(per.dtype)).mean()
        self.optim.zero_grad(set_to_none=False)
        loss.backward()
        if self.sync == "grad":
            self._allreduce_grads()
        self.optim.step()
        self.loss = loss.detach()
        if self.epoch % self.update_loop == 0:
            self.hard_update()
        return self.loss, delta.detach().abs().view(-1)

    def federated_average(self):
        """FedAvg over the ranks: q_local and q_target <- mean over ranks -- the rank-level counterpart of the reference's
        merge of its per-UAV trainers (Envs/PathPlan_City.py:590-601 adds the agents' state_dicts up; the division it
        writes at :597 never reaches the model, federated.py -- here the agents are the ranks, the sum is one all-reduce
        and the mean is applied).  The Adam moments stay local, as replace_param leaves each trainer's optimizer alone."""
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return
        with torch.no_grad():
            ps = list(self.q_local.parameters()) + list(self.q_target.parameters())
            flat = torch.cat([p.reshape(-1) for p in ps])
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            flat /= dist.get_world_size()
            off = 0
            for p in ps:
                n = p.numel()
                p.copy_(flat[off:off + n].view_as(p))
                off += n

    def learn(self, batch: dict) -> torch.Tensor:
        """One update on a sampled batch (dict with states/actions/rewards/next_states/dones[/valid])."""
        self.epoch += 1
        if self.amp_dtype is not None:
            with torch.autocast("cuda", dtype=self.amp_dtype):
                loss = self.td_loss(batch["states"], batch["actions"], batch["rewards"], batch["next_states"],
                                    batch["dones"], batch.get("valid"))
        else:
            loss = self.td_loss(batch["states"], batch["actions"], batch["rewards"], batch["next_states"],
                                batch["dones"], batch.get("valid"))
        self.optim.zero_grad(set_to_none=False)
        loss.backward()
        if self.sync == "grad":
            self._allreduce_grads()
        self.optim.step()
        self.loss = loss.detach()
        if self.epoch % self.update_loop == 0:                                         # DQN_Trainer.py:129-130
            self.hard_update()
        if self.sync == "fedavg" and self.epoch % self.fl_loop == 0:
            self.federated_average()
        return self.loss


class FusedDQNLearner:
    """Same update as DQNLearner, run by the hand-written kernels of csrc/learner.hip (k_dqn_act / k_dqn_grad on
    the f32 MFMA, k_dqn_reduce, k_dqn_adam) straight off the device replay ring: 3 launches per update instead of
    ~45.  Requires the reference's shapes (w=100, hiden_dim=64, output <= 15) and batch % 64 == 0.

    q_local / q_target are ordinary nn.Modules whose parameters are VIEWS into the flat blocks the kernels use, so
    state_dict() / load_state_dict() / checkpoints keep working (keys fc1, fc2 | fc_A, fc_V).
    """

    def __init__(self, param: dict, kind: str = "dqn", device="cuda:0", lr: Optional[float] = None,
                 gamma: Optional[float] = None, update_loop: Optional[int] = None, loss: str = "mse",
                 betas=(0.9, 0.999), eps: float = 1e-8, mfma: str = "f32"):
        """mfma = "f16": the matrix products of act / learn run on the f16 MFMA with f32 accumulation (BASELINE
        configs[2]); master weights, Adam and everything else stay f32.  Needs an f16 or packed ring, <= 4 outputs."""
        import ctypes as C
        from . import _lib
        assert kind in KINDS
        self._C, self._lib_mod = C, _lib
        self.lib = _lib.load()
        self.kind = kind
        self.device = torch.device(device)
        self.q_local = create_network(param).to(self.device)
        self.q_target = create_network(param).to(self.device)
        self.dueling = hasattr(self.q_local, "fc_A")
        if (kind == "dueling") != self.dueling:
            raise ValueError("kind 'dueling' goes with NetWork VAnet2 (and only with it)")
        self.n_actions = int(param.get("output"))
        w, hid = int(param.get("w")), int(param.get("hiden_dim"))
        self.lr = float(lr if lr is not None else (param.get("LEARNING_RATE") or 0.001))
        self.gamma = float(gamma if gamma is not None else (param.get("gamma") or 0.99))
        self.update_loop = int(update_loop if update_loop is not None else (param.get("Update_loop") or 3))
        self.betas, self.eps = betas, eps
        self.huber = 1 if loss == "huber" else 0
        self.epoch = 0
        n2 = self.n_actions + (1 if self.dueling else 0)
        self.P = hid * w + hid + n2 * hid + n2
        # local, target, m, v: rows padded to a multiple of 4 floats so that every block starts 16-byte aligned
        self._flat_pad = torch.zeros((4, (self.P + 3) & ~3), dtype=torch.float32, device=self.device)
        self.flat = self._flat_pad[:, :self.P]
        self._bind(self.q_local, self.flat[0], hid, w)
        self._bind(self.q_target, self.flat[1], hid, w)
        self.net = _lib.UavDqnNet(self.flat[0].data_ptr(), self.flat[1].data_ptr(), self.flat[2].data_ptr(),
                                  self.flat[3].data_ptr(), w, hid, self.n_actions, 1 if self.dueling else 0,
                                  _lib.MFMA_F16 if mfma == "f16" else _lib.MFMA_F32, 0)
        self.mfma = mfma
        rc = self.lib.uavenv_dqn_num_params(C.byref(self.net))
        if rc != self.P:
            raise _lib.UavEnvError(f"parameter count mismatch {rc} != {self.P}")
        self.raw = torch.zeros(self.P + 2, dtype=torch.float32, device=self.device)   # grad sums, loss sum, count
        self.loss = torch.zeros((), dtype=torch.float32, device=self.device)
        self._partials = None
        self.force_split = False       # tests: take the multi-GPU (reduce -> all-reduce -> adam) path on one GPU
        self.sync = "grad"             # or "fedavg": no per-update exchange, weights averaged every fl_loop updates
        self.fl_loop = int(param.get("FL_Loop") or 3)

    def _bind(self, net, flat, hid, w):
        """Move the module's parameters into the flat block (kernel layout) and make them views of it."""
        A = self.n_actions
        o_b1 = hid * w
        o_w2 = o_b1 + hid
        if self.dueling:
            n2 = A + 1
            o_b2 = o_w2 + n2 * hid
            slots = [(net.fc1.weight, 0), (net.fc1.bias, o_b1), (net.fc_A.weight, o_w2),
                     (net.fc_V.weight, o_w2 + A * hid), (net.fc_A.bias, o_b2), (net.fc_V.bias, o_b2 + A)]
        else:
            o_b2 = o_w2 + A * hid
            slots = [(net.fc1.weight, 0), (net.fc1.bias, o_b1), (net.fc2.weight, o_w2), (net.fc2.bias, o_b2)]
        with torch.no_grad():
            for p, off in slots:
                n = p.numel()
                flat[off:off + n].copy_(p.detach().reshape(-1))
                p.data = flat[off:off + n].view_as(p)

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def split_image(self) -> torch.Tensor:
        """fc1 / b1 of q_local and q_target in the split form the packed-row kernels stage (csrc/qnet_device.hpp), as the C loop
        keeps it for its own launches (csrc/dqn_internal.hpp): a snapshot of the parameters AS THEY ARE NOW -- for measuring those
        launches outside the loop; it does not follow later updates."""
        _lib = self._lib_mod
        img = torch.empty(2 * _lib.DQN_IMAGE_FLOATS, dtype=torch.float32, device=self.flat.device)
        s = torch.cuda.current_stream(self.flat.device).cuda_stream
        _lib.check(self.lib.uavenv_dqn_split_image(self._C.byref(self.net), img.data_ptr(), s), "uavenv_dqn_split_image")
        return img

    def new_partials(self, batch: int) -> torch.Tensor:
        """Scratch for uavenv_dqn_grad: partial_rows(batch) x partial_stride(net) floats."""
        rows = self.lib.uavenv_dqn_partial_rows(int(batch))
        stride = self.lib.uavenv_dqn_partial_stride(self._C.byref(self.net))
        if rows <= 0 or stride <= 0:
            raise ValueError("fused learner needs batch % 64 == 0")
        return torch.empty((rows, stride), dtype=torch.float32, device=self.device)

    def hard_update(self):
        self.flat[1].copy_(self.flat[0])

    def reset_optimizer(self):
        self.flat[2:].zero_()
        self.epoch = 0

    def federated_average(self):
        """FedAvg over the ranks: local and target blocks <- mean over ranks, one all-reduce of 2 x P floats over
        RCCL (DQNLearner.federated_average has the reference lines)."""
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return
        w = self._flat_pad[:2]             # contiguous (the pad column rides along)
        dist.all_reduce(w, op=dist.ReduceOp.SUM)
        w /= dist.get_world_size()

    # -- checkpoints: torch.optim.Adam's state-dict format around the kernels' flat moment blocks -------------------
    def _adam_view(self):
        """A torch.optim.Adam over q_local's parameters whose state mirrors the flat m / v blocks and the update count
        (never stepped: it exists so that save() / Load_Mod() exchange `optimizer` dicts with the reference's
        {'model', 'optimizer', 'epoch'} checkpoints, Trainer/DuelingDQN_Trainer.py:74-84)."""
        opt = torch.optim.Adam(self.q_local.parameters(), lr=self.lr, betas=self.betas, eps=self.eps)
        base = self.flat[0].data_ptr()
        for p in self.q_local.parameters():
            off = (p.data_ptr() - base) // 4
            n = p.numel()
            opt.state[p] = {"step": torch.tensor(float(self.epoch)),
                            "exp_avg": self.flat[2][off:off + n].view_as(p).clone(),
                            "exp_avg_sq": self.flat[3][off:off + n].view_as(p).clone()}
        return opt

    def optimizer_state_dict(self) -> dict:
        return self._adam_view().state_dict()

    def load_optimizer_state_dict(self, sd: dict):
        opt = self._adam_view()
        opt.load_state_dict(sd)
        base = self.flat[0].data_ptr()
        with torch.no_grad():
            for p in self.q_local.parameters():
                st = opt.state.get(p)
                if not st:
                    continue
                off = (p.data_ptr() - base) // 4
                n = p.numel()
                self.flat[2][off:off + n].copy_(st["exp_avg"].reshape(-1))
                self.flat[3][off:off + n].copy_(st["exp_avg_sq"].reshape(-1))

    class _OptimFacade:
        def __init__(self, owner):
            self._o = owner

        def state_dict(self):
            return self._o.optimizer_state_dict()

        def load_state_dict(self, sd):
            self._o.load_optimizer_state_dict(sd)

    @property
    def optim(self):
        return FusedDQNLearner._OptimFacade(self)

    def q_values(self, states: torch.Tensor) -> torch.Tensor:
        with torch.no_grad():
            return self.q_local(states.float())

    def learn(self, batch: dict) -> torch.Tensor:
        """One update on an arbitrary pre-sampled batch (the reference's update(transition_dict) path): the batch becomes
        a two-frame ring (frame 0 = states, frame 1 = next states), padded with invalid rows to a multiple of 64."""
        C, _lib = self._C, self._lib_mod
        st = batch["states"].to(self.device, torch.float32).reshape(-1, _lib.OBS_DIM)
        b = st.shape[0]
        bp = (b + 63) // 64 * 64
        if not hasattr(self, "_adhoc"):
            self._adhoc = {}
        buf = self._adhoc.get(bp)
        if buf is None:
            d = self.device
            # (the f16-MFMA kernels take f16 / packed rings: an arbitrary batch is rounded to f16 rows for them)
            odt, ocode = (torch.float16, _lib.OBS_F16) if self.mfma == "f16" else (torch.float32, _lib.OBS_F32)
            buf = dict(obs=torch.zeros((2, bp, _lib.OBS_DIM), dtype=odt, device=d),
                       action=torch.zeros((2, bp), dtype=torch.int32, device=d),
                       reward=torch.zeros((2, bp), dtype=torch.float32, device=d),
                       done=torch.zeros((2, bp), dtype=torch.uint8, device=d),
                       valid=torch.zeros((2, bp), dtype=torch.uint8, device=d),
                       idx=torch.stack([torch.zeros(bp, dtype=torch.int32), torch.arange(bp, dtype=torch.int32)], 1)
                       .contiguous().to(d))
            buf["c"] = _lib.UavReplayRing(buf["obs"].data_ptr(), buf["action"].data_ptr(), buf["reward"].data_ptr(),
                                          buf["done"].data_ptr(), buf["valid"].data_ptr(), 2, bp, ocode, 1)
            self._adhoc[bp] = buf
        buf["obs"][0, :b].copy_(st)
        buf["obs"][1, :b].copy_(batch["next_states"].to(self.device, torch.float32).reshape(-1, _lib.OBS_DIM))
        buf["action"][0, :b].copy_(batch["actions"].reshape(-1).to(self.device, torch.int32))
        buf["reward"][0, :b].copy_(batch["rewards"].reshape(-1).to(self.device, torch.float32))
        buf["done"][0, :b].copy_((batch["dones"].reshape(-1) != 0).to(self.device, torch.uint8))
        v = batch.get("valid")
        buf["valid"][0].zero_()
        if v is None:
            buf["valid"][0, :b] = 1
        else:
            buf["valid"][0, :b].copy_((v.reshape(-1) != 0).to(self.device, torch.uint8))

        class _R:            # what learn_from_ring needs of a ring
            pass
        r = _R()
        r._c, r.head, r.filled = buf["c"], 1, 1
        w = batch.get("is_weights")
        if w is None:
            return self.learn_from_ring(r, bp, 0, 0, explicit_idx=buf["idx"])
        if "isw" not in buf:
            buf["isw"] = torch.zeros(bp, dtype=torch.float32, device=self.device)
            buf["abs"] = torch.zeros(bp, dtype=torch.float32, device=self.device)
        buf["isw"].zero_()
        buf["isw"][:b].copy_(w.reshape(-1).to(self.device, torch.float32))
        loss = self.learn_from_ring(r, bp, 0, 0, explicit_idx=buf["idx"], is_weights=buf["isw"], abs_td_out=buf["abs"])
        self.abs_errors = buf["abs"][:b]
        return loss

    def learn_weighted(self, batch: dict, is_weights: torch.Tensor):
        """DQNLearner.learn_weighted on the fused kernels (uavenv_dqn_grad_w): loss = mean_i w_i * per_i over the batch
        rows, returns (loss, |TD error| per row)."""
        b = dict(batch)
        b["is_weights"] = is_weights
        b.pop("valid", None)
        loss = self.learn(b)
        return loss, self.abs_errors.clone()

    # -- multi-GPU: the gradient bucket summed over peer-mapped HBM instead of a collective call -------------------
    def enable_p2p(self, verify: bool = True, check_every: int = 256, spin_limit: int = 0) -> bool:
        """Set up csrc/p2p.hip between the ranks of the initialised process group (one rank per GPU of ONE node): every
        rank maps every other rank's receive area through HIP IPC.  With verify, one all-reduce of a known vector is
        compared with torch.distributed's result on every rank; on any failure (IPC not available, timeout, mismatch)
        the learner keeps the RCCL path.  Returns whether the peer-to-peer path is active on ALL ranks.
        check_every: the ranks compare a checksum of their weights on the device every that many updates (0 = never);
        spin_limit: polls before a flag wait gives up (0 = the library's default, about a second)."""
        C, _lib = self._C, self._lib_mod
        from . import exchange as ex
        self._p2p = None
        h = ex.open_p2p(self.lib, self.device, self.P + 2, check_every=check_every, spin_limit=spin_limit)
        if h is None:
            return False
        world, rank = dist.get_world_size(), dist.get_rank()
        ok, flags = True, [None] * world
        self.p2p_selftest_ms = None
        if ok and verify:
            # Round 6: the exchange has never crossed a link on this build's boxes, so it is not accepted on one
            # tidy vector.  FOUR exchanges back to back -- both receive slots, twice each -- of a multi-KB payload that is random per
            # rank and per iteration, each compared on every rank with torch.distributed's all-reduce of the same payload.
            import time as _time
            n = self.lib.uavenv_dqn_partial_rows(64)
            stride = self.lib.uavenv_dqn_partial_stride(C.byref(self.net))
            part = torch.zeros((n, stride), dtype=torch.float32, device=self.device)
            got = torch.zeros(self.P + 2, dtype=torch.float32, device=self.device)
            s = self._stream()
            # every rank's payload comes from a seed every rank knows, so the expected sum needs no second transport: the payloads of
            # all ranks are regenerated here and added in rank order, as the pull kernel adds the slots.  (A torch.distributed
            # all-reduce per iteration was the first form: with the gloo backend on device tensors, four of them per rank left an
            # 8-rank same-device run 100 x slower -- 22 ms per pass -- for the rest of the process's life.)
            gens = [torch.Generator(device="cpu").manual_seed(0xE7C4 + 7919 * r) for r in range(world)]
            torch.cuda.synchronize(self.device)
            t0 = _time.perf_counter()
            for it in range(4):
                payloads = [torch.rand(self.P + 2, generator=g_) * 2.0 - 1.0 for g_ in gens]
                want = payloads[0].clone()
                for r in range(1, world):
                    want += payloads[r]
                want = want.to(self.device)
                part[0, :self.P + 2] = payloads[rank].to(self.device)
                rc1 = self.lib.uavenv_dqn_reduce_p2p(C.byref(self.net), part.data_ptr(), n, h, s)
                rc2 = self.lib.uavenv_dqn_adam_p2p(C.byref(self.net), h, 0.0, 0.9, 0.999, 1e-8, 0, 0, None, got.data_ptr(), s)
                torch.cuda.synchronize(self.device)
                err = C.c_int32(0)
                self.lib.uavenv_p2p_errors(h, C.byref(err))
                ok = ok and rc1 == 0 and rc2 == 0 and err.value == 0 and bool(torch.allclose(got, want, rtol=1e-6, atol=1e-6))
            self.p2p_selftest_ms = (_time.perf_counter() - t0) * 1e3
            dist.all_gather_object(flags, ok)
            ok = all(flags)
        if not ok:
            if h.value:
                self.lib.uavenv_p2p_destroy(h)
            return False
        self._p2p = h
        return True

    def p2p_timeouts(self) -> int:
        if getattr(self, "_p2p", None) is None:
            return 0
        err = self._C.c_int32(0)
        self.lib.uavenv_p2p_errors(self._p2p, self._C.byref(err))
        return int(err.value)

    def p2p_status(self, synchronise: bool = True) -> dict:
        """{'code': sticky error (0 healthy, 1 timeout, 2 ranks diverged), 'timeouts', 'mismatches', 'checks'} of the
        peer exchange; synchronise=False reads only the host-mapped code (no device round trip)."""
        if getattr(self, "_p2p", None) is None:
            return {"code": 0, "timeouts": 0, "mismatches": 0, "checks": 0}
        out = (self._C.c_int32 * 4)()
        self._lib_mod.check(self.lib.uavenv_p2p_status(self._p2p, 1 if synchronise else 0, out), "uavenv_p2p_status")
        return {"code": int(out[0]), "timeouts": int(out[1]), "mismatches": int(out[2]), "checks": int(out[3])}

    def disable_p2p(self):
        if getattr(self, "_p2p", None) is not None:
            self.lib.uavenv_p2p_destroy(self._p2p)
            self._p2p = None

    def enable_coll(self) -> bool:
        """The RCCL fallback of the peer exchange, enqueued from C (csrc/coll.hip): rank 0 draws the communicator id,
        it travels over the process group, every rank joins.  With it the C loop (csrc/loop.hip) keeps driving the
        pass at N > 1 when csrc/p2p.hip is unavailable.  Returns whether the communicator is up on ALL ranks."""
        from . import exchange as ex
        self._coll = None
        h = ex.open_coll(self.lib, self.device, self.P + 2, self._stream())
        if h is None:
            return False
        self._coll = h
        return True

    def broadcast_weights(self, src: int = 0):
        """All four flat blocks (q_local, q_target, Adam's moments) of rank `src` to every rank: the recovery step after
        the peer exchange raised its sticky error (the ranks may have diverged by then)."""
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.broadcast(self._flat_pad, src=src)

    def weights_checksum(self) -> int:
        """64-bit sum of the weight blocks' bit patterns (q_local + q_target): equal on every rank iff they are bit-identical."""
        w = self._flat_pad[:2].contiguous().view(torch.int32).to(torch.int64)
        idx = torch.arange(1, w.numel() + 1, device=w.device, dtype=torch.int64)
        return int(((w.view(-1) & 0xFFFFFFFF) * idx).sum().item())

    def act(self, obs: torch.Tensor, eps: float, seed: int, counter: int, index_out: torch.Tensor = None,
            steer_out: torch.Tensor = None, q_out: torch.Tensor = None):
        """Q(s) + epsilon-greedy for all rows of obs [n,100] in one launch."""
        C, _lib = self._C, self._lib_mod
        dt = _lib.OBS_PACKED if obs.dtype == torch.int32 else (_lib.OBS_F16 if obs.dtype == torch.float16 else _lib.OBS_F32)
        rc = self.lib.uavenv_dqn_act(C.byref(self.net), obs.data_ptr(), dt, obs.shape[0], float(eps), int(seed),
                                     int(counter), None if index_out is None else index_out.data_ptr(),
                                     None if steer_out is None else steer_out.data_ptr(),
                                     None if q_out is None else q_out.data_ptr(), self._stream())
        _lib.check(rc, "uavenv_dqn_act")

    def learn_from_ring(self, ring, batch: int, seed: int, counter: int, explicit_idx: torch.Tensor = None,
                        is_weights: torch.Tensor = None, abs_td_out: torch.Tensor = None):
        """One learn_off_policy() on `batch` transitions drawn from the device ring (same draws as ring.sample).
        is_weights / abs_td_out (f32 [batch], prioritised replay): importance-sampling weights of the samples in the loss and
        the per-sample |TD error| coming back (uavenv_dqn_grad_w)."""
        C, _lib = self._C, self._lib_mod
        if batch % 64:
            raise ValueError("fused learner needs batch % 64 == 0")
        nblk = self.lib.uavenv_dqn_partial_rows(batch)
        if self._partials is None or self._partials.shape[0] != nblk:
            self._partials = self.new_partials(batch)
        self.epoch += 1
        s = self._stream()
        kind = 0 if self.kind == "dqn" else 1
        rc = self.lib.uavenv_dqn_grad_w(C.byref(ring._c), ring.head, ring.filled, batch, int(seed), int(counter),
                                        None if explicit_idx is None else explicit_idx.data_ptr(), C.byref(self.net),
                                        kind, self.gamma, self.huber,
                                        None if is_weights is None else is_weights.data_ptr(),
                                        None if abs_td_out is None else abs_td_out.data_ptr(), self._partials.data_ptr(), s)
        _lib.check(rc, "uavenv_dqn_grad_w")
        hard = 1 if self.epoch % self.update_loop == 0 else 0
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        fed = multi and self.sync == "fedavg"
        if (not multi or fed) and not self.force_split:      # reduce + Adam (+ hard target copy) in one launch
            rc = self.lib.uavenv_dqn_reduce_adam(C.byref(self.net), self._partials.data_ptr(), nblk, self.lr,
                                                 self.betas[0], self.betas[1], self.eps, self.epoch, hard,
                                                 self.loss.data_ptr(), None, s)
            _lib.check(rc, "uavenv_dqn_reduce_adam")
            if fed and self.epoch % self.fl_loop == 0:
                self.federated_average()
            return self.loss
        if multi and getattr(self, "_p2p", None) is not None:     # peer-to-peer sum over xGMI, on the stream
            _lib.check(self.lib.uavenv_dqn_reduce_p2p(C.byref(self.net), self._partials.data_ptr(), nblk, self._p2p, s),
                       "uavenv_dqn_reduce_p2p")
            _lib.check(self.lib.uavenv_dqn_adam_p2p(C.byref(self.net), self._p2p, self.lr, self.betas[0], self.betas[1],
                                                    self.eps, self.epoch, hard, self.loss.data_ptr(), self.raw.data_ptr(), s),
                       "uavenv_dqn_adam_p2p")
            return self.loss
        rc = self.lib.uavenv_dqn_reduce(C.byref(self.net), self._partials.data_ptr(), nblk, self.raw.data_ptr(), s)
        _lib.check(rc, "uavenv_dqn_reduce")
        # one ~26 KB bucket over RCCL / xGMI: gradient SUMS + loss sum + valid count, so the update is the mean over
        # the valid samples of all ranks (== single-GPU on the concatenated batch)
        if multi and getattr(self, "_coll", None) is not None:    # RCCL enqueued from C on the same stream (csrc/coll.hip)
            _lib.check(self.lib.uavenv_coll_allreduce_sum(self._coll, self.raw.data_ptr(), self.P + 2, s),
                       "uavenv_coll_allreduce_sum")
        elif multi:
            dist.all_reduce(self.raw, op=dist.ReduceOp.SUM)
        rc = self.lib.uavenv_dqn_adam(C.byref(self.net), self.raw.data_ptr(), self.lr, self.betas[0], self.betas[1],
                                      self.eps, self.epoch, hard, self.loss.data_ptr(), s)
        _lib.check(rc, "uavenv_dqn_adam")
        return self.loss
