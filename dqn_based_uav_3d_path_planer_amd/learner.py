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
        self._allreduce_grads()
        self.optim.step()
        self.loss = loss.detach()
        if self.epoch % self.update_loop == 0:                                         # DQN_Trainer.py:129-130
            self.hard_update()
        return self.loss
