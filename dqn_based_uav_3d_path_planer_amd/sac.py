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

    def learn(self, batch: dict, noise=None, is_weights=None):
        """One SAC_Trainer.update (continuous branch).  batch: states [B,100], actions [B,2], rewards, next_states,
        dones.  noise = (eps_next, eps_cur) optionally pins the two rsample() draws.  is_weights [B] (prioritised
        replay, IsPriority_Replay == 1): the critic losses become mean_i w_i * err_i^2 and self.abs_errors holds
        |min(Q1, Q2) - td_target| for ReplayTree.batch_update (SAC_Trainer.py:346-352; the reference's own
        `is_weights * critic_loss` there is a vector and its backward() raises)."""
        self.epoch += 1
        states, next_states = batch["states"].float(), batch["next_states"].float()
        actions = batch["actions"].float().reshape(len(states), -1)
        rewards, dones = batch["rewards"].float().view(-1, 1), batch["dones"].float().view(-1, 1)
        e_next, e_cur = noise if noise is not None else (None, None)
        td_target = self.calc_target(rewards, next_states, dones, e_next)
        q1, q2 = self.critic_1(states, actions), self.critic_2(states, actions)
        if is_weights is None:
            critic_1_loss = torch.mean(F.mse_loss(q1, td_target.detach()))
            critic_2_loss = torch.mean(F.mse_loss(q2, td_target.detach()))
        else:
            w = is_weights.to(q1.dtype).view(-1, 1)
            critic_1_loss = torch.mean(w * (q1 - td_target.detach()) ** 2)
            critic_2_loss = torch.mean(w * (q2 - td_target.detach()) ** 2)
            self.abs_errors = torch.abs(torch.min(q1, q2) - td_target).detach()[:, 0]      # :351 .squeeze() of [B,2] rows
        for opt, loss, net in ((self.critic_1_optimizer, critic_1_loss, self.critic_1),
                               (self.critic_2_optimizer, critic_2_loss, self.critic_2)):
            opt.zero_grad()
            loss.backward()
            self._sync_grads(list(net.parameters()))
            opt.step()
        new_actions, log_prob = self.actor(states, e_cur)
        entropy = -log_prob
        actor_loss = torch.mean(-self.log_alpha.exp() * entropy -
                                torch.min(self.critic_1(states, new_actions), self.critic_2(states, new_actions)))
        self.actor_optimizer.zero_grad()
        actor_loss.backward()
        self._sync_grads(list(self.actor.parameters()))
        self.actor_optimizer.step()
        alpha_loss = torch.mean((entropy - self.target_entropy).detach() * self.log_alpha.exp())
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
