"""Q-networks with the reference's module/parameter names so its checkpoints interchange
(BaseClass/BaseCNN.py:93-102 Qnet2, :120-139 VAnet2).  `param` is the XML dict (string leaves)."""
from __future__ import annotations

import torch
import torch.nn.functional as F


class Qnet2(torch.nn.Module):
    """fc2(relu(fc1(x)))  -- BaseCNN.py:93-102"""

    def __init__(self, param):
        super().__init__()
        self.fc1 = torch.nn.Linear(int(param.get("w")), int(param.get("hiden_dim")))
        self.fc2 = torch.nn.Linear(int(param.get("hiden_dim")), int(param.get("output")))

    def forward(self, x):
        return self.fc2(F.relu(self.fc1(x)))


class VAnet2(torch.nn.Module):
    """Dueling head: Q = V + A - mean(A)  -- BaseCNN.py:120-139 (incl. its 1-D input special case)."""

    def __init__(self, param):
        super().__init__()
        w, hid, out = int(param.get("w")), int(param.get("hiden_dim")), int(param.get("output"))
        self.fc1 = torch.nn.Linear(w, hid)
        self.fc_A = torch.nn.Linear(hid, out)
        self.fc_V = torch.nn.Linear(hid, 1)

    def forward(self, x):
        h = F.relu(self.fc1(x))
        A = self.fc_A(h)
        V = self.fc_V(h)
        if V.shape == torch.Size([1]):
            return (V + A - A.mean(0).view(-1, 1)).view(-1, 1)
        return V + A - A.mean(1).view(-1, 1)


NETWORKS = {"Qnet2": Qnet2, "VAnet2": VAnet2}


def create_network(param):
    """FactoryClass/NetworkFactory.py:10-22 -- look the class up by its `NetWork` name."""
    return NETWORKS[param.get("NetWork")](param)


class _SplitKLinearFn(torch.autograd.Function):
    """y = x W^T + b with the weight gradient dW = dY^T X computed as a batched product over row chunks + a sum.
    For the replay-sized batches of the vectorised trainers (B = 32 768 rows, 64 x 102 outputs) the plain GEMM is all
    reduction dimension: hipBLASLt runs it on a handful of workgroups (measured 95-132 us per call on MI355X, 38 % of a
    SAC update); 64 chunks give the library a batch dimension to spread over the chip."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        return F.linear(x, w, b)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        gx = gy @ w if ctx.needs_input_grad[0] else None
        gw = gb = None
        if ctx.needs_input_grad[1]:          # (the critics are only passed through in the actor loss: no dW needed there)
            n = x.shape[0]
            c = 512
            if n >= 8 * c and n % c == 0:    # reshape, not view: an upstream gradient may arrive transposed / strided
                gw = torch.bmm(gy.reshape(n // c, c, -1).transpose(1, 2), x.reshape(n // c, c, -1)).sum(0)
            else:
                gw = gy.t() @ x
        if ctx.needs_input_grad[2]:
            gb = gy.sum(0)
        return gx, gw, gb


class SplitKLinear(torch.nn.Linear):
    """nn.Linear (same parameter names, same forward values) whose backward is _SplitKLinearFn's."""

    def forward(self, x):
        if x.dim() == 2 and x.shape[0] >= 4096 and torch.is_grad_enabled():
            return _SplitKLinearFn.apply(x, self.weight, self.bias)
        return F.linear(x, self.weight, self.bias)


class PolicyNetContinuous_SAC(torch.nn.Module):
    """BaseCNN.py:459-483 incl. its quirks: std = tanh(softplus(.)), and the log-prob correction applies tanh to the
    already squashed action (`1 - tanh(action)^2`).  `eps` (optional) replaces Normal.rsample()'s N(0,1) draw so the
    same update can be replayed on another device."""

    def __init__(self, param):
        super().__init__()
        w, hid, out = int(param.get("w")), int(param.get("hiden_dim")), int(param.get("output"))
        self.fc1 = SplitKLinear(w, hid)
        self.fc_mu = SplitKLinear(hid, out)
        self.fc_std = SplitKLinear(hid, out)
        self.action_bound = float(param.get("action_bound"))

    def forward(self, x, eps=None):
        x = F.relu(self.fc1(x))
        mu = torch.tanh(self.fc_mu(x))
        std = torch.tanh(F.softplus(self.fc_std(x)))
        dist = torch.distributions.Normal(mu, std)
        normal_sample = dist.rsample() if eps is None else mu + std * eps
        log_prob = dist.log_prob(normal_sample)
        action = torch.tanh(normal_sample)
        log_prob = log_prob - torch.log(1 - torch.tanh(action).pow(2) + 1e-7)
        return action * self.action_bound, log_prob


class QValueNetContinuous_SAC(torch.nn.Module):
    """BaseCNN.py:486-500: Q(s, a) MLP whose output width is action_dim (2 values, as in the reference)."""

    def __init__(self, param):
        super().__init__()
        w, hid, ad = int(param.get("w")), int(param.get("hiden_dim")), int(param.get("action_dim"))
        self.fc1 = SplitKLinear(w + ad, hid)
        self.fc2 = SplitKLinear(hid, hid)
        self.fc_out = SplitKLinear(hid, ad)

    def forward(self, x, a):
        x = F.relu(self.fc1(torch.cat([x, a], dim=-1)))
        x = F.relu(self.fc2(x))
        return self.fc_out(x)


NETWORKS.update(PolicyNetContinuous_SAC=PolicyNetContinuous_SAC, QValueNetContinuous_SAC=QValueNetContinuous_SAC)
