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
