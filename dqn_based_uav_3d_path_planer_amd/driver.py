"""Host-side driver pieces mirroring simulator.py (reference root): the epsilon schedule and, below, a
`simulator`-compatible training loop over the vectorised env (see plugins/ for the reflection surface)."""
from __future__ import annotations


def epsilon_annealing(epoch: int, min_eps: float, max_eps_episode: float) -> float:
    """simulator.py:141-145 -- linear decay from 1.0 to min_eps over max_eps_episode episodes."""
    slope = (min_eps - 1.0) / (max_eps_episode + 0.1)
    return max(slope * epoch + 1.0, min_eps)
