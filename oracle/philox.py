"""TEST INFRASTRUCTURE (oracle): independent numpy restatement of the counter-based random streams the device uses.

* philox4x32_10: Salmon, Moraes, Dror, Shaw, "Parallel Random Numbers: As Easy as 1, 2, 3" (SC'11), the
  Philox-4x32 bijection with 10 rounds, multipliers 0xD2511F53 / 0xCD9E8D57, Weyl key increments 0x9E3779B9 /
  0xBB67AE85.  Pinned against the Random123 known-answer vectors in tests/test_oracle_philox.py.
* replay_draws: which stored transitions `uavenv_replay_sample` / `uavenv_dqn_grad` must pick for update
  (seed, counter): ReplayMemory.sample2 = random.sample(memory, batch) (BaseClass/replay_buffer.py:48-51) draws
  DISTINCT transitions; the device realises that as the first `batch` images of a keyed pseudo-random permutation
  of the D = filled * n_agents stored transitions (6-round alternating Feistel network over ceil(log2 D) bits,
  cycle-walked into [0, D); csrc/uavenv_device.hpp: replay_perm / replay_perm_apply).
* act_draws: the epsilon-greedy stream of uavenv_dqn_act / uavenv_select_actions (DuelingDQN_Trainer.py:86-97).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this module.
"""
from __future__ import annotations

import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
U32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(ctr, key):
    """ctr: (..., 4) uint32-valued, key: (2,) or (..., 2) -> (..., 4) uint32.  Vectorised over the leading axes."""
    c = np.asarray(ctr, dtype=np.uint64) & U32
    k = np.broadcast_to(np.asarray(key, dtype=np.uint64) & U32, c.shape[:-1] + (2,)).copy()
    c0, c1, c2, c3 = (c[..., i].copy() for i in range(4))
    k0, k1 = k[..., 0], k[..., 1]
    for _ in range(10):
        p0 = M0 * c0                      # 32 x 32 -> 64 bit products (fit in uint64)
        p1 = M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & U32
        hi1, lo1 = p1 >> np.uint64(32), p1 & U32
        c0, c1, c2, c3 = (hi1 ^ c1 ^ k0) & U32, lo1, (hi0 ^ c3 ^ k1) & U32, lo0
        k0 = (k0 + np.uint64(W0)) & U32
        k1 = (k1 + np.uint64(W1)) & U32
    return np.stack([c0, c1, c2, c3], axis=-1).astype(np.uint32)


def _key(seed: int):
    return np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=np.uint64)


def fmix32(h):
    """MurmurHash3's 32-bit finaliser (Appleby, public domain) on uint64-held 32-bit values."""
    h = np.asarray(h, dtype=np.uint64) & U32
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x85EBCA6B)) & U32
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(0xC2B2AE35)) & U32
    h ^= h >> np.uint64(16)
    return h


def replay_slots(batch: int, seed: int, counter: int, filled: int, n_agents: int) -> np.ndarray:
    """Transition slot (0 = newest frame's agent 0 ... D-1) of samples 0..batch-1."""
    D = filled * n_agents
    if D <= 1:
        return np.zeros(batch, dtype=np.int64)
    lo_c, hi_c = counter & 0xFFFFFFFF, (counter >> 32) & 0xFFFFFFFF
    a = philox4x32_10(np.array([0, lo_c, hi_c, 0x5A3B]), _key(seed))
    b = philox4x32_10(np.array([1, lo_c, hi_c, 0x5A3B]), _key(seed))
    keys = [np.uint64(v) for v in (a[0], a[1], a[2], a[3], b[0], b[1])]
    bits = max(2, int(D - 1).bit_length())
    la = bits // 2
    lb = bits - la
    ma, mb = np.uint64((1 << la) - 1), np.uint64((1 << lb) - 1)
    sa, sb = np.uint64(32 - la), np.uint64(32 - lb)
    x = (np.arange(batch, dtype=np.uint64) % np.uint64(D))
    todo = np.ones(batch, dtype=bool)
    while todo.any():
        v = x[todo]
        lo, hi = v & ma, v >> np.uint64(la)
        for r in range(0, 6, 2):
            lo = lo ^ (fmix32(hi ^ keys[r]) >> sa)             # even round: low part ^= F(high part)
            hi = hi ^ (fmix32(lo ^ keys[r + 1]) >> sb)         # odd round: high part ^= F(low part)
        v = (hi << np.uint64(la)) | lo
        x[todo] = v
        todo[todo] = v >= np.uint64(D)
    return x.astype(np.int64)


def replay_draws(batch: int, seed: int, counter: int, head: int, filled: int, frames: int, n_agents: int):
    """-> (frame [batch], agent [batch]) of the transitions update (seed, counter) must use."""
    slot = replay_slots(batch, seed, counter, filled, n_agents)
    back = slot // n_agents
    agent = slot - back * n_agents
    f = (head - 1 - back) % frames
    return f.astype(np.int64), agent.astype(np.int64)


def replay_draws_valid(batch: int, n_slots: int, uav_per_env: int, first_slot: int, valid: np.ndarray, max_tries: int, seed: int,
                       counter: int, head: int, filled: int, frames: int, n_envs: int):
    """uavenv_replay_draw_valid: -> (frame [n_slots * batch], env [n_slots * batch], found [n_slots * batch]).  valid: the ring's
    plane [frames][n_envs * uav_per_env].  Draw s of slot first_slot + s // batch walks permutation positions s, s + S, s + 2 S,
    ... (S = n_slots * batch, positions < filled * n_envs, at most max_tries) to the first row of ITS slot with valid != 0; a draw
    that finds none keeps its first row."""
    S, D = batch * n_slots, filled * n_envs
    n_pos = min(D, S * max_tries)
    f_all, e_all = replay_draws(n_pos, seed, counter, head, filled, frames, n_envs)
    valid = np.asarray(valid).reshape(frames, n_envs, uav_per_env)
    f_out, e_out, found = f_all[:S].copy(), e_all[:S].copy(), np.zeros(S, dtype=bool)
    slot = first_slot + np.arange(S) // batch
    for t in range(max_tries):
        q = np.arange(S) + t * S
        ok = (q < D) & ~found
        if not ok.any():
            break
        qq = q[ok]
        hit = valid[f_all[qq], e_all[qq], slot[ok]] != 0
        idx = np.nonzero(ok)[0][hit]
        f_out[idx], e_out[idx], found[idx] = f_all[qq[hit]], e_all[qq[hit]], True
    return f_out, e_out, found


def act_draws(n: int, seed: int, counter: int, n_actions: int):
    """-> (u [n] float32 in [0,1), random_action [n]) of the epsilon-greedy stream: greedy iff u > eps."""
    lo, hi = counter & 0xFFFFFFFF, (counter >> 32) & 0xFFFFFFFF
    ctr = np.stack([np.arange(n, dtype=np.uint64), np.full(n, lo, np.uint64), np.full(n, hi, np.uint64),
                    np.full(n, 0xAC7, np.uint64)], axis=-1)
    r = philox4x32_10(ctr, _key(seed)).astype(np.uint64)
    u = ((r[:, 0] >> np.uint64(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)).astype(np.float32)
    rnd = ((r[:, 1] * np.uint64(n_actions)) >> np.uint64(32)).astype(np.int64)
    return u, rnd
