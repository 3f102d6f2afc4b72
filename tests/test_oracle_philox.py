"""oracle/philox.py pinned: Random123's published known-answer vectors for philox4x32_10 (kat_vectors, Salmon et al.),
and the structural properties the replay draw relies on (a permutation: distinct draws, full coverage)."""
import numpy as np

from oracle import philox as px


def test_philox4x32_10_known_answers():
    kat = [
        ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
        ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
        ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
         (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
    ]
    for ctr, key, want in kat:
        got = px.philox4x32_10(np.array(ctr, dtype=np.uint64), np.array(key, dtype=np.uint64))
        assert tuple(int(v) for v in got) == want


def test_replay_slots_are_a_permutation():
    for filled, n in ((1, 64), (3, 100), (7, 1000), (63, 1024), (5, 1), (2, 3)):
        D = filled * n
        s = px.replay_slots(D, seed=9, counter=4, filled=filled, n_agents=n)
        assert sorted(s.tolist()) == list(range(D))                  # all D draws distinct = random.sample semantics
        s2 = px.replay_slots(D, seed=9, counter=5, filled=filled, n_agents=n)
        if D > 8:
            assert (s != s2).mean() > 0.5                            # another update, another permutation
    # batch > D wraps around (the reference raises there)
    s = px.replay_slots(10, seed=1, counter=0, filled=1, n_agents=4)
    assert (s[:4] == s[4:8]).all() and len(set(s[:4].tolist())) == 4


def test_replay_draws_map_slots_to_frames_behind_head():
    f, a = px.replay_draws(5000, seed=3, counter=1, head=2, filled=5, frames=8, n_agents=1000)
    assert set(np.unique(f).tolist()) == {1, 0, 7, 6, 5}             # 1..5 frames behind head 2, ring of 8
    assert a.min() >= 0 and a.max() < 1000
    # uniform over frames: 5000 of 5000 slots -> exactly 1000 per frame
    assert all((f == k).sum() == 1000 for k in (1, 0, 7, 6, 5))


def test_act_draws_range():
    u, r = px.act_draws(4096, seed=7, counter=3, n_actions=3)
    assert u.min() >= 0.0 and u.max() < 1.0 and set(np.unique(r).tolist()) == {0, 1, 2}
    assert abs(u.mean() - 0.5) < 0.03


def test_valid_only_draws_reject_over_the_permutation():
    """replay_draws_valid (the restatement uavenv_replay_draw_valid is tested against on the GPU): draws walk the permutation past
    rows whose valid flag is 0 -- accepted rows are valid rows of the draw's own slot, distinct within a slot, inside the stored
    frames; with every row valid they are replay_draws'; a draw that runs out of tries keeps its first row."""
    frames, n_envs, U, head, filled, batch = 17, 512, 4, 5, 16, 256
    rng = np.random.default_rng(1)
    valid = (rng.random((frames, n_envs * U)) >= 0.4).astype(np.uint8)
    f, e, found = px.replay_draws_valid(batch, U, U, 0, valid, 8, seed=5, counter=2, head=head, filled=filled, frames=frames, n_envs=n_envs)
    slot = np.arange(U * batch) // batch
    assert (valid.reshape(frames, n_envs, U)[f, e, slot][found] == 1).all()
    assert found.mean() > 0.995                                        # 0.4 ** 8 = 7e-4
    f0, e0 = px.replay_draws(U * batch, 5, 2, head, filled, frames, n_envs)
    assert (f[~found] == f0[~found]).all() and (e[~found] == e0[~found]).all()
    back = (head - 1 - f) % frames
    assert back.max() < filled
    for j in range(U):
        m = (slot == j) & found
        assert len(np.unique(back[m] * n_envs + e[m])) == m.sum()
    fa, ea, fo = px.replay_draws_valid(batch, U, U, 0, np.ones_like(valid), 8, 5, 2, head, filled, frames, n_envs)
    assert fo.all() and (fa == f0).all() and (ea == e0).all()
    # one slot alone, and a ring too small for a second try: positions past filled * n_envs are never looked at
    f1, e1, fo1 = px.replay_draws_valid(batch, 1, U, 3, valid, 8, 5, 2, head, filled, frames, n_envs)
    assert (valid.reshape(frames, n_envs, U)[f1, e1, 3][fo1] == 1).all()
    f2, e2, fo2 = px.replay_draws_valid(filled * n_envs, 1, U, 0, valid, 8, 5, 2, head, filled, frames, n_envs)
    fs, es = px.replay_draws(filled * n_envs, 5, 2, head, filled, frames, n_envs)
    assert (f2 == fs).all() and (e2 == es).all()                       # no second position exists: every draw keeps its first row
