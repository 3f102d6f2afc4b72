"""uavenv_replay_sample / uavenv_replay_draw / the fused learner's in-kernel draw against an INDEPENDENT prediction:
oracle/philox.py (Philox4x32-10 pinned to the Random123 vectors + the Feistel permutation restated in numpy) says
which (frame, agent) every sample must be; the returned rows must equal the ring's contents there.
ReplayMemory.sample2 = random.sample (BaseClass/replay_buffer.py:48-51): distinct transitions."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _filled_ring(n, frames_cap, steps, dtype, uav=1):
    from dqn_based_uav_3d_path_planer_amd.data import make_city26_env
    from dqn_based_uav_3d_path_planer_amd.replay import DeviceReplayRing
    env = make_city26_env(n, obs_dtype=dtype, uav_per_env=uav)
    ring = DeviceReplayRing(env, frames_cap * env.N, discrete=True)
    ring.reset(seed=5)
    gen = torch.Generator(device="cuda").manual_seed(2)
    for _ in range(steps):
        ring.current_action().copy_(torch.randint(0, 3, (env.N,), generator=gen, device="cuda", dtype=torch.int32))
        ring.step_env(auto_reset=True)
    return env, ring


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("steps", [4, 11])            # 11 > frames: the ring has wrapped, head is mid-ring
def test_sample_returns_the_ring_rows_the_oracle_predicts(dtype, steps):
    from oracle import philox as px
    env, ring = _filled_ring(600, 6, steps, dtype)
    n = env.N
    assert ring.frames == 7 and ring.filled == min(steps, 6)
    for batch, counter in ((1000, 0), (ring.filled * n, 3), (64, 2 ** 33 + 5)):
        b = ring.sample(batch, seed=0xABCDEF0123, counter=counter)
        torch.cuda.synchronize()
        f, a = px.replay_draws(batch, 0xABCDEF0123, counter, ring.head, ring.filled, ring.frames, n)
        ft, at = torch.tensor(f, device="cuda"), torch.tensor(a, device="cuda")
        fn = (ft + 1) % ring.frames
        assert torch.equal(b["states"], ring.obs[ft, at])
        assert torch.equal(b["next_states"], ring.obs[fn, at])
        assert torch.equal(b["actions"], ring.action[ft, at])
        assert torch.equal(b["rewards"], ring.reward[ft, at])
        assert torch.equal(b["dones"], ring.done[ft, at].float())
        assert torch.equal(b["valid"], ring.valid[ft, at].float())
        slots = f * n + a
        assert len(np.unique(slots)) == batch                          # random.sample: no transition twice
        # never the frame under construction (head) as a transition
        assert not (f == ring.head).any()
    env.close()


def test_draw_entry_point_matches_oracle_and_is_uniform():
    from oracle import philox as px
    from dqn_based_uav_3d_path_planer_amd import _lib
    lib = _lib.load()
    frames, n, head, filled, batch = 65, 16384, 17, 64, 16384
    out = torch.empty((batch, 2), dtype=torch.int32, device="cuda")
    hits = np.zeros(filled * n, dtype=np.int64)
    for counter in range(8):
        _lib.check(lib.uavenv_replay_draw(frames, n, head, filled, batch, 77, counter, out.data_ptr(), None), "draw")
        got = out.cpu().numpy()
        f, a = px.replay_draws(batch, 77, counter, head, filled, frames, n)
        assert (got[:, 0] == f).all() and (got[:, 1] == a).all()
        back = (head - 1 - f) % frames
        assert back.min() >= 0 and back.max() < filled
        np.add.at(hits, back * n + a, 1)
    # 8 x 16384 draws over 1 M slots: each update's draws are distinct, and across updates the per-frame load is flat
    per_frame = hits.reshape(filled, n).sum(1)
    assert abs(per_frame.mean() - 8 * batch / filled) < 1e-9
    assert per_frame.std() < 4 * np.sqrt(8 * batch / filled)
    assert hits.max() <= 5                  # 131 072 draws into 1 M slots: Poisson(1/8), P(>= 6 somewhere) ~ 1e-6


@pytest.mark.parametrize("dead", [0.0, 0.35, 0.9])
def test_draws_over_valid_rows_only(dead):
    """uavenv_replay_draw_valid (the draws of loops that skip finished agents: the reference's buffers never hold a row of a
    finished agent, Envs/PathPlan_City.py:456-459, BaseClass/replay_buffer.py:41-51) against oracle/philox.py's restatement:
    rejection over the permutation -- every accepted row is valid, rows stay distinct within a slot, every valid row is equally
    likely, and with every row valid the draws are uavenv_replay_draw's."""
    from oracle import philox as px
    from dqn_based_uav_3d_path_planer_amd import _lib
    lib = _lib.load()
    frames, n_envs, U, head, filled, batch = 33, 2048, 4, 9, 32, 1024
    rng = np.random.default_rng(4)
    valid = (rng.random((frames, n_envs * U)) >= dead).astype(np.uint8)
    vd = torch.tensor(valid, device="cuda")
    out = torch.empty((U * batch, 2), dtype=torch.int32, device="cuda")
    hits = np.zeros((U, filled * n_envs), dtype=np.int64)
    for counter in range(6):
        _lib.check(lib.uavenv_replay_draw_valid(frames, n_envs, head, filled, batch, U, U, 0, vd.data_ptr(), _lib.DRAW_MAX_TRIES, 91,
                                                counter, out.data_ptr(), None), "draw_valid")
        got = out.cpu().numpy()
        f, e, found = px.replay_draws_valid(batch, U, U, 0, valid, _lib.DRAW_MAX_TRIES, 91, counter, head, filled, frames, n_envs)
        assert (got[:, 0] == f).all() and (got[:, 1] == e).all()
        slot = np.arange(U * batch) // batch
        ok = valid.reshape(frames, n_envs, U)[f, e, slot] != 0
        assert (ok == found).all()
        if dead == 0.0:
            f0, e0 = px.replay_draws(U * batch, 91, counter, head, filled, frames, n_envs)
            assert found.all() and (f == f0).all() and (e == e0).all()
        elif dead < 0.5:
            assert found.mean() > 0.999                                   # 0.35 ** 8 = 2e-4 of the draws run out of tries
        back = (head - 1 - f) % frames
        assert back.min() >= 0 and back.max() < filled
        for j in range(U):
            m = (slot == j) & found
            rows = back[m] * n_envs + e[m]
            assert len(np.unique(rows)) == m.sum()                        # distinct within a slot
            np.add.at(hits[j], rows, 1)
        # one slot alone (the per-slot form: the ring does not hold U x batch transitions yet)
        _lib.check(lib.uavenv_replay_draw_valid(frames, n_envs, head, filled, batch, 1, U, 2, vd.data_ptr(), _lib.DRAW_MAX_TRIES, 91,
                                                counter, out.data_ptr(), None), "draw_valid")
        f1, e1, _ = px.replay_draws_valid(batch, 1, U, 2, valid, _lib.DRAW_MAX_TRIES, 91, counter, head, filled, frames, n_envs)
        got = out[:batch].cpu().numpy()
        assert (got[:, 0] == f1).all() and (got[:, 1] == e1).all()
    if dead == 0.35:        # uniform over the valid rows: hits per valid row ~ Poisson(6 * 1024 / #valid), never on an invalid one
        for j in range(U):
            back_all = np.arange(filled * n_envs) // n_envs
            f_all = (head - 1 - back_all) % frames
            v = valid.reshape(frames, n_envs, U)[f_all, np.arange(filled * n_envs) % n_envs, j] != 0
            assert hits[j][~v].sum() == 0
            lam = 6 * batch / v.sum()
            per_frame = np.array([hits[j][(back_all == b) & v].sum() for b in range(filled)])
            expect = np.array([lam * ((back_all == b) & v).sum() for b in range(filled)])
            assert np.abs(per_frame - expect).max() < 5 * np.sqrt(expect.max())


def test_fused_learner_draws_the_same_transitions_as_sample():
    """k_dqn_grad's in-kernel draw == uavenv_replay_sample's: learning from the ring with (seed, counter) equals
    learning from the explicit (frame, agent) list the oracle predicts -- bit for bit."""
    from oracle import philox as px
    from dqn_based_uav_3d_path_planer_amd.learner import FusedDQNLearner
    env, ring = _filled_ring(2048, 5, 5, torch.float32)
    param = {"NetWork": "Qnet2", "w": "100", "hiden_dim": "64", "output": "3"}
    torch.manual_seed(0)
    A = FusedDQNLearner(param, "dqn", device="cuda:0")
    B = FusedDQNLearner(param, "dqn", device="cuda:0")
    B.flat.copy_(A.flat)
    for it in range(3):
        f, a = px.replay_draws(4096, 21, it, ring.head, ring.filled, ring.frames, env.N)
        idx = torch.tensor(np.stack([f, a], 1).astype(np.int32), device="cuda").contiguous()
        la = float(A.learn_from_ring(ring, 4096, seed=21, counter=it))
        lb = float(B.learn_from_ring(ring, 4096, seed=21, counter=it, explicit_idx=idx))
        assert la == lb
    assert torch.equal(A.flat, B.flat)
    env.close()


def test_multi_uav_ring_marks_waiting_agents_invalid():
    """with several UAVs per env a finished agent waits for its team-mates (PathPlan_City.py:365-366);
    DeviceReplayRing.step_env must not step it, and its rows carry valid = 0."""
    env, ring = _filled_ring(256, 400, 0, torch.float32, uav=4)
    gen = torch.Generator(device="cuda").manual_seed(3)
    done_prev = torch.zeros(env.N, dtype=torch.bool, device="cuda")
    saw_waiting = 0
    for _ in range(330):
        ring.current_action().copy_(torch.randint(0, 3, (env.N,), generator=gen, device="cuda", dtype=torch.int32))
        t = ring.head
        ring.step_env(auto_reset=True)
        st = torch.tensor(env.get_state()[:, 10] != 0, device="cuda")          # agent done after this step
        valid = ring.valid[t].bool()
        # an agent that was already done before this step and whose env did not reset: not stepped, valid = 0
        assert not (valid & done_prev).any()
        saw_waiting += int((~valid).sum())
        done_prev = st
    assert saw_waiting > 0
    env.close()
