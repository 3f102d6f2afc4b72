"""Device prioritised replay (csrc/per.hip, replay.DevicePER) against the executed reference's ReplayTree
(tests/golden/per.npz, made by oracle/gen_golden_per.py) and its CPU restatement (oracle/per_oracle.py)."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


def test_selection_and_weights_match_the_reference_on_its_own_draws():
    from dqn_based_uav_3d_path_planer_amd.replay import DevicePER
    g = load_golden("per.npz")
    for ci in range(4):
        pre = f"c{ci}_"
        cap, batch = int(g[pre + "capacity"]), int(g[pre + "batch"])
        per = DevicePER(cap)
        per.set_priorities(torch.tensor(g[pre + "prio"]), n_entries=int(g[pre + "n_entries"]))
        assert abs(per.total() - float(g[pre + "total"])) <= 1e-12 * float(g[pre + "total"])
        assert len(per) == int(g[pre + "total_int"])
        for r in range(3):
            slots, w, p = per.sample(batch, draws=torch.tensor(g[pre + f"r{r}_draws"]))
            assert np.array_equal(slots.cpu().numpy(), g[pre + f"r{r}_tree_idx"] - (cap - 1))       # bit-exact indices
            assert np.allclose(w.cpu().numpy(), g[pre + f"r{r}_weights"], rtol=1e-12, atol=0)
            assert per.beta == float(g[pre + f"r{r}_beta"])


def test_push_and_batch_update_priorities():
    from dqn_based_uav_3d_path_planer_amd.replay import DevicePER
    g = load_golden("per.npz")
    pre = "c0_"
    cap = int(g[pre + "capacity"])
    per = DevicePER(cap)
    # push(error): (|e| + eps) ** alpha, no clip  -> same path as update() with clip off
    per.clip = 0.0
    per.update(torch.arange(cap), torch.tensor(g[pre + "errors"]))
    assert np.allclose(per.prio.cpu().numpy(), g[pre + "prio_after_push"], rtol=4e-16, atol=0)
    per.clip = 1.0
    per.update(torch.tensor(g[pre + "upd_data"]), torch.tensor(g[pre + "upd_err"]))
    assert np.allclose(per.prio.cpu().numpy(), g[pre + "prio"], rtol=4e-16, atol=0)
    per.fill(3, 5, error=0.0)                                   # fresh transitions: 0.01 ** 0.6
    assert np.allclose(per.prio[3:8].cpu().numpy(), 0.01 ** 0.6, rtol=1e-15)
    per.fill(3, 5, valid=torch.tensor([1, 0, 1, 0, 0], dtype=torch.uint8, device="cuda"))
    assert (per.prio[3:8] > 0).cpu().tolist() == [True, False, True, False, False]


def test_large_capacity_matches_the_cumsum_restatement_and_the_distribution():
    from dqn_based_uav_3d_path_planer_amd.replay import DevicePER
    from oracle.per_oracle import sample_by_cumsum
    rng = np.random.default_rng(5)
    for cap in (1 << 20, 1000003):
        prio = rng.uniform(0.0, 1.0, cap) ** 3
        prio[rng.integers(0, cap, cap // 10)] = 0.0
        per = DevicePER(cap)
        per.set_priorities(torch.tensor(prio))
        total = per.total()
        assert abs(total - prio.sum()) <= 1e-9 * prio.sum()
        batch = 4096
        seg = np.floor(total) / batch
        draws = seg * (np.arange(batch) + rng.uniform(0, 1, batch))
        slots, w, p = per.sample(batch, draws=torch.tensor(draws))
        want = sample_by_cumsum(prio, None, draws)
        got = slots.cpu().numpy()
        assert (got != want).sum() <= 2                        # only a draw within rounding of a leaf edge may differ
        assert np.array_equal(p.cpu().numpy(), prio[got]) and (prio[got] > 0).all()
        # Philox draws: stratified -> sample i comes from segment i; frequencies follow the priorities
        slots2, _, _ = per.sample(batch, seed=3, counter=9)
        s2 = slots2.cpu().numpy()
        assert (prio[s2] > 0).all() and len(np.unique(s2)) > batch * 0.95
        slots3, _, _ = per.sample(batch, seed=3, counter=9)
        assert torch.equal(slots2, slots3)                     # counter-based: reproducible
    # a heavy slot is drawn in proportion to its priority
    per = DevicePER(4096)
    pr = np.full(4096, 0.01)
    pr[1234] = 40.96 * 3                                       # 3/4 of the mass
    per.set_priorities(torch.tensor(pr))
    s, w, _ = per.sample(1024, seed=1, counter=1)
    frac = float((s == 1234).float().mean())
    assert 0.70 < frac < 0.80 and float(w.max()) == 1.0
    assert abs(float(w.min()) - (122.88 / 0.01) ** -per.beta) < 1e-12              # heavy slot: smallest weight


def test_ring_integration_new_frames_are_prioritised_and_the_head_is_retired():
    from dqn_based_uav_3d_path_planer_amd.data import make_city26_env
    from dqn_based_uav_3d_path_planer_amd.learner import DQNLearner
    from dqn_based_uav_3d_path_planer_amd.replay import DevicePER, DeviceReplayRing
    n = 512
    env = make_city26_env(n)
    ring = DeviceReplayRing(env, 4 * n)
    ring.reset(seed=3)
    per = DevicePER(ring.frames * n, tree_order=False)
    gen = torch.Generator(device="cuda").manual_seed(0)
    L = DQNLearner({"NetWork": "Qnet2", "w": "100", "hiden_dim": "64", "output": "3"}, "dqn", device="cuda:0")
    for t in range(9):                                          # wraps the 5-frame ring
        ring.current_action().copy_(torch.randint(0, 3, (n,), generator=gen, device="cuda", dtype=torch.int32))
        ring.step_env(auto_reset=True)
        per.on_frame(ring)
        pf = per.prio.view(ring.frames, n)
        assert float(pf[ring.head].abs().max()) == 0.0         # the head frame has no action / reward yet
        assert per.n_entries == ring.filled * n
        slots, w, _ = per.sample(256, seed=1, counter=t)
        f = torch.div(slots, n, rounding_mode="floor")
        assert bool((f != ring.head).all())
        batch = ring.gather(slots)
        assert bool((batch["valid"] == 1).all())
        # next_states of a gathered transition = the observation one frame later of the same agent
        a = slots - f * n
        assert torch.equal(batch["next_states"], ring.obs[(f + 1) % ring.frames, a])
        loss, abs_err = L.learn_weighted(batch, w.float())
        per.update(slots, abs_err)
        assert torch.isfinite(loss)
    # re-prioritised slots carry min(|err| + eps, 1) ** alpha
    want = torch.clamp(abs_err.double() + 0.01, max=1.0) ** 0.6
    assert torch.allclose(per.prio[slots], want, rtol=1e-12)
    env.close()


def test_zero_priority_leaves_are_never_picked_and_weights_stay_finite():
    """Rounding (or a draw on the boundary) used to fall back to the chunk's last leaf even when it was a retired slot
    with priority 0 -> pow(0, -beta) = inf -> NaN weights for the whole batch (the reference's tree has the same hole)."""
    from dqn_based_uav_3d_path_planer_amd.replay import DevicePER
    cap = 5000
    prio = torch.zeros(cap, dtype=torch.float64)
    prio[100:200] = 0.37                 # every chunk ends in zeros, most chunks are entirely zero
    prio[3000:3003] = 1e-9
    per = DevicePER(cap, tree_order=False)
    per.set_priorities(prio, n_entries=103)
    total = per.total()
    draws = torch.tensor([0.0, total, total * (1 + 1e-15), 37.0 - 1e-13, 37.0, 37.0 + 2e-9, total - 1e-12] + [1e-3 * k for k in range(57)],
                         dtype=torch.float64).clamp_(0.0, total * (1 + 1e-15))
    slots, w, p = per.sample(draws.numel(), draws=draws)
    s = slots.cpu().numpy()
    assert np.all(prio.numpy()[s] > 0) and np.all(p.cpu().numpy() > 0)
    assert torch.isfinite(w).all() and float(w.max()) == 1.0 and float(w.min()) > 0
    # priorities summing to less than 1: int(total) == 0 in the reference; here the divisor is clamped
    per2 = DevicePER(256, tree_order=False)
    q = torch.zeros(256, dtype=torch.float64)
    q[10:20] = 0.01
    per2.set_priorities(q, n_entries=10)
    _, w2, _ = per2.sample(32, seed=1, counter=0)
    assert torch.isfinite(w2).all() and float(w2.max()) == 1.0


PARAM = {"NetWork": "Qnet2", "w": "100", "hiden_dim": "64", "output": "3"}


def test_weights_kernel_matches_the_reference_weights():
    """uavenv_per_weights (the in-loop form of ReplayTree.sample's importance weights, :175-178) on the reference's own
    draws: equal to the golden weights at f32 resolution, and the (frame, agent) split of the slots is exact."""
    import ctypes as C
    from dqn_based_uav_3d_path_planer_amd import _lib
    from dqn_based_uav_3d_path_planer_amd.replay import DevicePER
    g = load_golden("per.npz")
    for ci in range(4):
        pre = f"c{ci}_"
        cap, batch = int(g[pre + "capacity"]), int(g[pre + "batch"])
        per = DevicePER(cap)
        per.set_priorities(torch.tensor(g[pre + "prio"]), n_entries=int(g[pre + "n_entries"]))
        slots, w, p = per.sample(batch, draws=torch.tensor(g[pre + "r0_draws"]))
        w32 = torch.zeros(batch, dtype=torch.float32, device="cuda")
        fa = torch.zeros((batch, 2), dtype=torch.int32, device="cuda")
        n_agents = 7
        p = torch.cat([p, torch.zeros((batch + 255) // 256, dtype=torch.float64, device="cuda")])     # + the call's scratch
        rc = per.lib.uavenv_per_weights(C.byref(per._c), slots.data_ptr(), p.data_ptr(), batch, per.n_entries, per.beta, n_agents,
                                        w32.data_ptr(), fa.data_ptr(), torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        torch.cuda.synchronize()
        assert np.allclose(w32.cpu().numpy(), g[pre + "r0_weights"].astype(np.float32), rtol=2e-7, atol=0)
        assert torch.equal(fa[:, 0].long() * n_agents + fa[:, 1].long(), slots)


def _ring(n=1024, frames=8, seed=4):
    from dqn_based_uav_3d_path_planer_amd.data import make_city26_env
    from dqn_based_uav_3d_path_planer_amd.replay import DeviceReplayRing
    env = make_city26_env(n)
    ring = DeviceReplayRing(env, (frames - 1) * n, discrete=True)
    ring.reset(seed=seed)
    return env, ring


@pytest.mark.parametrize("kind,net,huber", [("dqn", "Qnet2", False), ("dueling", "VAnet2", True)])
def test_weighted_fused_update_matches_the_torch_learner(kind, net, huber):
    """uavenv_dqn_grad_w: importance weights in the loss and |TD error| out, against DQNLearner.learn_weighted (the
    PyTorch statement of Trainer/SAC_Trainer.py:346-352 for the DQN family) on the same transitions."""
    from dqn_based_uav_3d_path_planer_amd.learner import DQNLearner, FusedDQNLearner
    env, ring = _ring()
    gen = torch.Generator(device="cuda").manual_seed(1)
    for _ in range(5):
        ring.current_action().copy_(torch.randint(0, 3, (env.N,), generator=gen, device="cuda", dtype=torch.int32))
        ring.step_env(auto_reset=True)
    param = dict(PARAM, NetWork=net)
    torch.manual_seed(0)
    T = DQNLearner(param, kind, device="cuda:0", loss="huber" if huber else "mse")
    F = FusedDQNLearner(param, kind, device="cuda:0", loss="huber" if huber else "mse")
    F.q_local.load_state_dict(T.q_local.state_dict())
    F.q_target.load_state_dict(T.q_target.state_dict())
    B = 2048
    for it in range(3):
        f = torch.randint(0, ring.filled, (B,), generator=gen, device="cuda")
        f = (ring.head - 1 - f) % ring.frames
        a = torch.randint(0, env.N, (B,), generator=gen, device="cuda")
        idx = torch.stack([f, a], 1).int().contiguous()
        w = torch.rand(B, generator=gen, device="cuda") * 0.9 + 0.1
        batch = ring.gather(f * env.N + a)
        lt, abs_t = T.learn_weighted(batch, w)
        abs_f = torch.zeros(B, device="cuda")
        lf = F.learn_from_ring(ring, B, 0, 0, explicit_idx=idx, is_weights=w.contiguous(), abs_td_out=abs_f)
        torch.cuda.synchronize()
        assert abs(float(lt) - float(lf)) <= 3e-5 * abs(float(lt)), (it, float(lt), float(lf))
        assert float((abs_f - abs_t).abs().max()) <= 2e-5 * max(1.0, float(abs_t.max()))
    for (k, x), (_, y) in zip(T.q_local.state_dict().items(), F.q_local.state_dict().items()):
        assert (x - y).abs().max().item() <= 3e-5, k
    # and the batch-dict entry point the trainer plugins use
    batch = ring.gather(f * env.N + a)
    lt, abs_t = T.learn_weighted(batch, w)
    lf, abs_f = F.learn_weighted(batch, w)
    assert abs(float(lt) - float(lf)) <= 3e-5 * abs(float(lt))
    assert float((abs_f - abs_t).abs().max()) <= 2e-5 * max(1.0, float(abs_t.max()))
    env.close()


def test_prioritised_replay_inside_the_c_loop_equals_the_stepwise_path():
    """IsPriority_Replay = 1 as a PATH: HotLoop(per=...) -- per pass new-frame priorities, rebuild, ReplayTree.sample,
    importance weights, the weighted fused update, batch_update, all enqueued from C -- against the same sequence issued
    call by call from Python through DevicePER (itself pinned to the executed reference's ReplayTree above): identical
    slots, priorities, beta, and weights after K passes."""
    from dqn_based_uav_3d_path_planer_amd.learner import FusedDQNLearner
    from dqn_based_uav_3d_path_planer_amd.loop import HotLoop
    from dqn_based_uav_3d_path_planer_amd.replay import DevicePER
    K, B, seed = 12, 512, 9
    out = []
    for mode in ("c", "py"):
        env, ring = _ring(n=1024, frames=6, seed=4)          # 12 passes wrap the 6-frame ring: retired frames are exercised
        torch.manual_seed(0)
        L = FusedDQNLearner(PARAM, "dqn", device="cuda:0")
        per = DevicePER(ring.frames * env.N, tree_order=False)
        if mode == "c":
            hot = HotLoop(ring, L, B, seed=seed, eps=0.2, per=per)
            hot.run(K)
            torch.cuda.synchronize()
            last_slots = hot._per_bufs[0].clone()
            hot.close()
        else:
            abs_f = torch.zeros(B, device="cuda")
            for t in range(K):
                L.act(ring.current_obs(), 0.2, seed, t, index_out=ring.current_action())
                ring.step_env(auto_reset=True)
                per.on_frame(ring)
                if ring.filled * env.N >= B:
                    slots, w, _ = per.sample(B, seed=seed, counter=t)
                    idx = torch.stack([slots // env.N, slots % env.N], 1).int().contiguous()
                    L.learn_from_ring(ring, B, seed, t, explicit_idx=idx, is_weights=w.float().contiguous(), abs_td_out=abs_f)
                    per.update(slots, abs_f)
                    last_slots = slots.clone()
            torch.cuda.synchronize()
        out.append((per.prio.clone(), per.beta, per.n_entries, L.flat.clone(), last_slots, L.epoch))
        env.close()
    (pc, bc, nc, wc, sc, ec), (pp, bp, npy, wp, sp, ep) = out
    assert ec == ep and ec >= K - 1 and nc == npy and abs(bc - bp) <= 1e-6
    assert torch.equal(sc, sp)                                     # the same transitions were drawn at the last update
    assert float((pc - pp).abs().max()) <= 1e-9                    # priorities (f64 pow of an f32 |TD error|)
    assert float((wc[:2] - wp[:2]).abs().max()) <= 1e-6            # weights
    live = pc > 0
    fresh = (0.0 + 0.01) ** 0.6
    assert 0 < int(live.sum()) <= (6 - 1) * 1024 and float(((pc[live] - fresh).abs() > 1e-9).float().mean()) > 0.05


def test_batch_update_with_repeated_leaves_is_last_wins_and_deterministic():
    """ReplayTree.batch_update (replay_buffer.py:215-222) is a sequential loop: a leaf drawn several times in one batch ends
    with the LAST sample's error.  One dominant priority makes the stratified sampler return the same leaf for most of the
    batch; uavenv_per_set / _set_f32 must then produce what the sequential restatement does, run after run."""
    from dqn_based_uav_3d_path_planer_amd.replay import DevicePER
    cap, batch = 4096, 512
    rng = np.random.default_rng(3)
    prio = rng.uniform(0.001, 0.01, cap)
    prio[1234] = 500.0                                        # ~95 % of the mass in one leaf
    prio[77] = 10.0
    results = []
    for rep in range(3):
        per = DevicePER(cap, tree_order=False)
        per.set_priorities(torch.tensor(prio), n_entries=cap)
        slots, w, p = per.sample(batch, seed=5, counter=1)
        sl = slots.cpu().numpy()
        assert (np.diff(sl) >= 0).all()                      # prefix order: equal slots are adjacent
        assert (sl == 1234).sum() > batch // 2 and len(set(sl.tolist())) < batch
        err = torch.tensor(rng.uniform(0.0, 2.0, batch)) if rep == 0 else err     # noqa: F821
        per.update(slots, err)
        want = prio.copy()
        for i in range(batch):                                # the reference's loop: later samples overwrite earlier ones
            want[sl[i]] = min(abs(float(err[i])) + per.epsilon, per.clip) ** per.alpha
        got = per.prio.cpu().numpy()
        assert np.allclose(got, want, rtol=4e-16, atol=0)
        results.append(got)
        per2 = DevicePER(cap, tree_order=rep == 1)            # the f32-error form the fused learners feed; rep 1: rotated slots
        per2.set_priorities(torch.tensor(prio), n_entries=cap)
        s2, _, _ = per2.sample(batch, seed=5, counter=1)
        per2.update_f32(s2, err.float().cuda())
        want32 = prio.copy()
        e32 = err.float().double().numpy()
        sl2 = s2.cpu().numpy()
        for i in range(batch):
            want32[sl2[i]] = min(abs(e32[i]) + per2.epsilon, per2.clip) ** per2.alpha
        assert np.allclose(per2.prio.cpu().numpy(), want32, rtol=4e-16, atol=0)
    assert np.array_equal(results[0], results[1]) and np.array_equal(results[1], results[2])
