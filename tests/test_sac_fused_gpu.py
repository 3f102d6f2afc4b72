"""Fused SAC update (csrc/sac.hip, sac.FusedSACLearner) against the PyTorch-ROCm SACLearner -- itself pinned to the executed
reference's SAC_Trainer.update by tests/test_sac_golden.py -- on the same sampled transitions and the same rsample() draws:
raw gradients of both phases, losses, then whole updates (parameters, targets, log_alpha)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

PARAM = {"actor": {"NetWork": "PolicyNetContinuous_SAC", "w": "100", "action_bound": "1", "hiden_dim": "64", "output": "2", "lr": "0.0001"},
         "critic": {"NetWork": "QValueNetContinuous_SAC", "w": "100", "hiden_dim": "64", "action_dim": "2", "lr": "0.001"},
         "SAC_param": {"IS_Continuous": "1", "alpha_lr": "0.0001", "target_entropy": "1", "gamma": "0.99", "tau": "0.05"}}
A_NAMES = ("fc1.weight", "fc1.bias", "fc_mu.weight", "fc_std.weight", "fc_mu.bias", "fc_std.bias")
C_NAMES = ("fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias", "fc_out.weight", "fc_out.bias")


@pytest.fixture(scope="module")
def world():
    """a continuous-action ring with a few hundred frames of random flying (packed rows), 2 UAVs per env"""
    from dqn_based_uav_3d_path_planer_amd.data import make_city26_env
    from dqn_based_uav_3d_path_planer_amd.replay import DeviceReplayRing
    env = make_city26_env(512, uav_per_env=2, obs_dtype="packed")
    ring = DeviceReplayRing(env, 40 * env.N, discrete=False)
    ring.reset(seed=5)
    a1 = torch.zeros((ring.frames, env.N), dtype=torch.float32, device="cuda")
    gen = torch.Generator(device="cuda").manual_seed(3)
    for _ in range(30):
        ring.current_action().copy_(torch.rand(env.N, generator=gen, device="cuda") * 2 - 1)
        a1[ring.head].copy_(torch.rand(env.N, generator=gen, device="cuda") * 2 - 1)
        ring.step_env(auto_reset=True)
    yield env, ring, a1
    env.close()


def _pair(seed=0):
    from dqn_based_uav_3d_path_planer_amd.sac import SACLearner, FusedSACLearner
    torch.manual_seed(seed)
    fused = FusedSACLearner(PARAM)
    ref = SACLearner(PARAM)
    for name in ("actor", "critic_1", "critic_2", "target_critic_1", "target_critic_2"):
        getattr(ref, name).load_state_dict(getattr(fused, name).state_dict())
    # decorrelate critic 2 / the targets from critic 1 (the constructor gives the targets the critics' weights)
    with torch.no_grad():
        for net in (fused.target_critic_1, fused.target_critic_2):
            for p in net.parameters():
                p.add_(0.01 * torch.randn_like(p))
        for name in ("target_critic_1", "target_critic_2"):
            getattr(ref, name).load_state_dict(getattr(fused, name).state_dict())
    return fused, ref


def _batch(world, fused, B, seed, slot=1):
    env, ring, a1 = world
    gen = torch.Generator(device="cuda").manual_seed(seed)
    f = torch.randint(0, ring.head - 1, (B,), generator=gen, device="cuda", dtype=torch.int32)
    e = torch.randint(0, env.N // 2, (B,), generator=gen, device="cuda", dtype=torch.int32)
    draws = torch.stack([f, e], 1).contiguous()
    rows = (f.long() * env.N + e.long() * 2 + slot)
    nxt = rows + env.N
    flat = ring.obs.view(-1, ring.obs.shape[-1])
    b = fused.make_batch(flat, ring.action.view(-1), a1.view(-1), ring.reward.view(-1), ring.done.view(-1),
                         valid=ring.valid.view(-1), draws=draws, n_agents=env.N, uav_per_env=2, slot=slot, frames=ring.frames)
    td = dict(states=env.unpack(flat[rows]), next_states=env.unpack(flat[nxt]),
              actions=torch.stack([ring.action.view(-1)[rows], a1.view(-1)[rows]], 1),
              rewards=ring.reward.view(-1)[rows], dones=ring.done.view(-1)[rows].float())
    w = ring.valid.view(-1)[rows].float()
    eps = torch.randn((2, B, 2), generator=gen, device="cuda")
    return b, td, w, (eps[0].contiguous(), eps[1].contiguous())


def _flat(tensors):
    return torch.cat([t.reshape(-1) for t in tensors])


def _rel(a, b):
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def test_gradients_of_both_phases_match_autograd(world):
    from dqn_based_uav_3d_path_planer_amd import _lib
    fused, ref = _pair()
    B = 2048
    b, td, w, (e_next, e_cur) = _batch(world, fused, B, seed=11)
    assert 0.5 < float(w.mean()) <= 1.0
    # ---- phase A: td target, the two (weighted) critic losses and their gradients
    states, nstates, actions = td["states"], td["next_states"], td["actions"]
    rewards, dones = td["rewards"].view(-1, 1), td["dones"].view(-1, 1)
    target = ref.calc_target(rewards, nstates, dones, e_next).detach()
    pc = fused.critic_grad(b, e_next).sum(0)
    for k, net in enumerate((ref.critic_1, ref.critic_2)):
        q = net(states, actions)
        loss = torch.mean(w.view(-1, 1) * (q - target) ** 2)
        params = dict(net.named_parameters())
        g = _flat(torch.autograd.grad(loss, [params[n] for n in C_NAMES]))
        mine = pc[k * _lib.SAC_CRITIC_PARAMS:(k + 1) * _lib.SAC_CRITIC_PARAMS]
        assert _rel(mine, g) <= 2e-5, (k, _rel(mine, g))
        assert float((mine - g).abs().max()) <= 2e-5 * float(g.abs().max())
        assert abs(float(pc[2 * _lib.SAC_CRITIC_PARAMS + k]) - float(loss.detach())) <= 2e-5 * abs(float(loss.detach()))
    # ---- phase B on the SAME critics (no critic step in between here): actor loss, its gradient, sum of log pi
    new_actions, log_prob = ref.actor(states, e_cur)
    # rows with valid = 0 carry weight 0 in the actor loss and in sum log pi too (the kernels keep the 1 / B of a plain mean;
    # the Adam kernel divides by the valid fraction)
    actor_loss = torch.mean(w.view(-1, 1) * (ref.log_alpha.exp() * log_prob -
                                             torch.min(ref.critic_1(states, new_actions), ref.critic_2(states, new_actions))))
    params = dict(ref.actor.named_parameters())
    g = _flat(torch.autograd.grad(actor_loss, [params[n] for n in A_NAMES]))
    pa = fused.actor_grad(b, e_cur).sum(0)
    mine = pa[:_lib.SAC_ACTOR_PARAMS]
    assert _rel(mine, g) <= 5e-5, _rel(mine, g)
    assert abs(float(pa[_lib.SAC_ACTOR_PARAMS]) - float(actor_loss)) <= 2e-5 * max(1.0, abs(float(actor_loss)))
    lp_sum = float((w.view(-1, 1) * log_prob).sum())
    assert abs(float(pa[_lib.SAC_ACTOR_PARAMS + 1]) - lp_sum) <= 2e-5 * abs(lp_sum)
    assert abs(float(pa[_lib.SAC_ACTOR_PARAMS + 2]) - float(w.mean())) <= 1e-6            # the valid fraction of the batch
    assert abs(float(pc[2 * _lib.SAC_CRITIC_PARAMS + 2]) - float(w.mean())) <= 1e-6


@pytest.mark.parametrize("B", [64, 4096, 20480])
def test_whole_updates_track_the_torch_learner(world, B):
    """B = 64: one workgroup, one tile; 4096: one tile per workgroup; 20480: 320 tiles -> two per workgroup."""
    fused, ref = _pair(seed=B)
    for it in range(4):
        b, td, w, noise = _batch(world, fused, B, seed=100 + it, slot=it & 1)
        ref.learn(td, noise=noise, valid=w)
        fused.learn(b, noise=noise)
        torch.cuda.synchronize()
        # losses of this update (computed before the steps)
        assert abs(float(fused.loss) - float(ref.loss)) <= 1e-4 * max(1.0, abs(float(ref.loss))), it
        for name, names, lr in (("actor", A_NAMES, 1e-4), ("critic_1", C_NAMES, 1e-3), ("critic_2", C_NAMES, 1e-3),
                                ("target_critic_1", C_NAMES, 1e-3), ("target_critic_2", C_NAMES, 1e-3)):
            pf, pr = dict(getattr(fused, name).named_parameters()), dict(getattr(ref, name).named_parameters())
            d = torch.cat([(pf[n] - pr[n]).abs().reshape(-1) for n in names])
            # Adam normalises the step: an element whose gradient is ~0 can move by up to lr in either learner; everywhere
            # else the two agree to rounding.  Bars: 99 % of the elements within 2 % of one step, none beyond 2 steps/update.
            assert float(torch.quantile(d, 0.99)) <= 0.02 * lr * (it + 1), (it, name, float(torch.quantile(d, 0.99)))
            assert float(d.max()) <= 2.0 * lr * (it + 1), (it, name, float(d.max()))
        assert abs(float(fused.log_alpha) - float(ref.log_alpha)) <= 1e-6 * (it + 1)
    # and the policies still act alike
    env, ring, _ = world
    s = env.unpack(ring.current_obs())[:256]
    e = torch.randn((256, 2), device="cuda")
    assert float((fused.act(s, e) - ref.act(s, e)).abs().max()) <= 2e-3


def test_act_rows_equals_the_module_forward(world):
    """uavenv_sac_act on the packed rows of the current frame (one UAV slot, strided rows, ragged count) against the
    PolicyNetContinuous_SAC module on the unpacked rows with the same rsample() draws."""
    from dqn_based_uav_3d_path_planer_amd.sac import FusedSACLearner
    env, ring, _ = world
    torch.manual_seed(1)
    fused = FusedSACLearner(PARAM)
    flat = ring.obs.view(-1, ring.obs.shape[-1])
    n_rows = flat.shape[0]
    for slot, count in ((0, env.N // 2), (1, 333)):
        a0 = torch.full((n_rows,), 7.0, device="cuda")
        a1 = torch.full((n_rows,), 7.0, device="cuda")
        eps = torch.randn((count, 2), device="cuda")
        first = ring.head * env.N + slot
        fused.act_rows(flat, first, 2, count, a0, a1, eps)
        rows = first + 2 * torch.arange(count, device="cuda")
        ref = fused.act(env.unpack(flat[rows]), eps)
        assert float((a0[rows] - ref[:, 0]).abs().max()) <= 2e-6 and float((a1[rows] - ref[:, 1]).abs().max()) <= 2e-6
        untouched = torch.ones(n_rows, dtype=torch.bool, device="cuda")
        untouched[rows] = False
        assert bool((a0[untouched] == 7.0).all()) and bool((a1[untouched] == 7.0).all())


def test_rejects_what_it_cannot_do(world):
    from dqn_based_uav_3d_path_planer_amd.sac import FusedSACLearner
    from dqn_based_uav_3d_path_planer_amd._lib import UavEnvError
    bad = {k: dict(v) for k, v in PARAM.items()}
    bad["actor"]["hiden_dim"] = "128"
    with pytest.raises(ValueError):
        FusedSACLearner(bad)
    fused = FusedSACLearner(PARAM)
    env, ring, a1 = world
    idx = torch.arange(100, dtype=torch.int32, device="cuda")
    flat = ring.obs.view(-1, ring.obs.shape[-1])
    b = fused.make_batch(flat, ring.action.view(-1), a1.view(-1), ring.reward.view(-1), ring.done.view(-1), idx_s=idx, idx_n=idx + env.N)
    with pytest.raises((UavEnvError, ValueError)):
        fused.learn(b, noise=(torch.zeros((100, 2), device="cuda"), torch.zeros((100, 2), device="cuda")))


def _sac_rank_worker(rank, world, port, out_dir, mode="dist"):
    import os
    import torch.distributed as dist
    from dqn_based_uav_3d_path_planer_amd.data import make_city26_env
    from dqn_based_uav_3d_path_planer_amd.replay import DeviceReplayRing
    from dqn_based_uav_3d_path_planer_amd.sac import FusedSACLearner
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    env = make_city26_env(256, uav_per_env=2, obs_dtype="packed")
    ring = DeviceReplayRing(env, 12 * env.N, discrete=False)
    ring.reset(seed=5)                                  # every rank builds the SAME ring
    a1 = torch.zeros((ring.frames, env.N), dtype=torch.float32, device="cuda")
    gen = torch.Generator(device="cuda").manual_seed(3)
    for _ in range(8):
        ring.current_action().copy_(torch.rand(env.N, generator=gen, device="cuda") * 2 - 1)
        a1[ring.head].copy_(torch.rand(env.N, generator=gen, device="cuda") * 2 - 1)
        ring.step_env(auto_reset=True)
    torch.manual_seed(0)                                # same initial weights everywhere
    L = FusedSACLearner(PARAM)
    if world > 1 and mode != "dist":                    # the on-stream exchange instead of torch.distributed per phase
        # (the library's default wait bound, ~4 s: these are healthy-exchange scenarios, and two processes taking turns to load their
        # code objects on a busy box can be further apart than the 2^18 polls this test used to allow -- one full-suite run in three
        # of round 6 timed out in the set-up's self-test, which now runs four exchanges instead of one)
        assert L.enable_exchange(mode, spin_limit=0) == mode
    g = torch.Generator().manual_seed(7)
    B = 512
    draws = torch.stack([torch.randint(0, 7, (B,), generator=g), torch.randint(0, 256, (B,), generator=g)], 1).int()
    eps = torch.randn((4, 2, B, 2), generator=g)
    per = B // world
    sl = slice(rank * per, (rank + 1) * per)
    flat = ring.obs.view(-1, ring.obs.shape[-1])
    b = L.make_batch(flat, ring.action.view(-1), a1.view(-1), ring.reward.view(-1), ring.done.view(-1), valid=ring.valid.view(-1),
                     draws=draws[sl].contiguous().cuda(), n_agents=env.N, uav_per_env=2, slot=1, frames=ring.frames)
    losses = []
    for it in range(4):
        L.learn(b, noise=(eps[it, 0, sl].contiguous().cuda(), eps[it, 1, sl].contiguous().cuda()))
        losses.append([float(L.loss), float(L.critic_losses[0]), float(L.critic_losses[1])])
    torch.cuda.synchronize()
    torch.save({"actor": L._blocks[0].cpu(), "critics": L._cblocks[:4].cpu(), "log_alpha": float(L.log_alpha), "losses": losses},
               os.path.join(out_dir, f"sac_{mode}_w{world}_r{rank}.pt"))
    if world > 1:
        dist.barrier()
        L.disable_exchange()
        dist.destroy_process_group()
    env.close()


def test_fused_sac_two_ranks_equal_one_process(tmp_path):
    """The N > 1 branch of FusedSACLearner (uavenv_sac_reduce -> all_reduce(sum) of the column sums -> Adam on that one
    row with grad_scale = 1 / world), two ranks sharing this GPU over gloo: each takes half of a 512-transition list and
    half of the rsample() draws; the result must be the update ONE process applies to the whole list."""
    import os
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_sac_rank_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    mp.spawn(_sac_rank_worker, args=(1, port, str(tmp_path)), nprocs=1, join=True)
    r0, r1 = torch.load(os.path.join(tmp_path, "sac_dist_w2_r0.pt")), torch.load(os.path.join(tmp_path, "sac_dist_w2_r1.pt"))
    one = torch.load(os.path.join(tmp_path, "sac_dist_w1_r0.pt"))
    assert torch.equal(r0["actor"], r1["actor"]) and torch.equal(r0["critics"], r1["critics"])     # ranks in lock-step, bit for bit
    assert r0["log_alpha"] == r1["log_alpha"] and r0["losses"] == r1["losses"]
    # vs one process: the same gradient up to the order of the sums; Adam can move an element with a ~0 gradient by a step
    da = (r0["actor"] - one["actor"]).abs()
    dc = (r0["critics"] - one["critics"]).abs()
    assert float(torch.quantile(da, 0.99)) <= 1e-5 and float(da.max()) <= 8e-4
    assert float(torch.quantile(dc.reshape(-1), 0.99)) <= 1e-4 and float(dc.max()) <= 8e-3
    assert abs(r0["log_alpha"] - one["log_alpha"]) <= 1e-6
    assert np.allclose(np.array(r0["losses"]), np.array(one["losses"]), rtol=2e-4, atol=1e-5)


def test_fused_sac_peer_exchange_equals_torch_distributed(tmp_path):
    """FusedSACLearner.enable_exchange("p2p"): each phase's column sums summed over peer-mapped HBM on the stream
    (uavenv_p2p_allreduce, csrc/p2p.hip) instead of a torch.distributed all-reduce per phase -- two ranks on this GPU, four
    updates: weights, log_alpha and losses bit-identical to the torch.distributed form on both ranks."""
    import os
    import socket
    import torch.multiprocessing as mp
    res = {}
    for mode in ("p2p", "dist"):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        mp.spawn(_sac_rank_worker, args=(2, port, str(tmp_path), mode), nprocs=2, join=True)
        res[mode] = [torch.load(os.path.join(tmp_path, f"sac_{mode}_w2_r{r}.pt")) for r in (0, 1)]
    for r in (0, 1):
        a, b = res["p2p"][r], res["dist"][r]
        assert torch.equal(a["actor"], b["actor"]) and torch.equal(a["critics"], b["critics"])
        assert a["log_alpha"] == b["log_alpha"] and a["losses"] == b["losses"]
    assert torch.equal(res["p2p"][0]["actor"], res["p2p"][1]["actor"]) and torch.equal(res["p2p"][0]["critics"], res["p2p"][1]["critics"])


def _sac_loop_worker(rank, world, port, out_dir, nudge=False):
    import os
    import torch.distributed as dist
    from dqn_based_uav_3d_path_planer_amd.data import make_city26_env
    from dqn_based_uav_3d_path_planer_amd.loop import SACHotLoop
    from dqn_based_uav_3d_path_planer_amd.replay import DeviceReplayRing
    from dqn_based_uav_3d_path_planer_amd.sac import FusedSACLearner
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    U, envs, B = 2, 256, 256
    env = make_city26_env(envs, uav_per_env=U, obs_dtype="packed")
    ring = DeviceReplayRing(env, 12 * env.N, discrete=False)
    ring.reset(seed=5)                                  # every rank holds the SAME shard: the sum over the ranks is 2 x one rank's
    a1 = torch.zeros((ring.frames, env.N), dtype=torch.float32, device="cuda")
    torch.manual_seed(0)
    Ls = [FusedSACLearner(PARAM) for _ in range(U)]
    loop = SACHotLoop(ring, Ls, B, seed=11, act1_plane=a1, exchange="p2p" if world > 1 else None, spin_limit=0, check_every=4)
    assert loop.exchange == ("p2p" if world > 1 else None)
    loop.run(9)
    if nudge:                                           # one rank's critic drifts by one ulp: the next checksum compare must notice
        from dqn_based_uav_3d_path_planer_amd.loop import P2PExchangeError
        torch.cuda.synchronize()
        st0 = loop.p2p_status()
        assert st0["code"] == 0 and st0["checks"] >= 1 and st0["mismatches"] == 0
        if rank == 1:
            w = Ls[1]._cblocks[0]
            w[77] = torch.nextafter(w[77], w[77] + 1)
        raised = False
        try:
            for _ in range(6):
                loop.run(4)
                torch.cuda.synchronize()
        except P2PExchangeError:
            raised = True
        st = loop.p2p_status()
        torch.save({"raised": raised, "status": st, "counts": [(x.epoch, x.adam_steps) for x in Ls], "cursor": (ring.head, ring.filled)},
                   os.path.join(out_dir, f"sacnudge_r{rank}.pt"))
        dist.barrier()
        loop.close()
        dist.destroy_process_group()
        env.close()
        return
    loop.run(8)
    torch.cuda.synchronize()
    if world > 1:
        st = loop.p2p_status()
        assert st["code"] == 0 and st["checks"] >= 2 and st["mismatches"] == 0, st
    torch.save({"blocks": [torch.cat([x._blocks.reshape(-1), x._cblocks.reshape(-1), x._alpha_mv, x.log_alpha.reshape(1)]).cpu() for x in Ls],
                "ring": {k: getattr(ring, k).cpu() for k in ("obs", "action", "reward", "done", "valid")}, "a1": a1.cpu(),
                "counts": [(x.epoch, x.adam_steps) for x in Ls], "cursor": (ring.head, ring.filled, loop.counter)},
               os.path.join(out_dir, f"sacloop_w{world}_r{rank}.pt"))
    if world > 1:
        dist.barrier()
    loop.close()
    if world > 1:
        dist.destroy_process_group()
    env.close()


def test_sac_c_loop_draws_valid_rows_only(world):
    """A loop that skips finished agents instead of restarting them (the plugin's episode loop) leaves their rows in the ring with
    valid = 0; the reference's buffers never hold such rows (Envs/PathPlan_City.py:456-459), so the batches are drawn over the
    valid rows only (UavSacLoopConfig.valid_draws -> uavenv_replay_draw_valid): every (frame, env) pair a slot's update used is a
    valid row of THAT slot, while the ring holds plenty that are not."""
    from dqn_based_uav_3d_path_planer_amd.data import make_city26_env
    from dqn_based_uav_3d_path_planer_amd.loop import SACHotLoop
    from dqn_based_uav_3d_path_planer_amd.replay import DeviceReplayRing
    from dqn_based_uav_3d_path_planer_amd.sac import FusedSACLearner
    U, envs, B = 2, 512, 256
    env = make_city26_env(envs, uav_per_env=U, obs_dtype="packed")
    ring = DeviceReplayRing(env, 200 * env.N, discrete=False)
    ring.reset(seed=5)
    a1 = torch.zeros((ring.frames, env.N), dtype=torch.float32, device="cuda")
    torch.manual_seed(0)
    Ls = [FusedSACLearner(PARAM) for _ in range(U)]
    loop = SACHotLoop(ring, Ls, B, seed=11, act1_plane=a1, auto_reset=False, skip_done=True, gate_updates=True)
    checked = dead_rows_seen = 0
    for t in range(150):
        loop.run(1)
        torch.cuda.synchronize()
        if t == 1:      # agents at every age from here on: they run out of steps (:456-465) one by one (the first step of an episode
            st, sub, alias = env.get_state(0, env.N, want_sub=True)      # pops the aliased start node and zeroes Step: not before it)
            step = np.random.default_rng(6).integers(0, 150, env.N).astype(np.int32)
            env.set_state(0, np.c_[st[:, 0:5], st[:, 6:9]], step, st[:, 11].astype(np.int32), sub, alias=alias)
            env.observe(ring.obs[ring.head])
        if t < 3 or t % 7:
            continue
        v = ring.valid.view(ring.frames, envs, U)
        d = loop._draws.view(U, B, 2).long()
        back = (ring.head - 1 - torch.arange(ring.filled, device="cuda")) % ring.frames
        stored_dead = int((v[back] == 0).sum())
        if not bool(v[(ring.head - 1) % ring.frames].any()):
            break                                       # nobody moved any more: the updates are gated off from here on
        for j in range(U):
            ok = v[d[j, :, 0], d[j, :, 1], j] != 0
            dead_frac = float((v[back][:, :, j] == 0).float().mean())
            # (a draw gives up after UAVENV_DRAW_MAX_TRIES = 8 invalid rows in a row and keeps its first one: weight 0)
            assert int((~ok).sum()) <= 3 + 3 * B * dead_frac ** 8, (t, j, int((~ok).sum()), dead_frac)
            assert len(torch.unique(d[j, :, 0] * envs + d[j, :, 1])) == B
        checked += 1
        dead_rows_seen = max(dead_rows_seen, stored_dead)
    assert checked >= 10 and dead_rows_seen > 20 * B, (checked, dead_rows_seen)
    loop.close()
    env.close()


def test_sac_c_loop_exchanges_on_the_stream(tmp_path):
    """uavenv_sac_loop_run at N > 1 (UavSacLoopConfig.p2p): per phase the column sums of every slot (uavenv_sac_reduce) are
    summed over the ranks by uavenv_p2p_allreduce on the stream and the Adam kernels take that one row.  Two ranks on this
    GPU hold the SAME shard, so the sum is exactly twice one rank's and the valid-fraction column normalises it back:
    ring, weights, targets, Adam moments and log_alpha of both slots must equal the ONE-process loop bit for bit, on both
    ranks (any lost, stale or doubly-added slot would show)."""
    import os
    import socket
    import torch.multiprocessing as mp
    for world in (2, 1):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        mp.spawn(_sac_loop_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    one = torch.load(os.path.join(tmp_path, "sacloop_w1_r0.pt"))
    assert one["counts"][0][1] > 5                       # updates did happen
    for r in (0, 1):
        two = torch.load(os.path.join(tmp_path, f"sacloop_w2_r{r}.pt"))
        assert two["cursor"] == one["cursor"] and two["counts"] == one["counts"]
        for k in one["ring"]:
            assert torch.equal(two["ring"][k], one["ring"][k]), k
        assert torch.equal(two["a1"], one["a1"])
        for j in range(2):
            assert torch.equal(two["blocks"][j], one["blocks"][j]), (r, j)


def test_sac_c_loop_notices_diverged_ranks(tmp_path):
    """The generic peer exchange carries no checksum words: uavenv_sac_loop_run hashes every slot's actor / critics /
    targets every check_every updates and the ranks compare the hashes on the device (uavenv_p2p_check_blocks).  Two ranks on
    this GPU, one critic weight of rank 1 moved by one ulp: within check_every updates BOTH ranks carry the sticky error
    UAVENV_P2P_ERR_DIVERGED, run() raises, and the cursor / update counts were written back before it did."""
    import os
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_sac_loop_worker, args=(2, port, str(tmp_path), True), nprocs=2, join=True)
    for r in (0, 1):
        d = torch.load(os.path.join(tmp_path, f"sacnudge_r{r}.pt"))
        assert d["raised"] and d["status"]["code"] == 2 and d["status"]["mismatches"] >= 1, d
        assert d["cursor"][1] > 9 and d["counts"][0][0] > 9                          # SACHotLoop.run synced before raising
    a, b = (torch.load(os.path.join(tmp_path, f"sacnudge_r{r}.pt")) for r in (0, 1))
    assert a["counts"] == b["counts"]


def test_fused_sac_against_the_executed_reference():
    """csrc/sac.hip on the reference's OWN vectors (tests/golden/learner_SAC_Trainer_packed.npz, written by executing
    Trainer/SAC_Trainer.py:325-379 update / :122-131 calc_target / :145-147 soft_update on observations its state_PathPlan
    produced, with the two rsample() draws recorded): same weights, same 128 transitions as packed rows, same noise.
    Bars: actor loss 5e-5 relative per update; log_alpha 1e-5; after the golden's five updates EVERY parameter of the
    actor, both critics and both soft-updated targets within 0.5 % of ONE Adam step (lr) of the executed reference's
    value (measured: 0.045 % worst, i.e. 4.5e-8 absolute on the actor), 99 % of them within 0.1 %."""
    from conftest import load_golden, pack_obs_rows
    from dqn_based_uav_3d_path_planer_amd import _lib
    from dqn_based_uav_3d_path_planer_amd.sac import FusedSACLearner
    g = load_golden("learner_SAC_Trainer_packed.npz")
    B = len(g["states"])
    rows = np.concatenate([pack_obs_rows(g["states"]), pack_obs_rows(g["next_states"])])
    flat = torch.tensor(rows, device="cuda")
    # the packed rows ARE the reference's inputs: expanding them gives back its f32 rows bit for bit
    back = torch.empty((2 * B, 100), dtype=torch.float32, device="cuda")
    lib = _lib.load()
    assert lib.uavenv_obs_unpack(flat.data_ptr(), 2 * B, back.data_ptr(), _lib.OBS_F32, torch.cuda.current_stream().cuda_stream) == 0
    assert torch.equal(back.cpu(), torch.tensor(np.concatenate([g["states"], g["next_states"]])))
    L = FusedSACLearner(PARAM)
    for name in ("actor", "critic_1", "critic_2", "target_critic_1", "target_critic_2"):
        getattr(L, name).load_state_dict({k[len(name) + 2:]: torch.tensor(v) for k, v in g.items() if k.startswith(f"{name}0_")})
    z = torch.zeros(B, device="cuda")
    act0 = torch.cat([torch.tensor(g["actions"][:, 0], device="cuda"), z]).contiguous()
    act1 = torch.cat([torch.tensor(g["actions"][:, 1], device="cuda"), z]).contiguous()
    rew = torch.cat([torch.tensor(g["rewards"], device="cuda"), z]).contiguous()
    done = torch.cat([torch.tensor(g["dones"], device="cuda"), z]).to(torch.uint8).contiguous()
    idx_s = torch.arange(B, device="cuda", dtype=torch.int32)
    idx_n = idx_s + B
    b = L.make_batch(flat, act0, act1, rew, done, idx_s=idx_s, idx_n=idx_n)
    K = len(g["losses"])
    for k in range(K):
        n = torch.tensor(g["noise"][k], device="cuda")
        L.learn(b, noise=(n[0].contiguous(), n[1].contiguous()))
        torch.cuda.synchronize()
        assert abs(float(L.loss) - g["losses"][k]) <= 5e-5 * max(1.0, abs(g["losses"][k])), (k, float(L.loss), g["losses"][k])
        assert abs(float(L.log_alpha) - g["log_alpha"][k]) <= 1e-5, k
    worst = {}
    for name, lr in (("actor", 1e-4), ("critic_1", 1e-3), ("critic_2", 1e-3), ("target_critic_1", 1e-3), ("target_critic_2", 1e-3)):
        d = torch.cat([(v.cpu() - torch.tensor(g[f"{name}1_{k}"])).abs().reshape(-1) for k, v in getattr(L, name).state_dict().items()])
        worst[name] = (float(d.max()) / lr, float(torch.quantile(d, 0.99)) / lr)
        assert float(d.max()) <= 5e-3 * lr, (name, worst[name])
        assert float(torch.quantile(d, 0.99)) <= 1e-3 * lr, (name, worst[name])
    print("fused SAC vs executed reference, (max, q99) |dw| in Adam steps:", worst)


def test_prioritised_replay_in_the_fused_sac_update(world):
    """Trainer/SAC_Trainer.py:336-352 on the fused kernels: importance weights multiply each sample in the two critic losses
    (mean_s w_s err_s^2 -- the per-sample form of `is_weights * critic_loss`), and |min(Q1, Q2) - td_target|[:, 0] comes
    back per sample for ReplayTree.batch_update.  Against autograd / SACLearner on the same transitions, weights and noise."""
    from dqn_based_uav_3d_path_planer_amd import _lib
    fused, ref = _pair(seed=5)
    B = 2048
    b, td, v, (e_next, e_cur) = _batch(world, fused, B, seed=21)
    gen = torch.Generator(device="cuda").manual_seed(3)
    isw = (torch.rand(B, generator=gen, device="cuda") * 0.9 + 0.1).contiguous()
    abs_f = torch.zeros(B, device="cuda")
    b.is_weights, b.abs_td_out = isw.data_ptr(), abs_f.data_ptr()
    states, nstates, actions = td["states"], td["next_states"], td["actions"]
    rewards, dones = td["rewards"].view(-1, 1), td["dones"].view(-1, 1)
    target = ref.calc_target(rewards, nstates, dones, e_next).detach()
    pc = fused.critic_grad(b, e_next).sum(0)
    torch.cuda.synchronize()
    q1, q2 = ref.critic_1(states, actions), ref.critic_2(states, actions)
    want_abs = (torch.min(q1, q2) - target).abs().detach()[:, 0]
    assert float((abs_f - want_abs).abs().max()) <= 2e-5 * max(1.0, float(want_abs.max()))
    ww = (v * isw).view(-1, 1)
    for k, (net, q) in enumerate(((ref.critic_1, q1), (ref.critic_2, q2))):
        loss = torch.mean(ww * (q - target) ** 2)
        params = dict(net.named_parameters())
        g = _flat(torch.autograd.grad(loss, [params[n] for n in C_NAMES]))
        mine = pc[k * _lib.SAC_CRITIC_PARAMS:(k + 1) * _lib.SAC_CRITIC_PARAMS]
        assert _rel(mine, g) <= 2e-5, (k, _rel(mine, g))
        assert abs(float(pc[2 * _lib.SAC_CRITIC_PARAMS + k]) - float(loss.detach())) <= 2e-5 * abs(float(loss.detach()))
    # one whole update: the torch learner with the same weights (validity AND importance) and noise
    ref.learn(td, noise=(e_next, e_cur), is_weights=isw, valid=v)
    fused.learn(b, noise=(e_next, e_cur))
    torch.cuda.synchronize()
    assert float((abs_f - ref.abs_errors).abs().max()) <= 2e-5 * max(1.0, float(ref.abs_errors.max()))
    for name, names, lr in (("actor", A_NAMES, 1e-4), ("critic_1", C_NAMES, 1e-3), ("critic_2", C_NAMES, 1e-3)):
        pf, pr = dict(getattr(fused, name).named_parameters()), dict(getattr(ref, name).named_parameters())
        d = torch.cat([(pf[n] - pr[n]).abs().reshape(-1) for n in names])
        assert float(torch.quantile(d, 0.99)) <= 0.02 * lr and float(d.max()) <= 2.0 * lr, (name, float(d.max()))


@pytest.mark.parametrize("B,tpw", [(64, 0), (4096, 0), (20480, 0), (8192, 8), (8192, 3)])
def test_td_targets_as_a_launch_of_their_own_change_nothing(world, B, tpw):
    """Round 4: with UavSacBatch.td_scratch the critic phase computes the td targets (Trainer/SAC_Trainer.py:122-131) in k_sac_td --
    eight wavefronts per workgroup, two tiles in flight, the staged actor + target critics shared -- and k_sac_critic_grad reads
    them from global memory; without it (UAVENV_SAC_FUSED_TD=1: the round-3 form) the gradient kernel computes them itself.
    k_sac_td runs layer 1 of its three forwards at f32 accuracy on the f16 matrix pipe (fc1 as two f16 terms against the exact
    0 / 1 flag columns, the scalar columns in f32: csrc/sac.hip "Layer 1 at f32 accuracy"): every product is exact and only the
    order of the f32 sums differs, so the td targets agree to ~1e-7 and the partial rows (gradients, losses, valid fraction)
    to 2e-6 relative L2 -- for one, several and an odd number of tiles per workgroup (the two halves of a workgroup then
    walk 2 and 1 tiles)."""
    import os
    fused, _ = _pair(seed=7)
    b, td, w, (e_next, e_cur) = _batch(world, fused, B, seed=21)
    b.tiles_per_wg = tpw
    assert b.td_scratch
    rows = []
    for fused_td in ("", "1"):
        if fused_td:
            os.environ["UAVENV_SAC_FUSED_TD"] = "1"
        else:
            os.environ.pop("UAVENV_SAC_FUSED_TD", None)
        try:
            pc = fused.critic_grad(b, e_next)
            torch.cuda.synchronize()
            n = fused._rows_launched
            rows.append(pc[:n].clone())
        finally:
            os.environ.pop("UAVENV_SAC_FUSED_TD", None)
    assert rows[0].shape == rows[1].shape and torch.isfinite(rows[0]).all()
    a, b_ = rows[0].double().sum(0), rows[1].double().sum(0)
    rel = float((a - b_).norm() / b_.norm())
    print("split vs fused td: partial rows rel L2", rel)
    assert rel <= 2e-6, rel
