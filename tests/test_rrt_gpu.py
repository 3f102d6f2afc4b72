"""GPU sub-goal planner (csrc/rrt.hip) vs the reference's RRT (PathPlan/RRT.py:63-105):
  * fed CPython's Mersenne stream for random.seed(k), it must reproduce the reference's reset (start, goal, sub-goal
    list) recorded in tests/golden/resets.npz;
  * fed arbitrary uniform streams it must agree with the CPU oracle's planner on the same stream;
  * in Philox mode every planned path satisfies the planner's own invariants at bank scale."""
import random

import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
KNOWN_FORKS = set()      # seeds of tests/golden/resets.npz whose tree forks on gfx950: none (round 4: all 44 reproduce)
TOL = 1e-9      # f64; the only differences are pow(x,2) vs x*x and OCML vs glibc sqrt-free arithmetic (last ulp)


@pytest.fixture(scope="module")
def env():
    from dqn_based_uav_3d_path_planer_amd.data import make_city26_env
    e = make_city26_env(64, max_subgoals=64)
    yield e
    e.close()


def _mt_stream(seed, n):
    r = random.Random(int(seed))
    return np.array([r.random() for _ in range(n)])


def test_replays_the_reference_reset_from_the_mersenne_stream(env):
    g = load_golden("resets.npz")
    seeds = g["seeds"]
    L = 6000
    u = np.stack([_mt_stream(s, L) for s in seeds])
    sg, sub, ns, it = env.rrt_plan(len(seeds), uniforms=u)
    sg, sub, ns = sg.cpu().numpy(), sub.cpu().numpy(), ns.cpu().numpy()
    st = g["state"]
    forks = []
    for k in range(len(seeds)):
        assert np.abs(sg[k] - np.r_[st[k, 0:3], st[k, 6:9]]).max() <= TOL           # start / goal draws (UAV.py:353-358)
        if not (ns[k] == g["n_sub"][k] and np.abs(sub[k, :ns[k]] - g["sub_goals"][k, :ns[k]]).max() <= TOL):
            forks.append(int(seeds[k]))
    # a last-ulp difference (x*x vs pow(x, 2), OCML vs glibc) can flip one `d < step` decision and fork the tree.  WHICH
    # seeds fork is a property of the arithmetic, not of scheduling: the set is recorded and must not change
    print("forking seeds:", forks)
    assert set(forks) <= KNOWN_FORKS, f"reference resets that no longer reproduce: {sorted(set(forks) - KNOWN_FORKS)}"


def test_matches_the_cpu_oracle_on_arbitrary_streams(env):
    from oracle import pyoracle as po
    w = load_golden("world_stock.npz")
    world = po.OracleWorld(w["buildings"])
    rng = np.random.default_rng(5)
    m, L = 96, 6000
    u = rng.random((m, L))
    start = np.c_[rng.uniform(10, 210, m), rng.uniform(1, 10, m), np.zeros(m)]
    goal = np.c_[rng.uniform(330, 490, m), rng.uniform(420, 490, m), np.zeros(m)]
    sg, sub, ns, it = env.rrt_plan(m, start_goal=np.c_[start, goal], uniforms=u)
    sub, ns, it = sub.cpu().numpy(), ns.cpu().numpy(), it.cpu().numpy()
    same = 0
    for k in range(m):
        path, iters = po.rrt_get_path(world, po.OracleRng(0).replay(u[k]), start[k], goal[k])
        if len(path) == abs(ns[k]) and iters == it[k] and (ns[k] < 0 or np.abs(sub[k, :ns[k]] - path).max() <= TOL):
            same += 1
    assert same >= m - 3, f"{same}/{m} plans identical to the oracle"


def test_philox_bank_invariants_and_feeds_the_env(env):
    from oracle import pyoracle as po
    w = load_golden("world_stock.npz")
    world = po.OracleWorld(w["buildings"])
    m = 4096
    sg, sub, ns, it = env.rrt_plan(m, seed=11)
    sg, sub, ns, it = sg.cpu().numpy(), sub.cpu().numpy(), ns.cpu().numpy(), it.cpu().numpy()
    ok = (ns >= 2) & (ns <= env.K)
    assert ok.mean() > 0.99 and 15 < ns[ok].mean() < 35            # reference: 19..34, mean 24.7
    assert np.all((sg[:, 0] >= 10) & (sg[:, 0] <= 210) & (sg[:, 1] >= 1) & (sg[:, 1] <= 10))
    assert np.all((sg[:, 3] >= 330) & (sg[:, 3] <= 490) & (sg[:, 4] >= 420) & (sg[:, 4] <= 490))
    for k in np.nonzero(ok)[0][:400]:
        p = sub[k, :ns[k]]
        assert np.array_equal(p[0], sg[k, :3]) and np.array_equal(p[-1], sg[k, 3:])     # start .. goal
        seg = np.linalg.norm(np.diff(p, axis=0), axis=1)
        assert seg.max() <= 30.0 + 1e-9                                                  # steer step / goal radius
        pts = np.concatenate([p[i] + (p[i + 1] - p[i]) * np.linspace(0, 1, 7)[:-1, None] for i in range(len(p) - 1)])
        assert world.threaten_rate_many(pts[::2]).sum() <= 2      # segments were obstacle-checked every 5 m
    # the planned bank drives resets: plan -> reset -> rollouts with auto-reset stay consistent
    env.plan_scenarios(2048, seed=3, max_iter=10000)
    obs = env.reset(seed=1)
    st, subs, alias = env.get_state(0, env.N, want_sub=True)
    assert np.all(st[:, 11] >= 2) and np.all(alias == 1) and torch.isfinite(obs).all()
    out = env.alloc_out()
    for t in range(200):
        env.step(torch.zeros(env.N, dtype=torch.int32, device="cuda") + 1, out, auto_reset=True)
    st = env.get_state(0, env.N)
    assert np.isfinite(st).all() and np.all(st[:, 11] >= 1)


def test_bank_stats_and_a_bank_nobody_could_plan(env):
    """Scenarios whose plan failed take a neighbour's: the count is reported; a bank without a single valid plan is
    refused (it would install n_total < 2 at every reset) and the bank in use stays."""
    from dqn_based_uav_3d_path_planer_amd._lib import UavEnvError
    env.plan_scenarios(2048, seed=3, max_iter=10000)
    m, replaced = env.bank_stats()
    assert m == 2048 and 0 <= replaced < 0.01 * m
    with pytest.raises(UavEnvError, match="none of the"):
        env.plan_scenarios(64, seed=3, max_iter=1)           # one RRT iteration never spans 300+ m in 30 m steps
    assert env.bank_stats() == (2048, replaced)
    obs = env.reset(seed=2)
    assert torch.isfinite(obs).all()


def test_rolling_refresh_of_the_bank():
    """uavenv_replan_begin / _ready / _commit (the reference plans a fresh path at EVERY reset, Agents/UAV.py:327-366; here the
    reset bank turns over in the background while a loop runs).
      * the background planner (no LDS, node list in global memory) computes what the two LDS tiers compute: a refresh with
        the bank's own (seed, row) streams leaves every row bit for bit as it was;
      * with a new seed the rows of the slice change, stay valid paths, and the rest of the bank is untouched;
      * a row an agent is flying keeps its old plan (the agent reads its sub-goal list from the bank) and is counted."""
    from dqn_based_uav_3d_path_planer_amd.data import make_city26_env
    e = make_city26_env(256, bank="gpu", bank_size=2048, bank_seed=5)
    sg0, sub0, ns0 = e.bank_read()
    assert ((ns0 >= 2) & (ns0 <= e.K)).all()
    assert e.replan_ready() == -1
    # (1) same streams, other kernel form: nothing changes
    e.replan_begin(256, 512, seed=5)
    while e.replan_ready() == 0:
        pass
    assert e.replan_ready() == 1
    e.replan_commit()
    sg1, sub1, ns1 = e.bank_read()
    st = e.replan_stats()
    assert np.array_equal(sg0, sg1) and np.array_equal(sub0, sub1) and np.array_equal(ns0, ns1)
    assert st["refreshes"] == 1 and st["rows_planned"] == 512 and st["rows_in_use"] == 0
    assert st["rows_committed"] + st["rows_without_plan"] == 512 and st["rows_committed"] > 480
    # (2) a new generation; the 256 agents are reset first so that some rows of the slice are in use
    e.reset(3)
    s16 = e.get_state(0, e.N)
    starts = {tuple(np.round(x, 9)) for x in np.c_[s16[:, 0:3], s16[:, 6:9]]}
    in_use_rows = [r for r in range(2048) if tuple(np.round(sg1[r], 9)) in starts]
    e.replan_begin(0, 2048, seed=77)
    torch.cuda.synchronize()
    assert e.replan_ready() == 1
    e.replan_commit()
    sg2, sub2, ns2 = e.bank_read()
    st2 = e.replan_stats()
    assert st2["refreshes"] == 2 and st2["rows_in_use"] >= 1 and st2["rows_in_use"] <= 256
    changed = np.any(sg2 != sg1, axis=1)
    assert changed.sum() == st2["rows_committed"] - st["rows_committed"] and changed.sum() > 1700
    for r in in_use_rows:                                     # flown rows kept their plan
        assert np.array_equal(sg2[r], sg1[r]) and np.array_equal(sub2[r], sub1[r]) and ns2[r] == ns1[r]
    assert ((ns2 >= 2) & (ns2 <= e.K)).all()
    for k in np.nonzero(changed)[0][:300]:
        p = sub2[k, :ns2[k]]
        assert np.array_equal(p[0], sg2[k, :3]) and np.array_equal(p[-1], sg2[k, 3:])       # start .. goal
        hop = np.linalg.norm(np.diff(p, axis=0), axis=1)
        assert hop.max() <= 30.0 + 1e-9
    # the agents that were flying go on undisturbed: their next steps match an env that never refreshed
    f = make_city26_env(256, bank="gpu", bank_size=2048, bank_seed=5)
    f.reset(3)
    oa, ob = e.alloc_out(), f.alloc_out()
    gen = torch.Generator(device="cuda").manual_seed(1)
    for t in range(40):
        a = torch.randint(0, 3, (e.N,), generator=gen, device="cuda", dtype=torch.int32)
        e.step(a, oa)
        f.step(a, ob)
        assert torch.equal(oa.reward, ob.reward) and torch.equal(oa.obs, ob.obs) and torch.equal(oa.info, ob.info), t
    # a slice outside the bank, and a second begin before the commit, are refused
    from dqn_based_uav_3d_path_planer_amd import _lib
    with pytest.raises(_lib.UavEnvError):
        e.replan_begin(2000, 100, seed=1)
    e.replan_begin(0, 64, seed=9)
    with pytest.raises(_lib.UavEnvError):
        e.replan_begin(64, 64, seed=9)
    torch.cuda.synchronize()
    e.replan_commit()
    e.close()
    f.close()


def test_loop_refreshes_the_bank_while_it_runs():
    """HotLoop(replan_every, replan_count): the C loop commits / starts a slice every replan_every passes on its own
    low-priority stream; the passes' results are those of the same loop on the bank contents it saw (checked loosely: the
    loop keeps training, the bank has turned over, no agent ever holds a broken list)."""
    from dqn_based_uav_3d_path_planer_amd.data import make_city26_env
    from dqn_based_uav_3d_path_planer_amd.learner import FusedDQNLearner
    from dqn_based_uav_3d_path_planer_amd.loop import HotLoop
    from dqn_based_uav_3d_path_planer_amd.replay import DeviceReplayRing
    n = 2048
    env = make_city26_env(n, bank="gpu", bank_size=4096, bank_seed=5, obs_dtype="packed")
    ring = DeviceReplayRing(env, 8 * n, discrete=True)
    ring.reset(seed=2)
    L = FusedDQNLearner({"w": "100", "hiden_dim": "64", "output": "3", "LEARNING_RATE": "0.001", "gamma": "0.99", "Update_loop": "3",
                         "NetWork": "Qnet2"}, "dqn", device="cuda:0")
    sg0, _, _ = env.bank_read()
    loop = HotLoop(ring, L, 2048, seed=4, eps=0.3, replan_every=16, replan_count=512)
    for _ in range(40):
        loop.run(64)
        torch.cuda.synchronize()
    st = env.replan_stats()
    sg1, sub1, ns1 = env.bank_read()
    assert st["refreshes"] >= 4 and st["rows_committed"] > 1000, st
    assert np.any(sg1 != sg0, axis=1).sum() > 1000 and ((ns1 >= 2) & (ns1 <= env.K)).all()
    assert torch.isfinite(L.flat).all() and L.epoch > 2000
    s16, sub, alias = env.get_state(0, n, want_sub=True)     # every agent's list still ends at its goal
    live = s16[:, 10] == 0
    for i in np.nonzero(live)[0][:400]:
        k = int(s16[i, 11])
        assert k >= 1 and np.array_equal(sub[i, k - 1], s16[i, 6:9]), i
    loop.close()
    env.close()
