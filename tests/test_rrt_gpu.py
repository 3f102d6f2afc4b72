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
