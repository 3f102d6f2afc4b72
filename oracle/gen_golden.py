"""Generate tests/golden/*.npz by EXECUTING the reference on a scratch copy.

TEST INFRASTRUCTURE.  Runs only in the build container (needs /root/reference);
the outputs are committed so that the CPU tests, the GPU parity tests, smoke()
and bench.py never need the reference at run time.

    python oracle/gen_golden.py            # env-path goldens
    python oracle/gen_golden.py --learner  # trainer goldens (see gen_golden_learner.py)

What pins what (reference file:line):
  world_stock.npz      config/buildings.xml, config/UAV.xml, config/PathPlan_City.xml as parsed by the reference
  threaten_kat.npz     Envs/PathPlan_City.py:215-223 + Obstacles/building.py:20-26
  angle_kat.npz        BaseClass/CalMod.py:89-102, :64-65
  resets.npz           Agents/UAV.py:327-366 + PathPlan/RRT.py:63-105 under random.seed(k)
  episodes.npz         Agents/UAV.py:397-567 whole episodes from reset, scripted actions
  injected.npz         Agents/UAV.py:397-567 single steps from injected states (all branches)
  apf.npz              Agents/UAV.py:156-210,448-453 with building.v injected
  apf_episodes.npz     the same, WHOLE episodes from uav.reset() with APF_Enabled = 1: every step's outputs, state and the
                       full sub-goal list (Adjust_subgoal shifts every sub-goal every step: accumulated drift is compared)
"""
from __future__ import annotations

import math
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "tests", "golden")
sys.path.insert(0, HERE)

from ref_harness import RefSession  # noqa: E402

INFO = {"normal": 0, "success": 1, "lose": 2}


def f(x):
    return float(x)


def uav_state_vec(uav):
    """[px,py,pz,vx,vy,V,gx,gy,gz,Step,done,n_sub,score,total_score,path_len,reach_goal]"""
    return [f(uav.position.x), f(uav.position.y), f(uav.position.z), f(uav.V_vector.x), f(uav.V_vector.y),
            f(uav.V), f(uav.goal.x), f(uav.goal.y), f(uav.goal.z), f(uav.Step), f(uav.done),
            f(len(uav.sub_goals)), f(uav.score), f(uav.total_score), f(uav.path_len), f(uav.reach_goal)]


def subgoals_arr(uav, kmax):
    a = np.zeros((kmax, 3))
    for k, g in enumerate(uav.sub_goals):
        a[k] = (f(g.x), f(g.y), f(g.z))
    return a


def gen_world(s, out):
    env, uav = s.env, s.uav
    b = np.array([[t.position.x, t.position.y, t.position.z, t._R, t._H] for t in env.buildings], dtype=np.float64)
    uav.V_vector = s.CalMod.Loc(0.6, 0.3, 0)
    p_fly_sub = uav.Calc_Fly_Power()
    uav.V_vector = s.CalMod.Loc(0.9, 0.8, 0)
    p_fly_clamped = uav.Calc_Fly_Power()
    np.savez(os.path.join(out, "world_stock.npz"),
             buildings=b, len=float(env.len), width=float(env.width), h=float(env.h),
             max_v=float(uav.Max_V), steering_angle=float(uav.Steering_angle), max_step=int(uav.Max_Step),
             sub_granularity=float(uav.sub_granularity), apf_enabled=int(uav.APF_Enabled),
             power=np.array([uav.P_i, uav.v_0, uav.d_0, uav.rho, uav.s, uav.A, uav.P_b, uav.F_b, uav.xi]),
             p_fly_v_06_03=float(p_fly_sub), p_fly_v_09_08=float(p_fly_clamped))
    return b


def gen_threaten(s, b, out):
    env, Loc = s.env, s.CalMod.Loc
    rng = random.Random(1234)
    pts = [
        (285.3311642197549, 454.6501406344541, 0), (285.33, 454.65, 14.12), (285.33, 454.65, 14.13),
        (322.21, 454.65, 0), (322.23, 454.65, 0), (0, 0, 0), (-0.0, 5, 0), (500, 500, 100),
        (500.0001, 10, 0), (10, 500.0001, 0), (250, 250, -1e-9), (250, 250, 100.0001),
        (129.24, 197.43, 31.1), (129.24, 197.43, 31.2),
    ]
    for _ in range(12000):   # uniform over a box slightly larger than the world
        pts.append((rng.uniform(-20, 520), rng.uniform(-20, 520), rng.uniform(-5, 105)))
    for i in range(len(b)):  # rim and roof of every cylinder, +-1e-9 .. +-1e-13 relative
        cx, cy, cz, R, H = b[i]
        for _ in range(120):
            th = rng.uniform(0, 2 * math.pi)
            eps = rng.choice([0.0, 1e-13, -1e-13, 1e-9, -1e-9, 1e-6, -1e-6, 1e-3, -1e-3])
            rr = R * (1 + eps)
            z = rng.choice([0.0, H, H * (1 + 1e-15), H * (1 - 1e-15), H + 1e-9, H - 1e-9, rng.uniform(0, 60)])
            pts.append((cx + rr * math.cos(th), cy + rr * math.sin(th), z))
    for _ in range(500):     # box faces
        e = rng.choice([0.0, 500.0, 500.00000000000006, -5e-324, 499.99999999999994])
        pts.append((e, rng.uniform(0, 500), rng.uniform(0, 100)))
        pts.append((rng.uniform(0, 500), e, rng.uniform(0, 100)))
        ez = rng.choice([0.0, 100.0, 100.00000000000001, -5e-324])
        pts.append((rng.uniform(0, 500), rng.uniform(0, 500), ez))
    pts = np.array(pts, dtype=np.float64)
    exp = np.array([env.Threaten_rate(Loc(float(p[0]), float(p[1]), float(p[2]))) for p in pts], dtype=np.int32)
    np.savez_compressed(os.path.join(out, "threaten_kat.npz"), points=pts, expected=exp)
    print("threaten_kat", len(pts), "hits", int(exp.sum()))


def gen_angles(s, out):
    cm, Loc = s.CalMod, s.CalMod.Loc
    rng = random.Random(99)
    pairs = [(0, 0, 1, 0), (0, 0, 0, 1), (0, 0, -1, 0), (0, 0, 0, -1), (0, 0, -1, -1), (0, 0, 0, 0),
             (0, 0, 1, -1e-12), (0, 0, 1, -1e-300), (0, 0, 1, -0.0), (0, 0, -1, -0.0), (0, 0, 1, 1e-17)]
    for _ in range(4000):
        pairs.append((rng.uniform(-500, 500), rng.uniform(-500, 500), rng.uniform(-500, 500), rng.uniform(-500, 500)))
    for _ in range(1000):
        th = rng.uniform(0, 2 * math.pi)
        pairs.append((0.0, 0.0, math.cos(th), math.sin(th)))
    pairs = np.array(pairs, dtype=np.float64)
    ang = np.array([cm.calculate_angle(Loc(float(p[0]), float(p[1]), 0), Loc(float(p[2]), float(p[3]), 0)) for p in pairs])
    dist = np.array([cm.Eu_Loc_distance(Loc(float(p[0]), float(p[1]), 3.25), Loc(float(p[2]), float(p[3]), -1.5)) for p in pairs])
    np.savez_compressed(os.path.join(out, "angle_kat.npz"), pairs=pairs, angle=ang, dist_z=dist)


def gen_resets(s, out, seeds, kmax=128):
    uav = s.uav
    rows, subs, nsub = [], [], []
    for k in seeds:
        random.seed(k)
        uav.reset()
        rows.append(uav_state_vec(uav) + [f(uav.V_dir), f(uav.start2goal), f(uav.len_Astar)])
        subs.append(subgoals_arr(uav, kmax))
        nsub.append(len(uav.sub_goals))
    np.savez_compressed(os.path.join(out, "resets.npz"), seeds=np.array(seeds, dtype=np.int64),
                        state=np.array(rows), sub_goals=np.array(subs), n_sub=np.array(nsub, dtype=np.int32))
    print("resets", len(seeds), "n_sub min/mean/max", min(nsub), sum(nsub) / len(nsub), max(nsub))


def seek_action(uav, cm, noise):
    if len(uav.sub_goals) == 0:
        return 0.0
    head = cm.calculate_angle(cm.Loc(0, 0, 0), uav.V_vector)
    want = cm.calculate_angle(uav.position, uav.sub_goals[0])
    d = (want - head + math.pi) % (2 * math.pi) - math.pi
    a = d / uav.Steering_angle + noise
    return max(-1.0, min(1.0, a))


def run_episode(s, seed, policy, kmax, max_steps=4000):
    """Whole episode from reset; returns dict of arrays.  policy in {random, seek, mixed}."""
    uav, cm = s.uav, s.CalMod
    random.seed(seed)
    uav.reset()
    init = uav_state_vec(uav) + [f(uav.V_dir)]
    sub0 = subgoals_arr(uav, kmax)
    obs0 = np.array(uav.state(), dtype=np.float64)
    prng = random.Random(seed * 7919 + 13)   # action stream independent of the reference's global RNG
    acts, outs, states, obss = [], [], [], []
    for t in range(max_steps):
        if policy == "random":
            a = prng.uniform(-1, 1)
        elif policy == "seek":
            a = seek_action(uav, cm, prng.gauss(0, 0.05))
        else:
            a = seek_action(uav, cm, prng.gauss(0, 0.3)) if prng.random() < 0.7 else prng.uniform(-1, 1)
        r, d, info = uav.update([a, 0.0])
        o = uav.state()
        acts.append(a)
        outs.append([f(r), f(d), f(uav.done), f(INFO[info])])
        states.append(uav_state_vec(uav))
        obss.append(np.array(o, dtype=np.float64))
        if uav.done:
            break
    return dict(init=np.array(init), sub_goals=sub0, obs0=obs0, actions=np.array(acts), outs=np.array(outs),
                states=np.array(states), obs=np.array(obss))


def gen_episodes(s, out, kmax=128):
    # seeds 105/110/114/131 reach the final goal under the seeking policy (scanned 100..139); the rest time out
    plan = [(5, "random"), (11, "random"), (23, "random"), (101, "seek"), (105, "seek"), (110, "seek"),
            (114, "seek"), (131, "seek"), (201, "mixed"), (202, "mixed"), (203, "mixed"), (204, "mixed")]
    eps = [run_episode(s, seed, pol, kmax) for seed, pol in plan]
    offs = np.cumsum([0] + [len(e["actions"]) for e in eps])
    counts = {}
    for e in eps:
        for row in e["outs"]:
            counts[int(row[3])] = counts.get(int(row[3]), 0) + 1
    np.savez_compressed(
        os.path.join(out, "episodes.npz"),
        seeds=np.array([p[0] for p in plan], dtype=np.int64), offsets=offs.astype(np.int64),
        init=np.array([e["init"] for e in eps]), sub_goals=np.array([e["sub_goals"] for e in eps]),
        obs0=np.array([e["obs0"] for e in eps]),
        actions=np.concatenate([e["actions"] for e in eps]), outs=np.concatenate([e["outs"] for e in eps]),
        states=np.concatenate([e["states"] for e in eps]), obs=np.concatenate([e["obs"] for e in eps]))
    print("episodes", len(eps), "steps", int(offs[-1]), "info counts", counts,
          "final infos", [int(e["outs"][-1][3]) for e in eps], "reach", [int(e["states"][-1][15]) for e in eps])


def inject(s, st, subs, apf=None):
    """st = [px,py,pz,vx,vy,gx,gy,gz,Step]"""
    uav, Loc = s.uav, s.CalMod.Loc
    uav.position = Loc(st[0], st[1], st[2])
    uav.V_vector = Loc(st[3], st[4], 0)
    uav.V = uav.Calc_V()
    uav.goal = Loc(st[5], st[6], st[7])
    uav.sub_goals = [Loc(float(g[0]), float(g[1]), float(g[2])) for g in subs]
    uav.Step = int(st[8])
    uav.done = False
    uav.score = 0
    uav.total_score = 0
    uav.path_len = 0
    uav.reach_goal = 0
    uav.path = []
    uav.V_record = []
    uav.R_record = []
    if apf is not None:
        uav.APF_Enabled = apf


def gen_injected(s, out, n, kmax=8):
    uav = s.uav
    rng = random.Random(777)
    b = np.array([[t.position.x, t.position.y, t._R] for t in s.env.buildings])
    ins, subs_all, nsubs, acts, outs, states, obss, subs_after = [], [], [], [], [], [], [], []
    for c in range(n):
        kind = c % 10
        px, py = rng.uniform(1, 499), rng.uniform(1, 499)
        pz = 0.0 if kind < 7 else rng.uniform(0, 60)
        if kind == 1:   # hug a building rim so the move collides or nearly collides
            i = rng.randrange(len(b))
            th = rng.uniform(0, 2 * math.pi)
            rr = b[i][2] + rng.uniform(0.0, 1.2)
            px, py = b[i][0] + rr * math.cos(th), b[i][1] + rr * math.sin(th)
        if kind == 2:   # near the world edge
            px = rng.choice([rng.uniform(0, 1.5), rng.uniform(498.5, 500)])
        th = rng.uniform(0, 2 * math.pi)
        speed = 1.0 if kind != 3 else rng.uniform(0.2, 1.7)     # kind 3: un-normalised velocity (Calc_V clamp)
        vx, vy = speed * math.cos(th), speed * math.sin(th)
        gx, gy, gz = rng.uniform(330, 490), rng.uniform(420, 490), 0.0
        nsub = rng.choice([0, 1, 1, 2, 3, 5]) if kind != 4 else 0
        subs = []
        for k in range(nsub):
            if k == 0 and kind in (5, 6):       # sub-goal within reach -> pop branches
                d = rng.uniform(0, 9)
                a = rng.uniform(0, 2 * math.pi)
                subs.append((px + d * math.cos(a), py + d * math.sin(a), rng.uniform(0, 3)))
            else:
                subs.append((rng.uniform(0, 500), rng.uniform(0, 500), rng.uniform(0, 99)))
        if kind == 9 and nsub >= 1:             # goal within 7 m but sub-goal nearer the goal than we are
            gx, gy, gz = px + rng.uniform(-4, 4), py + rng.uniform(-4, 4), 0.0
            subs[0] = (gx + rng.uniform(-1, 1), gy + rng.uniform(-1, 1), pz)
        if kind == 8 and nsub >= 1:             # final-goal branch: far sub-goal, close goal
            gx, gy, gz = px + rng.uniform(-4, 4), py + rng.uniform(-4, 4), pz
            far = rng.uniform(0, 2 * math.pi)
            subs[0] = (gx + 300 * math.cos(far), gy + 300 * math.sin(far), 0.0)
        step = rng.choice([0, 1, 17, 100, 148, 149, 150]) if kind != 7 else 149
        a0 = rng.uniform(-1, 1) if rng.random() < 0.9 else rng.choice([-1.0, 0.0, 1.0])
        st = [px, py, pz, vx, vy, gx, gy, gz, step]
        inject(s, st, subs)
        r, d, info = uav.update([a0, 0.0])
        o = uav.state()
        ins.append(st)
        sa = np.zeros((kmax, 3))
        sa[:nsub] = np.array(subs).reshape(-1, 3) if nsub else 0
        subs_all.append(sa)
        nsubs.append(nsub)
        acts.append(a0)
        outs.append([f(r), f(d), f(uav.done), f(INFO[info])])
        states.append(uav_state_vec(uav))
        obss.append(np.array(o, dtype=np.float64))
        subs_after.append(subgoals_arr(uav, kmax))
    outs = np.array(outs)
    print("injected", n, "info counts", {k: int((outs[:, 3] == k).sum()) for k in (0, 1, 2)},
          "agent_done", int(outs[:, 2].sum()), "ret_done", int(outs[:, 1].sum()))
    np.savez_compressed(os.path.join(out, "injected.npz"), inputs=np.array(ins), sub_goals=np.array(subs_all),
                        n_sub=np.array(nsubs, dtype=np.int32), actions=np.array(acts), outs=outs,
                        states=np.array(states), obs=np.array(obss), sub_goals_after=np.array(subs_after))


def gen_apf(s, out, n, kmax=8):
    """APF on: buildings get a velocity attribute (the stock class has none -> AttributeError)."""
    uav, env, Loc = s.uav, s.env, s.CalMod.Loc
    rng = random.Random(4242)
    nb = len(env.buildings)
    vel = np.zeros((nb, 3))
    for i in range(nb):
        if i % 3 != 0:   # a third of the buildings stay static (skipped by :180-182)
            vel[i] = (rng.uniform(-1, 1), rng.uniform(-1, 1), 0.0)
    for i, t in enumerate(env.buildings):
        t.v = Loc(float(vel[i][0]), float(vel[i][1]), float(vel[i][2]))
    ins, subs_all, nsubs, acts, outs, states, obss, subs_after = [], [], [], [], [], [], [], []
    tries = 0
    while len(ins) < n and tries < 20 * n:
        tries += 1
        px, py, pz = rng.uniform(1, 499), rng.uniform(1, 499), 0.0
        th = rng.uniform(0, 2 * math.pi)
        gx, gy, gz = rng.uniform(330, 490), rng.uniform(420, 490), 0.0
        nsub = rng.choice([1, 2, 3, 5])
        subs = [(rng.uniform(0, 500), rng.uniform(0, 500), rng.uniform(0, 99)) for _ in range(nsub)]
        step = rng.choice([0, 5, 100, 149])
        a0 = rng.uniform(-1, 1)
        st = [px, py, pz, math.cos(th), math.sin(th), gx, gy, gz, step]
        inject(s, st, subs, apf=1)
        try:
            r, d, info = uav.update([a0, 0.0])
        except TypeError:
            continue   # cum_force > 100 -> Cal_SubTask_Dynamic() raises in the reference (UAV.py:205-208)
        o = uav.state()
        ins.append(st)
        sa = np.zeros((kmax, 3))
        sa[:nsub] = np.array(subs)
        subs_all.append(sa)
        nsubs.append(nsub)
        acts.append(a0)
        outs.append([f(r), f(d), f(uav.done), f(INFO[info])])
        states.append(uav_state_vec(uav))
        obss.append(np.array(o, dtype=np.float64))
        subs_after.append(subgoals_arr(uav, kmax))
    uav.APF_Enabled = 0
    for t in env.buildings:
        del t.v
    outs = np.array(outs)
    print("apf", len(ins), "of", tries, "tries; info counts", {k: int((outs[:, 3] == k).sum()) for k in (0, 1, 2)})
    np.savez_compressed(os.path.join(out, "apf.npz"), velocities=vel, inputs=np.array(ins),
                        sub_goals=np.array(subs_all), n_sub=np.array(nsubs, dtype=np.int32),
                        actions=np.array(acts), outs=outs, states=np.array(states), obs=np.array(obss),
                        sub_goals_after=np.array(subs_after))


def run_apf_episode(s, seed, policy, kmax):
    """Whole episode from uav.reset() with APF on (env.buildings already carry .v).  Returns None when the reference raises at
    UAV.py:205-208 (cum_force > 100 -> Cal_SubTask_Dynamic() without its arguments -> TypeError)."""
    uav, cm = s.uav, s.CalMod
    random.seed(seed)
    uav.reset()
    if len(uav.sub_goals) > kmax:
        return None
    init = uav_state_vec(uav) + [f(uav.V_dir)]
    sub0 = subgoals_arr(uav, kmax)
    alias0 = int(len(uav.sub_goals) > 0 and uav.sub_goals[0] is uav.position)
    obs0 = np.array(uav.state(), dtype=np.float64)
    prng = random.Random(seed * 104729 + 71)
    acts, outs, states, obss, subs, nsubs = [], [], [], [], [], []
    for t in range(4000):
        if policy == "random":
            a = prng.uniform(-1, 1)
        elif policy == "seek":
            a = seek_action(uav, cm, prng.gauss(0, 0.05))
        else:
            a = seek_action(uav, cm, prng.gauss(0, 0.3)) if prng.random() < 0.7 else prng.uniform(-1, 1)
        try:
            r, d, info = uav.update([a, 0.0])
        except TypeError:
            return None
        o = uav.state()
        acts.append(a)
        outs.append([f(r), f(d), f(uav.done), f(INFO[info])])
        states.append(uav_state_vec(uav))
        obss.append(np.array(o, dtype=np.float64))
        subs.append(subgoals_arr(uav, kmax))
        nsubs.append(len(uav.sub_goals))
        if uav.done:
            break
    return dict(init=np.array(init), sub_goals=sub0, alias0=alias0, obs0=obs0, actions=np.array(acts), outs=np.array(outs),
                states=np.array(states), obs=np.array(obss), subs=np.array(subs), nsubs=np.array(nsubs, dtype=np.int32))


def gen_apf_episodes(s, out, kmax=48):
    """APF-on trajectories WITHOUT re-synchronisation.  Every building that gen_apf moves moves here too (the same
    seeded velocities); whole episodes from uav.reset(); seeds where the reference raises (UAV.py:205-208) are skipped and listed."""
    uav, env, Loc = s.uav, s.env, s.CalMod.Loc
    rng = random.Random(4242)
    nb = len(env.buildings)
    vel = np.zeros((nb, 3))
    for i in range(nb):
        if i % 3 != 0:
            vel[i] = (rng.uniform(-1, 1), rng.uniform(-1, 1), 0.0)
    for i, t in enumerate(env.buildings):
        t.v = Loc(float(vel[i][0]), float(vel[i][1]), float(vel[i][2]))
    uav.APF_Enabled = 1
    plan = ([(k, "seek") for k in range(300, 324)] + [(k, "random") for k in (5, 11, 23)] + [(k, "mixed") for k in (201, 202, 203, 204, 205)])
    eps, used, skipped = [], [], []
    n_success = n_lose = 0
    for seed, pol in plan:
        e = run_apf_episode(s, seed, pol, kmax)
        if e is None:
            skipped.append(seed)
            continue
        final = int(e["outs"][-1][3])
        # keep every random / mixed episode; of the seeking ones at most six successes and four time-outs
        if pol == "seek":
            if final == 1 and n_success >= 6:
                continue
            if final != 1 and n_lose >= 4:
                continue
        n_success += final == 1
        n_lose += final != 1
        eps.append(e)
        used.append((seed, pol))
    uav.APF_Enabled = 0
    for t in env.buildings:
        del t.v
    offs = np.cumsum([0] + [len(e["actions"]) for e in eps])
    pol_code = {"random": 0, "seek": 1, "mixed": 2}
    np.savez_compressed(
        os.path.join(out, "apf_episodes.npz"), velocities=vel,
        seeds=np.array([u[0] for u in used], dtype=np.int64), policy=np.array([pol_code[u[1]] for u in used], dtype=np.int32),
        skipped_seeds=np.array(skipped, dtype=np.int64), offsets=offs.astype(np.int64),
        init=np.array([e["init"] for e in eps]), sub_goals=np.array([e["sub_goals"] for e in eps]),
        alias0=np.array([e["alias0"] for e in eps], dtype=np.int32), obs0=np.array([e["obs0"] for e in eps]),
        actions=np.concatenate([e["actions"] for e in eps]), outs=np.concatenate([e["outs"] for e in eps]),
        states=np.concatenate([e["states"] for e in eps]), obs=np.concatenate([e["obs"] for e in eps]),
        subs=np.concatenate([e["subs"] for e in eps]), nsubs=np.concatenate([e["nsubs"] for e in eps]))
    print("apf_episodes", len(eps), "steps", int(offs[-1]), "final infos", [int(e["outs"][-1][3]) for e in eps],
          "lens", [len(e["actions"]) for e in eps], "skipped seeds (reference raised)", skipped)


def main():
    os.makedirs(OUT, exist_ok=True)
    s = RefSession()
    try:
        b = gen_world(s, OUT)
        gen_threaten(s, b, OUT)
        gen_angles(s, OUT)
        gen_resets(s, OUT, seeds=list(range(1, 41)) + [42, 1000, 2 ** 32 + 5, 2 ** 40 + 123])
        gen_episodes(s, OUT)
        gen_injected(s, OUT, 3000)
        gen_apf(s, OUT, 600)
        gen_apf_episodes(s, OUT)
    finally:
        s.close()


if __name__ == "__main__":
    main()
