"""Generate the committed scenario bank dqn_based_uav_3d_path_planer_amd/data/city26.npz.

A scenario is what UAV.reset() (Agents/UAV.py:327-366) draws besides the heading: start, goal and the RRT
sub-goal list (PathPlan/RRT.py:63-105).  Scenario k is exactly the reference's reset under random.seed(k)
(the C oracle reproduces it bit-for-bit, tests/test_oracle_golden.py::test_reset_and_rrt_golden).
The product only READS the resulting data file; it never links the oracle.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
from oracle import pyoracle as po  # noqa: E402

K = 48
M = int(sys.argv[1]) if len(sys.argv) > 1 else 1024


def main():
    w = np.load(os.path.join(HERE, "..", "tests", "golden", "world_stock.npz"))
    world = po.OracleWorld(w["buildings"], w["len"], w["width"], w["h"])
    params = po.default_uav_params(w)
    sg, subs, nsub, seeds = [], [], [], []
    seed = 0
    while len(sg) < M:
        seed += 1
        u = po.OracleUav(world, params)
        u.reset(po.OracleRng(seed), float(w["sub_granularity"]))
        n = u.u.n_sub
        if u.u.error or n < 2 or n > K:
            continue
        s = np.zeros((K, 3))
        s[:n] = u.sub_goals()
        sg.append([u.u.px, u.u.py, u.u.pz, u.u.gx, u.u.gy, u.u.gz])
        subs.append(s)
        nsub.append(n)
        seeds.append(seed)
    out = os.path.join(HERE, "..", "dqn_based_uav_3d_path_planer_amd", "data", "city26.npz")
    np.savez_compressed(out, buildings=w["buildings"], len=w["len"], width=w["width"], h=w["h"],
                        max_v=w["max_v"], steering_angle=w["steering_angle"], max_step=w["max_step"],
                        power=w["power"], start_goal=np.array(sg), sub_goals=np.array(subs),
                        n_sub=np.array(nsub, dtype=np.int32), seeds=np.array(seeds, dtype=np.int64))
    print(out, len(sg), "scenarios; n_sub min/mean/max", min(nsub), np.mean(nsub), max(nsub),
          os.path.getsize(out) / 1e6, "MB")


if __name__ == "__main__":
    main()
