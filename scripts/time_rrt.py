"""Time the GPU planner (csrc/rrt.hip): m scenarios, the two LDS tiers; and the LDS-free background form through the refresh API."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

from dqn_based_uav_3d_path_planer_amd.data import make_city26_env

env = make_city26_env(64, bank="packaged")
env.rrt_plan(1024, seed=1)
torch.cuda.synchronize()
for m in (4096, 16384, 65536):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    sg, sub, ns, it = env.rrt_plan(m, seed=100 + m)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    ok = ((ns >= 2) & (ns <= env.K)).float().mean().item()
    print("rrt_plan m=%d: %.2f ms, %.0f rows/s, usable %.4f, iters mean %.1f max %d" % (m, ms, m / ms * 1e3, ok, it.float().mean().item(), int(it.max())))
env.plan_scenarios(16384, seed=3)
for wgs in (128,):
    t0 = time.perf_counter()
    env.replan_begin(0, 4096, seed=9)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    env.replan_commit(force=True)
    torch.cuda.synchronize()
    print("background form (%d wavefronts, alone on the GPU): 4096 rows in %.1f ms = %.0f rows/s" % (wgs, dt * 1e3, 4096 / dt), env.replan_stats())
env.close()
