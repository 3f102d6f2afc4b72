"""Per-phase timeline of k_dqn_grad_packed8 AS THE C LOOP LAUNCHES IT (layer 1 staged from the loop's image, csrc/dqn_internal.hpp)
from in-kernel s_memtime stamps (needs a UAVENV_PHASE_PROFILE build).  python scripts/phase_profile_loop.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dqn_based_uav_3d_path_planer_amd.data import make_city26_env
from dqn_based_uav_3d_path_planer_amd.learner import FusedDQNLearner
from dqn_based_uav_3d_path_planer_amd.loop import HotLoop
from dqn_based_uav_3d_path_planer_amd.replay import DeviceReplayRing

n = B = int(os.environ.get("B", "16384"))
env = make_city26_env(n, obs_dtype="packed")
ring = DeviceReplayRing(env, 1 << 20)
ring.reset(seed=1)
L = FusedDQNLearner({"NetWork": "Qnet2", "w": "100", "hiden_dim": "64", "output": "3"}, "dqn")
hot = HotLoop(ring, L, B, seed=3, eps=0.1)
hot.run(40)
torch.cuda.synchronize()
nb = min(B // 64, 256)
buf = torch.zeros(2 * nb * 8, dtype=torch.int64, device="cuda")
env.lib.uavenv_dqn_set_debug_buffer(buf.data_ptr())
rows = []
for t in range(20):
    hot.run(1)
    torch.cuda.synchronize()
    rows.append(buf.cpu().numpy().reshape(2, nb, 8).astype(np.float64))
env.lib.uavenv_dqn_set_debug_buffer(None)
RR = np.stack(rows)
R, R2 = RR[:, 0], RR[:, 1]
print("  start -> weight loads issued: %.0f;  -> draw done: %.0f;  -> rows / scalars issued: %.0f" % ((R2[:, :, 0] - R[:, :, 0]).mean(), (R2[:, :, 1] - R2[:, :, 0]).mean(), (R[:, :, 6] - R2[:, :, 1]).mean()))
order = [0, 6, 7, 1, 2, 3, 4, 5]
names = ["start -> all loads issued (args, draw, addresses)", "-> layer-1 image in LDS (its loads arrived)", "-> W2 / b2 stored + barrier",
         "both forwards + layer 2", "hand-over barrier", "TD target, dL/dH -> LDS + barrier", "weight-gradient products"]
T = R[:, :, order]
d = np.diff(T, axis=2)
print("k_dqn_grad_packed8 in the C loop, batch", B, "- cycles per workgroup (mean / p95 / max)")
for k, nm in enumerate(names):
    x = d[:, :, k].ravel()
    print(f"  {nm:52s} {x.mean():9.0f} {np.percentile(x, 95):9.0f} {x.max():9.0f}")
tot = (R[:, :, 5] - R[:, :, 0]).ravel()
print(f"  {'total (stamp 0 -> 5)':52s} {tot.mean():9.0f} {np.percentile(tot, 95):9.0f} {tot.max():9.0f}")
span = R[:, :, 5].max(axis=1) - R[:, :, 0].min(axis=1)
first = R[:, :, 0].max(axis=1) - R[:, :, 0].min(axis=1)
print(f"  launch span (first stamp 0 -> last stamp 5): {span.mean():.0f} cycles; spread of the workgroups' start stamps: {first.mean():.0f}")
hot.close()
env.close()
