"""Per-phase timeline of k_sac_critic_grad from in-kernel s_memtime stamps (diagnostic, UAVENV_PHASE_PROFILE=1 build)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dqn_based_uav_3d_path_planer_amd.data import make_city26_env
from dqn_based_uav_3d_path_planer_amd.replay import DeviceReplayRing
from dqn_based_uav_3d_path_planer_amd.sac import FusedSACLearner

P = {"actor": {"NetWork": "PolicyNetContinuous_SAC", "w": "100", "action_bound": "1", "hiden_dim": "64", "output": "2", "lr": "0.0001"},
     "critic": {"NetWork": "QValueNetContinuous_SAC", "w": "100", "hiden_dim": "64", "action_dim": "2", "lr": "0.001"},
     "SAC_param": {"IS_Continuous": "1", "alpha_lr": "0.0001", "target_entropy": "1", "gamma": "0.99", "tau": "0.05"}}
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
env = make_city26_env(8192, uav_per_env=4, obs_dtype="packed")
ring = DeviceReplayRing(env, 20 * env.N, discrete=False)
ring.reset(seed=1)
a1 = torch.zeros((ring.frames, env.N), device="cuda")
for _ in range(12):
    ring.current_action().uniform_(-1, 1)
    ring.step_env(auto_reset=True)
L = FusedSACLearner(P)
f = torch.randint(0, ring.head - 1, (B,), device="cuda", dtype=torch.int32)
e = torch.randint(0, env.N // 4, (B,), device="cuda", dtype=torch.int32)
draws = torch.stack([f, e], 1).contiguous()
flat = ring.obs.view(-1, ring.obs.shape[-1])
b = L.make_batch(flat, ring.action.view(-1), a1.view(-1), ring.reward.view(-1), ring.done.view(-1), valid=ring.valid.view(-1),
                 draws=draws, n_agents=env.N, uav_per_env=4, slot=1, frames=ring.frames)
rows = env.lib.uavenv_sac_partial_rows(B)
buf = torch.zeros(rows * 16, dtype=torch.int64, device="cuda")
for _ in range(3):
    L.learn(b)
env.lib.uavenv_sac_set_debug_buffer(buf.data_ptr())
R = []
for _ in range(10):
    L.learn(b)
    torch.cuda.synchronize()
    R.append(buf.cpu().numpy().reshape(rows, 16).astype(np.float64))
env.lib.uavenv_sac_set_debug_buffer(None)
R = np.stack(R)
names = {(0, 1): "stage actor + 2 target critics (+ first rows)", (1, 2): "stage I: all tiles (actor, Qt1, Qt2 fwd)",
         (2, 3): "stage critic 1", (3, 4): "c1 tile 0: fwd + bwd + LDS", (4, 5): "c1 tile 0: dW1 products (packed decode)",
         (5, 6): "c1 tile 0: dW2 / dWout / db2 products", (6, 7): "c1: remaining tiles", (7, 8): "c1: partial row write-out",
         (8, 9): "stage critic 2", (9, 10): "c2 tile 0: fwd + bwd + LDS", (10, 11): "c2 tile 0: dW1 products",
         (11, 12): "c2 tile 0: other products", (12, 13): "c2: remaining tiles", (13, 14): "c2: write-out"}
print(f"k_sac_critic_grad, batch {B}, {rows} workgroups: s_memtime ticks per phase (mean / p95)")
for (a, c), nm in names.items():
    x = (R[:, :, c] - R[:, :, a]).ravel()
    print(f"  {nm:48s} {x.mean():9.0f} {np.percentile(x, 95):9.0f}")
tot = (R[:, :, 14] - R[:, :, 0]).ravel()
print(f"  {'total':48s} {tot.mean():9.0f} {np.percentile(tot, 95):9.0f}")
