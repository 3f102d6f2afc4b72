"""Per-phase timeline of k_dqn_grad from in-kernel s_memtime stamps (needs a UAVENV_PHASE_PROFILE build)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dqn_based_uav_3d_path_planer_amd.data import make_city26_env
from dqn_based_uav_3d_path_planer_amd.learner import FusedDQNLearner
from dqn_based_uav_3d_path_planer_amd.replay import DeviceReplayRing

n = B = int(os.environ.get("B", "16384"))
OBS = os.environ.get("OBS", "f32")
MFMA, KIND = os.environ.get("MFMA", "f32"), os.environ.get("KIND", "dqn")
env = make_city26_env(n, obs_dtype="packed" if OBS == "packed" else (torch.float16 if OBS == "f16" else torch.float32))
ring = DeviceReplayRing(env, 1 << 20)
ring.reset(seed=1)
L = FusedDQNLearner({"NetWork": "VAnet2" if KIND == "dueling" else "Qnet2", "w": "100", "hiden_dim": "64", "output": "3"}, KIND, mfma=MFMA)
for t in range(30):
    L.act(ring.current_obs(), 0.1, 1, t, index_out=ring.current_action())
    ring.step_env(auto_reset=True)
nb = min(B // 64, 256)
buf = torch.zeros(2 * nb * 8, dtype=torch.int64, device="cuda")      # two stamp banks (csrc/learner.hip: L_STAMP2 writes the second)
env.lib.uavenv_dqn_set_debug_buffer(buf.data_ptr())
rows = []
for t in range(20):
    L.learn_from_ring(ring, B, 3, t)
    torch.cuda.synchronize()
    rows.append(buf.cpu().numpy()[:nb * 8].reshape(nb, 8).astype(np.float64))
env.lib.uavenv_dqn_set_debug_buffer(None)
R = np.stack(rows)
print("h8 kernels: tile start -> rows committed:", (R[:, :, 7] - R[:, :, 6]).mean(), " -> next tile issued:", (R[:, :, 1] - R[:, :, 7]).mean(), " prev tile end -> tile start (sync):", (R[:, :, 6] - R[:, :, 5]).mean())
print("fwd_strip local (MFMA only):", (R[:, :, 6] - R[:, :, 1]).mean(), " fwd_strip target (MFMA only):", (R[:, :, 7] - R[:, :, 2]).mean())
d = np.diff(R[:, :, :6], axis=2)
names = ["draw, issue loads, commit s rows + local fc1", "forward q_local(s) + commit s' rows, target fc1",
         "forward q_target(s') [+ q_local(s')]", "TD target, dL/dH, H / dH / dout -> LDS", "dW1^T, dW2^T products (MFMA)"]
print("obs", OBS, "k_dqn_grad, batch", B, "- cycles per workgroup (mean / p95 / max)")
for k, nm in enumerate(names):
    x = d[:, :, k].ravel()
    print(f"  {nm:36s} {x.mean():9.0f} {np.percentile(x, 95):9.0f} {x.max():9.0f}")
tot = d.sum(axis=2).ravel()
print(f"  {'total':36s} {tot.mean():9.0f} {np.percentile(tot, 95):9.0f} {tot.max():9.0f}")
