"""Per-wavefront phase timeline of k_step_coop from in-kernel s_memtime stamps (diagnostic, -DUAVENV_PHASE_PROFILE).
python scripts/phase_profile_coop.py [envs]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dqn_based_uav_3d_path_planer_amd.data import make_city26_env
from dqn_based_uav_3d_path_planer_amd.replay import DeviceReplayRing

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
POLICY = os.environ.get("POLICY") in ("1", "2")      # 2: a -DUAVENV_PHASE_POLICY build, the stages of the policy prologue         # the launch the C loop issues: get_action in the step kernel's prologue
env = make_city26_env(n, obs_dtype="packed" if POLICY else torch.float32)
ring = DeviceReplayRing(env, 8 * n)
if POLICY:
    from dqn_based_uav_3d_path_planer_amd.learner import FusedDQNLearner
    L = FusedDQNLearner({"NetWork": "Qnet2", "w": "100", "hiden_dim": "64", "output": "3"}, "dqn")
ring.reset(seed=1)
gen = torch.Generator(device="cuda").manual_seed(0)
for _ in range(300):
    ring.current_action().copy_(torch.randint(0, 3, (n,), generator=gen, device="cuda", dtype=torch.int32))
    ring.step_env(auto_reset=True)
nb = (n + 63) // 64
buf = torch.zeros(nb * 4 * 8, dtype=torch.int64, device="cuda")
env.lib.uavenv_set_debug_buffer(env._h, buf.data_ptr())
rows = []
for it in range(20):
    if POLICY:
        assert ring.step_policy(L, 0.1, 3, 1000 + it)
    else:
        ring.current_action().copy_(torch.randint(0, 3, (n,), generator=gen, device="cuda", dtype=torch.int32))
        ring.step_env(auto_reset=True)
    torch.cuda.synchronize()
    rows.append(buf.cpu().numpy().reshape(nb, 4, 8).astype(np.float64))
env.lib.uavenv_set_debug_buffer(env._h, None)
t = np.stack(rows)                       # [iters, blocks, wave, 8]
t0 = t[:, :, :, 0].min(axis=2, keepdims=True)
names = ["start", "staged", "own work done", "after barrier 1", "after barrier 2", "own obs part done", "after barrier 3", "end"]
if os.environ.get("POLICY") == "2":
    names = ["start", "world + fc1 staged (barrier)", "layer-1 forward done", "layer 2 + eps-greedy done", "action barrier passed",
             "step_pre done", "update_PathPlan done (w0)", "end"]
print(f"{n} envs, {nb} workgroups x 4 waves; cycles since the workgroup's first stamp (mean over workgroups and 20 launches | p95)")
for k, nm in enumerate(names):
    x = t[:, :, :, k] - t0
    print(f"  {nm:30s} " + "  ".join(f"w{w}: {x[:, :, w].mean():7.0f} | {np.percentile(x[:, :, w], 95):7.0f}" for w in range(4)))
tot = (t[:, :, :, 7].max(axis=2) - t[:, :, :, 0].min(axis=2)).ravel()
print(f"  workgroup lifetime: mean {tot.mean():.0f}  p95 {np.percentile(tot, 95):.0f}  max {tot.max():.0f}")
