"""Per-phase timeline of k_step (the one-wavefront-per-64-agents kernel) from in-kernel s_memtime stamps (diagnostic,
-DUAVENV_PHASE_PROFILE build).  python scripts/phase_profile.py [envs]   (phase_profile_coop.py for k_step_coop)"""
import sys, os
os.environ.setdefault("UAVENV_COOP", "0")      # small launches would otherwise take the cooperative kernel
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dqn_based_uav_3d_path_planer_amd.data import make_city26_env
from dqn_based_uav_3d_path_planer_amd.replay import DeviceReplayRing

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
env = make_city26_env(n)
ring = DeviceReplayRing(env, 8 * n)
ring.reset(seed=1)
gen = torch.Generator(device="cuda").manual_seed(0)
for _ in range(300):      # desynchronise episodes: steady-state mix of pops and resets
    ring.current_action().copy_(torch.randint(0, 3, (n,), generator=gen, device="cuda", dtype=torch.int32))
    ring.step_env(auto_reset=True)
nw = (n + 63) // 64
buf = torch.zeros(nw * 8, dtype=torch.int64, device="cuda")
env.lib.uavenv_set_debug_buffer(env._h, buf.data_ptr())
rows = []
for _ in range(20):
    ring.current_action().copy_(torch.randint(0, 3, (n,), generator=gen, device="cuda", dtype=torch.int32))
    ring.step_env(auto_reset=True)
    torch.cuda.synchronize()
    rows.append(buf.cpu().numpy().reshape(nw, 8).astype(np.float64))
env.lib.uavenv_set_debug_buffer(env._h, None)
t = np.stack(rows)                       # [iters, waves, 8]
names = ["stage world (+state loads in flight)", "wait state", "step math", "auto-reset", "obs compute",
         "issue stores", "stores retire"]
d = np.diff(t, axis=2)
print(f"{n} envs, {nw} waves; shader-clock cycles per wave (mean / p95 / max over waves and 20 launches)")
for k, nm in enumerate(names):
    x = d[:, :, k].ravel()
    print(f"  {nm:40s} {x.mean():9.0f} {np.percentile(x, 95):9.0f} {x.max():9.0f}")
tot = (t[:, :, 7] - t[:, :, 0]).ravel()
print(f"  {'wave lifetime (stamp 0 -> 7)':40s} {tot.mean():9.0f} {np.percentile(tot, 95):9.0f} {tot.max():9.0f}")
span = (t[:, :, 7].max(axis=1) - t[:, :, 0].min(axis=1))
print(f"  launch span (first stamp 0 -> last stamp 7): mean {span.mean():.0f} cycles")
