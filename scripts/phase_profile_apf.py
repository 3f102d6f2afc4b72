"""Per-phase cycle totals of k_apf_adjust from in-kernel s_memtime stamps (diagnostic, UAVENV_PHASE_PROFILE=1 build).
python scripts/phase_profile_apf.py [envs] [uav_per_env]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dqn_based_uav_3d_path_planer_amd.data import make_city26_env
from dqn_based_uav_3d_path_planer_amd.replay import DeviceReplayRing

envs = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
U = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda:0")
env = make_city26_env(envs, bank="gpu", bank_size=max(envs, 4096), bank_seed=42, device=dev, obs_dtype="packed",
                      uav_per_env=U, apf_enabled=1)
v = np.random.default_rng(42).uniform(-1.0, 1.0, (len(env.buildings), 3)); v[:, 2] = 0.0
env.set_buildings(env.buildings, velocities=v)
ring = DeviceReplayRing(env, 9 * env.N, discrete=True); ring.reset(seed=1000)
gen = torch.Generator(device=dev).manual_seed(0)
def step():
    ring.current_action().copy_(torch.randint(0, 3, (env.N,), generator=gen, device=dev, dtype=torch.int32))
    ring.step_env(auto_reset=True)
for _ in range(300): step()
nw = (env.N + 63) // 64 * 4
buf = torch.zeros(nw * 8 + 4096, dtype=torch.int64, device=dev)
env.lib.uavenv_set_debug_buffer(env._h, buf.data_ptr())
rows = []
for _ in range(10):
    step(); torch.cuda.synchronize()
    rows.append(buf[:nw * 8].cpu().numpy().reshape(nw, 8).astype(np.float64))
env.lib.uavenv_set_debug_buffer(env._h, None)
t = np.stack(rows)[:, (env.N + 63) // 64:, :6]      # (k_step's own stamps overwrite the first N/64 wave slots)
names = ["setup (table, counts, scan)", "pass 1 loads + mask lookup", "histogram / keys / park", "sort", "pass 2 forces", "pass 3 stores"]
tot = t.sum(2)
print(f"{env.N} agents, {nw} waves; s_memtime ticks (100 MHz) per wave: mean / p95 / max")
for k, nm in enumerate(names):
    x = t[:, :, k].ravel(); print(f"  {nm:32s} {x.mean():9.1f} {np.percentile(x, 95):9.1f} {x.max():9.1f}")
print(f"  {'wave total':32s} {tot.mean():9.1f} {np.percentile(tot, 95):9.1f} {tot.max():9.1f}")
