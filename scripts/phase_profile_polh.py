"""Per-phase timeline of the one-wave k_step on an f16 ring, with the f16-MFMA policy in its prologue (uavenv_step_policy)
and without (random actions), from in-kernel s_memtime stamps (diagnostic, UAVENV_PHASE_PROFILE=1 build).
python scripts/phase_profile_polh.py [envs]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dqn_based_uav_3d_path_planer_amd.data import make_city26_env
from dqn_based_uav_3d_path_planer_amd.learner import FusedDQNLearner
from dqn_based_uav_3d_path_planer_amd.replay import DeviceReplayRing

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
PARAM = {"w": "100", "hiden_dim": "64", "output": "3", "LEARNING_RATE": "0.001", "gamma": "0.99", "Update_loop": "3", "NetWork": "VAnet2"}
env = make_city26_env(n, obs_dtype=torch.float16)
ring = DeviceReplayRing(env, 8 * n, discrete=True)
ring.reset(seed=1)
L = FusedDQNLearner(PARAM, "dueling", device="cuda:0", mfma="f16")
gen = torch.Generator(device="cuda").manual_seed(0)
for _ in range(300):
    ring.current_action().copy_(torch.randint(0, 3, (n,), generator=gen, device="cuda", dtype=torch.int32))
    ring.step_env(auto_reset=True)
nw = (n + 63) // 64
buf = torch.zeros(3 * nw * 8, dtype=torch.int64, device="cuda")
names = ["stage world (+loads in flight)", "wait state (+ policy)", "step math", "auto-reset", "obs compute",
         "issue stores", "stores retire"]
for policy in (False, True):
    env.lib.uavenv_set_debug_buffer(env._h, buf.data_ptr())
    rows = []
    for c in range(20):
        if policy:
            assert ring.step_policy(L, 0.2, 3, c)
        else:
            ring.current_action().copy_(torch.randint(0, 3, (n,), generator=gen, device="cuda", dtype=torch.int32))
            ring.step_env(auto_reset=True)
        torch.cuda.synchronize()
        full = buf.cpu().numpy().reshape(3 * nw, 8).astype(np.float64)
        rows.append(full[:nw])
        prow = full[nw:].reshape(nw, 2, 8)
    env.lib.uavenv_set_debug_buffer(env._h, None)
    t = np.stack(rows)
    d = np.diff(t, axis=2)
    print(f"policy={policy}: {n} envs, {nw} waves; cycles per wave (mean / p95 / max)")
    for k, nm in enumerate(names):
        x = d[:, :, k].ravel()
        print(f"  {nm:40s} {x.mean():9.0f} {np.percentile(x, 95):9.0f} {x.max():9.0f}")
    tot = (t[:, :, 7] - t[:, :, 0]).ravel()
    print(f"  {'wave lifetime (stamp 0 -> 7)':40s} {tot.mean():9.0f} {np.percentile(tot, 95):9.0f} {tot.max():9.0f}")
    span = (t[:, :, 7].max(axis=1) - t[:, :, 0].min(axis=1))
    print(f"  launch span: mean {span.mean():.0f} cycles")
    if policy:       # the policy wavefronts' own stamps (last launch), relative to their agent wavefront's stamp 0
        rel = (prow[:, :, :5] - rows[-1][:, None, :1]).reshape(-1, 5)
        for k, nm in enumerate(["policy: start", "loads issued", "past barrier 1 (fc1 tile in LDS)", "MFMAs done", "layer 2 done"]):
            print(f"  {nm:40s} {rel[:, k].mean():9.0f} {np.percentile(rel[:, k], 95):9.0f} {rel[:, k].max():9.0f}")
