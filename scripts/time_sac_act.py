"""Time k_sac_act (get_action of every UAV slot, one launch) at BASELINE configs[3]'s size."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C

from dqn_based_uav_3d_path_planer_amd import _lib
from dqn_based_uav_3d_path_planer_amd.sac import FusedSACLearner

P = {"actor": {"NetWork": "PolicyNetContinuous_SAC", "w": "100", "action_bound": "1", "hiden_dim": "64", "output": "2", "lr": "0.0001"},
     "critic": {"NetWork": "QValueNetContinuous_SAC", "w": "100", "hiden_dim": "64", "action_dim": "2", "lr": "0.001"},
     "SAC_param": {"IS_Continuous": "1", "alpha_lr": "0.0001", "target_entropy": "1", "gamma": "0.99", "tau": "0.05"}}
U, envs = 4, 32768
N = U * envs
lib = _lib.load()
vp = C.c_void_p
lib.uavenv_sac_act_multi.argtypes = [vp, vp, vp, C.c_int32, C.c_int32, vp, C.c_float, vp, vp, C.c_int32, vp]
Ls = [FusedSACLearner(P, "cuda:0") for _ in range(U)]
obs = torch.randint(0, 2**31 - 1, (N, 20), dtype=torch.int32, device="cuda")
obs[:, 3:18] = torch.randn((N, 15), device="cuda").view(torch.int32)
eps = torch.randn((U, envs, 2), device="cuda")
a0, a1 = torch.zeros(N, device="cuda"), torch.zeros(N, device="cuda")
actors = (C.c_void_p * U)(*[L._blocks[0].data_ptr() for L in Ls])
epss = (C.c_void_p * U)(*[eps[j].data_ptr() for j in range(U)])
first = (C.c_int32 * U)(*range(U))
s = torch.cuda.current_stream().cuda_stream
def run():
    rc = lib.uavenv_sac_act_multi(actors, obs.data_ptr(), first, U, envs, epss, 1.0, a0.data_ptr(), a1.data_ptr(), U, s)
    assert rc == 0, rc
run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200):
    run()
e1.record(); torch.cuda.synchronize()
print("k_sac_act x4 slots x 32768 agents: %.2f us  (UAVENV_SAC_ACT_WGS=%s)  checksum %.6f" % (e0.elapsed_time(e1) / 200 * 1e3, os.environ.get("UAVENV_SAC_ACT_WGS"), float(a0.double().sum() + a1.double().sum())))
