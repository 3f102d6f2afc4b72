"""A CPU stand-in for VecPathPlanEnv built on the oracle -- TEST INFRASTRUCTURE ONLY.  It lets the CPU suite
exercise the plugins' host logic (XML, factories, episode loop, replay, result dict) without a GPU.  The product
never imports it: plugins/_backend.make_backend constructs the HIP env and nothing else."""
import ctypes as C

import numpy as np
import torch

from dqn_based_uav_3d_path_planer_amd.env import StepOut
from oracle import pyoracle as po


class OracleVecEnv:
    def __init__(self, n_envs, buildings, *, uav_per_env=1, max_subgoals=48, max_step=150, apf_enabled=0,
                 obs_dtype=torch.float32, n_actions=3, length=500.0, width=500.0, h=100.0, max_v=1.0,
                 steering_angle=np.pi / 6, power=None, device="cpu", **_):
        self.device = torch.device("cpu")
        self.n_envs, self.uav_per_env, self.N = n_envs, uav_per_env, n_envs * uav_per_env
        self.K, self.n_actions, self.obs_dtype = max_subgoals, n_actions, obs_dtype
        self.world = po.OracleWorld(buildings, length, width, h)
        self.params = dict(max_v=float(max_v), steering_angle=float(steering_angle), max_step=int(max_step),
                           apf_enabled=int(apf_enabled))
        self.batch = po.OracleBatch(self.world, self.params, self.N)
        self.bank = None

    def load_scenarios(self, start_goal, sub_goals, n_sub):
        self.bank = (np.asarray(start_goal, dtype=np.float64), np.asarray(sub_goals, dtype=np.float64),
                     np.asarray(n_sub, dtype=np.int32))

    def reset(self, seed=0, obs=None):
        sg, sub, ns = self.bank
        rng = np.random.default_rng(seed)
        pick = rng.integers(0, len(sg), self.N)
        head = rng.uniform(0, 2 * np.pi, self.N)
        v = self.batch.view
        v["px"], v["py"], v["pz"] = sg[pick, 0], sg[pick, 1], sg[pick, 2]
        v["gx"], v["gy"], v["gz"] = sg[pick, 3], sg[pick, 4], sg[pick, 5]
        v["vx"], v["vy"], v["vz"], v["V"] = self.params["max_v"] * np.cos(head), self.params["max_v"] * np.sin(head), 0.0, 1.0
        for i in range(self.N):
            self.batch.arr[i].V = self.batch.lib.orc_calc_v(C.byref(self.batch.arr[i]))
        v["step"], v["done"], v["reach_goal"], v["error"] = 0, 0, 0, 0
        v["score"], v["total_score"], v["path_len"] = 0.0, 0.0, 0.0
        v["n_sub"] = ns[pick]
        v["sub0_alias"] = (ns[pick] >= 2).astype(np.int32)
        k = min(sub.shape[1], po.KMAX)
        v["sub"][:, :k] = sub[pick][:, :k]
        return self.observe()

    def observe(self, obs=None):
        out = np.zeros((self.N, 100))
        for i in range(self.N):
            self.batch.lib.orc_state_pathplan(C.byref(self.world.w), C.byref(self.batch.arr[i]), out[i].ctypes.data)
        return torch.tensor(out, dtype=self.obs_dtype)

    def step(self, actions, out=None, *, auto_reset=False, skip_done=False, want_energy=False, active=None):
        a = actions.cpu().numpy()
        steer = (-1.0 + 2.0 * a.astype(np.float64) / (self.n_actions - 1)) if actions.dtype == torch.int32 else a.astype(np.float64)
        v = self.batch.view
        skip = np.zeros(self.N, dtype=bool)
        if skip_done:
            skip |= v["done"] != 0
        if active is not None:
            skip |= active.cpu().numpy() == 0
        saved = v.copy()
        r, d, info, _ = self.batch.step(steer, want_obs=False)
        agent_done = v["done"].astype(np.uint8)
        if skip.any():
            v[skip] = saved[skip]
            r[skip], d[skip], info[skip] = 0.0, saved["done"][skip], 3
            agent_done[skip] = saved["done"][skip]
        obs = self.observe()
        t = torch.tensor
        return StepOut(obs=obs, reward=t(r), reward32=t(r.astype(np.float32)), ret_done=t(d.astype(np.uint8)),
                       agent_done=t(agent_done), info=t(info.astype(np.uint8)), valid=t((~skip).astype(np.uint8)),
                       energy=None)

    def get_state(self, first=0, count=None, want_sub=False):
        count = self.N - first if count is None else count
        v = self.batch.view[first:first + count]
        st = np.stack([v["px"], v["py"], v["pz"], v["vx"], v["vy"], v["V"], v["gx"], v["gy"], v["gz"],
                       v["step"].astype(float), v["done"].astype(float), v["n_sub"].astype(float), v["score"],
                       v["total_score"], v["path_len"], v["reach_goal"].astype(float)], axis=1)
        if not want_sub:
            return st
        return st, v["sub"][:, :self.K].copy(), v["sub0_alias"].copy()

    def threaten_rate(self, points, allpairs=False):
        return torch.tensor(self.world.threaten_rate_many(points.cpu().numpy()).astype(np.uint8))

    def close(self):
        pass
