"""Drop-in for Agents/UAV.py restricted to the PathPlan hot path.  One `UAV` object is agent slot j (UAV_j) of
EVERY vectorised env: the state lives in HBM inside libuavenv; this object is the host-side view with the
reference's attribute and method names.  Scalar accessors (`position`, `done`, `score`, `Step`, `state()`,
`update(a)`) refer to env 0, so single-env code written against the reference keeps its meaning; the batched
forms carry the `_all` / `_batch` suffix."""
from __future__ import annotations

import math
import time

import numpy as np
import torch

from dqn_based_uav_3d_path_planer_amd.compat import Loc

INFO_NAMES = ("normal", "success", "lose", "skipped")


class UAV:
    def __init__(self, param: dict, env=None) -> None:
        self.param = param
        self.env = env
        self.name = param.get("name")
        self.j = int(param.get("j") or 0)
        self.Max_V = int(param.get("Max_V"))                                       # UAV.py:25
        self.Steering_angle = float(param.get("Steering_angle")) / 180 * math.pi   # UAV.py:26
        self.Max_Step = int(param.get("Max_Step"))                                 # UAV.py:32
        self.APF_Enabled = int(param.get("APF_Enabled") or 0)
        self.sub_granularity = int(param.get("sub_granularity") or 30)
        self.update_function_name = param.get("update_function_name")
        self.state_function_name = param.get("state_function_name")
        if self.update_function_name not in (None, "update_PathPlan") or \
                self.state_function_name not in (None, "state_PathPlan"):
            raise ValueError("only update_PathPlan / state_PathPlan are on the MI355X hot path")
        fp = (param.get("Power_param") or {}).get("Fly_power") or {}
        g = lambda k, d: float(fp.get(k)) if fp.get(k) is not None else d   # noqa: E731
        self.P_i, self.v_0, self.d_0, self.rho = g("P_i", 89.0), g("v_0", 4.05), g("d_0", 0.6), g("rho", 1.225)
        self.s, self.A, self.P_b, self.F_b = g("s", 0.05), g("A", 0.5) + 0.03 * self.j, g("P_b", 79.0), g("F_b", 120.0)
        self.xi = 0.8 + 0.02 * self.j
        self.Trainer = None
        self.transition_dict = {"states": [], "actions": [], "next_states": [], "rewards": [], "dones": []}
        self.Train_time = 0
        self.Testing_time = 0
        self.UEs = []
        self.energy_cost_total = 0
        self.task_collect = 0
        self.Train_start = time.time()
        self.infos = []
        self.CsvWriter = None
        if int(param.get("record_csv") or 0):
            self.Init_Record_Mod()

    # ---- device-backed state (env 0 scalar view) --------------------------------------------------
    def _row(self, e: int = 0):
        return self.env._state_row(e, self.j)

    @property
    def position(self):
        r = self._row()
        return Loc(r[0], r[1], r[2])

    @property
    def V_vector(self):
        r = self._row()
        return Loc(r[3], r[4], 0)

    @property
    def V(self):
        return self._row()[5]

    @property
    def goal(self):
        r = self._row()
        return Loc(r[6], r[7], r[8])

    @property
    def Step(self):
        return int(self._row()[9])

    @property
    def done(self):
        return bool(self._row()[10])

    @property
    def score(self):
        return float(self._row()[12])

    @property
    def total_score(self):
        return float(self._row()[13])

    @property
    def path_len(self):
        return float(self._row()[14])

    @property
    def reach_goal(self):
        return int(self._row()[15])

    @property
    def sub_goals(self):
        return [Loc(*p) for p in self.env._subgoals(0, self.j)]

    @property
    def Train_epoch(self):
        return self.Trainer.epoch if self.Trainer is not None else 0

    @property
    def start2goal(self):
        r = self._row()
        return math.sqrt((r[0] - r[6]) ** 2 + (r[1] - r[7]) ** 2 + (r[2] - r[8]) ** 2)

    len_Astar = 0

    # ---- reference methods ----------------------------------------------------------------------------
    def reset(self, option=None):
        """UAV.py:327-366.  Resets are batched: any full reset request restarts the whole scene."""
        if option == "local reset":
            return
        self.env.Scene_Random_Reset()

    def state(self):
        """state_PathPlan (UAV.py:515-567) of env 0's UAV_j as float64[100]."""
        return self.env._obs_rows(self.j)[0].astype(np.float64)

    def state_all(self) -> torch.Tensor:
        return self.env._obs_slot(self.j)

    def update(self, action):
        """update_PathPlan (UAV.py:397-513) for this slot in every env -> env 0's (reward, done, info)."""
        r, d, info = self.env._step_slot(self.j, action)
        return float(r[0]), bool(d[0]), INFO_NAMES[int(info[0])]

    def get_action(self, state, eps):
        start = time.time()
        action = self.Trainer.get_action(state, eps)
        self.Testing_time += time.time() - start
        return action

    def Train_nn(self):
        start = time.time()
        re = self.Trainer.update(self.transition_dict)
        self.Train_time += time.time() - start
        return re

    @property
    def path(self):
        """UAV.py:42,431: the positions env 0's UAV_j has visited since its last reset."""
        return self.env._paths[self.j]

    RECORD_HEADER = ["sum_Episode", "Episode", " Score", " Avg.Score", "eps-greedy", "success", "failed", "meet_threaten",
                     "loss", "ALL_UEs_D", "ALL_UEs_F", "energy_cost", "task_collect", "Energy_Efficent", "UE_waiting_time",
                     "Covered_rate", "task_executed", "executed_rate", "KL", "Train_time", "Testing_time"]

    def Init_Record_Mod(self, directory="logs"):
        """UAV.py:268-277: logs/<name>_<time>.csv with the reference's 21-column header.  Opt-in here
        (<record_csv>1</record_csv> in UAV.xml): the reference opens one file per UAV at construction."""
        import csv
        import datetime
        import os
        os.makedirs(directory, exist_ok=True)
        cur = datetime.datetime.now().strftime("%m_%d_%Y(%H_%M_%S)")
        self._record_file = open(os.path.join(directory, "%s_%s.csv" % (self.name, cur)), "a+", newline="")
        self.CsvWriter = csv.writer(self._record_file)
        self.CsvWriter.writerow(self.RECORD_HEADER)

    def record_list(self):
        """UAV.py:279-310.  The UE / task / KL columns belong to the data-collection scenarios (out of the PathPlan
        path): zero here, exactly what the reference writes for a PathPlan run.  Score columns are env 0's, as the
        scalar accessors are."""
        if self.CsvWriter is not None:
            epoch = self.Trainer.epoch if self.Trainer is not None else 0
            sc = self.score
            ee = self.task_collect / (self.energy_cost_total + 0.001)
            self.CsvWriter.writerow([epoch, epoch, sc, sc, 0, 0, 0, 0, 0, 0, 0, self.energy_cost_total, self.task_collect, ee,
                                     0, 0.0, 0, 0.0, [], self.Train_time, self.Testing_time])
            self._record_file.flush()
        self.infos = []

    def Calc_Fly_Power(self):
        """UAV.py:239-245 at env 0's current speed."""
        V = self.V
        induced = self.P_i * math.sqrt(math.sqrt(1 + (V ** 4) / (4 * (self.v_0 ** 4))) - (V ** 2) / (2 * (self.v_0 ** 2)))
        parasite = 0.5 * self.d_0 * self.rho * self.s * self.A * (V ** 3)
        blade = self.xi * self.P_b * (1 + 3 * (V ** 2) / (self.F_b ** 2))
        return induced + parasite + blade
