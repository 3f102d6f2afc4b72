"""Drop-in for Envs/PathPlan_City.py on the MI355X hot path: same constructor contract (the <env> XML dict),
same duck type simulator.py touches (run_eposide, Agents, Trainer, run_XML_scene, Threaten_rate, Move_Agent,
update, Check_uav_Done, result-dict keys), backed by `num_envs` vectorised envs in HBM.  Optional new tags
(all default to the reference's behaviour): <num_envs>, <device>, <obs_dtype>, <scenario_bank>, <seed>.
Out of scope and therefore absent: rendering, MySQL.

Two ways through run_eposide:
  * the FUSED path (one UAV per env, a DQN-family trainer on the fused learner, a real GPU backend; <fast_path>0</fast_path>
    turns it off): observations are packed rows in a DeviceReplayRing, and the act -> step -> store -> sample -> learn
    cycle of Envs/PathPlan_City.py:364-385 is enqueued by csrc/loop.hip (HotLoop) -- the same kernels, the same order and
    the same replay as bench.py.  The host looks at the device every <done_check> steps (default 8) to see whether every
    agent has finished; the episode's counters come from the ring's info / valid planes;
  * the general path: every trainer plugin (SAC, several UAVs per env, prioritised replay, the PyTorch learner) through
    batched tensors, one sync per step."""
from __future__ import annotations

import os

import numpy as np
import torch

import _backend
from dqn_based_uav_3d_path_planer_amd.compat import Loc, None2Value, XML2Dict
from dqn_based_uav_3d_path_planer_amd.factories import AgentFactory, ThreatenFactory, TrainerFactory


class _RingMemoryView:
    """What the reference's callers read off `Trainer.replay_memory` when the replay is the device ring."""

    class _Len:
        def __init__(self, view):
            self._v = view

        def __len__(self):
            return len(self._v)

    def __init__(self, ring, per_frame: int = None):
        """per_frame: transitions of this memory per ring frame (one UAV slot of a multi-UAV ring: num_envs)."""
        self.ring = ring
        self.per_frame = per_frame
        self.buffer = self.memory = _RingMemoryView._Len(self)
        self.capacity = ring.capacity if per_frame is None else (ring.frames - 1) * per_frame

    def __len__(self):
        return len(self.ring) if self.per_frame is None else self.ring.filled * self.per_frame

    def sample_tensors(self, batch_size: int) -> dict:
        self._n = getattr(self, "_n", 0) + 1
        b = self.ring.sample(batch_size, seed=0x51ED, counter=self._n)
        return dict(states=b["states"], actions=b["actions"].long(), rewards=b["rewards"], next_states=b["next_states"],
                    dones=b["dones"], valid=b["valid"])


class PathPlan_City:
    def __init__(self, param: dict) -> None:
        # ---- BaseEnv.__init__ (BaseClass/BaseEnv.py:17-34)
        self.len = int(None2Value(param.get("len"), 100))
        self.width = int(None2Value(param.get("width"), 100))
        self.h = int(None2Value(param.get("h"), 20))
        self.Agents, self.Threatens, self.Trainer = [], [], None
        self.AgentFactory, self.ThreatenFactory, self.TrainerFactory = AgentFactory(), ThreatenFactory(), TrainerFactory()
        self.Is_AC = int(None2Value(param.get("Is_AC"), 0))
        # ---- PathPlan_City.__init__ (Envs/PathPlan_City.py:31-103)
        self.eps = float(None2Value(param.get("eps"), 0.1))
        self.Is_On_Policy = int(None2Value(param.get("Is_On_Policy"), 0))
        # Is_On_Policy = 1 (Envs/PathPlan_City.py:386-436): every UAV collects its whole episode in transition_dict (nothing goes to the
        # replay memory) and trains ONCE after the episode -- the general per-step path here (_run_eposide_on_policy); the fused loops
        # are the off-policy ones (run_thread_OffPolicy, :364-385) and stay off for such a config.
        self.param = param
        self.buildings = []
        threaten_params = param.get("Obstacles")
        self.buildings_param = None
        if threaten_params is not None:
            cfg = XML2Dict(os.path.normpath(threaten_params.get("buildings")))
            self.buildings_param = cfg.get("buildings")
        if self.buildings_param is not None:
            items = self.buildings_param["Threaten"]
            for p in (items if isinstance(items, list) else [items]):
                self.buildings.append(self.ThreatenFactory.Create_Threaten(p))
        self.num_UAV = int(param.get("num_UAV"))
        agents_params = param.get("Agent")
        uav_params = XML2Dict(os.path.normpath(agents_params["xml_path_agent"])).get("Agent")
        self.num_envs = int(None2Value(param.get("num_envs"), 1))
        self.seed = int(None2Value(param.get("seed"), 42))
        b = np.array([[t.position.x, t.position.y, t.position.z, t._R, t._H] for t in self.buildings], dtype=np.float64)
        obs_dtype = torch.float16 if (param.get("obs_dtype") or "f32") == "f16" else torch.float32
        trainer_xml = os.path.normpath(agents_params["Trainer"].get("Trainer_path"))
        tcfg = XML2Dict(trainer_xml).get("Trainer")
        n_actions = int(None2Value(tcfg.get("output"), 3))
        # the fused path keeps observations packed (15 scalars + 80 flag bits per row); decided before the backend exists
        self._want_fast = (int(None2Value(param.get("fast_path"), 1)) != 0 and self.num_UAV == 1 and
                           (tcfg.get("Trainer_Type") in ("DQN_Trainer", "DDQN_Trainer", "DuelingDQN_Trainer")) and
                           param.get("obs_dtype") is None and
                           int(None2Value(tcfg.get("Batch_Size"), 128)) % 64 == 0 and
                           int(None2Value(tcfg.get("fused"), 1)) != 0 and torch.cuda.is_available())
        # ... and so does the SAC fast path (any number of UAVs per env: one fused trainer per UAV slot, csrc/sac.hip)
        sacp = tcfg.get("SAC_param") or {}
        self._want_fast_sac = (int(None2Value(param.get("fast_path"), 1)) != 0 and tcfg.get("Trainer_Type") == "SAC_Trainer" and
                               int(None2Value(sacp.get("IS_Continuous"), 0)) == 1 and param.get("obs_dtype") is None and
                               int(None2Value(tcfg.get("Batch_Size"), 128)) % 64 == 0 and
                               int(None2Value(tcfg.get("fused"), 1)) != 0 and torch.cuda.is_available())
        if self.Is_On_Policy == 1:
            self._want_fast = self._want_fast_sac = False
        if self._want_fast or self._want_fast_sac:
            obs_dtype = "packed"
        fp = (uav_params.get("Power_param") or {}).get("Fly_power") or {}
        power = tuple(float(fp.get(k)) for k in ("P_i", "v_0", "d_0", "rho", "s", "A", "P_b", "F_b")) \
            if all(fp.get(k) is not None for k in ("P_i", "v_0", "d_0", "rho", "s", "A", "P_b", "F_b")) else None
        kw = dict(uav_per_env=self.num_UAV, max_step=int(uav_params.get("Max_Step")),
                  apf_enabled=int(uav_params.get("APF_Enabled") or 0), obs_dtype=obs_dtype, n_actions=n_actions,
                  length=float(self.len), width=float(self.width), h=float(self.h), max_v=float(int(uav_params.get("Max_V"))),
                  steering_angle=float(uav_params.get("Steering_angle")) / 180 * np.pi)
        if power is not None:
            kw["power"] = power
        if param.get("device"):
            kw["device"] = param.get("device")
        self.backend = _backend.make_backend(self.num_envs, b, **kw)
        self._load_bank(param.get("scenario_bank"), b)
        for i in range(self.num_UAV):
            up = dict(uav_params)
            up["name"] = "UAV_" + str(i)
            up["j"] = i
            agent = self.AgentFactory.Create_Agent(up, self)
            tp = dict(XML2Dict(trainer_xml).get("Trainer"))
            tp["name"] = agent.name
            if param.get("device") and not tp.get("device"):
                tp["device"] = param.get("device")
            agent.Trainer = self.TrainerFactory.Create_Trainer(tp)
            self.Agents.append(agent)
        self.result = {"success": 0, "failed:": 0, "meet_threaten": 0, "normal": 0, "loss": None, "sum_epoch": 0, "eps": 0.1}
        self.epoch = 0
        self.print_loop = int(None2Value(param.get("print_loop"), 2))
        self.Is_FL = int(None2Value(param.get("Is_FL"), 0))
        self.FL_Loop = int(None2Value(param.get("FL_Loop"), 3))
        # <FL_Aggregate>: "reference" (default) = the merge as the reference executes it -- the SUM of the UAVs' models: its
        # division at Envs/PathPlan_City.py:597 never reaches the model -- or "mean" (dqn_based_uav_3d_path_planer_amd/federated.py)
        self.FL_Aggregate = str(None2Value(param.get("FL_Aggregate"), "reference"))
        # (absent: the actor-critic merge keeps the executed reference's SUM and says so once; the DQN-family merge, for which the
        # reference has no executed behaviour -- its Federated_Learning raises AttributeError -- takes the mean: a SUM of U
        # Q-networks scales every Q-value by ~U.  ADVICE r4)
        self._fl_aggregate_given = param.get("FL_Aggregate") is not None
        self._fl_warned = False
        from dqn_based_uav_3d_path_planer_amd import federated as _fed
        if self.FL_Aggregate not in _fed.AGGREGATES:
            raise ValueError(f"<FL_Aggregate>{self.FL_Aggregate}</FL_Aggregate>: expected one of {_fed.AGGREGATES}")
        self.fl_merges = 0             # federated merges done so far (diagnostics / tests)
        self.executed_time = 0
        self._state_cache = None
        self._state0 = None
        self._obs_raw = None
        self._episode = 0
        self._ring = self._hot = self._info = None
        self.done_check = max(1, int(None2Value(param.get("done_check"), 8)))
        # <fresh_plans> (default 1 on a GPU backend): UAV.reset plans a new RRT path at EVERY reset (Agents/UAV.py:327-366); here a
        # slice of <fresh_plans_rows> bank rows is planned in the background of each episode and handed over at the next
        # episode boundary, where every agent is reset anyway (_refresh_bank)
        self.fresh_plans = int(None2Value(param.get("fresh_plans"), 1)) != 0 and hasattr(self.backend, "replan_begin")
        self.fresh_plans_rows = int(None2Value(param.get("fresh_plans_rows"), 256))   # ~what the planner finishes beside one episode
        self._replan_next, self._plan_stream = 0, None
        tr0 = self.Agents[0].Trainer
        self.fast = bool(self._want_fast and getattr(self.backend, "packed", False) and getattr(tr0, "fused", False))
        # fusion is decided HERE: a trainer that came up fused but whose env cannot offer the packed ring is moved to the
        # PyTorch learner (same weights) instead of failing later inside update()
        if self.fast:
            from dqn_based_uav_3d_path_planer_amd.replay import DeviceReplayRing
            self._ring = DeviceReplayRing(self.backend, max(tr0.replay_size, 2 * self.backend.N), discrete=True)
            self._info = torch.zeros((self._ring.frames, self.backend.N), dtype=torch.uint8, device=self.backend.device)
            tr0.replay_memory = _RingMemoryView(self._ring)
            self._per = None
            if getattr(tr0, "IsPriority_Replay", 0) == 1:
                # IsPriority_Replay = 1 stays on the fused path: one priority per ring slot (frame * N + agent), ReplayTree's
                # hyper-parameters, sampled / updated inside the C loop (csrc/loop.hip, csrc/per.hip)
                from dqn_based_uav_3d_path_planer_amd.replay import DevicePER
                self._per = DevicePER(self._ring.frames * self.backend.N, device=self.backend.device, tree_order=False)
        self.fast_sac = bool(self._want_fast_sac and getattr(self.backend, "packed", False) and
                             all(getattr(u.Trainer, "fused", False) for u in self.Agents))
        if not self.fast_sac:
            for u in self.Agents:
                if type(u.Trainer).__name__ == "SAC_Trainer" and getattr(u.Trainer, "fused", False):
                    u.Trainer.downgrade()
        if self.fast_sac:
            # one packed ring for all UAV slots: slot j's replay memory is rows e * num_UAV + j of every frame -- num_envs
            # transitions per frame, replay_size transitions per trainer as in the reference (one ReplayMemory per UAV)
            from dqn_based_uav_3d_path_planer_amd.replay import DeviceReplayRing
            N, d = self.backend.N, self.backend.device
            self._ring = DeviceReplayRing(self.backend, max(tr0.replay_size * self.num_UAV, 2 * N), discrete=False)
            ring = self._ring
            self._info = torch.zeros((ring.frames, N), dtype=torch.uint8, device=d)
            self._a1 = torch.zeros((ring.frames, N), dtype=torch.float32, device=d)          # second action component (:444-448)
            ring.attach_action1(self._a1)                        # ... recorded with every transition (DeviceReplayRing.meta)
            bs = [u.Trainer.Batch_Size for u in self.Agents]
            self._draws_all = torch.empty((sum(bs), 2), dtype=torch.int32, device=d)       # one draw launch per step for all slots
            offs = np.cumsum([0] + bs)
            self._draws = [self._draws_all[offs[j]:offs[j + 1]] for j in range(self.num_UAV)]
            flat = ring.obs.view(-1, ring.obs.shape[-1])
            self._flat = flat
            # IsPriority_Replay = 1 (the reference's ReplayTree use, Trainer/SAC_Trainer.py:336-352): one priority per stored
            # transition of each UAV slot (data slot = frame * num_envs + env); the slot's draws then come from its tree
            self._sac_per, self._sac_per_bufs = [None] * self.num_UAV, [None] * self.num_UAV
            for j, u in enumerate(self.Agents):
                if getattr(u.Trainer, "IsPriority_Replay", 0) == 1:
                    from dqn_based_uav_3d_path_planer_amd.replay import DevicePER
                    self._sac_per[j] = DevicePER(ring.frames * self.num_envs, device=d, tree_order=False)
                    self._sac_per_bufs[j] = self._sac_per[j].make_bufs(u.Trainer.Batch_Size, pairs=self._draws[j])
            self._sac_batches = [u.Trainer.learner.make_batch(flat, ring.action.view(-1), self._a1.view(-1), ring.reward.view(-1),
                                                              ring.done.view(-1), valid=ring.valid.view(-1), draws=self._draws[j],
                                                              n_agents=N, uav_per_env=self.num_UAV, slot=j, frames=ring.frames,
                                                              is_weights=None if self._sac_per[j] is None else self._sac_per_bufs[j]["w"],
                                                              abs_td_out=None if self._sac_per[j] is None else self._sac_per_bufs[j]["abs"],
                                                              meta=None if ring.meta is None else ring.meta.view(-1, 4))
                                 for j, u in enumerate(self.Agents)]
            self._sac_counter = 0
            for u in self.Agents:
                u.Trainer.replay_memory = _RingMemoryView(ring, per_frame=self.num_envs)
        # UAV.path of env 0 (UAV.py:431) and the path.csv the reference rewrites at every terminal (:461-464,:479-483,
        # :505-509).  One 16-double read-back of env 0's agents per step; <record_path>0</record_path> turns it off.
        self.record_path = int(None2Value(param.get("record_path"), 0 if (self.fast or self.fast_sac) else 1))
        self.path_csv = None2Value(param.get("path_csv"), "path.csv")
        self._paths = [[] for _ in range(self.num_UAV)]
        self._path_done = [False] * self.num_UAV
        self.Scene_Random_Reset()

    # ---- scenario bank (the RRT part of UAV.reset, pre-planned) ------------------------------------------
    def _load_bank(self, path, b):
        if path:
            z = np.load(os.path.normpath(path))
            self.backend.load_scenarios(z["start_goal"], z["sub_goals"], z["n_sub"])
            return
        from dqn_based_uav_3d_path_planer_amd.data import load_city26
        c = load_city26()
        if c["buildings"].shape == b.shape and np.allclose(c["buildings"], b, rtol=0, atol=1e-9):
            self.backend.load_scenarios(c["start_goal"], c["sub_goals"], c["n_sub"])
        elif hasattr(self.backend, "plan_scenarios"):
            # any other world: plan the bank on the GPU (UAV.reset's draws + RRT, csrc/rrt.hip)
            self.backend.plan_scenarios(max(1024, min(65536, self.num_envs)), seed=self.seed)
        else:
            raise ValueError("no <scenario_bank> given and no planner available for this world")

    # ---- helpers used by the UAV views -------------------------------------------------------------------
    @property
    def _obs(self):
        """[N, 100] observation rows as the host-side code reads them (packed backends are expanded on demand)."""
        if self._obs_raw is None:
            return None
        return self.backend.unpack(self._obs_raw) if getattr(self.backend, "packed", False) else self._obs_raw

    @_obs.setter
    def _obs(self, v):
        self._obs_raw = v

    def _invalidate(self):
        self._state_cache = None
        self._state0 = None

    def _states(self):
        if self._state_cache is None:
            self._state_cache = self.backend.get_state(0, self.backend.N, want_sub=True)
        return self._state_cache

    def _state_row(self, e, j):
        if e == 0 and self._state_cache is None:            # the scalar accessors of the UAV views read env 0 only
            if self._state0 is None:
                self._state0 = self.backend.get_state(0, self.num_UAV)
            return self._state0[j]
        return self._states()[0][e * self.num_UAV + j]

    def _subgoals(self, e, j):
        st, sub, _ = self._states()
        i = e * self.num_UAV + j
        return sub[i][: int(st[i][11])]

    def _obs_slot(self, j) -> torch.Tensor:
        return self._obs.view(self.num_envs, self.num_UAV, -1)[:, j]

    def _obs_rows(self, j) -> np.ndarray:
        return self._obs_slot(j).float().cpu().numpy()

    def _slot_mask(self, j) -> torch.Tensor:
        m = torch.zeros((self.num_envs, self.num_UAV), dtype=torch.uint8, device=self._obs.device)
        m[:, j] = 1
        return m.reshape(-1).contiguous()

    def _record_paths(self, stepped):
        """Append env 0's new positions (UAV.py:431) for the slots in `stepped`; dump path.csv when one terminates."""
        if not self.record_path:
            return
        st = self.backend.get_state(0, self.num_UAV)
        for j in stepped:
            if self._path_done[j]:
                continue                                    # done agents are skipped by run_thread_OffPolicy (:365)
            self._paths[j].append([float(st[j][0]), float(st[j][1]), float(st[j][2])])
            if st[j][10] != 0:
                self._path_done[j] = True
                try:
                    import csv
                    with open(self.path_csv, "w", newline="") as f:
                        csv.writer(f).writerows(self._paths[j])
                except OSError as e:
                    print(e)

    def _step_slot(self, j, action):
        """Move UAV_j of every env (BaseEnv.Move_Agent semantics for a batch); other slots are left untouched."""
        dev = self._obs.device
        a = torch.as_tensor(np.asarray(action.cpu() if torch.is_tensor(action) else action))
        if a.dtype in (torch.int64, torch.int32, torch.int16, torch.uint8) or isinstance(action, int):
            kind_int = True
            col = a.reshape(-1).to(torch.int32)
        else:
            kind_int = False
            col = a.reshape(-1)[:1].to(torch.float64) if a.numel() in (1, 2) and self.num_envs == 1 else a.reshape(-1).to(torch.float64)
        if col.numel() == 1:
            col = col.expand(self.num_envs)
        full = torch.zeros((self.num_envs, self.num_UAV), dtype=torch.int32 if kind_int else torch.float64, device=dev)
        full[:, j] = col.to(dev)
        out = self.backend.step(full.reshape(-1).contiguous(), active=self._slot_mask(j), skip_done=False)
        self._obs = out.obs
        self._invalidate()
        self._record_paths([j])
        sl = slice(j, None, self.num_UAV)
        return (out.reward.cpu().numpy()[sl], out.ret_done.cpu().numpy()[sl], out.info.cpu().numpy()[sl])

    # ---- reference methods ---------------------------------------------------------------------------------
    def Threaten_rate(self, p: Loc):
        """PathPlan_City.py:215-223 on the device broad phase + exact narrow phase."""
        pts = torch.tensor([[float(p.x), float(p.y), float(p.z)]], dtype=torch.float64)
        return int(self.backend.threaten_rate(pts)[0])

    def _refresh_bank(self):
        """Call right before the reset of every agent that opens an episode: the slice planned while the last episode ran is
        handed over in full (no agent flies a list at an episode boundary), and the next slice of the bank starts planning on
        a low-priority stream beside this episode (csrc/rrt.hip, the LDS-free background form)."""
        if not self.fresh_plans:
            return
        b = self.backend
        m = getattr(b, "n_scenarios", 0)
        if m <= 0:
            return
        if self._plan_stream is None:
            lo = torch.cuda.Stream.priority_range()[0] if hasattr(torch.cuda.Stream, "priority_range") else 0
            self._plan_stream = torch.cuda.Stream(device=b.device, priority=lo)
        if b.replan_ready() >= 0:       # (a slice still being planned is waited for ON THE STREAM: what the bank holds after N
            b.replan_commit(force=True)  #  episodes depends on the seeds only, never on timing -- runs stay reproducible)
        count = min(m, max(1, self.fresh_plans_rows))
        if self._replan_next + count > m:
            self._replan_next = 0
        b.replan_begin(self._replan_next, count, seed=(self.seed + 1) * 1000003 + self._episode, stream=self._plan_stream)
        self._replan_next += count

    def Scene_Random_Reset(self):
        self._episode += 1
        self._refresh_bank()
        self._obs = self.backend.reset(self.seed + self._episode)
        self._invalidate()
        self._paths = [[] for _ in range(self.num_UAV)]                                       # UAV.py:338
        self._path_done = [False] * self.num_UAV

    def Check_uav_Done(self):
        if self._state_cache is not None:
            return bool(self._state_cache[0][:, 10].all())
        return bool(self.backend.get_state(0, self.backend.N)[:, 10].all())

    def Move_Agent(self, index: int, action):
        """BaseEnv.py:123-137 -> (next_state, reward, done, info) of env 0."""
        reward, done, info = self.Agents[index].update(action)
        return self.Agents[index].state(), reward, done, info

    def run(self):
        pass

    def run_XML_scene(self):
        pass

    def Choose_Action2(self, index: int, eps=0.2):
        state = self.Agents[index].state()
        return self.Agents[index].Trainer.get_action(state, eps)

    def Reset_Result(self, eps_rate):
        self.result = {"success": 0, "lose": 0, "meet_threaten": 0, "normal": 0, "loss": 0, "sum_epoch": 0,
                       "eps": eps_rate, "score": 0, "average_score": 0, "step": 0}

    def Run_statistics(self, info):
        self.result[info] = self.result[info] + 1

    def update(self):
        """PathPlan_City.py:757-776: every agent trains on its current transition_dict."""
        re = []
        for uav in self.Agents:
            item = dict(uav.Train_nn())
            item["score"] = uav.score
            item["average_score"] = uav.score
            item["step"] = uav.Step
            item["energy_cost"] = uav.energy_cost_total
            item["task_collect"] = uav.task_collect
            item["Energy_Efficent"] = uav.task_collect / (uav.energy_cost_total + 0.001)
            item["UE_waiting_time"] = 0
            re.append(item)
        return re

    def Train_statistics(self, train_info):
        for info in train_info:
            loss = info["loss"]
            self.result["loss"] += float(loss.item()) if torch.is_tensor(loss) else float(loss)
            self.result["sum_epoch"] += info["sum_epoch"]
            self.result["score"] += info["score"]
            self.result["average_score"] += info["average_score"] / max(len(train_info), 1)
            self.result["step"] += info["step"]

    def Federated_Learning_AC(self):
        """Envs/PathPlan_City.py:590-601: every UAV's actor <- the merge of all UAVs' actors (one launch over the flat
        parameter blocks when the trainers are fused; critics, targets and optimizers untouched)."""
        from dqn_based_uav_3d_path_planer_amd import federated
        if not self._fl_aggregate_given and not self._fl_warned:
            self._fl_warned = True
            print("PathPlan_City: Is_FL = 1 without <FL_Aggregate>: merging the actors as the reference EXECUTES it -- their SUM "
                  "(the division at Envs/PathPlan_City.py:597 never reaches the model).  <FL_Aggregate>mean</FL_Aggregate> averages.")
        self.fl_merged_on = federated.federated_learning_ac([u.Trainer for u in self.Agents], self.FL_Aggregate)

    def Federated_Learning(self):
        """The Is_AC = 0 branch of :469-475.  The reference's own Federated_Learning (:604-640) needs get_policy_DFRL /
        SPN_param / Update_SPN_Soft, which no trainer in the tree has (AttributeError); the merge of Federated_Learning_AC is
        applied to q_local instead (replace_param, Trainer/DuelingDQN_Trainer.py:204-207)."""
        from dqn_based_uav_3d_path_planer_amd import federated
        self.fl_merged_on = federated.federated_learning_q([u.Trainer for u in self.Agents],
                                                           self.FL_Aggregate if self._fl_aggregate_given else "mean")

    def _federated_merge(self):
        """:469-475, after epoch += 1: every FL_Loop episodes when Is_FL."""
        if self.Is_FL and self.FL_Loop > 0 and self.epoch % self.FL_Loop == 0:
            if self.Is_AC:
                self.Federated_Learning_AC()
            else:
                self.Federated_Learning()
            self.fl_merges += 1

    def _rewind_ring(self, surplus: int):
        """Take the passes enqueued behind the end of the episode out of the replay cursor: they stored rows with valid = 0
        (nobody moved) -- the reference stores nothing after its loop has ended.  head goes back by `surplus` frames; the frames
        those passes wrote into are no longer counted as filled (on a ring that had wrapped they overwrote its oldest frames)."""
        if surplus <= 0:
            return
        ring = self._ring
        ring.head = (ring.head - surplus) % ring.frames
        ring.filled = max(0, ring.filled - surplus)
        self._obs_raw = ring.obs[ring.head]
        self._invalidate()

    def _run_eposide_fused(self, eps_rate):
        """run_eposide on the fused path: HotLoop enqueues done_check steps of act -> step (+ replay write) -> sample ->
        learn at a time; the host then reads, in one transfer, how many agents each of those steps still moved and the
        info counts.  A step that moved nobody means every agent had finished before it: the episode ended there (the
        env kernels leave finished agents untouched, so the up to done_check - 1 surplus steps change no env state; the
        learner does take that many extra replay updates -- the one deviation from the reference's loop, which breaks
        right after the update of the last moving step, PathPlan_City.py:456-459)."""
        from dqn_based_uav_3d_path_planer_amd.loop import HotLoop
        self.Reset_Result(eps_rate)
        uav = self.Agents[0]
        tr, ring = uav.Trainer, self._ring
        self._episode += 1
        self._refresh_bank()
        self.backend.reset(self.seed + self._episode, obs=ring.obs[ring.head])      # UAV.reset everywhere; the replay stays
        self._obs_raw = ring.obs[ring.head]
        self._invalidate()
        self._paths, self._path_done = [[] for _ in range(self.num_UAV)], [False] * self.num_UAV
        batch_now = tr.Batch_Size if tr.Is_Train else 0
        if self._hot is not None and not self._hot.in_sync(batch_now):
            self._hot_counter = self._hot.counter   # the Philox stream goes on where the old loop stopped
            self._hot.close()            # Is_Train / Batch_Size / the cursor / the epoch changed behind the C object
            self._hot = None
        if self._hot is None:
            self._hot = HotLoop(ring, tr.learner, tr.Batch_Size if tr.Is_Train else 0, seed=self.seed, eps=eps_rate,
                                learn_start=tr.Batch_Size + 1, auto_reset=False, skip_done=True, info=self._info,
                                per=getattr(self, "_per", None), counter=getattr(self, "_hot_counter", 0), gate_updates=True)
        self._hot.set_eps(eps_rate if tr.Is_Train else 0.0)
        k = 1 if self.record_path else min(self.done_check, ring.frames - 2)
        n_steps, ended = 0, False
        dev = self.backend.device
        epoch0, passes = tr.learner.epoch, 0
        per = getattr(self, "_per", None)
        self.hot_loop_with_per = self._hot._per is not None
        beta0 = per.beta if per is not None else None
        while not ended:
            t0 = ring.head
            self._hot.run(k)
            passes += k
            fr = (t0 + torch.arange(k, device=dev)) % ring.frames
            v = ring.valid[fr].bool()                                                # [k, N] agents moved by each step
            inf = self._info[fr]
            stats = torch.stack([v.sum(1)] + [((inf == c) & v).sum(1) for c in range(3)], 1).cpu().numpy()   # one sync
            for i in range(k):
                if stats[i, 0] == 0:                                                 # nobody moved: finished before this step
                    ended = True
                    break
                n_steps += 1
                self.result["normal"] += int(stats[i, 1])
                self.result["success"] += int(stats[i, 2])
                self.result["lose"] += int(stats[i, 3])
            self._obs_raw = ring.obs[ring.head]
            self._invalidate()
            if self.record_path:
                self._record_paths(range(self.num_UAV))
        # The passes enqueued behind the last moving step changed nothing on the device (HotLoop(gate_updates): the Adam launch and
        # batch_update check the word the step kernel stamps) -- the counters follow: the learner has taken exactly one update per
        # moving step, as the reference's loop, which leaves right after the last one (Envs/PathPlan_City.py:456-459)
        undo = min(passes - n_steps, tr.learner.epoch - epoch0)
        if undo > 0:
            tr.learner.epoch -= undo
            if per is not None:
                per.beta = min(1.0, beta0 + (tr.learner.epoch - epoch0) * per.beta_inc)
        self.surplus_passes_last_episode = passes - n_steps
        if passes > n_steps:             # ... and neither the replay cursor nor the Philox counter remembers them
            self._hot_counter = self._hot.counter - (passes - n_steps)
            self._hot.close()
            self._hot = None
            self._rewind_ring(passes - n_steps)
            if per is not None:
                per.n_entries = ring.filled * self.backend.N
        tr.loss = tr.learner.loss
        item = {"loss": tr.learner.loss, "sum_epoch": tr.epoch, "score": uav.score, "average_score": uav.score,
                "step": uav.Step, "energy_cost": uav.energy_cost_total, "task_collect": uav.task_collect,
                "Energy_Efficent": uav.task_collect / (uav.energy_cost_total + 0.001), "UE_waiting_time": 0}
        self.Train_statistics([item])
        if tr.Is_Train and tr.epoch // tr.save_loop != getattr(self, "_saved_at", 0):   # save() as _learn does, per save_loop updates
            self._saved_at = tr.epoch // tr.save_loop
            tr.save()
        self.steps_last_episode = n_steps
        self.epoch += 1
        if self.epoch % self.print_loop == 0:
            uav.record_list()
        self._federated_merge()
        return self.result

    def _run_eposide_fused_sac(self, eps_rate):
        """run_eposide with one fused SAC trainer per UAV slot (BASELINE configs[3]'s shape): per time step, one
        uavenv_sac_act launch per slot on the packed rows of the current frame -> the env step (replay write included) ->
        per slot a draw of Batch_Size of ITS transitions and the four launches of the fused update (:364-385, :456).
        Nothing is read back inside the loop except, every done_check steps, how many agents each step moved and the info
        counts (as _run_eposide_fused; the same up-to-done_check - 1 surplus updates at the end of an episode)."""
        from dqn_based_uav_3d_path_planer_amd import _lib
        self.Reset_Result(eps_rate)
        ring, U, N = self._ring, self.num_UAV, self.backend.N
        self._episode += 1
        self._refresh_bank()
        self.backend.reset(self.seed + self._episode, obs=ring.obs[ring.head])
        self._obs_raw = ring.obs[ring.head]
        self._invalidate()
        self._paths, self._path_done = [[] for _ in range(U)], [False] * U
        k = 1 if self.record_path else min(self.done_check, ring.frames - 2)
        dev = self.backend.device
        lib = self.backend.lib
        act0, act1 = ring.action.view(-1), self._a1.view(-1)
        n_steps, ended = 0, False
        # the whole step sequence enqueued from C (csrc/loop.hip: uavenv_sac_loop_run) when the slots share one batch size and
        # their replay kind (uniform, or prioritised = the reference's ReplayTree use, SAC_Trainer.py:336-352); the Python loop
        # below issues the same launches one by one (mixed settings, several ranks, <sac_c_loop>0</sac_c_loop>)
        trs = [u.Trainer for u in self.Agents]
        all_per = all(p is not None for p in self._sac_per)
        use_c = ((all(p is None for p in self._sac_per) or all_per) and len({t_.Batch_Size for t_ in trs}) == 1 and
                 len({bool(t_.Is_Train) for t_ in trs}) == 1 and U <= _lib.SAC_LOOP_MAX_SLOTS and
                 int(None2Value(self.param.get("sac_c_loop"), 1)) != 0)
        multi = torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1
        if use_c and multi:
            use_c = False        # several ranks: the learners of the Python loop exchange per phase (sac.py: _exchange); the C loop's
                                 # on-stream exchange is set up by its owner (loop.SACHotLoop(exchange=...), bench.py --config 4 --gpus N)
        if use_c and getattr(self, "_sac_hot", None) is not None and \
                not self._sac_hot.in_sync(trs[0].Batch_Size, bool(trs[0].Is_Train)):
            self._sac_hot.close()        # Is_Train / Batch_Size / the cursor / a learner's counters changed behind the C object
            self._sac_hot = None
        if use_c and getattr(self, "_sac_hot", None) is None:
            from dqn_based_uav_3d_path_planer_amd.loop import SACHotLoop
            self._sac_hot = SACHotLoop(ring, [t_.learner for t_ in trs], trs[0].Batch_Size, seed=self.seed, act1_plane=self._a1,
                                       counter=self._sac_counter, info=self._info, is_train=bool(trs[0].Is_Train),
                                       auto_reset=False, skip_done=True, pers=self._sac_per if all_per else None,
                                       gate_updates=True)
        z = getattr(self, "_sac_noise", None)
        # no update behind a step that moved nobody (see _run_eposide_fused): the C loop gates on the device; the Python loop below
        # hands every learner the same (word, tick) pair
        Ls = [t_.learner for t_ in trs]
        count0 = [(L.epoch, L.adam_steps) for L in Ls]
        beta0 = [None if p is None else p.beta for p in self._sac_per]
        passes = 0
        # Several ranks: every update sits behind a gradient sum over the ranks, so every rank must take the SAME
        # updates.  A rank-local gate (this rank's moved word) or a rank-local end of the episode would let rank A skip -- or never
        # issue -- an update rank B applies: parameters, adam_steps and bias corrections diverge and the exchange loses its partner.
        # So at world_size > 1 nothing is gated on the device, and the episode ends for every rank at the first step that moved
        # nobody on ANY rank (one all_gather_object of the per-step moved counts per <done_check> steps); the up to done_check - 1
        # surplus updates then do happen, identically on every rank, and the counters keep them.
        gate_local = not use_c and not multi
        if gate_local:
            if getattr(self, "_sac_moved", None) is None:
                self._sac_moved = torch.zeros(1, dtype=torch.int32, device=dev)
            _lib.check(lib.uavenv_set_moved_word(self.backend._h, self._sac_moved.data_ptr()), "uavenv_set_moved_word")
        while not ended:
            t0 = ring.head
            nb = self._draws_all.shape[0]
            passes += k
            for _ in range(0 if use_c else k):
                t = ring.head
                self._sac_counter += 1
                # every N(0,1) draw of the step (U get_action's, 2 rsample()'s per update) in one launch
                n_z = U * 2 * self.num_envs + 4 * nb
                if z is None or z.numel() != n_z:
                    z = self._sac_noise = torch.empty(n_z, dtype=torch.float32, device=dev)
                _lib.check(lib.uavenv_randn(self.seed, self._sac_counter, n_z, z.data_ptr(), torch.cuda.current_stream(dev).cuda_stream),
                           "uavenv_randn")
                za = z[:U * 2 * self.num_envs].view(U, self.num_envs, 2)
                zl = z[U * 2 * self.num_envs:].view(2, nb, 2)
                for j, uav in enumerate(self.Agents):
                    uav.Trainer.learner.act_rows(self._flat, t * N + j, U, self.num_envs, act0, act1, eps=za[j])
                ring.step_env(auto_reset=False, skip_done=True, info=self._info)
                go = (self._sac_moved.data_ptr(), int(lib.uavenv_tick(self.backend._h))) if gate_local else None
                for L in Ls:
                    L.go = go
                for j, per in enumerate(self._sac_per):      # ReplayTree.push(error 0) for the slot's rows of the frame just written
                    if per is not None:
                        per.fill(t * self.num_envs, self.num_envs, 0.0, valid=ring.valid[t].view(self.num_envs, U)[:, j].contiguous())
                        per.fill(ring.head * self.num_envs, self.num_envs, zero=True)
                        per.n_entries = ring.filled * self.num_envs
                learn = [uav.Trainer.Is_Train and ring.filled * self.num_envs > uav.Trainer.Batch_Size for uav in self.Agents]   # :383-385
                uniform = [l and self._sac_per[j] is None for j, l in enumerate(learn)]
                nd = 0
                if any(uniform):    # distinct (frame, env) pairs for all slots at once (each slot reads its own rows of them)
                    nd = nb if ring.filled * self.num_envs >= nb and len({t_.Batch_Size for t_ in trs}) == 1 else 0
                    if nd:
                        # (over the valid rows only: finished agents are skipped here, and the reference never stores their rows)
                        _lib.check(lib.uavenv_replay_draw_valid(ring.frames, self.num_envs, ring.head, ring.filled, nd // U, U, U, 0,
                                                                ring.valid.data_ptr(), _lib.DRAW_MAX_TRIES, self.seed + 7,
                                                                self._sac_counter, self._draws_all.data_ptr(),
                                                                torch.cuda.current_stream(dev).cuda_stream), "uavenv_replay_draw_valid")
                o = 0
                for j, uav in enumerate(self.Agents):
                    tr = uav.Trainer
                    if learn[j] and self._sac_per[j] is not None:
                        # ReplayTree.sample -> importance weights -> the fused update (weights in the critic losses, |TD| out) ->
                        # batch_update, all on the stream (:336-352)
                        per, pb = self._sac_per[j], self._sac_per_bufs[j]
                        per.sample_into(tr.Batch_Size, self.seed + 7 + j, self._sac_counter, pb, self.num_envs)
                        tr.learner.learn(self._sac_batches[j], noise=(zl[0, o:o + tr.Batch_Size], zl[1, o:o + tr.Batch_Size]))
                        per.update_f32(pb["slots"], pb["abs"], go=go)
                    elif learn[j]:
                        if not nd:  # the ring does not hold sum(Batch_Size) transitions yet: one draw per slot
                            _lib.check(lib.uavenv_replay_draw_valid(ring.frames, self.num_envs, ring.head, ring.filled, tr.Batch_Size,
                                                                    1, U, j, ring.valid.data_ptr(), _lib.DRAW_MAX_TRIES,
                                                                    self.seed + 7 + j, self._sac_counter, self._draws[j].data_ptr(),
                                                                    torch.cuda.current_stream(dev).cuda_stream),
                                       "uavenv_replay_draw_valid")
                        tr.learner.learn(self._sac_batches[j], noise=(zl[0, o:o + tr.Batch_Size], zl[1, o:o + tr.Batch_Size]))
                    else:
                        tr.learner.epoch += 1                                                   # :322-333
                    o += tr.Batch_Size
            if use_c:
                self._sac_hot.run(k)
                self._sac_counter = self._sac_hot.counter
            fr = (t0 + torch.arange(k, device=dev)) % ring.frames
            v = ring.valid[fr].bool()
            inf = self._info[fr]
            stats = torch.stack([v.sum(1)] + [((inf == c) & v).sum(1) for c in range(3)], 1).cpu().numpy()   # one sync
            moved = stats[:, 0]
            if multi:               # a step counts while ANY rank still moves an agent: the ranks leave the episode together
                every = [None] * torch.distributed.get_world_size()
                torch.distributed.all_gather_object(every, [int(x) for x in moved])
                moved = np.sum(np.asarray(every, dtype=np.int64), axis=0)
            for i in range(k):
                if moved[i] == 0:
                    ended = True
                    break
                n_steps += 1
                self.result["normal"] += int(stats[i, 1])
                self.result["success"] += int(stats[i, 2])
                self.result["lose"] += int(stats[i, 3])
            self._obs_raw = ring.obs[ring.head]
            self._invalidate()
            if self.record_path:
                self._record_paths(range(U))
        if gate_local:
            _lib.check(lib.uavenv_set_moved_word(self.backend._h, None), "uavenv_set_moved_word")
        for L in Ls:
            L.go = None
        surplus = passes - n_steps
        for j, L in enumerate(Ls):           # the counters follow the device: update() ran once per moving step
            if surplus > 0 and not multi:    # (several ranks: the surplus updates were applied, on every rank alike -- see above)
                L.epoch -= min(surplus, L.epoch - count0[j][0])
                L.adam_steps -= min(surplus, L.adam_steps - count0[j][1])
                if self._sac_per[j] is not None:
                    p = self._sac_per[j]
                    p.beta = min(1.0, beta0[j] + (L.adam_steps - count0[j][1]) * p.beta_inc)
        self.surplus_passes_last_episode = surplus
        if surplus > 0:
            if not multi:
                self._sac_counter -= surplus
            if getattr(self, "_sac_hot", None) is not None:
                self._sac_hot.close()
                self._sac_hot = None
            self._rewind_ring(surplus)
            for p in self._sac_per:
                if p is not None:
                    p.n_entries = ring.filled * self.num_envs
        items = []
        for uav in self.Agents:
            tr = uav.Trainer
            tr.loss = tr.learner.loss
            items.append({"loss": tr.learner.loss, "sum_epoch": tr.epoch, "score": uav.score, "average_score": uav.score,
                          "step": uav.Step, "energy_cost": uav.energy_cost_total, "task_collect": uav.task_collect,
                          "Energy_Efficent": uav.task_collect / (uav.energy_cost_total + 0.001), "UE_waiting_time": 0})
            if tr.Is_Train and tr.epoch // tr.save_loop != getattr(tr, "_saved_at", 0):
                tr._saved_at = tr.epoch // tr.save_loop
                tr.save()
        self.Train_statistics(items)
        self.steps_last_episode = n_steps
        self.sac_c_loop_used = bool(use_c)          # (tests: which of the two issue paths ran)
        self.epoch += 1
        if self.epoch % self.print_loop == 0:
            for uav in self.Agents:
                uav.record_list()
        self._federated_merge()
        return self.result

    def _run_eposide_on_policy(self, eps_rate=0.1):
        """PathPlan_City.py:419-436 + run_thread_OnPolicy (:386-406), for all vectorised envs at once: per time step every UAV that is
        not done acts (Choose_Action2 -> Trainer.get_action, :338-346), moves, and APPENDS the transition to its own transition_dict;
        nothing is pushed to a replay memory and nothing is learnt until every agent is done -- then ONE update() (:436), i.e. one
        Train_nn per UAV on its whole episode.  (The reference starts one thread per UAV per step and joins them all: the UAVs do not
        interact, so the result is that of the loop below.)"""
        self.Reset_Result(eps_rate)
        self.Scene_Random_Reset()
        names = ("normal", "success", "lose")
        keys = ("states", "actions", "next_states", "rewards", "dones")
        acc = [{k: [] for k in keys} for _ in self.Agents]
        while True:
            self.run()
            states = self._obs
            s_view = states.view(self.num_envs, self.num_UAV, -1)
            chosen = [uav.Trainer.get_action_batch(s_view[:, j], eps_rate).to(states.device) for j, uav in enumerate(self.Agents)]
            continuous = any(c.dim() == 2 for c in chosen)
            actions = torch.zeros((self.num_envs, self.num_UAV), dtype=torch.float32 if continuous else torch.int32, device=states.device)
            for j, c in enumerate(chosen):
                if c.dim() == 2:
                    actions[:, j] = c[:, 0].float()
                elif continuous:
                    actions[:, j] = -1.0 + 2.0 * c.float() / (self.Agents[j].Trainer.act_num - 1)
                else:
                    actions[:, j] = c.to(torch.int32)
            out = self.backend.step(actions.reshape(-1).contiguous(), skip_done=True)     # `if uav.done: return` (:388-389)
            self._obs = out.obs
            self._invalidate()
            self._record_paths(range(self.num_UAV))
            info = out.info.view(self.num_envs, self.num_UAV)
            valid = out.valid.view(self.num_envs, self.num_UAV).bool()
            for k, nm in enumerate(names):
                self.result[nm] += int(((info == k) & valid).sum().item())                # Run_statistics (:395)
            n_view = out.obs.view(self.num_envs, self.num_UAV, -1)
            r_view = out.reward32.view(self.num_envs, self.num_UAV)
            d_view = out.ret_done.view(self.num_envs, self.num_UAV)
            for j in range(self.num_UAV):
                keep = valid[:, j]
                a = acc[j]
                a["states"].append(s_view[keep, j].float())
                a["actions"].append(chosen[j][keep])
                a["next_states"].append(n_view[keep, j].float())
                a["rewards"].append(r_view[keep, j].float())
                a["dones"].append(d_view[keep, j].float())
            if self.Check_uav_Done():
                break
        for j, uav in enumerate(self.Agents):
            uav.transition_dict = {k: torch.cat(acc[j][k]) for k in keys}
        train_info = self.update()                                                        # :436, after the episode
        self.Train_statistics(train_info)
        self.epoch += 1
        if self.epoch % self.print_loop == 0:
            for uav in self.Agents:
                uav.record_list()
        self._federated_merge()
        return self.result

    def run_eposide(self, eps_rate=0.1):
        """PathPlan_City.py:410-478 (off-policy branch) for all vectorised envs at once: reset, then
        act -> step -> store -> sample -> learn per time step until every agent of every env is done."""
        if self.Is_On_Policy == 1:
            return self._run_eposide_on_policy(eps_rate)
        if self.fast:
            return self._run_eposide_fused(eps_rate)
        if self.fast_sac:
            return self._run_eposide_fused_sac(eps_rate)
        self.Reset_Result(eps_rate)
        self.Scene_Random_Reset()
        names = ("normal", "success", "lose")
        train_info = []
        while True:
            self.run()
            states = self._obs
            s_view = states.view(self.num_envs, self.num_UAV, -1)
            chosen = []
            for j, uav in enumerate(self.Agents):
                uav.transition_dict = {"states": [], "actions": [], "next_states": [], "rewards": [], "dones": []}
                chosen.append(uav.Trainer.get_action_batch(s_view[:, j], eps_rate).to(states.device))
            continuous = any(c.dim() == 2 for c in chosen)          # SAC: [n, action_dim], only [:, 0] steers (UAV.py:414)
            actions = torch.zeros((self.num_envs, self.num_UAV), dtype=torch.float32 if continuous else torch.int32,
                                  device=states.device)
            for j, c in enumerate(chosen):
                if c.dim() == 2:
                    actions[:, j] = c[:, 0].float()
                elif continuous:                                     # mixed: index -> steer with the documented table
                    actions[:, j] = -1.0 + 2.0 * c.float() / (self.Agents[j].Trainer.act_num - 1)
                else:
                    actions[:, j] = c.to(torch.int32)
            out = self.backend.step(actions.reshape(-1).contiguous(), skip_done=True)     # done agents wait (:365-366)
            self._obs = out.obs
            self._invalidate()
            self._record_paths(range(self.num_UAV))
            info = out.info.view(self.num_envs, self.num_UAV)
            valid = out.valid.view(self.num_envs, self.num_UAV).bool()
            for k, nm in enumerate(names):
                self.result[nm] += int(((info == k) & valid).sum().item())
            n_view = out.obs.view(self.num_envs, self.num_UAV, -1)
            r_view = out.reward32.view(self.num_envs, self.num_UAV)
            d_view = out.ret_done.view(self.num_envs, self.num_UAV)
            for j, uav in enumerate(self.Agents):
                mem = uav.Trainer.replay_memory
                mem.add_batch(s_view[:, j].float(), chosen[j], r_view[:, j], n_view[:, j].float(), d_view[:, j],
                              valid=valid[:, j])
                if len(mem.buffer) > uav.Trainer.Batch_Size:                                  # :383-385
                    b = mem.sample_tensors(uav.Trainer.Batch_Size)
                    uav.transition_dict = {"states": b["states"], "actions": b["actions"], "next_states": b["next_states"],
                                           "rewards": b["rewards"], "dones": b["dones"], "idx": b.get("idx"),
                                           "weights": b.get("weights")}
            train_info = self.update()                                                        # :456
            if self.Check_uav_Done():
                break
        self.Train_statistics(train_info)
        self.epoch += 1
        if self.epoch % self.print_loop == 0:
            for uav in self.Agents:
                uav.record_list()
        self._federated_merge()
        return self.result
