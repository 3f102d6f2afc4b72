"""Shared implementation of the DQN-family trainer plugins (reference: BaseClass/BaseTrainer.py:20-47,
Trainer/DQN_Trainer.py, DDQN_Trainer.py, DuelingDQN_Trainer.py, BaseClass/replay_buffer.py:28-54).

Same constructor contract (`param` = the Trainer.xml dict of strings), same attribute and method names, same
result dict {'sum_epoch', 'loss'}; the arithmetic is dqn_based_uav_3d_path_planer_amd.learner (parity-tested
against the executed reference trainers).  Where the reference is broken as shipped (SURVEY.md App. C.1-C.4:
missing get_action/update on DQN/DDQN, Push_Replay arity, save() format string) the working behaviour of
DuelingDQN_Trainer is used for all three.
"""
from __future__ import annotations

import os
import random

import numpy as np
import torch

from dqn_based_uav_3d_path_planer_amd.compat import None2Value
from dqn_based_uav_3d_path_planer_amd.factories import NetworkFactory
from dqn_based_uav_3d_path_planer_amd.learner import DQNLearner, FusedDQNLearner


class _LenProxy:
    """`len(replay_memory.buffer)` / `len(replay_memory.memory)` as the reference's callers use them."""

    def __init__(self, owner):
        self._o = owner

    def __len__(self):
        return self._o.size


class VecReplayMemory:
    """ReplayMemory (replay_buffer.py:28-54) as a tensor ring on the trainer's device, with batched insert."""

    def __init__(self, capacity: int, device, obs_dim: int = 100, prioritized: bool = False):
        self.capacity, self.device, self.obs_dim = int(capacity), torch.device(device), obs_dim
        # IsPriority_Replay == 1 (BaseClass/replay_buffer.py:121-223 ReplayTree): priorities live in a DevicePER next to
        # the tensors; new transitions get the reference's push priority (|0| + epsilon) ** alpha
        self.per = None
        if prioritized:
            from dqn_based_uav_3d_path_planer_amd.replay import DevicePER
            self.per = DevicePER(self.capacity, device=self.device, tree_order=False)
        self._last_slots = None
        self.position = 0
        self.size = 0
        self.buffer = _LenProxy(self)
        self.memory = _LenProxy(self)
        d = self.device
        self.states = torch.zeros((self.capacity, obs_dim), dtype=torch.float32, device=d)
        self.next_states = torch.zeros((self.capacity, obs_dim), dtype=torch.float32, device=d)
        self.actions = torch.zeros(self.capacity, dtype=torch.int64, device=d)
        self.rewards = torch.zeros(self.capacity, dtype=torch.float32, device=d)
        self.dones = torch.zeros(self.capacity, dtype=torch.float32, device=d)

    def __len__(self):
        return self.size

    def add_batch(self, states, actions, rewards, next_states, dones, valid=None):
        t = lambda x, dt: torch.as_tensor(x, device=self.device).to(dt)   # noqa: E731
        s, ns = t(states, torch.float32).reshape(-1, self.obs_dim), t(next_states, torch.float32).reshape(-1, self.obs_dim)
        a = t(actions, self.actions.dtype).reshape(len(s), *self.actions.shape[1:])      # [n] indices or [n, action_dim]
        r, d = t(rewards, torch.float32).reshape(-1), t(dones, torch.float32).reshape(-1)
        if valid is not None:
            keep = torch.as_tensor(valid, device=self.device).reshape(-1).bool()
            s, ns, a, r, d = s[keep], ns[keep], a[keep], r[keep], d[keep]
        n = len(a)
        if n == 0:
            return
        if n > self.capacity:
            s, ns, a, r, d = s[-self.capacity:], ns[-self.capacity:], a[-self.capacity:], r[-self.capacity:], d[-self.capacity:]
            n = self.capacity
        idx = (self.position + torch.arange(n, device=self.device)) % self.capacity      # FIFO overwrite (deque maxlen)
        self.states[idx], self.next_states[idx], self.actions[idx], self.rewards[idx], self.dones[idx] = s, ns, a, r, d
        if self.per is not None:                                                   # ReplayTree.push(error = 0), :136-144
            first = self.position
            head = min(n, self.capacity - first)
            self.per.fill(first, head, 0.0)
            if n > head:
                self.per.fill(0, n - head, 0.0)
        self.position = (self.position + n) % self.capacity
        self.size = min(self.size + n, self.capacity)
        if self.per is not None:
            self.per.n_entries = self.size

    def add(self, state, action, reward, next_state, done):                       # replay_buffer.py:41-42
        a = int(action.item()) if torch.is_tensor(action) else int(np.asarray(action).reshape(-1)[0])
        self.add_batch(np.asarray(state, dtype=np.float32).reshape(1, -1), [a], [float(reward)],
                       np.asarray(next_state, dtype=np.float32).reshape(1, -1), [float(done)])

    def push(self, batch, error=0):                                               # replay_buffer.py:36-39
        state, action, reward, next_state, done = batch
        self.add_batch(state, torch.as_tensor(action).reshape(-1), torch.as_tensor(reward).reshape(-1), next_state,
                       torch.as_tensor(done).reshape(-1))

    def sample_tensors(self, batch_size: int) -> dict:
        if self.size <= 0:
            raise ValueError("sample from an empty replay memory")
        if self.per is not None:                                                   # ReplayTree.sample, :146-180
            self._per_calls = getattr(self, "_per_calls", 0) + 1
            slots, w, _ = self.per.sample(batch_size, seed=0x9E37, counter=self._per_calls)
            self._last_slots = slots
            return dict(states=self.states[slots], actions=self.actions[slots], rewards=self.rewards[slots],
                        next_states=self.next_states[slots], dones=self.dones[slots], idx=slots, weights=w.float())
        if batch_size <= self.size:
            idx = torch.randperm(self.size, device=self.device)[:batch_size]      # random.sample: without replacement
        else:
            idx = torch.randint(0, self.size, (batch_size,), device=self.device)
        return dict(states=self.states[idx], actions=self.actions[idx], rewards=self.rewards[idx],
                    next_states=self.next_states[idx], dones=self.dones[idx])

    def batch_update(self, slots, abs_errors):                                    # ReplayTree.batch_update, :215-222
        if self.per is not None:
            # the list ReplayTree.sample just returned is in prefix order already (equal slots adjacent): no ordering pass
            self.per.update(slots, abs_errors, assume_sorted=slots is getattr(self, "_last_slots", None))

    def sample2(self, batch_size):                                                # replay_buffer.py:48-51
        b = self.sample_tensors(batch_size)
        return (b["states"].cpu().numpy(), tuple(b["actions"].cpu().tolist()), tuple(b["rewards"].cpu().tolist()),
                b["next_states"].cpu().numpy(), tuple(b["dones"].cpu().tolist()), None, None)


class BaseDQNTrainer:
    KIND = "dqn"
    FILE_TAG = ""          # reference file names: q_local_<tag><name>.pth

    def __init__(self, param: dict) -> None:
        self.param = param
        self.h = int(None2Value(param.get("h"), 1))
        self.w = int(None2Value(param.get("w"), 1))
        self.channel = int(None2Value(param.get("channel"), 1))
        self.output = int(None2Value(param.get("output"), 1))
        self.act_num = self.output
        self.name = param.get("name")
        self.replay_size = int(None2Value(param.get("replay_size"), 1000))
        self.LEARNING_RATE = float(None2Value(param.get("LEARNING_RATE"), 0.001))
        self.Batch_Size = int(None2Value(param.get("Batch_Size"), 128))
        self.gamma = float(None2Value(param.get("gamma"), 0.99))
        self.max_epoch = int(None2Value(param.get("max_epoch"), 100000))
        self.save_loop = int(None2Value(param.get("save_loop"), 10))
        self.Is_Train = int(None2Value(param.get("Is_Train"), 1))
        self.Update_loop = int(None2Value(param.get("Update_loop"), 3))
        dev = param.get("device") or ("cuda:0" if torch.cuda.is_available() else "cpu")
        self.device = torch.device(dev)
        self.NetworkFactory = NetworkFactory()
        self.IsPriority_Replay = int(None2Value(param.get("IsPriority_Replay"), 0))
        # The fused HIP learner (csrc/learner.hip) when the net has the reference's shape and a GPU is there; the
        # PyTorch-ROCm learner otherwise (<fused>0</fused> forces it).  Same attribute surface either way.
        hid = int(None2Value(param.get("hiden_dim"), 64))
        n2 = self.output + (1 if self.KIND == "dueling" else 0)
        want = param.get("fused")
        fusable = (self.device.type == "cuda" and torch.cuda.is_available() and self.w == 100 and hid == 64 and
                   self.output >= 2 and n2 + 2 <= 16)
        self.fused = fusable and (want is None or int(want) != 0)
        loss_kind = param.get("loss") or "mse"
        if self.fused:
            self.learner = FusedDQNLearner(param, self.KIND, device=self.device, lr=self.LEARNING_RATE, gamma=self.gamma,
                                           update_loop=self.Update_loop, loss=loss_kind, mfma=param.get("mfma") or "f32")
        else:
            self.learner = DQNLearner(param, self.KIND, device=self.device, lr=self.LEARNING_RATE, gamma=self.gamma,
                                      update_loop=self.Update_loop, loss=loss_kind)
        self.replay_memory = VecReplayMemory(self.replay_size, self.device, self.w, prioritized=self.IsPriority_Replay == 1)
        self.mse_loss = torch.nn.MSELoss()
        self.model_dir = param.get("model_dir") or os.path.join(os.getcwd(), "Mod")
        self.loss = 0
        self.Load_Mod()

    # -- reference attribute surface --------------------------------------------------------------
    @property
    def q_local(self):
        return self.learner.q_local

    @property
    def q_target(self):
        return self.learner.q_target

    @property
    def optim(self):
        return self.learner.optim

    @property
    def epoch(self):
        return self.learner.epoch

    @epoch.setter
    def epoch(self, v):
        self.learner.epoch = int(v)

    # -- acting --------------------------------------------------------------------------------------
    def get_action(self, state, eps):
        """DuelingDQN_Trainer.py:86-97: greedy w.p. 1-eps (always if not training), else randrange(act_num)."""
        sample = random.random()
        if sample > eps or self.Is_Train == 0:
            s = torch.as_tensor(np.asarray(state, dtype=np.float32), device=self.device).reshape(1, -1)
            with torch.no_grad():
                return int(self.q_local(s).max(1)[1].item())
        return random.randrange(self.act_num)

    def get_action_batch(self, states: torch.Tensor, eps: float) -> torch.Tensor:
        """The same policy for [n, 100] states at once -> int32 [n]."""
        s = states.to(self.device).float()
        with torch.no_grad():
            greedy = self.q_local(s).max(1)[1].to(torch.int32)
        if self.Is_Train == 0 or eps <= 0:
            return greedy
        explore = torch.rand(len(s), device=self.device) <= eps
        rnd = torch.randint(0, self.act_num, (len(s),), device=self.device, dtype=torch.int32)
        return torch.where(explore, rnd, greedy)

    def get_policy(self, state):                                                   # DuelingDQN_Trainer.py:192-198
        s = torch.as_tensor(np.asarray(state, dtype=np.float32), device=self.device).reshape(1, -1)
        probs = self.q_local(s)
        out = torch.zeros_like(probs)
        out[0][torch.argmax(probs)] = 1
        return out

    # -- learning ------------------------------------------------------------------------------------
    def _learn(self, batch: dict):
        if self.Is_Train and batch.get("weights") is not None and batch.get("idx") is not None:
            # prioritised replay: importance weights into the loss, |TD error| back into the priorities
            self.loss, abs_err = self.learner.learn_weighted(batch, batch["weights"].to(self.device))
            self.replay_memory.batch_update(batch["idx"], abs_err)
        elif self.Is_Train:
            self.loss = self.learner.learn(batch)          # epoch += 1, hard update every Update_loop inside
        else:
            self.learner.epoch += 1
            if self.learner.epoch % self.Update_loop == 0:
                self.hard_update()
        if self.epoch % self.save_loop == 0:
            self.save()

    def update(self, transition_dict):
        """DuelingDQN_Trainer.py:150-190 (the 'general' update on a pre-sampled batch)."""
        states = transition_dict.get("states") if transition_dict else None
        if states is None or len(states) == 0:
            self.learner.epoch += 1
            return {"sum_epoch": self.epoch, "loss": self.loss}
        t = lambda x, dt: torch.as_tensor(np.asarray(x) if not torch.is_tensor(x) else x).to(self.device, dt)  # noqa
        batch = dict(states=t(states, torch.float32), actions=t(transition_dict["actions"], torch.int64).reshape(-1),
                     rewards=t(transition_dict["rewards"], torch.float32).reshape(-1),
                     next_states=t(transition_dict["next_states"], torch.float32),
                     dones=t(transition_dict["dones"], torch.float32).reshape(-1))
        if transition_dict.get("weights") is not None and transition_dict.get("idx") is not None:
            batch["weights"], batch["idx"] = transition_dict["weights"], transition_dict["idx"]
        result = {"sum_epoch": self.epoch + 1, "loss": self.loss}    # the reference reports the PREVIOUS loss
        self._learn(batch)
        return result

    def learn_off_policy(self):
        """DQN_Trainer.py:85-136: sample Batch_Size from the trainer's own replay memory and learn."""
        if len(self.replay_memory) < self.Batch_Size:
            self.learner.epoch += 1
            return {"sum_epoch": self.epoch, "loss": self.loss}
        result = {"sum_epoch": self.epoch + 1, "loss": self.loss}
        self._learn(self.replay_memory.sample_tensors(self.Batch_Size))
        return result

    def hard_update(self):
        self.learner.hard_update()

    def replace_param(self, target):
        for tp, p in zip(target.parameters(), self.q_local.parameters()):
            p.data.copy_(tp.data)

    def replace_target_param(self, target):
        for tp, p in zip(target.parameters(), self.q_target.parameters()):
            p.data.copy_(tp.data)

    def Push_Replay(self, Experience, error=0):
        self.replay_memory.push(Experience, error)

    # -- checkpoints: the reference's file names and dict keys (DuelingDQN_Trainer.py:74-84) --------
    def _paths(self, directory=None):
        d = directory or self.model_dir
        n = self.name if self.name is not None else ""
        return (os.path.join(d, f"q_target_{self.FILE_TAG}{n}.pth"), os.path.join(d, f"q_local_{self.FILE_TAG}{n}.pth"))

    def save(self, directory=None):
        pt, pl = self._paths(directory)
        os.makedirs(os.path.dirname(pt), exist_ok=True)
        cpu = lambda sd: {k: v.detach().cpu() for k, v in sd.items()}   # noqa: E731
        torch.save({"model": cpu(self.q_target.state_dict()), "optimizer": self.optim.state_dict(), "epoch": self.epoch}, pt)
        torch.save({"model": cpu(self.q_local.state_dict()), "optimizer": self.optim.state_dict(), "epoch": self.epoch}, pl)

    def Load_Mod(self, Mod_path=None):
        pt, pl = self._paths(Mod_path)
        if os.path.exists(pt) and os.path.exists(pl):
            try:
                mt, ml = torch.load(pt, map_location=self.device), torch.load(pl, map_location=self.device)
                self.q_target.load_state_dict(mt["model"])
                self.q_local.load_state_dict(ml["model"])
                self.optim.load_state_dict(ml["optimizer"])
                self.epoch = ml["epoch"]
            except Exception as e:
                print(e.args)

    # -- setters kept for API compatibility (BaseTrainer.py:68-92) -------------------------------
    def set_replay_size(self, replay_size: int):
        self.replay_size = replay_size

    def set_LEARNING_RATE(self, LEARNING_RATE: float):
        self.LEARNING_RATE = LEARNING_RATE

    def set_Batch_Size(self, Batch_Size: int):
        self.Batch_Size = Batch_Size

    def set_gamma(self, gamma: float):
        self.gamma = gamma

    def set_max_epoch(self, max_epoch: int):
        self.max_epoch = max_epoch

    def set_save_loop(self, save_loop: int):
        self.save_loop = save_loop
