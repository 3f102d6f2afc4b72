"""Generate tests/golden/learner_*.npz by EXECUTING the reference trainers on a scratch copy.

TEST INFRASTRUCTURE (build container only).  Pins Trainer/DQN_Trainer.py:85-136, Trainer/DDQN_Trainer.py:72-117,
Trainer/DuelingDQN_Trainer.py:99-190 (learn_off_policy and update) with injected weights and an injected batch,
so nothing depends on torch's RNG stream (reference pins torch==1.11, the container has 2.10).
"""
from __future__ import annotations

import os
import random
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "tests", "golden")
sys.path.insert(0, HERE)
from ref_harness import RefSession  # noqa: E402

B, W, HID, A = 64, 100, 64, 3
N_UPDATES = 7     # crosses two hard target copies (Update_loop = 3)


def make_param(net, trainer):
    return {"Trainer_Type": trainer, "NetWork": net, "w": str(W), "hiden_dim": str(HID), "output": str(A),
            "h": "1", "channel": "1", "Batch_Size": str(B), "LEARNING_RATE": "0.001", "gamma": "0.99",
            "replay_size": "1000", "save_loop": str(10 ** 9), "Update_loop": "3", "Is_Train": "1",
            "name": "golden"}


def sd_to_np(sd):
    return {k: v.detach().cpu().numpy().copy() for k, v in sd.items()}


def gen(s, trainer_name, net):
    from FactoryClass.TrainerFactory import TrainerFactory
    tr = TrainerFactory().Create_Trainer(make_param(net, trainer_name))
    assert tr is not None, trainer_name
    tr.save = lambda *a, **k: None
    g = torch.Generator().manual_seed(1234)
    # injected weights (independent local / target, like the reference's two separate inits)
    for netobj, scale in ((tr.q_local, 0.15), (tr.q_target, 0.12)):
        with torch.no_grad():
            for p in netobj.parameters():
                p.copy_(torch.randn(p.shape, generator=g) * scale)
    w_local0, w_target0 = sd_to_np(tr.q_local.state_dict()), sd_to_np(tr.q_target.state_dict())
    rng = np.random.default_rng(99)
    states = rng.normal(0, 1, (B, W)).astype(np.float32)
    next_states = rng.normal(0, 1, (B, W)).astype(np.float32)
    actions = rng.integers(0, A, B).astype(np.int64)
    rewards = (rng.normal(0, 1, B) * np.where(rng.random(B) < 0.1, 150.0, 1.0)).astype(np.float32)
    dones = (rng.random(B) < 0.25).astype(np.float32)
    losses = []
    if trainer_name == "DuelingDQN_Trainer":
        # lists, not ndarrays: `transition_dict['states']==[]` (DuelingDQN_Trainer.py:155) raises on numpy >= 2
        td = {"states": states.tolist(), "actions": tuple(int(a) for a in actions),
              "rewards": tuple(float(r) for r in rewards), "next_states": next_states.tolist(),
              "dones": tuple(float(d) for d in dones)}
        for _ in range(N_UPDATES):
            tr.update(td)
            losses.append(float(tr.loss))
    else:
        FloatTensor = torch.FloatTensor
        for i in range(B):
            exp = (FloatTensor(states[i:i + 1]), torch.tensor([[int(actions[i])]]), FloatTensor([[float(rewards[i])]]),
                   FloatTensor(next_states[i:i + 1]), FloatTensor([[float(dones[i])]]))
            tr.replay_memory.push(exp, 0)          # 2-arg form (Push_Replay's 1-arg call is broken)
        random.seed(7)
        for _ in range(N_UPDATES):
            tr.learn_off_policy()                  # random.sample(memory, B) with len(memory)==B: a permutation
            losses.append(float(tr.loss))
    out = dict(states=states, next_states=next_states, actions=actions, rewards=rewards, dones=dones,
               losses=np.array(losses), epoch=int(tr.epoch))
    for pref, d in (("l0_", w_local0), ("t0_", w_target0), ("l1_", sd_to_np(tr.q_local.state_dict())),
                    ("t1_", sd_to_np(tr.q_target.state_dict()))):
        for k, v in d.items():
            out[pref + k] = v
    np.savez_compressed(os.path.join(OUT, f"learner_{trainer_name}.npz"), **out)
    print(trainer_name, "losses", losses)


# The DQN-family updates on PACKED-REPRESENTABLE observations (round 5): rows the reference's own state_PathPlan produced
# (tests/golden/episodes.npz), so that the bench's learner kernels -- k_dqn_grad_packed8 / k_dqn_grad_packed + k_dqn_reduce_adam,
# which read 80-byte packed rows -- run on EXACTLY the inputs the executed learn_off_policy / update saw
# (tests/test_learner_fused_gpu.py::test_packed_row_learner_against_the_executed_reference).  Two 64-sample tiles.
def gen_packed(s, trainer_name, net):
    from FactoryClass.TrainerFactory import TrainerFactory
    Bp = 128
    param = make_param(net, trainer_name)
    param["Batch_Size"] = str(Bp)
    tr = TrainerFactory().Create_Trainer(param)
    assert tr is not None, trainer_name
    tr.save = lambda *a, **k: None
    g = torch.Generator().manual_seed(2468)
    for netobj, scale in ((tr.q_local, 0.15), (tr.q_target, 0.12)):
        with torch.no_grad():
            for p in netobj.parameters():
                p.copy_(torch.randn(p.shape, generator=g) * scale)
    w_local0, w_target0 = sd_to_np(tr.q_local.state_dict()), sd_to_np(tr.q_target.state_dict())
    ep = np.load(os.path.join(OUT, "episodes.npz"))
    obs, outs, off = ep["obs"], ep["outs"], ep["offsets"]
    rng = np.random.default_rng(13)
    ok = np.ones(len(obs), bool)
    ok[off[1:] - 1] = False                                        # s' must be the next record of the same episode
    pick = rng.choice(np.nonzero(ok)[0], Bp, replace=False)
    states = obs[pick].astype(np.float32)
    next_states = obs[pick + 1].astype(np.float32)
    actions = rng.integers(0, A, Bp).astype(np.int64)
    rewards = outs[pick, 0].astype(np.float32)
    dones = (rng.random(Bp) < 0.2).astype(np.float32)
    losses = []
    if trainer_name == "DuelingDQN_Trainer":
        td = {"states": states.tolist(), "actions": tuple(int(a) for a in actions),
              "rewards": tuple(float(r) for r in rewards), "next_states": next_states.tolist(),
              "dones": tuple(float(d) for d in dones)}
        for _ in range(N_UPDATES):
            tr.update(td)
            losses.append(float(tr.loss))
    else:
        FloatTensor = torch.FloatTensor
        for i in range(Bp):
            exp = (FloatTensor(states[i:i + 1]), torch.tensor([[int(actions[i])]]), FloatTensor([[float(rewards[i])]]),
                   FloatTensor(next_states[i:i + 1]), FloatTensor([[float(dones[i])]]))
            tr.replay_memory.push(exp, 0)
        random.seed(7)
        for _ in range(N_UPDATES):
            tr.learn_off_policy()                  # random.sample(memory, B) with len(memory) == B: a permutation
            losses.append(float(tr.loss))
    out = dict(states=states, next_states=next_states, actions=actions, rewards=rewards, dones=dones,
               losses=np.array(losses), epoch=int(tr.epoch), episode_rows=pick)
    for pref, d in (("l0_", w_local0), ("t0_", w_target0), ("l1_", sd_to_np(tr.q_local.state_dict())),
                    ("t1_", sd_to_np(tr.q_target.state_dict()))):
        for k, v in d.items():
            out[pref + k] = v
    np.savez_compressed(os.path.join(OUT, f"learner_{trainer_name}_packed.npz"), **out)
    print(trainer_name, "(packed-representable rows) losses", losses)


PACKED_CASES = (("DQN_Trainer", "Qnet2"), ("DDQN_Trainer", "Qnet2"), ("DuelingDQN_Trainer", "VAnet2"))


def main():
    s = RefSession()
    try:
        gen(s, "DQN_Trainer", "Qnet2")
        gen(s, "DDQN_Trainer", "Qnet2")
        gen(s, "DuelingDQN_Trainer", "VAnet2")
        gen_sac(s)
        gen_sac_packed(s)
        for _name, _net in PACKED_CASES:
            gen_packed(s, _name, _net)
        # epsilon schedule, simulator.py:141-145
        sim = s.sim
        eps = []
        for ep in range(0, 40):
            sim.epoch = ep
            eps.append(sim.epsilon_annealing())
        np.savez(os.path.join(OUT, "epsilon.npz"), epoch=np.arange(40), eps=np.array(eps),
                 min_eps=sim.min_eps, max_eps_episode=sim.max_eps_episode)
    finally:
        s.close()




# ---------------------------------------------------------------------------------------------------------------
# SAC continuous (Trainer/SAC_Trainer.py:325-379, nets BaseClass/BaseCNN.py:459-500): BASELINE config 4's trainer.
# The update is stochastic (Normal.rsample); torch.manual_seed(SEED + k) before update k pins the two draws
# (actor(next_states) in calc_target, then actor(states)); the noise itself is recorded too.
def gen_sac(s):
    from FactoryClass.TrainerFactory import TrainerFactory
    param = {"Trainer_Type": "SAC_Trainer", "Is_Train": "1", "IsPriority_Replay": "0",
             "actor": {"NetWork": "PolicyNetContinuous_SAC", "h": "1", "w": "100", "channel": "1", "action_bound": "1",
                       "hiden_dim": "64", "output": "2", "lr": "0.0001"},
             "critic": {"NetWork": "QValueNetContinuous_SAC", "h": "1", "w": "100", "channel": "1", "hiden_dim": "64",
                        "action_dim": "2", "lr": "0.001"},
             "SAC_param": {"IS_Continuous": "1", "alpha_lr": "0.0001", "target_entropy": "1", "gamma": "0.99", "tau": "0.05"},
             "Priority_Replay": "0", "replay_size": "10000", "LEARNING_RATE": "0.0005", "Batch_Size": "64",
             "max_epoch": "100", "save_loop": str(10 ** 9), "name": "golden"}
    tr = TrainerFactory().Create_Trainer(param)
    assert tr is not None
    tr.save = lambda *a, **k: None
    g = torch.Generator().manual_seed(4321)
    nets = {"actor": tr.actor, "critic_1": tr.critic_1, "critic_2": tr.critic_2}
    for net in nets.values():
        with torch.no_grad():
            for p in net.parameters():
                p.copy_(torch.randn(p.shape, generator=g) * 0.1)
    tr.target_critic_1.load_state_dict(tr.critic_1.state_dict())
    tr.target_critic_2.load_state_dict(tr.critic_2.state_dict())
    out = {}
    for name, net in nets.items():
        for k, v in sd_to_np(net.state_dict()).items():
            out[f"{name}0_{k}"] = v
    rng = np.random.default_rng(7)
    B = 64
    states = rng.normal(0, 1, (B, 100)).astype(np.float32)
    next_states = rng.normal(0, 1, (B, 100)).astype(np.float32)
    actions = rng.uniform(-1, 1, (B, 2)).astype(np.float32)
    rewards = (rng.normal(0, 1, B) * np.where(rng.random(B) < 0.1, 150.0, 1.0)).astype(np.float32)
    dones = (rng.random(B) < 0.25).astype(np.float32)
    for i in range(B):
        tr.replay_memory.push((0, 0, 0, 0, 0), 0)     # update() only checks len(memory) >= Batch_Size (:333)
    td = {"states": states.tolist(), "actions": actions.tolist(), "rewards": rewards.tolist(),
          "next_states": next_states.tolist(), "dones": dones.tolist()}
    SEED, K = 1000, 5
    losses, alphas, noise = [], [], []
    for k in range(K):
        torch.manual_seed(SEED + k)
        n1, n2 = torch.randn(B, 2), torch.randn(B, 2)     # what the two rsample() calls will draw
        noise.append(np.stack([n1.numpy(), n2.numpy()]))
        torch.manual_seed(SEED + k)
        res = tr.update(td)
        losses.append(float(res["loss"]))
        alphas.append(float(tr.log_alpha))
    # deterministic acting check: get_action under a seed
    torch.manual_seed(77)
    act = tr.get_action(states[0].tolist(), 0.0)
    out.update(states=states, next_states=next_states, actions=actions, rewards=rewards, dones=dones,
               losses=np.array(losses), log_alpha=np.array(alphas), noise=np.array(noise), seed=SEED,
               act_seed77=np.array(act), epoch=int(tr.epoch))
    for name, net in {**nets, "target_critic_1": tr.target_critic_1, "target_critic_2": tr.target_critic_2}.items():
        for k, v in sd_to_np(net.state_dict()).items():
            out[f"{name}1_{k}"] = v
    np.savez_compressed(os.path.join(OUT, "learner_SAC_Trainer.npz"), **out)
    print("SAC_Trainer losses", losses, "log_alpha", alphas, "act", act)


# The same update on PACKED-REPRESENTABLE observations: rows the reference's own state_PathPlan produced (tests/golden/
# episodes.npz, generated by gen_golden.py from executed episodes) -- 80 of their 100 columns are 0/1 flags, 5 are zeros,
# 15 are scalars -- so that csrc/sac.hip, which reads 80-byte packed rows, can be run on EXACTLY the inputs the executed
# SAC_Trainer.update saw (tests/test_sac_fused_gpu.py::test_fused_sac_against_the_executed_reference).
def gen_sac_packed(s):
    from FactoryClass.TrainerFactory import TrainerFactory
    B = 128                                                       # two 64-sample tiles of the fused kernels
    param = {"Trainer_Type": "SAC_Trainer", "Is_Train": "1", "IsPriority_Replay": "0",
             "actor": {"NetWork": "PolicyNetContinuous_SAC", "h": "1", "w": "100", "channel": "1", "action_bound": "1",
                       "hiden_dim": "64", "output": "2", "lr": "0.0001"},
             "critic": {"NetWork": "QValueNetContinuous_SAC", "h": "1", "w": "100", "channel": "1", "hiden_dim": "64",
                        "action_dim": "2", "lr": "0.001"},
             "SAC_param": {"IS_Continuous": "1", "alpha_lr": "0.0001", "target_entropy": "1", "gamma": "0.99", "tau": "0.05"},
             "Priority_Replay": "0", "replay_size": "10000", "LEARNING_RATE": "0.0005", "Batch_Size": str(B),
             "max_epoch": "100", "save_loop": str(10 ** 9), "name": "golden"}
    tr = TrainerFactory().Create_Trainer(param)
    assert tr is not None
    tr.save = lambda *a, **k: None
    g = torch.Generator().manual_seed(8642)
    nets = {"actor": tr.actor, "critic_1": tr.critic_1, "critic_2": tr.critic_2}
    for net in nets.values():
        with torch.no_grad():
            for p in net.parameters():
                p.copy_(torch.randn(p.shape, generator=g) * 0.1)
    with torch.no_grad():                                         # targets away from the critics (as after some training)
        for tgt, src in ((tr.target_critic_1, tr.critic_1), (tr.target_critic_2, tr.critic_2)):
            for pt, p in zip(tgt.parameters(), src.parameters()):
                pt.copy_(p + 0.01 * torch.randn(p.shape, generator=g))
    out = {}
    for name, net in {**nets, "target_critic_1": tr.target_critic_1, "target_critic_2": tr.target_critic_2}.items():
        for k, v in sd_to_np(net.state_dict()).items():
            out[f"{name}0_{k}"] = v
    ep = np.load(os.path.join(OUT, "episodes.npz"))
    obs, outs, off = ep["obs"], ep["outs"], ep["offsets"]
    rng = np.random.default_rng(11)
    ok = np.ones(len(obs), bool)
    ok[off[1:] - 1] = False                                        # s' must be the next record of the same episode
    pick = rng.choice(np.nonzero(ok)[0], B, replace=False)
    states = obs[pick].astype(np.float32)
    next_states = obs[pick + 1].astype(np.float32)
    actions = rng.uniform(-1, 1, (B, 2)).astype(np.float32)
    rewards = outs[pick, 0].astype(np.float32)
    dones = (rng.random(B) < 0.2).astype(np.float32)
    for i in range(B):
        tr.replay_memory.push((0, 0, 0, 0, 0), 0)     # update() only checks len(memory) >= Batch_Size (:333)
    td = {"states": states.tolist(), "actions": actions.tolist(), "rewards": rewards.tolist(),
          "next_states": next_states.tolist(), "dones": dones.tolist()}
    SEED, K = 2000, 5
    losses, alphas, noise = [], [], []
    for k in range(K):
        torch.manual_seed(SEED + k)
        n1, n2 = torch.randn(B, 2), torch.randn(B, 2)
        noise.append(np.stack([n1.numpy(), n2.numpy()]))
        torch.manual_seed(SEED + k)
        res = tr.update(td)
        losses.append(float(res["loss"]))
        alphas.append(float(tr.log_alpha))
    out.update(states=states, next_states=next_states, actions=actions, rewards=rewards, dones=dones,
               losses=np.array(losses), log_alpha=np.array(alphas), noise=np.array(noise), seed=SEED, epoch=int(tr.epoch),
               episode_rows=pick)
    for name, net in {**nets, "target_critic_1": tr.target_critic_1, "target_critic_2": tr.target_critic_2}.items():
        for k, v in sd_to_np(net.state_dict()).items():
            out[f"{name}1_{k}"] = v
    np.savez_compressed(os.path.join(OUT, "learner_SAC_Trainer_packed.npz"), **out)
    print("SAC_Trainer (packed-representable rows) losses", losses, "log_alpha", alphas)


if __name__ == "__main__":
    if "--sac-packed" in sys.argv:
        _s = RefSession()
        try:
            gen_sac_packed(_s)
        finally:
            _s.close()
    elif "--dqn-packed" in sys.argv:
        _s = RefSession()
        try:
            for _name, _net in PACKED_CASES:
                gen_packed(_s, _name, _net)
        finally:
            _s.close()
    elif "--sac" in sys.argv:
        _s = RefSession()
        try:
            gen_sac(_s)
        finally:
            _s.close()
    else:
        main()
