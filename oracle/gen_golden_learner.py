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


def main():
    s = RefSession()
    try:
        gen(s, "DQN_Trainer", "Qnet2")
        gen(s, "DDQN_Trainer", "Qnet2")
        gen(s, "DuelingDQN_Trainer", "VAnet2")
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


if __name__ == "__main__":
    main()
