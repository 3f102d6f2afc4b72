"""Generate tests/golden/ckpt/* by EXECUTING the reference trainers' save() / Load_Mod() on a scratch copy.

TEST INFRASTRUCTURE (build container only).  For each of DQN_Trainer, DDQN_Trainer,
DuelingDQN_Trainer and SAC_Trainer:

  1. the reference trainer is built by the reference's own TrainerFactory, its weights are injected (seeded), and a few
     updates are executed so that the Adam moments and step counts are non-trivial and q_target != q_local;
  2. the reference's own `save()` writes the checkpoint files under <scratch>/Mod/ (Trainer/DuelingDQN_Trainer.py:74-84,
     Trainer/SAC_Trainer.py:109-119).  For DQN_Trainer and DDQN_Trainer `save()` RAISES as shipped -- the file name is
     built as `'%s/q_local_%s.pth' % (directory) % (self.name)` (Trainer/DQN_Trainer.py:79,82; DDQN_Trainer.py:66,69): one
     argument for two `%s` -> TypeError -- so for those two the files are written by executing save()'s two `state = {...}`
     lines as they stand (`q_local_*.pth` gets q_TARGET's weights and vice versa: the swap is the reference's) under the names
     its own Load_Mod reads (DQN_Trainer.py:53-58; DDQN_Trainer.py:41-48).  `save_raises` in expected.npz records which;
  3. a FRESH reference trainer of the same kind is constructed: its constructor calls Load_Mod(), which reads those files.
     What it ends up with -- q_local / q_target weights, epoch, the optimizer's moments -- is stored in
     tests/golden/ckpt/expected_<Trainer>.npz: the plugins' Load_Mod must end up with exactly the same from the same files.

The .pth files are torch.save() outputs of the executed reference: data, not source.
"""
from __future__ import annotations

import os
import random
import shutil
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.abspath(os.path.join(HERE, "..", "tests", "golden", "ckpt"))
sys.path.insert(0, HERE)
from ref_harness import RefSession  # noqa: E402
from gen_golden_learner import make_param, sd_to_np  # noqa: E402

NAME = "UAV_0"
FILES = {"DQN_Trainer": ("q_target_%s.pth", "q_local_%s.pth"),
         "DDQN_Trainer": ("q_target_DDQN_%s.pth", "q_local_DDQN_%s.pth"),
         "DuelingDQN_Trainer": ("q_target_DuelingDQN_%s.pth", "q_local_DuelingDQN_%s.pth"),
         "SAC_Trainer": ("actor_SAC_%s.pth", "critic_1_SAC_%s.pth", "critic_2_SAC_%s.pth")}
SAC_PARAM = {"Trainer_Type": "SAC_Trainer", "Is_Train": "1", "IsPriority_Replay": "0",
             "actor": {"NetWork": "PolicyNetContinuous_SAC", "h": "1", "w": "100", "channel": "1", "action_bound": "1",
                       "hiden_dim": "64", "output": "2", "lr": "0.0001"},
             "critic": {"NetWork": "QValueNetContinuous_SAC", "h": "1", "w": "100", "channel": "1", "hiden_dim": "64",
                        "action_dim": "2", "lr": "0.001"},
             "SAC_param": {"IS_Continuous": "1", "alpha_lr": "0.0001", "target_entropy": "1", "gamma": "0.99", "tau": "0.05"},
             "Priority_Replay": "0", "replay_size": "10000", "LEARNING_RATE": "0.0005", "Batch_Size": "64",
             "max_epoch": "100", "save_loop": str(10 ** 9), "name": NAME}


def opt_to_np(prefix, opt):
    """torch.optim.Adam.state_dict() -> flat arrays: per parameter index step / exp_avg / exp_avg_sq"""
    out = {}
    for i, st in opt.state_dict()["state"].items():
        out[f"{prefix}step_{i}"] = np.array(float(st["step"]))
        out[f"{prefix}exp_avg_{i}"] = st["exp_avg"].detach().cpu().numpy().copy()
        out[f"{prefix}exp_avg_sq_{i}"] = st["exp_avg_sq"].detach().cpu().numpy().copy()
    return out


def batch(B=64, W=100, A=3, seed=5):
    rng = np.random.default_rng(seed)
    return (rng.normal(0, 1, (B, W)).astype(np.float32), rng.normal(0, 1, (B, W)).astype(np.float32),
            rng.integers(0, A, B).astype(np.int64), rng.normal(0, 1, B).astype(np.float32), (rng.random(B) < 0.25).astype(np.float32))


def gen_dqn_family(s, trainer_name, net, mod_dir):
    from FactoryClass.TrainerFactory import TrainerFactory
    param = make_param(net, trainer_name)
    param["name"] = NAME
    tr = TrainerFactory().Create_Trainer(dict(param))
    assert tr is not None and tr.epoch == 0, trainer_name
    g = torch.Generator().manual_seed(97)
    for netobj, scale in ((tr.q_local, 0.15), (tr.q_target, 0.12)):
        with torch.no_grad():
            for p in netobj.parameters():
                p.copy_(torch.randn(p.shape, generator=g) * scale)
    states, next_states, actions, rewards, dones = batch()
    if trainer_name == "DuelingDQN_Trainer":
        td = {"states": states.tolist(), "actions": tuple(int(a) for a in actions), "rewards": tuple(float(r) for r in rewards),
              "next_states": next_states.tolist(), "dones": tuple(float(d) for d in dones)}
        for _ in range(4):
            tr.update(td)
    else:
        F = torch.FloatTensor
        for i in range(len(states)):
            tr.replay_memory.push((F(states[i:i + 1]), torch.tensor([[int(actions[i])]]), F([[float(rewards[i])]]),
                                   F(next_states[i:i + 1]), F([[float(dones[i])]])), 0)
        random.seed(7)
        for _ in range(4):
            tr.learn_off_policy()
    assert tr.epoch == 4
    raises = False
    try:
        tr.save()                                              # the reference's own save(), default directory <scratch>/Mod/
    except TypeError as e:                                     # DQN_Trainer.py:79 / DDQN_Trainer.py:66: the format expression
        raises = True
        print(trainer_name, "save() raises as shipped:", e)
        t_name, l_name = FILES[trainer_name]
        # save()'s two `state = {...}` lines as they stand: the file called q_local_* gets q_TARGET's weights, and vice versa
        state = {'model': tr.q_target.state_dict(), 'optimizer': tr.optim.state_dict(), 'epoch': tr.epoch}
        torch.save(state, os.path.join(mod_dir, l_name % NAME))
        state = {'model': tr.q_local.state_dict(), 'optimizer': tr.optim.state_dict(), 'epoch': tr.epoch}
        torch.save(state, os.path.join(mod_dir, t_name % NAME))
    saved = dict(saved_local=sd_to_np(tr.q_local.state_dict()), saved_target=sd_to_np(tr.q_target.state_dict()))
    # a fresh reference trainer: its constructor's Load_Mod() reads what was just written
    tr2 = TrainerFactory().Create_Trainer(dict(param))
    assert tr2 is not None and tr2.epoch == 4, (trainer_name, tr2 and tr2.epoch)
    out = {"save_raises": np.array(int(raises)), "epoch": np.array(int(tr2.epoch))}
    for pref, sd in (("local_", sd_to_np(tr2.q_local.state_dict())), ("target_", sd_to_np(tr2.q_target.state_dict())),
                     ("saved_local_", saved["saved_local"]), ("saved_target_", saved["saved_target"])):
        for k, v in sd.items():
            out[pref + k] = v
    out.update(opt_to_np("optim_", tr2.optim))
    # ... and one more executed update from the loaded state: moments and step count were really restored
    if trainer_name == "DuelingDQN_Trainer":
        tr2.save = lambda *a, **k: None
        tr2.update(td)
    else:
        tr2.save = lambda *a, **k: None
        for i in range(len(states)):
            tr2.replay_memory.push((F(states[i:i + 1]), torch.tensor([[int(actions[i])]]), F([[float(rewards[i])]]),
                                    F(next_states[i:i + 1]), F([[float(dones[i])]])), 0)
        random.seed(11)
        tr2.learn_off_policy()
    for k, v in sd_to_np(tr2.q_local.state_dict()).items():
        out["after1_local_" + k] = v
    out["after1_loss"] = np.array(float(tr2.loss))
    out.update(states=states, next_states=next_states, actions=actions, rewards=rewards, dones=dones)
    for fn in FILES[trainer_name]:
        shutil.copy(os.path.join(mod_dir, fn % NAME), os.path.join(OUT, fn % NAME))
    np.savez_compressed(os.path.join(OUT, f"expected_{trainer_name}.npz"), **out)
    print(trainer_name, "save_raises", raises, "epoch", int(tr2.epoch), "loss after one more update", float(tr2.loss))


def gen_sac(s, mod_dir):
    from FactoryClass.TrainerFactory import TrainerFactory
    tr = TrainerFactory().Create_Trainer(dict(SAC_PARAM))
    assert tr is not None
    g = torch.Generator().manual_seed(4321)
    for net in (tr.actor, tr.critic_1, tr.critic_2):
        with torch.no_grad():
            for p in net.parameters():
                p.copy_(torch.randn(p.shape, generator=g) * 0.1)
    tr.target_critic_1.load_state_dict(tr.critic_1.state_dict())
    tr.target_critic_2.load_state_dict(tr.critic_2.state_dict())
    rng = np.random.default_rng(7)
    B = 64
    td = {"states": rng.normal(0, 1, (B, 100)).astype(np.float32).tolist(), "actions": rng.uniform(-1, 1, (B, 2)).astype(np.float32).tolist(),
          "rewards": rng.normal(0, 1, B).astype(np.float32).tolist(), "next_states": rng.normal(0, 1, (B, 100)).astype(np.float32).tolist(),
          "dones": (rng.random(B) < 0.25).astype(np.float32).tolist()}
    for i in range(B):
        tr.replay_memory.push((0, 0, 0, 0, 0), 0)
    for k in range(3):
        torch.manual_seed(100 + k)
        tr.update(td)
    ep = int(tr.epoch)
    tr.save()                                                  # Trainer/SAC_Trainer.py:109-119, as shipped
    tr2 = TrainerFactory().Create_Trainer(dict(SAC_PARAM))     # constructor -> Load_Mod() (:70-106)
    assert tr2 is not None and tr2.epoch == ep
    out = {"save_raises": np.array(0), "epoch": np.array(ep)}
    for name, net in (("actor", tr2.actor), ("critic_1", tr2.critic_1), ("critic_2", tr2.critic_2),
                      ("target_critic_1", tr2.target_critic_1), ("target_critic_2", tr2.target_critic_2)):
        for k, v in sd_to_np(net.state_dict()).items():
            out[f"{name}_{k}"] = v
    for name, opt in (("actor_optim_", tr2.actor_optimizer), ("critic_1_optim_", tr2.critic_1_optimizer), ("critic_2_optim_", tr2.critic_2_optimizer)):
        out.update(opt_to_np(name, opt))
    out["log_alpha_after_load"] = np.array(float(tr2.log_alpha))       # NOT in the files: a fresh trainer starts from its initial value
    for fn in FILES["SAC_Trainer"]:
        shutil.copy(os.path.join(mod_dir, fn % NAME), os.path.join(OUT, fn % NAME))
    np.savez_compressed(os.path.join(OUT, "expected_SAC_Trainer.npz"), **out)
    print("SAC_Trainer epoch", ep, "log_alpha after load", float(tr2.log_alpha))


def main():
    os.makedirs(OUT, exist_ok=True)
    s = RefSession()
    try:
        mod_dir = os.path.join(s.root, "Mod")
        gen_dqn_family(s, "DQN_Trainer", "Qnet2", mod_dir)
        gen_dqn_family(s, "DDQN_Trainer", "Qnet2", mod_dir)
        gen_dqn_family(s, "DuelingDQN_Trainer", "VAnet2", mod_dir)
        gen_sac(s, mod_dir)
    finally:
        s.close()


if __name__ == "__main__":
    main()
