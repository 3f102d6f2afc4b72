"""Have the EXECUTED reference load a checkpoint directory and dump what it ended up with.

TEST INFRASTRUCTURE (build container only: needs /root/reference).  Used by tests/test_checkpoint_interchange.py (test B, row f2):
the files a PLUGIN trainer wrote are copied into a scratch copy's Mod/, the reference's own trainer is constructed by its own
TrainerFactory -- its constructor calls Load_Mod() (Trainer/DQN_Trainer.py:44-69, DDQN_Trainer.py:31-57,
DuelingDQN_Trainer.py:41-71, SAC_Trainer.py:70-106) -- and the loaded weights, epoch and optimizer moments go to an .npz.
Runs as its own process: the reference's factories import modules called DQN_Trainer, SAC_Trainer, ... -- the plugins' names.

    python oracle/ref_load_checkpoint.py <Trainer name> <dir with the .pth files> <out.npz>
"""
from __future__ import annotations

import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


def main(trainer_name: str, ckpt_dir: str, out_path: str) -> None:
    from ref_harness import RefSession
    from gen_golden_checkpoint import FILES, NAME, SAC_PARAM, opt_to_np
    from gen_golden_learner import make_param, sd_to_np
    ckpt_dir, out_path = os.path.abspath(ckpt_dir), os.path.abspath(out_path)
    s = RefSession()
    try:
        for fn in FILES[trainer_name]:
            shutil.copy(os.path.join(ckpt_dir, fn % NAME), os.path.join(s.root, "Mod", fn % NAME))
        from FactoryClass.TrainerFactory import TrainerFactory
        if trainer_name == "SAC_Trainer":
            tr = TrainerFactory().Create_Trainer(dict(SAC_PARAM))
            assert tr is not None
            out = {"epoch": np.array(int(tr.epoch))}
            for name, net in (("actor", tr.actor), ("critic_1", tr.critic_1), ("critic_2", tr.critic_2),
                              ("target_critic_1", tr.target_critic_1), ("target_critic_2", tr.target_critic_2)):
                for k, v in sd_to_np(net.state_dict()).items():
                    out[f"{name}_{k}"] = v
            for name, opt in (("actor_optim_", tr.actor_optimizer), ("critic_1_optim_", tr.critic_1_optimizer),
                              ("critic_2_optim_", tr.critic_2_optimizer)):
                out.update(opt_to_np(name, opt))
        else:
            net = "VAnet2" if trainer_name == "DuelingDQN_Trainer" else "Qnet2"
            param = make_param(net, trainer_name)
            param["name"] = NAME
            tr = TrainerFactory().Create_Trainer(dict(param))
            assert tr is not None
            out = {"epoch": np.array(int(tr.epoch))}
            for pref, sd in (("local_", sd_to_np(tr.q_local.state_dict())), ("target_", sd_to_np(tr.q_target.state_dict()))):
                for k, v in sd.items():
                    out[pref + k] = v
            out.update(opt_to_np("optim_", tr.optim))
        np.savez(out_path, **out)
    finally:
        s.close()


if __name__ == "__main__":
    main(*sys.argv[1:4])
