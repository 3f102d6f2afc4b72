"""Generate tests/golden/federated_ac.npz by EXECUTING the reference's Federated_Learning_AC (Envs/PathPlan_City.py:590-601)
on a scratch copy: four SAC_Trainers (Trainer/SAC_Trainer.py) with injected actor weights hung on the env's Agents list, the
method called as run_eposide calls it (:469-472), every actor recorded before and after.

TEST INFRASTRUCTURE (build container only).  What the execution shows (and the golden pins): the loop at :593-597 adds the
other agents' tensors IN PLACE into the deep copy's parameters (state_dict() hands out the parameters' storage), but the
division at :597 assigns into the throw-away dict state_dict() returned -- it never reaches the model.  The "global model" every
UAV receives through replace_param (:456-459) is therefore the SUM of the actors, not their mean.  The golden records
`divides` = whether the executed result equals the mean (False on torch 2.10; the statement is version-independent: a dict
item assignment cannot write through)."""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "tests", "golden")
sys.path.insert(0, HERE)
from ref_harness import RefSession  # noqa: E402

N_AGENTS = 4


def main():
    s = RefSession()
    try:
        from FactoryClass.TrainerFactory import TrainerFactory
        param = {"Trainer_Type": "SAC_Trainer", "Is_Train": "1", "IsPriority_Replay": "0",
                 "actor": {"NetWork": "PolicyNetContinuous_SAC", "h": "1", "w": "100", "channel": "1", "action_bound": "1",
                           "hiden_dim": "64", "output": "2", "lr": "0.0001"},
                 "critic": {"NetWork": "QValueNetContinuous_SAC", "h": "1", "w": "100", "channel": "1", "hiden_dim": "64",
                            "action_dim": "2", "lr": "0.001"},
                 "SAC_param": {"IS_Continuous": "1", "alpha_lr": "0.0001", "target_entropy": "1", "gamma": "0.99", "tau": "0.05"},
                 "Priority_Replay": "0", "replay_size": "10000", "LEARNING_RATE": "0.0005", "Batch_Size": "64",
                 "max_epoch": "100", "save_loop": str(10 ** 9), "name": "golden"}
        g = torch.Generator().manual_seed(777)
        agents, out = [], {}
        for j in range(N_AGENTS):
            tr = TrainerFactory().Create_Trainer(dict(param, name=f"golden{j}"))
            assert tr is not None
            with torch.no_grad():
                for p in tr.actor.parameters():
                    p.copy_(torch.randn(p.shape, generator=g) * 0.1)
            for k, v in tr.actor.state_dict().items():
                out[f"a{j}_before_{k}"] = v.detach().cpu().numpy().copy()
            for k, v in tr.critic_1.state_dict().items():
                out[f"c{j}_before_{k}"] = v.detach().cpu().numpy().copy()
            agents.append(types.SimpleNamespace(Trainer=tr))
        env = s.env
        env.Agents = agents
        type(env).Federated_Learning_AC(env)                       # Envs/PathPlan_City.py:590-601, as :469-472 calls it
        for j, a in enumerate(agents):
            for k, v in a.Trainer.actor.state_dict().items():
                out[f"a{j}_after_{k}"] = v.detach().cpu().numpy().copy()
            for k, v in a.Trainer.critic_1.state_dict().items():   # the critics are not touched
                assert np.array_equal(out[f"c{j}_before_{k}"], v.detach().cpu().numpy())
        keys = list(agents[0].Trainer.actor.state_dict().keys())
        total = {k: sum(out[f"a{j}_before_{k}"].astype(np.float64) for j in range(N_AGENTS)) for k in keys}
        is_sum = all(np.allclose(out[f"a0_after_{k}"], total[k], rtol=0, atol=1e-6) for k in keys)
        is_mean = all(np.allclose(out[f"a0_after_{k}"], total[k] / N_AGENTS, rtol=0, atol=1e-6) for k in keys)
        same = all(np.array_equal(out[f"a0_after_{k}"], out[f"a{j}_after_{k}"]) for k in keys for j in range(N_AGENTS))
        print("after == sum:", is_sum, " after == mean:", is_mean, " all agents equal:", same)
        out["n_agents"] = np.int64(N_AGENTS)
        out["divides"] = np.bool_(is_mean)
        out["keys"] = np.array(keys)
        np.savez_compressed(os.path.join(OUT, "federated_ac.npz"), **out)
    finally:
        s.close()


if __name__ == "__main__":
    main()
