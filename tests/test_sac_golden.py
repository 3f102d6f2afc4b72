"""SAC (continuous) parity: same weights + same batch + the same two N(0,1) draws -> same actor loss, log_alpha and
post-update weights (actor, twin critics, soft-updated targets) as the EXECUTED reference SAC_Trainer.update
(Trainer/SAC_Trainer.py:325-379).  Golden: oracle/gen_golden_learner.py::gen_sac."""
import numpy as np
import torch

from conftest import load_golden

PARAM = {"Trainer_Type": "SAC_Trainer", "Is_Train": "1", "IsPriority_Replay": "0",
         "actor": {"NetWork": "PolicyNetContinuous_SAC", "w": "100", "action_bound": "1", "hiden_dim": "64", "output": "2",
                   "lr": "0.0001"},
         "critic": {"NetWork": "QValueNetContinuous_SAC", "w": "100", "hiden_dim": "64", "action_dim": "2", "lr": "0.001"},
         "SAC_param": {"IS_Continuous": "1", "alpha_lr": "0.0001", "target_entropy": "1", "gamma": "0.99", "tau": "0.05"},
         "replay_size": "10000", "Batch_Size": "64", "save_loop": "1000000", "name": "UAV_0"}


def _load(L, g, suffix):
    for name in ("actor", "critic_1", "critic_2"):
        getattr(L, name).load_state_dict({k[len(name) + 2:]: torch.tensor(v) for k, v in g.items()
                                          if k.startswith(f"{name}{suffix}_")})
    L.target_critic_1.load_state_dict(L.critic_1.state_dict())
    L.target_critic_2.load_state_dict(L.critic_2.state_dict())


def _check(L, g, tol):
    for name in ("actor", "critic_1", "critic_2", "target_critic_1", "target_critic_2"):
        for k, v in getattr(L, name).state_dict().items():
            assert np.abs(v.cpu().numpy() - g[f"{name}1_{k}"]).max() <= tol, (name, k)


def test_sac_updates_match_reference_with_seeded_rng():
    from dqn_based_uav_3d_path_planer_amd.sac import SACLearner
    g = load_golden("learner_SAC_Trainer.npz")
    L = SACLearner(PARAM, device="cpu")
    _load(L, g, "0")
    batch = {k: torch.tensor(g[k]) for k in ("states", "actions", "rewards", "next_states", "dones")}
    for k in range(len(g["losses"])):
        torch.manual_seed(int(g["seed"]) + k)            # the reference drew its rsample() noise from this stream
        loss = float(L.learn(batch))
        assert abs(loss - g["losses"][k]) <= 2e-5 * max(1.0, abs(g["losses"][k])), k
        assert abs(float(L.log_alpha) - g["log_alpha"][k]) <= 1e-6
    assert L.epoch == int(g["epoch"])
    _check(L, g, 2e-6)


def test_sac_updates_match_reference_with_injected_noise():
    from dqn_based_uav_3d_path_planer_amd.sac import SACLearner
    g = load_golden("learner_SAC_Trainer.npz")
    L = SACLearner(PARAM, device="cpu")
    _load(L, g, "0")
    batch = {k: torch.tensor(g[k]) for k in ("states", "actions", "rewards", "next_states", "dones")}
    for k in range(len(g["losses"])):
        n = torch.tensor(g["noise"][k])
        loss = float(L.learn(batch, noise=(n[0], n[1])))
        assert abs(loss - g["losses"][k]) <= 2e-5 * max(1.0, abs(g["losses"][k])), k
    _check(L, g, 2e-6)


def _load_all(L, g):
    for name in ("actor", "critic_1", "critic_2", "target_critic_1", "target_critic_2"):
        getattr(L, name).load_state_dict({k[len(name) + 2:]: torch.tensor(v) for k, v in g.items() if k.startswith(f"{name}0_")})


def test_sac_updates_match_reference_on_packed_representable_rows():
    """The second SAC golden (oracle/gen_golden_learner.py::gen_sac_packed): observations the reference's own
    state_PathPlan produced, targets away from the critics, batch 128 -- the vectors the fused HIP update is checked
    against on the GPU (tests/test_sac_fused_gpu.py); here they pin the PyTorch learner too."""
    from dqn_based_uav_3d_path_planer_amd.sac import SACLearner
    g = load_golden("learner_SAC_Trainer_packed.npz")
    L = SACLearner(PARAM, device="cpu")
    _load_all(L, g)
    batch = {k: torch.tensor(g[k]) for k in ("states", "actions", "rewards", "next_states", "dones")}
    for k in range(len(g["losses"])):
        n = torch.tensor(g["noise"][k])
        loss = float(L.learn(batch, noise=(n[0], n[1])))
        assert abs(loss - g["losses"][k]) <= 2e-5 * max(1.0, abs(g["losses"][k])), k
        assert abs(float(L.log_alpha) - g["log_alpha"][k]) <= 1e-6
    _check(L, g, 2e-6)


def test_sac_trainer_plugin_surface(tmp_path):
    from dqn_based_uav_3d_path_planer_amd import factories
    g = load_golden("learner_SAC_Trainer.npz")
    p = dict(PARAM, device="cpu", model_dir=str(tmp_path))
    tr = factories.TrainerFactory().Create_Trainer(p)
    assert type(tr).__name__ == "SAC_Trainer" and tr.IS_Continuous == 1
    _load(tr.learner, g, "1")                    # the golden action was drawn from the trained actor
    torch.manual_seed(77)
    a = tr.get_action(g["states"][0].tolist(), 0.0)
    assert np.allclose(a, g["act_seed77"], atol=1e-6) and len(a) == 2
    _load(tr.learner, g, "0")
    td = {k: g[k] for k in ("states", "actions", "rewards", "next_states", "dones")}
    assert tr.update(td)["sum_epoch"] == 1 and tr.loss == 0          # memory < Batch_Size: no learning (:333)
    for i in range(64):
        tr.Push_Replay((g["states"][i:i + 1], g["actions"][i], [[float(g["rewards"][i])]], g["next_states"][i:i + 1],
                        [[float(g["dones"][i])]]))
    assert len(tr.replay_memory.memory) == 64
    assert torch.allclose(tr.replay_memory.actions[:64].cpu(), torch.tensor(g["actions"]))
    tr.learner.epoch = 0
    torch.manual_seed(int(g["seed"]))
    out = tr.update(td)
    assert abs(float(out["loss"]) - g["losses"][0]) <= 2e-5
    tr.save()
    assert {"model", "optimizer", "epoch"} == set(torch.load(tmp_path / "actor_SAC_UAV_0.pth"))
    assert (tmp_path / "critic_1_SAC_UAV_0.pth").exists() and (tmp_path / "critic_2_SAC_UAV_0.pth").exists()


import pytest  # noqa: E402


@pytest.mark.gpu
def test_sac_updates_match_reference_on_the_gpu():
    """The same executed-reference goldens with SACLearner on the MI355X (PyTorch-ROCm kernels, hipBLASLt GEMMs):
    injected rsample() noise pins the two N(0,1) draws, so everything is deterministic up to f32 summation order.
    Bars: 5e-5 relative on the actor loss, 1e-5 on weights after the golden's updates (CPU bars: 2e-5 / 2e-6)."""
    from dqn_based_uav_3d_path_planer_amd.sac import SACLearner
    g = load_golden("learner_SAC_Trainer.npz")
    L = SACLearner(PARAM, device="cuda:0")
    _load(L, g, "0")
    batch = {k: torch.tensor(g[k]).cuda() for k in ("states", "actions", "rewards", "next_states", "dones")}
    for k in range(len(g["losses"])):
        n = torch.tensor(g["noise"][k]).cuda()
        loss = float(L.learn(batch, noise=(n[0], n[1])))
        assert abs(loss - g["losses"][k]) <= 5e-5 * max(1.0, abs(g["losses"][k])), k
        assert abs(float(L.log_alpha) - g["log_alpha"][k]) <= 1e-5
    assert L.epoch == int(g["epoch"])
    _check(L, g, 1e-5)
