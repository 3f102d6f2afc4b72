"""the federated merge run_eposide fires every FL_Loop episodes when Is_FL is set
(Envs/PathPlan_City.py:469-475 -> Federated_Learning_AC :590-601) against the EXECUTED reference
(tests/golden/federated_ac.npz, oracle/gen_golden_federated.py: four SAC_Trainers with injected actors).  Host side here
(torch on CPU, the oracle-backed fake env); the one-launch device merge is in test_federated_gpu.py."""
import os
import re
import sys

import numpy as np
import pytest
import torch

from conftest import load_golden
from dqn_based_uav_3d_path_planer_amd import driver, factories, federated  # noqa: F401  (factories puts plugins/ on sys.path)

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fake_backend  # noqa: E402


def _inject(actor, g, j, when="before"):
    actor.load_state_dict({str(k): torch.tensor(g[f"a{j}_{when}_{k}"]) for k in g["keys"]})


def test_the_executed_reference_merges_to_the_sum_not_the_mean():
    """What the golden pins: Federated_Learning_AC's division never reaches the model."""
    g = load_golden("federated_ac.npz")
    n = int(g["n_agents"])
    assert n == 4 and not bool(g["divides"])
    for k in g["keys"]:
        total = g[f"a0_before_{k}"].copy()
        for j in range(1, n):
            total += g[f"a{j}_before_{k}"]            # f32 adds in agent order, as :593-596
        for j in range(n):
            assert np.array_equal(g[f"a{j}_after_{k}"], total), (k, j)


@pytest.mark.parametrize("aggregate", ["reference", "mean"])
def test_merge_modules_equals_the_executed_reference(aggregate):
    from dqn_based_uav_3d_path_planer_amd.nets import create_network
    g = load_golden("federated_ac.npz")
    n = int(g["n_agents"])
    p = {"NetWork": "PolicyNetContinuous_SAC", "w": "100", "action_bound": "1", "hiden_dim": "64", "output": "2"}
    actors = [create_network(p) for _ in range(n)]
    for j, a in enumerate(actors):
        _inject(a, g, j)
    federated.merge_modules(actors, 1.0 if aggregate == "reference" else 1.0 / n)
    for j, a in enumerate(actors):
        for k, v in a.state_dict().items():
            want = g[f"a{j}_after_{k}"] if aggregate == "reference" else g[f"a{j}_after_{k}"] * np.float32(0.25)
            assert np.array_equal(v.numpy(), want), (j, k)          # bit for bit (x 0.25 is exact)
    with pytest.raises(ValueError):
        federated._scale("median", 4)


def _sac_sim(tmp_path, monkeypatch, num_envs, **env_tags):
    import _backend
    monkeypatch.setattr(_backend, "make_backend", lambda n, b, **kw: fake_backend.OracleVecEnv(n, b, **kw))
    monkeypatch.chdir(tmp_path)
    xml = driver.make_config_dir(str(tmp_path), "SAC", num_envs=num_envs, num_uav=4)
    s = open(xml).read()
    for k, v in env_tags.items():
        if re.search(rf"<{k}>[^<]*</{k}>", s):
            s = re.sub(rf"<{k}>[^<]*</{k}>", f"<{k}>{v}</{k}>", s, count=1)
        else:
            s = s.replace("<num_UAV>", f"<{k}>{v}</{k}>\n        <num_UAV>", 1)
    open(xml, "w").write(s)
    return driver.simulator(xml)


def test_is_fl_merges_the_actors_every_fl_loop_episodes(tmp_path, monkeypatch):
    """Is_FL = 1, Is_AC = 4, FL_Loop = 2 through the PathPlan_City plugin: the bookkeeping of :469-472 (epoch % FL_Loop after
    the epoch increment) and the merge itself against the executed reference -- the injected actors end as the golden's."""
    g = load_golden("federated_ac.npz")
    sim = _sac_sim(tmp_path, monkeypatch, 2, Is_FL=1, Is_AC=4, FL_Loop=2)
    env = sim.env
    assert env is not None and env.Is_FL == 1 and env.FL_Loop == 2 and env.FL_Aggregate == "reference"
    trs = [u.Trainer for u in env.Agents]
    c_before = [t.critic_1.fc1.weight.detach().clone() for t in trs]
    for j, t in enumerate(trs):
        _inject(t.actor, g, j)
    env.epoch = 1
    env._federated_merge()                              # 1 % 2 != 0: nothing happens
    assert env.fl_merges == 0
    for j, t in enumerate(trs):
        assert np.array_equal(t.actor.fc1.weight.detach().numpy(), g[f"a{j}_before_fc1.weight"])
    env.epoch = 2
    env._federated_merge()
    assert env.fl_merges == 1 and env.fl_merged_on == "torch"
    for j, t in enumerate(trs):
        for k, v in t.actor.state_dict().items():
            assert np.array_equal(v.numpy(), g[f"a{j}_after_{k}"]), (j, k)
        assert torch.equal(t.critic_1.fc1.weight.detach(), c_before[j])       # critics / targets / optimizers untouched
    # replace_param, the reference's hand-over (Trainer/SAC_Trainer.py:456-459)
    from dqn_based_uav_3d_path_planer_amd.nets import create_network
    other = create_network({"NetWork": "PolicyNetContinuous_SAC", "w": "100", "action_bound": "1", "hiden_dim": "64", "output": "2"})
    trs[1].replace_param(other)
    assert all(torch.equal(a, b) for a, b in zip(trs[1].actor.parameters(), other.parameters()))


def test_is_fl_in_a_real_episode_and_off_by_default(tmp_path, monkeypatch):
    """One whole run_eposide with Is_FL = 1, FL_Loop = 1, <FL_Aggregate>mean: afterwards all four actors are equal (the mean
    of what the episode trained), the critics are not; the stock config (Is_FL = 0) never merges."""
    sim = _sac_sim(tmp_path, monkeypatch, 2, Is_FL=1, Is_AC=4, FL_Loop=1, FL_Aggregate="mean")
    env = sim.env
    assert env.FL_Aggregate == "mean"
    torch.manual_seed(2)
    env.run_eposide(0.1)
    assert env.fl_merges == 1 and env.epoch == 1
    trs = [u.Trainer for u in env.Agents]
    for t in trs[1:]:
        for a, b in zip(trs[0].actor.parameters(), t.actor.parameters()):
            assert torch.equal(a, b)
        assert not torch.equal(trs[0].critic_1.fc1.weight, t.critic_1.fc1.weight)
    sim0 = _sac_sim(tmp_path, monkeypatch, 1)
    assert sim0.env.Is_FL == 0
    sim0.env.epoch = 3
    sim0.env._federated_merge()
    assert sim0.env.fl_merges == 0


@pytest.mark.parametrize("tag", ["mean", None])
def test_dqn_family_merges_q_local(tmp_path, monkeypatch, tag):
    """Is_AC = 0: the reference's Federated_Learning cannot run (no trainer has get_policy_DFRL); the same merge on q_local.
    tag None: <FL_Aggregate> absent -- there is no executed behaviour to reproduce here, so the default is the MEAN (a sum of
    U Q-networks would scale every Q-value by ~U), unlike the actor-critic merge, whose default is the executed reference's sum."""
    import _backend
    monkeypatch.setattr(_backend, "make_backend", lambda n, b, **kw: fake_backend.OracleVecEnv(n, b, **kw))
    monkeypatch.chdir(tmp_path)
    xml = driver.make_config_dir(str(tmp_path), "DuelingDQN", num_envs=2, num_uav=3)
    s = open(xml).read().replace("<Is_FL>0</Is_FL>", "<Is_FL>1</Is_FL>").replace("<FL_Loop>3</FL_Loop>", "<FL_Loop>1</FL_Loop>")
    if tag is not None:
        s = s.replace("<num_UAV>", f"<FL_Aggregate>{tag}</FL_Aggregate>\n        <num_UAV>", 1)
    open(xml, "w").write(s)
    env = driver.simulator(xml).env
    assert env.FL_Aggregate == ("mean" if tag else "reference") and env._fl_aggregate_given == (tag is not None)
    trs = [u.Trainer for u in env.Agents]
    torch.manual_seed(0)
    with torch.no_grad():
        for t in trs:
            for p in t.q_local.parameters():
                p.copy_(torch.randn_like(p))
    want = [sum(ps).detach() / 3 for ps in zip(*[list(t.q_local.parameters()) for t in trs])]
    tgt = [p.detach().clone() for p in trs[0].q_target.parameters()]
    env.epoch = 1
    env._federated_merge()
    assert env.fl_merges == 1
    for t in trs:
        for p, w in zip(t.q_local.parameters(), want):
            assert torch.allclose(p, w, rtol=0, atol=1e-6)
    assert all(torch.equal(a, b) for a, b in zip(trs[0].q_target.parameters(), tgt))     # replace_param writes q_local only
