"""The reflection-level drop-in boundary: reference-named plugin modules, XML config
contract, factories, episode-loop ordering and result dict.  Host logic only: the vectorised env is the
oracle-backed fake from tests/fake_backend.py (the GPU run of the same plugins is in test_plugins_gpu.py).
When /root/reference is present the UNMODIFIED reference simulator.py drives our PathPlan_City plugin."""
import os
import random
import sys

import numpy as np
import pytest
import torch

from conftest import load_golden
from dqn_based_uav_3d_path_planer_amd import driver, factories
from dqn_based_uav_3d_path_planer_amd.compat import Loc, XML2Dict, calculate_angle, Eu_Loc_distance

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fake_backend  # noqa: E402


@pytest.fixture()
def cfg_dir(tmp_path, monkeypatch):
    import _backend                                    # plugins/_backend.py (on sys.path via factories)
    monkeypatch.setattr(_backend, "make_backend", lambda n, b, **kw: fake_backend.OracleVecEnv(n, b, **kw))
    monkeypatch.chdir(tmp_path)
    return tmp_path


def test_compat_geometry_matches_reference_goldens():
    g = load_golden("angle_kat.npz")
    for p, a in list(zip(g["pairs"], g["angle"]))[:500]:
        assert calculate_angle(Loc(float(p[0]), float(p[1]), 0), Loc(float(p[2]), float(p[3]), 0)) == a
    assert Eu_Loc_distance(Loc(0, 0, 0), Loc(3, 4, 12)) == 13.0
    assert Loc(1, 2, 3) + Loc(1, 1, 1) == Loc(2, 3, 4)
    with pytest.raises(ValueError):
        Loc(1, 2, 3) + 5


def test_xml2dict_contract(tmp_path):
    p = tmp_path / "t.xml"
    p.write_text("<?xml version='1.0'?>\n<a><b>1</b><c><d> x </d><e/></c><f>1</f><f>2</f></a>")
    assert XML2Dict(str(p)) == {"a": {"b": "1", "c": {"d": "x", "e": None}, "f": ["1", "2"]}}


def test_factories_reflect_and_swallow_errors(capsys):
    b = factories.ThreatenFactory().Create_Threaten(
        {"Threaten_Type": "building", "position": {"x": "1.5", "y": "2", "z": "0"}, "_R": "3", "_H": "4"})
    assert type(b).__name__ == "building" and b._R == 3.0
    assert b.check_threaten(Loc(1.5, 2, 4)) == 1 and b.check_threaten(Loc(1.5, 2, 4.01)) == 0
    assert b.check_threaten(Loc(4.49, 2, 0)) == 1 and b.check_threaten(Loc(4.5, 2, 0)) == 0
    assert factories.EnvFactory().Create_Env({"Env_Type": "No_Such_Env"}) is None      # EnvFactory.py:21-23
    assert "No_Such_Env" in capsys.readouterr().out
    assert factories.TrainerFactory().Create_Trainer({"Trainer_Type": "DQN_Trainer"}) is None   # bad param -> None


@pytest.mark.parametrize("name,net", [("DQN", "Qnet2"), ("DDQN", "Qnet2"), ("DuelingDQN", "VAnet2")])
def test_trainer_plugins_surface_and_reference_numbers(name, net, tmp_path):
    g = load_golden(f"learner_{name}_Trainer.npz")
    p = dict(XML2Dict(os.path.join(os.path.dirname(driver.__file__), "configs", f"Trainer_{name}.xml"))["Trainer"])
    p.update(name="UAV_0", LEARNING_RATE="0.001", device="cpu", model_dir=str(tmp_path), save_loop="4")
    tr = factories.TrainerFactory().Create_Trainer(p)
    assert type(tr).__name__ == f"{name}_Trainer" and tr.Batch_Size == 64 and tr.epoch == 0
    tr.q_local.load_state_dict({k[3:]: torch.tensor(v) for k, v in g.items() if k.startswith("l0_")})
    tr.q_target.load_state_dict({k[3:]: torch.tensor(v) for k, v in g.items() if k.startswith("t0_")})
    td = {"states": g["states"], "actions": tuple(int(a) for a in g["actions"]), "rewards": tuple(map(float, g["rewards"])),
          "next_states": g["next_states"], "dones": tuple(map(float, g["dones"]))}
    out = None
    for _ in range(len(g["losses"])):
        out = tr.update(td)                                     # DuelingDQN_Trainer.update contract
    assert out["sum_epoch"] == len(g["losses"]) == tr.epoch
    assert abs(float(tr.loss) - g["losses"][-1]) <= 1e-5 * g["losses"][-1]
    assert abs(float(out["loss"]) - g["losses"][-2]) <= 1e-5 * g["losses"][-2]   # the reference returns the PREVIOUS loss
    for k, v in tr.q_local.state_dict().items():
        assert np.abs(v.numpy() - g["l1_" + k]).max() <= 2e-6
    # empty transition_dict still counts an epoch (DuelingDQN_Trainer.py:152-156)
    assert tr.update({"states": []})["sum_epoch"] == tr.epoch
    # checkpoints: reference file names + dict keys, and a fresh trainer auto-loads them (Load_Mod)
    tag = {"DQN": "", "DDQN": "DDQN_", "DuelingDQN": "DuelingDQN_"}[name]
    ck = torch.load(tmp_path / f"q_local_{tag}UAV_0.pth")
    assert set(ck) == {"model", "optimizer", "epoch"} and (tmp_path / f"q_target_{tag}UAV_0.pth").exists()
    tr2 = factories.TrainerFactory().Create_Trainer(p)
    assert tr2.epoch == ck["epoch"]
    # replay surface: push (tensor tuple), add (raw), len(buffer), sample2 7-tuple, learn_off_policy
    random.seed(0)
    for i in range(70):
        tr.Push_Replay((torch.tensor(g["states"][i % 64:i % 64 + 1]), torch.tensor([[int(g["actions"][i % 64])]]),
                        torch.tensor([[float(g["rewards"][i % 64])]]), torch.tensor(g["next_states"][i % 64:i % 64 + 1]),
                        torch.tensor([[float(g["dones"][i % 64])]])))
    tr.replay_memory.add(g["states"][0], 1, 0.5, g["next_states"][0], False)
    assert len(tr.replay_memory.buffer) == 71 == len(tr.replay_memory.memory)
    s2 = tr.replay_memory.sample2(8)
    assert len(s2) == 7 and s2[0].shape == (8, 100) and s2[5] is None
    e0 = tr.epoch
    assert tr.learn_off_policy()["sum_epoch"] == e0 + 1
    a = tr.get_action(g["states"][0], 0.0)
    assert a == int(tr.q_local(torch.tensor(g["states"][:1])).argmax())
    assert set(int(tr.get_action(g["states"][0], 1.0)) for _ in range(60)) == {0, 1, 2}


def test_env_plugin_single_env_matches_oracle(cfg_dir):
    from oracle import pyoracle as po
    xml = driver.make_config_dir(str(cfg_dir), "DuelingDQN", num_envs=1)
    sim = driver.simulator(xml)
    env = sim.env
    assert type(env).__name__ == "PathPlan_City" and (env.len, env.width, env.h) == (500, 500, 100)
    assert len(env.buildings) == 26 and len(env.Agents) == 1
    uav = env.Agents[0]
    assert type(uav).__name__ == "UAV" and type(uav.Trainer).__name__ == "DuelingDQN_Trainer" and uav.name == "UAV_0"
    assert uav.Max_Step == 150 and abs(uav.Steering_angle - np.pi / 6) < 1e-15
    # Threaten_rate: SURVEY App. F.1
    assert env.Threaten_rate(Loc(285.3311642197549, 454.6501406344541, 0)) == 1
    assert env.Threaten_rate(Loc(322.23, 454.65, 0)) == 0 and env.Threaten_rate(Loc(500.0001, 10, 0)) == 1
    # Move_Agent == the oracle's update + state from the same state
    w = load_golden("world_stock.npz")
    o = po.OracleUav(po.OracleWorld(w["buildings"]), po.default_uav_params(w))
    st, sub, alias = env.backend.get_state(0, 1, want_sub=True)
    o.set_state(*st[0][:5], *st[0][6:9], int(st[0][9]), sub[0][: int(st[0][11])])
    o.u.sub0_alias = int(alias[0])
    for a in (2, 0, 1, 1, 2):
        nxt, r, d, info = env.Move_Agent(0, a)
        ro, do, io = o.update(-1.0 + a)
        assert (r, d, info) == (ro, do, po.INFO_NAMES[io])
        assert (np.abs(nxt - o.state()) / np.maximum(1.0, np.abs(o.state()))).max() <= 1e-6      # obs is stored f32
        assert uav.Step == o.u.step and abs(uav.position.x - o.u.px) < 1e-12 and uav.done == bool(o.u.done)


def test_run_eposide_contract_vectorised(cfg_dir):
    xml = driver.make_config_dir(str(cfg_dir), "DQN", num_envs=24, num_uav=2)
    sim = driver.simulator(xml)
    env = sim.env
    assert env.backend.N == 48 and len(env.Agents) == 2
    random.seed(1)
    torch.manual_seed(1)
    res = env.run_eposide(0.9)
    for k in ("success", "lose", "meet_threaten", "normal", "loss", "sum_epoch", "eps", "score", "average_score", "step"):
        assert k in res                                                            # PathPlan_City.py:361
    steps = res["normal"] + res["success"] + res["lose"]
    assert res["lose"] + res["success"] >= 48 and steps > 48 * 100               # every agent ran to a terminal
    assert env.Check_uav_Done() and res["eps"] == 0.9
    for uav in env.Agents:
        assert len(uav.Trainer.replay_memory) == min(steps // 2, 10000) or len(uav.Trainer.replay_memory) > 1000
        assert uav.Trainer.epoch > 150 and uav.Train_time > 0                      # one update per env time step
    # StartAndTrain: epsilon schedule + loop bookkeeping (simulator.py:105-145)
    sim.num_episodes = 10
    sim.StartAndTrain()
    assert sim.epoch == 10 and len(sim.infos) == 10 and sim.Max_score >= max(i["average_score"] for i in sim.infos) - 1e-9


def test_on_policy_episode_collects_then_trains_once(cfg_dir):
    """<Is_On_Policy>1</Is_On_Policy> = the reference's run_thread_OnPolicy branch (Envs/PathPlan_City.py:386-436): every UAV appends
    each of its transitions to its own transition_dict, NOTHING goes to the replay memory, and there is exactly ONE update() -- one
    Train_nn per UAV on the whole episode -- after every agent is done; the off-policy loop (:364-385) learns at every step instead."""
    import re
    xml = driver.make_config_dir(str(cfg_dir), "DuelingDQN", num_envs=4, num_uav=2)
    s = open(xml).read()
    s, n = re.subn(r"<Is_On_Policy>\s*0\s*</Is_On_Policy>", "<Is_On_Policy>1</Is_On_Policy>", s)
    assert n == 1
    open(xml, "w").write(s)
    sim = driver.simulator(xml)
    env = sim.env
    assert env is not None and env.Is_On_Policy == 1 and not env.fast and not env.fast_sac
    random.seed(1)
    torch.manual_seed(1)
    seen = []
    for uav in env.Agents:                                   # count the update() calls and what they are given
        tr = uav.Trainer
        orig = tr.update
        tr.update = (lambda td, _o=orig, _j=uav.j: (seen.append((_j, len(td["states"]))), _o(td))[1])
    e0 = [u.Trainer.epoch for u in env.Agents]
    res = env.run_eposide(0.3)
    moved = res["normal"] + res["success"] + res["lose"]
    assert env.Check_uav_Done() and moved >= 4 * 2 * 2
    assert sorted(j for j, _ in seen) == [0, 1]                                   # ONE update per UAV, after the episode
    assert sum(n for _, n in seen) == moved                                       # ... on every transition its agents made
    for u, e in zip(env.Agents, e0):
        assert u.Trainer.epoch == e + 1 and len(u.Trainer.replay_memory) == 0     # the replay memory is not part of this branch
        td = u.transition_dict
        assert len(td["states"]) == len(td["actions"]) == len(td["next_states"]) == len(td["rewards"]) == len(td["dones"])
        assert tuple(td["states"].shape[1:]) == (100,) and td["dones"].max() <= 1
    assert set(res) >= {"success", "lose", "normal", "loss", "sum_epoch", "average_score"} and np.isfinite(float(res["loss"]))
    # a second episode starts from empty transition lists (:421-422)
    seen.clear()
    res2 = env.run_eposide(0.3)
    assert sum(n for _, n in seen) == res2["normal"] + res2["success"] + res2["lose"]
    # the same file with the flag at 0 runs the off-policy loop: many updates per episode
    open(xml, "w").write(s.replace("<Is_On_Policy>1</Is_On_Policy>", "<Is_On_Policy>0</Is_On_Policy>"))
    env0 = driver.simulator(xml).env
    assert env0 is not None and env0.Is_On_Policy == 0


def test_run_eposide_with_sac_continuous_actions(cfg_dir):
    """BASELINE config 4's trainer (and the reference's shipped default) through the same episode loop."""
    xml = driver.make_config_dir(str(cfg_dir), "SAC", num_envs=6, num_uav=4)
    # Trainer.xml has no <output> at top level for SAC: n_actions falls back to 3 (unused for steer actions)
    sim = driver.simulator(xml)
    env = sim.env
    assert env is not None and type(env.Agents[0].Trainer).__name__ == "SAC_Trainer" and env.backend.N == 24
    torch.manual_seed(3)
    res = env.run_eposide(0.1)
    assert res["lose"] + res["success"] >= 24 and env.Check_uav_Done()
    tr = env.Agents[2].Trainer
    assert tr.replay_memory.actions.shape[1] == 2 and len(tr.replay_memory) > 150
    assert float(tr.replay_memory.actions[: len(tr.replay_memory)].abs().max()) <= 1.0
    assert tr.epoch > 150 and np.isfinite(float(res["loss"]))


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="needs the reference checkout")
def test_unmodified_reference_simulator_drives_our_plugins(cfg_dir, monkeypatch):
    """Drop-in proof: the reference's own simulator.py + factories (scratch copy), our plugin dir first on
    sys.path, our PathPlan_City / UAV / building / DuelingDQN_Trainer behind its reflection."""
    import shutil
    from oracle import ref_harness
    ref = cfg_dir / "ref"
    shutil.copytree("/root/reference", ref, ignore=shutil.ignore_patterns("*.gif", "*.jpg", "doc", "Mod", "Envs", "Agents",
                                                                          "Obstacles", "Trainer"))
    os.makedirs(ref / "logs", exist_ok=True)
    driver.make_config_dir(str(ref), "DuelingDQN", num_envs=8, num_episodes=10)   # overwrites ref/config/*.xml
    monkeypatch.chdir(ref)
    ref_harness.install_shims()
    for m in [m for m in sys.modules if m in ("simulator", "PathPlan_City", "UAV", "building", "CalMod") or
              m.startswith(("BaseClass", "FactoryClass"))]:
        monkeypatch.delitem(sys.modules, m, raising=False)
    monkeypatch.syspath_prepend(str(ref))
    monkeypatch.syspath_prepend(factories.PLUGIN_DIR)
    import importlib
    simulator = importlib.import_module("simulator")
    assert simulator.__file__.startswith(str(ref))
    sim = simulator.simulator()
    assert sim.env is not None and type(sim.env).__module__ == "PathPlan_City"
    assert sys.modules["PathPlan_City"].__file__.startswith(factories.PLUGIN_DIR)
    sim.StartAndTrain()                                       # 10 x 1 episodes through the reference's own loop
    assert sim.epoch == 10 and sim.Max_score > -9999999999
    assert os.path.exists(sim.result_path)                    # the reference wrote its score CSV


def test_path_record_and_csv_contracts(cfg_dir):
    """UAV.path / path.csv (UAV.py:431,461-464), the per-UAV log (UAV.py:268-310) and the simulator's
    score CSV (simulator.py:72-80,163-166) keep the reference's file names, headers and row shapes."""
    import csv
    import re
    xml = driver.make_config_dir(str(cfg_dir), "DQN", num_envs=3, num_uav=2)
    uav_xml = cfg_dir / "config" / "UAV.xml"
    uav_xml.write_text(re.sub(r"</Agent>\s*$", "<record_csv>1</record_csv></Agent>", uav_xml.read_text().rstrip()))
    sim = driver.simulator(xml)
    env = sim.env
    sim.Init_Record_Mod()
    random.seed(3)
    torch.manual_seed(3)
    sim.num_episodes = 10
    sim.StartAndTrain()
    # UAV.path: one position per step env 0's UAV actually took, ending where the state says it is
    for uav in env.Agents:
        assert 1 <= len(uav.path) <= 150 * 40 and len(uav.path[0]) == 3
        assert abs(uav.path[-1][0] - uav.position.x) < 1e-12 and abs(uav.path[-1][1] - uav.position.y) < 1e-12
    rows = list(csv.reader(open("path.csv")))
    assert rows and len(rows[0]) == 3
    assert any(len(rows) == len(u.path) and abs(float(rows[-1][0]) - u.path[-1][0]) < 1e-9 for u in env.Agents)
    # simulator score log
    rows = list(csv.reader(open(sim.result_path)))
    assert rows[0] == ["sum_Episode", "Episode", " Score", " Avg.Score", "eps-greedy", "success", "failed", "meet_threaten",
                       "loss", "step", "avg_trainning_time", "avg_testing_time", "total_time"]
    assert len(rows) == 11 and [int(r[1]) for r in rows[1:]] == list(range(1, 11)) and all(len(r) == 13 for r in rows[1:])
    assert abs(float(rows[1][4]) - driver.epsilon_annealing(1, sim.min_eps, sim.max_eps_episode)) < 1e-12
    # per-UAV log: header + one row every print_loop episodes
    logs = sorted(f for f in os.listdir("logs") if f.startswith("UAV_"))
    assert len(logs) == 2
    rows = list(csv.reader(open(os.path.join("logs", logs[0]))))
    assert rows[0][:4] == ["sum_Episode", "Episode", " Score", " Avg.Score"] and len(rows[0]) == 21 and rows[0][-1] == "Testing_time"
    assert len(rows) == 1 + 10 // env.print_loop and all(len(r) == 21 for r in rows[1:])
