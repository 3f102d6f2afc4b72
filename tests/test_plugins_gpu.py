"""The reference-named plugins on the real HIP backend (same contract checks as tests/test_plugins.py)."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


def test_move_agent_on_device_matches_oracle(tmp_path, monkeypatch):
    from dqn_based_uav_3d_path_planer_amd import driver
    from oracle import pyoracle as po
    monkeypatch.chdir(tmp_path)
    sim = driver.simulator(driver.make_config_dir(str(tmp_path), "DuelingDQN", num_envs=1))
    env, uav = sim.env, sim.env.Agents[0]
    assert type(env.backend).__name__ == "VecPathPlanEnv"
    w = load_golden("world_stock.npz")
    o = po.OracleUav(po.OracleWorld(w["buildings"]), po.default_uav_params(w))
    st, sub, alias = env.backend.get_state(0, 1, want_sub=True)
    o.set_state(*st[0][:5], *st[0][6:9], int(st[0][9]), sub[0][: int(st[0][11])])
    o.u.sub0_alias = int(alias[0])
    assert (np.abs(uav.state() - o.state()) / np.maximum(1, np.abs(o.state()))).max() <= 2e-6
    for a in (2, 0, 1, 1, 2, 0, 0, 1):
        nxt, r, d, info = env.Move_Agent(0, a)
        ro, do, io = o.update(-1.0 + a)
        assert abs(r - ro) <= 1e-9 and d == do and info == po.INFO_NAMES[io]
        assert (np.abs(nxt - o.state()) / np.maximum(1, np.abs(o.state()))).max() <= 2e-6
        assert uav.Step == o.u.step and abs(uav.position.x - o.u.px) < 1e-9


def test_run_eposide_on_device(tmp_path, monkeypatch):
    from dqn_based_uav_3d_path_planer_amd import driver
    monkeypatch.chdir(tmp_path)
    sim = driver.simulator(driver.make_config_dir(str(tmp_path), "DDQN", num_envs=512, num_uav=2))
    env = sim.env
    torch.manual_seed(0)
    res = env.run_eposide(0.5)
    for k in ("success", "lose", "meet_threaten", "normal", "loss", "sum_epoch", "eps", "score", "average_score", "step"):
        assert k in res
    assert res["lose"] + res["success"] >= 1024 and env.Check_uav_Done()
    assert all(u.Trainer.epoch > 150 for u in env.Agents)
    assert np.isfinite(res["loss"])


def test_config4_shape_sac_apf_multi_uav_on_device(tmp_path, monkeypatch):
    """BASELINE configs[3] in miniature: 4 UAVs per env, APF on, SAC continuous actions, through the plugins."""
    import re
    from dqn_based_uav_3d_path_planer_amd import driver
    monkeypatch.chdir(tmp_path)
    xml = driver.make_config_dir(str(tmp_path), "SAC", num_envs=128, num_uav=4)
    uav_xml = tmp_path / "config" / "UAV.xml"
    uav_xml.write_text(re.sub(r"<APF_Enabled>0</APF_Enabled>", "<APF_Enabled>1</APF_Enabled>", uav_xml.read_text()))
    sim = driver.simulator(xml)
    env = sim.env
    assert env.backend.cfg.apf_enabled == 1 and env.backend.N == 512
    torch.manual_seed(0)
    res = env.run_eposide(0.1)
    assert res["lose"] + res["success"] >= 512 and env.Check_uav_Done()
    tr = env.Agents[3].Trainer
    assert type(tr).__name__ == "SAC_Trainer" and tr.replay_memory.actions.shape[1] == 2 and tr.epoch > 150
    assert np.isfinite(float(res["loss"]))
