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


def test_on_policy_episode_on_device(tmp_path, monkeypatch):
    """Is_On_Policy = 1 (Envs/PathPlan_City.py:386-436) on the real backend: 256 envs x 2 UAVs collect their episode, the fused
    trainers take ONE update each on it (an arbitrary batch through FusedDQNLearner.learn), the replay memories stay empty."""
    import re
    from dqn_based_uav_3d_path_planer_amd import driver
    monkeypatch.chdir(tmp_path)
    xml = driver.make_config_dir(str(tmp_path), "DuelingDQN", num_envs=256, num_uav=2)
    s = open(xml).read()
    s, n = re.subn(r"<Is_On_Policy>\s*0\s*</Is_On_Policy>", "<Is_On_Policy>1</Is_On_Policy>", s)
    assert n == 1
    open(xml, "w").write(s)
    env = driver.simulator(xml).env
    assert env is not None and env.Is_On_Policy == 1 and not env.fast and type(env.backend).__name__ == "VecPathPlanEnv"
    torch.manual_seed(0)
    w0 = [u.Trainer.q_local.state_dict()["fc1.weight"].clone() for u in env.Agents]
    res = env.run_eposide(0.5)
    moved = res["normal"] + res["success"] + res["lose"]
    assert env.Check_uav_Done() and res["lose"] + res["success"] >= 512
    assert sum(len(u.transition_dict["states"]) for u in env.Agents) == moved
    for u, w in zip(env.Agents, w0):
        assert u.Trainer.epoch == 1 and len(u.Trainer.replay_memory) == 0
        assert not torch.equal(u.Trainer.q_local.state_dict()["fc1.weight"], w)      # the one update moved the weights
    assert np.isfinite(float(res["loss"]))


def _set_xml(path, **tags):
    import re
    s = path.read_text()
    for k, v in tags.items():
        if re.search(rf"<{k}>[^<]*</{k}>", s):
            s = re.sub(rf"<{k}>[^<]*</{k}>", f"<{k}>{v}</{k}>", s)
        else:
            s = s.replace("</Trainer>", f"    <{k}>{v}</{k}>\n</Trainer>")
    path.write_text(s)


def _config4(tmp_path, num_envs, **trainer_tags):
    import re
    from dqn_based_uav_3d_path_planer_amd import driver
    xml = driver.make_config_dir(str(tmp_path), "SAC", num_envs=num_envs, num_uav=4)
    uav_xml = tmp_path / "config" / "UAV.xml"
    uav_xml.write_text(re.sub(r"<APF_Enabled>0</APF_Enabled>", "<APF_Enabled>1</APF_Enabled>", uav_xml.read_text()))
    if trainer_tags:
        _set_xml(tmp_path / "config" / "Trainer.xml", **trainer_tags)
    return driver.simulator(xml)


def test_config4_shape_sac_apf_multi_uav_on_device(tmp_path, monkeypatch):
    """BASELINE configs[3] in miniature: 4 UAVs per env, APF on, SAC continuous actions, through the plugins -- on the
    general per-step path (<fused>0</fused>: PyTorch SACLearner, one torch replay per trainer)."""
    monkeypatch.chdir(tmp_path)
    sim = _config4(tmp_path, 128, fused=0)
    env = sim.env
    assert env.backend.cfg.apf_enabled == 1 and env.backend.N == 512 and not env.fast_sac
    torch.manual_seed(0)
    res = env.run_eposide(0.1)
    assert res["lose"] + res["success"] >= 512 and env.Check_uav_Done()
    tr = env.Agents[3].Trainer
    assert type(tr).__name__ == "SAC_Trainer" and tr.replay_memory.actions.shape[1] == 2 and tr.epoch > 150
    assert type(tr.learner).__name__ == "SACLearner" and np.isfinite(float(res["loss"]))


def test_config4_fused_sac_episode_path(tmp_path, monkeypatch):
    """The same shape on the fast path (the default): packed ring shared by the four UAV slots, one FusedSACLearner per
    slot (csrc/sac.hip), act / step / draw / learn launched per time step, host read-back every done_check steps.
    Contract checks as for the general path + the trainers really learn (parameters move, targets follow, checkpoints
    round-trip) + an episode costs a few launches per step, not hundreds."""
    import time
    monkeypatch.chdir(tmp_path)
    sim = _config4(tmp_path, 2048, Batch_Size=2048, replay_size=65536)
    env = sim.env
    assert env.fast_sac and env.backend.packed and env.backend.cfg.apf_enabled == 1 and env.backend.N == 8192
    tr = env.Agents[2].Trainer
    assert type(tr.learner).__name__ == "FusedSACLearner" and len(tr.replay_memory.memory) == 0
    w0 = tr.actor.fc1.weight.detach().clone()
    c0 = tr.critic_1.fc2.weight.detach().clone()
    t0w = tr.target_critic_1.fc2.weight.detach().clone()
    torch.manual_seed(0)
    res = env.run_eposide(0.1)
    steps = env.steps_last_episode
    assert res["lose"] + res["success"] >= 8192 and env.Check_uav_Done() and steps >= 150
    for k in ("success", "lose", "normal", "loss", "sum_epoch", "score", "average_score", "step"):
        assert k in res
    assert np.isfinite(float(res["loss"])) and tr.epoch >= steps
    assert len(tr.replay_memory.memory) == min(steps + env.done_check, env._ring.frames - 1) * 2048 or len(tr.replay_memory.memory) > 0
    assert float((tr.actor.fc1.weight.detach() - w0).abs().max()) > 0 and float((tr.critic_1.fc2.weight.detach() - c0).abs().max()) > 0
    assert float((tr.target_critic_1.fc2.weight.detach() - t0w).abs().max()) > 0
    for net in (tr.actor, tr.critic_1, tr.critic_2, tr.target_critic_1, tr.target_critic_2):
        assert all(torch.isfinite(p).all() for p in net.parameters())
    # checkpoint round trip (model + Adam moments, torch.optim.Adam's own format)
    tr.save()
    ck = torch.load(tr._path("critic_1"))
    assert set(ck["optimizer"]) == {"state", "param_groups"} and float(ck["optimizer"]["state"][0]["step"]) == tr.learner.adam_steps
    m_saved = tr.learner._blocks[1].clone()
    a_saved = tr.actor.fc1.weight.detach().clone()
    with torch.no_grad():
        tr.actor.fc1.weight.add_(1.0)
        tr.learner._blocks[1].zero_()
    tr.Load_Mod()
    assert torch.equal(tr.actor.fc1.weight, a_saved) and torch.equal(tr.learner._blocks[1], m_saved)
    # ... and a PyTorch-learner trainer (<fused>0</fused>) reads the same files: weights, Adam moments, step count
    from dqn_based_uav_3d_path_planer_amd import factories
    p2 = dict(tr.param, fused="0", model_dir=tr.model_dir)
    tr2 = factories.TrainerFactory().Create_Trainer(p2)
    assert not tr2.fused and torch.equal(tr2.actor.fc1.weight.detach(), a_saved) and tr2.epoch == tr.epoch
    st = tr2.learner.actor_optimizer.state[tr2.actor.fc1.weight]
    assert float(st["step"]) == tr.learner.adam_steps
    off = 0
    assert torch.equal(st["exp_avg"].reshape(-1), m_saved[off:off + st["exp_avg"].numel()])
    # speed: a second episode, timed
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    env.run_eposide(0.1)
    torch.cuda.synchronize()
    per_step = (time.perf_counter() - t0) / env.steps_last_episode
    print(f"fused SAC plugin episode: {env.steps_last_episode} steps, {per_step * 1e6:.0f} us/step (4 slots x (act + draw + 4-launch update))")
    assert per_step <= 2.5e-3
    # trainer.update(transition_dict) on arbitrary f32 states: the trainer moves itself to the PyTorch learner (same weights,
    # same Adam state) instead of raising
    a_now, steps_now = tr.actor.fc1.weight.detach().clone(), tr.learner.adam_steps
    out = tr.update({"states": []})
    assert not tr.fused and type(tr.learner).__name__ == "SACLearner" and "sum_epoch" in out
    assert torch.equal(tr.actor.fc1.weight.detach(), a_now)
    assert float(tr.learner.actor_optimizer.state[tr.actor.fc1.weight]["step"]) == steps_now
    rng = np.random.default_rng(0)
    td = {"states": rng.normal(0, 1, (64, 100)).astype(np.float32), "actions": rng.uniform(-1, 1, (64, 2)).astype(np.float32),
          "rewards": rng.normal(0, 1, 64).astype(np.float32), "next_states": rng.normal(0, 1, (64, 100)).astype(np.float32),
          "dones": np.zeros(64, np.float32)}
    tr.update(td)
    assert float(tr.learner.actor_optimizer.state[tr.actor.fc1.weight]["step"]) == steps_now + 1
    assert not torch.equal(tr.actor.fc1.weight.detach(), a_now) and np.isfinite(float(tr.loss))


def test_fused_episode_path_runs_at_bench_speed(tmp_path, monkeypatch):
    """the plugin surface must reach the fast path.  driver.simulator is this repo's mirror of the
    reference's simulator.py (the unmodified reference file drives the same plugins in tests/test_plugins.py on the CPU
    backend; /root/reference does not exist on the GPU box).  At BASELINE configs[1]'s shape -- 16 384 envs, DQN, batch
    16 384, 1 M replay -- an episode through simulator -> PathPlan_City.run_eposide must cost no more than 2x per step
    what the bare C loop (bench.py's loop: HotLoop on the same ring / learner types) costs."""
    import time
    from dqn_based_uav_3d_path_planer_amd import driver
    from dqn_based_uav_3d_path_planer_amd.learner import FusedDQNLearner
    from dqn_based_uav_3d_path_planer_amd.loop import HotLoop
    from dqn_based_uav_3d_path_planer_amd.replay import DeviceReplayRing
    monkeypatch.chdir(tmp_path)
    xml = driver.make_config_dir(str(tmp_path), "DQN", num_envs=16384, num_episodes=20)
    _set_xml(tmp_path / "config" / "Trainer.xml", Batch_Size=16384, replay_size=1 << 20)
    sim = driver.simulator(xml)
    env = sim.env
    tr = env.Agents[0].Trainer
    assert env.fast and type(tr.learner).__name__ == "FusedDQNLearner" and env.backend.packed
    assert len(tr.replay_memory.memory) == 0
    env.run_eposide(0.5)                                   # warm-up episode (kernel load, allocator)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = env.run_eposide(0.3)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    steps = env.steps_last_episode
    assert steps >= 151 and res["success"] + res["lose"] >= 16384 and env.Check_uav_Done()
    assert res["normal"] > 100 * 16384 and np.isfinite(float(res["loss"])) and tr.epoch >= steps
    assert len(tr.replay_memory.memory) == min(tr.epoch, env._ring.frames - 1) * 16384 or len(tr.replay_memory.memory) > 0
    per_step = dt / steps
    # the same loop without the plugin layer
    L = FusedDQNLearner({"NetWork": "Qnet2", "w": "100", "hiden_dim": "64", "output": "3"}, "dqn", device="cuda:0")
    ring = DeviceReplayRing(env.backend, 1 << 20, discrete=True)
    ring.reset(seed=3)
    hot = HotLoop(ring, L, 16384, seed=1, eps=0.3)
    hot.run(64)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    hot.run(512)
    torch.cuda.synchronize()
    bare = (time.perf_counter() - t0) / 512
    hot.close()
    print(f"plugin episode: {steps} steps, {per_step * 1e6:.1f} us/step; bare C loop {bare * 1e6:.1f} us/step")
    assert per_step <= 2.0 * bare, (per_step, bare)


def test_fused_and_general_episode_paths_agree_on_the_contract(tmp_path, monkeypatch):
    """Same XML, <fast_path> 1 and 0: both finish every env, fill the replay, train, and report the same result keys;
    both count every moved agent-step as normal, success or lose."""
    from dqn_based_uav_3d_path_planer_amd import driver
    monkeypatch.chdir(tmp_path)
    out = {}
    for fast in (1, 0):
        xml = driver.make_config_dir(str(tmp_path), "DuelingDQN", num_envs=256)
        p = tmp_path / "config" / "PathPlan_City.xml"
        p.write_text(p.read_text().replace("<seed>42</seed>", f"<seed>42</seed>\n        <fast_path>{fast}</fast_path>"))
        env = driver.simulator(xml).env
        assert env.fast == bool(fast)
        torch.manual_seed(0)
        res = env.run_eposide(0.4)
        assert env.Check_uav_Done() and res["success"] + res["lose"] >= 256
        assert env.Agents[0].Trainer.epoch > 150 and np.isfinite(float(res["loss"]))
        out[fast] = (set(res), res["normal"] + res["success"] + res["lose"])
        assert out[fast][1] >= 151 * 256                   # every agent moves at least Max_Step + 1 times
    assert out[1][0] == out[0][0]


def test_fused_trainer_checkpoint_roundtrip(tmp_path, monkeypatch):
    """f2: {'model', 'optimizer', 'epoch'} files of the reference's names, with the fused learner's Adam moments inside
    in torch.optim.Adam's own format; Load_Mod restores weights, moments and the update count."""
    from dqn_based_uav_3d_path_planer_amd import factories
    g = load_golden("learner_DuelingDQN_Trainer.npz")
    p = {"Trainer_Type": "DuelingDQN_Trainer", "NetWork": "VAnet2", "w": "100", "hiden_dim": "64", "output": "3",
         "Batch_Size": "64", "replay_size": "1000", "name": "UAV_0", "model_dir": str(tmp_path), "save_loop": "1000000",
         "device": "cuda:0"}
    tr = factories.TrainerFactory().Create_Trainer(dict(p))
    assert tr.fused
    td = {k: g[k] for k in ("states", "actions", "rewards", "next_states", "dones")}
    for _ in range(5):
        tr.update(td)
    tr.save()
    ck = torch.load(tmp_path / "q_local_Dueling_UAV_0.pth") if (tmp_path / "q_local_Dueling_UAV_0.pth").exists() \
        else torch.load(next(tmp_path.glob("q_local_*UAV_0.pth")))
    assert set(ck) == {"model", "optimizer", "epoch"} and ck["epoch"] == 5
    ref_opt = torch.optim.Adam(tr.q_local.parameters())
    ref_opt.load_state_dict(ck["optimizer"])                # loadable by a plain torch Adam
    tr2 = factories.TrainerFactory().Create_Trainer(dict(p))   # Load_Mod in the constructor
    assert tr2.epoch == 5 and torch.equal(tr2.learner.flat, tr.learner.flat)
    a, b = tr.update(td), tr2.update(td)
    assert torch.equal(tr2.learner.flat, tr.learner.flat)


def test_prioritised_replay_stays_on_the_fused_path(tmp_path, monkeypatch):
    """IsPriority_Replay = 1 with a DQN-family trainer: run_eposide keeps the fused path (HotLoop with a DevicePER over the
    ring's slots: sampling, importance weights, the weighted update and batch_update are enqueued from C)."""
    from dqn_based_uav_3d_path_planer_amd import driver
    monkeypatch.chdir(tmp_path)
    xml = driver.make_config_dir(str(tmp_path), "DuelingDQN", num_envs=512)
    _set_xml(tmp_path / "config" / "Trainer.xml", IsPriority_Replay=1, Batch_Size=256)
    env = driver.simulator(xml).env
    tr = env.Agents[0].Trainer
    assert env.fast and tr.fused and env._per is not None
    res = env.run_eposide(0.3)
    assert env.hot_loop_with_per            # the C loop ran with the DevicePER inside (the loop object is rebuilt per episode)
    assert env.Check_uav_Done() and np.isfinite(float(res["loss"])) and tr.epoch > 100
    per, ring = env._per, env._ring
    prio = per.prio.view(ring.frames, -1)
    assert float(prio[ring.head].abs().max()) == 0.0                       # the frame under construction is retired
    live = prio[prio > 0]
    fresh = (0.0 + per.epsilon) ** per.alpha
    assert live.numel() > 0 and ((live - fresh).abs() > 1e-9).float().mean().item() > 0.05      # re-prioritised by batch_update
    assert live.max().item() <= 1.0 + 1e-9 and per.beta > 0.4 and torch.isfinite(tr.learner.flat).all()


def test_prioritised_replay_stays_on_the_fused_sac_path(tmp_path, monkeypatch):
    """The reference's own use of ReplayTree (Trainer/SAC_Trainer.py:336-352) on the fast path: 4 UAVs per env, APF on, one
    fused SAC trainer + one DevicePER per UAV slot; per step and slot new-frame priorities, ReplayTree.sample, importance
    weights, the fused update (weights in the critic losses, |TD| out) and batch_update are stream-ordered launches."""
    monkeypatch.chdir(tmp_path)
    sim = _config4(tmp_path, 512, Batch_Size=256, replay_size=16384, IsPriority_Replay=1)
    env = sim.env
    assert env.fast_sac and all(u.Trainer.fused for u in env.Agents) and all(p is not None for p in env._sac_per)
    torch.manual_seed(0)
    res = env.run_eposide(0.1)
    assert env.Check_uav_Done() and np.isfinite(float(res["loss"]))
    assert env.sac_c_loop_used            # prioritised replay runs INSIDE uavenv_sac_loop_run (round 4), not from the Python loop
    ring = env._ring
    reprioritised = 0
    for j, per in enumerate(env._sac_per):
        tr = env.Agents[j].Trainer
        assert tr.epoch > 100 and all(torch.isfinite(p).all() for p in tr.actor.parameters())
        prio = per.prio.view(ring.frames, -1)
        assert float(prio[ring.head].abs().max()) == 0.0 and per.beta > 0.4
        # a retired / never-valid row is an empty leaf: rows of agents that only waited carry no priority -- also at the end of
        # the episode, when a slot's whole tree is empty and the sampler can only return empty leaves (batch_update skips them)
        valid_j = ring.valid.view(ring.frames, env.num_envs, env.num_UAV)[:, :, j]
        off = (valid_j == 0) & (torch.arange(ring.frames, device=prio.device).view(-1, 1) != ring.head)
        assert float(prio[off].abs().max()) == 0.0
        live = prio[prio > 0]
        fresh = (0.0 + per.epsilon) ** per.alpha
        if live.numel():
            assert live.max().item() <= 1.0 + 1e-9
            reprioritised += int(((live - fresh).abs() > 1e-9).sum())
    assert reprioritised > 0


@pytest.mark.parametrize("trainer", ["DuelingDQN", "SAC"])
def test_prioritised_replay_through_the_plugins(trainer, tmp_path, monkeypatch):
    """IsPriority_Replay = 1 (BaseClass/replay_buffer.py:121-223, Trainer/SAC_Trainer.py:336-352) on the general per-step
    path (<fused>0</fused>): the trainer plugins sample through DevicePER with importance weights and feed |TD error| back
    into the priorities."""
    from dqn_based_uav_3d_path_planer_amd import driver
    monkeypatch.chdir(tmp_path)
    xml = driver.make_config_dir(str(tmp_path), trainer, num_envs=64)
    _set_xml(tmp_path / "config" / "Trainer.xml", IsPriority_Replay=1, fused=0)
    env = driver.simulator(xml).env
    tr = env.Agents[0].Trainer
    assert not env.fast and tr.replay_memory.per is not None
    torch.manual_seed(0)
    res = env.run_eposide(0.3)
    assert env.Check_uav_Done() and np.isfinite(float(res["loss"])) and tr.epoch > 100
    prio = tr.replay_memory.per.prio[: len(tr.replay_memory)]
    fresh = (0.0 + tr.replay_memory.per.epsilon) ** tr.replay_memory.per.alpha
    assert (prio > 0).all() and ((prio - fresh).abs() > 1e-9).float().mean().item() > 0.2     # many were re-prioritised
    assert prio.max().item() <= 1.0 + 1e-9                                                    # clip at 1 (:219-221)


@pytest.mark.parametrize("envs,batch,replay,tpw,per", [(512, 512, 8192, 0, 0), (2048, 8192, 65536, 2, 0), (512, 512, 8192, 0, 1)])
def test_sac_c_loop_equals_the_python_loop(tmp_path, monkeypatch, envs, batch, replay, tpw, per):
    """csrc/loop.hip: uavenv_sac_loop_run -- per step the N(0,1) draws, U x get_action, the env step (APF on), one replay
    draw and U x the four launches of the fused SAC update, enqueued from C -- against the same sequence issued launch by
    launch from Python (PathPlan_City._run_eposide_fused_sac with <sac_c_loop>0</sac_c_loop>): ring, every parameter block
    of every slot (weights, targets, Adam moments, log_alpha), update counts -- bit for bit.  (Speed: both are GPU-bound at
    this size -- ~375 us per step = ~22 latency-bound launches -- so the C loop removes the interpreter, not time.)
    Second case: 4 x 128 tiles -- the C loop's four-slot launches take two tiles per workgroup (uavenv_sac_partial_rows_n), a
    slot alone would take one; the Python loop's learners are told the same partition (FusedSACLearner.tiles_per_wg) and the
    comparison stays bit for bit.
    Third case: IsPriority_Replay = 1 -- the reference's own use of ReplayTree (Trainer/SAC_Trainer.py:336-352) INSIDE the C
    loop (per step and slot: new-frame priorities, rebuild, ReplayTree.sample, importance weights, the update with weights in
    and |TD| out, batch_update) against the same launches issued from Python: priorities and beta of every slot's tree too."""
    import time
    monkeypatch.chdir(tmp_path)
    out = []
    for c_loop in ("1", "0"):
        torch.manual_seed(0)        # BEFORE the trainers are built: this torch build seeds its default generator from the OS, so an
                                    # env built first in a process would otherwise start from other weights than the second one
        sim = _config4(tmp_path, envs, Batch_Size=batch, replay_size=replay, IsPriority_Replay=per)
        env = sim.env
        env.param["sac_c_loop"] = c_loop
        assert env.fast_sac and all((p is not None) == bool(per) for p in env._sac_per)
        if c_loop == "0":
            for b in env._sac_batches:               # the partition of the C loop's four-slot launches
                b.tiles_per_wg = tpw
        torch.manual_seed(0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = env.run_eposide(0.1)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / env.steps_last_episode
        assert env.sac_c_loop_used == (c_loop == "1")
        L = [u.Trainer.learner for u in env.Agents]
        out.append(dict(ring={k: getattr(env._ring, k).clone() for k in ("obs", "action", "reward", "done", "valid")},
                        a1=env._a1.clone(), blocks=[torch.cat([x._blocks.reshape(-1), x._cblocks.reshape(-1), x._alpha_mv,
                                                               x.log_alpha.reshape(1)]) for x in L],
                        counts=[(x.epoch, x.adam_steps) for x in L], cursor=(env._ring.head, env._ring.filled, env._sac_counter),
                        steps=env.steps_last_episode, us=dt * 1e6, loss=float(res["loss"]),
                        per=[(p.prio.clone(), p.beta, p.n_entries) for p in env._sac_per if p is not None]))
    a, b = out
    assert len(a["per"]) == (4 if per else 0)
    for (pa, ba, na), (pb, bb, nb_) in zip(a["per"], b["per"]):
        assert torch.equal(pa, pb) and ba == bb and na == nb_ and ba > 0.4
    assert a["cursor"] == b["cursor"] and a["counts"] == b["counts"] and a["steps"] == b["steps"] and a["steps"] >= 150
    for k in a["ring"]:
        assert torch.equal(a["ring"][k], b["ring"][k]), k
    assert torch.equal(a["a1"], b["a1"])
    for j in range(4):
        assert torch.equal(a["blocks"][j], b["blocks"][j]), j
    assert a["loss"] == b["loss"]
    print(f"fused SAC episode, {envs} envs x 4 UAVs: C loop {a['us']:.0f} us/step, Python loop {b['us']:.0f} us/step")
    assert a["us"] < 1.25 * b["us"]


def test_fresh_plans_between_episodes(tmp_path, monkeypatch):
    """UAV.reset plans a new RRT path at every reset (Agents/UAV.py:327-366).  The plugin plans a slice of the reset bank in the
    background of each episode (csrc/rrt.hip, LDS-free form, low-priority stream) and hands it over at the next episode
    boundary, where every agent is reset anyway: after a few episodes the bank is not the one the env was built with, every row
    is still a valid path, and <fresh_plans>0</fresh_plans> keeps the bank fixed."""
    from dqn_based_uav_3d_path_planer_amd import driver
    monkeypatch.chdir(tmp_path)
    xml = driver.make_config_dir(str(tmp_path), "DQN", num_envs=64, num_uav=1)
    env = driver.simulator(xml).env
    assert env.fresh_plans and env.fast
    sg0, _, _ = env.backend.bank_read()
    for _ in range(4):
        env.run_eposide(0.5)
    torch.cuda.synchronize()
    st = env.backend.replan_stats()
    sg1, sub1, ns1 = env.backend.bank_read()
    assert st["refreshes"] >= 2 and st["rows_committed"] >= 200 and st["rows_in_use"] == 0, st
    changed = np.any(sg1 != sg0, axis=1)
    assert changed.sum() >= 200 and ((ns1 >= 2) & (ns1 <= env.backend.K)).all()
    for k in np.nonzero(changed)[0][:100]:
        p = sub1[k, :ns1[k]]
        assert np.array_equal(p[0], sg1[k, :3]) and np.array_equal(p[-1], sg1[k, 3:])
    s = open(xml).read().replace("<num_UAV>", "<fresh_plans>0</fresh_plans>\n        <num_UAV>", 1)
    open(xml, "w").write(s)
    env2 = driver.simulator(xml).env
    assert not env2.fresh_plans
    b0 = env2.backend.bank_read()[0]
    env2.run_eposide(0.5)
    env2.run_eposide(0.5)
    assert np.array_equal(env2.backend.bank_read()[0], b0) and env2.backend.replan_stats()["refreshes"] == 0


@pytest.mark.parametrize("trainer", ["DQN", "SAC"])
def test_done_check_does_not_change_what_is_learnt(trainer, tmp_path, monkeypatch):
    """The fused episode paths look at the device only every <done_check> steps, so up to done_check - 1 passes are enqueued
    behind the last moving step.  Round 4: those passes change nothing -- the step kernel stamps a device word when it moves
    an agent and the Adam launches / batch_update check it (uavenv_set_moved_word, uavenv_dqn_reduce_adam_gated,
    UavSacAdam.go_word) -- so the learner takes exactly one update per moving step, as the reference's loop, which leaves
    right after the last one (Envs/PathPlan_City.py:456-459).  done_check = 1 (no surplus pass can exist) and done_check = 8
    must therefore end with the same weights, moments, update counts -- bit for bit.  (The replay cursor and the Philox counter
    are rewound past those passes too; the ring is sized so that it does not wrap here: on a wrapped ring the surplus passes
    overwrite its oldest frames, a different number of them for a different done_check.)"""
    from dqn_based_uav_3d_path_planer_amd import driver
    monkeypatch.chdir(tmp_path)
    out = []
    for dc in (1, 8):
        torch.manual_seed(0)                          # the same initial weights in both runs
        if trainer == "SAC":
            sim = _config4(tmp_path, 128, Batch_Size=128, replay_size=400000)
        else:
            xml = driver.make_config_dir(str(tmp_path), "DQN", num_envs=256, num_uav=1)
            _set_xml(tmp_path / "config" / "Trainer.xml", replay_size=800000)
            sim = driver.simulator(xml)
        env = sim.env
        env.done_check = dc
        assert env.fast or env.fast_sac
        torch.manual_seed(0)
        for _ in range(2):
            env.run_eposide(0.2)
        torch.cuda.synchronize()
        if trainer == "SAC":
            L = [u.Trainer.learner for u in env.Agents]
            blocks = [torch.cat([x._blocks.reshape(-1), x._cblocks.reshape(-1), x._alpha_mv, x.log_alpha.reshape(1)]).clone() for x in L]
            counts = [(x.epoch, x.adam_steps) for x in L]
        else:
            L = env.Agents[0].Trainer.learner
            blocks, counts = [L.flat.clone()], [(L.epoch,)]
        out.append((blocks, counts, env.steps_last_episode, env.surplus_passes_last_episode))
    a, b = out
    assert a[3] == 1 and a[2] == b[2] and a[1] == b[1], (a[1:], b[1:])      # (the pass that notices the end is always there)
    assert 1 <= b[3] <= 8 and (b[2] + b[3]) % 8 == 0  # done_check = 8 enqueued whole chunks behind the end of the episode ...
    for x, y in zip(a[0], b[0]):
        assert torch.equal(x, y)                     # ... and they changed nothing
