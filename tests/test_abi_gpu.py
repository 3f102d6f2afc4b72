"""Error behaviour of the C ABI on a live device: every misuse returns a negative code with a message, never a
crash or a silent fallback."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cfg(_lib, **kw):
    c = _lib.UavEnvConfig(abi_version=_lib.ABI_VERSION, device=0, n_envs=64, uav_per_env=1, max_subgoals=8, max_step=150,
                          apf_enabled=0, obs_dtype=_lib.OBS_F32, n_actions=3, len=500.0, width=500.0, h=100.0, max_v=1.0,
                          steering_angle=np.pi / 6)
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def test_calls_out_of_order_and_bad_arguments_are_refused():
    from dqn_based_uav_3d_path_planer_amd import _lib
    lib = _lib.load()
    h = C.c_void_p()
    assert lib.uavenv_create(C.byref(_cfg(_lib)), C.byref(h)) == 0
    obs = torch.zeros((64, 100), device="cuda")
    act = torch.zeros(64, dtype=torch.int32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    none = None
    # step / observe / reset before the world and the scenario bank exist
    rc = lib.uavenv_step(h, act.data_ptr(), _lib.ACT_INDEX_I32, obs.data_ptr(), none, none, none, none, none, none, none,
                         none, 0, s)
    assert rc == _lib.EINVAL and b"set_buildings" in lib.uavenv_last_error()
    assert lib.uavenv_observe(h, obs.data_ptr(), s) == _lib.EINVAL
    b = np.array([[100.0, 100.0, 0.0, 20.0, 30.0]])
    assert lib.uavenv_set_buildings(h, b.ctypes.data, None, 1) == 0
    assert lib.uavenv_reset_all(h, 1, s) == _lib.EINVAL and lib.uavenv_last_error() != b""
    # too many buildings; scenario with more sub-goals than the capacity; null pointers
    big = np.zeros((65, 5))
    assert lib.uavenv_set_buildings(h, big.ctypes.data, None, 65) == _lib.EINVAL
    sg, sub, ns = np.zeros((1, 6)), np.zeros((1, 8, 3)), np.array([9], dtype=np.int32)
    assert lib.uavenv_load_scenarios(h, sg.ctypes.data, sub.ctypes.data, ns.ctypes.data, 1) == _lib.EINVAL
    assert lib.uavenv_load_scenarios(h, None, sub.ctypes.data, ns.ctypes.data, 1) == _lib.EINVAL
    ns[0] = 2
    assert lib.uavenv_load_scenarios(h, sg.ctypes.data, sub.ctypes.data, ns.ctypes.data, 1) == 0
    assert lib.uavenv_reset_all(h, 1, s) == 0
    # unknown action encoding, null action pointer
    assert lib.uavenv_step(h, act.data_ptr(), 17, obs.data_ptr(), none, none, none, none, none, none, none, none, 0,
                           s) == _lib.EINVAL
    assert lib.uavenv_step(h, None, _lib.ACT_INDEX_I32, obs.data_ptr(), none, none, none, none, none, none, none, none, 0,
                           s) == _lib.EINVAL
    # and the handle still works after all those refusals
    assert lib.uavenv_step(h, act.data_ptr(), _lib.ACT_INDEX_I32, obs.data_ptr(), none, none, none, none, none, none,
                           none, none, 0, s) == 0
    torch.cuda.synchronize()
    assert torch.isfinite(obs).all()
    # state windows out of range
    out = np.zeros((4, 16))
    assert lib.uavenv_get_state(h, 62, 4, out.ctypes.data, None, None) == _lib.EINVAL
    assert lib.uavenv_get_state(h, -1, 1, out.ctypes.data, None, None) == _lib.EINVAL
    assert lib.uavenv_destroy(h) == 0 and lib.uavenv_destroy(None) == 0


def test_learner_and_replay_entry_points_validate_their_arguments():
    from dqn_based_uav_3d_path_planer_amd import _lib
    lib = _lib.load()
    flat = torch.zeros((4, 6659), device="cuda")
    good = _lib.UavDqnNet(flat[0].data_ptr(), flat[1].data_ptr(), flat[2].data_ptr(), flat[3].data_ptr(), 100, 64, 3, 0)
    assert lib.uavenv_dqn_num_params(C.byref(good)) == 6659
    obs = torch.zeros((64, 100), device="cuda")
    idx = torch.zeros(64, dtype=torch.int32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    for bad in (_lib.UavDqnNet(flat[0].data_ptr(), flat[1].data_ptr(), flat[2].data_ptr(), flat[3].data_ptr(), 128, 64, 3, 0),
                _lib.UavDqnNet(flat[0].data_ptr(), flat[1].data_ptr(), flat[2].data_ptr(), flat[3].data_ptr(), 100, 32, 3, 0),
                _lib.UavDqnNet(flat[0].data_ptr(), flat[1].data_ptr(), flat[2].data_ptr(), flat[3].data_ptr(), 100, 64, 15, 0),
                _lib.UavDqnNet(None, flat[1].data_ptr(), flat[2].data_ptr(), flat[3].data_ptr(), 100, 64, 3, 0)):
        assert lib.uavenv_dqn_act(C.byref(bad), obs.data_ptr(), _lib.OBS_F32, 64, 0.1, 1, 1, idx.data_ptr(), None, None,
                                  s) == _lib.EINVAL
    assert lib.uavenv_dqn_act(C.byref(good), None, _lib.OBS_F32, 64, 0.1, 1, 1, idx.data_ptr(), None, None, s) == _lib.EINVAL
    assert lib.uavenv_dqn_act(C.byref(good), obs.data_ptr(), _lib.OBS_F32, 64, 0.1, 1, 1, idx.data_ptr(), None, None, s) == 0
    ring = _lib.UavReplayRing(obs.data_ptr(), idx.data_ptr(), obs.data_ptr(), obs.data_ptr(), None, 3, 16, _lib.OBS_F32, 1)
    part = torch.zeros((4, 6661), device="cuda")
    # batch not a multiple of 64, head out of range, nothing filled
    for batch, head, filled in ((100, 0, 1), (64, 3, 1), (64, 0, 0), (64, 0, 3)):
        assert lib.uavenv_dqn_grad(C.byref(ring), head, filled, batch, 1, 1, None, C.byref(good), 0, 0.99, 0,
                                   part.data_ptr(), s) == _lib.EINVAL
    per = _lib.UavPer(None, None, None, 100, 0)
    assert lib.uavenv_per_rebuild(C.byref(per), s) == _lib.EINVAL
    assert lib.uavenv_per_num_chunks(0) == 0 and lib.uavenv_per_num_chunks(1025) == 2 and lib.uavenv_per_rotation(100) == 28
    torch.cuda.synchronize()
