"""Packed observation rows (UAVENV_OBS_PACKED: 15 f32 scalars + 80 flag bits, 80 B per row) are a LOSSLESS image of the
f32 rows of state_PathPlan (Agents/UAV.py:515-567): every kernel that writes or reads them must agree bit for bit with
the f32-row path on the same trajectory."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

PARAM = {"w": "100", "hiden_dim": "64", "output": "3", "LEARNING_RATE": "0.001", "gamma": "0.99", "Update_loop": "3"}


def _pair(n, frames, uav=1, **kw):
    from dqn_based_uav_3d_path_planer_amd.data import make_city26_env
    from dqn_based_uav_3d_path_planer_amd.replay import DeviceReplayRing
    out = []
    for dt in (torch.float32, "packed"):
        env = make_city26_env(n, obs_dtype=dt, uav_per_env=uav, **kw)
        ring = DeviceReplayRing(env, frames * env.N, discrete=True)
        ring.reset(seed=5)
        out.append((env, ring))
    return out


@pytest.mark.parametrize("n,one_wave", [(1000, False), (1000, True), (70000, False)])   # coop kernel / one-wave rows / tile kernel
def test_env_kernels_write_packed_rows_that_unpack_to_the_f32_rows(n, one_wave):
    from dqn_based_uav_3d_path_planer_amd import _lib
    (e32, r32), (epk, rpk) = _pair(n, 3)
    assert rpk.obs.shape[-1] == 20 and rpk.obs.dtype == torch.int32
    assert torch.equal(epk.unpack(rpk.current_obs()), r32.current_obs())          # k_observe
    if one_wave:
        r32.extra_flags = rpk.extra_flags = _lib.STEP_ONE_WAVE
    gen = torch.Generator(device="cuda").manual_seed(2)
    for t in range(160):                      # past the first timeouts: auto-resets in the mix
        a = torch.randint(0, 3, (n,), generator=gen, device="cuda", dtype=torch.int32)
        r32.current_action().copy_(a)
        rpk.current_action().copy_(a)
        r32.step_env(auto_reset=True)
        rpk.step_env(auto_reset=True)
        if t % 20 == 0 or t > 150:
            assert torch.equal(epk.unpack(rpk.current_obs()), r32.current_obs()), t
            assert torch.equal(rpk.reward[(rpk.head - 1) % rpk.frames], r32.reward[(r32.head - 1) % r32.frames])
    # reserved dwords stay zero, f16 unpack = rounding of the f32 unpack
    assert int(rpk.current_obs()[:, 3].abs().max()) == 0 and int(rpk.current_obs()[:, 19].abs().max()) == 0
    assert torch.equal(epk.unpack(rpk.current_obs(), torch.float16), r32.current_obs().half())
    e32.close()
    epk.close()


@pytest.mark.parametrize("kind,net", [("dqn", "Qnet2"), ("dueling", "VAnet2")])
def test_fused_act_and_learner_on_packed_rows_equal_the_f32_row_path(kind, net):
    from dqn_based_uav_3d_path_planer_amd.learner import FusedDQNLearner
    (e32, r32), (epk, rpk) = _pair(2048, 6)
    torch.manual_seed(0)
    A = FusedDQNLearner(dict(PARAM, NetWork=net), kind, device="cuda:0")
    B = FusedDQNLearner(dict(PARAM, NetWork=net), kind, device="cuda:0")
    B.flat.copy_(A.flat)
    for t in range(9):                        # act -> step -> learn on both; the ring wraps
        A.act(r32.current_obs(), 0.3, 7, t, index_out=r32.current_action())
        B.act(rpk.current_obs(), 0.3, 7, t, index_out=rpk.current_action())
        assert torch.equal(r32.current_action(), rpk.current_action()), t
        r32.step_env(auto_reset=True)
        rpk.step_env(auto_reset=True)
        la = float(A.learn_from_ring(r32, 2048, 7, t))
        lb = float(B.learn_from_ring(rpk, 2048, 7, t))
        # (round 4: on packed rows layer 1 runs in the split form -- the flag columns against fc1 as two f16 terms on the f16
        # matrix pipe, csrc/qnet_device.hpp -- every product exact, the f32 sums in another order than the f32-row kernels': the two
        # paths agree to rounding, no longer bit for bit; the actions above still have to be the same ones)
        assert abs(la - lb) <= 2e-6 * max(1.0, abs(la)), (t, la, lb)
    dw = (A.flat[0] - B.flat[0]).abs()
    lr = float(PARAM["LEARNING_RATE"])          # nine Adam steps of lr each: the weight sets stay within a hundredth of ONE step,
    assert dw.max().item() <= 1e-2 * lr and dw.double().mean().item() <= 1e-5 * lr, (dw.max().item(), dw.double().mean().item())
    # the sampler returns the same transitions, unpacked
    s32, spk = r32.sample(512, 3, 1), rpk.sample(512, 3, 1)
    for k in ("states", "next_states", "actions", "rewards", "dones", "valid"):
        assert torch.equal(s32[k], spk[k]), k
    assert spk["packed_states"].shape == (512, 20)
    e32.close()
    epk.close()


def test_ragged_and_multi_uav_packed():
    (e32, r32), (epk, rpk) = _pair(37, 4, uav=4)          # 148 agents: ragged last tile everywhere
    gen = torch.Generator(device="cuda").manual_seed(3)
    for t in range(40):
        a = torch.randint(0, 3, (e32.N,), generator=gen, device="cuda", dtype=torch.int32)
        r32.current_action().copy_(a)
        rpk.current_action().copy_(a)
        r32.step_env(auto_reset=True)
        rpk.step_env(auto_reset=True)
        assert torch.equal(epk.unpack(rpk.current_obs()), r32.current_obs()), t
    e32.close()
    epk.close()
