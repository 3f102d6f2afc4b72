"""Transition records (ABI 5: UavReplayRing.meta -- {a1, a0, reward, done | valid << 8 | info << 16} per (frame, agent), written by the
step kernels next to the action / reward / done / valid planes): every step path writes records that agree with the planes bit for
bit, and a learner that gathers the records takes exactly the update it takes from the planes."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu

PARAM = {"w": "100", "hiden_dim": "64", "output": "3", "LEARNING_RATE": "0.001", "gamma": "0.99", "Update_loop": "3"}


def _check(ring, frames, info=None, a1=None):
    m = ring.meta[frames]
    act = ring.action[frames]
    assert torch.equal(m[..., 1], act if act.dtype == torch.int32 else act.view(torch.int32))
    assert torch.equal(m[..., 2], ring.reward[frames].view(torch.int32))
    assert torch.equal(m[..., 3] & 0xff, ring.done[frames].int())
    assert torch.equal((m[..., 3] >> 8) & 0xff, ring.valid[frames].int())
    if info is not None:
        assert torch.equal((m[..., 3] >> 16) & 0xff, info[frames].int())
    assert torch.equal(m[..., 0], torch.zeros_like(m[..., 0]) if a1 is None else a1[frames].view(torch.int32))


@pytest.mark.parametrize("n,dtype,one_wave", [(1000, "packed", False), (1000, torch.float32, True), (70000, torch.float16, False)])
def test_every_step_kernel_writes_the_records(n, dtype, one_wave):
    from dqn_based_uav_3d_path_planer_amd import _lib
    from dqn_based_uav_3d_path_planer_amd.data import make_city26_env
    from dqn_based_uav_3d_path_planer_amd.replay import DeviceReplayRing
    env = make_city26_env(n, obs_dtype=dtype, uav_per_env=2 if n == 1000 else 1)
    ring = DeviceReplayRing(env, 40 * env.N, discrete=True)
    ring.reset(seed=3)
    if one_wave:
        ring.extra_flags = _lib.STEP_ONE_WAVE
    info = torch.zeros((ring.frames, env.N), dtype=torch.uint8, device="cuda")
    gen = torch.Generator(device="cuda").manual_seed(1)
    for t in range(30):
        ring.current_action().copy_(torch.randint(0, 3, (env.N,), generator=gen, device="cuda", dtype=torch.int32))
        ring.step_env(auto_reset=(t % 2 == 0), info=info)          # (without auto-reset finished agents are skipped: valid = 0 rows)
    torch.cuda.synchronize()
    _check(ring, slice(0, 30), info=info)
    # a step issued around the ring (env.step) leaves the records alone: the request is for ONE launch
    before = ring.meta.clone()
    out = env.alloc_out()
    env.step(torch.zeros(env.N, dtype=torch.int32, device="cuda"), out, auto_reset=True)
    torch.cuda.synchronize()
    assert torch.equal(before, ring.meta)
    env.close()


def test_policy_step_c_loop_and_the_learner_on_records():
    """k_step_coop<policy> (Python-issued and from the C loop) writes the chosen action into the record; and a learner that is
    handed the SAME ring without its records (meta = NULL: the four plane gathers) takes the identical update."""
    from dqn_based_uav_3d_path_planer_amd.data import make_city26_env
    from dqn_based_uav_3d_path_planer_amd.learner import FusedDQNLearner
    from dqn_based_uav_3d_path_planer_amd.loop import HotLoop
    from dqn_based_uav_3d_path_planer_amd.replay import DeviceReplayRing
    n = 2048
    env = make_city26_env(n, obs_dtype="packed")
    ring = DeviceReplayRing(env, 24 * n, discrete=True)
    ring.reset(seed=5)
    torch.manual_seed(2)
    L = FusedDQNLearner(dict(PARAM, NetWork="Qnet2"), "dqn", device="cuda:0")
    for t in range(6):
        assert ring.step_policy(L, 0.3, 11, t, auto_reset=True)
    hot = HotLoop(ring, L, 1024, seed=11, eps=0.3, counter=6)
    hot.run(10)
    torch.cuda.synchronize()
    _check(ring, slice(0, 16))
    hot.close()
    # the same update from the records and from the planes
    A = FusedDQNLearner(dict(PARAM, NetWork="Qnet2"), "dqn", device="cuda:0")
    B = FusedDQNLearner(dict(PARAM, NetWork="Qnet2"), "dqn", device="cuda:0")
    A.flat.copy_(L.flat)
    B.flat.copy_(L.flat)
    meta = ring._c.meta
    for k in range(3):
        ring._c.meta = meta
        la = float(A.learn_from_ring(ring, 1024, 4, k))
        ring._c.meta = None
        lb = float(B.learn_from_ring(ring, 1024, 4, k))
        assert la == lb, (k, la, lb)
    ring._c.meta = meta
    assert torch.equal(A.flat, B.flat)
    env.close()


def test_sac_records_carry_both_action_components():
    from dqn_based_uav_3d_path_planer_amd.data import make_city26_env
    from dqn_based_uav_3d_path_planer_amd.replay import DeviceReplayRing
    env = make_city26_env(300, uav_per_env=4, obs_dtype="packed")
    ring = DeviceReplayRing(env, 12 * env.N, discrete=False)
    ring.reset(seed=8)
    a1 = torch.zeros((ring.frames, env.N), dtype=torch.float32, device="cuda")
    ring.attach_action1(a1)
    gen = torch.Generator(device="cuda").manual_seed(4)
    for t in range(10):
        ring.current_action().copy_(torch.rand(env.N, generator=gen, device="cuda") * 2 - 1)
        a1[ring.head].copy_(torch.rand(env.N, generator=gen, device="cuda") * 2 - 1)
        ring.step_env(auto_reset=True)
    torch.cuda.synchronize()
    _check(ring, slice(0, 10), a1=a1)
    env.close()


def test_the_f16_policy_step_kernel_writes_the_records():
    """k_step_polh (f16 rows, the f16-MFMA policy inside the one-wave step kernel, 49 152 < agents <= 65 536)."""
    from dqn_based_uav_3d_path_planer_amd.data import make_city26_env
    from dqn_based_uav_3d_path_planer_amd.learner import FusedDQNLearner
    from dqn_based_uav_3d_path_planer_amd.replay import DeviceReplayRing
    n = 50048
    env = make_city26_env(n, obs_dtype=torch.float16)
    ring = DeviceReplayRing(env, 6 * n, discrete=True)
    ring.reset(seed=6)
    torch.manual_seed(3)
    L = FusedDQNLearner(dict(PARAM, NetWork="VAnet2"), "dueling", device="cuda:0", mfma="f16")
    for t in range(5):
        assert ring.step_policy(L, 0.3, 13, t, auto_reset=True)
    torch.cuda.synchronize()
    _check(ring, slice(0, 5))
    env.close()


@pytest.mark.parametrize("discrete", [True, False])
def test_rewrite_records_follows_hand_edited_planes(discrete):
    """DeviceReplayRing.rewrite_records: the fused learners read the RECORDS, so code that edits an action / reward / done /
    valid plane by hand rebuilds them from the planes afterwards -- for the given frames only, the info byte (which has no plane in the
    ring) kept, a second action component taken from the attached plane."""
    from dqn_based_uav_3d_path_planer_amd.data import make_city26_env
    from dqn_based_uav_3d_path_planer_amd.replay import DeviceReplayRing
    n = 512
    env = make_city26_env(n, obs_dtype="packed")
    ring = DeviceReplayRing(env, 12 * n, discrete=discrete)
    ring.reset(seed=9)
    a1 = None
    if not discrete:
        a1 = torch.rand((ring.frames, n), device="cuda") * 2 - 1
        ring.attach_action1(a1)
    info = torch.zeros((ring.frames, n), dtype=torch.uint8, device="cuda")
    gen = torch.Generator(device="cuda").manual_seed(2)
    for _ in range(8):
        if discrete:
            ring.current_action().copy_(torch.randint(0, 3, (n,), generator=gen, device="cuda", dtype=torch.int32))
        else:
            ring.current_action().copy_(torch.rand(n, generator=gen, device="cuda") * 2 - 1)
        ring.step_env(auto_reset=True, info=info)
    torch.cuda.synchronize()
    _check(ring, slice(0, 8), info=info, a1=a1)
    before = ring.meta.clone()
    # hand-edit frames 2 and 5: rewards, dones, a block of actions, a few valid bytes
    ring.reward[2] += 1.5
    ring.done[5] ^= 1
    ring.valid[5, :7] = 0
    if discrete:
        ring.action[2, :64] = 2
    else:
        ring.action[2, :64] = 0.25
    assert not torch.equal(ring.meta[2][..., 2], ring.reward[2].view(torch.int32))          # the records are stale now
    ring.rewrite_records([2, 5])
    torch.cuda.synchronize()
    _check(ring, slice(0, 8), info=info, a1=a1)                                             # planes == records again, info kept
    untouched = [f for f in range(ring.frames) if f not in (2, 5)]
    assert torch.equal(ring.meta[untouched], before[untouched])
    ring.reward[:] = 0.0
    ring.rewrite_records()                                                                  # all frames
    assert int(ring.meta[..., 2].abs().max()) == 0
    env.close()
