"""The fused HIP learner (csrc/learner.hip: f32-MFMA forward/backward, reduce, Adam) against
  * the EXECUTED reference trainers (tests/golden/learner_*.npz, oracle/gen_golden_learner.py), and
  * the PyTorch-ROCm learner (learner.DQNLearner) on batches drawn from a real device replay ring.
Tolerance: f32 arithmetic with different summation order: 2e-5 relative on losses, 5e-6 absolute on weights after
7 updates (the same bars the torch learner meets against the reference on CPU)."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu

CASES = [("DQN_Trainer", "dqn", "Qnet2"), ("DDQN_Trainer", "ddqn", "Qnet2"), ("DuelingDQN_Trainer", "dueling", "VAnet2")]
PARAM = {"w": "100", "hiden_dim": "64", "output": "3", "LEARNING_RATE": "0.001", "gamma": "0.99", "Update_loop": "3"}


class HandRing:
    """A 2-frame ring holding exactly one hand-made batch: frame 0 = states, frame 1 = next states."""

    def __init__(self, states, next_states, actions, rewards, dones):
        import ctypes as C
        from dqn_based_uav_3d_path_planer_amd import _lib
        n = len(actions)
        dev = "cuda"
        self.obs = torch.stack([torch.tensor(states), torch.tensor(next_states)]).to(dev).contiguous()
        z = lambda x, dt: torch.stack([torch.tensor(x).to(dt), torch.zeros(n, dtype=dt)]).to(dev).contiguous()  # noqa
        self.action, self.reward = z(actions.astype(np.int32), torch.int32), z(rewards, torch.float32)
        self.done, self.valid = z(dones.astype(np.uint8), torch.uint8), torch.ones((2, n), dtype=torch.uint8, device=dev)
        self.head, self.filled = 1, 1
        self._c = _lib.UavReplayRing(self.obs.data_ptr(), self.action.data_ptr(), self.reward.data_ptr(),
                                     self.done.data_ptr(), self.valid.data_ptr(), 2, n, _lib.OBS_F32, 1)
        idx = np.stack([np.zeros(n, dtype=np.int32), np.arange(n, dtype=np.int32)], axis=1)
        self.idx = torch.tensor(idx, device=dev).contiguous()


def _load(net, g, pref):
    net.load_state_dict({k[len(pref):]: torch.tensor(v) for k, v in g.items() if k.startswith(pref)})


@pytest.mark.parametrize("ref_name,kind,net", CASES)
def test_fused_updates_match_reference(ref_name, kind, net):
    from dqn_based_uav_3d_path_planer_amd.learner import FusedDQNLearner
    g = load_golden(f"learner_{ref_name}.npz")
    L = FusedDQNLearner(dict(PARAM, NetWork=net), kind, device="cuda:0")
    _load(L.q_local, g, "l0_")
    _load(L.q_target, g, "t0_")
    ring = HandRing(g["states"], g["next_states"], g["actions"], g["rewards"], g["dones"])
    losses = []
    for _ in range(len(g["losses"])):
        losses.append(float(L.learn_from_ring(ring, 64, 0, 0, explicit_idx=ring.idx)))
    assert L.epoch == int(g["epoch"])
    assert np.allclose(losses, g["losses"], rtol=2e-5, atol=0), (losses, g["losses"])
    for pref, netobj in (("l1_", L.q_local), ("t1_", L.q_target)):
        for k, v in netobj.state_dict().items():
            assert np.abs(v.cpu().numpy() - g[pref + k]).max() <= 5e-6, (pref, k)


@pytest.mark.parametrize("kind,net", [("dqn", "Qnet2"), ("ddqn", "Qnet2"), ("dueling", "VAnet2")])
def test_fused_matches_torch_learner_on_a_real_ring(kind, net):
    from dqn_based_uav_3d_path_planer_amd.data import make_city26_env
    from dqn_based_uav_3d_path_planer_amd.learner import DQNLearner, FusedDQNLearner
    from dqn_based_uav_3d_path_planer_amd.replay import DeviceReplayRing
    n = 2048
    env = make_city26_env(n)
    ring = DeviceReplayRing(env, 6 * n, discrete=True)
    ring.reset(seed=4)
    gen = torch.Generator(device="cuda").manual_seed(1)
    for _ in range(5):
        ring.current_action().copy_(torch.randint(0, 3, (n,), generator=gen, device="cuda", dtype=torch.int32))
        ring.step_env(auto_reset=True)
    torch.manual_seed(0)
    T = DQNLearner(dict(PARAM, NetWork=net), kind, device="cuda:0")
    F = FusedDQNLearner(dict(PARAM, NetWork=net), kind, device="cuda:0")
    F.q_local.load_state_dict(T.q_local.state_dict())
    F.q_target.load_state_dict(T.q_target.state_dict())
    F2 = FusedDQNLearner(dict(PARAM, NetWork=net), kind, device="cuda:0")      # the split (multi-GPU) kernel path
    F2.q_local.load_state_dict(T.q_local.state_dict())
    F2.q_target.load_state_dict(T.q_target.state_dict())
    F2.force_split = True
    B = 4096
    for it in range(4):
        batch = ring.sample(B, seed=11, counter=it)
        lt = float(T.learn(batch))
        lf = float(F.learn_from_ring(ring, B, seed=11, counter=it))
        lf2 = float(F2.learn_from_ring(ring, B, seed=11, counter=it))
        assert abs(lt - lf) <= 2e-5 * abs(lt), (it, lt, lf)
        assert abs(lf2 - lf) <= 1e-6 * abs(lf), (it, lf, lf2)
    assert (F.flat[:2] - F2.flat[:2]).abs().max().item() <= 1e-6
    for (k, a), (_, b) in zip(T.q_local.state_dict().items(), F.q_local.state_dict().items()):
        assert (a - b).abs().max().item() <= 2e-5, k
    for (k, a), (_, b) in zip(T.q_target.state_dict().items(), F.q_target.state_dict().items()):
        assert (a - b).abs().max().item() <= 2e-5, k
    env.close()


def test_fused_act_matches_torch_forward():
    from dqn_based_uav_3d_path_planer_amd.learner import FusedDQNLearner
    for net, kind in (("Qnet2", "dqn"), ("VAnet2", "dueling")):
        F = FusedDQNLearner(dict(PARAM, NetWork=net), kind, device="cuda:0")
        n = 5000          # not a multiple of 64: ragged last tile
        obs = torch.randn(n, 100, device="cuda")
        q = torch.empty(n, 3, device="cuda")
        idx = torch.empty(n, dtype=torch.int32, device="cuda")
        steer = torch.empty(n, device="cuda")
        F.act(obs, 0.0, 3, 0, index_out=idx, steer_out=steer, q_out=q)
        with torch.no_grad():
            ref = F.q_local(obs)
        assert (q - ref).abs().max().item() <= 2e-5
        clear = (ref.topk(2, dim=1).values[:, 0] - ref.topk(2, dim=1).values[:, 1]) > 1e-4
        assert torch.equal(idx[clear].long(), ref.argmax(1)[clear])
        assert torch.allclose(steer, idx.float() - 1.0)
        F.act(obs, 1.0, 3, 1, index_out=idx)          # eps = 1: uniform random
        counts = torch.bincount(idx.long(), minlength=3).float() / n
        assert (counts - 1 / 3).abs().max().item() < 0.03
        F.act(obs.half(), 0.0, 3, 0, index_out=idx, q_out=q)   # f16 observations
        assert (q - F.q_local(obs.half().float())).abs().max().item() <= 2e-5
