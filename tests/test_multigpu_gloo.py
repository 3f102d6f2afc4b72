"""N > 1 path on CPU: gloo ranks (world sizes 2 and 4), each with its share of a batch, must apply exactly the update one process
applies on the concatenated batch (learner._allreduce_grads: flat-bucket all-reduce of the gradients before Adam.step)."""
import pytest
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import load_golden

PARAM = {"NetWork": "Qnet2", "w": "100", "hiden_dim": "64", "output": "3", "LEARNING_RATE": "0.001",
         "gamma": "0.99", "Update_loop": "3"}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, kind, net, out_dir):
    from dqn_based_uav_3d_path_planer_amd.learner import DQNLearner
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    g = load_golden("learner_DQN_Trainer.npz" if net == "Qnet2" else "learner_DuelingDQN_Trainer.npz")
    L = DQNLearner(dict(PARAM, NetWork=net), kind, device="cpu")
    L.q_local.load_state_dict({k[3:]: torch.tensor(v) for k, v in g.items() if k.startswith("l0_")})
    L.q_target.load_state_dict({k[3:]: torch.tensor(v) for k, v in g.items() if k.startswith("t0_")})
    n = len(g["actions"]) // world
    sl = slice(rank * n, (rank + 1) * n)
    batch = dict(states=torch.tensor(g["states"][sl]), next_states=torch.tensor(g["next_states"][sl]),
                 actions=torch.tensor(g["actions"][sl].astype(np.int32)), rewards=torch.tensor(g["rewards"][sl]),
                 dones=torch.tensor(g["dones"][sl]))
    for _ in range(4):
        L.learn(batch)
    torch.save({k: v.clone() for k, v in L.q_local.state_dict().items()}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def _single(kind, net):
    from dqn_based_uav_3d_path_planer_amd.learner import DQNLearner
    g = load_golden("learner_DQN_Trainer.npz" if net == "Qnet2" else "learner_DuelingDQN_Trainer.npz")
    L = DQNLearner(dict(PARAM, NetWork=net), kind, device="cpu")
    L.q_local.load_state_dict({k[3:]: torch.tensor(v) for k, v in g.items() if k.startswith("l0_")})
    L.q_target.load_state_dict({k[3:]: torch.tensor(v) for k, v in g.items() if k.startswith("t0_")})
    batch = dict(states=torch.tensor(g["states"]), next_states=torch.tensor(g["next_states"]),
                 actions=torch.tensor(g["actions"].astype(np.int32)), rewards=torch.tensor(g["rewards"]),
                 dones=torch.tensor(g["dones"]))
    for _ in range(4):
        L.learn(batch)
    return L.q_local.state_dict()


def _run(kind, net, tmp_path, world=2):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, kind, net, str(tmp_path)), nprocs=world, join=True)
    rs = [torch.load(os.path.join(tmp_path, f"rank{r}.pt")) for r in range(world)]
    ref = _single(kind, net)
    for k in ref:
        for r in rs[1:]:
            assert torch.equal(rs[0][k], r[k]), k                    # ranks stay in lock-step
        assert (rs[0][k] - ref[k]).abs().max().item() <= 2e-6, k     # == single process on the concatenated batch


@pytest.mark.parametrize("world", [2, 4])
def test_ranks_equal_one_process_dqn(tmp_path, world):
    _run("dqn", "Qnet2", tmp_path, world)


def test_two_ranks_equal_one_process_dueling(tmp_path):
    _run("dueling", "VAnet2", tmp_path)


def _fed_worker(rank, world, port, out_dir):
    from dqn_based_uav_3d_path_planer_amd.learner import DQNLearner
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    g = load_golden("learner_DQN_Trainer.npz")
    L = DQNLearner(dict(PARAM, FL_Loop="2"), "dqn", device="cpu")
    L.sync = "fedavg"
    L.q_local.load_state_dict({k[3:]: torch.tensor(v) for k, v in g.items() if k.startswith("l0_")})
    L.q_target.load_state_dict({k[3:]: torch.tensor(v) for k, v in g.items() if k.startswith("t0_")})
    n = len(g["actions"]) // world
    sl = slice(rank * n, (rank + 1) * n)
    batch = dict(states=torch.tensor(g["states"][sl]), next_states=torch.tensor(g["next_states"][sl]),
                 actions=torch.tensor(g["actions"][sl].astype(np.int32)), rewards=torch.tensor(g["rewards"][sl]),
                 dones=torch.tensor(g["dones"][sl]))
    snaps = []
    for _ in range(4):
        L.learn(batch)
        snaps.append({k: v.clone() for k, v in L.q_local.state_dict().items()})
    torch.save(snaps, os.path.join(out_dir, f"fed{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_federated_averaging_every_fl_loop_updates(tmp_path, world):
    """sync="fedavg": ranks train alone (different shards -> different weights after update 1) and hold the same,
    averaged weights after every FL_Loop-th update; the average is the mean of what each rank would have had."""
    from dqn_based_uav_3d_path_planer_amd.learner import DQNLearner
    port = _free_port()
    mp.spawn(_fed_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    snaps = [torch.load(tmp_path / f"fed{r}.pt") for r in range(world)]
    a = snaps[0]
    k = "fc1.weight"
    for b in snaps[1:]:
        assert not torch.equal(a[0][k], b[0][k])                       # update 1: no exchange
        assert all(torch.equal(a[1][q], b[1][q]) for q in a[1])        # update 2: averaged
        assert not torch.equal(a[2][k], b[2][k]) and all(torch.equal(a[3][q], b[3][q]) for q in a[3])
    # the averaged weights equal the mean of `world` single-process learners run on the shards for two updates
    g = load_golden("learner_DQN_Trainer.npz")
    outs = []
    for rank in range(world):
        L = DQNLearner(dict(PARAM), "dqn", device="cpu")
        L.q_local.load_state_dict({q[3:]: torch.tensor(v) for q, v in g.items() if q.startswith("l0_")})
        L.q_target.load_state_dict({q[3:]: torch.tensor(v) for q, v in g.items() if q.startswith("t0_")})
        n = len(g["actions"]) // world
        sl = slice(rank * n, (rank + 1) * n)
        batch = dict(states=torch.tensor(g["states"][sl]), next_states=torch.tensor(g["next_states"][sl]),
                     actions=torch.tensor(g["actions"][sl].astype(np.int32)), rewards=torch.tensor(g["rewards"][sl]),
                     dones=torch.tensor(g["dones"][sl]))
        L.learn(batch)
        L.learn(batch)
        outs.append(L.q_local.state_dict())
    for q in a[1]:
        assert torch.allclose(a[1][q], sum(o[q] for o in outs) / world, rtol=0, atol=2e-7)


def _exchange_worker(rank, world, port, out_dir):
    from dqn_based_uav_3d_path_planer_amd import _lib, exchange
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = _lib.load()
    h1 = exchange.open_p2p(lib, torch.device("cpu"), 1024)
    h2 = exchange.open_coll(lib, torch.device("cpu"), 64, None)
    with open(os.path.join(out_dir, f"ex{rank}.txt"), "w") as f:
        f.write(f"{h1 is None} {h2 is None}")
    dist.barrier()
    dist.destroy_process_group()


def test_exchange_setup_fails_closed_without_a_gpu(tmp_path):
    """exchange.open_p2p / open_coll (the set-up of csrc/p2p.hip and csrc/coll.hip between torch.distributed ranks) on a
    host without a HIP device: every rank gets None -- the callers then keep the torch.distributed path -- and nobody is left
    waiting in a collective of the handshake."""
    mp.spawn(_exchange_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    for r in (0, 1):
        assert open(os.path.join(tmp_path, f"ex{r}.txt")).read() == "True True"
