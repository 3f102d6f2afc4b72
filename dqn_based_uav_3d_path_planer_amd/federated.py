"""The federated merge of the per-UAV trainers that PathPlan_City.run_eposide fires every FL_Loop episodes when Is_FL is set
(Envs/PathPlan_City.py:469-475): Federated_Learning_AC (:590-601) for actor-critic trainers -- the actors of all UAVs are
added up in agent order into a deep copy of agent 0's and every UAV takes the result through replace_param
(Trainer/SAC_Trainer.py:456-459; optimizers, critics and targets stay as they are).

What the reference EXECUTES differs from its comment: the division by len(Agents) at :597 assigns into the dict state_dict()
returned and never reaches the model, so the merged model is the SUM of the actors (tests/golden/federated_ac.npz, generated
by oracle/gen_golden_federated.py, pins that).  `aggregate`:
  "reference"  the sum, as executed -- the drop-in default (<FL_Aggregate> absent);
  "mean"       the sum / number of UAVs -- what the comment at :597 intends and what keeps training stable.

For DQN-family trainers (Is_AC = 0) the reference calls Federated_Learning (:604-640), whose get_policy_DFRL / SPN_param /
Update_SPN_Soft exist on no trainer in the tree (it raises AttributeError): there is no executed behaviour to match, so the
same merge is applied to q_local (replace_param of Trainer/DuelingDQN_Trainer.py:204-207 writes q_local only).

Fused trainers (flat f32 parameter blocks in HBM) merge in ONE launch (csrc/fed.hip: uavenv_fed_aggregate, summation in the
reference's agent order -> bit-identical to torch's f32 adds); anything else merges through torch on whatever device it is."""
from __future__ import annotations

import ctypes as C
from typing import Sequence

import torch

AGGREGATES = ("reference", "mean")


def _scale(aggregate: str, n: int) -> float:
    if aggregate not in AGGREGATES:
        raise ValueError(f"FL_Aggregate {aggregate!r}: expected one of {AGGREGATES}")
    return 1.0 if aggregate == "reference" else 1.0 / n


def merge_blocks(blocks: Sequence[torch.Tensor], scale: float, n_floats: int = None) -> None:
    """Every flat f32 block <- scale * (blocks[0] + blocks[1] + ...) on the device, one launch (uavenv_fed_aggregate).
    Fails loudly without the HIP library / a GPU: there is no CPU path here (merge_modules is the torch form)."""
    from . import _lib
    lib = _lib.load()
    n = len(blocks)
    if not 1 <= n <= _lib.FED_MAX_BLOCKS:
        raise ValueError(f"{n} blocks: uavenv_fed_aggregate takes 1..{_lib.FED_MAX_BLOCKS}")
    n_floats = int(blocks[0].numel() if n_floats is None else n_floats)
    for b in blocks:
        if not (b.is_cuda and b.dtype == torch.float32 and b.is_contiguous() and b.numel() >= n_floats and b.device == blocks[0].device):
            raise ValueError("blocks must be contiguous f32 CUDA tensors of one device")
    arr = (C.c_void_p * n)(*[b.data_ptr() for b in blocks])
    _lib.check(lib.uavenv_fed_aggregate(arr, n, n_floats, float(scale), torch.cuda.current_stream(blocks[0].device).cuda_stream),
               "uavenv_fed_aggregate")


def merge_modules(modules: Sequence[torch.nn.Module], scale: float) -> None:
    """The same merge through torch, parameter by parameter in agent order (:593-596), written back with copy_ (:458-459)."""
    with torch.no_grad():
        for ps in zip(*[list(m.parameters()) for m in modules]):
            total = ps[0].detach().clone()
            for p in ps[1:]:
                total += p.detach().to(total.device)
            if scale != 1.0:
                total *= scale
            for p in ps:
                p.copy_(total.to(p.device))


def federated_learning_ac(trainers: Sequence, aggregate: str = "reference") -> str:
    """Federated_Learning_AC over `trainers` (objects with .actor, optionally .learner with the fused flat blocks).
    -> "device" when the one-launch kernel ran, "torch" otherwise."""
    scale = _scale(aggregate, len(trainers))
    learners = [getattr(t, "learner", None) for t in trainers]
    if len(trainers) <= 8 and all(getattr(t, "fused", False) and hasattr(L, "_blocks") for t, L in zip(trainers, learners)) \
            and len({L._blocks.device for L in learners}) == 1 and len({id(L) for L in learners}) == len(learners):
        from . import _lib
        merge_blocks([L._blocks[0] for L in learners], scale, n_floats=_lib.SAC_ACTOR_PARAMS)
        return "device"
    merge_modules([t.actor for t in trainers], scale)
    return "torch"


def federated_learning_q(trainers: Sequence, aggregate: str = "reference") -> str:
    """The merge applied to q_local of DQN-family trainers (see the module docstring)."""
    scale = _scale(aggregate, len(trainers))
    learners = [getattr(t, "learner", None) for t in trainers]
    if len(trainers) <= 8 and all(getattr(t, "fused", False) and hasattr(L, "flat") and hasattr(L, "P") for t, L in zip(trainers, learners)) \
            and len({L.flat.device for L in learners}) == 1 and len({L.P for L in learners}) == 1 \
            and len({id(L) for L in learners}) == len(learners):
        merge_blocks([L.flat[0] for L in learners], scale, n_floats=learners[0].P)
        return "device"
    merge_modules([t.q_local for t in trainers], scale)
    return "torch"
