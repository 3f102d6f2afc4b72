"""Set-up of the on-stream gradient exchanges between the ranks of an initialised torch.distributed group (one process per
GPU of ONE node): the peer-mapped receive areas of csrc/p2p.hip and the RCCL communicator of csrc/coll.hip.  Both are
verified against torch.distributed's own all-reduce before they are handed out; any failure on any rank returns None on
EVERY rank (the callers then keep the torch.distributed path).  The learners' bucket-specific protocols (FusedDQNLearner:
uavenv_dqn_reduce_p2p / uavenv_dqn_adam_p2p) and the plain buffer form (FusedSACLearner, the SAC loop: uavenv_p2p_allreduce)
run on the handles made here."""
import ctypes as C

import torch
import torch.distributed as dist

from . import _lib


def _group():
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return None
    return dist.get_world_size(), dist.get_rank()


def _all(ok: bool, world: int) -> bool:
    flags = [None] * world
    dist.all_gather_object(flags, bool(ok))
    return all(flags)


def open_p2p(lib, device: torch.device, bucket_floats: int, check_every: int = 256, spin_limit: int = 0):
    """-> a connected UavP2P handle (c_void_p) whose receive slots hold bucket_floats floats, or None.  Nothing is mapped
    unless this device may address every peer's memory (xGMI / PCIe peer access)."""
    g = _group()
    if g is None:
        return None
    world, rank = g
    ok, h = True, C.c_void_p()
    try:
        _lib.check(lib.uavenv_p2p_create(world, rank, int(bucket_floats), C.byref(h)), "uavenv_p2p_create")
        _lib.check(lib.uavenv_p2p_configure(h, int(check_every), int(spin_limit)), "uavenv_p2p_configure")
        mine = (C.c_ubyte * _lib.P2P_HANDLE_BYTES)()
        _lib.check(lib.uavenv_p2p_handle(h, mine), "uavenv_p2p_handle")
    except Exception:
        ok, mine = False, (C.c_ubyte * _lib.P2P_HANDLE_BYTES)()
    allh = [None] * world
    my_dev = device.index if device.index is not None else (torch.cuda.current_device() if torch.cuda.is_available() else -1)
    dist.all_gather_object(allh, (ok, bytes(mine), my_dev))
    ok = all(o for o, _, _ in allh)
    if ok:
        ok = all(lib.uavenv_p2p_can_reach(my_dev, d) == 1 for _, _, d in allh)
    ok = _all(ok, world)
    if ok:
        blob = b"".join(b for _, b, _ in allh)
        ok = lib.uavenv_p2p_connect(h, C.create_string_buffer(blob, len(blob))) == 0
    ok = _all(ok, world)
    if not ok:
        if h.value:
            lib.uavenv_p2p_destroy(h)
        return None
    return h


def verify_p2p_allreduce(lib, h, device: torch.device, n: int, stream, iters: int = 4) -> bool:
    """`iters` uavenv_p2p_allreduce calls back to back (both receive slots, twice each) of a payload that is random per rank and per
    iteration, each compared on every rank with the sum of ALL ranks' payloads -- regenerated locally from seeds every rank knows and
    added in rank order, as the pull kernel adds the slots (no second transport: a gloo all-reduce of a device tensor per iteration
    left an 8-rank same-device run 100 x slower for the rest of the process's life).  selftest_ms[0] = the wall time of the last call."""
    import time
    world, rank = dist.get_world_size(), dist.get_rank()
    n = max(4, (int(n) // 4) * 4)
    gens = [torch.Generator(device="cpu").manual_seed(0x51AC + 7919 * r) for r in range(world)]
    ok = True
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(max(1, iters)):
        payloads = [torch.rand(n, generator=g) * 2.0 - 1.0 for g in gens]
        want = payloads[0].clone()
        for r in range(1, world):
            want += payloads[r]
        t = payloads[rank].to(device)
        rc = lib.uavenv_p2p_allreduce(h, t.data_ptr(), n, stream)
        torch.cuda.synchronize(device)
        err = C.c_int32(0)
        lib.uavenv_p2p_errors(h, C.byref(err))
        ok = ok and rc == 0 and err.value == 0 and bool(torch.allclose(t, want.to(device), rtol=1e-6, atol=1e-6))
    selftest_ms[0] = (time.perf_counter() - t0) * 1e3
    return _all(ok, world)


selftest_ms = [None]


def open_coll(lib, device: torch.device, verify_floats: int, stream):
    """-> a UavColl handle (RCCL communicator driven from C, csrc/coll.hip), verified, or None."""
    g = _group()
    if g is None:
        return None
    world, rank = g
    path = _lib.rccl_path()
    buf = (C.c_ubyte * _lib.COLL_ID_BYTES)()
    # every rank first shows that it can load RCCL at all (dlopen + the symbols + one ncclGetUniqueId, thrown away): a rank that
    # cannot would never enter ncclCommInitRank below and its peers would block inside it -- vote BEFORE anyone does
    probe = (C.c_ubyte * _lib.COLL_ID_BYTES)()
    if not _all(bool(path) and torch.cuda.is_available() and lib.uavenv_coll_unique_id(path, probe) == 0, world):
        return None
    ok = True
    if rank == 0:
        ok = lib.uavenv_coll_unique_id(path, buf) == 0
    box = [(ok, bytes(buf))]
    dist.broadcast_object_list(box, src=0)
    ok, idb = box[0]
    h = C.c_void_p()
    if ok and not torch.cuda.is_available():
        ok = False
    if ok:
        torch.cuda.set_device(device)
        ok = lib.uavenv_coll_create(path, world, rank, C.create_string_buffer(idb, len(idb)), C.byref(h)) == 0
    if not _all(ok, world):
        if h.value:
            lib.uavenv_coll_destroy(h)
        return None
    n = int(verify_floats)
    t = torch.arange(n, device=device, dtype=torch.float32) * 1e-3 + (rank + 1)
    want = t.clone()
    dist.all_reduce(want, op=dist.ReduceOp.SUM)
    rc = lib.uavenv_coll_allreduce_sum(h, t.data_ptr(), n, stream)
    torch.cuda.synchronize(device)
    if not _all(rc == 0 and bool(torch.allclose(t, want, rtol=1e-6, atol=1e-6)), world):
        lib.uavenv_coll_destroy(h)
        return None
    return h
