// coll.hip -- the gradient bucket's all-reduce as an RCCL collective enqueued FROM C on the loop's stream.
//
// csrc/p2p.hip is the fast exchange (one hop over peer-mapped HBM).  When it cannot be set up -- no HIP IPC between the
// ranks' devices, a container that forbids it -- or after it raised its sticky error, the exchange falls back to
// ncclAllReduce(sum) on the same ~26 KB bucket.  Calling it through torch.distributed would take the loop back to
// Python (one host round trip per update: the pass becomes host-bound); here the collective is one more enqueue between
// uavenv_dqn_reduce and uavenv_dqn_adam inside uavenv_loop_run, so csrc/loop.hip keeps driving the pass at any world
// size.  RCCL is reached through dlopen (no link-time dependency: the library loads without it, and the process keeps
// the single RCCL instance PyTorch already mapped when the caller hands over that path).
//
// Rendezvous: rank 0 calls uavenv_coll_unique_id, the 128 bytes travel to the other ranks by whatever the caller has
// (bench.py / learner.py: a torch.distributed broadcast), every rank calls uavenv_coll_create.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <new>

#include <rccl/rccl.h>

#include "../../include/uavenv.h"

namespace {

struct RcclApi {
    void *so = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

RcclApi g_api;
thread_local char g_err[256] = "";

// dlopen once: the caller's path first (PyTorch's own librccl.so), then whatever the loader finds
int load_api(const char *path)
{
    if (g_api.so) return UAVENV_OK;
    const char *cands[] = {path, "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void *so = nullptr;
    for (const char *c : cands) {
        if (!c || !*c) continue;
        so = dlopen(c, RTLD_NOW | RTLD_GLOBAL);
        if (so) break;
    }
    if (!so) {
        snprintf(g_err, sizeof(g_err), "librccl.so not found: %s", dlerror());
        return UAVENV_ENODEV;
    }
    RcclApi a;
    a.so = so;
    a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(so, "ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))dlsym(so, "ncclCommInitRank");
    a.AllReduce = (decltype(a.AllReduce))dlsym(so, "ncclAllReduce");
    a.CommDestroy = (decltype(a.CommDestroy))dlsym(so, "ncclCommDestroy");
    a.GetErrorString = (decltype(a.GetErrorString))dlsym(so, "ncclGetErrorString");
    if (!a.GetUniqueId || !a.CommInitRank || !a.AllReduce || !a.CommDestroy) {
        snprintf(g_err, sizeof(g_err), "librccl.so lacks ncclGetUniqueId / ncclCommInitRank / ncclAllReduce / ncclCommDestroy");
        return UAVENV_ENODEV;
    }
    g_api = a;
    return UAVENV_OK;
}

int fail(const char *what, ncclResult_t r)
{
    snprintf(g_err, sizeof(g_err), "%s: %s", what, g_api.GetErrorString ? g_api.GetErrorString(r) : "rccl error");
    return UAVENV_EHIP;
}

}  // namespace

struct UavColl {
    ncclComm_t comm = nullptr;
    int world = 0, rank = 0;
};

extern "C" {

const char *uavenv_coll_last_error(void) { return g_err; }

int uavenv_coll_unique_id(const char *rccl_path, void *id_out_host)
{
    if (!id_out_host) return UAVENV_EINVAL;
    const int rc = load_api(rccl_path);
    if (rc != UAVENV_OK) return rc;
    static_assert(sizeof(ncclUniqueId) == UAVENV_COLL_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId id;
    const ncclResult_t r = g_api.GetUniqueId(&id);
    if (r != ncclSuccess) return fail("ncclGetUniqueId", r);
    memcpy(id_out_host, &id, sizeof(id));
    return UAVENV_OK;
}

int uavenv_coll_create(const char *rccl_path, int32_t world, int32_t rank, const void *id_host, UavColl **out)
{
    if (!out || !id_host || world < 1 || rank < 0 || rank >= world) return UAVENV_EINVAL;
    const int rc = load_api(rccl_path);
    if (rc != UAVENV_OK) return rc;
    UavColl *c = new (std::nothrow) UavColl();
    if (!c) return UAVENV_ENOMEM;
    ncclUniqueId id;
    memcpy(&id, id_host, sizeof(id));
    const ncclResult_t r = g_api.CommInitRank(&c->comm, world, id, rank);      // the current HIP device is this rank's GPU
    if (r != ncclSuccess) {
        delete c;
        return fail("ncclCommInitRank", r);
    }
    c->world = world;
    c->rank = rank;
    *out = c;
    return UAVENV_OK;
}

int uavenv_coll_destroy(UavColl *c)
{
    if (!c) return UAVENV_OK;
    if (c->comm) (void)g_api.CommDestroy(c->comm);
    delete c;
    return UAVENV_OK;
}

int uavenv_coll_allreduce_sum(UavColl *c, float *buf_dev, int64_t n, void *stream)
{
    if (!c || !c->comm || !buf_dev || n <= 0) return UAVENV_EINVAL;
    const ncclResult_t r = g_api.AllReduce(buf_dev, buf_dev, (size_t)n, ncclFloat32, ncclSum, c->comm, (hipStream_t)stream);
    return r == ncclSuccess ? UAVENV_OK : fail("ncclAllReduce", r);
}

}  // extern "C"
