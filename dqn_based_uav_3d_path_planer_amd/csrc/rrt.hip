// rrt.hip -- the sub-goal planner of UAV.reset() on the GPU.
//
// Reference: Agents/UAV.py:353-360 (start / goal draws, Cal_SubTask) -> PathPlan/RRT.py:63-105 (RRTPlanner.getPath:
// 50 % goal bias, nearest node, steer by step 30 m, obstacle test every 5 m, one rewire pass over all nodes, stop when
// within 30 m of the goal, path = start .. goal).  The algorithm is sequential in its iterations, so ONE WAVEFRONT
// plans one scenario and spreads each iteration's inner loops over its 64 lanes:
//   nearest node     lanes stride the node list (LDS), wave arg-min with first-index tie break (python min());
//   obstacle_free    the <= 7 sample points of a segment are tested by 7 lanes at once through the same LDS broad
//                    phase + exact narrow phase as the env kernels (probe());
//   rewire           lanes test "d < step" for 64 nodes at a time, the surviving candidates are then visited in list
//                    order (the reference's sequential cost update).
// Thousands of scenarios plan concurrently (one per wavefront), which is what feeds the auto-reset of the env kernels
// without a host-side bank.
//
// Residency (round 4).  The node list lives in LDS.  Round 3 reserved 2 048 nodes (74 KB) per wavefront: ONE wavefront per CU,
// 256 plans in flight on the whole chip, every LDS / f64 latency of the sequential iteration exposed -- 29 ms for 16 384
// plans.  But the reference's trees are small (measured on 1 500 resets: iterations mean 118, median 84, p99 509, max 2 917;
// nodes <= iterations), so the launch is now TWO TIERS: tier A plans every scenario with room for kTierANodes nodes (21 KB of
// LDS with the world: seven wavefronts per CU); a scenario whose tree outgrows that is appended to a list and re-planned
// from scratch -- same stream of draws, so the same result -- by tier B with the full 2 048-node list.  Scenarios are
// handed out through an atomic counter (the iteration counts spread over two orders of magnitude: a static split left most
// wavefronts idle behind the longest plan).  Random numbers: a counter-based Philox stream per scenario, or -- for parity tests against
// the CPU oracle -- an explicit U[0,1) stream per scenario.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/uavenv.h"
#include "uavenv_device.hpp"

using namespace uav;

namespace {

struct RrtArgs {
    const unsigned char *world_blob;
    int32_t world_bytes, aux_off, grid_off, grid_stride, gn;
    double inv_cell, W, Hbox;        // Threaten_rate bounds
    double len, width, h;            // sampling box (RRT.py:26-32)
    int32_t m, K, max_iter, max_nodes;
    double step_size, obstacle_step;
    const double *start_goal_in;     // nullable [m][6]
    const double *uniforms;          // nullable [m][stream_len]
    int32_t stream_len;
    uint64_t seed;
    double *out_start_goal;          // [m][6]
    double *out_sub;                 // [m][K][3]
    int32_t *out_nsub;               // [m]  (<0: -needed when the path does not fit K; 1: planner gave up -> [goal])
    int32_t *out_iters;              // nullable [m]
    int32_t first;                   // scenario index offset of the Philox stream (row r plans scenario first + r)
    int32_t tier_b;                  // 0: tier A (every scenario); 1: tier B (plans the `flagged` list)
    int32_t defer;                   // tier A: a scenario whose tree outgrows max_nodes is appended to `flagged` instead of giving up
    int32_t *queue;                  // [0] next scenario of tier A, [1] number of flagged scenarios, [2] next entry of tier B
    int32_t *flagged;                // [m] scenarios whose tree outgrew tier A's node list
    unsigned char *node_scratch;     // LDS-free variant: scratch_wgs x kTierBNodes x (Node + parent) bytes of global memory
    int32_t scratch_wgs;
};
constexpr int kTierANodes = 320;
constexpr int kTierBNodes = 2048;

struct Node {
    double x, y, z, cost;
};

struct Stream {
    const double *ext;
    int ext_n, t;
    uint64_t seed;
    uint32_t scn, attempt;
    __device__ double next()
    {
        const int k = t++;
        if (ext) return k < ext_n ? ext[k] : 0.75;
        const uint4 r = philox4x32_10(make_uint4((uint32_t)k, scn, attempt, 0x7272u),
                                      make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
        return u53(r.x, r.y);
    }
    __device__ double uniform(double a, double b) { return a + (b - a) * next(); }    // random.uniform
};

// Loc.distance (BaseClass/CalMod.py:41-42)
__device__ __forceinline__ double ldist(double ax, double ay, double az, double bx, double by, double bz)
{
    const double dx = ax - bx, dy = ay - by, dz = az - bz;
    return sqrt(dx * dx + dy * dy + dz * dz);
}

// Squared distance: the argument ldist() hands to sqrt, bit for bit (same expression, -ffp-contract=off).
__device__ __forceinline__ double ldist2(double ax, double ay, double az, double bx, double by, double bz)
{
    const double dx = ax - bx, dy = ay - by, dz = az - bz;
    return dx * dx + dy * dy + dz * dz;
}
// Comparisons of sqrt values decided on their arguments wherever the arguments are clearly apart (round 4: the sqrt -- ~30 f64
// instructions on gfx950 -- was most of a node's cost in the nearest-node and rewire scans).  The device sqrt is within one ulp
// (2^-52 relative) of the true root, so arguments that differ by more than kBand = 2^-48 relative have roots ordered the same
// way; closer arguments -- practically only exact ties -- fall back to comparing the roots themselves.  The decisions are
// therefore EXACTLY those of the round-3 code that took the root of every distance.
constexpr double kBand = 1.0 / 281474976710656.0;     // 2^-48
// is sqrt(a2) < sqrt(b2) ?
__device__ __forceinline__ bool root_less(double a2, double b2)
{
    if (a2 < b2 * (1.0 - kBand)) return true;
    if (a2 > b2 * (1.0 + kBand)) return false;
    return sqrt(a2) < sqrt(b2);
}
// is sqrt(a2) < s ?   (s2 = s * s: its rounding, 2^-53 relative, is far inside the band)
__device__ __forceinline__ bool root_less_than(double a2, double s, double s2)
{
    if (a2 < s2 * (1.0 - kBand)) return true;
    if (a2 > s2 * (1.0 + kBand)) return false;
    return sqrt(a2) < s;
}
__device__ __forceinline__ bool root_equal(double a2, double b2)
{
    if (a2 < b2 * (1.0 - kBand) || a2 > b2 * (1.0 + kBand)) return false;
    return sqrt(a2) == sqrt(b2);
}

// RRT.obstacle_free (RRT.py:48-56): sample points a + (b-a)*i/(steps+1), i = 0..steps, one lane each.
template <typename MaskT>
__device__ __forceinline__ bool obstacle_free(const WorldLds<MaskT> &w, double ax, double ay, double az, double bx,
                                              double by, double bz, double obstacle_step)
{
    const int lane = (int)threadIdx.x & 63;
    const int steps = (int)(ldist(ax, ay, az, bx, by, bz) / obstacle_step);
    bool hit_any = false;
    for (int base = 0; base <= steps; base += 64) {
        const int i = base + lane;
        int hit = 0;
        if (i <= steps) {
            const double x = ax + (bx - ax) * (double)i / (double)(steps + 1);
            const double y = ay + (by - ay) * (double)i / (double)(steps + 1);
            const double z = az + (bz - az) * (double)i / (double)(steps + 1);
            hit = probe(w, x, y, z);
        }
        if (__ballot(hit) != 0ull) hit_any = true;
    }
    return !hit_any;
}

// IN_LDS = false: the BACKGROUND form (uavenv_replan_begin, beside a running loop): no LDS at all -- the world is read from its
// global blob (10 KB: it lives in the L2 / vector caches) and the node list sits in global scratch -- so that its wavefronts
// fit beside ANY resident kernel.  (The learner's gradient kernel takes 140 KB of a CU's 160 KB: a planner workgroup holding
// 21 KB of LDS on every CU would keep those workgroups waiting for milliseconds.)  Slower per plan, invisible to the passes.
template <typename MaskT, bool IN_LDS>
__global__ void __launch_bounds__(64) k_rrt_plan(RrtArgs a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = (int)threadIdx.x;
    WorldLds<MaskT> w;
    // the node list, structure of arrays: [x | y | z | cost] of max_nodes doubles each, then the parents.  (An array of 32-byte
    // structs made every scan read with a 32-byte lane stride: half the LDS banks idle; and one node per lane per round left each
    // round waiting out a full LDS round trip -- the scans below take four rounds' operands at once.)
    double *NX, *NY, *NZ, *NC;
    int *parent;
    if (IN_LDS) {   // stage the world (same blob as the env kernels)
        const uint4 *src = reinterpret_cast<const uint4 *>(a.world_blob);
        uint4 *dst = reinterpret_cast<uint4 *>(smem);
        for (int k = lane; k < a.world_bytes / 16; k += 64) dst[k] = src[k];
        __syncthreads();
        w.b = reinterpret_cast<const BldLds *>(smem);
        w.aux = reinterpret_cast<const BldAux *>(smem + a.aux_off);
        for (int h = 0; h < 3; ++h) w.g[h] = reinterpret_cast<const MaskT *>(smem + a.grid_off + h * a.grid_stride);
        const int node_off = (a.world_bytes + 15) & ~15;
        NX = reinterpret_cast<double *>(smem + node_off);
        parent = reinterpret_cast<int *>(smem + node_off + (size_t)a.max_nodes * sizeof(Node));
    } else {
        w.b = reinterpret_cast<const BldLds *>(a.world_blob);
        w.aux = reinterpret_cast<const BldAux *>(a.world_blob + a.aux_off);
        for (int h = 0; h < 3; ++h) w.g[h] = reinterpret_cast<const MaskT *>(a.world_blob + a.grid_off + h * a.grid_stride);
        unsigned char *mine = a.node_scratch + (size_t)blockIdx.x * (size_t)a.max_nodes * (sizeof(Node) + sizeof(int));
        NX = reinterpret_cast<double *>(mine);
        parent = reinterpret_cast<int *>(mine + (size_t)a.max_nodes * sizeof(Node));
    }
    NY = NX + a.max_nodes; NZ = NY + a.max_nodes; NC = NZ + a.max_nodes;
    w.gn = a.gn; w.inv_cell = a.inv_cell; w.W = a.W; w.Hbox = a.Hbox;

    const int n_work = a.tier_b ? __hip_atomic_load(a.queue + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : a.m;
    for (;;) {
        int scn = 0;
        if (lane == 0) scn = atomicAdd(a.queue + (a.tier_b ? 2 : 0), 1);
        scn = __shfl(scn, 0, 64);
        if (scn >= n_work) break;
        if (a.tier_b) scn = a.flagged[scn];
        Stream rs;
        rs.ext = a.uniforms ? a.uniforms + (size_t)scn * a.stream_len : nullptr;
        rs.ext_n = a.stream_len; rs.t = 0; rs.seed = a.seed; rs.scn = (uint32_t)(a.first + scn); rs.attempt = 0;
        int n_path = 0, iters = 0;
        bool overflow = false;
        double sx, sy, sz, gx, gy, gz;
        const int max_attempts = a.uniforms ? 1 : 4;
        for (int attempt = 0; attempt < max_attempts; ++attempt) {
            rs.attempt = (uint32_t)attempt;
            rs.t = 0;
            if (a.start_goal_in) {
                const double *sg = a.start_goal_in + (size_t)scn * 6;
                sx = sg[0]; sy = sg[1]; sz = sg[2]; gx = sg[3]; gy = sg[4]; gz = sg[5];
            } else {      // UAV.py:344,353-358 (the heading draw comes first in UAV.reset; it is not part of a scenario)
                (void)rs.uniform(0, kTwoPi);
                sx = rs.uniform(10, 210); sy = rs.uniform(1, 10); sz = 0.0;
                gx = rs.uniform(330, 490); gy = rs.uniform(420, 490); gz = 0.0;
            }
            if (lane == 0) { NX[0] = sx; NY[0] = sy; NZ[0] = sz; NC[0] = 0.0; parent[0] = -1; }
            __syncthreads();
            int nn = 1, goal_parent = -1;
            int it = 0;
            for (; it < a.max_iter && nn < a.max_nodes; ++it) {
                // ---- get_random_point (RRT.py:26-34)
                double rx, ry, rz;
                if (rs.uniform(0, 1) > 0.5) {
                    rx = rs.uniform(0, a.len); ry = rs.uniform(0, a.width); rz = rs.uniform(0, a.h);
                } else {
                    rx = gx; ry = gy; rz = gz;
                }
                // ---- nearest_node (RRT.py:36-37): first minimum
                double bd = 1e300;                    // SQUARED distance of the best node so far (see root_less)
                int bi = 0x7fffffff;
                for (int k0 = lane; k0 < nn; k0 += 256) {          // four rounds' operands in flight together
                    double qx[4], qy[4], qz[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int k = k0 + 64 * u < nn ? k0 + 64 * u : k0;     // (a repeat of k0 can never win: same root, not a lower index)
                        qx[u] = NX[k]; qy[u] = NY[k]; qz[u] = NZ[k];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const double d = ldist2(qx[u], qy[u], qz[u], rx, ry, rz);
                        if (k0 + 64 * u < nn && root_less(d, bd)) { bd = d; bi = k0 + 64 * u; }
                    }
                }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    const double od = __shfl_xor(bd, off, 64);
                    const int oi = __shfl_xor(bi, off, 64);
                    if (root_less(od, bd) || (oi < bi && root_equal(od, bd))) { bd = od; bi = oi; }
                }
                const Node nr = Node{NX[bi], NY[bi], NZ[bi], NC[bi]};
                // ---- steer (RRT.py:39-46); np.linalg.norm == sqrt(ddot) == an FMA chain on the reference platform
                double dx = rx - nr.x, dy = ry - nr.y, dz = rz - nr.z;
                const double length = sqrt(fma(dz, dz, fma(dy, dy, dx * dx)));
                double nx, ny, nz;
                if (length < a.step_size) {
                    nx = rx; ny = ry; nz = rz;
                } else {
                    dx = dx / length; dy = dy / length; dz = dz / length;
                    nx = nr.x + dx * a.step_size; ny = nr.y + dy * a.step_size; nz = nr.z + dz * a.step_size;
                }
                if (!obstacle_free(w, nr.x, nr.y, nr.z, nx, ny, nz, a.obstacle_step)) continue;     // :77-78
                const int me = nn;
                const double step2 = a.step_size * a.step_size;
                double my_cost = nr.cost + ldist(nr.x, nr.y, nr.z, nx, ny, nz);
                int my_parent = bi;
                // ---- rewire (RRT.py:84-90), sequential over the list, 64 distance tests at a time
                for (int base4 = 0; base4 < nn; base4 += 256) {
                  double qx[4], qy[4], qz[4];
#pragma unroll
                  for (int u = 0; u < 4; ++u) {
                      const int k = base4 + 64 * u + lane < nn ? base4 + 64 * u + lane : 0;
                      qx[u] = NX[k]; qy[u] = NY[k]; qz[u] = NZ[k];
                  }
                  unsigned long long masks[4];
#pragma unroll
                  for (int u = 0; u < 4; ++u) {
                      const bool near = base4 + 64 * u + lane < nn &&
                                        root_less_than(ldist2(qx[u], qy[u], qz[u], nx, ny, nz), a.step_size, step2);   // Loc.distance < step (RRT.py:86)
                      masks[u] = __ballot(near);
                  }
#pragma unroll
                  for (int u = 0; u < 4; ++u) {
                    const int base = base4 + 64 * u;
                    unsigned long long mask = masks[u];
                    while (mask) {
                        const int b = __builtin_ctzll(mask);
                        mask &= mask - 1;
                        const Node q = Node{NX[base + b], NY[base + b], NZ[base + b], NC[base + b]};
                        const double d = ldist(q.x, q.y, q.z, nx, ny, nz);
                        if (my_cost > q.cost + d) {
                            if (obstacle_free(w, q.x, q.y, q.z, nx, ny, nz, a.obstacle_step)) {
                                my_parent = base + b;
                                my_cost = q.cost + d;
                            }
                        }
                    }
                  }
                }
                if (lane == 0) { NX[me] = nx; NY[me] = ny; NZ[me] = nz; NC[me] = my_cost; parent[me] = my_parent; }
                nn += 1;
                __syncthreads();
                if (ldist(nx, ny, nz, gx, gy, gz) <= a.step_size) {                       // :92-94
                    goal_parent = me;
                    ++it;
                    break;
                }
            }
            iters = it;
            if (a.defer && goal_parent < 0 && nn >= a.max_nodes && it < a.max_iter) {    // the tree outgrew tier A's list: tier B
                overflow = true;
                break;
            }
            // ---- path = [start .. new_node, goal] (RRT.py:96-103)
            int count = 1;
            for (int p = goal_parent; p >= 0; p = parent[p]) count++;
            n_path = count;
            if ((count >= 2 && count <= a.K) || attempt + 1 == max_attempts) {
                if (count <= a.K) {
                    double *out = a.out_sub + (size_t)scn * a.K * 3;
                    if (lane == 0) {
                        int k = count - 1;
                        out[3 * k] = gx; out[3 * k + 1] = gy; out[3 * k + 2] = gz;
                        for (int p = goal_parent; p >= 0; p = parent[p]) {
                            --k;
                            out[3 * k] = NX[p]; out[3 * k + 1] = NY[p]; out[3 * k + 2] = NZ[p];
                        }
                    }
                    for (int q = count * 3 + lane; q < a.K * 3; q += 64) out[q] = 0.0;
                }
                break;
            }
            __syncthreads();
        }
        if (overflow) {
            if (lane == 0) a.flagged[atomicAdd(a.queue + 1, 1)] = scn;
            __syncthreads();
            continue;
        }
        if (lane == 0) {
            a.out_nsub[scn] = n_path <= a.K ? n_path : -n_path;
            if (a.out_iters) a.out_iters[scn] = iters;
            double *sg = a.out_start_goal + (size_t)scn * 6;
            sg[0] = sx; sg[1] = sy; sg[2] = sz; sg[3] = gx; sg[4] = gy; sg[5] = gz;
        }
        __syncthreads();
    }
}

}  // namespace

// declared in uavenv.hip (shares the env's world blob)
extern "C" int uavenv__world_view(const UavEnv *env, const unsigned char **blob, int32_t *bytes, int32_t *aux_off,
                                  int32_t *grid_off, int32_t *grid_stride, int32_t *gn, double *inv_cell, double *W, double *Hbox,
                                  double *len, int32_t *mask_bytes, int32_t *K);

namespace {

// queue counters + the list of tier-A overflows: device scratch of this translation unit, one per device, grown on demand
// (no internal threads, one stream at a time per env: the contract of include/uavenv.h).  A launch pair owns it from its
// memset to the end of tier B, both on the caller's stream; two envs planning on DIFFERENT streams at once would share it,
// so every call takes its own region of the ring of kScratchSlots.
constexpr int kScratchSlots = 4;
struct RrtScratch {
    int32_t *buf = nullptr;
    int cap = 0;            // scenarios per slot
    int next = 0;
};
RrtScratch g_scratch[16];

int32_t *scratch_for(int m)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    RrtScratch &sc = g_scratch[dev];
    if (sc.cap < m) {
        if (sc.buf) { (void)hipDeviceSynchronize(); (void)hipFree(sc.buf); sc.buf = nullptr; }
        const int cap = m < 4096 ? 4096 : m;
        if (hipMalloc((void **)&sc.buf, (size_t)kScratchSlots * (size_t)(cap + 4) * sizeof(int32_t)) != hipSuccess) { sc.cap = 0; return nullptr; }
        sc.cap = cap;
    }
    int32_t *p = sc.buf + (size_t)sc.next * (size_t)(sc.cap + 4);
    sc.next = (sc.next + 1) % kScratchSlots;
    return p;
}

template <typename MaskT>
int launch_tiers(RrtArgs a, hipStream_t s)
{
    static bool attr = false;
    const size_t world = (size_t)((a.world_bytes + 15) & ~15);
    const size_t lds_b = world + (size_t)kTierBNodes * (sizeof(Node) + sizeof(int));
    if (!attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_rrt_plan<MaskT, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds_b) != hipSuccess)
            return UAVENV_EHIP;
        attr = true;
    }
    static const int tier_a_nodes = [] {          // A/B knob (UAVENV_RRT_TIER_A=2048: the round-3 single tier)
        const char *e = getenv("UAVENV_RRT_TIER_A");
        const int v = e ? atoi(e) : 0;
        return v >= 16 && v <= kTierBNodes ? v : kTierANodes;
    }();
    int32_t *scr = scratch_for(a.m);
    if (!scr) return UAVENV_ENOMEM;
    a.queue = scr;
    a.flagged = scr + 4;
    if (hipMemsetAsync(scr, 0, 4 * sizeof(int32_t), s) != hipSuccess) return UAVENV_EHIP;
    if (a.node_scratch) {                                     // the background form: one launch, no LDS, the full node list
        a.max_nodes = kTierBNodes;
        a.tier_b = 0;
        a.defer = 0;
        const int grid = a.m < a.scratch_wgs ? a.m : a.scratch_wgs;
        hipLaunchKernelGGL((k_rrt_plan<MaskT, false>), dim3(grid), dim3(64), 0, s, a);
        return hipGetLastError() == hipSuccess ? UAVENV_OK : UAVENV_EHIP;
    }
    // tier A: seven wavefronts per CU (21 KB of LDS each); more workgroups than that only queue up behind them
    a.max_nodes = tier_a_nodes;
    a.tier_b = 0;
    a.defer = tier_a_nodes < kTierBNodes ? 1 : 0;            // (a full-size tier A is the round-3 single launch: nothing to defer)
    const size_t lds_a = world + (size_t)a.max_nodes * (sizeof(Node) + sizeof(int));
    const int per_cu = (int)(160 * 1024 / (lds_a + 512));
    int grid_a = 256 * (per_cu < 1 ? 1 : (per_cu > 8 ? 8 : per_cu));
    if (grid_a > a.m) grid_a = a.m;
    hipLaunchKernelGGL((k_rrt_plan<MaskT, true>), dim3(grid_a), dim3(64), lds_a, s, a);
    if (hipGetLastError() != hipSuccess) return UAVENV_EHIP;
    if (a.max_nodes < kTierBNodes) {
        a.max_nodes = kTierBNodes;
        a.tier_b = 1;
        a.defer = 0;
        const int grid_b = a.m < 256 ? a.m : 256;
        hipLaunchKernelGGL((k_rrt_plan<MaskT, true>), dim3(grid_b), dim3(64), lds_b, s, a);
        if (hipGetLastError() != hipSuccess) return UAVENV_EHIP;
    }
    return UAVENV_OK;
}

}  // namespace

extern "C" int64_t uavenv_rrt_scratch_bytes(int32_t wgs) { return (int64_t)wgs * kTierBNodes * (int64_t)(sizeof(Node) + sizeof(int)); }

// node_scratch != NULL (uavenv_rrt_scratch_bytes(scratch_wgs) bytes of device memory): the LDS-free background form on
// scratch_wgs wavefronts; NULL: the two LDS tiers.
extern "C" int uavenv_rrt_plan_at(UavEnv *env, int32_t first, int32_t m, const double *start_goal_dev, const double *uniforms_dev,
                                  int32_t stream_len, uint64_t seed, int32_t max_iter, double step_size, double obstacle_step,
                                  double *out_start_goal_dev, double *out_sub_dev, int32_t *out_nsub_dev,
                                  int32_t *out_iters_dev, void *node_scratch, int32_t scratch_wgs, void *stream)
{
    if (node_scratch && scratch_wgs <= 0) return UAVENV_EINVAL;
    if (!env || m <= 0 || first < 0 || !out_start_goal_dev || !out_sub_dev || !out_nsub_dev || max_iter <= 0 || step_size <= 0 ||
        obstacle_step <= 0 || (uniforms_dev && stream_len <= 0))
        return UAVENV_EINVAL;
    RrtArgs a;
    int32_t mask_bytes = 8;
    double len = 0;
    if (uavenv__world_view(env, &a.world_blob, &a.world_bytes, &a.aux_off, &a.grid_off, &a.grid_stride, &a.gn, &a.inv_cell, &a.W,
                           &a.Hbox, &len, &mask_bytes, &a.K) != UAVENV_OK)
        return UAVENV_EINVAL;
    a.len = len; a.width = a.W; a.h = a.Hbox;
    a.m = m; a.max_iter = max_iter;
    a.max_nodes = kTierBNodes;
    a.step_size = step_size; a.obstacle_step = obstacle_step;
    a.start_goal_in = start_goal_dev; a.uniforms = uniforms_dev; a.stream_len = stream_len; a.seed = seed;
    a.out_start_goal = out_start_goal_dev; a.out_sub = out_sub_dev; a.out_nsub = out_nsub_dev; a.out_iters = out_iters_dev;
    a.first = first; a.tier_b = 0; a.defer = 0; a.queue = nullptr; a.flagged = nullptr;
    a.node_scratch = (unsigned char *)node_scratch; a.scratch_wgs = scratch_wgs;
    hipStream_t s = (hipStream_t)stream;
    return mask_bytes == 4 ? launch_tiers<uint32_t>(a, s) : launch_tiers<uint64_t>(a, s);
}

extern "C" int uavenv_rrt_plan(UavEnv *env, int32_t m, const double *start_goal_dev, const double *uniforms_dev,
                               int32_t stream_len, uint64_t seed, int32_t max_iter, double step_size, double obstacle_step,
                               double *out_start_goal_dev, double *out_sub_dev, int32_t *out_nsub_dev,
                               int32_t *out_iters_dev, void *stream)
{
    // UAVENV_RRT_BACKGROUND=<wavefronts> (tests): run the LDS-free background form here too, so that the parity tests of the
    // planner (Mersenne replay, oracle streams) cover it
    static const int bg_wgs = [] { const char *e = getenv("UAVENV_RRT_BACKGROUND"); const int v = e ? atoi(e) : 0; return v > 0 && v <= 4096 ? v : 0; }();
    static void *bg_scratch = nullptr;
    if (bg_wgs && !bg_scratch && hipMalloc(&bg_scratch, (size_t)uavenv_rrt_scratch_bytes(bg_wgs)) != hipSuccess) return UAVENV_ENOMEM;
    return uavenv_rrt_plan_at(env, 0, m, start_goal_dev, uniforms_dev, stream_len, seed, max_iter, step_size, obstacle_step,
                              out_start_goal_dev, out_sub_dev, out_nsub_dev, out_iters_dev, bg_wgs ? bg_scratch : nullptr, bg_wgs, stream);
}
