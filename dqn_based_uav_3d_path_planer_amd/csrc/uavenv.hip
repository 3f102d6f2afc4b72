// uavenv.hip -- kernels + C ABI of the MI355X-native PathPlan_City env hot path (see include/uavenv.h).
//
// Data layout in HBM (per device, N = n_envs * uav_per_env agents)
//   hot state  : structure-of-arrays, one f64/i32/u8 array of length N per field, so a wavefront's
//                64 lanes read/write 64 consecutive elements (512 B per f64 field) -- fully coalesced.
//                Includes a 2-entry window (s0, s1) of the agent's sub-goal list: the step and the
//                observation only ever read sub_goals[0] and sub_goals[1] (UAV.py:412-440, 519-531).
//   cold state : sub-goal lists, array-of-structures [N][K][3] f64; touched only when a sub-goal is
//                popped (one 24 B gather), at reset (copy from the scenario bank) or by APF.
//   world      : <= 64 cylinders (32 B each) + a gnx x gny grid of candidate bit-masks, staged into
//                LDS once per workgroup; every occupancy probe is then 1 LDS mask read + ~0-2 LDS
//                cylinder reads instead of a loop over all buildings.
//   outputs    : observation rows [N][100] (f32/f16) written straight into the caller's buffer --
//                point it at replay frame t+1 and the replay append costs no extra bytes.
// Roofline: HBM-bound by design (604 algorithmic B per agent-step); the f64
// transcendental chain of the step (4 atan2, 2 sin/cos pairs, ~8 sqrt) is the compute floor.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <utility>
#include <vector>

#include "../../include/uavenv.h"
#include "uavenv_device.hpp"
#include "qnet_device.hpp"
#include "dqn_internal.hpp"

using namespace uav;

extern "C" int uavenv_rrt_plan_at(UavEnv *env, int32_t first, int32_t m, const double *start_goal_dev, const double *uniforms_dev,
                                  int32_t stream_len, uint64_t seed, int32_t max_iter, double step_size, double obstacle_step,
                                  double *out_start_goal_dev, double *out_sub_dev, int32_t *out_nsub_dev,
                                  int32_t *out_iters_dev, void *node_scratch, int32_t scratch_wgs, void *stream);
extern "C" int64_t uavenv_rrt_scratch_bytes(int32_t wgs);
extern "C" int uavenv_rrt_plan(UavEnv *env, int32_t m, const double *start_goal_dev, const double *uniforms_dev,
                               int32_t stream_len, uint64_t seed, int32_t max_iter, double step_size, double obstacle_step,
                               double *out_start_goal_dev, double *out_sub_dev, int32_t *out_nsub_dev,
                               int32_t *out_iters_dev, void *stream);

// ------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                               \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess) return fail(UAVENV_EHIP, "%s -> %s", #expr, hipGetErrorString(e_));   \
    } while (0)

// ------------------------------------------------------------------------------------------------
// device-side state
// ------------------------------------------------------------------------------------------------
// One slab, three typed planes; field k of a plane is the npad-element array at base + k*npad.  Only the
// three base pointers (+ npad, sub) travel as kernel arguments: 25 separate pointers would not fit the SGPR
// budget and the compiler would spill them through v_writelane/v_readlane.
enum F64Field { F_PX, F_PY, F_PZ, F_VX, F_VY, F_V, F_GX, F_GY, F_GZ, F_S0X, F_S0Y, F_S0Z, F_S1X, F_S1Y, F_S1Z,
                F_SCORE, F_TOTAL, F_PATHLEN, F_HEAD };
enum I32Field { I_STEP, I_SUBIDX, I_NTOTAL, I_EPOCH, I_SCN, I_FLAGS };
// I_FLAGS packs done (bit 0), alias (bit 1), reach_goal (bit 2) in ONE dword: byte-sized planes made the compiler
// zero-extend each byte right after its load, i.e. wait for every earlier load before the LDS staging could start.
constexpr int kFlagDone = 1, kFlagAlias = 2, kFlagReach = 4;
struct DevState {
    double *f64;     // [19][npad]
    int32_t *i32;    // [6][npad]
    size_t npad;
    double *sub;     // [N][K][3]
    __host__ __device__ double *F(int k) const { return f64 + (size_t)k * npad; }
    __host__ __device__ int32_t *I(int k) const { return i32 + (size_t)k * npad; }
};
constexpr int kNumF64 = 19, kNumI32 = 6;

struct Bank {
    const double *start_goal;   // [M][6]
    const double *sub;          // [M][K][3]
    const int32_t *nsub;        // [M]
    int32_t m;
};

struct StepArgs {
    DevState st;
    // world
    const unsigned char *world_blob;   // [BldLds x nb][grid halo 2][grid halo 10][grid halo 20]  (16-byte multiples)
    int32_t world_bytes, aux_off, grid_off, grid_stride;
    int32_t nb, gn;
    double inv_cell, W, Hbox;
    const BldApf *apf_b;
    const uint64_t *apf_grid;          // [apf_nz][apf_gn][apf_gn] masks of the MOVING cylinders whose 60 m force range can
                                       // reach the cell (its own, finer, 3-D resolution: see uavenv_set_buildings)
    int32_t apf_gn, apf_nz;
    double apf_inv_cell, apf_inv_cz;
    uint64_t apf_all;                  // mask of all moving cylinders (points outside the box)
    // agent params
    double max_v, steer;
    PowerParams pw;
    int32_t max_step, K, U, N, n_actions;
    int32_t tile_off;                  // LDS byte offset of the obs tile (TILE kernels)
    int32_t wave_slot;                 // bytes of per-wavefront LDS (work queue / observation tile share it)
    int32_t obsq_off;                  // LDS byte offset of the per-wave observation work queues (ObsWaveLds[waves])
    int32_t apf_split;                 // != 0: k_apf_adjust ran before this k_step launch (see there)
    int32_t block;                     // workgroup size, passed as an argument: reading blockDim.x costs a vector load
                                       // from the dispatch packet + s_waitcnt vmcnt(0) in front of the staging barrier
    // io
    const void *actions;
    int32_t action_kind;
    void *obs;
    double *reward64;
    float *reward32;
    uint8_t *ret_done, *agent_done, *info, *valid;
    double *energy64;
    const uint2 *emit_lut;             // [25][64] packed (row, mask word, shift, scalar base, scalar nibble) of ctile_emit
    const uint8_t *active;             // nullable per-agent mask: 0 -> the agent is left untouched (valid = 0)
    unsigned long long *dbg;           // diagnostics: per-wave phase timestamps (s_memtime), 8 slots per wave
    uint32_t flags;
    // auto reset
    Bank bank;
    uint64_t seed, tick;
    // the policy in the prologue of k_step_coop (uavenv_step_policy): Q(s) + epsilon-greedy of Trainer/DuelingDQN_Trainer.py:86-97
    const float *pol_local;            // nullable: q_local's flat parameter block
    const float *pol_img;              // nullable: its layer 1 in the split form (csrc/dqn_internal.hpp; packed rows only)
    const uint32_t *pol_obs;           // rows of the CURRENT frame: [N][20] packed (k_step_coop) or [N][100] halfs (k_step)
    int32_t *pol_act;                  // [N] chosen action indices out (the replay ring's action plane)
    int32_t pol_dueling, pol_off;      // pol_off: LDS byte offset of the policy's tiles
    float pol_eps;
    uint64_t pol_seed, pol_counter;
    // uavenv_set_moved_word: stamped with moved_value by a step launch that moved at least one agent (valid = 1)
    uint32_t *moved_word;
    uint32_t moved_value;
    // uavenv_set_step_meta: the transition records of this launch ({a1, a0, reward, done | valid << 8 | info << 16}; include/uavenv.h)
    uint4 *meta;                       // nullable [N]
    const float *meta_a1;              // nullable [N]: the second action component to record
};

struct UavEnv {
    UavEnvConfig cfg;
    int N;
    DevState st;
    void *slab = nullptr;
    size_t slab_bytes = 0;
    // world
    unsigned char *world_blob = nullptr;
    int world_bytes = 0, aux_off = 0, grid_off = 0, grid_stride = 0, nb = 0, gn = 0, mask_bytes = 8;
    double cell = 10.0;
    BldApf *apf_b = nullptr;
    uint64_t *apf_grid = nullptr;
    uint64_t apf_all = 0;
    int bank_replaced = 0;
    int apf_gn = 1, apf_nz = 1;
    double apf_cell = 10.0, apf_cz = 10.0;
    uint2 *emit_lut = nullptr;
    bool have_world = false;
    // bank
    double *bank_sg = nullptr, *bank_sub = nullptr;
    int32_t *bank_nsub = nullptr;
    int bank_m = 0;
    uint64_t seed = 0, tick = 0;
    unsigned long long *dbg = nullptr;
    uint32_t *moved_word = nullptr;    // uavenv_set_moved_word
    void *step_meta = nullptr;         // uavenv_set_step_meta: consumed by the next step launch
    const float *step_meta_a1 = nullptr;
    // rolling refresh of the bank (uavenv_replan_*): staged plans of one slice + bookkeeping
    double *rp_sg = nullptr, *rp_sub = nullptr;
    int32_t *rp_nsub = nullptr;        // [cap] staged n_sub, then [cap] in-use flags
    int32_t *rp_counters = nullptr;    // [4] rows committed / skipped: in use / skipped: no plan (cumulative)
    int rp_cap = 0, rp_first = 0, rp_count = 0;
    void *rp_nodes = nullptr;          // node lists of the background planner's wavefronts (global memory: it uses no LDS)
    int rp_wgs = 0;
    bool rp_pending = false;
    hipEvent_t rp_done = nullptr;
    hipEvent_t rp_committed = nullptr; // recorded behind k_bank_commit: the next planner launch may not touch the staging area before it
    bool rp_commit_recorded = false;
    int world_gen = 0, rp_world_gen = 0;
    long long rp_calls = 0, rp_rows_planned = 0;
};

// ------------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------------
// global -> registers -> LDS by `nthr` threads (this thread is number `tid` of them), no barrier
__device__ __forceinline__ void stage_copy(unsigned char *smem, const StepArgs &a, int tid, int nthr)
{
    // up to kInFlight 16-byte loads per lane in flight (a one-load-per-iteration loop serialises one L2 round trip per
    // KiB and was ~20 % of the kernel at 16 384 envs)
    constexpr int kInFlight = 12;
    const uint4 *src = reinterpret_cast<const uint4 *>(a.world_blob);
    uint4 *dst = reinterpret_cast<uint4 *>(smem);
    const int n16 = a.world_bytes / 16;
    for (int base = 0; base < n16; base += nthr * kInFlight) {
        uint4 tmp[kInFlight];
#pragma unroll
        for (int r = 0; r < kInFlight; ++r) {
            const int k = base + r * nthr + tid;
            tmp[r] = src[k < n16 ? k : n16 - 1];      // unconditional (clamped) so all loads issue back to back
        }
#pragma unroll
        for (int r = 0; r < kInFlight; ++r)           // pin the loads here: keeps hipcc from sinking each one into
            asm volatile("" : "+v"(tmp[r].x), "+v"(tmp[r].y), "+v"(tmp[r].z), "+v"(tmp[r].w));   // its guarded store
#pragma unroll
        for (int r = 0; r < kInFlight; ++r) {         // unconditional too: a piece past the end was loaded from the last piece and is
            const int k = base + r * nthr + tid;      // stored onto it (same bytes) -- a guarded store is a divergent branch each
            dst[k < n16 ? k : n16 - 1] = tmp[r];
        }
    }
}

template <typename MaskT>
__device__ __forceinline__ WorldLds<MaskT> world_view(unsigned char *smem, const StepArgs &a)
{
    WorldLds<MaskT> w;
    w.b = reinterpret_cast<const BldLds *>(smem);
    w.aux = reinterpret_cast<const BldAux *>(smem + a.aux_off);
    for (int h = 0; h < 3; ++h) w.g[h] = reinterpret_cast<const MaskT *>(smem + a.grid_off + h * a.grid_stride);
    w.gn = a.gn;
    w.inv_cell = a.inv_cell;
    w.W = a.W;
    w.Hbox = a.Hbox;
    return w;
}

template <typename MaskT>
__device__ __forceinline__ WorldLds<MaskT> stage_world(unsigned char *smem, const StepArgs &a)
{
    stage_copy(smem, a, (int)threadIdx.x, a.block);
    __syncthreads();
    return world_view<MaskT>(smem, a);
}

// Agents/UAV.py:174-210  cal_force(point): attraction/repulsion + motion force of every MOVING building
// within 60 m of its rim.  Returns false where the reference would call Cal_SubTask_Dynamic() (raises).
// One (point, moving cylinder) pair of cal_force.  Returns false when the accumulated force passed 100 (:205-208).
struct ForceTerms {
    double s, f1x, f1y, f2x, f2y;      // f1 + f2, and the two force vectors (UAV.py:190-201)
};
// false: the cylinder is out of range (dis - R > 60, :186-187) and contributes nothing
__device__ __forceinline__ bool cal_force_terms(const BldApf &B, double x, double y, double z, ForceTerms &t)
{
    const double dx = B.cx - x, dy = B.cy - y, dz = B.cz - z;
    const double h2 = dx * dx + dy * dy;
    if (h2 + dz * dz > B.far2) return false;                                // certainly dis - R > 60
    const double dis = sqrt(h2 + dz * dz);                                  // :183 Eu_Loc_distance ((a-b)^2 == (b-a)^2 exactly)
    const double d2e = dis - B.R;
    if (d2e > 60.0) return false;                                           // :186-187
    // :190,:200  both magnitudes divide by dis2edge^2: one reciprocal, two products (each within 1 ulp of the
    // reference's quotient; every pair runs through here, and an f64 division is ~12 instructions)
    const double inv_dd = 1.0 / (d2e * d2e);
    const double q = B.R * inv_dd;
    double f1 = (q < 1.0) ? q : 1.0;                                        // :190
    if (d2e < 0.0) f1 = (-d2e > 2.0) ? -d2e : 2.0;                          // :196-197
    // :191,:198  direction sub-goal -> centre: (cos, sin) of calculate_angle = the unit vector of (dx, dy); the
    // zero vector has angle 0.  (The reference goes atan2 -> cos/sin; same direction to ~1e-16.)
    double cx1 = 1.0, sy1 = 0.0;
    if (h2 > 0.0) {
        const double inv = rsqrt(h2);
        cx1 = dx * inv;
        sy1 = dy * inv;
    }
    t.f1x = -f1 * cx1;
    t.f1y = -f1 * sy1;
    const double q2 = (B.vnorm * B.R) * inv_dd;
    const double f2 = (q2 < 1.0) ? q2 : 1.0;                                // :200
    t.f2x = f2 * B.ux;                                                      // :201
    t.f2y = f2 * B.uy;
    t.s = f1 + f2;
    return true;
}
// adds one in-range cylinder's terms; false when the accumulated force passed 100 (:205-208)
__device__ __forceinline__ bool force_accumulate(const ForceTerms &t, double &cum, double &tx, double &ty)
{
    cum += t.s;
    tx = (tx + t.f1x) + t.f2x;
    ty = (ty + t.f1y) + t.f2y;
    return !(cum > 100.0);
}
__device__ __forceinline__ bool cal_force_pair(const BldApf &B, double x, double y, double z, double &cum, double &tx,
                                               double &ty)
{
    ForceTerms t;
    if (!cal_force_terms(B, x, y, z, t)) return true;
    return force_accumulate(t, cum, tx, ty);
}

// cal_force's broad phase.  Inside the box only the cylinders listed for the point's cell are visited (ascending index
// = the reference's loop order with the out-of-range ones, which contribute nothing, left out); outside it, all the
// moving ones.
__device__ __forceinline__ uint64_t apf_mask(const StepArgs &a, double x, double y, double z)
{
    if (a.apf_grid && x >= 0.0 && x <= a.W && y >= 0.0 && y <= a.W && z >= 0.0 && z <= a.Hbox) {
        int ix = (int)(x * a.apf_inv_cell), iy = (int)(y * a.apf_inv_cell), iz = (int)(z * a.apf_inv_cz);
        ix = ix > a.apf_gn - 1 ? a.apf_gn - 1 : ix;
        iy = iy > a.apf_gn - 1 ? a.apf_gn - 1 : iy;
        iz = iz > a.apf_nz - 1 ? a.apf_nz - 1 : iz;
        return a.apf_grid[((size_t)iz * a.apf_gn + iy) * a.apf_gn + ix];
    }
    return a.apf_all;
}

__device__ __noinline__ bool cal_force_masked(const BldApf *__restrict__ b, uint64_t m, double x, double y, double z,
                                              double &fx, double &fy, double &fz)
{
    double cum = 0.0, tx = 0.0, ty = 0.0;
    bool ok = true;
    while (m) {
        const int i = __builtin_ctzll(m);
        m &= m - 1;
        if (!cal_force_pair(b[i], x, y, z, cum, tx, ty)) { ok = false; break; }
    }
    fx = tx; fy = ty; fz = 0.0;
    return ok;
}
__device__ __forceinline__ bool cal_force(const StepArgs &a, double x, double y, double z, double &fx, double &fy, double &fz)
{
    return cal_force_masked(a.apf_b, apf_mask(a, x, y, z), x, y, z, fx, fy, fz);
}

// Actions are fetched as raw bits (so the load can be issued early, with no dependent conversion) and decoded
// at first use.
struct RawAction {
    uint32_t lo, hi;
};
// the action as a transition record keeps it (include/uavenv.h: UavReplayRing.meta): the int32 index or the f32 steer bits as
// read; an f64 steer is narrowed to f32
__device__ __forceinline__ uint32_t meta_action_bits(const RawAction &ra, int kind, double a0)
{
    return kind == UAVENV_ACT_STEER_F64 ? __float_as_uint((float)a0) : ra.lo;
}
__device__ __forceinline__ RawAction load_action_raw(const void *actions, int kind, int i)
{
    RawAction r;
    r.hi = 0;
    if (kind == UAVENV_ACT_STEER_F64) {
        const uint2 v = reinterpret_cast<const uint2 *>(actions)[i];
        r.lo = v.x;
        r.hi = v.y;
    } else {
        r.lo = reinterpret_cast<const uint32_t *>(actions)[i];
    }
    return r;
}
__device__ __forceinline__ double decode_action(RawAction r, int kind, int n_actions)
{
    if (kind == UAVENV_ACT_STEER_F32) return (double)__uint_as_float(r.lo);
    if (kind == UAVENV_ACT_STEER_F64) return __longlong_as_double((long long)(((unsigned long long)r.hi << 32) | r.lo));
    return -1.0 + 2.0 * (double)(int)r.lo / (double)(n_actions - 1);
}

// The registers an agent lives in during a step.
struct Agent {
    ObsIn o;
    double head;                 // cached calculate_angle(0, V_vector) of the CURRENT velocity
    double score, total, path_len;
    int sub_idx, n_total, scn;   // scn >= 0: the sub-goal list is bank scenario scn (read-only); -1: private list
    int done, alias, reach;      // unpacked from I_FLAGS by unpack_flags() at first use
    int flags_raw;
    uint32_t epoch;
};

__device__ __forceinline__ void load_agent(const DevState &S, int i, Agent &g)
{
    ObsIn &o = g.o;
    o.px = S.F(F_PX)[i]; o.py = S.F(F_PY)[i]; o.pz = S.F(F_PZ)[i];
    o.vx = S.F(F_VX)[i]; o.vy = S.F(F_VY)[i]; o.V = S.F(F_V)[i];
    o.gx = S.F(F_GX)[i]; o.gy = S.F(F_GY)[i]; o.gz = S.F(F_GZ)[i];
    o.s0x = S.F(F_S0X)[i]; o.s0y = S.F(F_S0Y)[i]; o.s0z = S.F(F_S0Z)[i];
    o.s1x = S.F(F_S1X)[i]; o.s1y = S.F(F_S1Y)[i]; o.s1z = S.F(F_S1Z)[i];
    g.head = S.F(F_HEAD)[i];
    g.score = S.F(F_SCORE)[i]; g.total = S.F(F_TOTAL)[i]; g.path_len = S.F(F_PATHLEN)[i];
    o.step = S.I(I_STEP)[i];
    g.sub_idx = S.I(I_SUBIDX)[i]; g.n_total = S.I(I_NTOTAL)[i]; g.scn = S.I(I_SCN)[i];
    g.epoch = (uint32_t)S.I(I_EPOCH)[i];
    g.flags_raw = S.I(I_FLAGS)[i];
}

__device__ __forceinline__ void unpack_flags(Agent &g)
{
    g.done = g.flags_raw & kFlagDone ? 1 : 0;
    g.alias = g.flags_raw & kFlagAlias ? 1 : 0;
    g.reach = g.flags_raw & kFlagReach ? 1 : 0;
    g.o.n_rem = g.n_total - g.sub_idx;
}

__device__ __forceinline__ void store_agent(const DevState &S, int i, const Agent &g)
{
    const ObsIn &o = g.o;
    S.F(F_PX)[i] = o.px; S.F(F_PY)[i] = o.py; S.F(F_PZ)[i] = o.pz;
    S.F(F_VX)[i] = o.vx; S.F(F_VY)[i] = o.vy; S.F(F_V)[i] = o.V;
    S.F(F_GX)[i] = o.gx; S.F(F_GY)[i] = o.gy; S.F(F_GZ)[i] = o.gz;
    S.F(F_S0X)[i] = o.s0x; S.F(F_S0Y)[i] = o.s0y; S.F(F_S0Z)[i] = o.s0z;
    S.F(F_S1X)[i] = o.s1x; S.F(F_S1Y)[i] = o.s1y; S.F(F_S1Z)[i] = o.s1z;
    S.F(F_HEAD)[i] = g.head;
    S.F(F_SCORE)[i] = g.score; S.F(F_TOTAL)[i] = g.total; S.F(F_PATHLEN)[i] = g.path_len;
    S.I(I_STEP)[i] = o.step; S.I(I_SUBIDX)[i] = g.sub_idx; S.I(I_NTOTAL)[i] = g.n_total; S.I(I_SCN)[i] = g.scn;
    S.I(I_EPOCH)[i] = (int32_t)g.epoch;
    S.I(I_FLAGS)[i] = (g.done ? kFlagDone : 0) | (g.alias ? kFlagAlias : 0) | (g.reach ? kFlagReach : 0);
}

// Where agent i's sub-goal list lives: a bank scenario (shared, read-only) or its private [K][3] slot.
__device__ __forceinline__ const double *list_of(const StepArgs &a, int i, int scn)
{
    return scn >= 0 ? a.bank.sub + (size_t)scn * a.K * 3 : a.st.sub + (size_t)i * a.K * 3;
}

// UAV.reset() from the scenario bank (UAV.py:327-366 with the RRT result pre-planned), in two halves.
// reset_candidate needs nothing but (seed, tick, agent index): the Philox draw, the scenario's start / goal / first
// two sub-goals and the heading's sincos + calc_angle.  apply_reset installs it.  (The cooperative small-N kernel lets
// a helper wavefront prepare every agent's candidate while the first wavefront runs update_PathPlan.)  The sub-goal
// list is NOT copied: the agent just remembers the scenario id (APF mutates sub-goals, so there it is copied).
struct ResetCand {
    double f[16];     // px py pz gx gy gz s0x s0y s0z s1x s1y s1z vx vy V head
    int n_total, scn;
};

// the draw: scenario index + heading (random.uniform(0, 2*pi) = a + (b-a)*random(), a = 0)
__device__ __forceinline__ double reset_candidate_draw(const StepArgs &a, int i, ResetCand &c)
{
    const uint4 r = philox4x32_10(make_uint4((uint32_t)i, (uint32_t)a.tick, (uint32_t)(a.tick >> 32), 0x5eedu),
                                  make_uint2((uint32_t)a.seed, (uint32_t)(a.seed >> 32)));
    c.scn = (int)(uint32_t)(((uint64_t)r.z * (uint64_t)a.bank.m) >> 32);
    return (kTwoPi)*u53(r.x, r.y);
}

// the scenario's bank rows: start, goal, first two sub-goals (always K x 3 doubles, zero padded: no dependent branches)
__device__ __forceinline__ void reset_candidate_fetch(const StepArgs &a, ResetCand &c)
{
    const double *sg = a.bank.start_goal + (size_t)c.scn * 6;
    const double *src = a.bank.sub + (size_t)c.scn * a.K * 3;
    c.n_total = a.bank.nsub[c.scn];
#pragma unroll
    for (int k = 0; k < 6; ++k) c.f[k] = sg[k];
#pragma unroll
    for (int k = 0; k < 6; ++k) c.f[6 + k] = src[k];
}

__device__ __forceinline__ double reset_candidate_issue(const StepArgs &a, int i, ResetCand &c)
{
    const double heading = reset_candidate_draw(a, i, c);
    reset_candidate_fetch(a, c);
    return heading;
}

__device__ __forceinline__ void reset_candidate_finish(const StepArgs &a, double heading, ResetCand &c)
{
    double sn, cs;
    sincos(heading, &sn, &cs);
    double vx = a.max_v * cs, vy = a.max_v * sn;
    c.f[14] = calc_v(vx, vy, a.max_v);
    c.f[12] = vx;
    c.f[13] = vy;
    c.f[15] = calc_angle(vx, vy);
}

__device__ __forceinline__ void reset_candidate(const StepArgs &a, int i, ResetCand &c)
{
    const double heading = reset_candidate_issue(a, i, c);
    reset_candidate_finish(a, heading, c);
}

// COPY = false: the caller copies the list itself (k_step with the wave-level APF path: copy_reset_lists_wave)
template <bool APF, bool COPY = true>
__device__ __forceinline__ void apply_reset(const StepArgs &a, int i, Agent &g, const ResetCand &c)
{
    ObsIn &o = g.o;
    o.px = c.f[0]; o.py = c.f[1]; o.pz = c.f[2];
    o.gx = c.f[3]; o.gy = c.f[4]; o.gz = c.f[5];
    o.s0x = c.f[6]; o.s0y = c.f[7]; o.s0z = c.f[8];
    o.s1x = c.f[9]; o.s1y = c.f[10]; o.s1z = c.f[11];
    o.vx = c.f[12]; o.vy = c.f[13]; o.V = c.f[14];
    g.head = c.f[15];
    o.step = 0;
    g.score = 0.0; g.total = 0.0; g.path_len = 0.0;
    g.done = 0; g.reach = 0; g.epoch = 0;
    g.n_total = c.n_total;
    g.sub_idx = 0;
    g.alias = c.n_total >= 2 ? 1 : 0;      // path[0] is the start node == the position object (RRT.py:69)
    if (APF) {
        const double *src = a.bank.sub + (size_t)c.scn * a.K * 3;
        double *dst = a.st.sub + (size_t)i * a.K * 3;
        if (COPY) for (int k = 0; k < c.n_total * 3; ++k) dst[k] = src[k];
        g.scn = -1;
    } else {
        g.scn = c.scn;
    }
    o.n_rem = c.n_total;
}

template <bool APF>
__device__ __forceinline__ void reset_agent(const StepArgs &a, int i, Agent &g)
{
    ResetCand c;
    reset_candidate(a, i, c);
    apply_reset<APF>(a, i, g, c);
}

// APF resets of one wavefront: the reset itself per lane, then the scenario's sub-goal list into the agent's private
// slot by the whole wavefront, one reset agent after the other (a lane copying its own <= K x 3 doubles alone is a
// chain of dependent load -> store round trips: ~15 us of a 131 072-agent launch).  ALL lanes must call it.
__device__ __forceinline__ void reset_agents_wave(const StepArgs &a, int first_agent, int ii, Agent &g, bool do_reset)
{
    int scn = 0;
    if (do_reset) {
        ResetCand c;
        reset_candidate(a, ii, c);
        apply_reset<true, false>(a, ii, g, c);
        scn = c.scn;
    }
    const int lane = (int)threadIdx.x & 63;
    unsigned long long rm = __ballot(do_reset);
    while (rm) {
        const int l = __builtin_ctzll(rm);
        rm &= rm - 1;
        const int n3 = __shfl(g.n_total, l, 64) * 3;
        const double *src = a.bank.sub + (size_t)__shfl(scn, l, 64) * a.K * 3;
        double *dst = a.st.sub + (size_t)(first_agent + l) * a.K * 3;
        for (int k = lane; k < n3; k += 64) dst[k] = src[k];
    }
}

// Agents/UAV.py:397-513  update_PathPlan(action), first and second half.
// Split in two for the cooperative kernel: step_pre needs only the agent's own state and action (no world, no
// heading angle), step_post everything else.  k_step calls pre, the heading, post back to back (== the one function
// this used to be).
struct PreStep {
    double ox, oy, oz, dis_old, g_old, tgx, tgy;
    bool moved;                // false: :400-406, no sub-goal left -- nothing moves
};

// step_pre in two pieces, so that the policy form of the cooperative kernel can take the action-INdependent half in front of its
// staging barrier, while the weights are in flight (k_step_coop<POLICY>).  (Round 6 also had the helper wavefronts evaluate the
// sincos and the heading after the move for every discrete action speculatively -- bit-identical, all parity tests green, and
// SLOWER: 10.9 -> 11.95 us per launch.  No wavefront of the workgroup idles in front of the action barrier -- each runs a strip of
// the forward -- so the 1.8 k cycles of f64 moved the staging barrier by as much: profiles/r06_phase_k_step_coop_policy.txt.)
//   step_pre_dists   what does not depend on the action: :400-406's "nothing left", the old position, dis_old, g_old (:409-413)
//   step_pre_move    everything behind sincos(seta_new) (:415-422), given that sine and cosine
// step_pre = dists, sincos, move: the same operations on the same operands in the same order, whoever evaluates the sincos.
__device__ __forceinline__ void step_pre_dists(Agent &g, PreStep &P)
{
    ObsIn &o = g.o;
    P.moved = g.sub_idx < g.n_total;
    P.ox = o.px; P.oy = o.py; P.oz = o.pz;                                     // :409
    P.dis_old = P.g_old = P.tgx = P.tgy = 0.0;
    if (!P.moved) return;
    P.dis_old = dist3(o.px, o.py, o.pz, o.s0x, o.s0y, o.s0z);                  // :412
    P.g_old = dist3(o.px, o.py, o.pz, o.gx, o.gy, o.gz);                       // :413
}
__device__ __forceinline__ void step_pre_move(const StepArgs &a, Agent &g, PreStep &P, double sn, double cs)
{
    ObsIn &o = g.o;
    if (!P.moved) return;
    o.step += 1;                                                               // :408
    o.vx = a.max_v * cs;                                                       // :415
    o.vy = a.max_v * sn;                                                       // :416
    o.V = calc_v(o.vx, o.vy, a.max_v);                                         // :417
    o.px += o.vx;                                                              // :419
    o.py += o.vy;                                                              // :420
    if (g.alias) { o.s0x = o.px; o.s0y = o.py; o.s0z = o.pz; }                 // sub_goals[0] IS position after reset
    // :422-428  tri_goal = angle(sub0 - pos), tri_V = angle(V) -- or angle(sub0 - old pos) after a collision.  Only
    // cos|tri_goal - tri_V| is ever used, so the vectors are kept and cos_between() replaces two atan2 chains.
    P.tgx = o.s0x - o.px; P.tgy = o.s0y - o.py;                                // :422
}
__device__ __forceinline__ void step_pre(const StepArgs &a, double a0, Agent &g, PreStep &P)
{
    step_pre_dists(g, P);
    if (!P.moved) return;
    const double seta_new = g.head + a0 * a.steer;                             // :411 (cached: same V_vector), :414
    double sn, cs;
    sincos(seta_new, &sn, &cs);
    step_pre_move(a, g, P, sn, cs);
}

// The heading after the move, :423 (and obs[7], and the next step's :411): depends on the OLD heading and the action
// only, so the cooperative kernel lets another wavefront compute it.
template <bool INL>
__device__ __forceinline__ double heading_after(const StepArgs &a, double head_old, double a0)
{
    const double seta_new = head_old + a0 * a.steer;
    double sn, cs;
    sincos(seta_new, &sn, &cs);
    double vx = a.max_v * cs, vy = a.max_v * sn;
    (void)calc_v(vx, vy, a.max_v);
    return angle_of<INL>(vx, vy);
}

// update_PathPlan's second half in three pieces, so that the APF sub-goal adjustment in the middle can be done by the
// kernel in front (k_apf_adjust + adjust_subgoals_split) instead of one lane per agent:
//   step_post_a   :400-406 (no sub-goal left) and :425-444 -- collision, reward terms, path length
//   adjust_subgoals_lane / _split  :448-449, Adjust_subgoal :156-166 -- every remaining sub-goal += cal_force(sub-goal)
//   step_post_b   :450-453 (force on the UAV) and the termination cascade :456-513
struct PostMid {
    double r, tvx, tvy, dis_new, g_new;
    bool live;                 // false: the step ended in step_post_a (:400-406)
};

template <typename MaskT, bool APF, bool INL>
__device__ __forceinline__ void step_post_a(const StepArgs &a, const WorldLds<MaskT> &w, int ii, double a0, Agent &g,
                                            const PreStep &P, PostMid &M, int &ret_done, int &info, bool &head_set)
{
    ObsIn &o = g.o;
    M.r = 0.0; M.tvx = M.tvy = M.dis_new = M.g_new = 0.0;
    ret_done = 0; info = UAVENV_INFO_NORMAL;
    head_set = false;
    M.live = P.moved;
    if (!P.moved) {                                                            // :400-406
        g.done = 1;
        M.r += (double)(a.max_step - o.step);
        g.score += M.r;
        ret_done = 1; info = UAVENV_INFO_SUCCESS;
        return;
    }
    double r = 0.0;
    // :425-428 as selects, not a branch.  (hipcc 7.2 was seen to merge the divergent `if (collided) { pos = old; }` with
    // a stale copy of the new position feeding the distances below -- in one kernel, not in the other, same source.
    // Selects leave it nothing to merge.)
    const bool hit = probe(w, o.px, o.py, o.pz);
    r = hit ? r - 0.3 : r;
    o.px = hit ? P.ox : o.px;
    o.py = hit ? P.oy : o.py;
    o.pz = hit ? P.oz : o.pz;
    g.alias = hit ? 0 : g.alias;
    M.tvx = hit ? o.s0x - o.px : o.vx;
    M.tvy = hit ? o.s0y - o.py : o.vy;
    M.dis_new = dist3(o.px, o.py, o.pz, o.s0x, o.s0y, o.s0z);                  // :429
    M.g_new = dist3(o.px, o.py, o.pz, o.gx, o.gy, o.gz);                       // :430
    r -= 0.13 * fabs(a0);                                                      // :434
    r += 0.2 * cos_between(P.tgx, P.tgy, M.tvx, M.tvy);                        // :435
    r += 0.4 * (P.dis_old - M.dis_new);                                        // :436
    r += 0.4 * (P.g_old - M.g_new);                                            // :437
    r -= 0.1;                                                                  // :438
    r -= 0.01 * fabs(o.pz - o.s0z);                                            // :439-440
    g.path_len += o.V;                                                         // :443
    g.epoch += 1;                                                              // :444
    M.r = r;
}

// Adjust_subgoal (:156-166) for ONE agent by its own lane (private list only)
__device__ __forceinline__ void adjust_subgoals_lane(const StepArgs &a, int ii, Agent &g)
{
    ObsIn &o = g.o;
    double *lst = a.st.sub + (size_t)ii * a.K * 3;
    if (g.alias) { lst[g.sub_idx * 3] = o.s0x; lst[g.sub_idx * 3 + 1] = o.s0y; lst[g.sub_idx * 3 + 2] = o.s0z; }
    for (int k = g.sub_idx; k < g.n_total; ++k) {
        double fx, fy, fz;
        const double sx = lst[k * 3], sy = lst[k * 3 + 1], sz = lst[k * 3 + 2];
        cal_force(a, sx, sy, sz, fx, fy, fz);
        lst[k * 3] = sx + fx; lst[k * 3 + 1] = sy + fy; lst[k * 3 + 2] = sz + fz;
    }
    g.alias = 0;
    o.s0x = lst[g.sub_idx * 3]; o.s0y = lst[g.sub_idx * 3 + 1]; o.s0z = lst[g.sub_idx * 3 + 2];
    if (g.sub_idx + 1 < g.n_total) {
        o.s1x = lst[g.sub_idx * 3 + 3]; o.s1y = lst[g.sub_idx * 3 + 4]; o.s1z = lst[g.sub_idx * 3 + 5];
    }
}

// Adjust_subgoal for a whole launch as its OWN kernel, in front of k_step (k_step's registers allow two wavefronts per
// SIMD and a 131 072-agent launch has only 2 048 of them: the per-agent loop over ~20 sub-goals x the cylinders of each
// one's cell was a chain of dependent loads with nothing to overlap it -- 210 of 249 us).
//
// What makes the split legal: sub-goal k's new value depends on its old value and the (static) force field alone
// (:156-166, cal_force :174-210), not on the UAV -- EXCEPT the first remaining one while it still IS the position object
// (alias: the first step after a reset; Adjust_subgoal rebinds the list entries, so the alias ends there).  Whether the
// adjustment happens at all (:447: the step got past :400-406) is known before the step: the agent steps (not masked,
// not skip-done) and has a sub-goal left.  So this kernel adjusts every remaining sub-goal of every such agent, minus the
// aliased first one, which k_step does itself after the move (one cal_force on that lane); k_step then reads the new
// sub_goals[0] / [1] back from the list.  It must run BEFORE k_step (which pops sub-goals and, on auto-reset, rewrites
// the lists).  The values are computed by the same cal_force_pair calls on the same operands in the same order as
// adjust_subgoals_lane's: bit-identical lists.
//
// Geometry: a workgroup of four wavefronts takes 64 agents; their (agent, sub-goal) pairs form one list (prefix sum of
// the counts, owner of pair p by binary search in LDS), dealt to the four wavefronts 64 at a time, up to 256 in work per wavefront -- so the wavefronts of a workgroup finish together whatever the agents' list lengths are.  A pair costs one
// cal_force_pair (~130 f64-heavy instructions: two square roots, three divisions) per cylinder of its cell's mask, and
// the masks have 2.6 bits on average but 6-7 for the longest of 64 consecutive pairs, which the other 63 lanes would wait
// for.  So per chunk:
//   pass 1  lane p mod 64 reads pair p (consecutive lanes on consecutive 24-byte entries), looks its mask up and parks
//           (x, y, z, mask) in LDS; a histogram of min(popcount, 15) through ds_add_rtn gives bucket + rank in the bucket;
//   sort    counting sort of the chunk's pairs by mask size (16-bit indices scattered to their places);
//   pass 2  the forces, 64 pairs of (nearly) equal trip count at a time, LDS to LDS;
//   pass 3  the new entries back to the lists, again in list order.
// The order in which pairs are processed is free: every pair is an independent read-modify-write of its own entry.
// The moving-cylinder table sits in LDS.
constexpr int kApfAgentsPerBlock = 64;
constexpr int kApfBuckets = 16;
constexpr int kApfChunk = 192;
struct ApfPairLds {
    double x, y, z;
    unsigned long long m;
};
struct ApfSortLds {
    ApfPairLds rec[kApfChunk];
    uint16_t off[kApfChunk];       // list offset (agent-in-workgroup * K + k)
    uint16_t key[kApfChunk];       // bucket << 12 | rank
    uint16_t sorted[kApfChunk];
    uint32_t hist[kApfBuckets];
    int32_t ends[64], rels[64];    // inclusive prefix sum of the agents' pair counts; list offset - pair index
};
#ifdef UAVENV_PHASE_PROFILE   // diagnostic build only (scripts/phase_profile_apf.py): cycles per phase, summed per wavefront
#define APF_T0() unsigned long long apf_t = __builtin_amdgcn_s_memtime(), apf_acc[6] = {0, 0, 0, 0, 0, 0}
#define APF_LAP(k) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long n_ = __builtin_amdgcn_s_memtime(); apf_acc[k] += n_ - apf_t; apf_t = n_; } while (0)
#define APF_DUMP() do { if (a.dbg && lane == 0) for (int k_ = 0; k_ < 6; ++k_) a.dbg[((size_t)blockIdx.x * 4 + wave) * 8 + k_] = apf_acc[k_]; } while (0)
#else
#define APF_T0() do { } while (0)
#define APF_LAP(k) do { } while (0)
#define APF_DUMP() do { } while (0)
#endif
__global__ void __launch_bounds__(256) k_apf_adjust(StepArgs a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const DevState &S = a.st;
    APF_T0();
    {
        const double *src = reinterpret_cast<const double *>(a.apf_b);
        double *dst = reinterpret_cast<double *>(smem);
        for (int k = (int)threadIdx.x; k < a.nb * (int)(sizeof(BldApf) / 8); k += 256) dst[k] = src[k];
    }
    const BldApf *bl = reinterpret_cast<const BldApf *>(smem);
    const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6;
    ApfSortLds *L = reinterpret_cast<ApfSortLds *>(smem + (((a.nb > 0 ? a.nb : 1) * (int)sizeof(BldApf) + 31) & ~31)) + wave;
    const int first = blockIdx.x * kApfAgentsPerBlock;
    int cnt = 0, k0 = 0;
    if (first + lane < a.N) {                          // (every wavefront computes the workgroup's prefix sum for itself)
        const int ii = first + lane;
        const int sub = S.I(I_SUBIDX)[ii], nt = S.I(I_NTOTAL)[ii], fl = S.I(I_FLAGS)[ii];
        const bool masked = a.active && a.active[ii] == 0;
        const bool stepping = !(masked || ((a.flags & UAVENV_STEP_SKIP_DONE) && (fl & kFlagDone)));
        if (stepping && sub < nt) {
            k0 = sub + ((fl & kFlagAlias) ? 1 : 0);
            cnt = nt - k0;
        }
    }
    int end = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(end, d, 64);
        if (lane >= d) end += t;
    }
    L->ends[lane] = end;
    L->rels[lane] = lane * a.K + k0 - (end - cnt);     // list offset (agent-in-workgroup * K + k) = p + rel
    const int total = __shfl(end, 63, 64);
    double *lists = a.st.sub + (size_t)first * a.K * 3;
    __syncthreads();                                   // the table (and ends / rels)
    APF_LAP(0);
    // The list is dealt to the four wavefronts 64 pairs (one "group": 64 consecutive entries) at a time, wavefront w
    // taking groups w, w + 4, ...; a wavefront works on up to kApfChunk / 64 of its groups at once.
    constexpr int kPer = kApfChunk / 64;
    const int n_groups = (total + 63) >> 6;
    const int my_groups = (n_groups - wave + 3) >> 2;
    const int n_chunks = (my_groups + kPer - 1) / kPer;
    const int gpc = n_chunks > 0 ? (my_groups + n_chunks - 1) / n_chunks : 1;     // groups per chunk, evened out
    for (int g0 = 0; g0 < my_groups; g0 += gpc) {
        const int ng = my_groups - g0 < gpc ? my_groups - g0 : gpc;
        const int last_base = (((g0 + ng - 1) << 2) + wave) << 6;                 // only the list's last group can be short
        const int n = (ng - 1) * 64 + (total - last_base < 64 ? total - last_base : 64);
        if (lane < kApfBuckets) L->hist[lane] = 0;
        wave_lds_sync();
        // ---- pass 1 (the chunk's four loads per lane issued back to back, then the four table lookups: a wavefront's
        // chunks are a serial chain of HBM round trips otherwise)
        double ex[kPer], ey[kPer], ez[kPer];
        int offv[kPer];
#pragma unroll
        for (int u = 0; u < kPer; ++u) {
            const int i = lane + 64 * u;
            const int p = i < n ? ((((g0 + u) << 2) + wave) << 6) + lane : 0;     // slot i of the chunk <-> pair p
            int own = 0;                                       // ends[] is non-decreasing: the owner is #{j : ends[j] <= p}
#pragma unroll
            for (int st = 32; st > 0; st >>= 1)
                if (L->ends[own + st - 1] <= p) own += st;
            offv[u] = p + L->rels[own];
            const double *e = lists + (size_t)offv[u] * 3;
            ex[u] = e[0]; ey[u] = e[1]; ez[u] = e[2];
        }
        unsigned long long mv[kPer];
#pragma unroll
        for (int u = 0; u < kPer; ++u) mv[u] = apf_mask(a, ex[u], ey[u], ez[u]);
        APF_LAP(1);
#pragma unroll
        for (int u = 0; u < kPer; ++u) {
            const int i = lane + 64 * u;
            if (i < n) {
                const int pc = __popcll(mv[u]) < kApfBuckets - 1 ? __popcll(mv[u]) : kApfBuckets - 1;
                const uint32_t rank = atomicAdd(&L->hist[pc], 1u);
                L->rec[i] = ApfPairLds{ex[u], ey[u], ez[u], mv[u]};
                L->off[i] = (uint16_t)offv[u];
                L->key[i] = (uint16_t)((pc << 12) | rank);
            }
        }
        wave_lds_sync();
        APF_LAP(2);
        {   // bucket starts (exclusive scan of the histogram), in place
            const uint32_t h = lane < kApfBuckets ? L->hist[lane] : 0;
            uint32_t inc = h;
#pragma unroll
            for (int d = 1; d < kApfBuckets; d <<= 1) {
                const uint32_t t = __shfl_up(inc, d, 64);
                if (lane >= d) inc += t;
            }
            wave_lds_sync();
            if (lane < kApfBuckets) L->hist[lane] = inc - h;
        }
        wave_lds_sync();
        for (int i = lane; i < n; i += 64) {
            const int key = L->key[i];
            L->sorted[L->hist[key >> 12] + (key & 4095)] = (uint16_t)i;
        }
        wave_lds_sync();
        APF_LAP(3);
        // ---- pass 2: the forces, in bucket order
        for (int q = lane; q < n; q += 64) {
            const int i = L->sorted[q];
            const ApfPairLds rec = L->rec[i];
            unsigned long long m = rec.m;
            double cum = 0.0, tx = 0.0, ty = 0.0;
            while (m) {
                const int bi = __builtin_ctzll(m);
                m &= m - 1;
                if (!cal_force_pair(bl[bi], rec.x, rec.y, rec.z, cum, tx, ty)) break;
            }
            L->rec[i].x = rec.x + tx;
            L->rec[i].y = rec.y + ty;
        }
        wave_lds_sync();
        APF_LAP(4);
        // ---- pass 3
        for (int i = lane; i < n; i += 64) {
            double *e = lists + (size_t)L->off[i] * 3;
            e[0] = L->rec[i].x; e[1] = L->rec[i].y; e[2] = L->rec[i].z + 0.0;
        }
        wave_lds_sync();
        APF_LAP(5);
    }
    APF_DUMP();
}

// k_step's share of the adjustment when k_apf_adjust ran: the aliased first sub-goal (`pre_alias`: the alias flag before
// the move), then the new sub_goals[0] / [1] from the list.
__device__ __forceinline__ void adjust_subgoals_split(const StepArgs &a, int ii, Agent &g, bool pre_alias)
{
    ObsIn &o = g.o;
    double *lst = a.st.sub + ((size_t)ii * a.K + g.sub_idx) * 3;
    if (pre_alias) {
        // (a collision this step ended the alias before the adjustment: then the list's own entry is the operand)
        const double sx = g.alias ? o.s0x : lst[0], sy = g.alias ? o.s0y : lst[1], sz = g.alias ? o.s0z : lst[2];
        double fx, fy, fz;
        cal_force(a, sx, sy, sz, fx, fy, fz);
        o.s0x = sx + fx; o.s0y = sy + fy; o.s0z = sz + fz;
        lst[0] = o.s0x; lst[1] = o.s0y; lst[2] = o.s0z;
    } else {
        o.s0x = lst[0]; o.s0y = lst[1]; o.s0z = lst[2];
    }
    g.alias = 0;
    if (g.sub_idx + 1 < g.n_total) { o.s1x = lst[3]; o.s1y = lst[4]; o.s1z = lst[5]; }
}

// `head_set`: true when this function assigned g.head itself (:489, Calc_V rescaled the velocity again).
template <typename MaskT, bool APF, bool INL>
__device__ __forceinline__ void step_post_b(const StepArgs &a, const WorldLds<MaskT> &w, int ii, Agent &g, const PostMid &M,
                                            double &r, int &ret_done, int &info, bool &head_set)
{
    ObsIn &o = g.o;
    const int max_step = a.max_step;
    r = M.r;
    if (!M.live) return;
    const double tvx = M.tvx, tvy = M.tvy, dis_new = M.dis_new, g_new = M.g_new;
    if (APF) {                                                                 // :450-453
        double fx, fy, fz;
        cal_force(a, o.px, o.py, o.pz, fx, fy, fz);
        const double force = sqrt(fx * fx + fy * fy + fz * fz);
        r += 0.2 * force * cos_between(fx, fy, tvx, tvy);
    }

    const double d_sub = APF ? dist3(o.px, o.py, o.pz, o.s0x, o.s0y, o.s0z) : dis_new;   // same operands when !APF
    const double d_goal = g_new;
    if (o.step >= max_step) {                                                  // :456-465
        g.done = 1;
        r += (50.0 - d_sub);
        g.score += r; g.total += r;
        ret_done = 1; info = UAVENV_INFO_LOSE;
    } else if (d_sub < 7.0 || (d_goal < dist3(o.s0x, o.s0y, o.s0z, o.gx, o.gy, o.gz))) {   // :466
        r += (50.0 - d_sub);                                                   // :468
        g.sub_idx += 1;                                                        // :469 pop(0)
        g.alias = 0;
        if (g.sub_idx >= g.n_total) {                                          // :470-483
            r += 50.0;
            g.done = 1;
            r += (double)(max_step - o.step);
            g.score += r;
            g.reach = 1;
            g.total += r;
            ret_done = 1; info = UAVENV_INFO_SUCCESS;
        } else {                                                               // :484-495
            o.step = 0;                                                        // reset("local reset") :328-332
            g.score = 0.0;
            const double vx0 = o.vx, vy0 = o.vy;
            o.V = calc_v(o.vx, o.vy, a.max_v);
            o.s0x = o.s1x; o.s0y = o.s1y; o.s0z = o.s1z;
            if (g.sub_idx + 1 < g.n_total) {
                const double *nx = list_of(a, ii, g.scn) + (size_t)(g.sub_idx + 1) * 3;
                o.s1x = nx[0]; o.s1y = nx[1]; o.s1z = nx[2];
            }
            if (o.vx != vx0 || o.vy != vy0) { g.head = angle_of<INL>(o.vx, o.vy); head_set = true; }   // :489
            r += 0.2 * cos_between(o.s0x - o.px, o.s0y - o.py, o.vx, o.vy);    // :488-490
            r += (double)(max_step - o.step);                                  // :491
            g.score += r; g.total += r;
            ret_done = 1; info = UAVENV_INFO_SUCCESS;                          // returned done; agent NOT done
        }
    } else if (d_goal < 7.0) {                                                 // :496-509
        g.done = 1;
        r += 50.0;
        r += (double)(max_step - o.step);
        g.score += r;
        g.reach = 1;
        g.total += r;
        ret_done = 1; info = UAVENV_INFO_SUCCESS;
    } else {                                                                   // :510-513
        g.score += r; g.total += r;
    }
}

// the whole second half by one lane (cooperative kernel, and k_step when the wave-level adjustment is not available)
template <typename MaskT, bool APF, bool INL>
__device__ __forceinline__ void step_post(const StepArgs &a, const WorldLds<MaskT> &w, int ii, double a0, Agent &g,
                                          const PreStep &P, double &r, int &ret_done, int &info, bool &head_set)
{
    PostMid M;
    step_post_a<MaskT, APF, INL>(a, w, ii, a0, g, P, M, ret_done, info, head_set);
    if (APF && M.live) adjust_subgoals_lane(a, ii, g);
    step_post_b<MaskT, APF, INL>(a, w, ii, g, M, r, ret_done, info, head_set);
}

// Agents/UAV.py:397-513  update_PathPlan(action) on the register copy of one agent.
template <typename MaskT, bool APF, bool INL>
__device__ __forceinline__ void step_agent(const StepArgs &a, const WorldLds<MaskT> &w, int ii, double a0, Agent &g,
                                           double &r, int &ret_done, int &info)
{
    PreStep P;
    step_pre(a, a0, g, P);
    if (P.moved) g.head = angle_of<INL>(g.o.vx, g.o.vy);                       // :423
    bool head_set;
    step_post<MaskT, APF, INL>(a, w, ii, a0, g, P, r, ret_done, info, head_set);
}

// ------------------------------------------------------------------------------------------------
// the fused step kernel: update_PathPlan + (auto-reset) + state_PathPlan + output/replay write
// ------------------------------------------------------------------------------------------------
#ifdef UAVENV_PHASE_PROFILE   // diagnostic build only (scripts/phase_profile.py): the stamps perturb scheduling
#define UAV_STAMP(slot)                                                                                       \
    do {                                                                                                     \
        if (a.dbg && (threadIdx.x & 63) == 0)                                                                \
            a.dbg[((size_t)blockIdx.x * (a.block >> 6) + (threadIdx.x >> 6)) * 8 + (slot)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#define UAV_DRAIN() do { if (a.dbg) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); } } while (0)
#else
#define UAV_STAMP(slot) do { } while (0)
#define UAV_DRAIN() do { } while (0)
#endif
// -DUAVENV_PHASE_PROFILE -DUAVENV_PHASE_POLICY: slots 1..5 of k_step_coop hold the stages of the POLICY prologue instead of the
// step's own phases (scripts/phase_profile_coop.py, POLICY=2): 1 after the staging barrier, 2 after the layer-1 forward, 3 after layer
// 2 + epsilon-greedy, 4 after the action barrier, 5 after step_pre; 0 = start, 6 = update_PathPlan done (wave 0), 7 = end
#if defined(UAVENV_PHASE_PROFILE) && defined(UAVENV_PHASE_POLICY)
#define UAV_PSTAMP(slot) UAV_STAMP(slot)
#define UAV_CSTAMP(slot) do { if ((slot) == 0 || (slot) == 7) UAV_STAMP(slot); else if ((slot) == 2) UAV_STAMP(6); } while (0)
#else
#define UAV_PSTAMP(slot) do { } while (0)
#define UAV_CSTAMP(slot) UAV_STAMP(slot)
#endif

#ifndef UAVENV_KSTEP_WAVES
#define UAVENV_KSTEP_WAVES 1      // min waves per SIMD the register allocator must leave room for (A/B knob)
#endif
// POLH (f16 rings, uavenv_step_policy on launches too large for k_step_coop; k_step_polh below): 64 agents per workgroup
// and THREE wavefronts -- the agents and two policy wavefronts, which compute Q(s) of the 64 agents (qnet_device.hpp:
// polh_wave: the f16-MFMA forward of k_dqn_act_h, bit-identical) while the agent wavefront waits for its state and stages
// the world; the Q values cross through LDS and the agent lanes pick their own epsilon-greedy action.  The act launch this
// replaces cost 7.7 us + a launch boundary at 65 536 agents.
template <typename MaskT, bool APF, int OBS, bool POLH>
__device__ __forceinline__ void k_step_body(const StepArgs &a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const DevState &S = a.st;
    const int N = a.N;
    const int n_round = (N + 63) & ~63;
    if (POLH && (int)threadIdx.x >= 64) {                  // ---- a policy wavefront (a.block == 64)
        // (the team's tiles share the bytes of the agent wavefront's work queue / tile slot, first touched after the barriers)
        uavq::polh_wave(((int)threadIdx.x >> 6) - 1, a.pol_local, a.n_actions, a.pol_dueling, a.pol_obs, (int)blockIdx.x * 64, N,
                        smem + a.obsq_off,
                        a.dbg ? a.dbg + (((size_t)gridDim.x + blockIdx.x) * 2 + ((threadIdx.x >> 6) - 1)) * 8 : nullptr);
        return;
    }
    const int i = blockIdx.x * a.block + threadIdx.x;
    UAV_STAMP(0);

    // issue the first tile's state loads BEFORE the world is staged: their HBM latency overlaps the LDS fill
    Agent g;
    RawAction ra = {0u, 0u};
    uint32_t meta_a1 = 0u;                                                     // (second action component for the transition record)
    if (i < n_round) {
        const int ii = i < N ? i : N - 1;
        load_agent(S, ii, g);
        if (!POLH) ra = load_action_raw(a.actions, a.action_kind, ii);
        if (a.meta_a1) meta_a1 = __float_as_uint(a.meta_a1[ii]);
    }
    uint4 pol_rn = make_uint4(0u, 0u, 0u, 0u);
    if (POLH) pol_rn = uavq::policy_philox(i, a.pol_seed, a.pol_counter);       // under the state loads
    const WorldLds<MaskT> w = stage_world<MaskT>(smem, a);
    if (POLH) {
        __syncthreads();                                                       // the policy wavefronts' second barrier
        const uavq::floatx4 q = reinterpret_cast<const uavq::floatx4 *>(smem + a.obsq_off + uavq::kPolWBytes + uavq::kPolXBytes)[threadIdx.x];
        const int act = uavq::policy_select(q, pol_rn, a.pol_eps, a.n_actions);
        if (i < N) a.pol_act[i] = act;
        ra.lo = (uint32_t)act;
        wave_lds_sync();                                                       // (the slot is the work queue from here on)
    }
    UAV_STAMP(1);

    // Every launch covers N with exactly one agent per thread: a straight-line body.  (A grid-stride loop made the
    // compiler hoist ~25 plane addresses out of it; they did not fit the SGPR file and were spilled through ~100
    // v_writelane right after the staging barrier, and the looped form needed 256 VGPRs instead of 128.)
    if (i < n_round) {
        const bool active = i < N;
        const int ii = active ? i : N - 1;
        unpack_flags(g);
        const double a0 = decode_action(ra, POLH ? UAVENV_ACT_INDEX_I32 : a.action_kind, a.n_actions);
        UAV_DRAIN();
        UAV_STAMP(2);
        double r = 0.0;
        int ret_done = 0, info = UAVENV_INFO_NORMAL, valid = 1;
        const bool masked = a.active && a.active[ii] == 0;
        const bool stepping = !(masked || ((a.flags & UAVENV_STEP_SKIP_DONE) && g.done));
        if (APF && a.apf_split) {
            // update_PathPlan in three pieces: k_apf_adjust has done the sub-goal lists (see there)
            if (stepping) {
                PreStep P;
                PostMid M;
                bool head_set = false;
                const bool pre_alias = g.alias != 0;
                step_pre(a, a0, g, P);
                if (P.moved) g.head = angle_of<true>(g.o.vx, g.o.vy);              // :423
                step_post_a<MaskT, APF, true>(a, w, ii, a0, g, P, M, ret_done, info, head_set);
                if (M.live) adjust_subgoals_split(a, ii, g, pre_alias);
                step_post_b<MaskT, APF, true>(a, w, ii, g, M, r, ret_done, info, head_set);
            } else {
                ret_done = g.done; info = UAVENV_INFO_SKIPPED; valid = 0;         // PathPlan_City.py:365-366
            }
        } else if (!stepping) {
            ret_done = g.done; info = UAVENV_INFO_SKIPPED; valid = 0;             // PathPlan_City.py:365-366
        } else {
            step_agent<MaskT, APF, true>(a, w, ii, a0, g, r, ret_done, info);
        }
        UAV_STAMP(3);
        g.o.n_rem = g.n_total - g.sub_idx;
        const int agent_done = g.done;
        const double energy = a.energy64 ? fly_power(a.pw, g.o.V, ii % a.U) : 0.0;

        // ---- auto reset: the env restarts when ALL of its U agents are done (PathPlan_City.py:252-259,416-417)
        bool did_reset = false;
        if (a.flags & UAVENV_STEP_AUTO_RESET) {
            const unsigned long long dm = __ballot(active && g.done);
            const int lane = threadIdx.x & 63;
            const int g0 = lane & ~(a.U - 1);          // U is a power of two (uavenv_create)
            const unsigned long long gm = (a.U >= 64) ? ~0ull : (((1ull << a.U) - 1ull) << g0);
            did_reset = active && ((dm & gm) == gm);
            if (APF && a.apf_split) reset_agents_wave(a, i - lane, ii, g, did_reset);
            // (Tried in round 3 and dropped: requesting the reset candidate's bank rows speculatively before update_PathPlan
            // for every agent that CAN finish this step, as k_step_coop's helper wavefront does -- the draw + 13 extra loads
            // per candidate lane cost more than the hidden round trip saved: 65 536 agents 12.7 -> 13.2 us, 98 304: 14.9 -> 15.3.)
            else if (did_reset) reset_agent<APF>(a, ii, g);
        }

        UAV_STAMP(4);
        const bool want_obs = a.obs && !(a.flags & UAVENV_STEP_NO_OBS);
        ObsBits bits;
        ObsScalars sc;
        // ---- observation of the (possibly reset) state: state_PathPlan, UAV.py:515-567.  Computed BEFORE any store is
        // issued (a wait on a later load would also wait for the stores in flight); wave-cooperative (work queue), so
        // every lane of the wavefront takes part, active or not.
        if (want_obs) {
            ObsWaveLds *L = reinterpret_cast<ObsWaveLds *>(smem + a.obsq_off + (threadIdx.x >> 6) * a.wave_slot);
            bits = obs_bits_queued(w, L, g.o.px, g.o.py, g.o.pz, active);
            sc = obs_scalars(g.o, g.head);                                         // :526 heading == cached angle
        }
        if (active) {
            UAV_STAMP(5);
            // ---- outputs of the transition
            if (a.reward64) a.reward64[i] = r;
            if (a.reward32) a.reward32[i] = (float)r;
            if (a.ret_done) a.ret_done[i] = (uint8_t)ret_done;
            if (a.agent_done) a.agent_done[i] = (uint8_t)agent_done;
            if (a.info) a.info[i] = (uint8_t)info;
            if (a.valid) a.valid[i] = (uint8_t)valid;
            if (a.meta)                                                      // the transition record (one 16-byte store per agent)
                a.meta[i] = make_uint4(meta_a1, meta_action_bits(ra, POLH ? UAVENV_ACT_INDEX_I32 : a.action_kind, a0),
                                       __float_as_uint((float)r), (uint32_t)ret_done | ((uint32_t)valid << 8) | ((uint32_t)info << 16));
            if (a.moved_word && valid)                                       // (every writer stores the same value)
                __hip_atomic_store(a.moved_word, a.moved_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (a.energy64) a.energy64[i] = energy;
            // ---- state write-back
            if (valid || did_reset) store_agent(S, i, g);
            if (want_obs && a.tile_off < 0) store_obs_row<OBS>(a.obs, i, sc, bits);
            UAV_STAMP(6);
            UAV_DRAIN();
            UAV_STAMP(7);
        }
        if (a.tile_off >= 0 && want_obs) {      // wave-cooperative coalesced tile store (opt-in)
            const int first = i - ((int)threadIdx.x & 63);
            uint32_t *tile = reinterpret_cast<uint32_t *>(smem + a.tile_off + (threadIdx.x >> 6) * a.wave_slot);
            store_obs_ctile<OBS>(a.obs, first, N - first, tile, sc, bits);
        }
    }
}

template <typename MaskT, bool APF, int OBS>
__global__ void __launch_bounds__(256, UAVENV_KSTEP_WAVES) k_step(StepArgs a)
{
    k_step_body<MaskT, APF, OBS, false>(a);
}

// 64 agents per workgroup: the agent wavefront + two policy wavefronts; three wavefronts per SIMD at 65 536 agents
template <typename MaskT>
__global__ void __launch_bounds__(192, 3) k_step_polh(StepArgs a)
{
    k_step_body<MaskT, false, OBS_KIND_F16, true>(a);
}

// ------------------------------------------------------------------------------------------------
// The same step for SMALL launches (<= 2 wavefronts' worth of agents per CU): four wavefronts per 64 agents.
// With one wavefront per CU the launch is as long as its slowest wavefront's serial chain (staging -> f64 step math ->
// reset -> observation -> 50+ stores) and three of the CU's four SIMDs idle.  Here wavefront 0 is the agent (lane ==
// agent, exactly the code above), and the other three take the work that does not depend on its serial chain:
//   while wave 0 waits for its state and runs update_PathPlan:  waves 2-3 stage the world blob, wave 1 prepares the
//       reset candidate of every agent that can finish this step (Philox, bank rows, heading), wave 2 the heading
//       after the move (calc_angle of the new velocity: old heading + action only)
//   barrier; every wave derives the reset decision and the final positions itself (stepped position, done flags and
//       candidates are all in LDS), so nobody waits for wave 0's reset bookkeeping:
//       waves 1-3: the cylinder part of the three occupancy stencils, every third candidate cylinder of each agent
//                  (+ the out-of-box bits of one stencil each)
//       wave 0:    installs resets, writes the state planes and step outputs, the below-probes and the 15 scalars
//       each ORs its bits into the compact tile
//   barrier; all four stream the 64 observation rows out of the tile (coalesced 1 KiB runs).
// Two workgroup barriers after the staging one; every value is computed by the same device functions on the same
// operands as in k_step.
struct CoopLds {
    double cand[16][64];
    int32_t cand_n[64], cand_scn[64];
    double pos[64][4];                 // position after update_PathPlan (x, y, z, pad); after a fallback reset: final
    double head[64];                   // heading after the move, from wave 2
    int32_t done[64];                  // agent done after update_PathPlan (what the reset decision looks at)
    ObsWaveLds q[3];                   // work queues of waves 1..3
    uint32_t tile[64 * kCTileLd];
};

// POLICY: the actions are not read from memory but computed in the prologue from the packed observation rows of the
// current frame -- the forward pass of k_dqn_act_packed (qnet_device.hpp: same code, same Philox stream, bit-identical
// actions), every wavefront taking the 16 agents of its strip.  The weights, the packed rows and this kernel's own state
// / world loads are in flight together, and the ~5.6 k-cycle forward runs under the state and world round trips; a
// separate act launch cost 6 us plus a launch boundary in front of this kernel's 8 us.
// PAHEAD (with POLICY): the policy's layer-1 forward with every operand of the strip requested first (116 more registers: one
// wavefront per SIMD) -- for launches of at most one workgroup per CU, where nothing else hides an LDS round trip per K block.
template <typename MaskT, bool APF, int OBS, bool POLICY = false, bool PAHEAD = false>
__global__ void __launch_bounds__(256) k_step_coop(StepArgs a)
{
    UAV_HOT_PRIO();
    extern __shared__ __align__(16) unsigned char smem[];
    const DevState &S = a.st;
    const int N = a.N;
    // (the wavefront index through readfirstlane: the four wavefronts of a workgroup do different jobs, and the compiler can only let
    // their live ranges share registers -- and branch on a scalar -- when it knows the index is wavefront-uniform)
    const int lane = (int)threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int first = (int)blockIdx.x * 64;
    const int i = first + lane;
    const bool active = i < N;
    const int ii = active ? i : N - 1;
    CoopLds *C = reinterpret_cast<CoopLds *>(smem + a.obsq_off);
    const bool want_obs = a.obs && !(a.flags & UAVENV_STEP_NO_OBS);
    const bool auto_reset = (a.flags & UAVENV_STEP_AUTO_RESET) != 0;
    UAV_CSTAMP(0);

    // ---- policy prologue, part 1: fc1 + the packed row of this lane's agent (strip wv, row lane & 15) in flight
    const uavq::W1Split pW1 = uavq::w1split_at(reinterpret_cast<float *>(smem + a.pol_off));      // fc1 + b1, split form (kTileF floats)
    float *pW2 = reinterpret_cast<float *>(smem + a.pol_off) + uavq::kSplitF;                     // [16][64]
    float *pb2 = pW2 + uavq::kMaxOut * uavq::kHid;                          // [16]
    int32_t *pact = reinterpret_cast<int32_t *>(pb2 + uavq::kMaxOut);       // [64] chosen actions of the workgroup's agents
    uavq::floatx4 vW[uavq::kStageIters];
    uavq::PRow prow;
    uavq::SplitScRegs pol_sc;
    float pol_w2[4] = {0, 0, 0, 0}, pol_b2v = 0.0f;
    uint4 pol_rn = make_uint4(0u, 0u, 0u, 0u);
    const int pol_i = first + wv * 16 + (lane & 15);
    const int pol_n2 = a.n_actions + (a.pol_dueling ? 1 : 0);
    if (POLICY) {
        const int tid = (int)threadIdx.x;
        const uavq::NetDev nl = uavq::net_view(a.pol_local, pol_n2);
        if (a.pol_img) uavq::img_issue(vW, a.pol_img);
        else { uavq::w_issue(vW, a.pol_local); uavq::w_issue_sc(pol_sc, a.pol_local, nl.b1); }
#pragma unroll
        for (int k = 0; k < 4; ++k) pol_w2[k] = nl.W2[tid + 256 * k < pol_n2 * uavq::kHid ? tid + 256 * k : pol_n2 * uavq::kHid - 1];
        pol_b2v = nl.b2[tid < pol_n2 ? tid : 0];
        uavq::prow_load(prow, a.pol_obs + (size_t)(pol_i < N ? pol_i : N - 1) * kPackedDwords);
        pol_rn = philox4x32_10(make_uint4((uint32_t)pol_i, (uint32_t)a.pol_counter, (uint32_t)(a.pol_counter >> 32), 0xac7u),
                               make_uint2((uint32_t)a.pol_seed, (uint32_t)(a.pol_seed >> 32)));
    }
    Agent g;
    RawAction ra = {0u, 0u};
    double head_old = 0.0;
    uint32_t meta_a1 = 0u;               // (second action component for the transition record)
    if (wv == 0) {                       // the state loads fly while the other three wavefronts stage the world blob
        load_agent(S, ii, g);
        if (!POLICY) ra = load_action_raw(a.actions, a.action_kind, ii);
        if (a.meta_a1) meta_a1 = __float_as_uint(a.meta_a1[ii]);
    } else if (wv == 2) {                // wave 2 computes the heading after the move (:423): old heading + action only
        head_old = S.F(F_HEAD)[ii];
        if (!POLICY) ra = load_action_raw(a.actions, a.action_kind, ii);
    }
    // wave 1: every agent's reset candidate.  Philox and the bank rows need nothing but the agent index: their round
    // trip runs under the world staging.
    // Only for agents that CAN be done after this step (already done, on their last sub-goal, out of steps, or within
    // reach of the goal): fetching bank rows for everyone tripled the launch's HBM reads.  Wave 0 falls back to
    // reset_agent for an agent without a candidate (cannot happen with this predicate; kept for safety).
    ResetCand cand;
    double cand_heading = 0.0;
    bool cand_want = false;
    if (wv == 1 && auto_reset) {
        const int c_flags = S.I(I_FLAGS)[ii], c_sub = S.I(I_SUBIDX)[ii], c_n = S.I(I_NTOTAL)[ii], c_step = S.I(I_STEP)[ii];
        const double dx = S.F(F_GX)[ii] - S.F(F_PX)[ii], dy = S.F(F_GY)[ii] - S.F(F_PY)[ii],
                     dz = S.F(F_GZ)[ii] - S.F(F_PZ)[ii];
        cand_heading = reset_candidate_draw(a, ii, cand);                    // under the loads' round trip
        const double reach = 7.0 + a.max_v + 1.0;                            // :496 d_goal < 7 after a move of <= max_v
        cand_want = (c_flags & kFlagDone) || c_sub >= c_n - 1 || c_step + 1 >= a.max_step ||
                    dx * dx + dy * dy + dz * dz < reach * reach;
        if (cand_want) reset_candidate_fetch(a, cand);
    }
    // waves 1-3: their rows of the copy-out table (depends on (instruction, lane) only)
    uint2 lut[7];
    {
        const int lo = wv == 0 ? 0 : 7 + (wv - 1) * 6;
#pragma unroll
        for (int j = 0; j < 6; ++j) lut[j] = a.emit_lut[(lo + j) * 64 + lane];
        lut[6] = a.emit_lut[6 * 64 + lane];          // wave 0's seventh
    }
    uint32_t *trow = C->tile + lane * kCTileLd;
    if (wv == 3) { trow[0] = 0u; trow[1] = 0u; trow[2] = 0u; }               // mask words: OR targets of the four waves
    double r = 0.0, a0 = 0.0;
    int ret_done = 0, info = UAVENV_INFO_NORMAL, valid = 1, agent_done = 0;
    double energy = 0.0;
    bool did_reset = false, skip = false, head_set = false;
    PreStep pre;
    pre.moved = false;
    if (POLICY) {                        // policy prologue, part 2: weights into LDS
        const int tid = (int)threadIdx.x;
        if (a.pol_img) uavq::img_commit(reinterpret_cast<float *>(smem + a.pol_off), vW); else uavq::w_commit_split(pW1, vW, pol_sc);
#pragma unroll
        for (int k = 0; k < 4; ++k) {    // (clamped like the loads: past the end, the last element onto itself -- no guarded stores)
            const int at = tid + 256 * k < pol_n2 * uavq::kHid ? tid + 256 * k : pol_n2 * uavq::kHid - 1;
            pW2[at] = pol_w2[k];
        }
        if (tid < pol_n2) pb2[tid] = pol_b2v;
    }
    if (wv >= 2) {                       // the world blob comes in through waves 2 and 3 (wave 1's loads depend on
        stage_copy(smem, a, (int)threadIdx.x - 128, 128);                    // its state loads: it would hold the barrier)
    } else if (wv == 0) {                             // first half of update_PathPlan: needs the agent's own state only
        unpack_flags(g);
        const bool masked = a.active && a.active[ii] == 0;
        skip = masked || ((a.flags & UAVENV_STEP_SKIP_DONE) && g.done);
        if (!POLICY) {
            a0 = decode_action(ra, a.action_kind, a.n_actions);
            if (!skip) step_pre(a, a0, g, pre);
        }
#ifndef UAVENV_PRE_DISTS_LATE            // (A/B knob: the round-5 placement, all of step_pre behind the action barrier)
        else if (!skip) {
            step_pre_dists(g, pre);      // the action-independent half, while the policy's weights are still in flight
        }
#endif
    }
    __syncthreads();                                                         // world (and fc1) staged
    UAV_PSTAMP(1);
    const WorldLds<MaskT> w = world_view<MaskT>(smem, a);
    if (POLICY) {                        // policy prologue, part 3: forward, layer 2, epsilon-greedy
        uavq::floatx4 h[4];
        uavq::W2Frag<4> F;
        // (bit-identical either way: the same MFMAs in the same order per accumulator; measured at 16 384 agents, round 5: 29.26 ->
        // 29.08 us per configs[1] pass with the operands requested first)
        if (PAHEAD) uavq::fwd_strip_split_ahead<false>(pW1, prow, h);
        else uavq::fwd_strip_split<false>(pW1, prow, h);
        UAV_PSTAMP(2);
        float q[4];
        uavq::w2_load<4>(F, pW2, pb2, pol_n2);     // (requested in front of the forward it costs 52 registers across it: 227 + 32 -> 255 + 76, and the forward 1.2 k cycles)
        uavq::q_strip<4>(h, F, pol_n2, a.n_actions, a.pol_dueling, q);
        if (lane < 16) {
            const float sample = (float)(pol_rn.x >> 8) * (1.0f / 16777216.0f);
            int act;
            if (sample > a.pol_eps) {
                act = 0;
                float bq = q[0];
#pragma unroll
                for (int k = 1; k < 4; ++k)
                    if (k < a.n_actions && q[k] > bq) { bq = q[k]; act = k; }
            } else {
                act = (int)(((uint64_t)pol_rn.y * (uint64_t)a.n_actions) >> 32);
            }
            pact[wv * 16 + lane] = act;
            if (pol_i < N) a.pol_act[pol_i] = act;
        }
        UAV_PSTAMP(3);
        __syncthreads();                                                     // the 64 actions of the workgroup are known
        UAV_PSTAMP(4);
        if (wv == 0 || wv == 2) ra.lo = (uint32_t)pact[lane];
        if (wv == 0) {
            a0 = decode_action(ra, UAVENV_ACT_INDEX_I32, a.n_actions);
#ifdef UAVENV_PRE_DISTS_LATE
            if (!skip) step_pre(a, a0, g, pre);
#else
            if (!skip && pre.moved) {    // (step_pre's other half: the distances were taken in front of the staging barrier)
                double sn, cs;
                sincos(g.head + a0 * a.steer, &sn, &cs);
                step_pre_move(a, g, pre, sn, cs);
            }
#endif
        }
    }
    UAV_PSTAMP(5);
    UAV_CSTAMP(1);

    if (wv == 1 && auto_reset) {
        if (cand_want) {
            reset_candidate_finish(a, cand_heading, cand);
#pragma unroll
            for (int k = 0; k < 16; ++k) C->cand[k][lane] = cand.f[k];
            C->cand_n[lane] = cand.n_total;
            C->cand_scn[lane] = cand.scn;
        } else {
            C->cand_scn[lane] = -1;                                          // no candidate prepared
        }
    }
    if (wv == 2) C->head[lane] = heading_after<true>(a, head_old, decode_action(ra, POLICY ? UAVENV_ACT_INDEX_I32 : a.action_kind, a.n_actions));
    if (wv == 0) {
        if (skip) {
            ret_done = g.done; info = UAVENV_INFO_SKIPPED; valid = 0;             // PathPlan_City.py:365-366
        } else {
            step_post<MaskT, APF, true>(a, w, ii, a0, g, pre, r, ret_done, info, head_set);
        }
    }
    if (wv == 0) {                       // what the other waves need to place every agent: where it is, whether it is done
        C->pos[lane][0] = g.o.px;
        C->pos[lane][1] = g.o.py;
        C->pos[lane][2] = g.o.pz;
        C->done[lane] = g.done;
    }
    UAV_CSTAMP(2);
    __syncthreads();                                                         // candidates ready, step done
    UAV_CSTAMP(3);
    // ---- auto reset: the env restarts when ALL of its U agents are done (PathPlan_City.py:252-259,416-417).  Every
    // wave takes the decision from the same LDS data.
    bool will_reset = false;
    if (auto_reset) {
        const unsigned long long dm = __ballot(active && C->done[lane] != 0);
        const int g0 = lane & ~(a.U - 1);          // U is a power of two (uavenv_create)
        const unsigned long long gm = (a.U >= 64) ? ~0ull : (((1ull << a.U) - 1ull) << g0);
        will_reset = active && ((dm & gm) == gm);
    }
    // an agent that resets without a prepared candidate (the predicate above makes that impossible; kept for safety):
    // wave 0 plans it the slow way and publishes the position behind one more barrier.  Workgroup-uniform.
    const bool fallback = __ballot(will_reset && C->cand_scn[lane] < 0) != 0ull;
    if (wv == 0) {
        if (!skip && pre.moved && !head_set) g.head = C->head[lane];         // :423, computed by wave 2
        g.o.n_rem = g.n_total - g.sub_idx;
        agent_done = g.done;
        energy = a.energy64 ? fly_power(a.pw, g.o.V, ii % a.U) : 0.0;
        if (will_reset) {
            ResetCand c;
            c.scn = C->cand_scn[lane];
            if (c.scn >= 0) {
#pragma unroll
                for (int k = 0; k < 16; ++k) c.f[k] = C->cand[k][lane];
                c.n_total = C->cand_n[lane];
                apply_reset<APF>(a, ii, g, c);
            } else {
                reset_agent<APF>(a, ii, g);
            }
            did_reset = true;
        }
        if (fallback) {
            C->pos[lane][0] = g.o.px;
            C->pos[lane][1] = g.o.py;
            C->pos[lane][2] = g.o.pz;
        }
    }
    if (fallback) __syncthreads();
    UAV_CSTAMP(4);
    if (wv == 0) {
        if (active) {
            // ---- outputs of the transition
            if (a.reward64) a.reward64[i] = r;
            if (a.reward32) a.reward32[i] = (float)r;
            if (a.ret_done) a.ret_done[i] = (uint8_t)ret_done;
            if (a.agent_done) a.agent_done[i] = (uint8_t)agent_done;
            if (a.info) a.info[i] = (uint8_t)info;
            if (a.valid) a.valid[i] = (uint8_t)valid;
            if (a.meta)                                                      // the transition record (one 16-byte store per agent)
                a.meta[i] = make_uint4(meta_a1, meta_action_bits(ra, POLICY ? UAVENV_ACT_INDEX_I32 : a.action_kind, a0),
                                       __float_as_uint((float)r), (uint32_t)ret_done | ((uint32_t)valid << 8) | ((uint32_t)info << 16));
            if (a.moved_word && valid)                                       // (every writer stores the same value)
                __hip_atomic_store(a.moved_word, a.moved_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (a.energy64) a.energy64[i] = energy;
            // ---- state write-back
            if (valid || did_reset) store_agent(S, i, g);
        }
        if (want_obs) {                  // the cylinder-free part of state_PathPlan
            ctile_write_scalars(trow, obs_scalars(g.o, g.head));             // :526 heading == cached angle
            ObsBits part = {0u, 0u, 0u, 0u};
            part.below = obs_below_bits(w, g.o.px, g.o.py, g.o.pz);
            uint32_t m0, m1, m2;
            ctile_mask_words(part, m0, m1, m2);
            if (m0) atomicOr(&trow[0], m0);
            if (m1) atomicOr(&trow[1], m1);
            if (m2) atomicOr(&trow[2], m2);
        }
    } else if (want_obs) {               // the cylinders: every third candidate of each agent, from its (wave - 1)-th
        const bool from_cand = will_reset && !fallback;
        const double px = from_cand ? C->cand[0][lane] : C->pos[lane][0];
        const double py = from_cand ? C->cand[1][lane] : C->pos[lane][1];
        const double pz = from_cand ? C->cand[2][lane] : C->pos[lane][2];
        ObsBits part = obs_cand_queued<MaskT>(w, &C->q[wv - 1], px, py, pz, active, wv - 1);
        // + the out-of-box part of one stencil each
        const uint32_t box = obs_box_bits(w.W, w.Hbox, wv == 1 ? 1.0 : (wv == 2 ? 5.0 : 10.0), px, py, pz);
        if (wv == 1) part.s1 |= box;
        else if (wv == 2) part.s5 |= box;
        else part.s10 |= box;
        uint32_t m0, m1, m2;
        ctile_mask_words(part, m0, m1, m2);
        if (m0) atomicOr(&trow[0], m0);
        if (m1) atomicOr(&trow[1], m1);
        if (m2) atomicOr(&trow[2], m2);
    }
    UAV_CSTAMP(5);
    __syncthreads();                                                         // tile complete
    UAV_CSTAMP(6);
    if (want_obs && OBS == OBS_KIND_PACKED) {
        // packed rows: the 64 x 80 B image of the tile is 320 16-byte chunks, 5 KiB contiguous: two store instructions
        const int nv = N - first < 64 ? N - first : 64;
        uint4 *dst = reinterpret_cast<uint4 *>(reinterpret_cast<uint32_t *>(a.obs) + (int64_t)first * kPackedDwords);
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int chunk = it * 256 + (int)threadIdx.x;
            if (chunk < 5 * nv) dst[chunk] = ctile_packed_chunk(C->tile, chunk);
        }
    } else if (want_obs) {
        constexpr bool F16 = OBS == OBS_KIND_F16;
        // 25 coalesced store instructions: 7 for wave 0, 6 each for waves 1..3 (fixed trip counts: unrolled, so the
        // LDS reads of one instruction overlap the selects of the previous)
        const int nv = N - first;
        const int lo = wv == 0 ? 0 : 7 + (wv - 1) * 6;
        if (nv >= 64) {                     // workgroup-uniform: all but the last workgroup take the unguarded form
#pragma unroll
            for (int j = 0; j < 6; ++j) ctile_emit_lut<F16>(a.obs, first, nv, C->tile, lo + j, lut[j], false);
            if (wv == 0) ctile_emit_lut<F16>(a.obs, first, nv, C->tile, 6, lut[6], false);
        } else {
#pragma unroll
            for (int j = 0; j < 6; ++j) ctile_emit_lut<F16>(a.obs, first, nv, C->tile, lo + j, lut[j], true);
            if (wv == 0) ctile_emit_lut<F16>(a.obs, first, nv, C->tile, 6, lut[6], true);
        }
    }
    UAV_CSTAMP(7);
}

// state_PathPlan only
template <typename MaskT, int OBS>
__global__ void __launch_bounds__(256) k_observe(StepArgs a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const WorldLds<MaskT> w = stage_world<MaskT>(smem, a);
    for (int i = blockIdx.x * a.block + threadIdx.x; i < a.N; i += gridDim.x * a.block) {
        Agent g;
        load_agent(a.st, i, g);
        unpack_flags(g);
        const ObsBits bits = obs_bits(w, g.o.px, g.o.py, g.o.pz);      // lane-per-agent form (reference for the queue)
        const ObsScalars sc = obs_scalars(g.o, g.head);
        store_obs_row<OBS>(a.obs, i, sc, bits);
    }
}

// BaseClass/CalMod.py:89-102 calculate_angle and :64-65 Eu_Loc_distance for n point pairs, with the device functions the
// step kernels use (calc_angle / dist3): the direct known-answer check of the geometry layer.
__global__ void k_geometry(const double *__restrict__ ab, double *__restrict__ angle_out, double *__restrict__ dist_out, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double ax = ab[6 * i], ay = ab[6 * i + 1], az = ab[6 * i + 2], bx = ab[6 * i + 3], by = ab[6 * i + 4], bz = ab[6 * i + 5];
    if (angle_out) angle_out[i] = calc_angle(bx - ax, by - ay);
    if (dist_out) dist_out[i] = dist3(ax, ay, az, bx, by, bz);
}

template <typename MaskT, bool ALLPAIRS>
__global__ void k_threaten(StepArgs a, const double *__restrict__ xyz, uint8_t *__restrict__ out, int64_t n)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const WorldLds<MaskT> w = stage_world<MaskT>(smem, a);
    for (int64_t i = (int64_t)blockIdx.x * a.block + threadIdx.x; i < n; i += (int64_t)gridDim.x * a.block) {
        double x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
        out[i] = (uint8_t)(ALLPAIRS ? probe_allpairs(w.b, a.nb, a.W, a.Hbox, x, y, z) : probe(w, x, y, z));
    }
}

// reset every agent from the bank
template <bool APF>
__global__ void k_reset_all(StepArgs a)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < a.N; i += gridDim.x * blockDim.x) {
        Agent g;
        reset_agent<APF>(a, i, g);
        store_agent(a.st, i, g);
    }
}

// Detach every agent from the scenario bank (copy its list into the private slot) -- run before a bank is replaced.
__global__ void k_privatize(StepArgs a)
{
    const DevState &S = a.st;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < a.N; i += gridDim.x * blockDim.x) {
        const int scn = S.I(I_SCN)[i];
        if (scn < 0) continue;
        const double *src = a.bank.sub + (size_t)scn * a.K * 3;
        double *dst = S.sub + (size_t)i * a.K * 3;
        const int n = S.I(I_NTOTAL)[i];
        for (int q = 0; q < n * 3; ++q) dst[q] = src[q];
        S.I(I_SCN)[i] = -1;
    }
}

// parity injection: scatter host-provided AoS rows into the SoA state
__global__ void k_set_state(StepArgs a, int first, int count, const double *__restrict__ kin,
                            const int32_t *__restrict__ step, const int32_t *__restrict__ n_sub,
                            const int32_t *__restrict__ alias, const double *__restrict__ sub)
{
    const DevState &S = a.st;
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < count; c += gridDim.x * blockDim.x) {
        const int i = first + c;
        const double *k = kin + (size_t)c * 8;
        Agent g;
        ObsIn &o = g.o;
        o.px = k[0]; o.py = k[1]; o.pz = k[2];
        o.vx = k[3]; o.vy = k[4];
        o.V = calc_v(o.vx, o.vy, a.max_v);              // inject() does uav.V = uav.Calc_V()
        g.head = calc_angle(o.vx, o.vy);
        o.gx = k[5]; o.gy = k[6]; o.gz = k[7];
        const int n = n_sub[c];
        const double *src = sub + (size_t)c * a.K * 3;
        double *dst = S.sub + (size_t)i * a.K * 3;
        for (int q = 0; q < n * 3; ++q) dst[q] = src[q];
        o.s0x = n >= 1 ? src[0] : 0.0; o.s0y = n >= 1 ? src[1] : 0.0; o.s0z = n >= 1 ? src[2] : 0.0;
        o.s1x = n >= 2 ? src[3] : 0.0; o.s1y = n >= 2 ? src[4] : 0.0; o.s1z = n >= 2 ? src[5] : 0.0;
        o.step = step[c];
        g.sub_idx = 0; g.n_total = n; g.scn = -1;
        g.done = 0; g.alias = alias ? alias[c] : 0; g.reach = 0;
        g.score = 0.0; g.total = 0.0; g.path_len = 0.0; g.epoch = 0;
        store_agent(S, i, g);
    }
}

__global__ void k_get_state(StepArgs a, int first, int count, double *__restrict__ out16, double *__restrict__ out_sub,
                            int32_t *__restrict__ out_alias)
{
    const DevState &S = a.st;
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < count; c += gridDim.x * blockDim.x) {
        const int i = first + c;
        double *o = out16 + (size_t)c * 16;
        const int sub_idx = S.I(I_SUBIDX)[i], n_total = S.I(I_NTOTAL)[i];
        o[0] = S.F(F_PX)[i]; o[1] = S.F(F_PY)[i]; o[2] = S.F(F_PZ)[i]; o[3] = S.F(F_VX)[i]; o[4] = S.F(F_VY)[i]; o[5] = S.F(F_V)[i];
        o[6] = S.F(F_GX)[i]; o[7] = S.F(F_GY)[i]; o[8] = S.F(F_GZ)[i]; o[9] = (double)S.I(I_STEP)[i]; o[10] = (double)((S.I(I_FLAGS)[i] & kFlagDone) ? 1 : 0);
        o[11] = (double)(n_total - sub_idx); o[12] = S.F(F_SCORE)[i]; o[13] = S.F(F_TOTAL)[i]; o[14] = S.F(F_PATHLEN)[i];
        o[15] = (double)((S.I(I_FLAGS)[i] & kFlagReach) ? 1 : 0);
        if (out_alias) out_alias[c] = (S.I(I_FLAGS)[i] & kFlagAlias) ? 1 : 0;
        if (out_sub) {
            double *d = out_sub + (size_t)c * a.K * 3;
            const double *src = list_of(a, i, S.I(I_SCN)[i]);
            for (int k = 0; k < a.K; ++k) {
                const int q = sub_idx + k;
                const bool ok = q < n_total;
                // the hot window is authoritative for the current sub-goal (it tracks the position alias)
                d[k * 3] = ok ? (k == 0 ? S.F(F_S0X)[i] : src[q * 3]) : 0.0;
                d[k * 3 + 1] = ok ? (k == 0 ? S.F(F_S0Y)[i] : src[q * 3 + 1]) : 0.0;
                d[k * 3 + 2] = ok ? (k == 0 ? S.F(F_S0Z)[i] : src[q * 3 + 2]) : 0.0;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static bool is_pow2(int x) { return x > 0 && (x & (x - 1)) == 0; }

static StepArgs base_args(const UavEnv *e)
{
    StepArgs a;
    memset(&a, 0, sizeof(a));
    a.st = e->st;
    a.world_blob = e->world_blob;
    a.world_bytes = e->world_bytes;
    a.aux_off = e->aux_off;
    a.grid_off = e->grid_off;
    a.grid_stride = e->grid_stride;
    a.nb = e->nb;
    a.gn = e->gn;
    a.inv_cell = 1.0 / e->cell;
    a.W = e->cfg.width;
    a.Hbox = e->cfg.h;
    a.apf_b = e->apf_b;
    a.apf_grid = e->apf_grid;
    a.apf_all = e->apf_all;
    a.apf_gn = e->apf_gn;
    a.apf_nz = e->apf_nz;
    a.apf_inv_cell = 1.0 / e->apf_cell;
    a.apf_inv_cz = 1.0 / e->apf_cz;
    a.emit_lut = e->emit_lut;
    a.max_v = e->cfg.max_v;
    a.steer = e->cfg.steering_angle;
    a.pw = PowerParams{e->cfg.power[0], e->cfg.power[1], e->cfg.power[2], e->cfg.power[3],
                       e->cfg.power[4], e->cfg.power[5], e->cfg.power[6], e->cfg.power[7]};
    a.max_step = e->cfg.max_step;
    a.K = e->cfg.max_subgoals;
    a.U = e->cfg.uav_per_env;
    a.N = e->N;
    a.n_actions = e->cfg.n_actions;
    a.bank.start_goal = e->bank_sg;
    a.bank.sub = e->bank_sub;
    a.bank.nsub = e->bank_nsub;
    a.bank.m = e->bank_m;
    a.seed = e->seed;
    a.tick = e->tick;
    a.moved_word = e->moved_word;
    a.moved_value = (uint32_t)(e->tick + 1);
    a.meta = nullptr;                  // (the step entry points hand the pending records over: take_step_meta)
    a.meta_a1 = nullptr;
    a.dbg = e->dbg;
    return a;
}

// Launch geometry.  Every k_step launch is "one agent per thread" (straight-line body, 128 VGPRs):
//  * N <= 131 072: single-wavefront workgroups, so the wavefronts spread over all 256 CUs;
//  * beyond that: 256-thread workgroups (the ~10 KB world blob is staged once per 256 agents, not once per 64).
// Observation rows: small launches (N <= 49 152) go to k_step_coop; above that the wavefront's 64 rows go out through the compact LDS tile (store_obs_ctile), which shares its bytes
// with the by-then-dead observation work queue: ~16 KB per single-wave workgroup, ~34 KB per 256-thread one, so
// LDS allows as many wavefronts per CU as the VGPR budget does (16).
// MEASURED (round 1, us per launch, row-per-lane -> compact tile): 16 384 envs 12.6 -> 14.7 and 32 768: 15.4 -> 16.2
// (kept row-per-lane); 49 152: 18.2 -> 16.5; 65 536: 21.0 -> 17.0; 131 072: 32.5 -> 22.8; 262 144: 55.1 -> 36.6;
// 524 288: 117 -> 74; 1 M: 260 -> 174.  The earlier design (grid-stride
// loop, 256 VGPRs, full 25.9 KB f32 tile: one workgroup per CU) took 65 us at 262 144 and 252 us at 1 M.
// UAVENV_BLOCK / UAVENV_BLOCK64_MAX / UAVENV_TILE_STORE are A/B knobs.
static int env_int(const char *name, int dflt)
{
    const char *v = getenv(name);
    return v ? atoi(v) : dflt;
}

// Launch with `lds` bytes of dynamic LDS.  Above the 64 KB default a kernel needs its
// hipFuncAttributeMaxDynamicSharedMemorySize raised first (a fine broad-phase grid makes the world blob that large);
// the attribute is raised whenever a launch asks for more than any earlier launch of the same kernel on this device.
template <typename K>
static void launch_lds(K kernel, int grid, int block, size_t lds, hipStream_t s, const StepArgs &a)
{
    if (lds > 65536) {
        static std::mutex mu;
        static std::map<std::pair<const void *, int>, size_t> raised;
        int dev = 0;
        (void)hipGetDevice(&dev);
        std::lock_guard<std::mutex> lock(mu);
        size_t &cur = raised[std::make_pair(reinterpret_cast<const void *>(kernel), dev)];
        if (lds > cur) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            cur = lds;
        }
    }
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), lds, s, a);
}

static void launch_geometry(int n, int &block, int &grid)
{
    static const int thr = env_int("UAVENV_BLOCK64_MAX", 131072);
    static const int big = env_int("UAVENV_BLOCK", 256);
    block = n <= thr ? 64 : big;
    grid = (n + block - 1) / block;
    if (grid < 1) grid = 1;
}

// the records asked for by uavenv_set_step_meta go to THIS launch and no further (a stale frame pointer must never be written again)
struct PendingMeta {
    void *meta;
    const float *a1;
};
static PendingMeta take_step_meta(UavEnv *e)
{
    PendingMeta p{e->step_meta, e->step_meta_a1};
    e->step_meta = nullptr;
    e->step_meta_a1 = nullptr;
    return p;
}

template <typename MaskT>
static void launch_step(const UavEnv *e, const StepArgs &a_in, hipStream_t s)
{
    int block, grid;
    launch_geometry(e->N, block, grid);
    StepArgs a = a_in;
    a.block = block;
    const int obs = e->cfg.obs_dtype;
    const bool apf = e->cfg.apf_enabled == 1;
    static const int tile_env = env_int("UAVENV_TILE_STORE", -1);
    // the wave-cooperative tile store pays once a SIMD holds several wavefronts (MEASURED, round 3, us per launch with /
    // without it -- packed rows: 65 536 agents 13.8 / 12.7, 131 072: 16.9 / 16.4, 262 144: 25.8 / 26.4; f16 rows: 65 536:
    // 16.2 / 15.0, 131 072: 21.2 / 22.0, 262 144: 34.7 / 38.3; f32 rows keep the round-1 threshold)
    const int tile_from = obs == UAVENV_OBS_PACKED ? 196608 : (obs == UAVENV_OBS_F16 ? 131072 : 32769);
    const bool tile_store = tile_env >= 0 ? tile_env != 0 : e->N >= tile_from;
    const int nw = block / 64;
    a.obsq_off = (e->world_bytes + 15) & ~15;
    int slot = (int)sizeof(ObsWaveLds);            // per-wave LDS slot: the work queue, then (same bytes) the tile
    if (tile_store && kCTileBytes > slot) slot = kCTileBytes;
    slot = (slot + 15) & ~15;
    a.wave_slot = slot;
    a.tile_off = tile_store ? a.obsq_off : -1;
    const size_t lds = (size_t)a.obsq_off + (size_t)nw * slot;
    // <= 49 152 agents (three wavefronts' worth of agents per CU): the cooperative four-wavefronts-per-64-agents kernel
    // (see k_step_coop).  MEASURED (us per launch, k_step -> k_step_coop): 4 096 envs 10.6 -> 7.8; 16 384: 11.7 -> 8.7;
    // 32 768: 14.6 -> 10.1; 49 152: 16.1 -> 12.3; 65 536: 16.6 -> 19.5 (k_step kept from there on).
    static const int coop_env = env_int("UAVENV_COOP", -1);
    // (APF on: k_apf_adjust + k_step at every size -- the cooperative kernel adjusts the sub-goals per lane: 105 / 119 us
    // against 35 / 41 us for 8 192 / 32 768 agents -- unless the per-lane path is asked for)
    const bool apf_split_ok = apf && !(a.flags & UAVENV_STEP_APF_LANE) && kApfAgentsPerBlock * a.K <= 65536;
    const bool coop = !(a.flags & UAVENV_STEP_ONE_WAVE) && (coop_env >= 0 ? coop_env != 0 : (e->N <= 49152 && !apf_split_ok));
#define UAV_LAUNCH(KERNEL, GRID, BLOCK, LDS)                                                                        \
    do {                                                                                                            \
        if (apf) {                                                                                                  \
            if (obs == UAVENV_OBS_F16) launch_lds((KERNEL<MaskT, true, OBS_KIND_F16>), GRID, BLOCK, LDS, s, a);     \
            else if (obs == UAVENV_OBS_PACKED) launch_lds((KERNEL<MaskT, true, OBS_KIND_PACKED>), GRID, BLOCK, LDS, s, a); \
            else launch_lds((KERNEL<MaskT, true, OBS_KIND_F32>), GRID, BLOCK, LDS, s, a);                           \
        } else {                                                                                                    \
            if (obs == UAVENV_OBS_F16) launch_lds((KERNEL<MaskT, false, OBS_KIND_F16>), GRID, BLOCK, LDS, s, a);    \
            else if (obs == UAVENV_OBS_PACKED) launch_lds((KERNEL<MaskT, false, OBS_KIND_PACKED>), GRID, BLOCK, LDS, s, a); \
            else launch_lds((KERNEL<MaskT, false, OBS_KIND_F32>), GRID, BLOCK, LDS, s, a);                          \
        }                                                                                                           \
    } while (0)
    if (coop) {
        a.block = 256;
        a.obsq_off = (e->world_bytes + 15) & ~15;
        const size_t clds = (size_t)a.obsq_off + sizeof(CoopLds);
        const int cgrid = (e->N + 63) / 64;
        if (a.pol_local) {               // uavenv_step_policy (validated there: packed rows, APF off)
            a.pol_off = (int32_t)((clds + 15) & ~(size_t)15);
            const size_t plds = (size_t)a.pol_off + (size_t)(uavq::kTileF + uavq::kMaxOut * uavq::kHid + uavq::kMaxOut + 64) * 4;
            // (up to one workgroup per CU the register-hungry form of the forward; beyond, two or three workgroups share a CU)
            if (cgrid <= 256) launch_lds((k_step_coop<MaskT, false, OBS_KIND_PACKED, true, true>), cgrid, 256, plds, s, a);
            else launch_lds((k_step_coop<MaskT, false, OBS_KIND_PACKED, true, false>), cgrid, 256, plds, s, a);
            return;
        }
        UAV_LAUNCH(k_step_coop, cgrid, 256, clds);
        return;
    }
    if (a.pol_local) {                   // uavenv_step_policy on an f16 ring (validated there: APF off, f16 MFMA net, block 64)
        if (slot < uavq::kPolBytes) slot = uavq::kPolBytes;               // the policy team's tiles, then the agent wavefront's slot
        a.wave_slot = slot;
        launch_lds(k_step_polh<MaskT>, grid, 192, (size_t)a.obsq_off + (size_t)slot, s, a);
        return;
    }
    if (apf_split_ok) {                                                   // the lists first, in their own kernel (16-bit offsets)
        a.apf_split = 1;
        const int agents_per_block = kApfAgentsPerBlock;
        const size_t albs = ((size_t)((e->nb > 0 ? e->nb : 1) * sizeof(BldApf) + 31) & ~(size_t)31) + 4 * sizeof(ApfSortLds);
        launch_lds(k_apf_adjust, (e->N + agents_per_block - 1) / agents_per_block, 256, albs, s, a);
    }
    UAV_LAUNCH(k_step, grid, block, lds);
#undef UAV_LAUNCH
}

extern "C" {

int uavenv_abi_version(void) { return UAVENV_ABI_VERSION; }
const char *uavenv_last_error(void) { return g_err; }

int uavenv_create(const UavEnvConfig *cfg, UavEnv **out)
{
    if (!cfg || !out) return fail(UAVENV_EINVAL, "null argument");
    if (cfg->abi_version != UAVENV_ABI_VERSION)
        return fail(UAVENV_EINVAL, "abi_version %d != %d", cfg->abi_version, UAVENV_ABI_VERSION);
    if (cfg->n_envs <= 0 || cfg->max_subgoals < 2 || cfg->max_step <= 0)
        return fail(UAVENV_EINVAL, "n_envs/max_subgoals/max_step out of range");
    if (!is_pow2(cfg->uav_per_env) || cfg->uav_per_env > 64)
        return fail(UAVENV_EINVAL, "uav_per_env must be a power of two <= 64 (got %d)", cfg->uav_per_env);
    if (cfg->obs_dtype != UAVENV_OBS_F32 && cfg->obs_dtype != UAVENV_OBS_F16 && cfg->obs_dtype != UAVENV_OBS_PACKED)
        return fail(UAVENV_EINVAL, "obs_dtype");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(UAVENV_ENODEV, "no HIP device visible: libuavenv has no CPU fallback");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(UAVENV_EINVAL, "device %d of %d", cfg->device, ndev);
    HIP_TRY(hipSetDevice(cfg->device));
    UavEnv *e = new (std::nothrow) UavEnv();
    if (!e) return fail(UAVENV_ENOMEM, "host alloc");
    e->cfg = *cfg;
    if (e->cfg.n_actions < 2) e->cfg.n_actions = 3;
    e->cell = cfg->cell_size > 0 ? cfg->cell_size : 20.0;
    const long long n64 = (long long)cfg->n_envs * cfg->uav_per_env;
    if (n64 > (1ll << 30)) { delete e; return fail(UAVENV_EINVAL, "too many agents"); }
    e->N = (int)n64;
    const size_t n = (size_t)e->N, npad = (n + 63) & ~(size_t)63;
    const size_t K = (size_t)cfg->max_subgoals;
    size_t bytes = npad * 8 * kNumF64 + npad * 4 * kNumI32 + n * K * 3 * 8 + 256;
    hipError_t er = hipMalloc(&e->slab, bytes);
    if (er != hipSuccess) { delete e; return fail(UAVENV_ENOMEM, "hipMalloc(%zu): %s", bytes, hipGetErrorString(er)); }
    e->slab_bytes = bytes;
    (void)hipMemset(e->slab, 0, bytes);
    unsigned char *p = (unsigned char *)e->slab;
    e->st.npad = npad;
    e->st.f64 = (double *)p; p += npad * 8 * kNumF64;
    e->st.sub = (double *)p; p += n * K * 3 * 8;
    p = (unsigned char *)(((uintptr_t)p + 63) & ~(uintptr_t)63);
    e->st.i32 = (int32_t *)p; p += npad * 4 * kNumI32;
    (void)hipMemset(e->st.I(I_SCN), 0xff, npad * 4);   // scn = -1: private (empty) list
    {
        std::vector<uint2> lut(25 * 64);
        for (int it = 0; it < 25; ++it)
            for (int l = 0; l < 64; ++l) lut[(size_t)it * 64 + l] = ctile_emit_lut_entry(it, l);
        if (hipMalloc((void **)&e->emit_lut, lut.size() * sizeof(uint2)) != hipSuccess ||
            hipMemcpy(e->emit_lut, lut.data(), lut.size() * sizeof(uint2), hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipFree(e->slab);
            delete e;
            return fail(UAVENV_ENOMEM, "emit table");
        }
    }
    *out = e;
    return UAVENV_OK;
}

int uavenv_destroy(UavEnv *e)
{
    if (!e) return UAVENV_OK;
    (void)hipSetDevice(e->cfg.device);
    (void)hipFree(e->slab);
    (void)hipFree(e->world_blob);
    (void)hipFree(e->apf_b);
    (void)hipFree(e->apf_grid);
    (void)hipFree(e->emit_lut);
    (void)hipFree(e->bank_sg);
    (void)hipFree(e->bank_sub);
    (void)hipFree(e->bank_nsub);
    (void)hipFree(e->rp_sg);
    (void)hipFree(e->rp_sub);
    (void)hipFree(e->rp_nsub);
    (void)hipFree(e->rp_nodes);
    (void)hipFree(e->rp_counters);
    if (e->rp_done) (void)hipEventDestroy(e->rp_done);
    if (e->rp_committed) (void)hipEventDestroy(e->rp_committed);
    delete e;
    return UAVENV_OK;
}

int uavenv_num_agents(const UavEnv *e) { return e ? e->N : UAVENV_EINVAL; }

int uavenv_set_buildings(UavEnv *e, const double *b5, const double *v3, int32_t nb)
{
    if (!e || (nb > 0 && !b5)) return fail(UAVENV_EINVAL, "null argument");
    if (nb < 0 || nb > UAVENV_MAX_BUILDINGS)
        return fail(UAVENV_EINVAL, "nb=%d exceeds UAVENV_MAX_BUILDINGS=%d", nb, UAVENV_MAX_BUILDINGS);
    HIP_TRY(hipSetDevice(e->cfg.device));
    const double W = e->cfg.width, cell = e->cell;
    const int gn = (int)std::ceil(W / cell) > 0 ? (int)std::ceil(W / cell) : 1;
    const int mask_bytes = nb <= 32 ? 4 : 8;
    std::vector<BldLds> bl((size_t)(nb > 0 ? nb : 1));
    std::vector<BldApf> ba((size_t)(nb > 0 ? nb : 1));
    for (int i = 0; i < nb; ++i) {
        const double cx = b5[5 * i], cy = b5[5 * i + 1], cz = b5[5 * i + 2], R = b5[5 * i + 3], H = b5[5 * i + 4];
        // thr = min{ t : sqrt_rn(t) >= R }  =>  (s < thr) <=> (sqrt_rn(s) < R)
        double t = R * R;
        if (R > 0) {
            while (std::sqrt(t) >= R) t = std::nextafter(t, -INFINITY);
            while (std::sqrt(t) < R) t = std::nextafter(t, INFINITY);
        } else {
            t = 0.0;   // sqrt(s) < R<=0 is never true for s >= 0 (R<0) / s<0 impossible
        }
        bl[i] = BldLds{cx, cy, t, H};
        double vx = v3 ? v3[3 * i] : 0.0, vy = v3 ? v3[3 * i + 1] : 0.0, vz = v3 ? v3[3 * i + 2] : 0.0;
        // direction of motion = (cos, sin)(calculate_angle(0, v)) (UAV.py:192,201): a per-building constant
        const bool moving = !(vx == 0.0 && vy == 0.0 && vz == 0.0);
        double ang = std::atan2(vy, vx) * (180.0 / M_PI);
        ang = std::fmod(ang + 360.0, 360.0) / 180.0 * M_PI;
        const double far = R + 60.0 + 1e-6;
        ba[i] = BldApf{cx, cy, cz, R, vx, vy, vz, std::sqrt(vx * vx + vy * vy + vz * vz), far * far, std::cos(ang),
                       std::sin(ang), moving ? 1.0 : 0.0};
    }
    // conservative rasterisation: for halo h the cell rectangle is grown by h (+ margin) on every side and the
    // radius by a margin, so rounding of the cell index (x * inv_cell) or of the distance can never drop a
    // cylinder that the exact test would hit for any point within L-inf distance h of the cell.
    static const double kHalo[3] = {2.0, 10.0, 20.0};
    const double margin = 1e-6 * (cell > 1.0 ? cell : 1.0);
    const int bld_only = (int)(((size_t)(nb > 0 ? nb : 1) * sizeof(BldLds) + 15) & ~(size_t)15);
    const int aux_bytes = (int)(((size_t)(nb > 0 ? nb : 1) * sizeof(BldAux) + 15) & ~(size_t)15);
    const int bld_bytes = bld_only + aux_bytes;            // [BldLds x nb][BldAux x nb] then the three grids
    const int grid_bytes = (int)((((size_t)gn * gn * mask_bytes) + 15) & ~(size_t)15);
    std::vector<unsigned char> blob((size_t)bld_bytes + 3 * (size_t)grid_bytes, 0);
    memcpy(blob.data(), bl.data(), (size_t)(nb > 0 ? nb : 0) * sizeof(BldLds));
    {
        static const double kSpacing[3] = {1.0, 5.0, 10.0};
        BldAux *aux = reinterpret_cast<BldAux *>(blob.data() + bld_only);
        for (int i = 0; i < nb; ++i)
            for (int k = 0; k < 3; ++k) {
                const double R = b5[5 * i + 3], diag = 2.0 * std::sqrt(2.0) * kSpacing[k], guard = 1e-6;
                const double ro = R + diag + guard, ri = R - diag - guard;
                aux[i].rej2[k] = ro > 0 ? ro * ro : 0.0;
                aux[i].acc2[k] = ri > 0 ? ri * ri : -1.0;      // -1: never "entirely inside"
            }
    }
    for (int h = 0; h < 3; ++h) {
        unsigned char *gdst = blob.data() + bld_bytes + (size_t)h * grid_bytes;
        for (int iy = 0; iy < gn; ++iy)
            for (int ix = 0; ix < gn; ++ix) {
                const double x0 = ix * cell - kHalo[h] - margin, x1 = (ix + 1) * cell + kHalo[h] + margin;
                const double y0 = iy * cell - kHalo[h] - margin, y1 = (iy + 1) * cell + kHalo[h] + margin;
                uint64_t m = 0;
                for (int i = 0; i < nb; ++i) {
                    const double qx = bl[i].cx < x0 ? x0 : (bl[i].cx > x1 ? x1 : bl[i].cx);
                    const double qy = bl[i].cy < y0 ? y0 : (bl[i].cy > y1 ? y1 : bl[i].cy);
                    const double d = std::hypot(qx - bl[i].cx, qy - bl[i].cy);
                    if (d < b5[5 * i + 3] + margin) m |= (1ull << i);
                }
                const size_t k = (size_t)iy * gn + ix;
                if (mask_bytes == 8) reinterpret_cast<uint64_t *>(gdst)[k] = m;
                else reinterpret_cast<uint32_t *>(gdst)[k] = (uint32_t)m;
            }
    }
    if (blob.size() > 150 * 1024) return fail(UAVENV_EINVAL, "world blob %zu B does not fit LDS; raise cell_size", blob.size());
    (void)hipFree(e->world_blob);
    (void)hipFree(e->apf_b);
    e->world_blob = nullptr;
    e->apf_b = nullptr;
    HIP_TRY(hipMalloc((void **)&e->world_blob, blob.size()));
    HIP_TRY(hipMemcpy(e->world_blob, blob.data(), blob.size(), hipMemcpyHostToDevice));
    HIP_TRY(hipMalloc((void **)&e->apf_b, ba.size() * sizeof(BldApf)));
    HIP_TRY(hipMemcpy(e->apf_b, ba.data(), ba.size() * sizeof(BldApf), hipMemcpyHostToDevice));
    {   // APF broad phase: per cell, the moving cylinders whose force range (60 m beyond the rim, measured from the
        // CENTRE point (cx, cy, cz) in 3-D: UAV.py:183) can reach the cell.  Its own table, 3-D and finer than the
        // collision grids (it stays in global memory): the per-point loop is over the mask's bits and the lanes of a
        // wavefront wait for the longest one, so every spurious candidate costs.  Same conservative rasterisation
        // (minimum distance from the centre to the grown cell box, 1e-6-scale margins).
        double acell = 10.0, acz = 10.0;
        const double Hbox = e->cfg.h > 0 ? e->cfg.h : 1.0;
        auto cells = [&](double c, double cz) { return std::ceil(W / c) * std::ceil(W / c) * std::ceil(Hbox / cz); };
        while (cells(acell, acz) > 1048576.0) { acell *= 2.0; acz *= 2.0; }
        const int agn = std::max(1, (int)std::ceil(W / acell)), anz = std::max(1, (int)std::ceil(Hbox / acz));
        const double am = 1e-6 * (acell > 1.0 ? acell : 1.0);
        std::vector<uint64_t> ag((size_t)anz * agn * agn, 0);
        for (int iz = 0; iz < anz; ++iz)
            for (int iy = 0; iy < agn; ++iy)
                for (int ix = 0; ix < agn; ++ix) {
                    const double x0 = ix * acell - am, x1 = (ix + 1) * acell + am;
                    const double y0 = iy * acell - am, y1 = (iy + 1) * acell + am;
                    const double z0 = iz * acz - am, z1 = (iz + 1) * acz + am;
                    uint64_t m = 0;
                    for (int i = 0; i < nb; ++i) {
                        if (ba[i].moving == 0.0) continue;
                        const double qx = ba[i].cx < x0 ? x0 : (ba[i].cx > x1 ? x1 : ba[i].cx);
                        const double qy = ba[i].cy < y0 ? y0 : (ba[i].cy > y1 ? y1 : ba[i].cy);
                        const double qz = ba[i].cz < z0 ? z0 : (ba[i].cz > z1 ? z1 : ba[i].cz);
                        const double dx = qx - ba[i].cx, dy = qy - ba[i].cy, dz = qz - ba[i].cz;
                        if (std::sqrt(dx * dx + dy * dy + dz * dz) < ba[i].R + 60.0 + 1e-6 + am) m |= (1ull << i);
                    }
                    ag[((size_t)iz * agn + iy) * agn + ix] = m;
                }
        (void)hipFree(e->apf_grid);
        e->apf_grid = nullptr;
        e->apf_all = 0;
        for (int i = 0; i < nb; ++i)
            if (ba[i].moving != 0.0) e->apf_all |= 1ull << i;
        HIP_TRY(hipMalloc((void **)&e->apf_grid, ag.size() * 8));
        HIP_TRY(hipMemcpy(e->apf_grid, ag.data(), ag.size() * 8, hipMemcpyHostToDevice));
        e->apf_gn = agn; e->apf_nz = anz; e->apf_cell = acell; e->apf_cz = acz;
    }
    e->world_bytes = (int)blob.size();
    e->aux_off = bld_only;
    e->grid_off = bld_bytes;
    e->grid_stride = grid_bytes;
    e->nb = nb;
    e->gn = gn;
    e->mask_bytes = mask_bytes;
    e->have_world = true;
    e->world_gen += 1;                 // a refresh planned for the old world is dropped at commit
    return UAVENV_OK;
}

int uavenv_load_scenarios(UavEnv *e, const double *sg, const double *sub, const int32_t *nsub, int32_t m)
{
    if (!e || !sg || !sub || !nsub || m <= 0) return fail(UAVENV_EINVAL, "null/empty scenario bank");
    const int K = e->cfg.max_subgoals;
    for (int i = 0; i < m; ++i)
        if (nsub[i] < 0 || nsub[i] > K) return fail(UAVENV_EINVAL, "scenario %d has %d sub-goals > K=%d", i, nsub[i], K);
    HIP_TRY(hipSetDevice(e->cfg.device));
    if (e->bank_m > 0) {      // agents may still point into the old bank: give them private copies first
        StepArgs a = base_args(e);
        HIP_TRY(hipDeviceSynchronize());
        hipLaunchKernelGGL(k_privatize, dim3((e->N + 255) / 256), dim3(256), 0, 0, a);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipDeviceSynchronize());
    }
    (void)hipFree(e->bank_sg); (void)hipFree(e->bank_sub); (void)hipFree(e->bank_nsub);
    e->bank_sg = e->bank_sub = nullptr; e->bank_nsub = nullptr; e->bank_m = 0;
    HIP_TRY(hipMalloc((void **)&e->bank_sg, (size_t)m * 6 * 8));
    HIP_TRY(hipMalloc((void **)&e->bank_sub, (size_t)m * K * 3 * 8));
    HIP_TRY(hipMalloc((void **)&e->bank_nsub, (size_t)m * 4));
    HIP_TRY(hipMemcpy(e->bank_sg, sg, (size_t)m * 6 * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(e->bank_sub, sub, (size_t)m * K * 3 * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(e->bank_nsub, nsub, (size_t)m * 4, hipMemcpyHostToDevice));
    e->bank_m = m;
    e->bank_replaced = 0;
    return UAVENV_OK;
}

// internal: lets rrt.hip stage the same world blob
int uavenv__world_view(const UavEnv *e, const unsigned char **blob, int32_t *bytes, int32_t *aux_off, int32_t *grid_off, int32_t *grid_stride,
                       int32_t *gn, double *inv_cell, double *W, double *Hbox, double *len, int32_t *mask_bytes, int32_t *K)
{
    if (!e || !e->have_world) return fail(UAVENV_EINVAL, "planner before uavenv_set_buildings");
    *blob = e->world_blob; *bytes = e->world_bytes; *aux_off = e->aux_off; *grid_off = e->grid_off; *grid_stride = e->grid_stride;
    *gn = e->gn; *inv_cell = 1.0 / e->cell; *W = e->cfg.width; *Hbox = e->cfg.h; *len = e->cfg.len;
    *mask_bytes = e->mask_bytes; *K = e->cfg.max_subgoals;
    return UAVENV_OK;
}

// Scenarios the planner could not fit (n_sub outside [2, K]) take the next valid scenario's row.
// counters[0]: scenarios without a usable path (planner gave up, or the path needs more than K slots)
__global__ void k_bank_fix(double *sg, double *sub, int32_t *nsub, int m, int K, int32_t *counters)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) {
        const int n = nsub[i];
        if (n >= 2 && n <= K) continue;
        atomicAdd(&counters[0], 1);
        int j = -1;
        for (int d = 1; d < m; ++d) {
            const int c = (i + d) % m;
            const int nc = nsub[c];
            if (nc >= 2 && nc <= K) { j = c; break; }      // valid rows are never written by this kernel
        }
        if (j < 0) continue;
        for (int q = 0; q < 6; ++q) sg[(size_t)i * 6 + q] = sg[(size_t)j * 6 + q];
        for (int q = 0; q < K * 3; ++q) sub[(size_t)i * K * 3 + q] = sub[(size_t)j * K * 3 + q];
        nsub[i] = -1000000 - j;      // marker; resolved below
    }
}
__global__ void k_bank_fix2(int32_t *nsub, int m)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) {
        const int n = nsub[i];
        if (n <= -1000000) nsub[i] = nsub[-1000000 - n];
    }
}

int uavenv_plan_scenarios(UavEnv *e, int32_t m, uint64_t seed, int32_t max_iter, void *stream)
{
    if (!e || m <= 0 || max_iter <= 0) return fail(UAVENV_EINVAL, "uavenv_plan_scenarios: bad argument");
    if (!e->have_world) return fail(UAVENV_EINVAL, "uavenv_plan_scenarios before uavenv_set_buildings");
    HIP_TRY(hipSetDevice(e->cfg.device));
    const int K = e->cfg.max_subgoals;
    if (e->bank_m > 0) {      // agents may still point into the old bank
        StepArgs a = base_args(e);
        HIP_TRY(hipDeviceSynchronize());
        hipLaunchKernelGGL(k_privatize, dim3((e->N + 255) / 256), dim3(256), 0, 0, a);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipDeviceSynchronize());
    }
    double *sg = nullptr, *sub = nullptr;
    int32_t *ns = nullptr;
    HIP_TRY(hipMalloc((void **)&sg, (size_t)m * 6 * 8));
    HIP_TRY(hipMalloc((void **)&sub, (size_t)m * K * 3 * 8));
    HIP_TRY(hipMalloc((void **)&ns, (size_t)m * 4));
    // UAV.py:216-218: RRTPlanner(step = sub_granularity = 30, obstacle_step 5)
    int rc = uavenv_rrt_plan(e, m, nullptr, nullptr, 0, seed, max_iter, 30.0, 5.0, sg, sub, ns, nullptr, stream);
    if (rc != UAVENV_OK) { (void)hipFree(sg); (void)hipFree(sub); (void)hipFree(ns); return fail(rc, "uavenv_rrt_plan failed"); }
    int32_t *counters = nullptr, replaced = 0;
    HIP_TRY(hipMalloc((void **)&counters, 8));
    HIP_TRY(hipMemsetAsync(counters, 0, 8, (hipStream_t)stream));
    hipLaunchKernelGGL(k_bank_fix, dim3((m + 255) / 256), dim3(256), 0, (hipStream_t)stream, sg, sub, ns, m, K, counters);
    hipLaunchKernelGGL(k_bank_fix2, dim3((m + 255) / 256), dim3(256), 0, (hipStream_t)stream, ns, m);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(&replaced, counters, 4, hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    (void)hipFree(counters);
    if (replaced >= m) {      // nothing valid to copy from: publishing this bank would install n_total < 2 at every reset
        (void)hipFree(sg); (void)hipFree(sub); (void)hipFree(ns);
        return fail(UAVENV_EINVAL, "uavenv_plan_scenarios: none of the %d scenarios could be planned (max_iter=%d, K=%d)", m,
                    max_iter, K);
    }
    (void)hipFree(e->bank_sg); (void)hipFree(e->bank_sub); (void)hipFree(e->bank_nsub);   // hipFree waits for the device
    e->bank_replaced = replaced;
    e->bank_sg = sg; e->bank_sub = sub; e->bank_nsub = ns; e->bank_m = m;
    return UAVENV_OK;
}

// ---- rolling refresh of the bank ------------------------------------------------------------------------------------------
// The reference plans a fresh path at EVERY reset (Agents/UAV.py:327-366 -> PathPlan/RRT.py:63-105); the env kernels reset
// from a bank planned in advance.  These three calls keep the bank turning over while the loop runs: PLAN a slice of new
// scenarios into a staging area (any stream: it touches neither the bank nor an agent, so it can run beside the step
// kernels), then COMMIT on the step kernels' stream: a bank row of the slice takes its new plan unless an agent is flying it
// (agents keep a scenario id and read their sub-goal list from the bank) or the new plan is unusable.
__global__ void k_bank_mark(StepArgs a, int first, int count, int32_t *inuse)
{
    const DevState &S = a.st;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < a.N; i += gridDim.x * blockDim.x) {
        const int scn = S.I(I_SCN)[i];
        if (scn >= first && scn < first + count) inuse[scn - first] = 1;
    }
}

// one wavefront-sized group of threads per row: K x 3 doubles + 6 + 1
__global__ void __launch_bounds__(256) k_bank_commit(double *bank_sg, double *bank_sub, int32_t *bank_nsub, int first, int count, int K,
                                                     const double *sg, const double *sub, const int32_t *nsub, const int32_t *inuse,
                                                     int32_t *counters)
{
    const int lane = (int)threadIdx.x & 63;
    const int r = (int)(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (r >= count) return;
    const int n = nsub[r];
    const bool used = inuse && inuse[r] != 0;
    const bool ok = n >= 2 && n <= K && !used;
    if (lane == 0) atomicAdd(&counters[ok ? 0 : (used ? 1 : 2)], 1);
    if (!ok) return;
    const size_t row = (size_t)(first + r);
    for (int q = lane; q < K * 3; q += 64) bank_sub[row * K * 3 + q] = sub[(size_t)r * K * 3 + q];
    if (lane < 6) bank_sg[row * 6 + lane] = sg[(size_t)r * 6 + lane];
    if (lane == 6) bank_nsub[row] = n;
}

int uavenv_replan_begin(UavEnv *e, int32_t first, int32_t count, uint64_t seed, int32_t max_iter, void *plan_stream)
{
    if (!e || first < 0 || count <= 0 || max_iter <= 0) return fail(UAVENV_EINVAL, "uavenv_replan_begin: bad argument");
    if (!e->have_world || e->bank_m <= 0) return fail(UAVENV_EINVAL, "uavenv_replan_begin: no world / no bank to refresh");
    if (first + count > e->bank_m) return fail(UAVENV_EINVAL, "uavenv_replan_begin: rows [%d, %d) outside the bank of %d", first, first + count, e->bank_m);
    if (e->rp_pending) return fail(UAVENV_EINVAL, "uavenv_replan_begin: the previous slice has not been committed");
    HIP_TRY(hipSetDevice(e->cfg.device));
    const int K = e->cfg.max_subgoals;
    if (e->rp_cap < count) {           // (first call, or a larger slice: the one place this API allocates)
        HIP_TRY(hipDeviceSynchronize());
        (void)hipFree(e->rp_sg); (void)hipFree(e->rp_sub); (void)hipFree(e->rp_nsub);
        e->rp_sg = e->rp_sub = nullptr; e->rp_nsub = nullptr; e->rp_cap = 0;
        HIP_TRY(hipMalloc((void **)&e->rp_sg, (size_t)count * 6 * 8));
        HIP_TRY(hipMalloc((void **)&e->rp_sub, (size_t)count * K * 3 * 8));
        HIP_TRY(hipMalloc((void **)&e->rp_nsub, (size_t)2 * count * 4));
        e->rp_cap = count;
    }
    if (!e->rp_counters) {
        HIP_TRY(hipMalloc((void **)&e->rp_counters, 16));
        HIP_TRY(hipMemset(e->rp_counters, 0, 16));
    }
    if (!e->rp_done) HIP_TRY(hipEventCreateWithFlags(&e->rp_done, hipEventDisableTiming));
    if (!e->rp_nodes) {                // UAVENV_REPLAN_WGS: how many wavefronts plan in the background (default 128: half a wavefront per CU)
        const char *ev = getenv("UAVENV_REPLAN_WGS");
        const int v = ev ? atoi(ev) : 0;
        e->rp_wgs = v > 0 && v <= 4096 ? v : 128;
        HIP_TRY(hipMalloc(&e->rp_nodes, (size_t)uavenv_rrt_scratch_bytes(e->rp_wgs)));
    }
    // The previous slice's k_bank_commit was only ENQUEUED on the step stream (behind whatever backlog of passes that stream
    // carries) and reads rp_sg / rp_sub / rp_nsub: the planner, which writes a row as soon as its tree is done, must not start
    // before that kernel has run -- otherwise bank rows are torn and the bank's contents depend on timing.
    if (e->rp_commit_recorded) HIP_TRY(hipStreamWaitEvent((hipStream_t)plan_stream, e->rp_committed, 0));
    // the Philox stream of row r is keyed by (seed, first + r): every refresh passes its own seed (generation)
    int rc = uavenv_rrt_plan_at(e, first, count, nullptr, nullptr, 0, seed, max_iter, 30.0, 5.0, e->rp_sg, e->rp_sub, e->rp_nsub,
                                nullptr, e->rp_nodes, e->rp_wgs, plan_stream);
    if (rc != UAVENV_OK) return fail(rc, "uavenv_replan_begin: planner launch failed");
    HIP_TRY(hipEventRecord(e->rp_done, (hipStream_t)plan_stream));
    e->rp_first = first; e->rp_count = count; e->rp_pending = true; e->rp_world_gen = e->world_gen;
    e->rp_calls += 1; e->rp_rows_planned += count;
    return UAVENV_OK;
}

int uavenv_replan_ready(UavEnv *e)
{
    if (!e) return UAVENV_EINVAL;
    if (!e->rp_pending) return -1;
    const hipError_t q = hipEventQuery(e->rp_done);
    if (q == hipSuccess) return 1;
    if (q == hipErrorNotReady) { (void)hipGetLastError(); return 0; }
    return fail(UAVENV_EHIP, "uavenv_replan_ready: %s", hipGetErrorString(q));
}

int uavenv_replan_commit(UavEnv *e, int32_t force, void *stream)
{
    if (!e) return fail(UAVENV_EINVAL, "null env");
    if (!e->rp_pending) return fail(UAVENV_EINVAL, "uavenv_replan_commit: nothing planned");
    e->rp_pending = false;
    if (e->rp_world_gen != e->world_gen || e->rp_first + e->rp_count > e->bank_m) return UAVENV_OK;   // planned for another world / bank: dropped
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipStreamWaitEvent(s, e->rp_done, 0));
    const int cap = e->rp_cap, count = e->rp_count;
    int32_t *inuse = e->rp_nsub + cap, *counters = e->rp_counters;
    if (force) {                       // the caller resets EVERY agent next (an episode boundary): no list is being flown
        inuse = nullptr;
    } else {
        HIP_TRY(hipMemsetAsync(inuse, 0, (size_t)count * 4, s));
        StepArgs a = base_args(e);
        hipLaunchKernelGGL(k_bank_mark, dim3((e->N + 255) / 256 < 1024 ? (e->N + 255) / 256 : 1024), dim3(256), 0, s, a, e->rp_first, count, inuse);
    }
    hipLaunchKernelGGL(k_bank_commit, dim3((count + 3) / 4), dim3(256), 0, s, e->bank_sg, e->bank_sub, e->bank_nsub, e->rp_first, count,
                       e->cfg.max_subgoals, e->rp_sg, e->rp_sub, e->rp_nsub, inuse, counters);
    HIP_TRY(hipGetLastError());
    if (!e->rp_committed) HIP_TRY(hipEventCreateWithFlags(&e->rp_committed, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(e->rp_committed, s));      // uavenv_replan_begin makes the next planner launch wait for this
    e->rp_commit_recorded = true;
    return UAVENV_OK;
}

int uavenv_set_step_meta(UavEnv *e, void *meta_frame_dev, const float *action1_frame_dev)
{
    if (!e) return fail(UAVENV_EINVAL, "null env");
    if ((((uintptr_t)meta_frame_dev) & 15u) != 0) return fail(UAVENV_EINVAL, "uavenv_set_step_meta: records are 16-byte aligned");
    e->step_meta = meta_frame_dev;
    e->step_meta_a1 = meta_frame_dev ? action1_frame_dev : nullptr;
    return UAVENV_OK;
}

int uavenv_set_moved_word(UavEnv *e, uint32_t *dev_word)
{
    if (!e) return fail(UAVENV_EINVAL, "null env");
    e->moved_word = dev_word;
    return UAVENV_OK;
}

uint64_t uavenv_tick(const UavEnv *e) { return e ? e->tick : 0; }

int uavenv_bank_read(UavEnv *e, int32_t first, int32_t count, double *host_sg, double *host_sub, int32_t *host_nsub)
{
    if (!e || first < 0 || count <= 0 || first + count > e->bank_m) return fail(UAVENV_EINVAL, "uavenv_bank_read: rows outside the bank");
    HIP_TRY(hipSetDevice(e->cfg.device));
    HIP_TRY(hipDeviceSynchronize());
    const size_t K3 = (size_t)e->cfg.max_subgoals * 3;
    if (host_sg) HIP_TRY(hipMemcpy(host_sg, e->bank_sg + (size_t)first * 6, (size_t)count * 6 * 8, hipMemcpyDeviceToHost));
    if (host_sub) HIP_TRY(hipMemcpy(host_sub, e->bank_sub + (size_t)first * K3, (size_t)count * K3 * 8, hipMemcpyDeviceToHost));
    if (host_nsub) HIP_TRY(hipMemcpy(host_nsub, e->bank_nsub + first, (size_t)count * 4, hipMemcpyDeviceToHost));
    return UAVENV_OK;
}

int uavenv_replan_stats(UavEnv *e, int64_t *out5)
{
    if (!e || !out5) return fail(UAVENV_EINVAL, "null argument");
    out5[0] = e->rp_calls; out5[1] = e->rp_rows_planned; out5[2] = out5[3] = out5[4] = 0;
    if (e->rp_counters) {
        int32_t c[4] = {0, 0, 0, 0};
        HIP_TRY(hipMemcpy(c, e->rp_counters, 16, hipMemcpyDeviceToHost));     // synchronises
        out5[2] = c[0]; out5[3] = c[1]; out5[4] = c[2];
    }
    return UAVENV_OK;
}

int uavenv_bank_stats(const UavEnv *e, int32_t *m, int32_t *replaced)
{
    if (!e) return fail(UAVENV_EINVAL, "null env");
    if (m) *m = e->bank_m;
    if (replaced) *replaced = e->bank_replaced;
    return UAVENV_OK;
}

int uavenv_reset_all(UavEnv *e, uint64_t seed, void *stream)
{
    if (!e) return fail(UAVENV_EINVAL, "null env");
    if (e->bank_m <= 0) return fail(UAVENV_EINVAL, "uavenv_reset_all: no scenario bank loaded");
    HIP_TRY(hipSetDevice(e->cfg.device));
    e->seed = seed;
    e->tick = 0;
    StepArgs a = base_args(e);
    const int block = 256, grid = (e->N + block - 1) / block;
    if (e->cfg.apf_enabled == 1) hipLaunchKernelGGL(k_reset_all<true>, dim3(grid), dim3(block), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(k_reset_all<false>, dim3(grid), dim3(block), 0, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    e->tick = 1;
    return UAVENV_OK;
}

int uavenv_set_state(UavEnv *e, int32_t first, int32_t count, const double *kin, const int32_t *step,
                     const int32_t *n_sub, const int32_t *alias, const double *sub)
{
    if (!e || !kin || !step || !n_sub || !sub) return fail(UAVENV_EINVAL, "null argument");
    if (first < 0 || count <= 0 || first + count > e->N) return fail(UAVENV_EINVAL, "range [%d,+%d) of %d", first, count, e->N);
    const int K = e->cfg.max_subgoals;
    for (int c = 0; c < count; ++c)
        if (n_sub[c] < 0 || n_sub[c] > K) return fail(UAVENV_EINVAL, "n_sub[%d]=%d > K=%d", c, n_sub[c], K);
    HIP_TRY(hipSetDevice(e->cfg.device));
    double *d_kin = nullptr, *d_sub = nullptr;
    int32_t *d_i = nullptr;
    const size_t c = (size_t)count;
    HIP_TRY(hipMalloc((void **)&d_kin, c * 8 * 8));
    HIP_TRY(hipMalloc((void **)&d_sub, c * K * 3 * 8));
    HIP_TRY(hipMalloc((void **)&d_i, c * 4 * 3));
    HIP_TRY(hipMemcpy(d_kin, kin, c * 64, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_sub, sub, c * K * 24, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_i, step, c * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_i + c, n_sub, c * 4, hipMemcpyHostToDevice));
    if (alias) HIP_TRY(hipMemcpy(d_i + 2 * c, alias, c * 4, hipMemcpyHostToDevice));
    StepArgs a = base_args(e);
    const int block = 128, grid = (count + block - 1) / block;
    hipLaunchKernelGGL(k_set_state, dim3(grid), dim3(block), 0, 0, a, first, count, d_kin, d_i, d_i + c,
                       alias ? d_i + 2 * c : (const int32_t *)nullptr, d_sub);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    (void)hipFree(d_kin); (void)hipFree(d_sub); (void)hipFree(d_i);
    return UAVENV_OK;
}

int uavenv_get_state(UavEnv *e, int32_t first, int32_t count, double *out16, double *out_sub, int32_t *out_alias)
{
    if (!e || !out16) return fail(UAVENV_EINVAL, "null argument");
    if (first < 0 || count <= 0 || first + count > e->N) return fail(UAVENV_EINVAL, "range [%d,+%d) of %d", first, count, e->N);
    HIP_TRY(hipSetDevice(e->cfg.device));
    const int K = e->cfg.max_subgoals;
    const size_t c = (size_t)count;
    double *d16 = nullptr, *dsub = nullptr;
    int32_t *dal = nullptr;
    HIP_TRY(hipMalloc((void **)&d16, c * 16 * 8));
    if (out_sub) HIP_TRY(hipMalloc((void **)&dsub, c * K * 24));
    if (out_alias) HIP_TRY(hipMalloc((void **)&dal, c * 4));
    StepArgs a = base_args(e);
    const int block = 128, grid = (count + block - 1) / block;
    HIP_TRY(hipDeviceSynchronize());
    hipLaunchKernelGGL(k_get_state, dim3(grid), dim3(block), 0, 0, a, first, count, d16, dsub, dal);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out16, d16, c * 128, hipMemcpyDeviceToHost));
    if (out_sub) HIP_TRY(hipMemcpy(out_sub, dsub, c * K * 24, hipMemcpyDeviceToHost));
    if (out_alias) HIP_TRY(hipMemcpy(out_alias, dal, c * 4, hipMemcpyDeviceToHost));
    (void)hipFree(d16); (void)hipFree(dsub); (void)hipFree(dal);
    return UAVENV_OK;
}

int uavenv_step(UavEnv *e, const void *actions, int32_t action_kind, void *obs, double *reward64, float *reward32,
                uint8_t *ret_done, uint8_t *agent_done, uint8_t *info, uint8_t *valid, double *energy64,
                const uint8_t *active, uint32_t flags, void *stream)
{
    if (!e) return fail(UAVENV_EINVAL, "null env");
    const PendingMeta pm = take_step_meta(e);         // (consumed by THIS call, whatever it returns: taken before any other check)
    if (!actions) return fail(UAVENV_EINVAL, "null actions");
    if (!e->have_world) return fail(UAVENV_EINVAL, "uavenv_step before uavenv_set_buildings");
    if (action_kind < 0 || action_kind > 2) return fail(UAVENV_EINVAL, "action_kind %d", action_kind);
    if ((flags & UAVENV_STEP_AUTO_RESET) && e->bank_m <= 0)
        return fail(UAVENV_EINVAL, "AUTO_RESET needs a scenario bank (uavenv_load_scenarios)");
    StepArgs a = base_args(e);
    a.actions = actions;
    a.action_kind = action_kind;
    a.obs = obs;
    a.reward64 = reward64;
    a.reward32 = reward32;
    a.ret_done = ret_done;
    a.agent_done = agent_done;
    a.info = info;
    a.valid = valid;
    a.energy64 = energy64;
    a.active = active;
    a.flags = flags;
    a.meta = reinterpret_cast<uint4 *>(pm.meta);
    a.meta_a1 = pm.a1;
    if (e->mask_bytes == 4) launch_step<uint32_t>(e, a, (hipStream_t)stream);
    else launch_step<uint64_t>(e, a, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    e->tick += 1;
    return UAVENV_OK;
}

int uavenv_step_policy(UavEnv *e, const UavDqnNet *net, const void *obs_cur, float eps, uint64_t seed, uint64_t counter,
                       int32_t *action_out, void *obs, double *reward64, float *reward32, uint8_t *ret_done, uint8_t *agent_done,
                       uint8_t *info, uint8_t *valid, double *energy64, const uint8_t *active, uint32_t flags, void *stream)
{
    return uavenv_step_policy_img(e, net, obs_cur, eps, seed, counter, action_out, obs, reward64, reward32, ret_done, agent_done, info, valid,
                                  energy64, active, flags, nullptr, stream);
}

int uavenv_step_policy_img(UavEnv *e, const UavDqnNet *net, const void *obs_cur, float eps, uint64_t seed, uint64_t counter,
                           int32_t *action_out, void *obs, double *reward64, float *reward32, uint8_t *ret_done, uint8_t *agent_done,
                           uint8_t *info, uint8_t *valid, double *energy64, const uint8_t *active, uint32_t flags,
                           const float *image_dev, void *stream)
{
    if (!e) return fail(UAVENV_EINVAL, "null env");
    const PendingMeta pm = take_step_meta(e);         // (consumed by THIS call, whatever it returns: a caller that falls back to act + step asks again)
    if (!net || !net->local || !obs_cur || !action_out) return fail(UAVENV_EINVAL, "null argument");
    if (!e->have_world) return fail(UAVENV_EINVAL, "uavenv_step_policy before uavenv_set_buildings");
    if ((flags & UAVENV_STEP_AUTO_RESET) && e->bank_m <= 0)
        return fail(UAVENV_EINVAL, "AUTO_RESET needs a scenario bank (uavenv_load_scenarios)");
    static const int coop_env = env_int("UAVENV_COOP", -1);
    const int n2 = net->n_actions + (net->dueling ? 1 : 0);
    const bool one_wave = (flags & UAVENV_STEP_ONE_WAVE) || (coop_env >= 0 ? coop_env == 0 : e->N > 49152);   // launch_step's choice
    const bool shape_ok = e->cfg.apf_enabled != 1 && net->w == uavq::kW && net->hid == uavq::kHid && n2 <= 4 && net->n_actions >= 2 &&
                          net->n_actions == e->cfg.n_actions && (((uintptr_t)net->local) & 15u) == 0;
    // packed rows + f32 MFMA in k_step_coop's prologue; f16 rows + f16 MFMA in the one-wave k_step's
    const bool coop_packed = !one_wave && e->cfg.obs_dtype == UAVENV_OBS_PACKED && net->mfma_dtype == UAVENV_MFMA_F32 &&
                             (((uintptr_t)obs_cur) & 15u) == 0;
    int blk = 0, grd = 0;
    launch_geometry(e->N, blk, grd);
    // (its workgroup -- world + the policy team's 27 KB -- must leave room for four per CU, and three wavefronts per 64 agents
    // at 152 registers fit the chip up to 65 536 agents: beyond, the separate act launch is the faster form)
    // (an even agent count: every frame and every wavefront's 32-row half then starts and ends on a 16-byte boundary)
    const bool wave_f16 = one_wave && blk == 64 && e->N <= 65536 && (e->N & 1) == 0 && e->cfg.obs_dtype == UAVENV_OBS_F16 && net->mfma_dtype == UAVENV_MFMA_F16 &&
                          (((uintptr_t)obs_cur) & 15u) == 0 && env_int("UAVENV_POLH", 1) != 0 &&
                          (size_t)e->world_bytes + 16 + uavq::kPolBytes <= 40960;
    if (!shape_ok || !(coop_packed || wave_f16))
        return fail(UAVENV_EINVAL, "uavenv_step_policy: this env / net takes uavenv_dqn_act + uavenv_step");
    StepArgs a = base_args(e);
    a.actions = action_out;
    a.action_kind = UAVENV_ACT_INDEX_I32;
    a.obs = obs;
    a.reward64 = reward64;
    a.reward32 = reward32;
    a.ret_done = ret_done;
    a.agent_done = agent_done;
    a.info = info;
    a.valid = valid;
    a.energy64 = energy64;
    a.active = active;
    a.flags = flags;
    a.pol_local = net->local;
    a.pol_img = (coop_packed && image_dev && (((uintptr_t)image_dev) & 15u) == 0) ? image_dev : nullptr;
    a.pol_obs = reinterpret_cast<const uint32_t *>(obs_cur);
    a.pol_act = action_out;
    a.pol_dueling = net->dueling;
    a.pol_eps = eps;
    a.pol_seed = seed;
    a.pol_counter = counter;
    a.meta = reinterpret_cast<uint4 *>(pm.meta);
    a.meta_a1 = pm.a1;
    if (e->mask_bytes == 4) launch_step<uint32_t>(e, a, (hipStream_t)stream);
    else launch_step<uint64_t>(e, a, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    e->tick += 1;
    return UAVENV_OK;
}

int uavenv_set_debug_buffer(UavEnv *e, unsigned long long *dev_buf)
{
    if (!e) return fail(UAVENV_EINVAL, "null env");
    e->dbg = dev_buf;
    return UAVENV_OK;
}

int uavenv_observe(UavEnv *e, void *obs, void *stream)
{
    if (!e || !obs) return fail(UAVENV_EINVAL, "null argument");
    if (!e->have_world) return fail(UAVENV_EINVAL, "uavenv_observe before uavenv_set_buildings");
    StepArgs a = base_args(e);
    a.obs = obs;
    int block, grid;
    launch_geometry(e->N, block, grid);
    a.block = block;
    const size_t lds = (size_t)e->world_bytes;
    const int obs_kind = e->cfg.obs_dtype;
    hipStream_t s = (hipStream_t)stream;
#define UAV_OBSERVE(MASK)                                                                                        \
    do {                                                                                                         \
        if (obs_kind == UAVENV_OBS_F16) launch_lds((k_observe<MASK, OBS_KIND_F16>), grid, block, lds, s, a);     \
        else if (obs_kind == UAVENV_OBS_PACKED) launch_lds((k_observe<MASK, OBS_KIND_PACKED>), grid, block, lds, s, a); \
        else launch_lds((k_observe<MASK, OBS_KIND_F32>), grid, block, lds, s, a);                                \
    } while (0)
    if (e->mask_bytes == 4) UAV_OBSERVE(uint32_t);
    else UAV_OBSERVE(uint64_t);
#undef UAV_OBSERVE
    HIP_TRY(hipGetLastError());
    return UAVENV_OK;
}

static int threaten_impl(UavEnv *e, const double *xyz, uint8_t *out, int64_t n, void *stream, bool allpairs)
{
    if (!e || !xyz || !out || n < 0) return fail(UAVENV_EINVAL, "null argument");
    if (!e->have_world) return fail(UAVENV_EINVAL, "threaten_rate before uavenv_set_buildings");
    if (n == 0) return UAVENV_OK;
    StepArgs a = base_args(e);
    const int block = 256;
    a.block = block;
    int64_t g = (n + block - 1) / block;
    const int grid = (int)(g > 2048 ? 2048 : g);
    const size_t lds = (size_t)e->world_bytes;
    hipStream_t s = (hipStream_t)stream;
    if (e->mask_bytes == 4) {
        if (allpairs) hipLaunchKernelGGL((k_threaten<uint32_t, true>), dim3(grid), dim3(block), lds, s, a, xyz, out, n);
        else hipLaunchKernelGGL((k_threaten<uint32_t, false>), dim3(grid), dim3(block), lds, s, a, xyz, out, n);
    } else {
        if (allpairs) hipLaunchKernelGGL((k_threaten<uint64_t, true>), dim3(grid), dim3(block), lds, s, a, xyz, out, n);
        else hipLaunchKernelGGL((k_threaten<uint64_t, false>), dim3(grid), dim3(block), lds, s, a, xyz, out, n);
    }
    HIP_TRY(hipGetLastError());
    return UAVENV_OK;
}

int uavenv_geometry(const double *ab, double *angle_out, double *dist_out, int64_t n, void *stream)
{
    if (!ab || n < 0 || (!angle_out && !dist_out)) return UAVENV_EINVAL;
    if (n == 0) return UAVENV_OK;
    hipLaunchKernelGGL(k_geometry, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ab, angle_out, dist_out, n);
    return hipGetLastError() == hipSuccess ? UAVENV_OK : UAVENV_EHIP;
}

int uavenv_threaten_rate(UavEnv *e, const double *xyz, uint8_t *out, int64_t n, void *stream)
{
    return threaten_impl(e, xyz, out, n, stream, false);
}

int uavenv_threaten_rate_allpairs(UavEnv *e, const double *xyz, uint8_t *out, int64_t n, void *stream)
{
    return threaten_impl(e, xyz, out, n, stream, true);
}

}  // extern "C"
