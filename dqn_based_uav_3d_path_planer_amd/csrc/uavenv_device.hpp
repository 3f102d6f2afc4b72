// uavenv_device.hpp -- device-side building blocks of the PathPlan_City hot path (gfx950).
//
// Numerics contract: everything that decides a reward, a termination or an occupancy bit is
// IEEE f64 evaluated in the reference's operation order (no FMA contraction: the TU is built
// with -ffp-contract=off), so that results agree with the CPython reference to last-ulp
// libm differences (OCML vs glibc atan2/sin/cos), far inside the 1e-5 parity tolerance.
// Only the stored observation is narrowed (f32 / f16).
#pragma once
#ifndef UAVENV_HOT_PRIO_DEFINED
#define UAVENV_HOT_PRIO_DEFINED
// Wave priority of the loop's own kernels (s_setprio, 0..3; the hardware's default is 0).  The background planner that turns the
// reset bank over (k_rrt_plan<.., false>, csrc/rrt.hip) shares SIMDs with them: a step / gradient launch is as long as its slowest
// workgroup, so a planner wavefront taking every other issue slot of ONE SIMD stretches the whole launch.  At priority 3 the loop's
// wavefronts issue first and the planner fills the slots they leave (round 6: see DESIGN 3.5 for the measured difference).
#ifndef UAVENV_HOT_PRIO
#define UAVENV_HOT_PRIO 3
#endif
#define UAV_HOT_PRIO() __builtin_amdgcn_s_setprio(UAVENV_HOT_PRIO)
#endif
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

namespace uav {

constexpr double kPi = 3.141592653589793;                 // math.pi
constexpr double kRad2Deg = 180.0 / 3.141592653589793;    // CPython math.degrees factor
constexpr double kTwoPi = 2.0 * 3.141592653589793;

// One cylinder as the narrow-phase test needs it (LDS resident, 32 B).
// Obstacles/building.py:20-26:  z > H -> miss;  sqrt(dx^2+dy^2+0) < R -> hit.
// `thr` is the smallest double t with sqrt_rn(t) >= R, so (s < thr) == (sqrt_rn(s) < R) EXACTLY
// (sqrt_rn is monotone): the square root leaves the inner loop without changing any result.
struct BldLds {
    double cx, cy, thr, H;
};

// Per-cylinder quick-accept / quick-reject radii for the three stencils (LDS resident, 48 B).  With c the stencil
// centre and diag_k = 2*sqrt(2)*spacing_k the farthest stencil point:
//   |c - centre_b|^2 >= rej2[k] = (R + diag_k + 1e-6)^2  =>  NO stencil point can be inside the disc  -> skip;
//   |c - centre_b|^2 <  acc2[k] = (R - diag_k - 1e-6)^2  =>  EVERY stencil point is inside           -> all 25 bits.
// Both are implications with a 1e-6 m guard band (>> f64 rounding), so they never change a result of the exact
// 25-point test; they only skip it.  A 20 m cell + halo lists every cylinder that MIGHT touch the stencil; most do not.
struct BldAux {
    double rej2[3], acc2[3];
};

// Cylinder as the APF force needs it (Agents/UAV.py:174-210); global memory, uniform index.
// far2 = (R + 60 + 1e-6)^2: beyond it `dis - R > 60` holds for sure (1e-6 m guard >> rounding), so the sqrt and
// everything after it is skipped; ux, uy = (cos, sin) of calculate_angle(0, v): the motion force's direction, a
// per-building constant the reference recomputes for every (sub-goal, building) pair (:192).
struct BldApf {
    double cx, cy, cz, R, vx, vy, vz, vnorm, far2, ux, uy, moving;
};

// The world as a workgroup sees it in LDS: the cylinder table and three candidate grids.
// grid h (h = 2, 10, 20 m) stores, per cell, the bit-mask of cylinders that can hit ANY point within an
// L-infinity distance h of the cell: one mask read then serves a whole 5x5 stencil of spacing h/2 centred in
// the cell (or, for h = 2, any single point of the cell).  Masks are conservative supersets (host side,
// uavenv_set_buildings), every candidate still takes the exact narrow-phase test, so the result is identical to
// the reference's loop over all buildings.
template <typename MaskT>
struct WorldLds {
    const BldLds *b;       // LDS
    const BldAux *aux;     // LDS
    const MaskT *g[3];     // LDS: halo 2 m / 10 m / 20 m, gn*gn masks each
    int gn;
    double inv_cell;
    double W, Hbox;        // Threaten_rate bounds (x and y both use `width`; PathPlan_City.py:218)
};

// BaseClass/CalMod.py:89-102  calculate_angle(p1, p2, mod=1) with (dx,dy) = p2 - p1.
// atan2 -> degrees -> (a + 360) % 360 -> / 180 * pi.  For a in [180, 540] the float modulo is
// exactly (a >= 360 ? a - 360 : a).
__device__ __forceinline__ double calc_angle_body(double dx, double dy)
{
    double a = atan2(dy, dx) * kRad2Deg;
    a = a + 360.0;
    double m = (a >= 360.0) ? (a - 360.0) : a;
    return m / 180.0 * kPi;
}
// Two instantiations of the same arithmetic: out-of-line (small code, best when many waves share a SIMD) and inline
// (lets the scheduler interleave the independent atan2 chains of one step: ~10 % on the single-wave-per-CU launches).
__device__ __noinline__ double calc_angle(double dx, double dy) { return calc_angle_body(dx, dy); }
template <bool INL>
__device__ __forceinline__ double angle_of(double dx, double dy)
{
    return INL ? calc_angle_body(dx, dy) : calc_angle(dx, dy);
}

// cos(|calculate_angle(a) - calculate_angle(b)|) without the two atan2 chains: the cosine of the angle between the two
// vectors, (a.b) / (|a||b|).  calculate_angle maps the zero vector to angle 0 (atan2(0, 0) = 0), i.e. to the direction
// (1, 0); the degree/modulo normalisation only shifts angles by multiples of 2 pi, which the cosine ignores.  Agrees
// with the reference's chain to ~1e-15 (both are a few ulp from the exact value); used ONLY for the reward's heading
// terms (UAV.py:435, :453, :490) -- the heading that is stored and fed to the next step still goes through calc_angle.
__device__ __forceinline__ double cos_between(double ax, double ay, double bx, double by)
{
    if (ax == 0.0 && ay == 0.0) ax = 1.0;
    if (bx == 0.0 && by == 0.0) bx = 1.0;
    const double num = ax * bx + ay * by;
    const double den = sqrt((ax * ax + ay * ay) * (bx * bx + by * by));
    return num / den;
}

// BaseClass/CalMod.py:64-65  Eu_Loc_distance
__device__ __forceinline__ double dist3(double ax, double ay, double az, double bx, double by, double bz)
{
    double dx = ax - bx, dy = ay - by, dz = az - bz;
    return sqrt(dx * dx + dy * dy + dz * dz);
}

// Agents/UAV.py:246-253  Calc_V: speed with the clamp that rescales (x, y).
__device__ __forceinline__ double calc_v(double &vx, double &vy, double max_v)
{
    double V = sqrt(vx * vx + vy * vy + 0.0);
    if (V > max_v) {
        double k = max_v / V;
        vx = vx * k;
        vy = vy * k;
        V = max_v;
    }
    return V;
}

// Agents/UAV.py:239-245  Calc_Fly_Power at speed V (V already clamped).
struct PowerParams {
    double P_i, v_0, d_0, rho, s, A, P_b, F_b;
};
__device__ __forceinline__ double fly_power(const PowerParams &p, double V, int j)
{
    double A = p.A + 0.03 * (double)j;       // UAV.py:55
    double xi = 0.8 + 0.02 * (double)j;      // UAV.py:58
    double V2 = V * V, v02 = p.v_0 * p.v_0;
    double induced = p.P_i * sqrt(sqrt(1.0 + (V2 * V2) / (4.0 * (v02 * v02))) - V2 / (2.0 * v02));
    double parasite = 0.5 * p.d_0 * p.rho * p.s * A * (V2 * V);
    double blade = xi * p.P_b * (1.0 + 3.0 * V2 / (p.F_b * p.F_b));
    return induced + parasite + blade;
}

// LDS hand-off between the lanes of ONE wavefront (DS operations of a wave complete in order; this only has to stop
// the compiler from reordering across it).
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

template <typename MaskT>
__device__ __forceinline__ int ctz_mask(MaskT m)
{
    return (sizeof(MaskT) == 8) ? __builtin_ctzll((unsigned long long)m) : __builtin_ctz((unsigned)m);
}

// Cell of (x, y), clamped into the grid.  For a point outside the box the clamped cell is the cell of its
// projection onto the box, which is within the same L-infinity halo of every in-bounds stencil point.
template <typename MaskT>
__device__ __forceinline__ int cell_of(const WorldLds<MaskT> &w, double x, double y)
{
    int ix = (int)(x * w.inv_cell), iy = (int)(y * w.inv_cell);
    ix = ix < 0 ? 0 : (ix > w.gn - 1 ? w.gn - 1 : ix);
    iy = iy < 0 ? 0 : (iy > w.gn - 1 ? w.gn - 1 : iy);
    return iy * w.gn + ix;
}

// Envs/PathPlan_City.py:215-223  Threaten_rate(p) for one point through the exact broad phase.
template <typename MaskT>
__device__ __forceinline__ int probe(const WorldLds<MaskT> &w, double x, double y, double z)
{
    if ((x < 0.0) | (x > w.W) | (y < 0.0) | (y > w.W) | (z < 0.0) | (z > w.Hbox)) return 1;
    MaskT m = w.g[0][cell_of(w, x, y)];
    int hit = 0;
    while (m) {
        const int b = ctz_mask(m);
        m &= (MaskT)(m - 1);
        const BldLds B = w.b[b];
        const double dx = x - B.cx, dy = y - B.cy;
        const double s = dx * dx + dy * dy;       // + (bz-bz)^2 == + 0.0
        hit |= (int)(!(z > B.H)) & (int)(s < B.thr);
    }
    return hit;
}

// All-pairs variant (no grid): the literal reference loop, for the culling-exactness test.
__device__ __forceinline__ int probe_allpairs(const BldLds *b, int nb, double W, double Hbox, double x, double y,
                                              double z)
{
    if ((x < 0.0) | (x > W) | (y < 0.0) | (y > W) | (z < 0.0) | (z > Hbox)) return 1;
    int hit = 0;
    for (int i = 0; i < nb; ++i) {
        BldLds B = b[i];
        double dx = x - B.cx, dy = y - B.cy;
        double s = dx * dx + dy * dy;
        hit |= (int)(!(z > B.H)) & (int)(s < B.thr);
    }
    return hit;
}

// The per-agent values an observation is built from.
struct ObsIn {
    double px, py, pz, vx, vy, V, gx, gy, gz;
    double s0x, s0y, s0z, s1x, s1y, s1z;
    int step, n_rem;   // n_rem = len(sub_goals)
};

struct ObsBits {
    uint32_t s1, s5, s10, below;
};

// All 80 occupancy probes of state_PathPlan (UAV.py:533-566) in ONE pass over the candidate cylinders of the widest
// (halo 20 m) mask, which contains the candidates of the two narrower stencils: each candidate costs one LDS round
// trip (its 80 bytes) and then feeds three independent quick-reject / quick-accept / 25-point tests, instead of three
// separate mask -> cylinder -> test dependency chains (the observation was LDS-latency bound, not ALU bound).
template <typename MaskT>
__device__ __forceinline__ ObsBits obs_bits(const WorldLds<MaskT> &w, double px, double py, double pz)
{
    const double kSp[3] = {1.0, 5.0, 10.0};
    uint32_t bits[3];
    const bool z_out = (pz < 0.0) | (pz > w.Hbox);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        uint32_t bk = 0;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const double xi = px + kSp[k] * (double)(i - 2), yi = py + kSp[k] * (double)(i - 2);
            bk |= ((xi < 0.0) | (xi > w.W)) ? (0x1Fu << (5 * i)) : 0u;        // whole row i out of the box
            bk |= ((yi < 0.0) | (yi > w.W)) ? (0x108421u << i) : 0u;          // whole column j=i out of the box
        }
        bits[k] = z_out ? 0x1FFFFFFu : bk;
    }
    // UAV.py:562-566: Threaten_rate(px, py, pz - k), k = 1..5
    uint32_t bl = 0;
    const bool xy_out = (px < 0.0) | (px > w.W) | (py < 0.0) | (py > w.W);
#pragma unroll
    for (int k = 1; k <= 5; ++k) {
        const double z = pz - (double)k;
        bl |= (xy_out | (z < 0.0) | (z > w.Hbox)) ? (1u << (k - 1)) : 0u;
    }
    const bool need_below = bl != 0x1Fu;      // only when the UAV flies above ground level

    MaskT m = w.g[2][cell_of(w, px, py)];
    while (m) {
        const int b = ctz_mask(m);
        m &= (MaskT)(m - 1);
        const BldLds B = w.b[b];
        const BldAux X = w.aux[b];
        const double dcx = px - B.cx, dcy = py - B.cy;
        const double dc2 = dcx * dcx + dcy * dcy;
        if (need_below & (dc2 < B.thr)) {
#pragma unroll
            for (int k = 1; k <= 5; ++k) bl |= !((pz - (double)k) > B.H) ? (1u << (k - 1)) : 0u;
        }
        if (pz > B.H) continue;                                   // UAV above this roof: no stencil point can hit
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (dc2 >= X.rej2[k]) continue;                       // cannot touch this stencil
            if (dc2 < X.acc2[k]) { bits[k] = 0x1FFFFFFu; continue; }   // stencil entirely inside the disc
            double dx2[5], dy2[5];
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const double dx = (px + kSp[k] * (double)(i - 2)) - B.cx, dy = (py + kSp[k] * (double)(i - 2)) - B.cy;
                dx2[i] = dx * dx;
                dy2[i] = dy * dy;
            }
            uint32_t bk = bits[k];
#pragma unroll
            for (int i = 0; i < 5; ++i)
#pragma unroll
                for (int j = 0; j < 5; ++j) bk |= ((dx2[i] + dy2[j]) < B.thr) ? (1u << (5 * i + j)) : 0u;
            bits[k] = bk;
        }
    }
    ObsBits o;
    o.s1 = bits[0];
    o.s5 = bits[1];
    o.s10 = bits[2];
    o.below = bl;
    return o;
}

// ---- the same 80 probes with a wave-level WORK QUEUE -----------------------------------------------------------
// With lane == agent, the 25-point test of a (cylinder, stencil) pair is needed by only ~1 lane in 3 per candidate
// iteration, but a divergent branch makes the whole wavefront pay for it every time.  Here phase A only CLASSIFIES
// each (agent, candidate, stencil) as reject / accept / "needs the 25 tests" and pushes the latter as packed items
// into a per-wave LDS queue (ballot + mbcnt prefix, no atomics); phase B runs the 25 tests with one ITEM per lane
// -- dense, no divergence -- and ORs the result into the owner's bit words in LDS; phase C reads them back.
// Same tests on the same operands => bit-identical to obs_bits().
constexpr int kObsQueueCap = 512;
struct ObsWaveLds {
    double pxy[64][2];
    uint32_t bits[64][3];
    uint32_t queue[kObsQueueCap];
};

template <typename MaskT>
__device__ __forceinline__ void obs_queue_drain(const WorldLds<MaskT> &w, ObsWaveLds *L, int count)
{
    const int lane = (int)threadIdx.x & 63;
    wave_lds_sync();
    for (int base = 0; base < count; base += 64) {
        const int it = base + lane;
        if (it < count) {
            const uint32_t item = L->queue[it];
            const int owner = (int)(item & 63u), b = (int)((item >> 6) & 63u), k = (int)(item >> 12);
            const double sp = k == 0 ? 1.0 : (k == 1 ? 5.0 : 10.0);
            const double px = L->pxy[owner][0], py = L->pxy[owner][1];
            const BldLds B = w.b[b];
            double dx2[5], dy2[5];
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const double dx = (px + sp * (double)(i - 2)) - B.cx, dy = (py + sp * (double)(i - 2)) - B.cy;
                dx2[i] = dx * dx;
                dy2[i] = dy * dy;
            }
            uint32_t bk = 0;
#pragma unroll
            for (int i = 0; i < 5; ++i)
#pragma unroll
                for (int j = 0; j < 5; ++j) bk |= ((dx2[i] + dy2[j]) < B.thr) ? (1u << (5 * i + j)) : 0u;
            if (bk) atomicOr(&L->bits[owner][k], bk);
        }
    }
    wave_lds_sync();
}

template <typename MaskT>
__device__ __forceinline__ ObsBits obs_bits_queued(const WorldLds<MaskT> &w, ObsWaveLds *L, double px, double py,
                                                   double pz, bool active)
{
    const int lane = (int)threadIdx.x & 63;
    const double kSp[3] = {1.0, 5.0, 10.0};
    uint32_t bits[3];
    const bool z_out = (pz < 0.0) | (pz > w.Hbox);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        uint32_t bk = 0;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const double xi = px + kSp[k] * (double)(i - 2), yi = py + kSp[k] * (double)(i - 2);
            bk |= ((xi < 0.0) | (xi > w.W)) ? (0x1Fu << (5 * i)) : 0u;
            bk |= ((yi < 0.0) | (yi > w.W)) ? (0x108421u << i) : 0u;
        }
        bits[k] = z_out ? 0x1FFFFFFu : bk;
    }
    uint32_t bl = 0;
    const bool xy_out = (px < 0.0) | (px > w.W) | (py < 0.0) | (py > w.W);
#pragma unroll
    for (int k = 1; k <= 5; ++k) {
        const double z = pz - (double)k;
        bl |= (xy_out | (z < 0.0) | (z > w.Hbox)) ? (1u << (k - 1)) : 0u;
    }
    const bool need_below = bl != 0x1Fu;

    L->pxy[lane][0] = px;
    L->pxy[lane][1] = py;
    L->bits[lane][0] = 0u; L->bits[lane][1] = 0u; L->bits[lane][2] = 0u;
    MaskT m = active ? w.g[2][cell_of(w, px, py)] : (MaskT)0;
    int count = 0;                                   // wave-uniform
    while (__ballot(m != 0) != 0ull) {
        if (count > kObsQueueCap - 192) {            // room for one more iteration's worst case (64 lanes x 3)
            obs_queue_drain(w, L, count);
            count = 0;
        }
        const bool have = m != 0;
        const int b = have ? ctz_mask(m) : 0;
        if (have) m &= (MaskT)(m - 1);
        bool full[3];
        {   // (predicated, not branched: the if / else-if ladder per stencil compiled to one divergent branch per arm, ~20 cycles each
            // with one wavefront per SIMD; a lane without a candidate reads cylinder 0 and masks everything with `have`)
            const BldLds B = w.b[b];
            const BldAux X = w.aux[b];
            const double dcx = px - B.cx, dcy = py - B.cy;
            const double dc2 = dcx * dcx + dcy * dcy;
            const bool under = have & need_below & (dc2 < B.thr);
#pragma unroll
            for (int k = 1; k <= 5; ++k) bl |= (under & !((pz - (double)k) > B.H)) ? (1u << (k - 1)) : 0u;
            const bool low = have & !(pz > B.H);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const bool in = dc2 < X.acc2[k];                           // stencil entirely inside the disc
                bits[k] |= (low & in) ? 0x1FFFFFFu : 0u;
                full[k] = low & !in & (dc2 < X.rej2[k]);                   // can touch: needs the 25 exact tests
            }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const unsigned long long bm = __ballot(full[k]);
            if (bm) {
                const int pos = count + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bm >> 32),
                                                                       __builtin_amdgcn_mbcnt_lo((uint32_t)bm, 0u));
                if (full[k]) L->queue[pos] = (uint32_t)lane | ((uint32_t)b << 6) | ((uint32_t)k << 12);
                count += __builtin_popcountll(bm);
            }
        }
    }
    obs_queue_drain(w, L, count);
    ObsBits o;
    o.s1 = bits[0] | L->bits[lane][0];
    o.s5 = bits[1] | L->bits[lane][1];
    o.s10 = bits[2] | L->bits[lane][2];
    o.below = bl;
    return o;
}

// ---- the cooperative small-N kernel's split of the same work --------------------------------------------------------
// obs_box_bits: the out-of-box part of ONE stencil (every probe of a row / column that leaves [0, W], or of a UAV
// outside [0, H], is 1), no cylinders.  obs_cand_queued: the cylinder part of the three stencils for every third
// candidate cylinder of each agent, starting with its `phase`-th -- three wavefronts (phase 0, 1, 2) cover all of them
// with a third of the classify iterations and a third of the queue each, balanced agent by agent.  OR of all parts ==
// obs_bits().
__device__ __forceinline__ uint32_t obs_box_bits(double W, double Hbox, double sp, double px, double py, double pz)
{
    uint32_t bk = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const double xi = px + sp * (double)(i - 2), yi = py + sp * (double)(i - 2);
        bk |= ((xi < 0.0) | (xi > W)) ? (0x1Fu << (5 * i)) : 0u;
        bk |= ((yi < 0.0) | (yi > W)) ? (0x108421u << i) : 0u;
    }
    return ((pz < 0.0) | (pz > Hbox)) ? 0x1FFFFFFu : bk;
}

template <typename MaskT>
__device__ __forceinline__ ObsBits obs_cand_queued(const WorldLds<MaskT> &w, ObsWaveLds *L, double px, double py,
                                                   double pz, bool active, int phase)
{
    const int lane = (int)threadIdx.x & 63;
    uint32_t bits[3] = {0u, 0u, 0u};
    L->pxy[lane][0] = px;
    L->pxy[lane][1] = py;
    L->bits[lane][0] = 0u; L->bits[lane][1] = 0u; L->bits[lane][2] = 0u;
    MaskT m = active ? w.g[2][cell_of(w, px, py)] : (MaskT)0;
    for (int q = 0; q < phase; ++q) m &= (MaskT)(m - 1);          // skip the candidates the earlier phases start with
    int count = 0;                                   // wave-uniform
    while (__ballot(m != 0) != 0ull) {
        if (count > kObsQueueCap - 192) {            // room for one more iteration's worst case (64 lanes x 3)
            obs_queue_drain(w, L, count);
            count = 0;
        }
        const bool have = m != 0;
        const int b = have ? ctz_mask(m) : 0;
        m &= (MaskT)(m - 1);                         // this candidate, and the two the other phases take
        m &= (MaskT)(m - 1);
        m &= (MaskT)(m - 1);
        bool full[3];
        {   // (predicated, not branched: see obs_bits_queued)
            const BldLds B = w.b[b];
            const BldAux X = w.aux[b];
            const double dcx = px - B.cx, dcy = py - B.cy;
            const double dc2 = dcx * dcx + dcy * dcy;
            const bool low = have & !(pz > B.H);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const bool in = dc2 < X.acc2[k];                           // stencil entirely inside the disc
                bits[k] |= (low & in) ? 0x1FFFFFFu : 0u;
                full[k] = low & !in & (dc2 < X.rej2[k]);                   // can touch: needs the 25 exact tests
            }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const unsigned long long bm = __ballot(full[k]);
            if (bm) {
                const int at = count + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bm >> 32),
                                                                      __builtin_amdgcn_mbcnt_lo((uint32_t)bm, 0u));
                if (full[k]) L->queue[at] = (uint32_t)lane | ((uint32_t)b << 6) | ((uint32_t)k << 12);
                count += __builtin_popcountll(bm);
            }
        }
    }
    obs_queue_drain(w, L, count);
    ObsBits o;
    o.s1 = bits[0] | L->bits[lane][0];
    o.s5 = bits[1] | L->bits[lane][1];
    o.s10 = bits[2] | L->bits[lane][2];
    o.below = 0u;
    return o;
}

// ---- one stencil per wavefront (the cooperative small-N kernel) ---------------------------------------------------
// Stencil k (spacing kSp[k]) of state_PathPlan for the 64 agents of a workgroup, by ONE wavefront: the same
// classify -> queue -> dense 25-point tests as obs_bits_queued, but over the stencil's own halo grid g[k] (fewer
// candidates than the widest grid) and with a single test per candidate.  pos = LDS [64][4] doubles (x, y, z, pad),
// acc = LDS [64] words this function zeroes and ORs into.  Same exact tests on the same operands => the same bits.
template <typename MaskT>
__device__ __forceinline__ void stencil_queue_drain(const WorldLds<MaskT> &w, const uint32_t *queue, uint32_t *acc,
                                                    const double *pos, double sp, int count)
{
    const int lane = (int)threadIdx.x & 63;
    wave_lds_sync();
    for (int base = 0; base < count; base += 64) {
        const int it = base + lane;
        if (it < count) {
            const uint32_t item = queue[it];
            const int owner = (int)(item & 63u), b = (int)(item >> 6);
            const double px = pos[owner * 4], py = pos[owner * 4 + 1];
            const BldLds B = w.b[b];
            double dx2[5], dy2[5];
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const double dx = (px + sp * (double)(i - 2)) - B.cx, dy = (py + sp * (double)(i - 2)) - B.cy;
                dx2[i] = dx * dx;
                dy2[i] = dy * dy;
            }
            uint32_t bk = 0;
#pragma unroll
            for (int i = 0; i < 5; ++i)
#pragma unroll
                for (int j = 0; j < 5; ++j) bk |= ((dx2[i] + dy2[j]) < B.thr) ? (1u << (5 * i + j)) : 0u;
            if (bk) atomicOr(&acc[owner], bk);
        }
    }
    wave_lds_sync();
}

// `subset` selects which candidate cylinders this wavefront handles (all ones: every candidate); two wavefronts with
// complementary subsets split one stencil between them and OR their results.
template <typename MaskT>
__device__ __forceinline__ uint32_t obs_stencil_queued(const WorldLds<MaskT> &w, uint32_t *queue, uint32_t *acc,
                                                       const double *pos, int k, double px, double py, double pz,
                                                       bool active, MaskT subset)
{
    const int lane = (int)threadIdx.x & 63;
    const double sp = k == 0 ? 1.0 : (k == 1 ? 5.0 : 10.0);
    uint32_t mine = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const double xi = px + sp * (double)(i - 2), yi = py + sp * (double)(i - 2);
        mine |= ((xi < 0.0) | (xi > w.W)) ? (0x1Fu << (5 * i)) : 0u;
        mine |= ((yi < 0.0) | (yi > w.W)) ? (0x108421u << i) : 0u;
    }
    if ((pz < 0.0) | (pz > w.Hbox)) mine = 0x1FFFFFFu;
    acc[lane] = 0u;
    MaskT m = active ? (MaskT)(w.g[k][cell_of(w, px, py)] & subset) : (MaskT)0;
    int count = 0;                                   // wave-uniform
    while (__ballot(m != 0) != 0ull) {
        if (count > kObsQueueCap - 64) {
            stencil_queue_drain(w, queue, acc, pos, sp, count);
            count = 0;
        }
        const bool have = m != 0;
        const int b = have ? ctz_mask(m) : 0;
        if (have) m &= (MaskT)(m - 1);
        bool full = false;
        if (have) {
            const BldLds B = w.b[b];
            const double rej2 = w.aux[b].rej2[k], acc2 = w.aux[b].acc2[k];
            const double dcx = px - B.cx, dcy = py - B.cy;
            const double dc2 = dcx * dcx + dcy * dcy;
            if (!(pz > B.H)) {
                if (dc2 < acc2) mine = 0x1FFFFFFu;                  // stencil entirely inside the disc
                else if (dc2 < rej2) full = true;                   // can touch: needs the 25 exact tests
            }
        }
        const unsigned long long bm = __ballot(full);
        if (bm) {
            const int at = count + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bm >> 32),
                                                                  __builtin_amdgcn_mbcnt_lo((uint32_t)bm, 0u));
            if (full) queue[at] = (uint32_t)lane | ((uint32_t)b << 6);
            count += __builtin_popcountll(bm);
        }
    }
    stencil_queue_drain(w, queue, acc, pos, sp, count);
    return mine | acc[lane];
}

// UAV.py:562-566: Threaten_rate(px, py, pz - k), k = 1..5 (all 1 while the UAV is on the ground: z - k < 0)
template <typename MaskT>
__device__ __forceinline__ uint32_t obs_below_bits(const WorldLds<MaskT> &w, double px, double py, double pz)
{
    uint32_t bl = 0;
    const bool xy_out = (px < 0.0) | (px > w.W) | (py < 0.0) | (py > w.W);
#pragma unroll
    for (int k = 1; k <= 5; ++k) {
        const double z = pz - (double)k;
        bl |= (xy_out | (z < 0.0) | (z > w.Hbox)) ? (1u << (k - 1)) : 0u;
    }
    if (bl != 0x1Fu) {                                   // only when the UAV flies above ground level
        MaskT m = w.g[0][cell_of(w, px, py)];
        while (m) {
            const int b = ctz_mask(m);
            m &= (MaskT)(m - 1);
            const BldLds B = w.b[b];
            const double dcx = px - B.cx, dcy = py - B.cy;
            if (dcx * dcx + dcy * dcy < B.thr) {
#pragma unroll
                for (int k = 1; k <= 5; ++k) bl |= !((pz - (double)k) > B.H) ? (1u << (k - 1)) : 0u;
            }
        }
    }
    return bl;
}

// The 20 scalar features of state_PathPlan (UAV.py:518-531, 557-560), as float.
struct ObsScalars {
    float f[20];   // 0..10 -> cols 0..10 ; 11..14 -> cols 86..89
};

// The divisions by 10 / 100 of the reference are multiplications by the f64 reciprocal here: the <= 1 ulp(f64)
// difference vanishes in the f32 / f16 store (an f64 division costs ~15 dependent instructions, there are 14).
__device__ __forceinline__ ObsScalars obs_scalars(const ObsIn &a, double heading)
{
    ObsScalars o;
    o.f[0] = (float)((double)a.step * 0.01);
    bool h0 = a.n_rem >= 1, h1 = a.n_rem >= 2;
    o.f[1] = h0 ? (float)((a.s0x - a.px) * 0.1) : 0.0f;
    o.f[2] = h0 ? (float)((a.s0y - a.py) * 0.1) : 0.0f;
    o.f[3] = h0 ? (float)((a.s0z - a.pz) * 0.1) : 0.0f;
    o.f[4] = (float)a.V;
    o.f[5] = (float)a.vx;
    o.f[6] = (float)a.vy;
    o.f[7] = (float)heading;
    o.f[8] = h1 ? (float)((a.s1x - a.px) * 0.1) : 0.0f;
    o.f[9] = h1 ? (float)((a.s1y - a.py) * 0.1) : 0.0f;
    o.f[10] = h1 ? (float)((a.s1z - a.pz) * 0.1) : 0.0f;
    o.f[11] = (float)((a.gx - a.px) * 0.1);
    o.f[12] = (float)((a.gy - a.py) * 0.1);
    o.f[13] = (float)((a.gz - a.pz) * 0.1);
    o.f[14] = (float)(a.pz * 0.1);
    return o;
}

// Column c (0..99) of the observation row.  `c` is a compile-time
// constant after unrolling, so this folds to one select per column.
__device__ __forceinline__ float obs_col(const ObsScalars &s, const ObsBits &b, int c)
{
    if (c < 11) return s.f[c];
    if (c < 36) return (float)((b.s1 >> (c - 11)) & 1u);
    if (c < 61) return (float)((b.s5 >> (c - 36)) & 1u);
    if (c < 86) return (float)((b.s10 >> (c - 61)) & 1u);
    if (c < 90) return s.f[11 + (c - 86)];
    if (c < 95) return (float)((b.below >> (c - 90)) & 1u);
    return 0.0f;
}

// ---- packed observation rows (UAVENV_OBS_PACKED) -------------------------------------------------------------------
// 75 + 5 of the 100 columns of state_PathPlan are 0/1 occupancy flags and 5 are constant zeros (UAV.py:533-566,517):
// a row is fully described by 15 scalars and 80 bits.  The packed row keeps exactly that, losslessly, in 20 dwords
// (80 B instead of 400 B):
//   dword 0..2   mask words: bit (c & 31) of word (c >> 5) = column c, for the flag columns 11..85 and 90..94
//   dword 3      0
//   dword 4..14  columns 0..10 as f32
//   dword 15..18 columns 86..89 as f32
//   dword 19     0
// Rows are 16-byte aligned (5 chunks); a wavefront's 64 rows are 5 contiguous KiB.
constexpr int kPackedDwords = 20;
constexpr int OBS_KIND_F32 = 0, OBS_KIND_F16 = 1, OBS_KIND_PACKED = 2;

__device__ __forceinline__ void ctile_mask_words(const ObsBits &b, uint32_t &m0, uint32_t &m1, uint32_t &m2);

// Column c (0..99) of the row a packed row stands for; p = its 20 dwords (any address space).
__host__ __device__ __forceinline__ float packed_col(const uint32_t *p, int c)
{
    if (c < 11) return __builtin_bit_cast(float, p[4 + c]);
    if (c >= 86 && c < 90) return __builtin_bit_cast(float, p[15 + (c - 86)]);
    if (c >= 95) return 0.0f;
    return (float)((p[c >> 5] >> (c & 31)) & 1u);
}

// Row-per-lane store of one packed observation: 5 x 16-byte stores.
__device__ __forceinline__ void store_obs_row_packed(void *obs_base, int64_t agent, const ObsScalars &s, const ObsBits &b)
{
    uint4 *row = reinterpret_cast<uint4 *>(reinterpret_cast<uint32_t *>(obs_base) + agent * kPackedDwords);
    uint32_t m0, m1, m2;
    ctile_mask_words(b, m0, m1, m2);
    row[0] = make_uint4(m0, m1, m2, 0u);
    row[1] = make_uint4(__float_as_uint(s.f[0]), __float_as_uint(s.f[1]), __float_as_uint(s.f[2]), __float_as_uint(s.f[3]));
    row[2] = make_uint4(__float_as_uint(s.f[4]), __float_as_uint(s.f[5]), __float_as_uint(s.f[6]), __float_as_uint(s.f[7]));
    row[3] = make_uint4(__float_as_uint(s.f[8]), __float_as_uint(s.f[9]), __float_as_uint(s.f[10]), __float_as_uint(s.f[11]));
    row[4] = make_uint4(__float_as_uint(s.f[12]), __float_as_uint(s.f[13]), __float_as_uint(s.f[14]), 0u);
}

// Row-per-lane store of one observation: 25 x 16-byte stores (f32), 25 x 8-byte (f16) or 5 x 16-byte (packed).
template <int OBS>
__device__ __forceinline__ void store_obs_row(void *obs_base, int64_t agent, const ObsScalars &s, const ObsBits &b)
{
    constexpr bool F16 = OBS == OBS_KIND_F16;
    if (OBS == OBS_KIND_PACKED) {
        store_obs_row_packed(obs_base, agent, s, b);
    } else if (!F16) {
        float4 *row = reinterpret_cast<float4 *>(reinterpret_cast<float *>(obs_base) + agent * 100);
#pragma unroll
        for (int k = 0; k < 25; ++k) {
            float4 v;
            v.x = obs_col(s, b, 4 * k + 0);
            v.y = obs_col(s, b, 4 * k + 1);
            v.z = obs_col(s, b, 4 * k + 2);
            v.w = obs_col(s, b, 4 * k + 3);
            row[k] = v;
        }
    } else {
        uint2 *row = reinterpret_cast<uint2 *>(reinterpret_cast<__half *>(obs_base) + agent * 100);
#pragma unroll
        for (int k = 0; k < 25; ++k) {
            __half2 lo = __floats2half2_rn(obs_col(s, b, 4 * k + 0), obs_col(s, b, 4 * k + 1));
            __half2 hi = __floats2half2_rn(obs_col(s, b, 4 * k + 2), obs_col(s, b, 4 * k + 3));
            uint2 v;
            v.x = *reinterpret_cast<uint32_t *>(&lo);
            v.y = *reinterpret_cast<uint32_t *>(&hi);
            row[k] = v;
        }
    }
}

// Wave-cooperative, fully coalesced store of a wavefront's 64 observation rows (25.6 KB contiguous = 25 x 1 KiB
// store instructions).  The row-per-lane form (store_obs_row) issues 25 store instructions of 64 separate 16-byte
// segments each: ~64 TA cycles per instruction instead of ~16, and partial-line writes at scale (1 M envs: 294 us
// against 198 us).  A full f32 tile in LDS (64 x 101 dwords = 25.9 KB per wavefront) was the first form of this; it
// limited a CU to four single-wave workgroups.  The compact form below holds, per row, only what is not a 0/1 flag
// -- 15 scalars -- plus a column-aligned bit mask of the 80 occupancy flags (23 dwords instead of 101), and the
// streaming pass rebuilds each lane's four columns from them: 5.9 KB per wavefront, ~30 more VALU per store
// instruction.
constexpr int kCTileLd = 23;                      // 3 mask words + 20 scalar slots; odd -> conflict-free column writes
constexpr int kCTileBytes = 64 * kCTileLd * 4;    // 5 888 B per wavefront
// mask words of a compact-tile row from the four bit fields (bit (col & 31) of word (col >> 5) = column col's flag)
__device__ __forceinline__ void ctile_mask_words(const ObsBits &b, uint32_t &m0, uint32_t &m1, uint32_t &m2)
{
    m0 = b.s1 << 11;
    m1 = (b.s1 >> 21) | (b.s5 << 4) | (b.s10 << 29);
    m2 = (b.s10 >> 3) | (b.below << 26);
}

__device__ __forceinline__ void ctile_write_scalars(uint32_t *row, const ObsScalars &s)
{
    float *sl = reinterpret_cast<float *>(row + 3);   // slot = col for cols 0..10, col - 72 for cols 86..89
#pragma unroll
    for (int c = 0; c < 11; ++c) sl[c] = s.f[c];
#pragma unroll
    for (int c = 0; c < 4; ++c) sl[14 + c] = s.f[11 + c];
}

// store instruction `it` (0..24) of the 64-row block: lane handles flat elements it*256 + lane*4 .. +3
template <bool F16>
__device__ __forceinline__ void ctile_emit(void *obs_base, int64_t first_agent, int n_valid, const uint32_t *tile, int it,
                                           bool guard)
{
    const int lane = (int)threadIdx.x & 63;
    const int e = it * 256 + lane * 4;            // flat element index inside the 64 x 100 block of rows
    const int r = e / 100, c = e - r * 100, cg = c >> 2;
    // branch-free: every lane reads one mask word and four scalar slots (clamped addresses), then selects
    const uint32_t *src = tile + r * kCTileLd;
    const int w = c >> 5;
    const uint32_t mw = src[w < 3 ? w : 2];
    const bool edge = (cg == 21) | (cg == 22);
    const int sb = cg < 3 ? c : edge ? c - 72 : 0;
    const float s0 = reinterpret_cast<const float *>(src + 3 + sb)[0];
    const float s1 = reinterpret_cast<const float *>(src + 3 + sb)[1];
    const float s2 = reinterpret_cast<const float *>(src + 3 + sb)[2];
    const float s3 = reinterpret_cast<const float *>(src + 3 + sb)[3];
    const uint32_t nib = w < 3 ? (mw >> (c & 31)) : 0u;
    const uint32_t snib = cg < 2 ? 15u : cg == 2 ? 7u : cg == 21 ? 12u : cg == 22 ? 3u : 0u;
    float v[4];
    v[0] = (snib & 1u) ? s0 : (float)(nib & 1u);
    v[1] = (snib & 2u) ? s1 : (float)((nib >> 1) & 1u);
    v[2] = (snib & 4u) ? s2 : (float)((nib >> 2) & 1u);
    v[3] = (snib & 8u) ? s3 : (float)((nib >> 3) & 1u);
    if (!guard || r < n_valid) {
        if (!F16) {
            *reinterpret_cast<float4 *>(reinterpret_cast<float *>(obs_base) + first_agent * 100 + e) =
                make_float4(v[0], v[1], v[2], v[3]);
        } else {
            __half2 lo = __floats2half2_rn(v[0], v[1]);
            __half2 hi = __floats2half2_rn(v[2], v[3]);
            uint2 o;
            o.x = *reinterpret_cast<uint32_t *>(&lo);
            o.y = *reinterpret_cast<uint32_t *>(&hi);
            *reinterpret_cast<uint2 *>(reinterpret_cast<__half *>(obs_base) + first_agent * 100 + e) = o;
        }
    }
}

// The same store instruction with everything that depends only on (it, lane) -- the row, the mask word and shift, the
// scalar slot base and which of the four columns are scalars -- read from two precomputed words (host side:
// ctile_emit_lut_entry; LDS byte offsets included): ~25 VALU per instruction instead of ~60.  Columns 96..99 take mask word 2 shifted by 31
// (bit 31 of that word is never set), so they need no special case.
__host__ __device__ inline uint2 ctile_emit_lut_entry(int it, int lane)
{
    const int e = it * 256 + lane * 4;
    const int r = e / 100, c = e - r * 100, cg = c >> 2;
    const int w = c >> 5;
    const uint32_t wi = w < 3 ? (uint32_t)w : 2u, sh = w < 3 ? (uint32_t)(c & 31) : 31u;
    const bool edge = (cg == 21) | (cg == 22);
    const uint32_t sb = (uint32_t)(cg < 3 ? c : edge ? c - 72 : 0);
    const uint32_t snib = cg < 2 ? 15u : cg == 2 ? 7u : cg == 21 ? 12u : cg == 22 ? 3u : 0u;
    uint2 L;
    L.x = (uint32_t)(r * kCTileLd + (int)wi) * 4u | (sh << 16) | (snib << 24);       // mask word byte offset, shift, nibble
    L.y = (uint32_t)(r * kCTileLd + 3 + (int)sb) * 4u | ((uint32_t)r << 16);          // scalar slots byte offset, row
    return L;
}

template <bool F16>
__device__ __forceinline__ void ctile_emit_lut(void *obs_base, int64_t first_agent, int n_valid, const uint32_t *tile,
                                               int it, uint2 L, bool guard)
{
    const int lane = (int)threadIdx.x & 63;
    const int e = it * 256 + lane * 4;
    const unsigned char *tb = reinterpret_cast<const unsigned char *>(tile);
    const uint32_t mw = *reinterpret_cast<const uint32_t *>(tb + (L.x & 0xFFFFu));
    const float *ss = reinterpret_cast<const float *>(tb + (L.y & 0xFFFFu));
    float s0 = ss[0], s1 = ss[1], s2 = ss[2], s3 = ss[3];
    asm volatile("" : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3));       // all four LDS reads unconditional: hipcc otherwise
    const uint32_t nib = mw >> ((L.x >> 16) & 31u);                   // wraps some in a branch on their select bit
    float v[4];
    v[0] = (L.x & (1u << 24)) ? s0 : (float)(nib & 1u);
    v[1] = (L.x & (1u << 25)) ? s1 : (float)((nib >> 1) & 1u);
    v[2] = (L.x & (1u << 26)) ? s2 : (float)((nib >> 2) & 1u);
    v[3] = (L.x & (1u << 27)) ? s3 : (float)((nib >> 3) & 1u);
    if (!guard || (int)(L.y >> 16) < n_valid) {
        if (!F16) {
            *reinterpret_cast<float4 *>(reinterpret_cast<float *>(obs_base) + first_agent * 100 + e) =
                make_float4(v[0], v[1], v[2], v[3]);
        } else {
            __half2 lo = __floats2half2_rn(v[0], v[1]);
            __half2 hi = __floats2half2_rn(v[2], v[3]);
            uint2 o;
            o.x = *reinterpret_cast<uint32_t *>(&lo);
            o.y = *reinterpret_cast<uint32_t *>(&hi);
            *reinterpret_cast<uint2 *>(reinterpret_cast<__half *>(obs_base) + first_agent * 100 + e) = o;
        }
    }
}

// 16-byte chunk `chunk` (0 .. 5 * rows - 1) of the packed image of compact-tile rows: row = chunk / 5, part = chunk % 5.
// The tile's scalar slots are column-indexed (slot c for columns 0..10, slot c - 72 for 86..89).
__device__ __forceinline__ uint4 ctile_packed_chunk(const uint32_t *tile, int chunk)
{
    const int r = chunk / 5, part = chunk - r * 5;
    const uint32_t *src = tile + r * kCTileLd;
    // part 0: masks | part 1: cols 0..3 | part 2: cols 4..7 | part 3: cols 8, 9, 10, 86 | part 4: cols 87, 88, 89, 0
    // (arithmetic + an unconditional fourth load: the five-way ?: and the conditional load were a divergent branch each, in every
    // packed-row step kernel's copy-out; b3 <= 21 stays inside the row of kCTileLd = 23 dwords)
    const int b0 = 4 * part - (part != 0 ? 1 : 0) + (part == 4 ? 3 : 0);          // 0, 3, 7, 11, 18
    const int b3 = part == 3 ? 17 : b0 + 3;
    uint4 v;
    v.x = src[b0];
    v.y = src[b0 + 1];
    v.z = src[b0 + 2];
    const uint32_t w3 = src[b3];
    v.w = (part == 0 || part == 4) ? 0u : w3;
    return v;
}

// a wavefront's 64 packed rows from its compact tile: 5 fully coalesced 1 KiB stores
__device__ __forceinline__ void ctile_emit_packed_wave(void *obs_base, int64_t first_agent, int n_valid, const uint32_t *tile)
{
    const int lane = (int)threadIdx.x & 63;
    uint4 *dst = reinterpret_cast<uint4 *>(reinterpret_cast<uint32_t *>(obs_base) + first_agent * kPackedDwords);
    if (n_valid >= 64) {                  // wavefront-uniform: every wavefront but a launch's last stores unguarded
#pragma unroll
        for (int it = 0; it < 5; ++it) dst[it * 64 + lane] = ctile_packed_chunk(tile, it * 64 + lane);
        return;
    }
#pragma unroll
    for (int it = 0; it < 5; ++it) {
        const int chunk = it * 64 + lane;
        const uint4 v = ctile_packed_chunk(tile, chunk);
        if (chunk < 5 * n_valid) dst[chunk] = v;
    }
}

template <int OBS>
__device__ __forceinline__ void store_obs_ctile(void *obs_base, int64_t first_agent, int n_valid, uint32_t *tile,
                                                const ObsScalars &s, const ObsBits &b)
{
    constexpr bool F16 = OBS == OBS_KIND_F16;
    const int lane = (int)threadIdx.x & 63;
    uint32_t *row = tile + lane * kCTileLd;
    ctile_mask_words(b, row[0], row[1], row[2]);
    ctile_write_scalars(row, s);
    wave_lds_sync();
    const int nv = __builtin_amdgcn_readfirstlane(n_valid);      // same in every lane; tell the compiler
    if (OBS == OBS_KIND_PACKED) {
        ctile_emit_packed_wave(obs_base, first_agent, nv < 64 ? nv : 64, tile);
    } else if (nv >= 64) {             // every wavefront but the last takes the unguarded, straight-line form
#pragma unroll
        for (int it = 0; it < 25; ++it) ctile_emit<F16>(obs_base, first_agent, n_valid, tile, it, false);
    } else {
#pragma unroll
        for (int it = 0; it < 25; ++it) ctile_emit<F16>(obs_base, first_agent, n_valid, tile, it, true);
    }
    wave_lds_sync();
}

// Philox4x32-10 (counter-based; one independent stream per (seed, agent, tick)).
__host__ __device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint2 k)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c.x, p1 = (uint64_t)0xCD9E8D57u * c.z;
        uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
        k.x += 0x9E3779B9u;
        k.y += 0xBB67AE85u;
    }
    return c;
}

// 53-bit uniform in [0,1) from two words (same construction as CPython's random()).
__device__ __forceinline__ double u53(uint32_t a, uint32_t b)
{
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * (1.0 / 9007199254740992.0);
}

// ReplayMemory.sample2 = random.sample(self.memory, batch_size) (BaseClass/replay_buffer.py:48-51): `batch` DISTINCT
// stored transitions, every subset equally likely.  On the device: sample s of update (seed, counter) is P(s), where P
// is a keyed pseudo-random permutation of the D = filled * n_agents stored transitions -- an alternating (unbalanced)
// Feistel network over bits = ceil(log2 D) bits, split la = bits / 2 low bits | lb = bits - la high bits: six rounds,
// even rounds  lo ^= F(hi),  odd rounds  hi ^= F(lo)  (each round is its own inverse, so the whole is a bijection of
// [0, 2^bits)), F = murmur3's 32-bit finaliser keyed per round from Philox(seed, counter); the result is cycle-walked
// back into [0, D) (2^bits < 2 D: fewer than two passes expected, exactly one when D is a power of two).
// Distinct s < D give distinct transitions; s >= D wraps (the reference raises there).  oracle/philox.py restates it.
struct ReplayPerm {
    uint32_t k[6];
    uint32_t D, la, lb;
    uint32_t n_agents, magic;      // slot / n_agents by multiplication: magic = floor(2^32 / n_agents)
};

__host__ __device__ __forceinline__ uint32_t fmix32(uint32_t h)
{
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}

// (host-callable: the round keys are the same for every sample of an update, so the launch code derives them once and
// passes them as kernel arguments -- two Philox chains less on every wavefront's critical path)
__host__ __device__ __forceinline__ ReplayPerm replay_perm(uint64_t seed, uint64_t counter, uint32_t D, uint32_t n_agents)
{
    ReplayPerm p;
    const uint2 key = make_uint2((uint32_t)seed, (uint32_t)(seed >> 32));
    const uint4 a = philox4x32_10(make_uint4(0u, (uint32_t)counter, (uint32_t)(counter >> 32), 0x5a3bu), key);
    const uint4 b = philox4x32_10(make_uint4(1u, (uint32_t)counter, (uint32_t)(counter >> 32), 0x5a3bu), key);
    p.k[0] = a.x; p.k[1] = a.y; p.k[2] = a.z; p.k[3] = a.w; p.k[4] = b.x; p.k[5] = b.y;
    p.D = D;
    uint32_t bits = 2u;                                                            // ceil(log2 D), at least 2
    while (bits < 32u && (1ull << bits) < (uint64_t)D) ++bits;
    p.la = bits >> 1;
    p.lb = bits - p.la;
    p.n_agents = n_agents ? n_agents : 1u;
    const uint64_t mg = (1ull << 32) / p.n_agents;
    p.magic = mg > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)mg;
    return p;
}

__device__ __forceinline__ uint32_t replay_perm_apply(const ReplayPerm &p, uint32_t s)
{
    if (p.D <= 1u) return 0u;
    uint32_t x = s < p.D ? s : s % p.D;
    const uint32_t la = p.la, lb = p.lb, ma = (1u << la) - 1u, mb = (1u << lb) - 1u;
    do {
        uint32_t lo = x & ma, hi = x >> la;
#pragma unroll
        for (int r = 0; r < 6; r += 2) {
            lo ^= fmix32(hi ^ p.k[r]) >> (32u - la);
            hi ^= fmix32(lo ^ p.k[r + 1]) >> (32u - lb);
        }
        x = (hi << la) | lo;
    } while (x >= p.D);
    return x;
}

// transition slot -> (frame, agent): slot / n_agents + 1 frames behind the ring head
__device__ __forceinline__ void replay_slot_to_frame(const ReplayPerm &p, uint32_t slot, int head, int frames, int &f, int &agent)
{
    uint32_t back = __umulhi(slot, p.magic);           // floor(slot * floor(2^32 / n) / 2^32): at most 1 short (slot < 2^32)
    uint32_t rem = slot - back * p.n_agents;
    if (rem >= p.n_agents) { rem -= p.n_agents; back += 1u; }
    if (rem >= p.n_agents) { rem -= p.n_agents; back += 1u; }
    agent = (int)rem;
    f = head - 1 - (int)back;
    if (f < 0) f += frames;
}

}  // namespace uav
