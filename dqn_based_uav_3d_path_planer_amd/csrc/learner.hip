// learner.hip -- fused DQN / DDQN / Dueling-DQN update for the reference's Q-MLPs (100-64-A) on gfx950.
//
// Replaces, per update, the ~45 PyTorch launches of  sample -> q_local(s), q_target(s') [, q_local(s')] ->
// TD target -> MSE -> backward -> Adam -> hard target copy  (Trainer/DQN_Trainer.py:85-136,
// DDQN_Trainer.py:72-117, DuelingDQN_Trainer.py:99-190, nets BaseClass/BaseCNN.py:93-139) by three kernels:
//
//   k_dqn_grad   one workgroup per 64 sampled transitions (persistent over tiles when the batch has more than 512):
//                draws the samples (same permutation as uavenv_replay_sample), gathers the two 400 B rows straight
//                from the replay ring into LDS; each of the four wavefronts then runs the forward passes, the TD
//                target and dL/dH of ITS 16 samples without leaving its registers (v_mfma_f32_16x16x4_f32 on
//                H^T = W1 X^T: lane = sample, registers = hidden units), the weight-gradient products run on the
//                same MFMA over all 64 samples, and ONE partial-gradient row is written per workgroup -- no
//                atomics, deterministic.  f32 MFMA = exact f32 FMA chains (the reference's precision).
//   k_dqn_reduce sums the partial rows -> raw[P+2] = gradient sums, loss sum, valid count (this flat vector is
//                what multi-GPU all-reduces over RCCL: the mean is then over the valid samples of ALL ranks).
//   k_dqn_adam   normalises by the valid count, torch.optim.Adam step (+ hard target copy every Update_loop).
//   k_dqn_act    Q(s) for all N envs + epsilon-greedy in one launch (same wave-strip MFMA forward).
//
// The generic PyTorch-ROCm learner (learner.py) remains for other shapes and as the numerical cross-check.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/uavenv.h"
#include "uavenv_device.hpp"
#include "qnet_device.hpp"
#include "dqn_internal.hpp"

using namespace uav;
using namespace uavq;

namespace {

struct GradArgs {
    UavReplayRing ring;
    int head, filled, batch;
    uint64_t seed, counter;
    ReplayPerm perm;                 // the update's sample permutation (round keys derived on the host)
    const int32_t *explicit_idx;     // nullable: batch x (frame, agent) overriding the draws
    const float *local, *target;     // flat parameter blocks
    int n_actions, dueling, kind;    // kind 0: max_a Q_target(s')   1: Q_target(s', argmax_a Q_local(s'))
    float gamma;
    int huber;
    float *partials;                 // [gridDim.x][stride]: parameter layout, then loss sum and valid count
    int P;
    unsigned long long *dbg;         // diagnostics build: 8 s_memtime stamps per workgroup
    // prioritised replay (nullable): importance-sampling weight of sample s in the loss (mean_s w_s loss_s); |TD error| of
    // sample s out (ReplayTree.batch_update's input)
    const float *is_w;
    float *abs_td;
    const float *img;                // nullable: q_local's and q_target's layer 1 in the split form (2 x kSplitF floats, qnet_device.hpp)
};

#ifdef UAVENV_PHASE_PROFILE
#define L_STAMP(slot) do { if (g.dbg && threadIdx.x == 0) g.dbg[(size_t)blockIdx.x * 8 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
// a second bank of eight stamps per workgroup, behind the first (scripts/phase_profile_loop.py allocates both)
#define L_STAMP2(slot) do { if (g.dbg && threadIdx.x == 0) g.dbg[((size_t)gridDim.x + blockIdx.x) * 8 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define L_STAMP(slot) do { } while (0)
#define L_STAMP2(slot) do { } while (0)
#endif


// The scalar fields of the transition in ring row `row`: from its ONE 16-byte record when the ring has them (UavReplayRing.meta,
// ABI 5: {a1, a0, reward, done | valid << 8 | info << 16}, written by the step kernels), else a load each from the action, reward,
// done and valid planes -- four scattered 128-byte lines per sample, ~45 % of a gradient launch's HBM fetch (round 5 counters).
// done / valid come back as integers: converted where they are used.
__device__ __forceinline__ void load_transition(const UavReplayRing &R, uint32_t row, int &act, float &rew, uint32_t &done, uint32_t &valid)
{
    if (R.meta) {
        const uint4 m = reinterpret_cast<const uint4 *>(R.meta)[row];
        act = (int)m.y;
        rew = __uint_as_float(m.z);
        done = m.w & 0xffu;
        valid = (m.w >> 8) & 0xffu;
    } else {
        act = reinterpret_cast<const int32_t *>(R.action)[row];
        rew = R.reward[row];
        done = R.done[row];
        valid = R.valid ? (uint32_t)R.valid[row] : 1u;
    }
}

// =====================================================================================================================
// k_dqn_grad, wave-strip formulation.
//
// Layer 1 is computed transposed, H^T[j][s] = sum_k W1[j][k] X[s][k], on v_mfma_f32_16x16x4_f32: the C/D fragment puts
// sample s = lane & 15 in the lane and hidden units j = 16 t + 4 (lane >> 4) + r (t = tile 0..3, r = register 0..3) in
// its registers.  A wavefront owning 16 samples therefore holds, per lane, one sample's share of the hidden layer,
// and layer 2 (n2 <= 16 outputs), the dueling combine, the TD target, dL/dout and dL/dH are per-lane register
// arithmetic plus two cross-lane adds -- no LDS round trip, no workgroup barrier between the phases (the previous
// formulation spent 2/3 of its 44 k cycles on those).  K is padded to 104: column 100 of the X tile is 1 and column
// 100 of the W1 tile is b1, so the MFMA adds the bias and the weight-gradient product yields db1 as its column 100.
// K index of MFMA step i in lane group g = 26 g + 2 i (+1): each lane reads its operands as 13 aligned float2 per row.
// =====================================================================================================================
// one 16-byte piece (four columns) of a ring row of floats or halfs
template <typename T>
__device__ __forceinline__ floatx4 x_load(const T *obs, uint32_t ring_row, int q)
{
    if (sizeof(T) == 4) {
        return *reinterpret_cast<const floatx4 *>(reinterpret_cast<const float *>(obs) + (size_t)ring_row * kW + 4 * q);
    } else {
        const uint2 raw = *reinterpret_cast<const uint2 *>(reinterpret_cast<const __half *>(obs) + (size_t)ring_row * kW + 4 * q);
        return floatx4{__uint_as_float(raw.x), __uint_as_float(raw.y), 0.0f, 0.0f};            // decoded at commit
    }
}

// The 16 gathered observation rows of this wavefront's strip (f32 / f16 rings; packed rings never build this tile, see
// below): lane l & 15 holds the ring row index of sample l & 15; 400 chunks of four columns, 7 per lane.
template <typename T>
__device__ __forceinline__ void x_issue(floatx4 (&v)[kXIters], const T *obs, uint32_t my_row)
{
    const int lane = (int)threadIdx.x & 63;
#pragma unroll
    for (int it = 0; it < kXIters; ++it) {
        int c = it * 64 + lane;
        c = c < 400 ? c : 399;
        const int row = c / 25, q = c - row * 25;
        const uint32_t ring_row = (uint32_t)__shfl((int)my_row, row, 64);
        v[it] = x_load<T>(obs, ring_row, q);
    }
}

template <typename T>
__device__ __forceinline__ void x_commit(float *strip, floatx4 (&v)[kXIters])
{
    const int lane = (int)threadIdx.x & 63;
#pragma unroll
    for (int it = 0; it < kXIters; ++it) {
        const int c = it * 64 + lane;
        if (c < 400) {
            const int row = c / 25, q = c - row * 25;
            floatx4 w = v[it];
            if (sizeof(T) != 4) {
                const uint32_t rx = __float_as_uint(v[it][0]), ry = __float_as_uint(v[it][1]);
                const __half2 lo = *reinterpret_cast<const __half2 *>(&rx);
                const __half2 hi = *reinterpret_cast<const __half2 *>(&ry);
                w = floatx4{__low2float(lo), __high2float(lo), __low2float(hi), __high2float(hi)};
            }
            *reinterpret_cast<floatx4 *>(strip + row * kLd + 4 * q) = w;
        }
    }
    if (lane < 16) *reinterpret_cast<floatx4 *>(strip + lane * kLd + kW) = floatx4{1.0f, 0.0f, 0.0f, 0.0f};
}

// acc[t][r] = b1[j] + sum_k W1[j][k] X[s][k],  j = 16 t + 4 g + r,  s = this lane's sample (strip row lane & 15).
// One wavefront per SIMD: nothing else hides the LDS latency, so the operands of step i + 1 are requested before the
// eight MFMAs of step i issue (256 cycles of cover), and the scheduler is fenced so that it keeps that order and never
// puts two MFMAs on the same accumulator back to back (40-cycle dependent latency against a 32-cycle issue interval).
__device__ __forceinline__ void fwd_strip(const float *W, const float *strip, floatx4 (&acc)[4])
{
    const int lane = (int)threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    const float *xp = strip + r * kLd + 26 * g;
    const float *wp = W + r * kLd + 26 * g;
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
    float2 b = *reinterpret_cast<const float2 *>(xp);
    float2 a[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) a[t] = *reinterpret_cast<const float2 *>(wp + t * 16 * kLd);
#pragma unroll
    for (int i = 0; i < 13; ++i) {
        float2 bn = b, an[4] = {a[0], a[1], a[2], a[3]};
        if (i + 1 < 13) {
            bn = *reinterpret_cast<const float2 *>(xp + 2 * (i + 1));
#pragma unroll
            for (int t = 0; t < 4; ++t) an[t] = *reinterpret_cast<const float2 *>(wp + t * 16 * kLd + 2 * (i + 1));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = mfma16(a[t].x, b.x, acc[t]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = mfma16(a[t].y, b.y, acc[t]);
        __builtin_amdgcn_sched_barrier(0);
        b = bn;
#pragma unroll
        for (int t = 0; t < 4; ++t) a[t] = an[t];
    }
}

// sum over the 16 lanes of a DPP row (the 16 samples of a strip), result in every lane of the row
__device__ __forceinline__ float row_sum16(float v)
{
#define UAV_ROW_ROR(x, n) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (x)), 0x120 + (n), 0xf, 0xf, false))
    v += UAV_ROW_ROR(v, 8);
    v += UAV_ROW_ROR(v, 4);
    v += UAV_ROW_ROR(v, 2);
    v += UAV_ROW_ROR(v, 1);
#undef UAV_ROW_ROR
    return v;
}

struct Grad2Args {
    GradArgs g;
    int n_tiles, stride;             // tiles of 64 samples in the batch; floats per partial row (multiple of 4)
};

struct GradLds {
    float *W1l, *W1t, *Xs, *Xn, *dHs, *douts, *W2l, *W2t, *b2l, *b2t, *red;
};

template <int NMAX>
struct GradAcc {
    floatx4 acc1[7], acc2;           // dW1^T tiles (k columns 16 u + 4 gq + rr, hidden unit 16 wv + r), dW2^T tile
    float csum[NMAX + 2];            // column sums of dout (db2), loss sum, valid count -- of this wave's strips
};

// TD target, loss, dL/dout of this lane's sample, the strip's column sums, dL/dH and the forward H into LDS -- everything
// between the forward passes and the weight-gradient products; common to the f32-tile and the packed-row kernels.
// HALF_T: H, dH and dout go to LDS transposed and as f16 (hrow / drow / dout_row then point at column s of the [j][s],
// [j][s] and [a][s] tiles of row stride kLdT halfs) -- the operand layout of the f16 weight-gradient MFMAs.
constexpr int kLdT = 72;                    // halfs per row of the transposed f16 tiles (64 samples + 8: 16-byte rows)
// the bootstrap value of the TD target: max_a Q_target(s', a), or Q_target(s', a*) with a* from q_local (double DQN)
template <int NMAX>
__device__ __forceinline__ float pick_qn(const GradArgs &g, const float (&qt)[NMAX], int best)
{
    float qn = qt[0];
#pragma unroll
    for (int a = 1; a < NMAX; ++a) {
        if (g.kind == 1) { if (a == best) qn = qt[a]; }
        else if (a < g.n_actions && qt[a] > qn) qn = qt[a];
    }
    return qn;
}

template <int NMAX, bool HALF_T = false>
__device__ __forceinline__ void td_backward(const GradArgs &g, const float *W2l, const floatx4 (&hl)[4], const W2Frag<NMAX> &Fl,
                                            const float (&ql)[NMAX], float qn, int p_act, float p_rew,
                                            float p_done, float p_valid, float p_w, int smp, GradAcc<NMAX> &A, float *hrow,
                                            float *drow, float *dout_row, float *lane_amax = nullptr)
{
    constexpr bool kKeepW2 = NMAX <= 4;
    float amax = 0.0f;                                // max |dL/dH| of this lane's 16 hidden units (lane_amax != nullptr only)
    const int gq = ((int)threadIdx.x & 63) >> 4;
    const int n2 = g.n_actions + (g.dueling ? 1 : 0);
    // ---- TD target, loss, dL/dout of this lane's sample (Trainer/DQN_Trainer.py:107-119)
    // Q(s, a_taken) as a one-hot dot product (exact: one term times 1, the others times 0): a compare-and-select chain
    // over ql[] is turned by the optimiser into an indexed load from a scratch-memory copy of the array
    float qa = 0.0f;
#pragma unroll
    for (int a = 0; a < NMAX; ++a) qa = fmaf(ql[a], a == p_act ? 1.0f : 0.0f, qa);
    const float y = p_rew + (g.gamma * qn * (1.0f - p_done));
    const float delta = qa - y;
    float per, dq;
    if (g.huber) {
        const float ad = fabsf(delta);
        per = ad < 1.0f ? 0.5f * delta * delta : ad - 0.5f;
        dq = ad < 1.0f ? delta : (delta > 0.0f ? 1.0f : -1.0f);
    } else {
        per = delta * delta;                          // MSELoss (BaseTrainer.py:40)
        dq = 2.0f * delta;
    }
    dq *= p_valid;
    dq *= p_w;                                        // (x 1 without prioritised replay: bit-identical)
    if (g.abs_td && gq == 0) g.abs_td[smp] = fabsf(delta);
    float dv[NMAX + 2];
    const float inv_a = 1.0f / (float)g.n_actions;
#pragma unroll
    for (int a = 0; a < NMAX; ++a) {
        float d = 0.0f;
        if (g.dueling) {                              // Q = V + A - mean(A)
            if (a < g.n_actions) d = dq * ((a == p_act ? 1.0f : 0.0f) - inv_a);
            else if (a == g.n_actions) d = dq;
        } else if (a == p_act) {
            d = dq;
        }
        dv[a] = d;
    }
    dv[NMAX] = per * p_valid * p_w;                   // loss and valid count ride along as two more columns
    dv[NMAX + 1] = p_valid;
    // column sums over the strip's 16 samples (db2, loss sum, valid count): four DPP row rotations each
#pragma unroll
    for (int a = 0; a < NMAX + 2; ++a) {
        if (a < n2 || a >= NMAX) {
            A.csum[a] += row_sum16(dv[a]);
        }
    }
    // ---- dL/dH = relu'(h) * W2^T dout, and the forward H, into LDS for the weight-gradient products
    wave_lds_sync();
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        floatx4 dh = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int a = 0; a < NMAX; ++a) {
            if (a < n2) {
                const floatx4 wv4 = kKeepW2 ? Fl.w[a][t]
                                            : *reinterpret_cast<const floatx4 *>(W2l + a * kHid + 16 * t + 4 * gq);
                dh[0] = fmaf(dv[a], wv4[0], dh[0]); dh[1] = fmaf(dv[a], wv4[1], dh[1]);
                dh[2] = fmaf(dv[a], wv4[2], dh[2]); dh[3] = fmaf(dv[a], wv4[3], dh[3]);
            }
        }
        floatx4 hh;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            hh[rr] = hl[t][rr] > 0.0f ? hl[t][rr] : 0.0f;
            dh[rr] = hl[t][rr] > 0.0f ? dh[rr] : 0.0f;
            if (lane_amax) amax = fmaxf(amax, fabsf(dh[rr]));
        }
        if (HALF_T) {
            _Float16 *hT = reinterpret_cast<_Float16 *>(hrow), *dT = reinterpret_cast<_Float16 *>(drow);
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                hT[(16 * t + 4 * gq + rr) * kLdT] = (_Float16)hh[rr];
                dT[(16 * t + 4 * gq + rr) * kLdT] = (_Float16)dh[rr];
            }
        } else {
            *reinterpret_cast<floatx4 *>(hrow + 16 * t + 4 * gq) = hh;
            *reinterpret_cast<floatx4 *>(drow + 16 * t + 4 * gq) = dh;
        }
    }
    if (HALF_T) {       // doutT[a][s], a < NMAX (rows above stay zero); every lane group writes the same values
        _Float16 *oT = reinterpret_cast<_Float16 *>(dout_row);
#pragma unroll
        for (int a = 0; a < NMAX; ++a)
            if ((a & 3) == gq) oT[a * kLdT] = (_Float16)dv[a];
    } else {            // douts[s][4 gq .. 4 gq + 3]
        floatx4 d4 = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int a = 0; a < NMAX; ++a)
            if ((a >> 2) == gq) d4[a & 3] = dv[a];
        *reinterpret_cast<floatx4 *>(dout_row + 4 * gq) = d4;
    }
    if (lane_amax) *lane_amax = amax;
}

// One tile of 64 transitions.  FIRST: also stages the weights (their loads fly together with the observation rows).
template <typename ObsT, int NMAX, bool FIRST>
__device__ __forceinline__ void grad_tile(const GradArgs &g, const GradLds &L, int tile, bool more, GradAcc<NMAX> &A)
{
    const int tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6, r = lane & 15, gq = lane >> 4;
    const int n2 = g.n_actions + (g.dueling ? 1 : 0);
    const ObsT *obs = reinterpret_cast<const ObsT *>(g.ring.obs);
    float *xs_strip = L.Xs + wv * kStripF, *xn_strip = L.Xn + wv * kStripF;
    // every global load in flight before the first LDS write.  The weights go first: they do not depend on the draw,
    // whose Feistel rounds (a serial chain of ~100 integer operations) then run under their round trip.
    floatx4 vXs[kXIters], vXn[kXIters], vWl[kStageIters], vWt[kStageIters];
    float pb1 = 0.0f, pb1t = 0.0f, pw[4] = {0, 0, 0, 0}, pt[4] = {0, 0, 0, 0}, pb2 = 0.0f, pb2t = 0.0f;
    if (FIRST) {
        // (the small vectors right behind the local fc1: vmcnt retires in order, so anything issued after the s' rows
        // and the target fc1 would make the first LDS commit wait for ALL of the tile's loads)
        w_issue(vWl, g.local);
        const NetDev nl = net_view(g.local, n2), nt = net_view(g.target, n2);
        const int kb = tid < kHid ? tid : kHid - 1;
        pb1 = nl.b1[kb]; pb1t = nt.b1[kb];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int idx = tid + 256 * k < n2 * kHid ? tid + 256 * k : 0;
            pw[k] = nl.W2[idx]; pt[k] = nt.W2[idx];
        }
        const int kq = tid < n2 ? tid : 0;
        pb2 = nl.b2[kq]; pb2t = nt.b2[kq];
    }
    // ---- the 16 transitions of this wavefront's strip (lanes l, l + 16, l + 32, l + 48 hold the same sample)
    const int smp = tile * kTile + wv * 16 + r;
    int f, agent;
    if (g.explicit_idx) {
        f = g.explicit_idx[2 * smp];
        agent = g.explicit_idx[2 * smp + 1];
    } else {
        replay_slot_to_frame(g.perm, replay_perm_apply(g.perm, (uint32_t)smp), g.head, g.ring.frames, f, agent);
    }
    int fn = f + 1;
    if (fn >= g.ring.frames) fn = 0;
    const uint32_t row_s = (uint32_t)f * (uint32_t)g.ring.n_agents + (uint32_t)agent;
    const uint32_t row_n = (uint32_t)fn * (uint32_t)g.ring.n_agents + (uint32_t)agent;
    x_issue<ObsT>(vXs, obs, row_s);
    // this sample's scalar fields (needed at the TD target; behind the s rows, ahead of everything the first commit
    // does not wait for)
    int p_act;
    float p_rew;
    uint32_t raw_done, raw_valid;
    load_transition(g.ring, row_s, p_act, p_rew, raw_done, raw_valid);
    const float p_done = (float)raw_done, p_valid = (float)raw_valid;
    const float p_w = g.is_w ? g.is_w[smp] : 1.0f;
    x_issue<ObsT>(vXn, obs, row_n);
    if (FIRST) w_issue(vWt, g.target);

    if (FIRST) w_commit(L.W1l, vWl, pb1);
    x_commit<ObsT>(xs_strip, vXs);
    wave_lds_sync();
    if (FIRST) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (tid + 256 * k < n2 * kHid) { L.W2l[tid + 256 * k] = pw[k]; L.W2t[tid + 256 * k] = pt[k]; }
        if (tid < n2) { L.b2l[tid] = pb2; L.b2t[tid] = pb2t; }
        __syncthreads();                      // local weights staged (the target fc1 and the s' rows are still in flight)
    }
    L_STAMP(1);
    // ---- forward of q_local on s: the pre-activations stay in registers for the backward pass
    floatx4 hl[4];
    fwd_strip(L.W1l, xs_strip, hl);
    L_STAMP(6);
    // fc2 of q_local in registers: layer 2 of s (and s'), and dL/dH below.  (The general variant, up to 14 outputs,
    // cannot afford to keep 238 registers alive across the forward passes and re-reads LDS each time.)
    constexpr bool kKeepW2 = NMAX <= 4;
    W2Frag<NMAX> Fl;
    w2_load<NMAX>(Fl, L.W2l, L.b2l, n2);
    float ql[NMAX];
    q_strip<NMAX>(hl, Fl, n2, g.n_actions, g.dueling, ql);
    x_commit<ObsT>(xn_strip, vXn);
    if (FIRST) {
        w_commit(L.W1t, vWt, pb1t);
        __syncthreads();                      // target weights staged
    }
    wave_lds_sync();
    L_STAMP(2);
    int best = 0;
    floatx4 ht[4];
    if (g.kind == 1) {                        // double DQN: a* = argmax_a Q_local(s', a)   (DDQN_Trainer.py:94)
        fwd_strip(L.W1l, xn_strip, ht);
        float qn_l[NMAX];
        if (!kKeepW2) w2_load<NMAX>(Fl, L.W2l, L.b2l, n2);
        q_strip<NMAX>(ht, Fl, n2, g.n_actions, g.dueling, qn_l);
        float bq = qn_l[0];
#pragma unroll
        for (int a = 1; a < NMAX; ++a)
            if (a < g.n_actions && qn_l[a] > bq) { bq = qn_l[a]; best = a; }           // torch.max: first maximum
    }
    fwd_strip(L.W1t, xn_strip, ht);
    L_STAMP(7);
    float qt[NMAX];
    {
        W2Frag<NMAX> Ft;
        w2_load<NMAX>(Ft, L.W2t, L.b2t, n2);
        q_strip<NMAX>(ht, Ft, n2, g.n_actions, g.dueling, qt);
    }
    L_STAMP(3);
    // H of sample r of this strip goes where its s' rows were (dead now)
    td_backward<NMAX>(g, L.W2l, hl, Fl, ql, pick_qn<NMAX>(g, qt, best), p_act, p_rew, p_done, p_valid, p_w, smp, A, xn_strip + r * kLh,
                      L.dHs + (wv * 16 + r) * kLh, L.douts + (wv * 16 + r) * kMaxOut);
    __syncthreads();                                  // H, dH, dout of all 64 samples visible
    L_STAMP(4);
    // ---- weight gradients over the 64 samples (K index of MFMA step kk in lane group g: sample (kk & 3) + 16 (kk >> 2) + 4 g)
    //   dW1^T[k][j] += sum_s X[s][k] dH[s][j]   (7 tiles of 16 k-columns, hidden units 16 wv .. 16 wv + 15)
    //   dW2^T[j][a] += sum_s H[s][j] dout[s][a]
    {
        const float *xa = L.Xs + 4 * gq * kLd + r;
        const float *db = L.dHs + 4 * gq * kLh + 16 * wv + r;
        const float *ha = L.Xn + 4 * gq * kLh + 16 * wv + r;              // + strip * kStripF + (kk & 3) * kLh
        const float *ob = L.douts + 4 * gq * kMaxOut + r;
        // operands of step kk + 1 are requested before the eight MFMAs of step kk (same reasoning as fwd_strip)
        float bdh = db[0], ah = ha[0], bo = ob[0], ax[7];
#pragma unroll
        for (int u = 0; u < 7; ++u) ax[u] = xa[16 * u];
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            float bdh_n = bdh, ah_n = ah, bo_n = bo, ax_n[7];
#pragma unroll
            for (int u = 0; u < 7; ++u) ax_n[u] = ax[u];
            if (kk + 1 < 16) {
                const int k1 = kk + 1, s1 = (k1 & 3) + 16 * (k1 >> 2);
                bdh_n = db[s1 * kLh];
#pragma unroll
                for (int u = 0; u < 7; ++u) ax_n[u] = xa[s1 * kLd + 16 * u];
                ah_n = ha[(k1 >> 2) * kStripF + (k1 & 3) * kLh];
                bo_n = ob[s1 * kMaxOut];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 7; ++u) A.acc1[u] = mfma16(ax[u], bdh, A.acc1[u]);
            A.acc2 = mfma16(ah, bo, A.acc2);
            __builtin_amdgcn_sched_barrier(0);
            bdh = bdh_n; ah = ah_n; bo = bo_n;
#pragma unroll
            for (int u = 0; u < 7; ++u) ax[u] = ax_n[u];
        }
    }
    if (more) __syncthreads();                        // the next tile overwrites Xs / Xn / dHs / douts
}

// the partial-gradient row of this workgroup: dW1 | db1 | dW2 | db2 | loss sum | valid count
template <int NMAX>
__device__ __forceinline__ void grad_write_partials(const GradArgs &g, int stride, float *red, const GradAcc<NMAX> &A)
{
    const int tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6, r = lane & 15, gq = lane >> 4;
    const int n2 = g.n_actions + (g.dueling ? 1 : 0);
    // ---- the partial-gradient row of this workgroup
    float *out = g.partials + (size_t)blockIdx.x * stride;
    const int oW2 = kHid * kW + kHid, ob2 = oW2 + n2 * kHid;
    {
        const int j = 16 * wv + r;
#pragma unroll
        for (int u = 0; u < 6; ++u)
            *reinterpret_cast<floatx4 *>(out + j * kW + 16 * u + 4 * gq) = A.acc1[u];
        if (gq == 0) *reinterpret_cast<floatx4 *>(out + j * kW + 96) = A.acc1[6];
        else if (gq == 1) out[kHid * kW + j] = A.acc1[6][0];            // column 100 = X's ones column -> db1[j]
        if (r < n2) *reinterpret_cast<floatx4 *>(out + oW2 + r * kHid + 16 * wv + 4 * gq) = A.acc2;
    }
    // column sums: every lane of a wave holds the wave's sums; add the four waves through LDS
    if (lane == 0) {
#pragma unroll
        for (int a = 0; a < NMAX + 2; ++a) red[wv * (kMaxOut + 2) + a] = A.csum[a];
    }
    __syncthreads();
    if (tid < NMAX + 2) {
        const float s = (red[tid] + red[(kMaxOut + 2) + tid]) + (red[2 * (kMaxOut + 2) + tid] + red[3 * (kMaxOut + 2) + tid]);
        if (tid < n2) out[ob2 + tid] = s;
        else if (tid == NMAX) out[g.P] = s;
        else if (tid == NMAX + 1) out[g.P + 1] = s;
    }
}

// =====================================================================================================================
// Packed rings (UAVENV_OBS_PACKED): the kernels consume the 80-byte rows in place.
//
// Expanding packed rows into f32 tiles in LDS costs more VALU + LDS time on a one-wave-per-SIMD kernel than the gather it
// saves.  Instead every lane loads the packed row of ITS sample into registers (five 16-byte loads, no cross-lane
// traffic) and produces the B operand of each forward MFMA step -- two consecutive columns of its sample's row -- with
// a handful of VALU instructions that issue in the shadow of the MFMAs: a flag column is one bit-field extract of the
// 26-bit slice of the flag words belonging to the lane group's column range, a scalar column is a select.  For the
// weight-gradient product, where the observations are the A operand over all 64 samples, the packed rows of s are kept
// in LDS (5 KB instead of 27 KB) and each lane extracts its column from them.  No f32 observation tile exists.
// =====================================================================================================================
struct GradLdsP {
    float *W1l, *W1t, *Hs, *dHs, *douts, *W2l, *W2t, *b2l, *b2t, *red;
    uint32_t *Ps;                    // [64][20] packed rows of s (+ 4 dwords of pad)
};

template <int NMAX, bool FIRST>
__device__ __forceinline__ void grad_tile_packed(const GradArgs &g, const GradLdsP &L, int tile, bool more, GradAcc<NMAX> &A)
{
    const int tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6, r = lane & 15, gq = lane >> 4;
    const int n2 = g.n_actions + (g.dueling ? 1 : 0);
    const uint32_t *obs = reinterpret_cast<const uint32_t *>(g.ring.obs);
    floatx4 vWl[kStageIters], vWt[kStageIters];
    SplitScRegs vSl, vSt;
    float pw[4] = {0, 0, 0, 0}, pt[4] = {0, 0, 0, 0}, pb2 = 0.0f, pb2t = 0.0f;
    if (FIRST) {
        if (g.img) img_issue(vWl, g.img); else w_issue(vWl, g.local);
        const NetDev nl = net_view(g.local, n2), nt = net_view(g.target, n2);
        if (!g.img) { w_issue_sc(vSl, g.local, nl.b1); w_issue_sc(vSt, g.target, nt.b1); }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int idx = tid + 256 * k < n2 * kHid ? tid + 256 * k : 0;
            pw[k] = nl.W2[idx]; pt[k] = nt.W2[idx];
        }
        const int kq = tid < n2 ? tid : 0;
        pb2 = nl.b2[kq]; pb2t = nt.b2[kq];
    }
    // ---- this lane's transition (lanes l, l + 16, l + 32, l + 48 hold the same sample) and its two packed rows
    const int smp = tile * kTile + wv * 16 + r;
    int f, agent;
    if (g.explicit_idx) {
        f = g.explicit_idx[2 * smp];
        agent = g.explicit_idx[2 * smp + 1];
    } else {
        replay_slot_to_frame(g.perm, replay_perm_apply(g.perm, (uint32_t)smp), g.head, g.ring.frames, f, agent);
    }
    int fn = f + 1;
    if (fn >= g.ring.frames) fn = 0;
    const uint32_t row_s = (uint32_t)f * (uint32_t)g.ring.n_agents + (uint32_t)agent;
    const uint32_t row_n = (uint32_t)fn * (uint32_t)g.ring.n_agents + (uint32_t)agent;
    PRow Rs, Rn;
    prow_load(Rs, obs + (size_t)row_s * kPackedDwords);
    int p_act;
    float p_rew;
    uint32_t raw_done, raw_valid;
    load_transition(g.ring, row_s, p_act, p_rew, raw_done, raw_valid);
    const float p_done = (float)raw_done, p_valid = (float)raw_valid;
    const float p_w = g.is_w ? g.is_w[smp] : 1.0f;
    prow_load(Rn, obs + (size_t)row_n * kPackedDwords);
    if (FIRST) { if (g.img) img_issue(vWt, g.img + kSplitF); else w_issue(vWt, g.target); }

    if (FIRST) {
        if (g.img) img_commit(L.W1l, vWl); else w_commit_split(w1split_at(L.W1l), vWl, vSl);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (tid + 256 * k < n2 * kHid) { L.W2l[tid + 256 * k] = pw[k]; L.W2t[tid + 256 * k] = pt[k]; }
        if (tid < n2) { L.b2l[tid] = pb2; L.b2t[tid] = pb2t; }
        __syncthreads();                      // local weights staged (the target fc1 is still in flight)
    }
    L_STAMP(1);
    floatx4 hl[4];
    fwd_strip_split<false>(w1split_at(L.W1l), Rs, hl);
    L_STAMP(6);
    constexpr bool kKeepW2 = NMAX <= 4;
    W2Frag<NMAX> Fl;
    w2_load<NMAX>(Fl, L.W2l, L.b2l, n2);
    float ql[NMAX];
    q_strip<NMAX>(hl, Fl, n2, g.n_actions, g.dueling, ql);
    if (gq == 0) prow_store_lds(L.Ps + (wv * 16 + r) * kPackedDwords, Rs);     // the s rows, for the dW1 product
    if (FIRST) {
        if (g.img) img_commit(L.W1t, vWt); else w_commit_split(w1split_at(L.W1t), vWt, vSt);
        __syncthreads();                      // target weights staged
    }
    L_STAMP(2);
    int best = 0;
    floatx4 ht[4];
    if (g.kind == 1) {                        // double DQN: a* = argmax_a Q_local(s', a)   (DDQN_Trainer.py:94)
        fwd_strip_split<false>(w1split_at(L.W1l), Rn, ht);
        float qn_l[NMAX];
        if (!kKeepW2) w2_load<NMAX>(Fl, L.W2l, L.b2l, n2);
        q_strip<NMAX>(ht, Fl, n2, g.n_actions, g.dueling, qn_l);
        float bq = qn_l[0];
#pragma unroll
        for (int a = 1; a < NMAX; ++a)
            if (a < g.n_actions && qn_l[a] > bq) { bq = qn_l[a]; best = a; }           // torch.max: first maximum
    }
    fwd_strip_split<false>(w1split_at(L.W1t), Rn, ht);
    L_STAMP(7);
    float qt[NMAX];
    {
        W2Frag<NMAX> Ft;
        w2_load<NMAX>(Ft, L.W2t, L.b2t, n2);
        q_strip<NMAX>(ht, Ft, n2, g.n_actions, g.dueling, qt);
    }
    L_STAMP(3);
    td_backward<NMAX>(g, L.W2l, hl, Fl, ql, pick_qn<NMAX>(g, qt, best), p_act, p_rew, p_done, p_valid, p_w, smp, A, L.Hs + (wv * 16 + r) * kLh,
                      L.dHs + (wv * 16 + r) * kLh, L.douts + (wv * 16 + r) * kMaxOut);
    __syncthreads();                                  // H, dH, dout and the packed s rows of all 64 samples visible
    L_STAMP(4);
    // ---- weight gradients over the 64 samples (K index of MFMA step kk in lane group g: sample (kk & 3) + 16 (kk >> 2) + 4 g)
    //   dW1^T[k][j] += sum_s X[s][k] dH[s][j]: X[s][16 u + r] comes out of the packed row of s --
    //     u = 0: scalar (r < 11) or flag; u = 1..4: flags of word u >> 1; u = 5: flags, or scalars 86..89 (r = 6..9);
    //     u = 6: the ones column (r = 4), else 0: a per-lane constant
    //   dW2^T[j][a] += sum_s H[s][j] dout[s][a]
    {
        const uint32_t *pr = L.Ps + 4 * gq * kPackedDwords;
        const int ia = 4 + (r < 11 ? r : 10), ib = 15 + (r < 6 ? 0 : (r > 9 ? 3 : r - 6));
        const bool sc0 = r < 11, sc5 = r >= 6 && r <= 9;
        const float ones = r == 4 ? 1.0f : 0.0f;
        const float *db = L.dHs + 4 * gq * kLh + 16 * wv + r;
        const float *ha = L.Hs + 4 * gq * kLh + 16 * wv + r;
        const float *ob = L.douts + 4 * gq * kMaxOut + r;
        // three-stage pipeline: the LDS reads of step kk + 2, the decode of step kk + 1 (VALU) and the MFMAs of step kk
        // share a scheduling region, VALU and LDS instructions placed in the shadows of the MFMAs
        struct Raw { uintx4 mk; float sa, sb, bdh, ah, bo; };
        auto load = [&](int k) {
            const int s1 = (k & 3) + 16 * (k >> 2);
            Raw w;
            w.mk = *reinterpret_cast<const uintx4 *>(pr + s1 * kPackedDwords);
            w.sa = __uint_as_float(pr[s1 * kPackedDwords + ia]);
            w.sb = __uint_as_float(pr[s1 * kPackedDwords + ib]);
            w.bdh = db[s1 * kLh];
            w.ah = ha[s1 * kLh];
            w.bo = ob[s1 * kMaxOut];
            return w;
        };
        auto decode = [&](const Raw &w, float (&ax)[6]) {
            ax[0] = sc0 ? w.sa : (float)((w.mk[0] >> r) & 1u);
            ax[1] = (float)((w.mk[0] >> (16 + r)) & 1u);
            ax[2] = (float)((w.mk[1] >> r) & 1u);
            ax[3] = (float)((w.mk[1] >> (16 + r)) & 1u);
            ax[4] = (float)((w.mk[2] >> r) & 1u);
            ax[5] = sc5 ? w.sb : (float)((w.mk[2] >> (16 + r)) & 1u);
        };
        Raw cur = load(0), nxt = load(1);
        float ax[6];
        decode(cur, ax);
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            Raw nn = nxt;
            if (kk + 2 < 16) nn = load(kk + 2);
            float axn[6];
            decode(nxt, axn);
#pragma unroll
            for (int u = 0; u < 6; ++u) A.acc1[u] = mfma16(ax[u], cur.bdh, A.acc1[u]);
            A.acc1[6] = mfma16(ones, cur.bdh, A.acc1[6]);
            A.acc2 = mfma16(cur.ah, cur.bo, A.acc2);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                if (k < 6) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            cur = nxt;
            nxt = nn;
#pragma unroll
            for (int u = 0; u < 6; ++u) ax[u] = axn[u];
        }
    }
    if (more) __syncthreads();                        // the next tile overwrites Ps / Hs / dHs / douts
}

template <int NMAX>
__global__ void __launch_bounds__(256) k_dqn_grad_packed(Grad2Args ga)
{
    const GradArgs &g = ga.g;
    extern __shared__ __align__(16) float lds[];
    GradLdsP L;
    L.W1l = lds;                            // [64][108] local fc1 (+ b1 in column 100)
    L.W1t = L.W1l + kTileF;                 // target fc1
    L.Hs = L.W1t + kTileF;                  // [64][68] relu(fc1(s))
    L.dHs = L.Hs + kTile * kLh;             // [64][68] dL/dH
    L.douts = L.dHs + kTile * kLh;          // [64][16] dL/d(layer-2 output)
    L.W2l = L.douts + kTile * kMaxOut;      // [16][64]
    L.W2t = L.W2l + kMaxOut * kHid;
    L.b2l = L.W2t + kMaxOut * kHid;         // [16]
    L.b2t = L.b2l + kMaxOut;
    L.red = L.b2t + kMaxOut;                // [4][18] per-wave column sums
    L.Ps = reinterpret_cast<uint32_t *>(L.red + 4 * (kMaxOut + 2));          // 16-byte aligned (25 704 floats in)
    const int tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6, r = lane & 15, gq = lane >> 4;
    const int n2 = g.n_actions + (g.dueling ? 1 : 0);
    GradAcc<NMAX> A;
    A.acc2 = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int u = 0; u < 7; ++u) A.acc1[u] = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int a = 0; a < NMAX + 2; ++a) A.csum[a] = 0.0f;
    L_STAMP(0);
    const int step = (int)gridDim.x;
    int tile = (int)blockIdx.x;
    grad_tile_packed<NMAX, true>(g, L, tile, tile + step < ga.n_tiles, A);
    for (tile += step; tile < ga.n_tiles; tile += step)
        grad_tile_packed<NMAX, false>(g, L, tile, tile + step < ga.n_tiles, A);
    L_STAMP(5);
    grad_write_partials<NMAX>(g, ga.stride, L.red, A);
}

// ---------------------------------------------------------------------------------------------------------------------
// The packed kernel with EIGHT wavefronts per workgroup (two per SIMD), for at most 4 layer-2 outputs.
// At one 64-sample tile per CU (batch 16 384) a 4-wave workgroup leaves every SIMD with ONE wavefront: each LDS /
// HBM latency, each dependent VALU chain (layer 2, the TD target) is exposed, and the f32 MFMA does not overlap with
// VALU work of the same wavefront.  Here wavefronts 0..3 (group 0) own q_local(s) and the backward pass of strip
// w & 3, wavefronts 4..7 (group 1) own the bootstrap value of the same strip -- q_target(s') (and q_local(s') for the
// double-DQN action choice) -- and hand it over as ONE float per sample; the two groups run concurrently, each SIMD
// interleaving a group-0 and a group-1 wavefront.  Both groups stage weights (group 0 the local fc1, group 1 the target
// fc1) and share the weight-gradient products (group 0: k-column tiles 0, 2, 4, 6; group 1: tiles 1, 3, 5 and dW2).
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void w_issue_half(floatx4 (&v)[kStageIters], const float *W, int t256)
{
#pragma unroll
    for (int it = 0; it < kStageIters; ++it) {
        int c = it * 256 + t256;
        c = c < kStageChunks ? c : kStageChunks - 1;
        v[it] = *reinterpret_cast<const floatx4 *>(W + 4 * c);
    }
}

struct GradAcc8 {
    floatx4 acc[6];                  // group 0: dW1^T k-column tiles 0, 2, 4, 6; group 1: tiles 1, 3, 5 and the dW2^T tile
                                     // (form B of the split products: group 1 all six flag tiles, group 0 the gathered tile and dW2^T)
    float csum[6];                   // group 0: column sums of dout (NMAX = 4), loss sum, valid count of its strips
};

// ---------------------------------------------------------------------------------------------------------------------
// Round 5: the weight-gradient products of k_dqn_grad_packed8 in the SPLIT form.
//   dW1^T[c][j] = sum_s X[s][c] dH[s][j] over the tile's 64 samples.  80 of the 100 columns of X are 0 / 1 flags -- exact in f16 --
//   so for them the product runs on v_mfma_f32_16x16x32_f16 with dH as TWO f16 terms, hi = f16(dH 2^S) and mid = f16((dH 2^S - hi)
//   2^11, met by flags worth 2^-11 so that both terms share one accumulator) (22-23 significant bits relative to the tile's largest |dH|; 2^S, a power of two taken from that maximum, puts it at
//   2^13..2^14: no overflow whatever the loss does, and the scaling is exact): every product is exact, the sums are f32.  The 15
//   scalar columns (0..10, 86..89) and the ones column (-> db1) form ONE gathered 16-column tile on v_mfma_f32_16x16x4_f32 with the
//   unscaled f32 dH, as before; the scalar columns' bits are 0 in the packed words, so the f16 products leave exact zeros there.
//   Per wavefront: 12 f16 MFMAs (16 cycles) + 16 f32 MFMAs (32 cycles) instead of 64 f32 MFMAs -- 1.4 k matrix cycles per SIMD
//   and tile instead of 4.1 k.
// K (= sample) index of a lane group: sample(q, g, i) = 32 q + 16 (i >> 2) + 4 g + (i & 3) for f16 step q, K element i of lane
// group g -- lane groups 4 samples apart are 16 LDS banks apart in the hidden tiles (kLh = 68) and in the packed rows (20 dwords);
// f32 step kk = 8 q + i takes the same sample, so the lane's sixteen dH values serve both products.
// Register layout out (the partial-row writer of the split kernel knows it): group 0: acc[0..2] = column tiles 0, 2, 4 (rows
// 16 u + 4 g + e), acc[3] = the gathered tile (entry 4 g + e: columns 0..10 | 86..89 | ones); group 1: acc[0..2] = tiles 1, 3, 5,
// acc[3] = dW2^T as before.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void grad_products_split8(const GradLdsP &L, const float *amax4, int grp, int strip, int r, int gq,
                                                     GradAcc8 &A)
{
    // ---- the tile's scale (qnet_device.hpp: split_scale)
    float up, down;
    split_scale(fmaxf(fmaxf(amax4[0], amax4[1]), fmaxf(amax4[2], amax4[3])), up, down);
    // ---- every operand of the phase is requested first (one wavefront's dependent LDS round trips are not hidden by anything)
    const uint32_t *pr = L.Ps + 4 * gq * kPackedDwords;
    const float *db = L.dHs + 4 * gq * kLh + 16 * strip + r;
    uintx4 mk[2][8];
    float dh[2][8], a32[2][8], b32[2][8] = {{0, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0}};
    // group 0: A of the f32 product = entry r of the gathered tile: packed dword 4 + r (columns 0..10), 15 + (r - 11) (86..89); entry
    // 15 is the ones column (dword 19 is a zero pad: replaced below).  group 1: A = H[s][16 strip + r], B = dout[s][r]
    const int dsc = r < 15 ? 4 + r : 19;
    const float *ha = L.Hs + 4 * gq * kLh + 16 * strip + r;
    const float *ob = L.douts + 4 * gq * kMaxOut + r;
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int s1 = 32 * q + 16 * (i >> 2) + (i & 3);           // + 4 gq through the base pointers
            mk[q][i] = *reinterpret_cast<const uintx4 *>(pr + s1 * kPackedDwords);
            dh[q][i] = db[s1 * kLh];
            if (grp == 0) {
                a32[q][i] = __uint_as_float(pr[s1 * kPackedDwords + dsc]);
            } else {
                a32[q][i] = ha[s1 * kLh];
                b32[q][i] = ob[s1 * kMaxOut];
            }
        }
    // ---- B of the f16 products: the lane's sixteen dH values, scaled and split
    half8 bh[2], bm[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) split_half8(dh[q], up, bh[q], bm[q]);
    // ---- the flag columns: tiles u = 2 v + grp, v = 0..2 (columns 16 u + r: word u >> 1, bit 16 (u & 1) + r of the row's flag words)
    floatx4 ch[3];
#pragma unroll
    for (int v = 0; v < 3; ++v) ch[v] = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
    const uint32_t sh = (uint32_t)(grp == 0 ? r : 16 + r);            // u & 1 == grp
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        half8 a1[3], at[3];
#pragma unroll
        for (int v = 0; v < 3; ++v) flags_to_half8(mk[q], v, sh, a1[v], at[v]);                  // u >> 1 == v
#pragma unroll
        for (int v = 0; v < 3; ++v) ch[v] = mfma16h(a1[v], bh[q], ch[v]);
#pragma unroll
        for (int v = 0; v < 3; ++v) ch[v] = mfma16h(at[v], bm[q], ch[v]);      // (mid x 2^-11: same accumulator, three MFMAs later)
    }
    // ---- the f32 product: group 0 the gathered scalar tile x dH, group 1 H x dout (dW2^T)
    floatx4 c32 = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
    const bool ones = grp == 0 && r == 15;
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int i = 0; i < 8; ++i)
            c32 = mfma16(ones ? 1.0f : a32[q][i], grp == 0 ? dh[q][i] : b32[q][i], c32);
    // ---- into the persistent accumulators: (hi + mid 2^-11) 2^-S, exact scalings
#pragma unroll
    for (int v = 0; v < 3; ++v)
#pragma unroll
        for (int k = 0; k < 4; ++k) A.acc[v][k] += ch[v][k] * down;
#pragma unroll
    for (int k = 0; k < 4; ++k) A.acc[3][k] += c32[k];
}

// Form B of the split products (SPLIT == 2): the two groups' work is not symmetric -- group 0 carries the TD target and dL/dH while
// group 1 waits -- so group 1 takes ALL six flag tiles and builds their A operands (the flag bits of every sample: they depend on the
// packed rows only) while group 0 is still in td_backward; behind the barrier group 1 has 24 f16 MFMAs + the split of its sixteen dH
// values left, group 0 the two f32 products (the gathered scalar tile x dH, H x dout).  Same operands in the same K order per
// accumulator as form A: bit-identical results.
struct FlagOps {
    half8 one[2][6];                 // (the 2^-11 operand of the mid term is derived from it behind the barrier: two instructions per dword)
};
__device__ __forceinline__ half8 tiny_of(const half8 &one)
{
    uintx4 d = *reinterpret_cast<const uintx4 *>(&one);
#pragma unroll
    for (int k = 0; k < 4; ++k) d[k] = (d[k] << 2) & 0x10001000u;          // 0x3c00 (1.0) -> 0x1000 (2^-11), per half
    return *reinterpret_cast<const half8 *>(&d);
}
__device__ __forceinline__ void flag_ops_build(const GradLdsP &L, int r, int gq, FlagOps &F)
{
    const uint32_t *pr = L.Ps + 4 * gq * kPackedDwords;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        uintx4 mk[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) mk[i] = *reinterpret_cast<const uintx4 *>(pr + (32 * q + 16 * (i >> 2) + (i & 3)) * kPackedDwords);
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            half8 tiny;
            flags_to_half8(mk, u >> 1, (uint32_t)(16 * (u & 1) + r), F.one[q][u], tiny);
        }
    }
}
__device__ __forceinline__ void grad_products_split8b(const GradLdsP &L, const float *amax4, int grp, int strip, int r, int gq,
                                                      const FlagOps &F, GradAcc8 &A)
{
    const float *db = L.dHs + 4 * gq * kLh + 16 * strip + r;
    float dh[2][8];
    if (grp == 1) {
        float up, down;
        split_scale(fmaxf(fmaxf(amax4[0], amax4[1]), fmaxf(amax4[2], amax4[3])), up, down);
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int i = 0; i < 8; ++i) dh[q][i] = db[(32 * q + 16 * (i >> 2) + (i & 3)) * kLh];
        floatx4 ch[6];
#pragma unroll
        for (int u = 0; u < 6; ++u) ch[u] = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            half8 bh, bm;
            split_half8(dh[q], up, bh, bm);
#pragma unroll
            for (int u = 0; u < 6; ++u) ch[u] = mfma16h(F.one[q][u], bh, ch[u]);
#pragma unroll
            for (int u = 0; u < 6; ++u) ch[u] = mfma16h(tiny_of(F.one[q][u]), bm, ch[u]);
        }
#pragma unroll
        for (int u = 0; u < 6; ++u)
#pragma unroll
            for (int k = 0; k < 4; ++k) A.acc[u][k] += ch[u][k] * down;
        return;
    }
    // group 0: the gathered scalar tile (entry r: packed dword 4 + r, 15 = the ones column) x dH, and H x dout (dW2^T)
    const uint32_t *pr = L.Ps + 4 * gq * kPackedDwords;
    const int dsc = r < 15 ? 4 + r : 19;
    const float *ha = L.Hs + 4 * gq * kLh + 16 * strip + r;
    const float *ob = L.douts + 4 * gq * kMaxOut + r;
    float xs[2][8], hh[2][8], bo[2][8];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int s1 = 32 * q + 16 * (i >> 2) + (i & 3);
            dh[q][i] = db[s1 * kLh];
            xs[q][i] = __uint_as_float(pr[s1 * kPackedDwords + dsc]);
            hh[q][i] = ha[s1 * kLh];
            bo[q][i] = ob[s1 * kMaxOut];
        }
    floatx4 csc = floatx4{0.0f, 0.0f, 0.0f, 0.0f}, cw2 = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
    const bool ones = r == 15;
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            csc = mfma16(ones ? 1.0f : xs[q][i], dh[q][i], csc);
            cw2 = mfma16(hh[q][i], bo[q][i], cw2);
        }
#pragma unroll
    for (int k = 0; k < 4; ++k) { A.acc[0][k] += csc[k]; A.acc[1][k] += cw2[k]; }
}

// SPLIT (round 5): the weight-gradient products in the split form -- see grad_products_split8 below.
template <bool FIRST, int SPLIT>
__device__ __forceinline__ void grad_tile_packed8(const GradArgs &g, const GradLdsP &L, float *qn_lds, int tile, bool more,
                                                  GradAcc8 &A)
{
    constexpr int NMAX = 4;
    // (the wavefront index through readfirstlane: the compiler then knows that `grp` branches are wavefront-uniform and lets the two
    // groups' live ranges share registers)
    const int tid = (int)threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), grp = wv >> 2, strip = wv & 3;
    const int r = lane & 15, gq = lane >> 4, t256 = tid & 255;
    const int n2 = g.n_actions + (g.dueling ? 1 : 0);
    const uint32_t *obs = reinterpret_cast<const uint32_t *>(g.ring.obs);
    const float *net = grp == 0 ? g.local : g.target;
    floatx4 vW[kStageIters];
    SplitScRegs vS;
    float pw[4] = {0, 0, 0, 0}, pb2 = 0.0f;
    if (FIRST) {                              // group 0 stages q_local's weights, group 1 q_target's
        if (g.img) img_issue(vW, g.img + (grp == 0 ? 0 : kSplitF)); else w_issue_half(vW, net, t256);
        const NetDev nv = net_view(net, n2);
        if (!g.img) w_issue_sc(vS, net, nv.b1);
#pragma unroll
        for (int k = 0; k < 4; ++k) pw[k] = nv.W2[t256 + 256 * k < n2 * kHid ? t256 + 256 * k : 0];
        pb2 = nv.b2[t256 < n2 ? t256 : 0];
    }
    // (round 5, measured: the weights FIRST -- issuing the row loads ahead of them, so that the rows' HBM round trip runs under the
    // image's, cost 0.35 us per launch: the CU's vector-memory path is the limit while ~100 KB per workgroup are requested, and the
    // draw's serial chain hides under the weights' round trip only in this order)
    L_STAMP2(0);                              // (diagnostics: weight loads issued)
    // ---- this lane's transition: group 0 needs the s row and the transition scalars, group 1 the s' row
    const int smp = tile * kTile + strip * 16 + r;
    int f, agent;
    if (g.explicit_idx) {
        f = g.explicit_idx[2 * smp];
        agent = g.explicit_idx[2 * smp + 1];
    } else {
        replay_slot_to_frame(g.perm, replay_perm_apply(g.perm, (uint32_t)smp), g.head, g.ring.frames, f, agent);
    }
    int fn = f + 1;
    if (fn >= g.ring.frames) fn = 0;
    const uint32_t row_s = (uint32_t)f * (uint32_t)g.ring.n_agents + (uint32_t)agent;
    const uint32_t row_n = (uint32_t)fn * (uint32_t)g.ring.n_agents + (uint32_t)agent;
    L_STAMP2(1);                              // (diagnostics: the draw done)
    PRow R;
    prow_load(R, obs + (size_t)(grp == 0 ? row_s : row_n) * kPackedDwords);
    int p_act = 0;
    float p_rew = 0.0f, p_w = 1.0f;
    uint32_t raw_done = 0u, raw_valid = 1u;   // (converted where they are used: a conversion here would wait for every load above)
    if (grp == 0) {
        load_transition(g.ring, row_s, p_act, p_rew, raw_done, raw_valid);
        p_w = g.is_w ? g.is_w[smp] : 1.0f;
    }
    L_STAMP(6);                               // (diagnostics: every load of the tile issued)
    if (FIRST) {
        if (g.img) img_commit(grp == 0 ? L.W1l : L.W1t, vW); else w_commit_split(w1split_at(grp == 0 ? L.W1l : L.W1t), vW, vS);
        L_STAMP(7);                           // (diagnostics: layer 1 committed -- its loads have arrived)
        float *W2 = grp == 0 ? L.W2l : L.W2t, *b2 = grp == 0 ? L.b2l : L.b2t;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (t256 + 256 * k < n2 * kHid) W2[t256 + 256 * k] = pw[k];
        if (t256 < n2) b2[t256] = pb2;
        __syncthreads();                      // both weight sets staged
    }
    L_STAMP(1);
    floatx4 hl[4];
    W2Frag<NMAX> Fl;
    float ql[NMAX];
    if (grp == 0) {
        // ---- q_local(s): pre-activations stay in registers for the backward pass
        fwd_strip_split<false>(w1split_at(L.W1l), R, hl);
        w2_load<NMAX>(Fl, L.W2l, L.b2l, n2);
        q_strip<NMAX>(hl, Fl, n2, g.n_actions, g.dueling, ql);
        if (gq == 0) prow_store_lds(L.Ps + (strip * 16 + r) * kPackedDwords, R);    // the s rows, for the dW1 product
    } else {
        // ---- the bootstrap value of the TD target from s'
        int best = 0;
        floatx4 ht[4];
        if (g.kind == 1) {                    // double DQN: a* = argmax_a Q_local(s', a)   (DDQN_Trainer.py:94)
            fwd_strip_split<false>(w1split_at(L.W1l), R, ht);
            W2Frag<NMAX> F;
            w2_load<NMAX>(F, L.W2l, L.b2l, n2);
            float qn_l[NMAX];
            q_strip<NMAX>(ht, F, n2, g.n_actions, g.dueling, qn_l);
            float bq = qn_l[0];
#pragma unroll
            for (int a = 1; a < NMAX; ++a)
                if (a < g.n_actions && qn_l[a] > bq) { bq = qn_l[a]; best = a; }       // torch.max: first maximum
        }
        fwd_strip_split<false>(w1split_at(L.W1t), R, ht);
        W2Frag<NMAX> Ft;
        w2_load<NMAX>(Ft, L.W2t, L.b2t, n2);
        float qt[NMAX];
        q_strip<NMAX>(ht, Ft, n2, g.n_actions, g.dueling, qt);
        if (gq == 0) qn_lds[strip * 16 + r] = pick_qn<NMAX>(g, qt, best);
    }
    L_STAMP(2);
    __syncthreads();                                  // bootstrap values handed over
    L_STAMP(3);
    FlagOps FO;
    if (SPLIT == 2 && grp == 1) flag_ops_build(L, r, gq, FO);        // (while group 0 is in td_backward)
    if (grp == 0) {
        GradAcc<NMAX> T;                              // td_backward's accumulator interface: only csum is used here
#pragma unroll
        for (int a = 0; a < NMAX + 2; ++a) T.csum[a] = A.csum[a];
        float amax = 0.0f;
        const float p_done = (float)raw_done, p_valid = (float)raw_valid;
        td_backward<NMAX>(g, L.W2l, hl, Fl, ql, qn_lds[strip * 16 + r], p_act, p_rew, p_done, p_valid, p_w, smp, T,
                          L.Hs + (strip * 16 + r) * kLh, L.dHs + (strip * 16 + r) * kLh, L.douts + (strip * 16 + r) * kMaxOut,
                          SPLIT ? &amax : nullptr);
#pragma unroll
        for (int a = 0; a < NMAX + 2; ++a) A.csum[a] = T.csum[a];
        if (SPLIT) {                                  // max |dL/dH| of this strip: the tile's f16 scale is derived from the four
            amax = wave_max64(amax);
            if (lane == 0) qn_lds[kTile + strip] = amax;
        }
    }
    __syncthreads();                                  // H, dH, dout and the packed s rows of all 64 samples visible
    L_STAMP(4);
    if (SPLIT) {
        if (SPLIT == 2) grad_products_split8b(L, qn_lds + kTile, grp, strip, r, gq, FO, A);
        else grad_products_split8(L, qn_lds + kTile, grp, strip, r, gq, A);
        if (more) __syncthreads();
        return;
    }
    // ---- weight gradients over the 64 samples (MFMA step kk, lane group gq: sample (kk & 3) + 16 (kk >> 2) + 4 gq);
    // hidden units 16 strip + r; group 0: X columns 16 u + r for u = 0, 2, 4, 6; group 1: u = 1, 3, 5, and dW2^T
    {
        const uint32_t *pr = L.Ps + 4 * gq * kPackedDwords;
        const int ia = 4 + (r < 11 ? r : 10), ib = 15 + (r < 6 ? 0 : (r > 9 ? 3 : r - 6));
        const bool sc0 = r < 11, sc5 = r >= 6 && r <= 9;
        const float ones = r == 4 ? 1.0f : 0.0f;
        const int sh = grp == 0 ? r : 16 + r;
        const float *db = L.dHs + 4 * gq * kLh + 16 * strip + r;
        const float *ha = L.Hs + 4 * gq * kLh + 16 * strip + r;
        const float *ob = L.douts + 4 * gq * kMaxOut + r;
        struct Raw { uintx4 mk; float sx, bdh, ah, bo; };
        auto load = [&](int k) {
            const int s1 = (k & 3) + 16 * (k >> 2);
            Raw w;
            w.mk = *reinterpret_cast<const uintx4 *>(pr + s1 * kPackedDwords);
            w.sx = __uint_as_float(pr[s1 * kPackedDwords + (grp == 0 ? ia : ib)]);
            w.bdh = db[s1 * kLh];
            w.ah = ha[s1 * kLh];
            w.bo = ob[s1 * kMaxOut];
            return w;
        };
        auto decode = [&](const Raw &w, float (&ax)[4]) {
            const float b0 = (float)((w.mk[0] >> sh) & 1u), b1 = (float)((w.mk[1] >> sh) & 1u), b2 = (float)((w.mk[2] >> sh) & 1u);
            if (grp == 0) { ax[0] = sc0 ? w.sx : b0; ax[1] = b1; ax[2] = b2; ax[3] = ones; }
            else { ax[0] = b0; ax[1] = b1; ax[2] = sc5 ? w.sx : b2; ax[3] = w.ah; }
        };
        Raw cur = load(0), nxt = load(1);
        float ax[4];
        decode(cur, ax);
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            Raw nn = nxt;
            if (kk + 2 < 16) nn = load(kk + 2);
            float axn[4];
            decode(nxt, axn);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 3; ++u) A.acc[u] = mfma16(ax[u], cur.bdh, A.acc[u]);
            A.acc[3] = mfma16(ax[3], grp == 0 ? cur.bdh : cur.bo, A.acc[3]);      // group 0: ones column; group 1: H x dout
            __builtin_amdgcn_sched_barrier(0);
            cur = nxt;
            nxt = nn;
#pragma unroll
            for (int u = 0; u < 4; ++u) ax[u] = axn[u];
        }
    }
    if (more) __syncthreads();                        // the next tile overwrites Ps / Hs / dHs / douts / qn
}

template <int SPLIT>
__global__ void __launch_bounds__(512) k_dqn_grad_packed8(Grad2Args ga)
{
    UAV_HOT_PRIO();
    constexpr int NMAX = 4;
    const GradArgs &g = ga.g;
    extern __shared__ __align__(16) float lds[];
    GradLdsP L;
    L.W1l = lds;
    L.W1t = L.W1l + kTileF;
    L.Hs = L.W1t + kTileF;
    L.dHs = L.Hs + kTile * kLh;
    L.douts = L.dHs + kTile * kLh;
    L.W2l = L.douts + kTile * kMaxOut;
    L.W2t = L.W2l + kMaxOut * kHid;
    L.b2l = L.W2t + kMaxOut * kHid;
    L.b2t = L.b2l + kMaxOut;
    L.red = L.b2t + kMaxOut;
    L.Ps = reinterpret_cast<uint32_t *>(L.red + 4 * (kMaxOut + 2));
    float *qn_lds = reinterpret_cast<float *>(L.Ps + kTile * kPackedDwords + 4);
    const int tid = (int)threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), grp = wv >> 2, strip = wv & 3;
    const int r = lane & 15, gq = lane >> 4;
    const int n2 = g.n_actions + (g.dueling ? 1 : 0);
    GradAcc8 A;
#pragma unroll
    for (int u = 0; u < 6; ++u) A.acc[u] = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int a = 0; a < 6; ++a) A.csum[a] = 0.0f;
    L_STAMP(0);
    const int step = (int)gridDim.x;
    int tile = (int)blockIdx.x;
    grad_tile_packed8<true, SPLIT>(g, L, qn_lds, tile, tile + step < ga.n_tiles, A);
    for (tile += step; tile < ga.n_tiles; tile += step)
        grad_tile_packed8<false, SPLIT>(g, L, qn_lds, tile, tile + step < ga.n_tiles, A);
    L_STAMP(5);
    // ---- the partial-gradient row of this workgroup: dW1 | db1 | dW2 | db2 | loss sum | valid count
    float *out = g.partials + (size_t)blockIdx.x * ga.stride;
    const int oW2 = kHid * kW + kHid, ob2 = oW2 + n2 * kHid;
    const int j = 16 * strip + r;
    // (round 5, measured: non-temporal stores of the row lengthen this kernel by 0.8 us and shorten the next by as much -- the 6.8 MB
    // of partial rows cross the memory system once either way)
#define UAV_ROW_ST4(ptr, val) (*reinterpret_cast<floatx4 *>(ptr) = (val))
    if (SPLIT == 2) {
        // (grad_products_split8b's register layout) group 1: acc[u] = tile u of the flag columns (exact zeros at the scalar columns, which
        // group 0 stores from its gathered tile: columns 0..10 and 86..89 are left out here); group 0: acc[0] = the gathered tile
        // (entries 0..10 = columns 0..10, 11..14 = 86..89, 15 = db1), acc[1] = dW2^T; columns 96..99 are constant zero
        if (grp == 1) {
            float *t0 = out + j * kW + 4 * gq;
            if (gq == 3) UAV_ROW_ST4(t0, A.acc[0]);                      // columns 12..15
            else if (gq == 2) t0[3] = A.acc[0][3];                       // column 11
#pragma unroll
            for (int u = 1; u < 5; ++u) UAV_ROW_ST4(out + j * kW + 16 * u + 4 * gq, A.acc[u]);
            float *t5 = out + j * kW + 80 + 4 * gq;
            if (gq == 0 || gq == 3) UAV_ROW_ST4(t5, A.acc[5]);
            else if (gq == 1) { t5[0] = A.acc[5][0]; t5[1] = A.acc[5][1]; }
            else { t5[2] = A.acc[5][2]; t5[3] = A.acc[5][3]; }
        } else {
            const floatx4 sc = A.acc[0];
            if (gq < 2) UAV_ROW_ST4(out + j * kW + 4 * gq, sc);                                            // columns 0..7
            else if (gq == 2) { out[j * kW + 8] = sc[0]; out[j * kW + 9] = sc[1]; out[j * kW + 10] = sc[2]; out[j * kW + 86] = sc[3]; }
            else { out[j * kW + 87] = sc[0]; out[j * kW + 88] = sc[1]; out[j * kW + 89] = sc[2]; out[kHid * kW + j] = sc[3]; }
            if (gq == 0) UAV_ROW_ST4(out + j * kW + 96, (floatx4{0.0f, 0.0f, 0.0f, 0.0f}));
            if (r < n2) UAV_ROW_ST4(out + oW2 + r * kHid + 16 * strip + 4 * gq, A.acc[1]);                  // dW2^T
        }
    } else if (SPLIT) {
        // (grad_products_split8's register layout) group 0: tiles 0, 2, 4 + the gathered tile, whose entries 0..10 are columns
        // 0..10 of tile 0 -- held by the same lanes, and exact zeros in the f16 result --, 11..14 columns 86..89 and 15 db1;
        // group 1: tiles 1, 3, 5 without columns 86..89, and dW2^T.  Columns 96..99 are constant zero in every row.
        if (grp == 0) {
            floatx4 t0 = A.acc[0];
#pragma unroll
            for (int k = 0; k < 4; ++k) t0[k] += (4 * gq + k <= 10) ? A.acc[3][k] : 0.0f;
            UAV_ROW_ST4(out + j * kW + 4 * gq, t0);
            UAV_ROW_ST4(out + j * kW + 32 + 4 * gq, A.acc[1]);
            UAV_ROW_ST4(out + j * kW + 64 + 4 * gq, A.acc[2]);
            if (gq == 2) out[j * kW + 86] = A.acc[3][3];
            if (gq == 3) { out[j * kW + 87] = A.acc[3][0]; out[j * kW + 88] = A.acc[3][1]; out[j * kW + 89] = A.acc[3][2];
                           out[kHid * kW + j] = A.acc[3][3]; }
            if (gq == 0) *reinterpret_cast<floatx4 *>(out + j * kW + 96) = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
        } else {
            UAV_ROW_ST4(out + j * kW + 16 + 4 * gq, A.acc[0]);
            UAV_ROW_ST4(out + j * kW + 48 + 4 * gq, A.acc[1]);
            float *t5 = out + j * kW + 80 + 4 * gq;                      // tile 5 = columns 80..95; 86..89 belong to group 0
            if (gq == 0 || gq == 3) *reinterpret_cast<floatx4 *>(t5) = A.acc[2];
            else if (gq == 1) { t5[0] = A.acc[2][0]; t5[1] = A.acc[2][1]; }
            else { t5[2] = A.acc[2][2]; t5[3] = A.acc[2][3]; }
            if (r < n2) *reinterpret_cast<floatx4 *>(out + oW2 + r * kHid + 16 * strip + 4 * gq) = A.acc[3];      // dW2^T
        }
    } else if (grp == 0) {
#pragma unroll
        for (int u = 0; u < 3; ++u) *reinterpret_cast<floatx4 *>(out + j * kW + 32 * u + 4 * gq) = A.acc[u];     // tiles 0, 2, 4
        if (gq == 0) *reinterpret_cast<floatx4 *>(out + j * kW + 96) = A.acc[3];                                // tile 6: columns 96..99
        else if (gq == 1) out[kHid * kW + j] = A.acc[3][0];                                                      // column 100 -> db1[j]
    } else {
#pragma unroll
        for (int u = 0; u < 3; ++u) *reinterpret_cast<floatx4 *>(out + j * kW + 16 + 32 * u + 4 * gq) = A.acc[u]; // tiles 1, 3, 5
        if (r < n2) *reinterpret_cast<floatx4 *>(out + oW2 + r * kHid + 16 * strip + 4 * gq) = A.acc[3];          // dW2^T
    }
    if (grp == 0 && lane == 0) {
#pragma unroll
        for (int a = 0; a < NMAX + 2; ++a) L.red[strip * (kMaxOut + 2) + a] = A.csum[a];
    }
    __syncthreads();
    if (tid < NMAX + 2) {
        const float s = (L.red[tid] + L.red[(kMaxOut + 2) + tid]) + (L.red[2 * (kMaxOut + 2) + tid] + L.red[3 * (kMaxOut + 2) + tid]);
        if (tid < n2) out[ob2 + tid] = s;
        else if (tid == NMAX) out[g.P] = s;
        else if (tid == NMAX + 1) out[g.P + 1] = s;
    }
}

constexpr size_t kGradP8Lds = (size_t)(2 * kTileF + 2 * kTile * kLh + kTile * kMaxOut + 2 * kMaxOut * kHid + 2 * kMaxOut +
                                       4 * (kMaxOut + 2) + kTile * kPackedDwords + 4 + kTile + 4) * 4;      // (+ 4: the strips' max |dH|)

constexpr size_t kGradPLds = (size_t)(2 * kTileF + 2 * kTile * kLh + kTile * kMaxOut + 2 * kMaxOut * kHid + 2 * kMaxOut +
                                      4 * (kMaxOut + 2) + kTile * kPackedDwords + 4) * 4;
static_assert((2 * kTileF + 2 * kTile * kLh + kTile * kMaxOut + 2 * kMaxOut * kHid + 2 * kMaxOut + 4 * (kMaxOut + 2)) % 4 == 0,
              "packed-row staging must start 16-byte aligned");

// =====================================================================================================================
// f16 MFMA learner (UavDqnNet.mfma_dtype = 1; BASELINE configs[2]: "fp16 Q-net MFMA").
//
// Mixed precision as a tensor-core trainer does it: master weights, Adam moments, layer 2, the TD target, the loss and
// dL/dH stay f32; the three big products -- layer 1 of q_local / q_target and dW1 = dH^T X -- run on
// v_mfma_f32_16x16x32_f16 (f16 operands, f32 accumulate, 16x the f32 MFMA rate): fc1 weights are rounded to f16 when
// they are staged, observations are f16 (f16 rings as stored, packed rings converted), H and dH are rounded to f16 for
// the gradient products only.  Same wave-strip structure as above: lane = sample, registers = hidden units.
// Operand tiles (halfs): rows of kLdH = 136 for the K = 128 products of the forward (100 inputs, column 100 = 1 | b1,
// then zeros), rows of kLdT = 72 for the K = 64 (samples) products of the backward, whose operands are TRANSPOSED
// ([column][sample]: an MFMA lane supplies 8 consecutive K elements of one row).
// =====================================================================================================================
constexpr int kLdH = 136;            // (half8 / half4 / mfma16h: qnet_device.hpp)
constexpr int kStageW = 16 * kPackedDwords + 4;   // dwords of packed-row staging per wavefront (+ pad: the expansion over-reads 1)

// four columns 4 q .. 4 q + 3 of the row whose packed image is at pr (LDS)
__device__ __forceinline__ floatx4 packed_expand4(const uint32_t *pr, int q)
{
    const int c = 4 * q, w = c >> 5;
    const uint32_t mw = pr[w < 3 ? w : 2];
    const uint32_t nib = w < 3 ? mw >> (c & 31) : 0u;
    const uint32_t snib = q < 2 ? 15u : q == 2 ? 7u : q == 21 ? 12u : q == 22 ? 3u : 0u;     // which of the four are scalars
    const int sb = q < 3 ? 4 + c : q == 21 ? 13 : q == 22 ? 17 : 4;                          // and where they sit in the row
    const float s0 = __uint_as_float(pr[sb]), s1 = __uint_as_float(pr[sb + 1]);
    const float s2 = __uint_as_float(pr[sb + 2]), s3 = __uint_as_float(pr[sb + 3]);
    floatx4 v;
    v[0] = (snib & 1u) ? s0 : (float)(nib & 1u);
    v[1] = (snib & 2u) ? s1 : (float)((nib >> 1) & 1u);
    v[2] = (snib & 4u) ? s2 : (float)((nib >> 2) & 1u);
    v[3] = (snib & 8u) ? s3 : (float)((nib >> 3) & 1u);
    return v;
}

// fc1 (64 x 100 f32 in HBM) -> f16 tile [64][kLdH]: column 100 = bias, 101..127 = 0
__device__ __forceinline__ void wh_commit(_Float16 *dst, floatx4 (&v)[kStageIters], float bias)
{
#pragma unroll
    for (int it = 0; it < kStageIters; ++it) {
        const int c = it * 256 + (int)threadIdx.x;
        if (c < kStageChunks) {
            const int row = c / 25, q = c - row * 25;
            *reinterpret_cast<half4 *>(dst + row * kLdH + 4 * q) =
                half4{(_Float16)v[it][0], (_Float16)v[it][1], (_Float16)v[it][2], (_Float16)v[it][3]};
        }
    }
    if (threadIdx.x < kHid) {
        _Float16 *row = dst + (int)threadIdx.x * kLdH;
        *reinterpret_cast<half4 *>(row + 100) = half4{(_Float16)bias, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f};
#pragma unroll
        for (int k = 0; k < 3; ++k) *reinterpret_cast<half8 *>(row + 104 + 8 * k) = half8{0, 0, 0, 0, 0, 0, 0, 0};
    }
}

// KIND: OBS_KIND_F16 (rows of 100 halfs) or OBS_KIND_PACKED.  Loads of this wavefront's 16 rows (2 or 7 per lane).
template <int KIND>
__device__ __forceinline__ void xh_issue(floatx4 (&v)[kXIters], const void *obs, uint32_t my_row)
{
    constexpr int per_row = KIND == OBS_KIND_PACKED ? 5 : 25, iters = KIND == OBS_KIND_PACKED ? 2 : kXIters;
    const int lane = (int)threadIdx.x & 63;
#pragma unroll
    for (int it = 0; it < iters; ++it) {
        int c = it * 64 + lane;
        c = c < 16 * per_row ? c : 16 * per_row - 1;
        const int row = c / per_row, q = c - row * per_row;
        const uint32_t ring_row = (uint32_t)__shfl((int)my_row, row, 64);
        if (KIND == OBS_KIND_PACKED) {
            v[it] = *reinterpret_cast<const floatx4 *>(reinterpret_cast<const uint32_t *>(obs) + (size_t)ring_row * kPackedDwords + 4 * q);
        } else {
            const uint2 raw = *reinterpret_cast<const uint2 *>(reinterpret_cast<const __half *>(obs) + (size_t)ring_row * kW + 4 * q);
            v[it] = floatx4{__uint_as_float(raw.x), __uint_as_float(raw.y), 0.0f, 0.0f};
        }
    }
}

// ... into the f16 strip [16][kLdH] (+ column 100 = 1, 101..127 = 0) and, if xT != nullptr, transposed into
// xT[column][sample] (sample column of this strip's first row = xT + 16 w).
template <int KIND>
__device__ __forceinline__ void xh_commit(_Float16 *strip, floatx4 (&v)[kXIters], uint32_t *stage, _Float16 *xT)
{
    const int lane = (int)threadIdx.x & 63;
    if (KIND == OBS_KIND_PACKED) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int c = it * 64 + lane;
            if (c < 80) *reinterpret_cast<floatx4 *>(stage + 4 * c) = v[it];
        }
        wave_lds_sync();
    }
#pragma unroll
    for (int it = 0; it < kXIters; ++it) {
        const int c = it * 64 + lane;
        if (c < 400) {
            const int row = c / 25, q = c - row * 25;
            half4 h;
            if (KIND == OBS_KIND_PACKED) {
                const floatx4 f = packed_expand4(stage + row * kPackedDwords, q);
                h = half4{(_Float16)f[0], (_Float16)f[1], (_Float16)f[2], (_Float16)f[3]};
            } else {
                const uint2 raw = make_uint2(__float_as_uint(v[it][0]), __float_as_uint(v[it][1]));
                h = *reinterpret_cast<const half4 *>(&raw);
            }
            *reinterpret_cast<half4 *>(strip + row * kLdH + 4 * q) = h;
            if (xT) {
#pragma unroll
                for (int e = 0; e < 4; ++e) xT[(4 * q + e) * kLdT + row] = h[e];
            }
        }
    }
    if (lane < 16) {
        _Float16 *row = strip + lane * kLdH;
        *reinterpret_cast<half4 *>(row + 100) = half4{(_Float16)1.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f};
#pragma unroll
        for (int k = 0; k < 3; ++k) *reinterpret_cast<half8 *>(row + 104 + 8 * k) = half8{0, 0, 0, 0, 0, 0, 0, 0};
        if (xT) xT[100 * kLdT + lane] = (_Float16)1.0f;
    }
}

// acc[t][r] = b1[j] + sum_k W1[j][k] X[s][k] on the f16 MFMA: 4 K-steps of 32, 16 MFMAs
__device__ __forceinline__ void fwd_strip_h(const _Float16 *W, const _Float16 *strip, floatx4 (&acc)[4])
{
    const int lane = (int)threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    const _Float16 *xp = strip + r * kLdH + 8 * g;
    const _Float16 *wp = W + r * kLdH + 8 * g;
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const half8 b = *reinterpret_cast<const half8 *>(xp + 32 * kk);
#pragma unroll
        for (int t = 0; t < 4; ++t)
            acc[t] = mfma16h(*reinterpret_cast<const half8 *>(wp + t * 16 * kLdH + 32 * kk), b, acc[t]);
    }
}

struct GradLdsH {
    _Float16 *W1l, *W1t, *Xs, *Xn, *XsT, *HT, *dHT, *doutT;     // HT / dHT alias the Xn tile (dead after the forward passes)
    float *W2l, *W2t, *b2l, *b2t, *red;
    uint32_t *stage;
};

// what a tile needs from HBM: its 2 x 16 observation rows per wavefront and each lane's transition scalars
struct TileLoads {
    floatx4 vXs[kXIters], vXn[kXIters];
    int p_act, smp;
    float p_rew, p_done, p_valid, p_w;
};

template <int KIND>
__device__ __forceinline__ void tile_issue(const GradArgs &g, int tile, TileLoads &T)
{
    const int lane = (int)threadIdx.x & 63, wv = (int)threadIdx.x >> 6, r = lane & 15;
    const int smp = tile * kTile + wv * 16 + r;
    int f, agent;
    if (g.explicit_idx) {
        f = g.explicit_idx[2 * smp];
        agent = g.explicit_idx[2 * smp + 1];
    } else {
        replay_slot_to_frame(g.perm, replay_perm_apply(g.perm, (uint32_t)smp), g.head, g.ring.frames, f, agent);
    }
    int fn = f + 1;
    if (fn >= g.ring.frames) fn = 0;
    const uint32_t row_s = (uint32_t)f * (uint32_t)g.ring.n_agents + (uint32_t)agent;
    const uint32_t row_n = (uint32_t)fn * (uint32_t)g.ring.n_agents + (uint32_t)agent;
    xh_issue<KIND>(T.vXs, g.ring.obs, row_s);
    {
        uint32_t raw_done, raw_valid;
        load_transition(g.ring, row_s, T.p_act, T.p_rew, raw_done, raw_valid);
        T.p_done = (float)raw_done;
        T.p_valid = (float)raw_valid;
    }
    T.p_w = g.is_w ? g.is_w[smp] : 1.0f;
    T.smp = smp;
    xh_issue<KIND>(T.vXn, g.ring.obs, row_n);
}

// One tile.  T holds this tile's loads (issued by the caller / the previous tile); when `more`, the NEXT tile's loads are
// issued into T as soon as this tile's rows have been committed to LDS, so that their HBM round trip runs under the
// TD / backward / gradient-product phases (a workgroup of BASELINE configs[2] walks four tiles).
template <int KIND, int NMAX, bool FIRST>
__device__ __forceinline__ void grad_tile_h(const GradArgs &g, const GradLdsH &L, int tile, int next_tile, bool more,
                                            TileLoads &T, GradAcc<NMAX> &A)
{
    const int tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6, r = lane & 15, gq = lane >> 4;
    const int n2 = g.n_actions + (g.dueling ? 1 : 0);
    _Float16 *xs_strip = L.Xs + wv * 16 * kLdH, *xn_strip = L.Xn + wv * 16 * kLdH;
    floatx4 vWl[kStageIters], vWt[kStageIters];
    float pb1 = 0.0f, pb1t = 0.0f, pw[4] = {0, 0, 0, 0}, pt[4] = {0, 0, 0, 0}, pb2 = 0.0f, pb2t = 0.0f;
    if (FIRST) {
        w_issue(vWl, g.local);
        const NetDev nl = net_view(g.local, n2), nt = net_view(g.target, n2);
        const int kb = tid < kHid ? tid : kHid - 1;
        pb1 = nl.b1[kb]; pb1t = nt.b1[kb];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int idx = tid + 256 * k < n2 * kHid ? tid + 256 * k : 0;
            pw[k] = nl.W2[idx]; pt[k] = nt.W2[idx];
        }
        const int kq = tid < n2 ? tid : 0;
        pb2 = nl.b2[kq]; pb2t = nt.b2[kq];
        tile_issue<KIND>(g, tile, T);
        w_issue(vWt, g.target);
    }
    const int p_act = T.p_act, smp = T.smp;
    const float p_rew = T.p_rew, p_done = T.p_done, p_valid = T.p_valid, p_w = T.p_w;

    uint32_t *stage = L.stage + wv * kStageW;
    if (FIRST) wh_commit(L.W1l, vWl, pb1);
    xh_commit<KIND>(xs_strip, T.vXs, stage, L.XsT + wv * 16);
    wave_lds_sync();
    if (FIRST) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (tid + 256 * k < n2 * kHid) { L.W2l[tid + 256 * k] = pw[k]; L.W2t[tid + 256 * k] = pt[k]; }
        if (tid < n2) { L.b2l[tid] = pb2; L.b2t[tid] = pb2t; }
        for (int k = tid; k < 16 * kLdT / 2; k += 256) reinterpret_cast<uint32_t *>(L.doutT)[k] = 0u;     // rows >= n2 stay zero
        __syncthreads();
    }
    L_STAMP(1);
    floatx4 hl[4];
    fwd_strip_h(L.W1l, xs_strip, hl);
    W2Frag<NMAX> Fl;
    w2_load<NMAX, 3>(Fl, L.W2l, L.b2l, n2);
    float ql[NMAX];
    q_strip<NMAX, 3>(hl, Fl, n2, g.n_actions, g.dueling, ql);
    xh_commit<KIND>(xn_strip, T.vXn, stage, nullptr);
    if (FIRST) {
        wh_commit(L.W1t, vWt, pb1t);
        __syncthreads();
    }
    wave_lds_sync();
    if (more) tile_issue<KIND>(g, next_tile, T);      // the next tile's HBM round trip starts here
    L_STAMP(2);
    int best = 0;
    floatx4 ht[4];
    if (g.kind == 1) {                        // double DQN: a* = argmax_a Q_local(s', a)   (DDQN_Trainer.py:94)
        fwd_strip_h(L.W1l, xn_strip, ht);
        float qn_l[NMAX];
        q_strip<NMAX, 3>(ht, Fl, n2, g.n_actions, g.dueling, qn_l);
        float bq = qn_l[0];
#pragma unroll
        for (int a = 1; a < NMAX; ++a)
            if (a < g.n_actions && qn_l[a] > bq) { bq = qn_l[a]; best = a; }
    }
    fwd_strip_h(L.W1t, xn_strip, ht);
    float qt[NMAX];
    {
        W2Frag<NMAX> Ft;
        w2_load<NMAX, 3>(Ft, L.W2t, L.b2t, n2);
        q_strip<NMAX, 3>(ht, Ft, n2, g.n_actions, g.dueling, qt);
    }
    L_STAMP(3);
    __syncthreads();                                  // every wave is done with its s' rows: HT / dHT overwrite that tile
    const int s = wv * 16 + r;
    td_backward<NMAX, true>(g, L.W2l, hl, Fl, ql, pick_qn<NMAX>(g, qt, best), p_act, p_rew, p_done, p_valid, p_w, smp, A,
                            reinterpret_cast<float *>(L.HT + s), reinterpret_cast<float *>(L.dHT + s),
                            reinterpret_cast<float *>(L.doutT + s));
    __syncthreads();
    L_STAMP(4);
    // ---- weight gradients, K = 64 samples = 2 MFMA steps of 32:
    //   dW1^T[k][j] += sum_s XsT[k][s] dHT[j][s]   (7 tiles of 16 k-columns, hidden units 16 wv + r)
    //   dW2^T[j][a] += sum_s HT[j][s] doutT[a][s]
    {
        const _Float16 *xa = L.XsT + r * kLdT + 8 * gq;
        const _Float16 *db = L.dHT + (16 * wv + r) * kLdT + 8 * gq;
        const _Float16 *ha = L.HT + (16 * wv + r) * kLdT + 8 * gq;
        const _Float16 *ob = L.doutT + r * kLdT + 8 * gq;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const half8 bdh = *reinterpret_cast<const half8 *>(db + 32 * kk);
#pragma unroll
            for (int u = 0; u < 7; ++u)
                A.acc1[u] = mfma16h(*reinterpret_cast<const half8 *>(xa + 16 * u * kLdT + 32 * kk), bdh, A.acc1[u]);
            A.acc2 = mfma16h(*reinterpret_cast<const half8 *>(ha + 32 * kk), *reinterpret_cast<const half8 *>(ob + 32 * kk), A.acc2);
        }
    }
    if (more) __syncthreads();
}

template <int KIND, int NMAX>
__global__ void __launch_bounds__(256) k_dqn_grad_h(Grad2Args ga)
{
    const GradArgs &g = ga.g;
    extern __shared__ __align__(16) float lds[];
    GradLdsH L;
    _Float16 *hb = reinterpret_cast<_Float16 *>(lds);
    L.W1l = hb;                              // [64][136]
    L.W1t = L.W1l + kTile * kLdH;
    L.Xs = L.W1t + kTile * kLdH;
    L.Xn = L.Xs + kTile * kLdH;              // [64][136] = 8 704 halfs; later HT [64][72] + dHT [64][72] = 9 216 halfs
    L.HT = L.Xn;
    L.dHT = L.HT + kTile * kLdT;
    L.XsT = L.Xn + 2 * kTile * kLdT;         // [112][72]
    L.doutT = L.XsT + 112 * kLdT;            // [16][72]
    L.W2l = reinterpret_cast<float *>(L.doutT + 16 * kLdT);
    L.W2t = L.W2l + kMaxOut * kHid;
    L.b2l = L.W2t + kMaxOut * kHid;
    L.b2t = L.b2l + kMaxOut;
    L.red = L.b2t + kMaxOut;
    L.stage = reinterpret_cast<uint32_t *>(L.red + 4 * (kMaxOut + 2));
    // XsT rows 101..111 feed output rows nobody stores; zeroed once so that no NaN bit pattern ever enters an MFMA
    for (int k = (int)threadIdx.x; k < 11 * kLdT / 2; k += 256) reinterpret_cast<uint32_t *>(L.XsT + 101 * kLdT)[k] = 0u;
    GradAcc<NMAX> A;
    A.acc2 = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int u = 0; u < 7; ++u) A.acc1[u] = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int a = 0; a < NMAX + 2; ++a) A.csum[a] = 0.0f;
    L_STAMP(0);
    const int step = (int)gridDim.x;
    int tile = (int)blockIdx.x;
    TileLoads T;
    grad_tile_h<KIND, NMAX, true>(g, L, tile, tile + step, tile + step < ga.n_tiles, T, A);
    for (tile += step; tile < ga.n_tiles; tile += step)
        grad_tile_h<KIND, NMAX, false>(g, L, tile, tile + step, tile + step < ga.n_tiles, T, A);
    L_STAMP(5);
    grad_write_partials<NMAX>(g, ga.stride, L.red, A);
}

// ---------------------------------------------------------------------------------------------------------------------
// The f16 kernel with EIGHT wavefronts per workgroup (two per SIMD), for at most 4 layer-2 outputs -- same division of
// labour as k_dqn_grad_packed8: wavefronts 0..3 own the s rows, q_local(s) and the backward pass of strip w & 3,
// wavefronts 4..7 the s' rows and the bootstrap value (q_target(s'), and q_local(s') for double DQN), handed over as one
// float per sample; both groups stage weights (group 0 the local fc1 / fc2, group 1 the target's) and split the
// weight-gradient products (group 0: k-column tiles 0, 2, 4, 6; group 1: 1, 3, 5 and dW2^T).  With the matrix work down to
// ~1 k cycles per tile, a 4-wave workgroup is one serial VALU / LDS chain per SIMD (phase stamps at batch 65 536: row
// commit 8 k, the two forwards + heads 8.4 k, TD / dL/dH 2.8 k, products 1.3 k cycles per tile); here the two halves of that
// chain run on the two wavefronts of each SIMD.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void wh_commit_half(_Float16 *dst, floatx4 (&v)[kStageIters], float bias, int t256)
{
#pragma unroll
    for (int it = 0; it < kStageIters; ++it) {
        const int c = it * 256 + t256;
        if (c < kStageChunks) {
            const int row = c / 25, q = c - row * 25;
            *reinterpret_cast<half4 *>(dst + row * kLdH + 4 * q) =
                half4{(_Float16)v[it][0], (_Float16)v[it][1], (_Float16)v[it][2], (_Float16)v[it][3]};
        }
    }
    if (t256 < kHid) {
        _Float16 *row = dst + t256 * kLdH;
        *reinterpret_cast<half4 *>(row + 100) = half4{(_Float16)bias, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f};
#pragma unroll
        for (int k = 0; k < 3; ++k) *reinterpret_cast<half8 *>(row + 104 + 8 * k) = half8{0, 0, 0, 0, 0, 0, 0, 0};
    }
}

struct TileLoads8 {
    floatx4 vX[kXIters];             // group 0: the s rows of this wavefront's strip, group 1: the s' rows
    int p_act, smp;
    float p_rew, p_done, p_valid, p_w;
};

template <int KIND>
__device__ __forceinline__ void tile_issue8(const GradArgs &g, int tile, int grp, int strip, TileLoads8 &T)
{
    const int lane = (int)threadIdx.x & 63, r = lane & 15;
    const int smp = tile * kTile + strip * 16 + r;
    int f, agent;
    if (g.explicit_idx) {
        f = g.explicit_idx[2 * smp];
        agent = g.explicit_idx[2 * smp + 1];
    } else {
        replay_slot_to_frame(g.perm, replay_perm_apply(g.perm, (uint32_t)smp), g.head, g.ring.frames, f, agent);
    }
    int fn = f + 1;
    if (fn >= g.ring.frames) fn = 0;
    const uint32_t row_s = (uint32_t)f * (uint32_t)g.ring.n_agents + (uint32_t)agent;
    const uint32_t row_n = (uint32_t)fn * (uint32_t)g.ring.n_agents + (uint32_t)agent;
    xh_issue<KIND>(T.vX, g.ring.obs, grp == 0 ? row_s : row_n);
    T.p_act = 0; T.p_rew = 0.0f; T.p_done = 0.0f; T.p_valid = 1.0f; T.p_w = 1.0f; T.smp = smp;
    if (grp == 0) {
        uint32_t raw_done, raw_valid;
        load_transition(g.ring, row_s, T.p_act, T.p_rew, raw_done, raw_valid);
        T.p_done = (float)raw_done;
        T.p_valid = (float)raw_valid;
        T.p_w = g.is_w ? g.is_w[smp] : 1.0f;
    }
}

// the partial-gradient row of an 8-wave workgroup (GradAcc8): dW1 | db1 | dW2 | db2 | loss sum | valid count
__device__ __forceinline__ void grad_write_partials8(const GradArgs &g, int stride, float *red, const GradAcc8 &A)
{
    constexpr int NMAX = 4;
    const int tid = (int)threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), grp = wv >> 2, strip = wv & 3;
    const int r = lane & 15, gq = lane >> 4;
    const int n2 = g.n_actions + (g.dueling ? 1 : 0);
    float *out = g.partials + (size_t)blockIdx.x * stride;
    const int oW2 = kHid * kW + kHid, ob2 = oW2 + n2 * kHid;
    const int j = 16 * strip + r;
    if (grp == 0) {
#pragma unroll
        for (int u = 0; u < 3; ++u) *reinterpret_cast<floatx4 *>(out + j * kW + 32 * u + 4 * gq) = A.acc[u];     // tiles 0, 2, 4
        if (gq == 0) *reinterpret_cast<floatx4 *>(out + j * kW + 96) = A.acc[3];                                // tile 6: columns 96..99
        else if (gq == 1) out[kHid * kW + j] = A.acc[3][0];                                                      // column 100 -> db1[j]
    } else {
#pragma unroll
        for (int u = 0; u < 3; ++u) *reinterpret_cast<floatx4 *>(out + j * kW + 16 + 32 * u + 4 * gq) = A.acc[u]; // tiles 1, 3, 5
        if (r < n2) *reinterpret_cast<floatx4 *>(out + oW2 + r * kHid + 16 * strip + 4 * gq) = A.acc[3];          // dW2^T
    }
    if (grp == 0 && lane == 0) {
#pragma unroll
        for (int a = 0; a < NMAX + 2; ++a) red[strip * (kMaxOut + 2) + a] = A.csum[a];
    }
    __syncthreads();
    if (tid < NMAX + 2) {
        const float s = (red[tid] + red[(kMaxOut + 2) + tid]) + (red[2 * (kMaxOut + 2) + tid] + red[3 * (kMaxOut + 2) + tid]);
        if (tid < n2) out[ob2 + tid] = s;
        else if (tid == NMAX) out[g.P] = s;
        else if (tid == NMAX + 1) out[g.P + 1] = s;
    }
}

// Schedule.  Group 1 runs ONE TILE AHEAD of group 0: while group 0 turns tile i's bootstrap values into dL/dH (TD) and
// then commits + forwards the s rows of tile i + 1, group 1 commits the s' rows of tile i + 1 and computes its bootstrap
// values -- so the TD never waits for the (longer, two forwards with double DQN) chain of group 1.  Two barriers per tile
// bracket the weight-gradient products, which both groups share.  Double buffers: the bootstrap values and the
// transposed X tile (tile i + 1's is written while tile i's still feeds the products); HT / dHT have their own space.
template <int KIND>
__global__ void __launch_bounds__(512) k_dqn_grad_h8(Grad2Args ga)
{
    constexpr int NMAX = 4;
    const GradArgs &g = ga.g;
    extern __shared__ __align__(16) float lds[];
    _Float16 *hb = reinterpret_cast<_Float16 *>(lds);
    _Float16 *W1l = hb;                                  // [64][136]
    _Float16 *W1t = W1l + kTile * kLdH;
    _Float16 *Xs = W1t + kTile * kLdH;                   // s rows (group 0), s' rows (group 1)
    _Float16 *Xn = Xs + kTile * kLdH;
    _Float16 *HT = Xn + kTile * kLdH;                    // [64][72] each
    _Float16 *dHT = HT + kTile * kLdT;
    _Float16 *XsT = dHT + kTile * kLdT;                  // 2 x [112][72]
    _Float16 *doutT = XsT + 2 * 112 * kLdT;              // [16][72]
    float *W2l = reinterpret_cast<float *>(doutT + 16 * kLdT);
    float *W2t = W2l + kMaxOut * kHid;
    float *b2l = W2t + kMaxOut * kHid;
    float *b2t = b2l + kMaxOut;
    float *red = b2t + kMaxOut;
    float *qn_lds = red + 4 * (kMaxOut + 2);             // [2][64]
    uint32_t *stage_all = reinterpret_cast<uint32_t *>(qn_lds + 2 * kTile);
    const int tid = (int)threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), grp = wv >> 2, strip = wv & 3;
    const int r = lane & 15, gq = lane >> 4, t256 = tid & 255;
    const int n2 = g.n_actions + (g.dueling ? 1 : 0);
    _Float16 *x_strip = (grp == 0 ? Xs : Xn) + strip * 16 * kLdH;
    uint32_t *stage = stage_all + wv * kStageW;
    // XsT rows 101..111 (both buffers) feed output rows nobody stores; zeroed once so that no NaN bit pattern enters an MFMA
    for (int k = tid; k < 11 * kLdT / 2; k += 512) {
        reinterpret_cast<uint32_t *>(XsT + 101 * kLdT)[k] = 0u;
        reinterpret_cast<uint32_t *>(XsT + (112 + 101) * kLdT)[k] = 0u;
    }
    GradAcc8 A;
#pragma unroll
    for (int u = 0; u < 4; ++u) A.acc[u] = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int a = 0; a < 6; ++a) A.csum[a] = 0.0f;
    L_STAMP(0);
    const int step = (int)gridDim.x;
    const int n_my = ((int)ga.n_tiles - (int)blockIdx.x + step - 1) / step;      // tiles blockIdx.x + i * step, i < n_my
    auto tile_of = [&](int i) { return (int)blockIdx.x + i * step; };
    TileLoads8 T;
    {   // ---- weights: group 0 stages q_local's, group 1 q_target's; the first tile's rows are in flight meanwhile
        const float *net = grp == 0 ? g.local : g.target;
        floatx4 vW[kStageIters];
        w_issue_half(vW, net, t256);
        const NetDev nv = net_view(net, n2);
        const float pb1 = nv.b1[t256 < kHid ? t256 : kHid - 1];
        float pw[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) pw[k] = nv.W2[t256 + 256 * k < n2 * kHid ? t256 + 256 * k : 0];
        const float pb2 = nv.b2[t256 < n2 ? t256 : 0];
        tile_issue8<KIND>(g, tile_of(0), grp, strip, T);
        wh_commit_half(grp == 0 ? W1l : W1t, vW, pb1, t256);
        float *W2 = grp == 0 ? W2l : W2t, *b2 = grp == 0 ? b2l : b2t;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (t256 + 256 * k < n2 * kHid) W2[t256 + 256 * k] = pw[k];
        if (t256 < n2) b2[t256] = pb2;
        if (grp == 1)
            for (int k = t256; k < 16 * kLdT / 2; k += 256) reinterpret_cast<uint32_t *>(doutT)[k] = 0u;     // rows >= n2 stay zero
    }
    __syncthreads();
    L_STAMP(1);
    floatx4 hl[4];                                        // group 0: pre-activations / Q of the tile whose TD comes next
    float ql[NMAX];
    int p_act = 0, smp = 0;
    float p_rew = 0.0f, p_done = 0.0f, p_valid = 1.0f, p_w = 1.0f;
    // (the fc2 fragments are re-read from LDS where they are used: two of them live across the whole loop spill)
    // commit this wavefront's rows of tile i, request tile i + 1's, and (group 0) Q_local(s) / (group 1) the bootstrap values
    auto front = [&](int i) {
        if (grp == 0) { p_act = T.p_act; p_rew = T.p_rew; p_done = T.p_done; p_valid = T.p_valid; p_w = T.p_w; smp = T.smp; }
        xh_commit<KIND>(x_strip, T.vX, stage, grp == 0 ? XsT + (i & 1) * 112 * kLdT + strip * 16 : nullptr);
        wave_lds_sync();
        if (i + 1 < n_my) tile_issue8<KIND>(g, tile_of(i + 1), grp, strip, T);
        if (grp == 0) {
            fwd_strip_h(W1l, x_strip, hl);
            W2Frag<NMAX> Fl;
            w2_load<NMAX, 3>(Fl, W2l, b2l, n2);
            q_strip<NMAX, 3>(hl, Fl, n2, g.n_actions, g.dueling, ql);
        } else {
            int best = 0;
            floatx4 ht[4];
            if (g.kind == 1) {                // double DQN: a* = argmax_a Q_local(s', a)   (DDQN_Trainer.py:94)
                fwd_strip_h(W1l, x_strip, ht);
                W2Frag<NMAX> Fl;
                w2_load<NMAX, 3>(Fl, W2l, b2l, n2);
                float qn_l[NMAX];
                q_strip<NMAX, 3>(ht, Fl, n2, g.n_actions, g.dueling, qn_l);
                float bq = qn_l[0];
#pragma unroll
                for (int a = 1; a < NMAX; ++a)
                    if (a < g.n_actions && qn_l[a] > bq) { bq = qn_l[a]; best = a; }
            }
            fwd_strip_h(W1t, x_strip, ht);
            W2Frag<NMAX> Ft;
            w2_load<NMAX, 3>(Ft, W2t, b2t, n2);
            float qt[NMAX];
            q_strip<NMAX, 3>(ht, Ft, n2, g.n_actions, g.dueling, qt);
            if (gq == 0) qn_lds[(i & 1) * kTile + strip * 16 + r] = pick_qn<NMAX>(g, qt, best);
            wave_lds_sync();                  // (the strip is rewritten by this wavefront's next commit)
        }
    };
    front(0);
    __syncthreads();                          // tile 0's bootstrap values handed over
    L_STAMP(2);
    for (int i = 0; i < n_my; ++i) {
        if (grp == 0) {
            GradAcc<NMAX> Tc;                 // td_backward's accumulator interface: only csum is used here
#pragma unroll
            for (int a = 0; a < NMAX + 2; ++a) Tc.csum[a] = A.csum[a];
            const int s = strip * 16 + r;
            W2Frag<NMAX> Fl;
            w2_load<NMAX, 3>(Fl, W2l, b2l, n2);
            td_backward<NMAX, true>(g, W2l, hl, Fl, ql, qn_lds[(i & 1) * kTile + s], p_act, p_rew, p_done, p_valid, p_w, smp, Tc,
                                    reinterpret_cast<float *>(HT + s), reinterpret_cast<float *>(dHT + s),
                                    reinterpret_cast<float *>(doutT + s));
#pragma unroll
            for (int a = 0; a < NMAX + 2; ++a) A.csum[a] = Tc.csum[a];
        }
        if (i + 1 < n_my) front(i + 1);       // both groups, one tile ahead of the products below
        __syncthreads();                      // HT / dHT / doutT of tile i complete (and tile i + 1's bootstrap values)
        if (i == n_my - 1) L_STAMP(3);
        // ---- weight gradients of tile i, K = 64 samples = 2 MFMA steps of 32 (hidden units 16 strip + r):
        //   group 0: dW1^T k-column tiles 0, 2, 4, 6; group 1: tiles 1, 3, 5 and dW2^T[j][a] += sum_s HT[j][s] doutT[a][s]
        {
            const _Float16 *xa = XsT + (i & 1) * 112 * kLdT + r * kLdT + 8 * gq;
            const _Float16 *db = dHT + (16 * strip + r) * kLdT + 8 * gq;
            const _Float16 *ha = HT + (16 * strip + r) * kLdT + 8 * gq;
            const _Float16 *ob = doutT + r * kLdT + 8 * gq;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const half8 bdh = *reinterpret_cast<const half8 *>(db + 32 * kk);
#pragma unroll
                for (int u = 0; u < 3; ++u)
                    A.acc[u] = mfma16h(*reinterpret_cast<const half8 *>(xa + 16 * (2 * u + grp) * kLdT + 32 * kk), bdh, A.acc[u]);
                if (grp == 0) A.acc[3] = mfma16h(*reinterpret_cast<const half8 *>(xa + 16 * 6 * kLdT + 32 * kk), bdh, A.acc[3]);
                else A.acc[3] = mfma16h(*reinterpret_cast<const half8 *>(ha + 32 * kk), *reinterpret_cast<const half8 *>(ob + 32 * kk), A.acc[3]);
            }
        }
        if (i + 1 < n_my) __syncthreads();    // the next TD overwrites HT / dHT / doutT
    }
    L_STAMP(5);
    grad_write_partials8(g, ga.stride, red, A);
}

constexpr size_t kGradH8Lds = (size_t)(4 * kTile * kLdH + 2 * kTile * kLdT + 2 * 112 * kLdT + 16 * kLdT) * 2 +
                              (size_t)(2 * kMaxOut * kHid + 2 * kMaxOut + 4 * (kMaxOut + 2) + 2 * kTile + 8 * kStageW) * 4;

constexpr size_t kGradHLds = (size_t)(3 * kTile * kLdH + 2 * kTile * kLdT + 112 * kLdT + 16 * kLdT) * 2 +
                             (size_t)(2 * kMaxOut * kHid + 2 * kMaxOut + 4 * (kMaxOut + 2) + 4 * kStageW) * 4;

template <typename ObsT, int NMAX>
__global__ void __launch_bounds__(256) k_dqn_grad(Grad2Args ga)
{
    const GradArgs &g = ga.g;
    extern __shared__ __align__(16) float lds[];
    GradLds L;
    L.W1l = lds;                            // [64][108] local fc1 (+ b1 in column 100)
    L.W1t = L.W1l + kTileF;                 // target fc1
    L.Xs = L.W1t + kTileF;                  // states        (strip w = rows 16 w ..)
    L.Xn = L.Xs + kTileF;                   // next states; after the forward passes strip w's first 16 x 68 floats hold H of its samples
    L.dHs = L.Xn + kTileF;                  // [64][68] dL/dH
    L.douts = L.dHs + kTile * kLh;          // [64][16] dL/d(layer-2 output)
    L.W2l = L.douts + kTile * kMaxOut;      // [16][64]
    L.W2t = L.W2l + kMaxOut * kHid;
    L.b2l = L.W2t + kMaxOut * kHid;         // [16]
    L.b2t = L.b2l + kMaxOut;
    L.red = L.b2t + kMaxOut;                // [4][18] per-wave column sums
    const int tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6, r = lane & 15, gq = lane >> 4;
    const int n2 = g.n_actions + (g.dueling ? 1 : 0);
    GradAcc<NMAX> A;
    A.acc2 = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int u = 0; u < 7; ++u) A.acc1[u] = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int a = 0; a < NMAX + 2; ++a) A.csum[a] = 0.0f;

    L_STAMP(0);
    const int step = (int)gridDim.x;
    int tile = (int)blockIdx.x;                       // the grid never exceeds the number of tiles
    grad_tile<ObsT, NMAX, true>(g, L, tile, tile + step < ga.n_tiles, A);
    for (tile += step; tile < ga.n_tiles; tile += step)
        grad_tile<ObsT, NMAX, false>(g, L, tile, tile + step < ga.n_tiles, A);
    L_STAMP(5);
    grad_write_partials<NMAX>(g, ga.stride, L.red, A);
}

constexpr int kMaxGradGrid = 256;           // one workgroup per CU (140 KB of LDS each)
constexpr size_t kGrad2Lds = (size_t)(4 * kTileF + kTile * kLh + kTile * kMaxOut + 2 * kMaxOut * kHid + 2 * kMaxOut +
                                      4 * (kMaxOut + 2)) * 4;

// Column sums of the partial rows for the 32 parameters of this workgroup: 8 float4 column groups x 32 row groups, every
// thread's loads (<= 8 rows of 16 bytes per 256 rows) independent and in flight together -- one memory round trip (a
// dependent walk over the rows was 8 round trips: 6 us for 256 rows).  Returns the sum of column blockIdx.x * 32 + tid
// to threads tid < 32 (others: 0).  `c_out`: this thread's share of the valid-count column.
__device__ __forceinline__ float reduce_columns(const float *__restrict__ partials, int nblk, int P, int stride, float &c_out)
{
    __shared__ float red[32][33];
    __shared__ float red2[8][33];
    const int tid = (int)threadIdx.x;
    const int cg = tid & 7, rg = tid >> 3;                       // column group (4 columns), row group
    const int p4 = (int)blockIdx.x * 32 + cg * 4;                // first of this thread's four columns (stride % 4 == 0)
    const bool col_ok = p4 < stride;
    floatx4 acc = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
    float c = 0.0f;
    for (int b0 = 0; b0 < nblk; b0 += 256) {
        floatx4 t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int b = b0 + rg + 32 * k;
            const int bb = b < nblk ? b : nblk - 1;
            t[k] = *reinterpret_cast<const floatx4 *>(partials + (size_t)bb * stride + (col_ok ? p4 : 0));
        }
        const int bc = b0 + tid;
        const float cv = partials[(size_t)(bc < nblk ? bc : nblk - 1) * stride + P + 1];
        c += bc < nblk ? cv : 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const bool ok = b0 + rg + 32 * k < nblk;
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += ok ? t[k][e] : 0.0f;
        }
    }
    c_out = c;
#pragma unroll
    for (int e = 0; e < 4; ++e) red[rg][cg * 4 + e] = acc[e];
    __syncthreads();
    {
        const int px = tid & 31, part = tid >> 5;                // 32 row groups -> 8 -> 1
        red2[part][px] = (red[4 * part][px] + red[4 * part + 1][px]) + (red[4 * part + 2][px] + red[4 * part + 3][px]);
    }
    __syncthreads();
    float t = 0.0f;
    if (tid < 32) {
#pragma unroll
        for (int k = 0; k < 8; ++k) t += red2[k][tid];
    }
    return t;
}

// raw[p] = sum_b partial[b][p] for p in [0, P+2): gradient sums, loss sum, valid count (the multi-GPU all-reduce payload).
__global__ void __launch_bounds__(256) k_dqn_reduce(const float *__restrict__ partials, int nblk, int P, int stride,
                                                    float *__restrict__ raw)
{
    float c;
    const float t = reduce_columns(partials, nblk, P, stride, c);
    const int p = (int)blockIdx.x * 32 + (int)threadIdx.x;
    if (threadIdx.x < 32 && p < P + 2) raw[p] = t;
}

// torch.optim.Adam (amsgrad off, weight_decay 0) on grad = raw / max(valid count, 1), + optional hard target copy
// (DQN_Trainer.py:121-130,138-141).  raw[P] = loss sum, raw[P+1] = valid count.
__global__ void k_dqn_adam(float *__restrict__ local, float *__restrict__ target, float *__restrict__ m,
                           float *__restrict__ v, const float *__restrict__ raw, int P, float lr, float beta1,
                           float beta2, float eps, float bc1, float bc2_sqrt, int hard_update, float *__restrict__ loss,
                           float *__restrict__ img)
{
    const int p = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    const float cnt = raw[P + 1];
    const float inv = 1.0f / (cnt > 1.0f ? cnt : 1.0f);
    if (p == 0 && loss) *loss = raw[P] * inv;
    if (p >= P) return;
    const float gp = raw[p] * inv;
    const float mp = m[p] + (gp - m[p]) * (1.0f - beta1);          // exp_avg.lerp_(grad, 1 - beta1)
    const float vp = v[p] * beta2 + (1.0f - beta2) * gp * gp;      // exp_avg_sq.mul_(beta2).addcmul_(g, g, 1 - beta2)
    m[p] = mp;
    v[p] = vp;
    const float denom = sqrtf(vp) / bc2_sqrt + eps;
    const float np = local[p] - (lr / bc1) * (mp / denom);
    local[p] = np;
    if (hard_update) target[p] = np;
    if (img) {
        img_store_param(img, p, np);
        if (hard_update) img_store_param(img + kSplitF, p, np);
    }
}

// Single-GPU fast path: k_dqn_reduce + k_dqn_adam in one launch (each workgroup owns 32 parameters end to end; the
// valid count is re-derived per workgroup from the count cells, one load per thread).
__global__ void __launch_bounds__(256) k_dqn_reduce_adam(const float *__restrict__ partials, int nblk, int P, int stride,
                                                         float *__restrict__ local, float *__restrict__ target,
                                                         float *__restrict__ m, float *__restrict__ v, float lr,
                                                         float beta1, float beta2, float eps, float bc1, float bc2_sqrt,
                                                         int hard_update, float *__restrict__ loss, float *__restrict__ raw,
                                                         const uint32_t *__restrict__ go_word, uint32_t go_value, float *__restrict__ img)
{
    // gated update (uavenv_dqn_reduce_adam_gated): the step kernel of this pass stamps go_word with go_value when it moved at
    // least one agent; a pass in which every agent had already finished leaves the learner exactly as it is (the reference's loop
    // has left run_eposide by then, Envs/PathPlan_City.py:456-459)
    UAV_HOT_PRIO();
    if (go_word && __hip_atomic_load(go_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != go_value) return;
    __shared__ float cnt_part[4];
    const int tid = (int)threadIdx.x;
    const int p = (int)blockIdx.x * 32 + tid;
    // this parameter's moments and value are requested FIRST: their round trip runs under the partial rows' (behind the
    // reduction's barriers they were a second, dependent round trip of the launch)
    float m0 = 0.0f, v0 = 0.0f, w0 = 0.0f;
    if (tid < 32 && p < P) { m0 = m[p]; v0 = v[p]; w0 = local[p]; }
    float c;
    const float t = reduce_columns(partials, nblk, P, stride, c);
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
    if ((tid & 63) == 0) cnt_part[tid >> 6] = c;
    __syncthreads();
    const float cnt = (cnt_part[0] + cnt_part[1]) + (cnt_part[2] + cnt_part[3]);
    const float inv = 1.0f / (cnt > 1.0f ? cnt : 1.0f);
    if (tid < 32 && p < P + 2) {
        if (raw) raw[p] = t;
        if (p < P) {
            const float gp = t * inv;
            const float mp = m0 + (gp - m0) * (1.0f - beta1);
            const float vp = v0 * beta2 + (1.0f - beta2) * gp * gp;
            m[p] = mp;
            v[p] = vp;
            const float np = w0 - (lr / bc1) * (mp / (sqrtf(vp) / bc2_sqrt + eps));
            local[p] = np;
            if (hard_update) target[p] = np;
            if (img) {                        // the C loop's layer-1 image follows the parameters (qnet_device.hpp: img_store_param)
                img_store_param(img, p, np);
                if (hard_update) img_store_param(img + kSplitF, p, np);
            }
        } else if (p == P && loss) {
            *loss = t * inv;
        }
    }
}

struct ActArgs {
    const void *obs;       // [n][100]
    int n, n_actions, dueling;
    const float *local;
    float eps;
    uint64_t seed, counter;
    int32_t *index_out;
    float *steer_out;
    float *q_out;          // nullable [n][A]
    const float *img;      // nullable: q_local's layer 1 in the split form (packed rows, f32 MFMA)
};

// Q(s) + epsilon-greedy for a tile of 64 envs (Trainer/DuelingDQN_Trainer.py:86-97), same wave-strip forward as
// k_dqn_grad: wavefront w owns rows 16 w .. 16 w + 15 of the tile (consecutive observation rows: one 6.4 KB run).
template <typename ObsT, int NMAX>
__global__ void __launch_bounds__(256) k_dqn_act(ActArgs g)
{
    extern __shared__ __align__(16) float lds[];
    float *W1 = lds;                        // [64][108] fc1 (+ b1 in column 100)
    float *Xs = W1 + kTileF;                // [64][108]
    float *W2 = Xs + kTileF;                // [16][64]
    float *b2 = W2 + kMaxOut * kHid;        // [16]
    const int tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6, r = lane & 15;
    const int n2 = g.n_actions + (g.dueling ? 1 : 0);
    const NetDev nl = net_view(g.local, n2);
    const int first = (int)blockIdx.x * kTile + wv * 16;              // first row of this wavefront's strip
    float *strip = Xs + wv * kStripF;
    // one memory round trip: observation strip, fc1, the small vectors
    floatx4 vX[kXIters], vW[kStageIters];
    {
        const ObsT *obs = reinterpret_cast<const ObsT *>(g.obs);
#pragma unroll
        for (int it = 0; it < kXIters; ++it) {
            int c = it * 64 + lane;
            c = c < 400 ? c : 399;
            const int row = c / 25, q = c - row * 25;
            int i = first + row;
            i = i < g.n ? i : g.n - 1;                                 // ragged last tile: clamp (results discarded)
            vX[it] = x_load<ObsT>(obs, (uint32_t)i, q);
        }
    }
    w_issue(vW, g.local);
    // the epsilon-greedy draw of this lane's env: a serial chain, computed under the loads' round trip
    const int i = first + r;
    const uint4 rn = philox4x32_10(make_uint4((uint32_t)i, (uint32_t)g.counter, (uint32_t)(g.counter >> 32), 0xac7u),
                                   make_uint2((uint32_t)g.seed, (uint32_t)(g.seed >> 32)));
    const float pb1 = nl.b1[tid < kHid ? tid : kHid - 1];
    float pw[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) pw[k] = nl.W2[tid + 256 * k < n2 * kHid ? tid + 256 * k : 0];
    const float pb2 = nl.b2[tid < n2 ? tid : 0];
    x_commit<ObsT>(strip, vX);
    w_commit(W1, vW, pb1);
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (tid + 256 * k < n2 * kHid) W2[tid + 256 * k] = pw[k];
    if (tid < n2) b2[tid] = pb2;
    __syncthreads();
    floatx4 h[4];
    fwd_strip(W1, strip, h);
    float q[NMAX];
    {
        W2Frag<NMAX> F;
        w2_load<NMAX>(F, W2, b2, n2);
        q_strip<NMAX>(h, F, n2, g.n_actions, g.dueling, q);
    }
    if (lane < 16 && i < g.n) {
        if (g.q_out) {
#pragma unroll
            for (int a = 0; a < NMAX; ++a)
                if (a < g.n_actions) g.q_out[(size_t)i * g.n_actions + a] = q[a];
        }
        const float sample = (float)(rn.x >> 8) * (1.0f / 16777216.0f);
        int a;
        if (sample > g.eps) {
            a = 0;
            float bq = q[0];
#pragma unroll
            for (int k = 1; k < NMAX; ++k)
                if (k < g.n_actions && q[k] > bq) { bq = q[k]; a = k; }
        } else {
            a = (int)(((uint64_t)rn.y * (uint64_t)g.n_actions) >> 32);
        }
        if (g.index_out) g.index_out[i] = a;
        if (g.steer_out) g.steer_out[i] = (float)(-1.0 + 2.0 * (double)a / (double)(g.n_actions - 1));
    }
}

// Packed observations: no observation tile -- every lane loads the packed row of its env and generates the MFMA operands
// from it (fwd_strip_split, qnet_device.hpp "Layer 1 at f32 accuracy on the f16 matrix pipe"); LDS holds the weights only.
template <int NMAX>
__global__ void __launch_bounds__(256) k_dqn_act_packed(ActArgs g)
{
    extern __shared__ __align__(16) float lds[];
    const W1Split W1 = w1split_at(lds);     // fc1 + b1 in the split form (kTileF floats)
    float *W2 = lds + kSplitF;              // [16][64]
    float *b2 = W2 + kMaxOut * kHid;        // [16]
    const int tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6, r = lane & 15;
    const int n2 = g.n_actions + (g.dueling ? 1 : 0);
    const NetDev nl = net_view(g.local, n2);
    const int i = (int)blockIdx.x * kTile + wv * 16 + r;
    floatx4 vW[kStageIters];
    SplitScRegs vS;
    if (g.img) img_issue(vW, g.img); else { w_issue(vW, g.local); w_issue_sc(vS, g.local, nl.b1); }
    float pw[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) pw[k] = nl.W2[tid + 256 * k < n2 * kHid ? tid + 256 * k : 0];
    const float pb2 = nl.b2[tid < n2 ? tid : 0];
    PRow R;
    prow_load(R, reinterpret_cast<const uint32_t *>(g.obs) + (size_t)(i < g.n ? i : g.n - 1) * kPackedDwords);
    // the epsilon-greedy draw of this lane's env: a serial chain, computed under the loads' round trip
    const uint4 rn = philox4x32_10(make_uint4((uint32_t)i, (uint32_t)g.counter, (uint32_t)(g.counter >> 32), 0xac7u),
                                   make_uint2((uint32_t)g.seed, (uint32_t)(g.seed >> 32)));
    if (g.img) img_commit(lds, vW); else w_commit_split(W1, vW, vS);
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (tid + 256 * k < n2 * kHid) W2[tid + 256 * k] = pw[k];
    if (tid < n2) b2[tid] = pb2;
    __syncthreads();
    floatx4 h[4];
    fwd_strip_split<false>(W1, R, h);
    float q[NMAX];
    {
        W2Frag<NMAX> F;
        w2_load<NMAX>(F, W2, b2, n2);
        q_strip<NMAX>(h, F, n2, g.n_actions, g.dueling, q);
    }
    if (lane < 16 && i < g.n) {
        if (g.q_out) {
#pragma unroll
            for (int a = 0; a < NMAX; ++a)
                if (a < g.n_actions) g.q_out[(size_t)i * g.n_actions + a] = q[a];
        }
        const float sample = (float)(rn.x >> 8) * (1.0f / 16777216.0f);
        int a;
        if (sample > g.eps) {
            a = 0;
            float bq = q[0];
#pragma unroll
            for (int k = 1; k < NMAX; ++k)
                if (k < g.n_actions && q[k] > bq) { bq = q[k]; a = k; }
        } else {
            a = (int)(((uint64_t)rn.y * (uint64_t)g.n_actions) >> 32);
        }
        if (g.index_out) g.index_out[i] = a;
        if (g.steer_out) g.steer_out[i] = (float)(-1.0 + 2.0 * (double)a / (double)(g.n_actions - 1));
    }
}

// f16 MFMA forward (UavDqnNet.mfma_dtype = 1) on f16 or packed observations.  Workgroups are persistent over tiles (the
// grid is capped at 512): fc1 is staged and converted once per workgroup, not once per 64 agents, and the next tile's rows
// are requested before the current tile is computed.
template <int KIND>
__device__ __forceinline__ void act_rows_issue(const ActArgs &g, int first, floatx4 (&vX)[kXIters])
{
    constexpr int per_row = KIND == OBS_KIND_PACKED ? 5 : 25, iters = KIND == OBS_KIND_PACKED ? 2 : kXIters;
    const int lane = (int)threadIdx.x & 63;
#pragma unroll
    for (int it = 0; it < iters; ++it) {
        int c = it * 64 + lane;
        c = c < 16 * per_row ? c : 16 * per_row - 1;
        const int row = c / per_row, q = c - row * per_row;
        int i = first + row;
        i = i < g.n ? i : g.n - 1;
        if (KIND == OBS_KIND_PACKED) {
            vX[it] = *reinterpret_cast<const floatx4 *>(reinterpret_cast<const uint32_t *>(g.obs) + (size_t)i * kPackedDwords + 4 * q);
        } else {
            const uint2 raw = *reinterpret_cast<const uint2 *>(reinterpret_cast<const __half *>(g.obs) + (size_t)i * kW + 4 * q);
            vX[it] = floatx4{__uint_as_float(raw.x), __uint_as_float(raw.y), 0.0f, 0.0f};
        }
    }
}

template <int KIND, int NMAX>
__global__ void __launch_bounds__(256) k_dqn_act_h(ActArgs g)
{
    extern __shared__ __align__(16) float lds[];
    _Float16 *W1 = reinterpret_cast<_Float16 *>(lds);       // [64][136]
    _Float16 *Xs = W1 + kTile * kLdH;                         // [64][136]
    float *W2 = reinterpret_cast<float *>(Xs + kTile * kLdH); // [16][64]
    float *b2 = W2 + kMaxOut * kHid;                          // [16]
    uint32_t *stage = reinterpret_cast<uint32_t *>(b2 + kMaxOut);
    const int tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6, r = lane & 15;
    const int n2 = g.n_actions + (g.dueling ? 1 : 0);
    const NetDev nl = net_view(g.local, n2);
    const int n_tiles = (g.n + kTile - 1) / kTile;
    _Float16 *strip = Xs + wv * 16 * kLdH;
    floatx4 vX[kXIters];
    int tile = (int)blockIdx.x;
    act_rows_issue<KIND>(g, tile * kTile + wv * 16, vX);
    {
        floatx4 vW[kStageIters];
        w_issue(vW, g.local);
        const float pb1 = nl.b1[tid < kHid ? tid : kHid - 1];
        float pw[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) pw[k] = nl.W2[tid + 256 * k < n2 * kHid ? tid + 256 * k : 0];
        const float pb2 = nl.b2[tid < n2 ? tid : 0];
        wh_commit(W1, vW, pb1);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (tid + 256 * k < n2 * kHid) W2[tid + 256 * k] = pw[k];
        if (tid < n2) b2[tid] = pb2;
    }
    __syncthreads();
    W2Frag<NMAX> F;
    w2_load<NMAX, 3>(F, W2, b2, n2);
    for (; tile < n_tiles; tile += (int)gridDim.x) {
        const int i = tile * kTile + wv * 16 + r;
        xh_commit<KIND>(strip, vX, stage + wv * kStageW, nullptr);
        wave_lds_sync();
        if (tile + (int)gridDim.x < n_tiles) act_rows_issue<KIND>(g, (tile + (int)gridDim.x) * kTile + wv * 16, vX);
        const uint4 rn = philox4x32_10(make_uint4((uint32_t)i, (uint32_t)g.counter, (uint32_t)(g.counter >> 32), 0xac7u),
                                       make_uint2((uint32_t)g.seed, (uint32_t)(g.seed >> 32)));
        floatx4 h[4];
        fwd_strip_h(W1, strip, h);
        float q[NMAX];
        q_strip<NMAX, 3>(h, F, n2, g.n_actions, g.dueling, q);
        if (lane < 16 && i < g.n) {
            if (g.q_out) {
#pragma unroll
                for (int a = 0; a < NMAX; ++a)
                    if (a < g.n_actions) g.q_out[(size_t)i * g.n_actions + a] = q[a];
            }
            const float sample = (float)(rn.x >> 8) * (1.0f / 16777216.0f);
            int a;
            if (sample > g.eps) {
                a = 0;
                float bq = q[0];
#pragma unroll
                for (int k = 1; k < NMAX; ++k)
                    if (k < g.n_actions && q[k] > bq) { bq = q[k]; a = k; }
            } else {
                a = (int)(((uint64_t)rn.y * (uint64_t)g.n_actions) >> 32);
            }
            if (g.index_out) g.index_out[i] = a;
            if (g.steer_out) g.steer_out[i] = (float)(-1.0 + 2.0 * (double)a / (double)(g.n_actions - 1));
        }
        wave_lds_sync();                      // the strip is rewritten by the next tile's commit
    }
}

constexpr size_t kActHLds = (size_t)(2 * kTile * kLdH) * 2 + (size_t)(kMaxOut * kHid + kMaxOut + 4 * kStageW) * 4;
constexpr size_t kAct2Lds = (size_t)(2 * kTileF + kMaxOut * kHid + kMaxOut) * 4;
constexpr size_t kActPLds = (size_t)(kTileF + kMaxOut * kHid + kMaxOut) * 4;

unsigned long long *g_learner_dbg = nullptr;

bool net_ok(const UavDqnNet *n)
{
    return n && n->local && n->w == kW && n->hid == kHid && n->n_actions >= 2 &&
           n->n_actions + (n->dueling ? 1 : 0) + 2 <= kMaxOut;     // + 2 spare dout columns (loss sum, valid count)
}

template <int KIND>
static int launch_grad_h(const Grad2Args &ga, int grid, hipStream_t s)
{
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_dqn_grad_h<KIND, 4>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGradHLds) != hipSuccess)
            return UAVENV_EHIP;
        attr = true;
    }
    hipLaunchKernelGGL((k_dqn_grad_h<KIND, 4>), dim3(grid), dim3(256), kGradHLds, s, ga);
    return UAVENV_OK;
}

template <int KIND>
static int launch_grad_h8(const Grad2Args &ga, int grid, hipStream_t s)
{
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_dqn_grad_h8<KIND>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGradH8Lds) != hipSuccess)
            return UAVENV_EHIP;
        attr = true;
    }
    hipLaunchKernelGGL((k_dqn_grad_h8<KIND>), dim3(grid), dim3(512), kGradH8Lds, s, ga);
    return UAVENV_OK;
}

template <int KIND>
static int launch_act_h(const ActArgs &g, int grid, hipStream_t s)
{
    hipLaunchKernelGGL((k_dqn_act_h<KIND, 4>), dim3(grid < 512 ? grid : 512), dim3(256), kActHLds, s, g);   // 44 KB: no attribute needed
    return UAVENV_OK;
}

template <int NMAX>
static int launch_act_packed(const ActArgs &g, int grid, hipStream_t s)
{
    hipLaunchKernelGGL((k_dqn_act_packed<NMAX>), dim3(grid), dim3(256), kActPLds, s, g);      // 32 KB: no attribute needed
    return UAVENV_OK;
}

template <typename ObsT, int NMAX>
static int launch_act(const ActArgs &g, int grid, hipStream_t s)
{
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_dqn_act<ObsT, NMAX>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)kAct2Lds) != hipSuccess)
            return UAVENV_EHIP;
        attr = true;
    }
    hipLaunchKernelGGL((k_dqn_act<ObsT, NMAX>), dim3(grid), dim3(256), kAct2Lds, s, g);
    return UAVENV_OK;
}

template <int NMAX>
static int launch_grad_packed(const Grad2Args &ga, int grid, hipStream_t s)
{
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_dqn_grad_packed<NMAX>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGradPLds) != hipSuccess)
            return UAVENV_EHIP;
        attr = true;
    }
    hipLaunchKernelGGL((k_dqn_grad_packed<NMAX>), dim3(grid), dim3(256), kGradPLds, s, ga);
    return UAVENV_OK;
}

template <typename ObsT, int NMAX>
static int launch_grad(const Grad2Args &ga, int grid, hipStream_t s)
{
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_dqn_grad<ObsT, NMAX>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGrad2Lds) != hipSuccess)
            return UAVENV_EHIP;
        attr = true;
    }
    hipLaunchKernelGGL((k_dqn_grad<ObsT, NMAX>), dim3(grid), dim3(256), kGrad2Lds, s, ga);
    return UAVENV_OK;
}


}  // namespace

extern "C" {

int uavenv_dqn_set_debug_buffer(unsigned long long *dev_buf)
{
    g_learner_dbg = dev_buf;
    return UAVENV_OK;
}

int uavenv_dqn_num_params(const UavDqnNet *net)
{
    if (!net) return UAVENV_EINVAL;
    const int n2 = net->n_actions + (net->dueling ? 1 : 0);
    return net->hid * net->w + net->hid + n2 * net->hid + n2;
}

int uavenv_dqn_partial_stride(const UavDqnNet *net)
{
    if (!net_ok(net)) return UAVENV_EINVAL;
    // rows start on a 128-byte line: the reduction reads 128-byte column runs of every row, and a run that straddles two lines
    // is fetched by two workgroups, i.e. usually by two XCDs' L2s (round 4: 12.4 MB fetched per launch for 6.8 MB of rows)
    return (uavenv_dqn_num_params(net) + 2 + 31) & ~31;
}

int uavenv_dqn_partial_rows(int32_t batch)
{
    if (batch <= 0 || batch % kTile != 0) return UAVENV_EINVAL;
    const int tiles = batch / kTile;
    return tiles < kMaxGradGrid ? tiles : kMaxGradGrid;       // more tiles than that: persistent workgroups loop over them
}

int uavenv_dqn_grad(const UavReplayRing *ring, int32_t head, int32_t filled, int32_t batch, uint64_t seed,
                    uint64_t counter, const int32_t *explicit_idx, const UavDqnNet *net, int32_t kind, float gamma,
                    int32_t huber, float *partials, void *stream)
{
    return uavenv_dqn_grad_w(ring, head, filled, batch, seed, counter, explicit_idx, net, kind, gamma, huber, nullptr, nullptr,
                             partials, stream);
}

int uavenv_dqn_grad_w(const UavReplayRing *ring, int32_t head, int32_t filled, int32_t batch, uint64_t seed,
                      uint64_t counter, const int32_t *explicit_idx, const UavDqnNet *net, int32_t kind, float gamma,
                      int32_t huber, const float *is_weights, float *abs_td_out, float *partials, void *stream)
{
    return uavenv_dqn_grad_img(ring, head, filled, batch, seed, counter, explicit_idx, net, kind, gamma, huber, is_weights, abs_td_out,
                               partials, nullptr, stream);
}

// (internal, csrc/dqn_internal.hpp) image_dev: uavenv_dqn_split_image's output for THIS net as it is now -- packed rings with the
// f32 MFMA net stage layer 1 of both nets from it; ignored (may be null) everywhere else
int uavenv_dqn_grad_img(const UavReplayRing *ring, int32_t head, int32_t filled, int32_t batch, uint64_t seed,
                        uint64_t counter, const int32_t *explicit_idx, const UavDqnNet *net, int32_t kind, float gamma,
                        int32_t huber, const float *is_weights, float *abs_td_out, float *partials, const float *image_dev, void *stream)
{
    if (!ring || !ring->obs || !ring->action || !ring->reward || !ring->done || !partials || !net_ok(net) || !net->target)
        return UAVENV_EINVAL;
    if (batch <= 0 || batch % kTile != 0 || ring->frames < 2 || head < 0 || head >= ring->frames) return UAVENV_EINVAL;
    if (!explicit_idx && (filled <= 0 || filled > ring->frames - 1)) return UAVENV_EINVAL;
    if ((uint64_t)ring->frames * (uint64_t)ring->n_agents >= (1ull << 32)) return UAVENV_EINVAL;
    if (!ring->action_is_index) return UAVENV_EINVAL;
    if ((((uintptr_t)partials | (uintptr_t)net->local | (uintptr_t)net->target | (uintptr_t)ring->meta) & 15u) != 0) return UAVENV_EINVAL;
    Grad2Args ga;
    GradArgs &g = ga.g;
    g.ring = *ring;
    g.head = head; g.filled = filled; g.batch = batch;
    g.seed = seed; g.counter = counter;
    g.explicit_idx = explicit_idx;
    g.perm = replay_perm(seed, counter, explicit_idx ? 1u : (uint32_t)filled * (uint32_t)ring->n_agents, (uint32_t)ring->n_agents);
    g.local = net->local; g.target = net->target;
    g.n_actions = net->n_actions; g.dueling = net->dueling; g.kind = kind;
    g.gamma = gamma; g.huber = huber;
    g.partials = partials;
    g.P = uavenv_dqn_num_params(net);
    g.dbg = g_learner_dbg;
    g.is_w = is_weights;
    g.abs_td = abs_td_out;
    g.img = (image_dev && (((uintptr_t)image_dev) & 15u) == 0) ? image_dev : nullptr;
    ga.n_tiles = batch / kTile;
    ga.stride = uavenv_dqn_partial_stride(net);
    const int grid = uavenv_dqn_partial_rows(batch);
    hipStream_t s = (hipStream_t)stream;
    const bool small = net->n_actions + (net->dueling ? 1 : 0) <= 4;
    int rc;
    if (net->mfma_dtype == UAVENV_MFMA_F16) {         // f16 operands, f32 accumulate: f16 / packed rings, <= 4 layer-2 outputs
        if (!small || ring->obs_dtype == UAVENV_OBS_F32) return UAVENV_EINVAL;
        static const bool four_waves_h = getenv("UAVENV_GRAD_4WAVES") != nullptr;    // A/B knob
        if (four_waves_h)
            rc = ring->obs_dtype == UAVENV_OBS_PACKED ? launch_grad_h<OBS_KIND_PACKED>(ga, grid, s) : launch_grad_h<OBS_KIND_F16>(ga, grid, s);
        else
            rc = ring->obs_dtype == UAVENV_OBS_PACKED ? launch_grad_h8<OBS_KIND_PACKED>(ga, grid, s) : launch_grad_h8<OBS_KIND_F16>(ga, grid, s);
    } else if (ring->obs_dtype == UAVENV_OBS_F32)
        rc = small ? launch_grad<float, 4>(ga, grid, s) : launch_grad<float, kMaxOut - 2>(ga, grid, s);
    else if (ring->obs_dtype == UAVENV_OBS_PACKED) {
        static const bool four_waves = getenv("UAVENV_GRAD_4WAVES") != nullptr;      // A/B knob
        if (small && !four_waves) {
            // UAVENV_DW1_F32=1: the round-4 form of the weight-gradient products (all on the f32 matrix pipe) -- A/B knob
            static const bool dw1_f32 = getenv("UAVENV_DW1_F32") != nullptr;
            static bool attr8 = false;
            // form B of the split products (group 1 all six flag tiles, their operands built while group 0 is in td_backward) is the
            // default: 11.48 -> 10.75 us per launch, 28.76 -> 28.2 us per configs[1] pass; UAVENV_DW1_SPLITA=1 selects form A
            static const bool dw1_b = getenv("UAVENV_DW1_SPLITA") == nullptr;
            if (!attr8) {
                if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_dqn_grad_packed8<1>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGradP8Lds) != hipSuccess ||
                    hipFuncSetAttribute(reinterpret_cast<const void *>(&k_dqn_grad_packed8<2>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGradP8Lds) != hipSuccess ||
                    hipFuncSetAttribute(reinterpret_cast<const void *>(&k_dqn_grad_packed8<0>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGradP8Lds) != hipSuccess)
                    return UAVENV_EHIP;
                attr8 = true;
            }
            if (dw1_f32) hipLaunchKernelGGL(k_dqn_grad_packed8<0>, dim3(grid), dim3(512), kGradP8Lds, s, ga);
            else if (dw1_b) hipLaunchKernelGGL(k_dqn_grad_packed8<2>, dim3(grid), dim3(512), kGradP8Lds, s, ga);
            else hipLaunchKernelGGL(k_dqn_grad_packed8<1>, dim3(grid), dim3(512), kGradP8Lds, s, ga);
            rc = UAVENV_OK;
        } else {
            rc = small ? launch_grad_packed<4>(ga, grid, s) : launch_grad_packed<kMaxOut - 2>(ga, grid, s);
        }
    }
    else
        rc = small ? launch_grad<__half, 4>(ga, grid, s) : launch_grad<__half, kMaxOut - 2>(ga, grid, s);
    if (rc != UAVENV_OK) return rc;
    return hipGetLastError() == hipSuccess ? UAVENV_OK : UAVENV_EHIP;
}

int uavenv_dqn_reduce(const UavDqnNet *net, const float *partials, int32_t n_partials, float *raw_out, void *stream)
{
    if (!net_ok(net) || !partials || !raw_out || n_partials <= 0) return UAVENV_EINVAL;
    const int P = uavenv_dqn_num_params(net);
    hipLaunchKernelGGL(k_dqn_reduce, dim3((P + 2 + 31) / 32), dim3(256), 0, (hipStream_t)stream, partials, n_partials, P,
                       uavenv_dqn_partial_stride(net), raw_out);
    return hipGetLastError() == hipSuccess ? UAVENV_OK : UAVENV_EHIP;
}

int uavenv_dqn_adam(const UavDqnNet *net, const float *raw, float lr, float beta1, float beta2, float eps, int32_t step_t,
                    int32_t hard_update, float *loss_out, void *stream)
{
    return uavenv_dqn_adam_img(net, raw, lr, beta1, beta2, eps, step_t, hard_update, loss_out, nullptr, stream);
}

int uavenv_dqn_adam_img(const UavDqnNet *net, const float *raw, float lr, float beta1, float beta2, float eps, int32_t step_t,
                        int32_t hard_update, float *loss_out, float *image_dev, void *stream)
{
    if (!net_ok(net) || !net->target || !net->m || !net->v || !raw || step_t <= 0) return UAVENV_EINVAL;
    if (image_dev && (net->w != kW || net->hid != kHid || (((uintptr_t)image_dev) & 15u) != 0)) return UAVENV_EINVAL;
    const int P = uavenv_dqn_num_params(net);
    const float bc1 = 1.0f - powf(beta1, (float)step_t);
    const float bc2 = 1.0f - powf(beta2, (float)step_t);
    hipLaunchKernelGGL(k_dqn_adam, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, net->local, net->target,
                       net->m, net->v, raw, P, lr, beta1, beta2, eps, bc1, sqrtf(bc2), hard_update, loss_out, image_dev);
    return hipGetLastError() == hipSuccess ? UAVENV_OK : UAVENV_EHIP;
}

int uavenv_dqn_reduce_adam(const UavDqnNet *net, const float *partials, int32_t n_partials, float lr, float beta1,
                           float beta2, float eps, int32_t step_t, int32_t hard_update, float *loss_out, float *raw_out,
                           void *stream)
{
    return uavenv_dqn_reduce_adam_gated(net, partials, n_partials, lr, beta1, beta2, eps, step_t, hard_update, loss_out, raw_out,
                                        nullptr, 0u, stream);
}

int uavenv_dqn_reduce_adam_gated(const UavDqnNet *net, const float *partials, int32_t n_partials, float lr, float beta1,
                                 float beta2, float eps, int32_t step_t, int32_t hard_update, float *loss_out, float *raw_out,
                                 const uint32_t *go_word_dev, uint32_t go_value, void *stream)
{
    return uavenv_dqn_reduce_adam_img(net, partials, n_partials, lr, beta1, beta2, eps, step_t, hard_update, loss_out, raw_out, go_word_dev,
                                      go_value, nullptr, stream);
}

int uavenv_dqn_reduce_adam_img(const UavDqnNet *net, const float *partials, int32_t n_partials, float lr, float beta1, float beta2,
                               float eps, int32_t step_t, int32_t hard_update, float *loss_out, float *raw_out,
                               const uint32_t *go_word_dev, uint32_t go_value, float *image_dev, void *stream)
{
    if (image_dev && (net->w != kW || net->hid != kHid || (((uintptr_t)image_dev) & 15u) != 0)) return UAVENV_EINVAL;
    if (!net_ok(net) || !net->target || !net->m || !net->v || !partials || n_partials <= 0 || step_t <= 0)
        return UAVENV_EINVAL;
    const int P = uavenv_dqn_num_params(net);
    const float bc1 = 1.0f - powf(beta1, (float)step_t);
    const float bc2 = 1.0f - powf(beta2, (float)step_t);
    hipLaunchKernelGGL(k_dqn_reduce_adam, dim3((P + 2 + 31) / 32), dim3(256), 0, (hipStream_t)stream, partials, n_partials,
                       P, uavenv_dqn_partial_stride(net), net->local, net->target, net->m, net->v, lr, beta1, beta2, eps, bc1, sqrtf(bc2), hard_update,
                       loss_out, raw_out, go_word_dev, go_value, image_dev);
    return hipGetLastError() == hipSuccess ? UAVENV_OK : UAVENV_EHIP;
}

// q_local's and q_target's fc1 / b1 -> the split form in memory (workgroup 0 / 1): exactly the staging of the kernels that convert
// for themselves (w_issue + w_issue_sc + w_commit_split), aimed at memory instead of LDS
__global__ void __launch_bounds__(256) k_dqn_split_image(const float *__restrict__ local, const float *__restrict__ target,
                                                         float *__restrict__ img)
{
    const float *net = blockIdx.x == 0 ? local : target;
    floatx4 v[kStageIters];
    SplitScRegs sc;
    w_issue(v, net);
    w_issue_sc(sc, net, net + kHid * kW);
    w_commit_split(w1split_at(img + (size_t)blockIdx.x * kSplitF), v, sc);
}

int uavenv_dqn_split_image(const UavDqnNet *net, float *image_dev, void *stream)
{
    if (!net_ok(net) || !net->target || !image_dev || net->w != kW || net->hid != kHid || (((uintptr_t)image_dev) & 15u) != 0)
        return UAVENV_EINVAL;
    static_assert(UAVENV_DQN_IMAGE_FLOATS == kSplitF, "dqn_internal.hpp");
    hipLaunchKernelGGL(k_dqn_split_image, dim3(2), dim3(256), 0, (hipStream_t)stream, net->local, net->target, image_dev);
    return hipGetLastError() == hipSuccess ? UAVENV_OK : UAVENV_EHIP;
}

int uavenv_dqn_act(const UavDqnNet *net, const void *obs_dev, int32_t obs_dtype, int32_t n, float eps, uint64_t seed,
                   uint64_t counter, int32_t *index_out, float *steer_out, float *q_out, void *stream)
{
    if (!net_ok(net) || !obs_dev || n <= 0) return UAVENV_EINVAL;
    if ((((uintptr_t)net->local | (uintptr_t)obs_dev) & 15u) != 0) return UAVENV_EINVAL;
    ActArgs g;
    g.obs = obs_dev; g.n = n; g.n_actions = net->n_actions; g.dueling = net->dueling;
    g.local = net->local; g.eps = eps; g.seed = seed; g.counter = counter;
    g.index_out = index_out; g.steer_out = steer_out; g.q_out = q_out;
    g.img = nullptr;
    const int grid = (n + kTile - 1) / kTile;
    hipStream_t s = (hipStream_t)stream;
    const bool small = net->n_actions + (net->dueling ? 1 : 0) <= 4;
    int rc;
    if (net->mfma_dtype == UAVENV_MFMA_F16) {
        if (!small || obs_dtype == UAVENV_OBS_F32) return UAVENV_EINVAL;
        rc = obs_dtype == UAVENV_OBS_PACKED ? launch_act_h<OBS_KIND_PACKED>(g, grid, s) : launch_act_h<OBS_KIND_F16>(g, grid, s);
    } else if (obs_dtype == UAVENV_OBS_F32) rc = small ? launch_act<float, 4>(g, grid, s) : launch_act<float, kMaxOut - 2>(g, grid, s);
    else if (obs_dtype == UAVENV_OBS_PACKED)
        rc = small ? launch_act_packed<4>(g, grid, s) : launch_act_packed<kMaxOut - 2>(g, grid, s);
    else rc = small ? launch_act<__half, 4>(g, grid, s) : launch_act<__half, kMaxOut - 2>(g, grid, s);
    if (rc != UAVENV_OK) return rc;
    return hipGetLastError() == hipSuccess ? UAVENV_OK : UAVENV_EHIP;
}

}  // extern "C"
