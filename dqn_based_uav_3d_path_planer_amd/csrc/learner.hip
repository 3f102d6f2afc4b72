// learner.hip -- fused DQN / DDQN / Dueling-DQN update for the reference's Q-MLPs (100-64-A) on gfx950.
//
// Replaces, per update, the ~45 PyTorch launches of  sample -> q_local(s), q_target(s') [, q_local(s')] ->
// TD target -> MSE -> backward -> Adam -> hard target copy  (Trainer/DQN_Trainer.py:85-136,
// DDQN_Trainer.py:72-117, DuelingDQN_Trainer.py:99-190, nets BaseClass/BaseCNN.py:93-139) by three kernels:
//
//   k_dqn_grad   one workgroup per 64 sampled transitions: draws the samples (same Philox stream as
//                uavenv_replay_sample), gathers the two 400 B rows straight from the replay ring into LDS,
//                runs the three 64x64x100 layer-1 products and the 64x100x64 weight-gradient product on the
//                f32 MFMA (v_mfma_f32_32x32x2_f32: exact f32 FMA chains, 157 TF class), everything else
//                (layer 2 with A <= 15 outputs, TD target, ReLU masks, bias/W2 gradients) on the VALU, and
//                writes ONE partial-gradient row per workgroup -- no atomics, deterministic.
//   k_dqn_reduce sums the partial rows -> raw[P+2] = gradient sums, loss sum, valid count (this flat vector is
//                what multi-GPU all-reduces over RCCL: the mean is then over the valid samples of ALL ranks).
//   k_dqn_adam   normalises by the valid count, torch.optim.Adam step (+ hard target copy every Update_loop).
//   k_dqn_act    Q(s) for all N envs + epsilon-greedy in one launch (same MFMA forward).
//
// The generic PyTorch-ROCm learner (learner.py) remains for other shapes and as the numerical cross-check.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/uavenv.h"
#include "uavenv_device.hpp"

using namespace uav;

namespace {

constexpr int kW = 100;        // input width   (config/Trainer.xml <w>)
constexpr int kHid = 64;       // hidden width  (<hiden_dim>)
constexpr int kTile = 64;      // samples per workgroup
constexpr int kLdx = 101;      // LDS leading dim of the 64 x 100 tiles: odd -> the 32 rows a wave reads hit 32 banks
constexpr int kLdh = 65;
constexpr int kMaxOut = 16;    // layer-2 outputs: A (+1 for the dueling value head)
constexpr int kXTile = kTile * kLdx + 32;   // + pad: the dW1 product reads 28 columns past the last row

typedef float floatx16 __attribute__((ext_vector_type(16)));

struct NetDev {
    const float *W1, *b1, *W2, *b2;
};

__device__ __forceinline__ NetDev net_view(const float *flat, int n2)
{
    NetDev n;
    n.W1 = flat;
    n.b1 = flat + kHid * kW;
    n.W2 = n.b1 + kHid;
    n.b2 = n.W2 + n2 * kHid;
    return n;
}

// 64 rows x 100 floats -> LDS tile [64][kLdx].  Rows are given by a per-row base pointer (gathered from the ring)
// or are consecutive (weights).  16-byte global loads, all issued before the LDS writes.
constexpr int kStageChunks = kTile * 25;                 // 25 chunks of 4 elements per row
constexpr int kStageIters = (kStageChunks + 255) / 256;  // 7 per thread

// Phase 1 of staging a 64 x 100 tile: put this thread's 7 global loads in flight (no wait).  Rows come from a
// per-row pointer table in LDS (gathered replay rows) or are consecutive (weights).
template <typename T>
__device__ __forceinline__ void stage_issue(float4 (&v)[kStageIters], const T *const *row_ptr_lds, const T *base_consecutive,
                                            int last_row = kTile - 1)
{
#pragma unroll
    for (int it = 0; it < kStageIters; ++it) {
        int c = it * 256 + (int)threadIdx.x;
        c = c < kStageChunks ? c : kStageChunks - 1;
        const int row = c / 25, q = c - row * 25;
        const T *src = row_ptr_lds ? row_ptr_lds[row] : base_consecutive + (size_t)(row < last_row ? row : last_row) * kW;
        if (sizeof(T) == 4) {
            v[it] = *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(src) + 4 * q);
        } else {
            const uint2 raw = *reinterpret_cast<const uint2 *>(reinterpret_cast<const __half *>(src) + 4 * q);
            v[it] = make_float4(__uint_as_float(raw.x), __uint_as_float(raw.y), 0.0f, 0.0f);   // decoded at commit
        }
    }
}

// Phase 2: LDS writes (the first use of v[] is where the compiler waits for the loads).
template <typename T>
__device__ __forceinline__ void stage_commit(float *dst, float4 (&v)[kStageIters])
{
#pragma unroll
    for (int it = 0; it < kStageIters; ++it)
        asm volatile("" : "+v"(v[it].x), "+v"(v[it].y), "+v"(v[it].z), "+v"(v[it].w));
#pragma unroll
    for (int it = 0; it < kStageIters; ++it) {
        const int c = it * 256 + (int)threadIdx.x;
        if (c < kStageChunks) {
            const int row = c / 25, q = c - row * 25;
            float4 w = v[it];
            if (sizeof(T) != 4) {
                const uint32_t rx = __float_as_uint(v[it].x), ry = __float_as_uint(v[it].y);
                const __half2 lo = *reinterpret_cast<const __half2 *>(&rx);
                const __half2 hi = *reinterpret_cast<const __half2 *>(&ry);
                w = make_float4(__low2float(lo), __high2float(lo), __low2float(hi), __high2float(hi));
            }
            float *d = dst + row * kLdx + 4 * q;
            d[0] = w.x; d[1] = w.y; d[2] = w.z; d[3] = w.w;
        }
    }
}

template <typename T>
__device__ __forceinline__ void stage_rows(float *dst, const T *const *row_ptr_lds, const T *base_consecutive)
{
    float4 v[kStageIters];
    stage_issue<T>(v, row_ptr_lds, base_consecutive);
    stage_commit<T>(dst, v);
}

// One wave: acc(32x32) += A(32 x K) * B(K x 32) with  A[i][k] = a[(i)*lda + k],  B[k][j] = b[(j)*ldb + k]
// (both operands stored "row = output index, contiguous k"): lane l feeds A[l&31][k + (l>>5)], B[k + (l>>5)][l&31].
__device__ __forceinline__ floatx16 mfma_rows(const float *a, int lda, const float *b, int ldb, int K)
{
    const int l = (int)threadIdx.x & 63;
    const float *ap = a + (l & 31) * lda + (l >> 5);
    const float *bp = b + (l & 31) * ldb + (l >> 5);
    floatx16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll 10
    for (int k = 0; k < K; k += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[k], bp[k], acc, 0, 0, 0);
    return acc;
}

// H[64][kLdh] = relu(X[64][kLdx] * W1^T + b1): wave w owns the 32x32 quadrant (w>>1, w&1).
__device__ __forceinline__ void layer1(const float *X, const float *W1, const float *b1, float *H)
{
    const int wv = (int)threadIdx.x >> 6, l = (int)threadIdx.x & 63;
    const int r0 = (wv >> 1) * 32, c0 = (wv & 1) * 32;
    const floatx16 acc = mfma_rows(X + r0 * kLdx, kLdx, W1 + c0 * kLdx, kLdx, kW);
    const int col = c0 + (l & 31);
    const float bias = b1[col];
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int row = r0 + (reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5);
        const float v = acc[reg] + bias;
        H[row * kLdh + col] = v > 0.0f ? v : 0.0f;
    }
}

// Layer 2 for a whole 64-sample tile with ALL threads of the workgroup: one (sample, output) dot product per thread
// (consecutive threads -> consecutive samples -> conflict-free H reads; W2 row broadcast), four accumulators so the
// LDS latency overlaps.  out2[s][a] = W2[a] . H[s] + b2[a].  (A thread-per-sample loop over all outputs ran on 64
// lanes only and cost 11 k cycles per tile.)
__device__ __forceinline__ void layer2_block(const float *H, const float *W2, const float *b2, int n2, float *out2)
{
    for (int item = (int)threadIdx.x; item < kTile * n2; item += 256) {
        const int smp = item & (kTile - 1), a = item >> 6;
        const float *h = H + smp * kLdh, *wr = W2 + a * kHid;
        float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
#pragma unroll
        for (int j = 0; j < kHid; j += 4) {
            s0 = fmaf(h[j], wr[j], s0);
            s1 = fmaf(h[j + 1], wr[j + 1], s1);
            s2 = fmaf(h[j + 2], wr[j + 2], s2);
            s3 = fmaf(h[j + 3], wr[j + 3], s3);
        }
        out2[smp * kMaxOut + a] = ((s0 + s1) + (s2 + s3)) + b2[a];
    }
}

// Q from the layer-2 outputs of one sample (dueling: V + A - mean A, BaseCNN.py:131-138)
__device__ __forceinline__ void q_from_out(const float *o, int n_actions, int dueling, float *q)
{
    if (dueling) {
        float mean = 0.0f;
        for (int a = 0; a < n_actions; ++a) mean += o[a];
        mean /= (float)n_actions;
        for (int a = 0; a < n_actions; ++a) q[a] = o[n_actions] + o[a] - mean;
    } else {
        for (int a = 0; a < n_actions; ++a) q[a] = o[a];
    }
}

struct GradArgs {
    UavReplayRing ring;
    int head, filled, batch;
    uint64_t seed, counter;
    const int32_t *explicit_idx;     // nullable: batch x (frame, agent) overriding the Philox draws
    const float *local, *target;     // flat parameter blocks
    int n_actions, dueling, kind;    // kind 0: max_a Q_target(s')   1: Q_target(s', argmax_a Q_local(s'))
    float gamma;
    int huber;
    float *partials;                 // [gridDim.x][P + 2]
    int P;
    unsigned long long *dbg;         // diagnostics build: 8 s_memtime stamps per workgroup
};

#ifdef UAVENV_PHASE_PROFILE
#define L_STAMP(slot) do { if (g.dbg && threadIdx.x == 0) g.dbg[(size_t)blockIdx.x * 8 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define L_STAMP(slot) do { } while (0)
#endif

template <typename ObsT>
__global__ void __launch_bounds__(256) k_dqn_grad(GradArgs g)
{
    extern __shared__ __align__(16) float lds[];
    float *Xs = lds;                       // states       [64][101]
    float *Xn = Xs + kXTile;               // next states
    float *W1 = Xn + kXTile;               // local fc1
    float *W1t = W1 + kXTile;              // target fc1
    float *Hs = W1t + kXTile;              // relu(fc1(s)) -> later dH
    float *Ht = Hs + kTile * kLdh;         // scratch hidden (local(s') then target(s'))
    float *small = Ht + kTile * kLdh;
    const int n2 = g.n_actions + (g.dueling ? 1 : 0);
    float *b1 = small;                     // [64]
    float *b1t = b1 + kHid;                // [64]
    float *W2 = b1t + kHid;                // [16][64]
    float *W2t = W2 + kMaxOut * kHid;      // [16][64]
    float *b2 = W2t + kMaxOut * kHid;      // [16]
    float *b2t = b2 + kMaxOut;             // [16]
    float *dout = b2t + kMaxOut;           // [64][16]  dL/d(layer-2 output) per sample
    float *red = dout + kTile * kMaxOut;   // [64] per-sample loss, [64] per-sample weight
    const ObsT **rows_s = reinterpret_cast<const ObsT **>(red + 2 * kTile);   // [64] row pointers
    const ObsT **rows_n = rows_s + kTile;
    int *aux = reinterpret_cast<int *>(rows_n + kTile);                       // [64] action, [64] argmax, [64] slot
    const int tid = (int)threadIdx.x;
    const NetDev nl = net_view(g.local, n2), nt = net_view(g.target, n2);

    L_STAMP(0);
    // ---- P0: draw / look up the 64 transitions of this tile, then stage everything
    if (tid < kTile) {
        const int s = (int)blockIdx.x * kTile + tid;
        int f, agent;
        if (g.explicit_idx) {
            f = g.explicit_idx[2 * s];
            agent = g.explicit_idx[2 * s + 1];
        } else {
            const ReplayPerm perm = replay_perm(g.seed, g.counter, (uint32_t)g.filled * (uint32_t)g.ring.n_agents);
            replay_slot_to_frame(replay_perm_apply(perm, (uint32_t)s), g.head, g.ring.frames, g.ring.n_agents, f, agent);
        }
        int fn = f + 1;
        if (fn >= g.ring.frames) fn = 0;
        const ObsT *obs = reinterpret_cast<const ObsT *>(g.ring.obs);
        rows_s[tid] = obs + ((size_t)f * g.ring.n_agents + agent) * kW;
        rows_n[tid] = obs + ((size_t)fn * g.ring.n_agents + agent) * kW;
        aux[2 * kTile + tid] = f * g.ring.n_agents + agent;
    }
    __syncthreads();      // row pointers visible
    // every global load of the tile in flight at once: 4 x 7 wide loads + the small vectors, ONE memory round trip
    float4 vW1[kStageIters], vW1t[kStageIters], vXs[kStageIters], vXn[kStageIters];
    stage_issue<float>(vW1, nullptr, nl.W1);
    stage_issue<float>(vW1t, nullptr, nt.W1);
    stage_issue<ObsT>(vXs, rows_s, nullptr);
    stage_issue<ObsT>(vXn, rows_n, nullptr);
    const int kb = tid < kHid ? tid : kHid - 1;
    const float pb1 = nl.b1[kb], pb1t = nt.b1[kb];
    const int k0 = tid < n2 * kHid ? tid : 0, k1 = tid + 256 < n2 * kHid ? tid + 256 : 0;
    const int k2 = tid + 512 < n2 * kHid ? tid + 512 : 0, k3 = tid + 768 < n2 * kHid ? tid + 768 : 0;
    const float pw0 = nl.W2[k0], pw1 = nl.W2[k1], pw2 = nl.W2[k2], pw3 = nl.W2[k3];
    const float pt0 = nt.W2[k0], pt1 = nt.W2[k1], pt2 = nt.W2[k2], pt3 = nt.W2[k3];
    const int kq = tid < n2 ? tid : 0;
    const float pb2 = nl.b2[kq], pb2t = nt.b2[kq];
    // this sample's scalar fields (needed only in P4: fetched now so their latency is long gone by then)
    int p_act = 0;
    float p_rew = 0.0f, p_done = 0.0f, p_valid = 1.0f;
    if (tid < kTile) {
        const int slot = aux[2 * kTile + tid];
        p_act = g.ring.action_is_index ? reinterpret_cast<const int32_t *>(g.ring.action)[slot] : 0;
        p_rew = g.ring.reward[slot];
        p_done = (float)g.ring.done[slot];
        p_valid = g.ring.valid ? (float)g.ring.valid[slot] : 1.0f;
    }
    stage_commit<float>(W1, vW1);
    stage_commit<float>(W1t, vW1t);
    stage_commit<ObsT>(Xs, vXs);
    stage_commit<ObsT>(Xn, vXn);
    if (tid < kHid) { b1[tid] = pb1; b1t[tid] = pb1t; }
    if (tid < n2 * kHid) { W2[tid] = pw0; W2t[tid] = pt0; }
    if (tid + 256 < n2 * kHid) { W2[tid + 256] = pw1; W2t[tid + 256] = pt1; }
    if (tid + 512 < n2 * kHid) { W2[tid + 512] = pw2; W2t[tid + 512] = pt2; }
    if (tid + 768 < n2 * kHid) { W2[tid + 768] = pw3; W2t[tid + 768] = pt3; }
    if (tid < n2) { b2[tid] = pb2; b2t[tid] = pb2t; }
    if (tid < 32) {       // zero the pad the dW1 product over-reads
        Xs[kTile * kLdx + tid] = 0.0f;
        Xn[kTile * kLdx + tid] = 0.0f;
    }
    if (tid < kTile) Xs[tid * kLdx + kW] = 1.0f;     // ones column: the dW1 product then yields db1 as its column 100
    __syncthreads();

    L_STAMP(1);
    // ---- P1: hidden layers on the matrix cores
    layer1(Xs, W1, b1, Hs);
    if (g.kind == 1) layer1(Xn, W1, b1, Ht);         // local net on s' (double-DQN action choice)
    __syncthreads();
    if (g.kind == 1) {
        layer2_block(Ht, W2, b2, n2, dout);           // dout is free until P4: scratch for Q_local(s')
        __syncthreads();
        if (tid < kTile) {
            float q[kMaxOut];
            q_from_out(dout + tid * kMaxOut, g.n_actions, g.dueling, q);
            int best = 0;
            for (int a = 1; a < g.n_actions; ++a) if (q[a] > q[best]) best = a;     // torch.max: first maximum
            aux[kTile + tid] = best;
        }
        __syncthreads();
    }
    layer1(Xn, W1t, b1t, Ht);                        // target net on s'
    __syncthreads();
    float *outl = Xn, *outt = Xn + kTile * kMaxOut;  // Xn (and W1t) are dead from here on: 2 x [64][16] scratch
    layer2_block(Hs, W2, b2, n2, outl);
    layer2_block(Ht, W2t, b2t, n2, outt);
    __syncthreads();

    L_STAMP(2);
    // ---- P4: TD target, loss, dL/dout per sample (Trainer/DQN_Trainer.py:107-119)
    if (tid < kTile) {
        const int act = p_act;
        const float r = p_rew, d = p_done, v = p_valid;
        float ql[kMaxOut], qt[kMaxOut];
        q_from_out(outl + tid * kMaxOut, g.n_actions, g.dueling, ql);
        q_from_out(outt + tid * kMaxOut, g.n_actions, g.dueling, qt);
        float qn;
        if (g.kind == 1) {
            qn = qt[aux[kTile + tid]];
        } else {
            qn = qt[0];
            for (int a = 1; a < g.n_actions; ++a) qn = qt[a] > qn ? qt[a] : qn;
        }
        const float y = r + (g.gamma * qn * (1.0f - d));
        const float delta = ql[act] - y;
        float per, dq;
        if (g.huber) {
            const float ad = fabsf(delta);
            per = ad < 1.0f ? 0.5f * delta * delta : ad - 0.5f;
            dq = ad < 1.0f ? delta : (delta > 0.0f ? 1.0f : -1.0f);
        } else {
            per = delta * delta;                      // MSELoss (BaseTrainer.py:40)
            dq = 2.0f * delta;
        }
        dq *= v;
        float dvals[kMaxOut];                         // compile-time indexed only (stays in registers)
        const float inv_a = 1.0f / (float)g.n_actions;
#pragma unroll
        for (int a = 0; a < kMaxOut; ++a) {
            float dv = 0.0f;
            if (g.dueling) {                          // Q = V + A - mean(A)
                if (a < g.n_actions) dv = dq * ((a == act ? 1.0f : 0.0f) - inv_a);
                else if (a == g.n_actions) dv = dq;
            } else if (a == act) {
                dv = dq;
            }
            dvals[a] = dv;
            dout[tid * kMaxOut + a] = dv;
        }
        // the loss sum and the valid count ride along as two extra dout columns: P5 sums every column over the samples
        dout[tid * kMaxOut + n2] = per * v;
        dout[tid * kMaxOut + n2 + 1] = v;
    }
    __syncthreads();

    L_STAMP(3);
    // ---- P5: layer-2 gradients and dH on the matrix cores (both are small dense products over the 64 samples; as
    // thread-per-output VALU loops they cost ~10 k cycles of LDS round trips):
    //   waves 0, 1:  dW2[a][j] = sum_s dout[s][a] * Hs[s][j]   (A[m = a][k = s], B[k = s][n = j]; M padded 16 -> 32)
    //   waves 2, 3:  dH[s][j]  = sum_a dout[s][a] * W2[a][j]   (A[m = s][k = a], B[k = a][n = j]; K = 16, a >= n2 masked)
    //   db2 / loss sum / valid count = column sums of dout (16 threads of wave 3 afterwards).
    // dH stays in registers until every wave has finished reading the forward Hs, then goes in place under the ReLU mask.
    // (f32 MFMA runs at 64 FLOP/cycle/SIMD -- no faster than the VALU: what this buys is instruction count and LDS
    // round trips, 10 k -> 6.5 k cycles.  Layer 2 and the column sums were tried the same way and were slower.)
    float *out = g.partials + (size_t)blockIdx.x * (g.P + 2);
    const int oW2 = kHid * kW + kHid, ob2 = oW2 + n2 * kHid;
    floatx16 dh0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, dh1 = dh0;
    {
        const int wv = tid >> 6, l = tid & 63, lm = l & 31, h = l >> 5;
        if (wv < 2) {
            const int n0 = wv * 32;
            floatx16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll 8
            for (int k0 = 0; k0 < kTile; k0 += 2) {
                const int smp = k0 + h;
                const float av = dout[smp * kMaxOut + (lm & (kMaxOut - 1))];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(lm < kMaxOut ? av : 0.0f, Hs[smp * kLdh + n0 + lm], acc, 0, 0, 0);
            }
#pragma unroll
            for (int reg = 0; reg < 8; ++reg) {                   // rows m < 16 live in registers 0..7
                const int m = (reg & 3) + 8 * (reg >> 2) + 4 * h;
                if (m < n2) out[oW2 + m * kHid + n0 + lm] = acc[reg];
            }
        } else {
            const int m0 = (wv - 2) * 32;
#pragma unroll
            for (int k0 = 0; k0 < kMaxOut; k0 += 2) {
                const int kk = k0 + h;
                const float dv = dout[(m0 + lm) * kMaxOut + kk];
                const bool use = kk < n2;                         // columns n2, n2+1 carry the loss / count, not a gradient;
                const float av = use ? dv : 0.0f;                 // W2 rows >= n2 are not staged (0 x garbage could be NaN)
                const float b0 = W2[(use ? kk : 0) * kHid + lm], b1 = W2[(use ? kk : 0) * kHid + 32 + lm];
                dh0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, use ? b0 : 0.0f, dh0, 0, 0, 0);
                dh1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, use ? b1 : 0.0f, dh1, 0, 0, 0);
            }
        }
        if (tid >= 192 && tid < 192 + n2 + 2) {                   // db2[a], loss sum, valid count: column sums of dout
            const int a = tid - 192;
            float sa = 0.0f;
            for (int smp = 0; smp < kTile; ++smp) sa += dout[smp * kMaxOut + a];
            if (a < n2) out[ob2 + a] = sa;
            else out[g.P + (a - n2)] = sa;
        }
    }
    __syncthreads();
    {
        const int wv = tid >> 6, l = tid & 63, lm = l & 31, h = l >> 5;
        if (wv >= 2) {
            const int m0 = (wv - 2) * 32;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int smp = m0 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                float *p0 = Hs + smp * kLdh + lm, *p1 = p0 + 32;
                *p0 = *p0 > 0.0f ? dh0[reg] : 0.0f;               // ReLU mask
                *p1 = *p1 > 0.0f ? dh1[reg] : 0.0f;
            }
        }
    }
    __syncthreads();

    L_STAMP(4);
    // ---- P6: dW1[j][k] = sum_s dH[s][j] X[s][k] on the matrix cores; db1[j] = sum_s dH[s][j]
    {
        const int wv = tid >> 6, l = tid & 63;
        const int m0 = (wv & 1) * 32;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int n0 = ((wv >> 1) * 2 + t) * 32;
            // A[m = j][kk = s] = dH[s][j]  (address s*kLdh + j);  B[kk = s][n = k] = Xs[s][k]
            const float *ap = Hs + (l >> 5) * kLdh + m0 + (l & 31);
            const float *bp = Xs + (l >> 5) * kLdx + n0 + (l & 31);
            floatx16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll 8
            for (int kk = 0; kk < kTile; kk += 2)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[kk * kLdh], bp[kk * kLdx], acc, 0, 0, 0);
            const int n = n0 + (l & 31);
            if (n <= kW) {
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int m = m0 + (reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5);
                    if (n < kW) out[m * kW + n] = acc[reg];
                    else out[kHid * kW + m] = acc[reg];          // column 100 = X's ones column -> db1[m]
                }
            }
        }
    }
    L_STAMP(5);
}

// raw[p] = sum_b partial[b][p] for p in [0, P+2): gradient sums, loss sum, valid count.  32 columns x 8 row-groups
// per workgroup so that each lane keeps nblk/8 independent, coalesced loads in flight (a one-thread-per-column loop
// over 256 partial rows was latency-bound at 63 us).
__global__ void __launch_bounds__(256) k_dqn_reduce(const float *__restrict__ partials, int nblk, int P,
                                                    float *__restrict__ raw)
{
    __shared__ float red[8][33];
    const int stride = P + 2;
    const int px = (int)threadIdx.x & 31, gy = (int)threadIdx.x >> 5;
    const int p = (int)blockIdx.x * 32 + px;
    float s = 0.0f;
    if (p < stride) {
        int b = gy;
        for (; b + 24 < nblk; b += 32) {
            const float v0 = partials[(size_t)b * stride + p], v1 = partials[(size_t)(b + 8) * stride + p];
            const float v2 = partials[(size_t)(b + 16) * stride + p], v3 = partials[(size_t)(b + 24) * stride + p];
            s += (v0 + v1) + (v2 + v3);
        }
        for (; b < nblk; b += 8) s += partials[(size_t)b * stride + p];
    }
    red[gy][px] = s;
    __syncthreads();
    if (gy == 0 && p < stride) {
        float t = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += red[k][px];
        raw[p] = t;
    }
}

// torch.optim.Adam (amsgrad off, weight_decay 0) on grad = raw / max(valid count, 1), + optional hard target copy
// (DQN_Trainer.py:121-130,138-141).  raw[P] = loss sum, raw[P+1] = valid count.
__global__ void k_dqn_adam(float *__restrict__ local, float *__restrict__ target, float *__restrict__ m,
                           float *__restrict__ v, const float *__restrict__ raw, int P, float lr, float beta1,
                           float beta2, float eps, float bc1, float bc2_sqrt, int hard_update, float *__restrict__ loss)
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
}

// Single-GPU fast path: k_dqn_reduce + k_dqn_adam in one launch (each workgroup owns 32 parameters end to end; the
// valid count is re-derived per workgroup from the nblk count cells, 1 load per thread).
__global__ void __launch_bounds__(256) k_dqn_reduce_adam(const float *__restrict__ partials, int nblk, int P,
                                                         float *__restrict__ local, float *__restrict__ target,
                                                         float *__restrict__ m, float *__restrict__ v, float lr,
                                                         float beta1, float beta2, float eps, float bc1, float bc2_sqrt,
                                                         int hard_update, float *__restrict__ loss, float *__restrict__ raw)
{
    __shared__ float red[8][33];
    __shared__ float cnt_part[4];
    __shared__ float s_inv;
    const int stride = P + 2;
    const int tid = (int)threadIdx.x;
    float c = 0.0f;
    for (int b = tid; b < nblk; b += 256) c += partials[(size_t)b * stride + P + 1];
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
    if ((tid & 63) == 0) cnt_part[tid >> 6] = c;
    const int px = tid & 31, gy = tid >> 5;
    const int p = (int)blockIdx.x * 32 + px;
    float s = 0.0f;
    if (p < stride) {
        int b = gy;
        for (; b + 24 < nblk; b += 32) {
            const float v0 = partials[(size_t)b * stride + p], v1 = partials[(size_t)(b + 8) * stride + p];
            const float v2 = partials[(size_t)(b + 16) * stride + p], v3 = partials[(size_t)(b + 24) * stride + p];
            s += (v0 + v1) + (v2 + v3);
        }
        for (; b < nblk; b += 8) s += partials[(size_t)b * stride + p];
    }
    red[gy][px] = s;
    __syncthreads();
    if (tid == 0) {
        const float cnt = (cnt_part[0] + cnt_part[1]) + (cnt_part[2] + cnt_part[3]);
        s_inv = 1.0f / (cnt > 1.0f ? cnt : 1.0f);
    }
    __syncthreads();
    if (gy == 0 && p < stride) {
        float t = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += red[k][px];
        if (raw) raw[p] = t;
        if (p < P) {
            const float gp = t * s_inv;
            const float mp = m[p] + (gp - m[p]) * (1.0f - beta1);
            const float vp = v[p] * beta2 + (1.0f - beta2) * gp * gp;
            m[p] = mp;
            v[p] = vp;
            const float np = local[p] - (lr / bc1) * (mp / (sqrtf(vp) / bc2_sqrt + eps));
            local[p] = np;
            if (hard_update) target[p] = np;
        } else if (p == P && loss) {
            *loss = t * s_inv;
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
};

// Q(s) + epsilon-greedy for a tile of 64 envs (Trainer/DuelingDQN_Trainer.py:86-97)
template <typename ObsT>
__global__ void __launch_bounds__(256) k_dqn_act(ActArgs g)
{
    extern __shared__ __align__(16) float lds[];
    float *Xs = lds;
    float *W1 = Xs + kXTile;
    float *Hs = W1 + kXTile;
    float *b1 = Hs + kTile * kLdh;
    float *W2 = b1 + kHid;
    float *b2 = W2 + kMaxOut * kHid;
    const int tid = (int)threadIdx.x;
    const int n2 = g.n_actions + (g.dueling ? 1 : 0);
    const NetDev nl = net_view(g.local, n2);
    // one memory round trip: the weight tile, the 64 observation rows (consecutive; the last tile clamps to row n-1)
    // and the small vectors are all in flight before the first LDS write
    float4 vW[kStageIters], vX[kStageIters];
    const int first = (int)blockIdx.x * kTile;
    stage_issue<float>(vW, nullptr, nl.W1);
    stage_issue<ObsT>(vX, nullptr, reinterpret_cast<const ObsT *>(g.obs) + (size_t)first * kW, g.n - 1 - first);
    const float pb1 = nl.b1[tid < kHid ? tid : kHid - 1];
    const int k0 = tid < n2 * kHid ? tid : 0, k1 = tid + 256 < n2 * kHid ? tid + 256 : 0;
    const int k2 = tid + 512 < n2 * kHid ? tid + 512 : 0, k3 = tid + 768 < n2 * kHid ? tid + 768 : 0;
    const float pw0 = nl.W2[k0], pw1 = nl.W2[k1], pw2 = nl.W2[k2], pw3 = nl.W2[k3];
    const float pb2 = nl.b2[tid < n2 ? tid : 0];
    stage_commit<float>(W1, vW);
    stage_commit<ObsT>(Xs, vX);
    if (tid < kHid) b1[tid] = pb1;
    if (tid < n2 * kHid) W2[tid] = pw0;
    if (tid + 256 < n2 * kHid) W2[tid + 256] = pw1;
    if (tid + 512 < n2 * kHid) W2[tid + 512] = pw2;
    if (tid + 768 < n2 * kHid) W2[tid + 768] = pw3;
    if (tid < n2) b2[tid] = pb2;
    __syncthreads();
    layer1(Xs, W1, b1, Hs);
    __syncthreads();
    float *out2 = Xs;                                 // the staged observations are dead after layer 1
    layer2_block(Hs, W2, b2, n2, out2);
    __syncthreads();
    if (tid < kTile) {
        const int i = (int)blockIdx.x * kTile + tid;
        if (i < g.n) {
            float q[kMaxOut];
            q_from_out(out2 + tid * kMaxOut, g.n_actions, g.dueling, q);
            if (g.q_out)
                for (int a = 0; a < g.n_actions; ++a) g.q_out[(size_t)i * g.n_actions + a] = q[a];
            const uint4 r = philox4x32_10(make_uint4((uint32_t)i, (uint32_t)g.counter, (uint32_t)(g.counter >> 32), 0xac7u),
                                          make_uint2((uint32_t)g.seed, (uint32_t)(g.seed >> 32)));
            const float sample = (float)(r.x >> 8) * (1.0f / 16777216.0f);
            int a;
            if (sample > g.eps) {
                a = 0;
                for (int k = 1; k < g.n_actions; ++k) if (q[k] > q[a]) a = k;
            } else {
                a = (int)(((uint64_t)r.y * (uint64_t)g.n_actions) >> 32);
            }
            if (g.index_out) g.index_out[i] = a;
            if (g.steer_out) g.steer_out[i] = (float)(-1.0 + 2.0 * (double)a / (double)(g.n_actions - 1));
        }
    }
}

constexpr size_t kGradLds = (size_t)(4 * kXTile + 2 * kTile * kLdh + 2 * kHid + 2 * kMaxOut * kHid + 2 * kMaxOut +
                                     kTile * kMaxOut + 2 * kTile) * 4 + 2 * kTile * 8 + 3 * kTile * 4;
constexpr size_t kActLds = (size_t)(2 * kXTile + kTile * kLdh + kHid + kMaxOut * kHid + kMaxOut) * 4 + kTile * 8;

unsigned long long *g_learner_dbg = nullptr;

bool net_ok(const UavDqnNet *n)
{
    return n && n->local && n->w == kW && n->hid == kHid && n->n_actions >= 2 &&
           n->n_actions + (n->dueling ? 1 : 0) + 2 <= kMaxOut;     // + 2 spare dout columns (loss sum, valid count)
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

int uavenv_dqn_grad(const UavReplayRing *ring, int32_t head, int32_t filled, int32_t batch, uint64_t seed,
                    uint64_t counter, const int32_t *explicit_idx, const UavDqnNet *net, int32_t kind, float gamma,
                    int32_t huber, float *partials, void *stream)
{
    if (!ring || !ring->obs || !ring->action || !ring->reward || !ring->done || !partials || !net_ok(net) || !net->target)
        return UAVENV_EINVAL;
    if (batch <= 0 || batch % kTile != 0 || ring->frames < 2 || head < 0 || head >= ring->frames) return UAVENV_EINVAL;
    if (!explicit_idx && (filled <= 0 || filled > ring->frames - 1)) return UAVENV_EINVAL;
    if (!ring->action_is_index) return UAVENV_EINVAL;
    GradArgs g;
    g.ring = *ring;
    g.head = head; g.filled = filled; g.batch = batch;
    g.seed = seed; g.counter = counter;
    g.explicit_idx = explicit_idx;
    g.local = net->local; g.target = net->target;
    g.n_actions = net->n_actions; g.dueling = net->dueling; g.kind = kind;
    g.gamma = gamma; g.huber = huber;
    g.partials = partials;
    g.P = uavenv_dqn_num_params(net);
    g.dbg = g_learner_dbg;
    const int grid = batch / kTile;
    hipStream_t s = (hipStream_t)stream;
    if (ring->obs_dtype == UAVENV_OBS_F32) {
        static bool attr32 = false;
        if (!attr32) {
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_dqn_grad<float>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGradLds) != hipSuccess)
                return UAVENV_EHIP;
            attr32 = true;
        }
        hipLaunchKernelGGL((k_dqn_grad<float>), dim3(grid), dim3(256), kGradLds, s, g);
    } else {
        static bool attr16 = false;
        if (!attr16) {
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_dqn_grad<__half>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGradLds) != hipSuccess)
                return UAVENV_EHIP;
            attr16 = true;
        }
        hipLaunchKernelGGL((k_dqn_grad<__half>), dim3(grid), dim3(256), kGradLds, s, g);
    }
    return hipGetLastError() == hipSuccess ? UAVENV_OK : UAVENV_EHIP;
}

int uavenv_dqn_reduce(const UavDqnNet *net, const float *partials, int32_t n_partials, float *raw_out, void *stream)
{
    if (!net_ok(net) || !partials || !raw_out || n_partials <= 0) return UAVENV_EINVAL;
    const int P = uavenv_dqn_num_params(net);
    hipLaunchKernelGGL(k_dqn_reduce, dim3((P + 2 + 31) / 32), dim3(256), 0, (hipStream_t)stream, partials, n_partials, P,
                       raw_out);
    return hipGetLastError() == hipSuccess ? UAVENV_OK : UAVENV_EHIP;
}

int uavenv_dqn_adam(const UavDqnNet *net, const float *raw, float lr, float beta1, float beta2, float eps, int32_t step_t,
                    int32_t hard_update, float *loss_out, void *stream)
{
    if (!net_ok(net) || !net->target || !net->m || !net->v || !raw || step_t <= 0) return UAVENV_EINVAL;
    const int P = uavenv_dqn_num_params(net);
    const float bc1 = 1.0f - powf(beta1, (float)step_t);
    const float bc2 = 1.0f - powf(beta2, (float)step_t);
    hipLaunchKernelGGL(k_dqn_adam, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, net->local, net->target,
                       net->m, net->v, raw, P, lr, beta1, beta2, eps, bc1, sqrtf(bc2), hard_update, loss_out);
    return hipGetLastError() == hipSuccess ? UAVENV_OK : UAVENV_EHIP;
}

int uavenv_dqn_reduce_adam(const UavDqnNet *net, const float *partials, int32_t n_partials, float lr, float beta1,
                           float beta2, float eps, int32_t step_t, int32_t hard_update, float *loss_out, float *raw_out,
                           void *stream)
{
    if (!net_ok(net) || !net->target || !net->m || !net->v || !partials || n_partials <= 0 || step_t <= 0)
        return UAVENV_EINVAL;
    const int P = uavenv_dqn_num_params(net);
    const float bc1 = 1.0f - powf(beta1, (float)step_t);
    const float bc2 = 1.0f - powf(beta2, (float)step_t);
    hipLaunchKernelGGL(k_dqn_reduce_adam, dim3((P + 2 + 31) / 32), dim3(256), 0, (hipStream_t)stream, partials, n_partials,
                       P, net->local, net->target, net->m, net->v, lr, beta1, beta2, eps, bc1, sqrtf(bc2), hard_update,
                       loss_out, raw_out);
    return hipGetLastError() == hipSuccess ? UAVENV_OK : UAVENV_EHIP;
}

int uavenv_dqn_act(const UavDqnNet *net, const void *obs_dev, int32_t obs_dtype, int32_t n, float eps, uint64_t seed,
                   uint64_t counter, int32_t *index_out, float *steer_out, float *q_out, void *stream)
{
    if (!net_ok(net) || !obs_dev || n <= 0) return UAVENV_EINVAL;
    ActArgs g;
    g.obs = obs_dev; g.n = n; g.n_actions = net->n_actions; g.dueling = net->dueling;
    g.local = net->local; g.eps = eps; g.seed = seed; g.counter = counter;
    g.index_out = index_out; g.steer_out = steer_out; g.q_out = q_out;
    const int grid = (n + kTile - 1) / kTile;
    hipStream_t s = (hipStream_t)stream;
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_dqn_act<float>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)kActLds) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(&k_dqn_act<__half>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)kActLds) != hipSuccess)
            return UAVENV_EHIP;
        attr = true;
    }
    if (obs_dtype == UAVENV_OBS_F32) hipLaunchKernelGGL((k_dqn_act<float>), dim3(grid), dim3(256), kActLds, s, g);
    else hipLaunchKernelGGL((k_dqn_act<__half>), dim3(grid), dim3(256), kActLds, s, g);
    return hipGetLastError() == hipSuccess ? UAVENV_OK : UAVENV_EHIP;
}

}  // extern "C"
