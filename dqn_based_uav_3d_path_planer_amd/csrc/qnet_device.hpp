// qnet_device.hpp -- device-side building blocks of the reference's Q-MLP (100-64-A, BaseClass/BaseCNN.py:93-139) on
// gfx950, shared by the fused learner / act kernels (learner.hip) and by the env step kernel that runs the policy in its
// prologue (uavenv.hip): the wave-strip forward on v_mfma_f32_16x16x4_f32 (lane = sample, registers = hidden units),
// layer 2 + the dueling combine from registers, weight staging, packed observation rows as MFMA operands.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "uavenv_device.hpp"

namespace uavq {

using namespace uav;

constexpr int kW = 100;        // input width   (config/Trainer.xml <w>)
constexpr int kHid = 64;       // hidden width  (<hiden_dim>)
constexpr int kTile = 64;      // samples per workgroup
constexpr int kMaxOut = 16;    // layer-2 outputs: A (+1 for the dueling value head)

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

// a 64 x 100 f32 matrix as 16-byte chunks (staging of the fc1 weights)
constexpr int kStageChunks = kTile * 25;                 // 25 chunks of 4 elements per row
constexpr int kStageIters = (kStageChunks + 255) / 256;  // 7 per thread

// K is padded to 104: column 100 of the observation is 1 and column 100 of the fc1 tile is b1, so the MFMA adds the bias.
// K index of MFMA step i in lane group g = 26 g + 2 i (+1): each lane reads its operands as 13 aligned float2 per row.
constexpr int kK = 104;                     // padded K of the layer-1 products
constexpr int kLd = 108;                    // LDS row stride (floats) of the X / W1 tiles: 16-byte rows; row * 108 mod 64 walks
                                            // the multiples of 4, so the 32 lanes of a ds_read_b64 phase hit 64 distinct banks
constexpr int kLh = 68;                     // LDS row stride of the hidden tiles: rows 4 apart are 16 banks apart
constexpr int kTileF = kTile * kLd;         // floats per X / W1 tile
constexpr int kStripF = 16 * kLd;           // floats per 16-row strip of a tile
constexpr int kXIters = 7;                  // 16 rows x 25 chunks = 400 chunks per strip = 7 per lane

typedef float floatx4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ floatx4 mfma16(float a, float b, floatx4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// fc1 weights (64 x 100 f32, rows consecutive in HBM): 1 600 16-byte chunks, 7 per thread, all in flight
// (floatx4 = a plain vector value: HIP's float4 is a struct, and arrays of it copied whole become memcpys through
// scratch memory instead of registers)
__device__ __forceinline__ void w_issue(floatx4 (&v)[kStageIters], const float *W)
{
#pragma unroll
    for (int it = 0; it < kStageIters; ++it) {
        int c = it * 256 + (int)threadIdx.x;
        c = c < kStageChunks ? c : kStageChunks - 1;
        v[it] = *reinterpret_cast<const floatx4 *>(W + 4 * c);         // parameter blocks are 16-byte aligned (checked by the host)
    }
}

__device__ __forceinline__ void w_commit(float *dst, floatx4 (&v)[kStageIters], float bias)
{
#pragma unroll
    for (int it = 0; it < kStageIters; ++it) {
        const int c = it * 256 + (int)threadIdx.x;
        if (c < kStageChunks) {
            const int row = c / 25, q = c - row * 25;
            *reinterpret_cast<floatx4 *>(dst + row * kLd + 4 * q) = v[it];
        }
    }
    if (threadIdx.x < kHid)                                     // column 100 = b1, 101..103 = 0
        *reinterpret_cast<floatx4 *>(dst + (int)threadIdx.x * kLd + kW) = floatx4{bias, 0.0f, 0.0f, 0.0f};
}

// Sum over the four lane groups (lanes l, l ^ 16, l ^ 32, l ^ 48), result in all of them, on the VALU: gfx950's
// v_permlane16_swap exchanges the odd 16-lane rows of its first operand with the even rows of its second,
// v_permlane32_swap the upper half of the first with the lower half of the second; with both operands equal the two
// results are the two halves of the sum.  (Inline asm: hipcc 7.2 maps both results of the builtin to one register.
// ds_bpermute, which __shfl_xor compiles to, costs an LDS round trip per step -- ~1 k cycles per layer-2 evaluation.)
__device__ __forceinline__ float group_sum4(float v)
{
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    a += b;
    b = a;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}

// This lane's share of fc2 (+ bias): rows a < n2, hidden units 16 t + 4 g .. + 3.
template <int NMAX>
struct W2Frag {
    floatx4 w[NMAX][4];
    float b[NMAX];
};

template <int NMAX>
__device__ __forceinline__ void w2_load_n(W2Frag<NMAX> &F, const float *W2, const float *b2, int n2);
template <int NMAX>
__device__ __forceinline__ void q_strip_n(const floatx4 (&acc)[4], const W2Frag<NMAX> &F, int n2, int n_actions, int dueling, float (&q)[NMAX]);

// n2 (the number of layer-2 rows) is a run-time value, so every `a < n2` below is a scalar compare + branch -- dozens per gradient
// launch.  The reference's own shape (three actions, no value head: Qnet2 100-64-3) takes a copy in which it is a constant.
// SPEC: which shapes get their own copy -- bit 0: Qnet2 100-64-3; bit 1: VAnet2 with three actions (3 + the value head), taken by the
// f16 kernels of BASELINE configs[2] only: a third copy in the f32 kernels of configs[1] cost them more in code size than the branches
// it removes (k_step_coop<policy> 10.05 -> 10.4 us), while configs[2] gained 1.5 us per pass.
template <int NMAX, int SPEC = 1>
__device__ __forceinline__ void w2_load(W2Frag<NMAX> &F, const float *W2, const float *b2, int n2)
{
    if ((SPEC & 1) && NMAX >= 3 && n2 == 3) w2_load_n<NMAX>(F, W2, b2, 3);
    else if ((SPEC & 2) && NMAX >= 4 && n2 == 4) w2_load_n<NMAX>(F, W2, b2, 4);
    else w2_load_n<NMAX>(F, W2, b2, n2);
}

template <int NMAX>
__device__ __forceinline__ void w2_load_n(W2Frag<NMAX> &F, const float *W2, const float *b2, int n2)
{
    const int g = ((int)threadIdx.x & 63) >> 4;
#pragma unroll
    for (int a = 0; a < NMAX; ++a) {
        F.b[a] = 0.0f;
#pragma unroll
        for (int t = 0; t < 4; ++t) F.w[a][t] = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
        if (a < n2) {
            F.b[a] = b2[a];
#pragma unroll
            for (int t = 0; t < 4; ++t) F.w[a][t] = *reinterpret_cast<const floatx4 *>(W2 + a * kHid + 16 * t + 4 * g);
        }
    }
}

// layer 2 + (dueling) Q for this lane's sample from its registers: q[a], a < n_actions.  acc holds pre-activations.
template <int NMAX, int SPEC = 1>
__device__ __forceinline__ void q_strip(const floatx4 (&acc)[4], const W2Frag<NMAX> &F, int n2, int n_actions, int dueling,
                                        float (&q)[NMAX])
{
    if ((SPEC & 1) && NMAX >= 3 && n2 == 3 && n_actions == 3 && dueling == 0) q_strip_n<NMAX>(acc, F, 3, 3, 0, q);
    else if ((SPEC & 2) && NMAX >= 4 && n2 == 4 && n_actions == 3 && dueling != 0) q_strip_n<NMAX>(acc, F, 4, 3, 1, q);
    else q_strip_n<NMAX>(acc, F, n2, n_actions, dueling, q);
}

template <int NMAX>
__device__ __forceinline__ void q_strip_n(const floatx4 (&acc)[4], const W2Frag<NMAX> &F, int n2, int n_actions, int dueling,
                                          float (&q)[NMAX])
{
    floatx4 h[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) h[t][r] = acc[t][r] > 0.0f ? acc[t][r] : 0.0f;
    float o[NMAX];
#pragma unroll
    for (int a = 0; a < NMAX; ++a) {
        o[a] = 0.0f;
        if (a < n2) {
            float st[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const floatx4 wv = F.w[a][t];
                st[t] = fmaf(h[t][3], wv[3], fmaf(h[t][2], wv[2], fmaf(h[t][1], wv[1], h[t][0] * wv[0])));
            }
            o[a] = group_sum4((st[0] + st[1]) + (st[2] + st[3])) + F.b[a];
        }
    }
    if (dueling) {                                            // Q = V + A - mean(A)   (BaseCNN.py:131-138)
        float mean = 0.0f, val = 0.0f;
#pragma unroll
        for (int a = 0; a < NMAX; ++a) {
            if (a < n_actions) mean += o[a];
            if (a == n_actions) val = o[a];
        }
        mean /= (float)n_actions;
#pragma unroll
        for (int a = 0; a < NMAX; ++a) q[a] = val + o[a] - mean;
    } else {
#pragma unroll
        for (int a = 0; a < NMAX; ++a) q[a] = o[a];
    }
}

typedef uint32_t uintx4 __attribute__((ext_vector_type(4)));

struct PRow {                        // one packed observation row in registers
    uint32_t m0, m1, m2;
    float sc[11];                    // columns 0..10
    float sg[4];                     // columns 86..89
};

__device__ __forceinline__ void prow_load(PRow &R, const uint32_t *p)       // p: 16-byte aligned packed row in HBM
{
    const uintx4 a = reinterpret_cast<const uintx4 *>(p)[0], b = reinterpret_cast<const uintx4 *>(p)[1];
    const uintx4 c = reinterpret_cast<const uintx4 *>(p)[2], d = reinterpret_cast<const uintx4 *>(p)[3];
    const uintx4 e = reinterpret_cast<const uintx4 *>(p)[4];
    R.m0 = a[0]; R.m1 = a[1]; R.m2 = a[2];
#pragma unroll
    for (int k = 0; k < 4; ++k) { R.sc[k] = __uint_as_float(b[k]); R.sc[4 + k] = __uint_as_float(c[k]); }
    R.sc[8] = __uint_as_float(d[0]); R.sc[9] = __uint_as_float(d[1]); R.sc[10] = __uint_as_float(d[2]);
    R.sg[0] = __uint_as_float(d[3]);
    R.sg[1] = __uint_as_float(e[0]); R.sg[2] = __uint_as_float(e[1]); R.sg[3] = __uint_as_float(e[2]);
}

__device__ __forceinline__ void prow_store_lds(uint32_t *dst, const PRow &R)
{
    reinterpret_cast<uintx4 *>(dst)[0] = uintx4{R.m0, R.m1, R.m2, 0u};
    reinterpret_cast<uintx4 *>(dst)[1] = uintx4{__float_as_uint(R.sc[0]), __float_as_uint(R.sc[1]), __float_as_uint(R.sc[2]), __float_as_uint(R.sc[3])};
    reinterpret_cast<uintx4 *>(dst)[2] = uintx4{__float_as_uint(R.sc[4]), __float_as_uint(R.sc[5]), __float_as_uint(R.sc[6]), __float_as_uint(R.sc[7])};
    reinterpret_cast<uintx4 *>(dst)[3] = uintx4{__float_as_uint(R.sc[8]), __float_as_uint(R.sc[9]), __float_as_uint(R.sc[10]), __float_as_uint(R.sg[0])};
    reinterpret_cast<uintx4 *>(dst)[4] = uintx4{__float_as_uint(R.sg[1]), __float_as_uint(R.sg[2]), __float_as_uint(R.sg[3]), 0u};
}

// the flag bits of columns [26 g, 26 g + 26) of a row, column 26 g at bit 0 (scalar and constant columns read 0)
__device__ __forceinline__ uint32_t prow_slice(const PRow &R, int g)
{
    const uint32_t lo = g < 2 ? R.m0 : (g == 2 ? R.m1 : R.m2);
    const uint32_t hi = g < 2 ? R.m1 : (g == 2 ? R.m2 : 0u);
    const uint32_t sh = g == 0 ? 0u : g == 1 ? 26u : g == 2 ? 20u : 14u;
    return (uint32_t)(((((uint64_t)hi) << 32) | (uint64_t)lo) >> sh);
}

// columns 26 g + 2 i and 26 g + 2 i + 1 of the row (column 100 = the ones column, 101..103 = 0); i is a constant after unrolling.
// EXT: columns 101 and 102 carry two extra inputs (the action of the SAC critics' cat([state, action]), sac.hip).
template <bool EXT = false>
__device__ __forceinline__ float2 prow_pair(const PRow &R, uint32_t slice, bool g0, bool g3, int i, float e0 = 0.0f, float e1 = 0.0f)
{
    float x0 = (float)((slice >> (2 * i)) & 1u), x1 = (float)((slice >> (2 * i + 1)) & 1u);
    if (i <= 4) { x0 = g0 ? R.sc[i <= 4 ? 2 * i : 0] : x0; x1 = g0 ? R.sc[i <= 4 ? 2 * i + 1 : 0] : x1; }   // columns 0..9
    if (i == 5) x0 = g0 ? R.sc[10] : x0;                                                                     // column 10
    if (i == 4) { x0 = g3 ? R.sg[0] : x0; x1 = g3 ? R.sg[1] : x1; }                                          // 86, 87
    if (i == 5) { x0 = g3 ? R.sg[2] : x0; x1 = g3 ? R.sg[3] : x1; }                                          // 88, 89
    if (i == 11) x0 = g3 ? 1.0f : x0;                                                                        // 100
    if (EXT && i == 11) x1 = g3 ? e0 : x1;                                                                   // 101
    if (EXT && i == 12) x0 = g3 ? e1 : x0;                                                                   // 102
    return make_float2(x0, x1);
}

// fwd_strip with the B operand generated from the lane's packed row
template <bool EXT = false>
__device__ __forceinline__ void fwd_strip_packed(const float *W, const PRow &R, floatx4 (&acc)[4], float e0 = 0.0f, float e1 = 0.0f)
{
    const int lane = (int)threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    const float *wp = W + r * kLd + 26 * g;
    const uint32_t slice = prow_slice(R, g);
    const bool g0 = g == 0, g3 = g == 3;
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
    float2 b = prow_pair<EXT>(R, slice, g0, g3, 0, e0, e1);
    float2 a[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) a[t] = *reinterpret_cast<const float2 *>(wp + t * 16 * kLd);
#pragma unroll
    for (int i = 0; i < 13; ++i) {
        float2 bn = b, an[4] = {a[0], a[1], a[2], a[3]};
        if (i + 1 < 13) {
#pragma unroll
            for (int t = 0; t < 4; ++t) an[t] = *reinterpret_cast<const float2 *>(wp + t * 16 * kLd + 2 * (i + 1));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = mfma16(a[t].x, b.x, acc[t]);
        if (i + 1 < 13) bn = prow_pair<EXT>(R, slice, g0, g3, i + 1, e0, e1);       // ~8 VALU: issue in the shadow of the MFMAs around them
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = mfma16(a[t].y, b.y, acc[t]);
        // order inside this region: MFMA, then two VALU, ... (an MFMA occupies the matrix pipe for 32 cycles; the VALU pipe is free)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        b = bn;
#pragma unroll
        for (int t = 0; t < 4; ++t) a[t] = an[t];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// f16-MFMA policy for ONE wavefront = 64 agents of an f16 ring (the forward of k_dqn_act_h, learner.hip: same operands in
// the same K positions of the same four v_mfma_f32_16x16x32_f16 per accumulator, so the Q values -- and the actions -- are
// bit-identical), for the prologue of the one-wave env step kernel (uavenv_step_policy on f16 rings).  Nothing is staged
// through LDS: a lane's share of the fc1 operand (rows 16 t + r, columns 32 kk + 8 g .. + 7: the whole matrix exactly once
// per wavefront) and of the four 16-agent strips' observation rows (row r of strip st, the same columns) come straight
// from memory into registers, all in flight together with the caller's own state loads.
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ floatx4 mfma16h(half8 a, half8 b, floatx4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// ---------------------------------------------------------------------------------------------------------------------
// Layer 1 at f32 accuracy on the f16 matrix pipe (round 4: the SAC kernels of sac.hip; the policy of k_dqn_act_packed and of k_step_coop's prologue).
// 80 of the 100 observation columns are 0 / 1 flags (Agents/UAV.py:533-566): exact in f16.  fc1 is staged as TWO f16 terms,
// hi = f16(w) and mid = f16((w - hi) * 2^11) (22-23 significant bits together; the power-of-two scale keeps the residual out
// of f16's subnormals), so that the flag part of H^T = W1 X^T is three v_mfma_f32_16x16x32_f16 per term and 16 hidden units
// (K = columns 8..95 + eight idle positions: bit c of the row's 96 flag bits is column c, and columns 0..7 are scalars whose bits
// are always 0, so the term tiles drop them; the B operand of a lane = the eight flag bits of its sample's columns
// 8 + 32 kk + 8 g .., expanded to 0 / 1.0 halves -- the other scalar columns' bits are 0 in the packed words too, so their f16
// weights meet zeros) instead of twenty f32 MFMAs of twice the latency; the 16 scalar columns (0..10, 86..89), the ones column (bias) and the critics' two action
// columns go through four / five v_mfma_f32_16x16x4_f32 steps against an f32 [64][20] block.  Two accumulators per 16 hidden
// units (the mid term is summed on its own and scaled back by 2^-11 at the end); every product is exact, the sums are f32.
// MFMA cycles of one layer-1 strip: 24 x 16 + 16..20 x 32 = 0.9-1.0 k instead of 104 x 32 = 3.3 k.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kLdHs = 88;                           // halves per row of a term tile: columns 8..95 (44 dwords: the sixteen rows of a
                                                    // ds_read_b128 phase start 4 banks apart mod 64 -- all banks, once).  A staged
                                                    // layer 1 is then exactly as large as the f32 tile it replaces (kTileF floats)
constexpr int kFlagC0 = 8;                          // first column held by the term tiles
constexpr int kScK = 20;                            // scalar K: columns 0..10 | 86..89 | 100 (ones / b1) | 101 | 102 | 0 | 0
constexpr int kSplitBytes = 2 * kHid * kLdHs * 2 + kHid * kScK * 4;          // 27 648 per staged layer 1
constexpr int kSplitF = kSplitBytes / 4;
static_assert(kSplitF == kTileF, "a staged layer 1 in the split form fits where the f32 tile was");
struct W1Split {
    _Float16 *hi, *mid;
    float *sc;
};
__device__ __forceinline__ W1Split w1split_at(float *p)
{
    W1Split S;
    S.hi = reinterpret_cast<_Float16 *>(p);
    S.mid = S.hi + kHid * kLdHs;
    S.sc = reinterpret_cast<float *>(S.mid + kHid * kLdHs);
    return S;
}
// scalar-block index of observation column c (-1: a flag or constant-zero column)
__device__ __forceinline__ int sc_index(int c)
{
    return c <= 10 ? c : (c >= 86 && c <= 89 ? c - 75 : -1);
}
__device__ __forceinline__ void split_store(const W1Split &S, int row, int col, float w)
{
    if (col >= kFlagC0 && col < 96) {
        const _Float16 h = (_Float16)w;
        S.hi[row * kLdHs + col - kFlagC0] = h;
        S.mid[row * kLdHs + col - kFlagC0] = (_Float16)((w - (float)h) * 2048.0f);
    }
    const int k = sc_index(col);
    if (k >= 0) S.sc[row * kScK + k] = w;
}
// fc1 (64 x 100 f32, the registers of w_issue) + b1 -> the split form; 256 threads.  A thread's piece = four consecutive columns
// 4 q .. 4 q + 3 of one row: pieces 2..23 go to the term tiles as two 8-byte stores.  The f32 block (the 15 scalar columns, b1,
// four zeros) is loaded a second time, straight from memory, five entries per thread (w_issue_sc, in flight with w_issue's loads):
// picking those columns out of the pieces took eight conditional LDS stores per piece -- 1.4 us of a 6 us launch, measured on
// k_dqn_act_packed; columns 95..99 are constant zero in every row (their weights never meet a non-zero operand).
struct SplitScRegs {
    float v[5];
};
__device__ __forceinline__ void w_issue_sc(SplitScRegs &R, const float *W, const float *b1)
{
    const int tid = (int)threadIdx.x, row = (tid >> 2) & 63, part = tid & 3;      // (any 256 consecutive threads of a workgroup)
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int k = 5 * part + i;                             // scalar-block entry: columns 0..10 | 86..89 | b1 | 0 0 0 0
        const int col = k < 11 ? k : 75 + k;
        const float *src = k < 15 ? W + row * kW + col : b1 + row;
        const float x = *src;
        R.v[i] = k <= 15 ? x : 0.0f;
    }
}
__device__ __forceinline__ void w_commit_split(const W1Split &S, const floatx4 (&v)[kStageIters], const SplitScRegs &R)
{
    const int tid = (int)threadIdx.x & 255;                                       // (the 256 threads that called w_issue / w_issue_sc)
    // every piece is converted first, branch-free: seven independent chains for the scheduler to interleave
    half4 h[kStageIters], m[kStageIters];
#pragma unroll
    for (int it = 0; it < kStageIters; ++it)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            h[it][e] = (_Float16)v[it][e];
            m[it][e] = (_Float16)((v[it][e] - (float)h[it][e]) * 2048.0f);
        }
#pragma unroll
    for (int it = 0; it < kStageIters; ++it) {
        const int c = it * 256 + tid;
        const int row = c / 25, q = c - row * 25;
        if (c < kStageChunks && q >= 2 && q < 24) {
            *reinterpret_cast<half4 *>(S.hi + row * kLdHs + 4 * q - kFlagC0) = h[it];
            *reinterpret_cast<half4 *>(S.mid + row * kLdHs + 4 * q - kFlagC0) = m[it];
        }
    }
    float *d = S.sc + (tid >> 2) * kScK + 5 * (tid & 3);
#pragma unroll
    for (int i = 0; i < 5; ++i) d[i] = R.v[i];
}
// A staged layer 1 kept in MEMORY in the split form -- the LDS image, byte for byte (kSplitF floats) -- so that staging it is a
// straight copy: the conversion above costs a 256-thread staging team ~1.5 k cycles in front of the barrier, as much as the split
// forward saves a 16-agent strip.  The C loop keeps one image per net (built when a run starts, kept current by its own Adam
// launches: csrc/loop.hip); every other caller converts while staging -- same values, same results.
constexpr int kImgPieces = kSplitBytes / 16;                  // 1 728 pieces of 16 bytes: 7 per thread of 256
static_assert((kImgPieces + 255) / 256 == kStageIters, "the image is staged through the registers of w_issue");
__device__ __forceinline__ void img_issue(floatx4 (&v)[kStageIters], const float *img)
{
    const int tid = (int)threadIdx.x & 255;
#pragma unroll
    for (int it = 0; it < kStageIters; ++it) {
        int c = it * 256 + tid;
        c = c < kImgPieces ? c : kImgPieces - 1;
        v[it] = *reinterpret_cast<const floatx4 *>(img + 4 * c);
    }
}
__device__ __forceinline__ void img_commit(float *lds_image, const floatx4 (&v)[kStageIters])
{
    const int tid = (int)threadIdx.x & 255;
#pragma unroll
    for (int it = 0; it < kStageIters; ++it) {
        const int c = it * 256 + tid;
        if (c < kImgPieces) *reinterpret_cast<floatx4 *>(lds_image + 4 * c) = v[it];
    }
}
// parameter p of a flat block (fc1 [64][100] | b1 [64] | ...) = w into an image in memory (the Adam kernels: one thread per parameter)
__device__ __forceinline__ void img_store_param(float *img, int p, float w)
{
    const W1Split S = w1split_at(img);
    if (p < kHid * kW) {
        const int row = p / kW;
        split_store(S, row, p - row * kW, w);
    } else if (p < kHid * kW + kHid) {
        S.sc[(p - kHid * kW) * kScK + 15] = w;
    }
}

// eight flag bits -> eight halves (0 / 1.0) as the B operand of v_mfma_f32_16x16x32_f16
__device__ __forceinline__ half8 bits_to_half8(uint32_t byte)
{
    uintx4 d;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        d[k] = ((byte >> (2 * k)) & 1u ? 0x3c00u : 0u) | ((byte >> (2 * k + 1)) & 1u ? 0x3c000000u : 0u);
    return *reinterpret_cast<const half8 *>(&d);
}
// K slot q = 4 kk + g of the flag part (lane group g, K block kk): columns 8 (q + 1) .. 8 (q + 1) + 7 = byte q + 1 of the row's
// twelve flag bytes, at halves 8 q of a term-tile row; slot 11 is idle (B = 0, A re-reads slot 0: finite values)
__device__ __forceinline__ uint32_t flag_byte(const PRow &R, int kk, int g)
{
    const uint32_t lo = kk == 0 ? R.m0 : (kk == 1 ? R.m1 : R.m2), hi = kk == 0 ? R.m1 : (kk == 1 ? R.m2 : 0u);
    return g == 3 ? (hi & 0xffu) : ((lo >> (8 * (g + 1))) & 0xffu);
}
__device__ __forceinline__ int flag_aoff(int kk, int g)
{
    return (kk == 2 && g == 3) ? 0 : 32 * kk + 8 * g;
}
// the pre-activations of layer 1 for this lane's sample (the C/D layout of fwd_strip_packed: registers = hidden units
// 16 t + 4 g + reg).  EXT: scalar-block entries 16 / 17 carry the critics' action columns.
// x[4 i + g] for a lane's group g = lane >> 4 (the K index of the f32 MFMA step i): a four-way select on per-lane data.  Written as
// nested ?: it compiles to DIVERGENT CONTROL FLOW on gfx950 (s_and_saveexec / s_cbranch_execz / v_mov per arm: ~15 instructions
// per select, sixteen selects per strip); as two levels of bit-field inserts it is three v_bfi_b32.  m1 / m2 = all ones where bit 0 /
// bit 1 of g is set.
__device__ __forceinline__ float sel4(float a0, float a1, float a2, float a3, uint32_t m1, uint32_t m2)
{
    const uint32_t lo = (__float_as_uint(a1) & m1) | (__float_as_uint(a0) & ~m1);
    const uint32_t hi = (__float_as_uint(a3) & m1) | (__float_as_uint(a2) & ~m1);
    return __uint_as_float((hi & m2) | (lo & ~m2));
}

template <bool EXT>
__device__ __forceinline__ void fwd_strip_split_ahead(const W1Split &S, const PRow &R, floatx4 (&acc)[4], float e0 = 0.0f, float e1 = 0.0f)
{
    const int lane = (int)threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    constexpr int NS = EXT ? 5 : 4;
    // every operand of the strip is requested first (24 x 16 bytes + 16 / 20 floats per lane, all in flight): with one wavefront per
    // SIMD nothing else hides an LDS round trip per K block (stage II of the critic phase: the forward cost 2.0 k cycles with
    // load -> MFMA per block against 3.3 k on the f32 pipe; ~1.1 k like this)
    half8 ah[3][4], al[3][4];
    float as[NS][4];
#pragma unroll
    for (int kk = 0; kk < 3; ++kk)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            ah[kk][t] = *reinterpret_cast<const half8 *>(S.hi + (16 * t + r) * kLdHs + flag_aoff(kk, g));
            al[kk][t] = *reinterpret_cast<const half8 *>(S.mid + (16 * t + r) * kLdHs + flag_aoff(kk, g));
        }
#pragma unroll
    for (int i = 0; i < NS; ++i)
#pragma unroll
        for (int t = 0; t < 4; ++t) as[i][t] = S.sc[(16 * t + r) * kScK + 4 * i + g];
    const float x[kScK] = {R.sc[0], R.sc[1], R.sc[2], R.sc[3], R.sc[4], R.sc[5], R.sc[6], R.sc[7], R.sc[8], R.sc[9], R.sc[10],
                           R.sg[0], R.sg[1], R.sg[2], R.sg[3], 1.0f, EXT ? e0 : 0.0f, EXT ? e1 : 0.0f, 0.0f, 0.0f};
    half8 b[3];
#pragma unroll
    for (int kk = 0; kk < 3; ++kk) b[kk] = bits_to_half8(flag_byte(R, kk, g));
    float bv[NS];
    const uint32_t gm1 = 0u - (uint32_t)(g & 1), gm2 = 0u - (uint32_t)(g >> 1);
#pragma unroll
    for (int i = 0; i < NS; ++i) bv[i] = sel4(x[4 * i], x[4 * i + 1], x[4 * i + 2], x[4 * i + 3], gm1, gm2);
    floatx4 am[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { acc[t] = floatx4{0.0f, 0.0f, 0.0f, 0.0f}; am[t] = floatx4{0.0f, 0.0f, 0.0f, 0.0f}; }
    // ---- the flag columns: K = 96 slots (88 columns) in three blocks of 32, two terms
#pragma unroll
    for (int kk = 0; kk < 3; ++kk)
#pragma unroll
        for (int t = 0; t < 4; ++t) { acc[t] = mfma16h(ah[kk][t], b[kk], acc[t]); am[t] = mfma16h(al[kk][t], b[kk], am[t]); }
    // ---- the scalar columns, the ones column and (EXT) the action: f32, K index 4 i + g
#pragma unroll
    for (int i = 0; i < NS; ++i)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = mfma16(as[i][t], bv[i], acc[t]);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[t][q] = fmaf(am[t][q], 1.0f / 2048.0f, acc[t][q]);
}

// the same with the operands of one K block in flight at a time: for the kernels that run two wavefronts per SIMD (k_sac_td,
// k_sac_act), where the other wavefront hides the LDS round trips and 116 more registers per lane would not fit
template <bool EXT>
__device__ __forceinline__ void fwd_strip_split(const W1Split &S, const PRow &R, floatx4 (&acc)[4], float e0 = 0.0f, float e1 = 0.0f)
{
    const int lane = (int)threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    floatx4 am[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { acc[t] = floatx4{0.0f, 0.0f, 0.0f, 0.0f}; am[t] = floatx4{0.0f, 0.0f, 0.0f, 0.0f}; }
#pragma unroll
    for (int kk = 0; kk < 3; ++kk) {
        const half8 b = bits_to_half8(flag_byte(R, kk, g));
        half8 ah[4], al[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            ah[t] = *reinterpret_cast<const half8 *>(S.hi + (16 * t + r) * kLdHs + flag_aoff(kk, g));
            al[t] = *reinterpret_cast<const half8 *>(S.mid + (16 * t + r) * kLdHs + flag_aoff(kk, g));
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) { acc[t] = mfma16h(ah[t], b, acc[t]); am[t] = mfma16h(al[t], b, am[t]); }
    }
    const float x[kScK] = {R.sc[0], R.sc[1], R.sc[2], R.sc[3], R.sc[4], R.sc[5], R.sc[6], R.sc[7], R.sc[8], R.sc[9], R.sc[10],
                           R.sg[0], R.sg[1], R.sg[2], R.sg[3], 1.0f, EXT ? e0 : 0.0f, EXT ? e1 : 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int i = 0; i < (EXT ? 5 : 4); ++i) {
        const float bv = sel4(x[4 * i], x[4 * i + 1], x[4 * i + 2], x[4 * i + 3], 0u - (uint32_t)(g & 1), 0u - (uint32_t)(g >> 1));
        float a[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) a[t] = S.sc[(16 * t + r) * kScK + 4 * i + g];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = mfma16(a[t], bv, acc[t]);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[t][q] = fmaf(am[t][q], 1.0f / 2048.0f, acc[t][q]);
}


// ---------------------------------------------------------------------------------------------------------------------
// Round 5: building blocks of the weight-gradient product dW1^T = X^T dH in the split form (learner.hip: grad_products_split8,
// sac.hip: wgrad_x_split) -- the flag columns of X against dH as two f16 terms on v_mfma_f32_16x16x32_f16.
// ---------------------------------------------------------------------------------------------------------------------
// Max over the 64 lanes of a wavefront, result in every lane: DPP row rotations inside the 16-lane rows, then the two permlane
// swaps of group_sum4 (qnet_device.hpp) across the four rows -- no LDS round trip
__device__ __forceinline__ float wave_max64(float v)
{
#define UAV_ROW_ROR(x, n) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (x)), 0x120 + (n), 0xf, 0xf, false))
    v = fmaxf(v, UAV_ROW_ROR(v, 8));
    v = fmaxf(v, UAV_ROW_ROR(v, 4));
    v = fmaxf(v, UAV_ROW_ROR(v, 2));
    v = fmaxf(v, UAV_ROW_ROR(v, 1));
#undef UAV_ROW_ROR
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    a = fmaxf(a, b);
    b = a;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return fmaxf(a, b);
}

// The flag bits of eight samples for one column as the eight halves of an f16 MFMA operand: `one` holds 1.0 where the flag is set,
// `tiny` 2^-11 (a normal f16 number) -- the operand the MID term of a split value meets, so that hi x one + mid x tiny lands in ONE
// accumulator in the units of the hi term (mid = (w - hi) 2^11; the products are exact).  mk[i] = the first dwords of sample i's
// packed row, w = flag word, sh = bit of the column in it.
template <typename MaskVec>
__device__ __forceinline__ void flags_to_half8(const MaskVec (&mk)[8], int w, uint32_t sh, half8 &one, half8 &tiny)
{
    uintx4 d, t;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t pair = ((mk[2 * k][w] >> sh) & 1u) | (((mk[2 * k + 1][w] >> sh) & 1u) << 16);
        d[k] = pair * 0x3c00u;                               // two halves: 1.0
        t[k] = pair * 0x1000u;                               // two halves: 2^-11
    }
    one = *reinterpret_cast<const half8 *>(&d);
    tiny = *reinterpret_cast<const half8 *>(&t);
}
// The f16 scale of a tile whose largest |dH| is m: up = 2^(13 - e), down = 2^(e - 13), e = exponent of m (clamped: an all-zero or
// denormal tile scales by 2^113) -- the largest scaled value sits in [2^13, 2^14): no overflow whatever the loss does, exact scaling.
__device__ __forceinline__ void split_scale(float m, float &up, float &down)
{
    int e = (int)((__float_as_uint(m) >> 23) & 0xffu) - 127;
    e = e < -100 ? -100 : e;
    up = __uint_as_float((uint32_t)(127 + 13 - e) << 23);
    down = __uint_as_float((uint32_t)(127 - 13 + e) << 23);
}
// eight f32 values, scaled, as two f16 terms: hi = f16(w up), mid = f16((w up - hi) 2^11)
__device__ __forceinline__ void split_half8(const float (&w)[8], float up, half8 &hi, half8 &mid)
{
    _Float16 h[8], l[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float v = w[i] * up;
        h[i] = (_Float16)v;
        l[i] = (_Float16)((v - (float)h[i]) * 2048.0f);
    }
    hi = half8{h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]};
    mid = half8{l[0], l[1], l[2], l[3], l[4], l[5], l[6], l[7]};
}

// LDS of a 64-agent policy team: the fc1 tile in f16 (row stride 104 halfs = 52 dwords: the 16 rows of a ds_read_b128 phase
// start 4 banks apart and cover all 64), the 64 raw observation rows (200 bytes each, as in memory), 64 x 16 bytes of Q values.
constexpr int kPolLdW = 104;
constexpr int kPolWBytes = kHid * kPolLdW * 2;            // 13 312
constexpr int kPolXBytes = 64 * kW * 2;                   // 12 800
constexpr int kPolBytes = kPolWBytes + kPolXBytes + 64 * 16;

#ifdef UAVENV_PHASE_PROFILE
#define POL_STAMP(k) do { if (dbg && ((int)threadIdx.x & 63) == 0) dbg[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define POL_STAMP(k) do { } while (0)
#endif

// One of the TWO policy wavefronts (p = 0, 1) of a 64-agent team (first .. first + 63; the third wavefront of the workgroup
// is the agents themselves): Q(s) of agents 32 p .. 32 p + 31 into qs[agent] -- the forward of k_dqn_act_h, same operands in
// the same K positions of the same four v_mfma_f32_16x16x32_f16 per accumulator, so the Q values (and the actions the agent
// wavefront picks from them) are bit-identical.  flat: q_local's parameter block; obs: [N][100] halfs of the CURRENT frame
// (16-byte aligned; first is a multiple of 64, so the rows start on a 16-byte boundary); lds: kPolBytes owned by the team.
// Everything comes in as full 1 KiB runs (64 lanes x 16 consecutive bytes): each wavefront converts half of fc1 and copies
// its 32 rows; the MFMA lanes take their operands from LDS.  Calls __syncthreads() twice -- the agent wavefront's staging
// barrier and the hand-over of the Q values -- and so must every other wavefront of the workgroup.
// History (65 536 agents, cycles of the agent wavefront until it has its action, 4.3 k without a policy): the agent
// wavefront running the policy itself 19 k; one policy wavefront with its operand pieces loaded straight into registers
// (59 scattered loads per lane) 14.7 k, through LDS 15.1 k -- a single wavefront's dependent instruction stream (four strips x
// (16 MFMAs + ~135 VALU of layer 2), ~10 cycles per instruction) is the cost, so it is split over two.
__device__ __forceinline__ void polh_wave(int p, const float *flat, int n_actions, int dueling, const void *obs, int first,
                                          int N, unsigned char *lds, unsigned long long *dbg = nullptr)
{
    const int lane = (int)threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    const int n2 = n_actions + (dueling ? 1 : 0);
    const NetDev nl = net_view(flat, n2);
    _Float16 *Wt = reinterpret_cast<_Float16 *>(lds);
    unsigned char *xs = lds + kPolWBytes + p * (32 * kW * 2);               // this wavefront's 32 rows
    floatx4 *qs = reinterpret_cast<floatx4 *>(lds + kPolWBytes + kPolXBytes) + 32 * p;
    POL_STAMP(0);
    {
        floatx4 vw[13];
        uintx4 vx[7];
        int rows = N - (first + 32 * p);
        rows = rows < 0 ? 0 : (rows < 32 ? rows : 32);
        const int nx = rows * (kW * 2) / 16;                                  // whole 16-byte chunks inside the frame
        const uintx4 *xsrc = reinterpret_cast<const uintx4 *>(reinterpret_cast<const unsigned char *>(obs) +
                                                              (size_t)(rows > 0 ? first + 32 * p : first) * (kW * 2));
#pragma unroll
        for (int it = 0; it < 13; ++it) {
            const int c = it * 128 + p * 64 + lane;
            vw[it] = *reinterpret_cast<const floatx4 *>(nl.W1 + 4 * (c < kStageChunks ? c : kStageChunks - 1));
        }
#pragma unroll
        for (int it = 0; it < 7; ++it) {
            const int c = it * 64 + lane;
            vx[it] = xsrc[c < nx ? c : 0];
        }
        const float pb1 = nl.b1[lane];
        POL_STAMP(1);
#pragma unroll
        for (int it = 0; it < 13; ++it) {
            const int c = it * 128 + p * 64 + lane, row = c / 25, q = c - row * 25;
            if (c < kStageChunks)
                *reinterpret_cast<half4 *>(Wt + row * kPolLdW + 4 * q) =
                    half4{(_Float16)vw[it][0], (_Float16)vw[it][1], (_Float16)vw[it][2], (_Float16)vw[it][3]};
        }
        if (p == 0)                                                             // column 100 = b1, 101..103 = 0
            *reinterpret_cast<half4 *>(Wt + lane * kPolLdW + kW) = half4{(_Float16)pb1, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f};
#pragma unroll
        for (int it = 0; it < 7; ++it) {
            const int c = it * 64 + lane;
            if (c < 400) reinterpret_cast<uintx4 *>(xs)[c] = vx[it];
        }
    }
    W2Frag<4> F;
    w2_load<4, 3>(F, nl.W2, nl.b2, n2);          // fc2 fragments straight from memory (16-byte aligned: 6 464 floats in)
    __syncthreads();                             // fc1 tile complete (both halves); the agent wavefront: world staged
    POL_STAMP(2);
    const _Float16 z = (_Float16)0.0f;
    floatx4 acc[2][4];
    half8 b[2][4];
#pragma unroll
    for (int st = 0; st < 2; ++st) {
        const unsigned char *xr = xs + (16 * st + r) * (kW * 2);
#pragma unroll
        for (int kk = 0; kk < 3; ++kk) {
            const uint2 lo = *reinterpret_cast<const uint2 *>(xr + 64 * kk + 16 * g), hi = *reinterpret_cast<const uint2 *>(xr + 64 * kk + 16 * g + 8);
            const uintx4 raw = uintx4{lo.x, lo.y, hi.x, hi.y};
            b[st][kk] = *reinterpret_cast<const half8 *>(&raw);
        }
        const uint2 lo = *reinterpret_cast<const uint2 *>(xr + 192);
        const uintx4 raw = uintx4{lo.x, lo.y, 0x3c00u, 0u};                              // column 100 = 1.0 (f16 0x3c00)
        b[st][3] = g == 0 ? *reinterpret_cast<const half8 *>(&raw) : half8{z, z, z, z, z, z, z, z};
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const _Float16 *wr = Wt + (16 * t + r) * kPolLdW;
        half8 A[4];
#pragma unroll
        for (int kk = 0; kk < 3; ++kk) A[kk] = *reinterpret_cast<const half8 *>(wr + 32 * kk + 8 * g);
        // K-step 3: columns 96..99, column 100 = b1 (against the ones column of the observation), 101..103 = 0
        A[3] = g == 0 ? *reinterpret_cast<const half8 *>(wr + 96) : half8{z, z, z, z, z, z, z, z};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int st = 0; st < 2; ++st)
                acc[st][t] = mfma16h(A[kk], b[st][kk], kk == 0 ? floatx4{0.0f, 0.0f, 0.0f, 0.0f} : acc[st][t]);
    }
    POL_STAMP(3);
#pragma unroll
    for (int st = 0; st < 2; ++st) {
        float q[4];
        q_strip<4, 3>(acc[st], F, n2, n_actions, dueling, q);
        if (lane < 16) qs[16 * st + lane] = floatx4{q[0], q[1], q[2], q[3]};
    }
    POL_STAMP(4);
    __syncthreads();                             // Q values handed over
}

// epsilon-greedy over the Q values (Trainer/DuelingDQN_Trainer.py:86-97; the Philox stream of k_dqn_act*: rn of agent i)
__device__ __forceinline__ uint4 policy_philox(int i, uint64_t seed, uint64_t counter)
{
    return philox4x32_10(make_uint4((uint32_t)i, (uint32_t)counter, (uint32_t)(counter >> 32), 0xac7u),
                         make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
}

__device__ __forceinline__ int policy_select(floatx4 q, uint4 rn, float eps, int n_actions)
{
    const float sample = (float)(rn.x >> 8) * (1.0f / 16777216.0f);
    if (sample > eps) {
        int act = 0;
        float bq = q[0];
#pragma unroll
        for (int k = 1; k < 4; ++k)
            if (k < n_actions && q[k] > bq) { bq = q[k]; act = k; }
        return act;
    }
    return (int)(((uint64_t)rn.y * (uint64_t)n_actions) >> 32);
}

}  // namespace uavq
