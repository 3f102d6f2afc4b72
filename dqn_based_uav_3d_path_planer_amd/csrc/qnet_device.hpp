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
__device__ __forceinline__ void w2_load(W2Frag<NMAX> &F, const float *W2, const float *b2, int n2)
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
template <int NMAX>
__device__ __forceinline__ void q_strip(const floatx4 (&acc)[4], const W2Frag<NMAX> &F, int n2, int n_actions, int dueling,
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

}  // namespace uavq
