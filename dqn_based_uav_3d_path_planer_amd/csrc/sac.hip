// sac.hip -- fused SAC (continuous actions) update for the reference's nets on gfx950.
//
// Replaces, per update, the ~270 PyTorch launches of Trainer/SAC_Trainer.py:325-379 (update, continuous branch), :122-131
// (calc_target), :145-147 (soft_update) on the nets of BaseClass/BaseCNN.py:459-500 (PolicyNetContinuous_SAC 100-64-(2+2),
// QValueNetContinuous_SAC 102-64-64-2) by four kernels:
//
//   k_sac_critic_grad   a' , log pi(a'|s') from the actor; Q_target1/2(s', a'); td target; Q1/Q2(s, a), their (weighted)
//                       MSE against it, backward -> one partial-gradient row (both critics) per workgroup
//   k_sac_reduce_adam   column sums of the partial rows -> Adam on critic_1 / critic_2 -> soft update of the targets
//   k_sac_actor_grad    a~, log pi(a~|s) from the actor; Q1/Q2(s, a~) with the UPDATED critics; dL/da~ back through the
//                       critic that holds the minimum; actor backward -> one partial row per workgroup (+ sum of log pi)
//   k_sac_reduce_adam   ... -> Adam on the actor, Adam on log_alpha
//
// Same wave-strip formulation as learner.hip (qnet_device.hpp): H^T = W X^T on v_mfma_f32_16x16x4_f32, lane = sample,
// registers = hidden units, observation operands generated in registers from the packed 80-byte rows.  The 64 -> 64
// layer chains on the MFMA without leaving the registers: the K index of a step may be ANY permutation of the hidden
// units, so step (t, reg) takes hidden unit 16 t + 4 (lane >> 4) + reg -- which is exactly the register the previous
// layer's C/D fragment left in this lane.  Weight gradients: H / dH tiles of the 64 samples and their packed rows through
// LDS, MFMA over K = samples (the observation operand decoded from the packed row), persistent accumulators, no atomics,
// deterministic.  A workgroup owns up to kTMax tiles.  Critic phase: actor + both target critics stay in LDS for one pass
// over the tiles (-> td targets, two floats per sample), then critic 1 and critic 2 one after the other, each staged once
// per workgroup.  Actor phase: actor + both critics resident, one pass.  The next tile's rows are requested before the
// current tile is computed.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/uavenv.h"
#include "uavenv_device.hpp"
#include "qnet_device.hpp"

using namespace uav;
using namespace uavq;

namespace {

thread_local char g_sac_err[256];
int sac_fail(int code, const char *msg)
{
    snprintf(g_sac_err, sizeof g_sac_err, "%s", msg);
    return code;
}

constexpr int kIn = UAVENV_SAC_CRITIC_IN;          // 102 = cat([state, action])
// flat parameter blocks (the partial-gradient rows use the same layout)
//   actor : fc1.w 64x100 | fc1.b 64 | fc_mu.w 2x64 | fc_std.w 2x64 | fc_mu.b 2 | fc_std.b 2         (= net_view(flat, 4))
//   critic: fc1.w 64x102 | fc1.b 64 | fc2.w 64x64 | fc2.b 64 | fc_out.w 2x64 | fc_out.b 2
constexpr int kPa = UAVENV_SAC_ACTOR_PARAMS;       // 6724
constexpr int kAoW2 = kHid * kW + kHid, kAob2 = kAoW2 + 4 * kHid;
constexpr int kCob1 = kHid * kIn, kCoW2 = kCob1 + kHid, kCob2 = kCoW2 + kHid * kHid, kCoWo = kCob2 + kHid, kCobo = kCoWo + 2 * kHid;
constexpr int kPc = UAVENV_SAC_CRITIC_PARAMS;      // 10882
static_assert(kAob2 + 4 == kPa && kCobo + 2 == kPc, "flat layouts");
constexpr int kStrideA = UAVENV_SAC_ACTOR_STRIDE;  // kPa + [actor loss sum, sum of log pi, 0, 0]
constexpr int kStrideC = UAVENV_SAC_CRITIC_STRIDE; // 2 kPc + [loss 1 sum, loss 2 sum, 0, 0]
static_assert(kStrideA == kPa + 4 && kStrideC == 2 * kPc + 4, "partial-row strides");
constexpr int kTMax = 8;                           // tiles per workgroup
// The 64 -> 64 layer's forward as a three-term f16 product in k_sac_td and the critic phase (layer2_fwd_h): BUILT, MEASURED, OFF.
// configs[3], one MI355X: 0.4980 ms per pass with the layer on the f32 matrix pipe, 0.5108 ms with the f16 form (all 18 SAC tests
// green either way) -- 24 f16 MFMAs replace 64 f32 ones, but splitting the lane's 16 activations (max, scale, two conversions each)
// and the second accumulator's fix-up are ~150 VALU instructions of the SAME wavefront, which do not overlap its MFMAs: with one
// wavefront per SIMD the phase is bound by its instruction stream, not by the matrix pipe.  -DUAVENV_SAC_W2_HALF builds it.
#ifdef UAVENV_SAC_W2_HALF
constexpr bool kSacW2Half = true;
#else
constexpr bool kSacW2Half = false;
#endif
#ifdef UAVENV_SAC_DW1_F32
constexpr bool kSacSplitDw1 = false;               // A/B build: the round-4 form of dW1 (all on the f32 matrix pipe)
#else
constexpr bool kSacSplitDw1 = true;                // dW1 in the split form (wgrad_x_split)
#endif
// Tiles per workgroup of a launch that covers n_slots trainers of n_tiles tiles each: as many as it takes to bring the launch
// down to one workgroup per CU (the staged nets -- ~26 k cycles per workgroup -- and the partial row are paid per workgroup,
// not per tile), at most kTMax.  BASELINE configs[3]: 4 slots x 512 tiles -> 8 tiles per workgroup, 256 workgroups; one
// slot alone -> 2.  MEASURED (configs[3] pass): 2 -> 4 tiles per workgroup 668 -> 607 us.
static int tiles_per_wg_of(int n_tiles, int n_slots, int asked)
{
    // (UAVENV_SAC_WGS: a test / A-B knob read once per process -- it changes the partition and with it the summation order;
    // anything that is not a positive number falls back to the default)
    static const int div = [] {
        const char *e = getenv("UAVENV_SAC_WGS");
        const int v = e ? atoi(e) : 0;
        return v > 0 ? v : 256;
    }();
    int tpw = asked > 0 ? asked : (int)(((long long)n_tiles * n_slots + div - 1) / div);
    if (tpw > kTMax) tpw = kTMax;
    return tpw < 1 ? 1 : tpw;
}

struct SacArgs {
    const uint32_t *obs;                  // packed rows
    const int32_t *idx_s, *idx_n;         // row of s / s' per sample (when draws == nullptr)
    const int32_t *draws;                 // batch x (frame, env): row = frame * n_agents + env * uav + slot (s' : next frame)
    int n_agents, uav, slot, frames;
    const float *act0, *act1, *reward;    // planes indexed by the row of s
    const uint8_t *done, *valid;          // valid nullable (= all 1)
    const uint4 *meta;                    // nullable: the ring's transition records {a1, a0, reward, done | valid << 8 | info << 16}, indexed
                                          // by the row of s -- ONE 16-byte gather instead of a line from each of the five planes above
    const float *eps;                     // [batch][2] N(0,1) draws of this phase's rsample()
    int batch, tiles_per_wg;
    const float *actor, *c1, *c2, *t1, *t2, *log_alpha;
    float gamma, bound;
    float *partials;
    unsigned long long *dbg;              // diagnostics build: 16 s_memtime stamps per workgroup
    // prioritised replay (Trainer/SAC_Trainer.py:336-352; nullable): importance weight of sample s in the critic losses;
    // |min(Q1, Q2)(s, a) - td_target|[:, 0] of sample s out (ReplayTree.batch_update's input, :351)
    const float *is_w;
    float *abs_td;
    int grid;                             // workgroups of this slot (a batched launch's grid.x is the maximum over its slots)
    float *td;                            // nullable [batch][2]: the td targets in global memory (k_sac_td writes, stage II reads)
};
// A launch covers up to kSlots independent SAC trainers (one per UAV slot, Envs/PathPlan_City.py:59-69): blockIdx.y = slot.
// One trainer at BASELINE configs[3]'s batch fills the chip by itself; small runs (a few tiles per slot) are a chain of
// latency-bound launches, and four slots side by side cost what one does.
constexpr int kSlots = UAVENV_SAC_LOOP_MAX_SLOTS;
struct SacArgsN {
    SacArgs s[kSlots];
};

#ifdef UAVENV_PHASE_PROFILE
#define S_STAMP(slot) do { if (g.dbg && threadIdx.x == 0) g.dbg[(size_t)blockIdx.x * 16 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define S_STAMP(slot) do { } while (0)
#endif
unsigned long long *g_sac_dbg = nullptr;

// One staged critic in LDS: fc1 as a [64][kLd] tile (column 100 = b1, 101 / 102 = the action columns), fc2 [64][kLh], b2
struct WSet {
    float *W1s, *W2s, *b2s;
};
constexpr int kWSetF = kTileF + kHid * kLh + kHid;
__device__ __forceinline__ WSet wset_at(float *p)
{
    return WSet{p, p + kTileF, p + kTileF + kHid * kLh};
}
constexpr int kPsLd = 28;                          // dwords per sample of the packed-row tile: 20 + {1, a0, a1, 0} + pad (rows
                                                   // 4 apart are 48 banks apart)
// LDS maps (floats).  Critic phase, stage I: actor fc1 | target 1 | target 2; stage II: critic | X | H1 H2 dH1 dH2 | dq | red;
// the td targets of the workgroup's tiles live behind both.  Actor phase: actor fc1 | critic 1 | critic 2 | Ps | H dH | dq | red.
constexpr int kTdSetF_ = 2 * kHid * 88 * 2 / 4 + kHid * 20 + kHid * kLh + kHid;       // = kTdSetF (defined with the split layer 1 below)
constexpr int kCritStage2F = kTdSetF_ + kTileF + 4 * kTile * kLh + kTile * 4 + 64;   // (the critic phase keeps an f32 X tile)
constexpr int kCritTdOff = kCritStage2F > kTileF + 2 * kWSetF ? kCritStage2F : kTileF + 2 * kWSetF;
constexpr size_t kSacCriticLds = (size_t)(kCritTdOff + kTMax * kTile * 3) * 4;      // td targets [.][2] + critic 1's Q[0] per sample
constexpr size_t kSacActorLds = (size_t)(kTileF + 2 * kWSetF + kTile * kPsLd + 2 * kTile * kLh + kTile * 4 + 64) * 4;   // (the split forms: same sizes)
static_assert(kSacActorLds <= 160 * 1024 && kSacCriticLds <= 160 * 1024, "LDS budget");

__device__ __forceinline__ void sample_rows(const SacArgs &g, int smp, uint32_t &rs, uint32_t &rn)
{
    if (g.draws) {
        const int f = g.draws[2 * smp], e = g.draws[2 * smp + 1];
        const int fn = f + 1 < g.frames ? f + 1 : 0;
        rs = (uint32_t)(f * g.n_agents + e * g.uav + g.slot);
        rn = (uint32_t)(fn * g.n_agents + e * g.uav + g.slot);
    } else {
        rs = (uint32_t)g.idx_s[smp];
        rn = (uint32_t)g.idx_n[smp];
    }
}

// actor fc1 (64 x 100, 16-byte aligned rows) -> W1s, column 100 = b1
__device__ __forceinline__ void stage_actor(float *W1s, const float *flat)
{
    floatx4 v[kStageIters];
    w_issue(v, flat);
    const float bias = threadIdx.x < kHid ? flat[kHid * kW + threadIdx.x] : 0.0f;
    w_commit(W1s, v, bias);
}


// actor fc1 (64 x 100 f32, rows consecutive) + b1 -> the split form; 256 threads (the first 256 of the workgroup)
__device__ __forceinline__ void stage_actor_split(const W1Split &S, const float *flat)
{
    floatx4 v[kStageIters];
    SplitScRegs sc;
    w_issue(v, flat);
    w_issue_sc(sc, flat, flat + kHid * kW);
    w_commit_split(S, v, sc);
}

// critic fc1 (64 x 102: input column c < 100 -> tile column c, the two action columns -> 101, 102, b1 -> 100) and fc2,
// in two halves so that the loads of one net are in flight while another is being written: every thread's 13 + 4 + 2
// loads are issued back to back (a load -> store loop pays one L2 round trip per iteration: 6.6 us per net, measured).
// (fc1 rows are 408 bytes: 8-byte pieces, 51 per row)
constexpr int kCritPieces = kHid * kIn / 2;                        // 3 264
constexpr int kCritIters = (kCritPieces + 255) / 256;              // 13
struct CritRegs {
    float2 w1[kCritIters];
    floatx4 w2[4];
    float b1, b2;
};
__device__ __forceinline__ void critic_issue(CritRegs &C, const float *flat)
{
    const int tid = (int)threadIdx.x;
#pragma unroll
    for (int it = 0; it < kCritIters; ++it) {
        int idx = it * 256 + tid;
        idx = idx < kCritPieces ? idx : kCritPieces - 1;
        C.w1[it] = *reinterpret_cast<const float2 *>(flat + 2 * idx);
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) C.w2[it] = *reinterpret_cast<const floatx4 *>(flat + kCoW2 + 4 * (it * 256 + tid));
    C.b1 = flat[kCob1 + (tid & 63)];
    C.b2 = flat[kCob2 + (tid & 63)];
}
__device__ __forceinline__ void critic_commit(const WSet &S, const CritRegs &C)
{
    const int tid = (int)threadIdx.x;
#pragma unroll
    for (int it = 0; it < kCritIters; ++it) {
        const int idx = it * 256 + tid;
        if (idx < kCritPieces) {
            const int row = idx / (kIn / 2), q = idx - row * (kIn / 2);
            float *dst = S.W1s + row * kLd + 2 * q;
            if (q < kW / 2) *reinterpret_cast<float2 *>(dst) = C.w1[it];
            else { dst[1] = C.w1[it].x; dst[2] = C.w1[it].y; }     // input columns 100, 101 -> tile columns 101, 102
        }
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int c = it * 256 + tid, row = c >> 4, q = c & 15;
        *reinterpret_cast<floatx4 *>(S.W2s + row * kLh + 4 * q) = C.w2[it];
    }
    if (tid < kHid) {
        S.W1s[tid * kLd + kW] = C.b1;
        S.W1s[tid * kLd + kW + 3] = 0.0f;
        S.b2s[tid] = C.b2;
    }
}
__device__ __forceinline__ void stage_critic(const WSet &S, const float *flat)
{
    CritRegs C;
    critic_issue(C, flat);
    critic_commit(S, C);
}

__device__ __forceinline__ void relu4(const floatx4 (&a)[4], floatx4 (&h)[4])
{
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) h[t][r] = a[t][r] > 0.0f ? a[t][r] : 0.0f;
}

// pre-activations of the 64 -> 64 layer from the (post-ReLU) registers of the layer below: step (t, reg) of the K loop
// is hidden unit 16 t + 4 g + reg -- the register this lane already holds
__device__ __forceinline__ void layer2_fwd(const WSet &S, const floatx4 (&h1)[4], floatx4 (&acc2)[4])
{
    const int lane = (int)threadIdx.x & 63, r = lane & 15, g = lane >> 4;
#pragma unroll
    for (int t2 = 0; t2 < 4; ++t2) acc2[t2] = *reinterpret_cast<const floatx4 *>(S.b2s + 16 * t2 + 4 * g);
    // the A operands of K block t + 1 are requested before the sixteen MFMAs of block t (one wavefront per SIMD: nothing else
    // hides the LDS round trip)
    floatx4 a[4], an[4];
#pragma unroll
    for (int t2 = 0; t2 < 4; ++t2) a[t2] = *reinterpret_cast<const floatx4 *>(S.W2s + (16 * t2 + r) * kLh + 4 * g);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        if (t + 1 < 4) {
#pragma unroll
            for (int t2 = 0; t2 < 4; ++t2) an[t2] = *reinterpret_cast<const floatx4 *>(S.W2s + (16 * t2 + r) * kLh + 16 * (t + 1) + 4 * g);
        }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg)
#pragma unroll
            for (int t2 = 0; t2 < 4; ++t2) acc2[t2] = mfma16(a[t2][reg], h1[t][reg], acc2[t2]);
#pragma unroll
        for (int t2 = 0; t2 < 4; ++t2) a[t2] = an[t2];
    }
}

// dL/dh1 = W2^T dL/dh2 (before the ReLU mask of layer 1), same trick with the roles of the two index sets swapped
__device__ __forceinline__ void layer2_bwd(const WSet &S, const floatx4 (&dh2)[4], floatx4 (&dh1)[4])
{
    const int lane = (int)threadIdx.x & 63, r = lane & 15, g = lane >> 4;
#pragma unroll
    for (int t = 0; t < 4; ++t) dh1[t] = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
    // sixteen operands (one K block: four rows of W2, four column tiles) in flight ahead of the sixteen MFMAs that use them
    float w[4][4], wn[4][4];
#pragma unroll
    for (int reg = 0; reg < 4; ++reg)
#pragma unroll
        for (int t = 0; t < 4; ++t) w[reg][t] = S.W2s[(4 * g + reg) * kLh + r + 16 * t];
#pragma unroll
    for (int t2 = 0; t2 < 4; ++t2) {
        if (t2 + 1 < 4) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg)
#pragma unroll
                for (int t = 0; t < 4; ++t) wn[reg][t] = S.W2s[(16 * (t2 + 1) + 4 * g + reg) * kLh + r + 16 * t];
        }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg)
#pragma unroll
            for (int t = 0; t < 4; ++t) dh1[t] = mfma16(w[reg][t], dh2[t2][reg], dh1[t]);
#pragma unroll
        for (int reg = 0; reg < 4; ++reg)
#pragma unroll
            for (int t = 0; t < 4; ++t) w[reg][t] = wn[reg][t];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 5: the 64 -> 64 layer's FORWARD off the f32 matrix pipe.  Neither operand is exact in f16 here, so
// both are split: fc2 as hi + mid 2^-11 (staged once per workgroup, W2Half), the lane's post-ReLU activations as hi + mid 2^-11 of
// h 2^S, S from the SAMPLE's largest activation (an MFMA's B column is one sample: a per-sample scale factors out of its output
// column exactly).  Three v_mfma_f32_16x16x32_f16 per (16 outputs, 32 inputs): hi x hi into one accumulator, mid x hi and hi x mid
// into a second one worth 2^-11 (mid x mid, 2^-22 relative, is dropped): 24 MFMAs of 16 cycles instead of 64 of 32, sums in f32,
// ~2^-21 relative per product.  The K index of a lane group keeps the chaining trick of layer2_fwd: element i of lane group g in K
// block kb is hidden unit 16 (2 kb + (i >> 2)) + 4 g + (i & 3) -- the registers this lane holds.  The staged image is stored in
// MFMA order, [t2][kb][lane][8 halves]: every A operand is one conflict-free 16-byte read.  Forward only: the backward (W2^T) would
// need a transposed image the LDS maps have no room for.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kW2hTerm = 4 * 2 * 64 * 8;                   // halves per term: 8 KB; hi + mid = 16 KB (4 096 floats)
struct W2Half {
    _Float16 *hi, *mid;
};
__device__ __forceinline__ W2Half w2half_at(float *p)
{
    W2Half I;
    I.hi = reinterpret_cast<_Float16 *>(p);
    I.mid = I.hi + kW2hTerm;
    return I;
}
typedef _Float16 half4v __attribute__((ext_vector_type(4)));
// fc2 from the registers of critic_issue (thread tid, piece it: row (it 256 + tid) >> 4, input units 4 q .. 4 q + 3); 256 threads
__device__ __forceinline__ void layer2_commit_h(const W2Half &I, const CritRegs &C)
{
    const int tid = (int)threadIdx.x & 255;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int c = it * 256 + tid, row = c >> 4, q = c & 15;
        const int t2 = row >> 4, r = row & 15, t = q >> 2, g = q & 3, kb = t >> 1;
        const int idx = (((t2 * 2 + kb) * 64 + 16 * g + r) * 8) + 4 * (t & 1);
        half4v h, l;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float w = C.w2[it][k];
            h[k] = (_Float16)w;
            l[k] = (_Float16)((w - (float)h[k]) * 2048.0f);
        }
        *reinterpret_cast<half4v *>(I.hi + idx) = h;
        *reinterpret_cast<half4v *>(I.mid + idx) = l;
    }
}
// max over the four lane groups of a sample (lanes l, l ^ 16, l ^ 32, l ^ 48), in all of them (group_sum4 with max)
__device__ __forceinline__ float group_max4(float v)
{
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    a = fmaxf(a, b);
    b = a;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return fmaxf(a, b);
}
__device__ __forceinline__ void layer2_fwd_h(const W2Half &I, const float *b2s, const floatx4 (&h1)[4], floatx4 (&acc2)[4])
{
    const int lane = (int)threadIdx.x & 63, g = lane >> 4;
    float m = 0.0f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int k = 0; k < 4; ++k) m = fmaxf(m, h1[t][k]);                  // (post-ReLU: >= 0)
    float up, down;
    split_scale(group_max4(m), up, down);
    floatx4 cm[4], cc[4];
#pragma unroll
    for (int t2 = 0; t2 < 4; ++t2) { cm[t2] = floatx4{0.0f, 0.0f, 0.0f, 0.0f}; cc[t2] = floatx4{0.0f, 0.0f, 0.0f, 0.0f}; }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        half8 ah[4], am[4];
#pragma unroll
        for (int t2 = 0; t2 < 4; ++t2) {
            ah[t2] = *reinterpret_cast<const half8 *>(I.hi + ((t2 * 2 + kb) * 64 + lane) * 8);
            am[t2] = *reinterpret_cast<const half8 *>(I.mid + ((t2 * 2 + kb) * 64 + lane) * 8);
        }
        const float w[8] = {h1[2 * kb][0], h1[2 * kb][1], h1[2 * kb][2], h1[2 * kb][3],
                            h1[2 * kb + 1][0], h1[2 * kb + 1][1], h1[2 * kb + 1][2], h1[2 * kb + 1][3]};
        half8 bh, bm;
        split_half8(w, up, bh, bm);
#pragma unroll
        for (int t2 = 0; t2 < 4; ++t2) cm[t2] = mfma16h(ah[t2], bh, cm[t2]);
#pragma unroll
        for (int t2 = 0; t2 < 4; ++t2) cc[t2] = mfma16h(am[t2], bh, cc[t2]);
#pragma unroll
        for (int t2 = 0; t2 < 4; ++t2) cc[t2] = mfma16h(ah[t2], bm, cc[t2]);
    }
#pragma unroll
    for (int t2 = 0; t2 < 4; ++t2) {
        const floatx4 b = *reinterpret_cast<const floatx4 *>(b2s + 16 * t2 + 4 * g);
#pragma unroll
        for (int k = 0; k < 4; ++k) acc2[t2][k] = fmaf(fmaf(cc[t2][k], 1.0f / 2048.0f, cm[t2][k]), down, b[k]);
    }
}

// Q(s, a) of a staged critic for this lane's sample; acc1 / acc2 keep the pre-activations
__device__ __forceinline__ void critic_fwd(const WSet &S, const PRow &R, float a0, float a1, const W2Frag<2> &Fo,
                                           floatx4 (&acc1)[4], floatx4 (&acc2)[4], float (&q)[2])
{
    fwd_strip_packed<true>(S.W1s, R, acc1, a0, a1);
    floatx4 h1[4];
    relu4(acc1, h1);
    layer2_fwd(S, h1, acc2);
    q_strip<2>(acc2, Fo, 2, 2, 0, q);
}

// dL/dq -> dL/dh2, dL/dh1 (both after their ReLU masks)
__device__ __forceinline__ void critic_bwd(const WSet &S, const W2Frag<2> &Fo, const floatx4 (&acc1)[4], const floatx4 (&acc2)[4],
                                           float dq0, float dq1, floatx4 (&dh1)[4], floatx4 (&dh2)[4])
{
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) dh2[t][r] = acc2[t][r] > 0.0f ? fmaf(Fo.w[1][t][r], dq1, Fo.w[0][t][r] * dq0) : 0.0f;
    layer2_bwd(S, dh2, dh1);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) dh1[t][r] = acc1[t][r] > 0.0f ? dh1[t][r] : 0.0f;
}

// this lane's share of dL/da_k = sum_j fc1.w[j][100 + k] dL/dh1[j] (sum over the four lane groups still to be taken)
__device__ __forceinline__ void action_grad_part(const WSet &S, const floatx4 (&dh1)[4], float &da0, float &da1)
{
    const int g = ((int)threadIdx.x & 63) >> 4;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const float *wr = S.W1s + (16 * t + 4 * g + reg) * kLd + kW + 1;
            da0 = fmaf(wr[0], dh1[t][reg], da0);
            da1 = fmaf(wr[1], dh1[t][reg], da1);
        }
}

// PolicyNetContinuous_SAC.forward after fc_mu / fc_std (BaseCNN.py:470-483, quirks included: std = tanh(softplus(.)),
// the log-prob correction applies tanh to the already squashed action)
struct ActorOut {
    float act[2], lp[2], mu[2], sd[2], spre[2];
};
__device__ __forceinline__ void actor_head(const float (&o)[4], float e0, float e1, ActorOut &A)
{
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        const float m = o[d], s = o[2 + d], e = d ? e1 : e0;
        const float mu = tanhf(m);
        const float sp = s > 20.0f ? s : log1pf(expf(s));          // F.softplus (beta 1, threshold 20)
        const float sd = tanhf(sp);
        const float ns = mu + sd * e;                               // rsample()
        const float df = ns - mu;
        float lp = -(df * df) / (2.0f * (sd * sd)) - logf(sd) - 0.9189385332046727f;     // Normal.log_prob
        const float act = tanhf(ns);
        const float th = tanhf(act);
        lp -= logf(1.0f - th * th + 1e-7f);
        A.act[d] = act; A.lp[d] = lp; A.mu[d] = mu; A.sd[d] = sd; A.spre[d] = s;
    }
}

// One action dimension of the head (the forward-only kernels: the four lane groups of a sample hold the same four head outputs, so
// group g evaluates dimension g & 1 only -- half the transcendental work; the full head above made k_sac_act VALU-bound once
// layer 1 had left the f32 matrix pipe).  Same arithmetic as actor_head for that dimension.
struct ActorOne {
    float act, lp;
};
__device__ __forceinline__ ActorOne actor_head_one(float m, float sraw, float e)
{
    const float mu = tanhf(m);
    const float sp = sraw > 20.0f ? sraw : log1pf(expf(sraw));
    const float sd = tanhf(sp);
    const float ns = mu + sd * e;
    const float df = ns - mu;
    float lp = -(df * df) / (2.0f * (sd * sd)) - logf(sd) - 0.9189385332046727f;
    const float act = tanhf(ns);
    const float th = tanhf(act);
    lp -= logf(1.0f - th * th + 1e-7f);
    return ActorOne{act, lp};
}

// the lane's packed row (+ {1, a0, a1, 0}: tile columns 100..103) into the packed-row tile; one lane per sample calls it
__device__ __forceinline__ void ps_store(uint32_t *Ps, int row, const PRow &R, float a0, float a1)
{
    uint32_t *dst = Ps + row * kPsLd;
    prow_store_lds(dst, R);
    reinterpret_cast<uintx4 *>(dst)[5] = uintx4{__float_as_uint(1.0f), __float_as_uint(a0), __float_as_uint(a1), 0u};
}

__device__ __forceinline__ void h_strip_store(float *Hs, const floatx4 (&h)[4])
{
    const int lane = (int)threadIdx.x & 63, wv = (int)threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
#pragma unroll
    for (int t = 0; t < 4; ++t) *reinterpret_cast<floatx4 *>(Hs + (16 * wv + r) * kLh + 16 * t + 4 * g) = h[t];
}

// Weight-gradient products over the tile's 64 samples (K = samples): acc[reg] += sum_s A[s][16 wave + 4 g + reg] B[s][n].
// MFMA step k, lane group g: sample (k & 3) + 16 (k >> 2) + 4 g -- rows 4 apart are 16 (48 for the packed rows) banks apart.
// The operands of four steps are requested together, then their MFMAs run (one ds_read -> wait -> MFMA per step left
// the matrix pipe idle two thirds of the time: 109 cycles per MFMA, measured).
//
// dW1: B = the f32 row the packed-row tile stands for (tile columns 0..111; 100 = 1, 101 / 102 = the action when EXT).
// The lane's column per u is fixed, so where its value sits in a row (a scalar dword, or a bit of a flag word) is
// worked out once, outside the sample loop.
template <bool EXT>
__device__ __forceinline__ void wgrad_x(const float *As, const uint32_t *Ps, floatx4 (&acc)[7])
{
    const int lane = (int)threadIdx.x & 63, wv = (int)threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
    int dw[7], sh[7];
    bool sc[7];
#pragma unroll
    for (int u = 0; u < 7; ++u) {
        const int c = 16 * u + r;
        sh[u] = 0;
        sc[u] = true;
        if (c < 11) dw[u] = 4 + c;
        else if (c >= 86 && c < 90) dw[u] = 15 + (c - 86);
        else if (c == kW) dw[u] = 20;
        else if (EXT && (c == kW + 1 || c == kW + 2)) dw[u] = 20 + (c - kW);
        else if (c >= 95) dw[u] = 3;                                // constant-zero columns (dword 3 of a row is 0)
        else { dw[u] = c >> 5; sh[u] = c & 31; sc[u] = false; }
    }
    const float *ap = As + 4 * g * kLh + 16 * wv + r;
    const uint32_t *pp = Ps + 4 * g * kPsLd;
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
        float a[4];
        uint32_t w[4][7];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int s = kk + 16 * kc;
            a[kk] = ap[s * kLh];
#pragma unroll
            for (int u = 0; u < 7; ++u) w[kk][u] = pp[s * kPsLd + dw[u]];
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int u = 0; u < 7; ++u) {
                const float b = sc[u] ? __uint_as_float(w[kk][u]) : (float)((w[kk][u] >> sh[u]) & 1u);
                acc[u] = mfma16(a[kk], b, acc[u]);
            }
    }
}
// this strip's 16 rows of an f32 X tile [64][kLd] (columns 0..103; 100 = 1, 101 / 102 = the action) from the lanes' packed rows
__device__ __forceinline__ void x_strip_store(float *Xs, const PRow &R, float a0, float a1)
{
    const int lane = (int)threadIdx.x & 63, wv = (int)threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
    const uint32_t slice = prow_slice(R, g);
    float *dst = Xs + (16 * wv + r) * kLd + 26 * g;
#pragma unroll
    for (int i = 0; i < 13; ++i) *reinterpret_cast<float2 *>(dst + 2 * i) = prow_pair<true>(R, slice, g == 0, g == 3, i, a0, a1);
}
// dW1 with B read from that tile: no decode between the MFMAs (the VALU of a wavefront does not overlap its own MFMAs: the
// packed-row decode made this product 7.5 k cycles per tile against 3.6 k of MFMA issue; the actor phase has no LDS for an
// f32 tile next to its three resident nets and keeps wgrad_x)
__device__ __forceinline__ void wgrad_xf(const float *As, const float *Xs, floatx4 (&acc)[7])
{
    const int lane = (int)threadIdx.x & 63, wv = (int)threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
    const float *ap = As + 4 * g * kLh + 16 * wv + r;
    const float *bp = Xs + 4 * g * kLd + r;
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
        float a[4], b[4][7];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int s = kk + 16 * kc;
            a[kk] = ap[s * kLh];
#pragma unroll
            for (int u = 0; u < 7; ++u) b[kk][u] = bp[s * kLd + 16 * u];
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int u = 0; u < 7; ++u) acc[u] = mfma16(a[kk], b[kk][u], acc[u]);
    }
}
// ---------------------------------------------------------------------------------------------------------------------
// Round 5: dW1 in the SPLIT form (as learner.hip: grad_products_split8).  80 of the 100 observation columns are 0 / 1 flags -- exact
// in f16 -- so their part of dW1[j][c] = sum_s dH1[s][j] X[s][c] runs on v_mfma_f32_16x16x32_f16 with dH1 as two f16 terms (hi +
// mid 2^-11 of dH1 2^S; 2^S from the tile's largest |dH1|: no overflow, 22-23 significant bits relative to it; every product
// exact, f32 sums); the 15 scalar columns and the ones column (-> db1) form one gathered 16-entry tile on v_mfma_f32_16x16x4_f32
// with the f32 dH1; the critics' two action columns are 32 FMAs per lane.  Per wavefront and tile: 24 f16 MFMAs of 16 cycles + 16
// f32 MFMAs of 32 instead of 112 f32 MFMAs -- 0.9 k matrix cycles instead of 3.6 k (measured before: 5.5 k cycles per tile for
// the product on an f32 X tile, 7.5 k with the packed-row decode).
// A = dH1 (rows = this wavefront's 16 hidden units), B = the flags; sample of K element i, lane group g, step q: 32 q + 16 (i >> 2)
// + 4 g + (i & 3) -- lane groups 4 samples apart are 16 banks apart in both tiles.  fl[u][reg] = dW1[16 wave + 4 g + reg][16 u + r]
// (exact zeros at the scalar columns: their bits are 0 in the packed words); sc[reg] = the gathered entry r of those hidden units
// (entries 0..10 = columns 0..10, 11..14 = 86..89, 15 = ones); act[0 / 1] = this lane's partial sums for hidden unit 16 wave + r
// and the two action columns (summed over the lane groups at write-out).
// ---------------------------------------------------------------------------------------------------------------------
struct Dw1Split {
    floatx4 fl[6], sc;
    float act[2];
};
__device__ __forceinline__ void dw1_zero(Dw1Split &W)
{
#pragma unroll
    for (int u = 0; u < 6; ++u) W.fl[u] = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
    W.sc = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
    W.act[0] = W.act[1] = 0.0f;
}
// max |dH1| of this wavefront's strip, for the tile's scale: lane 0 leaves it in amax4[wave] (read after the tile's barrier)
__device__ __forceinline__ void dh_strip_amax(const floatx4 (&dh)[4], float *amax4)
{
    float m = 0.0f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int k = 0; k < 4; ++k) m = fmaxf(m, fabsf(dh[t][k]));
    m = wave_max64(m);
    if (((int)threadIdx.x & 63) == 0) amax4[(int)threadIdx.x >> 6] = m;
}
template <bool EXT>
__device__ __forceinline__ void wgrad_x_split(const float *dHs, const uint32_t *Ps, const float *amax4, Dw1Split &W)
{
    const int lane = (int)threadIdx.x & 63, wv = (int)threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
    float up, down;
    split_scale(fmaxf(fmaxf(amax4[0], amax4[1]), fmaxf(amax4[2], amax4[3])), up, down);
    const float *ap = dHs + 4 * g * kLh + 16 * wv + r;
    const uint32_t *pp = Ps + 4 * g * kPsLd;
    const int dsc = r < 15 ? 4 + r : 20;                        // gathered entry r of a row: dwords 4..18, the ones at dword 20
    floatx4 ch[6], c32 = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int u = 0; u < 6; ++u) ch[u] = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
    float s0 = 0.0f, s1a = 0.0f;
    // the two K blocks one after the other (both at once: ~180 live registers on top of the phase's persistent accumulators --
    // the critic kernel spilled 23 of them)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        uintx4 mk[8];
        float dh[8], xs[8], xa0[8], xa1[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int s1 = 32 * q + 16 * (i >> 2) + (i & 3);    // + 4 g through the base pointers
            mk[i] = *reinterpret_cast<const uintx4 *>(pp + s1 * kPsLd);
            dh[i] = ap[s1 * kLh];
            xs[i] = __uint_as_float(pp[s1 * kPsLd + dsc]);
            if (EXT) {
                xa0[i] = __uint_as_float(pp[s1 * kPsLd + 21]);
                xa1[i] = __uint_as_float(pp[s1 * kPsLd + 22]);
            }
        }
        half8 ah, am;
        split_half8(dh, up, ah, am);
#pragma unroll
        for (int u0 = 0; u0 < 6; u0 += 3) {                  // three column tiles at a time: six operands live
            half8 b1[3], bt[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) flags_to_half8(mk, (u0 + k) >> 1, (uint32_t)(16 * ((u0 + k) & 1) + r), b1[k], bt[k]);
#pragma unroll
            for (int k = 0; k < 3; ++k) ch[u0 + k] = mfma16h(ah, b1[k], ch[u0 + k]);
#pragma unroll
            for (int k = 0; k < 3; ++k) ch[u0 + k] = mfma16h(am, bt[k], ch[u0 + k]);    // (mid x 2^-11: same accumulator, three MFMAs later)
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) c32 = mfma16(dh[i], xs[i], c32);
        if (EXT) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { s0 = fmaf(dh[i], xa0[i], s0); s1a = fmaf(dh[i], xa1[i], s1a); }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int u = 0; u < 6; ++u)
#pragma unroll
        for (int k = 0; k < 4; ++k) W.fl[u][k] += ch[u][k] * down;
#pragma unroll
    for (int k = 0; k < 4; ++k) W.sc[k] += c32[k];
    if (EXT) {
        W.act[0] += s0;
        W.act[1] += s1a;
    }
}
// ... -> the partial row (the layout of store_dw1: input column c < 100 at [j][c], b1 at ob1 + j, the action columns at [j][100], [j][101])
template <bool EXT>
__device__ __forceinline__ void store_dw1_split(float *out, int in_dim, int ob1, Dw1Split &W)
{
    const int lane = (int)threadIdx.x & 63, wv = (int)threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
#pragma unroll
    for (int u = 0; u < 6; ++u) {
        const int c = 16 * u + r;
        const bool scalar = c <= 10 || (c >= 86 && c <= 89);          // written from the gathered tile below
#pragma unroll
        for (int reg = 0; reg < 4; ++reg)
            if (!scalar) out[(16 * wv + 4 * g + reg) * in_dim + c] = W.fl[u][reg];
    }
    const int cg = r <= 10 ? r : 86 + (r - 11);                       // gathered entry r -> its column (entry 15: b1)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int j = 16 * wv + 4 * g + reg;
        if (r < 15) out[j * in_dim + cg] = W.sc[reg];
        else out[ob1 + j] = W.sc[reg];
        if (r < 4) out[j * in_dim + 96 + r] = 0.0f;                   // columns 96..99 are constant zero in every row
    }
    if (EXT) {
        const float a0 = group_sum4(W.act[0]), a1 = group_sum4(W.act[1]);
        if (g == 0) { out[(16 * wv + r) * in_dim + kW] = a0; out[(16 * wv + r) * in_dim + kW + 1] = a1; }
    }
}

// the critic's other products in one sweep: dW2 = dH2^T H1 (4 tiles), dWout^T = H2^T dq, db2 = column sums of dH2
__device__ __forceinline__ void wgrad_critic_rest(const float *dH2s, const float *H1s, const float *H2s, const float *dqs,
                                                  floatx4 (&aw2)[4], floatx4 &awo, floatx4 &ab2)
{
    const int lane = (int)threadIdx.x & 63, wv = (int)threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
    const int ao = 4 * g * kLh + 16 * wv + r;
    const float *bp = H1s + 4 * g * kLh + r;
    const float one = r == 0 ? 1.0f : 0.0f;
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
        float ad[4], ah[4], b[4][4], bq[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int s = kk + 16 * kc;
            ad[kk] = dH2s[ao + s * kLh];
            ah[kk] = H2s[ao + s * kLh];
#pragma unroll
            for (int u = 0; u < 4; ++u) b[kk][u] = bp[s * kLh + 16 * u];
            bq[kk] = r < 2 ? dqs[(s + 4 * g) * 4 + r] : 0.0f;
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
            for (int u = 0; u < 4; ++u) aw2[u] = mfma16(ad[kk], b[kk][u], aw2[u]);
            awo = mfma16(ah[kk], bq[kk], awo);
            ab2 = mfma16(ad[kk], one, ab2);
        }
    }
}
// acc += A^T small, small: [64][4], columns n >= nb read as 0
__device__ __forceinline__ void wgrad_small(const float *As, const float *small, int nb, floatx4 &acc)
{
    const int lane = (int)threadIdx.x & 63, wv = (int)threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
    const float *ap = As + 4 * g * kLh + 16 * wv + r;
    float a[16], b[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int s0 = (k & 3) + 16 * (k >> 2);
        a[k] = ap[s0 * kLh];
        b[k] = r < nb ? small[(s0 + 4 * g) * 4 + r] : 0.0f;
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) acc = mfma16(a[k], b[k], acc);
}

// fc1 gradient tiles -> the partial row: tile column c < 100 -> input column c, 100 -> b1, 101 / 102 -> the action columns
__device__ __forceinline__ void store_dw1(float *out, int in_dim, int ob1, const floatx4 (&acc)[7])
{
    const int lane = (int)threadIdx.x & 63, wv = (int)threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
#pragma unroll
    for (int u = 0; u < 7; ++u) {
        const int c = 16 * u + r;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int j = 16 * wv + 4 * g + reg;
            if (c < kW) out[j * in_dim + c] = acc[u][reg];
            else if (c == kW) out[ob1 + j] = acc[u][reg];
            else if (c < kW + 1 + (in_dim - kW)) out[j * in_dim + c - 1] = acc[u][reg];
        }
    }
}

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// what a tile's pass needs from HBM for this lane's sample (requested one tile ahead)
struct TileIn {
    PRow R;
    float a0, a1, rew, nd, w, e0, e1, isw;
};
template <bool NEXT, bool CRITIC>
__device__ __forceinline__ void tile_in(const SacArgs &g, int tile, TileIn &T)
{
    const int lane = (int)threadIdx.x & 63, wv = (int)threadIdx.x >> 6, r = lane & 15;
    const int smp = tile * kTile + 16 * wv + r;
    uint32_t rs, rn;
    sample_rows(g, smp, rs, rn);
    prow_load(T.R, g.obs + (size_t)(NEXT ? rn : rs) * kPackedDwords);
    T.a0 = T.a1 = T.rew = T.nd = 0.0f;
    T.w = 1.0f;
    T.isw = (CRITIC && !NEXT && g.is_w) ? g.is_w[smp] : 1.0f;
    if (g.meta) {                         // the transition's scalars from its one record (include/uavenv.h: UavReplayRing.meta)
        const uint4 m = g.meta[rs];
        if (CRITIC) {
            if (NEXT) { T.rew = __uint_as_float(m.z); T.nd = 1.0f - (float)(m.w & 0xffu); }
            else { T.a0 = __uint_as_float(m.y); T.a1 = __uint_as_float(m.x); }
        }
        if (!NEXT) T.w = (float)((m.w >> 8) & 0xffu);
    } else {
        if (CRITIC) {
            if (NEXT) { T.rew = g.reward[rs]; T.nd = 1.0f - (float)g.done[rs]; }
            else { T.a0 = g.act0[rs]; T.a1 = g.act1[rs]; }
        }
        // rows of agents that were only waiting for their team-mates (valid = 0) are not replay memory: weight 0 in every loss
        if (!NEXT) T.w = g.valid ? (float)g.valid[rs] : 1.0f;
    }
    if (NEXT || !CRITIC) { T.e0 = g.eps[2 * smp]; T.e1 = g.eps[2 * smp + 1]; } else { T.e0 = T.e1 = 0.0f; }
}

// one staged target critic of k_sac_td: layer 1 in the split form, fc2 [64][kLh] and b2 as in WSet
struct TdSet {
    W1Split W1;
    float *W2s, *b2s;
};
constexpr int kTdSetF = kSplitF + kHid * kLh + kHid;
static_assert(kTdSetF == kTdSetF_, "LDS map");
static_assert(2 * kW2hTerm / 2 <= kHid * kLh && kTile * kPsLd + 2 * kW2hTerm / 2 <= kTileF,
              "fc2's f16 image fits where the f32 fc2 tile was (k_sac_td) and behind the packed rows in the critic phase's X area");
constexpr size_t kSacTdLds = (size_t)(kSplitF + 2 * kTdSetF) * 4;      // actor layer 1 + both target critics: 120 KB
__device__ __forceinline__ TdSet tdset_at(float *p)
{
    return TdSet{w1split_at(p), p + kSplitF, p + kSplitF + kHid * kLh};
}
// critic fc1 (64 x 102: the two action columns -> scalar-block entries 16 / 17, b1 -> 15), fc2, b2 from the registers of
// critic_issue; 256 threads
// W2H: fc2 as the f16 image of layer2_fwd_h IN PLACE of the f32 tile (forward-only kernels: k_sac_td)
template <bool W2H = false>
__device__ __forceinline__ void critic_commit_split(const TdSet &S, const CritRegs &C)
{
    const int tid = (int)threadIdx.x;
#pragma unroll
    for (int it = 0; it < kCritIters; ++it) {
        const int idx = it * 256 + tid;
        if (idx < kCritPieces) {
            const int row = idx / (kIn / 2), q = idx - row * (kIn / 2);
            if (q < kW / 2) {
                split_store(S.W1, row, 2 * q, C.w1[it].x);
                split_store(S.W1, row, 2 * q + 1, C.w1[it].y);
            } else {                                    // input columns 100, 101: the action
                S.W1.sc[row * kScK + 16] = C.w1[it].x;
                S.W1.sc[row * kScK + 17] = C.w1[it].y;
            }
        }
    }
    if (W2H) {
        layer2_commit_h(w2half_at(S.W2s), C);
    } else {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int c = it * 256 + tid, row = c >> 4, q = c & 15;
            *reinterpret_cast<floatx4 *>(S.W2s + row * kLh + 4 * q) = C.w2[it];
        }
    }
    if (tid < kHid) {
        float *d = S.W1.sc + tid * kScK;
        d[15] = C.b1; d[18] = 0.0f; d[19] = 0.0f;
        S.b2s[tid] = C.b2;
    }
}
// w2h: the 64 -> 64 layer from this f16 image (layer2_fwd_h) instead of S.W2s on the f32 pipe (nullptr)
template <bool AHEAD = false>
__device__ __forceinline__ void critic_fwd_split(const TdSet &S, const PRow &R, float a0, float a1, const W2Frag<2> &Fo,
                                                 floatx4 (&acc1)[4], floatx4 (&acc2)[4], float (&q)[2], float *w2h = nullptr)
{
    if (AHEAD) fwd_strip_split_ahead<true>(S.W1, R, acc1, a0, a1);
    else fwd_strip_split<true>(S.W1, R, acc1, a0, a1);
    floatx4 h1[4];
    relu4(acc1, h1);
    if (w2h) layer2_fwd_h(w2half_at(w2h), S.b2s, h1, acc2);
    else layer2_fwd(WSet{nullptr, S.W2s, S.b2s}, h1, acc2);
    q_strip<2>(acc2, Fo, 2, 2, 0, q);
}

// action_grad_part on a critic staged in the split form (its action columns are scalar-block entries 16 / 17)
__device__ __forceinline__ void action_grad_part_split(const TdSet &S, const floatx4 (&dh1)[4], float &da0, float &da1)
{
    const int g = ((int)threadIdx.x & 63) >> 4;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const float *wr = S.W1.sc + (16 * t + 4 * g + reg) * kScK + 16;
            da0 = fmaf(wr[0], dh1[t][reg], da0);
            da1 = fmaf(wr[1], dh1[t][reg], da1);
        }
}

// stage I's inputs for this lane's sample of tile `tile` (wavefront wv of the four that share the tile)
struct TileInS {
    PRow R;
    float rew, nd, e0, e1;
};
__device__ __forceinline__ void tile_in_s(const SacArgs &g, int tile, int wv, TileInS &T)
{
    const int lane = (int)threadIdx.x & 63, r = lane & 15;
    const int smp = tile * kTile + 16 * wv + r;
    uint32_t rs, rn;
    sample_rows(g, smp, rs, rn);
    prow_load(T.R, g.obs + (size_t)rn * kPackedDwords);
    if (g.meta) {
        const uint4 m = g.meta[rs];
        T.rew = __uint_as_float(m.z);
        T.nd = 1.0f - (float)(m.w & 0xffu);
    } else {
        T.rew = g.reward[rs];
        T.nd = 1.0f - (float)g.done[rs];
    }
    T.e0 = g.eps[2 * smp];
    T.e1 = g.eps[2 * smp + 1];
}

// ---------------------------------------------------------------------------------------------------------------------
// phase A, stage I as a launch of its own (round 4): the td targets of the whole batch into SacArgs.td (global, [batch][2]).
// Stage I keeps three nets resident (118 KB) but needs no per-tile scratch, so TWO tiles can be in flight per workgroup: eight
// wavefronts -- waves 0-3 take the even tiles of the workgroup's range, waves 4-7 the odd ones -- share the staged nets, two
// wavefronts per SIMD, and an f32 MFMA of one overlaps the VALU work (actor head, layer 2, the row decode) of the other.  Same
// device functions on the same operands as the fused stage I of k_sac_critic_grad: bit-identical td targets.
// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(512) k_sac_td(SacArgsN slots)
{
    const SacArgs &g = slots.s[blockIdx.y];
    if ((int)blockIdx.x >= g.grid) return;
    extern __shared__ __align__(16) float lds[];
    const int tid = (int)threadIdx.x, lane = tid & 63, wv8 = __builtin_amdgcn_readfirstlane(tid >> 6), wv = wv8 & 3, half = wv8 >> 2,
              r = lane & 15, gq = lane >> 4;
    const int n_tiles = g.batch / kTile;
    const int t0 = (int)blockIdx.x * g.tiles_per_wg;
    const int nt = n_tiles - t0 < g.tiles_per_wg ? n_tiles - t0 : g.tiles_per_wg;
    const float alpha = expf(*g.log_alpha);
    // LDS: actor layer 1 (split form) | target 1: layer 1 (split), fc2, b2 | target 2: the same
    const W1Split Wa = w1split_at(lds);
    const TdSet S1 = tdset_at(lds + kSplitF), S2 = tdset_at(lds + kSplitF + kTdSetF);
    TileInS T;
    if (half < nt) tile_in_s(g, t0 + half, wv, T);
    if (half == 0) {                                   // waves 0-3 stage (the helpers are written for 256 threads)
        CritRegs C1, C2;
        critic_issue(C1, g.t1);
        critic_issue(C2, g.t2);
        stage_actor_split(Wa, g.actor);
        critic_commit_split<kSacW2Half>(S1, C1);
        critic_commit_split<kSacW2Half>(S2, C2);
    }
    __syncthreads();
    for (int j = half; j < nt; j += 2) {
        TileInS Tn = T;
        if (j + 2 < nt) tile_in_s(g, t0 + j + 2, wv, Tn);
        floatx4 acc[4], acc2[4];
        fwd_strip_split<false>(Wa, T.R, acc);
        float o[4];
        {
            W2Frag<4> Fa;                          // (re-read per tile, like the critics' heads below: registers)
            w2_load<4>(Fa, g.actor + kAoW2, g.actor + kAob2, 4);
            q_strip<4>(acc, Fa, 4, 4, 0, o);
        }
        const int d = gq & 1;                      // this lane group's action dimension (the critics' layer 1 takes a0 from the
        const ActorOne A = actor_head_one(d ? o[1] : o[0], d ? o[3] : o[2], d ? T.e1 : T.e0);      // lanes of group 0, a1 from group 1)
        const float a0 = A.act * g.bound, a1 = a0;
        float q1[2], q2[2];
        {   // (the two heads' fragments are re-read per use: 64 registers that two wavefronts per SIMD do not have; L1 hits)
            W2Frag<2> Fo;
            w2_load<2>(Fo, g.t1 + kCoWo, g.t1 + kCobo, 2);
            critic_fwd_split(S1, T.R, a0, a1, Fo, acc, acc2, q1, kSacW2Half ? S1.W2s : nullptr);
            w2_load<2>(Fo, g.t2 + kCoWo, g.t2 + kCobo, 2);
            critic_fwd_split(S2, T.R, a0, a1, Fo, acc, acc2, q2, kSacW2Half ? S2.W2s : nullptr);
        }
        if (gq < 2)                                // group d writes component d
            g.td[((size_t)(t0 + j) * kTile + 16 * wv + r) * 2 + d] =
                T.rew + g.gamma * (fminf(d ? q1[1] : q1[0], d ? q2[1] : q2[0]) + alpha * (-A.lp)) * T.nd;
        T = Tn;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// phase A: the critics
// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_sac_critic_grad(SacArgsN slots)
{
    const SacArgs &g = slots.s[blockIdx.y];
    if ((int)blockIdx.x >= g.grid) return;
    extern __shared__ __align__(16) float lds[];
    const int tid = (int)threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 15, gq = lane >> 4;
    const int n_tiles = g.batch / kTile;
    const int t0 = (int)blockIdx.x * g.tiles_per_wg;
    const int nt = n_tiles - t0 < g.tiles_per_wg ? n_tiles - t0 : g.tiles_per_wg;
    const float alpha = expf(*g.log_alpha);
    const float inv_b = 1.0f / (float)g.batch;
    float *tds = lds + kCritTdOff;                     // [tile][sample][2]
    float *q1s = tds + kTMax * kTile * 2;              // [tile][sample]: critic 1's Q(s, a)[0] (for |TD| of prioritised replay)
    S_STAMP(0);

    // ---- stage I: td target = r + gamma (min(Q_t1, Q_t2)(s', a') - alpha log pi(a' | s')) (1 - done)      (:122-131)
    // (g.td: k_sac_td has already left the td targets of the whole batch in global memory -- stage I is skipped)
    if (!g.td) {
        float *Wa = lds;
        const WSet S1 = wset_at(lds + kTileF), S2 = wset_at(lds + kTileF + kWSetF);
        TileIn T;
        tile_in<true, true>(g, t0, T);                 // in flight under the staging
        {
            CritRegs C1, C2;
            critic_issue(C1, g.t1);
            critic_issue(C2, g.t2);
            stage_actor(Wa, g.actor);
            critic_commit(S1, C1);
            critic_commit(S2, C2);
        }
        W2Frag<4> Fa;
        w2_load<4>(Fa, g.actor + kAoW2, g.actor + kAob2, 4);
        W2Frag<2> Fo1, Fo2;
        w2_load<2>(Fo1, g.t1 + kCoWo, g.t1 + kCobo, 2);
        w2_load<2>(Fo2, g.t2 + kCoWo, g.t2 + kCobo, 2);
        __syncthreads();
        S_STAMP(1);
        for (int j = 0; j < nt; ++j) {
            TileIn Tn = T;
            if (j + 1 < nt) tile_in<true, true>(g, t0 + j + 1, Tn);
            floatx4 acc[4], acc2[4];
            fwd_strip_packed(Wa, T.R, acc);
            float o[4];
            q_strip<4>(acc, Fa, 4, 4, 0, o);
            ActorOut A;
            actor_head(o, T.e0, T.e1, A);
            const float a0 = A.act[0] * g.bound, a1 = A.act[1] * g.bound;
            float q1[2], q2[2];
            critic_fwd(S1, T.R, a0, a1, Fo1, acc, acc2, q1);
            critic_fwd(S2, T.R, a0, a1, Fo2, acc, acc2, q2);
            if (gq == 0) {
#pragma unroll
                for (int d = 0; d < 2; ++d)
                    tds[(j * kTile + 16 * wv + r) * 2 + d] = T.rew + g.gamma * (fminf(q1[d], q2[d]) + alpha * (-A.lp[d])) * T.nd;
            }
            T = Tn;
        }
    }
    __syncthreads();
    S_STAMP(2);
    // ---- stage II: Q1 / Q2 (s, a): loss, backward, weight gradients
    // (the critic's layer 1 in the split form: its forward runs at f32 accuracy on the f16 matrix pipe, "Layer 1 at f32 accuracy")
    const TdSet SS = tdset_at(lds);
    const WSet S = WSet{nullptr, SS.W2s, SS.b2s};
    float *Xs = lds + kTdSetF;
    float *H1s = Xs + kTileF, *H2s = H1s + kTile * kLh, *dH1s = H2s + kTile * kLh, *dH2s = dH1s + kTile * kLh;
    float *dqs = dH2s + kTile * kLh, *red = dqs + kTile * 4;
    for (int c = 0; c < 2; ++c) {
        const float *flat = c ? g.c2 : g.c1;
        TileIn T;
        tile_in<false, true>(g, t0, T);
        // (split dW1: the f32 X tile is gone, the packed rows take 1 792 of its 6 912 floats -- fc2's f16 image of layer2_fwd_h sits
        // behind them; the backward keeps the f32 fc2 of SS)
        float *w2h = (kSacSplitDw1 && kSacW2Half) ? Xs + kTile * kPsLd : nullptr;
        {
            CritRegs C;
            critic_issue(C, flat);
            critic_commit_split(SS, C);
            if (w2h) layer2_commit_h(w2half_at(w2h), C);
        }
        W2Frag<2> Fo;
        w2_load<2>(Fo, flat + kCoWo, flat + kCobo, 2);
        __syncthreads();
        S_STAMP(3 + 6 * c);
        floatx4 aw1[7], aw2[4], awo = floatx4{0.0f, 0.0f, 0.0f, 0.0f}, ab2 = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
        Dw1Split ws1;
        dw1_zero(ws1);
#pragma unroll
        for (int u = 0; u < 7; ++u) aw1[u] = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int u = 0; u < 4; ++u) aw2[u] = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
        float s_bo0 = 0.0f, s_bo1 = 0.0f, s_loss = 0.0f, s_cnt = 0.0f;
        for (int j = 0; j < nt; ++j) {
            TileIn Tn = T;
            if (j + 1 < nt) tile_in<false, true>(g, t0 + j + 1, Tn);
            const float *td = g.td ? g.td + ((size_t)(t0 + j) * kTile + 16 * wv + r) * 2 : tds + (j * kTile + 16 * wv + r) * 2;
            floatx4 acc1[4], acc2[4];
            float q[2];
            critic_fwd_split<true>(SS, T.R, T.a0, T.a1, Fo, acc1, acc2, q, w2h);
            const float e0 = q[0] - td[0], e1 = q[1] - td[1];
            const float ww = T.w * T.isw;                                // validity x importance-sampling weight (1 without PER)
            const float dq0 = ww * e0 * inv_b, dq1 = ww * e1 * inv_b;    // d mean_{[B,2]}(w err^2) / dq = 2 w err / (2 B)
            if (gq == 0) { s_loss += ww * (e0 * e0 + e1 * e1); s_bo0 += dq0; s_bo1 += dq1; s_cnt += T.w; }
            if (g.abs_td && gq == 0) {                                   // :351 |min(Q1, Q2) - td_target|, column 0
                if (c == 0) q1s[j * kTile + 16 * wv + r] = q[0];
                else g.abs_td[(size_t)(t0 + j) * kTile + 16 * wv + r] = fabsf(fminf(q1s[j * kTile + 16 * wv + r], q[0]) - td[0]);
            }
            floatx4 dh1[4], dh2[4], h1[4], h2[4];
            critic_bwd(S, Fo, acc1, acc2, dq0, dq1, dh1, dh2);
            relu4(acc1, h1);
            relu4(acc2, h2);
            h_strip_store(H1s, h1);
            h_strip_store(H2s, h2);
            h_strip_store(dH1s, dh1);
            h_strip_store(dH2s, dh2);
            if (kSacSplitDw1) {
                dh_strip_amax(dh1, red + 48);
                if (gq == 0) ps_store(reinterpret_cast<uint32_t *>(Xs), 16 * wv + r, T.R, T.a0, T.a1);
            } else {
                x_strip_store(Xs, T.R, T.a0, T.a1);
            }
            if (gq == 0) { dqs[(16 * wv + r) * 4] = dq0; dqs[(16 * wv + r) * 4 + 1] = dq1; }
            __syncthreads();
            if (j == 0) S_STAMP(4 + 6 * c);
            if (kSacSplitDw1) wgrad_x_split<true>(dH1s, reinterpret_cast<const uint32_t *>(Xs), red + 48, ws1);
            else wgrad_xf(dH1s, Xs, aw1);              // dW1 (+ db1 as column 100)
            if (j == 0) S_STAMP(5 + 6 * c);
            wgrad_critic_rest(dH2s, H1s, H2s, dqs, aw2, awo, ab2);     // dW2, dWout^T, db2
            __syncthreads();
            if (j == 0) S_STAMP(6 + 6 * c);
            T = Tn;
        }
        S_STAMP(7 + 6 * c);
        float *out = g.partials + (size_t)blockIdx.x * kStrideC + c * kPc;
        if (kSacSplitDw1) store_dw1_split<true>(out, kIn, kCob1, ws1);
        else store_dw1(out, kIn, kCob1, aw1);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) out[kCoW2 + (16 * wv + 4 * gq + reg) * kHid + 16 * u + r] = aw2[u][reg];
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            if (r < 2) out[kCoWo + r * kHid + 16 * wv + 4 * gq + reg] = awo[reg];
            if (r == 0) out[kCob2 + 16 * wv + 4 * gq + reg] = ab2[reg];
        }
        s_bo0 = wave_sum(s_bo0); s_bo1 = wave_sum(s_bo1); s_loss = wave_sum(s_loss); s_cnt = wave_sum(s_cnt);
        if (lane == 0) { red[wv * 4] = s_bo0; red[wv * 4 + 1] = s_bo1; red[wv * 4 + 2] = s_loss; red[wv * 4 + 3] = s_cnt; }
        __syncthreads();
        if (tid < 4) {
            const float s = (red[tid] + red[4 + tid]) + (red[8 + tid] + red[12 + tid]);
            if (tid < 2) out[kCobo + tid] = s;
            else if (tid == 2) g.partials[(size_t)blockIdx.x * kStrideC + 2 * kPc + c] = s * 0.5f * inv_b;      // mean over [B, 2]
            else if (c == 0) {
                // the valid FRACTION of the batch this workgroup saw (column sum = sum of valid / B, summed over the ranks
                // after an all-reduce): the Adam kernel divides every column by it -> means over the valid samples
                g.partials[(size_t)blockIdx.x * kStrideC + 2 * kPc + 2] = s * inv_b;
                g.partials[(size_t)blockIdx.x * kStrideC + 2 * kPc + 3] = 0.0f;
            }
        }
        __syncthreads();
        S_STAMP(8 + 6 * c);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// phase B: the actor (critics already updated).  Actor fc1 and both critics stay in LDS; one pass over the tiles.
// ---------------------------------------------------------------------------------------------------------------------
// Round 4: all three layer-1 forwards (the actor's, both critics') in the split form -- with 88-halves term rows a staged layer 1
// is exactly as large as the f32 tile it replaces, so the phase keeps its LDS map.  (An earlier form with 104-halves rows needed
// 170 KB; the two-pass variant built around that limit passed every test and was no faster -- the matrix cycles it saved were
// what its second pass over the rows cost.)
__global__ void __launch_bounds__(256) k_sac_actor_grad(SacArgsN slots)
{
    const SacArgs &g = slots.s[blockIdx.y];
    if ((int)blockIdx.x >= g.grid) return;
    extern __shared__ __align__(16) float lds[];
    const int tid = (int)threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 15, gq = lane >> 4;
    const int n_tiles = g.batch / kTile;
    const int t0 = (int)blockIdx.x * g.tiles_per_wg;
    const int nt = n_tiles - t0 < g.tiles_per_wg ? n_tiles - t0 : g.tiles_per_wg;
    const float alpha = expf(*g.log_alpha);
    const float inv_2b = 0.5f / (float)g.batch;
    const float g_lp = alpha * inv_2b;                                 // d loss / d log pi per element
    const W1Split Wa = w1split_at(lds);
    const TdSet S1 = tdset_at(lds + kSplitF), S2 = tdset_at(lds + kSplitF + kTdSetF);
    const WSet L1 = WSet{nullptr, S1.W2s, S1.b2s}, L2 = WSet{nullptr, S2.W2s, S2.b2s};      // (the 64 x 64 layers, for the backward)
    uint32_t *Ps = reinterpret_cast<uint32_t *>(lds + kSplitF + 2 * kTdSetF);
    float *H1s = lds + kSplitF + 2 * kTdSetF + kTile * kPsLd, *dH1s = H1s + kTile * kLh, *dqs = dH1s + kTile * kLh, *red = dqs + kTile * 4;
    TileIn T;
    tile_in<false, false>(g, t0, T);
    {
        CritRegs C1, C2;
        critic_issue(C1, g.c1);
        critic_issue(C2, g.c2);
        stage_actor_split(Wa, g.actor);
        critic_commit_split(S1, C1);
        critic_commit_split(S2, C2);
    }
    W2Frag<4> Fa;
    w2_load<4>(Fa, g.actor + kAoW2, g.actor + kAob2, 4);
    W2Frag<2> Fo1, Fo2;
    w2_load<2>(Fo1, g.c1 + kCoWo, g.c1 + kCobo, 2);
    w2_load<2>(Fo2, g.c2 + kCoWo, g.c2 + kCobo, 2);
    __syncthreads();
    floatx4 aw1[7], awo = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
    Dw1Split ws1;
    dw1_zero(ws1);
#pragma unroll
    for (int u = 0; u < 7; ++u) aw1[u] = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
    float s_b[4] = {0.0f, 0.0f, 0.0f, 0.0f}, s_lp = 0.0f, s_loss = 0.0f, s_cnt = 0.0f;
    for (int j = 0; j < nt; ++j) {
        TileIn Tn = T;
        if (j + 1 < nt) tile_in<false, false>(g, t0 + j + 1, Tn);
        const float w_lp = g_lp * T.w, w_2b = inv_2b * T.w;               // this sample's weight in the two loss terms
        // a~, log pi(a~ | s)
        floatx4 acc[4];
        fwd_strip_split<false>(Wa, T.R, acc);       // (the all-operands-first form spills here: 0.529 against 0.522 ms per configs[3] pass)
        float o[4];
        q_strip<4>(acc, Fa, 4, 4, 0, o);
        // The four lane groups of a sample hold the same head outputs: group g evaluates action dimension g & 1 only (half the
        // transcendental work of the full head) and the groups exchange what the others need -- group_sum4 over values that are
        // zero in the groups of the other dimension is twice the value (two groups per dimension), so x 0.5 is exact.
        const int hd = gq & 1;
        const float hm = hd ? o[1] : o[0], hs = hd ? o[3] : o[2], he = hd ? T.e1 : T.e0;
        const float h_mu = tanhf(hm);
        const float h_sp = hs > 20.0f ? hs : log1pf(expf(hs));
        const float h_sd = tanhf(h_sp);
        const float h_ns = h_mu + h_sd * he;
        const float h_df = h_ns - h_mu;
        float h_lp = -(h_df * h_df) / (2.0f * (h_sd * h_sd)) - logf(h_sd) - 0.9189385332046727f;
        const float h_act = tanhf(h_ns);
        const float h_th = tanhf(h_act);
        const float h_u = 1.0f - h_th * h_th + 1e-7f;
        h_lp -= logf(h_u);
        const float a0 = 0.5f * group_sum4(hd ? 0.0f : h_act) * g.bound, a1 = 0.5f * group_sum4(hd ? h_act : 0.0f) * g.bound;
        const float lp_both = 0.5f * group_sum4(h_lp);                      // log pi of both dimensions, in every lane
        // Q1, Q2 (s, a~): the minimum picks, per (sample, output), the critic that dL/dq = -1 / (2B) flows into (ties: critic 1)
        floatx4 c1a[4], c1b[4], c2a[4], c2b[4];
        float q1[2], q2[2];
        critic_fwd_split<false>(S1, T.R, a0, a1, Fo1, c1a, c1b, q1);
        critic_fwd_split<false>(S2, T.R, a0, a1, Fo2, c2a, c2b, q2);
        const bool m0 = q2[0] < q1[0], m1 = q2[1] < q1[1];
        if (gq == 0) {
            s_lp += T.w * lp_both;
            s_loss += T.w * (alpha * lp_both - ((m0 ? q2[0] : q1[0]) + (m1 ? q2[1] : q1[1])));       // :364-365
            s_cnt += T.w;
        }
        float da0 = 0.0f, da1 = 0.0f;
        {
            floatx4 dh1[4], dh2[4];
            critic_bwd(L2, Fo2, c2a, c2b, m0 ? -w_2b : 0.0f, m1 ? -w_2b : 0.0f, dh1, dh2);
            action_grad_part_split(S2, dh1, da0, da1);
            critic_bwd(L1, Fo1, c1a, c1b, m0 ? 0.0f : -w_2b, m1 ? 0.0f : -w_2b, dh1, dh2);
            action_grad_part_split(S1, dh1, da0, da1);
        }
        da0 = group_sum4(da0);
        da1 = group_sum4(da1);
        // the actor's backward
        float dout[4];
        {   // this lane group's dimension, then all four head gradients into every lane
            const float dav = hd ? da1 : da0;
            const float dact = w_lp * (2.0f * h_th * (1.0f - h_th * h_th) / h_u) + g.bound * dav;
            const float dns = dact * (1.0f - h_act * h_act);
            const float dsd = dns * he - w_lp / h_sd;
            const float sig = hs > 20.0f ? 1.0f : 1.0f / (1.0f + expf(-hs));
            const float d_mu = dns * (1.0f - h_mu * h_mu);                    // fc_mu pre-activation
            const float d_sd = dsd * (1.0f - h_sd * h_sd) * sig;              // fc_std pre-activation
            dout[0] = 0.5f * group_sum4(hd ? 0.0f : d_mu);
            dout[1] = 0.5f * group_sum4(hd ? d_mu : 0.0f);
            dout[2] = 0.5f * group_sum4(hd ? 0.0f : d_sd);
            dout[3] = 0.5f * group_sum4(hd ? d_sd : 0.0f);
        }
        floatx4 h[4], dh[4];
        relu4(acc, h);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const float v = fmaf(Fa.w[3][t][reg], dout[3], fmaf(Fa.w[2][t][reg], dout[2],
                                fmaf(Fa.w[1][t][reg], dout[1], Fa.w[0][t][reg] * dout[0])));
                dh[t][reg] = acc[t][reg] > 0.0f ? v : 0.0f;
            }
        h_strip_store(H1s, h);
        h_strip_store(dH1s, dh);
        if (kSacSplitDw1) dh_strip_amax(dh, red + 48);
        if (gq == 0) {
            ps_store(Ps, 16 * wv + r, T.R, 0.0f, 0.0f);
#pragma unroll
            for (int a = 0; a < 4; ++a) { dqs[(16 * wv + r) * 4 + a] = dout[a]; s_b[a] += dout[a]; }
        }
        __syncthreads();
        if (kSacSplitDw1) wgrad_x_split<false>(dH1s, Ps, red + 48, ws1);
        else wgrad_x<false>(dH1s, Ps, aw1);
        wgrad_small(H1s, dqs, 4, awo);
        __syncthreads();
        T = Tn;
    }
    float *out = g.partials + (size_t)blockIdx.x * kStrideA;
    if (kSacSplitDw1) store_dw1_split<false>(out, kW, kHid * kW, ws1);
    else store_dw1(out, kW, kHid * kW, aw1);
#pragma unroll
    for (int reg = 0; reg < 4; ++reg)
        if (r < 4) out[kAoW2 + r * kHid + 16 * wv + 4 * gq + reg] = awo[reg];
#pragma unroll
    for (int a = 0; a < 4; ++a) s_b[a] = wave_sum(s_b[a]);
    s_lp = wave_sum(s_lp);
    s_loss = wave_sum(s_loss);
    s_cnt = wave_sum(s_cnt);
    if (lane == 0) {
#pragma unroll
        for (int a = 0; a < 4; ++a) red[wv * 8 + a] = s_b[a];
        red[wv * 8 + 4] = s_loss;
        red[wv * 8 + 5] = s_lp;
        red[wv * 8 + 6] = s_cnt;
    }
    __syncthreads();
    if (tid < 7) {
        const float s = (red[tid] + red[8 + tid]) + (red[16 + tid] + red[24 + tid]);
        if (tid < 4) out[kAob2 + tid] = s;
        else if (tid == 4) out[kPa] = s * inv_2b;
        else if (tid == 5) out[kPa + 1] = s;
        else out[kPa + 2] = s * 2.0f * inv_2b;               // valid fraction of the batch (see k_sac_critic_grad)
    }
    if (tid == 7) out[kPa + 3] = 0.0f;
}

// ---------------------------------------------------------------------------------------------------------------------
// get_action (SAC_Trainer.py:444-448): a = actor(s) for `count` agents whose packed rows are first + i * stride; the two
// action components go to act0[row] / act1[row] (the planes the env step and the learner read)
// ---------------------------------------------------------------------------------------------------------------------
struct SacActArgs {
    const float *actor;
    const uint32_t *obs;
    int first, stride, count;
    const float *eps;
    float bound;
    float *act0, *act1;
};
struct SacActArgsN {
    SacActArgs s[kSlots];
};
constexpr size_t kSacActLds = (size_t)kSplitBytes;

__global__ void __launch_bounds__(256) k_sac_act(SacActArgsN slots)
{
    const SacActArgs &g = slots.s[blockIdx.y];
    extern __shared__ __align__(16) float lds[];
    const W1Split W1s = w1split_at(lds);
    const int lane = (int)threadIdx.x & 63, wv = (int)threadIdx.x >> 6, r = lane & 15, gq = lane >> 4;
    stage_actor_split(W1s, g.actor);
    __syncthreads();
    W2Frag<4> Fa;
    w2_load<4>(Fa, g.actor + kAoW2, g.actor + kAob2, 4);
    const int n_tiles = (g.count + kTile - 1) / kTile;
    // A workgroup walks several tiles (the launch is sized to two workgroups per CU: round 3 gave every 64-agent tile a workgroup
    // of its own, and staging fc1 -- 25.6 KB per workgroup -- was most of its 27 us at 4 x 32 768 agents); the next tile's row and
    // draws are requested before this tile is computed.
    auto fetch = [&](int tile, PRow &R, float &e0, float &e1, size_t &row, int &i) {
        i = tile * kTile + 16 * wv + r;
        const int ii = i < g.count ? i : g.count - 1;
        row = (size_t)g.first + (size_t)ii * g.stride;
        prow_load(R, g.obs + row * kPackedDwords);
        e0 = g.eps[2 * ii + (gq & 1)];          // lane group g evaluates action dimension g & 1
        e1 = 0.0f;
    };
    int tile = (int)blockIdx.x;
    if (tile >= n_tiles) return;
    PRow R;
    float e0, e1;
    size_t row;
    int i;
    fetch(tile, R, e0, e1, row, i);
    for (; tile < n_tiles; tile += (int)gridDim.x) {
        PRow Rn = R;
        float e0n = e0, e1n = e1;
        size_t rown = row;
        int in = i;
        if (tile + (int)gridDim.x < n_tiles) fetch(tile + (int)gridDim.x, Rn, e0n, e1n, rown, in);
        floatx4 acc[4];
        fwd_strip_split<false>(W1s, R, acc);
        float o[4];
        q_strip<4>(acc, Fa, 4, 4, 0, o);
        const int d = gq & 1;
        const ActorOne A = actor_head_one(d ? o[1] : o[0], d ? o[3] : o[2], e0);
        if (gq < 2 && i < g.count) (d ? g.act1 : g.act0)[row] = A.act * g.bound;
        R = Rn; e0 = e0n; e1 = e1n; row = rown; i = in;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// column sums of the partial rows + torch.optim.Adam (+ soft target update, + the log_alpha step)
// ---------------------------------------------------------------------------------------------------------------------
struct AdamSeg {
    float *p, *m, *v, *tgt;      // tgt nullable: tgt = tgt (1 - tau) + p tau after the step (soft_update, :145-147)
    int n;
    float lr;
};
struct AdamArgs {
    const float *partials;
    int rows, stride, nseg, extras;          // columns: seg 0 | seg 1 | `extras` scalars
    AdamSeg seg[2];
    float beta1, beta2, eps, bc1, bc2_sqrt, tau;
    float *scalars_out;                      // [extras] column sums of the extras (losses, sum of log pi)
    float *raw_out;                          // non-null: only write the column sums there (multi-GPU: all-reduce them, then
                                             // run the kernel again on that one row) -- no parameter is touched
    float grad_scale;                        // gradient = column sum * grad_scale (1 / world size after an all-reduce SUM)
    // log_alpha step (actor phase): extras column 1 holds sum log pi over [B, 2]
    float *log_alpha, *alpha_mv;             // nullable; alpha_mv = {exp_avg, exp_avg_sq}
    float alpha_lr, target_entropy, inv_2b;
    int frac_col;                            // column holding the valid fraction of the batch (-1: scale by grad_scale instead)
    int n_loss;                              // the first n_loss extras are loss means
    const uint32_t *skip;                    // nullable: a non-zero word makes the launch a no-op (UavSacAdam.skip_word)
    const uint32_t *go;                      // nullable: the launch is a no-op unless *go == go_value (UavSacAdam.go_word)
    uint32_t go_value;
};

__device__ __forceinline__ float adam_step(float p, float g, float &m, float &v, const AdamArgs &a, float lr)
{
    m = m + (g - m) * (1.0f - a.beta1);                           // exp_avg.lerp_(grad, 1 - beta1)
    v = v * a.beta2 + (1.0f - a.beta2) * g * g;
    return p - (lr / a.bc1) * (m / (sqrtf(v) / a.bc2_sqrt + a.eps));
}

// 64 columns per workgroup, four row groups (wavefront w sums rows w, w + 4, ...: 8 loads in flight per thread), combined
// through LDS in a fixed order -- deterministic.
struct AdamArgsN {
    AdamArgs s[kSlots];
};

__global__ void __launch_bounds__(256) k_sac_reduce_adam(AdamArgsN slots)
{
    const AdamArgs &a = slots.s[blockIdx.y];
    __shared__ float part[4][64];
    __shared__ float fpart[4];
    const int cl = (int)threadIdx.x & 63, rg = (int)threadIdx.x >> 6;
    const int col = (int)blockIdx.x * 64 + cl;
    const int total = a.seg[0].n + (a.nseg > 1 ? a.seg[1].n : 0) + a.extras;
    const float *src = a.partials + (col < total ? col : total - 1);
    const float *fsrc = a.partials + (a.frac_col >= 0 ? a.frac_col : 0);      // the valid-fraction column (every lane: same address)
    float s = 0.0f, f = 0.0f;
    int b = rg;
    for (; b + 28 < a.rows; b += 32) {
        float t[8], u[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { t[k] = src[(size_t)(b + 4 * k) * a.stride]; u[k] = fsrc[(size_t)(b + 4 * k) * a.stride]; }
#pragma unroll
        for (int k = 0; k < 8; ++k) { s += t[k]; f += u[k]; }
    }
    for (; b < a.rows; b += 4) { s += src[(size_t)b * a.stride]; f += fsrc[(size_t)b * a.stride]; }
    part[rg][cl] = s;
    if (cl == 0) fpart[rg] = f;
    __syncthreads();
    if (rg != 0 || col >= total) return;
    s = (part[0][cl] + part[1][cl]) + (part[2][cl] + part[3][cl]);
    if (a.raw_out) {
        a.raw_out[col] = s;
        return;
    }
    // behind a peer exchange that raised its sticky error the row holds rank-local sums: step nothing (the rank's
    // parameters, moments, targets and log_alpha freeze until the caller re-synchronises them; csrc/p2p.hip)
    if (a.skip && __hip_atomic_load(a.skip, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
    // ... and behind a step that moved no agent there is no update at all (the reference has left its episode loop)
    if (a.go && __hip_atomic_load(a.go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != a.go_value) return;
    // every column was accumulated as sum_i w_i (...) / B per rank; frac = sum of the ranks' (valid / B): dividing by it
    // gives the mean over the valid samples of all ranks (all valid: 1 / world size)
    const float frac = (fpart[0] + fpart[1]) + (fpart[2] + fpart[3]);
    const float norm = a.frac_col >= 0 ? (frac > 0.0f ? 1.0f / frac : 0.0f) : a.grad_scale;
    int c = col;
    for (int k = 0; k < a.nseg; ++k) {
        const AdamSeg &sg = a.seg[k];
        if (c < sg.n) {
            float m = sg.m[c], v = sg.v[c];
            const float np = adam_step(sg.p[c], s * norm, m, v, a, sg.lr);
            sg.m[c] = m; sg.v[c] = v; sg.p[c] = np;
            if (sg.tgt) sg.tgt[c] = sg.tgt[c] * (1.0f - a.tau) + np * a.tau;
            return;
        }
        c -= sg.n;
    }
    // extras: losses (means: normalised like the gradients), then raw sums (sum log pi; the valid fraction)
    if (a.scalars_out) a.scalars_out[c] = c < a.n_loss ? s * norm : s;
    if (c == 1 && a.log_alpha) {
        // alpha_loss = mean((entropy - target_entropy).detach() * exp(log_alpha))   (:372-375), over the valid samples
        const float la = *a.log_alpha;
        const float gl = expf(la) * (-s * a.inv_2b * norm - a.target_entropy);
        float m = a.alpha_mv[0], v = a.alpha_mv[1];
        const float nla = adam_step(la, gl, m, v, a, a.alpha_lr);
        a.alpha_mv[0] = m; a.alpha_mv[1] = v;
        *a.log_alpha = nla;
    }
}

bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

int fill_args(const UavSacNets *n, const UavSacBatch *b, float *partials, SacArgs &g, int &grid, int n_slots)
{
    if (!n || !b || !partials) return sac_fail(UAVENV_EINVAL, "uavenv_sac: null argument");
    if (!n->actor || !n->critic1 || !n->critic2 || !n->target1 || !n->target2 || !n->log_alpha)
        return sac_fail(UAVENV_EINVAL, "uavenv_sac: null parameter block");
    if (!aligned16(n->actor) || !aligned16(n->critic1) || !aligned16(n->critic2) || !aligned16(n->target1) || !aligned16(n->target2) ||
        !aligned16(partials) || !aligned16(b->obs_packed))
        return sac_fail(UAVENV_EINVAL, "uavenv_sac: parameter blocks, partial rows and the observation rows must be 16-byte aligned");
    if (b->batch <= 0 || b->batch % kTile) return sac_fail(UAVENV_EINVAL, "uavenv_sac: batch must be a positive multiple of 64");
    if (!b->obs_packed || !b->eps || (!b->meta && (!b->act0 || !b->act1 || !b->reward || !b->done)))
        return sac_fail(UAVENV_EINVAL, "uavenv_sac: null batch plane");
    if (!aligned16(b->meta)) return sac_fail(UAVENV_EINVAL, "uavenv_sac: the transition records are 16-byte aligned");
    if (!b->draws && (!b->idx_s || !b->idx_n)) return sac_fail(UAVENV_EINVAL, "uavenv_sac: neither draws nor row indices");
    if (b->draws && (b->n_agents <= 0 || b->uav_per_env <= 0 || b->slot < 0 || b->slot >= b->uav_per_env || b->frames < 2))
        return sac_fail(UAVENV_EINVAL, "uavenv_sac: draws need n_agents / uav_per_env / slot / frames");
    g.obs = reinterpret_cast<const uint32_t *>(b->obs_packed);
    g.idx_s = b->idx_s; g.idx_n = b->idx_n; g.draws = b->draws;
    g.n_agents = b->n_agents; g.uav = b->uav_per_env; g.slot = b->slot; g.frames = b->frames;
    g.act0 = b->act0; g.act1 = b->act1; g.reward = b->reward; g.done = b->done; g.valid = b->valid;
    g.meta = reinterpret_cast<const uint4 *>(b->meta);
    g.eps = b->eps;
    g.is_w = b->is_weights;
    g.abs_td = b->abs_td_out;
    g.td = b->td_scratch;
    g.batch = b->batch;
    const int n_tiles = b->batch / kTile;
    const int tpw = tiles_per_wg_of(n_tiles, n_slots, b->tiles_per_wg);
    g.tiles_per_wg = tpw;
    grid = (n_tiles + tpw - 1) / tpw;
    g.grid = grid;
    g.actor = n->actor; g.c1 = n->critic1; g.c2 = n->critic2; g.t1 = n->target1; g.t2 = n->target2; g.log_alpha = n->log_alpha;
    g.partials = partials;
    g.dbg = g_sac_dbg;
    return UAVENV_OK;
}

template <typename K>
int launch_phase(K kernel, bool &attr, size_t lds, const SacArgsN &slots, int n, int grid, hipStream_t s)
{
    if (!attr) {                         // (once per kernel; one process = one device for this library's learners)
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
            hipSuccess)
            return sac_fail(UAVENV_EHIP, "uavenv_sac: cannot raise the dynamic LDS limit");
        attr = true;
    }
    hipLaunchKernelGGL(kernel, dim3(grid, n), dim3(256), lds, s, slots);
    return hipGetLastError() == hipSuccess ? UAVENV_OK : sac_fail(UAVENV_EHIP, "uavenv_sac: launch failed");
}

// the two grad phases for n slots in one launch (n = 1: the single-trainer entry points)
int grad_phase(bool critic, const UavSacNets *nets, const UavSacBatch *batches, int n, float gamma, float action_bound,
               float *const *partials, hipStream_t s)
{
    if (n < 1 || n > kSlots || !nets || !batches || !partials) return sac_fail(UAVENV_EINVAL, "uavenv_sac: 1 .. 8 slots per launch");
    SacArgsN slots;
    memset(&slots, 0, sizeof(slots));
    int grid = 0;
    for (int j = 0; j < n; ++j) {
        int gj = 0;
        const int rc = fill_args(&nets[j], &batches[j], partials[j], slots.s[j], gj, n);
        if (rc != UAVENV_OK) return rc;
        slots.s[j].gamma = critic ? gamma : 0.0f;
        slots.s[j].bound = action_bound;
        grid = gj > grid ? gj : grid;
    }
    static bool attr_c = false, attr_a = false, attr_t = false;
    if (critic) {
        // stage I as a launch of its own (two tiles in flight per workgroup) when EVERY slot brought a td scratch; otherwise the
        // fused kernel computes the td targets itself (bit-identical either way: the same device functions on the same operands)
        bool split = getenv("UAVENV_SAC_FUSED_TD") == nullptr;
        for (int j = 0; j < n; ++j) split = split && slots.s[j].td != nullptr;
        if (split) {
            if (!attr_t) {
                if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_sac_td), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)kSacTdLds) != hipSuccess)
                    return sac_fail(UAVENV_EHIP, "uavenv_sac: cannot raise the dynamic LDS limit");
                attr_t = true;
            }
            hipLaunchKernelGGL(k_sac_td, dim3(grid, n), dim3(512), kSacTdLds, s, slots);
            if (hipGetLastError() != hipSuccess) return sac_fail(UAVENV_EHIP, "uavenv_sac: launch failed");
        } else {
            for (int j = 0; j < n; ++j) slots.s[j].td = nullptr;
        }
    }
    return critic ? launch_phase(k_sac_critic_grad, attr_c, kSacCriticLds, slots, n, grid, s)
                  : launch_phase(k_sac_actor_grad, attr_a, kSacActorLds, slots, n, grid, s);
}

int adam_launch(const AdamArgs *args, int n, int total, hipStream_t s, const char *what)
{
    AdamArgsN slots;
    memset(&slots, 0, sizeof(slots));
    for (int j = 0; j < n; ++j) slots.s[j] = args[j];
    hipLaunchKernelGGL(k_sac_reduce_adam, dim3((total + 63) / 64, n), dim3(256), 0, s, slots);
    return hipGetLastError() == hipSuccess ? UAVENV_OK : sac_fail(UAVENV_EHIP, what);
}

int critic_adam_args(const UavSacNets *nets, const float *partials, int32_t rows, float *m1, float *v1, float *m2, float *v2,
                     const UavSacAdam *h, float *losses_out, AdamArgs &a)
{
    if (!nets || !partials || rows <= 0 || !m1 || !v1 || !m2 || !v2 || !h) return sac_fail(UAVENV_EINVAL, "uavenv_sac_critic_adam: null argument");
    a = AdamArgs();
    a.partials = partials; a.rows = rows; a.stride = kStrideC; a.nseg = 2; a.extras = 4;
    a.seg[0] = AdamSeg{nets->critic1, m1, v1, nets->target1, kPc, h->lr};
    a.seg[1] = AdamSeg{nets->critic2, m2, v2, nets->target2, kPc, h->lr};
    a.beta1 = h->beta1; a.beta2 = h->beta2; a.eps = h->eps; a.bc1 = h->bias_correction1; a.bc2_sqrt = h->bias_correction2_sqrt;
    a.tau = h->tau;
    a.grad_scale = 1.0f;
    a.frac_col = 2 * kPc + 2;
    a.n_loss = 2;
    a.scalars_out = losses_out;
    a.skip = h->skip_word;
    a.go = h->go_word; a.go_value = h->go_value;
    return UAVENV_OK;
}

int actor_adam_args(const UavSacNets *nets, const float *partials, int32_t rows, int32_t batch, float *m, float *v, float *alpha_mv,
                    const UavSacAdam *h, float alpha_lr, float target_entropy, float *scalars_out, AdamArgs &a)
{
    if (!nets || !partials || rows <= 0 || batch <= 0 || !m || !v || !alpha_mv || !h)
        return sac_fail(UAVENV_EINVAL, "uavenv_sac_actor_adam: null argument");
    a = AdamArgs();
    a.partials = partials; a.rows = rows; a.stride = kStrideA; a.nseg = 1; a.extras = 4;
    a.seg[0] = AdamSeg{nets->actor, m, v, nullptr, kPa, h->lr};
    a.beta1 = h->beta1; a.beta2 = h->beta2; a.eps = h->eps; a.bc1 = h->bias_correction1; a.bc2_sqrt = h->bias_correction2_sqrt;
    a.tau = 0.0f;
    a.grad_scale = 1.0f;
    a.frac_col = kPa + 2;
    a.n_loss = 1;
    a.scalars_out = scalars_out;
    a.log_alpha = nets->log_alpha; a.alpha_mv = alpha_mv; a.alpha_lr = alpha_lr; a.target_entropy = target_entropy;
    a.inv_2b = 0.5f / (float)batch;
    a.skip = h->skip_word;
    a.go = h->go_word; a.go_value = h->go_value;
    return UAVENV_OK;
}

}  // namespace

extern "C" {

const char *uavenv_sac_last_error(void) { return g_sac_err; }

int uavenv_sac_set_debug_buffer(unsigned long long *dev_buf)
{
    g_sac_dbg = dev_buf;
    return UAVENV_OK;
}

int uavenv_sac_partial_rows(int32_t batch)
{
    return uavenv_sac_partial_rows_n(batch, 1, 0);
}

int uavenv_sac_partial_rows_n(int32_t batch, int32_t n_slots, int32_t tiles_per_wg)
{
    if (batch <= 0 || batch % kTile || n_slots < 1 || n_slots > kSlots || tiles_per_wg < 0) return UAVENV_EINVAL;
    const int n_tiles = batch / kTile;
    const int tpw = tiles_per_wg_of(n_tiles, n_slots, tiles_per_wg);
    return (n_tiles + tpw - 1) / tpw;
}

int uavenv_sac_act_multi(const float *const *actors, const void *obs_packed, const int32_t *first_rows, int32_t row_stride,
                         int32_t count, const float *const *eps, float action_bound, float *act0, float *act1, int32_t n, void *stream)
{
    if (!actors || !obs_packed || !first_rows || !eps || !act0 || !act1 || count <= 0 || row_stride <= 0 || n < 1 || n > kSlots)
        return sac_fail(UAVENV_EINVAL, "uavenv_sac_act: bad argument");
    if (!aligned16(obs_packed)) return sac_fail(UAVENV_EINVAL, "uavenv_sac_act: 16-byte alignment");
    SacActArgsN slots;
    memset(&slots, 0, sizeof(slots));
    for (int j = 0; j < n; ++j) {
        if (!actors[j] || !eps[j] || first_rows[j] < 0 || !aligned16(actors[j])) return sac_fail(UAVENV_EINVAL, "uavenv_sac_act: bad slot argument");
        slots.s[j] = SacActArgs{actors[j], reinterpret_cast<const uint32_t *>(obs_packed), first_rows[j], row_stride, count, eps[j],
                                action_bound, act0, act1};
    }
    const int n_tiles = (count + kTile - 1) / kTile;
    // two workgroups per CU over all slots of the launch (UAVENV_SAC_ACT_WGS: A/B knob, workgroups per launch)
    static const int act_wgs = [] { const char *e = getenv("UAVENV_SAC_ACT_WGS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 512; }();
    int gx = act_wgs / n;
    gx = gx < 1 ? 1 : gx;
    hipLaunchKernelGGL(k_sac_act, dim3(n_tiles < gx ? n_tiles : gx, n), dim3(256), kSacActLds, (hipStream_t)stream, slots);
    return hipGetLastError() == hipSuccess ? UAVENV_OK : sac_fail(UAVENV_EHIP, "uavenv_sac_act: launch failed");
}

int uavenv_sac_act(const float *actor, const void *obs_packed, int32_t first_row, int32_t row_stride, int32_t count,
                   const float *eps, float action_bound, float *act0, float *act1, void *stream)
{
    return uavenv_sac_act_multi(&actor, obs_packed, &first_row, row_stride, count, &eps, action_bound, act0, act1, 1, stream);
}

int uavenv_sac_critic_grad(const UavSacNets *nets, const UavSacBatch *batch, float gamma, float action_bound, float *partials,
                           void *stream)
{
    return grad_phase(true, nets, batch, 1, gamma, action_bound, &partials, (hipStream_t)stream);
}

int uavenv_sac_actor_grad(const UavSacNets *nets, const UavSacBatch *batch, float action_bound, float *partials, void *stream)
{
    return grad_phase(false, nets, batch, 1, 0.0f, action_bound, &partials, (hipStream_t)stream);
}

int uavenv_sac_critic_grad_multi(const UavSacNets *nets, const UavSacBatch *batches, int32_t n, float gamma, float action_bound,
                                 float *const *partials, void *stream)
{
    return grad_phase(true, nets, batches, n, gamma, action_bound, partials, (hipStream_t)stream);
}

int uavenv_sac_actor_grad_multi(const UavSacNets *nets, const UavSacBatch *batches, int32_t n, float action_bound,
                                float *const *partials, void *stream)
{
    return grad_phase(false, nets, batches, n, 0.0f, action_bound, partials, (hipStream_t)stream);
}

// multi-GPU: column sums only (rows x stride -> raw[stride]); the caller all-reduces raw and passes it back as ONE row
int uavenv_sac_reduce(const float *partials, int32_t rows, int32_t stride, float *raw, void *stream)
{
    if (!partials || !raw || rows <= 0 || (stride != kStrideA && stride != kStrideC)) return sac_fail(UAVENV_EINVAL, "uavenv_sac_reduce: bad argument");
    AdamArgs a = AdamArgs();
    a.partials = partials; a.rows = rows; a.stride = stride; a.nseg = 1; a.extras = 0;
    a.seg[0].n = stride;
    a.raw_out = raw;
    a.frac_col = -1;
    return adam_launch(&a, 1, stride, (hipStream_t)stream, "uavenv_sac_reduce: launch failed");
}

int uavenv_sac_critic_adam(const UavSacNets *nets, const float *partials, int32_t rows, float *m1, float *v1, float *m2, float *v2,
                           const UavSacAdam *h, float *losses_out, void *stream)
{
    AdamArgs a;
    const int rc = critic_adam_args(nets, partials, rows, m1, v1, m2, v2, h, losses_out, a);
    if (rc != UAVENV_OK) return rc;
    return adam_launch(&a, 1, 2 * kPc + 4, (hipStream_t)stream, "uavenv_sac_critic_adam: launch failed");
}

int uavenv_sac_actor_adam(const UavSacNets *nets, const float *partials, int32_t rows, int32_t batch, float *m, float *v,
                          float *alpha_mv, const UavSacAdam *h, float alpha_lr, float target_entropy, float *scalars_out,
                          void *stream)
{
    AdamArgs a;
    const int rc = actor_adam_args(nets, partials, rows, batch, m, v, alpha_mv, h, alpha_lr, target_entropy, scalars_out, a);
    if (rc != UAVENV_OK) return rc;
    return adam_launch(&a, 1, kPa + 4, (hipStream_t)stream, "uavenv_sac_actor_adam: launch failed");
}

int uavenv_sac_critic_adam_multi(const UavSacNets *nets, float *const *partials, int32_t rows, float *const *m1, float *const *v1,
                                 float *const *m2, float *const *v2, const UavSacAdam *h, float *const *losses_out, int32_t n,
                                 void *stream)
{
    if (n < 1 || n > kSlots || !nets || !partials || !m1 || !v1 || !m2 || !v2 || !h) return sac_fail(UAVENV_EINVAL, "uavenv_sac_critic_adam_multi: bad argument");
    AdamArgs a[kSlots];
    for (int j = 0; j < n; ++j) {
        const int rc = critic_adam_args(&nets[j], partials[j], rows, m1[j], v1[j], m2[j], v2[j], &h[j], losses_out ? losses_out[j] : nullptr, a[j]);
        if (rc != UAVENV_OK) return rc;
    }
    return adam_launch(a, n, 2 * kPc + 4, (hipStream_t)stream, "uavenv_sac_critic_adam_multi: launch failed");
}

int uavenv_sac_actor_adam_multi(const UavSacNets *nets, float *const *partials, int32_t rows, int32_t batch, float *const *m,
                                float *const *v, float *const *alpha_mv, const UavSacAdam *h, float alpha_lr, float target_entropy,
                                float *const *scalars_out, int32_t n, void *stream)
{
    if (n < 1 || n > kSlots || !nets || !partials || !m || !v || !alpha_mv || !h) return sac_fail(UAVENV_EINVAL, "uavenv_sac_actor_adam_multi: bad argument");
    AdamArgs a[kSlots];
    for (int j = 0; j < n; ++j) {
        const int rc = actor_adam_args(&nets[j], partials[j], rows, batch, m[j], v[j], alpha_mv[j], &h[j], alpha_lr, target_entropy,
                                       scalars_out ? scalars_out[j] : nullptr, a[j]);
        if (rc != UAVENV_OK) return rc;
    }
    return adam_launch(a, n, kPa + 4, (hipStream_t)stream, "uavenv_sac_actor_adam_multi: launch failed");
}

}  // extern "C"
