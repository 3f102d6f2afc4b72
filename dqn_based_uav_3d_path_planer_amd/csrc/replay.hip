// replay.hip -- device-resident replay sampling and epsilon-greedy acting (see include/uavenv.h).
//
// The replay ring itself is written by k_step (uavenv.hip): the observation of frame t+1 and the
// reward/done/valid of frame t go straight from the env kernel into the caller's ring tensors, so
// ReplayMemory.add (BaseClass/replay_buffer.py:41-42) costs no extra HBM traffic.  What is left of
// ReplayMemory on the learner side is sample2 (replay_buffer.py:48-51): uniform draws + a gather of
// 2 x 400 B rows per sample, done here as one launch (HBM-bound: 813 B per sample).
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "../../include/uavenv.h"
#include "uavenv_device.hpp"

using namespace uav;

extern "C" const char *uavenv_last_error(void);

namespace {

// One 16-byte chunk per thread: chunk q of sample s, q in [0, 2*CH) = obs row then next_obs row.
// CH = 25 (f32 rows, 400 B) or 13 (f16 rows, 200 B = 12.5 chunks -> handled as 8-byte units, CH8 = 25).
template <int UNIT /*bytes per copy unit*/, int ROW_UNITS = 25>
__global__ void k_sample_gather(UavReplayRing ring, int head, int batch, ReplayPerm perm,
                                unsigned char *__restrict__ obs_b, unsigned char *__restrict__ next_b,
                                unsigned char *__restrict__ act_b, float *__restrict__ rew_b,
                                float *__restrict__ done_b, float *__restrict__ valid_b)
{
    // ROW_UNITS x UNIT bytes per row: 25 x 16 B (f32), 25 x 8 B (f16), 5 x 16 B (packed)
    const int row_bytes = ROW_UNITS * UNIT;
    const int64_t total = (int64_t)batch * (2 * ROW_UNITS);
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int s = (int)(t / (2 * ROW_UNITS));
        const int q = (int)(t - (int64_t)s * (2 * ROW_UNITS));
        // every thread of a sample re-derives the same draw (SIMD lanes: no extra instructions per wavefront)
        int f, agent;
        replay_slot_to_frame(perm, replay_perm_apply(perm, (uint32_t)s), head, ring.frames, f, agent);
        int fn = f + 1;
        if (fn >= ring.frames) fn = 0;
        const bool is_next = q >= ROW_UNITS;
        const int u = is_next ? q - ROW_UNITS : q;
        const size_t src_row = ((size_t)(is_next ? fn : f) * ring.n_agents + agent) * (size_t)row_bytes;
        const unsigned char *src = reinterpret_cast<const unsigned char *>(ring.obs) + src_row + (size_t)u * UNIT;
        unsigned char *dst = (is_next ? next_b : obs_b) + (size_t)s * row_bytes + (size_t)u * UNIT;
        if (UNIT == 16) *reinterpret_cast<uint4 *>(dst) = *reinterpret_cast<const uint4 *>(src);
        else *reinterpret_cast<uint2 *>(dst) = *reinterpret_cast<const uint2 *>(src);
        if (q == 0) {
            const size_t k = (size_t)f * ring.n_agents + agent;
            reinterpret_cast<uint32_t *>(act_b)[s] = reinterpret_cast<const uint32_t *>(ring.action)[k];
            rew_b[s] = ring.reward[k];
            done_b[s] = (float)ring.done[k];
            if (valid_b) valid_b[s] = ring.valid ? (float)ring.valid[k] : 1.0f;
        }
    }
}

// packed rows -> f32 / f16 rows: one thread per group of four columns (coalesced 16 / 8 byte stores)
template <bool F16>
__global__ void k_obs_unpack(const uint32_t *__restrict__ packed, int64_t n, void *__restrict__ out)
{
    const int64_t total = n * 25;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = t / 25;
        const int q = (int)(t - row * 25);
        const uint32_t *p = packed + row * kPackedDwords;
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = packed_col(p, 4 * q + k);
        if (!F16) {
            reinterpret_cast<float4 *>(out)[t] = make_float4(v[0], v[1], v[2], v[3]);
        } else {
            __half2 lo = __floats2half2_rn(v[0], v[1]), hi = __floats2half2_rn(v[2], v[3]);
            uint2 o;
            o.x = *reinterpret_cast<uint32_t *>(&lo);
            o.y = *reinterpret_cast<uint32_t *>(&hi);
            reinterpret_cast<uint2 *>(out)[t] = o;
        }
    }
}

// the draws alone: (frame, agent) of samples 0 .. batch-1 (tests, prioritised replay bookkeeping)
__global__ void k_replay_draw(int frames, int n_agents, int head, int batch, ReplayPerm perm, int32_t *__restrict__ out)
{
    for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < batch; s += gridDim.x * blockDim.x) {
        int f, agent;
        replay_slot_to_frame(perm, replay_perm_apply(perm, (uint32_t)s), head, frames, f, agent);
        out[2 * s] = f;
        out[2 * s + 1] = agent;
    }
}

// The draws over VALID rows only.  The reference's buffers never hold a row of a finished agent (run_eposide stops pushing for it,
// Envs/PathPlan_City.py:456-459; BaseClass/replay_buffer.py:41-51 samples what was pushed); the ring keeps such rows with valid = 0.
// Draw s of slot j (s in [j * batch, (j + 1) * batch)) looks at permutation positions s, s + S, s + 2 S, ... (S = n_slots * batch,
// positions < D) until the row (frame, env, first_slot + j) is valid: rejection sampling over a bijection, so accepted rows are
// uniform over the valid ones and stay distinct within a slot.  A draw that runs out of tries keeps its first row (valid = 0: weight
// 0 in the update, the behaviour before this kernel).
__global__ void k_replay_draw_valid(int frames, int n_envs, int head, int batch, int n_slots, int uav, int first_slot,
                                    const unsigned char *__restrict__ valid, int max_tries, ReplayPerm perm, int32_t *__restrict__ out)
{
    const int total = batch * n_slots;
    for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < total; s += gridDim.x * blockDim.x) {
        const int slot = first_slot + s / batch;
        int f0 = 0, e0 = 0;
        bool found = false;
        for (int t = 0; t < max_tries && !found; ++t) {
            const uint64_t q = (uint64_t)s + (uint64_t)t * (uint64_t)total;
            if (q >= (uint64_t)perm.D) break;
            int f, e;
            replay_slot_to_frame(perm, replay_perm_apply(perm, (uint32_t)q), head, frames, f, e);
            if (t == 0) { f0 = f; e0 = e; }
            if (valid[((size_t)f * n_envs + e) * uav + slot]) { f0 = f; e0 = e; found = true; }
        }
        out[2 * s] = f0;
        out[2 * s + 1] = e0;
    }
}

// Trainer/DuelingDQN_Trainer.py:86-97: sample > eps -> argmax_a Q(s,a) (first maximum, as torch.max), else randrange(A).
__global__ void k_select_actions(const float *__restrict__ q, int n, int A, float eps, uint64_t seed, uint64_t counter,
                                 int32_t *__restrict__ idx_out, float *__restrict__ steer_out)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint4 r = philox4x32_10(make_uint4((uint32_t)i, (uint32_t)counter, (uint32_t)(counter >> 32), 0xac7u),
                                      make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
        const float sample = (float)(r.x >> 8) * (1.0f / 16777216.0f);
        int a;
        if (sample > eps) {
            const float *row = q + (size_t)i * A;
            float best = row[0];
            a = 0;
            for (int k = 1; k < A; ++k) {
                const float v = row[k];
                if (v > best) { best = v; a = k; }
            }
        } else {
            a = (int)(((uint64_t)r.y * (uint64_t)A) >> 32);
        }
        if (idx_out) idx_out[i] = a;
        if (steer_out) steer_out[i] = (float)(-1.0 + 2.0 * (double)a / (double)(A - 1));
    }
}

}  // namespace

extern "C" {

int uavenv_replay_sample(const UavReplayRing *ring, int32_t head, int32_t filled, int32_t batch, uint64_t seed,
                         uint64_t counter, void *obs_b, void *next_obs_b, void *action_b, float *reward_b,
                         float *done_b, float *valid_b, void *stream)
{
    if (!ring || !ring->obs || !ring->action || !ring->reward || !ring->done || !obs_b || !next_obs_b || !action_b ||
        !reward_b || !done_b)
        return UAVENV_EINVAL;
    if (ring->frames < 2 || ring->n_agents <= 0 || batch <= 0 || filled <= 0 || filled > ring->frames - 1 ||
        head < 0 || head >= ring->frames || (uint64_t)filled * (uint64_t)ring->n_agents >= (1ull << 32))
        return UAVENV_EINVAL;
    const int64_t total = (int64_t)batch * (ring->obs_dtype == UAVENV_OBS_PACKED ? 10 : 50);
    const int block = 256;
    int64_t g = (total + block - 1) / block;
    const int grid = (int)(g > 4096 ? 4096 : g);
    hipStream_t s = (hipStream_t)stream;
    const ReplayPerm perm = replay_perm(seed, counter, (uint32_t)filled * (uint32_t)ring->n_agents, (uint32_t)ring->n_agents);
    if (ring->obs_dtype == UAVENV_OBS_F32)
        hipLaunchKernelGGL((k_sample_gather<16>), dim3(grid), dim3(block), 0, s, *ring, head, batch, perm,
                           (unsigned char *)obs_b, (unsigned char *)next_obs_b, (unsigned char *)action_b, reward_b,
                           done_b, valid_b);
    else if (ring->obs_dtype == UAVENV_OBS_PACKED)
        hipLaunchKernelGGL((k_sample_gather<16, 5>), dim3(grid), dim3(block), 0, s, *ring, head, batch, perm,
                           (unsigned char *)obs_b, (unsigned char *)next_obs_b, (unsigned char *)action_b, reward_b,
                           done_b, valid_b);
    else
        hipLaunchKernelGGL((k_sample_gather<8>), dim3(grid), dim3(block), 0, s, *ring, head, batch, perm,
                           (unsigned char *)obs_b, (unsigned char *)next_obs_b, (unsigned char *)action_b, reward_b,
                           done_b, valid_b);
    return hipGetLastError() == hipSuccess ? UAVENV_OK : UAVENV_EHIP;
}

int uavenv_obs_unpack(const void *packed_dev, int64_t n, void *out_dev, int32_t out_dtype, void *stream)
{
    if (!packed_dev || !out_dev || n <= 0 || (out_dtype != UAVENV_OBS_F32 && out_dtype != UAVENV_OBS_F16)) return UAVENV_EINVAL;
    const int block = 256;
    int64_t g = (n * 25 + block - 1) / block;
    const int grid = (int)(g > 8192 ? 8192 : g);
    if (out_dtype == UAVENV_OBS_F32)
        hipLaunchKernelGGL((k_obs_unpack<false>), dim3(grid), dim3(block), 0, (hipStream_t)stream,
                           (const uint32_t *)packed_dev, n, out_dev);
    else
        hipLaunchKernelGGL((k_obs_unpack<true>), dim3(grid), dim3(block), 0, (hipStream_t)stream,
                           (const uint32_t *)packed_dev, n, out_dev);
    return hipGetLastError() == hipSuccess ? UAVENV_OK : UAVENV_EHIP;
}

int uavenv_replay_draw(int32_t frames, int32_t n_agents, int32_t head, int32_t filled, int32_t batch, uint64_t seed,
                       uint64_t counter, int32_t *frame_agent_out, void *stream)
{
    if (!frame_agent_out || frames < 2 || n_agents <= 0 || batch <= 0 || filled <= 0 || filled > frames - 1 || head < 0 ||
        head >= frames || (uint64_t)filled * (uint64_t)n_agents >= (1ull << 32))
        return UAVENV_EINVAL;
    const int block = 256;
    int grid = (batch + block - 1) / block;
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(k_replay_draw, dim3(grid), dim3(block), 0, (hipStream_t)stream, frames, n_agents, head, batch,
                       replay_perm(seed, counter, (uint32_t)filled * (uint32_t)n_agents, (uint32_t)n_agents), frame_agent_out);
    return hipGetLastError() == hipSuccess ? UAVENV_OK : UAVENV_EHIP;
}

int uavenv_replay_draw_valid(int32_t frames, int32_t n_envs, int32_t head, int32_t filled, int32_t batch, int32_t n_slots,
                             int32_t uav_per_env, int32_t first_slot, const uint8_t *valid, int32_t max_tries, uint64_t seed,
                             uint64_t counter, int32_t *frame_agent_out, void *stream)
{
    if (!frame_agent_out || !valid || frames < 2 || n_envs <= 0 || batch <= 0 || n_slots <= 0 || uav_per_env <= 0 || first_slot < 0 ||
        first_slot + n_slots > uav_per_env || max_tries <= 0 || filled <= 0 || filled > frames - 1 || head < 0 || head >= frames ||
        (uint64_t)filled * (uint64_t)n_envs >= (1ull << 32) || (int64_t)batch * n_slots >= (1ll << 31))
        return UAVENV_EINVAL;
    const int block = 256;
    int grid = (batch * n_slots + block - 1) / block;
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(k_replay_draw_valid, dim3(grid), dim3(block), 0, (hipStream_t)stream, frames, n_envs, head, batch, n_slots,
                       uav_per_env, first_slot, valid, max_tries,
                       replay_perm(seed, counter, (uint32_t)filled * (uint32_t)n_envs, (uint32_t)n_envs), frame_agent_out);
    return hipGetLastError() == hipSuccess ? UAVENV_OK : UAVENV_EHIP;
}

int uavenv_select_actions(const float *q, int32_t n, int32_t n_actions, float eps, uint64_t seed, uint64_t counter,
                          int32_t *index_out, float *steer_out, void *stream)
{
    if (!q || n <= 0 || n_actions < 2 || (!index_out && !steer_out)) return UAVENV_EINVAL;
    const int block = 256;
    int grid = (n + block - 1) / block;
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(k_select_actions, dim3(grid), dim3(block), 0, (hipStream_t)stream, q, n, n_actions, eps, seed,
                       counter, index_out, steer_out);
    return hipGetLastError() == hipSuccess ? UAVENV_OK : UAVENV_EHIP;
}

}  // extern "C"
