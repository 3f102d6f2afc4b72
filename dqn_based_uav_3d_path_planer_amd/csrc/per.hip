// per.hip -- prioritised experience replay on the device (see include/uavenv.h, UavPer).
//
// The reference (BaseClass/replay_buffer.py:57-223) keeps a binary sum tree in a flat array and walks it from the
// root once per sample, one Python loop iteration per level.  On a GPU the tree is the wrong shape: an update would
// touch log2(c) dependent nodes per leaf, with write conflicts between leaves that share ancestors.  Here the
// priorities stay a flat f64 array in HBM (one per ring slot) and selection is a two-level prefix search:
//
//   k_per_chunk_sum   one workgroup per 1024 in-order leaf positions -> chunk_sum[b]        (reads c x 8 B)
//   k_per_prefix      exclusive prefix over the chunk sums -> chunk_prefix[n_chunks + 1]    (one workgroup)
//   k_per_sample      one WAVEFRONT per sample: binary search in chunk_prefix, then the 1024 leaves of that chunk
//                     are summed 16 per lane, scanned across the 64 lanes, and the owning lane walks its 16
//   k_per_set         batch_update / push: p[slot] = min(|err| + eps, clip) ** alpha
//   k_per_fill        the N slots of a freshly written ring frame get the "new transition" priority
//
// Selection rule = the tree's: the first in-order leaf whose inclusive cumulative priority reaches v (the descent
// goes left when v <= left sum, replay_buffer.py:106).  "In-order" matters: with a capacity that is not a power of
// two the flat tree visits its leaves rotated by rot = 2^floor(log2(2c-1)) - c slots, and position q holds slot
// (q + rot) % c.  Results equal the reference's for the same draws except where v lands within rounding of a leaf
// boundary (the tree's inner nodes are history-dependent float sums; these are fresh sums).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/uavenv.h"
#include "uavenv_device.hpp"

using namespace uav;

namespace {

constexpr int kChunk = 1024;

__device__ __forceinline__ int64_t slot_of(const UavPer &p, int64_t q)
{
    int64_t s = q + p.rot;
    return s >= p.capacity ? s - p.capacity : s;
}

// the two fills of a replay step (k_per_fill2) applied while the priorities are being read anyway: count == 0 -> none
struct PerFill {
    int64_t first, count, retire_first;
    double value;
    const uint8_t *valid;
};

__global__ void __launch_bounds__(256) k_per_chunk_sum(UavPer p, PerFill f)
{
    __shared__ double red[256];
    __shared__ double leaf[kChunk];
    const int64_t q0 = (int64_t)blockIdx.x * kChunk + (int64_t)threadIdx.x * 4;
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t q = q0 + j;
        double v = 0.0;
        if (q < p.capacity) {
            const int64_t sl = slot_of(p, q);
            if (sl >= f.first && sl < f.first + f.count) {
                v = (!f.valid || f.valid[sl - f.first]) ? f.value : 0.0;
                p.prio[sl] = v;
            } else if (sl >= f.retire_first && sl < f.retire_first + f.count) {
                p.prio[sl] = 0.0;
            } else {
                v = p.prio[sl];
            }
        }
        leaf[threadIdx.x * 4 + j] = v;
        s += v;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    // the chunk's 64 group sums: 16 consecutive in-order leaves each, added in leaf order from 0.0 -- exactly the sum a lane
    // of k_per_sample forms over its 16 leaves, so the sampler can read one double per lane instead of sixteen
    if (p.group_sum && threadIdx.x < 64) {
        double g = 0.0;
#pragma unroll
        for (int j = 0; j < 16; ++j) g += leaf[threadIdx.x * 16 + j];
        p.group_sum[(int64_t)blockIdx.x * 64 + threadIdx.x] = g;
    }
    for (int w = 128; w > 0; w >>= 1) {            // fixed pairing: the sum does not depend on scheduling
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) p.chunk_sum[blockIdx.x] = red[0];
}

__global__ void __launch_bounds__(256) k_per_prefix(UavPer p, int n_chunks)
{
    __shared__ double part[256];
    const int per = (n_chunks + 255) / 256;
    const int b0 = (int)threadIdx.x * per;
    double s = 0.0;
    for (int b = b0; b < b0 + per && b < n_chunks; ++b) s += p.chunk_sum[b];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double run = 0.0;
        for (int t = 0; t < 256; ++t) {
            const double v = part[t];
            part[t] = run;
            run += v;
        }
    }
    __syncthreads();
    double run = part[threadIdx.x];
    for (int b = b0; b < b0 + per && b < n_chunks; ++b) {
        p.chunk_prefix[b] = run;
        run += p.chunk_sum[b];
    }
    if (b0 < n_chunks && b0 + per >= n_chunks) p.chunk_prefix[n_chunks] = run;     // the total
}

// One wavefront per sample.
__global__ void __launch_bounds__(256) k_per_sample(UavPer p, int n_chunks, int batch, const double *__restrict__ draws,
                                                    uint64_t seed, uint64_t counter, int64_t *__restrict__ out_slot,
                                                    double *__restrict__ out_prio)
{
    const int lane = (int)threadIdx.x & 63;
    const int i = (int)blockIdx.x * 4 + ((int)threadIdx.x >> 6);
    if (i >= batch) return;
    double v;
    if (draws) {
        v = draws[i];
    } else {      // random.uniform(seg * i, seg * (i + 1)) with seg = int(total) / batch (replay_buffer.py:147,160-162)
        const double seg = floor(p.chunk_prefix[n_chunks]) / (double)batch;
        const uint4 r = philox4x32_10(make_uint4((uint32_t)i, (uint32_t)counter, (uint32_t)(counter >> 32), 0x9e7u),
                                      make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
        const double a = seg * (double)i, b = seg * (double)(i + 1);
        v = a + (b - a) * u53(r.x, r.y);
    }
    // chunk: the first b with v <= prefix[b + 1] and something in front of that boundary (wave-uniform binary search).
    // A draw past the total (caller's stream, or rounding of seg * batch) is pulled back onto it, and a draw of exactly 0
    // skips leading all-zero chunks: either would otherwise end in a chunk without a single positive priority.
    const double total = p.chunk_prefix[n_chunks];
    v = v < total ? v : total;
    // The predicate "v <= prefix[b + 1] and prefix[b + 1] > 0" is monotone in b; the first b that satisfies it is found by the
    // 64 lanes together in two loads (64 segments of `stride` chunks, then the chunks of the winning segment) instead of a
    // ten-deep chain of dependent loads per wavefront (26 of the 84 us of a prioritised pass).
    const int stride = (n_chunks + 63) / 64;
    int lo;
    {
        int last = (lane + 1) * stride;                       // one past the last chunk of this lane's segment
        last = last < n_chunks ? last : n_chunks;
        const double up = p.chunk_prefix[last];
        const unsigned long long hit1 = __ballot(v <= up && up > 0.0);
        const int seg = hit1 ? __builtin_ctzll(hit1) : 63;
        lo = n_chunks - 1;                                    // (nothing satisfies it: the last chunk, as the binary search ended)
        for (int base = seg * stride; base < (seg + 1) * stride && base < n_chunks; base += 64) {
            const int bq = base + lane;
            const bool in = bq < (seg + 1) * stride && bq < n_chunks;
            const double u2 = p.chunk_prefix[in ? bq + 1 : n_chunks];
            const unsigned long long hit2 = __ballot(in && v <= u2 && u2 > 0.0);
            if (hit2) { lo = base + __builtin_ctzll(hit2); break; }
        }
        if (!hit1) lo = n_chunks - 1;
    }
    const int b = lo;
    const double r = v - p.chunk_prefix[b];
    // 16 leaves per lane, then an inclusive scan over the 64 lane sums (fixed order)
    const int64_t q0 = (int64_t)b * kChunk + (int64_t)lane * 16;
    double pv[16];
    double s = 0.0;
    const bool grouped = p.group_sum != nullptr;
    if (grouped) {                           // one double per lane (the rebuild summed the lane's 16 leaves in this order)
        s = p.group_sum[(int64_t)b * 64 + lane];
    } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int64_t q = q0 + j;
            pv[j] = q < p.capacity ? p.prio[slot_of(p, q)] : 0.0;
            s += pv[j];
        }
    }
    double incl = s;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const double up = __shfl_up(incl, off, 64);
        if (lane >= off) incl += up;
    }
    const double excl = incl - s;
    const unsigned long long hit = __ballot(r <= incl);
    int owner = hit ? __builtin_ctzll(hit) : 63;
    int64_t q = -1;
    double pq = 0.0;
    bool found = false;
    if (grouped) {                           // only the lane that owns the pick reads its 16 leaves
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int64_t q = q0 + j;
            pv[j] = (hit && lane == owner && q < p.capacity) ? p.prio[slot_of(p, q)] : 0.0;
        }
    }
    if (hit && lane == owner) {
        double run = excl;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            run += pv[j];
            if (!found && r <= run && pv[j] > 0.0 && q0 + j < p.capacity) { q = q0 + j; pq = pv[j]; found = true; }
        }
    }
    if (__ballot(found) == 0) {
        if (grouped) {                       // (rare) the fall-back looks at every leaf of the chunk
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int64_t q = q0 + j;
                pv[j] = q < p.capacity ? p.prio[slot_of(p, q)] : 0.0;
            }
        }
        // Rounding can leave r above the chunk's fresh sum (or on a zero-priority leaf): take the chunk's LAST leaf with
        // p > 0 -- never a retired / invalid slot (priority 0), whose importance weight pow(0, -beta) would be inf.
        // Only a chunk that is all zero falls through to its last leaf with priority 0 (the caller masks those).
        int jl = -1;
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (pv[j] > 0.0 && q0 + j < p.capacity) { jl = j; pq = pv[j]; }
        const unsigned long long pos = __ballot(jl >= 0);
        if (pos) {
            owner = 63 - __builtin_clzll(pos);
            q = q0 + jl;
        } else {
            owner = 0;
            int64_t last = (int64_t)b * kChunk + kChunk - 1;
            if (last >= p.capacity) last = p.capacity - 1;
            q = last;
            pq = 0.0;
        }
    }
    if (lane == owner) {
        out_slot[i] = slot_of(p, q);
        if (out_prio) out_prio[i] = pq;
    }
}

__global__ void k_per_set(UavPer p, const int64_t *__restrict__ slots, const double *__restrict__ abs_err, int n,
                          double epsilon, double alpha, double clip)
{
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n) return;
    const int64_t s = slots[i];
    if (s < 0 || s >= p.capacity) return;
    // batch_update (clip > 0) re-prioritises SAMPLED transitions: a slot whose priority is 0 is an empty leaf (a retired or
    // never-valid ring row; the sampler only returns one when the whole tree is empty) and stays empty
    if (clip > 0.0 && p.prio[s] == 0.0) return;
    // ReplayTree.batch_update (:215-222) walks the batch in order: when a leaf was drawn more than once the LAST sample's error
    // wins.  Stratified draws come back in non-decreasing prefix order, so equal slots are ADJACENT: a sample followed by the
    // same slot leaves the write to its successor (one writer per slot: deterministic, and the reference's winner)
    if (i + 1 < n && slots[i + 1] == s) return;
    double e = fabs(abs_err[i]) + epsilon;
    if (clip > 0.0 && e > clip) e = clip;
    p.prio[s] = pow(e, alpha);
}

__global__ void k_per_set_f32(UavPer p, const int64_t *__restrict__ slots, const float *__restrict__ abs_err, int n,
                              double epsilon, double alpha, double clip, const uint32_t *__restrict__ go_word, uint32_t go_value)
{
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n) return;
    if (go_word && __hip_atomic_load(go_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != go_value) return;   // (gated: see uavenv_per_set_f32_gated)
    const int64_t s = slots[i];
    if (s < 0 || s >= p.capacity) return;
    if (clip > 0.0 && p.prio[s] == 0.0) return;          // (an empty leaf stays empty: see k_per_set)
    if (i + 1 < n && slots[i + 1] == s) return;          // (the last of a run of equal slots writes: see k_per_set)
    double e = fabs((double)abs_err[i]) + epsilon;
    if (clip > 0.0 && e > clip) e = clip;
    p.prio[s] = pow(e, alpha);
}

// ReplayTree.sample's importance weights (:175-178): w_i = (n p_i / int(total)) ** -beta, then / max_i w_i.  Two launches:
// the powers (one f64 pow per thread) with one maximum per workgroup, then the division by the maximum of those maxima
// (fixed-order LDS trees: deterministic).  `w64` holds the unnormalised weights in between (it may alias the priorities),
// `wg_max` one double per workgroup of the first launch.
__global__ void __launch_bounds__(256) k_per_weights_pow(UavPer p, int n_chunks, const double *__restrict__ prio, int batch,
                                                         double n_entries, double beta, double *__restrict__ w64,
                                                         double *__restrict__ wg_max)
{
    __shared__ double red[256];
    double total = floor(p.chunk_prefix[n_chunks]);
    total = total < 1.0 ? 1.0 : total;
    const int i = (int)(blockIdx.x * 256 + threadIdx.x);
    double w = 0.0;
    if (i < batch) {
        const double pi = prio[i];
        w = pi > 0.0 ? pow(n_entries * (pi / total), -beta) : 0.0;
        w64[i] = w;
    }
    red[threadIdx.x] = w;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) red[threadIdx.x] = red[threadIdx.x] > red[threadIdx.x + k] ? red[threadIdx.x] : red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) wg_max[blockIdx.x] = red[0];
}

__global__ void __launch_bounds__(256) k_per_weights_norm(const int64_t *__restrict__ slots, const double *__restrict__ w64,
                                                          const double *__restrict__ wg_max, int n_wg, int batch, int n_agents,
                                                          float *__restrict__ w_out, int32_t *__restrict__ fa_out)
{
    __shared__ double red[256];
    double mx = 0.0;
    for (int k = (int)threadIdx.x; k < n_wg; k += 256) mx = wg_max[k] > mx ? wg_max[k] : mx;
    red[threadIdx.x] = mx;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) red[threadIdx.x] = red[threadIdx.x] > red[threadIdx.x + k] ? red[threadIdx.x] : red[threadIdx.x + k];
        __syncthreads();
    }
    mx = red[0];
    mx = mx > 2.2250738585072014e-308 ? mx : 2.2250738585072014e-308;
    const int i = (int)(blockIdx.x * 256 + threadIdx.x);
    if (i >= batch) return;
    w_out[i] = (float)(w64[i] / mx);
    if (fa_out) {
        const int64_t s = slots[i];
        const int64_t f = s / n_agents;
        fa_out[2 * i] = (int32_t)f;
        fa_out[2 * i + 1] = (int32_t)(s - f * n_agents);
    }
}

__global__ void k_per_fill(UavPer p, int64_t first, int64_t count, double value, const uint8_t *__restrict__ valid)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    p.prio[first + i] = (!valid || valid[i]) ? value : 0.0;
}

// the two fills of a replay step in one launch: [first, first + count) as k_per_fill, [zero_first, zero_first + count) <- 0
__global__ void k_per_fill2(UavPer p, int64_t first, int64_t count, double value, const uint8_t *__restrict__ valid, int64_t zero_first,
                            int64_t valid_stride)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) p.prio[first + i] = (!valid || valid[i * valid_stride]) ? value : 0.0;
    else if (i < 2 * count) p.prio[zero_first + (i - count)] = 0.0;
}

bool per_ok(const UavPer *p)
{
    return p && p->prio && p->chunk_sum && p->chunk_prefix && p->capacity > 0 && p->rot >= 0 && p->rot < p->capacity;
}

}  // namespace

extern "C" {

int uavenv_per_num_chunks(int64_t capacity) { return capacity > 0 ? (int)((capacity + kChunk - 1) / kChunk) : 0; }

int uavenv_per_rotation(int64_t capacity)
{
    if (capacity <= 0) return 0;
    int depth = 0;
    while ((2 * capacity - 1) >> (depth + 1)) ++depth;          // floor(log2(2c - 1))
    return (int)(((int64_t)1 << depth) - capacity);
}

int uavenv_per_rebuild(const UavPer *p, void *stream)
{
    return uavenv_per_rebuild_frame(p, 0, 0, 0.0, nullptr, 0, stream);
}

int uavenv_per_rebuild_frame(const UavPer *p, int64_t first, int64_t count, double priority, const uint8_t *valid_dev,
                             int64_t retire_first, void *stream)
{
    if (!per_ok(p) || first < 0 || retire_first < 0 || count < 0 || first + count > p->capacity || retire_first + count > p->capacity)
        return UAVENV_EINVAL;
    const int nc = uavenv_per_num_chunks(p->capacity);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_per_chunk_sum, dim3(nc), dim3(256), 0, s, *p, PerFill{first, count, retire_first, priority, valid_dev});
    hipLaunchKernelGGL(k_per_prefix, dim3(1), dim3(256), 0, s, *p, nc);
    return hipGetLastError() == hipSuccess ? UAVENV_OK : UAVENV_EHIP;
}

int uavenv_per_sample(const UavPer *p, int32_t batch, const double *draws_dev, uint64_t seed, uint64_t counter,
                      int64_t *out_slot_dev, double *out_prio_dev, void *stream)
{
    if (!per_ok(p) || batch <= 0 || !out_slot_dev) return UAVENV_EINVAL;
    const int nc = uavenv_per_num_chunks(p->capacity);
    hipLaunchKernelGGL(k_per_sample, dim3((batch + 3) / 4), dim3(256), 0, (hipStream_t)stream, *p, nc, batch, draws_dev,
                       seed, counter, out_slot_dev, out_prio_dev);
    return hipGetLastError() == hipSuccess ? UAVENV_OK : UAVENV_EHIP;
}

int uavenv_per_set(const UavPer *p, const int64_t *slots_dev, const double *abs_err_dev, int32_t n, double epsilon,
                   double alpha, double clip, void *stream)
{
    if (!per_ok(p) || !slots_dev || !abs_err_dev || n < 0) return UAVENV_EINVAL;
    if (n == 0) return UAVENV_OK;
    hipLaunchKernelGGL(k_per_set, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, *p, slots_dev, abs_err_dev, n,
                       epsilon, alpha, clip);
    return hipGetLastError() == hipSuccess ? UAVENV_OK : UAVENV_EHIP;
}

int uavenv_per_fill_frame(const UavPer *p, int64_t first, int64_t count, double priority, const uint8_t *valid_dev,
                          int64_t retire_first, void *stream)
{
    if (!per_ok(p) || first < 0 || retire_first < 0 || count < 0 || first + count > p->capacity || retire_first + count > p->capacity)
        return UAVENV_EINVAL;
    if (count == 0) return UAVENV_OK;
    hipLaunchKernelGGL(k_per_fill2, dim3((unsigned)((2 * count + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *p, first, count,
                       priority, valid_dev, retire_first, (int64_t)1);
    return hipGetLastError() == hipSuccess ? UAVENV_OK : UAVENV_EHIP;
}

int uavenv_per_fill_frame_strided(const UavPer *p, int64_t first, int64_t count, double priority, const uint8_t *valid_dev,
                                  int64_t valid_stride, int64_t retire_first, void *stream)
{
    if (!per_ok(p) || first < 0 || retire_first < 0 || count < 0 || first + count > p->capacity || retire_first + count > p->capacity ||
        valid_stride < 1)
        return UAVENV_EINVAL;
    if (count == 0) return UAVENV_OK;
    hipLaunchKernelGGL(k_per_fill2, dim3((unsigned)((2 * count + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *p, first, count,
                       priority, valid_dev, retire_first, valid_stride);
    return hipGetLastError() == hipSuccess ? UAVENV_OK : UAVENV_EHIP;
}

int uavenv_per_set_f32(const UavPer *p, const int64_t *slots_dev, const float *abs_err_dev, int32_t n, double epsilon,
                       double alpha, double clip, void *stream)
{
    return uavenv_per_set_f32_gated(p, slots_dev, abs_err_dev, n, epsilon, alpha, clip, nullptr, 0u, stream);
}

int uavenv_per_set_f32_gated(const UavPer *p, const int64_t *slots_dev, const float *abs_err_dev, int32_t n, double epsilon,
                             double alpha, double clip, const uint32_t *go_word_dev, uint32_t go_value, void *stream)
{
    if (!per_ok(p) || !slots_dev || !abs_err_dev || n < 0) return UAVENV_EINVAL;
    if (n == 0) return UAVENV_OK;
    hipLaunchKernelGGL(k_per_set_f32, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, *p, slots_dev, abs_err_dev, n,
                       epsilon, alpha, clip, go_word_dev, go_value);
    return hipGetLastError() == hipSuccess ? UAVENV_OK : UAVENV_EHIP;
}

int uavenv_per_weights(const UavPer *p, const int64_t *slots_dev, double *prio_dev, int32_t batch, int64_t n_entries,
                       double beta, int32_t n_agents, float *is_weights_out_dev, int32_t *frame_agent_out_dev, void *stream)
{
    if (!per_ok(p) || !slots_dev || !prio_dev || !is_weights_out_dev || batch <= 0 || n_entries < 0) return UAVENV_EINVAL;
    if (frame_agent_out_dev && n_agents <= 0) return UAVENV_EINVAL;
    const int n_wg = (batch + 255) / 256;
    double *wg_max = prio_dev + batch;                         // the scratch tail the caller left behind the priorities
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_per_weights_pow, dim3(n_wg), dim3(256), 0, s, *p, uavenv_per_num_chunks(p->capacity), prio_dev, batch,
                       (double)n_entries, beta, prio_dev, wg_max);
    hipLaunchKernelGGL(k_per_weights_norm, dim3(n_wg), dim3(256), 0, s, slots_dev, prio_dev, wg_max, n_wg, batch,
                       n_agents > 0 ? n_agents : 1, is_weights_out_dev, frame_agent_out_dev);
    return hipGetLastError() == hipSuccess ? UAVENV_OK : UAVENV_EHIP;
}

int uavenv_per_fill(const UavPer *p, int64_t first, int64_t count, double priority, const uint8_t *valid_dev, void *stream)
{
    if (!per_ok(p) || first < 0 || count < 0 || first + count > p->capacity) return UAVENV_EINVAL;
    if (count == 0) return UAVENV_OK;
    hipLaunchKernelGGL(k_per_fill, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *p, first,
                       count, priority, valid_dev);
    return hipGetLastError() == hipSuccess ? UAVENV_OK : UAVENV_EHIP;
}

}  // extern "C"
