// counter_calibration.hip -- known byte counts in the step kernel's own access patterns, to calibrate rocprofv3's FETCH_SIZE and
// WRITE_SIZE on gfx950 (MI355X_MICROARCH.md, HBM: "calibrate on a known byte count in your own access pattern before trusting an
// absolute").  DIAGNOSTIC, not product code.  One kernel per pattern, so that a --pmc pass reports each by name:
//   cal_read16      16 B per lane, coalesced streaming read            (the guide's calibrated case: FETCH_SIZE = 1/2 of the bytes)
//   cal_read8       one f64 per lane from each of 15 planes            (the step's state read: pos, V_vector, V, goal, s0, s1)
//   cal_read4       one i32 per lane from each of 4 planes
//   cal_read_rows80 80-byte rows, a wavefront's 64 rows as 5 KiB of dwordx4  (the policy's row read)
//   cal_write8      one f64 per lane into each of 6 planes             (pos, V_vector, V)
//   cal_write4      one i32 / f32 per lane into each of 3 planes       (Step, sub_idx, reward32)
//   cal_write1      one byte per lane into each of 5 planes            (done, valid, info, ret_done, agent_done)
//   cal_write_rows80  80-byte rows, a wavefront's 64 rows as 5 KiB of dwordx4
//   cal_write_rec16   one 16-byte record per lane
//   cal_step_like   all of the writes + reads above in one launch      (what k_step moves per agent, minus the world / bank gathers)
// Every launch works on a FRESH frame of a ring of frames (like the replay ring), n agents per frame.
//   hipcc --offload-arch=gfx950 -O3 -o counter_calibration counter_calibration.hip ; ./counter_calibration <n_agents> <launches>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void cal_read16(const uint4 *__restrict__ src, size_t n16, uint32_t *__restrict__ sink)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = src[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
template <typename T, int PLANES>
__device__ void cal_read_planes(const T *__restrict__ src, int n, uint32_t *__restrict__ sink)
{
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n) return;
    T acc = 0;
#pragma unroll
    for (int p = 0; p < PLANES; ++p) acc += src[(size_t)p * n + i];
    if (acc == (T)0x12345678) sink[0] = 1;
}
__global__ void cal_read8(const double *s, int n, uint32_t *k) { cal_read_planes<double, 15>(s, n, k); }
__global__ void cal_read4(const int32_t *s, int n, uint32_t *k) { cal_read_planes<int32_t, 4>(s, n, k); }
__global__ void cal_read_rows80(const uint4 *__restrict__ rows, int n, uint32_t *__restrict__ sink)
{
    // 64 rows of 80 B = 320 dwordx4 per wavefront: 5 loads per lane, consecutive lanes consecutive 16-byte pieces
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (wave * 64 >= (size_t)n) return;
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < 5; ++k) { const uint4 v = rows[wave * 320 + k * 64 + lane]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) sink[0] = acc;
}
template <typename T, int PLANES>
__device__ void write_planes(T *__restrict__ dst, int n, T v)
{
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n) return;
#pragma unroll
    for (int p = 0; p < PLANES; ++p) dst[(size_t)p * n + i] = v + (T)p;
}
__global__ void cal_write8(double *d, int n, double v) { write_planes<double, 6>(d, n, v); }
__global__ void cal_write4(int32_t *d, int n, int32_t v) { write_planes<int32_t, 3>(d, n, v); }
__global__ void cal_write1(uint8_t *d, int n, uint8_t v) { write_planes<uint8_t, 5>(d, n, v); }
__global__ void cal_write_rows80(uint4 *__restrict__ rows, int n, uint32_t v)
{
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (wave * 64 >= (size_t)n) return;
#pragma unroll
    for (int k = 0; k < 5; ++k) rows[wave * 320 + k * 64 + lane] = make_uint4(v, v + k, lane, 7u);
}
__global__ void cal_write_rec16(uint4 *__restrict__ rec, int n, uint32_t v)
{
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i < n) rec[i] = make_uint4(v, i, 3u, 1u);
}
struct StepLike {
    const double *r8; const int32_t *r4; const uint4 *rrow;
    double *w8; int32_t *w4; uint8_t *w1; uint4 *wrow; uint4 *wrec;
};
__global__ void cal_step_like(StepLike a, int n, uint32_t v, uint32_t *__restrict__ sink)
{
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (i >= n) return;
    double acc = 0;
#pragma unroll
    for (int p = 0; p < 15; ++p) acc += a.r8[(size_t)p * n + i];
    int32_t ai = 0;
#pragma unroll
    for (int p = 0; p < 4; ++p) ai += a.r4[(size_t)p * n + i];
    uint32_t ar = 0;
#pragma unroll
    for (int k = 0; k < 5; ++k) { const uint4 q = a.rrow[wave * 320 + k * 64 + lane]; ar ^= q.x ^ q.w; }
#pragma unroll
    for (int p = 0; p < 6; ++p) a.w8[(size_t)p * n + i] = acc + p;
#pragma unroll
    for (int p = 0; p < 3; ++p) a.w4[(size_t)p * n + i] = ai + p;
#pragma unroll
    for (int p = 0; p < 5; ++p) a.w1[(size_t)p * n + i] = (uint8_t)(ar + p);
#pragma unroll
    for (int k = 0; k < 5; ++k) a.wrow[wave * 320 + k * 64 + lane] = make_uint4(v, ar, lane, k);
    a.wrec[i] = make_uint4(v, ai, ar, 1u);
    if (acc == 1.2345e300) sink[0] = 1;
}

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 16384;
    const int launches = argc > 2 ? atoi(argv[2]) : 32;
    const int frames = 64;                              // every launch a fresh frame (frames x n x 600 B: 0.63 GB at n = 16384 ... )
    const size_t fr = (size_t)frames;
    double *r8, *w8; int32_t *r4, *w4; uint8_t *w1; uint4 *rrow, *wrow, *wrec, *big; uint32_t *sink;
    CHECK(hipMalloc(&r8, fr * 15 * n * 8)); CHECK(hipMalloc(&w8, fr * 6 * n * 8));
    CHECK(hipMalloc(&r4, fr * 4 * n * 4)); CHECK(hipMalloc(&w4, fr * 3 * n * 4));
    CHECK(hipMalloc(&w1, fr * 5 * n)); CHECK(hipMalloc(&rrow, fr * n * 80)); CHECK(hipMalloc(&wrow, fr * n * 80));
    CHECK(hipMalloc(&wrec, fr * n * 16)); CHECK(hipMalloc(&sink, 64));
    const size_t big16 = (size_t)n * 80 / 16 * 8;       // cal_read16: 8 x the row bytes per launch
    CHECK(hipMalloc(&big, fr * big16 * 16));
    CHECK(hipMemset(r8, 1, fr * 15 * n * 8)); CHECK(hipMemset(r4, 1, fr * 4 * n * 4)); CHECK(hipMemset(rrow, 1, fr * n * 80));
    CHECK(hipMemset(big, 1, fr * big16 * 16));
    // flush what the memsets left in the caches: touch 512 MB
    void *junk; CHECK(hipMalloc(&junk, (size_t)512 << 20)); CHECK(hipMemset(junk, 0, (size_t)512 << 20)); CHECK(hipDeviceSynchronize());
    const int block = 256, grid = (n + block - 1) / block;
    for (int it = 0; it < launches; ++it) {
        const size_t f = (size_t)(it % frames);
        hipLaunchKernelGGL(cal_read16, dim3(1024), dim3(256), 0, 0, big + f * big16, big16, sink);
        hipLaunchKernelGGL(cal_read8, dim3(grid), dim3(block), 0, 0, r8 + f * 15 * n, n, sink);
        hipLaunchKernelGGL(cal_read4, dim3(grid), dim3(block), 0, 0, r4 + f * 4 * n, n, sink);
        hipLaunchKernelGGL(cal_read_rows80, dim3(grid), dim3(block), 0, 0, rrow + f * n * 5, n, sink);
        hipLaunchKernelGGL(cal_write8, dim3(grid), dim3(block), 0, 0, w8 + f * 6 * n, n, (double)it);
        hipLaunchKernelGGL(cal_write4, dim3(grid), dim3(block), 0, 0, w4 + f * 3 * n, n, it);
        hipLaunchKernelGGL(cal_write1, dim3(grid), dim3(block), 0, 0, w1 + f * 5 * n, n, (uint8_t)it);
        hipLaunchKernelGGL(cal_write_rows80, dim3(grid), dim3(block), 0, 0, wrow + f * n * 5, n, (uint32_t)it);
        hipLaunchKernelGGL(cal_write_rec16, dim3(grid), dim3(block), 0, 0, wrec + f * n, n, (uint32_t)it);
        CHECK(hipMemsetAsync(junk, it & 0xff, (size_t)512 << 20, 0));        // evict: the step-like launch starts cold like the others' first touch
        const size_t g = (size_t)((it + frames / 2) % frames);
        StepLike a{r8 + g * 15 * n, r4 + g * 4 * n, rrow + g * n * 5, w8 + g * 6 * n, w4 + g * 3 * n, w1 + g * 5 * n, wrow + g * n * 5, wrec + g * n};
        hipLaunchKernelGGL(cal_step_like, dim3(grid), dim3(block), 0, 0, a, n, (uint32_t)it, sink);
    }
    CHECK(hipDeviceSynchronize());
    printf("n_agents %d launches %d\n", n, launches);
    printf("known_bytes cal_read16 read %zu write 0\n", big16 * 16);
    printf("known_bytes cal_read8 read %zu write 0\n", (size_t)15 * n * 8);
    printf("known_bytes cal_read4 read %zu write 0\n", (size_t)4 * n * 4);
    printf("known_bytes cal_read_rows80 read %zu write 0\n", (size_t)n * 80);
    printf("known_bytes cal_write8 read 0 write %zu\n", (size_t)6 * n * 8);
    printf("known_bytes cal_write4 read 0 write %zu\n", (size_t)3 * n * 4);
    printf("known_bytes cal_write1 read 0 write %zu\n", (size_t)5 * n);
    printf("known_bytes cal_write_rows80 read 0 write %zu\n", (size_t)n * 80);
    printf("known_bytes cal_write_rec16 read 0 write %zu\n", (size_t)n * 16);
    printf("known_bytes cal_step_like read %zu write %zu\n", (size_t)n * (15 * 8 + 4 * 4 + 80), (size_t)n * (6 * 8 + 3 * 4 + 5 + 80 + 16));
    return 0;
}
