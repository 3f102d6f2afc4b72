// chain_floor.hip -- what the pieces of a 64-agents-per-CU latency chain cost on this part, measured in isolation.
// DIAGNOSTIC (not product code): the configs[1] step launch is 256 workgroups x 4 wavefronts, one wavefront per SIMD, so a
// workgroup's time is the SUM of its dependent round trips and instruction chains (nothing overlaps a wavefront's own waits).
// This program prices those pieces one at a time, in the launch geometry of k_step_coop (256 x 256 threads), each launch on
// planes a "writer" launch stored just before (as the previous pass's step kernel did):
//   fl_empty        nothing: dispatch of 256 workgroups + kernel end
//   fl_rt<K>        K DEPENDENT global round trips (the address of hop k + 1 comes out of hop k) + one store
//   fl_f64          one wavefront's f64 chain of update_PathPlan's first half: sincos, sqrt x 3, atan2 (no memory)
//   fl_lds<K>       K x (LDS store, workgroup barrier, LDS load)
//   fl_stage        27.6 KB + 8.5 KB from global memory into LDS by 256 threads + barrier (the policy image + the world blob)
// rocprofv3 --kernel-trace --stats gives each kernel's average duration; differences between fl_rt<K> price one round trip.
//   hipcc --offload-arch=gfx950 -O3 -o chain_floor chain_floor.hip ; ./chain_floor <launches>
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int kN = 16384;        // agents
constexpr int kPlanes = 8;

__global__ void fl_writer(int32_t *__restrict__ planes, double *__restrict__ fplanes, int it)
{
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= kN) return;
    // plane p holds, for agent i, the index of another agent of the SAME 64-agent block (so that a hop stays on the CU's lines'
    // neighbourhood like the state planes do) -- a different permutation per plane and per launch
#pragma unroll
    for (int p = 0; p < kPlanes; ++p) planes[(size_t)p * kN + i] = (i & ~63) | ((i * (2 * p + 5) + it + 7 * p) & 63);
#pragma unroll
    for (int p = 0; p < 4; ++p) fplanes[(size_t)p * kN + i] = 0.001 * i + p + 1e-3 * it;
}

__global__ void __launch_bounds__(256) fl_empty(int32_t *out) { if (out == nullptr && threadIdx.x == 999) out[0] = 1; }

template <int K>
__global__ void __launch_bounds__(256) fl_rt(const int32_t *__restrict__ planes, int32_t *__restrict__ out)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (wv != 0) return;                                   // wave 0 = the agents, as in k_step_coop
    int j = (int)blockIdx.x * 64 + lane;
#pragma unroll
    for (int k = 0; k < K; ++k) j = planes[(size_t)k * kN + j];          // hop k + 1 needs hop k's result
    out[(int)blockIdx.x * 64 + lane] = j;
}

__global__ void __launch_bounds__(256) fl_f64(const double *__restrict__ f, double *__restrict__ out, double steer)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (wv != 0) return;
    const int i = (int)blockIdx.x * 64 + lane;
    const double head = f[i], px = f[kN + i], py = f[2 * kN + i], gx = f[3 * kN + i];
    // step_pre: two distances, sincos of the new heading, the speed norm; heading_after: atan2 of the new velocity
    const double d0 = sqrt((px - gx) * (px - gx) + (py - 1.0) * (py - 1.0) + 0.0);
    const double d1 = sqrt((px - 3.0) * (px - 3.0) + (py - gx) * (py - gx) + 0.0);
    double sn, cs;
    sincos(head + steer, &sn, &cs);
    const double V = sqrt(cs * cs + sn * sn + 0.0);
    double ang = atan2(sn, cs) * (180.0 / 3.14159265358979323846);
    ang = fmod(ang + 360.0, 360.0) / 180.0 * 3.14159265358979323846;
    out[i] = d0 + d1 + V + ang;
}

template <int K>
__global__ void __launch_bounds__(256) fl_lds(double *__restrict__ out)
{
    __shared__ double sh[256];
    double v = (double)threadIdx.x;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        sh[threadIdx.x] = v;
        __syncthreads();
        v += sh[(threadIdx.x + 64) & 255];
        __syncthreads();
    }
    if (threadIdx.x < 64) out[(int)blockIdx.x * 64 + threadIdx.x] = v;
}

__global__ void __launch_bounds__(256) fl_stage(const uint4 *__restrict__ img, double *__restrict__ out)
{
    __shared__ uint4 sh[2304];                              // 36 864 B >= 27 648 (layer-1 image) + 8 704 (world blob)
    uint4 v[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) v[k] = img[k * 256 + threadIdx.x];        // every load in flight, then the stores (as img_issue / img_commit)
#pragma unroll
    for (int k = 0; k < 9; ++k) sh[k * 256 + threadIdx.x] = v[k];
    __syncthreads();
    if (threadIdx.x < 64) out[(int)blockIdx.x * 64 + threadIdx.x] = (double)sh[(threadIdx.x * 37) % 2304].x;
}

int main(int argc, char **argv)
{
    const int launches = argc > 1 ? atoi(argv[1]) : 200;
    int32_t *planes, *out; double *fplanes, *fout; uint4 *img;
    CHECK(hipMalloc(&planes, (size_t)kPlanes * kN * 4)); CHECK(hipMalloc(&out, kN * 4));
    CHECK(hipMalloc(&fplanes, (size_t)4 * kN * 8)); CHECK(hipMalloc(&fout, kN * 8));
    CHECK(hipMalloc(&img, 2304 * 16)); CHECK(hipMemset(img, 1, 2304 * 16));
    for (int it = 0; it < launches; ++it) {
        // (the writer runs in front of EVERY measured launch: its planes are what the previous pass's step kernel left behind --
        // written by 64 workgroups, read by 256: mostly another XCD's L2, i.e. through the fabric, like the loop's state planes)
#define FRESH() do { hipLaunchKernelGGL(fl_writer, dim3(64), dim3(256), 0, 0, planes, fplanes, it); } while (0)
        FRESH(); hipLaunchKernelGGL(fl_empty, dim3(256), dim3(256), 0, 0, out);
        FRESH(); hipLaunchKernelGGL(fl_rt<1>, dim3(256), dim3(256), 0, 0, planes, out);
        FRESH(); hipLaunchKernelGGL(fl_rt<2>, dim3(256), dim3(256), 0, 0, planes, out);
        FRESH(); hipLaunchKernelGGL(fl_rt<3>, dim3(256), dim3(256), 0, 0, planes, out);
        FRESH(); hipLaunchKernelGGL(fl_rt<5>, dim3(256), dim3(256), 0, 0, planes, out);
        FRESH(); hipLaunchKernelGGL(fl_rt<8>, dim3(256), dim3(256), 0, 0, planes, out);
        FRESH(); hipLaunchKernelGGL(fl_f64, dim3(256), dim3(256), 0, 0, fplanes, fout, 0.5235987755982988);
        FRESH(); hipLaunchKernelGGL(fl_lds<1>, dim3(256), dim3(256), 0, 0, fout);
        FRESH(); hipLaunchKernelGGL(fl_lds<5>, dim3(256), dim3(256), 0, 0, fout);
        FRESH(); hipLaunchKernelGGL(fl_stage, dim3(256), dim3(256), 0, 0, img, fout);
    }
    CHECK(hipDeviceSynchronize());
    printf("chain_floor: %d launches of each kernel, 256 workgroups x 256 threads, %d agents\n", launches, kN);
    return 0;
}
