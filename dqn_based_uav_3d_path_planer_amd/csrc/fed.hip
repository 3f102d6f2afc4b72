// fed.hip -- the federated merge of the per-UAV trainers (Envs/PathPlan_City.py:469-475, :590-601) on the device.
//
// Federated_Learning_AC deep-copies agent 0's actor, adds every other agent's tensors to it in agent order, and hands the
// result to every agent through replace_param (Trainer/SAC_Trainer.py:456-459).  With one flat f32 parameter block per
// trainer (sac.py: FusedSACLearner._blocks[0]; learner.py: FusedDQNLearner.flat[0]) that is one pass over U blocks: each
// thread owns four consecutive floats, reads them from every block (independent 16-byte loads, all in flight together),
// adds them in agent order -- ((w_0 + w_1) + w_2) + ..., the reference's order, so the f32 sums are bit-identical to
// torch's -- scales, and stores the result into every block.  HBM-bound: 2 x U x n x 4 bytes per call (U = 4 actors of
// 6 724 floats: 215 KB), one launch, no atomics.
//
// `scale`: the executed reference never applies its division (the `state_dict()[k] = torch.div(...)` of :597 assigns into a
// temporary dict; tests/golden/federated_ac.npz, oracle/gen_golden_federated.py), so its merge is the SUM: scale = 1 is the
// reference as executed, scale = 1 / U the mean its comment ("local平均") intends.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/uavenv.h"

namespace {

struct FedArgs {
    float *block[UAVENV_FED_MAX_BLOCKS];
    int32_t n_blocks;
    int32_t n_floats;
    float scale;
};

__global__ __launch_bounds__(256) void k_fed_aggregate(FedArgs a)
{
    const int i4 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (i4 >= a.n_floats) return;
    if (i4 + 4 <= a.n_floats) {
        float4 v[UAVENV_FED_MAX_BLOCKS];
#pragma unroll
        for (int j = 0; j < UAVENV_FED_MAX_BLOCKS; ++j)
            if (j < a.n_blocks) v[j] = *reinterpret_cast<const float4 *>(a.block[j] + i4);
        float4 s = v[0];
#pragma unroll
        for (int j = 1; j < UAVENV_FED_MAX_BLOCKS; ++j)
            if (j < a.n_blocks) { s.x += v[j].x; s.y += v[j].y; s.z += v[j].z; s.w += v[j].w; }
        if (a.scale != 1.0f) { s.x *= a.scale; s.y *= a.scale; s.z *= a.scale; s.w *= a.scale; }
#pragma unroll
        for (int j = 0; j < UAVENV_FED_MAX_BLOCKS; ++j)
            if (j < a.n_blocks) *reinterpret_cast<float4 *>(a.block[j] + i4) = s;
    } else {
        for (int i = i4; i < a.n_floats; ++i) {
            float s = a.block[0][i];
            for (int j = 1; j < a.n_blocks; ++j) s += a.block[j][i];
            if (a.scale != 1.0f) s *= a.scale;
            for (int j = 0; j < a.n_blocks; ++j) a.block[j][i] = s;
        }
    }
}

}  // namespace

extern "C" int uavenv_fed_aggregate(float *const *blocks_dev, int32_t n_blocks, int32_t n_floats, float scale, void *stream)
{
    if (!blocks_dev || n_blocks <= 0 || n_blocks > UAVENV_FED_MAX_BLOCKS || n_floats <= 0) return UAVENV_EINVAL;
    FedArgs a;
    for (int j = 0; j < UAVENV_FED_MAX_BLOCKS; ++j) a.block[j] = nullptr;
    for (int j = 0; j < n_blocks; ++j) {
        if (!blocks_dev[j] || ((uintptr_t)blocks_dev[j] & 15u) != 0) return UAVENV_EINVAL;
        for (int k = 0; k < j; ++k)                               // two trainers sharing a block would race with themselves
            if (blocks_dev[k] == blocks_dev[j]) return UAVENV_EINVAL;
        a.block[j] = blocks_dev[j];
    }
    a.n_blocks = n_blocks;
    a.n_floats = n_floats;
    a.scale = scale;
    const int threads = (n_floats + 3) / 4;
    hipLaunchKernelGGL(k_fed_aggregate, dim3((threads + 255) / 256), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? UAVENV_OK : UAVENV_EHIP;
}
