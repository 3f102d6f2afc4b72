// dqn_internal.hpp -- entry points shared by the translation units of libuavenv.so that are NOT part of the C ABI
// (include/uavenv.h): the C loop (loop.hip) hands the DQN kernels a layer-1 image in the split form (qnet_device.hpp) that it owns
// and keeps current; every ABI entry point forwards here with a null image.
#pragma once
#include <stdint.h>

#include "../../include/uavenv.h"

extern "C" {

// Floats of ONE net's image (= qnet_device.hpp: kSplitF); an image buffer holds q_local's, then q_target's.
#define UAVENV_DQN_IMAGE_FLOATS 6912

// image_dev[0 .. 2 x UAVENV_DQN_IMAGE_FLOATS) <- fc1 / b1 of net->local and net->target in the split form (16-byte aligned).
int uavenv_dqn_split_image(const UavDqnNet *net, float *image_dev, void *stream);

// uavenv_dqn_grad_w with the image of THIS net as it is now (null: the kernels convert while staging -- same values).
int uavenv_dqn_grad_img(const UavReplayRing *ring, int32_t head, int32_t filled, int32_t batch, uint64_t seed, uint64_t counter,
                        const int32_t *explicit_idx, const UavDqnNet *net, int32_t kind, float gamma, int32_t huber,
                        const float *is_weights, float *abs_td_out, float *partials, const float *image_dev, void *stream);

// uavenv_dqn_reduce_adam_gated that also writes every fc1 / b1 value it steps into the image (and, on a hard update, into the
// target half): the image stays the split form of the parameters without a launch of its own.
int uavenv_dqn_reduce_adam_img(const UavDqnNet *net, const float *partials, int32_t n_partials, float lr, float beta1, float beta2,
                               float eps, int32_t step_t, int32_t hard_update, float *loss_out, float *raw_out,
                               const uint32_t *go_word, uint32_t go_value, float *image_dev, void *stream);

// ... and the Adam launches behind an exchange (csrc/p2p.hip, the RCCL path's uavenv_dqn_adam) do the same.
int uavenv_dqn_adam_p2p_img(const UavDqnNet *net, struct UavP2P *c, float lr, float beta1, float beta2, float eps, int32_t step_t,
                            int32_t hard_update, float *loss_out, float *raw_out, float *image_dev, void *stream);
int uavenv_dqn_adam_img(const UavDqnNet *net, const float *raw, float lr, float beta1, float beta2, float eps, int32_t step_t,
                        int32_t hard_update, float *loss_out, float *image_dev, void *stream);

// uavenv_step_policy with q_local's image (first half of an image buffer).
int uavenv_step_policy_img(UavEnv *e, const UavDqnNet *net, const void *obs_cur, float eps, uint64_t seed, uint64_t counter,
                           int32_t *action_out, void *obs, double *reward64, float *reward32, uint8_t *ret_done, uint8_t *agent_done,
                           uint8_t *info, uint8_t *valid, double *energy64, const uint8_t *active, uint32_t flags,
                           const float *image_dev, void *stream);

}  // extern "C"
