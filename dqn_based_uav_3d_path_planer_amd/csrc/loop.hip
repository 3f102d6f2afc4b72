// loop.hip -- the off-policy loop of PathPlan_City.run_thread_OffPolicy (Envs/PathPlan_City.py:364-385), all envs of
// the shard at once, enqueued from C: act -> step (+ replay write) -> learn, K times per call.
//
// Why this exists: driven from Python through four ctypes calls per step the host needed ~33 us to enqueue a step
// whose kernels take ~45 us; any kernel improvement beyond that made the loop host-bound.  From C the same four
// launches cost the HIP runtime's ~3 us each and nothing else.  Every launch goes through the library's own extern "C"
// entry points (same validation, same kernels), so a loop of K steps is bit-identical to K rounds of
// uavenv_dqn_act / uavenv_step / uavenv_dqn_grad / uavenv_dqn_reduce_adam issued by the caller.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <math.h>
#include <new>
#include <vector>

#include "../../include/uavenv.h"
#include "dqn_internal.hpp"

#include <chrono>
#include <stdio.h>

struct UavLoop {
    double t_host[4] = {0, 0, 0, 0};   // UAVENV_LOOP_PROFILE: host seconds spent in the act/step, grad, adam calls, calls
    bool prof = false;
    UavLoopConfig c;
    int32_t head, filled, epoch;
    uint64_t counter;
    size_t obs_row_bytes;
    bool fuse_act = true;            // act in the step kernel's prologue (uavenv_step_policy) until it says it cannot
    bool per = false;                // prioritised replay on (c.per.prio != NULL)
    double per_beta = 0.4;
    // sample_lag = 1 (experiment): the gradient kernel of pass t runs on a second stream beside the step kernel of pass t
    hipStream_t aux = nullptr;
    hipEvent_t ev_grad = nullptr, ev_join = nullptr;
    std::vector<hipEvent_t> ev;      // pairs (start, stop), recorded so far
    std::vector<hipEvent_t> pool;    // idle events
    // rolling refresh of the reset bank: the planner runs on a low-priority stream beside the passes
    hipStream_t plan = nullptr;
    int32_t replan_next = 0, bank_m = 0;
    uint64_t replan_gen = 0;
    // fc1 / b1 of q_local and q_target in the split form (csrc/dqn_internal.hpp, qnet_device.hpp): built from the parameters when a
    // run starts, kept current by the loop's own Adam launches, read by its step-policy and gradient launches instead of converting
    // fc1 while staging it (packed ring + f32 MFMA net, the serial order; null otherwise).  Behind an exchange every rank's Adam
    // launch applies the same summed gradient, so every rank's image follows its own -- identical -- parameters
    float *img = nullptr;
};

// Every replan_every passes: hand over the slice whose planning has finished (on the passes' stream: no reset runs meanwhile),
// then start planning the next slice.  A slice still being planned is left alone until the next visit.
static int replan_tick(UavLoop *l, hipStream_t s)
{
    const UavLoopConfig &c = l->c;
    const int ready = uavenv_replan_ready(c.env);
    if (ready == 0) return UAVENV_OK;
    if (ready < -1) return ready;
    if (ready == 1) {
        const int rc = uavenv_replan_commit(c.env, 0, s);
        if (rc != UAVENV_OK) return rc;
    }
    int32_t count = c.replan_count < l->bank_m ? c.replan_count : l->bank_m;
    if (l->replan_next + count > l->bank_m) l->replan_next = 0;
    l->replan_gen += 1;
    const int rc = uavenv_replan_begin(c.env, l->replan_next, count, c.seed * 0x9E3779B97F4A7C15ull + l->replan_gen,
                                       c.replan_max_iter > 0 ? c.replan_max_iter : 10000, l->plan);
    if (rc != UAVENV_OK) return rc;
    l->replan_next += count;
    return UAVENV_OK;
}

extern "C" {

int uavenv_loop_create(const UavLoopConfig *cfg, UavLoop **out)
{
    if (!cfg || !out || !cfg->env || !cfg->ring.obs || !cfg->ring.action || !cfg->ring.reward || !cfg->ring.done ||
        !cfg->net.local)
        return UAVENV_EINVAL;
    if (!cfg->ring.action_is_index || cfg->ring.frames < 3 || cfg->ring.n_agents != uavenv_num_agents(cfg->env))
        return UAVENV_EINVAL;
    if (cfg->head < 0 || cfg->head >= cfg->ring.frames || cfg->filled < 0 || cfg->filled > cfg->ring.frames - 1)
        return UAVENV_EINVAL;
    if (cfg->batch < 0 || cfg->batch % 64 != 0 || (cfg->batch > 0 && (!cfg->partials_dev || !cfg->net.target ||
                                                                        !cfg->net.m || !cfg->net.v)))
        return UAVENV_EINVAL;
    if (cfg->update_loop <= 0 || cfg->epoch < 0) return UAVENV_EINVAL;
    if (!cfg->p2p && cfg->coll && !cfg->raw_dev) return UAVENV_EINVAL;
    if (cfg->per.prio) {
        if (cfg->per.capacity != (int64_t)cfg->ring.frames * cfg->ring.n_agents || cfg->batch <= 0 || !cfg->per_slots_dev ||
            !cfg->per_prio_dev || !cfg->per_w_dev || !cfg->per_abs_dev || !cfg->per_idx_dev || !cfg->ring.valid)
            return UAVENV_EINVAL;
    }
    UavLoop *l = new (std::nothrow) UavLoop();
    if (!l) return UAVENV_ENOMEM;
    l->c = *cfg;
    l->head = cfg->head;
    l->filled = cfg->filled;
    l->epoch = cfg->epoch;
    l->counter = cfg->counter;
    l->per = cfg->per.prio != nullptr;
    l->per_beta = cfg->per_beta;
    {   // the layer-1 image (see UavLoop.img): the serial order, packed rows, the f32-MFMA net of the reference's shape
        static const bool off = getenv("UAVENV_LOOP_IMAGE") && atoi(getenv("UAVENV_LOOP_IMAGE")) == 0;     // A/B knob
        if (!off && cfg->sample_lag == 0 && cfg->ring.obs_dtype == UAVENV_OBS_PACKED &&
            cfg->net.mfma_dtype == UAVENV_MFMA_F32 && cfg->net.w == 100 && cfg->net.hid == 64 && cfg->net.local && cfg->net.target) {
            if (hipMalloc((void **)&l->img, 2 * (size_t)UAVENV_DQN_IMAGE_FLOATS * sizeof(float)) != hipSuccess) { delete l; return UAVENV_ENOMEM; }
        }
    }
    if (cfg->sample_lag != 0) {
        if (cfg->sample_lag != 1 || l->per) { delete l; return UAVENV_EINVAL; }
        if (hipStreamCreateWithFlags(&l->aux, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&l->ev_grad, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&l->ev_join, hipEventDisableTiming) != hipSuccess) {
            uavenv_loop_destroy(l);
            return UAVENV_EHIP;
        }
    }
    if (cfg->moved_dev) {
        if (cfg->p2p || cfg->coll || cfg->sample_lag != 0) { uavenv_loop_destroy(l); return UAVENV_EINVAL; }
        if (uavenv_set_moved_word(cfg->env, cfg->moved_dev) != UAVENV_OK) { uavenv_loop_destroy(l); return UAVENV_EINVAL; }
    }
    if (cfg->replan_every > 0) {
        int32_t m = 0;
        int lo = 0, hi = 0;
        if (cfg->replan_count <= 0 || uavenv_bank_stats(cfg->env, &m, nullptr) != UAVENV_OK || m <= 0) { uavenv_loop_destroy(l); return UAVENV_EINVAL; }
        l->bank_m = m;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);          // lo = the numerically largest value = the LOWEST priority
        if (hipStreamCreateWithPriority(&l->plan, hipStreamNonBlocking, lo) != hipSuccess) { uavenv_loop_destroy(l); return UAVENV_EHIP; }
    }
    l->fuse_act = getenv("UAVENV_NO_FUSED_ACT") == nullptr;
    l->prof = getenv("UAVENV_LOOP_PROFILE") != nullptr;
    l->obs_row_bytes = (size_t)cfg->ring.n_agents * (cfg->ring.obs_dtype == UAVENV_OBS_PACKED ? UAVENV_OBS_PACKED_DWORDS * 4
                                                     : UAVENV_OBS_DIM * (cfg->ring.obs_dtype == UAVENV_OBS_F16 ? 2 : 4));
    *out = l;
    return UAVENV_OK;
}

int uavenv_loop_destroy(UavLoop *l)
{
    if (!l) return UAVENV_OK;
    if (l->prof && l->t_host[3] > 0)
        fprintf(stderr, "uavenv_loop host us per pass: act+step %.2f grad %.2f adam %.2f (%.0f passes)\n", 1e6 * l->t_host[0] / l->t_host[3],
                1e6 * l->t_host[1] / l->t_host[3], 1e6 * l->t_host[2] / l->t_host[3], l->t_host[3]);
    if (l->c.moved_dev) (void)uavenv_set_moved_word(l->c.env, nullptr);
    for (hipEvent_t e : l->ev) (void)hipEventDestroy(e);
    for (hipEvent_t e : l->pool) (void)hipEventDestroy(e);
    if (l->aux) { (void)hipStreamSynchronize(l->aux); (void)hipStreamDestroy(l->aux); }
    if (l->plan) {                   // a slice still in flight is finished and dropped with the env's next begin / destroy
        (void)hipStreamSynchronize(l->plan);
        (void)hipStreamDestroy(l->plan);
    }
    for (hipEvent_t e : {l->ev_grad, l->ev_join}) if (e) (void)hipEventDestroy(e);
    if (l->img) (void)hipFree(l->img);
    delete l;
    return UAVENV_OK;
}

int uavenv_loop_set_eps(UavLoop *l, float eps)
{
    if (!l) return UAVENV_EINVAL;
    l->c.eps = eps;
    return UAVENV_OK;
}

int uavenv_loop_get_per(const UavLoop *l, double *beta_out)
{
    if (!l || !beta_out) return UAVENV_EINVAL;
    *beta_out = l->per_beta;
    return UAVENV_OK;
}

int uavenv_loop_get(const UavLoop *l, UavLoopCursor *out)
{
    if (!l || !out) return UAVENV_EINVAL;
    out->head = l->head;
    out->filled = l->filled;
    out->epoch = l->epoch;
    out->reserved0 = 0;
    out->counter = l->counter;
    return UAVENV_OK;
}

static hipEvent_t take_event(UavLoop *l)
{
    hipEvent_t e = nullptr;
    if (!l->pool.empty()) {
        e = l->pool.back();
        l->pool.pop_back();
    } else if (hipEventCreate(&e) != hipSuccess) {
        e = nullptr;
    }
    return e;
}

int uavenv_loop_run(UavLoop *l, int32_t n_steps, void *stream)
{
    if (!l || n_steps < 0) return UAVENV_EINVAL;
    const UavLoopConfig &c = l->c;
    const UavReplayRing &R = c.ring;
    const size_t n = (size_t)R.n_agents;
    hipStream_t s = (hipStream_t)stream;
    if (c.p2p) {                              // the exchange's sticky error: stop enqueueing on frozen weights
        int32_t st[4];
        if (uavenv_p2p_status(c.p2p, 0, st) == UAVENV_OK && st[0] != 0) return UAVENV_EP2P;
    }
    const bool lag = l->aux != nullptr;
    if (l->img && n_steps > 0) {              // whatever happened to the parameters since the last run: the image is rebuilt from them
        const int ri = uavenv_dqn_split_image(&c.net, l->img, s);
        if (ri != UAVENV_OK) return ri;
    }
    for (int k = 0; k < n_steps; ++k) {
        const int t = l->head, nxt = t + 1 == R.frames ? 0 : t + 1;
        if (l->plan && l->counter % (uint64_t)c.replan_every == 0) {
            const int rr = replan_tick(l, s);
            if (rr != UAVENV_OK) return rr;
        }
        bool lag_update = false;
        if (lag && c.batch > 0 && l->filled > 0 &&
            (int64_t)l->filled * (int64_t)n >= (int64_t)(c.learn_start > c.batch ? c.learn_start : c.batch)) {
            // update t samples the transitions stored BEFORE step t (frames <= t - 1: the cursor as it stands), so its
            // gradient depends on adam(t - 1) only, like step t -- the two run side by side
            // (the second stream starts behind the main stream's tail: Adam t - 1, or step t - 1 when that pass had no update)
            if (hipEventRecord(l->ev_join, s) != hipSuccess || hipStreamWaitEvent(l->aux, l->ev_join, 0) != hipSuccess) return UAVENV_EHIP;
            // (a full ring's oldest frame is the one step t is overwriting: it is left out -- one frame less replay in this mode)
            const int filled_lag = l->filled < R.frames - 2 ? l->filled : R.frames - 2;
            int rcg = uavenv_dqn_grad(&R, l->head, filled_lag, c.batch, c.seed, l->counter, nullptr, &c.net, c.kind, c.gamma,
                                      c.huber, c.partials_dev, l->aux);
            if (rcg != UAVENV_OK) return rcg;
            if (hipEventRecord(l->ev_grad, l->aux) != hipSuccess) return UAVENV_EHIP;
            lag_update = true;
        }
        unsigned char *obs_t = (unsigned char *)R.obs + (size_t)t * l->obs_row_bytes;
        unsigned char *obs_n = (unsigned char *)R.obs + (size_t)nxt * l->obs_row_bytes;
        int32_t *act_t = (int32_t *)R.action + (size_t)t * n;
        const bool timed = c.time_every > 0 && (l->counter % (uint64_t)c.time_every) == 0;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        int rc = UAVENV_EINVAL;
        std::chrono::steady_clock::time_point tp0, tp1, tp2, tp3;
        if (l->prof) tp0 = std::chrono::steady_clock::now();
        uint8_t *info_t = c.info_dev ? c.info_dev + (size_t)t * n : nullptr;
        uint8_t *valid_t = R.valid ? R.valid + (size_t)t * n : nullptr;
        // the transition records of this frame (UavReplayRing.meta): the learner then gathers one record per sample
        void *meta_t = R.meta ? (unsigned char *)R.meta + (size_t)t * n * UAVENV_META_BYTES : nullptr;
        if (l->fuse_act) {               // Q(s) + epsilon-greedy in the prologue of the step kernel: one launch less
            if (timed) {
                e0 = take_event(l);
                e1 = take_event(l);
                if (e0 && e1) (void)hipEventRecord(e0, s);
            }
            rc = uavenv_set_step_meta(c.env, meta_t, nullptr);
            if (rc != UAVENV_OK) return rc;
            rc = uavenv_step_policy_img(c.env, &c.net, obs_t, c.eps, c.seed, l->counter, act_t, obs_n, nullptr, R.reward + (size_t)t * n,
                                        R.done + (size_t)t * n, nullptr, info_t, valid_t, nullptr, nullptr, c.step_flags, l->img, s);
            if (rc == UAVENV_EINVAL) l->fuse_act = false;         // not this env / net: the two-launch form from here on
            else if (rc != UAVENV_OK) return rc;
        }
        if (!l->fuse_act) {
            rc = uavenv_dqn_act(&c.net, obs_t, R.obs_dtype, R.n_agents, c.eps, c.seed, l->counter, act_t, nullptr, nullptr, s);
            if (rc != UAVENV_OK) return rc;
            if (timed && !e0) {
                e0 = take_event(l);
                e1 = take_event(l);
                if (e0 && e1) (void)hipEventRecord(e0, s);
            }
            rc = uavenv_set_step_meta(c.env, meta_t, nullptr);
            if (rc != UAVENV_OK) return rc;
            rc = uavenv_step(c.env, act_t, UAVENV_ACT_INDEX_I32, obs_n, nullptr, R.reward + (size_t)t * n, R.done + (size_t)t * n,
                             nullptr, info_t, valid_t, nullptr, nullptr, c.step_flags, s);
        }
        if (timed && e0 && e1) {
            (void)hipEventRecord(e1, s);
            l->ev.push_back(e0);
            l->ev.push_back(e1);
        }
        if (rc != UAVENV_OK) return rc;
        if (l->prof) tp1 = std::chrono::steady_clock::now();
        l->head = nxt;
        if (l->filled < R.frames - 1) l->filled += 1;
        const bool learn_now = !lag && c.batch > 0 &&
                               (int64_t)l->filled * (int64_t)n >= (int64_t)(c.learn_start > c.batch ? c.learn_start : c.batch);
        const double per_new = l->per ? pow(0.0 + (double)c.per_eps, (double)c.per_alpha) : 0.0;
        if (l->per && !learn_now) {       // ReplayTree.push(error 0) for the frame just written; the new head's rows are retired
            rc = uavenv_per_fill_frame(&c.per, (int64_t)t * (int64_t)n, (int64_t)n, per_new, R.valid + (size_t)t * n,
                                       (int64_t)nxt * (int64_t)n, s);           // (an update's rebuild applies them itself)
            if (rc != UAVENV_OK) return rc;
        }
        if (lag ? lag_update
                : (c.batch > 0 && (int64_t)l->filled * (int64_t)n >= (int64_t)(c.learn_start > c.batch ? c.learn_start : c.batch))) {
            if (lag) {                    // the gradient is already in flight on the second stream: join it in front of Adam
                if (hipStreamWaitEvent(s, l->ev_grad, 0) != hipSuccess) return UAVENV_EHIP;
                rc = UAVENV_OK;
            } else if (l->per) {          // ReplayTree.sample: beta first (:155), selection, importance weights
                l->per_beta = l->per_beta + (double)c.per_beta_inc < 1.0 ? l->per_beta + (double)c.per_beta_inc : 1.0;
                rc = uavenv_per_rebuild_frame(&c.per, (int64_t)t * (int64_t)n, (int64_t)n, per_new, R.valid + (size_t)t * n,
                                              (int64_t)nxt * (int64_t)n, s);
                if (rc != UAVENV_OK) return rc;
                rc = uavenv_per_sample(&c.per, c.batch, nullptr, c.seed, l->counter, c.per_slots_dev, c.per_prio_dev, s);
                if (rc != UAVENV_OK) return rc;
                rc = uavenv_per_weights(&c.per, c.per_slots_dev, c.per_prio_dev, c.batch, (int64_t)l->filled * (int64_t)n,
                                        l->per_beta, R.n_agents, c.per_w_dev, c.per_idx_dev, s);
                if (rc != UAVENV_OK) return rc;
                rc = uavenv_dqn_grad_img(&R, l->head, l->filled, c.batch, c.seed, l->counter, c.per_idx_dev, &c.net, c.kind, c.gamma,
                                         c.huber, c.per_w_dev, c.per_abs_dev, c.partials_dev, l->img, s);
            } else if ((c.step_flags & UAVENV_STEP_SKIP_DONE) != 0 && c.per_idx_dev && R.valid) {
                // finished agents are skipped, not restarted: their rows stay in the ring with valid = 0 (the reference stores
                // nothing for them) -- the batch is drawn over the valid rows only and handed to the update as explicit pairs
                rc = uavenv_replay_draw_valid(R.frames, n, l->head, l->filled, c.batch, 1, 1, 0, R.valid, UAVENV_DRAW_MAX_TRIES, c.seed,
                                              l->counter, c.per_idx_dev, s);
                if (rc != UAVENV_OK) return rc;
                rc = uavenv_dqn_grad_img(&R, l->head, l->filled, c.batch, c.seed, l->counter, c.per_idx_dev, &c.net, c.kind, c.gamma,
                                         c.huber, nullptr, nullptr, c.partials_dev, l->img, s);
            } else {
                rc = uavenv_dqn_grad_img(&R, l->head, l->filled, c.batch, c.seed, l->counter, nullptr, &c.net, c.kind, c.gamma,
                                         c.huber, nullptr, nullptr, c.partials_dev, l->img, s);
            }
            if (rc != UAVENV_OK) return rc;
            if (l->prof) tp2 = std::chrono::steady_clock::now();
            l->epoch += 1;
            const int hard = l->epoch % c.update_loop == 0 ? 1 : 0;
            if (c.p2p) {                      // multi-GPU: every rank's column sums to every rank, then the same Adam step
                rc = uavenv_dqn_reduce_p2p(&c.net, c.partials_dev, uavenv_dqn_partial_rows(c.batch), c.p2p, s);
                if (rc != UAVENV_OK) {        // (EP2P: nothing was enqueued for this update; the step itself is in the ring)
                    l->epoch -= 1;
                    l->counter += 1;
                    return rc;
                }
                rc = uavenv_dqn_adam_p2p_img(&c.net, c.p2p, c.lr, c.beta1, c.beta2, c.adam_eps, l->epoch, hard, c.loss_dev, nullptr, l->img, s);
                if (rc == UAVENV_EP2P) l->counter += 1;   // enqueued, but the kernel keeps the weights frozen
            } else if (c.coll) {              // the same bucket through an RCCL all-reduce enqueued from here (csrc/coll.hip)
                rc = uavenv_dqn_reduce(&c.net, c.partials_dev, uavenv_dqn_partial_rows(c.batch), c.raw_dev, s);
                if (rc != UAVENV_OK) return rc;
                rc = uavenv_coll_allreduce_sum(c.coll, c.raw_dev, (int64_t)uavenv_dqn_num_params(&c.net) + 2, s);
                if (rc != UAVENV_OK) return rc;
                rc = uavenv_dqn_adam_img(&c.net, c.raw_dev, c.lr, c.beta1, c.beta2, c.adam_eps, l->epoch, hard, c.loss_dev, l->img, s);
            } else {
                // (moved_dev: no update behind a step that moved nobody -- the word was stamped with this pass's tick by the step)
                rc = uavenv_dqn_reduce_adam_img(&c.net, c.partials_dev, uavenv_dqn_partial_rows(c.batch), c.lr, c.beta1, c.beta2,
                                                c.adam_eps, l->epoch, hard, c.loss_dev, nullptr, c.moved_dev,
                                                (uint32_t)uavenv_tick(c.env), l->img, s);
            }
            if (rc != UAVENV_OK) return rc;
            if (l->per) {                 // ReplayTree.batch_update (:215-222)
                rc = uavenv_per_set_f32_gated(&c.per, c.per_slots_dev, c.per_abs_dev, c.batch, (double)c.per_eps, (double)c.per_alpha,
                                              (double)c.per_clip, c.moved_dev, (uint32_t)uavenv_tick(c.env), s);
                if (rc != UAVENV_OK) return rc;
            }
            if (l->prof) {
                tp3 = std::chrono::steady_clock::now();
                l->t_host[0] += std::chrono::duration<double>(tp1 - tp0).count();
                l->t_host[1] += std::chrono::duration<double>(tp2 - tp1).count();
                l->t_host[2] += std::chrono::duration<double>(tp3 - tp2).count();
                l->t_host[3] += 1.0;
            }
        }
        l->counter += 1;
    }
    return UAVENV_OK;
}

int uavenv_loop_step_times(UavLoop *l, float *ms_out, int32_t max_n, int32_t *n_out)
{
    if (!l || !n_out || (max_n > 0 && !ms_out)) return UAVENV_EINVAL;
    int n = 0;
    for (size_t i = 0; i + 1 < l->ev.size(); i += 2) {
        if (hipEventSynchronize(l->ev[i + 1]) != hipSuccess) return UAVENV_EHIP;
        float ms = 0.0f;
        if (n < max_n && hipEventElapsedTime(&ms, l->ev[i], l->ev[i + 1]) == hipSuccess) ms_out[n++] = ms;
        l->pool.push_back(l->ev[i]);
        l->pool.push_back(l->ev[i + 1]);
    }
    l->ev.clear();
    *n_out = n;
    return UAVENV_OK;
}

}  // extern "C"

// =====================================================================================================================
// The same loop for SAC_Trainer with continuous actions, one trainer per UAV slot (Envs/PathPlan_City.py:59-69, :364-385,
// BASELINE configs[3]'s shape): per time step
//     one launch of N(0,1) draws for every rsample() of the step (the U get_action's, the 2 U rsample()'s of the updates)
//     U x uavenv_sac_act            get_action (SAC_Trainer.py:444-448) of slot j's agents from the packed rows of frame t
//     uavenv_step                   Move_Agent for every agent (+ k_apf_adjust in front with APF on), replay write included
//     uavenv_replay_draw            distinct (frame, env) pairs for all slots at once (each slot reads ITS rows of them)
//     U x (critic_grad, critic_adam, actor_grad, actor_adam)      SAC_Trainer.update (:325-379)
// Driven from Python this was ~24 ctypes / torch launches per step: 389 us per step at 2 048 envs x 4 UAVs, host-bound.
// Every launch goes through the library's own entry points, in the order and with the arguments the Python loop of
// plugins/PathPlan_City._run_eposide_fused_sac uses: K steps from here == K steps from there, bit for bit.
// =====================================================================================================================
#include "uavenv_device.hpp"
#include "dqn_internal.hpp"

namespace {

// out[i] ~ N(0, 1), i < n: Philox4x32-10 keyed by seed, counter (step, block of four), Box-Muller on two uniforms each.
__global__ void __launch_bounds__(256) k_randn(uint64_t seed, uint64_t counter, int64_t n, float *__restrict__ out)
{
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (4 * q >= n) return;
    const uint4 r = uav::philox4x32_10(make_uint4((uint32_t)q, (uint32_t)(q >> 32), (uint32_t)counter, (uint32_t)(counter >> 32) ^ 0x6a55u),
                                       make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
    const float u1 = ((float)(r.x >> 8) + 1.0f) * (1.0f / 16777216.0f);      // (0, 1]
    const float u2 = (float)(r.y >> 8) * (1.0f / 16777216.0f);              // [0, 1)
    const float u3 = ((float)(r.z >> 8) + 1.0f) * (1.0f / 16777216.0f);
    const float u4 = (float)(r.w >> 8) * (1.0f / 16777216.0f);
    const float ra = sqrtf(-2.0f * logf(u1)), rb = sqrtf(-2.0f * logf(u3));
    float sa, ca, sb, cb;
    sincosf(6.283185307179586f * u2, &sa, &ca);
    sincosf(6.283185307179586f * u4, &sb, &cb);
    const float v[4] = {ra * ca, ra * sa, rb * cb, rb * sb};
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (4 * q + e < n) out[4 * q + e] = v[e];
}

}  // namespace

struct UavSacLoop {
    UavSacLoopConfig c;
    int32_t head, filled;
    int32_t epoch[UAVENV_SAC_LOOP_MAX_SLOTS], adam_steps[UAVENV_SAC_LOOP_MAX_SLOTS];
    uint64_t counter;
    size_t obs_row_bytes;
    bool per = false;                          // prioritised replay: every slot carries a UavPer
    double per_beta[UAVENV_SAC_LOOP_MAX_SLOTS];
};

extern "C" {

int uavenv_randn(uint64_t seed, uint64_t counter, int64_t n, float *out_dev, void *stream)
{
    if (!out_dev || n < 0) return UAVENV_EINVAL;
    if (n == 0) return UAVENV_OK;
    const int64_t quads = (n + 3) / 4;
    hipLaunchKernelGGL(k_randn, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, seed, counter, n, out_dev);
    return hipGetLastError() == hipSuccess ? UAVENV_OK : UAVENV_EHIP;
}

int64_t uavenv_sac_loop_noise_floats(int32_t n_slots, int32_t n_envs, int32_t batch)
{
    return (int64_t)n_slots * 2 * n_envs + 4 * (int64_t)n_slots * batch;
}

int uavenv_sac_loop_create(const UavSacLoopConfig *cfg, UavSacLoop **out)
{
    if (!cfg || !out || !cfg->env || !cfg->ring.obs || !cfg->ring.action || !cfg->ring.reward || !cfg->ring.done || !cfg->ring.valid)
        return UAVENV_EINVAL;
    if (cfg->ring.action_is_index || cfg->ring.obs_dtype != UAVENV_OBS_PACKED || cfg->ring.frames < 3 ||
        cfg->ring.n_agents != uavenv_num_agents(cfg->env))
        return UAVENV_EINVAL;
    if (cfg->n_slots < 1 || cfg->n_slots > UAVENV_SAC_LOOP_MAX_SLOTS || cfg->ring.n_agents % cfg->n_slots != 0) return UAVENV_EINVAL;
    if (cfg->batch <= 0 || cfg->batch % 64 != 0 || !cfg->act1_plane || !cfg->draws_dev || !cfg->noise_dev) return UAVENV_EINVAL;
    if (cfg->head < 0 || cfg->head >= cfg->ring.frames || cfg->filled < 0 || cfg->filled > cfg->ring.frames - 1) return UAVENV_EINVAL;
    for (int j = 0; j < cfg->n_slots; ++j) {
        const UavSacLoopSlot &sl = cfg->slot[j];
        if (!sl.nets.actor || !sl.nets.critic1 || !sl.nets.critic2 || !sl.nets.target1 || !sl.nets.target2 || !sl.nets.log_alpha ||
            !sl.m_actor || !sl.v_actor || !sl.alpha_mv || !sl.m1 || !sl.v1 || !sl.m2 || !sl.v2 || !sl.scalars || !sl.partials_critic ||
            !sl.partials_actor || sl.epoch < 0 ||
            sl.adam_steps < 0)
            return UAVENV_EINVAL;
    }
    if ((cfg->p2p || cfg->coll) && (!cfg->xbuf_dev || (((uintptr_t)cfg->xbuf_dev) & 15u) != 0)) return UAVENV_EINVAL;
    int n_per = 0;
    for (int j = 0; j < cfg->n_slots; ++j) {
        const UavSacLoopSlot &sl = cfg->slot[j];
        if (!sl.per.prio) continue;
        n_per += 1;
        const int64_t envs = cfg->ring.n_agents / cfg->n_slots;
        if (sl.per.capacity != (int64_t)cfg->ring.frames * envs || sl.per.rot != 0 || !sl.per_slots_dev || !sl.per_prio_dev || !sl.per_w_dev ||
            !sl.per_abs_dev)
            return UAVENV_EINVAL;
    }
    if (n_per != 0 && n_per != cfg->n_slots) return UAVENV_EINVAL;          // one Trainer.xml: all slots or none
    if (n_per && (cfg->p2p || cfg->coll)) return UAVENV_EINVAL;             // (prioritised replay is a one-GPU path here)
    if (cfg->moved_dev && (cfg->p2p || cfg->coll)) return UAVENV_EINVAL;    // (so is the gated update)
    if (cfg->moved_dev && uavenv_set_moved_word(cfg->env, cfg->moved_dev) != UAVENV_OK) return UAVENV_EINVAL;
    UavSacLoop *l = new (std::nothrow) UavSacLoop();
    if (!l) return UAVENV_ENOMEM;
    l->c = *cfg;
    l->per = n_per != 0;
    for (int j = 0; j < UAVENV_SAC_LOOP_MAX_SLOTS; ++j) l->per_beta[j] = j < cfg->n_slots ? cfg->slot[j].per_beta : 0.0;
    l->head = cfg->head;
    l->filled = cfg->filled;
    l->counter = cfg->counter;
    for (int j = 0; j < cfg->n_slots; ++j) { l->epoch[j] = cfg->slot[j].epoch; l->adam_steps[j] = cfg->slot[j].adam_steps; }
    l->obs_row_bytes = (size_t)cfg->ring.n_agents * UAVENV_OBS_PACKED_DWORDS * 4;
    *out = l;
    return UAVENV_OK;
}

int uavenv_sac_loop_destroy(UavSacLoop *l)
{
    if (l && l->c.moved_dev) (void)uavenv_set_moved_word(l->c.env, nullptr);
    delete l;
    return UAVENV_OK;
}

int uavenv_sac_loop_get_per(const UavSacLoop *l, double *beta_out)
{
    if (!l || !beta_out) return UAVENV_EINVAL;
    for (int j = 0; j < l->c.n_slots; ++j) beta_out[j] = l->per_beta[j];
    return UAVENV_OK;
}

int uavenv_sac_loop_get(const UavSacLoop *l, UavSacLoopCursor *out)
{
    if (!l || !out) return UAVENV_EINVAL;
    out->head = l->head;
    out->filled = l->filled;
    out->counter = l->counter;
    for (int j = 0; j < UAVENV_SAC_LOOP_MAX_SLOTS; ++j) {
        out->epoch[j] = j < l->c.n_slots ? l->epoch[j] : 0;
        out->adam_steps[j] = j < l->c.n_slots ? l->adam_steps[j] : 0;
    }
    return UAVENV_OK;
}

int uavenv_sac_loop_run(UavSacLoop *l, int32_t n_steps, void *stream)
{
    if (!l || n_steps < 0) return UAVENV_EINVAL;
    const UavSacLoopConfig &c = l->c;
    const UavReplayRing &R = c.ring;
    const int U = c.n_slots, B = c.batch;
    const size_t n = (size_t)R.n_agents;
    const int envs = R.n_agents / U;
    const int64_t nb = (int64_t)U * B;
    const int64_t n_noise = uavenv_sac_loop_noise_floats(U, envs, B);
    float *act0 = (float *)R.action;
    hipStream_t s = (hipStream_t)stream;
    for (int k = 0; k < n_steps; ++k) {
        const int t = l->head, nxt = t + 1 == R.frames ? 0 : t + 1;
        l->counter += 1;                                                  // (the Python loop counts before it draws)
        int rc = uavenv_randn(c.seed, l->counter, n_noise, c.noise_dev, s);
        if (rc != UAVENV_OK) return rc;
        const float *za = c.noise_dev;                                     // [U][envs][2]   get_action
        const float *zl = c.noise_dev + (size_t)U * 2 * envs;              // [2][U * B][2]  rsample() of calc_target / of the actor phase
        {                                                                 // get_action of every slot: one launch
            const float *actors[UAVENV_SAC_LOOP_MAX_SLOTS], *eps[UAVENV_SAC_LOOP_MAX_SLOTS];
            int32_t first[UAVENV_SAC_LOOP_MAX_SLOTS];
            for (int j = 0; j < U; ++j) {
                actors[j] = c.slot[j].nets.actor;
                eps[j] = za + (size_t)j * 2 * envs;
                first[j] = (int32_t)((size_t)t * n + j);
            }
            rc = uavenv_sac_act_multi(actors, R.obs, first, U, envs, eps, c.action_bound, act0, c.act1_plane, U, s);
            if (rc != UAVENV_OK) return rc;
        }
        rc = uavenv_set_step_meta(c.env, R.meta ? (unsigned char *)R.meta + (size_t)t * n * UAVENV_META_BYTES : nullptr,
                                  R.meta ? c.act1_plane + (size_t)t * n : nullptr);
        if (rc != UAVENV_OK) return rc;
        rc = uavenv_step(c.env, act0 + (size_t)t * n, UAVENV_ACT_STEER_F32, (unsigned char *)R.obs + (size_t)nxt * l->obs_row_bytes, nullptr,
                         R.reward + (size_t)t * n, R.done + (size_t)t * n, nullptr, c.info_dev ? c.info_dev + (size_t)t * n : nullptr,
                         R.valid + (size_t)t * n, nullptr, nullptr, c.step_flags, s);
        if (rc != UAVENV_OK) return rc;
        l->head = nxt;
        if (l->filled < R.frames - 1) l->filled += 1;
        if (l->per) {                     // ReplayTree.push(error 0) for every slot's rows of the frame just written (its column of
            const double per_new = pow(0.0 + c.per_eps, c.per_alpha);          // the valid plane); the new head's rows are retired
            for (int j = 0; j < U; ++j) {
                rc = uavenv_per_fill_frame_strided(&c.slot[j].per, (int64_t)t * envs, envs, per_new, R.valid + (size_t)t * n + j, U,
                                                   (int64_t)nxt * envs, s);
                if (rc != UAVENV_OK) return rc;
            }
        }
        const bool learn = c.is_train && (int64_t)l->filled * envs > (int64_t)B;            // :383-385
        bool one_draw = false;
        const bool valid_only = c.valid_draws != 0;
        if (!l->per && learn && (int64_t)l->filled * envs >= nb) {
            // finished agents are skipped, not restarted: their rows stay in the ring with valid = 0, and the draws go over the
            // valid rows only (the reference's buffers never hold such rows)
            rc = valid_only ? uavenv_replay_draw_valid(R.frames, envs, l->head, l->filled, B, U, U, 0, R.valid, UAVENV_DRAW_MAX_TRIES,
                                                       c.seed + 7, l->counter, c.draws_dev, s)
                            : uavenv_replay_draw(R.frames, envs, l->head, l->filled, (int32_t)nb, c.seed + 7, l->counter, c.draws_dev, s);
            if (rc != UAVENV_OK) return rc;
            one_draw = true;
        }
        for (int j = 0; j < U; ++j) l->epoch[j] += 1;                     // update() is called either way (:322-333)
        if (!learn) continue;
        UavSacNets nets[UAVENV_SAC_LOOP_MAX_SLOTS];
        UavSacBatch bt[UAVENV_SAC_LOOP_MAX_SLOTS];
        UavSacAdam hc[UAVENV_SAC_LOOP_MAX_SLOTS], ha[UAVENV_SAC_LOOP_MAX_SLOTS];
        float *pc[UAVENV_SAC_LOOP_MAX_SLOTS], *pa[UAVENV_SAC_LOOP_MAX_SLOTS], *m1[UAVENV_SAC_LOOP_MAX_SLOTS], *v1[UAVENV_SAC_LOOP_MAX_SLOTS],
              *m2[UAVENV_SAC_LOOP_MAX_SLOTS], *v2[UAVENV_SAC_LOOP_MAX_SLOTS], *ma[UAVENV_SAC_LOOP_MAX_SLOTS], *va[UAVENV_SAC_LOOP_MAX_SLOTS],
              *amv[UAVENV_SAC_LOOP_MAX_SLOTS], *sc_c[UAVENV_SAC_LOOP_MAX_SLOTS], *sc_a[UAVENV_SAC_LOOP_MAX_SLOTS];
        for (int j = 0; j < U; ++j) {
            const UavSacLoopSlot &sl = c.slot[j];
            int32_t *draws = c.draws_dev + (size_t)j * B * 2;
            if (l->per) {             // ReplayTree.sample (:146-180): beta first (:155), selection, importance weights + (frame, env)
                l->per_beta[j] = l->per_beta[j] + c.per_beta_inc < 1.0 ? l->per_beta[j] + c.per_beta_inc : 1.0;
                rc = uavenv_per_rebuild(&sl.per, s);
                if (rc != UAVENV_OK) return rc;
                rc = uavenv_per_sample(&sl.per, B, nullptr, c.seed + 7 + (uint64_t)j, l->counter, sl.per_slots_dev, sl.per_prio_dev, s);
                if (rc != UAVENV_OK) return rc;
                rc = uavenv_per_weights(&sl.per, sl.per_slots_dev, sl.per_prio_dev, B, (int64_t)l->filled * envs, l->per_beta[j], envs,
                                        sl.per_w_dev, draws, s);
                if (rc != UAVENV_OK) return rc;
            } else if (!one_draw) {   // the ring does not hold U x B transitions yet: one draw per slot
                rc = valid_only ? uavenv_replay_draw_valid(R.frames, envs, l->head, l->filled, B, 1, U, j, R.valid, UAVENV_DRAW_MAX_TRIES,
                                                           c.seed + 7 + (uint64_t)j, l->counter, draws, s)
                                : uavenv_replay_draw(R.frames, envs, l->head, l->filled, B, c.seed + 7 + (uint64_t)j, l->counter, draws, s);
                if (rc != UAVENV_OK) return rc;
            }
            l->adam_steps[j] += 1;
            const double tt = (double)l->adam_steps[j];
            UavSacBatch b = UavSacBatch();
            b.obs_packed = R.obs;
            b.draws = draws;
            b.n_agents = R.n_agents; b.uav_per_env = U; b.slot = j; b.frames = R.frames;
            b.act0 = act0; b.act1 = c.act1_plane; b.reward = R.reward; b.done = R.done; b.valid = R.valid;
            b.meta = R.meta;
            b.batch = B;
            if (l->per) { b.is_weights = sl.per_w_dev; b.abs_td_out = sl.per_abs_dev; }
            b.td_scratch = sl.td_dev;
            b.eps = zl + ((size_t)j * B) * 2;                             // rsample() of calc_target; the actor phase's below
            bt[j] = b;
            nets[j] = sl.nets;
            UavSacAdam h;
            h.beta1 = (float)c.beta1; h.beta2 = (float)c.beta2; h.eps = (float)c.adam_eps;
            h.bias_correction1 = (float)(1.0 - pow(c.beta1, tt));
            h.bias_correction2_sqrt = (float)sqrt(1.0 - pow(c.beta2, tt));
            h.grad_scale = 0.0f;
            h.skip_word = c.p2p ? uavenv_p2p_error_word(c.p2p) : nullptr;   // Adam behind a failed pull: a no-op
            h.go_word = c.moved_dev;                                          // ... and behind a step that moved nobody
            h.go_value = (uint32_t)uavenv_tick(c.env);
            h.reserved0 = 0;
            h.lr = c.critic_lr; h.tau = c.tau;
            hc[j] = h;
            h.lr = c.actor_lr; h.tau = 0.0f;
            ha[j] = h;
            pc[j] = sl.partials_critic; pa[j] = sl.partials_actor;
            m1[j] = sl.m1; v1[j] = sl.v1; m2[j] = sl.m2; v2[j] = sl.v2; ma[j] = sl.m_actor; va[j] = sl.v_actor; amv[j] = sl.alpha_mv;
            sc_c[j] = sl.scalars; sc_a[j] = sl.scalars + 4;
        }
        // critics (:340-357), then the actor and log_alpha on the updated critics (:359-377); the soft update (:378-379) rides
        // in critic_adam -- every phase for all U slots in one launch
        int rows = uavenv_sac_partial_rows_n(B, U, 0);       // workgroups (= partial rows) per slot of a U-slot launch
        const bool multi = c.p2p || c.coll;
        // N > 1: the U slots' column sums side by side in xbuf, summed over the ranks on the stream, one row per slot from there
        auto exchange = [&](float **part, int stride) -> int {
            for (int j = 0; j < U; ++j) {
                const int r2 = uavenv_sac_reduce(part[j], rows, stride, c.xbuf_dev + (size_t)j * stride, s);
                if (r2 != UAVENV_OK) return r2;
                part[j] = c.xbuf_dev + (size_t)j * stride;
            }
            return c.p2p ? uavenv_p2p_allreduce(c.p2p, c.xbuf_dev, (int64_t)U * stride, s)
                         : uavenv_coll_allreduce_sum(c.coll, c.xbuf_dev, (int64_t)U * stride, s);
        };
        rc = uavenv_sac_critic_grad_multi(nets, bt, U, c.gamma, c.action_bound, pc, s);
        if (rc != UAVENV_OK) return rc;
        if (multi) {
            rc = exchange(pc, UAVENV_SAC_CRITIC_STRIDE);
            if (rc == UAVENV_EP2P) {                     // sticky error: no Adam launch of this update steps (skip_word) -- the
                for (int j = 0; j < U; ++j) l->adam_steps[j] -= 1;       // update did not happen, so it is not counted either
                return rc;
            }
            if (rc != UAVENV_OK) return rc;
        }
        rc = uavenv_sac_critic_adam_multi(nets, pc, multi ? 1 : rows, m1, v1, m2, v2, hc, sc_c, U, s);
        if (rc != UAVENV_OK) return rc;
        for (int j = 0; j < U; ++j) bt[j].eps = zl + ((size_t)nb + (size_t)j * B) * 2;
        rc = uavenv_sac_actor_grad_multi(nets, bt, U, c.action_bound, pa, s);
        if (rc != UAVENV_OK) return rc;
        if (multi) {
            rc = exchange(pa, UAVENV_SAC_ACTOR_STRIDE);
            if (rc != UAVENV_OK && rc != UAVENV_EP2P) return rc;
            // EP2P here: the critics of this update did step (their exchange succeeded); the actor launch below is enqueued all the
            // same and skips itself on the device (skip_word) -- the caller re-synchronises every block from one rank anyway
        }
        const int rc_x = rc;
        rc = uavenv_sac_actor_adam_multi(nets, pa, multi ? 1 : rows, B, ma, va, amv, ha, c.alpha_lr, c.target_entropy, sc_a, U, s);
        if (rc != UAVENV_OK) return rc;
        if (rc_x == UAVENV_EP2P) return rc_x;
        if (c.p2p && c.check_every > 0 && l->adam_steps[0] % c.check_every == 0) {
            // the ranks apply bit-identical updates by construction: every check_every updates the parameter blocks of all slots
            // (actor, both critics, both targets) are hashed and compared across the ranks on the device (csrc/p2p.hip)
            const float *blk[UAVENV_P2P_CHECK_MAX_BLOCKS];
            int32_t nf[UAVENV_P2P_CHECK_MAX_BLOCKS];
            int nb_ = 0;
            for (int j = 0; j < U; ++j) {
                const UavSacNets &nn = c.slot[j].nets;
                const float *ps[5] = {nn.actor, nn.critic1, nn.critic2, nn.target1, nn.target2};
                for (int q = 0; q < 5; ++q) { blk[nb_] = ps[q]; nf[nb_] = q == 0 ? UAVENV_SAC_ACTOR_PARAMS : UAVENV_SAC_CRITIC_PARAMS; ++nb_; }
            }
            rc = uavenv_p2p_check_blocks(c.p2p, blk, nf, nb_, s);
            if (rc != UAVENV_OK) return rc;
        }
        if (l->per) {                     // ReplayTree.batch_update (:215-222, :352) with the |TD| the critic phase left
            for (int j = 0; j < U; ++j) {
                rc = uavenv_per_set_f32_gated(&c.slot[j].per, c.slot[j].per_slots_dev, c.slot[j].per_abs_dev, B, c.per_eps, c.per_alpha,
                                              c.per_clip, c.moved_dev, (uint32_t)uavenv_tick(c.env), s);
                if (rc != UAVENV_OK) return rc;
            }
        }
    }
    return UAVENV_OK;
}

}  // extern "C"
