// coevo_mpe_rollout - a whole batch of games in ONE C-ABI call: the replacement for the reference's per-game
// play_game()/play_MPE() loop (utils/game_logic_functions.py:123-228) at batch scale.
//
// Per env-cycle and cohort it enqueues ONE merged launch (coevo_mpe_policy_cycle_merged: shared-opponent workgroups on
// the matrix cores + per-individual streaming workgroups, env step fused in); cohorts are independent chains on their
// own streams (cohort 0: the caller's).  With d->merged == 0 the older two-launch form is used instead: [side stream]
// the shared-opponent launch || [main stream] the per-individual launch, joined by an event.  Nothing here synchronises
// with the host; a single-cohort sequence can also be captured into a hipGraph by the caller.
#include <cstdlib>
#include <vector>

#include "coevo_common.hip.h"

static_assert(sizeof(coevo_rollout_desc) == 192, "layout mirrored by coevonet_amd/lib.py RolloutDesc");

__global__ void stamps_init_kernel(uint64_t *stamps, int n)
{
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        stamps[2 * i] = ~0ull;
        stamps[2 * i + 1] = 0ull;
    }
}

// stream of one cohort beyond the first (the first cohort uses the caller's stream)
struct cohort_lane {
    hipStream_t s = nullptr;
    hipEvent_t done = nullptr;
};

struct coevo_rollout_ctx {
    hipStream_t side = nullptr;
    hipEvent_t fork = nullptr, join = nullptr, start = nullptr;
    std::vector<cohort_lane> lanes;  // created on first use
    std::vector<hipEvent_t> timing;  // pairs (start, end) around the light launch of each cycle
    int pairs_used = 0;
};

extern "C" void *coevo_rollout_ctx_create(int n_timing_pairs)
{
    auto *c = new coevo_rollout_ctx();
    int prio_lo = 0, prio_hi = 0;  // numerically lowest = highest priority
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    if (hipStreamCreateWithPriority(&c->side, hipStreamNonBlocking, prio_hi) != hipSuccess ||
        hipEventCreateWithFlags(&c->fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->join, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->start, hipEventDisableTiming) != hipSuccess) {
        delete c;
        return nullptr;
    }
    c->timing.resize(2 * (size_t)(n_timing_pairs > 0 ? n_timing_pairs : 0));
    for (auto &e : c->timing)
        if (hipEventCreate(&e) != hipSuccess) { delete c; return nullptr; }
    return c;
}

extern "C" void coevo_rollout_ctx_destroy(void *ctx)
{
    auto *c = static_cast<coevo_rollout_ctx *>(ctx);
    if (!c) return;
    for (auto e : c->timing) (void)hipEventDestroy(e);
    for (auto &ln : c->lanes) {
        if (ln.done) (void)hipEventDestroy(ln.done);
        if (ln.s) (void)hipStreamDestroy(ln.s);
    }
    if (c->start) (void)hipEventDestroy(c->start);
    if (c->fork) (void)hipEventDestroy(c->fork);
    if (c->join) (void)hipEventDestroy(c->join);
    if (c->side) (void)hipStreamDestroy(c->side);
    delete c;
}

// streams for up to n_cohorts independent cohorts (coevo_rollout_desc.n_cohorts); call outside graph capture
extern "C" int coevo_rollout_ctx_reserve_cohorts(void *ctx, int n_cohorts)
{
    auto *c = static_cast<coevo_rollout_ctx *>(ctx);
    if (!c || n_cohorts < 1 || n_cohorts > COEVO_MAX_COHORTS) return COEVO_ERR_ARG;
    while ((int)c->lanes.size() < n_cohorts - 1) {
        cohort_lane ln;
        if (hipStreamCreateWithFlags(&ln.s, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&ln.done, hipEventDisableTiming) != hipSuccess)
            return COEVO_ERR_HIP;
        c->lanes.push_back(ln);
    }
    return COEVO_OK;
}

// the stream cohort k (>= 1) of this context runs on (cohort 0 runs on the caller's stream): for callers that enqueue
// per-cohort work (breeding, resets) in front of a cohort's chain themselves
extern "C" void *coevo_rollout_ctx_cohort_stream(void *ctx, int k)
{
    auto *c = static_cast<coevo_rollout_ctx *>(ctx);
    if (!c || k < 1 || k > (int)c->lanes.size()) return nullptr;
    return c->lanes[k - 1].s;
}

// generic use of the context's timing event pairs: bracket whatever is enqueued on `stream` between the two calls (eager
// enqueue only - events cannot be recorded inside a captured graph on this runtime); read with
// coevo_rollout_ctx_light_times.  Used by coevo_dqn_forward_argmax_timed around its dominant kernel.
extern "C" int coevo_timing_begin(void *ctx, void *stream)
{
    auto *c = static_cast<coevo_rollout_ctx *>(ctx);
    if (!c) return COEVO_ERR_ARG;
    if (2 * (size_t)c->pairs_used + 1 >= c->timing.size()) return COEVO_OK;   // out of pairs: stop sampling, keep running
    COEVO_HIP_CHECK(hipEventRecord(c->timing[2 * c->pairs_used], (hipStream_t)stream));
    return COEVO_OK;
}

extern "C" int coevo_timing_end(void *ctx, void *stream)
{
    auto *c = static_cast<coevo_rollout_ctx *>(ctx);
    if (!c) return COEVO_ERR_ARG;
    if (2 * (size_t)c->pairs_used + 1 >= c->timing.size()) return COEVO_OK;
    COEVO_HIP_CHECK(hipEventRecord(c->timing[2 * c->pairs_used + 1], (hipStream_t)stream));
    c->pairs_used += 1;
    return COEVO_OK;
}

extern "C" int coevo_rollout_ctx_reset_timing(void *ctx)
{
    auto *c = static_cast<coevo_rollout_ctx *>(ctx);
    if (!c) return COEVO_ERR_ARG;
    c->pairs_used = 0;
    return COEVO_OK;
}

// elapsed milliseconds of every timed light launch so far (blocks until they have completed); returns the count
extern "C" int coevo_rollout_ctx_light_times(void *ctx, float *ms_out, int max_out)
{
    auto *c = static_cast<coevo_rollout_ctx *>(ctx);
    if (!c || !ms_out) return COEVO_ERR_ARG;
    int n = c->pairs_used < max_out ? c->pairs_used : max_out;
    for (int i = 0; i < n; ++i) {
        if (hipEventSynchronize(c->timing[2 * i + 1]) != hipSuccess) return COEVO_ERR_HIP;
        if (hipEventElapsedTime(&ms_out[i], c->timing[2 * i], c->timing[2 * i + 1]) != hipSuccess) return COEVO_ERR_HIP;
    }
    return n;
}

extern "C" int coevo_mpe_rollout(const coevo_rollout_desc *d, void *ctx, int time_light, void *stream)
{
    if (!d || !d->slab || !d->state || !d->row_game || !d->row_slot || !d->status) return COEVO_ERR_ARG;
    if ((d->state_alt == nullptr) != (d->actions_by_game == nullptr)) return COEVO_ERR_ARG;
    if (!d->state_alt && (!d->game_rows || !d->actions)) return COEVO_ERR_ARG;
    if (d->n_games <= 0 || d->n_cycles < 0 || d->n_heavy < 0 || d->n_light < 0) return COEVO_ERR_ARG;
    if ((d->n_heavy > 0 && !d->heavy) || (d->n_light > 0 && !d->light)) return COEVO_ERR_ARG;
    auto *c = static_cast<coevo_rollout_ctx *>(ctx);
    hipStream_t main_s = (hipStream_t)stream;
    // HIP events cannot be read back from inside a captured graph (external event-record nodes are rejected by this
    // runtime), so event timing is for eager enqueues only; graph replays use the kernels' own clock stamps instead
    auto record_timing = [&](hipEvent_t e) { return hipEventRecord(e, main_s); };
    if (d->light_stamps && d->n_cycles > 0 && !d->stamps_armed) {  // [cycle][2] = {UINT64_MAX, 0}, re-armed by every enqueue / replay
        hipLaunchKernelGGL(stamps_init_kernel, dim3(1), dim3(256), 0, main_s, d->light_stamps,
                           (d->n_cohorts > 1 ? d->n_cohorts : 1) * d->n_cycles * COEVO_STAMP_SLOTS);
        COEVO_HIP_CHECK(hipGetLastError());
    }
    const bool fused = d->state_alt != nullptr && d->actions_by_game != nullptr;
    static const bool persistent_ok = !(getenv("COEVO_PERSISTENT") && getenv("COEVO_PERSISTENT")[0] == '0');   // A/B
    const size_t act_stride = 3 * (size_t)d->n_games;
    const int K = d->n_cohorts > 1 ? d->n_cohorts : 1;
    if (K > 1) {
        if (!c || !fused || !d->heavy_begin || !d->light_begin || K > COEVO_MAX_COHORTS) return COEVO_ERR_ARG;
        if (d->heavy_begin[0] != 0 || d->light_begin[0] != 0 || d->heavy_begin[K] != d->n_heavy ||
            d->light_begin[K] != d->n_light)
            return COEVO_ERR_ARG;
        for (int k = 0; k < K; ++k)
            if (d->heavy_begin[k + 1] < d->heavy_begin[k] || d->light_begin[k + 1] < d->light_begin[k]) return COEVO_ERR_ARG;
        // the lanes must exist already: streams cannot be created while the caller is capturing a graph
        if ((int)c->lanes.size() < K - 1) return COEVO_ERR_ARG;
        COEVO_HIP_CHECK(hipEventRecord(c->start, main_s));
        // One stream per cohort, forked from / joined to the caller's stream only.  (Under graph capture this runtime
        // tolerates mutual waits between the origin stream and a forked stream - the K = 1 pair below - but two forked
        // streams that wait on each other send hip::Stream::EndCapture into endless recursion.)  Inside a cohort the
        // shared-opponent launch and the per-individual launch of a cycle therefore run back to back; the overlap of
        // matrix-core work with weight streaming now comes from the other cohorts' launches.
        for (int k = 1; k < K; ++k) COEVO_HIP_CHECK(hipStreamWaitEvent(c->lanes[k - 1].s, c->start, 0));
    }
    bool closed_in_launch = false;
    // one cohort's chain of cycles: per cycle the shared-opponent launch on `hs` || the per-individual launch on `ls`
    // (hs == ls: back to back)
    auto chain = [&](int k, const coevo_fc_task *heavy, int n_heavy, const coevo_fc_task *light, int n_light,
                     hipStream_t ls, hipStream_t hs, hipEvent_t fork, hipEvent_t join) -> int {
        const bool two = c && hs != ls && n_heavy > 0 && n_light > 0;
        // a cohort whose workgroups are all resident at once (coevo_mpe_persistent_fits): its n_cycles as ONE persistent launch
        if (persistent_ok && fused && d->merged && d->sync_words && n_heavy + n_light > 0 && d->n_cycles > 0) {
            const int conc = d->concurrent_hint > K ? d->concurrent_hint : K;
            if (coevo_mpe_persistent_fits(n_heavy, n_light, d->heavy_max_rows, d->light_max_rows, conc) == 1) {
                // one cohort = every game of the call: its launch closes the books too (no closing launch below)
                const bool closes = K == 1 && d->rewards != nullptr;
                closed_in_launch = closes;
                return coevo_mpe_rollout_persistent(
                    d->slab, heavy, n_heavy, light, n_light, d->light_max_rows, d->heavy_max_rows, d->state, d->state_alt,
                    d->n_games, d->row_game, d->row_slot, d->actions_by_game, d->game_limit, d->n_cycles, d->pos_first, d->status,
                    d->light_stamps ? d->light_stamps + 2 * COEVO_STAMP_SLOTS * ((size_t)k * d->n_cycles) : nullptr,
                    d->sync_words + (size_t)k * (size_t)(4 + 6 * (size_t)d->n_games), conc, closes ? d->rewards : nullptr,
                    closes ? d->pack : nullptr, d->sync_cleared, ls);
            }
        }
        for (int cyc = 0; cyc < d->n_cycles; ++cyc) {
            int rc;
            // fused env step: cycle c reads the state of cycle c-1 (buffer (c-1)&1; buffer 0 holds the reset state)
            // and the actions of cycle c-1, derives its own state in registers, the owner rows write it to buffer c&1
            const double *st_prev = (cyc == 0) ? d->state : (((cyc - 1) & 1) ? d->state_alt : d->state);
            double *st_next = (cyc & 1) ? d->state_alt : d->state;
            if (cyc == 0) st_next = d->state_alt;  // never written in cycle 0; only has to differ from st_prev
            const int32_t *act_prev = fused ? d->actions_by_game + (size_t)((cyc + 1) & 1) * act_stride : nullptr;
            int32_t *act_cur = fused ? d->actions_by_game + (size_t)(cyc & 1) * act_stride : nullptr;
            auto policy = [&](const coevo_fc_task *tasks, int n_tasks, int max_rows, uint64_t *stamps, hipStream_t s) {
                if (fused)
                    return coevo_mpe_policy_cycle_fused(d->slab, tasks, n_tasks, max_rows, st_prev, st_next, d->n_games,
                                                        d->row_game, d->row_slot, act_prev, act_cur, d->game_limit,
                                                        cyc, d->pos_first, d->status, stamps, s);
                return coevo_mpe_policy_cycle_stamped(d->slab, tasks, n_tasks, max_rows, d->state, d->n_games,
                                                      d->row_game, d->row_slot, d->actions, d->status, stamps, s);
            };
            if (fused && d->merged && n_heavy > 0 && n_light > 0 && d->light_max_rows <= 8) {
                rc = coevo_mpe_policy_cycle_merged(
                    d->slab, heavy, n_heavy, light, n_light, d->light_max_rows, st_prev, st_next, d->n_games, d->row_game,
                    d->row_slot, act_prev, act_cur, d->game_limit, cyc, d->pos_first, d->status,
                    d->light_stamps ? d->light_stamps + 2 * COEVO_STAMP_SLOTS * ((size_t)k * d->n_cycles + cyc) : nullptr,
                    d->concurrent_hint > K ? d->concurrent_hint : K, d->heavy_max_rows, ls);
                if (rc) return rc;
                continue;
            }
            hipStream_t heavy_s = two ? hs : ls;
            if (two) {
                COEVO_HIP_CHECK(hipEventRecord(fork, ls));
                COEVO_HIP_CHECK(hipStreamWaitEvent(hs, fork, 0));
            }
            if (n_heavy > 0) {
                rc = policy(heavy, n_heavy, d->heavy_max_rows, nullptr, heavy_s);
                if (rc) return rc;
                if (two) COEVO_HIP_CHECK(hipEventRecord(join, hs));
            }
            if (n_light > 0) {
                const bool timed = K == 1 && time_light && c && (size_t)(2 * c->pairs_used + 1) < c->timing.size();
                if (timed) COEVO_HIP_CHECK(record_timing(c->timing[2 * c->pairs_used]));
                rc = policy(light, n_light, d->light_max_rows,
                            d->light_stamps ? d->light_stamps + 2 * COEVO_STAMP_SLOTS * ((size_t)k * d->n_cycles + cyc)
                                            : nullptr, ls);
                if (rc) return rc;
                if (timed) {
                    COEVO_HIP_CHECK(record_timing(c->timing[2 * c->pairs_used + 1]));
                    ++c->pairs_used;
                }
            }
            if (two) COEVO_HIP_CHECK(hipStreamWaitEvent(ls, join, 0));
            if (!fused) {
                rc = coevo_mpe_step(d->state, d->n_games, d->game_rows, d->actions, cyc, d->game_limit, d->pos_first, ls);
                if (rc) return rc;
            }
        }
        return COEVO_OK;
    };
    if (K == 1) {
        const int rc = chain(0, d->heavy, d->n_heavy, d->light, d->n_light, main_s, c ? c->side : main_s,
                             c ? c->fork : nullptr, c ? c->join : nullptr);
        if (rc) return rc;
    } else {
        // enqueue order = cohort by cohort; the chains only meet again at the `done` events below
        for (int k = 0; k < K; ++k) {
            const int hb = d->heavy_begin[k], lb = d->light_begin[k];
            hipStream_t ks = k ? c->lanes[k - 1].s : main_s;
            const int rc = chain(k, d->heavy ? d->heavy + hb : nullptr, d->heavy_begin[k + 1] - hb,
                                 d->light ? d->light + lb : nullptr, d->light_begin[k + 1] - lb, ks, ks, nullptr, nullptr);
            if (rc) return rc;
        }
        for (int k = 1; k < K; ++k) {
            COEVO_HIP_CHECK(hipEventRecord(c->lanes[k - 1].done, c->lanes[k - 1].s));
            COEVO_HIP_CHECK(hipStreamWaitEvent(main_s, c->lanes[k - 1].done, 0));
        }
    }
    if (fused) {
        if (!d->rewards) return COEVO_OK;  // the caller closes the rollout itself (coevo_mpe_final_step), e.g. after
                                            // several per-cohort calls on different streams
        if (closed_in_launch) return COEVO_OK;   // (the persistent launch did)
        const int last = d->n_cycles - 1;  // -1: no cycle ran, the books are the reset state's zeros
        const double *st_last = (last <= 0) ? d->state : ((last & 1) ? d->state_alt : d->state);
        const int32_t *act_last = d->actions_by_game + (size_t)((last < 0 ? 0 : last) & 1) * act_stride;
        if (d->pack)
            return coevo_mpe_final_step_pack(st_last, d->n_games, act_last, last, d->game_limit, d->pos_first, d->rewards,
                                             d->pack->out, d->pack->dist, d->pack->n_roles, d->pack->n_local, d->pack->hof,
                                             d->pack->dist_pitch, d->pack->dist_first, main_s);
        return coevo_mpe_final_step(st_last, d->n_games, act_last, last, d->game_limit, d->pos_first, d->rewards, main_s);
    }
    if (d->rewards) return coevo_mpe_rewards(d->state, d->n_games, d->rewards, main_s);
    return COEVO_OK;
}

COEVO_DEFINE_TU_FLAGS(rollout_api)
