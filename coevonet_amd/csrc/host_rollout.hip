// coevo_mpe_host_rollout - north_star's first configuration as ONE C-ABI call: "vectorised env stepping runs on the host
// cores", the policy steps on the GPU.  Replaces, for all of a rank's games at once, the per-game loop of play_MPE
// (utils/game_logic_functions.py:123-212: env.observe :138, the forward :152-163, env.step :179, env.last :181,
// rewards[agent] += reward :190, the limit / truncation breaks :197-204).
//
// The games are cut into K cohorts (contiguous row ranges of the observation / action buffers).  Per env-cycle and cohort:
//   host cores   step the cohort's games with the actions of the previous cycle, write their observations  (T threads,
//                games are independent: thread t owns a contiguous slice of the cohort's game list)
//   cohort stream  obs rows host -> device, ONE policy launch (coevo_fc_forward_merged, or one launch per task table when the
//                lean merged kernel does not hold them), action rows device -> host, an event
// and the caller's thread moves on to the next cohort: while cohort k's copies and launch are in flight the cores step
// cohort k+1, and the cohorts' launches run side by side on the GPU.  The only waits are per-cohort event polls; nothing
// synchronises a whole stream or the device.  Results do not depend on K or T: every game is stepped by exactly one thread
// with the device env's own bodies (csrc/mpe_env.hip, coevo_common.hip.h; fp64, -ffp-contract=off on both sides).
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "coevo_common.hip.h"

namespace {

struct HostPool {
    typedef void (*fn_t)(void *, int, int);
    int T = 1;
    std::vector<std::thread> th;
    std::mutex m;
    std::condition_variable cv;
    std::atomic<uint64_t> epoch{0};
    std::atomic<int> pending{0};
    std::atomic<bool> stop{false};
    fn_t fn = nullptr;
    void *arg = nullptr;

    explicit HostPool(int threads) : T(threads < 1 ? 1 : threads)
    {
        for (int i = 1; i < T; ++i) th.emplace_back([this, i] { worker(i); });
    }
    ~HostPool()
    {
        {
            std::lock_guard<std::mutex> g(m);
            stop.store(true, std::memory_order_release);
        }
        cv.notify_all();
        for (auto &t : th) t.join();
    }
    void worker(int i)
    {
        uint64_t seen = 0;
        for (;;) {
            // between two cycles of a rollout the next piece of work is tens of microseconds away: poll first, sleep
            // only when nothing came for a while (between rollouts)
            bool got = false;
            for (int spin = 0; spin < 20000; ++spin) {
                if (epoch.load(std::memory_order_acquire) != seen || stop.load(std::memory_order_acquire)) { got = true; break; }
                __builtin_ia32_pause();
            }
            if (!got) {
                std::unique_lock<std::mutex> g(m);
                cv.wait(g, [&] { return epoch.load(std::memory_order_acquire) != seen || stop.load(std::memory_order_acquire); });
            }
            if (stop.load(std::memory_order_acquire)) return;
            seen = epoch.load(std::memory_order_acquire);
            fn(arg, i, T);
            pending.fetch_sub(1, std::memory_order_acq_rel);
        }
    }
    // fn(arg, part, parts) on every thread (the caller is part 0); returns when all parts are done
    void run(fn_t f, void *a)
    {
        if (T == 1) { f(a, 0, 1); return; }
        {
            std::lock_guard<std::mutex> g(m);
            fn = f;
            arg = a;
            pending.store(T - 1, std::memory_order_release);
            epoch.fetch_add(1, std::memory_order_acq_rel);
        }
        cv.notify_all();
        f(a, 0, T);
        while (pending.load(std::memory_order_acquire) != 0) __builtin_ia32_pause();
    }
};

struct HostLane {
    hipStream_t s = nullptr;
    hipEvent_t done = nullptr;
    hipEvent_t t[4] = {nullptr, nullptr, nullptr, nullptr};   // timing: before h2d, after h2d, after launch, after d2h
};

struct HostRollout {
    HostPool pool;
    std::vector<HostLane> lanes;
    hipEvent_t start = nullptr;
    int max_cohorts = 1;
    explicit HostRollout(int threads) : pool(threads) {}
};

struct StepJob {
    const coevo_host_rollout_desc *d;
    const coevo_host_cohort *c;
    int step_cycle;   // >= 0: apply that cycle's actions first
    int observe;
};

void step_part(void *arg, int part, int parts)
{
    const StepJob *j = static_cast<const StepJob *>(arg);
    const int n = j->c->n_games;
    const int lo = (int)((int64_t)n * part / parts), hi = (int)((int64_t)n * (part + 1) / parts);
    if (hi > lo)
        (void)coevo_mpe_host_step_games(j->d->state, j->d->n_games, j->d->game_rows, j->d->actions_host, j->step_cycle,
                                        j->d->game_limit, j->d->pos_first, j->c->games, lo, hi, j->observe, j->d->obs_host);
}

inline double now_us()
{
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// poll (the copy-back of a few KB finishes within microseconds of the launch; a blocking wait costs a wake-up)
int wait_event(hipEvent_t e)
{
    for (int i = 0; i < 4000000; ++i) {
        const hipError_t q = hipEventQuery(e);
        if (q == hipSuccess) return COEVO_OK;
        if (q != hipErrorNotReady) return COEVO_ERR_HIP;
        __builtin_ia32_pause();
    }
    return hipEventSynchronize(e) == hipSuccess ? COEVO_OK : COEVO_ERR_HIP;
}

}  // namespace

extern "C" void *coevo_host_rollout_create(int n_threads, int n_cohorts)
{
    if (n_threads < 1 || n_threads > 256 || n_cohorts < 1 || n_cohorts > COEVO_MAX_COHORTS) return nullptr;
    auto *h = new HostRollout(n_threads);
    h->max_cohorts = n_cohorts;   // streams and events are created by the first rollout (the context itself needs no GPU)
    return h;
}

static int ensure_lanes(HostRollout *h)
{
    if (!h->start) COEVO_HIP_CHECK(hipEventCreateWithFlags(&h->start, hipEventDisableTiming));
    while ((int)h->lanes.size() < h->max_cohorts) {
        HostLane ln;
        bool ok = hipStreamCreateWithFlags(&ln.s, hipStreamNonBlocking) == hipSuccess &&
                  hipEventCreateWithFlags(&ln.done, hipEventDisableTiming) == hipSuccess;
        for (int i = 0; ok && i < 4; ++i) ok = hipEventCreate(&ln.t[i]) == hipSuccess;
        h->lanes.push_back(ln);   // (destroy() releases whatever was created)
        if (!ok) return COEVO_ERR_HIP;
    }
    return COEVO_OK;
}

extern "C" void coevo_host_rollout_destroy(void *handle)
{
    auto *h = static_cast<HostRollout *>(handle);
    if (!h) return;
    for (auto &ln : h->lanes) {
        for (auto e : ln.t)
            if (e) (void)hipEventDestroy(e);
        if (ln.done) (void)hipEventDestroy(ln.done);
        if (ln.s) (void)hipStreamDestroy(ln.s);
    }
    if (h->start) (void)hipEventDestroy(h->start);
    delete h;
}

extern "C" int coevo_host_rollout_threads(void *handle)
{
    auto *h = static_cast<HostRollout *>(handle);
    return h ? h->pool.T : COEVO_ERR_ARG;
}

// world step `cycle` (< 0: none) + observations (observe != 0) of the listed games on the context's host cores: what a
// rollout does per cohort and env-cycle, exposed for callers that drive the cycle themselves (and for the CPU tests:
// n_threads slices == one thread, bit for bit)
extern "C" int coevo_host_rollout_step(void *handle, double *state, int n_games, const int32_t *game_rows,
                                       const int32_t *actions, int n_rows, int cycle, const int32_t *game_limit, int pos_first,
                                       const int32_t *games, int n_list, int observe, float *obs)
{
    auto *h = static_cast<HostRollout *>(handle);
    if (!h || !state || !game_rows || !games || n_games <= 0 || n_rows <= 0 || n_list < 0 || (cycle >= 0 && !actions) ||
        (observe && !obs))
        return COEVO_ERR_ARG;
    for (int i = 0; i < n_list; ++i) {
        const int g = games[i];
        if (g < 0 || g >= n_games) return COEVO_ERR_ARG;
        for (int sl = 0; sl < 3; ++sl)
            if (game_rows[3 * g + sl] < 0 || game_rows[3 * g + sl] >= n_rows) return COEVO_ERR_ARG;
    }
    coevo_host_rollout_desc d{};
    d.state = state;
    d.n_games = n_games;
    d.game_rows = game_rows;
    d.actions_host = const_cast<int32_t *>(actions);
    d.game_limit = game_limit;
    d.pos_first = pos_first;
    d.obs_host = obs;
    coevo_host_cohort c{};
    c.games = games;
    c.n_games = n_list;
    StepJob job{&d, &c, cycle, observe};
    h->pool.run(step_part, &job);
    return COEVO_OK;
}

extern "C" int coevo_mpe_host_rollout(void *handle, const coevo_host_rollout_desc *d, void *stream)
{
    auto *h = static_cast<HostRollout *>(handle);
    if (!h || !d || !d->slab || !d->state || !d->game_rows || !d->obs_host || !d->obs_dev || !d->actions_host ||
        !d->actions_dev || !d->status || !d->cohorts)
        return COEVO_ERR_ARG;
    if (d->n_games <= 0 || d->n_rows != 3 * d->n_games || d->n_cycles < 0 || d->n_cohorts < 1 ||
        d->n_cohorts > h->max_cohorts)
        return COEVO_ERR_ARG;
    const int K = d->n_cohorts;
    {
        const int rc_l = ensure_lanes(h);
        if (rc_l) return rc_l;
    }
    // every game in exactly one cohort, its three rows inside the cohort's row range; the ranges disjoint
    {
        std::vector<uint8_t> seen((size_t)d->n_games, 0);
        int64_t games = 0;
        for (int k = 0; k < K; ++k) {
            const coevo_host_cohort &c = d->cohorts[k];
            if (c.n_games < 0 || (c.n_games > 0 && !c.games) || c.row_first < 0 || c.n_rows != 3 * c.n_games ||
                c.row_first + c.n_rows > d->n_rows || c.n_heavy < 0 || c.n_light < 0 || (c.n_heavy > 0 && !c.heavy) ||
                (c.n_light > 0 && !c.light))
                return COEVO_ERR_ARG;
            for (int q = 0; q < k; ++q) {
                const coevo_host_cohort &o = d->cohorts[q];
                if (c.n_rows > 0 && o.n_rows > 0 && c.row_first < o.row_first + o.n_rows && o.row_first < c.row_first + c.n_rows)
                    return COEVO_ERR_ARG;
            }
            for (int i = 0; i < c.n_games; ++i) {
                const int g = c.games[i];
                if (g < 0 || g >= d->n_games || seen[g]) return COEVO_ERR_ARG;
                seen[g] = 1;
                for (int s = 0; s < 3; ++s) {
                    const int r = d->game_rows[3 * g + s];
                    if (r < c.row_first || r >= c.row_first + c.n_rows) return COEVO_ERR_ARG;
                }
            }
            games += c.n_games;
        }
        if (games != d->n_games) return COEVO_ERR_ARG;
    }
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
        return COEVO_ERR_HIP;
    int wgs = 0;
    bool lean = true;   // the lean merged kernel needs every workgroup of every cohort in flight resident at four per CU
    for (int k = 0; k < K; ++k) {
        const coevo_host_cohort &c = d->cohorts[k];
        if (c.n_rows == 0) continue;
        wgs += c.n_heavy + c.n_light;
        if (c.n_heavy <= 0 || c.n_light <= 0 || c.heavy_max_rows > 16 || c.light_max_rows > 8) lean = false;
    }
    if (wgs > 4 * cus) lean = false;
    const bool timed = d->phase_us != nullptr;
    double acc_wait = 0.0, acc_step = 0.0, acc_enq = 0.0;
    double acc_gpu[3] = {0.0, 0.0, 0.0};
    int acc_n = 0;

    // the cohort streams see everything the caller's stream has enqueued so far (the offspring written into the slab)
    COEVO_HIP_CHECK(hipEventRecord(h->start, (hipStream_t)stream));
    for (int k = 0; k < K; ++k) COEVO_HIP_CHECK(hipStreamWaitEvent(h->lanes[k].s, h->start, 0));

    auto enqueue = [&](int k) -> int {
        const coevo_host_cohort &c = d->cohorts[k];
        HostLane &ln = h->lanes[k];
        const size_t r0 = (size_t)c.row_first;
        const float *obs = d->zero_copy ? d->obs_host : d->obs_dev;
        int32_t *act = d->zero_copy ? d->actions_host : d->actions_dev;
        if (timed) COEVO_HIP_CHECK(hipEventRecord(ln.t[0], ln.s));
        if (!d->zero_copy)
            COEVO_HIP_CHECK(hipMemcpyAsync(d->obs_dev + r0 * COEVO_OBS_STRIDE, d->obs_host + r0 * COEVO_OBS_STRIDE,
                                           (size_t)c.n_rows * COEVO_OBS_STRIDE * sizeof(float), hipMemcpyHostToDevice, ln.s));
        if (timed) COEVO_HIP_CHECK(hipEventRecord(ln.t[1], ln.s));
        int rc = COEVO_OK;
        if (lean) {
            rc = coevo_fc_forward_merged(d->slab, c.heavy, c.n_heavy, c.heavy_max_rows, c.light, c.n_light, c.light_max_rows,
                                         obs, act, nullptr, d->status, ln.s);
        } else {
            if (c.n_heavy > 0)
                rc = coevo_fc_forward_argmax(d->slab, c.heavy, c.n_heavy, c.heavy_max_rows, obs, act, nullptr, d->status, ln.s);
            if (rc == COEVO_OK && c.n_light > 0)
                rc = coevo_fc_forward_argmax(d->slab, c.light, c.n_light, c.light_max_rows, obs, act, nullptr, d->status, ln.s);
        }
        if (rc) return rc;
        if (timed) COEVO_HIP_CHECK(hipEventRecord(ln.t[2], ln.s));
        if (!d->zero_copy)
            COEVO_HIP_CHECK(hipMemcpyAsync(d->actions_host + r0, d->actions_dev + r0, (size_t)c.n_rows * sizeof(int32_t),
                                           hipMemcpyDeviceToHost, ln.s));
        if (timed) COEVO_HIP_CHECK(hipEventRecord(ln.t[3], ln.s));
        COEVO_HIP_CHECK(hipEventRecord(ln.done, ln.s));
        return COEVO_OK;
    };

    int rc = COEVO_OK;
    for (int cyc = 0; cyc <= d->n_cycles && rc == COEVO_OK; ++cyc) {
        for (int k = 0; k < K && rc == COEVO_OK; ++k) {
            const coevo_host_cohort &c = d->cohorts[k];
            if (c.n_rows == 0) continue;
            double t0 = timed ? now_us() : 0.0, t1 = t0;
            if (cyc > 0) {
                rc = wait_event(h->lanes[k].done);   // the actions of cycle cyc-1 are in actions_host
                if (rc) break;
                if (timed) {
                    t1 = now_us();
                    acc_wait += t1 - t0;
                    float ms;
                    for (int i = 0; i < 3; ++i)
                        if (hipEventElapsedTime(&ms, h->lanes[k].t[i], h->lanes[k].t[i + 1]) == hipSuccess) acc_gpu[i] += 1e3 * ms;
                    ++acc_n;
                }
            }
            StepJob job{d, &c, cyc - 1, cyc < d->n_cycles ? 1 : 0};
            h->pool.run(step_part, &job);
            const double t2 = timed ? now_us() : 0.0;
            if (cyc < d->n_cycles) rc = enqueue(k);
            if (timed) {
                acc_step += t2 - t1;
                acc_enq += now_us() - t2;
            }
        }
    }
    if (rc != COEVO_OK) {   // leave no launch of ours in flight behind an error
        for (int k = 0; k < K; ++k) (void)hipStreamSynchronize(h->lanes[k].s);
        return rc;
    }
    if (timed) {   // microseconds per cohort-cycle: host wait, host env, host enqueue, then the stream's h2d / launch / d2h
        const double n = acc_n > 0 ? (double)acc_n : 1.0;
        d->phase_us[0] = acc_wait / n;
        d->phase_us[1] = acc_step / n;
        d->phase_us[2] = acc_enq / n;
        d->phase_us[3] = acc_gpu[0] / n;
        d->phase_us[4] = acc_gpu[1] / n;
        d->phase_us[5] = acc_gpu[2] / n;
    }
    return COEVO_OK;
}

COEVO_DEFINE_TU_FLAGS(host_rollout)
