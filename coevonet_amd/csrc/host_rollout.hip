// coevo_mpe_host_rollout - north_star's first configuration as ONE C-ABI call: "vectorised env stepping runs on the host
// cores", the policy steps on the GPU.  Replaces, for all of a rank's games at once, the per-game loop of play_MPE
// (utils/game_logic_functions.py:123-212: env.observe :138, the forward :152-163, env.step :179, env.last :181,
// rewards[agent] += reward :190, the limit / truncation breaks :197-204).
//
// The games are cut into K cohorts (contiguous row ranges of the observation / action buffers), the cohorts dealt to T host
// cores (the caller's thread + T-1 workers of the context).  Per env-cycle and cohort its core
//   - polls the cohort's event (the actions of the previous cycle are back), steps the cohort's games with them and writes
//     their observations (the env bodies of csrc/mpe_env.hip, fp64, -ffp-contract=off on both sides),
//   - enqueues on the cohort's own stream: obs rows host -> device, ONE policy launch (coevo_fc_forward_merged, or one launch
//     per task table when the lean merged kernel does not hold them), action rows device -> host, the event
//     (zero_copy: the launch reads / writes the mapped page-locked buffers itself - same PCIe bytes, no copy engine),
// and turns to its next cohort: while one cohort is on the GPU the core steps another, the cores run independently of each
// other, and the cohorts' launches run side by side on the GPU.  Nothing synchronises a whole stream or the device.  Results
// do not depend on K or T: every game is stepped by exactly one thread, by the same code.
#include <pthread.h>
#include <sched.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "coevo_common.hip.h"

namespace {

// T host cores on one job at a time: thread t runs part t of T (the caller is part 0).  The split is STATIC on purpose: part t
// of a cohort's game list is stepped by the same core every cycle, so the games' state stays in that core's cache - handing
// out chunks dynamically made two threads four times SLOWER than one on the 256-core host of a GPU box (every line of the
// 1 MB of game state migrating between cores each cycle: profiles/r04_experiments.md).  Signalling costs one cache-line
// transfer per worker each way: the job number on one line all workers poll, a completion word per worker on its own line.
struct HostPool {
    typedef void (*fn_t)(void *, int, int);   // fn(arg, part, parts)
    struct alignas(64) Slot {
        std::atomic<uint64_t> done{0};        // the last job this worker completed
    };
    int T = 1;
    std::vector<std::thread> th;
    std::mutex m;
    std::condition_variable cv;
    alignas(64) std::atomic<uint64_t> epoch{0};
    alignas(64) std::atomic<bool> stop{false};
    fn_t fn = nullptr;
    void *arg = nullptr;
    std::vector<Slot> slots;

    explicit HostPool(int threads) : T(threads < 1 ? 1 : threads), slots((size_t)(threads < 1 ? 1 : threads))
    {
        // workers next to the creating thread: the cores of its own 8-core complex share an L3 and a memory controller (on
        // the 2-socket host of a GPU box an unpinned worker landed on the far socket and stepped its games at half speed).
        // COEVO_HOST_PIN=0 leaves the placement to the scheduler; a core outside the affinity mask is simply not taken.
        const char *pin_env = getenv("COEVO_HOST_PIN");
        const bool pin = !(pin_env && pin_env[0] == '0');
        const int me = sched_getcpu();
        for (int i = 1; i < T; ++i) {
            th.emplace_back([this, i] { worker(i); });
            if (pin && me >= 0 && T <= 8) {
                const int base = me & ~7;
                cpu_set_t set;
                CPU_ZERO(&set);
                CPU_SET(base + ((me - base + i) & 7), &set);
                (void)pthread_setaffinity_np(th.back().native_handle(), sizeof(set), &set);
            }
        }
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
            // between two cycles of a rollout the next job is tens of microseconds away: poll (plain loads - under a
            // hypervisor a long `pause` loop gets the virtual core descheduled for milliseconds), sleep only when nothing
            // came for 2 ms (between rollouts)
            bool got = false;
            const auto t_idle = std::chrono::steady_clock::now();
            for (unsigned spin = 0;; ++spin) {
                if (epoch.load(std::memory_order_acquire) != seen || stop.load(std::memory_order_relaxed)) {
                    got = true;
                    break;
                }
                if ((spin & 1023u) == 1023u && std::chrono::steady_clock::now() - t_idle > std::chrono::milliseconds(2)) break;
            }
            if (!got) {
                std::unique_lock<std::mutex> g(m);
                cv.wait(g, [&] { return epoch.load(std::memory_order_acquire) != seen || stop.load(std::memory_order_acquire); });
            }
            if (stop.load(std::memory_order_acquire)) return;
            seen = epoch.load(std::memory_order_acquire);
            fn(arg, i, T);
            slots[(size_t)i].done.store(seen, std::memory_order_release);
        }
    }
    // fn(arg, part, T) on every thread (the caller is part 0); returns when all parts are done
    void run(fn_t f, void *a)
    {
        if (T == 1) {
            f(a, 0, 1);
            return;
        }
        uint64_t e;
        {
            std::lock_guard<std::mutex> g(m);   // (orders fn / arg before the job number for a worker inside cv.wait)
            fn = f;
            arg = a;
            e = epoch.load(std::memory_order_relaxed) + 1;
            epoch.store(e, std::memory_order_release);
        }
        cv.notify_all();
        f(a, 0, T);
        for (int i = 1; i < T; ++i)
            while (slots[(size_t)i].done.load(std::memory_order_acquire) != e) {}
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
    // completion signal of a cohort-cycle: a 32-bit sequence number the stream writes into page-locked host memory behind
    // the launch (hipStreamWriteValue32) - the core polls plain memory, no runtime call (and none of the runtime's locks)
    // while it waits; events (hipEventQuery polls) when COEVO_HOST_SIGNAL=event or the write command is refused
    volatile uint32_t *flags = nullptr;   // one per cohort, 64 bytes apart
    uint32_t seq[COEVO_MAX_COHORTS] = {0, 0, 0, 0, 0, 0, 0, 0};
    std::atomic<bool> use_flags{false};
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
    // slices in units of 8 games: one 64-byte line of every fp64 state field belongs to one thread
    const int u = (n + 7) / 8;
    const int lo8 = (int)((int64_t)u * part / parts) * 8, hi8 = (int)((int64_t)u * (part + 1) / parts) * 8;
    const int lo = lo8 < n ? lo8 : n, hi = hi8 < n ? hi8 : n;
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
    if (!h->start) {
        COEVO_HIP_CHECK(hipEventCreateWithFlags(&h->start, hipEventDisableTiming));
        const char *sig = getenv("COEVO_HOST_SIGNAL");   // "event": A/B runs (374 vs 364 generations/s with two cores)
        if (!(sig && sig[0] == 'e')) {
            void *p = nullptr;
            if (hipHostMalloc(&p, 64 * COEVO_MAX_COHORTS, hipHostMallocDefault) == hipSuccess && p) {
                memset(p, 0, 64 * COEVO_MAX_COHORTS);
                h->flags = static_cast<volatile uint32_t *>(p);
                h->use_flags.store(true);
            } else {
                (void)hipGetLastError();
            }
        }
    }
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
    if (h->flags) (void)hipHostFree(const_cast<uint32_t *>(h->flags));
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

namespace {

struct DriveJob {
    HostRollout *h;
    const coevo_host_rollout_desc *d;
    const float *obs;        // what the launches read / write: the device twins, or the mapped host buffers (zero_copy)
    int32_t *act;
    int device;
    bool lean, timed;
    std::atomic<int> rc{COEVO_OK};
    bool flag_mode = false;                       // this rollout signals through the host flags
    bool flag_fallback[COEVO_MAX_COHORTS] = {};   // ... except this cohort, whose last enqueue had to record an event
    double acc[COEVO_MAX_COHORTS][6];   // per cohort: wait, step, enqueue (host clock), h2d, launch, d2h (HIP events)
    int acc_n[COEVO_MAX_COHORTS];
};

int enqueue_cohort(DriveJob &J, int k)
{
    const coevo_host_rollout_desc *d = J.d;
    const coevo_host_cohort &c = d->cohorts[k];
    HostLane &ln = J.h->lanes[k];
    const size_t r0 = (size_t)c.row_first;
    if (J.timed) COEVO_HIP_CHECK(hipEventRecord(ln.t[0], ln.s));
    if (!d->zero_copy)
        COEVO_HIP_CHECK(hipMemcpyAsync(d->obs_dev + r0 * COEVO_OBS_STRIDE, d->obs_host + r0 * COEVO_OBS_STRIDE,
                                       (size_t)c.n_rows * COEVO_OBS_STRIDE * sizeof(float), hipMemcpyHostToDevice, ln.s));
    if (J.timed) COEVO_HIP_CHECK(hipEventRecord(ln.t[1], ln.s));
    int rc = COEVO_OK;
    if (J.lean) {
        rc = coevo_fc_forward_merged(d->slab, c.heavy, c.n_heavy, c.heavy_max_rows, c.light, c.n_light, c.light_max_rows,
                                     J.obs, J.act, nullptr, d->status, ln.s);
    } else {
        if (c.n_heavy > 0)
            rc = coevo_fc_forward_argmax(d->slab, c.heavy, c.n_heavy, c.heavy_max_rows, J.obs, J.act, nullptr, d->status, ln.s);
        if (rc == COEVO_OK && c.n_light > 0)
            rc = coevo_fc_forward_argmax(d->slab, c.light, c.n_light, c.light_max_rows, J.obs, J.act, nullptr, d->status, ln.s);
    }
    if (rc) return rc;
    if (J.timed) COEVO_HIP_CHECK(hipEventRecord(ln.t[2], ln.s));
    if (!d->zero_copy)
        COEVO_HIP_CHECK(hipMemcpyAsync(d->actions_host + r0, d->actions_dev + r0, (size_t)c.n_rows * sizeof(int32_t),
                                       hipMemcpyDeviceToHost, ln.s));
    if (J.timed) COEVO_HIP_CHECK(hipEventRecord(ln.t[3], ln.s));
    if (J.flag_mode && !J.flag_fallback[k]) {
        const uint32_t v = ++J.h->seq[k];
        if (hipStreamWriteValue32(ln.s, const_cast<uint32_t *>(J.h->flags + 16 * k), v, 0) == hipSuccess) return COEVO_OK;
        (void)hipGetLastError();
        J.flag_fallback[k] = true;   // refused here: this cohort signals through its event from now on (this enqueue included)
        J.h->use_flags.store(false);
    }
    COEVO_HIP_CHECK(hipEventRecord(ln.done, ln.s));
    return COEVO_OK;
}

// the cohort's previous enqueue has completed: its actions are in actions_host
int wait_cohort(DriveJob &J, int k)
{
    HostRollout *h = J.h;
    if (h->flags && !J.flag_fallback[k] && J.flag_mode) {
        const volatile uint32_t *f = h->flags + 16 * k;
        const uint32_t want = h->seq[k];
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned spin = 0;; ++spin) {
            if (*f == want) {
                std::atomic_thread_fence(std::memory_order_acquire);
                return COEVO_OK;
            }
            if ((spin & 0xffffu) == 0xffffu && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(20))
                return COEVO_ERR_HIP;   // a launch that never completes: report, do not hang the caller
        }
    }
    return wait_event(h->lanes[k].done);
}

// One host core's share of a rollout: it DRIVES cohorts part, part + parts, ... from the first observation to the last
// world step - wait for a cohort's actions, step its games, write their observations, enqueue its next copies + launch -
// and alternates between its cohorts, so that one is on the GPU while it works on the other.  The cores never wait for each
// other (a per-cycle fork / join across cores cost ~15 us on the 2-socket host of a GPU box, as much as stepping 1000 games),
// and a cohort's game state stays in the cache of the core that owns it, cycle after cycle and generation after generation.
void drive_part(void *arg, int part, int parts)
{
    DriveJob &J = *static_cast<DriveJob *>(arg);
    const coevo_host_rollout_desc *d = J.d;
    const int K = d->n_cohorts;
    if (part >= K) return;
    if (part > 0 && hipSetDevice(J.device) != hipSuccess) {   // (the worker threads start on device 0)
        J.rc.store(COEVO_ERR_HIP);
        return;
    }
    for (int cyc = 0; cyc <= d->n_cycles; ++cyc) {
        if (J.rc.load(std::memory_order_relaxed) != COEVO_OK) return;   // another core failed: stop enqueueing
        for (int k = part; k < K; k += parts) {
            const coevo_host_cohort &c = d->cohorts[k];
            if (c.n_rows == 0) continue;
            const double t0 = J.timed ? now_us() : 0.0;
            double t1 = t0;
            if (cyc > 0) {
                const int rc = wait_cohort(J, k);   // the actions of cycle cyc-1 are in actions_host
                if (rc) { J.rc.store(rc); return; }
                if (J.timed) {
                    t1 = now_us();
                    J.acc[k][0] += t1 - t0;
                    float ms;
                    for (int i = 0; i < 3; ++i)
                        if (hipEventElapsedTime(&ms, J.h->lanes[k].t[i], J.h->lanes[k].t[i + 1]) == hipSuccess)
                            J.acc[k][3 + i] += 1e3 * ms;
                    ++J.acc_n[k];
                }
            }
            if (cyc > 0) {   // the device has just written these rows: pull them in as one burst, not miss by miss
                const char *a0 = reinterpret_cast<const char *>(d->actions_host + c.row_first);
                for (size_t b = 0; b < (size_t)c.n_rows * sizeof(int32_t); b += 64) __builtin_prefetch(a0 + b, 0, 3);
            }
            (void)coevo_mpe_host_step_games(d->state, d->n_games, d->game_rows, d->actions_host, cyc - 1, d->game_limit,
                                            d->pos_first, c.games, 0, c.n_games, cyc < d->n_cycles ? 1 : 0, d->obs_host);
            const double t2 = J.timed ? now_us() : 0.0;
            if (cyc < d->n_cycles) {
                const int rc = enqueue_cohort(J, k);
                if (rc) { J.rc.store(rc); return; }
            }
            if (J.timed) {
                J.acc[k][1] += t2 - t1;
                J.acc[k][2] += now_us() - t2;
            }
        }
    }
}

}  // namespace

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
    DriveJob J;
    J.h = h;
    J.d = d;
    J.obs = d->obs_dev;
    J.act = d->actions_dev;
    J.timed = d->phase_us != nullptr;
    J.flag_mode = h->use_flags.load();
    for (int k = 0; k < COEVO_MAX_COHORTS; ++k) {
        J.acc_n[k] = 0;
        for (int i = 0; i < 6; ++i) J.acc[k][i] = 0.0;
    }
    int cus = 0;
    if (hipGetDevice(&J.device) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, J.device) != hipSuccess || cus <= 0)
        return COEVO_ERR_HIP;
    if (d->zero_copy) {   // the launches address the page-locked host buffers themselves: they must be mapped for this device
        void *po = nullptr, *pa = nullptr;
        if (hipHostGetDevicePointer(&po, d->obs_host, 0) != hipSuccess ||
            hipHostGetDevicePointer(&pa, d->actions_host, 0) != hipSuccess || !po || !pa) {
            (void)hipGetLastError();
            return COEVO_ERR_ARG;
        }
        J.obs = static_cast<const float *>(po);
        J.act = static_cast<int32_t *>(pa);
    }
    int wgs = 0;
    J.lean = true;   // the lean merged kernel needs every workgroup of every cohort in flight resident at four per CU
    for (int k = 0; k < K; ++k) {
        const coevo_host_cohort &c = d->cohorts[k];
        if (c.n_rows == 0) continue;
        wgs += c.n_heavy + c.n_light;
        if (c.n_heavy <= 0 || c.n_light <= 0 || c.heavy_max_rows > 16 || c.light_max_rows > 8) J.lean = false;
    }
    if (wgs > 4 * cus) J.lean = false;

    // the cohort streams see everything the caller's stream has enqueued so far (the offspring written into the slab)
    COEVO_HIP_CHECK(hipEventRecord(h->start, (hipStream_t)stream));
    for (int k = 0; k < K; ++k) COEVO_HIP_CHECK(hipStreamWaitEvent(h->lanes[k].s, h->start, 0));

    h->pool.run(drive_part, &J);

    const int rc = J.rc.load();
    if (rc != COEVO_OK) {   // leave no launch of ours in flight behind an error
        for (int k = 0; k < K; ++k) (void)hipStreamSynchronize(h->lanes[k].s);
        return rc;
    }
    if (J.timed) {   // mean microseconds per cohort-cycle: host wait, host env, host enqueue, then the stream's h2d / launch / d2h
        double tot[6] = {0, 0, 0, 0, 0, 0};
        int n = 0;
        for (int k = 0; k < K; ++k) {
            n += J.acc_n[k];
            for (int i = 0; i < 6; ++i) tot[i] += J.acc[k][i];
        }
        for (int i = 0; i < 6; ++i) d->phase_us[i] = tot[i] / (n > 0 ? n : 1);
    }
    return COEVO_OK;
}

COEVO_DEFINE_TU_FLAGS(host_rollout)
