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

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "coevo_common.hip.h"
#include "host_placement.hip.h"

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
        for (int i = 1; i < T; ++i) th.emplace_back([this, i] { worker(i); });
    }
    // worker i -> cpus[i] (cpus[0] is the caller's, pinned per rollout): the placement is chosen once the device is known
    // (host_placement.hip); workers beyond the list stay where the scheduler puts them
    void pin_workers(const std::vector<int> &cpus)
    {
        for (int i = 1; i < T && i < (int)cpus.size(); ++i) {
            cpu_set_t set;
            CPU_ZERO(&set);
            CPU_SET(cpus[(size_t)i], &set);
            (void)pthread_setaffinity_np(th[(size_t)i - 1].native_handle(), sizeof(set), &set);
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
    // ... and the other direction: a cohort's NEXT launch is queued while the core still steps the games, behind a stream
    // wait on a word of the same page (hipStreamWaitValue32, >=), and released by a plain store when the observations stand
    // - no runtime call between the last observation and the launch's start (release -> completion of an empty kernel 6 us,
    // against 16 us for launching it then: round 4 probe, profiles/r04_experiments.md).  COEVO_HOST_PREQUEUE=0 for A/B.
    volatile uint32_t *gates = nullptr;   // one per cohort, 64 bytes apart (second half of the page)
    uint32_t gate_issued[COEVO_MAX_COHORTS] = {0, 0, 0, 0, 0, 0, 0, 0};     // target of the newest queued wait
    uint32_t gate_released[COEVO_MAX_COHORTS] = {0, 0, 0, 0, 0, 0, 0, 0};   // value last stored
    uint32_t wait_target[COEVO_MAX_COHORTS] = {0, 0, 0, 0, 0, 0, 0, 0};     // completion number of the launch that RUNS
    uint32_t pending_seq[COEVO_MAX_COHORTS] = {0, 0, 0, 0, 0, 0, 0, 0};     // ... of the one queued behind its gate
    std::atomic<bool> use_gates{false};
    // where the cores run (host_placement.hip): chosen at the first rollout / allocation, when the device is known
    bool placed = false, pinned = false;
    std::vector<int> cpus;        // cpus[0]: the caller's thread during a rollout, cpus[i]: worker i
    int gpu_node = -1, node_used = -1, place_flags = 0, ctx_index = 0;
    char pin_mode = '1';          // COEVO_HOST_PIN: '1' the GPU's node (default), 'f' ("far") another node on purpose, '0' none
    std::vector<void *> host_allocs;   // coevo_host_rollout_alloc
    explicit HostRollout(int threads) : pool(threads) {}
};

std::atomic<int> g_ctx_counter{0};

// the caller's thread on its CPU for the duration of a rollout / an allocation; the mask it came with is put back
struct CallerPin {
    cpu_set_t saved;
    bool active = false;
    explicit CallerPin(const HostRollout *h)
    {
        if (!h->pinned || h->cpus.empty()) return;
        if (sched_getaffinity(0, sizeof(saved), &saved) != 0) return;
        cpu_set_t set;
        CPU_ZERO(&set);
        CPU_SET(h->cpus[0], &set);
        active = sched_setaffinity(0, sizeof(set), &set) == 0;
    }
    ~CallerPin()
    {
        if (active) (void)sched_setaffinity(0, sizeof(saved), &saved);
    }
};

// choose the context's CPUs (once): the L3 complex of the GPU's NUMA node the affinity mask allows
void ensure_placement(HostRollout *h)
{
    if (h->placed) return;
    h->placed = true;
    const char *pin_env = getenv("COEVO_HOST_PIN");
    h->pin_mode = (pin_env && pin_env[0] == '0') ? '0' : (pin_env && pin_env[0] == 'f') ? 'f' : '1';
    int dev = 0;
    char bdf[64] = {0};
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetPCIBusId(bdf, (int)sizeof(bdf), dev) == hipSuccess)
        h->gpu_node = coevo::numa_node_of_pci(bdf);
    else
        (void)hipGetLastError();
    if (h->pin_mode == '0') return;
    coevo::HostTopology topo;
    coevo::probe_topology(topo);
    const int me = sched_getcpu();
    int want = h->gpu_node;
    if (want < 0 || want >= (int)topo.node_cpus.size() || topo.node_cpus[(size_t)want].empty())
        want = coevo::node_of_cpu(topo, me);   // unknown (a VM that hides it): the caller's own node
    if (h->pin_mode == 'f') {   // A/B: a node that is NOT the wanted one, to reproduce the far-socket step on purpose
        const std::vector<int> allowed = coevo::parse_cpulist(topo.allowed.c_str());
        for (int n = 0; n < (int)topo.node_cpus.size(); ++n) {
            if (n == want) continue;
            bool any = false;
            for (int c : coevo::parse_cpulist(topo.node_cpus[(size_t)n].c_str()))
                if (std::binary_search(allowed.begin(), allowed.end(), c)) { any = true; break; }
            if (any) { want = n; break; }
        }
    }
    const char *lr = getenv("LOCAL_RANK");
    h->ctx_index = 2 * (lr ? atoi(lr) : 0) + (g_ctx_counter.fetch_add(1) & 1);
    const char *node_s = (want >= 0 && want < (int)topo.node_cpus.size()) ? topo.node_cpus[(size_t)want].c_str() : "";
    const int n = coevo::choose_placement(topo.allowed.c_str(), node_s, topo.l3_groups.c_str(), topo.smt_groups.c_str(), me,
                                          h->pool.T, h->ctx_index, h->cpus, h->place_flags);
    h->node_used = (n > 0) ? coevo::node_of_cpu(topo, h->cpus[0]) : -1;
    h->pinned = n > 0;
    if (h->pinned) h->pool.pin_workers(h->cpus);
}


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
            if (hipHostMalloc(&p, 128 * COEVO_MAX_COHORTS, hipHostMallocDefault) == hipSuccess && p) {
                memset(p, 0, 128 * COEVO_MAX_COHORTS);
                h->flags = static_cast<volatile uint32_t *>(p);
                h->gates = h->flags + 16 * COEVO_MAX_COHORTS;
                h->use_flags.store(true);
                int can_wait = 0;
                int dev_now = 0;
                const char *pq = getenv("COEVO_HOST_PREQUEUE");
                if (!(pq && pq[0] == '0') && hipGetDevice(&dev_now) == hipSuccess &&
                    hipDeviceGetAttribute(&can_wait, hipDeviceAttributeCanUseStreamWaitValue, dev_now) == hipSuccess && can_wait)
                    h->use_gates.store(true);
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
    for (void *p : h->host_allocs) (void)hipHostFree(p);
    delete h;
}

extern "C" int coevo_host_rollout_threads(void *handle)
{
    auto *h = static_cast<HostRollout *>(handle);
    return h ? h->pool.T : COEVO_ERR_ARG;
}

// Page-locked, device-mapped host memory for the observation / action staging buffers, allocated FROM the context's first CPU
// and zeroed there: the pages are first touched on the NUMA node the rollout's cores run on (hipHostMallocNumaUser leaves the
// placement to the calling thread; without it the runtime decides).  Freed by coevo_host_rollout_destroy.
extern "C" void *coevo_host_rollout_alloc(void *handle, size_t bytes)
{
    auto *h = static_cast<HostRollout *>(handle);
    if (!h || bytes == 0) return nullptr;
    ensure_placement(h);
    CallerPin pin(h);
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocMapped | hipHostMallocNumaUser) != hipSuccess || !p) {
        (void)hipGetLastError();
        p = nullptr;
        if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess || !p) {
            (void)hipGetLastError();
            return nullptr;
        }
    }
    memset(p, 0, bytes);
    h->host_allocs.push_back(p);
    return p;
}

// what the context chose: cpus_out[0 .. n) (cpus_out[0] = the caller's thread during a rollout), info_out[0..4) = {NUMA node
// of the GPU (-1 unknown), node of the chosen CPUs, COEVO_PLACE_* flags, pinned (0 / 1)}.  Returns n (0: not pinned).
// Before the first rollout / allocation of the context the choice is made now (needs the current device).
extern "C" int coevo_host_rollout_placement(void *handle, int32_t *cpus_out, int max_cpus, int32_t *info_out)
{
    auto *h = static_cast<HostRollout *>(handle);
    if (!h || max_cpus < 0 || (max_cpus > 0 && !cpus_out)) return COEVO_ERR_ARG;
    ensure_placement(h);
    const int n = h->pinned ? (int)h->cpus.size() : 0;
    for (int i = 0; i < n && i < max_cpus; ++i) cpus_out[i] = h->cpus[(size_t)i];
    if (info_out) {
        info_out[0] = h->gpu_node;
        info_out[1] = h->node_used;
        info_out[2] = h->place_flags;
        info_out[3] = h->pinned ? 1 : 0;
    }
    return n;
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
    bool gate_mode = false;                       // launches are queued ahead behind a stream wait, released by a store
    bool flag_mode = false;                       // this rollout signals through the host flags
    bool flag_fallback[COEVO_MAX_COHORTS] = {};   // ... except this cohort, whose last enqueue had to record an event
    bool gate_off[COEVO_MAX_COHORTS] = {};        // ... or whose stream wait was refused: it launches late from then on
    double acc[COEVO_MAX_COHORTS][6];   // per cohort: wait, step, enqueue (host clock), h2d, launch, d2h (HIP events)
    int acc_n[COEVO_MAX_COHORTS];
};

int enqueue_cohort(DriveJob &J, int k, bool gated = false)
{
    const coevo_host_rollout_desc *d = J.d;
    const coevo_host_cohort &c = d->cohorts[k];
    HostLane &ln = J.h->lanes[k];
    const size_t r0 = (size_t)c.row_first;
    if (gated) {   // everything below waits in the stream until release_cohort() stores this target
        const uint32_t target = J.h->gate_issued[k] + 1;
        if (hipStreamWaitValue32(ln.s, const_cast<uint32_t *>(J.h->gates + 16 * k), target, hipStreamWaitValueGte,
                                 0xFFFFFFFFu) != hipSuccess) {
            (void)hipGetLastError();
            return COEVO_ERR_UNSUPPORTED;   // (nothing was queued: the caller falls back to launching late)
        }
        J.h->gate_issued[k] = target;
    }
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
        if (hipStreamWriteValue32(ln.s, const_cast<uint32_t *>(J.h->flags + 16 * k), v, 0) == hipSuccess) {
            if (gated) J.h->pending_seq[k] = v;
            else J.h->wait_target[k] = v;
            return COEVO_OK;
        }
        (void)hipGetLastError();
        if (gated) return COEVO_ERR_HIP;   // (a launch waits at its gate without a completion word: give up, the caller opens the gates)
        J.flag_fallback[k] = true;   // refused here: this cohort signals through its event from now on (this enqueue included)
        J.gate_off[k] = true;        // ... and launches late (an event re-recorded behind a gate would be waited for too early)
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
        const uint32_t want = h->wait_target[k];
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

// the observations of the cohort's queued launch stand: let it go (x86 stores keep their order; the word is the last one)
inline void release_cohort(HostRollout *h, int k)
{
    std::atomic_thread_fence(std::memory_order_release);
    h->gate_released[k] = h->gate_issued[k];
    h->wait_target[k] = h->pending_seq[k];   // the launch that now runs is the one the next wait is for
    h->gates[16 * k] = h->gate_released[k];
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
            bool queued = false;   // this cycle's launch already sits in the stream, behind its gate
            if (J.gate_mode && !J.gate_off[k]) {
                if (cyc == 0 && d->n_cycles > 0) {   // the first launch is queued before the resets are even drawn
                    const int rc = enqueue_cohort(J, k, true);
                    if (rc == COEVO_ERR_UNSUPPORTED) J.gate_off[k] = true;
                    else if (rc) { J.rc.store(rc); return; }
                }
                queued = !J.gate_off[k] && cyc < d->n_cycles;
            }
            if (cyc == 0 && d->reset_ordinals)   // play_game's env.reset() of this cohort's games
                (void)coevo_mpe_host_reset_games(d->state, d->n_games, d->reset_rng, d->reset_ordinals, c.games, 0, c.n_games);
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
                if (queued) {
                    release_cohort(J.h, k);   // the queued launch starts now
                    if (cyc + 1 < d->n_cycles) {   // ... and the next one is queued while it runs
                        const int rc = enqueue_cohort(J, k, true);
                        if (rc == COEVO_ERR_UNSUPPORTED) J.gate_off[k] = true;
                        else if (rc) { J.rc.store(rc); return; }
                    }
                } else {
                    const int rc = enqueue_cohort(J, k);
                    if (rc) { J.rc.store(rc); return; }
                }
            }
            if (J.timed) {
                J.acc[k][1] += t2 - t1;
                J.acc[k][2] += now_us() - t2;
            }
        }
    }
}

}  // namespace

// tests only: leave the context's completion / gate numbering where `value` earlier cycles would have left it (a rollout
// restarts the numbering while its lanes are idle, so it must not matter: ADVICE r4 - near 2^31 / 2^32 a stream wait on ">="
// would otherwise be satisfied by the OLD word and a launch would run on the previous cycle's observations)
extern "C" int coevo_host_rollout_debug_seed_counters(void *handle, uint32_t value)
{
    auto *h = static_cast<HostRollout *>(handle);
    if (!h) return COEVO_ERR_ARG;
    const int rc = ensure_lanes(h);
    if (rc) return rc;
    if (!h->flags) return COEVO_ERR_UNSUPPORTED;
    for (int k = 0; k < COEVO_MAX_COHORTS; ++k) {
        h->flags[16 * k] = value;
        h->gates[16 * k] = value;
        h->seq[k] = h->gate_issued[k] = h->gate_released[k] = h->wait_target[k] = h->pending_seq[k] = value;
    }
    return COEVO_OK;
}

extern "C" int coevo_mpe_host_rollout(void *handle, const coevo_host_rollout_desc *d, void *stream)
{
    auto *h = static_cast<HostRollout *>(handle);
    if (!h || !d || !d->slab || !d->state || !d->game_rows || !d->obs_host || !d->obs_dev || !d->actions_host ||
        !d->actions_dev || !d->status || !d->cohorts)
        return COEVO_ERR_ARG;
    if (d->n_games <= 0 || d->n_rows != 3 * d->n_games || d->n_cycles < 0 || d->n_cycles > (1 << 24) || d->n_cohorts < 1 ||
        d->n_cohorts > h->max_cohorts)
        return COEVO_ERR_ARG;
    const int K = d->n_cohorts;
    {
        const int rc_l = ensure_lanes(h);
        if (rc_l) return rc_l;
    }
    ensure_placement(h);
    CallerPin caller_pin(h);   // the caller's thread next to its workers for this rollout; its own mask comes back at return
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
        if (d->reset_ordinals)
            for (int g = 0; g < d->n_games; ++g)
                if (d->reset_ordinals[g] < 0) return COEVO_ERR_ARG;
    }
    DriveJob J;
    J.h = h;
    J.d = d;
    J.obs = d->obs_dev;
    J.act = d->actions_dev;
    J.timed = d->phase_us != nullptr;
    J.flag_mode = h->use_flags.load();
    // (a timed rollout launches late: its per-launch events would be re-recorded by the pre-queued next launch before they
    // are read; the breakdown it reports is that of the late-launch form, ~10 us of enqueue + launch latency per cycle more)
    J.gate_mode = J.flag_mode && h->use_gates.load() && !J.timed;
    for (int k = 0; k < COEVO_MAX_COHORTS; ++k) {
        J.acc_n[k] = 0;
        for (int i = 0; i < 6; ++i) J.acc[k][i] = 0.0;
    }
    int cus = 0;
    if (hipGetDevice(&J.device) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, J.device) != hipSuccess || cus <= 0)
        return COEVO_ERR_HIP;
    {   // both staging buffers must be page-locked and mapped for this device - zero_copy: the launches address them; staged
        // copies: a hipMemcpyAsync from pageable memory BLOCKS the calling core, here on a stream only that core can release
        void *po = nullptr, *pa = nullptr;
        if (hipHostGetDevicePointer(&po, d->obs_host, 0) != hipSuccess ||
            hipHostGetDevicePointer(&pa, d->actions_host, 0) != hipSuccess || !po || !pa) {
            (void)hipGetLastError();
            return COEVO_ERR_ARG;
        }
        if (d->zero_copy) {
            J.obs = static_cast<const float *>(po);
            J.act = static_cast<int32_t *>(pa);
        }
    }
    // every lane is idle here (the previous rollout saw its last completion word, or drained its streams on error): the
    // sequence numbers restart, so that they never come near a 32-bit wrap (hipStreamWaitValueGte compares as written)
    if (h->flags) {
        for (int k = 0; k < COEVO_MAX_COHORTS; ++k) {
            h->flags[16 * k] = 0;
            h->gates[16 * k] = 0;
            h->seq[k] = h->gate_issued[k] = h->gate_released[k] = h->wait_target[k] = h->pending_seq[k] = 0;
        }
        std::atomic_thread_fence(std::memory_order_seq_cst);
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
    if (rc != COEVO_OK) {   // leave no launch of ours in flight (or waiting at a gate) behind an error
        if (h->gates)
            for (int k = 0; k < K; ++k) release_cohort(h, k);
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


// =====================================================================================================================
// coevo_dqn_host_frames_rollout - the DeepQN games with the env on the HOST: what the reference's play_atari drives through
// env.observe / env.step / env.last (utils/game_logic_functions.py:84-120) with its ALE + SuperSuit env living in host memory
// (:47-53), for all of a rank's games at once.  The env here is the synthetic Atari-shaped one (coevonet_amd/
// atari_synthetic.py; device twin synth_step_kernel, csrc/dqn_engine.hip): per agent-step and cohort the host cores book the
// previous action (hit / zero-sum reward, fp64) and render the next 84 x 84 x C uint8 frame of every live game into a
// page-locked buffer (Philox4x32-10 keyed by game ordinal, step and previous action: the SAME bytes the device twin writes,
// so rewards are bit-identical), the cohort's stream copies the frames up (28 224 bytes per game and step at C = 4), runs
// conv stack + fc1 + output layer (coevo_dqn_forward_argmax) and copies the actions down.  Cohorts alternate: one cohort's
// copy and launches run while the cores render the next cohort's frames.  This is the PCIe-inclusive form of the cfg 4 /
// cfg 5 legs; the device-resident form (frames never leave HBM) is the measured headline of those configs.
namespace {

constexpr uint32_t PHILOX_M0 = 0xD2511F53u, PHILOX_M1 = 0xCD9E8D57u, PHILOX_W0 = 0x9E3779B9u, PHILOX_W1 = 0xBB67AE85u;

inline void philox10_host(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4])
{
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)PHILOX_M0 * c0, p1 = (uint64_t)PHILOX_M1 * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += PHILOX_W0; k1 += PHILOX_W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// 16 bytes per counter i = 0 .. n16-1 (synth_frame of atari_synthetic.py / the frame loop of synth_step_kernel), eight
// counters at a time in struct-of-arrays form so that the host compiler turns the rounds into 8-wide vector code
#define COEVO_FRAME_BODY(W)                                                                                            \
    for (int i0 = 0; i0 < n16; i0 += W) {                                                                             \
        uint32_t a[W], b[W], c[W], d[W];                                                                              \
        for (int j = 0; j < W; ++j) { a[j] = (uint32_t)(i0 + j); b[j] = c1; c[j] = c2; d[j] = c3; }                   \
        uint32_t ka = k0, kb = k1;                                                                                    \
        for (int r = 0; r < 10; ++r) {                                                                                \
            for (int j = 0; j < W; ++j) {                                                                             \
                const uint64_t p0 = (uint64_t)PHILOX_M0 * a[j], p1 = (uint64_t)PHILOX_M1 * c[j];                      \
                const uint32_t n0 = (uint32_t)(p1 >> 32) ^ b[j] ^ ka, n2 = (uint32_t)(p0 >> 32) ^ d[j] ^ kb;          \
                a[j] = n0; b[j] = (uint32_t)p1; c[j] = n2; d[j] = (uint32_t)p0;                                       \
            }                                                                                                         \
            ka += PHILOX_W0; kb += PHILOX_W1;                                                                         \
        }                                                                                                             \
        const int lim = n16 - i0 < W ? n16 - i0 : W;                                                                  \
        uint32_t *o = reinterpret_cast<uint32_t *>(dst) + 4 * (size_t)i0;                                             \
        for (int j = 0; j < lim; ++j) { o[4 * j] = a[j]; o[4 * j + 1] = b[j]; o[4 * j + 2] = c[j]; o[4 * j + 3] = d[j]; } \
    }

#if !defined(__HIP_DEVICE_COMPILE__)
__attribute__((target("avx2"))) void synth_frame_avx2(uint8_t *dst, int n16, uint32_t c1, uint32_t c2, uint32_t c3,
                                                      uint32_t k0, uint32_t k1)
{
    COEVO_FRAME_BODY(8)
}
__attribute__((target("avx512f,avx512vl,avx512dq"))) void synth_frame_avx512(
    uint8_t *dst, int n16, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1)
{
    COEVO_FRAME_BODY(16)
}
#endif
void synth_frame_base(uint8_t *dst, int n16, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1)
{
    COEVO_FRAME_BODY(8)
}

inline void synth_frame_host(uint8_t *dst, int nbytes, uint64_t seed, int64_t ordinal, int t, uint32_t last)
{
    const uint32_t c1 = (uint32_t)t | (last << 16), c2 = (uint32_t)ordinal,
                   c3 = (uint32_t)((uint64_t)ordinal >> 32) ^ 0x66726d65u;
#if !defined(__HIP_DEVICE_COMPILE__)
    static const bool avx512 = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512vl") &&
                               __builtin_cpu_supports("avx512dq");
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx512) {
        synth_frame_avx512(dst, nbytes / 16, c1, c2, c3, (uint32_t)seed, (uint32_t)(seed >> 32));
        return;
    }
    if (avx2) {
        synth_frame_avx2(dst, nbytes / 16, c1, c2, c3, (uint32_t)seed, (uint32_t)(seed >> 32));
        return;
    }
#endif
    synth_frame_base(dst, nbytes / 16, c1, c2, c3, (uint32_t)seed, (uint32_t)(seed >> 32));
}

inline uint32_t synth_target_host(uint64_t seed, int64_t ordinal, int t, int n_actions)
{
    uint32_t o[4];
    philox10_host(0xFFFFFFFFu, (uint32_t)t, (uint32_t)ordinal, (uint32_t)((uint64_t)ordinal >> 32) ^ 0x74617267u,
                  (uint32_t)seed, (uint32_t)(seed >> 32), o);
    return o[0] % (uint32_t)n_actions;
}

struct FrameJob {
    const coevo_frames_rollout_desc *d;
    const coevo_frame_cohort *c;
    int t;
};

// the env's share of agent-step t for a slice of a cohort's games: book the action of step t-1, render frame t
void frame_part(void *arg, int part, int parts)
{
    const FrameJob *j = static_cast<const FrameJob *>(arg);
    const coevo_frames_rollout_desc *d = j->d;
    const coevo_frame_cohort &c = *j->c;
    const int t = j->t, p = t & 1, q = (t - 1) & 1;
    const int nbytes = 84 * 84 * (d->C & 0xff);
    const int lo = (int)((int64_t)c.n_games * part / parts), hi = (int)((int64_t)c.n_games * (part + 1) / parts);
    for (int i = lo; i < hi; ++i) {
        const int g = c.game_first + i;
        const int64_t ordinal = d->game_ordinal0[g] + d->generation * d->ordinals_per_gen;
        const int lim = ordinal < 0 ? 0 : d->limit[g];
        int last = d->game_state[4 * g], prev_hit = d->game_state[4 * g + 1];
        if (t == 0) {
            last = 0xFF; prev_hit = 0;
            d->acc[3 * (size_t)g] = 0.0; d->acc[3 * (size_t)g + 1] = 0.0; d->acc[3 * (size_t)g + 2] = 0.0;
        } else if (t - 1 < lim) {
            const int a = c.actions_host[c.rows[q][i]];
            const int hit = ((uint32_t)a == synth_target_host(d->seed, ordinal, t - 1, d->n_actions)) ? 1 : 0;
            const size_t slot = 3 * (size_t)g + ((t - 1) & 1);
            d->acc[slot] = d->acc[slot] + (double)(prev_hit - hit);
            prev_hit = hit;
            last = a;
        }
        d->game_state[4 * g] = last;
        d->game_state[4 * g + 1] = prev_hit;
        if (t < d->T && t < lim)
            synth_frame_host(c.frames_host + (size_t)c.rows[p][i] * nbytes, nbytes, d->seed, ordinal, t, (uint32_t)last);
    }
}

}  // namespace

extern "C" int coevo_dqn_host_frames_rollout(void *handle, const coevo_frames_rollout_desc *d, void *stream)
{
    auto *h = static_cast<HostRollout *>(handle);
    if (!h || !d || !d->slab || !d->status || !d->game_state || !d->acc || !d->game_ordinal0 || !d->limit || !d->cohorts)
        return COEVO_ERR_ARG;
    const int C = d->C & 0xff;   // (d->C may carry COEVO_DQN_FC1_TILED for the forward launches)
    if (d->n_games <= 0 || d->T < 0 || d->T > 65535 || C < 1 || C > 6 || (d->C & ~(0xff | COEVO_DQN_FC1_TILED)) ||
        d->n_actions < 1 || d->n_cohorts < 1 || d->n_cohorts > h->max_cohorts)
        return COEVO_ERR_ARG;
    const int K = d->n_cohorts;
    const size_t nbytes = (size_t)84 * 84 * C;
    int covered = 0;
    for (int k = 0; k < K; ++k) {
        const coevo_frame_cohort &c = d->cohorts[k];
        if (c.n_games <= 0 || c.game_first != covered || !c.frames_host || !c.frames_dev || !c.actions_host || !c.actions_dev ||
            !c.workspace)
            return COEVO_ERR_ARG;
        for (int p = 0; p < 2; ++p) {
            if (!c.tasks[p] || !c.rows[p] || c.n_tasks[p] <= 0 || c.max_rows[p] < 1) return COEVO_ERR_ARG;
            for (int i = 0; i < c.n_games; ++i)
                if (c.rows[p][i] < 0 || c.rows[p][i] >= c.n_games) return COEVO_ERR_ARG;
        }
        covered += c.n_games;
    }
    if (covered != d->n_games) return COEVO_ERR_ARG;
    {
        const int rc_l = ensure_lanes(h);
        if (rc_l) return rc_l;
    }
    ensure_placement(h);
    CallerPin caller_pin(h);
    const bool timed = d->phase_us != nullptr;
    double acc_host = 0.0, acc_wait = 0.0, acc_gpu[3] = {0.0, 0.0, 0.0};
    int acc_n = 0;
    COEVO_HIP_CHECK(hipEventRecord(h->start, (hipStream_t)stream));
    for (int k = 0; k < K; ++k) COEVO_HIP_CHECK(hipStreamWaitEvent(h->lanes[k].s, h->start, 0));
    int rc = COEVO_OK;
    for (int t = 0; t <= d->T && rc == COEVO_OK; ++t) {
        for (int k = 0; k < K && rc == COEVO_OK; ++k) {
            const coevo_frame_cohort &c = d->cohorts[k];
            HostLane &ln = h->lanes[k];
            const double t0 = timed ? now_us() : 0.0;
            if (t > 0) {   // the actions of step t-1 are in actions_host
                rc = wait_event(ln.done);
                if (rc) break;
                if (timed) {
                    float ms;
                    for (int i = 0; i < 3; ++i)
                        if (hipEventElapsedTime(&ms, ln.t[i], ln.t[i + 1]) == hipSuccess) acc_gpu[i] += 1e3 * ms;
                    ++acc_n;
                }
            }
            const double t1 = timed ? now_us() : 0.0;
            FrameJob job{d, &c, t};
            h->pool.run(frame_part, &job);
            if (timed) {
                acc_wait += t1 - t0;
                acc_host += now_us() - t1;
            }
            if (t == d->T) continue;
            const int p = t & 1;
            // (a HIP error leaves the loop through `rc`, so that the lanes are drained below before the caller - who may free
            // frames_host / actions_host on an error - sees it)
            auto ok = [&](hipError_t e) { if (e != hipSuccess && rc == COEVO_OK) rc = COEVO_ERR_HIP; return rc == COEVO_OK; };
            if (timed && !ok(hipEventRecord(ln.t[0], ln.s))) break;
            if (!ok(hipMemcpyAsync(c.frames_dev, c.frames_host, (size_t)c.n_games * nbytes, hipMemcpyHostToDevice, ln.s))) break;
            if (timed && !ok(hipEventRecord(ln.t[1], ln.s))) break;
            rc = coevo_dqn_forward_argmax(d->slab, c.tasks[p], c.n_tasks[p], c.max_rows[p], c.n_games, d->C, d->n_actions,
                                          c.frames_dev, c.actions_dev, nullptr, d->status, c.workspace, ln.s);
            if (rc) break;
            if (timed && !ok(hipEventRecord(ln.t[2], ln.s))) break;
            if (!ok(hipMemcpyAsync(c.actions_host, c.actions_dev, (size_t)c.n_games * sizeof(int32_t), hipMemcpyDeviceToHost,
                                   ln.s)))
                break;
            if (timed && !ok(hipEventRecord(ln.t[3], ln.s))) break;
            if (!ok(hipEventRecord(ln.done, ln.s))) break;
        }
    }
    if (rc != COEVO_OK) {
        for (int k = 0; k < K; ++k) (void)hipStreamSynchronize(h->lanes[k].s);
        return rc;
    }
    if (timed) {   // mean microseconds per cohort-step: host wait, host env (book + render), frames up, forward, actions down
        const double n = acc_n > 0 ? (double)acc_n : 1.0;
        d->phase_us[0] = acc_wait / n;
        d->phase_us[1] = acc_host / n;
        d->phase_us[2] = acc_gpu[0] / n;
        d->phase_us[3] = acc_gpu[1] / n;
        d->phase_us[4] = acc_gpu[2] / n;
    }
    return COEVO_OK;
}

// one frame on the calling thread: the bytes coevo_dqn_host_frames_rollout renders (tests: == atari_synthetic.synth_frame)
extern "C" int coevo_synth_frame_host(uint8_t *dst, int C, uint64_t seed, int64_t ordinal, int t, int last_action)
{
    if (!dst || C < 1 || C > 6 || t < 0 || t > 65535) return COEVO_ERR_ARG;
    synth_frame_host(dst, 84 * 84 * C, seed, ordinal, t, (uint32_t)last_action);
    return COEVO_OK;
}

COEVO_DEFINE_TU_FLAGS(host_rollout)
