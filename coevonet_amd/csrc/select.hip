// K6/K7 - fitness sharing, GA fitness and ranking on the device, so that selection never waits on the host.
//
// Replaces (reference file:line): diversity_penalty (utils/game_logic_functions.py:12-37) as called from
// genetic_algorithm.py:131-133 / evolutionary_strategy.py:128; the GA fitness expression
// (genetic_algorithm.py:140-146, :172-178, :205-211) and np.argsort(fitness)[::-1] (:223-225).
#include "coevo_common.hip.h"
#include <stdio.h>

namespace coevo {

// dist[i] = || w_i - w_ref ||_2 over the Linear weights and biases only (get_weights_ES default layers,
// MPE/fcnetwork.py:161: fc1, fc2, output).  One workgroup per population member; squares accumulate in fp64 so the
// fp32 result does not depend on the summation order.
__global__ __launch_bounds__(256) void fc_distance_kernel(const float *ref_net, const float *pop_slab, int D,
                                                           float *dist)
{
    __shared__ double scratch[4];
    const int64_t stride = fc_stride(D), P = fc_params(D);
    const int64_t g1 = fc_off_b1(D) + H1, g2 = fc_off_b2(D) + H2;
    const float *wi = pop_slab + (int64_t)blockIdx.x * stride;
    double acc = 0.0;
    for (int64_t s0 = (int64_t)threadIdx.x * 4; s0 < stride; s0 += 1024) {
        const float4 a = *reinterpret_cast<const float4 *>(wi + s0);
        const float4 b = *reinterpret_cast<const float4 *>(ref_net + s0);
        const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int64_t s = s0 + c;
            const bool ln = (s >= g1 && s < g1 + 2 * H1) || (s >= g2 && s < g2 + 2 * H2);
            if (s < P && !ln) {
                const float d = av[c] - bv[c];
                acc += (double)d * (double)d;
            }
        }
    }
    const double tot = block_sum_f64(acc, scratch);
    if (threadIdx.x == 0) dist[blockIdx.x] = (float)sqrt(tot);
}

// dist[first + c] = sqrt(sum_b partial[c][b]) - finishes the squared distances the perturb kernel accumulated per
// block; one wavefront per child (lane-strided partial sums, then the xor tree: a fixed order).  If head is given,
// dist[first - 1] = *head (the unchanged best individual keeps the distance it had in the previous population).
__global__ __launch_bounds__(64) void dist_finalize_kernel(const double *partial, int n_blocks, float *dist, int first,
                                                            const float *head)
{
    const int c = blockIdx.x, l = threadIdx.x;
    double v = 0.0;
    for (int b = l; b < n_blocks; b += 64) v += partial[(size_t)c * n_blocks + b];
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v = v + __shfl_xor(v, m, 64);
    if (l == 0) {
        dist[first + c] = (float)sqrt(v);
        if (c == 0 && head) dist[first - 1] = *head;
    }
}

static_assert(sizeof(coevo_fc_finalize_job) == 40, "layout mirrored by coevonet_amd/lib.py FinalizeJob");
struct FinalizeJobs { coevo_fc_finalize_job j[COEVO_MAX_JOBS]; };
__global__ __launch_bounds__(64) void dist_finalize_multi_kernel(FinalizeJobs jobs, int32_t *tick)
{
    const coevo_fc_finalize_job &jb = jobs.j[blockIdx.y];
    const int c = blockIdx.x, l = threadIdx.x;
    // (the generation counter's tick rides in the generation's last launch: nothing in this launch reads it)
    if (tick && blockIdx.x == 0 && blockIdx.y == 0 && l == 0) *tick += 1;
    if (c >= jb.n) return;
    double v = 0.0;
    for (int b = l; b < jb.n_blocks; b += 64) v += jb.dist_partial[(size_t)c * jb.n_blocks + b];
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v = v + __shfl_xor(v, m, 64);
    if (l == 0) {
        jb.dist[jb.first + c] = (float)sqrt(v);
        if (c == 0 && jb.head) jb.dist[jb.first - 1] = *jb.head;
    }
}

__global__ void gather_f32_kernel(float *dst, const float *src, const int32_t *idx, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}

// score = sum_i max(0, 1 - d_i / mean(d)), mean rounded to fp32 as np.mean of an fp32 array is
__global__ __launch_bounds__(256) void sharing_score_kernel(const float *dist, int n, float *score)
{
    __shared__ double scratch[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += (double)dist[i];
    const float sigma = (float)(block_sum_f64(s, scratch) / (double)n);
    double sc = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float sh = 1.0f - dist[i] / sigma;
        if (sh > 0.0f) sc += (double)sh;
    }
    const double tot = block_sum_f64(sc, scratch);
    if (threadIdx.x == 0) *score = (float)tot;
}

// quirk Q2: the reward variable is overwritten by every HoF game, only the LAST one (k = hof-1) survives; it is
// still divided by hof_size and by (1 + diversity).  numpy >= 2 evaluates python_float / np.float32 in float32.
__global__ void ga_fitness_kernel(const double *rewards, int game_first, int pop, int games_per_individual,
                                  int hof, int slot, const float *diversity, float *fitness)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= pop) return;
    const int gpi = games_per_individual;
    const double last = rewards[3 * (size_t)(game_first + i * gpi + gpi - 1) + slot];
    const float total = (float)(last / (double)hof);
    fitness[i] = total / (1.0f + *diversity);
}

// order = argsort(fitness)[::-1] with a stable ascending sort (ties: higher index first after the reversal);
// NaN sorts last in ascending order like numpy.  Rank counting, n <= 4096, one workgroup.
__device__ inline bool rank_less(float a, int ia, float b, int ib)
{
    const bool an = __builtin_isnan(a), bn = __builtin_isnan(b);
    if (an || bn) return (!an && bn) || (an && bn && ia < ib);
    return (a < b) || (a == b && ia < ib);
}

__global__ __launch_bounds__(256) void rank_desc_kernel(const float *fitness, int n, int32_t *order)
{
    __shared__ float f[4096];
    for (int i = threadIdx.x; i < n; i += 256) f[i] = fitness[i];
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 256) {
        const float fi = f[i];
        int rank = 0;
        for (int j = 0; j < n; ++j) rank += rank_less(f[j], j, fi, i) ? 1 : 0;
        order[n - 1 - rank] = i;
    }
}

// ---- the three per-role selection launches (score, fitness, rank) + the best individual's distance, for up to three
// roles in ONE launch: block r serves role r with exactly the arithmetic of the kernels above -----------------------
struct GaSelectArgs {
    coevo_ga_select_role role[3];
    int pop, games_per_individual, hof;
    // gathered form (coevo_ga_select_gathered): the all-gathered buffer [world][n_roles][n_local][4] fp64 = per rank, role and
    // local individual {reward triple of its last HoF game, distance to the stale agent}; individual i lives on rank
    // i / n_local.  NULL: distances / rewards through the role's own pointers.
    const double *gathered;
    int n_local, n_roles;
    // the sigma rule in the same launch (coevo_ga_select_adapt): block (n_roles, 0) runs it beside the roles' blocks
    coevo_ga_adapt_args adapt;
    int with_adapt, n_roles_launched;
};

static_assert(sizeof(coevo_ga_adapt_args) == 88, "layout mirrored by coevonet_amd/lib.py GaAdaptArgs");
__device__ inline void ga_adapt_body(const coevo_ga_adapt_args &a);

__global__ __launch_bounds__(256) void ga_select_kernel(GaSelectArgs a)
{
    // grid (role, slice): every workgroup recomputes the O(pop) parts (score, all fitness values into LDS) and ranks
    // only its own 256 individuals - rank counting is O(pop^2), and weak scaling multiplies pop by the number of GPUs
    // (one workgroup per role took 0.19 ms at pop 1600)
    __shared__ double scratch[4];
    __shared__ float f[4096];
    __shared__ float div_s;
    if ((int)blockIdx.x >= a.n_roles_launched) {   // the extra block: evaluation means + adaptive sigma (ga_adapt_kernel)
        if (a.with_adapt && blockIdx.y == 0 && threadIdx.x == 0) ga_adapt_body(a.adapt);
        return;
    }
    const coevo_ga_select_role R = a.role[blockIdx.x];
    const int n = a.pop;
    const bool first_slice = blockIdx.y == 0;
    // where individual i's record sits in the gathered buffer (its distance was an fp32 value: the cast back is exact)
    auto rec = [&](int i) {
        return a.gathered + 4 * ((size_t)((i / a.n_local) * a.n_roles + (int)blockIdx.x) * a.n_local + i % a.n_local);
    };
    auto dist_of = [&](int i) { return a.gathered ? (float)rec(i)[3] : R.dist[i]; };
    // sharing score (sharing_score_kernel)
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += (double)dist_of(i);
    const float sigma = (float)(block_sum_f64(s, scratch) / (double)n);
    double sc = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float sh = 1.0f - dist_of(i) / sigma;
        if (sh > 0.0f) sc += (double)sh;
    }
    const double tot = block_sum_f64(sc, scratch);
    if (threadIdx.x == 0) {
        div_s = (float)tot;
        if (first_slice) *R.diversity = (float)tot;
    }
    __syncthreads();
    // fitness (ga_fitness_kernel)
    const int gpi = a.games_per_individual;
    for (int i = threadIdx.x; i < n; i += 256) {
        const double last = a.gathered ? rec(i)[R.slot] : R.rewards[3 * (size_t)(R.game_first + i * gpi + gpi - 1) + R.slot];
        const float total = (float)(last / (double)a.hof);
        const float fit = total / (1.0f + div_s);
        f[i] = fit;
        if (first_slice) R.fitness[i] = fit;
    }
    __syncthreads();
    // rank (rank_desc_kernel) of this slice's individuals + the best individual's distance (gather_f32 of order[0])
    const int i = blockIdx.y * 256 + threadIdx.x;
    if (i < n) {
        const float fi = f[i];
        int rank = 0;
        for (int j = 0; j < n; ++j) rank += rank_less(f[j], j, fi, i) ? 1 : 0;
        R.order[n - 1 - rank] = i;
        if (rank == n - 1 && R.best_dist) *R.best_dist = dist_of(i);
    }
}

// np.mean of n <= 10 float64 values exactly as numpy computes it (pairwise_sum: n < 8 sequential from 0; otherwise
// eight running sums combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), remainder added sequentially), then / n
__device__ inline double np_mean_le10(const double *a, int n)
{
    double res;
    if (n < 8) {
        res = 0.0;
        for (int i = 0; i < n; ++i) res += a[i];
    } else {
        res = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
        for (int i = 8; i < n; ++i) res += a[i];
    }
    return res / (double)n;
}

// evaluate_current_weights' means (genetic_algorithm.py:12-29) of the generation that finished one generation ago +
// the adaptive mutation power rule (:323-345, quirk Q5: agent_0's increase starts from agent_1's sigma), on the device
// so that a generation needs no host round trip.  *gen_dev = g: the evaluation games in `rewards` belong to g-1.
__device__ inline void ga_adapt_body(const coevo_ga_adapt_args &a)
{
    const double *rewards = a.rewards;
    double *hist = a.hist, *sig_hist = a.sig_hist, *sigma64 = a.sigma64;
    const int cap = a.cap, eval_first = a.eval_first_game;
    const int g = *a.gen_dev;
    if (a.sigma32_prev)   // what the children evaluated in this generation were bred with (elite rebuild of a sharded run)
        for (int s = 0; s < 3; ++s) a.sigma32_prev[s] = a.sigma32[s];
    if (g > 0 && g - 1 < cap) {
        const int e = g - 1;
        for (int s = 0; s < 3; ++s) {
            double tot = 0.0;
            for (int i = 0; i < 10; ++i) tot += rewards[3 * (size_t)(eval_first + i) + s];
            hist[(size_t)s * cap + e] = tot / 10;
        }
        if (a.adaptive) {
            bool worse[3];
            for (int s = 0; s < 3; ++s) {
                const double *h = hist + (size_t)s * cap;
                const int len = e + 1;
                if (e > 10) {
                    const int n_old = (len >= 20) ? 10 : len - 10;   // h[-20:-10]
                    const double m_new = np_mean_le10(h + len - 10, 10);
                    const double m_old = np_mean_le10(h + (len >= 20 ? len - 20 : 0), n_old);
                    worse[s] = m_new < m_old;
                } else {
                    worse[s] = false;
                }
            }
            const double s1_before = sigma64[1];
            sigma64[0] = worse[0] ? fmin(s1_before * 1.2, a.sig_max) : fmax(sigma64[0] * 0.95, a.sig_min);
            sigma64[1] = worse[1] ? fmin(sigma64[1] * 1.2, a.sig_max) : fmax(sigma64[1] * 0.95, a.sig_min);
            sigma64[2] = worse[2] ? fmin(sigma64[2] * 1.2, a.sig_max) : fmax(sigma64[2] * 0.95, a.sig_min);
        }
        for (int s = 0; s < 3; ++s) sig_hist[(size_t)s * cap + e] = sigma64[s];
    }
    for (int s = 0; s < 3; ++s) a.sigma32[s] = (float)sigma64[s];
}

__global__ void ga_adapt_kernel(coevo_ga_adapt_args a)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    ga_adapt_body(a);
}

// cfg 3 extension mode (NOT in the reference, which has its normalisation commented out at
// evolutionary_strategy.py:133-135): centered ranks u_i = rank_i / (n-1) - 0.5, rank_i = number of individuals that sort
// before i in a stable ascending sort (ties: lower index first).  One thread per individual, any n.
__global__ __launch_bounds__(256) void centered_rank_kernel(const float *f, int n, float *out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float fi = f[i];
    int rank = 0;
    for (int j = 0; j < n; ++j) {
        const float fj = f[j];
        rank += (fj < fi || (fj == fi && j < i)) ? 1 : 0;
    }
    out[i] = (n > 1) ? (float)rank / (float)(n - 1) - 0.5f : 0.0f;
}

__global__ void counter_add_kernel(int32_t *p, int v) { if (threadIdx.x == 0 && blockIdx.x == 0) *p += v; }

}  // namespace coevo

using namespace coevo;

extern "C" int coevo_ga_adapt_sigma(const double *rewards, int eval_first_game, const int32_t *gen_dev, double *hist,
                                    double *sig_hist, int cap, double *sigma64, float *sigma32, double sig_min,
                                    double sig_max, int adaptive, void *stream)
{
    if (!rewards || !gen_dev || !hist || !sig_hist || !sigma64 || !sigma32 || cap <= 0 || eval_first_game < 0)
        return COEVO_ERR_ARG;
    const coevo_ga_adapt_args a{rewards, gen_dev, hist, sig_hist, sigma64, sigma32, nullptr, sig_min, sig_max, eval_first_game,
                                cap, adaptive, 0};
    hipLaunchKernelGGL(ga_adapt_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a);
    COEVO_HIP_CHECK(hipGetLastError());
    return COEVO_OK;
}

extern "C" int coevo_centered_ranks(const float *fitness, int n, float *out, void *stream)
{
    if (!fitness || !out || n <= 0 || fitness == out) return COEVO_ERR_ARG;
    hipLaunchKernelGGL(centered_rank_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, fitness, n, out);
    COEVO_HIP_CHECK(hipGetLastError());
    return COEVO_OK;
}

extern "C" int coevo_counter_add(int32_t *counter, int value, void *stream)
{
    if (!counter) return COEVO_ERR_ARG;
    hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, counter, value);
    COEVO_HIP_CHECK(hipGetLastError());
    return COEVO_OK;
}

extern "C" int coevo_fc_diversity(const float *ref_net, const float *pop_slab, int n, int D, float *dist,
                                  float *score, void *stream)
{
    if (!ref_net || !pop_slab || !dist || !score || n <= 0 || (D != 8 && D != 10)) return COEVO_ERR_ARG;
    hipLaunchKernelGGL(fc_distance_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, ref_net, pop_slab, D, dist);
    hipLaunchKernelGGL(sharing_score_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, dist, n, score);
    COEVO_HIP_CHECK(hipGetLastError());
    return COEVO_OK;
}

extern "C" int coevo_fc_distance(const float *ref_net, const float *pop_slab, int n, int D, float *dist, void *stream)
{
    if (!ref_net || !pop_slab || !dist || n <= 0 || (D != 8 && D != 10)) return COEVO_ERR_ARG;
    hipLaunchKernelGGL(fc_distance_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, ref_net, pop_slab, D, dist);
    COEVO_HIP_CHECK(hipGetLastError());
    return COEVO_OK;
}

extern "C" int coevo_fc_distance_finalize(const double *partial, int n_blocks, int n, float *dist, int first,
                                          const float *head, void *stream)
{
    if (!partial || !dist || n_blocks <= 0 || n <= 0 || first < 0 || (head && first < 1)) return COEVO_ERR_ARG;
    hipLaunchKernelGGL(dist_finalize_kernel, dim3(n), dim3(64), 0, (hipStream_t)stream, partial, n_blocks, dist, first,
                       head);
    COEVO_HIP_CHECK(hipGetLastError());
    return COEVO_OK;
}

static int finalize_multi_launch(const coevo_fc_finalize_job *jobs, int n_jobs, int32_t *tick, void *stream);

extern "C" int coevo_fc_distance_finalize_multi(const coevo_fc_finalize_job *jobs, int n_jobs, void *stream)
{
    return finalize_multi_launch(jobs, n_jobs, nullptr, stream);
}

// ... with the generation counter's tick (coevo_counter_add(counter, 1)) in the same launch: the last launch of a generation
extern "C" int coevo_fc_distance_finalize_multi_tick(const coevo_fc_finalize_job *jobs, int n_jobs, int32_t *counter, void *stream)
{
    if (!counter) return COEVO_ERR_ARG;
    return finalize_multi_launch(jobs, n_jobs, counter, stream);
}

static int finalize_multi_launch(const coevo_fc_finalize_job *jobs, int n_jobs, int32_t *tick, void *stream)
{
    if (!jobs || n_jobs < 1 || n_jobs > COEVO_MAX_JOBS) return COEVO_ERR_ARG;
    FinalizeJobs fj{};
    int nmax = 0;
    for (int i = 0; i < n_jobs; ++i) {
        const coevo_fc_finalize_job &j = jobs[i];
        if (!j.dist_partial || !j.dist || j.n_blocks <= 0 || j.n < 0 || j.first < 0 || (j.head && j.first < 1)) return COEVO_ERR_ARG;
        fj.j[i] = j;
        nmax = j.n > nmax ? j.n : nmax;
    }
    if (nmax == 0 && !tick) return COEVO_OK;
    hipLaunchKernelGGL(dist_finalize_multi_kernel, dim3(nmax > 0 ? nmax : 1, n_jobs), dim3(64), 0, (hipStream_t)stream, fj, tick);
    COEVO_HIP_CHECK(hipGetLastError());
    return COEVO_OK;
}

extern "C" int coevo_gather_f32(float *dst, const float *src, const int32_t *idx, int n, void *stream)
{
    if (!dst || !src || !idx || n <= 0) return COEVO_ERR_ARG;
    hipLaunchKernelGGL(gather_f32_kernel, dim3((n + 127) / 128), dim3(128), 0, (hipStream_t)stream, dst, src, idx, n);
    COEVO_HIP_CHECK(hipGetLastError());
    return COEVO_OK;
}

extern "C" int coevo_sharing_score(const float *dist, int n, float *score, void *stream)
{
    if (!dist || !score || n <= 0) return COEVO_ERR_ARG;
    hipLaunchKernelGGL(sharing_score_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, dist, n, score);
    COEVO_HIP_CHECK(hipGetLastError());
    return COEVO_OK;
}

extern "C" int coevo_ga_fitness(const double *rewards, int game_first, int pop, int games_per_individual, int hof,
                                int slot, const float *diversity, float *fitness, void *stream)
{
    if (!rewards || !diversity || !fitness || pop <= 0 || hof <= 0 || games_per_individual <= 0 || slot < 0 ||
        slot > 2 || game_first < 0)
        return COEVO_ERR_ARG;
    hipLaunchKernelGGL(ga_fitness_kernel, dim3((pop + 127) / 128), dim3(128), 0, (hipStream_t)stream, rewards,
                       game_first, pop, games_per_individual, hof, slot, diversity, fitness);
    COEVO_HIP_CHECK(hipGetLastError());
    return COEVO_OK;
}

extern "C" int coevo_rank_desc(const float *fitness, int n, int32_t *order, void *stream)
{
    if (!fitness || !order || n <= 0 || n > 4096) return COEVO_ERR_ARG;
    hipLaunchKernelGGL(rank_desc_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, fitness, n, order);
    COEVO_HIP_CHECK(hipGetLastError());
    return COEVO_OK;
}

static int ga_select_launch(const coevo_ga_select_role *roles, int n_roles, int pop, int games_per_individual, int hof,
                            const double *gathered, int n_local, const coevo_ga_adapt_args *adapt, void *stream);

extern "C" int coevo_ga_select(const coevo_ga_select_role *roles, int n_roles, int pop, int games_per_individual,
                               int hof, void *stream)
{
    return ga_select_launch(roles, n_roles, pop, games_per_individual, hof, nullptr, 0, nullptr, stream);
}

// coevo_ga_select (gathered == NULL) or coevo_ga_select_gathered, with coevo_ga_adapt_sigma's work in the same launch (an extra
// block beside the roles' blocks; the two are independent: the rule reads the evaluation games, the selection the training
// games): one launch less on the chain selection -> promotion -> offspring (genetic_algorithm.py:223-225 + :323-345).
extern "C" int coevo_ga_select_adapt(const coevo_ga_select_role *roles, int n_roles, int pop, int games_per_individual, int hof,
                                     const double *gathered, int n_local, const coevo_ga_adapt_args *adapt, void *stream)
{
    if (!adapt || !adapt->rewards || !adapt->gen_dev || !adapt->hist || !adapt->sig_hist || !adapt->sigma64 || !adapt->sigma32 ||
        adapt->cap <= 0 || adapt->eval_first_game < 0)
        return COEVO_ERR_ARG;
    if (gathered && (n_local <= 0 || pop % n_local)) return COEVO_ERR_ARG;
    return ga_select_launch(roles, n_roles, pop, gathered ? 1 : games_per_individual, hof, gathered, n_local, adapt, stream);
}

// The selection of a population-sharded run straight off the all-gathered buffer: gathered[rank][role][j][0..2] = the
// play_game triple of the last HoF game of rank `rank`'s j-th individual (quirk Q2), [3] = its distance to the stale agent
// (quirk Q3); individual i = rank i / n_local, j = i % n_local (pop = world * n_local).  The roles' `dist` / `rewards`
// pointers are not read.  Same arithmetic as coevo_ga_select (genetic_algorithm.py:140-146, 223-225).
extern "C" int coevo_ga_select_gathered(const coevo_ga_select_role *roles, int n_roles, int pop, int hof,
                                        const double *gathered, int n_local, void *stream)
{
    if (!gathered || n_local <= 0 || pop % n_local) return COEVO_ERR_ARG;
    return ga_select_launch(roles, n_roles, pop, 1, hof, gathered, n_local, nullptr, stream);
}

static int ga_select_launch(const coevo_ga_select_role *roles, int n_roles, int pop, int games_per_individual, int hof,
                            const double *gathered, int n_local, const coevo_ga_adapt_args *adapt, void *stream)
{
    if (!roles || n_roles < 1 || n_roles > 3 || pop <= 0 || pop > 4096 || hof <= 0 || games_per_individual <= 0)
        return COEVO_ERR_ARG;
    coevo::GaSelectArgs a{};
    for (int r = 0; r < n_roles; ++r) {
        const coevo_ga_select_role &R = roles[r];
        if ((!gathered && (!R.dist || !R.rewards)) || !R.diversity || !R.fitness || !R.order || R.slot < 0 || R.slot > 2 ||
            R.game_first < 0)
            return COEVO_ERR_ARG;
        a.role[r] = R;
    }
    a.gathered = gathered; a.n_local = n_local; a.n_roles = n_roles;
    a.pop = pop; a.games_per_individual = games_per_individual; a.hof = hof;
    a.n_roles_launched = n_roles;
    if (adapt) { a.adapt = *adapt; a.with_adapt = 1; }
    hipLaunchKernelGGL(coevo::ga_select_kernel, dim3(n_roles + (adapt ? 1 : 0), (pop + 255) / 256), dim3(256), 0,
                       (hipStream_t)stream, a);
    COEVO_HIP_CHECK(hipGetLastError());
    return COEVO_OK;
}

extern "C" int coevo_version(void) { return COEVO_VERSION; }

COEVO_DEFINE_TU_FLAGS(select)
namespace coevo {
const char *tu_flags_fc_forward(); const char *tu_flags_mpe_env(); const char *tu_flags_offspring();
const char *tu_flags_rollout_api(); const char *tu_flags_deepqn(); const char *tu_flags_dqn_engine();
const char *tu_flags_host_rollout(); const char *tu_flags_host_placement();
}

extern "C" const char *coevo_build_flags(void)
{
    static char buf[1024];
    static bool done = false;
    if (!done) {   // (idempotent: a race between first callers writes the same bytes)
        const char *names[] = {"fc_forward", "mpe_env", "offspring", "select", "rollout_api", "deepqn", "dqn_engine", "host_rollout",
                               "host_placement"};
        const char *flags[] = {coevo::tu_flags_fc_forward(), coevo::tu_flags_mpe_env(), coevo::tu_flags_offspring(),
                               coevo::tu_flags_select(), coevo::tu_flags_rollout_api(), coevo::tu_flags_deepqn(),
                               coevo::tu_flags_dqn_engine(), coevo::tu_flags_host_rollout(), coevo::tu_flags_host_placement()};
        size_t n = 0;
        buf[0] = 0;
        for (int i = 0; i < 9; ++i) {
            if (!flags[i][0]) continue;
            const int w = snprintf(buf + n, sizeof(buf) - n, "%s%s: %s", n ? "; " : "", names[i], flags[i]);
            if (w < 0 || (size_t)w >= sizeof(buf) - n) break;
            n += (size_t)w;
        }
        done = true;
    }
    return buf;
}
