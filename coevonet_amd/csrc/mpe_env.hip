// Device-side MPE simple_adversary for E env copies (fp64, bit-identical with the host env in
// coevonet_amd/mpe/simple_adversary.py and with oracle/coevo_oracle.c).
//
// Replaces, for a whole batch of games at once, what play_MPE drives through the AEC API
// (reference utils/game_logic_functions.py:123-212: env.observe :138, env.step :179, env.last :181,
// rewards[agent] += reward :190, the limit/truncation breaks :197-204) and play_game's env.reset() (:217).
//
// State is struct-of-arrays [COEVO_MPE_STATE_DOUBLES][n_games] so that a wavefront touching 64 consecutive games
// reads 512 contiguous bytes per field:
//   0..5   ppos[agent][xy]   (agents: 0 adversary_0, 1 agent_0, 2 agent_1)
//   6..11  pvel[agent][xy]
//   12..15 landmark[l][xy]
//   16..17 goal landmark position
//   18     rg_prev: good reward of the latest world step (0 before the first)
//   19..21 acc[slot]: rewards credited to the acting agent of that slot (quirk Q1)
//   22     goal index, 23 spare
#include "coevo_common.hip.h"

namespace coevo {

typedef unsigned __int128 u128;

__host__ __device__ inline u128 pcg_mult()
{
    return (((u128)0x2360ED051FC65DA4ULL) << 64) | (u128)0x4385DF649FCCF645ULL;
}

__host__ __device__ inline uint64_t pcg_output(u128 s)
{
    const uint64_t hi = (uint64_t)(s >> 64), lo = (uint64_t)s;
    const uint64_t x = hi ^ lo;
    const unsigned r = (unsigned)(hi >> 58);
    return (x >> r) | (x << ((64 - r) & 63));
}

// state after `delta` further steps of the 128-bit LCG (O(log delta))
__host__ __device__ inline u128 pcg_advance(u128 state, u128 inc, uint64_t delta)
{
    u128 acc_mult = 1, acc_plus = 0, cur_mult = pcg_mult(), cur_plus = inc;
    while (delta > 0) {
        if (delta & 1) {
            acc_mult *= cur_mult;
            acc_plus = acc_plus * cur_mult + cur_plus;
        }
        cur_plus = (cur_mult + 1) * cur_plus;
        cur_mult *= cur_mult;
        delta >>= 1;
    }
    return acc_mult * state + acc_plus;
}

// numpy Generator semantics of one PettingZoo reset: choice(2) = one buffered 32-bit draw (the low half of a
// 64-bit output for even ordinals, the kept high half for odd ones; Lemire range 2 => top bit), then ten
// uniform(-1,1) doubles.  Two resets consume 21 raw 64-bit outputs.
__host__ __device__ __forceinline__ void mpe_reset_game(double *st, int n, int g, coevo_pcg64 rng, uint64_t ordinal);

__global__ void mpe_reset_kernel(double *st, int n, int game_first, int count, coevo_pcg64 rng,
                                 int64_t first_ordinal, const int32_t *gen_dev, int64_t ordinals_per_gen)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    int64_t first = first_ordinal + (gen_dev ? (int64_t)(*gen_dev) * ordinals_per_gen : 0);
    if (first < 0) first = 0;  // a batch that is disabled in this generation (e.g. no previous evaluation yet)
    mpe_reset_game(st, n, game_first + i, rng, (uint64_t)first + (uint64_t)i);
}

static_assert(sizeof(coevo_reset_seg) == 16, "layout mirrored by coevonet_amd/lib.py ResetSeg");
struct ResetSegs { coevo_reset_seg s[COEVO_MAX_JOBS]; };
__global__ void mpe_reset_multi_kernel(double *st, int n, ResetSegs segs, coevo_pcg64 rng, uint64_t *stamps, int n_stamps,
                                       int32_t *zero_words, int n_zero)
{
    const coevo_reset_seg &sg = segs.s[blockIdx.y];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    // ... and the sync words of the persistent rollout that follows (what sync_clear_kernel did in a launch of its own)
    if (zero_words && blockIdx.y == gridDim.y - 1)
        for (int j = i; j < n_zero; j += gridDim.x * blockDim.x) zero_words[j] = 0;
    // the clock-stamp slot pairs of the rollout that follows, re-armed to {UINT64_MAX, 0} (what stamps_init_kernel did in a
    // launch of its own in front of every timed rollout)
    if (stamps && blockIdx.y == 0)
        for (int j = i; j < n_stamps; j += gridDim.x * blockDim.x) {
            stamps[2 * j] = ~0ull;
            stamps[2 * j + 1] = 0ull;
        }
    if (i >= sg.count) return;
    mpe_reset_game(st, n, sg.game_first + i, rng, sg.first_ordinal + (uint64_t)i);
}

__host__ __device__ __forceinline__ void mpe_reset_game(double *st, int n, int g, coevo_pcg64 rng, uint64_t ordinal)
{
    const u128 inc = ((u128)rng.pcg_inc_hi << 64) | rng.pcg_inc_lo;
    u128 s = ((u128)rng.pcg_state_hi << 64) | rng.pcg_state_lo;
    s = pcg_advance(s, inc, (ordinal >> 1) * 21);
    s = s * pcg_mult() + inc;
    const uint64_t c = pcg_output(s);
    const uint32_t half = (ordinal & 1) ? (uint32_t)(c >> 32) : (uint32_t)c;
    const int goal = (int)(((uint64_t)half * 2) >> 32);
    if (ordinal & 1) s = pcg_advance(s, inc, 10);
    double d[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        s = s * pcg_mult() + inc;
        d[i] = -1.0 + 2.0 * ((double)(pcg_output(s) >> 11) * (1.0 / 9007199254740992.0));
    }
    const size_t N = (size_t)n;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        st[i * N + g] = d[i];        // agent positions
        st[(6 + i) * N + g] = 0.0;   // velocities
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) st[(12 + i) * N + g] = d[6 + i];
    st[16 * N + g] = d[6 + 2 * goal];
    st[17 * N + g] = d[7 + 2 * goal];
    st[18 * N + g] = 0.0;
    st[19 * N + g] = 0.0;
    st[20 * N + g] = 0.0;
    st[21 * N + g] = 0.0;
    st[22 * N + g] = (double)goal;
    st[23 * N + g] = 0.0;
}

// element i of the observation buffer [n_rows][COEVO_OBS_STRIDE] (the kernel's thread / the host loop's iteration)
__host__ __device__ inline float mpe_observe_at(const double *st, int n, const int32_t *row_game, const int32_t *row_slot, int i)
{
    const int row = i / COEVO_OBS_STRIDE, k = i % COEVO_OBS_STRIDE;
    const int slot = row_slot[row];
    const int D = (slot == COEVO_SLOT_ADVERSARY) ? 8 : 10;
    return (k < D) ? mpe_obs_element(st, n, row_game[row], slot, k) : 0.0f;
}

__global__ void mpe_observe_kernel(const double *st, int n, const int32_t *row_game, const int32_t *row_slot,
                                   int n_rows, float *obs)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows * COEVO_OBS_STRIDE) return;
    obs[i] = mpe_observe_at(st, n, row_game, row_slot, i);
}

// One world cycle (three agent-steps) of every game.  Agent-step index of slot s in cycle c is t = 3c + s; it
// happens only while t < limit (play_MPE breaks at timesteps >= limit, :197) and c < max_cycles (truncation, :204,
// enforced by the caller never launching more cycles).  Credits (quirk Q1):
//   adversary_0 acting at 3c   receives agent_0's cumulative reward  = good reward of world step c
//   agent_0     acting at 3c+1 receives agent_1's cumulative reward  = good reward of world step c
//   agent_1     acting at 3c+2 triggers world step c+1 and receives the adversary's reward of that step
__host__ __device__ inline void mpe_step_game(double *st, int n, const int32_t *game_rows, const int32_t *actions,
                                              int cycle, const int32_t *game_limit, int pos_first, int g)
{
    const size_t N = (size_t)n;
    const int limit = game_limit ? game_limit[g] : 0x7fffffff;
    const int t0 = 3 * cycle;
    if (t0 >= limit) return;
    const double rg_prev = st[18 * N + g];
    st[19 * N + g] = st[19 * N + g] + rg_prev;                      // adversary_0 acted
    if (t0 + 1 < limit) st[20 * N + g] = st[20 * N + g] + rg_prev;  // agent_0 acted
    if (t0 + 2 >= limit) return;                                    // agent_1 did not act: no world step
    double d[3];
    const double gx = st[16 * N + g], gy = st[17 * N + g];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int act = actions[game_rows[3 * g + i]];
        double u[2] = {0.0, 0.0};
        if (act == 1) u[0] = -1.0;
        if (act == 2) u[0] = +1.0;
        if (act == 3) u[1] = -1.0;
        if (act == 4) u[1] = +1.0;
        double p[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const double f = (((u[c] * 5.0) + 0.0) / 1.0) * 0.1;
            double pos = st[(2 * i + c) * N + g], vel = st[(6 + 2 * i + c) * N + g];
            if (pos_first) pos = pos + vel * 0.1;
            vel = vel * 0.75;
            vel = vel + f;
            if (!pos_first) pos = pos + vel * 0.1;
            st[(2 * i + c) * N + g] = pos;
            st[(6 + 2 * i + c) * N + g] = vel;
            p[c] = pos;
        }
        const double dx = p[0] - gx, dy = p[1] - gy;
        d[i] = sqrt(dx * dx + dy * dy);
    }
    const double r_adv = -d[0];
    const double m = (d[2] < d[1]) ? d[2] : d[1];
    const double r_good = -m + d[0];
    st[21 * N + g] = st[21 * N + g] + r_adv;  // agent_1 acted
    st[18 * N + g] = r_good;
}

__global__ void mpe_step_kernel(double *st, int n, const int32_t *game_rows, const int32_t *actions, int cycle,
                                const int32_t *game_limit, int pos_first)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < n) mpe_step_game(st, n, game_rows, actions, cycle, game_limit, pos_first, g);
}

// The last cycle's step in the fused scheme: its actions are applied only to close the books (credits of cycle
// `cycle`), the play_game triples go straight to rewards[n][3]; no state is written.
// pack (a population-sharded run, coevo_mpe_final_step_pack): the games are laid out [role][local individual][hof game]; the
// thread that closes the LAST HoF game of an individual (the one that counts, quirk Q2) also writes that individual's record
// of the fitness all-gather: pack[role][j] = {its play_game triple, the individual's distance to the stale agent}
typedef coevo_final_pack FinalPack;   // {out [n_roles][n_local][4] or NULL, dist [n_roles][dist_pitch], n_roles, n_local, hof, dist_pitch, dist_first}
static_assert(sizeof(coevo_final_pack) == 40, "layout mirrored by coevonet_amd/lib.py FinalPack");

__global__ void mpe_final_step_kernel(const double *st, int n, const int32_t *act, int cycle,
                                      const int32_t *game_limit, int pos_first, double *rewards, FinalPack pk)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    const size_t N = (size_t)n;
    double a_adv = st[19 * N + g], a_a0 = st[20 * N + g], a_a1 = st[21 * N + g];
    if (cycle >= 0) {
        const int limit = game_limit ? game_limit[g] : 0x7fffffff;
        const int t0 = 3 * cycle;
        const double rg_prev = st[18 * N + g];
        if (t0 < limit) a_adv = a_adv + rg_prev;
        if (t0 + 1 < limit) a_a0 = a_a0 + rg_prev;
        if (t0 + 2 < limit) {
            MpeGame s;
            mpe_load_game(st, n, g, s);
            double r_good, r_adv;
            mpe_world_step(s, act[3 * g], act[3 * g + 1], act[3 * g + 2], pos_first, r_good, r_adv);
            a_a1 = a_a1 + r_adv;
        }
    }
    rewards[3 * (size_t)g + 0] = a_a0;
    rewards[3 * (size_t)g + 1] = a_a1;
    rewards[3 * (size_t)g + 2] = a_adv;
    if (pk.out && g < pk.n_roles * pk.n_local * pk.hof && g % pk.hof == pk.hof - 1) {
        const int ij = g / pk.hof, role = ij / pk.n_local, j = ij % pk.n_local;
        double *o = pk.out + 4 * (size_t)ij;
        o[0] = a_a0;
        o[1] = a_a1;
        o[2] = a_adv;
        o[3] = (double)pk.dist[(size_t)role * pk.dist_pitch + pk.dist_first + j];
    }
}

__global__ void mpe_rewards_kernel(const double *st, int n, double *rewards)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    const size_t N = (size_t)n;
    rewards[3 * (size_t)g + 0] = st[20 * N + g];  // agent_0
    rewards[3 * (size_t)g + 1] = st[21 * N + g];  // agent_1
    rewards[3 * (size_t)g + 2] = st[19 * N + g];  // adversary_0
}

}  // namespace coevo

extern "C" int coevo_mpe_reset_gen(double *state, int n_games, int game_first, int count, coevo_pcg64 rng,
                                   int64_t first_ordinal, const int32_t *gen_dev, int64_t ordinals_per_gen,
                                   void *stream)
{
    if (!state || n_games <= 0 || game_first < 0 || count < 0 || game_first + count > n_games) return COEVO_ERR_ARG;
    if (count == 0) return COEVO_OK;
    hipLaunchKernelGGL(coevo::mpe_reset_kernel, dim3((count + 127) / 128), dim3(128), 0, (hipStream_t)stream,
                       state, n_games, game_first, count, rng, first_ordinal, gen_dev, ordinals_per_gen);
    COEVO_HIP_CHECK(hipGetLastError());
    return COEVO_OK;
}

static int reset_multi_launch(double *state, int n_games, const coevo_reset_seg *segs, int n_segs, coevo_pcg64 rng,
                              uint64_t *stamps, int n_stamps, int32_t *zero_words, int n_zero, void *stream);

extern "C" int coevo_mpe_reset_multi(double *state, int n_games, const coevo_reset_seg *segs, int n_segs, coevo_pcg64 rng,
                                     void *stream)
{
    return reset_multi_launch(state, n_games, segs, n_segs, rng, nullptr, 0, nullptr, 0, stream);
}

// ... that also re-arms n_stamps {start, end} clock-stamp slot pairs for the timed rollout that follows (coevo_rollout_desc.
// stamps_armed): a timed rollout then needs no launch of its own for it
extern "C" int coevo_mpe_reset_multi_arm(double *state, int n_games, const coevo_reset_seg *segs, int n_segs, coevo_pcg64 rng,
                                         uint64_t *stamps, int n_stamps, void *stream)
{
    if (!stamps || n_stamps <= 0) return COEVO_ERR_ARG;
    return reset_multi_launch(state, n_games, segs, n_segs, rng, stamps, n_stamps, nullptr, 0, stream);
}

extern "C" int coevo_mpe_reset_multi_prep(double *state, int n_games, const coevo_reset_seg *segs, int n_segs, coevo_pcg64 rng,
                                          uint64_t *stamps, int n_stamps, int32_t *zero_words, int n_zero, void *stream)
{
    if ((stamps && n_stamps <= 0) || (zero_words && n_zero <= 0)) return COEVO_ERR_ARG;
    return reset_multi_launch(state, n_games, segs, n_segs, rng, stamps, stamps ? n_stamps : 0, zero_words, zero_words ? n_zero : 0,
                              stream);
}

static int reset_multi_launch(double *state, int n_games, const coevo_reset_seg *segs, int n_segs, coevo_pcg64 rng,
                              uint64_t *stamps, int n_stamps, int32_t *zero_words, int n_zero, void *stream)
{
    if (!state || n_games <= 0 || !segs || n_segs < 1 || n_segs > COEVO_MAX_JOBS) return COEVO_ERR_ARG;
    coevo::ResetSegs rs{};
    int cmax = 0;
    for (int i = 0; i < n_segs; ++i) {
        if (segs[i].game_first < 0 || segs[i].count < 0 || segs[i].game_first + segs[i].count > n_games) return COEVO_ERR_ARG;
        rs.s[i] = segs[i];
        cmax = segs[i].count > cmax ? segs[i].count : cmax;
    }
    if (cmax == 0 && !stamps && !zero_words) return COEVO_OK;
    hipLaunchKernelGGL(coevo::mpe_reset_multi_kernel, dim3((cmax > 0 ? cmax + 127 : 128) / 128, n_segs), dim3(128), 0,
                       (hipStream_t)stream, state, n_games, rs, rng, stamps, n_stamps, zero_words, n_zero);
    COEVO_HIP_CHECK(hipGetLastError());
    return COEVO_OK;
}

extern "C" int coevo_mpe_reset(double *state, int n_games, int game_first, int count, coevo_pcg64 rng,
                               uint64_t first_ordinal, void *stream)
{
    if (!state || n_games <= 0 || game_first < 0 || count < 0 || game_first + count > n_games) return COEVO_ERR_ARG;
    if (count == 0) return COEVO_OK;
    hipLaunchKernelGGL(coevo::mpe_reset_kernel, dim3((count + 127) / 128), dim3(128), 0, (hipStream_t)stream,
                       state, n_games, game_first, count, rng, (int64_t)first_ordinal, (const int32_t *)nullptr,
                       (int64_t)0);
    COEVO_HIP_CHECK(hipGetLastError());
    return COEVO_OK;
}

extern "C" int coevo_mpe_observe(const double *state, int n_games, const int32_t *row_game,
                                 const int32_t *row_slot, int n_rows, float *obs, void *stream)
{
    if (!state || !row_game || !row_slot || !obs || n_games <= 0 || n_rows <= 0) return COEVO_ERR_ARG;
    const int total = n_rows * COEVO_OBS_STRIDE;
    hipLaunchKernelGGL(coevo::mpe_observe_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       state, n_games, row_game, row_slot, n_rows, obs);
    COEVO_HIP_CHECK(hipGetLastError());
    return COEVO_OK;
}

extern "C" int coevo_mpe_step(double *state, int n_games, const int32_t *game_rows, const int32_t *actions,
                              int cycle, const int32_t *game_limit, int pos_first, void *stream)
{
    if (!state || !game_rows || !actions || n_games <= 0 || cycle < 0) return COEVO_ERR_ARG;
    hipLaunchKernelGGL(coevo::mpe_step_kernel, dim3((n_games + 127) / 128), dim3(128), 0, (hipStream_t)stream,
                       state, n_games, game_rows, actions, cycle, game_limit, pos_first);
    COEVO_HIP_CHECK(hipGetLastError());
    return COEVO_OK;
}

// The same two functions on the host cores (env_mode "host": the env is stepped by the rank's own host process, the
// observations go up and the actions come back over PCIe every cycle).  Plain host pointers; the bodies are the kernels'.
// Host side: the games of a list take their resets; a game whose ordinal follows its predecessor's continues the stream
// where that one stopped (two resets consume 21 raw outputs: the shared choice word, then ten doubles each) instead of
// jumping ahead from the seed again - the same stream positions, so the same bits as mpe_reset_game.
namespace coevo {
static void mpe_reset_list_host(double *st, int n, coevo_pcg64 rng, const int64_t *ordinals, const int32_t *games, int lo, int hi)
{
    const u128 inc = ((u128)rng.pcg_inc_hi << 64) | rng.pcg_inc_lo;
    const u128 s0 = ((u128)rng.pcg_state_hi << 64) | rng.pcg_state_lo;
    const size_t N = (size_t)n;
    u128 s = 0;
    uint64_t c = 0, prev = 0;
    bool have = false;
    for (int i = lo; i < hi; ++i) {
        const int g = games ? games[i] : i;
        const uint64_t ordinal = (uint64_t)ordinals[g];
        if (have && ordinal == prev + 1) {
            if (!(ordinal & 1)) {   // a new pair: its choice word is the next output
                s = s * pcg_mult() + inc;
                c = pcg_output(s);
            }                       // (odd: the pair's word is `c`, the stream stands behind the even game's ten doubles)
        } else {
            s = pcg_advance(s0, inc, (ordinal >> 1) * 21);
            s = s * pcg_mult() + inc;
            c = pcg_output(s);
            if (ordinal & 1) s = pcg_advance(s, inc, 10);
        }
        const uint32_t half = (ordinal & 1) ? (uint32_t)(c >> 32) : (uint32_t)c;
        const int goal = (int)(((uint64_t)half * 2) >> 32);
        double d[10];
        for (int k = 0; k < 10; ++k) {
            s = s * pcg_mult() + inc;
            d[k] = -1.0 + 2.0 * ((double)(pcg_output(s) >> 11) * (1.0 / 9007199254740992.0));
        }
        for (int k = 0; k < 6; ++k) {
            st[k * N + g] = d[k];
            st[(6 + k) * N + g] = 0.0;
        }
        for (int k = 0; k < 4; ++k) st[(12 + k) * N + g] = d[6 + k];
        st[16 * N + g] = d[6 + 2 * goal];
        st[17 * N + g] = d[7 + 2 * goal];
        st[18 * N + g] = 0.0;
        st[19 * N + g] = 0.0;
        st[20 * N + g] = 0.0;
        st[21 * N + g] = 0.0;
        st[22 * N + g] = (double)goal;
        st[23 * N + g] = 0.0;
        prev = ordinal;
        have = true;
    }
}
}  // namespace coevo

extern "C" int coevo_mpe_host_reset(double *state, int n_games, coevo_pcg64 rng, const int64_t *ordinals)
{
    if (!state || !ordinals || n_games <= 0) return COEVO_ERR_ARG;
    for (int g = 0; g < n_games; ++g)
        if (ordinals[g] < 0) return COEVO_ERR_ARG;
    coevo::mpe_reset_list_host(state, n_games, rng, ordinals, nullptr, 0, n_games);
    return COEVO_OK;
}

// ... of games[lo..hi) only (what a host core does for one of its cohorts at the start of coevo_mpe_host_rollout)
extern "C" int coevo_mpe_host_reset_games(double *state, int n_games, coevo_pcg64 rng, const int64_t *ordinals,
                                          const int32_t *games, int lo, int hi)
{
    if (!state || !ordinals || !games || n_games <= 0 || lo < 0 || hi < lo) return COEVO_ERR_ARG;
    for (int i = lo; i < hi; ++i)
        if (games[i] < 0 || games[i] >= n_games || ordinals[games[i]] < 0) return COEVO_ERR_ARG;
    coevo::mpe_reset_list_host(state, n_games, rng, ordinals, games, lo, hi);
    return COEVO_OK;
}

extern "C" int coevo_mpe_host_observe(const double *state, int n_games, const int32_t *row_game, const int32_t *row_slot,
                                      int n_rows, float *obs)
{
    if (!state || !row_game || !row_slot || !obs || n_games <= 0 || n_rows <= 0) return COEVO_ERR_ARG;
    for (int r = 0; r < n_rows; ++r)
        if (row_game[r] < 0 || row_game[r] >= n_games || row_slot[r] < 0 || row_slot[r] > 2) return COEVO_ERR_ARG;
    for (int r = 0; r < n_rows; ++r) {   // a row at a time: the game's state once (the same casts of the same differences)
        coevo::MpeGame s;
        coevo::mpe_load_game(state, n_games, row_game[r], s);
        float *o = obs + (size_t)r * COEVO_OBS_STRIDE;
        coevo::mpe_obs_from_game(s, row_slot[r], o);
        o[10] = 0.0f;
        o[11] = 0.0f;
    }
    return COEVO_OK;
}

extern "C" int coevo_mpe_host_step(double *state, int n_games, const int32_t *game_rows, const int32_t *actions,
                                   int n_rows, int cycle, const int32_t *game_limit, int pos_first)
{
    if (!state || !game_rows || !actions || n_games <= 0 || n_rows <= 0 || cycle < 0) return COEVO_ERR_ARG;
    for (int i = 0; i < 3 * n_games; ++i)
        if (game_rows[i] < 0 || game_rows[i] >= n_rows) return COEVO_ERR_ARG;
    for (int g = 0; g < n_games; ++g) coevo::mpe_step_game(state, n_games, game_rows, actions, cycle, game_limit, pos_first, g);
    return COEVO_OK;
}

// One slice of a game list on the calling thread: world step `cycle` (cycle < 0: none) of games[lo..hi), then (observe != 0)
// the observations of their three rows.  The unit of work a host core gets from coevo_mpe_host_rollout; no argument scan
// (the caller validated the tables once), so that T threads on T slices cost what one thread costs on the whole list / T.
extern "C" int coevo_mpe_host_step_games(double *state, int n_games, const int32_t *game_rows, const int32_t *actions,
                                         int cycle, const int32_t *game_limit, int pos_first, const int32_t *games, int lo,
                                         int hi, int observe, float *obs)
{
    if (!state || !game_rows || !games || n_games <= 0 || lo < 0 || hi < lo || (cycle >= 0 && !actions) || (observe && !obs))
        return COEVO_ERR_ARG;
    for (int i = lo; i < hi; ++i) {
        const int g = games[i];
        if (cycle >= 0) coevo::mpe_step_game(state, n_games, game_rows, actions, cycle, game_limit, pos_first, g);
        if (observe) {
            coevo::MpeGame s;
            coevo::mpe_load_game(state, n_games, g, s);
            for (int slot = 0; slot < 3; ++slot) {
                float *o = obs + (size_t)game_rows[3 * g + slot] * COEVO_OBS_STRIDE;
                coevo::mpe_obs_from_game(s, slot, o);
                o[10] = 0.0f;
                o[11] = 0.0f;
            }
        }
    }
    return COEVO_OK;
}

extern "C" int coevo_mpe_final_step(const double *state, int n_games, const int32_t *actions_by_game, int cycle,
                                    const int32_t *game_limit, int pos_first, double *rewards, void *stream)
{
    if (!state || !rewards || n_games <= 0 || (cycle >= 0 && !actions_by_game)) return COEVO_ERR_ARG;
    hipLaunchKernelGGL(coevo::mpe_final_step_kernel, dim3((n_games + 127) / 128), dim3(128), 0, (hipStream_t)stream,
                       state, n_games, actions_by_game, cycle, game_limit, pos_first, rewards, coevo::FinalPack{});
    COEVO_HIP_CHECK(hipGetLastError());
    return COEVO_OK;
}

// coevo_mpe_final_step + this rank's record of the fitness all-gather in the same launch (genetic_algorithm.py:140-146: only the
// last HoF game's reward survives, quirk Q2): games [0, n_roles * n_local * hof) are laid out [role][local individual][hof
// game]; pack[role][j] = {triple of game (role, j, hof - 1), (double)dist[role * dist_pitch + dist_first + j]}.
extern "C" int coevo_mpe_final_step_pack(const double *state, int n_games, const int32_t *actions_by_game, int cycle,
                                         const int32_t *game_limit, int pos_first, double *rewards, double *pack,
                                         const float *dist, int n_roles, int n_local, int hof, int dist_pitch, int dist_first,
                                         void *stream)
{
    if (!state || !rewards || n_games <= 0 || (cycle >= 0 && !actions_by_game)) return COEVO_ERR_ARG;
    if (!pack || !dist || n_roles < 1 || n_local < 1 || hof < 1 || (int64_t)n_roles * n_local * hof > n_games ||
        dist_first < 0 || dist_first + n_local > dist_pitch)
        return COEVO_ERR_ARG;
    const coevo::FinalPack pk{pack, dist, n_roles, n_local, hof, dist_pitch, dist_first, 0};
    hipLaunchKernelGGL(coevo::mpe_final_step_kernel, dim3((n_games + 127) / 128), dim3(128), 0, (hipStream_t)stream,
                       state, n_games, actions_by_game, cycle, game_limit, pos_first, rewards, pk);
    COEVO_HIP_CHECK(hipGetLastError());
    return COEVO_OK;
}

extern "C" int coevo_mpe_rewards(const double *state, int n_games, double *rewards, void *stream)
{
    if (!state || !rewards || n_games <= 0) return COEVO_ERR_ARG;
    hipLaunchKernelGGL(coevo::mpe_rewards_kernel, dim3((n_games + 127) / 128), dim3(128), 0, (hipStream_t)stream,
                       state, n_games, rewards);
    COEVO_HIP_CHECK(hipGetLastError());
    return COEVO_OK;
}

COEVO_DEFINE_TU_FLAGS(mpe_env)
