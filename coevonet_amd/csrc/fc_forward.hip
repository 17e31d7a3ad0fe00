// K1 - FCNetwork policy step for many (weight set x observation rows) tasks in one launch.
//
// Replaces FCNetwork.forward + determine_action (reference MPE/fcnetwork.py:37-90), which the reference calls
// once per agent-step at batch 1 (utils/game_logic_functions.py:152-163).
//
// One workgroup (4 wavefronts of 64) = one task = one weight set applied to up to R observation rows.
// HBM traffic per task is the weight set itself, read exactly once, fully coalesced:
//   fc1  W1t[k][512]           each lane owns outputs t and t+256, wave reads 256 B rows (20 KB total)
//   fc2  W2q[4][128][64][4]    wave w streams its 128 KiB block as 1 KiB wave-instructions (global_load_dwordx4),
//                              8-16 in flight per lane; each 16-byte piece is 4 k of the lane's own output column and
//                              feeds v_mfma_f32_4x4x1_16B_f32 directly (16 blocks x 4 columns = the wave's 64 columns,
//                              4 rows, one k per instruction); the A operands of a row group are one ds_read_b128
//   out  W3[5][256]            staged through LDS once
// Arithmetic per weight byte is tiny (R FMAs per 4 bytes), but as VALU FMAs fed by LDS broadcasts it still cost a lone
// workgroup 27 us per net against ~5 us of HBM time (round 1 probe; the stream rate of one CU: tools/stream_waves_probe.hip): the matrix pipe, otherwise idle here,
// does the same sequential-k fma chains (tools/mfma4_chain_probe.hip: bit-identical) in far fewer issue slots and LDS
// reads.  (v_mfma_f32_16x16x4_f32, tools/mfma16_chain_probe.hip, is exact as well but spends 16 tile rows on <= 8.)
// All fp32 math follows the canonical order in coevo_common.hip.h, so logits equal the oracle's bit for bit.
#include <cstdlib>

#include "coevo_common.hip.h"

namespace coevo {

#ifndef COEVO_HEAVY16_U
#define COEVO_HEAVY16_U 4  // 16-byte pieces per lane and buffer in the lean shared-opponent body (two buffers)
#endif
#ifndef COEVO_LIGHT_U
#define COEVO_LIGHT_U 8   // 16-byte pieces per lane and buffer (two buffers in ping-pong)
#endif

// A per-individual weight set is read exactly once per launch by exactly one CU: non-temporal loads keep that stream
// (335 MB per env-cycle through 32 MB of L2) from evicting what IS reused - the shared-opponent nets and the kernel's own
// code (MI355X_MICROARCH.md row nt-weights).  With the first, VALU-bound version of this kernel they changed nothing
// (350 vs 349 generations/s); with the current one: 539.5 vs 517 (three alternating runs each).  -DCOEVO_NO_NT builds
// the plain-load variant for A/B runs.
typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ inline float4 load_stream16(const float4 *p)
{
#ifdef COEVO_NO_NT
    return *p;
#else
    const f32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t *>(p));
    return make_float4(v[0], v[1], v[2], v[3]);
#endif
}

template <int R, int P>
struct FcSmem {
    static constexpr int NG = (R + 3) / 4;   // row groups of four = v_mfma_f32_4x4x1 issues per k
    static constexpr int RP = 4 * NG + 1;    // odd row pitch: conflict-free scatter of h1 into the k-quad image
    union {
        // fc1 activations of ONE net at a time, [k/4][row][k%4]: one ds_read_b128 hands lane l the A operands
        // x[4g + l%4][4q .. 4q+3] of row group g.  Rows >= R of the last group are never written: a tile row only
        // feeds its own output row, and nothing reads those.
        float h1q[128][RP][4];
        float h2[P][R][260];  // fc2 activations, row pitch 260 keeps 16-byte alignment, shifts banks by 4
    };
    float xs0[P][R][COEVO_OBS_STRIDE];
    float w3s[P][NACT][260];
    float red[P][R][8];       // LayerNorm partials per (row, 64-feature block)
    float logit[P][R][COEVO_LOGIT_STRIDE];
};

struct FcArgs {
    const float *slab;
    const coevo_fc_task *tasks;
    const float *obs;          // MODE 0: observations given
    const double *state;       // MODE 1: read from the game state [COEVO_MPE_STATE_DOUBLES][n_games];
                               // MODE 2 (fused env step): the PREVIOUS cycle's state buffer
    const int32_t *row_game;
    const int32_t *row_slot;
    int n_games;
    int32_t *actions;
    float *logits;  // may be null
    int32_t *status;
    unsigned long long *stamps;  // may be null: [COEVO_STAMP_SLOTS][2] = {min workgroup start, max workgroup end}
    // MODE 2 only
    double *state_next;          // the buffer this cycle's state is written to (by each game's owner row)
    const int32_t *act_prev;     // [n_games][3] actions of the previous cycle, by env slot
    int32_t *act_cur;            // [n_games][3] this cycle's actions
    const int32_t *game_limit;
    int cycle, pos_first;
    // merged launch only (fc_cycle_kernel): workgroups [0, n_heavy) run `tasks` on the matrix cores, the rest run
    // `light_tasks[blockIdx.x - n_heavy]` through the streaming path
    const coevo_fc_task *light_tasks;
    int n_heavy, n_light;
};
constexpr int MODE_OBS = 0, MODE_STATE = 1, MODE_FUSED = 2;

// 100 MHz wall clock; the first/last workgroup of a launch bracket its duration (used where HIP events cannot be:
// inside hipGraph replays)
// (workgroups spread over COEVO_STAMP_SLOTS slot pairs so that hundreds of simultaneous atomics do not serialise on
// one address; the host takes the min / max over the slots)
__device__ inline void stamp_begin(unsigned long long *stamps)
{
    if (stamps && threadIdx.x == 0)
        atomicMin(&stamps[2 * (blockIdx.x % COEVO_STAMP_SLOTS)], (unsigned long long)__builtin_amdgcn_s_memrealtime());
}
__device__ inline void stamp_end(unsigned long long *stamps)
{
    if (stamps && threadIdx.x == 0)
        atomicMax(&stamps[2 * (blockIdx.x % COEVO_STAMP_SLOTS) + 1],
                  (unsigned long long)__builtin_amdgcn_s_memrealtime());
}

#ifdef COEVO_PHASE_STAMPS
// diagnostic build only (tools/merged_wg_times.py): per-workgroup shader-clock stamps at the phase boundaries of the
// streaming kernel, written to a buffer nothing else reads
__device__ unsigned long long g_phase_stamps[4096 * 16];
#define COEVO_STAMP(i)                                                                          \
    do {                                                                                        \
        if (threadIdx.x == 0 && blockIdx.x < 4096) {                                            \
            /* 100 MHz constant clock: comparable across XCDs (s_memtime is per XCD) */          \
            g_phase_stamps[blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memrealtime();            \
        }                                                                                       \
    } while (0)
// per-wave stamps (lane 0 of every wave): where do the waves of a workgroup wait for each other?
__device__ unsigned long long g_wave_stamps[4096 * 4 * 16];
#define COEVO_WSTAMP(i)                                                                                         \
    do {                                                                                                        \
        if ((threadIdx.x & 63) == 0 && blockIdx.x < 4096)                                                       \
            g_wave_stamps[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
// shader-clock counter (s_memtime) next to the 100 MHz one: effective clock of a section = d(memtime) / d(memrealtime)
#define COEVO_CSTAMP(i)                                                                                         \
    do {                                                                                                        \
        if ((threadIdx.x & 63) == 0 && blockIdx.x < 4096)                                                       \
            g_wave_stamps[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 16 + (i)] = __builtin_amdgcn_s_memtime();     \
    } while (0)
#else
#define COEVO_STAMP(i) do { } while (0)
#define COEVO_WSTAMP(i) do { } while (0)
#define COEVO_CSTAMP(i) do { } while (0)
#endif

// One workgroup carries P nets (tasks first .. first+P-1).  P = 1 is the plain kernel.  P = 2 is used by the merged
// cycle launch when one net per workgroup would not fit the CUs in a single round: the entry chains (env step,
// observation) and both LayerNorm(512) of the two nets share their latency, the two 512 KiB weight streams run back to
// back, then both LayerNorm(256) / output layers share theirs.  Results do not depend on P.
template <int R, int MODE, int P>
__device__ __forceinline__ void fc_policy_body(const FcArgs &a, FcSmem<R, P> &sm, const coevo_fc_task *tasks, int first,
                                               int n_tasks)
{
#ifdef COEVO_LIGHT_PRIO
    __builtin_amdgcn_s_setprio(COEVO_LIGHT_PRIO);
#endif
    COEVO_STAMP(0);
    static_assert(P >= 1 && P <= 4 && R * NACT <= 64, "one wave per net in the per-row phases");
    const int t = threadIdx.x, w = t >> 6, l = t & 63;
    int D[P], nrows[P], row0[P];
    const float *net[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const bool active = first + p < n_tasks;  // a missing net repeats the last one with no rows: nothing is written
        const coevo_fc_task task = tasks[active ? first + p : n_tasks - 1];
        D[p] = task.D; nrows[p] = active ? task.n_rows : 0; row0[p] = task.row_begin;
        net[p] = a.slab + task.net_off;
    }
    int st = 0;

    // ---- every small parameter this thread will need, requested up front: one HBM round trip instead of one
    //      per layer (they would otherwise be loaded at first use, behind a barrier, with nothing else in flight).
    //      (Measured: requesting the row lanes' game state even before these - vmcnt completes in order - and marking
    //      consecutive-game tasks to skip the row table costs registers, spills in the paired kernel: 438 vs 451.)
    constexpr int DMAX = 10;
    float w1a[P][DMAX], w1b[P][DMAX];
    float p_b1a[P], p_b1b[P], p_g1a[P], p_g1b[P], p_be1a[P], p_be1b[P];
    float p_b2[P], p_g2[P], p_be2[P];  // fc2 column of this lane: t
#pragma unroll
    for (int p = 0; p < P; ++p) {
#pragma unroll
        for (int k = 0; k < DMAX; ++k) {
            w1a[p][k] = (k < D[p]) ? net[p][(size_t)k * H1 + t] : 0.0f;
            w1b[p][k] = (k < D[p]) ? net[p][(size_t)k * H1 + 256 + t] : 0.0f;
        }
        const float *b1p = net[p] + fc_off_b1(D[p]), *b2p = net[p] + fc_off_b2(D[p]);
        p_b1a[p] = b1p[t]; p_b1b[p] = b1p[t + 256];
        p_g1a[p] = b1p[H1 + t]; p_g1b[p] = b1p[H1 + t + 256];
        p_be1a[p] = b1p[2 * H1 + t]; p_be1b[p] = b1p[2 * H1 + t + 256];
        p_b2[p] = b2p[t]; p_g2[p] = b2p[H2 + t]; p_be2[p] = b2p[2 * H2 + t];
    }
    // per-row phases (env step + observation, output layer, argmax): wave p serves net p, lane = row or (row, action)
    const int pw = (w < P) ? w : 0;  // wave-uniform
    const bool row_wave = w < P;
    const float p_b3 = (row_wave && l < R * NACT) ? net[pw][fc_off_b3(D[pw]) + l % NACT] : 0.0f;

    // ---- stage observations (zero padded) and the output layer -------------------------------------------
    if constexpr (MODE >= MODE_FUSED) {
        if (row_wave && l < R) {  // one lane per row: advance its game in registers, observe, (owner row) publish
            float o[COEVO_OBS_STRIDE];
#pragma unroll
            for (int k = 0; k < COEVO_OBS_STRIDE; ++k) o[k] = 0.0f;
            if (l < nrows[pw]) {
                const int row = row0[pw] + l;
                mpe_fused_observe(a.state, a.state_next, a.act_prev, a.game_limit, a.n_games, a.row_game[row],
                                  a.row_slot[row], a.cycle, a.pos_first, o);
#pragma unroll
                for (int k = 0; k < 10; ++k)
                    if (!__builtin_isfinite(o[k])) st |= COEVO_ST_BAD_INPUT;
            }
#pragma unroll
            for (int k = 0; k < COEVO_OBS_STRIDE; ++k) sm.xs0[pw][l][k] = o[k];
        }
    } else {
#pragma unroll
        for (int p = 0; p < P; ++p) {
            for (int i = t; i < R * COEVO_OBS_STRIDE; i += 256) {
                const int r = i / COEVO_OBS_STRIDE, k = i % COEVO_OBS_STRIDE;
                float v = 0.0f;
                if (r < nrows[p] && k < D[p]) {
                    if constexpr (MODE == MODE_STATE) {
                        const int row = row0[p] + r;
                        v = mpe_obs_element(a.state, a.n_games, a.row_game[row], a.row_slot[row], k);
                    } else {
                        v = a.obs[(size_t)(row0[p] + r) * COEVO_OBS_STRIDE + k];
                    }
                    if (!__builtin_isfinite(v)) st |= COEVO_ST_BAD_INPUT;
                }
                sm.xs0[p][r][k] = v;
            }
        }
    }
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const float *W3 = net[p] + fc_off_w3(D[p]);
        for (int i = t; i < NACT * H2; i += 256) sm.w3s[p][i >> 8][i & 255] = W3[i];
    }
    __syncthreads();
    COEVO_STAMP(1);

    // ---- fc1: outputs j0 = t, j1 = t + 256; sequential-k chains from the bias -----------------------------
    float a0[P][R], a1[P][R];
#pragma unroll
    for (int p = 0; p < P; ++p) {
#pragma unroll
        for (int r = 0; r < R; ++r) { a0[p][r] = p_b1a[p]; a1[p][r] = p_b1b[p]; }
#pragma unroll
        for (int k = 0; k < DMAX; ++k) {
            if (k < D[p]) {  // wave-uniform
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const float x = sm.xs0[p][r][k];
                    a0[p][r] = __builtin_fmaf(w1a[p][k], x, a0[p][r]);
                    a1[p][r] = __builtin_fmaf(w1b[p][k], x, a1[p][r]);
                }
            }
        }
    }
    COEVO_STAMP(8);
    // ---- LayerNorm(512) + ReLU: block b of the canonical reduce is wave (b & 3), half (b >> 2) -------------
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float s0 = wave_tree_sum(a0[p][r]), s1 = wave_tree_sum(a1[p][r]);
            if (l == 0) { sm.red[p][r][w] = s0; sm.red[p][r][4 + w] = s1; }
        }
    COEVO_STAMP(9);
    __syncthreads();
    COEVO_STAMP(10);
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float tot = sm.red[p][r][0];
#pragma unroll
            for (int b = 1; b < 8; ++b) tot = tot + sm.red[p][r][b];
            const float mean = tot * (1.0f / H1);
            a0[p][r] = a0[p][r] - mean;
            a1[p][r] = a1[p][r] - mean;
        }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float s0 = wave_tree_sum(a0[p][r] * a0[p][r]), s1 = wave_tree_sum(a1[p][r] * a1[p][r]);
            if (l == 0) { sm.red[p][r][w] = s0; sm.red[p][r][4 + w] = s1; }
        }
    __syncthreads();
    // the normalised activations stay in registers (a0 / a1) until their net's turn at the single A-operand image
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float tot = sm.red[p][r][0];
#pragma unroll
            for (int b = 1; b < 8; ++b) tot = tot + sm.red[p][r][b];
            const float var = tot * (1.0f / H1);
            const float rstd = 1.0f / __builtin_sqrtf(var + LN_EPS);
            const float y0 = __builtin_fmaf(a0[p][r] * rstd, p_g1a[p], p_be1a[p]);
            const float y1 = __builtin_fmaf(a1[p][r] * rstd, p_g1b[p], p_be1b[p]);
            if (r < nrows[p] && (bad_post_relu(y0) || bad_post_relu(y1))) st |= COEVO_ST_BAD_FC1;
            a0[p][r] = relu_keep_nan(y0);
            a1[p][r] = relu_keep_nan(y1);
        }
    asm volatile("" : "+v"(st));   // this phase's status bits settled here (see fc_policy_mfma16_body)
    COEVO_STAMP(2);
    COEVO_STAMP(11);

    // ---- fc2 on the matrix cores: lane owns output column 64w + l; its streamed 16-byte piece is W2[64w + l][4q..4q+3],
    //      used as is as the B operand of four v_mfma_f32_4x4x1_16B_f32 per row group ------------------------------
    // (Measured: a single left-over row - R = 5 - as VALU FMAs instead of a second 4-row MFMA group halves these
    // workgroups' matrix-pipe use but its serial fma chain and extra broadcast read cost more: 449 vs 516.)
    typedef float f32x4_acc __attribute__((ext_vector_type(4)));
    constexpr int NG = FcSmem<R, P>::NG;
    f32x4_acc acc[P][NG];  // acc[p][g][i]: row 4g + i of net p, column 64w + l
#pragma unroll
    for (int p = 0; p < P; ++p) {
        if (p > 0) __syncthreads();  // every wave is done with the previous net's image
#pragma unroll
        for (int r = 0; r < R; ++r) {
            sm.h1q[t >> 2][r][t & 3] = a0[p][r];
            sm.h1q[(t + 256) >> 2][r][t & 3] = a1[p][r];
        }
        __syncthreads();
        COEVO_STAMP(12);
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[p][g][i] = p_b2[p];
        const float4 *wp = reinterpret_cast<const float4 *>(net[p] + fc_off_w2(D[p])) + (size_t)w * 128 * 64 + l;
        // Two register buffers of U pieces in ping-pong, the order pinned with sched_barrier: while one buffer's U
        // pieces feed the matrix pipe the other buffer's U loads are in flight.  (Left alone, the scheduler interleaves
        // loads and MFMAs pairwise with s_waitcnt vmcnt(1) in between - two loads in flight per lane, 28 us per net.
        // Earlier VALU versions of this loop: C++ double buffers were sunk below their consumers, inline-asm loads
        // spilled; with the activations in SGPRs through a global scratch and s_load a lone workgroup ran 22.7 instead
        // of 27 us but the 16 KiB scalar cache thrashed with several workgroups per CU.)
        constexpr int U = COEVO_LIGHT_U;
        static_assert(128 % (2 * U) == 0, "the k-quads are consumed in pairs of buffers");
        float4 bufA[U], bufB[U];
        auto issue = [&](float4 (&buf)[U], int kq) {
#pragma unroll
            for (int u = 0; u < U; ++u) buf[u] = load_stream16(wp + (size_t)(kq + u) * 64);
        };
        auto consume = [&](const float4 (&buf)[U], int kq) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float4 x[NG];
#pragma unroll
                for (int g = 0; g < NG; ++g) x[g] = *reinterpret_cast<const float4 *>(&sm.h1q[kq + u][4 * g + (l & 3)][0]);
#pragma unroll
                for (int g = 0; g < NG; ++g) acc[p][g] = __builtin_amdgcn_mfma_f32_4x4x1f32(x[g].x, buf[u].x, acc[p][g], 0, 0, 0);
#pragma unroll
                for (int g = 0; g < NG; ++g) acc[p][g] = __builtin_amdgcn_mfma_f32_4x4x1f32(x[g].y, buf[u].y, acc[p][g], 0, 0, 0);
#pragma unroll
                for (int g = 0; g < NG; ++g) acc[p][g] = __builtin_amdgcn_mfma_f32_4x4x1f32(x[g].z, buf[u].z, acc[p][g], 0, 0, 0);
#pragma unroll
                for (int g = 0; g < NG; ++g) acc[p][g] = __builtin_amdgcn_mfma_f32_4x4x1f32(x[g].w, buf[u].w, acc[p][g], 0, 0, 0);
            }
        };
        issue(bufA, 0);
        int kq = 0;
        for (; kq < 128 - 2 * U; kq += 2 * U) {
            issue(bufB, kq + U);
            __builtin_amdgcn_sched_barrier(0);
            consume(bufA, kq);
            issue(bufA, kq + 2 * U);
            __builtin_amdgcn_sched_barrier(0);
            consume(bufB, kq + U);
        }
        issue(bufB, kq + U);
        __builtin_amdgcn_sched_barrier(0);
        consume(bufA, kq);
        consume(bufB, kq + U);
    }
    COEVO_STAMP(3);
    // ---- LayerNorm(256) + ReLU: canonical block b = wave b ------------------------------------------------------
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float s = wave_tree_sum(acc[p][r >> 2][r & 3]);
            if (l == 0) sm.red[p][r][w] = s;
        }
    __syncthreads();  // also: every wave is done reading h1q, h2 may overwrite it below
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float *rr = sm.red[p][r];
            const float tot = ((rr[0] + rr[1]) + rr[2]) + rr[3];
            acc[p][r >> 2][r & 3] = acc[p][r >> 2][r & 3] - tot * (1.0f / H2);
        }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float d = acc[p][r >> 2][r & 3];
            const float s = wave_tree_sum(d * d);
            if (l == 0) sm.red[p][r][w] = s;
        }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float *rr = sm.red[p][r];
            const float tot = ((rr[0] + rr[1]) + rr[2]) + rr[3];
            const float rstd = 1.0f / __builtin_sqrtf(tot * (1.0f / H2) + LN_EPS);
            const float y = __builtin_fmaf(acc[p][r >> 2][r & 3] * rstd, p_g2[p], p_be2[p]);
            if (r < nrows[p] && bad_post_relu(y)) st |= COEVO_ST_BAD_FC2;
            sm.h2[p][r][t] = relu_keep_nan(y);
        }
    asm volatile("" : "+v"(st));   // this phase's status bits settled here (see fc_policy_mfma16_body)
    __syncthreads();
    COEVO_STAMP(4);

    // ---- output layer: one lane per (row, action), 256-long sequential chain out of LDS --------------------
    if (row_wave && l < R * NACT) {
        const int r = l / NACT, o = l % NACT;
        float y = p_b3;
        const float4 *wr = reinterpret_cast<const float4 *>(&sm.w3s[pw][o][0]);
        const float4 *xr = reinterpret_cast<const float4 *>(&sm.h2[pw][r][0]);
#pragma unroll 8
        for (int k = 0; k < H2 / 4; ++k) {
            const float4 wv = wr[k], xv = xr[k];
            y = __builtin_fmaf(wv.x, xv.x, y);
            y = __builtin_fmaf(wv.y, xv.y, y);
            y = __builtin_fmaf(wv.z, xv.z, y);
            y = __builtin_fmaf(wv.w, xv.w, y);
        }
        sm.logit[pw][r][o] = y;
    }
    __syncthreads();
    COEVO_STAMP(5);

    // ---- first-max action (strict '>' scan from -inf), status --------------------------------------------
    if (row_wave && l < nrows[pw]) {
        int best = -1;
        float cur = -__builtin_inff();
#pragma unroll
        for (int o = 0; o < NACT; ++o) {
            const float v = sm.logit[pw][l][o];
            if (!__builtin_isfinite(v)) st |= COEVO_ST_BAD_OUT;
            if (v > cur) { cur = v; best = o; }
        }
        if (best < 0) { st |= COEVO_ST_NO_ACTION; best = 0; }
        const int row = row0[pw] + l;
        if constexpr (MODE >= MODE_FUSED) {
            a.act_cur[3 * a.row_game[row] + a.row_slot[row]] = best;  // by (game, slot)
        } else {
            a.actions[row] = best;
        }
        if (a.logits) {
#pragma unroll
            for (int o = 0; o < NACT; ++o) a.logits[(size_t)row * COEVO_LOGIT_STRIDE + o] = sm.logit[pw][l][o];
        }
    }
    if (st) atomicOr(a.status, st);
    COEVO_STAMP(6);
}

// =====================================================================================================
// The per-individual body of the lean merged cycle kernel (COEVO_COMPACT, default), written for FEW ISSUED INSTRUCTIONS.
// In a cycle launch all sixteen waves of a CU run the same phase at the same time and the chip clocks at ~1.5 GHz under
// the weight stream, so everything outside the stream is bound by the SIMDs' issue slots, not by latency or instruction
// fetch (tools/merged_wg_times.py: per-wave stamps + shader-clock stamps): with fc1 as 100 VALU FMAs fed by LDS
// broadcasts and one 6-level butterfly per (row, block) value (fc_policy_body) a workgroup spends ~26 of its 67 us between
// the entry barrier and the stream.  Here
//   * fc1 runs on the matrix pipe like fc2 (v_mfma_f32_4x4x1, rows in groups of four: 40 MFMAs instead of 100 FMAs + 70
//     LDS reads), its weights and the LayerNorm parameters staged by LDS-DMA (no registers held across the env step);
//   * the 2 R block sums a wave owes per LayerNorm pass come from ONE packed butterfly (coevo_common.hip.h:
//     packed_totals, ~4 instructions per value instead of 11);
//   * mean / variance / rstd of a row are computed by one lane per row and broadcast with v_readlane, not by every thread;
//   * the stream loop carries its own tail (no peeled copy of the last iteration), the last barrier is a wave fence.
// The arithmetic - every fmaf chain, every reduction tree, the left-to-right block sums - is the canonical one, so the
// bits are those of fc_policy_body and of the oracle.
template <int R>
struct FcSmemC {
    static constexpr int NG = (R + 3) / 4;   // row groups of four = v_mfma_f32_4x4x1 issues per k
    static constexpr int RP = 4 * NG + 1;    // odd row pitch of the k-quad image (see FcSmem)
    union {
        float par[13 * H1];                  // W1t [D][512], fc1.bias, ln1.weight, ln1.bias: contiguous in the slab
        float h1q[128][RP][4];               // A operands of fc2 (written once every wave is done with `par`)
        float h1r[R][H1];                    // ... row-major for the vector-ALU fc2 of the small-launch kernel (FC2_DPP)
        struct { float h2[R][260]; float w3s[NACT][260]; } tail;   // after the stream
    };
    float xs0[4 * NG][COEVO_OBS_STRIDE];     // observations, rows padded to whole groups (zeros)
    float red[4][16], red2[4][16];           // per wave: its block sums of every row (sums, squared deviations)
    float logit[R][COEVO_LOGIT_STRIDE];
};

// fc2 of the per-individual body, two forms with the same bits (each output is the sequential-k fmaf chain from its bias):
//   FC2_MFMA  v_mfma_f32_4x4x1, rows in groups of four.  The form of the full launch, where several workgroups share a SIMD: a
//             dependent 4x4x1 issues only every 40-56 cycles (tools/mfma_rate_probe.hip chain: one accumulator 40, two 2 x 28),
//             which other workgroups' waves fill.
//   FC2_DPP   v_fmac_f32 with the activation as a DPP row_newbcast source: one VGPR holds 16 consecutive activations of a row
//             (lane % 16 = k), `row_newbcast:j` hands lane j's value to every lane of its 16-lane row inside the fmac - no
//             matrix instruction, no LDS read per k, R independent chains per lane.  The form of a SMALL launch (a rank of a
//             sharded population: fewer workgroups than CUs), where a workgroup has its SIMDs to itself and the two 4x4x1
//             chains of a wave leave the matrix pipe idle 70 % of the time: 512 k x 56 cycles = 12.1 us of a 19 us workgroup
//             (tools/merged_wg_times.py, COEVO_PROBE_SHARD=8), against 512 x R x 6.3 cycles = 6.8 us (tools/fmac_dpp_probe.hip).
//             Its weight stream uses PLAIN loads: a small launch's nets (<= 256 x 0.56 MB) are re-read every env-cycle out of
//             the 256 MiB Infinity Cache, whose shorter latency is bandwidth to a wave with ~16 loads in flight (a rank of 8:
//             16.4 against 18.7 us per launch with nt loads; the full population, 336 MB, streams from HBM and wants nt).
constexpr int FC2_MFMA = 0, FC2_DPP = 1;

#ifndef COEVO_SMALL_U
#define COEVO_SMALL_U 16   // 16-byte pieces per lane and buffer in the small-launch body (two buffers; <= 256 registers)
#endif

template <int J>
__device__ __forceinline__ void fmac_bcast(float &acc, float xv, float w)
{
    // acc = fmaf(xv[lane j of this lane's 16-lane row], w, acc): one rounding, like __builtin_fmaf
    asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(xv), "v"(w), "n"(J));
}

// four k-quads (16 consecutive k) of every row's chain: xv[r] holds activations k0 .. k0 + 15 of row r (lane % 16 = k - k0)
template <int R>
__device__ __forceinline__ void fmac_block16(float (&acc)[R], const float (&xv)[R], const float4 &p0, const float4 &p1,
                                             const float4 &p2, const float4 &p3)
{
#define COEVO_FMAC_K(J, W)                                   \
    _Pragma("unroll") for (int r = 0; r < R; ++r) fmac_bcast<J>(acc[r], xv[r], W);
    COEVO_FMAC_K(0, p0.x) COEVO_FMAC_K(1, p0.y) COEVO_FMAC_K(2, p0.z) COEVO_FMAC_K(3, p0.w)
    COEVO_FMAC_K(4, p1.x) COEVO_FMAC_K(5, p1.y) COEVO_FMAC_K(6, p1.z) COEVO_FMAC_K(7, p1.w)
    COEVO_FMAC_K(8, p2.x) COEVO_FMAC_K(9, p2.y) COEVO_FMAC_K(10, p2.z) COEVO_FMAC_K(11, p2.w)
    COEVO_FMAC_K(12, p3.x) COEVO_FMAC_K(13, p3.y) COEVO_FMAC_K(14, p3.z) COEVO_FMAC_K(15, p3.w)
#undef COEVO_FMAC_K
}

template <int R, int MODE, int FC2 = FC2_MFMA>
__device__ __forceinline__ void fc_policy_body_c(const FcArgs &a, FcSmemC<R> &sm, const coevo_fc_task *tasks, int first,
                                                 int n_tasks)
{
    static_assert((MODE == MODE_FUSED || MODE == MODE_OBS) && R * NACT <= 64 && R <= 8, "the lean merged cycle kernel");
    typedef float f32x4_acc __attribute__((ext_vector_type(4)));
    constexpr int NG = FcSmemC<R>::NG;
    COEVO_STAMP(0);
    const int t = threadIdx.x, w = t >> 6, l = t & 63;
    const bool active = first < n_tasks;
    const coevo_fc_task task = tasks[active ? first : n_tasks - 1];
    const int D = task.D, nrows = active ? task.n_rows : 0, row0 = task.row_begin;
    const float *net = a.slab + task.net_off;
    int st = 0;

    // ---- W1t, fc1.bias and the LayerNorm(512) affine: (D + 3) * 512 contiguous floats, by LDS-DMA in 16-byte pieces (a
    //      wave's 64 pieces land contiguously at its wave-uniform base; (D + 3) * 128 pieces are a multiple of 64) ------
    const int n_pieces = (D + 3) * (H1 / 4);
#pragma unroll
    for (int j = 0; j < 7; ++j)
        if (64 * w + 256 * j < n_pieces)   // wave-uniform
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(net + 4 * (t + 256 * j)),
                                             (__attribute__((address_space(3))) void *)(&sm.par[4 * (64 * w + 256 * j)]),
                                             16, 0, 0);
    const float *b2p = net + fc_off_b2(D);
    float w3r[5];
    {
        const float *W3 = net + fc_off_w3(D);
#pragma unroll
        for (int j = 0; j < 5; ++j) w3r[j] = W3[t + 256 * j];
    }
    const float p_b2 = b2p[t], p_g2 = b2p[H2 + t], p_be2 = b2p[2 * H2 + t];
    const float p_b3 = (w == 0 && l < R * NACT) ? net[fc_off_b3(D) + l % NACT] : 0.0f;
    constexpr int US = COEVO_SMALL_U;
    static_assert(US % 4 == 0 && 128 % (2 * US) == 0, "buffers of whole 16-k blocks, consumed in pairs");
    const float4 *wps = reinterpret_cast<const float4 *>(net + fc_off_w2(D)) + (size_t)w * 128 * 64 + l;
    float4 sbufA[FC2 == FC2_DPP ? US : 1], sbufB[FC2 == FC2_DPP ? US : 1];

    // ---- env step + observation: one lane per row (wave 0) ----------------------------------------------------
    if (w == 0 && l < 4 * NG) {
        float o[COEVO_OBS_STRIDE];
#pragma unroll
        for (int k = 0; k < COEVO_OBS_STRIDE; ++k) o[k] = 0.0f;
        if (l < nrows) {
            const int row = row0 + l;
            if constexpr (MODE == MODE_FUSED) {
                mpe_fused_observe(a.state, a.state_next, a.act_prev, a.game_limit, a.n_games, a.row_game[row],
                                  a.row_slot[row], a.cycle, a.pos_first, o);
            } else {   // observations given (env stepped elsewhere, e.g. on the host cores): columns >= D are not inputs
#pragma unroll
                for (int k = 0; k < 10; ++k) o[k] = (k < D) ? a.obs[(size_t)row * COEVO_OBS_STRIDE + k] : 0.0f;
            }
#pragma unroll
            for (int k = 0; k < 10; ++k)
                if (!__builtin_isfinite(o[k])) st |= COEVO_ST_BAD_INPUT;
        }
#pragma unroll
        for (int k = 0; k < COEVO_OBS_STRIDE; ++k) sm.xs0[l][k] = o[k];   // rows R .. 4 NG - 1: zeros (tile padding)
    }
    COEVO_WSTAMP(0);
    __syncthreads();   // (waits for the LDS-DMA too: the fence counts it on vmcnt)
    COEVO_WSTAMP(1);
    COEVO_CSTAMP(10);
    COEVO_STAMP(1);
    // FC2_DPP: the first two buffers of the fc2 stream (a quarter of the net at U = 16) are requested NOW - fc1 and
    // LayerNorm(512) take ~2 us in which nothing else is in flight; the barriers below do not drain them.  (Requested before
    // the entry barrier they delayed it by 1.7 us: that barrier's vmcnt(0) covers the LDS-DMA and, in order, everything else,
    // and wave 0's game-state reads queued behind 32 KiB of stream.)
    if constexpr (FC2 == FC2_DPP) {
#pragma unroll
        for (int uu = 0; uu < US; ++uu) sbufA[uu] = wps[(size_t)uu * 64];
#pragma unroll
        for (int uu = 0; uu < US; ++uu) sbufB[uu] = wps[(size_t)(US + uu) * 64];
    }

    // ---- fc1 on the matrix cores, like fc2: wave w owns features 64 w + l (block w) and 256 + 64 w + l (block 4 + w);
    //      sequential-k chains from the bias, one k per v_mfma_f32_4x4x1 (rows in groups of four) ---------------------
    const float *b1s = sm.par + D * H1;
    f32x4_acc c0[NG], c1[NG];
    {
        const float bia = b1s[t], bib = b1s[t + 256];
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int i = 0; i < 4; ++i) { c0[g][i] = bia; c1[g][i] = bib; }
        float4 xk[NG][3];   // observations of row 4 g + l % 4: 12 floats (10 used)
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int j = 0; j < 3; ++j) xk[g][j] = *reinterpret_cast<const float4 *>(&sm.xs0[4 * g + (l & 3)][4 * j]);
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            if (k < D) {   // wave-uniform
                const float wa = sm.par[k * H1 + t], wb = sm.par[k * H1 + 256 + t];
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const float4 xv = xk[g][k >> 2];
                    const float x = (k & 3) == 0 ? xv.x : (k & 3) == 1 ? xv.y : (k & 3) == 2 ? xv.z : xv.w;
                    c0[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(x, wa, c0[g], 0, 0, 0);
                    c1[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(x, wb, c1[g], 0, 0, 0);
                }
            }
        }
    }
    // LayerNorm parameters of this thread's two features: read before `par` may be overwritten
    const float p_g1a = b1s[H1 + t], p_g1b = b1s[H1 + t + 256], p_be1a = b1s[2 * H1 + t], p_be1b = b1s[2 * H1 + t + 256];
    COEVO_STAMP(8);
    // ---- LayerNorm(512) + ReLU.  Per wave 2 R block sums (rows x its two blocks): one packed butterfly; the eight
    //      partials of a row are combined and turned into mean / rstd by ONE lane per row, then broadcast ---------------
    float v[2 * R];
#pragma unroll
    for (int r = 0; r < R; ++r) { v[2 * r] = c0[r >> 2][r & 3]; v[2 * r + 1] = c1[r >> 2][r & 3]; }
    {
        const float s = packed_totals<2 * R>(v, l);
        if (l < 2 * R) sm.red[w][l] = s;          // red[wave][2 r + half]: block (4 half + wave) of row r
    }
    COEVO_STAMP(9);
    COEVO_WSTAMP(2);
    __syncthreads();
    COEVO_WSTAMP(3);
    COEVO_STAMP(10);
    const int lr = l < R ? l : R - 1;             // lane r < R works on row r (the other lanes repeat the last row)
    {
        float tot = sm.red[0][2 * lr];            // blocks left to right: 0..3 = first halves, 4..7 = second halves
#pragma unroll
        for (int b = 1; b < 8; ++b) tot = tot + sm.red[b & 3][2 * lr + (b >> 2)];
        const float meanv = tot * (1.0f / H1);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float m = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(meanv), r));
            v[2 * r] = v[2 * r] - m;
            v[2 * r + 1] = v[2 * r + 1] - m;
        }
    }
    {
        float sq[2 * R];
#pragma unroll
        for (int j = 0; j < 2 * R; ++j) sq[j] = v[j] * v[j];
        const float s = packed_totals<2 * R>(sq, l);
        if (l < 2 * R) sm.red2[w][l] = s;
    }
    COEVO_WSTAMP(4);
    __syncthreads();   // every wave is also done with `par` (its fc1 weights and LayerNorm parameters are in registers)
    COEVO_WSTAMP(5);
    {
        float tot = sm.red2[0][2 * lr];
#pragma unroll
        for (int b = 1; b < 8; ++b) tot = tot + sm.red2[b & 3][2 * lr + (b >> 2)];
        const float rstdv = 1.0f / __builtin_sqrtf(tot * (1.0f / H1) + LN_EPS);
        bool bad = false;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float rstd = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rstdv), r));
            const float y0 = __builtin_fmaf(v[2 * r] * rstd, p_g1a, p_be1a);
            const float y1 = __builtin_fmaf(v[2 * r + 1] * rstd, p_g1b, p_be1b);
            if (r < nrows) bad = bad || bad_post_relu(y0) || bad_post_relu(y1);
            if constexpr (FC2 == FC2_DPP) {
                sm.h1r[r][t] = relu_keep_nan(y0);
                sm.h1r[r][t + 256] = relu_keep_nan(y1);
            } else {
                sm.h1q[t >> 2][r][t & 3] = relu_keep_nan(y0);
                sm.h1q[(t + 256) >> 2][r][t & 3] = relu_keep_nan(y1);
            }
        }
        if (bad) st |= COEVO_ST_BAD_FC1;
    }
    COEVO_WSTAMP(6);
    __syncthreads();
    COEVO_WSTAMP(7);
    COEVO_CSTAMP(11);
    COEVO_STAMP(2);

    // ---- fc2 on the matrix cores, as in fc_policy_body; the ping-pong loop carries its own tail ------------------
    f32x4_acc acc[NG];
    float u[R];
    if constexpr (FC2 == FC2_MFMA) {
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[g][i] = p_b2;
    {
        const float4 *wp = reinterpret_cast<const float4 *>(net + fc_off_w2(D)) + (size_t)w * 128 * 64 + l;
        constexpr int U = COEVO_LIGHT_U;
        static_assert(128 % (2 * U) == 0, "the k-quads are consumed in pairs of buffers");
        float4 bufA[U], bufB[U];
        auto issue = [&](float4 (&buf)[U], int kq) {
#pragma unroll
            for (int u = 0; u < U; ++u) buf[u] = load_stream16(wp + (size_t)(kq + u) * 64);
        };
        auto consume = [&](const float4 (&buf)[U], int kq) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float4 x[NG];
#pragma unroll
                for (int g = 0; g < NG; ++g) x[g] = *reinterpret_cast<const float4 *>(&sm.h1q[kq + u][4 * g + (l & 3)][0]);
#pragma unroll
                for (int g = 0; g < NG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(x[g].x, buf[u].x, acc[g], 0, 0, 0);
#pragma unroll
                for (int g = 0; g < NG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(x[g].y, buf[u].y, acc[g], 0, 0, 0);
#pragma unroll
                for (int g = 0; g < NG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(x[g].z, buf[u].z, acc[g], 0, 0, 0);
#pragma unroll
                for (int g = 0; g < NG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(x[g].w, buf[u].w, acc[g], 0, 0, 0);
            }
        };
        issue(bufA, 0);
#pragma nounroll
        for (int kq = 0; kq < 128; kq += 2 * U) {
            issue(bufB, kq + U);
            __builtin_amdgcn_sched_barrier(0);
            consume(bufA, kq);
            if (kq + 2 * U < 128) issue(bufA, kq + 2 * U);   // wave-uniform
            __builtin_amdgcn_sched_barrier(0);
            consume(bufB, kq + U);
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) u[r] = acc[r >> 2][r & 3];
    } else {
        // ---- fc2 on the vector ALU (FC2_DPP, see above): lane = output column 64 w + l, R chains per lane; per 16 k one
        //      ds_read_b32 per row (16 consecutive activations, the same in all four 16-lane rows) and 16 R fmacs ----------
#pragma unroll
        for (int r = 0; r < R; ++r) u[r] = p_b2;
        const float *xrow = &sm.h1r[0][l & 15];
        auto issue_s = [&](float4 (&buf)[FC2 == FC2_DPP ? US : 1], int kq) {
#pragma unroll
            for (int uu = 0; uu < US; ++uu) buf[uu] = wps[(size_t)(kq + uu) * 64];   // plain, not nt: see FC2_DPP
        };
        auto consume_s = [&](const float4 (&buf)[FC2 == FC2_DPP ? US : 1], int kq) {
#pragma unroll
            for (int b = 0; b < US / 4; ++b) {
                float xv[R];
#pragma unroll
                for (int r = 0; r < R; ++r) xv[r] = xrow[r * H1 + 4 * (kq + 4 * b)];
                fmac_block16<R>(u, xv, buf[4 * b], buf[4 * b + 1], buf[4 * b + 2], buf[4 * b + 3]);
            }
        };
#pragma nounroll
        for (int kq = 0; kq < 128; kq += 2 * US) {
            consume_s(sbufA, kq);
            if (kq + 2 * US < 128) issue_s(sbufA, kq + 2 * US);   // wave-uniform
            __builtin_amdgcn_sched_barrier(0);
            consume_s(sbufB, kq + US);
            if (kq + 3 * US < 128) issue_s(sbufB, kq + 3 * US);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    COEVO_STAMP(3);
    COEVO_WSTAMP(8);
    COEVO_CSTAMP(12);
    // ---- LayerNorm(256) + ReLU: canonical block b = wave b; R block sums per wave by one packed butterfly ---------------
    {
        const float s = packed_totals<R>(u, l);
        if (l < R) sm.red[w][l] = s;   // (red / red2 live outside the union: no wave still needs the old contents)
    }
    __syncthreads();  // also: every wave is done reading h1q, the tail image may overwrite it below
    {
        const float tot = ((sm.red[0][lr] + sm.red[1][lr]) + sm.red[2][lr]) + sm.red[3][lr];
        const float meanv = tot * (1.0f / H2);
#pragma unroll
        for (int r = 0; r < R; ++r) u[r] = u[r] - __int_as_float(__builtin_amdgcn_readlane(__float_as_int(meanv), r));
        float sq[R];
#pragma unroll
        for (int r = 0; r < R; ++r) sq[r] = u[r] * u[r];
        const float s = packed_totals<R>(sq, l);
        if (l < R) sm.red2[w][l] = s;
    }
#pragma unroll
    for (int j = 0; j < 5; ++j) sm.tail.w3s[(t + 256 * j) >> 8][(t + 256 * j) & 255] = w3r[j];
    __syncthreads();
    {
        const float tot = ((sm.red2[0][lr] + sm.red2[1][lr]) + sm.red2[2][lr]) + sm.red2[3][lr];
        const float rstdv = 1.0f / __builtin_sqrtf(tot * (1.0f / H2) + LN_EPS);
        bool bad = false;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float rstd = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rstdv), r));
            const float y = __builtin_fmaf(u[r] * rstd, p_g2, p_be2);
            if (r < nrows) bad = bad || bad_post_relu(y);
            sm.tail.h2[r][t] = relu_keep_nan(y);
        }
        if (bad) st |= COEVO_ST_BAD_FC2;
    }
    __syncthreads();
    COEVO_STAMP(4);

    // ---- output layer (one lane per (row, action), 256-long sequential chain out of LDS), first-max action: wave 0 ----
    if (w == 0) {
        if (l < R * NACT) {
            const int r = l / NACT, o = l % NACT;
            float y = p_b3;
            const float4 *wr = reinterpret_cast<const float4 *>(&sm.tail.w3s[o][0]);
            const float4 *xr = reinterpret_cast<const float4 *>(&sm.tail.h2[r][0]);
            // (the small-launch form has registers to spare: 16 k-quads of both operands requested at a time, so that the
            // 256-step chain waits for LDS four times instead of sixteen)
            constexpr int OUT_UNROLL = FC2 == FC2_DPP ? 16 : 4;
#pragma unroll OUT_UNROLL
            for (int k = 0; k < H2 / 4; ++k) {
                const float4 wv = wr[k], xv = xr[k];
                y = __builtin_fmaf(wv.x, xv.x, y);
                y = __builtin_fmaf(wv.y, xv.y, y);
                y = __builtin_fmaf(wv.z, xv.z, y);
                y = __builtin_fmaf(wv.w, xv.w, y);
            }
            sm.logit[r][o] = y;
        }
        COEVO_STAMP(5);
        // the same wave wrote the logits: a wave-level fence orders its LDS accesses, no workgroup barrier needed
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (l < nrows) {   // strict '>' scan from -inf
            int best = -1;
            float cur = -__builtin_inff();
#pragma unroll
            for (int o = 0; o < NACT; ++o) {
                const float vv = sm.logit[l][o];
                if (!__builtin_isfinite(vv)) st |= COEVO_ST_BAD_OUT;
                if (vv > cur) { cur = vv; best = o; }
            }
            if (best < 0) { st |= COEVO_ST_NO_ACTION; best = 0; }
            const int row = row0 + l;
            if constexpr (MODE == MODE_FUSED) a.act_cur[3 * a.row_game[row] + a.row_slot[row]] = best;  // by (game, slot)
            else a.actions[row] = best;
            if (a.logits) {
#pragma unroll
                for (int o = 0; o < NACT; ++o) a.logits[(size_t)row * COEVO_LOGIT_STRIDE + o] = sm.logit[l][o];
            }
        }
    }
    if (st) atomicOr(a.status, st);
    COEVO_STAMP(6);
}

#ifndef COEVO_COMPACT
#define COEVO_COMPACT 1   // 0: the unrolled per-individual body everywhere (A/B runs)
#endif
template <int R, int MODE>
__global__ __launch_bounds__(256) void fc_policy_kernel(FcArgs a)
{
    stamp_begin(a.stamps);
    if constexpr (COEVO_COMPACT && MODE == MODE_FUSED && R <= 8) {   // fused env step: the body built for few instructions
        __shared__ __attribute__((aligned(16))) FcSmemC<R> smc;
        fc_policy_body_c<R, MODE>(a, smc, a.tasks, blockIdx.x, gridDim.x);
    } else {
        __shared__ __attribute__((aligned(16))) FcSmem<R, 1> sm;
        fc_policy_body<R, MODE, 1>(a, sm, a.tasks, blockIdx.x, gridDim.x);
    }
    stamp_end(a.stamps);
}

// =====================================================================================================
// Shared-opponent tasks (a HoF member / ES base net applied to many games): a real GEMM with M = rows, so the
// dense layers run on the matrix cores.  v_mfma_f32_32x32x2_f32 with C-in = bias is bit-for-bit the canonical
// sequential-k fmaf chain (tools/mfma_chain_probe.hip), so this kernel and the VALU kernel above give identical bits.
//
// One workgroup = one net x up to 32 rows (the M of a 32x32 tile); wave w owns output columns
//   fc1: [128w, 128w+128) = 4 tiles, fc2: [64w, 64w+64) = 2 tiles.
// A operand (activations, [row][k]) comes from LDS in an image laid out so that one ds_read_b128 per lane feeds four
// consecutive MFMAs; B operand (weights) is the same 16-byte row piece stream as the VALU kernel (W2q), turned into
// two k-pair operands for each of the wave's two column tiles by two v_permlane32_swap.
// Accumulator layout (32x32): col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

struct FcMfmaSmem {
    union {
        float h1a[64][64][4];  // [k/8][lane = 32*(k&1) + row][(k>>1)&3]: A operands of fc2, 64 KiB
        float h2[32][260];
    };
    float xst[COEVO_OBS_STRIDE][32];  // observations transposed [k][row]: A operands of fc1
    float w3s[NACT][260];
    float red[32][8];
    float logit[32][COEVO_LOGIT_STRIDE];
};

__device__ inline int mfma_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

template <int MODE>
__device__ __forceinline__ void fc_policy_mfma_body(const FcArgs &a, FcMfmaSmem &sm, const coevo_fc_task &task)
{
    // these workgroups are the long pole of a cycle when they share CUs with the streaming kernel's waves: let their
    // (few) waves win issue arbitration; the streaming waves are waiting on HBM most of the time anyway
#ifndef COEVO_HEAVY_PRIO
#define COEVO_HEAVY_PRIO 3
#endif
    __builtin_amdgcn_s_setprio(COEVO_HEAVY_PRIO);
    COEVO_STAMP(0);
    const int t = threadIdx.x, w = t >> 6, l = t & 63, lc = l & 31, lh = l >> 5;
    const int D = task.D, nrows = task.n_rows, row0 = task.row_begin;
    const float *net = a.slab + task.net_off;
    int st = 0;

    // ---- all small parameters and the whole fc1 B operand requested up front (one HBM/L2 round trip) ----------
    constexpr int KP = 5;  // k-pairs of fc1 (D <= 10)
    float w1[KP][4], p_b1[4], p_g1[4], p_be1[4], p_b2[2], p_g2[2], p_be2[2];
    {
        const float *b1p = net + fc_off_b1(D), *b2p = net + fc_off_b2(D);
#pragma unroll
        for (int kp = 0; kp < KP; ++kp)
#pragma unroll
            for (int tl = 0; tl < 4; ++tl)
                w1[kp][tl] = (2 * kp < D) ? net[(size_t)(2 * kp + lh) * H1 + 128 * w + 32 * tl + lc] : 0.0f;
#pragma unroll
        for (int tl = 0; tl < 4; ++tl) {
            const int j = 128 * w + 32 * tl + lc;
            p_b1[tl] = b1p[j]; p_g1[tl] = b1p[H1 + j]; p_be1[tl] = b1p[2 * H1 + j];
        }
#pragma unroll
        for (int tl = 0; tl < 2; ++tl) {
            const int j = 64 * w + 32 * tl + lc;
            p_b2[tl] = b2p[j]; p_g2[tl] = b2p[H2 + j]; p_be2[tl] = b2p[2 * H2 + j];
        }
    }
    const float p_b3 = (t < 32 * NACT) ? net[fc_off_b3(D) + t % NACT] : 0.0f;

    if constexpr (MODE >= MODE_FUSED) {
        if (t < 32) {
            float o[COEVO_OBS_STRIDE];
#pragma unroll
            for (int k = 0; k < COEVO_OBS_STRIDE; ++k) o[k] = 0.0f;
            if (t < nrows) {
                const int row = row0 + t;
                mpe_fused_observe(a.state, a.state_next, a.act_prev, a.game_limit, a.n_games, a.row_game[row],
                                  a.row_slot[row], a.cycle, a.pos_first, o);
#pragma unroll
                for (int k = 0; k < 10; ++k)
                    if (!__builtin_isfinite(o[k])) st |= COEVO_ST_BAD_INPUT;
            }
#pragma unroll
            for (int k = 0; k < COEVO_OBS_STRIDE; ++k) sm.xst[k][t] = o[k];
        }
    } else {
        for (int i = t; i < 32 * COEVO_OBS_STRIDE; i += 256) {
            const int r = i / COEVO_OBS_STRIDE, k = i % COEVO_OBS_STRIDE;
            float v = 0.0f;
            if (r < nrows && k < D) {
                if constexpr (MODE == MODE_STATE) {
                    const int row = row0 + r;
                    v = mpe_obs_element(a.state, a.n_games, a.row_game[row], a.row_slot[row], k);
                } else {
                    v = a.obs[(size_t)(row0 + r) * COEVO_OBS_STRIDE + k];
                }
                if (!__builtin_isfinite(v)) st |= COEVO_ST_BAD_INPUT;
            }
            sm.xst[k][r] = v;
        }
    }
    {
        const float *W3 = net + fc_off_w3(D);
        for (int i = t; i < NACT * H2; i += 256) sm.w3s[i >> 8][i & 255] = W3[i];
    }
    __syncthreads();
    COEVO_STAMP(1);

    // ---- fc1 on the matrix cores: 4 column tiles per wave, D/2 k-pairs ---------------------------------------
    f32x16 c1[4];
    {
#pragma unroll
        for (int tl = 0; tl < 4; ++tl) {
#pragma unroll
            for (int r = 0; r < 16; ++r) c1[tl][r] = p_b1[tl];
        }
#pragma unroll
        for (int kp = 0; kp < KP; ++kp) {
            if (2 * kp < D) {  // wave-uniform
                const float av = sm.xst[2 * kp + lh][lc];
#pragma unroll
                for (int tl = 0; tl < 4; ++tl)
                    c1[tl] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, w1[kp][tl], c1[tl], 0, 0, 0);
            }
        }
    }
    // ---- LayerNorm(512): canonical blocks 2w (tiles 0,1) and 2w+1 (tiles 2,3) --------------------------------
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float s01 = half_tree_sum(c1[0][r]) + half_tree_sum(c1[1][r]);
        const float s23 = half_tree_sum(c1[2][r]) + half_tree_sum(c1[3][r]);
        if (lc == 0) { sm.red[mfma_row(r, l)][2 * w] = s01; sm.red[mfma_row(r, l)][2 * w + 1] = s23; }
    }
    __syncthreads();
    float stat[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float *rr = sm.red[mfma_row(r, l)];
        float tot = rr[0];
#pragma unroll
        for (int b = 1; b < 8; ++b) tot = tot + rr[b];
        stat[r] = tot * (1.0f / H1);
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
        for (int tl = 0; tl < 4; ++tl) c1[tl][r] = c1[tl][r] - stat[r];
        const float s01 = half_tree_sum(c1[0][r] * c1[0][r]) + half_tree_sum(c1[1][r] * c1[1][r]);
        const float s23 = half_tree_sum(c1[2][r] * c1[2][r]) + half_tree_sum(c1[3][r] * c1[3][r]);
        if (lc == 0) { sm.red[mfma_row(r, l)][2 * w] = s01; sm.red[mfma_row(r, l)][2 * w + 1] = s23; }
    }
    __syncthreads();
    {
        const float *ga = p_g1, *bt = p_be1;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = mfma_row(r, l);
            const float *rr = sm.red[row];
            float tot = rr[0];
#pragma unroll
            for (int b = 1; b < 8; ++b) tot = tot + rr[b];
            const float rstd = 1.0f / __builtin_sqrtf(tot * (1.0f / H1) + LN_EPS);
#pragma unroll
            for (int tl = 0; tl < 4; ++tl) {
                const float y = __builtin_fmaf(c1[tl][r] * rstd, ga[tl], bt[tl]);
                if (row < nrows && bad_post_relu(y)) st |= COEVO_ST_BAD_FC1;
                const int k = 128 * w + 32 * tl + lc;  // this activation is input k of fc2
                sm.h1a[k >> 3][32 * (k & 1) + row][(k >> 1) & 3] = relu_keep_nan(y);
            }
        }
    }
    asm volatile("" : "+v"(st));   // this phase's status bits settled here (see fc_policy_mfma16_body)
    __syncthreads();
    COEVO_STAMP(2);

    // ---- fc2 on the matrix cores: 2 column tiles per wave, 256 k-pairs ---------------------------------------
    f32x16 c2[2];
    {
#pragma unroll
        for (int tl = 0; tl < 2; ++tl) {
#pragma unroll
            for (int r = 0; r < 16; ++r) c2[tl][r] = p_b2[tl];
        }
        const float4 *wp = reinterpret_cast<const float4 *>(net + fc_off_w2(D)) + (size_t)w * 128 * 64 + l;
        // software pipeline over k-octets, two register buffers in ping-pong: the other buffer's 2U weight pieces
        // are in flight while this buffer's U octets feed 8U MFMAs
        constexpr int U = 2;
        float4 bufA[2 * U], bufB[2 * U];
        auto issue = [&](float4 (&buf)[2 * U], int ko) {
#pragma unroll
            for (int u = 0; u < 2 * U; ++u) buf[u] = wp[(size_t)(2 * ko + u) * 64];
        };
        auto consume = [&](const float4 (&buf)[2 * U], int ko) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float4 av = *reinterpret_cast<const float4 *>(&sm.h1a[ko + u][l][0]);
                const float aop[4] = {av.x, av.y, av.z, av.w};
#pragma unroll
                for (int q = 0; q < 2; ++q) {  // the two k-quads of this octet
                    const float4 x = buf[2 * u + q];
                    const u32x2 s0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(x.x), __float_as_uint(x.y), false, false);
                    const u32x2 s1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(x.z), __float_as_uint(x.w), false, false);
                    c2[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(aop[2 * q], __uint_as_float(s0[0]), c2[0], 0, 0, 0);
                    c2[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(aop[2 * q], __uint_as_float(s0[1]), c2[1], 0, 0, 0);
                    c2[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(aop[2 * q + 1], __uint_as_float(s1[0]), c2[0], 0, 0, 0);
                    c2[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(aop[2 * q + 1], __uint_as_float(s1[1]), c2[1], 0, 0, 0);
                }
            }
        };
        issue(bufA, 0);
        int ko = 0;
        for (; ko < 64 - 2 * U; ko += 2 * U) {
            issue(bufB, ko + U);
            __builtin_amdgcn_sched_barrier(0);
            consume(bufA, ko);
            issue(bufA, ko + 2 * U);
            __builtin_amdgcn_sched_barrier(0);
            consume(bufB, ko + U);
        }
        issue(bufB, ko + U);
        __builtin_amdgcn_sched_barrier(0);
        consume(bufA, ko);
        consume(bufB, ko + U);
    }
    COEVO_STAMP(3);
    // ---- LayerNorm(256): canonical block w = this wave's two tiles -------------------------------------------
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float s = half_tree_sum(c2[0][r]) + half_tree_sum(c2[1][r]);
        if (lc == 0) sm.red[mfma_row(r, l)][w] = s;
    }
    __syncthreads();  // every wave is done with h1a: h2 may overwrite it below
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float *rr = sm.red[mfma_row(r, l)];
        stat[r] = (((rr[0] + rr[1]) + rr[2]) + rr[3]) * (1.0f / H2);
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        c2[0][r] = c2[0][r] - stat[r];
        c2[1][r] = c2[1][r] - stat[r];
        const float s = half_tree_sum(c2[0][r] * c2[0][r]) + half_tree_sum(c2[1][r] * c2[1][r]);
        if (lc == 0) sm.red[mfma_row(r, l)][w] = s;
    }
    __syncthreads();
    {
        const float *ga = p_g2, *bt = p_be2;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = mfma_row(r, l);
            const float *rr = sm.red[row];
            const float tot = ((rr[0] + rr[1]) + rr[2]) + rr[3];
            const float rstd = 1.0f / __builtin_sqrtf(tot * (1.0f / H2) + LN_EPS);
#pragma unroll
            for (int tl = 0; tl < 2; ++tl) {
                const float y = __builtin_fmaf(c2[tl][r] * rstd, ga[tl], bt[tl]);
                if (row < nrows && bad_post_relu(y)) st |= COEVO_ST_BAD_FC2;
                sm.h2[row][64 * w + 32 * tl + lc] = relu_keep_nan(y);
            }
        }
    }
    asm volatile("" : "+v"(st));   // this phase's status bits settled here (see fc_policy_mfma16_body)
    __syncthreads();
    COEVO_STAMP(4);

    // ---- output layer (N = 5: not worth a tile), argmax, status - as in the VALU kernel ----------------------
    if (t < 32 * NACT) {
        const int r = t / NACT, o = t % NACT;
        float y = p_b3;
        const float4 *wr = reinterpret_cast<const float4 *>(&sm.w3s[o][0]);
        const float4 *xr = reinterpret_cast<const float4 *>(&sm.h2[r][0]);
#pragma unroll 8
        for (int k = 0; k < H2 / 4; ++k) {
            const float4 wv = wr[k], xv = xr[k];
            y = __builtin_fmaf(wv.x, xv.x, y);
            y = __builtin_fmaf(wv.y, xv.y, y);
            y = __builtin_fmaf(wv.z, xv.z, y);
            y = __builtin_fmaf(wv.w, xv.w, y);
        }
        sm.logit[r][o] = y;
    }
    __syncthreads();
    COEVO_STAMP(5);
    if (t < nrows) {
        int best = -1;
        float cur = -__builtin_inff();
#pragma unroll
        for (int o = 0; o < NACT; ++o) {
            const float v = sm.logit[t][o];
            if (!__builtin_isfinite(v)) st |= COEVO_ST_BAD_OUT;
            if (v > cur) { cur = v; best = o; }
        }
        if (best < 0) { st |= COEVO_ST_NO_ACTION; best = 0; }
        if constexpr (MODE >= MODE_FUSED) {
            const int row = row0 + t;
            a.act_cur[3 * a.row_game[row] + a.row_slot[row]] = best;  // by (game, slot)
        } else {
            a.actions[row0 + t] = best;
        }
        if (a.logits) {
#pragma unroll
            for (int o = 0; o < NACT; ++o) a.logits[(size_t)(row0 + t) * COEVO_LOGIT_STRIDE + o] = sm.logit[t][o];
        }
    }
    if (st) atomicOr(a.status, st);
    COEVO_STAMP(6);
}

template <int MODE>
__global__ __launch_bounds__(256, 2) void fc_policy_mfma_kernel(FcArgs a)
{
    __shared__ __attribute__((aligned(16))) FcMfmaSmem sm;
    fc_policy_mfma_body<MODE>(a, sm, a.tasks[blockIdx.x]);
}

// =====================================================================================================
// The lean shared-opponent body: up to 16 rows per task on v_mfma_f32_16x16x4_f32 (tools/mfma16_chain_probe.hip: the
// same bits as the fmaf chain), <= 128 registers and < 40 KiB of LDS, so that FOUR workgroups fit a CU and the merged
// cycle launch keeps every streaming workgroup resident next to the shared-opponent ones (the 32-row body above costs
// 256 registers and 73 KiB: two workgroups per CU).  Wave w owns fc1 columns [128w, 128w+128) = 8 tiles and fc2 columns
// [64w, 64w+64) = 4 tiles.  Operands of one MFMA: lane (c = l%16, kk = l/16) supplies A[row c][k = 4q + kk] and
// B[k = 4q + kk][column c]; accumulator register i of lane (c, g = l/16) is row 4g + i, column c of the tile.
struct FcMfma16Smem {
    union {
        // A operands of fc2: h1a[k][row ^ ((k >> 2) & 3)] - the xor makes both the scatter from the accumulator layout
        // and the gather by (row, kk) conflict-free at a pitch of 16
        float h1a[H1][16];
        float h2[16][260];
    };
    float xst[12][16];  // observations transposed [k][row], zero padded to 12 inputs: A operands of fc1
    float w3s[NACT][260];
    float red[16][8];
    float logit[16][COEVO_LOGIT_STRIDE];
};
static_assert(sizeof(FcMfma16Smem) <= 40960, "four workgroups per CU");

template <int MODE>
__device__ __forceinline__ void fc_policy_mfma16_body(const FcArgs &a, FcMfma16Smem &sm, const coevo_fc_task &task)
{
    typedef float f32x4_acc __attribute__((ext_vector_type(4)));
    typedef unsigned u32x2_s __attribute__((ext_vector_type(2)));
    COEVO_STAMP(0);
    const int t = threadIdx.x, w = t >> 6, l = t & 63, lc = l & 15, lg = l >> 4;
    const int D = task.D, nrows = task.n_rows, row0 = task.row_begin;
    const float *net = a.slab + task.net_off;
    int st = 0;

    // ---- W1t, fc1.bias and the LayerNorm(512) affine - (D + 3) * 512 contiguous floats of the slab - by LDS-DMA into the
    //      region that later holds fc2's A operands (26 KiB of its 32), as the per-individual body stages them: requested
    //      first, so they do not queue behind the streaming workgroups' traffic, and NOT held in registers across the env
    //      step, whose fp64 state is this body's register peak (as 53 live registers they cost 24 bytes of scratch at four
    //      workgroups per CU).  Read into registers after the barrier below; the region is rewritten three barriers later.
    constexpr int KS = 3;  // k-steps of fc1 (D <= 12)
    float *par = &sm.h1a[0][0];
    {
        const int n_pieces = (D + 3) * (H1 / 4);
#pragma unroll
        for (int j = 0; j < 7; ++j)
            if (64 * w + 256 * j < n_pieces)   // wave-uniform
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(net + 4 * (t + 256 * j)),
                                                 (__attribute__((address_space(3))) void *)(par + 4 * (64 * w + 256 * j)),
                                                 16, 0, 0);
    }
    float p_b2[4], p_g2[4], p_be2[4];
    {
        const float *b2p = net + fc_off_b2(D);
#pragma unroll
        for (int T = 0; T < 4; ++T) p_b2[T] = b2p[64 * w + 16 * T + lc];
    }
    const float p_b3 = (t < 16 * NACT) ? net[fc_off_b3(D) + t % NACT] : 0.0f;

    if constexpr (MODE >= MODE_FUSED) {
        if (t < 16) {
            float o[COEVO_OBS_STRIDE];
#pragma unroll
            for (int k = 0; k < COEVO_OBS_STRIDE; ++k) o[k] = 0.0f;
            if (t < nrows) {
                const int row = row0 + t;
                mpe_fused_observe(a.state, a.state_next, a.act_prev, a.game_limit, a.n_games, a.row_game[row],
                                  a.row_slot[row], a.cycle, a.pos_first, o);
#pragma unroll
                for (int k = 0; k < 10; ++k)
                    if (!__builtin_isfinite(o[k])) st |= COEVO_ST_BAD_INPUT;
            }
#pragma unroll
            for (int k = 0; k < COEVO_OBS_STRIDE; ++k) sm.xst[k][t] = o[k];
        }
    } else {
        for (int i = t; i < 16 * COEVO_OBS_STRIDE; i += 256) {
            const int r = i / COEVO_OBS_STRIDE, k = i % COEVO_OBS_STRIDE;
            float v = 0.0f;
            if (r < nrows && k < D) {
                if constexpr (MODE == MODE_STATE) {
                    const int row = row0 + r;
                    v = mpe_obs_element(a.state, a.n_games, a.row_game[row], a.row_slot[row], k);
                } else {
                    v = a.obs[(size_t)(row0 + r) * COEVO_OBS_STRIDE + k];
                }
                if (!__builtin_isfinite(v)) st |= COEVO_ST_BAD_INPUT;
            }
            sm.xst[k][r] = v;
        }
    }
    {
        const float *W3 = net + fc_off_w3(D);
        for (int i = t; i < NACT * H2; i += 256) sm.w3s[i >> 8][i & 255] = W3[i];
    }
    // the LayerNorm(256) affine is not needed before the end of the fc2 stream: requested after the env step (whose fp64
    // state is the register peak of this body - held across it, these eight values were spilled to scratch), still long
    // before the stream's own loads
    {
        const float *b2p = net + fc_off_b2(D);
#pragma unroll
        for (int T = 0; T < 4; ++T) {
            const int j = 64 * w + 16 * T + lc;
            p_g2[T] = b2p[H2 + j]; p_be2[T] = b2p[2 * H2 + j];
        }
    }
    __syncthreads();   // (waits for the LDS-DMA too: the fence counts it on vmcnt)
    COEVO_STAMP(1);
    // (registers: the bias first, then one k-step's eight B operands at a time, the LayerNorm affine only after the matrix
    // instructions - all 48 parameter values at once next to the 32 accumulators were this body's new register peak)
    const float *b1s = par + (size_t)D * H1;

    // ---- fc1: 8 column tiles per wave, ceil(D/4) k-steps --------------------------------------------------
    f32x4_acc c1[8];
#pragma unroll
    for (int T = 0; T < 8; ++T) {
        const float b = b1s[128 * w + 16 * T + lc];
#pragma unroll
        for (int i = 0; i < 4; ++i) c1[T][i] = b;
    }
#pragma unroll
    for (int q = 0; q < KS; ++q) {
        if (4 * q < D) {  // wave-uniform
            const float av = sm.xst[4 * q + lg][lc];
            float w1[8];
#pragma unroll
            for (int T = 0; T < 8; ++T)  // a padded k contributes fma(0, -0, acc) = acc for every acc, -0 included
                w1[T] = (4 * q + lg < D) ? par[(size_t)(4 * q + lg) * H1 + 128 * w + 16 * T + lc] : -0.0f;
#pragma unroll
            for (int T = 0; T < 8; ++T) c1[T] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, w1[T], c1[T], 0, 0, 0);
        }
    }
    float p_g1[8], p_be1[8];
#pragma unroll
    for (int T = 0; T < 8; ++T) {
        const int j = 128 * w + 16 * T + lc;
        p_g1[T] = b1s[H1 + j]; p_be1[T] = b1s[2 * H1 + j];
    }
    // ---- LayerNorm(512): canonical blocks 2w (tiles 0..3) and 2w+1 (tiles 4..7); inside a block feature 16T' + lc:
    //      four tree levels inside the 16-lane row, then (T0 + T1) + (T2 + T3) ---------------------------------
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float s_lo = (row16_tree_sum(c1[0][i]) + row16_tree_sum(c1[1][i])) +
                           (row16_tree_sum(c1[2][i]) + row16_tree_sum(c1[3][i]));
        const float s_hi = (row16_tree_sum(c1[4][i]) + row16_tree_sum(c1[5][i])) +
                           (row16_tree_sum(c1[6][i]) + row16_tree_sum(c1[7][i]));
        if (lc == 0) { sm.red[4 * lg + i][2 * w] = s_lo; sm.red[4 * lg + i][2 * w + 1] = s_hi; }
    }
    __syncthreads();
    float stat[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float *rr = sm.red[4 * lg + i];
        float tot = rr[0];
#pragma unroll
        for (int b = 1; b < 8; ++b) tot = tot + rr[b];
        stat[i] = tot * (1.0f / H1);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int T = 0; T < 8; ++T) c1[T][i] = c1[T][i] - stat[i];
        const float s_lo = (row16_tree_sum(c1[0][i] * c1[0][i]) + row16_tree_sum(c1[1][i] * c1[1][i])) +
                           (row16_tree_sum(c1[2][i] * c1[2][i]) + row16_tree_sum(c1[3][i] * c1[3][i]));
        const float s_hi = (row16_tree_sum(c1[4][i] * c1[4][i]) + row16_tree_sum(c1[5][i] * c1[5][i])) +
                           (row16_tree_sum(c1[6][i] * c1[6][i]) + row16_tree_sum(c1[7][i] * c1[7][i]));
        if (lc == 0) { sm.red[4 * lg + i][2 * w] = s_lo; sm.red[4 * lg + i][2 * w + 1] = s_hi; }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 4 * lg + i;
        const float *rr = sm.red[row];
        float tot = rr[0];
#pragma unroll
        for (int b = 1; b < 8; ++b) tot = tot + rr[b];
        const float rstd = 1.0f / __builtin_sqrtf(tot * (1.0f / H1) + LN_EPS);
#pragma unroll
        for (int T = 0; T < 8; ++T) {
            const float y = __builtin_fmaf(c1[T][i] * rstd, p_g1[T], p_be1[T]);
            if (row < nrows && bad_post_relu(y)) st |= COEVO_ST_BAD_FC1;
            const int k = 128 * w + 16 * T + lc;  // this activation is input k of fc2
            sm.h1a[k][row ^ ((k >> 2) & 3)] = relu_keep_nan(y);
        }
    }
    // (the status bits of this phase settled HERE: left alone, the compiler keeps the 32 activations alive to test them at
    // the very end of the body and spills across the fc2 stream to make room)
    asm volatile("" : "+v"(st));
    __syncthreads();
    COEVO_STAMP(2);

    // ---- fc2: 4 column tiles per wave, 128 k-steps; B operands of the four tiles from the lane's own 16-byte piece by
    //      a 4x4 (register x 16-lane row) transpose -------------------------------------------------------------
    f32x4_acc c2[4];
    {
#pragma unroll
        for (int T = 0; T < 4; ++T)
#pragma unroll
            for (int i = 0; i < 4; ++i) c2[T][i] = p_b2[T];
        const float4 *wp = reinterpret_cast<const float4 *>(net + fc_off_w2(D)) + (size_t)w * 128 * 64 + l;
        constexpr int U = COEVO_HEAVY16_U;
        float4 bufA[U], bufB[U];
        auto issue = [&](float4 (&buf)[U], int kq) {
#pragma unroll
            for (int u = 0; u < U; ++u) buf[u] = wp[(size_t)(kq + u) * 64];
        };
        auto consume = [&](const float4 (&buf)[U], int kq) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int q = kq + u;
                const float av = sm.h1a[4 * q + lg][lc ^ (q & 3)];  // A[row lc][k = 4q + lg]
                const u32x2_s s02 = __builtin_amdgcn_permlane32_swap(__float_as_uint(buf[u].x), __float_as_uint(buf[u].z), false, false);
                const u32x2_s s13 = __builtin_amdgcn_permlane32_swap(__float_as_uint(buf[u].y), __float_as_uint(buf[u].w), false, false);
                const u32x2_s y01 = __builtin_amdgcn_permlane16_swap(s02[0], s13[0], false, false);
                const u32x2_s y23 = __builtin_amdgcn_permlane16_swap(s02[1], s13[1], false, false);
                c2[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, __uint_as_float(y01[0]), c2[0], 0, 0, 0);
                c2[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, __uint_as_float(y01[1]), c2[1], 0, 0, 0);
                c2[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, __uint_as_float(y23[0]), c2[2], 0, 0, 0);
                c2[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, __uint_as_float(y23[1]), c2[3], 0, 0, 0);
            }
        };
        issue(bufA, 0);
        int kq = 0;
        for (; kq < 128 - 2 * U; kq += 2 * U) {
            issue(bufB, kq + U);
            __builtin_amdgcn_sched_barrier(0);
            consume(bufA, kq);
            issue(bufA, kq + 2 * U);
            __builtin_amdgcn_sched_barrier(0);
            consume(bufB, kq + U);
        }
        issue(bufB, kq + U);
        __builtin_amdgcn_sched_barrier(0);
        consume(bufA, kq);
        consume(bufB, kq + U);
    }
    COEVO_STAMP(3);
    // ---- LayerNorm(256): canonical block w = this wave's four tiles --------------------------------------------
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float s = (row16_tree_sum(c2[0][i]) + row16_tree_sum(c2[1][i])) +
                        (row16_tree_sum(c2[2][i]) + row16_tree_sum(c2[3][i]));
        if (lc == 0) sm.red[4 * lg + i][w] = s;
    }
    __syncthreads();  // every wave is done with h1a: h2 may overwrite it below
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float *rr = sm.red[4 * lg + i];
        stat[i] = (((rr[0] + rr[1]) + rr[2]) + rr[3]) * (1.0f / H2);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int T = 0; T < 4; ++T) c2[T][i] = c2[T][i] - stat[i];
        const float s = (row16_tree_sum(c2[0][i] * c2[0][i]) + row16_tree_sum(c2[1][i] * c2[1][i])) +
                        (row16_tree_sum(c2[2][i] * c2[2][i]) + row16_tree_sum(c2[3][i] * c2[3][i]));
        if (lc == 0) sm.red[4 * lg + i][w] = s;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 4 * lg + i;
        const float *rr = sm.red[row];
        const float tot = ((rr[0] + rr[1]) + rr[2]) + rr[3];
        const float rstd = 1.0f / __builtin_sqrtf(tot * (1.0f / H2) + LN_EPS);
#pragma unroll
        for (int T = 0; T < 4; ++T) {
            const float y = __builtin_fmaf(c2[T][i] * rstd, p_g2[T], p_be2[T]);
            if (row < nrows && bad_post_relu(y)) st |= COEVO_ST_BAD_FC2;
            sm.h2[row][64 * w + 16 * T + lc] = relu_keep_nan(y);
        }
    }
    asm volatile("" : "+v"(st));
    __syncthreads();
    COEVO_STAMP(4);

    // ---- output layer, argmax, status - as in the other bodies ------------------------------------------------
    if (t < 16 * NACT) {
        const int r = t / NACT, o = t % NACT;
        float y = p_b3;
        const float4 *wr = reinterpret_cast<const float4 *>(&sm.w3s[o][0]);
        const float4 *xr = reinterpret_cast<const float4 *>(&sm.h2[r][0]);
#pragma unroll 8
        for (int k = 0; k < H2 / 4; ++k) {
            const float4 wv = wr[k], xv = xr[k];
            y = __builtin_fmaf(wv.x, xv.x, y);
            y = __builtin_fmaf(wv.y, xv.y, y);
            y = __builtin_fmaf(wv.z, xv.z, y);
            y = __builtin_fmaf(wv.w, xv.w, y);
        }
        sm.logit[r][o] = y;
    }
    __syncthreads();
    COEVO_STAMP(5);
    if (t < nrows) {
        int best = -1;
        float cur = -__builtin_inff();
#pragma unroll
        for (int o = 0; o < NACT; ++o) {
            const float v = sm.logit[t][o];
            if (!__builtin_isfinite(v)) st |= COEVO_ST_BAD_OUT;
            if (v > cur) { cur = v; best = o; }
        }
        if (best < 0) { st |= COEVO_ST_NO_ACTION; best = 0; }
        if constexpr (MODE >= MODE_FUSED) {
            const int row = row0 + t;
            a.act_cur[3 * a.row_game[row] + a.row_slot[row]] = best;  // by (game, slot)
        } else {
            a.actions[row0 + t] = best;
        }
        if (a.logits) {
#pragma unroll
            for (int o = 0; o < NACT; ++o) a.logits[(size_t)(row0 + t) * COEVO_LOGIT_STRIDE + o] = sm.logit[t][o];
        }
    }
    if (st) atomicOr(a.status, st);
    COEVO_STAMP(6);
}

// The lean merged cycle launch: shared-opponent tasks of <= 16 rows (fc_policy_mfma16_body) + one per-individual net per
// streaming workgroup, four workgroups per CU.
template <int R, int MODE = MODE_FUSED>
__global__ __launch_bounds__(256, 4) void fc_cycle16_kernel(FcArgs a)
{
    // (16-byte alignment stated: hipcc otherwise assumes 4 and splits every ds_read_b128 of the k-quad image into two
    // ds_read2_b32 with a vector add each - a third of a vector instruction per MFMA in the stream loop)
    __shared__ __attribute__((aligned(16))) union Cycle16Smem {
        FcMfma16Smem heavy;
#if COEVO_COMPACT
        FcSmemC<R> light;
#else
        FcSmem<R, 1> light;
#endif
    } sm;
    static_assert(sizeof(sm) <= 40960, "four workgroups per CU");
    stamp_begin(a.stamps);
    if ((int)blockIdx.x < a.n_heavy)  // workgroup-uniform
        fc_policy_mfma16_body<MODE>(a, sm.heavy, a.tasks[blockIdx.x]);
    else
#if COEVO_COMPACT
        fc_policy_body_c<R, MODE>(a, sm.light, a.light_tasks, (int)blockIdx.x - a.n_heavy, a.n_light);
#else
        fc_policy_body<R, MODE, 1>(a, sm.light, a.light_tasks, (int)blockIdx.x - a.n_heavy, a.n_light);
#endif
    stamp_end(a.stamps);
}

// The SMALL merged cycle launch: every task (shared-opponent chunk or per-individual net) has <= 8 rows and runs the
// per-individual body with its fc2 on the vector ALU (FC2_DPP).  For launches that leave CUs idle - one rank of a population
// sharded over 4 / 8 GPUs (genetic_algorithm.py:125-217 split by index: 25 or 50 individuals per role), a cohort of the
// host-stepped env - where a workgroup's life, not bytes, is the launch's duration.  <= 256 registers: two workgroups per CU.
template <int R, int MODE = MODE_FUSED>
__global__ __launch_bounds__(256, 2) void fc_cycle_small_kernel(FcArgs a)
{
    __shared__ __attribute__((aligned(16))) FcSmemC<R> sm;
    stamp_begin(a.stamps);
    const bool heavy = (int)blockIdx.x < a.n_heavy;   // workgroup-uniform
    fc_policy_body_c<R, MODE, FC2_DPP>(a, sm, heavy ? a.tasks : a.light_tasks, heavy ? (int)blockIdx.x : (int)blockIdx.x - a.n_heavy,
                                       heavy ? a.n_heavy : a.n_light);
    stamp_end(a.stamps);
}

// ---- The PERSISTENT form of the small launch: a whole rollout (n_cycles env-cycles) in ONE launch ---------------------------
// ("device-side env step fused with the forward into a persistent whole-rollout kernel".)  A launch that the small
// kernel serves has no more workgroups than the device has CUs, so all of them are resident at once (<= 256 registers, < 80 KiB
// of LDS: two fit a CU, i.e. even two such launches side by side are resident together) and a workgroup can keep its task for the
// whole rollout: W1t / the LayerNorm(512) affine / W3 stay in LDS, the small parameters in registers, every row's game (fp64
// state + the reward books) in LDS - no state buffer is read or written between the reset and the last cycle.  What the rows of
// a game owe each other per cycle is three small integers: a row posts its action as ONE 32-bit word (cycle + 1) << 8 | action
// (agent-scope atomic store, double buffered by cycle parity: a row can be at most one cycle ahead of the rows it plays with),
// and the rows of cycle c spin on the three words of their game until all carry the tag c (agent-scope atomic loads; bounded: a
// workgroup that waits too long - a third such launch squeezed its partners off the chip - raises COEVO_ST_SYNC_TIMEOUT and
// the abort word, which every waiter also watches, so the grid drains).  The data IS the flag: nothing else crosses between
// workgroups, so no fence and no cache maintenance is needed.
// The arithmetic is fc_policy_body_c<R, MODE_FUSED, FC2_DPP>'s and mpe_fused_observe's, operation for operation; the last cycle
// leaves the state buffer and the plain action words exactly as the per-cycle launches do, for coevo_mpe_final_step.
template <int R>
struct FcSmemP {
    static constexpr int NG = (R + 3) / 4;
    float par[13 * H1];                      // resident: W1t [D][512], fc1.bias, ln1.weight, ln1.bias
    float h1r[R][H1];                        // fc2's activations, row-major (FC2_DPP)
    float h2[R][260];
    float w3s[NACT][260];                    // resident
    float xs0[4 * NG][COEVO_OBS_STRIDE];
    float red[4][16], red2[4][16];
    float logit[R][COEVO_LOGIT_STRIDE];
    double gs[8][24];                        // per row: its game (MpeGame, 18 doubles) + the books [18..21]
    float p2[3][H2];                         // resident: fc2.bias, ln2.weight, ln2.bias (read where used: registers are short)
    float pb3[8];                            // output.bias
    int rowinfo[8][4];                       // per row: game, env slot, agent-step limit
    int ctl[4];                              // [0]: leave the cycle loop (a wait timed out somewhere)
};

#ifndef COEVO_SYNC_SPINS
#define COEVO_SYNC_SPINS (1 << 21)   // polls of ~0.5-1 us each before a waiting row gives up (seconds, not microseconds)
#endif

struct PersistArgs {
    int n_cycles;
    int32_t *sync;        // [4 + 2 * 3 * n_games]: [0] abort word, tagged actions [parity][game][slot] from [4]; zeroed before the launch
    double *state_alt;    // the second state buffer (buffer 1); a.state is buffer 0 (the reset state)
    int32_t *act_plain;   // actions_by_game [2][n_games][3]: only the last cycle's plain actions are written
    double *rewards;      // [n_games][3] or NULL: each game's owner row closes the books itself (mpe_final_step_kernel's arithmetic)
    coevo_final_pack pack;   // .out != NULL: ... and writes this rank's all-gather record (mpe_final_step_kernel's)
};

__global__ void sync_clear_kernel(int32_t *w, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) w[i] = 0;
}

template <int R>
__global__ __launch_bounds__(256, 2) void fc_rollout_small_kernel(FcArgs a, PersistArgs pa)
{
    __shared__ __attribute__((aligned(16))) FcSmemP<R> sm;
    static_assert(sizeof(FcSmemP<R>) <= 80 * 1024, "two workgroups per CU");
    constexpr int NG = FcSmemP<R>::NG;
    constexpr int US = COEVO_SMALL_U;
    typedef float f32x4_acc __attribute__((ext_vector_type(4)));
    const int t = threadIdx.x, w = t >> 6, l = t & 63;
    const bool heavy = (int)blockIdx.x < a.n_heavy;   // workgroup-uniform
    const coevo_fc_task task = heavy ? a.tasks[blockIdx.x] : a.light_tasks[(int)blockIdx.x - a.n_heavy];
    const int D = task.D, nrows = task.n_rows, row0 = task.row_begin;
    const float *net = a.slab + task.net_off;
    const size_t N = (size_t)a.n_games;
    int st = 0;

    // ---- once: parameters into LDS / registers, every row's game into LDS ---------------------------------------------------
    const int n_pieces = (D + 3) * (H1 / 4);
#pragma unroll
    for (int j = 0; j < 7; ++j)
        if (64 * w + 256 * j < n_pieces)   // wave-uniform
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(net + 4 * (t + 256 * j)),
                                             (__attribute__((address_space(3))) void *)(&sm.par[4 * (64 * w + 256 * j)]),
                                             16, 0, 0);
    {
        const float *W3 = net + fc_off_w3(D);
#pragma unroll
        for (int j = 0; j < 5; ++j) sm.w3s[(t + 256 * j) >> 8][(t + 256 * j) & 255] = W3[t + 256 * j];
    }
    {
        const float *b2p = net + fc_off_b2(D);
        sm.p2[0][t] = b2p[t]; sm.p2[1][t] = b2p[H2 + t]; sm.p2[2][t] = b2p[2 * H2 + t];
        if (t < NACT) sm.pb3[t] = net[fc_off_b3(D) + t];
    }
    const float4 *wps = reinterpret_cast<const float4 *>(net + fc_off_w2(D)) + (size_t)w * 128 * 64 + l;
    float4 sbufA[US], sbufB[US];
    if (w == 0 && l < nrows) {
        const int g = a.row_game[row0 + l];
        sm.rowinfo[l][0] = g;
        sm.rowinfo[l][1] = a.row_slot[row0 + l];
        sm.rowinfo[l][2] = a.game_limit ? a.game_limit[g] : 0x7fffffff;
#pragma unroll
        for (int f = 0; f < 22; ++f) sm.gs[l][f] = a.state[(size_t)f * N + g];
    }
    if (t == 0) sm.ctl[0] = 0;
    int32_t *const tags = pa.sync + 4;

    for (int c = 0; c < pa.n_cycles; ++c) {
        // (an opaque zero in the once-per-cycle global addresses of wave 0: left alone, the compiler computes all of them in
        // front of the loop and spills them across it)
        int zero;
        asm volatile("v_mov_b32 %0, 0" : "=v"(zero));
        // ---- env step + observation: one lane per row (wave 0); cycle c > 0 first waits for the three actions of cycle c - 1 --
        // (the first two buffers of this cycle's fc2 stream are requested by waves 1-3 as soon as they get here - under wave 0's
        // output chain of the previous cycle, its wait and its env step - and by wave 0 behind its env step, whose fp64 state is the
        // kernel's register peak: held across it they spilled)
        auto prefetch = [&]() {
#pragma unroll
            for (int uu = 0; uu < US; ++uu) sbufA[uu] = wps[(size_t)uu * 64];
#pragma unroll
            for (int uu = 0; uu < US; ++uu) sbufB[uu] = wps[(size_t)(US + uu) * 64];
        };
        if (w != 0) {
            prefetch();
        } else {
        if (l < 4 * NG) {
            float o[COEVO_OBS_STRIDE];
#pragma unroll
            for (int k = 0; k < COEVO_OBS_STRIDE; ++k) o[k] = 0.0f;
            if (l < nrows) {
                const int g = sm.rowinfo[l][0], slot = sm.rowinfo[l][1], limit = sm.rowinfo[l][2];
                MpeGame s;
                s.ax = sm.gs[l][0]; s.ay = sm.gs[l][1]; s.bx = sm.gs[l][2]; s.by = sm.gs[l][3]; s.cx = sm.gs[l][4]; s.cy = sm.gs[l][5];
                s.avx = sm.gs[l][6]; s.avy = sm.gs[l][7]; s.bvx = sm.gs[l][8]; s.bvy = sm.gs[l][9]; s.cvx = sm.gs[l][10];
                s.cvy = sm.gs[l][11]; s.l0x = sm.gs[l][12]; s.l0y = sm.gs[l][13]; s.l1x = sm.gs[l][14]; s.l1y = sm.gs[l][15];
                s.gx = sm.gs[l][16]; s.gy = sm.gs[l][17];
                if (c > 0) {
                    const int32_t *tp = tags + (size_t)((c - 1) & 1) * 3 * N + 3 * (size_t)(g + zero);
                    int w0 = 0, w1 = 0, w2 = 0;
                    bool ok = false;
                    for (int it = 0; it < COEVO_SYNC_SPINS; ++it) {
                        w0 = __hip_atomic_load(tp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        w1 = __hip_atomic_load(tp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        w2 = __hip_atomic_load(tp + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ok = (w0 >> 8) == c && (w1 >> 8) == c && (w2 >> 8) == c;
                        if (ok) break;
                        // (the abort word is ONE word for everybody - agent-scope loads are served by memory, and a thousand
                        // lanes on one line queue up: looked at every 16th poll only)
                        if ((it & 15) == 15 && __hip_atomic_load(pa.sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                        __builtin_amdgcn_s_sleep(1);
                    }
                    if (!ok) {   // timed out, or somebody else did: everybody leaves
                        st |= COEVO_ST_SYNC_TIMEOUT;
                        __hip_atomic_store(pa.sync, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        sm.ctl[0] = 1;
                    } else {
                        // mpe_fused_observe's step of cycle c from the state of cycle c - 1 (same operations, same order)
                        const int t0 = 3 * (c - 1);
                        const bool stepped = t0 + 2 < limit;
                        double r_good = 0.0, r_adv = 0.0;
                        if (stepped) {
                            const int a0 = w0 & 0xff, a1 = w1 & 0xff, a2 = w2 & 0xff;
                            if (slot == COEVO_SLOT_ADVERSARY) mpe_world_step(s, a0, a1, a2, a.pos_first, r_good, r_adv);
                            else mpe_world_move(s, a0, a1, a2, a.pos_first);
                        }
                        if (slot == COEVO_SLOT_ADVERSARY) {   // the game's owner row keeps the books
                            double rg_prev = sm.gs[l][18], a_adv = sm.gs[l][19], a_a0 = sm.gs[l][20], a_a1 = sm.gs[l][21];
                            if (t0 < limit) a_adv = a_adv + rg_prev;
                            if (t0 + 1 < limit) a_a0 = a_a0 + rg_prev;
                            if (stepped) { a_a1 = a_a1 + r_adv; rg_prev = r_good; }
                            sm.gs[l][18] = rg_prev; sm.gs[l][19] = a_adv; sm.gs[l][20] = a_a0; sm.gs[l][21] = a_a1;
                        }
                        sm.gs[l][0] = s.ax; sm.gs[l][1] = s.ay; sm.gs[l][2] = s.bx; sm.gs[l][3] = s.by; sm.gs[l][4] = s.cx;
                        sm.gs[l][5] = s.cy; sm.gs[l][6] = s.avx; sm.gs[l][7] = s.avy; sm.gs[l][8] = s.bvx; sm.gs[l][9] = s.bvy;
                        sm.gs[l][10] = s.cvx; sm.gs[l][11] = s.cvy;
                    }
                }
                mpe_obs_from_game(s, slot, o);
#pragma unroll
                for (int k = 0; k < 10; ++k)
                    if (!__builtin_isfinite(o[k])) st |= COEVO_ST_BAD_INPUT;
            }
#pragma unroll
            for (int k = 0; k < COEVO_OBS_STRIDE; ++k) sm.xs0[l][k] = o[k];
        }
        prefetch();
        }
        if (a.stamps && t == 0)
            atomicMin(&a.stamps[2 * (COEVO_STAMP_SLOTS * (size_t)c + blockIdx.x % COEVO_STAMP_SLOTS)],
                      (unsigned long long)__builtin_amdgcn_s_memrealtime());
        __syncthreads();   // (cycle 0: waits for the LDS-DMA too)
        if (sm.ctl[0]) break;   // workgroup-uniform

        // ---- fc1 (v_mfma_f32_4x4x1, rows in groups of four) ---------------------------------------------------------------
        const float *b1s = sm.par + D * H1;
        f32x4_acc c0[NG], c1[NG];
        {
            const float bia = b1s[t], bib = b1s[t + 256];
#pragma unroll
            for (int gq = 0; gq < NG; ++gq)
#pragma unroll
                for (int i = 0; i < 4; ++i) { c0[gq][i] = bia; c1[gq][i] = bib; }
            float4 xk[NG][3];
#pragma unroll
            for (int gq = 0; gq < NG; ++gq)
#pragma unroll
                for (int j = 0; j < 3; ++j) xk[gq][j] = *reinterpret_cast<const float4 *>(&sm.xs0[4 * gq + (l & 3)][4 * j]);
#pragma unroll
            for (int k = 0; k < 10; ++k) {
                if (k < D) {   // wave-uniform
                    const float wa = sm.par[k * H1 + t], wb = sm.par[k * H1 + 256 + t];
#pragma unroll
                    for (int gq = 0; gq < NG; ++gq) {
                        const float4 xv = xk[gq][k >> 2];
                        const float x = (k & 3) == 0 ? xv.x : (k & 3) == 1 ? xv.y : (k & 3) == 2 ? xv.z : xv.w;
                        c0[gq] = __builtin_amdgcn_mfma_f32_4x4x1f32(x, wa, c0[gq], 0, 0, 0);
                        c1[gq] = __builtin_amdgcn_mfma_f32_4x4x1f32(x, wb, c1[gq], 0, 0, 0);
                    }
                }
            }
        }
        const float p_g1a = b1s[H1 + t], p_g1b = b1s[H1 + t + 256], p_be1a = b1s[2 * H1 + t], p_be1b = b1s[2 * H1 + t + 256];
        // ---- LayerNorm(512) + ReLU ------------------------------------------------------------------------------------------
        float v[2 * R];
#pragma unroll
        for (int r = 0; r < R; ++r) { v[2 * r] = c0[r >> 2][r & 3]; v[2 * r + 1] = c1[r >> 2][r & 3]; }
        {
            const float s = packed_totals<2 * R>(v, l);
            if (l < 2 * R) sm.red[w][l] = s;
        }
        __syncthreads();
        const int lr = l < R ? l : R - 1;
        {
            float tot = sm.red[0][2 * lr];
#pragma unroll
            for (int b = 1; b < 8; ++b) tot = tot + sm.red[b & 3][2 * lr + (b >> 2)];
            const float meanv = tot * (1.0f / H1);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float m = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(meanv), r));
                v[2 * r] = v[2 * r] - m;
                v[2 * r + 1] = v[2 * r + 1] - m;
            }
        }
        {
            float sq[2 * R];
#pragma unroll
            for (int j = 0; j < 2 * R; ++j) sq[j] = v[j] * v[j];
            const float s = packed_totals<2 * R>(sq, l);
            if (l < 2 * R) sm.red2[w][l] = s;
        }
        __syncthreads();
        {
            float tot = sm.red2[0][2 * lr];
#pragma unroll
            for (int b = 1; b < 8; ++b) tot = tot + sm.red2[b & 3][2 * lr + (b >> 2)];
            const float rstdv = 1.0f / __builtin_sqrtf(tot * (1.0f / H1) + LN_EPS);
            bool bad = false;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float rstd = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rstdv), r));
                const float y0 = __builtin_fmaf(v[2 * r] * rstd, p_g1a, p_be1a);
                const float y1 = __builtin_fmaf(v[2 * r + 1] * rstd, p_g1b, p_be1b);
                if (r < nrows) bad = bad || bad_post_relu(y0) || bad_post_relu(y1);
                sm.h1r[r][t] = relu_keep_nan(y0);
                sm.h1r[r][t + 256] = relu_keep_nan(y1);
            }
            if (bad) st |= COEVO_ST_BAD_FC1;
        }
        asm volatile("" : "+v"(st));
        __syncthreads();

        // ---- fc2 on the vector ALU (FC2_DPP) ----------------------------------------------------------------------------------
        float u[R];
        {
            const float p_b2 = sm.p2[0][t];
#pragma unroll
            for (int r = 0; r < R; ++r) u[r] = p_b2;
        }
        {
            const float *xrow = &sm.h1r[0][l & 15];
            auto issue_s = [&](float4 (&buf)[US], int kq) {
#pragma unroll
                for (int uu = 0; uu < US; ++uu) buf[uu] = wps[(size_t)(kq + uu) * 64];
            };
            auto consume_s = [&](const float4 (&buf)[US], int kq) {
#pragma unroll
                for (int b = 0; b < US / 4; ++b) {
                    float xv[R];
#pragma unroll
                    for (int r = 0; r < R; ++r) xv[r] = xrow[r * H1 + 4 * (kq + 4 * b)];
                    fmac_block16<R>(u, xv, buf[4 * b], buf[4 * b + 1], buf[4 * b + 2], buf[4 * b + 3]);
                }
            };
#pragma nounroll
            for (int kq = 0; kq < 128; kq += 2 * US) {
                consume_s(sbufA, kq);
                if (kq + 2 * US < 128) issue_s(sbufA, kq + 2 * US);   // wave-uniform
                __builtin_amdgcn_sched_barrier(0);
                consume_s(sbufB, kq + US);
                if (kq + 3 * US < 128) issue_s(sbufB, kq + 3 * US);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- LayerNorm(256) + ReLU ------------------------------------------------------------------------------------------
        {
            const float s = packed_totals<R>(u, l);
            if (l < R) sm.red[w][l] = s;
        }
        __syncthreads();
        {
            const float tot = ((sm.red[0][lr] + sm.red[1][lr]) + sm.red[2][lr]) + sm.red[3][lr];
            const float meanv = tot * (1.0f / H2);
#pragma unroll
            for (int r = 0; r < R; ++r) u[r] = u[r] - __int_as_float(__builtin_amdgcn_readlane(__float_as_int(meanv), r));
            float sq[R];
#pragma unroll
            for (int r = 0; r < R; ++r) sq[r] = u[r] * u[r];
            const float s = packed_totals<R>(sq, l);
            if (l < R) sm.red2[w][l] = s;
        }
        __syncthreads();
        {
            const float tot = ((sm.red2[0][lr] + sm.red2[1][lr]) + sm.red2[2][lr]) + sm.red2[3][lr];
            const float rstdv = 1.0f / __builtin_sqrtf(tot * (1.0f / H2) + LN_EPS);
            const float p_g2 = sm.p2[1][t], p_be2 = sm.p2[2][t];
            bool bad = false;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float rstd = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rstdv), r));
                const float y = __builtin_fmaf(u[r] * rstd, p_g2, p_be2);
                if (r < nrows) bad = bad || bad_post_relu(y);
                sm.h2[r][t] = relu_keep_nan(y);
            }
            if (bad) st |= COEVO_ST_BAD_FC2;
        }
        asm volatile("" : "+v"(st));
        __syncthreads();

        // ---- output layer + first-max action: wave 0; the action is posted as the tagged word -------------------------------
        if (w == 0) {
            if (l < R * NACT) {
                const int r = l / NACT, o = l % NACT;
                float y = sm.pb3[o];
                const float4 *wr = reinterpret_cast<const float4 *>(&sm.w3s[o][0]);
                const float4 *xr = reinterpret_cast<const float4 *>(&sm.h2[r][0]);
#pragma unroll 8
                for (int k = 0; k < H2 / 4; ++k) {
                    const float4 wv = wr[k], xv = xr[k];
                    y = __builtin_fmaf(wv.x, xv.x, y);
                    y = __builtin_fmaf(wv.y, xv.y, y);
                    y = __builtin_fmaf(wv.z, xv.z, y);
                    y = __builtin_fmaf(wv.w, xv.w, y);
                }
                sm.logit[r][o] = y;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (l < nrows) {   // strict '>' scan from -inf
                int best = -1;
                float cur = -__builtin_inff();
#pragma unroll
                for (int o = 0; o < NACT; ++o) {
                    const float vv = sm.logit[l][o];
                    if (!__builtin_isfinite(vv)) st |= COEVO_ST_BAD_OUT;
                    if (vv > cur) { cur = vv; best = o; }
                }
                if (best < 0) { st |= COEVO_ST_NO_ACTION; best = 0; }
                const int g = sm.rowinfo[l][0], slot = sm.rowinfo[l][1];
                __hip_atomic_store(tags + (size_t)(c & 1) * 3 * N + 3 * (size_t)(g + zero) + slot, ((c + 1) << 8) | best, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
                if (c == pa.n_cycles - 1) pa.act_plain[(size_t)(c & 1) * 3 * N + 3 * (size_t)(g + zero) + slot] = best;
                if (a.logits) {
                    int r0 = row0;
                    asm volatile("" : "+s"(r0));   // (not an address to carry through the loop)
#pragma unroll
                    for (int o = 0; o < NACT; ++o) a.logits[(size_t)(r0 + l) * COEVO_LOGIT_STRIDE + o] = sm.logit[l][o];
                }
            }
        }
        if (a.stamps && t == 0)
            atomicMax(&a.stamps[2 * (COEVO_STAMP_SLOTS * (size_t)c + blockIdx.x % COEVO_STAMP_SLOTS) + 1],
                      (unsigned long long)__builtin_amdgcn_s_memrealtime());
    }
    // what the last per-cycle launch leaves for coevo_mpe_final_step: the state of cycle n_cycles - 1 + the books, in that
    // cycle's buffer, written by each game's owner row (cycle 0 writes nothing: its state is the reset state in buffer 0)
    if (w == 0 && l < nrows && sm.rowinfo[l][1] == COEVO_SLOT_ADVERSARY && pa.n_cycles > 1 && !sm.ctl[0]) {
        const int g = sm.rowinfo[l][0];
        double *sn = ((pa.n_cycles - 1) & 1) ? pa.state_alt : const_cast<double *>(a.state);
#pragma unroll
        for (int f = 0; f < 22; ++f) sn[(size_t)f * N + g] = sm.gs[l][f];
    }
    // ... and, asked to, the closing step itself (mpe_final_step_kernel, operation for operation): the owner row waits for the
    // last cycle's three actions, credits the last rewards and writes the game's return triple (+ the all-gather record)
    if (pa.rewards && w == 0 && l < nrows && sm.rowinfo[l][1] == COEVO_SLOT_ADVERSARY && !sm.ctl[0]) {
        const int g = sm.rowinfo[l][0], limit = sm.rowinfo[l][2], cyc = pa.n_cycles - 1;
        const int32_t *tp = tags + (size_t)(cyc & 1) * 3 * N + 3 * (size_t)g;
        int w0 = 0, w1 = 0, w2 = 0;
        bool ok = false;
        for (int it = 0; it < COEVO_SYNC_SPINS; ++it) {
            w0 = __hip_atomic_load(tp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            w1 = __hip_atomic_load(tp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            w2 = __hip_atomic_load(tp + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = (w0 >> 8) == cyc + 1 && (w1 >> 8) == cyc + 1 && (w2 >> 8) == cyc + 1;
            if (ok) break;
            if ((it & 15) == 15 && __hip_atomic_load(pa.sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
            __builtin_amdgcn_s_sleep(1);
        }
        if (!ok) {
            st |= COEVO_ST_SYNC_TIMEOUT;
            __hip_atomic_store(pa.sync, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            double a_adv = sm.gs[l][19], a_a0 = sm.gs[l][20], a_a1 = sm.gs[l][21];
            const int t0 = 3 * cyc;
            const double rg_prev = sm.gs[l][18];
            if (t0 < limit) a_adv = a_adv + rg_prev;
            if (t0 + 1 < limit) a_a0 = a_a0 + rg_prev;
            if (t0 + 2 < limit) {
                MpeGame s;
                s.ax = sm.gs[l][0]; s.ay = sm.gs[l][1]; s.bx = sm.gs[l][2]; s.by = sm.gs[l][3]; s.cx = sm.gs[l][4]; s.cy = sm.gs[l][5];
                s.avx = sm.gs[l][6]; s.avy = sm.gs[l][7]; s.bvx = sm.gs[l][8]; s.bvy = sm.gs[l][9]; s.cvx = sm.gs[l][10];
                s.cvy = sm.gs[l][11]; s.l0x = sm.gs[l][12]; s.l0y = sm.gs[l][13]; s.l1x = sm.gs[l][14]; s.l1y = sm.gs[l][15];
                s.gx = sm.gs[l][16]; s.gy = sm.gs[l][17];
                double r_good, r_adv;
                mpe_world_step(s, w0 & 0xff, w1 & 0xff, w2 & 0xff, a.pos_first, r_good, r_adv);
                a_a1 = a_a1 + r_adv;
            }
            pa.rewards[3 * (size_t)g + 0] = a_a0;
            pa.rewards[3 * (size_t)g + 1] = a_a1;
            pa.rewards[3 * (size_t)g + 2] = a_adv;
            const coevo_final_pack &pk = pa.pack;
            if (pk.out && g < pk.n_roles * pk.n_local * pk.hof && g % pk.hof == pk.hof - 1) {
                const int ij = g / pk.hof, role = ij / pk.n_local, j = ij % pk.n_local;
                double *o = pk.out + 4 * (size_t)ij;
                o[0] = a_a0;
                o[1] = a_a1;
                o[2] = a_adv;
                o[3] = (double)pk.dist[(size_t)role * pk.dist_pitch + pk.dist_first + j];
            }
        }
    }
    if (st) atomicOr(a.status, st);
}

// One launch = one env-cycle of one cohort of games (fused env step): the shared-opponent tasks first (lowest block
// indices: they are dispatched first and are the longer workgroups), then the per-individual tasks.  Both kinds of
// workgroup get the MFMA path's footprint (<= 256 registers, ~73 KiB LDS: two workgroups per CU), so a launch of a
// few hundred workgroups occupies the CUs in waves - together with a second cohort's launch on another stream the
// CUs' weight streams run out of phase and HBM stays busy through the non-streaming phases of any one workgroup.
template <int R, int P>
__global__ __launch_bounds__(256, 2) void fc_cycle_kernel(FcArgs a)
{
    __shared__ __attribute__((aligned(16))) union CycleSmem {
        FcMfmaSmem heavy;
        FcSmem<R, P> light;
    } sm;
    stamp_begin(a.stamps);
    if ((int)blockIdx.x < a.n_heavy)  // workgroup-uniform
        fc_policy_mfma_body<MODE_FUSED>(a, sm.heavy, a.tasks[blockIdx.x]);
    else
        fc_policy_body<R, MODE_FUSED, P>(a, sm.light, a.light_tasks, P * ((int)blockIdx.x - a.n_heavy), a.n_light);
    stamp_end(a.stamps);
}

template <int MODE>
static int launch_fc(const FcArgs &a, int n_tasks, int max_rows, hipStream_t s)
{
    if (n_tasks <= 0) return COEVO_OK;
    if (max_rows <= 1)
        hipLaunchKernelGGL((fc_policy_kernel<1, MODE>), dim3(n_tasks), dim3(256), 0, s, a);
    else if (max_rows <= 2)
        hipLaunchKernelGGL((fc_policy_kernel<2, MODE>), dim3(n_tasks), dim3(256), 0, s, a);
    else if (max_rows <= 5)
        hipLaunchKernelGGL((fc_policy_kernel<5, MODE>), dim3(n_tasks), dim3(256), 0, s, a);
    else if (max_rows <= 8)
        hipLaunchKernelGGL((fc_policy_kernel<8, MODE>), dim3(n_tasks), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((fc_policy_mfma_kernel<MODE>), dim3(n_tasks), dim3(256), 0, s, a);
    COEVO_HIP_CHECK(hipGetLastError());
    return COEVO_OK;
}

}  // namespace coevo

#ifdef COEVO_PHASE_STAMPS
extern "C" int coevo_debug_read_wave_stamps(unsigned long long *host_out, int n_words)
{
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(coevo::g_wave_stamps), sizeof(unsigned long long) * n_words) ==
                   hipSuccess ? 0 : -2;
}

extern "C" int coevo_debug_read_phase_stamps(unsigned long long *host_out, int n_words)
{
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(coevo::g_phase_stamps), sizeof(unsigned long long) * n_words) ==
                   hipSuccess ? 0 : -2;
}
#endif

extern "C" int coevo_fc_forward_argmax(const float *slab, const coevo_fc_task *tasks, int n_tasks,
                                       int max_rows_per_task, const float *obs, int32_t *actions, float *logits,
                                       int32_t *status, void *stream)
{
    if (!slab || !tasks || !obs || !actions || !status || n_tasks < 0) return COEVO_ERR_ARG;
    if (max_rows_per_task < 1 || max_rows_per_task > COEVO_FC_MAX_ROWS) return COEVO_ERR_ARG;
    coevo::FcArgs a{slab, tasks, obs, nullptr, nullptr, nullptr, 0, actions, logits, status, nullptr,
                    nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr, 0, 0};
    return coevo::launch_fc<coevo::MODE_OBS>(a, n_tasks, max_rows_per_task, (hipStream_t)stream);
}

extern "C" int coevo_mpe_policy_cycle_stamped(const float *slab, const coevo_fc_task *tasks, int n_tasks,
                                              int max_rows_per_task, const double *state, int n_games,
                                              const int32_t *row_game, const int32_t *row_slot, int32_t *actions,
                                              int32_t *status, uint64_t *stamps, void *stream);

extern "C" int coevo_mpe_policy_cycle(const float *slab, const coevo_fc_task *tasks, int n_tasks,
                                      int max_rows_per_task, const double *state, int n_games,
                                      const int32_t *row_game, const int32_t *row_slot, int32_t *actions,
                                      int32_t *status, void *stream)
{
    return coevo_mpe_policy_cycle_stamped(slab, tasks, n_tasks, max_rows_per_task, state, n_games, row_game, row_slot,
                                          actions, status, nullptr, stream);
}

extern "C" int coevo_mpe_policy_cycle_stamped(const float *slab, const coevo_fc_task *tasks, int n_tasks,
                                              int max_rows_per_task, const double *state, int n_games,
                                              const int32_t *row_game, const int32_t *row_slot, int32_t *actions,
                                              int32_t *status, uint64_t *stamps, void *stream)
{
    if (!slab || !tasks || !state || !row_game || !row_slot || !actions || !status) return COEVO_ERR_ARG;
    if (n_tasks < 0 || n_games <= 0) return COEVO_ERR_ARG;
    if (max_rows_per_task < 1 || max_rows_per_task > COEVO_FC_MAX_ROWS) return COEVO_ERR_ARG;
    coevo::FcArgs a{slab, tasks, nullptr, state, row_game, row_slot, n_games, actions, nullptr, status,
                    reinterpret_cast<unsigned long long *>(stamps), nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr, 0, 0};
    return coevo::launch_fc<coevo::MODE_STATE>(a, n_tasks, max_rows_per_task, (hipStream_t)stream);
}

extern "C" int coevo_mpe_policy_cycle_fused(const float *slab, const coevo_fc_task *tasks, int n_tasks,
                                            int max_rows_per_task, const double *state_prev, double *state_next,
                                            int n_games, const int32_t *row_game, const int32_t *row_slot,
                                            const int32_t *act_prev, int32_t *act_cur, const int32_t *game_limit,
                                            int cycle, int pos_first, int32_t *status, uint64_t *stamps, void *stream)
{
    if (!slab || !tasks || !state_prev || !state_next || !row_game || !row_slot || !act_prev || !act_cur || !status)
        return COEVO_ERR_ARG;
    if (n_tasks < 0 || n_games <= 0 || cycle < 0 || state_prev == state_next || act_prev == act_cur)
        return COEVO_ERR_ARG;
    if (max_rows_per_task < 1 || max_rows_per_task > COEVO_FC_MAX_ROWS) return COEVO_ERR_ARG;
    coevo::FcArgs a{slab, tasks, nullptr, state_prev, row_game, row_slot, n_games, nullptr, nullptr, status,
                    reinterpret_cast<unsigned long long *>(stamps), state_next, act_prev, act_cur, game_limit, cycle,
                    pos_first, nullptr, 0, 0};
    return coevo::launch_fc<coevo::MODE_FUSED>(a, n_tasks, max_rows_per_task, (hipStream_t)stream);
}

// Which kernel a merged cycle launch of this shape runs (the launcher's own decision, exposed for tests and bench records)
extern "C" int coevo_mpe_cycle_kernel_form(int n_heavy, int n_light, int heavy_max_rows, int light_max_rows,
                                           int concurrent_launches)
{
    if (n_heavy <= 0 || n_light <= 0 || light_max_rows < 1 || light_max_rows > 8) return COEVO_ERR_ARG;
    // workgroup slots of the 32-row kernel: two per CU (<= 256 registers, ~73 KiB LDS)
    static int slots = 0;
    if (slots == 0) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            return COEVO_ERR_HIP;
        slots = 2 * cus;
    }
    const int conc = concurrent_launches > 1 ? concurrent_launches : 1;
    const int wgs = (n_heavy + n_light) * conc;
    // every task of <= 8 rows and no more workgroups in flight than CUs: the small-launch kernel (COEVO_SMALL_KERNEL=0: A/B).
    // (Two per CU share their SIMDs' vector ALU: a rank of 4 - 360 workgroups of <= 8 rows - 27.6-30.5 us per launch against
    // 21.8 for the lean kernel's 270 with 16-row matrix-core tiles; a rank of 8 - 231 - 16.4 against 21.0.)
    static const bool small_ok = !(getenv("COEVO_SMALL_KERNEL") && getenv("COEVO_SMALL_KERNEL")[0] == '0');
    if (small_ok && heavy_max_rows >= 1 && heavy_max_rows <= 8 && 2 * wgs <= slots) return COEVO_CYCLE_FORM_SMALL;
    if (heavy_max_rows >= 1 && heavy_max_rows <= 16 && wgs <= 2 * slots) return COEVO_CYCLE_FORM_LEAN16;
    // If one net per streaming workgroup does not fit the slots in a single round, pair the nets: the second round would
    // otherwise wait for the slots of the (long) shared-opponent workgroups.
    return wgs > slots ? COEVO_CYCLE_FORM_TILE32_PAIRED : COEVO_CYCLE_FORM_TILE32;
}

// The persistent launch needs every workgroup resident at once: <= 256 registers and < 80 KiB of LDS, so two fit a CU.  (With
// more than one per CU the rollout is bound by what ONE CU streams - ~47 GB/s, tools/stream_waves_probe.hip: two 0.56 MB nets
// per cycle = 20 us - and still ahead of the per-cycle launches of the lean kernel: a rank of 4, 450 workgroups, 20.2 against
// 23.3 us per cycle.)
extern "C" int coevo_mpe_persistent_fits(int n_heavy, int n_light, int heavy_max_rows, int light_max_rows, int concurrent_launches)
{
    // (either list may be empty: the ten evaluation games of Co-ES are six per-individual tasks and nothing else)
    if (n_heavy < 0 || n_light < 0 || n_heavy + n_light <= 0) return COEVO_ERR_ARG;
    if ((n_light > 0 && (light_max_rows < 1 || light_max_rows > 8)) || (n_heavy > 0 && heavy_max_rows < 1)) return COEVO_ERR_ARG;
    if (n_heavy > 0 && heavy_max_rows > 8) return 0;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
        return COEVO_ERR_HIP;
    // what the runtime says the instantiation's residency is (two per CU as built; asked, not assumed: a launch that is not
    // all resident would wait for itself)
    const int hr = n_heavy > 0 ? heavy_max_rows : 0, lr = n_light > 0 ? light_max_rows : 0, rows = hr > lr ? hr : lr;
    static int per_cu[4] = {-1, -1, -1, -1};
    const int slot = rows <= 1 ? 0 : rows <= 2 ? 1 : rows <= 5 ? 2 : 3;
    if (per_cu[slot] < 0) {
        int n = 0;
        const hipError_t e =
            slot == 0   ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, coevo::fc_rollout_small_kernel<1>, 256, 0)
            : slot == 1 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, coevo::fc_rollout_small_kernel<2>, 256, 0)
            : slot == 2 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, coevo::fc_rollout_small_kernel<5>, 256, 0)
                        : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, coevo::fc_rollout_small_kernel<8>, 256, 0);
        if (e != hipSuccess) return COEVO_ERR_HIP;
        per_cu[slot] = n > 2 ? 2 : n;   // (never more than the two the kernel was sized for)
    }
    const int conc = concurrent_launches > 1 ? concurrent_launches : 1;
    return (n_heavy + n_light) * conc <= per_cu[slot] * cus ? 1 : 0;
}

extern "C" int coevo_mpe_persistent_sync_words(int n_games) { return n_games > 0 ? 4 + 6 * n_games : COEVO_ERR_ARG; }

extern "C" int coevo_mpe_rollout_persistent(const float *slab, const coevo_fc_task *heavy_tasks, int n_heavy,
                                            const coevo_fc_task *light_tasks, int n_light, int light_max_rows, int heavy_max_rows,
                                            double *state, double *state_alt, int n_games, const int32_t *row_game,
                                            const int32_t *row_slot, int32_t *actions_by_game, const int32_t *game_limit,
                                            int n_cycles, int pos_first, int32_t *status, uint64_t *stamps, int32_t *sync_words,
                                            int concurrent_launches, double *rewards, const coevo_final_pack *pack,
                                            int sync_cleared, void *stream)
{
    if (!slab || (n_heavy > 0 && !heavy_tasks) || (n_light > 0 && !light_tasks) || !state || !state_alt || state == state_alt ||
        !row_game || !row_slot || !actions_by_game || !status || !sync_words || n_games <= 0 || n_cycles < 1 || n_cycles > (1 << 22))
        return COEVO_ERR_ARG;
    const int fits = coevo_mpe_persistent_fits(n_heavy, n_light, heavy_max_rows, light_max_rows, concurrent_launches);
    if (fits < 0) return fits;
    if (!fits) return COEVO_ERR_UNSUPPORTED;   // not all resident at once: the per-cycle launches
    hipStream_t s = (hipStream_t)stream;
    // (a kernel, not hipMemsetAsync: captured into a hipGraph and replayed after other work had run, the memset node of this
    // runtime left device pointers in the buffer - the Co-ES evaluation rollout timed out on them)
    if (!sync_cleared) {
        hipLaunchKernelGGL(coevo::sync_clear_kernel, dim3((4 + 6 * n_games + 255) / 256), dim3(256), 0, s, sync_words, 4 + 6 * n_games);
        COEVO_HIP_CHECK(hipGetLastError());
    }
    coevo::FcArgs a{};
    a.slab = slab; a.tasks = heavy_tasks; a.state = state; a.row_game = row_game; a.row_slot = row_slot; a.n_games = n_games;
    a.status = status; a.stamps = reinterpret_cast<unsigned long long *>(stamps); a.game_limit = game_limit;
    a.pos_first = pos_first; a.light_tasks = light_tasks; a.n_heavy = n_heavy; a.n_light = n_light;
    if (pack && (!rewards || !pack->out || !pack->dist || pack->n_roles < 1 || pack->n_local < 1 || pack->hof < 1 ||
                 (int64_t)pack->n_roles * pack->n_local * pack->hof > n_games || pack->dist_first < 0 ||
                 pack->dist_first + pack->n_local > pack->dist_pitch))
        return COEVO_ERR_ARG;
    const coevo::PersistArgs pa{n_cycles, sync_words, state_alt, actions_by_game, rewards, pack ? *pack : coevo_final_pack{}};
    const int hr = n_heavy > 0 ? heavy_max_rows : 0, lr = n_light > 0 ? light_max_rows : 0;
    const int rows = hr > lr ? hr : lr;
    const dim3 grid(n_heavy + n_light), block(256);
    if (rows <= 1) hipLaunchKernelGGL((coevo::fc_rollout_small_kernel<1>), grid, block, 0, s, a, pa);
    else if (rows <= 2) hipLaunchKernelGGL((coevo::fc_rollout_small_kernel<2>), grid, block, 0, s, a, pa);
    else if (rows <= 5) hipLaunchKernelGGL((coevo::fc_rollout_small_kernel<5>), grid, block, 0, s, a, pa);
    else hipLaunchKernelGGL((coevo::fc_rollout_small_kernel<8>), grid, block, 0, s, a, pa);
    COEVO_HIP_CHECK(hipGetLastError());
    return COEVO_OK;
}

extern "C" int coevo_mpe_policy_cycle_merged(const float *slab, const coevo_fc_task *heavy_tasks, int n_heavy,
                                             const coevo_fc_task *light_tasks, int n_light, int light_max_rows,
                                             const double *state_prev, double *state_next, int n_games,
                                             const int32_t *row_game, const int32_t *row_slot, const int32_t *act_prev,
                                             int32_t *act_cur, const int32_t *game_limit, int cycle, int pos_first,
                                             int32_t *status, uint64_t *stamps, int concurrent_launches,
                                             int heavy_max_rows, void *stream)
{
    if (!slab || !heavy_tasks || !light_tasks || !state_prev || !state_next || !row_game || !row_slot || !act_prev ||
        !act_cur || !status)
        return COEVO_ERR_ARG;
    if (n_heavy <= 0 || n_light <= 0 || n_games <= 0 || cycle < 0 || state_prev == state_next || act_prev == act_cur)
        return COEVO_ERR_ARG;
    if (light_max_rows < 1 || light_max_rows > 8) return COEVO_ERR_ARG;
    coevo::FcArgs a{slab, heavy_tasks, nullptr, state_prev, row_game, row_slot, n_games, nullptr, nullptr, status,
                    reinterpret_cast<unsigned long long *>(stamps), state_next, act_prev, act_cur, game_limit, cycle,
                    pos_first, light_tasks, n_heavy, n_light};
    const int form = coevo_mpe_cycle_kernel_form(n_heavy, n_light, heavy_max_rows, light_max_rows, concurrent_launches);
    if (form < 0) return form;
    hipStream_t s = (hipStream_t)stream;
    if (form == COEVO_CYCLE_FORM_SMALL) {
        const int rows = heavy_max_rows > light_max_rows ? heavy_max_rows : light_max_rows;
        const dim3 grid_s(n_heavy + n_light), block_s(256);
        if (rows <= 1) hipLaunchKernelGGL((coevo::fc_cycle_small_kernel<1>), grid_s, block_s, 0, s, a);
        else if (rows <= 2) hipLaunchKernelGGL((coevo::fc_cycle_small_kernel<2>), grid_s, block_s, 0, s, a);
        else if (rows <= 5) hipLaunchKernelGGL((coevo::fc_cycle_small_kernel<5>), grid_s, block_s, 0, s, a);
        else hipLaunchKernelGGL((coevo::fc_cycle_small_kernel<8>), grid_s, block_s, 0, s, a);
        COEVO_HIP_CHECK(hipGetLastError());
        return COEVO_OK;
    }
    // (when not everything fits - Co-ES with 3000 nets - the 32-row tiles with two nets per streaming workgroup are
    // ahead: 109 vs 106 generations/s)
    if (form == COEVO_CYCLE_FORM_LEAN16) {
        // the lean kernel: four workgroups per CU hold everything at one net per streaming workgroup
        const dim3 grid16(n_heavy + n_light), block16(256);
        if (light_max_rows <= 1) hipLaunchKernelGGL((coevo::fc_cycle16_kernel<1>), grid16, block16, 0, s, a);
        else if (light_max_rows <= 2) hipLaunchKernelGGL((coevo::fc_cycle16_kernel<2>), grid16, block16, 0, s, a);
        else if (light_max_rows <= 5) hipLaunchKernelGGL((coevo::fc_cycle16_kernel<5>), grid16, block16, 0, s, a);
        else hipLaunchKernelGGL((coevo::fc_cycle16_kernel<8>), grid16, block16, 0, s, a);
        COEVO_HIP_CHECK(hipGetLastError());
        return COEVO_OK;
    }
    const bool pair = form == COEVO_CYCLE_FORM_TILE32_PAIRED;
    const dim3 grid(n_heavy + (pair ? (n_light + 1) / 2 : n_light)), block(256);
#define COEVO_LAUNCH_CYCLE(RR)                                                                    \
    do {                                                                                          \
        if (pair) hipLaunchKernelGGL((coevo::fc_cycle_kernel<RR, 2>), grid, block, 0, s, a);      \
        else hipLaunchKernelGGL((coevo::fc_cycle_kernel<RR, 1>), grid, block, 0, s, a);           \
    } while (0)
    if (light_max_rows <= 1) COEVO_LAUNCH_CYCLE(1);
    else if (light_max_rows <= 2) COEVO_LAUNCH_CYCLE(2);
    else if (light_max_rows <= 5) COEVO_LAUNCH_CYCLE(5);
    else COEVO_LAUNCH_CYCLE(8);
#undef COEVO_LAUNCH_CYCLE
    COEVO_HIP_CHECK(hipGetLastError());
    return COEVO_OK;
}

// The merged launch with the observations GIVEN (the env stepped elsewhere - on the host cores, env_mode "host"): shared-opponent
// tasks (<= 16 rows) and per-individual tasks (<= 8 rows) of one env-cycle in one launch of the lean kernel, instead of one
// coevo_fc_forward_argmax per task class.  Needs every workgroup resident at four per CU; COEVO_ERR_ARG otherwise (the caller
// then launches the classes one after the other).
extern "C" int coevo_fc_forward_merged(const float *slab, const coevo_fc_task *heavy_tasks, int n_heavy, int heavy_max_rows,
                                       const coevo_fc_task *light_tasks, int n_light, int light_max_rows, const float *obs,
                                       int32_t *actions, float *logits, int32_t *status, void *stream)
{
    if (!slab || !heavy_tasks || !light_tasks || !obs || !actions || !status) return COEVO_ERR_ARG;
    if (n_heavy <= 0 || n_light <= 0 || heavy_max_rows < 1 || heavy_max_rows > 16 || light_max_rows < 1 || light_max_rows > 8)
        return COEVO_ERR_ARG;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
        return COEVO_ERR_HIP;
    if (n_heavy + n_light > 4 * cus) return COEVO_ERR_ARG;
    coevo::FcArgs a{slab, heavy_tasks, obs, nullptr, nullptr, nullptr, 0, actions, logits, status, nullptr, nullptr, nullptr,
                    nullptr, nullptr, 0, 0, light_tasks, n_heavy, n_light};
    const dim3 grid(n_heavy + n_light), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (light_max_rows <= 1) hipLaunchKernelGGL((coevo::fc_cycle16_kernel<1, coevo::MODE_OBS>), grid, block, 0, s, a);
    else if (light_max_rows <= 2) hipLaunchKernelGGL((coevo::fc_cycle16_kernel<2, coevo::MODE_OBS>), grid, block, 0, s, a);
    else if (light_max_rows <= 5) hipLaunchKernelGGL((coevo::fc_cycle16_kernel<5, coevo::MODE_OBS>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((coevo::fc_cycle16_kernel<8, coevo::MODE_OBS>), grid, block, 0, s, a);
    COEVO_HIP_CHECK(hipGetLastError());
    return COEVO_OK;
}

COEVO_DEFINE_TU_FLAGS(fc_forward)
