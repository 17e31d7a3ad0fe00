/*
 * coevo.h - C ABI of libcoevo.so, the MI355X (gfx950) population-evaluation engine.
 *
 * The reference (CogSP/CoEvoNet) is pure Python and has no FFI; its boundary for this hot path is the Python
 * surface initialize_env / create_agent / play_game / Agent / FCNetwork / DeepQN.  This header is
 * what a binding for that surface binds instead of torch-CPU ops: every entry point names the reference code whose
 * arithmetic it replaces (file:line under the reference checkout).  INTEGRATION.md shows the ctypes stub.
 *
 * Conventions
 *   - plain C types only; every pointer is a DEVICE pointer owned by the caller unless named host_*;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); calls are asynchronous on it, allocate
 *     nothing, keep no global state and are re-entrant per stream (graph-capturable);
 *   - return value: COEVO_OK or a negative COEVO_ERR_* (argument errors are detected on the host before launch);
 *   - numerical faults the reference raises ValueError for (NaN/inf in the forward, no action) are OR-ed into a
 *     caller-provided device status word as COEVO_ST_* bits; the host facade turns them into ValueError;
 *   - fp32 arithmetic follows the canonical order documented in DESIGN.md (sequential-k fmaf chains, fixed
 *     reduction trees), fp64 for the MPE physics and rewards.
 */
#ifndef COEVO_H
#define COEVO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define COEVO_VERSION 103   /* 103: coevo_mpe_rollout_persistent, coevo_rollout_desc.sync_words / .pack (192 bytes), coevo_final_pack; 102: host-cores placement (coevo_host_placement_choose, coevo_host_rollout_placement / _alloc),
                             * wide small-shard cycle kernel; 101: offspring noise = Philox4x32-7 (100: -10; other numbers for the same seed), host-cores
                             * rollout entry points, coevo_noise_rounds */

#define COEVO_OK 0
#define COEVO_ERR_ARG (-1)   /* bad size / null pointer / unsupported shape */
#define COEVO_ERR_HIP (-2)   /* a HIP runtime call failed (hipGetLastError has the detail) */
#define COEVO_ERR_UNSUPPORTED (-3)  /* valid arguments this entry point cannot serve: use the general entry point instead */

/* device status word bits (MPE/fcnetwork.py:39,49,57,65,87; utils/game_logic_functions.py:172-177) */
#define COEVO_ST_BAD_INPUT 1
#define COEVO_ST_BAD_FC1 2
#define COEVO_ST_BAD_FC2 4
#define COEVO_ST_BAD_OUT 8
#define COEVO_ST_NO_ACTION 16
#define COEVO_ST_SYNC_TIMEOUT 32 /* coevo_mpe_rollout_persistent: a workgroup waited too long for the other rows of its games (no
                                  * reference counterpart: more such launches side by side than the device holds at once) */

/* FCNetwork geometry (MPE/fcnetwork.py:14-22) */
#define COEVO_FC_H1 512
#define COEVO_FC_H2 256
#define COEVO_FC_NACT 5
#define COEVO_OBS_STRIDE 12      /* floats per observation row in obs buffers (D = 8 or 10, zero padded) */
#define COEVO_LOGIT_STRIDE 8     /* floats per logits row */
#define COEVO_FC_MAX_ROWS 32     /* rows (observations) one task may carry */
#define COEVO_STAMP_SLOTS 32     /* clock-stamp slot pairs per timed launch (see coevo_mpe_policy_cycle_stamped) */

/* env slots, PettingZoo agent order of simple_adversary_v3 */
#define COEVO_SLOT_ADVERSARY 0
#define COEVO_SLOT_AGENT_0 1
#define COEVO_SLOT_AGENT_1 2

int coevo_version(void);
/* "" for the shipped configuration; otherwise the non-default build switches (A/B measurement builds), per translation unit.
 * A binding should refuse a library whose answer is not empty: some switches change results. */
const char *coevo_build_flags(void);

/* ---------------------------------------------------------------- weight slab ------------------------------ */
/* Canonical flat order = torch parameters() order of FCNetwork (fc1.w, fc1.b, ln1.w, ln1.b, fc2.w, fc2.b, ln2.w,
 * ln2.b, output.w, output.b): what Agent.mutate walks (agent.py:27) and state_dict() holds.  The device slab
 * re-tiles fc1.w / fc2.w for coalesced streaming (DESIGN.md "HBM layout"). */
int64_t coevo_fc_param_count(int D);   /* 139781 (D=10) / 138757 (D=8) */
int64_t coevo_fc_slab_stride(int D);   /* floats between consecutive nets in a slab */
/* flat[n][P] <-> slab[n][stride]; replaces state_dict()/load_state_dict() copies (MPE/mpe_agent.py:24-28) */
int coevo_fc_pack(const float *flat, float *slab, int n, int D, void *stream);
int coevo_fc_unpack(const float *slab, float *flat, int n, int D, void *stream);

/* ---------------------------------------------------------------- K1: policy step -------------------------- */
/* One task = one weight set applied to n_rows observations (rows row_begin .. row_begin+n_rows-1).  The rows of
 * one task are the env copies that share that net in this env-cycle (an individual's HoF games; a HoF member's
 * games against the whole population). */
typedef struct {
    int64_t net_off;   /* float offset of the net inside the slab */
    int32_t row_begin;
    int32_t n_rows;    /* 1 .. COEVO_FC_MAX_ROWS */
    int32_t D;         /* 8 or 10 */
    int32_t reserved;
} coevo_fc_task;

/* FCNetwork.forward + determine_action (MPE/fcnetwork.py:37-90) for every row of every task, one launch.
 *   obs      [rows][COEVO_OBS_STRIDE] fp32          actions [rows] int32 (first index of the maximum logit)
 *   logits   [rows][COEVO_LOGIT_STRIDE] or NULL     status  one int32, COEVO_ST_* bits OR-ed in
 * max_rows_per_task: upper bound of n_rows over the tasks (selects the 8- or 32-row kernel). */
int coevo_fc_forward_argmax(const float *slab, const coevo_fc_task *tasks, int n_tasks, int max_rows_per_task,
                            const float *obs, int32_t *actions, float *logits, int32_t *status, void *stream);
/* The same forward (MPE/fcnetwork.py:37-90) for TWO task tables in ONE launch of the lean cycle kernel: `heavy` = nets that
 * act in many games (tasks of <= 16 rows, matrix-core body), `light` = one task of <= 8 rows per individual (weight streaming
 * body).  What a host-stepped env (env_mode "host") launches per env-cycle.  Every workgroup must be resident at four per CU:
 * COEVO_ERR_ARG if n_heavy + n_light exceeds that (launch the two tables with coevo_fc_forward_argmax then). */
int coevo_fc_forward_merged(const float *slab, const coevo_fc_task *heavy_tasks, int n_heavy, int heavy_max_rows,
                            const coevo_fc_task *light_tasks, int n_light, int light_max_rows, const float *obs,
                            int32_t *actions, float *logits, int32_t *status, void *stream);

/* ---------------------------------------------------------------- device-side MPE simple_adversary ---------- */
/* Replaces env.reset/observe/step/last of play_MPE (utils/game_logic_functions.py:123-212, :217) for E env
 * copies at once.  State is struct-of-arrays, fp64, E-strided: see coevo_mpe_state_doubles(). */
#define COEVO_MPE_STATE_DOUBLES 24   /* per game: ppos[6] pvel[6] lm[4] goal_pos[2] rg_prev acc[3] spare[2] */
typedef struct {
    uint64_t pcg_state_hi, pcg_state_lo, pcg_inc_hi, pcg_inc_lo;  /* numpy PCG64(seed).state */
} coevo_pcg64;

/* state = [COEVO_MPE_STATE_DOUBLES][n_games] fp64.  Games game_first .. game_first+count-1 take reset ordinals
 * first_ordinal .. of the single seeded stream (quirk Q6; ordinal 0 is the reset inside initialize_env,
 * utils/game_logic_functions.py:54); each game jumps straight to its ordinal, so shards need no common prefix. */
int coevo_mpe_reset(double *state, int n_games, int game_first, int count, coevo_pcg64 rng, uint64_t first_ordinal,
                    void *stream);
/* up to COEVO_MAX_JOBS (game range, first ordinal) segments in one launch (a cohort's three role phases + the evaluation
 * games); same results as one coevo_mpe_reset per segment */
typedef struct { int32_t game_first, count; uint64_t first_ordinal; } coevo_reset_seg;
int coevo_mpe_reset_multi(double *state, int n_games, const coevo_reset_seg *segs /* host */, int n_segs,
                          coevo_pcg64 rng, void *stream);
/* ... that also re-arms n_stamps clock-stamp slot pairs ({UINT64_MAX, 0}) for the timed rollout it precedes
 * (coevo_rollout_desc.light_stamps with stamps_armed = 1) */
int coevo_mpe_reset_multi_arm(double *state, int n_games, const coevo_reset_seg *segs /* host */, int n_segs,
                              coevo_pcg64 rng, uint64_t *stamps, int n_stamps, void *stream);
/* ... and / or zeroes n_zero 32-bit words (the sync words of the persistent rollout it precedes: coevo_rollout_desc.sync_cleared
 * = 1); stamps and zero_words may each be NULL */
int coevo_mpe_reset_multi_prep(double *state, int n_games, const coevo_reset_seg *segs /* host */, int n_segs,
                               coevo_pcg64 rng, uint64_t *stamps, int n_stamps, int32_t *zero_words, int n_zero, void *stream);
/* the same with the generation taken from a device counter: first ordinal = first_ordinal + (*gen_dev) *
 * ordinals_per_gen (clamped at 0), so a captured hipGraph of a whole generation can be replayed unchanged */
int coevo_mpe_reset_gen(double *state, int n_games, int game_first, int count, coevo_pcg64 rng, int64_t first_ordinal,
                        const int32_t *gen_dev, int64_t ordinals_per_gen, void *stream);
/* obs rows for (game, slot) pairs: row r observes game row_game[r] as slot row_slot[r] (float32 casts of fp64
 * differences, PettingZoo SimpleEnv.observe). */
int coevo_mpe_observe(const double *state, int n_games, const int32_t *row_game, const int32_t *row_slot,
                      int n_rows, float *obs, void *stream);
/* One world cycle for every game: the three acting rows' actions are read through game_rows[g][3] (row index per
 * slot), physics advanced, rewards credited with the reference's attribution (quirk Q1) honouring the per-game
 * agent-step limit: cycle c credits adversary_0/agent_0 with the good reward of world step c and agent_1 with the
 * adversary reward of world step c+1, each only while 3c+slot < limit.  pos_first = PettingZoo >= 1.24 order. */
int coevo_mpe_step(double *state, int n_games, const int32_t *game_rows, const int32_t *actions, int cycle,
                   const int32_t *game_limit, int pos_first, void *stream);
/* coevo_mpe_observe / coevo_mpe_step on the HOST cores (every pointer is host memory, the call returns when the work is
 * done; same state layout, same arithmetic, the kernels' own bodies): the env of north_star's first configuration,
 * "vectorised env stepping runs on the host cores" - what play_MPE drives through env.observe / env.step / env.last
 * (utils/game_logic_functions.py:138,179-190), for all of a rank's games at once.  n_rows = length of `actions`. */
int coevo_mpe_host_reset(double *state, int n_games, coevo_pcg64 rng, const int64_t *ordinals /* [n_games] reset ordinal of
                         every game: play_game's env.reset(), utils/game_logic_functions.py:217, quirk Q6 */);
int coevo_mpe_host_reset_games(double *state, int n_games, coevo_pcg64 rng, const int64_t *ordinals /* [n_games] */,
                               const int32_t *games, int lo, int hi /* only games[lo..hi) are reset */);
int coevo_mpe_host_observe(const double *state, int n_games, const int32_t *row_game, const int32_t *row_slot,
                           int n_rows, float *obs);
int coevo_mpe_host_step(double *state, int n_games, const int32_t *game_rows, const int32_t *actions, int n_rows,
                        int cycle, const int32_t *game_limit, int pos_first);
/* world step `cycle` (cycle < 0: none) of games[lo..hi) on the calling thread, then (observe != 0) the observation rows of
 * those games: the unit of work one host core gets per env-cycle.  No table scan - the caller has validated game_rows. */
int coevo_mpe_host_step_games(double *state, int n_games, const int32_t *game_rows, const int32_t *actions, int cycle,
                              const int32_t *game_limit, int pos_first, const int32_t *games, int lo, int hi, int observe,
                              float *obs);

/* A whole batch of games with the env on the host cores in ONE call (north_star: "vectorised env stepping runs on the host
 * cores"): the per-game loop of play_MPE (utils/game_logic_functions.py:123-212) for all of a rank's games.  The games are cut
 * into cohorts; per env-cycle and cohort the host cores (the context's n_threads, the caller's thread included) step the
 * cohort's games and write their observations, the cohort's own stream carries obs up, one policy launch
 * (MPE/fcnetwork.py:37-90 for every row) and the actions down; while that is in flight the cores work on the next cohort.
 * Returns when the last cycle has been stepped; results are independent of n_threads and of the cohort partition. */
typedef struct {
    const coevo_fc_task *heavy;   /* DEVICE: this cohort's shared-opponent tasks (may be NULL / 0) */
    const coevo_fc_task *light;   /* DEVICE: this cohort's per-individual tasks */
    const int32_t *games;         /* HOST: ids of this cohort's games */
    int32_t n_heavy, heavy_max_rows, n_light, light_max_rows;
    int32_t n_games;
    int32_t row_first, n_rows;    /* the cohort's rows: one contiguous range of obs / actions; n_rows = 3 * n_games */
    int32_t reserved;
} coevo_host_cohort;
typedef struct {
    const float *slab;            /* DEVICE weight slab */
    double *state;                /* HOST [COEVO_MPE_STATE_DOUBLES][n_games], as coevo_mpe_host_reset leaves it */
    const int32_t *game_rows;     /* HOST [n_games][3]: row of each slot */
    const int32_t *game_limit;    /* HOST [n_games] agent-step limits, or NULL */
    float *obs_host;              /* HOST, page-locked: [n_rows][COEVO_OBS_STRIDE] */
    float *obs_dev;               /* DEVICE twin */
    int32_t *actions_host;        /* HOST, page-locked: [n_rows] */
    int32_t *actions_dev;         /* DEVICE twin */
    int32_t *status;              /* DEVICE status word */
    const coevo_host_cohort *cohorts;
    double *phase_us;             /* NULL, or HOST [6]: mean microseconds per cohort-cycle of {host wait for the actions, host
                                     env step + observe, host enqueue, obs host->device, policy launch, actions device->host} */
    int32_t n_games, n_rows, n_cycles, n_cohorts, pos_first;
    int32_t zero_copy;            /* != 0: the launch reads obs_host / writes actions_host directly (page-locked, mapped
                                     memory: same PCIe bytes, no copy engine in the chain) */
    const int64_t *reset_ordinals;   /* NULL: `state` holds the games' reset states already; else HOST [n_games]: every game is
                                        reset to this ordinal of the seeded stream first (play_game's env.reset(),
                                        utils/game_logic_functions.py:217) - by the core that drives its cohort, inside the
                                        rollout, so that only the first cohort's resets precede the first launch */
    coevo_pcg64 reset_rng;           /* the stream's seed state (used with reset_ordinals) */
} coevo_host_rollout_desc;
void *coevo_host_rollout_create(int n_threads, int n_cohorts);   /* worker threads + one stream per cohort */
void coevo_host_rollout_destroy(void *ctx);
int coevo_host_rollout_threads(void *ctx);                       /* host cores a rollout uses (caller's thread included) */
int coevo_mpe_host_rollout(void *ctx, const coevo_host_rollout_desc *desc, void *stream);
/* WHICH host cores (the reference steps its env on whatever core runs the interpreter, utils/game_logic_functions.py:138,
 * 179-190; on a two-socket GPU host that choice is a factor of four in the env step).  A context runs its cores in one L3
 * complex of the GPU's NUMA node, as far as the caller's affinity mask allows: the caller's thread is pinned to cpus[0] for
 * the duration of a rollout (its mask is put back at return), worker i to cpus[i].  COEVO_HOST_PIN=0: no pinning; =far: a
 * node that is NOT the GPU's (to reproduce the slow placement on purpose).
 * coevo_host_placement_choose is the choice as a pure function of sysfs-style strings (cpulist syntax "0-7,128-135"; groups
 * separated by ';'): -> number of CPUs written to cpus_out (<= n_threads; 0: nothing to pin to), COEVO_PLACE_* in *flags_out.
 * allowed = the affinity mask; node_cpus = the wanted node's CPUs ("" = unknown: every allowed CPU is a candidate);
 * l3_groups / smt_groups = the L3 complexes / SMT sibling sets; caller_cpu = where the caller runs now (its own complex is
 * preferred, and it keeps its CPU); ctx_index moves on by whole complexes (ranks / contexts side by side). */
#define COEVO_PLACE_ON_NODE 1       /* the CPUs are on the wanted node */
#define COEVO_PLACE_ONE_L3 2        /* ... all in one L3 complex */
#define COEVO_PLACE_NODE_UNKNOWN 4  /* no node given */
#define COEVO_PLACE_SHORT 8         /* fewer CPUs than threads: the last threads stay unpinned */
int coevo_host_placement_choose(const char *allowed, const char *node_cpus, const char *l3_groups, const char *smt_groups,
                                int caller_cpu, int n_threads, int ctx_index, int32_t *cpus_out, int32_t *flags_out);
/* what a context chose (made now if it has not run yet; needs the current device): cpus_out[0..n), info_out[0..4) = {NUMA node
 * of the GPU or -1, node of the chosen CPUs, COEVO_PLACE_* flags, pinned 0/1}; returns n (0: not pinned) */
int coevo_host_rollout_placement(void *ctx, int32_t *cpus_out, int max_cpus, int32_t *info_out);
/* page-locked, device-mapped host memory first touched on the context's node (for obs_host / actions_host / frames_host);
 * zero-filled; owned by the context (freed by coevo_host_rollout_destroy); NULL on failure */
void *coevo_host_rollout_alloc(void *ctx, size_t bytes);
/* tests only: leave the context's completion / gate sequence numbers where `value` earlier cycles would have left them */
int coevo_host_rollout_debug_seed_counters(void *ctx, uint32_t value);
/* one cohort-cycle's host share alone, on the context's cores: world step `cycle` (< 0: none) of the n_list listed games,
 * then (observe != 0) their observation rows.  Needs no GPU.  n_rows = length of `actions` / rows of `obs`. */
int coevo_host_rollout_step(void *ctx, double *state, int n_games, const int32_t *game_rows, const int32_t *actions,
                            int n_rows, int cycle, const int32_t *game_limit, int pos_first, const int32_t *games,
                            int n_list, int observe, float *obs);
/* play_game() return triples (agent_0, agent_1, adversary_0) -> rewards[n][3] fp64 */
int coevo_mpe_rewards(const double *state, int n_games, double *rewards, void *stream);

/* fused env-cycle: observe + policy step for all tasks (rows address (game,slot) pairs) in one launch */
int coevo_mpe_policy_cycle(const float *slab, const coevo_fc_task *tasks, int n_tasks, int max_rows_per_task,
                           const double *state, int n_games, const int32_t *row_game, const int32_t *row_slot,
                           int32_t *actions, int32_t *status, void *stream);

/* what coevo_mpe_final_step_pack writes beside the rewards (a population-sharded run: this rank's record of the all-gather) */
typedef struct {
    double *out;          /* [n_roles][n_local][4] */
    const float *dist;    /* [n_roles][dist_pitch]; this rank's individuals start at dist_first */
    int32_t n_roles, n_local, hof, dist_pitch, dist_first, reserved;
} coevo_final_pack;

/* A whole batch of games in one call (play_game/play_MPE at batch scale): n_cycles world cycles, each = the
 * shared-opponent policy launch (tasks `heavy`, > 8 rows each, matrix cores) on a side stream concurrently with the
 * per-individual policy launch (tasks `light`, <= 8 rows each, weight streaming) on `stream`, then coevo_mpe_step;
 * finally coevo_mpe_rewards if `rewards` is not NULL.  ctx (side stream + fork/join events + optional HIP timing
 * events around every light launch) comes from coevo_rollout_ctx_create and is owned by the caller; ctx == NULL runs
 * everything on `stream`.  Asynchronous; no host synchronisation. */
typedef struct {
    const float *slab;
    const coevo_fc_task *heavy; int32_t n_heavy; int32_t heavy_max_rows;
    const coevo_fc_task *light; int32_t n_light; int32_t light_max_rows;
    double *state; int32_t n_games; int32_t n_cycles;
    const int32_t *row_game; const int32_t *row_slot; const int32_t *game_rows;
    int32_t *actions; int32_t *status;
    const int32_t *game_limit;   /* per game agent-step limit, or NULL */
    double *rewards;             /* [n_games][3] or NULL (fused step: NULL = no closing coevo_mpe_final_step) */
    int32_t pos_first;
    int32_t n_cohorts;           /* 0/1: one lock-step chain.  K > 1: the games are partitioned into K cohorts whose
                                    cycle chains are independent (no task of one cohort touches a game of another);
                                    cohort k runs tasks [heavy_begin[k], heavy_begin[k+1]) and [light_begin[k], ...) on
                                    its own stream, so the cohorts drift out of phase and one cohort's weight
                                    streaming fills the HBM idle time of another's non-streaming phases */
    /* fused env step (both set, or both NULL for the separate coevo_mpe_step launch per cycle): a second state buffer
     * and the actions by (game, slot), double buffered [2][n_games][3].  The policy launches of cycle c then derive
     * the state of cycle c in registers from cycle c-1's buffer + actions (bit-identical arithmetic), each game's
     * adversary-seat row publishes it to the other buffer, and one coevo_mpe_final_step closes the books: one launch
     * less and one dependency less per cycle.  `rewards` is required; `actions` / `game_rows` are not used. */
    double *state_alt; int32_t *actions_by_game;
    uint64_t *light_stamps;      /* [n_cycles][COEVO_STAMP_SLOTS][2] or NULL: per light launch and slot {earliest
                                    workgroup start, latest workgroup end} in 100 MHz s_memrealtime ticks (min / max
                                    over the slots = the launch) - kernel timing that survives graph replay.
                                    With cohorts: [n_cohorts][n_cycles][COEVO_STAMP_SLOTS][2] */
    const int32_t *heavy_begin;  /* HOST arrays of n_cohorts + 1 task indices (used when n_cohorts > 1) */
    const int32_t *light_begin;
    int32_t merged;              /* fused env step only: a cohort's cycle is ONE launch (coevo_mpe_policy_cycle_merged)
                                    instead of the shared-opponent launch beside / before the per-individual one */
    int32_t concurrent_hint;     /* how many such rollouts the caller runs side by side on other streams (one call per
                                    cohort); 0 = none.  Only sizes the merged launch (coevo_mpe_policy_cycle_merged) */
    int32_t stamps_armed;        /* != 0: the caller has re-armed light_stamps to {UINT64_MAX, 0} itself (coevo_mpe_reset_multi_arm,
                                    in the reset launch that precedes the rollout anyway): no launch of its own for it */
    int32_t sync_cleared;        /* != 0: ... and zeroed sync_words (this call's cohort region) there too (coevo_mpe_reset_multi_prep) */
    int32_t *sync_words;         /* device int32 [n_cohorts][coevo_mpe_persistent_sync_words(n_games)] or NULL.  Given: a cohort
                                    whose workgroups are all resident at once (coevo_mpe_persistent_fits) runs its n_cycles as ONE persistent launch
                                    (coevo_mpe_rollout_persistent) */
    const coevo_final_pack *pack; /* fused step, `rewards` given: the closing step also writes this rank's record of the fitness
                                    all-gather (coevo_mpe_final_step_pack), or NULL.  With one cohort in a persistent launch the
                                    closing step itself runs inside that launch */
} coevo_rollout_desc;
#define COEVO_MAX_COHORTS 8
void *coevo_rollout_ctx_create(int n_timing_pairs);
void coevo_rollout_ctx_destroy(void *ctx);
/* create the streams for rollouts with up to n_cohorts cohorts; must not be called during graph capture */
int coevo_rollout_ctx_reserve_cohorts(void *ctx, int n_cohorts);
/* the hipStream_t cohort k >= 1 runs on (NULL for k = 0: the caller's stream, or when k was not reserved) */
void *coevo_rollout_ctx_cohort_stream(void *ctx, int k);
int coevo_rollout_ctx_reset_timing(void *ctx);
int coevo_rollout_ctx_light_times(void *ctx, float *host_ms_out, int max_out);  /* returns the count; blocks */
int coevo_mpe_rollout(const coevo_rollout_desc *desc, void *ctx, int time_light, void *stream);

/* coevo_mpe_policy_cycle that also brackets the launch with device clock stamps: stamps[COEVO_STAMP_SLOTS][2],
 * workgroup b folds its start into stamps[b % SLOTS][0] (min) and its end into [..][1] (max); caller-initialised to
 * {UINT64_MAX, 0}; stamps may be NULL */
int coevo_mpe_policy_cycle_stamped(const float *slab, const coevo_fc_task *tasks, int n_tasks, int max_rows_per_task,
                                   const double *state, int n_games, const int32_t *row_game,
                                   const int32_t *row_slot, int32_t *actions, int32_t *status, uint64_t *stamps,
                                   void *stream);

/* the policy launch of the fused scheme (see coevo_rollout_desc) and the closing step */
int coevo_mpe_policy_cycle_fused(const float *slab, const coevo_fc_task *tasks, int n_tasks, int max_rows_per_task,
                                 const double *state_prev, double *state_next, int n_games, const int32_t *row_game,
                                 const int32_t *row_slot, const int32_t *act_prev, int32_t *act_cur,
                                 const int32_t *game_limit, int cycle, int pos_first, int32_t *status, uint64_t *stamps,
                                 void *stream);
/* one env-cycle of one cohort in ONE launch: workgroups [0, n_heavy) run the shared-opponent tasks on the matrix
 * cores, the other n_light run the per-individual tasks (<= 8 rows each); same results as the two separate
 * coevo_mpe_policy_cycle_fused launches (replaces the per-agent-step forward of utils/game_logic_functions.py:152-163
 * for every row of the cohort at once).  The kernel holds two workgroups per CU; when the workgroups of
 * `concurrent_launches` such launches (cohorts running side by side, else 1) exceed those slots, a streaming workgroup
 * carries two nets so that everything is resident in one round.  heavy_max_rows <= 16 selects the lean kernel (four
 * workgroups per CU, 16-row shared-opponent tiles) when everything then fits at one net per workgroup; heavy_max_rows <= 8
 * with no more workgroups in flight than the device has CUs selects the small-launch kernel (every task through the per-individual body,
 * fc2 on the vector ALU: a rank of a population sharded over several GPUs, genetic_algorithm.py:125-217 split by index). */
#define COEVO_CYCLE_FORM_TILE32 0         /* 32-row shared-opponent tiles, one net per streaming workgroup, two per CU */
#define COEVO_CYCLE_FORM_TILE32_PAIRED 1  /* ... two nets per streaming workgroup */
#define COEVO_CYCLE_FORM_LEAN16 2         /* 16-row tiles, four workgroups per CU (the full-population launch) */
#define COEVO_CYCLE_FORM_SMALL 3          /* every task <= 8 rows, vector-ALU fc2 (launches that leave CUs idle) */
/* which of them a launch of this shape runs (needs the current device: the choice depends on its CU count) */
int coevo_mpe_cycle_kernel_form(int n_heavy, int n_light, int heavy_max_rows, int light_max_rows, int concurrent_launches);
int coevo_mpe_policy_cycle_merged(const float *slab, const coevo_fc_task *heavy_tasks, int n_heavy,
                                  const coevo_fc_task *light_tasks, int n_light, int light_max_rows,
                                  const double *state_prev, double *state_next, int n_games,
                                  const int32_t *row_game, const int32_t *row_slot, const int32_t *act_prev,
                                  int32_t *act_cur, const int32_t *game_limit, int cycle, int pos_first,
                                  int32_t *status, uint64_t *stamps, int concurrent_launches, int heavy_max_rows,
                                  void *stream);
/* n_cycles env-cycles of one cohort in ONE launch (the env step and the forward fused into a persistent
 * whole-rollout kernel; replaces n_cycles coevo_mpe_policy_cycle_merged launches, i.e. the whole per-step loop of
 * utils/game_logic_functions.py:138-212 for every game of the cohort).  Only for launch shapes whose workgroups are all
 * resident at once (coevo_mpe_persistent_fits: every task <= 8 rows, at most two workgroups per CU counting
 * `concurrent_launches` such launches side by side; anything else: COEVO_ERR_UNSUPPORTED): a workgroup
 * keeps its net's small layers in LDS and its rows' games in LDS for the whole rollout; per cycle a row posts its action as one
 * tagged 32-bit word and waits for the two other rows of its game (bounded: COEVO_ST_SYNC_TIMEOUT in the status word).
 * state = buffer 0 (the reset state), state_alt = buffer 1; on return they and actions_by_game [2][n_games][3] hold exactly
 * what the last of the per-cycle launches leaves for coevo_mpe_final_step(state of cycle n_cycles - 1, cycle n_cycles - 1).
 * sync_words: device int32 [coevo_mpe_persistent_sync_words(n_games)], scratch of this call (zeroed by it on `stream` unless
 * sync_cleared != 0: the caller zeroed them in a launch of its own that precedes this one, coevo_mpe_reset_multi_prep).
 * rewards ([n_games][3] fp64) given: each game's owner row also closes the books itself (coevo_mpe_final_step's arithmetic,
 * and coevo_mpe_final_step_pack's record when `pack` is given) - no closing launch is needed; NULL: the caller closes them.
 * stamps: [n_cycles][COEVO_STAMP_SLOTS][2] or NULL, per cycle {earliest start after the wait, latest action posted}.
 * Every task's games must have all three of their rows among the tasks of this call.  COEVO_PERSISTENT=0 in the environment
 * makes coevo_mpe_rollout keep the per-cycle launches (A/B). */
int coevo_mpe_persistent_sync_words(int n_games);
int coevo_mpe_persistent_fits(int n_heavy, int n_light, int heavy_max_rows, int light_max_rows, int concurrent_launches);
int coevo_mpe_rollout_persistent(const float *slab, const coevo_fc_task *heavy_tasks, int n_heavy,
                                 const coevo_fc_task *light_tasks, int n_light, int light_max_rows, int heavy_max_rows,
                                 double *state, double *state_alt, int n_games, const int32_t *row_game,
                                 const int32_t *row_slot, int32_t *actions_by_game, const int32_t *game_limit, int n_cycles,
                                 int pos_first, int32_t *status, uint64_t *stamps, int32_t *sync_words,
                                 int concurrent_launches, double *rewards, const coevo_final_pack *pack, int sync_cleared,
                                 void *stream);
int coevo_mpe_final_step(const double *state, int n_games, const int32_t *actions_by_game, int cycle,
                         const int32_t *game_limit, int pos_first, double *rewards, void *stream);
/* ... + this rank's record of the fitness all-gather in the same launch (a population-sharded run; genetic_algorithm.py:
 * 140-146: only the last HoF game's reward survives, quirk Q2).  Games [0, n_roles * n_local * hof) are laid out
 * [role][local individual][hof game]; pack[role][j][0..2] = the play_game triple of game (role, j, hof - 1), [3] =
 * (double)dist[role * dist_pitch + dist_first + j] (the individual's distance to the stale agent, quirk Q3).  The ranks'
 * packs, all-gathered rank-major, are what coevo_ga_select_gathered reads: no copy kernel between rollout and selection. */
int coevo_mpe_final_step_pack(const double *state, int n_games, const int32_t *actions_by_game, int cycle,
                              const int32_t *game_limit, int pos_first, double *rewards, double *pack, const float *dist,
                              int n_roles, int n_local, int hof, int dist_pitch, int dist_first, void *stream);

/* ---------------------------------------------------------------- K3/K4/K8: offspring on device ------------- */
/* The noise contract: rounds of the Philox4x32 generator behind every device-built offspring (7).  A checkpoint / a binding
 * that resumes a `device_philox` run must see the number it was started with - other rounds are other numbers. */
int coevo_noise_rounds(void);
/* the generator itself, for known-answer tests: n x (counter[4], key[2]) -> n x 4 words after `rounds` (7 or 10) rounds */
int coevo_philox4x32(int rounds, const uint32_t *ctr_key, int n, uint32_t *out, void *stream);
/* ... and the standard normals the offspring kernels add (agent.py:27-28's torch.normal / :52's np.random.normal in the
 * `device_philox` mode): out[4 i + k] = normal k of counter q_first + i of stream (stream_lo, stream_hi) under `seed` */
int coevo_philox_normals(uint64_t seed, uint32_t stream_lo, uint32_t stream_hi, uint32_t q_first, int n_quads, float *out,
                         void *stream);
/* child = parent + sigma * eps(seed, stream, p), p = canonical flat index; Philox4x32-7 (the Crush-resistant minimum) + Box-Muller with
 * fmaf-only polynomials (bit-reproducible against the oracle).  Replaces clone()+Agent.mutate (agent.py:25-29,
 * genetic_algorithm.py:32-48) and Agent.mutate_ES (agent.py:51-53).
 *   parent_slab/child_slab: slabs in device layout; parent_idx[c] = net index of child c's parent in parent_slab
 *   (device array, so elite ids chosen on device never visit the host); child c is written at net index
 *   child_first + c; its noise stream is (stream_lo = stream_lo_first + c, stream_hi).
 *   skip_layernorm != 0 leaves LayerNorm affine untouched (ES, MPE/fcnetwork.py:185-199). */
int coevo_fc_perturb(const float *parent_slab, const int32_t *parent_idx, float *child_slab, int child_first,
                     int n_children, int D, const float *sigma_dev, uint64_t seed, uint32_t stream_lo_first,
                     uint32_t stream_hi, int skip_layernorm, void *stream);
/* coevo_fc_perturb with flags: bit 0 = skip_layernorm, bit 1 = antithetic pairs (cfg 3 extension mode, NOT in the
 * reference: individuals 2m and 2m+1, counted from stream_lo_first + c, share noise stream m; the odd one gets -eps) */
#define COEVO_PERTURB_SKIP_LAYERNORM 1
#define COEVO_PERTURB_ANTITHETIC 2
int coevo_fc_perturb_flags(const float *parent_slab, const int32_t *parent_idx, float *child_slab, int child_first,
                           int n_children, int D, const float *sigma_dev, uint64_t seed, uint32_t stream_lo_first,
                           uint32_t stream_hi, int flags, void *stream);
/* coevo_fc_perturb with the generation taken from a device counter: stream_hi_eff = stream_hi + 4 * (*gen_dev) */
int coevo_fc_perturb_gen(const float *parent_slab, const int32_t *parent_idx, float *child_slab, int child_first,
                         int n_children, int D, const float *sigma_dev, uint64_t seed, uint32_t stream_lo_first,
                         uint32_t stream_hi, int skip_layernorm, const int32_t *gen_dev, void *stream);
/* ... and, while each child is in registers, its squared L2 distance (Linear weights/biases only) to the net
 * `dist_ref` (one net, slab layout): per-block partial sums go to dist_partial[n_children][coevo_fc_perturb_blocks(D)]
 * (fp64); coevo_fc_distance_finalize turns them into the distances diversity_penalty needs next generation, saving a
 * second pass over the population.  dist_ref and dist_partial are both NULL or both set. */
int64_t coevo_fc_perturb_blocks(int D);
int coevo_fc_perturb_dist(const float *parent_slab, const int32_t *parent_idx, float *child_slab, int child_first,
                          int n_children, int D, const float *sigma_dev, uint64_t seed, uint32_t stream_lo_first,
                          uint32_t stream_hi, int skip_layernorm, const int32_t *gen_dev, const float *dist_ref,
                          double *dist_partial, void *stream);
/* dist[first + c] = sqrt(sum of child c's partials); head != NULL: also dist[first-1] = *head (the unchanged best
 * individual keeps the distance it had in the previous population, stashed with coevo_gather_f32 before breeding) */
int coevo_fc_distance_finalize(const double *dist_partial, int n_blocks, int n, float *dist, int first,
                               const float *head, void *stream);
int coevo_gather_f32(float *dst, const float *src, const int32_t *idx, int n, void *stream);  /* dst[i]=src[idx[i]] */
/* The breeding of several roles in ONE launch each (a generation of genetic_algorithm.py:296-321 breeds every role
 * right after the other; as nine separate 10-40 us launches per cohort they sat in front of every rollout chain).
 * Job j = the arguments of one coevo_fc_perturb_dist / coevo_fc_distance_finalize call; results are identical to the
 * per-role calls.  `jobs` is host memory, n_jobs <= COEVO_MAX_JOBS. */
#define COEVO_MAX_JOBS 4
typedef struct {
    const float *parent_slab; const int32_t *parent_idx; float *child_slab; const float *sigma_dev;
    const float *dist_ref; double *dist_partial;
    int32_t child_first, n_children, D; uint32_t stream_lo_first, stream_hi; int32_t pad;
} coevo_fc_perturb_job;
int coevo_fc_perturb_dist_multi(const coevo_fc_perturb_job *jobs, int n_jobs, uint64_t seed, int skip_layernorm,
                                const int32_t *gen_dev, void *stream);
typedef struct {
    const double *dist_partial; float *dist; const float *head; int32_t n_blocks, n, first, pad;
} coevo_fc_finalize_job;
int coevo_fc_distance_finalize_multi(const coevo_fc_finalize_job *jobs, int n_jobs, void *stream);
/* ... with coevo_counter_add(counter, 1) in the same launch: the last launch of a device-resident generation */
int coevo_fc_distance_finalize_multi_tick(const coevo_fc_finalize_job *jobs, int n_jobs, int32_t *counter, void *stream);
/* Multi-GPU Co-GA: this generation's elites (ids order[0..E-1] on the device) rebuilt from LAST generation's elites
 * and the counter-based noise their children were bred with (id 0 = last best unchanged, id >= 1 = elite_prev[(id-1)%E]
 * + sigma_prev*eps(stream (id-1, stream_hi_prev))): no weight crosses xGMI, every rank gets identical bits.
 * GA mutation touches every parameter (LayerNorm included). elite_new must not alias elite_prev. */
int coevo_fc_rebuild_elites(const float *elite_prev, const int32_t *order, float *elite_new, int E, int D,
                            const float *sigma_prev_dev, uint64_t seed, uint32_t stream_hi_prev,
                            const int32_t *gen_dev /* or NULL; adds 4*(*gen_dev - 1) */, void *stream);
/* Selection of up to three roles in ONE launch = coevo_sharing_score + coevo_ga_fitness + coevo_rank_desc per role
 * (same arithmetic), plus best_dist = dist[order[0]] when best_dist is set (the distance the unchanged best individual
 * keeps, see coevo_fc_distance_finalize).  All pointers are device memory; pop <= 4096. */
typedef struct coevo_ga_select_role {
    const float *dist;       /* [pop] distances to the stale agent */
    const double *rewards;   /* play_game triples [n_games][3] */
    float *diversity;        /* [1] out: the sharing score */
    float *fitness;          /* [pop] out */
    int32_t *order;          /* [pop] out: argsort(fitness)[::-1] */
    float *best_dist;        /* [1] out, or NULL */
    int32_t game_first;      /* individual i's games are game_first + i*games_per_individual .. (last one counts, Q2) */
    int32_t slot;            /* which element of the triple is this role's return */
} coevo_ga_select_role;
int coevo_ga_select(const coevo_ga_select_role *roles, int n_roles, int pop, int games_per_individual, int hof,
                    void *stream);
/* ... of a population-sharded run, straight off the all-gathered buffer (genetic_algorithm.py:125-217 split by individual
 * index, :223-225 on every rank): gathered[rank][role][j][0..2] = play_game triple of the last HoF game of rank `rank`'s j-th
 * individual (what coevo_mpe_final_step_pack wrote on that rank), [3] = its distance to the stale agent; individual i =
 * rank i / n_local, j = i % n_local.  The roles' `dist` / `rewards` are not read. */
int coevo_ga_select_gathered(const coevo_ga_select_role *roles, int n_roles, int pop, int hof, const double *gathered,
                             int n_local, void *stream);
/* the arguments of coevo_ga_adapt_sigma as a block, + sigma32_prev: NULL, or device [3] that receives sigma32 as it was BEFORE
 * the rule ran (what the children evaluated in this generation were bred with: coevo_ga_promote_rebuild's sigma) */
typedef struct coevo_ga_adapt_args {
    const double *rewards; const int32_t *gen_dev; double *hist; double *sig_hist; double *sigma64; float *sigma32;
    float *sigma32_prev; double sig_min, sig_max; int32_t eval_first_game, cap, adaptive, reserved;
} coevo_ga_adapt_args;
/* coevo_ga_select (gathered == NULL) or coevo_ga_select_gathered with coevo_ga_adapt_sigma's work in the same launch */
int coevo_ga_select_adapt(const coevo_ga_select_role *roles, int n_roles, int pop, int games_per_individual, int hof,
                          const double *gathered, int n_local, const coevo_ga_adapt_args *adapt, void *stream);
/* Promotion of up to three roles in ONE launch (replaces five coevo_fc_gather launches per role;
 * genetic_algorithm.py:262-275): elite[k] = pop[order[k]] (k < E <= 8; skipped when elites_from_pop == 0: the elites
 * are already in `elite`, e.g. rebuilt by coevo_fc_rebuild_elites), hof.pop(0); hof.append(elite[0]) (hof <= 16 nets,
 * shifted in place), and pop[0] = elite[0] when best_to_pop0 != 0. */
typedef struct coevo_ga_promote_role {
    float *pop, *hof, *elite;   /* first net of each region (slab layout of width D) */
    const int32_t *order;       /* device, [>= E]; may be NULL when elites_from_pop == 0 */
    int32_t D, elites_from_pop, best_to_pop0, reserved;
} coevo_ga_promote_role;
int coevo_ga_promote(const coevo_ga_promote_role *roles, int n_roles, int E, int hof, void *stream);
/* ... with coevo_counter_add(counter, 1) in the same launch: the last launch of a device-resident generation's tail */
int coevo_ga_promote_tick(const coevo_ga_promote_role *roles, int n_roles, int E, int hof, int32_t *counter, void *stream);
/* ... with the elites REBUILT in the same launch (roles with elites_from_pop == 0; `order` required): elite[k] = individual
 * order[k] of the generation just evaluated = the unchanged best (id 0: old elite 0) or child c = id - 1 = old elite[c % E] +
 * sigma[role] * noise(stream (c, stream_hi_prev + role [+ 4 (g - 1) with gen_dev])), regenerated in place from the OLD elites
 * (a rank of a sharded population holds only its own individuals: no weight crosses xGMI).  One launch instead of, per role,
 * coevo_fc_gather + coevo_fc_rebuild_elites + the promotion.  sigma: device [n_roles]. */
int coevo_ga_promote_rebuild(const coevo_ga_promote_role *roles, int n_roles, int E, int hof, const float *sigma,
                             uint64_t seed, uint32_t stream_hi_prev, const int32_t *gen_dev, void *stream);
/* net copies inside/between slabs driven by device-resident indices: dst[dst_first+i] = src[src_idx[i]] */
int coevo_fc_gather(const float *src_slab, const int32_t *src_idx, float *dst_slab, int dst_first, int n, int D,
                    void *stream);

/* K5: theta += lr/(n*sigma) * sum_i fitness[i] * (pert_i - theta) over the Linear weights/biases, i ascending
 * (compute_weight_update, evolutionary_strategy.py:120-148; the reference multiplies the stored n x P noise matrix).
 * theta is ONE net in slab layout, pert_slab the n perturbed nets coevo_fc_perturb materialised from it. */
int coevo_es_update(float *theta_slab_net, const float *pert_slab, int D, const float *fitness, int n,
                    const float *sigma_dev, float lr, void *stream);

/* K5 in two steps - the canonical ES summation of this build: the n individuals are cut into chunks_total chunks (chunk
 * c = [c*n/C, (c+1)*n/C)), each chunk is summed i-ascending with one fmaf per term from 0 (coevo_es_partial), the chunk
 * sums are added left to right and applied (coevo_es_apply).  chunks_total = 1 is coevo_es_update.  More workgroups on
 * one GPU, and the unit a population shard owns on several (evolutionary_strategy.py:236-265 distributed as SURVEY 8e
 * describes: a rank computes the partial sums of its own individuals, partials are all-gathered, every rank applies the
 * identical update - N ranks give the bits of one).
 *   pert_slab_local: the caller's perturbed nets, the first one being global individual ind_first (= start of chunk
 *   chunk_first); fitness_all [n_total] indexed by the global individual; partial [n_chunks][slab stride]. */
int coevo_es_partial(const float *theta_net, const float *pert_slab_local, int ind_first, int D,
                     const float *fitness_all, int n_total, int chunks_total, int chunk_first, int n_chunks,
                     float *partial, void *stream);
/* partial of global chunk c at partials + (c / chunks_per_block) * block_stride_floats + (c % chunks_per_block) *
 * stride: one block per rank after the all-gather (block_stride_floats = floats each rank contributed) */
int coevo_es_apply(float *theta_net, const float *partials, int chunks_total, int chunks_per_block,
                   int64_t block_stride_floats, int D, int n_total, const float *sigma_dev, float lr, void *stream);
/* cfg 3 extension mode (BASELINE.json configs[2]; the reference's own normalisation is commented out at
 * evolutionary_strategy.py:133-135): out[i] = rank_i / (n-1) - 0.5, stable ascending rank (ties: lower index first) */
int coevo_centered_ranks(const float *fitness, int n, float *out, void *stream);

/* ---------------------------------------------------------------- K6/K7: fitness, sharing, selection -------- */
/* distances d[i] = || w_i - w_ref ||_2 over the Linear weights/biases (get_weights_ES default layers) and the
 * sharing score sum_i max(0, 1 - d_i/mean(d)) (utils/game_logic_functions.py:12-37). ref_net is one net in slab
 * layout, pop_slab holds n nets.  dist [n] fp32, score one fp32. */
int coevo_fc_diversity(const float *ref_net, const float *pop_slab, int n, int D, float *dist, float *score,
                       void *stream);
/* the distances alone (the part each GPU computes for its population shard) */
int coevo_fc_distance(const float *ref_net, const float *pop_slab, int n, int D, float *dist, void *stream);
/* the score alone from n distances (used when the distances were all-gathered from several GPUs) */
int coevo_sharing_score(const float *dist, int n, float *score, void *stream);
/* GA fitness of one role phase (genetic_algorithm.py:140-146, quirk Q2: only the LAST HoF game counts):
 * fitness[i] = float(rewards[game_first + i*gpi + gpi-1][slot] / hof) / (1 + *diversity), gpi = games_per_individual
 * (= hof on a raw rewards table, 1 on a table that already holds each individual's last game); float32, as
 * numpy >= 2 evaluates python_float / np.float32.  rewards = play_game triples [games][3] fp64. */
int coevo_ga_fitness(const double *rewards, int game_first, int pop, int games_per_individual, int hof, int slot,
                     const float *diversity, float *fitness, void *stream);
/* order = np.argsort(fitness)[::-1] (genetic_algorithm.py:223-225; stable ascending sort reversed, so ties put the
 * HIGHER index first, quirk Q13).  n <= 4096. */
int coevo_rank_desc(const float *fitness, int n, int32_t *order, void *stream);

/* evaluate_current_weights' 10-game means (genetic_algorithm.py:12-29) + the adaptive mutation power rule (:323-345,
 * quirk Q5) on the device, arithmetic identical to the reference's numpy/python float64 (np.mean's pairwise order
 * included).  *gen_dev = g: rewards[eval_first_game .. +9] are generation g-1's evaluation games; their means are
 * appended to hist[3][cap] at index g-1, sigma64[3] (agent_0, agent_1, adversary) is updated when `adaptive`, logged to
 * sig_hist[3][cap] and rounded into sigma32[3] for coevo_fc_perturb.  g == 0 only refreshes sigma32. */
int coevo_ga_adapt_sigma(const double *rewards, int eval_first_game, const int32_t *gen_dev, double *hist,
                         double *sig_hist, int cap, double *sigma64, float *sigma32, double sig_min, double sig_max,
                         int adaptive, void *stream);
int coevo_counter_add(int32_t *counter, int value, void *stream);   /* the device generation counter's tick */

/* ---------------------------------------------------------------- K2: DeepQN policy step -------------------- */
/* DeepQN.forward (Atari/deepqn.py:39-48) + first-max action (the rule its docstring :51-52 intends; the reference's
 * own determine_action does not run) for rows of uint8 frames [84][84][C] in HWC order, as the env hands
 * them over (the permute of preprocess_observation, utils/game_logic_functions.py:78, is folded into the load).
 * BatchNorm runs in training mode at batch 1: per-sample, per-channel spatial statistics; rows never mix.
 * Canonical flat order = torch parameters() order: conv1.w conv1.b conv2.w conv2.b conv3.w conv3.b fc1.w fc1.b
 * output.w output.b vbn1.w vbn1.b vbn2.w vbn2.b vbn3.w vbn3.b. */
#define COEVO_DQN_LOGIT_STRIDE 32    /* floats per logits row; n_actions <= 32 (6 pong, 18 boxing) */
#define COEVO_DQN_MAX_ROWS 16        /* frames one task (one weight set) may carry */
/* The fc1 block (3136 -> 512, 95 % of a net: Atari/deepqn.py:46) of a slab has one of two layouts, an attribute of the engine
 * that owns the slab; every entry point whose work depends on it - coevo_dqn_pack / _unpack / _perturb / _perturb_blocks and
 * the coevo_dqn_forward_* family (also through coevo_frames_rollout_desc.C) - takes it or-ed into its channel argument:
 *   C                        streamed: [out block of 64][k / 4][out % 64][k % 4] - v_mfma_f32_4x4x1, rows in groups of four
 *                            (Co-ES: one frame per task)
 *   C | COEVO_DQN_FC1_TILED  tiled for v_mfma_f32_16x16x4: [out block][k / 16][16-out tile][lane = 16 (k % 4) + out % 16][(k / 4) % 4]
 *                            - sixteen matrix instructions per 16 k of a <= 16-row task without a vector instruction
 *                            touching an operand (Co-GA: 10 / 16 frames per task)
 * Results do not depend on the layout (same sequential-k chains); sizes (param_count, slab_stride) do not either. */
#define COEVO_DQN_FC1_TILED 0x100
int64_t coevo_dqn_param_count(int C, int n_actions);      /* 1 687 526 for C=4, n=6 */
int64_t coevo_dqn_slab_stride(int C, int n_actions);
int64_t coevo_dqn_workspace_bytes(int n_rows_total);      /* conv3 activations + fc1 outputs of every row */
int coevo_dqn_pack(const float *flat, float *slab, int n, int C, int n_actions, void *stream);
typedef struct {
    int64_t net_off;   /* float offset of the net inside the slab */
    int32_t row_begin;
    int32_t n_rows;    /* 1 .. COEVO_DQN_MAX_ROWS frames that share this weight set */
} coevo_dqn_task;
/* frames [n_rows_total][84][84][C] uint8; actions [n_rows_total]; logits [n_rows_total][COEVO_DQN_LOGIT_STRIDE] or
 * NULL; workspace of coevo_dqn_workspace_bytes(n_rows_total) bytes.  The tasks must partition the rows 0 ..
 * n_rows_total-1 in ascending row_begin order (a frame's task is found by binary search).  Three launches on `stream`. */
int coevo_dqn_forward_argmax(const float *slab, const coevo_dqn_task *tasks, int n_tasks, int max_rows_per_task,
                             int n_rows_total, int C, int n_actions, const uint8_t *frames, int32_t *actions,
                             float *logits, int32_t *status, void *workspace, void *stream);

/* the same with one of its launches (timed_kernel: 0 = conv stack, 1 = fc1) bracketed by the next timing event pair of a
 * rollout context (coevo_rollout_ctx_create / coevo_rollout_ctx_light_times); eager enqueue only; timing_ctx == NULL =
 * untimed */
int coevo_dqn_forward_argmax_timed(const float *slab, const coevo_dqn_task *tasks, int n_tasks, int max_rows_per_task,
                                   int n_rows_total, int C, int n_actions, const uint8_t *frames, int32_t *actions,
                                   float *logits, int32_t *status, void *workspace, void *timing_ctx, int timed_kernel,
                                   void *stream);
/* conv stack + fc1 only: leaves the hidden rows (post-ReLU fc1 outputs) in the workspace for coevo_dqn_out_synth_step,
 * which runs the output layer in the env-step launch (two launches instead of three per agent-step) */
int coevo_dqn_forward_hidden_timed(const float *slab, const coevo_dqn_task *tasks, int n_tasks, int max_rows_per_task,
                                   int n_rows_total, int C, int n_actions, const uint8_t *frames, void *workspace,
                                   void *timing_ctx, int timed_kernel, void *stream);
/* bracket whatever is enqueued on `stream` between the two calls with the context's next timing event pair */
int coevo_timing_begin(void *ctx, void *stream);
int coevo_timing_end(void *ctx, void *stream);
int coevo_dqn_unpack(const float *slab, float *flat, int n, int C, int n_actions, void *stream);
/* n nets from a slab of one fc1 layout into a slab of the other (channel arguments of the two slabs; slab strides are equal) */
int coevo_dqn_relayout(const float *src_slab, float *dst_slab, int n, int C_src, int C_dst, int n_actions, void *stream);

/* ---------------------------------------------------------------- DeepQN population engine (cfg 4 / cfg 5) ---------- */
/* Offspring of the DeepQN layout on the device: child = parent +- sigma * eps(seed, stream, p), p = canonical flat index
 * (the order above), same Philox / Box-Muller as coevo_fc_perturb.  Replaces AtariAgent.clone + Agent.mutate
 * (Atari/atari_agent.py:27-30, agent.py:25-29: every parameter, BatchNorm affine included) and the ES perturbation of the
 * perturbable layers (Atari/deepqn.py:158-171: BatchNorm excluded).
 *   flags  COEVO_DQP_SKIP_BN     BatchNorm affine untouched (ES)
 *          COEVO_DQP_ANTITHETIC  individuals 2m / 2m+1 (counted from stream_lo_first + c) share stream m, odd one -eps
 *          COEVO_DQP_FROM_ORDER  elite rebuild (multi-GPU Co-GA, as coevo_fc_rebuild_elites): parent_idx is this
 *                                generation's ranking; child e = individual id = parent_idx[e] of the population bred
 *                                from the E nets of parent_slab: id 0 -> parent 0 unchanged, else parent (id-1) % E +
 *                                noise stream id-1; child_slab must not alias parent_slab
 *          COEVO_DQP_COPY        no noise: child = parent (child_slab may be NULL: distance only)
 *   gen_dev != NULL: stream_hi_eff = stream_hi + 4 * (*gen_dev + gen_bias)
 *   dist_ref / dist_partial (both or neither): squared L2 distance of every child to the net dist_ref over ALL parameters
 *   (DeepQN.get_weights_ES() default = self.layers, which holds the BatchNorm layers too, Atari/deepqn.py:14-37), as
 *   fp64 partial sums [n_children][coevo_dqn_perturb_blocks]; coevo_fc_distance_finalize turns them into distances. */
#define COEVO_DQP_SKIP_BN 1
#define COEVO_DQP_ANTITHETIC 2
#define COEVO_DQP_FROM_ORDER 4
#define COEVO_DQP_COPY 8
int64_t coevo_dqn_perturb_blocks(int C, int n_actions);
int coevo_dqn_perturb(const float *parent_slab, const int32_t *parent_idx, float *child_slab, int child_first,
                      int n_children, int C, int n_actions, const float *sigma_dev, uint64_t seed,
                      uint32_t stream_lo_first, uint32_t stream_hi, int flags, int E, const int32_t *gen_dev,
                      int gen_bias, const float *dist_ref, double *dist_partial, void *stream);
/* K5 for the DeepQN layout (see coevo_es_partial / coevo_es_apply): BatchNorm affine is never updated */
int coevo_dqn_es_partial(const float *theta_net, const float *pert_slab_local, int ind_first, int C, int n_actions,
                         const float *fitness_all, int n_total, int chunks_total, int chunk_first, int n_chunks,
                         float *partial, void *stream);
int coevo_dqn_es_apply(float *theta_net, const float *partials, int chunks_total, int chunks_per_block,
                       int64_t block_stride_floats, int C, int n_actions, int n_total, const float *sigma_dev, float lr,
                       void *stream);
/* dst[dst_first + i] = src[src_idx[i]] for nets of any slab layout, stride_floats apart (elites, HoF FIFO, best) */
int coevo_net_gather(const float *src_slab, const int32_t *src_idx, float *dst_slab, int dst_first, int n,
                     int64_t stride_floats, void *stream);

/* Synthetic two-player env in the shape of pettingzoo.atari pong_v3 / boxing_v2 (ALE is not in the image, SURVEY 8d cfg
 * 4/5): agents first_0 / second_0 alternate, the frame of agent-step t is uint8 [84][84][C] noise keyed by (seed, the
 * game's reset ordinal, t, the action of step t-1) - no game dynamics, but a real dependency chain from step to step.
 * hit(t) = [action_t == target(seed, ordinal, t)]; zero-sum rewards with PettingZoo's AEC bookkeeping, and play_atari's
 * crediting (utils/game_logic_functions.py:104-108: the actor receives what env.last() returns after env.step, i.e. the
 * NEXT agent's cumulative reward): credited(t) = hit(t-1) - hit(t) to actor t & 1.
 * One call = the env side of agent-step t for every game: books the action of step t-1 (read through row_prev /
 * actions_prev) and writes the frame of step t at frames[row_cur[g]] (frames == NULL: bookkeeping only, the closing call
 * with t = T).  t = 0 resets.  ordinal(g) = game_ordinal0[g] + *gen_dev * ordinals_per_gen; negative = disabled.
 *   game_state [n_games][4] int32, acc [n_games][3] fp64 = play_game's (first_0, second_0) returns + one unused slot,
 *   limit [n_games] = agent-step limit per game (max_timesteps_per_episode / max_evaluation_steps). */
int coevo_synth_step(int32_t *game_state, double *acc, int n_games, const int64_t *game_ordinal0, const int32_t *gen_dev,
                     int64_t ordinals_per_gen, int t, const int32_t *limit, const int32_t *row_prev,
                     const int32_t *actions_prev, const int32_t *row_cur, uint8_t *frames, int C, int n_actions,
                     uint64_t seed, void *stream);
/* coevo_synth_step for t >= 1 with the output layer of step t-1 fused in: game g's workgroup turns the hidden row of its
 * actor (row_prev[g] of the workspace coevo_dqn_forward_hidden_timed filled for tasks_prev / n_rows_total) into logits and
 * the first-max action (Atari/deepqn.py:48 + the argmax rule of MPE/fcnetwork.py:78-85), stores it in actions_prev, books
 * it and writes the next frame.  Same results as coevo_dqn_forward_argmax + coevo_synth_step. */
int coevo_dqn_out_synth_step(int32_t *game_state, double *acc, int n_games, const int64_t *game_ordinal0,
                             const int32_t *gen_dev, int64_t ordinals_per_gen, int t, const int32_t *limit,
                             const int32_t *row_prev, int32_t *actions_prev, const int32_t *row_cur, uint8_t *frames, int C,
                             int n_actions, uint64_t seed, const float *slab, const coevo_dqn_task *tasks_prev,
                             int n_tasks_prev, int n_rows_total, const void *workspace, int32_t *status, void *stream);

/* The DeepQN games with the env on the HOST (PCIe-inclusive form of cfg 4 / cfg 5): what play_atari drives through env.observe
 * / env.step / env.last (utils/game_logic_functions.py:84-120) when the env lives in host memory (:47-53), for all of a rank's
 * games.  Per agent-step and cohort the context's host cores book the previous action and render every live game's next frame
 * (the synthetic env of coevo_synth_step, same bytes) into page-locked memory; the cohort's stream copies the frames up, runs
 * coevo_dqn_forward_argmax and copies the actions down; cohorts alternate.  Returns when step T has been booked. */
typedef struct {
    const coevo_dqn_task *tasks[2];   /* DEVICE: task table per parity (who acts); rows = this cohort's games */
    const int32_t *rows[2];           /* HOST [n_games]: row of game (game_first + i) in the parity's frame / action arrays */
    uint8_t *frames_host;             /* HOST, page-locked: [n_games][84 * 84 * C] */
    uint8_t *frames_dev;              /* DEVICE twin */
    int32_t *actions_host;            /* HOST, page-locked: [n_games] */
    int32_t *actions_dev;             /* DEVICE twin */
    void *workspace;                  /* DEVICE: coevo_dqn_workspace_bytes(n_games) */
    int32_t n_tasks[2], max_rows[2];
    int32_t game_first, n_games;      /* this cohort's games: [game_first, game_first + n_games) of the arrays below */
} coevo_frame_cohort;
typedef struct {
    const float *slab;                /* DEVICE */
    int32_t *status;                  /* DEVICE */
    int32_t *game_state;              /* HOST [n_games][4], as coevo_synth_step keeps it */
    double *acc;                      /* HOST [n_games][3]: play_game's (first_0, second_0) returns + one unused slot */
    const int64_t *game_ordinal0;     /* HOST [n_games] */
    const int32_t *limit;             /* HOST [n_games] agent-step limits */
    const coevo_frame_cohort *cohorts;
    double *phase_us;                 /* NULL, or HOST [5]: mean microseconds per cohort-step of {host wait for the actions, host
                                         env (book + render), frames host->device, conv stack + fc1 + output layer, actions
                                         device->host} */
    int64_t generation, ordinals_per_gen;   /* ordinal(g) = game_ordinal0[g] + generation * ordinals_per_gen; < 0: disabled */
    uint64_t seed;
    int32_t n_games, n_cohorts, C, n_actions, T, reserved;
} coevo_frames_rollout_desc;
int coevo_dqn_host_frames_rollout(void *ctx, const coevo_frames_rollout_desc *desc, void *stream);
/* the frame of agent-step t of game `ordinal` after `last_action` (0xFF: none yet), on the calling thread */
int coevo_synth_frame_host(uint8_t *dst, int C, uint64_t seed, int64_t ordinal, int t, int last_action);

#ifdef __cplusplus
}
#endif
#endif /* COEVO_H */
