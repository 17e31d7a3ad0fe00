// DeepQN population engine pieces (BASELINE configs 4 / 5): offspring of the 1.69 M-parameter conv net built on the
// device, and the synthetic Atari-shaped env that stands in for ALE (absent from the image, SURVEY 8d cfg 4/5).
//
// Replaces (reference file:line): AtariAgent.clone's state_dict copy (Atari/atari_agent.py:27-30) + Agent.mutate
// (agent.py:25-29: EVERY parameter, BatchNorm affine included) for Co-GA; the perturbable-weights rule of Co-ES
// (Atari/deepqn.py:158-171, 214-217: BatchNorm excluded); diversity_penalty's distances over get_weights_ES() - for
// DeepQN that is all parameters (Atari/deepqn.py:14-37: self.layers holds the three BatchNorm layers too); and the
// env side of play_atari (utils/game_logic_functions.py:84-119).
//
// Noise: eps(seed, stream, p), p = canonical flat index in torch parameters() order (include/coevo.h), the same
// Philox / Box-Muller as the FCNetwork offspring - the oracle's generic flat perturb is the checker.
#include "dqn_common.hip.h"
#include "philox.hip.h"

namespace coevo {

// flags of dqn_perturb_kernel
constexpr int DQP_SKIP_BN = 1;      // ES: BatchNorm affine untouched
constexpr int DQP_ANTITHETIC = 2;   // extension mode: individuals 2m / 2m+1 share stream m, the odd one takes -eps
constexpr int DQP_FROM_ORDER = 4;   // rebuild elites: child e is individual order[e] of the population bred from `parent`
constexpr int DQP_COPY = 8;         // no noise (distance of an existing net / plain copy)

__global__ __launch_bounds__(256) void dqn_perturb_kernel(const float *parent_slab, const int32_t *parent_idx,
                                                           float *child_slab, int child_first, int C, int n_actions,
                                                           const float *sigma_dev, uint64_t seed,
                                                           uint32_t stream_lo_first, uint32_t stream_hi, int flags,
                                                           int E, const int32_t *gen_dev, int gen_bias,
                                                           const float *dist_ref, double *dist_partial, int fc1_tiled)
{
    __shared__ double scratch[4];
    if (gen_dev) stream_hi += 4u * (uint32_t)(*gen_dev + gen_bias);
    const int c = blockIdx.y;
    const DqnLayout L = dqn_layout(C, n_actions);
    // tiled fc1 block: the thread -> slab mapping is shifted by dqn_perturb_shift() quads, so that a wave's 64 lanes are the 64
    // lanes of ONE (output block, super-quad, tile) of the block (whose first quad is not a multiple of 64)
    const int64_t s0 = ((int64_t)blockIdx.x * 256 + threadIdx.x - (fc1_tiled ? dqn_perturb_shift(L) : 0)) * 4;
    double d2 = 0.0;
    if (s0 >= 0 && s0 < L.stride) {
        int parent = parent_idx ? parent_idx[c] : 0;
        uint32_t ind = stream_lo_first + (uint32_t)c;
        bool copy = (flags & DQP_COPY) != 0;
        if (flags & DQP_FROM_ORDER) {  // parent_idx = this generation's ranking: id 0 is last generation's best
            const int id = parent_idx[c];
            copy = id == 0;
            parent = copy ? 0 : (id - 1) % E;
            ind = (uint32_t)(id - 1);
        }
        const uint32_t slo = (flags & DQP_ANTITHETIC) ? (ind >> 1) : ind;
        const bool negate = (flags & DQP_ANTITHETIC) && (ind & 1u);
        const float sigma = copy ? 0.0f : *sigma_dev;
        const float4 pv = *reinterpret_cast<const float4 *>(parent_slab + (int64_t)parent * L.stride + s0);
        const float in[4] = {pv.x, pv.y, pv.z, pv.w};
        float out[4];
        bool live[4];
        if (copy) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { out[i] = in[i]; live[i] = s0 + i < L.total; }
        } else if (fc1_tiled && s0 >= L.wf && s0 < L.bf) {
            // tiled fc1 block: the quad of lane (cc = lane % 16, kk = lane / 16) of tile (ob, Q, T) holds k = 16 Q + 4 j + kk,
            // j = 0 .. 3, of output 64 ob + 16 T + cc - one element of each of FOUR Philox blocks (k-quads 4 Q + j).  The lane
            // draws the block of k-quad 4 Q + kk instead (whose four normals belong to lanes (cc, 0 .. 3), slot j = kk) and a
            // 4x4 (register x 16-lane row) transpose hands every lane its own four: one Philox block per lane, not four.
            const int64_t i = s0 - L.wf;
            const int lane = (int)((i >> 2) & 63), cc = lane & 15, kk = lane >> 4;
            const int64_t T = (i >> 8) & 3, Q = (i >> 10) % 196, ob = (i >> 10) / 196;
            const int64_t F_wf = 2048LL * C + 69792;
            const int64_t p0 = F_wf + (ob * 64 + 16 * T + cc) * DQ_FC1_IN + 16 * Q + 4 * kk;
            float zl[4], z[4];
            philox_normal4(seed, slo, stream_hi, (uint32_t)(p0 >> 2), zl);
            typedef unsigned u32x2_s __attribute__((ext_vector_type(2)));
            const u32x2_s s02 = __builtin_amdgcn_permlane32_swap(__float_as_uint(zl[0]), __float_as_uint(zl[2]), false, false);
            const u32x2_s s13 = __builtin_amdgcn_permlane32_swap(__float_as_uint(zl[1]), __float_as_uint(zl[3]), false, false);
            const u32x2_s y01 = __builtin_amdgcn_permlane16_swap(s02[0], s13[0], false, false);
            const u32x2_s y23 = __builtin_amdgcn_permlane16_swap(s02[1], s13[1], false, false);
            z[0] = __uint_as_float(y01[0]); z[1] = __uint_as_float(y01[1]); z[2] = __uint_as_float(y23[0]); z[3] = __uint_as_float(y23[1]);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float noise = sigma * z[k];
                out[k] = in[k] + (negate ? -noise : noise);
                live[k] = true;
            }
        } else if (s0 >= L.wf && s0 < L.bf) {
            // fc1 tile (95 % of a net): wfq[ob][kq][l][0..3] = fc1.w[64 ob + l][4 kq .. 4 kq + 3], four consecutive
            // canonical indices = one Philox block; never BatchNorm, never padding
            const int64_t i = s0 - L.wf;
            const int64_t l = (i >> 2) & 63, kk = i >> 8, kq = kk % 784, ob = kk / 784;
            const int64_t F_wf = 2048LL * C + 69792;  // conv1.w conv1.b conv2.w conv2.b conv3.w conv3.b come first
            const int64_t p0 = F_wf + (ob * 64 + l) * DQ_FC1_IN + kq * 4;
            float z[4];
            philox_normal4(seed, slo, stream_hi, (uint32_t)(p0 >> 2), z);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float noise = sigma * z[k];
                out[k] = in[k] + (negate ? -noise : noise);
                live[k] = true;
            }
        } else {
            float zz[4];
            int64_t have = -1;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int64_t s = s0 + k;
                const int64_t p = dqn_slab_to_flat(s, C, n_actions, fc1_tiled);
                live[k] = p >= 0;
                const bool keep = p < 0 || ((flags & DQP_SKIP_BN) && dqn_slab_is_batchnorm(s, L));
                if (keep) { out[k] = in[k]; continue; }
                const int64_t q = p >> 2;
                if (q != have) { philox_normal4(seed, slo, stream_hi, (uint32_t)q, zz); have = q; }
                const float noise = sigma * zz[p & 3];
                out[k] = in[k] + (negate ? -noise : noise);
            }
        }
        if (child_slab)
            *reinterpret_cast<float4 *>(child_slab + (int64_t)(child_first + c) * L.stride + s0) =
                make_float4(out[0], out[1], out[2], out[3]);
        if (dist_partial) {
            const float4 rv = *reinterpret_cast<const float4 *>(dist_ref + s0);
            const float ref[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (live[k]) {
                    const float d = out[k] - ref[k];
                    d2 += (double)d * (double)d;
                }
            }
        }
    }
    if (dist_partial) {
        const double tot = block_sum_f64(d2, scratch);
        if (threadIdx.x == 0) dist_partial[(size_t)c * gridDim.x + blockIdx.x] = tot;
    }
}

__global__ __launch_bounds__(256) void dqn_unpack_kernel(const float *slab, float *flat, int C, int n, int fc1_tiled)
{
    const DqnLayout L = dqn_layout(C, n);
    const int64_t P = dqn_param_count(C, n);
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= L.total) return;
    const int64_t f = dqn_slab_to_flat(s, C, n, fc1_tiled);
    if (f >= 0) flat[(int64_t)blockIdx.y * P + f] = slab[(int64_t)blockIdx.y * L.stride + s];
}

// dst (fc1 layout dst_tiled) := src (fc1 layout src_tiled): everything outside the fc1 block keeps its place
__global__ __launch_bounds__(256) void dqn_relayout_kernel(const float *src, float *dst, int C, int n, int src_tiled, int dst_tiled)
{
    const DqnLayout L = dqn_layout(C, n);
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= L.stride) return;
    const float *sn = src + (int64_t)blockIdx.y * L.stride;
    int64_t from = s;
    if (s >= L.wf && s < L.bf) {
        const int64_t f = dqn_fc1_slab_to_flat(s - L.wf, dst_tiled);
        from = L.wf + dqn_fc1_flat_to_slab(f / DQ_FC1_IN, f % DQ_FC1_IN, src_tiled);
    }
    dst[(int64_t)blockIdx.y * L.stride + s] = sn[from];
}

// ---------------------------------------------------------------------------------------------------------------
// Synthetic two-player env in the shape of pettingzoo.atari (agents first_0, second_0 alternate; uint8 84x84xC frames).
// It has no game dynamics, but it IS sequential: the frame an agent observes at agent-step t is keyed by the game's
// reset ordinal, t and the action taken at step t-1, so step t cannot start before step t-1's forward has finished
// (what makes a rollout a chain of dependent launches, as with a real emulator).  Rewards are zero-sum hits:
// hit(t) = [action_t == target(ordinal, t)]; the AEC bookkeeping is PettingZoo's (_cumulative_rewards of the actor
// reset, then every agent accumulates the step's rewards), and play_atari credits the ACTOR with what env.last()
// returns after env.step - the NEXT agent's cumulative reward (utils/game_logic_functions.py:104-108; the same quirk
// as Q1): credited(t) = hit(t-1) - hit(t).
//   game state [n_games][4] int32: last_action, prev_hit, spare, spare;  acc [n_games][3] fp64: first_0, second_0, 0
struct SynthKey { uint32_t k0, k1; };

__device__ inline uint32_t synth_target(uint64_t seed, int64_t ordinal, int t, int n_actions)
{
    const u32x4 o = philox4x32_10(0xFFFFFFFFu, (uint32_t)t, (uint32_t)ordinal, (uint32_t)((uint64_t)ordinal >> 32) ^ 0x74617267u,
                                  (uint32_t)seed, (uint32_t)(seed >> 32));
    return o.v[0] % (uint32_t)n_actions;
}

// FUSED = the output layer of step t-1 rides in this launch (coevo_dqn_out_synth_step): the workgroup of game g first
// turns the hidden row of its actor (hid[row_prev[g]], left by conv stack + fc1) into logits and the first-max action, then
// books that action and writes the next frame - one launch per agent-step less than output-layer launch + env launch.
template <bool FUSED>
__global__ __launch_bounds__(256) void synth_step_kernel(int32_t *gstate, double *acc, int n_games,
                                                          const int64_t *game_ordinal0, const int32_t *gen_dev,
                                                          int64_t ordinals_per_gen, int t, const int32_t *limit,
                                                          const int32_t *row_prev, int32_t *actions_prev,
                                                          const int32_t *row_cur, uint8_t *frames, int C,
                                                          int n_actions, uint64_t seed, const float *slab,
                                                          const coevo_dqn_task *tasks_prev, int n_tasks_prev,
                                                          const float *hid, int32_t *status)
{
    const int g = blockIdx.x;
    if (g >= n_games) return;
    int64_t ordinal = game_ordinal0[g] + (gen_dev ? (int64_t)(*gen_dev) * ordinals_per_gen : 0);
    const int lim = (ordinal < 0) ? 0 : limit[g];   // a negative ordinal disables the game
    __shared__ int s_last;
    __shared__ __attribute__((aligned(16))) float xs[FUSED ? DQ_FC1_OUT : 4];
    __shared__ float lg[FUSED ? 64 : 1];
    int action = 0;
    if constexpr (FUSED) {
        if (t - 1 < lim) {   // workgroup-uniform; (a finished game's row holds a stale frame: its action is never booked)
            const int row = row_prev[g];
            const coevo_dqn_task task = tasks_prev[task_of_row(tasks_prev, n_tasks_prev, row)];
            action = dqn_out_row(slab + task.net_off, dqn_layout(C, n_actions), n_actions, hid + (size_t)row * DQ_FC1_OUT,
                                 nullptr, status, xs, lg, threadIdx.x);
            if (threadIdx.x == 0) actions_prev[row] = action;
        }
    }
    if (threadIdx.x == 0) {
        int last = gstate[4 * g], prev_hit = gstate[4 * g + 1];
        if (t == 0) {
            last = 0xFF; prev_hit = 0;
            acc[3 * (size_t)g] = 0.0; acc[3 * (size_t)g + 1] = 0.0; acc[3 * (size_t)g + 2] = 0.0;
        } else if (t - 1 < lim) {   // book the action of step t-1 (actor = (t-1) & 1)
            const int a = FUSED ? action : actions_prev[row_prev[g]];
            const int hit = ((uint32_t)a == synth_target(seed, ordinal, t - 1, n_actions)) ? 1 : 0;
            const size_t slot = 3 * (size_t)g + ((t - 1) & 1);
            acc[slot] = acc[slot] + (double)(prev_hit - hit);
            prev_hit = hit;
            last = a;
        }
        gstate[4 * g] = last;
        gstate[4 * g + 1] = prev_hit;
        s_last = last;
    }
    __syncthreads();
    if (!frames || t >= lim) return;
    const uint32_t last = (uint32_t)s_last;
    const int nbytes = 84 * 84 * C;
    uint4 *dst = reinterpret_cast<uint4 *>(frames + (size_t)row_cur[g] * nbytes);
    for (int i = threadIdx.x; i < nbytes / 16; i += 256) {
        const u32x4 o = philox4x32_10((uint32_t)i, (uint32_t)t | (last << 16), (uint32_t)ordinal,
                                      (uint32_t)((uint64_t)ordinal >> 32) ^ 0x66726d65u, (uint32_t)seed,
                                      (uint32_t)(seed >> 32));
        dst[i] = make_uint4(o.v[0], o.v[1], o.v[2], o.v[3]);
    }
}

}  // namespace coevo

using namespace coevo;

static bool dqn_shape_ok2(int C, int n) { return C >= 1 && C <= 6 && n >= 1 && n <= COEVO_DQN_LOGIT_STRIDE; }

static int64_t dqn_perturb_grid(const DqnLayout &L, int tiled) { return (L.stride / 4 + (tiled ? dqn_perturb_shift(L) : 0) + 255) / 256; }
static bool dqn_carg_ok(int c_arg, int n) { return !(c_arg & ~(0xff | COEVO_DQN_FC1_TILED)) && dqn_shape_ok2(dqn_channels(c_arg), n); }

extern "C" int64_t coevo_dqn_perturb_blocks(int c_arg, int n_actions)
{
    return dqn_carg_ok(c_arg, n_actions) ? dqn_perturb_grid(dqn_layout(dqn_channels(c_arg), n_actions), dqn_fc1_tiled(c_arg))
                                        : COEVO_ERR_ARG;
}

extern "C" int coevo_dqn_perturb(const float *parent_slab, const int32_t *parent_idx, float *child_slab,
                                 int child_first, int n_children, int c_arg, int n_actions, const float *sigma_dev,
                                 uint64_t seed, uint32_t stream_lo_first, uint32_t stream_hi, int flags, int E,
                                 const int32_t *gen_dev, int gen_bias, const float *dist_ref, double *dist_partial,
                                 void *stream)
{
    if ((dist_ref == nullptr) != (dist_partial == nullptr)) return COEVO_ERR_ARG;
    if (!dqn_carg_ok(c_arg, n_actions)) return COEVO_ERR_ARG;
    const int C = dqn_channels(c_arg), tiled = dqn_fc1_tiled(c_arg);
    if (!parent_slab || flags < 0 || flags > 15) return COEVO_ERR_ARG;
    if (!child_slab && !dist_partial) return COEVO_ERR_ARG;
    if (!(flags & DQP_COPY) && !sigma_dev) return COEVO_ERR_ARG;
    if ((flags & DQP_FROM_ORDER) && (!parent_idx || E <= 0 || parent_slab == child_slab)) return COEVO_ERR_ARG;
    if (n_children < 0 || child_first < 0 || n_children > 65535) return COEVO_ERR_ARG;
    if (n_children == 0) return COEVO_OK;
    const dim3 grid((unsigned)dqn_perturb_grid(dqn_layout(C, n_actions), tiled), (unsigned)n_children);
    hipLaunchKernelGGL(dqn_perturb_kernel, grid, dim3(256), 0, (hipStream_t)stream, parent_slab, parent_idx, child_slab,
                       child_first, C, n_actions, sigma_dev, seed, stream_lo_first, stream_hi, flags, E, gen_dev, gen_bias,
                       dist_ref, dist_partial, tiled);
    COEVO_HIP_CHECK(hipGetLastError());
    return COEVO_OK;
}

// n nets from one fc1 layout into the other (c_arg_src / c_arg_dst: the channel arguments of the two slabs; same channels).
// A Co-ES engine keeps its nets streamed (one frame per task in the rollout) and plays the ten evaluation games of the updated
// base nets (evolutionary_strategy.py:272) - one 10-frame task per agent-step - on tiled twins of them.
extern "C" int coevo_dqn_relayout(const float *src_slab, float *dst_slab, int n, int c_arg_src, int c_arg_dst, int n_actions,
                                  void *stream)
{
    if (!src_slab || !dst_slab || src_slab == dst_slab || n <= 0 || n > 65535 || !dqn_carg_ok(c_arg_src, n_actions) ||
        !dqn_carg_ok(c_arg_dst, n_actions) || dqn_channels(c_arg_src) != dqn_channels(c_arg_dst))
        return COEVO_ERR_ARG;
    const int C = dqn_channels(c_arg_src);
    const dim3 grid((unsigned)((dqn_layout(C, n_actions).stride + 255) / 256), (unsigned)n);
    hipLaunchKernelGGL(dqn_relayout_kernel, grid, dim3(256), 0, (hipStream_t)stream, src_slab, dst_slab, C, n_actions,
                       dqn_fc1_tiled(c_arg_src), dqn_fc1_tiled(c_arg_dst));
    COEVO_HIP_CHECK(hipGetLastError());
    return COEVO_OK;
}

extern "C" int coevo_dqn_unpack(const float *slab, float *flat, int n, int c_arg, int n_actions, void *stream)
{
    if (!flat || !slab || n <= 0 || !dqn_carg_ok(c_arg, n_actions)) return COEVO_ERR_ARG;
    const int C = dqn_channels(c_arg);
    const dim3 grid((unsigned)((dqn_layout(C, n_actions).total + 255) / 256), (unsigned)n);
    hipLaunchKernelGGL(dqn_unpack_kernel, grid, dim3(256), 0, (hipStream_t)stream, slab, flat, C, n_actions, dqn_fc1_tiled(c_arg));
    COEVO_HIP_CHECK(hipGetLastError());
    return COEVO_OK;
}

extern "C" int coevo_synth_step(int32_t *game_state, double *acc, int n_games, const int64_t *game_ordinal0,
                                const int32_t *gen_dev, int64_t ordinals_per_gen, int t, const int32_t *limit,
                                const int32_t *row_prev, const int32_t *actions_prev, const int32_t *row_cur,
                                uint8_t *frames, int C, int n_actions, uint64_t seed, void *stream)
{
    if (!game_state || !acc || !game_ordinal0 || !limit || n_games <= 0 || t < 0 || t > 65535) return COEVO_ERR_ARG;
    if (!dqn_shape_ok2(C, n_actions)) return COEVO_ERR_ARG;
    if (t > 0 && (!row_prev || !actions_prev)) return COEVO_ERR_ARG;
    if (frames && !row_cur) return COEVO_ERR_ARG;
    hipLaunchKernelGGL(synth_step_kernel<false>, dim3(n_games), dim3(256), 0, (hipStream_t)stream, game_state, acc, n_games,
                       game_ordinal0, gen_dev, ordinals_per_gen, t, limit, row_prev, const_cast<int32_t *>(actions_prev),
                       row_cur, frames, C, n_actions, seed, nullptr, nullptr, 0, nullptr, nullptr);
    COEVO_HIP_CHECK(hipGetLastError());
    return COEVO_OK;
}

extern "C" int coevo_dqn_out_synth_step(int32_t *game_state, double *acc, int n_games, const int64_t *game_ordinal0,
                                        const int32_t *gen_dev, int64_t ordinals_per_gen, int t, const int32_t *limit,
                                        const int32_t *row_prev, int32_t *actions_prev, const int32_t *row_cur,
                                        uint8_t *frames, int C, int n_actions, uint64_t seed, const float *slab,
                                        const coevo_dqn_task *tasks_prev, int n_tasks_prev, int n_rows_total,
                                        const void *workspace, int32_t *status, void *stream)
{
    if (!game_state || !acc || !game_ordinal0 || !limit || n_games <= 0 || t < 1 || t > 65535) return COEVO_ERR_ARG;
    if (!dqn_shape_ok2(C, n_actions) || !row_prev || !actions_prev || (frames && !row_cur)) return COEVO_ERR_ARG;
    if (!slab || !tasks_prev || n_tasks_prev <= 0 || n_rows_total <= 0 || !workspace || !status) return COEVO_ERR_ARG;
    const float *hid = static_cast<const float *>(workspace) + (size_t)n_rows_total * DQ_FC1_IN;   // (deepqn.hip's layout)
    hipLaunchKernelGGL(synth_step_kernel<true>, dim3(n_games), dim3(256), 0, (hipStream_t)stream, game_state, acc, n_games,
                       game_ordinal0, gen_dev, ordinals_per_gen, t, limit, row_prev, actions_prev, row_cur, frames, C,
                       n_actions, seed, slab, tasks_prev, n_tasks_prev, hid, status);
    COEVO_HIP_CHECK(hipGetLastError());
    return COEVO_OK;
}

COEVO_DEFINE_TU_FLAGS(dqn_engine)
