// Shared device helpers for libcoevo (gfx950 only).
//
// Canonical fp32 arithmetic (the contract with oracle/coevo_oracle.c, bit for bit):
//   Linear     acc = bias[j]; for k = 0..K-1: acc = fmaf(W[j][k], x[k], acc)
//   Reduce(N)  blocks of 64 consecutive indices, balanced tree inside a block with adjacent pairs first
//              (lane xor 1,2,4,8,16,32 on one 64-wide wavefront), block sums added left to right
//   LayerNorm  mean = Reduce(x)/N; d = x-mean; var = Reduce(d*d)/N; rstd = 1/sqrtf(var+1e-5f);
//              y = fmaf(d*rstd, gamma, beta);  ReLU y>0?y:0
// The library is compiled with -ffp-contract=off: only the explicit __builtin_fmaf calls fuse.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/coevo.h"

namespace coevo {

constexpr int H1 = COEVO_FC_H1;      // 512
constexpr int H2 = COEVO_FC_H2;      // 256
constexpr int NACT = COEVO_FC_NACT;  // 5
constexpr float LN_EPS = 1e-5f;

// ---- device slab layout of one FCNetwork (floats), D = observation width (10 good / 8 adversary) -------------
//   W1t [D][512]            fc1.weight transposed: a wavefront reads 64 consecutive outputs of one input k
//   b1, g1, be1 [512] each  fc1.bias, ln1.weight, ln1.bias
//   W2q [4][128][64][4]     fc2.weight tiled: [output block of 64][k/4][output in block][k%4]; lane l of wave w
//                           streams its own output row 64w+l as consecutive 16-byte pieces, a wavefront reads
//                           1 KiB contiguous per instruction and one wave's whole stream (128 KiB) is contiguous
//   b2, g2, be2 [256] each
//   W3 [5][256], b3 [5]     output layer, row-major
// The stride between nets is padded to a multiple of 64 floats so every net (and W2q) is 256-byte aligned.
__host__ __device__ constexpr int64_t fc_off_b1(int D) { return (int64_t)D * H1; }
__host__ __device__ constexpr int64_t fc_off_w2(int D) { return fc_off_b1(D) + 3 * H1; }
__host__ __device__ constexpr int64_t fc_off_b2(int D) { return fc_off_w2(D) + (int64_t)H1 * H2; }
__host__ __device__ constexpr int64_t fc_off_w3(int D) { return fc_off_b2(D) + 3 * H2; }
__host__ __device__ constexpr int64_t fc_off_b3(int D) { return fc_off_w3(D) + NACT * H2; }
__host__ __device__ constexpr int64_t fc_params(int D) { return fc_off_b3(D) + NACT; }
__host__ __device__ constexpr int64_t fc_stride(int D) { return (fc_params(D) + 63) / 64 * 64; }

// slab position -> canonical flat index (torch parameters() order: fc1.w[512][D], fc1.b, ln1.w, ln1.b,
// fc2.w[256][512], fc2.b, ln2.w, ln2.b, output.w[5][256], output.b).  Only W1 and W2 are re-tiled.
__host__ __device__ inline int64_t fc_slab_to_flat(int64_t s, int D)
{
    const int64_t o_b1 = fc_off_b1(D), o_w2 = fc_off_w2(D), o_b2 = fc_off_b2(D);
    if (s < o_b1) {  // W1t[k][j] -> fc1.w[j][k]
        int64_t k = s / H1, j = s % H1;
        return j * D + k;
    }
    if (s < o_w2) return s;  // b1, g1, be1 keep their place
    if (s < o_b2) {          // W2q[jb][kq][l][c] -> fc2.w[64*jb + l][4*kq + c]
        int64_t t = s - o_w2;
        int64_t c = t & 3, l = (t >> 2) & 63, kq = (t >> 8) & 127, jb = t >> 15;
        return o_w2 + (jb * 64 + l) * H1 + kq * 4 + c;
    }
    return s;
}

// ---- canonical tree sums on one 64-wide wavefront -----------------------------------------------------------
// The tree is: adjacent pairs first (lane xor 1), then xor 2, 4, 8, 16, 32.  After level m every group of 2m lanes
// holds one value, so any lane permutation that maps each group onto its sibling serves the next level: DPP
// quad_perm for xor 1 / 2, row_half_mirror for 4, row_mirror for 8 (all in the VALU, no LDS crossbar), ds_swizzle
// for 16, v_permlane32_swap for 32.  fp32 addition is commutative, so both partners get the same bits.
template <int CTRL>
__device__ inline float dpp_move(float v)
{
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, true));
}

// levels 1..16: every lane ends with the tree sum of its 32-lane half
// the first four levels only: every lane returns the sum over its 16-lane row (DPP row operations, no cross-row step)
__device__ inline float row16_tree_sum(float v)
{
    v = v + dpp_move<0xB1>(v);   // quad_perm [1,0,3,2]
    v = v + dpp_move<0x4E>(v);   // quad_perm [2,3,0,1]
    v = v + dpp_move<0x141>(v);  // row_half_mirror
    v = v + dpp_move<0x140>(v);  // row_mirror
    return v;
}

__device__ inline float half_tree_sum(float v)
{
    v = v + dpp_move<0xB1>(v);   // quad_perm [1,0,3,2]
    v = v + dpp_move<0x4E>(v);   // quad_perm [2,3,0,1]
    v = v + dpp_move<0x141>(v);  // row_half_mirror
    v = v + dpp_move<0x140>(v);  // row_mirror
    // xor 16: v_permlane16_swap(v, v) leaves {row0,row0,row2,row2} and {row1,row1,row3,row3} (rows of 16 lanes); their
    // sum is the pairwise sum in every lane (fp add commutes, so the bits equal v + v[lane ^ 16]) - no LDS round trip
    typedef unsigned u32x2_h __attribute__((ext_vector_type(2)));
    const u32x2_h s = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(s[0]) + __uint_as_float(s[1]);
}

// all six levels: every lane returns the block (64-lane) sum
__device__ inline float wave_tree_sum(float v)
{
    v = half_tree_sum(v);
    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
    const u32x2_t s = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(s[0]) + __uint_as_float(s[1]);
}

// ---- packed butterfly: the canonical Reduce for a ROW held by ONE wave ----------------------------------------------
// true lane-xor exchanges (the packed butterfly needs partners that agree in the lower lane bits)
__device__ __forceinline__ float lane_xor4(float v)
{
    int x = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x104, 0xf, 0x5, false);   // row_shl:4 -> lanes 0-3, 8-11
    x = __builtin_amdgcn_update_dpp(x, __float_as_int(v), 0x114, 0xf, 0xa, false);       // row_shr:4 -> lanes 4-7, 12-15
    return __int_as_float(x);
}
__device__ __forceinline__ float add_xor16(float v)
{
    typedef unsigned u32x2_h __attribute__((ext_vector_type(2)));
    const u32x2_h s = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(s[0]) + __uint_as_float(s[1]);
}
__device__ __forceinline__ float add_xor32(float v)
{
    typedef unsigned u32x2_h __attribute__((ext_vector_type(2)));
    const u32x2_h s = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(s[0]) + __uint_as_float(s[1]);
}

// N independent 64-lane tree sums (lane xor 1, 2, 4, 8, 16, 32 - the canonical in-block tree) in one packed butterfly:
// after level m the 2m lanes of a group hold the same value, so two values share a register from there on (a lane bit
// selects which).  Returns a register whose lane j (j < N) holds the total of v[j].  N <= 16.  ~4 instructions per value
// instead of 11.
template <int N>
__device__ __forceinline__ float packed_totals(const float (&v)[N], int l)
{
    static_assert(N >= 1 && N <= 16, "four packing levels");
    constexpr int N1 = (N + 1) / 2, N2 = (N1 + 1) / 2, N3 = (N2 + 1) / 2;
    float p[N1];
#pragma unroll
    for (int j = 0; j < N1; ++j) {
        const float a = v[2 * j] + dpp_move<0xB1>(v[2 * j]);                       // xor 1
        if (2 * j + 1 < N) {
            const float b = v[2 * j + 1] + dpp_move<0xB1>(v[2 * j + 1]);
            p[j] = (l & 1) ? b : a;
        } else {
            p[j] = a;
        }
    }
    float q[N2];
#pragma unroll
    for (int j = 0; j < N2; ++j) {
        const float a = p[2 * j] + dpp_move<0x4E>(p[2 * j]);                       // xor 2
        if (2 * j + 1 < N1) {
            const float b = p[2 * j + 1] + dpp_move<0x4E>(p[2 * j + 1]);
            q[j] = (l & 2) ? b : a;
        } else {
            q[j] = a;
        }
    }
    float s3[N3];
#pragma unroll
    for (int j = 0; j < N3; ++j) {
        const float a = q[2 * j] + lane_xor4(q[2 * j]);                            // xor 4
        if (2 * j + 1 < N2) {
            const float b = q[2 * j + 1] + lane_xor4(q[2 * j + 1]);
            s3[j] = (l & 4) ? b : a;
        } else {
            s3[j] = a;
        }
    }
    float s = s3[0] + dpp_move<0x128>(s3[0]);                                      // xor 8: row_ror:8
    if constexpr (N3 > 1) {
        const float b = s3[1] + dpp_move<0x128>(s3[1]);
        s = (l & 8) ? b : s;
    }
    s = add_xor16(s);
    s = add_xor32(s);
    return s;
}

// Reduce(NB * 64) for one row held by one wave: v[b] = the lane's element of block b (lane l = position l of the block):
// the block trees by one packed butterfly, then the block sums left to right.  NB = 1..8.
template <int NB>
__device__ __forceinline__ float row_blocks_total(const float (&v)[NB], int l)
{
    static_assert(NB >= 1 && NB <= 8, "up to eight 64-wide blocks");
    const float s = packed_totals<NB>(v, l);
    float tot = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s), 0));
#pragma unroll
    for (int b = 1; b < NB; ++b) tot = tot + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s), b));
    return tot;
}

// ---- MPE observation element k of env slot `slot` from the fp64 struct-of-arrays game state -------------------
// (PettingZoo simple_adversary.observation + SimpleEnv.observe's float32 cast; field indices in mpe_env.hip)
__host__ __device__ inline float mpe_obs_element(const double *st, int n, int g, int slot, int k)
{
    const int c = k & 1, e = k >> 1;
    int src;  // fp64 field the agent's own position is subtracted from
    if (slot == COEVO_SLOT_ADVERSARY) {
        src = (e < 2) ? (12 + 2 * e) : (2 * (e - 1));  // lm0, lm1, agent_0, agent_1
    } else {
        if (e == 0) src = 16;                           // goal
        else if (e < 3) src = 12 + 2 * (e - 1);         // lm0, lm1
        else if (e == 3) src = 0;                       // adversary
        else src = 2 * (slot == COEVO_SLOT_AGENT_0 ? 2 : 1);  // the other good agent
    }
    return (float)(st[(size_t)(src + c) * n + g] - st[(size_t)(2 * slot + c) * n + g]);
}

// ---- one game in registers: the world step and the observation, same operation order as everywhere else ---------
// Named scalars, not arrays: hipcc turns a select over array elements back into a runtime-indexed load from a scratch
// copy of the array.
struct MpeGame {
    double ax, ay, bx, by, cx, cy;        // positions of adversary_0 (a), agent_0 (b), agent_1 (c)
    double avx, avy, bvx, bvy, cvx, cvy;  // velocities
    double l0x, l0y, l1x, l1y;            // landmarks
    double gx, gy;                        // goal landmark
};

__host__ __device__ __forceinline__ void mpe_load_game(const double *st, int n, int g, MpeGame &s)
{
    const size_t N = (size_t)n;
    s.ax = st[0 * N + g]; s.ay = st[1 * N + g];
    s.bx = st[2 * N + g]; s.by = st[3 * N + g];
    s.cx = st[4 * N + g]; s.cy = st[5 * N + g];
    s.avx = st[6 * N + g]; s.avy = st[7 * N + g];
    s.bvx = st[8 * N + g]; s.bvy = st[9 * N + g];
    s.cvx = st[10 * N + g]; s.cvy = st[11 * N + g];
    s.l0x = st[12 * N + g]; s.l0y = st[13 * N + g];
    s.l1x = st[14 * N + g]; s.l1y = st[15 * N + g];
    s.gx = st[16 * N + g]; s.gy = st[17 * N + g];
}

__device__ __forceinline__ void mpe_store_game(double *st, int n, int g, const MpeGame &s)
{
    const size_t N = (size_t)n;
    st[0 * N + g] = s.ax; st[1 * N + g] = s.ay;
    st[2 * N + g] = s.bx; st[3 * N + g] = s.by;
    st[4 * N + g] = s.cx; st[5 * N + g] = s.cy;
    st[6 * N + g] = s.avx; st[7 * N + g] = s.avy;
    st[8 * N + g] = s.bvx; st[9 * N + g] = s.bvy;
    st[10 * N + g] = s.cvx; st[11 * N + g] = s.cvy;
    st[12 * N + g] = s.l0x; st[13 * N + g] = s.l0y;
    st[14 * N + g] = s.l1x; st[15 * N + g] = s.l1y;
    st[16 * N + g] = s.gx; st[17 * N + g] = s.gy;
}

// one agent's integration step for one coordinate (discrete action -> force 5*u, dt 0.1, damping 0.25)
__device__ __forceinline__ void mpe_move(double &pos, double &vel, double u, int pos_first)
{
    const double f = (((u * 5.0) + 0.0) / 1.0) * 0.1;
    if (pos_first) pos = pos + vel * 0.1;
    vel = vel * 0.75;
    vel = vel + f;
    if (!pos_first) pos = pos + vel * 0.1;
}

__device__ __forceinline__ double mpe_ux(int act) { return act == 1 ? -1.0 : (act == 2 ? 1.0 : 0.0); }
__device__ __forceinline__ double mpe_uy(int act) { return act == 3 ? -1.0 : (act == 4 ? 1.0 : 0.0); }

// PettingZoo World.step for three discrete actions (no collisions, no noise) + the two reward values
// the movement alone (rows that are not their game's owner do not need the rewards: three fp64 square roots less)
__device__ __forceinline__ void mpe_world_move(MpeGame &s, int act_a, int act_b, int act_c, int pos_first)
{
    mpe_move(s.ax, s.avx, mpe_ux(act_a), pos_first);
    mpe_move(s.ay, s.avy, mpe_uy(act_a), pos_first);
    mpe_move(s.bx, s.bvx, mpe_ux(act_b), pos_first);
    mpe_move(s.by, s.bvy, mpe_uy(act_b), pos_first);
    mpe_move(s.cx, s.cvx, mpe_ux(act_c), pos_first);
    mpe_move(s.cy, s.cvy, mpe_uy(act_c), pos_first);
}

__device__ __forceinline__ void mpe_world_step(MpeGame &s, int act_a, int act_b, int act_c, int pos_first,
                                               double &r_good, double &r_adv)
{
    mpe_world_move(s, act_a, act_b, act_c, pos_first);
    double dx = s.ax - s.gx, dy = s.ay - s.gy;
    const double da = sqrt(dx * dx + dy * dy);
    dx = s.bx - s.gx; dy = s.by - s.gy;
    const double db = sqrt(dx * dx + dy * dy);
    dx = s.cx - s.gx; dy = s.cy - s.gy;
    const double dc = sqrt(dx * dx + dy * dy);
    r_adv = -da;
    const double m = (dc < db) ? dc : db;
    r_good = -m + da;
}

// the whole observation of env slot `slot` (float32 casts of fp64 differences, as SimpleEnv.observe does):
//   adversary: [lm0-p, lm1-p, agent_0-p, agent_1-p]           good: [goal-p, lm0-p, lm1-p, adversary-p, other good-p]
// Three branches with fixed fields rather than selects on `slot`: hipcc turns a select chain over the position values
// into a runtime-indexed load from a scratch copy.  Rows of one task almost always share a slot (wave-uniform branch).
__host__ __device__ __forceinline__ void mpe_obs_good(const MpeGame &s, double mex, double mey, double ox, double oy, float o[10])
{
    o[0] = (float)(s.gx - mex);  o[1] = (float)(s.gy - mey);
    o[2] = (float)(s.l0x - mex); o[3] = (float)(s.l0y - mey);
    o[4] = (float)(s.l1x - mex); o[5] = (float)(s.l1y - mey);
    o[6] = (float)(s.ax - mex);  o[7] = (float)(s.ay - mey);
    o[8] = (float)(ox - mex);    o[9] = (float)(oy - mey);
}

__host__ __device__ __forceinline__ void mpe_obs_from_game(const MpeGame &s, int slot, float o[10])
{
    if (slot == COEVO_SLOT_ADVERSARY) {
        o[0] = (float)(s.l0x - s.ax); o[1] = (float)(s.l0y - s.ay);
        o[2] = (float)(s.l1x - s.ax); o[3] = (float)(s.l1y - s.ay);
        o[4] = (float)(s.bx - s.ax);  o[5] = (float)(s.by - s.ay);
        o[6] = (float)(s.cx - s.ax);  o[7] = (float)(s.cy - s.ay);
        o[8] = 0.0f;                  o[9] = 0.0f;
    } else if (slot == COEVO_SLOT_AGENT_0) {
        mpe_obs_good(s, s.bx, s.by, s.cx, s.cy, o);
    } else {
        mpe_obs_good(s, s.cx, s.cy, s.bx, s.by, o);
    }
}

// Fused env step (coevo_mpe_rollout): the state a policy launch of cycle `cycle` observes is derived in registers
// from the previous cycle's state buffer and the previous cycle's actions; only the row in the adversary's seat (one
// per game) writes the new state and credits the rewards (quirk Q1), into the OTHER buffer.  obs_out[0..D) filled.
__device__ __forceinline__ void mpe_fused_observe(const double *st_prev, double *st_next, const int32_t *act_prev,
                                         const int32_t *game_limit, int n, int g, int slot, int cycle, int pos_first,
                                         float *obs_out)
{
    const size_t N = (size_t)n;
    MpeGame s;
    mpe_load_game(st_prev, n, g, s);
    if (cycle > 0) {
        const int limit = game_limit ? game_limit[g] : 0x7fffffff;
        const int t0 = 3 * (cycle - 1);
        const bool stepped = t0 + 2 < limit;  // agent_1 acted in the previous cycle: the world moved
        double r_good = 0.0, r_adv = 0.0;
        if (stepped) {
            const int a0 = act_prev[3 * g], a1 = act_prev[3 * g + 1], a2 = act_prev[3 * g + 2];
            if (slot == COEVO_SLOT_ADVERSARY) mpe_world_step(s, a0, a1, a2, pos_first, r_good, r_adv);
            else mpe_world_move(s, a0, a1, a2, pos_first);  // only the owner row credits rewards
        }
        if (slot == COEVO_SLOT_ADVERSARY) {  // the game's owner row: carry the bookkeeping into the new buffer
            double rg_prev = st_prev[18 * N + g], a_adv = st_prev[19 * N + g], a_a0 = st_prev[20 * N + g],
                   a_a1 = st_prev[21 * N + g];
            if (t0 < limit) a_adv = a_adv + rg_prev;
            if (t0 + 1 < limit) a_a0 = a_a0 + rg_prev;
            if (stepped) { a_a1 = a_a1 + r_adv; rg_prev = r_good; }
            mpe_store_game(st_next, n, g, s);
            st_next[18 * N + g] = rg_prev;
            st_next[19 * N + g] = a_adv;
            st_next[20 * N + g] = a_a0;
            st_next[21 * N + g] = a_a1;
        }
    }
    mpe_obs_from_game(s, slot, obs_out);
}

// fixed-order block reduction of one double per thread (256 threads): xor tree inside each wave, waves left to right
__device__ inline double block_sum_f64(double v, double *scratch)
{
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v = v + __shfl_xor(v, m, 64);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) scratch[w] = v;
    __syncthreads();
    const double tot = ((scratch[0] + scratch[1]) + scratch[2]) + scratch[3];
    __syncthreads();
    return tot;
}

__device__ inline bool bad_post_relu(float y) { return __builtin_isnan(y) || (__builtin_isinf(y) && y > 0.0f); }
// one compare: !(y <= 0) is true for y > 0 AND for NaN (unordered), so NaN passes through and -0 / negatives give +0
__device__ inline float relu_keep_nan(float y) { return !(y <= 0.0f) ? y : 0.0f; }

}  // namespace coevo

// Build switches (COEVO_EXTRA_FLAGS / tools/build_variant.sh) exist for A/B measurements only.  The build passes them to every
// translation unit it compiles under them as COEVO_TU_FLAGS; coevo_build_flags() reports the union, and
// coevonet_amd.lib.load() refuses a library with a non-empty answer unless COEVO_ALLOW_VARIANT=1 (tools/ set it).
#ifndef COEVO_TU_FLAGS
#define COEVO_TU_FLAGS ""
#endif
#define COEVO_DEFINE_TU_FLAGS(tu) \
    namespace coevo { const char *tu_flags_##tu() { return COEVO_TU_FLAGS; } }

#define COEVO_HIP_CHECK(expr)                        \
    do {                                             \
        hipError_t _e = (expr);                      \
        if (_e != hipSuccess) return COEVO_ERR_HIP;  \
    } while (0)
