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
__device__ inline float half_tree_sum(float v)
{
    v = v + dpp_move<0xB1>(v);   // quad_perm [1,0,3,2]
    v = v + dpp_move<0x4E>(v);   // quad_perm [2,3,0,1]
    v = v + dpp_move<0x141>(v);  // row_half_mirror
    v = v + dpp_move<0x140>(v);  // row_mirror
    v = v + __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x401F));  // xor 16 within 32 lanes
    return v;
}

// all six levels: every lane returns the block (64-lane) sum
__device__ inline float wave_tree_sum(float v)
{
    v = half_tree_sum(v);
    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
    const u32x2_t s = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(s[0]) + __uint_as_float(s[1]);
}

// ---- MPE observation element k of env slot `slot` from the fp64 struct-of-arrays game state -------------------
// (PettingZoo simple_adversary.observation + SimpleEnv.observe's float32 cast; field indices in mpe_env.hip)
__device__ inline float mpe_obs_element(const double *st, int n, int g, int slot, int k)
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
__device__ inline float relu_keep_nan(float y) { return (y > 0.0f) ? y : (__builtin_isnan(y) ? y : 0.0f); }

}  // namespace coevo

#define COEVO_HIP_CHECK(expr)                        \
    do {                                             \
        hipError_t _e = (expr);                      \
        if (_e != hipSuccess) return COEVO_ERR_HIP;  \
    } while (0)
