// DeepQN slab layout shared by the forward kernels (deepqn.hip) and the population engine (dqn_engine.hip).
#pragma once
#include "coevo_common.hip.h"

namespace coevo {

constexpr int DQ_FC1_IN = 3136, DQ_FC1_OUT = 512;

// the channel argument of the layout-dependent entry points: channels | COEVO_DQN_FC1_TILED
__host__ __device__ inline int dqn_channels(int c_arg) { return c_arg & 0xff; }
__host__ __device__ inline int dqn_fc1_tiled(int c_arg) { return (c_arg & COEVO_DQN_FC1_TILED) ? 1 : 0; }

struct DqnLayout {
    int64_t w1, b1, w2, b2, w3, b3, wf, bf, wo, bo, total, stride;
};

__host__ __device__ inline DqnLayout dqn_layout(int C, int n)
{
    DqnLayout L;
    L.w1 = 0;
    L.b1 = (int64_t)C * 64 * 32;          // b1, g1, be1 (32 each)
    L.w2 = L.b1 + 96;
    L.b2 = L.w2 + 512 * 64;               // b2, g2, be2 (64 each)
    L.w3 = L.b2 + 192;
    L.b3 = L.w3 + 576 * 64;
    L.wf = L.b3 + 192;                     // [8][784][64][4]
    L.bf = L.wf + (int64_t)DQ_FC1_OUT * DQ_FC1_IN;
    L.wo = L.bf + DQ_FC1_OUT;              // [n][512]
    L.bo = L.wo + (int64_t)n * DQ_FC1_OUT;
    L.total = L.bo + n;
    L.stride = (L.total + 63) / 64 * 64;
    return L;
}

// quads by which dqn_perturb_kernel shifts its thread -> slab mapping for the tiled fc1 block (see there): the block's first
// quad then falls on lane 0 of a wave
__host__ __device__ inline int dqn_perturb_shift(const DqnLayout &L) { return (int)((64 - ((L.wf >> 2) & 63)) & 63); }

__host__ __device__ inline int64_t dqn_param_count(int C, int n)
{
    return 32LL * C * 64 + 32 + 64LL * 512 + 64 + 64LL * 576 + 64 + 512LL * DQ_FC1_IN + 512 + 512LL * n + n + 320;
}

// Conv weights are stored in the order the implicit-GEMM kernel's lanes consume them (deepqn.hip conv16_mfma): position
// i = ((qp * NP + np) * 64 + lane) * 4 + e holds W[co = 32 np + 16 (e & 1) + lane % 16][tap = 4 (2 qp + e / 2) + lane / 16]
// (NP = COUT / 32, taps in the canonical (ci, ky, kx) order), so one 16-byte load per lane brings the B operands of two
// k-steps x two channel tiles and a wave reads 1 KB contiguously.  (As [tap][cout] every operand was its own dword
// load: the texture addresser spends 16 cycles on a wave's load whatever its width, and conv3 - two loads per two
// MFMAs - ran at its pace, not the matrix pipe's.)  Returns co * taps + tap.
__host__ __device__ inline int64_t dqn_conv_slab_to_flat(int64_t i, int cout, int64_t taps)
{
    const int64_t np_count = cout / 32;
    const int64_t e = i & 3, lane = (i >> 2) & 63, blk = i >> 8, np = blk % np_count, qp = blk / np_count;
    const int64_t co = 32 * np + 16 * (e & 1) + (lane & 15), tap = 4 * (2 * qp + (e >> 1)) + (lane >> 4);
    return co * taps + tap;
}

// The fc1 block has two layouts, an attribute of the engine that owns the slab (every entry point that depends on it takes it
// as COEVO_DQN_FC1_TILED or-ed into its channel argument, include/coevo.h):
//   streamed (0)  wfq[ob][kq][l][c] = fc1.w[64 ob + l][4 kq + c]: lane l of a wave owns output 64 ob + l and streams its row as
//                 16-byte pieces - the B operand of v_mfma_f32_4x4x1 as it is (rows in groups of four; Co-ES: one row per task)
//   tiled (1)     wft[ob][Q][T][l][j] = fc1.w[64 ob + 16 T + l % 16][16 Q + 4 j + l / 16]: the 16-byte piece of lane (c = l % 16,
//                 kk = l / 16) holds, for tile T and the four k-quads 4 Q + j, exactly the B operands of v_mfma_f32_16x16x4 -
//                 B[k = 4 q + kk][column c] - so a 16-row task runs sixteen matrix instructions per 16 k with no vector
//                 instruction touching an operand (Co-GA: 10 / 16 rows per task; profiles/r04_experiments.md section 8)
__host__ __device__ inline int64_t dqn_fc1_slab_to_flat(int64_t i, int tiled)   // i: position inside the block -> out * 3136 + k
{
    if (tiled) {
        const int64_t j = i & 3, l = (i >> 2) & 63, T = (i >> 8) & 3, Q = (i >> 10) % 196, ob = (i >> 10) / 196;
        return (ob * 64 + 16 * T + (l & 15)) * DQ_FC1_IN + 16 * Q + 4 * j + (l >> 4);
    }
    const int64_t c = i & 3, l = (i >> 2) & 63, kq = (i >> 8) % 784, ob = (i >> 8) / 784;
    return (ob * 64 + l) * DQ_FC1_IN + kq * 4 + c;
}

// ... and back: fc1.w[out][k] -> position inside the block
__host__ __device__ inline int64_t dqn_fc1_flat_to_slab(int64_t out, int64_t k, int tiled)
{
    const int64_t ob = out >> 6, o = out & 63;
    if (tiled) return ((((ob * 196 + (k >> 4)) * 4 + (o >> 4)) * 64 + 16 * (k & 3) + (o & 15)) << 2) + ((k >> 2) & 3);
    return (((ob * 784 + (k >> 2)) * 64 + o) << 2) + (k & 3);
}

// slab position -> canonical flat index (parameters() order: conv1.w conv1.b conv2.w conv2.b conv3.w conv3.b fc1.w
// fc1.b output.w output.b vbn1.w vbn1.b vbn2.w vbn2.b vbn3.w vbn3.b); -1 for padding
__host__ __device__ inline int64_t dqn_slab_to_flat(int64_t s, int C, int n, int fc1_tiled = 0)
{
    const DqnLayout L = dqn_layout(C, n);
    const int64_t T1 = (int64_t)C * 64;
    const int64_t F_w1 = 0, F_b1 = 32 * T1, F_w2 = F_b1 + 32, F_b2 = F_w2 + 64 * 512, F_w3 = F_b2 + 64,
                  F_b3 = F_w3 + 64 * 576, F_wf = F_b3 + 64, F_bf = F_wf + 512LL * DQ_FC1_IN, F_wo = F_bf + 512,
                  F_bo = F_wo + 512LL * n, F_g1 = F_bo + n, F_be1 = F_g1 + 32, F_g2 = F_be1 + 32, F_be2 = F_g2 + 64,
                  F_g3 = F_be2 + 64, F_be3 = F_g3 + 64;
    if (s >= L.total) return -1;
    if (s < L.b1) return F_w1 + dqn_conv_slab_to_flat(s, 32, T1);
    if (s < L.w2) { const int64_t i = s - L.b1; return i < 32 ? F_b1 + i : (i < 64 ? F_g1 + i - 32 : F_be1 + i - 64); }
    if (s < L.b2) return F_w2 + dqn_conv_slab_to_flat(s - L.w2, 64, 512);
    if (s < L.w3) { const int64_t i = s - L.b2; return i < 64 ? F_b2 + i : (i < 128 ? F_g2 + i - 64 : F_be2 + i - 128); }
    if (s < L.b3) return F_w3 + dqn_conv_slab_to_flat(s - L.w3, 64, 576);
    if (s < L.wf) { const int64_t i = s - L.b3; return i < 64 ? F_b3 + i : (i < 128 ? F_g3 + i - 64 : F_be3 + i - 128); }
    if (s < L.bf) return F_wf + dqn_fc1_slab_to_flat(s - L.wf, fc1_tiled);
    if (s < L.wo) return F_bf + (s - L.bf);
    if (s < L.bo) return F_wo + (s - L.wo);
    return F_bo + (s - L.bo);
}

// is slab position s a BatchNorm affine parameter (vbn*.weight / vbn*.bias)?  ES leaves them untouched
// (Atari/deepqn.py:158-171: get_perturbable_layers skips nn.BatchNorm2d); GA mutates them (agent.py:27)
__host__ __device__ inline bool dqn_slab_is_batchnorm(int64_t s, const DqnLayout &L)
{
    return (s >= L.b1 + 32 && s < L.w2) || (s >= L.b2 + 64 && s < L.w3) || (s >= L.b3 + 64 && s < L.wf);
}

// the task that holds row `row`: the tasks partition the rows in ascending row_begin order (include/coevo.h)
__device__ __forceinline__ int task_of_row(const coevo_dqn_task *tasks, int n_tasks, int row)
{
    int lo = 0, hi = n_tasks - 1;
    while (lo < hi) {   // workgroup-uniform: scalar loads
        const int mid = (lo + hi + 1) >> 1;
        if (tasks[mid].row_begin <= row) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// Output layer (512 -> n logits) + first-max action of one row, by the first wavefront of the calling workgroup (lane =
// threadIdx.x < 64; the other threads only take part in the barriers).  Lane o < n_actions runs the canonical sequential-k
// chain of its logit; the hidden row is staged in LDS once (every lane reads the same element: a broadcast) and the
// lane's weight row comes in 16-byte pieces, 16 of them in flight (as a dword-at-a-time loop this took as long as a
// tenth of the conv stack).  Returns the action to every thread.  xs: 512 floats, 16-byte aligned; lg: 64 floats.
__device__ __forceinline__ int dqn_out_row(const float *net, const DqnLayout L, int n_actions, const float *hid_row,
                                           float *logits_row, int32_t *status, float *xs, float *lg, int tid)
{
    if (tid < 64) {
        const float4 *x4 = reinterpret_cast<const float4 *>(hid_row);
        reinterpret_cast<float4 *>(xs)[tid] = x4[tid];
        reinterpret_cast<float4 *>(xs)[tid + 64] = x4[tid + 64];
    }
    __syncthreads();
    if (tid < n_actions) {
        float y = net[L.bo + tid];
        const float4 *w4 = reinterpret_cast<const float4 *>(net + L.wo + (size_t)tid * DQ_FC1_OUT);
        constexpr int B = 16;
        for (int k0 = 0; k0 < DQ_FC1_OUT / 4; k0 += B) {
            float4 wv[B];
#pragma unroll
            for (int i = 0; i < B; ++i) wv[i] = w4[k0 + i];
#pragma unroll
            for (int i = 0; i < B; ++i) {
                const float4 xv = reinterpret_cast<const float4 *>(xs)[k0 + i];
                y = __builtin_fmaf(wv[i].x, xv.x, y);
                y = __builtin_fmaf(wv[i].y, xv.y, y);
                y = __builtin_fmaf(wv[i].z, xv.z, y);
                y = __builtin_fmaf(wv[i].w, xv.w, y);
            }
        }
        lg[tid] = y;
        if (logits_row) logits_row[tid] = y;
    }
    __syncthreads();
    if (tid == 0) {
        int best = -1;
        float cur = -__builtin_inff();
        for (int i = 0; i < n_actions; ++i)
            if (lg[i] > cur) { cur = lg[i]; best = i; }
        if (best < 0) { atomicOr(status, COEVO_ST_NO_ACTION); best = 0; }
        lg[63] = __int_as_float(best);   // (n_actions <= 32: slot 63 is free)
    }
    __syncthreads();
    return __float_as_int(lg[63]);
}

}  // namespace coevo
