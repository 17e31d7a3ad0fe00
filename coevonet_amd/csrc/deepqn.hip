// K2 - DeepQN policy step (reference Atari/deepqn.py:39-48: x/255 -> conv8s4(32) -> BN -> ReLU -> conv4s2(64) -> BN ->
// ReLU -> conv3s1(64) -> BN -> ReLU -> flatten(CHW) -> fc 3136->512 -> ReLU -> fc 512->n) + first-max action, for many
// (weight set x frames) tasks.  BatchNorm is in TRAINING mode at batch 1 in the reference (never .eval()ed), i.e.
// per-sample, per-channel statistics over the spatial positions: batching must not mix samples.
//
// Three launches per env step:
//   dqn_conv_kernel   one workgroup per frame: the uint8 HWC frame is staged in LDS once (coalesced 16-byte loads),
//                     /255 through a 256-entry LDS table (exact fp32 quotients), the three convolutions run as implicit
//                     GEMMs on v_mfma_f32_32x32x2_f32 (A = im2col gather out of LDS, B = weights [tap][cout] from L2),
//                     BN statistics are reduced in the accumulator layout in the canonical tree order, activations stay
//                     in LDS between layers; conv3's output goes to HBM as act[row][3136].
//   dqn_fc1_kernel    the 6.4 MB fc1 matrix of each net is streamed exactly once per task (<= 16 rows): a grouped GEMV
//                     like fc2 of the MPE net, [8][784][64][4] tiling, lane = output, rows in groups of four on
//                     v_mfma_f32_4x4x1_16B_f32 with the activations of a chunk staged in LDS.
//   dqn_out_kernel    512 -> n logits, first-max action.
// fp32 arithmetic follows the canonical order of oracle/coevo_oracle.c (taps in (ci,ky,kx) order, sequential-k fc
// chains), so logits equal the oracle's bit for bit.
#include "dqn_common.hip.h"

namespace coevo {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void dqn_pack_kernel(const float *flat, float *slab, int C, int n)
{
    const DqnLayout L = dqn_layout(C, n);
    const int64_t P = dqn_param_count(C, n);
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= L.stride) return;
    const int64_t f = dqn_slab_to_flat(s, C, n);
    slab[(int64_t)blockIdx.y * L.stride + s] = (f >= 0) ? flat[(int64_t)blockIdx.y * P + f] : 0.0f;
}

// ---------------------------------------------------------------------------------------------------------------
// canonical sum over the 64 positions of one block, per output channel, from two 32x32 accumulator tiles.
// position inside the block: i = 32*mt + (reg&3) + 8*(reg>>2) + 4*(lane>>5); tree = i xor 1, 2, 4, 8, 16, 32.
__device__ inline float block_tree_from_acc(const f32x16 &t0, const f32x16 &t1)
{
    float s[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const f32x16 &v = mt ? t1 : t0;
        float a[8], b[4];
#pragma unroll
        for (int q = 0; q < 8; ++q) a[q] = v[2 * q] + v[2 * q + 1];            // xor 1: reg bit 0
#pragma unroll
        for (int q = 0; q < 4; ++q) b[q] = a[2 * q] + a[2 * q + 1];            // xor 2: reg bit 1
#pragma unroll
        for (int q = 0; q < 4; ++q) {                                          // xor 4: lane half
            const u32x2 x = __builtin_amdgcn_permlane32_swap(__float_as_uint(b[q]), __float_as_uint(b[q]), false, false);
            b[q] = __uint_as_float(x[0]) + __uint_as_float(x[1]);
        }
        s[mt] = (b[0] + b[1]) + (b[2] + b[3]);                                  // xor 8, 16: reg bits 2, 3
    }
    return s[0] + s[1];                                                         // xor 32: the two row tiles
}

struct ConvGeom { int cin, ks, stride, hin, hout, cout; };

// One conv + BN(train, batch 1) + ReLU layer of one frame on the matrix cores.
//   UNITS_PER_WAVE: (position block, column tile) work units a wave owns (conv1: 2, conv2/3: 1); unit u of wave w is
//   global unit w + 4*u; unit -> (block = unit / NT, nt = unit % NT).
template <int TAPS_MAX, int KS, int STRIDE, int HIN, int HOUT, int COUT, int UNITS, bool U8IN>
__device__ inline void conv_bn_relu_mfma(const void *in_lds, const float *lut, int cin, const float *wt,
                                         const float *bias, const float *gamma, const float *beta, float *out,
                                         float *red, int w, int l)
{
    constexpr int NPOS = HOUT * HOUT, NBLK = (NPOS + 63) / 64, NT = COUT / 32, NUNIT = NBLK * NT;
    const int lc = l & 31, lh = l >> 5;
    const int taps = cin * KS * KS;
    f32x16 acc[UNITS][2];
    int base[UNITS][2];
    bool live[UNITS];
#pragma unroll
    for (int u = 0; u < UNITS; ++u) {
        const int unit = w + 4 * u;
        live[u] = unit < NUNIT;
        const int blk = unit / NT, nt = unit % NT;
        const float bb = live[u] ? bias[32 * nt + lc] : 0.0f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            int p = 64 * blk + 32 * mt + lc;
            if (!live[u] || p >= NPOS) p = 0;  // padded rows read position 0 and are masked out of every result
            const int oy = p / HOUT, ox = p % HOUT;
            base[u][mt] = U8IN ? ((oy * STRIDE) * HIN + ox * STRIDE) * cin : (oy * STRIDE) * HIN + ox * STRIDE;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[u][mt][r] = bb;
        }
    }
    // all units of one wave share the column tile when NT == 1 (conv1); otherwise a wave has a single unit
    const int nt0 = (w % NT);
    // taps in chunks of 8 k-pairs (all tap counts are multiples of 16): the chunk's 8 weight operands are requested
    // first (8 independent L2 loads in flight), then 8 x (im2col gather out of LDS, MFMAs)
    for (int t0 = 0; t0 < taps; t0 += 16) {
        float bv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) bv[j] = wt[(size_t)(t0 + 2 * j + lh) * COUT + 32 * nt0 + lc];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int t = t0 + 2 * j + lh;
            const int ci = t / (KS * KS), rem = t % (KS * KS), ky = rem / KS, kx = rem % KS;
            const int toff = U8IN ? (ky * HIN + kx) * cin + ci : (ci * HIN + ky) * HIN + kx;
#pragma unroll
            for (int u = 0; u < UNITS; ++u) {
                if (!live[u]) continue;  // wave-uniform
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    float av;
                    if constexpr (U8IN) av = lut[static_cast<const unsigned char *>(in_lds)[base[u][mt] + toff]];
                    else av = static_cast<const float *>(in_lds)[base[u][mt] + toff];
                    acc[u][mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[j], acc[u][mt], 0, 0, 0);
                }
            }
        }
    }
    // ---- BatchNorm statistics: canonical block sums -> LDS -> blocks left to right ------------------------------
    auto masked = [&](int u, int mt, int r) {
        const int unit = w + 4 * u, blk = unit / NT;
        const int p = 64 * blk + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * lh;
        return p < NPOS;
    };
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
        for (int u = 0; u < UNITS; ++u) {
            if (!live[u]) continue;
            const int unit = w + 4 * u, blk = unit / NT, nt = unit % NT;
            f32x16 v0, v1;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float x0 = acc[u][0][r], x1 = acc[u][1][r];
                v0[r] = masked(u, 0, r) ? (pass ? x0 * x0 : x0) : 0.0f;
                v1[r] = masked(u, 1, r) ? (pass ? x1 * x1 : x1) : 0.0f;
            }
            const float s = block_tree_from_acc(v0, v1);
            if (lh == 0) red[(nt * 8 + blk) * 32 + lc] = s;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < UNITS; ++u) {
            if (!live[u]) continue;
            const int unit = w + 4 * u, nt = unit % NT;
            float tot = red[(nt * 8 + 0) * 32 + lc];
            for (int b = 1; b < NBLK; ++b) tot = tot + red[(nt * 8 + b) * 32 + lc];
            const float stat = tot / (float)NPOS;
            if (pass == 0) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[u][mt][r] = acc[u][mt][r] - stat;
            } else {
                const int blk = unit / NT;
                const float rstd = 1.0f / __builtin_sqrtf(stat + LN_EPS);
                const float ga = gamma[32 * nt + lc], be = beta[32 * nt + lc];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int p = 64 * blk + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        if (p < NPOS) {
                            const float y = __builtin_fmaf(acc[u][mt][r] * rstd, ga, be);
                            out[(size_t)(32 * nt + lc) * NPOS + p] = relu_keep_nan(y);
                        }
                    }
            }
        }
        __syncthreads();
    }
}

struct DqnSmem {
    float lut[256];
    float a1[32 * 400];
    float a2[64 * 81];
    float red[2 * 8 * 32];
    unsigned char frame[84 * 84 * 6 + 16];
};

__global__ __launch_bounds__(256) void dqn_conv_kernel(const float *slab, const coevo_dqn_task *tasks, int C,
                                                        int n_actions, const uint8_t *frames, float *act)
{
    __shared__ __attribute__((aligned(16))) DqnSmem sm;
    const coevo_dqn_task task = tasks[blockIdx.x];
    if ((int)blockIdx.y >= task.n_rows) return;
    const int row = task.row_begin + blockIdx.y;
    const int t = threadIdx.x, w = t >> 6, l = t & 63;
    const float *net = slab + task.net_off;
    const DqnLayout L = dqn_layout(C, n_actions);
    // stage the frame (84*84*C bytes, a multiple of 16) and the /255 table
    const int nbytes = 84 * 84 * C;
    const uint4 *src = reinterpret_cast<const uint4 *>(frames + (size_t)row * nbytes);
    uint4 *dst = reinterpret_cast<uint4 *>(sm.frame);
    for (int i = t; i < nbytes / 16; i += 256) dst[i] = src[i];
    sm.lut[t] = (float)t / 255.0f;
    __syncthreads();
    conv_bn_relu_mfma<384, 8, 4, 84, 20, 32, 2, true>(sm.frame, sm.lut, C, net + L.w1, net + L.b1, net + L.b1 + 32,
                                                      net + L.b1 + 64, sm.a1, sm.red, w, l);
    conv_bn_relu_mfma<512, 4, 2, 20, 9, 64, 1, false>(sm.a1, nullptr, 32, net + L.w2, net + L.b2, net + L.b2 + 64,
                                                      net + L.b2 + 128, sm.a2, sm.red, w, l);
    conv_bn_relu_mfma<576, 3, 1, 9, 7, 64, 1, false>(sm.a2, nullptr, 64, net + L.w3, net + L.b3, net + L.b3 + 64,
                                                     net + L.b3 + 128, act + (size_t)row * DQ_FC1_IN, sm.red, w, l);
}

// fc1 + ReLU: grid (task, 8), ONE wavefront per workgroup: it owns outputs [64*ob, +64) and streams their 802 KB
// ([784][64][4] tile) exactly once for the task's <= 16 rows, 28 KiB in flight (only 8 wavefronts exist per net, so the
// memory-level parallelism has to come from depth).  The activations of a chunk (rows x 28 k-quads) are staged in LDS
// with coalesced loads and read back as broadcasts (scalar loads of them serialise: 12 500 dependent s_loads per wave).
constexpr int DQ_RMAX = 16;
template <int NG>  // row groups of four the launch provides for (max rows per task rounded up)
__global__ __launch_bounds__(64) void dqn_fc1_kernel(const float *slab, const coevo_dqn_task *tasks, int C,
                                                      int n_actions, const float *act, float *hid)
{
    constexpr int U = 28;  // k-quads per chunk; 784 = 28 * 28
    __shared__ __attribute__((aligned(16))) float xs[DQ_RMAX][U * 4];
    const coevo_dqn_task task = tasks[blockIdx.x];
    const int l = threadIdx.x;
    const int ob = blockIdx.y;
    const float *net = slab + task.net_off;
    const DqnLayout L = dqn_layout(C, n_actions);
    const int nrows = task.n_rows;
    const float bb = net[L.bf + 64 * ob + l];
    // rows in groups of four on v_mfma_f32_4x4x1_16B_f32 (16 blocks x 4 columns = the wave's 64 outputs, one k per
    // instruction; bit-identical to the fmaf chain, tools/mfma4_chain_probe.hip): the lane's streamed 16-byte piece is
    // the B operand as is, the A operand x[4g + l%4][4q..4q+3] is one ds_read_b128 per group.  (As VALU FMAs fed by one
    // LDS broadcast per row this kernel ran at 1.4 TB/s.)
    typedef float f32x4_acc __attribute__((ext_vector_type(4)));
    f32x4_acc acc[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[g][i] = bb;
    const float4 *wp = reinterpret_cast<const float4 *>(net + L.wf) + (size_t)ob * 784 * 64 + l;
    const float *arow = act + (size_t)task.row_begin * DQ_FC1_IN;
    for (int kq = 0; kq < 784; kq += U) {
        float4 wv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {  // read once per launch: non-temporal, keeps the conv weights / activations in L2
            typedef float f32x4_nt __attribute__((ext_vector_type(4)));
            const f32x4_nt v = __builtin_nontemporal_load(reinterpret_cast<const f32x4_nt *>(wp + (size_t)(kq + u) * 64));
            wv[u] = make_float4(v[0], v[1], v[2], v[3]);
        }
        // the chunk's activations (rows x 28 float4 pieces, coalesced per row; pad rows: zeros): all requested at once,
        // next to the weight loads (one iteration at a time they cost six serial memory latencies per chunk)
        constexpr int XI = (4 * NG * U + 63) / 64;
        float4 xr[XI];
#pragma unroll
        for (int j = 0; j < XI; ++j) {
            const int i = l + 64 * j, r = i / U, q = i % U;
            xr[j] = (i < 4 * NG * U && r < nrows)
                        ? *reinterpret_cast<const float4 *>(arow + (size_t)r * DQ_FC1_IN + 4 * (kq + q))
                        : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __syncthreads();  // the previous chunk's activations have been consumed
#pragma unroll
        for (int j = 0; j < XI; ++j) {
            const int i = l + 64 * j;
            if (i < 4 * NG * U) *reinterpret_cast<float4 *>(&xs[i / U][4 * (i % U)]) = xr[j];
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float4 x[NG];
#pragma unroll
            for (int g = 0; g < NG; ++g) x[g] = *reinterpret_cast<const float4 *>(&xs[4 * g + (l & 3)][4 * u]);
#pragma unroll
            for (int g = 0; g < NG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(x[g].x, wv[u].x, acc[g], 0, 0, 0);
#pragma unroll
            for (int g = 0; g < NG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(x[g].y, wv[u].y, acc[g], 0, 0, 0);
#pragma unroll
            for (int g = 0; g < NG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(x[g].z, wv[u].z, acc[g], 0, 0, 0);
#pragma unroll
            for (int g = 0; g < NG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(x[g].w, wv[u].w, acc[g], 0, 0, 0);
        }
    }
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (4 * g + i < nrows)
                hid[(size_t)(task.row_begin + 4 * g + i) * DQ_FC1_OUT + 64 * ob + l] = relu_keep_nan(acc[g][i]);
}

// output layer + first-max action: one 64-thread workgroup per (task, row)
__global__ __launch_bounds__(64) void dqn_out_kernel(const float *slab, const coevo_dqn_task *tasks, int C,
                                                      int n_actions, const float *hid, int32_t *actions,
                                                      float *logits, int32_t *status)
{
    __shared__ float lg[64];
    const coevo_dqn_task task = tasks[blockIdx.x];
    if ((int)blockIdx.y >= task.n_rows) return;
    const int row = task.row_begin + blockIdx.y, o = threadIdx.x;
    const float *net = slab + task.net_off;
    const DqnLayout L = dqn_layout(C, n_actions);
    if (o < n_actions) {
        float y = net[L.bo + o];
        const float *wr = net + L.wo + (size_t)o * DQ_FC1_OUT, *x = hid + (size_t)row * DQ_FC1_OUT;
        for (int k = 0; k < DQ_FC1_OUT; ++k) y = __builtin_fmaf(wr[k], x[k], y);
        lg[o] = y;
        if (logits) logits[(size_t)row * COEVO_DQN_LOGIT_STRIDE + o] = y;
    }
    __syncthreads();
    if (o == 0) {
        int best = -1;
        float cur = -__builtin_inff();
        for (int i = 0; i < n_actions; ++i)
            if (lg[i] > cur) { cur = lg[i]; best = i; }
        if (best < 0) { atomicOr(status, COEVO_ST_NO_ACTION); best = 0; }
        actions[row] = best;
    }
}

}  // namespace coevo

using namespace coevo;

static bool dqn_shape_ok(int C, int n) { return C >= 1 && C <= 6 && n >= 1 && n <= COEVO_DQN_LOGIT_STRIDE; }

extern "C" int64_t coevo_dqn_param_count(int C, int n_actions)
{
    return dqn_shape_ok(C, n_actions) ? dqn_param_count(C, n_actions) : COEVO_ERR_ARG;
}

extern "C" int64_t coevo_dqn_slab_stride(int C, int n_actions)
{
    return dqn_shape_ok(C, n_actions) ? dqn_layout(C, n_actions).stride : COEVO_ERR_ARG;
}

extern "C" int64_t coevo_dqn_workspace_bytes(int n_rows_total)
{
    return n_rows_total > 0 ? (int64_t)n_rows_total * (DQ_FC1_IN + DQ_FC1_OUT) * 4 : COEVO_ERR_ARG;
}

extern "C" int coevo_dqn_pack(const float *flat, float *slab, int n, int C, int n_actions, void *stream)
{
    if (!flat || !slab || n <= 0 || !dqn_shape_ok(C, n_actions)) return COEVO_ERR_ARG;
    const dim3 grid((unsigned)((dqn_layout(C, n_actions).stride + 255) / 256), (unsigned)n);
    hipLaunchKernelGGL(dqn_pack_kernel, grid, dim3(256), 0, (hipStream_t)stream, flat, slab, C, n_actions);
    COEVO_HIP_CHECK(hipGetLastError());
    return COEVO_OK;
}

extern "C" int coevo_dqn_forward_argmax(const float *slab, const coevo_dqn_task *tasks, int n_tasks,
                                        int max_rows_per_task, int n_rows_total, int C, int n_actions,
                                        const uint8_t *frames, int32_t *actions, float *logits, int32_t *status,
                                        void *workspace, void *stream)
{
    return coevo_dqn_forward_argmax_timed(slab, tasks, n_tasks, max_rows_per_task, n_rows_total, C, n_actions, frames,
                                          actions, logits, status, workspace, nullptr, stream);
}

extern "C" int coevo_dqn_forward_argmax_timed(const float *slab, const coevo_dqn_task *tasks, int n_tasks,
                                              int max_rows_per_task, int n_rows_total, int C, int n_actions,
                                              const uint8_t *frames, int32_t *actions, float *logits, int32_t *status,
                                              void *workspace, void *timing_ctx, void *stream)
{
    if (!slab || !tasks || !frames || !actions || !status || !workspace) return COEVO_ERR_ARG;
    if (n_tasks <= 0 || n_rows_total <= 0 || !dqn_shape_ok(C, n_actions)) return COEVO_ERR_ARG;
    if (max_rows_per_task < 1 || max_rows_per_task > DQ_RMAX) return COEVO_ERR_ARG;
    float *act = static_cast<float *>(workspace);
    float *hid = act + (size_t)n_rows_total * DQ_FC1_IN;
    hipStream_t s = (hipStream_t)stream;
    if (timing_ctx && coevo_timing_begin(timing_ctx, stream) != COEVO_OK) return COEVO_ERR_HIP;
    hipLaunchKernelGGL(dqn_conv_kernel, dim3(n_tasks, max_rows_per_task), dim3(256), 0, s, slab, tasks, C, n_actions,
                       frames, act);
    if (timing_ctx && coevo_timing_end(timing_ctx, stream) != COEVO_OK) return COEVO_ERR_HIP;
    const dim3 g1(n_tasks, 8), b1(64);
    switch ((max_rows_per_task + 3) / 4) {
    case 1: hipLaunchKernelGGL(dqn_fc1_kernel<1>, g1, b1, 0, s, slab, tasks, C, n_actions, act, hid); break;
    case 2: hipLaunchKernelGGL(dqn_fc1_kernel<2>, g1, b1, 0, s, slab, tasks, C, n_actions, act, hid); break;
    case 3: hipLaunchKernelGGL(dqn_fc1_kernel<3>, g1, b1, 0, s, slab, tasks, C, n_actions, act, hid); break;
    default: hipLaunchKernelGGL(dqn_fc1_kernel<4>, g1, b1, 0, s, slab, tasks, C, n_actions, act, hid); break;
    }
    hipLaunchKernelGGL(dqn_out_kernel, dim3(n_tasks, max_rows_per_task), dim3(64), 0, s, slab, tasks, C, n_actions, hid,
                       actions, logits, status);
    COEVO_HIP_CHECK(hipGetLastError());
    return COEVO_OK;
}
