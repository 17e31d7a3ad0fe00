// K2 - DeepQN policy step (reference Atari/deepqn.py:39-48: x/255 -> conv8s4(32) -> BN -> ReLU -> conv4s2(64) -> BN ->
// ReLU -> conv3s1(64) -> BN -> ReLU -> flatten(CHW) -> fc 3136->512 -> ReLU -> fc 512->n) + first-max action, for many
// (weight set x frames) tasks.  BatchNorm is in TRAINING mode at batch 1 in the reference (never .eval()ed), i.e.
// per-sample, per-channel statistics over the spatial positions: batching must not mix samples.
//
// Three launches per env step:
//   dqn_conv_kernel   one workgroup (8 waves) per frame, three per CU: the uint8 HWC frame is staged in LDS once (coalesced
//                     16-byte loads), /255 as an exact two-term product (u8_over_255), the three convolutions run as
//                     implicit GEMMs on v_mfma_f32_16x16x4_f32 (A = im2col gather out of LDS, B = weights in lane
//                     order from L2: dqn_conv_slab_to_flat), BatchNorm statistics are reduced in the canonical order
//                     (lane-strided sums, then one packed butterfly for a wave's channels), activations stay in LDS between layers (one
//                     region, each layer's output written over its input); conv3's output goes to HBM as act[row][3136].
//   dqn_fc1_kernel    the 6.4 MB fc1 matrix of each net is streamed exactly once per task (<= 16 rows): a grouped GEMV
//                     like fc2 of the MPE net, [8][784][64][4] tiling, lane = output, rows in groups of four on
//                     v_mfma_f32_4x4x1_16B_f32 with the activations of a chunk staged in LDS.
//   dqn_out_kernel    512 -> n logits, first-max action.
// fp32 arithmetic follows the canonical order of oracle/coevo_oracle.c (taps in (ci,ky,kx) order, sequential-k fc
// chains), so logits equal the oracle's bit for bit.
#include "dqn_common.hip.h"
#include <type_traits>

namespace coevo {

// build-time tuning switches (defaults = the shipped configuration; tools/build_variant.sh for A/B runs)
#ifndef DQ_LUT
#define DQ_LUT 0   // 1: /255 through a 256-entry LDS table instead of u8_over_255 (measured equal at 3 workgroups per CU)
#endif
#ifndef DQ_FC1_U
#define DQ_FC1_U 7    // fc1: k-quads per chunk of the weight stream (784 = 112 x 7)
#endif
#ifndef DQ_FC1_NB
#define DQ_FC1_NB 8    // fc1: chunks in the wave's register ring (NB - 1 in flight)
#endif
#ifndef DQ_SMALL_QU
#define DQ_SMALL_QU 16   // k-steps per chunk in the small-launch conv kernels: a wave is alone on its SIMD there, so the LDS
#endif                   // latency of its gathers is hidden by depth (16 gathers in flight), not by other waves
#ifndef DQ_SMALL_MAX_ROWS
#define DQ_SMALL_MAX_ROWS 32   // launches of at most this many frames take the three-launch conv stack
#endif
#ifndef DQ_FC1_NBN
#define DQ_FC1_NBN 4    // ring depth of the narrow kernel: chunks of DQ_FC1_U x 4 k-quads, two 16-byte loads per lane each
#endif
#ifndef DQ_FC1_NARROW_MAX_TASKS
#define DQ_FC1_NARROW_MAX_TASKS 16   // launches of at most this many tasks take the 32-waves-per-task kernel
#endif
#ifndef DQ_FC1_ALLNT
#define DQ_FC1_ALLNT 0   // 1: non-temporal weight loads for every task (A/B)
#endif
#ifndef DQ_FC1_HALF
#define DQ_FC1_HALF 0    // 1 (A/B, measured and not shipped): launches of <= 1024 waves as 32-output waves, two per SIMD - 104 vs
                         // 92 us at cfg 4: a wave's time is its 3136-step dependent 4x4x1 chain, whatever its width (r04_experiments.md)
#endif
#ifndef DQ_FC1_NB_HALF
#define DQ_FC1_NB_HALF 4   // ... and their ring depth (<= 256 registers)
#endif
#ifndef DQ_FC1_NB_MANY
#define DQ_FC1_NB_MANY 2   // ... of a launch with more waves than the chip has SIMDs (fewer registers: several waves per SIMD)
#endif
#ifndef DQ_FC1_TQ
#define DQ_FC1_TQ 2    // tiled fc1: super-quads (16 k: four 16-byte loads per lane, one per 16-output tile) per chunk of the ring
#endif
#ifndef DQ_FC1_TNB
#define DQ_FC1_TNB 7   // ... and chunks in the ring (196 / DQ_FC1_TQ a multiple of it): 6 x 8 KiB in flight per wave
#endif
#ifndef DQ_WPE
#define DQ_WPE 6   // waves per SIMD the register budget is set for: 3 workgroups x 8 waves / 4 SIMDs
#endif
#ifndef DQ_QU1
#define DQ_QU1 4
#endif
#ifndef DQ_QU2
#define DQ_QU2 4
#endif
#ifndef DQ_QU3
#define DQ_QU3 4
#endif


__global__ __launch_bounds__(256) void dqn_pack_kernel(const float *flat, float *slab, int C, int n, int fc1_tiled)
{
    const DqnLayout L = dqn_layout(C, n);
    const int64_t P = dqn_param_count(C, n);
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= L.stride) return;
    const int64_t f = dqn_slab_to_flat(s, C, n, fc1_tiled);
    slab[(int64_t)blockIdx.y * L.stride + s] = (f >= 0) ? flat[(int64_t)blockIdx.y * P + f] : 0.0f;
}

// ---------------------------------------------------------------------------------------------------------------
// One conv layer of one frame as an implicit GEMM on v_mfma_f32_16x16x4_f32 (bit-identical to the sequential-k fmaf chain
// from the bias: tools/mfma16_chain_probe.hip; taps in the canonical (ci, ky, kx) order).  M = output positions in tiles
// of 16, N = output channels in pairs of 16-wide tiles, K = taps, four per instruction.  A unit = (position tile, channel
// tile pair): one LDS gather per lane feeds two MFMAs.  Unit u = w + 8 i belongs to wave w (8 waves, two per SIMD: one
// wave's gathers hide behind the other's MFMAs), so all units of a wave share their channel pair and the pair's weight
// operands are loaded once per k-step pair (a chunk of QU k-steps ahead, from L2: the 16 frames of a task and every task
// of the same net read the same 0.3 MB).  Tile padding: 400 = 25 x 16 positions (0 %), 81 -> 96 (16 %), 49 -> 64 (23 %); the 32 x 32 tiles
// this replaces padded 400 -> 512, 81 -> 128, 49 -> 64 and left half of the waves idle in conv3.
//   operands of one MFMA: lane (c = l % 16, kk = l / 16): A[position c of the tile][tap 4 q + kk], B[tap 4 q + kk][channel c];
//   accumulator register r of lane (c, g = l / 16): position 4 g + r of the tile, channel c.
// The raw sums go to LDS as out[channel][position] (odd pitch); BatchNorm + ReLU then runs over them channel by channel.
typedef float f32x4_acc __attribute__((ext_vector_type(4)));

// x / 255.0f for x = 0 .. 255, correctly rounded, without the divide: 1/255 split into a float head and tail,
// fma(x, head, x * tail) equals the IEEE quotient for all 256 inputs (tests/test_host_logic_cpu.py checks the identity
// in numpy; an LDS table of the quotients cost a dependent, bank-conflicted read per gathered tap)
__device__ __forceinline__ float u8_over_255(unsigned b)
{
    constexpr float HEAD = (float)(1.0 / 255.0), TAIL = (float)(1.0 / 255.0 - (double)HEAD);
    const float x = (float)b;
    return __builtin_fmaf(x, HEAD, x * TAIL);
}

// Addressing is organised per chunk of QU k-steps (4 QU taps) so that the MFMA loop issues almost no address arithmetic
// (SQ counters of the first 16x16x4 version: 4 VALU instructions per MFMA - 64-bit weight addresses, tap decoding - kept
// the matrix pipe at 46 %): within a chunk the tap of k-step j, lane group kk is
//   conv1 (8 x 8 window, t = 4 q + kk):  ci = q0 / 16, ky = (q0 / 2) % 8 + j / 2, kx = kk + 4 (j % 2)
//   conv2 (4 x 4):                        ci = q0 / 4 + j / 4, ky = j % 4, kx = kk
// i.e. one per-lane chunk base + a compile-time offset per j (an instruction immediate); conv3's 3 x 3 window repeats every
// nine k-steps instead: nine per-lane addresses + an immediate (Conv16::run).
template <int KS, int HIN, bool U8IN, int IN_PITCH, int CT>
struct TapAddr {
    // offset of tap (chunk q0, step j, lane group kk) = chunk_base(q0, kk) + rel(j)
    __device__ static __forceinline__ int chunk_base(int q0, int kk, int cin)
    {
        if constexpr (KS == 8) {
            const int ci = q0 >> 4, ky0 = (q0 >> 1) & 7;
            return U8IN ? (ky0 * HIN + kk) * (CT ? CT : cin) + ci : ci * IN_PITCH + ky0 * HIN + kk;
        } else {
            return (q0 >> 2) * IN_PITCH + kk;
        }
    }
    __device__ static __forceinline__ int rel(int j, int cin)
    {
        if constexpr (KS == 8) return U8IN ? ((j >> 1) * HIN + 4 * (j & 1)) * (CT ? CT : cin) : (j >> 1) * HIN + 4 * (j & 1);
        else return (j >> 2) * IN_PITCH + (j & 3) * HIN;
    }
};

template <int KS, int STRIDE, int HIN, int HOUT, int COUT, bool U8IN, int IN_PITCH, int OUT_PITCH, int CT, int QU, bool OVER>
struct Conv16 {
    static constexpr int NPOS = HOUT * HOUT, NM = (NPOS + 15) / 16, NP = COUT / 32, NUNITS = NM * NP;
    // SPLIT (conv1: 25 units on 8 waves): every wave takes three whole units and the 25th is halved between waves 0 and 1
    // (they sit on different SIMDs), one channel tile each: 7 MFMAs per k-step on the critical waves instead of 8
    static constexpr bool SPLIT = (NP == 1) && (NUNITS % 8 == 1);
    static constexpr int NFULL = NUNITS / 8, REM = SPLIT ? 0 : NUNITS % 8;   // waves w < REM carry NFULL + 1 units
    static_assert(8 % NP == 0, "all units of a wave share their channel pair");
    using Tap = TapAddr<KS, HIN, U8IN, IN_PITCH, CT>;

    // NU whole units (+ the half unit when XL) of wave w, as straight-line code: the unit count is a template argument so
    // that the k-loop has no branches and the scheduler can move a k-step's gathers above the previous step's MFMAs
    // The units are given by the caller: channel pair np (all units of a wave share it), position tile mt[i] of unit i,
    // xh = which channel tile of the last position tile the half unit takes (XL).  in_shift: elements by which `in_lds`
    // starts inside the layer's input image (a kernel that stages only the rows its tile needs).
    template <int NU, bool XL>
    static __device__ __forceinline__ void run(const void *in_lds, const float *lut, int cin, int taps, const float *wt,
                                               const float *bias, float *out, int l, int np,
                                               const int (&mt)[NU > 0 ? NU : 1], int xh, int in_shift)
    {
        const int c = l & 15, kk = l >> 4;
        constexpr int NA = NU > 0 ? NU : 1;
        f32x4_acc acc[NA][2], accx;
        int base[NA], xbase = 0;
        auto tile_base = [&](int m) {
            int p = 16 * m + c;
            if (p >= NPOS) p = 0;   // padded rows read position 0; their results are never stored
            const int oy = p / HOUT, ox = p % HOUT;
            return (U8IN ? ((oy * STRIDE) * HIN + ox * STRIDE) * (CT ? CT : cin) : (oy * STRIDE) * HIN + ox * STRIDE) - in_shift;
        };
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            base[i] = tile_base(mt[i]);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float bb = bias[32 * np + 16 * h + c];
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][h][r] = bb;
                // pin the splat in a real register tuple: hipcc 7.2 otherwise kept only element 0 of some accumulators
                // live up to the first MFMA and reused elements 1..3 for the loop's address / operand temporaries
                // (seen in the ISA of the three-unit + half-unit instantiation; tests/test_deepqn_gpu.py caught it)
                asm volatile("" : "+v"(acc[i][h]));
            }
        }
        if constexpr (XL) {
            xbase = tile_base(NM - 1);
            const float bb = bias[16 * xh + c];
#pragma unroll
            for (int r = 0; r < 4; ++r) accx[r] = bb;
            asm volatile("" : "+v"(accx));
        }
        // QU = k-steps per chunk: their weight operands are requested together, one chunk AHEAD of the MFMAs that use them
        // (an L2 round trip per chunk would otherwise be exposed: conv2 / conv3 have only one or two units per wave to hide
        // it behind).  This lane's B operands: one float4 per k-step pair (layout: dqn_common.hip.h dqn_conv_slab_to_flat)
        // (buffer loads: wave-uniform descriptor + the chunk's byte offset in an SGPR + the lane's constant 32-bit offset - as
        // global loads every chunk cost a 64-bit vector add, and vector instructions are taken from the matrix pipe's time)
        typedef unsigned u32x4_b __attribute__((ext_vector_type(4)));
        const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float *>(wt) + np * 256, 0, 0x7fffffff, 0x00020000);
        float bvA[QU][2], bvB[QU][2];
        int cbA = 0, cbB = 0;    // chunk bases (conv1 / conv2)
        // conv3 (3 x 3 window): k-step q = 9 m + s, lane group kk holds tap t = 36 m + 4 s + kk, i.e. input channel
        // 4 m + (4 s + kk) / 9 and window cell (4 s + kk) % 9: nine per-lane addresses per unit (s = 0 .. 8) + a compile-time
        // 4 m channel pitches (an instruction immediate; the k loop is unrolled) - no vector instruction per gather.  (A
        // 16-bit offset table in LDS cost a table read and a shift-add per gather: half a vector instruction per MFMA.)
        int a9[KS == 3 ? NA : 1][9];
        if constexpr (KS == 3) {
#pragma unroll
            for (int sft = 0; sft < 9; ++sft) {
                const int t = 4 * sft + kk, cell = t % 9;
#pragma unroll
                for (int i = 0; i < NA; ++i) a9[i][sft] = base[i] + (t / 9) * IN_PITCH + (cell / 3) * HIN + cell % 3;
            }
        }
        auto issue = [&](float (&bv)[QU][2], int &cb, int q0) {
#pragma unroll
            for (int jp = 0; jp < QU / 2; ++jp) {
                const u32x4_b v = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, 16 * l + jp * (NP * 1024), (q0 >> 1) * (NP * 1024), 0);
                bv[2 * jp][0] = __uint_as_float(v[0]);
                bv[2 * jp][1] = __uint_as_float(v[1]);
                bv[2 * jp + 1][0] = __uint_as_float(v[2]);
                bv[2 * jp + 1][1] = __uint_as_float(v[3]);
            }
            if constexpr (KS != 3) cb = Tap::chunk_base(q0, kk, cin);
        };
        auto consume = [&](const float (&bv)[QU][2], int cb, int qc) {   // qc: the chunk's first k-step (conv3 only)
#pragma unroll
            for (int j = 0; j < QU; ++j) {
                auto gather = [&](int b0, int i) {
                    int off;
                    if constexpr (KS == 3) off = a9[i][(qc + j) % 9] + ((qc + j) / 9) * (4 * IN_PITCH);
                    else off = b0 + cb + Tap::rel(j, cin);
                    if constexpr (U8IN) return DQ_LUT ? lut[static_cast<const unsigned char *>(in_lds)[off]]
                                              : u8_over_255(static_cast<const unsigned char *>(in_lds)[off]);
                    else return static_cast<const float *>(in_lds)[off];
                };
#pragma unroll
                for (int i = 0; i < NU; ++i) {
                    const float av = gather(base[i], i);
                    acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[j][0], acc[i][0], 0, 0, 0);
                    acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[j][1], acc[i][1], 0, 0, 0);
                }
                if constexpr (XL) {
                    const float av = gather(xbase, NU);
                    accx = __builtin_amdgcn_mfma_f32_16x16x4f32(av, xh ? bv[j][1] : bv[j][0], accx, 0, 0, 0);
                }
            }
        };
        issue(bvA, cbA, 0);
        if constexpr (KS == 3) {
            constexpr int NQ = 144;   // 64 input channels x 9 cells / 4: fully unrolled (a9's index must be a constant)
            static_assert(!XL && NQ % QU == 0, "conv3: whole units, whole chunks");
#pragma unroll
            for (int q0 = 0; q0 < NQ; q0 += 2 * QU) {
                if (q0 + QU < NQ) issue(bvB, cbB, q0 + QU);
                consume(bvA, cbA, q0);
                if (q0 + QU >= NQ) break;
                if (q0 + 2 * QU < NQ) issue(bvA, cbA, q0 + 2 * QU);
                consume(bvB, cbB, q0 + QU);
            }
        } else {
            const int nq = taps / 4;   // a multiple of QU (C * 16, 128)
            for (int q0 = 0; q0 < nq; q0 += 2 * QU) {
                if (q0 + QU < nq) issue(bvB, cbB, q0 + QU);
                consume(bvA, cbA, 0);
                if (q0 + QU >= nq) break;
                if (q0 + 2 * QU < nq) issue(bvA, cbA, q0 + 2 * QU);
                consume(bvB, cbB, 0);
            }
        }
        if constexpr (OVER) __syncthreads();   // the output overwrites the input: every wave has gathered its last tap
#pragma unroll
        for (int i = 0; i < NU; ++i) {
            const int m = mt[i];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int p = 16 * m + 4 * kk + r;
                    if (p < NPOS) out[(32 * np + 16 * h + c) * OUT_PITCH + p] = acc[i][h][r];
                }
        }
        if constexpr (XL) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int p = 16 * (NM - 1) + 4 * kk + r;
                if (p < NPOS) out[(16 * xh + c) * OUT_PITCH + p] = accx[r];
            }
        }
    }
};

// The whole layer on the eight waves of a frame's workgroup: unit u = w + 8 i belongs to wave w.  w must be wave-uniform in
// an SGPR (readfirstlane): the dispatch below is a scalar branch (every wave runs exactly one instantiation, so the barrier
// inside is reached once by all of them)
template <int KS, int STRIDE, int HIN, int HOUT, int COUT, bool U8IN, int IN_PITCH, int OUT_PITCH, int CT, int QU, bool OVER>
__device__ __forceinline__ void conv16_mfma(const void *in_lds, const float *lut, int cin, int taps, const float *wt,
                                            const float *bias, float *out, int w, int l)
{
    using K = Conv16<KS, STRIDE, HIN, HOUT, COUT, U8IN, IN_PITCH, OUT_PITCH, CT, QU, OVER>;
    // every wave must enter run<> exactly once: the barrier of an OVER layer sits inside it
    static_assert(K::SPLIT || K::REM == 0 || K::NFULL > 0, "a wave without units would skip run<>'s barrier");
    const int np = w % K::NP;
    auto call = [&](auto nu, auto xl) {
        constexpr int NU = decltype(nu)::value;
        int mt[NU > 0 ? NU : 1];
#pragma unroll
        for (int i = 0; i < (NU > 0 ? NU : 1); ++i) mt[i] = (w + 8 * i) / K::NP;
        K::template run<NU, decltype(xl)::value>(in_lds, lut, cin, taps, wt, bias, out, l, np, mt, w & 1, 0);
    };
    using std::integral_constant;
    if constexpr (K::SPLIT) {
        if (w < 2) call(integral_constant<int, K::NFULL>{}, integral_constant<bool, true>{});
        else call(integral_constant<int, K::NFULL>{}, integral_constant<bool, false>{});
    } else if constexpr (K::REM == 0) {
        call(integral_constant<int, K::NFULL>{}, integral_constant<bool, false>{});
    } else {
        if (w < K::REM) call(integral_constant<int, K::NFULL + 1>{}, integral_constant<bool, false>{});
        else call(integral_constant<int, K::NFULL>{}, integral_constant<bool, false>{});
    }
}

// BatchNorm in training mode at batch 1 (per-sample, per-channel statistics over the NPOS positions) + ReLU, in place on
// x[channel][position] in LDS.  mean = S / N, var = S2 / N (biased), rstd = 1 / sqrtf(var + 1e-5f),
// y = fmaf(d * rstd, gamma, beta).  S = the canonical sum of a channel image (oracle/coevo_oracle.c reduce_strided64): lane l
// adds its positions l, l + 64, l + 128, ... left to right (pad = 0), then the canonical 64-lane tree over the lane sums.
// A wave owns COUT / 8 channels: their lane sums are plain vector adds and ALL the trees of a pass are ONE packed butterfly
// (coevo_common.hip.h: lane k then holds channel k's total); the IEEE divides and the square root run once per pass,
// lane-parallel, and each channel's mean / rstd comes back by v_readlane.  (f32 MFMA and VALU instructions share one
// issue resource - tools/mfma_rate_probe.hip - so every vector instruction of this pass is taken from the other
// workgroups' matrix time.  The first form summed 64-wide blocks by a tree each and chained the block sums: 7 trees per
// channel of conv1's image instead of one, 550 instead of ~250 vector instructions per wave for that pass.)
template <int NPOS, int PITCH, int COUT>
__device__ __forceinline__ void bn_relu_rows(float *x, const float *gamma, const float *beta, int w, int l)
{
    constexpr int NB = (NPOS + 63) / 64, CPW = COUT / 8;
    static_assert(CPW >= 1 && CPW <= 16, "one packed butterfly per pass");
    float v[CPW][NB], s1[CPW], s2[CPW];
#pragma unroll
    for (int k = 0; k < CPW; ++k) {
#pragma unroll
        for (int b = 0; b < NB; ++b) v[k][b] = (64 * b + l < NPOS) ? x[(w + 8 * k) * PITCH + 64 * b + l] : 0.0f;
        s1[k] = v[k][0];
#pragma unroll
        for (int b = 1; b < NB; ++b) s1[k] = s1[k] + v[k][b];
    }
    const float meanv = packed_totals<CPW>(s1, l) / (float)NPOS;   // lane k: channel k's mean
#pragma unroll
    for (int k = 0; k < CPW; ++k) {
        const float mean = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(meanv), k));
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            v[k][b] = v[k][b] - mean;
            const float sq = (64 * b + l < NPOS) ? v[k][b] * v[k][b] : 0.0f;
            s2[k] = (b == 0) ? sq : s2[k] + sq;
        }
    }
    const float varv = packed_totals<CPW>(s2, l) / (float)NPOS;
    const float rstdv = 1.0f / __builtin_sqrtf(varv + LN_EPS);
#pragma unroll
    for (int k = 0; k < CPW; ++k) {
        const int ch = w + 8 * k;
        const float rstd = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rstdv), k));
        const float ga = gamma[ch], be = beta[ch];
#pragma unroll
        for (int b = 0; b < NB; ++b)
            if (64 * b + l < NPOS) x[ch * PITCH + 64 * b + l] = relu_keep_nan(__builtin_fmaf(v[k][b] * rstd, ga, be));
    }
}

constexpr int DQ_P1 = 401, DQ_P2 = 81, DQ_P3 = 49;   // channel pitches of the activation images (odd: conflict-free columns)

// One region serves every layer: the frame, then conv1's output written OVER it (conv1 keeps its sums in registers until
// every wave has read its last tap: one extra barrier), conv2's output over that the same way, conv3's next to conv2's.
// 52.5 KB instead of 80.6 KB: three workgroups (24 waves) per CU - while one frame is in a barrier, a BatchNorm pass or its
// staging, two others feed the matrix pipe - and six-channel frames fit the same footprint.
template <int CMAX>
struct DqnSmem {
    float lut[DQ_LUT ? 256 : 1];                   // x / 255.0f for x = 0 .. 255
    union {
        unsigned char frame[84 * 84 * CMAX + 16];  // the uint8 HWC frame (dead after conv1's last gather)
        float a1[32 * DQ_P1];                      // conv1 activations (dead after conv2's last gather)
        struct {
            float a2[64 * DQ_P2];                  // conv2 activations
            float a3[64 * DQ_P3];                  // conv3 activations = the flattened CHW row
        };
    };
};
static_assert(sizeof(DqnSmem<6>) * 3 <= 160 * 1024, "three workgroups per CU");

#ifdef COEVO_PHASE_STAMPS
// diagnostic build only (tools/dqn_conv_phases.py): wave 0's arrival at each phase boundary, 100 MHz constant clock
__device__ unsigned long long g_dqn_stamps[2048 * 16];
#define DQ_STAMP(i)                                                                                   \
    do {                                                                                              \
        if (threadIdx.x == 0 && blockIdx.x < 2048) g_dqn_stamps[blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define DQ_STAMP(i) do { } while (0)
#endif

// CT: the channel count as a compile-time constant (4 or 6: the gather offsets become instruction immediates), 0 = run time
template <int CMAX, int CT>
__global__ __launch_bounds__(512, DQ_WPE) void dqn_conv_kernel(const float *slab, const coevo_dqn_task *tasks, int n_tasks,
                                                           int n_rows, int C, int n_actions, const uint8_t *frames, float *act)
{
    __shared__ __attribute__((aligned(16))) DqnSmem<CMAX> sm;
    // one workgroup per frame; its task by binary search (tasks ascend in row_begin).  Workgroups are dealt to the 8 XCDs
    // round robin: XCD x takes the contiguous rows [x * per, (x + 1) * per), so the frames of one net meet in ONE L2 (and,
    // dispatched back to back, on one CU's L1) instead of pulling every net's conv weights into all eight.
#ifdef DQ_NO_XCD_MAP
    const int row = blockIdx.x;
#else
    const int per = gridDim.x >> 3;
    const int row = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
#endif
    if (row >= n_rows) return;   // workgroup-uniform (the grid is rounded up to a multiple of 8)
    const coevo_dqn_task task = tasks[task_of_row(tasks, n_tasks, row)];
    const int t = threadIdx.x, w = __builtin_amdgcn_readfirstlane(t >> 6), l = t & 63;   // w: scalar (wave-uniform branches)
    const float *net = slab + task.net_off;
    const DqnLayout L = dqn_layout(C, n_actions);
    // stage the frame (84*84*C bytes, a multiple of 16)
    DQ_STAMP(0);
    const int nbytes = 84 * 84 * C;
    const uint4 *src = reinterpret_cast<const uint4 *>(frames + (size_t)row * nbytes);
    uint4 *dst = reinterpret_cast<uint4 *>(sm.frame);
    for (int i = t; i < nbytes / 16; i += 512) dst[i] = src[i];
    if (DQ_LUT && t < 256) sm.lut[t] = (float)t / 255.0f;
    __syncthreads();
    DQ_STAMP(1);
    conv16_mfma<8, 4, 84, 20, 32, true, 0, DQ_P1, CT, DQ_QU1, true>(sm.frame, sm.lut, C, C * 64, net + L.w1, net + L.b1, sm.a1, w, l);
    __syncthreads();
    DQ_STAMP(2);
    bn_relu_rows<400, DQ_P1, 32>(sm.a1, net + L.b1 + 32, net + L.b1 + 64, w, l);
    __syncthreads();
    DQ_STAMP(3);
    conv16_mfma<4, 2, 20, 9, 64, false, DQ_P1, DQ_P2, 0, DQ_QU2, true>(sm.a1, nullptr, 32, 512, net + L.w2, net + L.b2, sm.a2, w, l);
    __syncthreads();
    DQ_STAMP(4);
    bn_relu_rows<81, DQ_P2, 64>(sm.a2, net + L.b2 + 64, net + L.b2 + 128, w, l);
    __syncthreads();
    DQ_STAMP(5);
    conv16_mfma<3, 1, 9, 7, 64, false, DQ_P2, DQ_P3, 0, DQ_QU3, false>(sm.a2, nullptr, 64, 576, net + L.w3, net + L.b3, sm.a3, w, l);
    __syncthreads();
    DQ_STAMP(6);
    bn_relu_rows<49, DQ_P3, 64>(sm.a3, net + L.b3 + 64, net + L.b3 + 128, w, l);
    __syncthreads();
    DQ_STAMP(7);
    // flatten in CHW order (Atari/deepqn.py:45): channel pitch 49 = the flat layout itself
    float *dsta = act + (size_t)row * DQ_FC1_IN;
    for (int i = t; i < DQ_FC1_IN; i += 512) dsta[i] = sm.a3[i];
    DQ_STAMP(8);
}

// ---------------------------------------------------------------------------------------------------------------
// The conv stack of a SMALL launch (a Co-ES generation's evaluation games: ten frames per agent-step, 200 dependent steps;
// the reference's own population sizes of record are 20): one workgroup per frame leaves the chip empty and takes 51 us
// (28.6 us of matrix issue on the four SIMDs of ONE CU, eight barriers).  Here a frame is spread over the chip in three
// launches, the layers' images passing through L2:
//   dqn_small_conv1_kernel  (frame, position tile): one wave stages the 12 input rows its 16 positions need and runs the
//                           tile's 64 k-steps for all 32 channels; raw sums -> a1raw[frame][32][400]          (25 waves / frame)
//   dqn_small_conv2_kernel  (frame, channel pair): BatchNorm(32) + ReLU of conv1's sums (statistics need every position,
//                           so they are taken here, by each of the two workgroups), then conv2 + BatchNorm + ReLU of
//                           32 of the 64 channels -> a2[frame][64][81]
//   dqn_small_conv3_kernel  (frame, channel pair): conv3 + BatchNorm + ReLU of 32 channels -> act[frame][3136]
// Same Conv16 / bn_relu_rows code, same order of every sum: the bits of dqn_conv_kernel.
constexpr int DQ_SMALL_ROWS1 = 12;   // input rows a 16-position tile of conv1 touches: two output rows x stride 4 + 8 - 4

template <int CMAX, int CT>
__global__ __launch_bounds__(64) void dqn_small_conv1_kernel(const float *slab, const coevo_dqn_task *tasks, int n_tasks, int C,
                                                              int n_actions, const uint8_t *frames, float *a1raw)
{
    __shared__ __attribute__((aligned(16))) unsigned char rows[DQ_SMALL_ROWS1 * 84 * CMAX + 16];
    __shared__ float lut[DQ_LUT ? 256 : 1];
    const int row = blockIdx.x, m = blockIdx.y, l = threadIdx.x;
    const coevo_dqn_task task = tasks[task_of_row(tasks, n_tasks, row)];
    const float *net = slab + task.net_off;
    const DqnLayout L = dqn_layout(C, n_actions);
    const int y0 = 4 * ((16 * m) / 20);                               // first input row of the tile's first output row
    const int nrows = min(DQ_SMALL_ROWS1, 84 - y0), nbytes = nrows * 84 * C;   // (84 * C is a multiple of 16)
    const uint4 *src = reinterpret_cast<const uint4 *>(frames + (size_t)row * (84 * 84 * C) + (size_t)y0 * 84 * C);
    uint4 *dst = reinterpret_cast<uint4 *>(rows);
    for (int i = l; i < nbytes / 16; i += 64) dst[i] = src[i];
    if (DQ_LUT) for (int i = l; i < 256; i += 64) lut[i] = (float)i / 255.0f;
    __syncthreads();
    using K = Conv16<8, 4, 84, 20, 32, true, 0, 400, CT, DQ_SMALL_QU, false>;
    const int mt[1] = {m};
    K::template run<1, false>(rows, lut, C, C * 64, net + L.w1, net + L.b1, a1raw + (size_t)row * (32 * 400), l, 0, mt, 0,
                              y0 * 84 * C);
}

struct DqnSmallSmem2 {
    union {
        float a1[32 * DQ_P1];
        float a2[64 * DQ_P2];
    };
};

__global__ __launch_bounds__(512) void dqn_small_conv2_kernel(const float *slab, const coevo_dqn_task *tasks, int n_tasks, int C,
                                                               int n_actions, const float *a1raw, float *a2g)
{
    __shared__ __attribute__((aligned(16))) DqnSmallSmem2 sm;
    const int row = blockIdx.x, pr = blockIdx.y;
    const coevo_dqn_task task = tasks[task_of_row(tasks, n_tasks, row)];
    const int t = threadIdx.x, w = __builtin_amdgcn_readfirstlane(t >> 6), l = t & 63;
    const float *net = slab + task.net_off;
    const DqnLayout L = dqn_layout(C, n_actions);
    const float *src = a1raw + (size_t)row * (32 * 400);
    for (int i = t; i < 32 * 400; i += 512) sm.a1[(i / 400) * DQ_P1 + i % 400] = src[i];
    __syncthreads();
    bn_relu_rows<400, DQ_P1, 32>(sm.a1, net + L.b1 + 32, net + L.b1 + 64, w, l);
    __syncthreads();
    using K = Conv16<4, 2, 20, 9, 64, false, DQ_P1, DQ_P2, 0, DQ_SMALL_QU, true>;
    const int mt[1] = {w};
    if (w < K::NM) K::template run<1, false>(sm.a1, nullptr, 32, 512, net + L.w2, net + L.b2, sm.a2, l, pr, mt, 0, 0);
    else K::template run<0, false>(sm.a1, nullptr, 32, 512, net + L.w2, net + L.b2, sm.a2, l, pr, mt, 0, 0);
    __syncthreads();
    bn_relu_rows<81, DQ_P2, 32>(sm.a2 + 32 * pr * DQ_P2, net + L.b2 + 64 + 32 * pr, net + L.b2 + 128 + 32 * pr, w, l);
    __syncthreads();
    float *dst = a2g + (size_t)row * (64 * DQ_P2) + 32 * pr * DQ_P2;
    for (int i = t; i < 32 * DQ_P2; i += 512) dst[i] = sm.a2[32 * pr * DQ_P2 + i];
}

__global__ __launch_bounds__(512) void dqn_small_conv3_kernel(const float *slab, const coevo_dqn_task *tasks, int n_tasks, int C,
                                                               int n_actions, const float *a2g, float *act)
{
    __shared__ __attribute__((aligned(16))) float a2[64 * DQ_P2];
    __shared__ __attribute__((aligned(16))) float a3[64 * DQ_P3];
    const int row = blockIdx.x, pr = blockIdx.y;
    const coevo_dqn_task task = tasks[task_of_row(tasks, n_tasks, row)];
    const int t = threadIdx.x, w = __builtin_amdgcn_readfirstlane(t >> 6), l = t & 63;
    const float *net = slab + task.net_off;
    const DqnLayout L = dqn_layout(C, n_actions);
    const float *src = a2g + (size_t)row * (64 * DQ_P2);
    for (int i = t; i < 64 * DQ_P2; i += 512) a2[i] = src[i];
    __syncthreads();
    using K = Conv16<3, 1, 9, 7, 64, false, DQ_P2, DQ_P3, 0, DQ_SMALL_QU, false>;
    const int mt[1] = {w};
    if (w < K::NM) K::template run<1, false>(a2, nullptr, 64, 576, net + L.w3, net + L.b3, a3, l, pr, mt, 0, 0);
    __syncthreads();
    bn_relu_rows<49, DQ_P3, 32>(a3 + 32 * pr * DQ_P3, net + L.b3 + 64 + 32 * pr, net + L.b3 + 128 + 32 * pr, w, l);
    __syncthreads();
    float *dst = act + (size_t)row * DQ_FC1_IN + 32 * pr * DQ_P3;   // CHW flatten: channel pitch 49 = the flat layout
    for (int i = t; i < 32 * DQ_P3; i += 512) dst[i] = a3[32 * pr * DQ_P3 + i];
}

constexpr int DQ_RMAX = 16;
// NG = row groups of four this task needs (ceil(rows / 4)): a one-frame task (Co-ES) issues a quarter of a 16-frame task's
// MFMAs.  One launch serves tasks of every size: the kernel picks the instantiation by the task's own row count.
// NB = chunks of the weight stream in the wave's register ring: NB - 1 of them (14 KiB each) are in flight while one feeds
// the matrix pipe.
// HALF: a wave owns 32 outputs instead of 64 (`ob` then counts 32-output blocks): lanes l and l + 32 hold the SAME output
// column l % 32 and different row groups - the sixteen 4x4 blocks of an MFMA are 8 column quads x 2 row groups - so a 16-row
// task issues two matrix instructions per k and wave instead of four, and has 16 waves instead of 8.  For launches with fewer
// waves than the chip has SIMDs (a Co-GA shard: 90 tasks): a lone wave on a SIMD cannot hide its own LDS round trips and MFMA
// chain (100 k matrix-pipe cycles for 64 outputs x 16 rows); two half-size waves per SIMD overlap each other's.  The two
// lanes of a column request the same 16-byte piece (one 512-byte segment per wave-load): no extra bytes leave the L2.
// NG counts the MFMA groups of the wave (HALF: 8 rows each, else 4).  Same sequential-k chain per (row, output): same bits.
template <int NG, int NB, bool SHARED_NET, bool HALF = false>
__device__ __forceinline__ void dqn_fc1_body(const float *net, const DqnLayout &L, const coevo_dqn_task &task,
                                             const float *act, float *hid, float (*xs)[DQ_RMAX][DQ_FC1_U * 4], int ob, int l)
{
    constexpr int U = DQ_FC1_U;  // k-quads per chunk; 784 = 56 * 14
    constexpr int NCHUNK = 784 / U;
    static_assert(784 % U == 0 && NCHUNK % NB == 0 && NB >= 2, "whole rounds of the ring");
    constexpr int RG = HALF ? 2 * NG : NG;          // row groups of four staged per chunk
    static_assert(4 * RG <= DQ_RMAX, "rows of one task");
    const int nrows = task.n_rows;
    const int col = HALF ? 32 * ob + (l & 31) : 64 * ob + l;   // the lane's output
    const int hrow = HALF ? (l >> 5) : 0;                       // which of a group's two row quads this lane accumulates
    const float bb = net[L.bf + col];
    // rows in groups of four on v_mfma_f32_4x4x1_16B_f32 (16 blocks x 4 columns = the wave's 64 outputs, one k per
    // instruction; bit-identical to the fmaf chain, tools/mfma4_chain_probe.hip): the lane's streamed 16-byte piece is
    // the B operand as is, the A operand x[4g + l%4][4q..4q+3] is one ds_read_b128 per group.  (As VALU FMAs fed by one
    // LDS broadcast per row this kernel ran at 1.4 TB/s.)
    typedef float f32x4_acc1 __attribute__((ext_vector_type(4)));
    typedef float f32x4_nt __attribute__((ext_vector_type(4))) __attribute__((unused));
    f32x4_acc1 acc[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[g][i] = bb;
    const float4 *wp = reinterpret_cast<const float4 *>(net + L.wf) + (size_t)(col >> 6) * 784 * 64 + (col & 63);
    const float *arow = act + (size_t)task.row_begin * DQ_FC1_IN;
    constexpr int XI = (4 * RG * U + 63) / 64;
    float4 wv[NB][U], xr[NB][XI];
    // a chunk's weight pieces (read once per launch: non-temporal, keeps the conv weights / activations in L2 - unless the
    // neighbouring task streams the same net: then plain loads, so that the siblings' requests meet in the XCD's L2.
    // FETCH_SIZE of the Co-GA launch, 400 MB of distinct weights in 90 tasks: 481 MB non-temporal, 440 MB plain, and the
    // launch's time follows those bytes; plain loads for every task cost the conv launch 4 - 20 %) and its activations (rows x 14 float4 pieces, coalesced per row; pad rows: zeros), all requested together
    auto issue = [&](float4 (&w)[U], float4 (&x)[XI], int kq) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if constexpr (SHARED_NET) {
                w[u] = wp[(size_t)(kq + u) * 64];
            } else {
                const f32x4_nt v = __builtin_nontemporal_load(reinterpret_cast<const f32x4_nt *>(wp + (size_t)(kq + u) * 64));
                w[u] = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
#pragma unroll
        for (int j = 0; j < XI; ++j) {
            const int i = l + 64 * j, r = i / U, q = i % U;
            x[j] = (i < 4 * RG * U && r < nrows)
                       ? *reinterpret_cast<const float4 *>(arow + (size_t)r * DQ_FC1_IN + 4 * (kq + q))
                       : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto consume = [&](const float4 (&w)[U], const float4 (&xin)[XI], float (*x_lds)[DQ_FC1_U * 4]) {
#pragma unroll
        for (int j = 0; j < XI; ++j) {
            const int i = l + 64 * j;
            if (i < 4 * RG * U) *reinterpret_cast<float4 *>(&x_lds[i / U][4 * (i % U)]) = xin[j];
        }
        __syncthreads();   // one wave per workgroup: orders the LDS round trip
        // The broadcast reads of k-quad u + 1 are requested BEFORE the matrix instructions of k-quad u are issued (two
        // register sets, the order pinned): left to itself the compiler re-used one set and waited out every read - a lone
        // wave on its SIMD then spent ~150 of its ~290 cycles per k-quad on LDS latency, which is what bounded the launch
        // (its time did not follow the bytes, the wave count, the ring depth or the MFMA count: profiles/r04_experiments.md).
        float4 x[2][NG];
        auto read_x = [&](float4 (&dst)[NG], int u) {
#pragma unroll
            for (int g = 0; g < NG; ++g)
                dst[g] = *reinterpret_cast<const float4 *>(&x_lds[4 * ((HALF ? 2 * g : g) + hrow) + (l & 3)][4 * u]);
        };
        read_x(x[0], 0);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (u + 1 < U) read_x(x[(u + 1) & 1], u + 1);
            __builtin_amdgcn_sched_barrier(0);
            const float4 (&xc)[NG] = x[u & 1];
#pragma unroll
            for (int g = 0; g < NG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(xc[g].x, w[u].x, acc[g], 0, 0, 0);
#pragma unroll
            for (int g = 0; g < NG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(xc[g].y, w[u].y, acc[g], 0, 0, 0);
#pragma unroll
            for (int g = 0; g < NG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(xc[g].z, w[u].z, acc[g], 0, 0, 0);
#pragma unroll
            for (int g = 0; g < NG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(xc[g].w, w[u].w, acc[g], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // the ring: chunk c lives in buffer c % NB; before chunk c is consumed chunk c + NB - 1 is requested into the buffer
    // chunk c - 1 has just left (the order pinned with sched_barrier; serially - load, wait, compute - the 56 memory
    // latencies of a wave added up to half the kernel, and with two buffers a wave that is alone on its SIMD - a Co-GA
    // launch has fewer waves than the chip has SIMDs - still waited out most of each HBM round trip)
#pragma unroll
    for (int b = 0; b < NB - 1; ++b) issue(wv[b], xr[b], b * U);
#pragma nounroll
    for (int c0 = 0; c0 < NCHUNK; c0 += NB) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int nxt = c0 + b + NB - 1;
            if (nxt < NCHUNK) issue(wv[(b + NB - 1) % NB], xr[(b + NB - 1) % NB], nxt * U);   // wave-uniform
            __builtin_amdgcn_sched_barrier(0);
            consume(wv[b], xr[b], xs[b & 1]);
        }
    }
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 4 * ((HALF ? 2 * g : g) + hrow) + i;
            if (r < nrows) hid[(size_t)(task.row_begin + r) * DQ_FC1_OUT + col] = relu_keep_nan(acc[g][i]);
        }
}

// fc1 + ReLU: grid (task, 8), ONE wavefront per workgroup: it owns outputs [64*ob, +64) and streams their 802 KB
// ([784][64][4] tile) exactly once for the task's <= 16 rows (only 8 wavefronts exist per net, so the memory-level
// parallelism has to come from depth).  The activations of a chunk (rows x U k-quads) are staged in LDS with coalesced
// loads and read back as broadcasts (scalar loads of them serialise: 12 500 dependent s_loads per wave).
// The ring, two instantiations: a launch with at most one wave per SIMD (Co-GA shard: 90 tasks = 720 waves) keeps 8 chunks
// of 7 KiB (~300 registers, nothing else would hide a lone wave's round trips); a launch with more waves than SIMDs (a
// Co-ES cohort: 133 tasks = 1064 waves) keeps two (< 256 registers: every wave resident; with the deep ring its last 40
// waves ran as a second round, 171 against 141 us).  Chunks of 7 instead of 14 pieces: 170 -> 141 us for that launch.
template <int NB, bool HALF = false>
__global__ __launch_bounds__(64, (HALF || NB <= DQ_FC1_NB_MANY) ? 2 : 1) void dqn_fc1_kernel(const float *slab, const coevo_dqn_task *tasks, int n_tasks, int C,
                                                      int n_actions, const float *act, float *hid)
{
    __shared__ __attribute__((aligned(16))) float xs[2][DQ_RMAX][DQ_FC1_U * 4];
    // XCD x takes a contiguous range of tasks (gridDim.x is a multiple of 8, so the output block blockIdx.y does not change
    // the XCD): the several <= 16-row tasks of a net that acts in many games stream the same matrix block through ONE L2
#ifdef DQ_NO_XCD_MAP
    const int ti = blockIdx.x;
#else
    const int ti = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
#endif
    if (ti >= n_tasks) return;
    const coevo_dqn_task task = tasks[ti];
    // a net that acts in more than 16 games owns several tasks, adjacent in the table (workgroup-uniform)
    const bool shared_net = !DQ_FC1_ALLNT && ((ti > 0 && tasks[ti - 1].net_off == task.net_off) ||
                                              (ti + 1 < n_tasks && tasks[ti + 1].net_off == task.net_off));
    const float *net = slab + task.net_off;
    const DqnLayout L = dqn_layout(C, n_actions);
    const int ob = blockIdx.y, l = threadIdx.x;
#ifdef COEVO_PHASE_STAMPS
    // diagnostic build only (round 4: tools/dqn_fc1_clock.py, retired - profiles/r04_experiments.md): shader-clock and 100 MHz stamps of wave (task ti, block 0) -> the clock
    // the chip holds during this launch
    if (ob == 0 && l == 0 && ti < 1024) {
        g_dqn_stamps[ti * 16 + 10] = __builtin_amdgcn_s_memrealtime();
        g_dqn_stamps[ti * 16 + 11] = __builtin_amdgcn_s_memtime();
    }
#endif
    // MFMA groups of this wave (workgroup-uniform, as is shared_net: one straight-line instantiation each)
    const int ng = HALF ? (task.n_rows + 7) >> 3 : (task.n_rows + 3) >> 2;
    if constexpr (HALF) {
        if (shared_net) {
            if (ng == 1) dqn_fc1_body<1, NB, true, true>(net, L, task, act, hid, xs, ob, l);
            else dqn_fc1_body<2, NB, true, true>(net, L, task, act, hid, xs, ob, l);
        } else {
            if (ng == 1) dqn_fc1_body<1, NB, false, true>(net, L, task, act, hid, xs, ob, l);
            else dqn_fc1_body<2, NB, false, true>(net, L, task, act, hid, xs, ob, l);
        }
    } else if (shared_net) {
        if (ng == 1) dqn_fc1_body<1, NB, true>(net, L, task, act, hid, xs, ob, l);
        else if (ng == 2) dqn_fc1_body<2, NB, true>(net, L, task, act, hid, xs, ob, l);
        else if (ng == 3) dqn_fc1_body<3, NB, true>(net, L, task, act, hid, xs, ob, l);
        else dqn_fc1_body<4, NB, true>(net, L, task, act, hid, xs, ob, l);
    } else {
        if (ng == 1) dqn_fc1_body<1, NB, false>(net, L, task, act, hid, xs, ob, l);
        else if (ng == 2) dqn_fc1_body<2, NB, false>(net, L, task, act, hid, xs, ob, l);
        else if (ng == 3) dqn_fc1_body<3, NB, false>(net, L, task, act, hid, xs, ob, l);
        else dqn_fc1_body<4, NB, false>(net, L, task, act, hid, xs, ob, l);
    }
#ifdef COEVO_PHASE_STAMPS
    if (ob == 0 && l == 0 && ti < 1024) {
        g_dqn_stamps[ti * 16 + 12] = __builtin_amdgcn_s_memrealtime();
        g_dqn_stamps[ti * 16 + 13] = __builtin_amdgcn_s_memtime();
    }
#endif
}

// fc1 + ReLU over the TILED fc1 block (dqn_common.hip.h: the layout of a Co-GA engine, whose tasks carry 10 / 16 rows): grid
// (task, 8), one wavefront per workgroup as dqn_fc1_kernel, but the wave's 64 outputs are four 16-column tiles on
// v_mfma_f32_16x16x4 (bit-identical to the sequential-k chain: tools/mfma16_chain_probe.hip).  Per super-quad (16 k): four
// 16-byte weight loads per lane - piece j of lane (c, kk) of tile T IS the B operand of k-quad 4 Q + j - one ds_read_b128 of
// the activations (staged per chunk as [Q][kk][row][j]: lane (c = row, kk) reads its four A operands in one piece) and sixteen
// matrix instructions: no vector instruction touches an operand.  Round 4 found the streamed form's 16-row waves bound by
// their own matrix issue (12 544 v_mfma_f32_4x4x1 at 13.6 cycles for a lone wave = 71 us of an 84 us wave life); here a wave
// issues 3 136 instructions of 32 cycles on four independent accumulators = 43 us, and the launch sits on its weight stream.
template <int NB, bool SHARED_NET>
__device__ __forceinline__ void dqn_fc1_tiled_body(const float *net, const DqnLayout &L, const coevo_dqn_task &task,
                                                   const float *act, float *hid, float (*xs)[DQ_FC1_TQ][4][16][4], int ob, int l)
{
    constexpr int CQ = DQ_FC1_TQ, NSQ = 196, NCHUNK = NSQ / CQ;
    static_assert(NSQ % CQ == 0 && NCHUNK % NB == 0 && NB >= 2, "whole rounds of the ring");
    constexpr int XI = (16 * 4 * CQ + 63) / 64;   // activation pieces (row, k-quad) of a chunk per lane
    typedef float f32x4_nt __attribute__((ext_vector_type(4))) __attribute__((unused));
    const int nrows = task.n_rows, c = l & 15, kk = l >> 4;
    f32x4_acc acc[4];
#pragma unroll
    for (int T = 0; T < 4; ++T) {
        const float bb = net[L.bf + 64 * ob + 16 * T + c];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[T][i] = bb;
    }
    const float4 *wp = reinterpret_cast<const float4 *>(net + L.wf) + (size_t)ob * NSQ * 4 * 64 + l;
    const float *arow = act + (size_t)task.row_begin * DQ_FC1_IN;
    float4 wv[NB][CQ][4], xr[NB][XI];
    auto issue = [&](float4 (&w)[CQ][4], float4 (&x)[XI], int sq) {
#pragma unroll
        for (int q = 0; q < CQ; ++q)
#pragma unroll
            for (int T = 0; T < 4; ++T) {
                const float4 *p = wp + (size_t)((sq + q) * 4 + T) * 64;
                if constexpr (SHARED_NET) {
                    w[q][T] = *p;
                } else {
                    const f32x4_nt v = __builtin_nontemporal_load(reinterpret_cast<const f32x4_nt *>(p));
                    w[q][T] = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
#pragma unroll
        for (int j = 0; j < XI; ++j) {   // piece i: row i / (4 CQ), k-quad 4 sq + i % (4 CQ) of the chunk (coalesced per row)
            const int i = l + 64 * j, r = i / (4 * CQ), q = i % (4 * CQ);
            x[j] = (i < 16 * 4 * CQ && r < nrows)
                       ? *reinterpret_cast<const float4 *>(arow + (size_t)r * DQ_FC1_IN + 4 * (4 * sq + q))
                       : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto consume = [&](const float4 (&w)[CQ][4], const float4 (&xin)[XI], float (*x_lds)[4][16][4]) {
#pragma unroll
        for (int j = 0; j < XI; ++j) {   // element e of k-quad 4 Q + jq is k = 16 Q + 4 jq + e: plane kk = e, slot jq
            const int i = l + 64 * j, r = i / (4 * CQ), q = i % (4 * CQ);
            if (i < 16 * 4 * CQ) {
                x_lds[q >> 2][0][r][q & 3] = xin[j].x;
                x_lds[q >> 2][1][r][q & 3] = xin[j].y;
                x_lds[q >> 2][2][r][q & 3] = xin[j].z;
                x_lds[q >> 2][3][r][q & 3] = xin[j].w;
            }
        }
        __syncthreads();   // one wave per workgroup: orders the LDS round trip
        float4 a[CQ];
#pragma unroll
        for (int q = 0; q < CQ; ++q) a[q] = *reinterpret_cast<const float4 *>(&x_lds[q][kk][c][0]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < CQ; ++q) {
#pragma unroll
            for (int T = 0; T < 4; ++T) acc[T] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].x, w[q][T].x, acc[T], 0, 0, 0);
#pragma unroll
            for (int T = 0; T < 4; ++T) acc[T] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].y, w[q][T].y, acc[T], 0, 0, 0);
#pragma unroll
            for (int T = 0; T < 4; ++T) acc[T] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].z, w[q][T].z, acc[T], 0, 0, 0);
#pragma unroll
            for (int T = 0; T < 4; ++T) acc[T] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].w, w[q][T].w, acc[T], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
#pragma unroll
    for (int b = 0; b < NB - 1; ++b) issue(wv[b], xr[b], b * CQ);
#pragma nounroll
    for (int c0 = 0; c0 < NCHUNK; c0 += NB) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int nxt = c0 + b + NB - 1;
            if (nxt < NCHUNK) issue(wv[(b + NB - 1) % NB], xr[(b + NB - 1) % NB], nxt * CQ);   // wave-uniform
            __builtin_amdgcn_sched_barrier(0);
            consume(wv[b], xr[b], xs[b & 1]);
        }
    }
#pragma unroll
    for (int T = 0; T < 4; ++T)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 4 * kk + i;
            if (r < nrows) hid[(size_t)(task.row_begin + r) * DQ_FC1_OUT + 64 * ob + 16 * T + c] = relu_keep_nan(acc[T][i]);
        }
}

template <int NB>
__global__ __launch_bounds__(64, 1) void dqn_fc1_tiled_kernel(const float *slab, const coevo_dqn_task *tasks, int n_tasks, int C,
                                                              int n_actions, const float *act, float *hid)
{
    __shared__ __attribute__((aligned(16))) float xs[2][DQ_FC1_TQ][4][16][4];
#ifdef DQ_NO_XCD_MAP
    const int ti = blockIdx.x;
#else
    const int ti = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);   // as dqn_fc1_kernel: a net's tasks meet in one L2
#endif
    if (ti >= n_tasks) return;
    const coevo_dqn_task task = tasks[ti];
    const bool shared_net = !DQ_FC1_ALLNT && ((ti > 0 && tasks[ti - 1].net_off == task.net_off) ||
                                              (ti + 1 < n_tasks && tasks[ti + 1].net_off == task.net_off));
    const float *net = slab + task.net_off;
    const DqnLayout L = dqn_layout(C, n_actions);
    if (shared_net) dqn_fc1_tiled_body<NB, true>(net, L, task, act, hid, xs, blockIdx.y, threadIdx.x);
    else dqn_fc1_tiled_body<NB, false>(net, L, task, act, hid, xs, blockIdx.y, threadIdx.x);
}

// fc1 + ReLU of a SMALL launch (a Co-ES generation's ten evaluation games: one task per agent-step, 600 dependent launches
// per generation).  Such a launch is bound by the LATENCY of the k chain, not by throughput: 8 waves with 3 - 4 independent
// accumulators each advance one k per dependent v_mfma_f32_4x4x1 (~40 cycles): 3136 x 40 cycles = 52 us + the stream =
// 75 us per launch on an empty chip.  Here a wave owns 16 outputs x all <= 16 rows as ONE v_mfma_f32_16x16x4 tile, which
// advances FOUR k per dependent instruction (784 x 40 cycles = 13 us; the same sequential-k bits, tools/mfma16_chain_probe),
// and a task has 32 waves.  Operands: lane (c = l % 16, kk = l / 16): A = x[row c][4 q + kk] (LDS), B = W[16 jb + c][4 q + kk]
// = element kk of the tile's 16-byte piece of lane 16 jb + c (one dword per lane, 256 contiguous bytes per wave).
// Measured on the evaluation launch (1 task x 10 rows): 75.5 us (wide kernel) -> 54.7 (16 outputs per wave on 4x4x1, one
// accumulator: the same 40-cycle chain) -> 37.1 (16x16x4, activations staged in LDS per chunk) -> 36.1 (both operands
// as one dword per k-quad from memory) -> this form.
template <int NB>
__global__ __launch_bounds__(64) void dqn_fc1_narrow_kernel(const float *slab, const coevo_dqn_task *tasks, int n_tasks, int C,
                                                             int n_actions, const float *act, float *hid)
{
    constexpr int U = DQ_FC1_U;
    const int ti = blockIdx.x;
    const coevo_dqn_task task = tasks[ti];
    const float *net = slab + task.net_off;
    const DqnLayout L = dqn_layout(C, n_actions);
    const int ob = blockIdx.y >> 2, jb = blockIdx.y & 3, l = threadIdx.x, c = l & 15, kk = l >> 4, col = 16 * jb + c;
    const int nrows = task.n_rows;
    f32x4_acc acc;
    {
        const float bb = net[L.bf + 64 * ob + col];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = bb;
    }
    // Both operands straight from memory (the activations are L2 hits; rows past the task's last repeat it - their sums
    // are never stored): no LDS staging, no barrier between the 784 dependent matrix instructions.  A lane requests FOUR
    // k-quads at a time - lane (c, g) the 16-byte pieces W[col c][4 (4 Q + g) ..+3] and x[row c][..] - and a 4x4 (register x
    // 16-lane row) transpose (two v_permlane32_swap + two v_permlane16_swap each) turns them into the operands of k-quads
    // 4 Q .. 4 Q + 3.  (One dword per lane and k-quad: 1568 loads per wave, and a wave's 63 countable outstanding loads
    // covered 31 k-quads = 0.5 us of chain: 33 us per launch instead of the chain's 13.)
    typedef unsigned u32x2_s __attribute__((ext_vector_type(2)));
    constexpr int NSQ = 784 / 4, NCH = NSQ / U;   // super-quads (four k-quads), chunks of U of them
    static_assert(NSQ % U == 0 && NCH % NB == 0, "whole rounds of the ring");
    const float4 *wp = reinterpret_cast<const float4 *>(net + L.wf) + (size_t)ob * 784 * 64 + col + (size_t)kk * 64;
    const float4 *xp = reinterpret_cast<const float4 *>(act + (size_t)(task.row_begin + min(c, nrows - 1)) * DQ_FC1_IN) + kk;
    float4 wv[NB][U], xv[NB][U];
    auto issue = [&](float4 (&w)[U], float4 (&x)[U], int sq) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            w[u] = wp[(size_t)(sq + u) * 256];   // piece (k-quad 4 (sq + u) + kk, column col); plain loads: read every step
            x[u] = xp[4 * (sq + u)];
        }
    };
    auto transpose4 = [](const float4 &v, float (&o)[4]) {
        const u32x2_s s02 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v.x), __float_as_uint(v.z), false, false);
        const u32x2_s s13 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v.y), __float_as_uint(v.w), false, false);
        const u32x2_s y01 = __builtin_amdgcn_permlane16_swap(s02[0], s13[0], false, false);
        const u32x2_s y23 = __builtin_amdgcn_permlane16_swap(s02[1], s13[1], false, false);
        o[0] = __uint_as_float(y01[0]); o[1] = __uint_as_float(y01[1]); o[2] = __uint_as_float(y23[0]); o[3] = __uint_as_float(y23[1]);
    };
#pragma unroll
    for (int b = 0; b < NB - 1; ++b) issue(wv[b], xv[b], b * U);
#pragma nounroll
    for (int c0 = 0; c0 < NCH; c0 += NB) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int nxt = min(c0 + b + NB - 1, NCH - 1);   // (clamped, unconditional: straight-line code)
            issue(wv[(b + NB - 1) % NB], xv[(b + NB - 1) % NB], nxt * U);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float bq[4], aq[4];
                transpose4(wv[b][u], bq);
                transpose4(xv[b][u], aq);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[j], bq[j], acc, 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (4 * kk + i < nrows)
            hid[(size_t)(task.row_begin + 4 * kk + i) * DQ_FC1_OUT + 64 * ob + col] = relu_keep_nan(acc[i]);
}

// dqn_fc1_narrow_kernel over the TILED fc1 block: the 16-byte piece of lane (c, kk) of tile (ob, Q, T) is the B operand of
// k-quads 4 Q .. 4 Q + 3 as it is; the A operands x[row c][16 Q + 4 j + kk] still come as ONE 16-byte load per lane and a 4x4
// transpose (four v_permlane swaps per four matrix instructions, half of the streamed form's eight: they issue in the shadow of
// the dependent matrix instructions).  Two loads per super-quad instead of the five of an all-dword form matter more than the
// swaps: a wave can count 63 loads in flight, i.e. 31 super-quads = 2.3 us of chain ahead instead of 0.9 (measured: all-dword
// 22.4 us per launch, interleaved or not; streamed 25.2; this form: profiles/r05_experiments.md).
template <int NB>
__global__ __launch_bounds__(64) void dqn_fc1_narrow_tiled_kernel(const float *slab, const coevo_dqn_task *tasks, int n_tasks, int C,
                                                                   int n_actions, const float *act, float *hid)
{
    constexpr int U = DQ_FC1_U;
    const int ti = blockIdx.x;
    const coevo_dqn_task task = tasks[ti];
    const float *net = slab + task.net_off;
    const DqnLayout L = dqn_layout(C, n_actions);
    const int ob = blockIdx.y >> 2, T = blockIdx.y & 3, l = threadIdx.x, c = l & 15, kk = l >> 4, col = 16 * T + c;
    const int nrows = task.n_rows;
    f32x4_acc acc;
    {
        const float bb = net[L.bf + 64 * ob + col];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = bb;
    }
    typedef unsigned u32x2_s __attribute__((ext_vector_type(2)));
    constexpr int NSQ = 196, NCH = NSQ / U;
    static_assert(NSQ % U == 0 && NCH % NB == 0, "whole rounds of the ring");
    const float4 *wp = reinterpret_cast<const float4 *>(net + L.wf) + ((size_t)ob * NSQ * 4 + T) * 64 + l;
    const float4 *xp = reinterpret_cast<const float4 *>(act + (size_t)(task.row_begin + min(c, nrows - 1)) * DQ_FC1_IN) + kk;
    float4 wv[NB][U], xv[NB][U];
    auto issue = [&](float4 (&w)[U], float4 (&x)[U], int sq) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            w[u] = wp[(size_t)(sq + u) * 256];   // (plain loads: the same two nets every step)
            x[u] = xp[4 * (sq + u)];             // x[row c][16 (sq + u) + 4 kk .. + 3]
        }
    };
#pragma unroll
    for (int b = 0; b < NB - 1; ++b) issue(wv[b], xv[b], b * U);
    // The transpose of super-quad s + 1 is issued one swap behind each matrix instruction of super-quad s (the order pinned): a
    // dependent v_mfma_f32_16x16x4 leaves ~45 cycles in which the in-order wave can issue an independent vector instruction
    // for free; as four swaps in front of their own four matrix instructions they sat on the chain (63 cycles per step).
    float a_cur[4], a_nxt[4];
    auto tr_a = [&](const float4 &v, u32x2_s &s02, u32x2_s &s13) {
        s02 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v.x), __float_as_uint(v.z), false, false);
        s13 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v.y), __float_as_uint(v.w), false, false);
    };
    {
        u32x2_s s02, s13;
        tr_a(xv[0][0], s02, s13);
        const u32x2_s y01 = __builtin_amdgcn_permlane16_swap(s02[0], s13[0], false, false);
        const u32x2_s y23 = __builtin_amdgcn_permlane16_swap(s02[1], s13[1], false, false);
        a_cur[0] = __uint_as_float(y01[0]); a_cur[1] = __uint_as_float(y01[1]);
        a_cur[2] = __uint_as_float(y23[0]); a_cur[3] = __uint_as_float(y23[1]);
    }
#pragma nounroll
    for (int c0 = 0; c0 < NCH; c0 += NB) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int nxt = min(c0 + b + NB - 1, NCH - 1);   // (clamped, unconditional: straight-line code)
            issue(wv[(b + NB - 1) % NB], xv[(b + NB - 1) % NB], nxt * U);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                // the super-quad after this one: the next of the chunk, or the first of the next chunk (the last one of all
                // transposes a piece nobody uses)
                const float4 &vn = (u + 1 < U) ? xv[b][u + 1] : xv[(b + 1) % NB][0];
                u32x2_s s02, s13;
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[0], wv[b][u].x, acc, 0, 0, 0);
                s02 = __builtin_amdgcn_permlane32_swap(__float_as_uint(vn.x), __float_as_uint(vn.z), false, false);
                __builtin_amdgcn_sched_barrier(0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[1], wv[b][u].y, acc, 0, 0, 0);
                s13 = __builtin_amdgcn_permlane32_swap(__float_as_uint(vn.y), __float_as_uint(vn.w), false, false);
                __builtin_amdgcn_sched_barrier(0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[2], wv[b][u].z, acc, 0, 0, 0);
                const u32x2_s y01 = __builtin_amdgcn_permlane16_swap(s02[0], s13[0], false, false);
                __builtin_amdgcn_sched_barrier(0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[3], wv[b][u].w, acc, 0, 0, 0);
                const u32x2_s y23 = __builtin_amdgcn_permlane16_swap(s02[1], s13[1], false, false);
                __builtin_amdgcn_sched_barrier(0);
                a_nxt[0] = __uint_as_float(y01[0]); a_nxt[1] = __uint_as_float(y01[1]);
                a_nxt[2] = __uint_as_float(y23[0]); a_nxt[3] = __uint_as_float(y23[1]);
#pragma unroll
                for (int j = 0; j < 4; ++j) a_cur[j] = a_nxt[j];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (4 * kk + i < nrows)
            hid[(size_t)(task.row_begin + 4 * kk + i) * DQ_FC1_OUT + 64 * ob + col] = relu_keep_nan(acc[i]);
}

// output layer + first-max action: one 64-thread workgroup per (task, row) (dqn_out_row, dqn_common.hip.h).  The
// population engine does not launch it: there the output layer rides in the env-step launch (coevo_dqn_out_synth_step).
__global__ __launch_bounds__(64) void dqn_out_kernel(const float *slab, const coevo_dqn_task *tasks, int n_tasks, int C,
                                                      int n_actions, const float *hid, int32_t *actions,
                                                      float *logits, int32_t *status)
{
    __shared__ __attribute__((aligned(16))) float xs[DQ_FC1_OUT];
    __shared__ float lg[64];
    const int row = blockIdx.x;
    const coevo_dqn_task task = tasks[task_of_row(tasks, n_tasks, row)];
    const int a = dqn_out_row(slab + task.net_off, dqn_layout(C, n_actions), n_actions, hid + (size_t)row * DQ_FC1_OUT,
                              logits ? logits + (size_t)row * COEVO_DQN_LOGIT_STRIDE : nullptr, status, xs, lg, threadIdx.x);
    if (threadIdx.x == 0) actions[row] = a;
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

// conv3 activations + fc1 outputs of every row; a small launch also passes conv1's sums and conv2's activations through it
static int64_t dqn_small_extra_floats(int n_rows_total)
{
    return n_rows_total <= DQ_SMALL_MAX_ROWS ? (int64_t)n_rows_total * (32 * 400 + 64 * DQ_P2) : 0;
}

extern "C" int64_t coevo_dqn_workspace_bytes(int n_rows_total)
{
    return n_rows_total > 0 ? ((int64_t)n_rows_total * (DQ_FC1_IN + DQ_FC1_OUT) + dqn_small_extra_floats(n_rows_total)) * 4
                            : COEVO_ERR_ARG;
}

extern "C" int coevo_dqn_pack(const float *flat, float *slab, int n, int c_arg, int n_actions, void *stream)
{
    const int C = dqn_channels(c_arg), tiled = dqn_fc1_tiled(c_arg);
    if (!flat || !slab || n <= 0 || !dqn_shape_ok(C, n_actions) || (c_arg & ~(0xff | COEVO_DQN_FC1_TILED))) return COEVO_ERR_ARG;
    const dim3 grid((unsigned)((dqn_layout(C, n_actions).stride + 255) / 256), (unsigned)n);
    hipLaunchKernelGGL(dqn_pack_kernel, grid, dim3(256), 0, (hipStream_t)stream, flat, slab, C, n_actions, tiled);
    COEVO_HIP_CHECK(hipGetLastError());
    return COEVO_OK;
}

extern "C" int coevo_dqn_forward_argmax(const float *slab, const coevo_dqn_task *tasks, int n_tasks,
                                        int max_rows_per_task, int n_rows_total, int C, int n_actions,
                                        const uint8_t *frames, int32_t *actions, float *logits, int32_t *status,
                                        void *workspace, void *stream)
{
    return coevo_dqn_forward_argmax_timed(slab, tasks, n_tasks, max_rows_per_task, n_rows_total, C, n_actions, frames,
                                          actions, logits, status, workspace, nullptr, 0, stream);
}

// conv stack + fc1 (+ the output layer when `actions` is given)
static int dqn_forward_launch(const float *slab, const coevo_dqn_task *tasks, int n_tasks, int max_rows_per_task,
                              int n_rows_total, int c_arg, int n_actions, const uint8_t *frames, int32_t *actions, float *logits,
                              int32_t *status, void *workspace, void *timing_ctx, int timed_kernel, void *stream)
{
    const int C = dqn_channels(c_arg), tiled = dqn_fc1_tiled(c_arg);   // (the slab's fc1 layout rides in the channel argument)
    if (c_arg & ~(0xff | COEVO_DQN_FC1_TILED)) return COEVO_ERR_ARG;
    if (!slab || !tasks || !frames || !workspace) return COEVO_ERR_ARG;
    if (n_tasks <= 0 || n_rows_total <= 0 || !dqn_shape_ok(C, n_actions)) return COEVO_ERR_ARG;
    if (max_rows_per_task < 1 || max_rows_per_task > DQ_RMAX) return COEVO_ERR_ARG;
    float *act = static_cast<float *>(workspace);
    float *hid = act + (size_t)n_rows_total * DQ_FC1_IN;
    hipStream_t s = (hipStream_t)stream;
    if (timing_ctx && (timed_kernel < 0 || timed_kernel > 1)) return COEVO_ERR_ARG;
    if (timing_ctx && timed_kernel == 0 && coevo_timing_begin(timing_ctx, stream) != COEVO_OK) return COEVO_ERR_HIP;
    const dim3 cg(8 * ((n_rows_total + 7) / 8)), cb(512);   // a multiple of 8: the kernel's XCD-aware row mapping
    if (n_rows_total <= DQ_SMALL_MAX_ROWS) {   // three launches that spread each frame over the chip
        float *a1raw = hid + (size_t)n_rows_total * DQ_FC1_OUT, *a2g = a1raw + (size_t)n_rows_total * (32 * 400);
        const dim3 g1(n_rows_total, 25), g2(n_rows_total, 2);
        if (C == 4) hipLaunchKernelGGL((dqn_small_conv1_kernel<4, 4>), g1, dim3(64), 0, s, slab, tasks, n_tasks, C, n_actions, frames, a1raw);
        else if (C < 4) hipLaunchKernelGGL((dqn_small_conv1_kernel<4, 0>), g1, dim3(64), 0, s, slab, tasks, n_tasks, C, n_actions, frames, a1raw);
        else if (C == 6) hipLaunchKernelGGL((dqn_small_conv1_kernel<6, 6>), g1, dim3(64), 0, s, slab, tasks, n_tasks, C, n_actions, frames, a1raw);
        else hipLaunchKernelGGL((dqn_small_conv1_kernel<6, 0>), g1, dim3(64), 0, s, slab, tasks, n_tasks, C, n_actions, frames, a1raw);
        hipLaunchKernelGGL(dqn_small_conv2_kernel, g2, cb, 0, s, slab, tasks, n_tasks, C, n_actions, a1raw, a2g);
        hipLaunchKernelGGL(dqn_small_conv3_kernel, g2, cb, 0, s, slab, tasks, n_tasks, C, n_actions, a2g, act);
    }
    else if (C == 4) hipLaunchKernelGGL((dqn_conv_kernel<4, 4>), cg, cb, 0, s, slab, tasks, n_tasks, n_rows_total, C, n_actions, frames, act);
    else if (C < 4) hipLaunchKernelGGL((dqn_conv_kernel<4, 0>), cg, cb, 0, s, slab, tasks, n_tasks, n_rows_total, C, n_actions, frames, act);
    else if (C == 6) hipLaunchKernelGGL((dqn_conv_kernel<6, 6>), cg, cb, 0, s, slab, tasks, n_tasks, n_rows_total, C, n_actions, frames, act);
    else hipLaunchKernelGGL((dqn_conv_kernel<6, 0>), cg, cb, 0, s, slab, tasks, n_tasks, n_rows_total, C, n_actions, frames, act);
    if (timing_ctx && timed_kernel == 0 && coevo_timing_end(timing_ctx, stream) != COEVO_OK) return COEVO_ERR_HIP;
    if (timing_ctx && timed_kernel == 1 && coevo_timing_begin(timing_ctx, stream) != COEVO_OK) return COEVO_ERR_HIP;
    const dim3 fg(8 * ((n_tasks + 7) / 8), 8);
    if (tiled && n_tasks <= DQ_FC1_NARROW_MAX_TASKS)   // the tiled layout: v_mfma_f32_16x16x4 without operand moves at every size
        hipLaunchKernelGGL(dqn_fc1_narrow_tiled_kernel<DQ_FC1_NBN>, dim3(n_tasks, 32), dim3(64), 0, s, slab, tasks, n_tasks, C,
                           n_actions, act, hid);
    else if (tiled)
        hipLaunchKernelGGL(dqn_fc1_tiled_kernel<DQ_FC1_TNB>, fg, dim3(64), 0, s, slab, tasks, n_tasks, C, n_actions, act, hid);
    else if (n_tasks <= DQ_FC1_NARROW_MAX_TASKS)
        hipLaunchKernelGGL(dqn_fc1_narrow_kernel<DQ_FC1_NBN>, dim3(n_tasks, 32), dim3(64), 0, s, slab, tasks, n_tasks, C, n_actions,
                           act, hid);
    else if (DQ_FC1_HALF && n_tasks * 8 <= 1024)
        hipLaunchKernelGGL((dqn_fc1_kernel<DQ_FC1_NB_HALF, true>), dim3(fg.x, 16), dim3(64), 0, s, slab, tasks, n_tasks, C, n_actions,
                           act, hid);
    else if (n_tasks * 8 <= 1024) hipLaunchKernelGGL(dqn_fc1_kernel<DQ_FC1_NB>, fg, dim3(64), 0, s, slab, tasks, n_tasks, C, n_actions, act, hid);
    else hipLaunchKernelGGL(dqn_fc1_kernel<DQ_FC1_NB_MANY>, fg, dim3(64), 0, s, slab, tasks, n_tasks, C, n_actions, act, hid);
    if (timing_ctx && timed_kernel == 1 && coevo_timing_end(timing_ctx, stream) != COEVO_OK) return COEVO_ERR_HIP;
    if (actions)
        hipLaunchKernelGGL(dqn_out_kernel, dim3(n_rows_total), dim3(64), 0, s, slab, tasks, n_tasks, C, n_actions, hid,
                           actions, logits, status);
    COEVO_HIP_CHECK(hipGetLastError());
    return COEVO_OK;
}

extern "C" int coevo_dqn_forward_argmax_timed(const float *slab, const coevo_dqn_task *tasks, int n_tasks,
                                              int max_rows_per_task, int n_rows_total, int C, int n_actions,
                                              const uint8_t *frames, int32_t *actions, float *logits, int32_t *status,
                                              void *workspace, void *timing_ctx, int timed_kernel, void *stream)
{
    if (!actions || !status) return COEVO_ERR_ARG;
    return dqn_forward_launch(slab, tasks, n_tasks, max_rows_per_task, n_rows_total, C, n_actions, frames, actions, logits,
                              status, workspace, timing_ctx, timed_kernel, stream);
}

extern "C" int coevo_dqn_forward_hidden_timed(const float *slab, const coevo_dqn_task *tasks, int n_tasks,
                                              int max_rows_per_task, int n_rows_total, int C, int n_actions,
                                              const uint8_t *frames, void *workspace, void *timing_ctx, int timed_kernel,
                                              void *stream)
{
    return dqn_forward_launch(slab, tasks, n_tasks, max_rows_per_task, n_rows_total, C, n_actions, frames, nullptr, nullptr,
                              nullptr, workspace, timing_ctx, timed_kernel, stream);
}

#ifdef COEVO_PHASE_STAMPS
extern "C" int coevo_debug_read_dqn_stamps(unsigned long long *host_out, int n_words)
{
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(coevo::g_dqn_stamps), sizeof(unsigned long long) * n_words) ==
                   hipSuccess ? 0 : -2;
}
#endif

COEVO_DEFINE_TU_FLAGS(deepqn)
