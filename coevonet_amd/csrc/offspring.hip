// K3/K4/K5/K8 - building offspring weights on the device, plus slab <-> canonical-order packing.
//
// Replaces (reference file:line): MPEAgent.clone's state_dict copy (MPE/mpe_agent.py:24-28), Agent.mutate
// (agent.py:25-29: noise = torch.normal(0, sigma, size); param += noise for EVERY parameter, LayerNorm affine
// included), Agent.mutate_ES (agent.py:51-53: Linear weights/biases only) and compute_weight_update
// (evolutionary_strategy.py:120-148).
//
// Noise is counter-based so that no weight ever crosses PCIe: eps(seed, stream, p) for canonical flat index p is
// element p%4 of Philox4x32-7(counter = (p/4, stream_lo, stream_hi, 'coev'), key = seed) pushed through a
// Box-Muller transform whose log / sincos are fmaf-only polynomials (bit-identical with oracle/coevo_oracle.c).
// A child is one streaming pass: read parent (L2/Infinity-Cache resident elite), write child once.
#include "coevo_common.hip.h"
#include "dqn_common.hip.h"
#include "philox.hip.h"

namespace coevo {

// is slab position s a LayerNorm affine parameter?
__device__ inline bool fc_slab_is_layernorm(int64_t s, int D)
{
    const int64_t g1 = fc_off_b1(D) + H1, g2 = fc_off_b2(D) + H2;
    return (s >= g1 && s < g1 + 2 * H1) || (s >= g2 && s < g2 + 2 * H2);
}

// Standard normals for the four slab positions s0..s0+3 (s0 % 4 == 0).  Everywhere except W1t the four positions
// map to four consecutive canonical indices inside one Philox block, so one block serves them; W1t (3.7 % of a
// net) pays one block per element.
__device__ inline void slab_quad_normals(uint64_t seed, uint32_t slo, uint32_t shi, int64_t s0, int D, int64_t P,
                                         float z[4])
{
    int64_t p[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) p[i] = (s0 + i < P) ? fc_slab_to_flat(s0 + i, D) : -1;
    float zz[4];
    int64_t have = -1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (p[i] < 0) { z[i] = 0.0f; continue; }
        const int64_t q = p[i] >> 2;
        if (q != have) { philox_normal4(seed, slo, shi, (uint32_t)q, zz); have = q; }
        z[i] = zz[p[i] & 3];
    }
}

// one workgroup's 1024 slab positions of child c (block bx of nblocks); shared by the single- and the multi-role launch
__device__ __forceinline__ void fc_perturb_block(const float *parent_slab, const int32_t *parent_idx, float *child_slab,
                                                 int child_first, int D, const float *sigma_dev, uint64_t seed,
                                                 uint32_t stream_lo_first, uint32_t stream_hi, int skip_layernorm,
                                                 const int32_t *gen_dev, const float *dist_ref, double *dist_partial,
                                                 int c, int bx, int nblocks, double *scratch)
{
    if (gen_dev) stream_hi += 4u * (uint32_t)(*gen_dev);  // generation-indexed noise stream without a host argument
    // flags: bit 0 = leave LayerNorm affine untouched (ES), bit 1 = antithetic pairs (extension mode): individuals 2m and
    // 2m+1 share noise stream m, the odd one takes -eps
    const bool antithetic = (skip_layernorm & 2) != 0;
    skip_layernorm &= 1;
    const uint32_t ind = stream_lo_first + (uint32_t)c;
    const uint32_t slo = antithetic ? (ind >> 1) : ind;
    const bool negate = antithetic && (ind & 1u);
    const int64_t stride = fc_stride(D), P = fc_params(D);
    const int64_t s0 = ((int64_t)bx * 256 + threadIdx.x) * 4;
    double d2 = 0.0;
    if (s0 < stride) {
        const float sigma = *sigma_dev;
        const float *par = parent_slab + (int64_t)parent_idx[c] * stride;
        float *ch = child_slab + (int64_t)(child_first + c) * stride;
        const float4 pv = *reinterpret_cast<const float4 *>(par + s0);
        float z[4];
        bool keep[4], in_dist[4];
        const int32_t o_w2 = (int32_t)fc_off_w2(D), o_b2 = (int32_t)fc_off_b2(D), s32 = (int32_t)s0;
        if (s32 >= o_w2 && s32 < o_b2) {
            // fc2 block (93.8 % of a net): W2q[jb][kq][l][0..3] are the four consecutive canonical indices
            // fc2.w[64*jb + l][4*kq .. 4*kq+3], one Philox block, never LayerNorm, never padding - 32-bit index math
            const int32_t t = s32 - o_w2;
            const int32_t l = (t >> 2) & 63, kq = (t >> 8) & 127, jb = t >> 15;
            const int32_t p0 = o_w2 + (jb * 64 + l) * H1 + kq * 4;  // multiple of 4
            philox_normal4(seed, slo, stream_hi, (uint32_t)(p0 >> 2), z);
#pragma unroll
            for (int i = 0; i < 4; ++i) { keep[i] = false; in_dist[i] = true; }
        } else {
            slab_quad_normals(seed, slo, stream_hi, s0, D, P, z);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int64_t s = s0 + i;
                const bool ln = fc_slab_is_layernorm(s, D);
                keep[i] = (s >= P) || (skip_layernorm && ln);
                in_dist[i] = s < P && !ln;
            }
        }
        float in[4] = {pv.x, pv.y, pv.z, pv.w}, out[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float noise = sigma * z[i];  // rounded first, then added (agent.py:28-29)
            out[i] = keep[i] ? in[i] : in[i] + (negate ? -noise : noise);
        }
        *reinterpret_cast<float4 *>(ch + s0) = make_float4(out[0], out[1], out[2], out[3]);
        if (dist_partial) {  // fitness-sharing distance of the new child to a reference net, while it is in registers
            const float4 rv = *reinterpret_cast<const float4 *>(dist_ref + s0);
            const float ref[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (in_dist[i]) {
                    const float d = out[i] - ref[i];
                    d2 += (double)d * (double)d;
                }
            }
        }
    }
    if (dist_partial) {  // wave-uniform
        const double tot = block_sum_f64(d2, scratch);
        if (threadIdx.x == 0) dist_partial[(size_t)c * nblocks + bx] = tot;
    }
}

__global__ __launch_bounds__(256) void fc_perturb_kernel(const float *parent_slab, const int32_t *parent_idx,
                                                          float *child_slab, int child_first, int D,
                                                          const float *sigma_dev, uint64_t seed,
                                                          uint32_t stream_lo_first, uint32_t stream_hi,
                                                          int skip_layernorm, const int32_t *gen_dev,
                                                          const float *dist_ref, double *dist_partial)
{
    __shared__ double scratch[4];
    fc_perturb_block(parent_slab, parent_idx, child_slab, child_first, D, sigma_dev, seed, stream_lo_first, stream_hi,
                     skip_layernorm, gen_dev, dist_ref, dist_partial, blockIdx.y, blockIdx.x, gridDim.x, scratch);
}

static_assert(sizeof(coevo_fc_perturb_job) == 72, "layout mirrored by coevonet_amd/lib.py PerturbJob");
struct PerturbJobs { coevo_fc_perturb_job j[COEVO_MAX_JOBS]; };
// blockIdx.z = job (role); the grid covers the largest job, the surplus workgroups of the others leave at once
__global__ __launch_bounds__(256) void fc_perturb_multi_kernel(PerturbJobs jobs, uint64_t seed, int skip_layernorm,
                                                                const int32_t *gen_dev)
{
    __shared__ double scratch[4];
    const coevo_fc_perturb_job &jb = jobs.j[blockIdx.z];
    const int nblocks = (int)((fc_stride(jb.D) / 4 + 255) / 256);
    if ((int)blockIdx.x >= nblocks || (int)blockIdx.y >= jb.n_children) return;   // workgroup-uniform
    fc_perturb_block(jb.parent_slab, jb.parent_idx, jb.child_slab, jb.child_first, jb.D, jb.sigma_dev, seed,
                     jb.stream_lo_first, jb.stream_hi, skip_layernorm, gen_dev, jb.dist_ref, jb.dist_partial, blockIdx.y,
                     blockIdx.x, nblocks, scratch);
}

// New elites without gathering them from whichever GPU evaluated them: elite e of this generation is individual
// id = order[e] of the population that was bred from `elite_prev` one generation ago, so any rank can rebuild it:
//   id == 0      -> last generation's best, elite_prev[0], unchanged
//   id >= 1      -> elite_prev[(id-1) % E] + sigma_prev * eps(seed, stream (id-1, stream_hi_prev))   (same bits as the
//                   child the owning rank materialised: same kernel arithmetic, same counters)
__global__ __launch_bounds__(256) void fc_rebuild_elites_kernel(const float *elite_prev, const int32_t *order,
                                                                 float *elite_new, int E, int D,
                                                                 const float *sigma_prev_dev, uint64_t seed,
                                                                 uint32_t stream_hi_prev, const int32_t *gen_dev)
{
    if (gen_dev) stream_hi_prev += 4u * (uint32_t)(*gen_dev - 1);
    const int e = blockIdx.y;
    const int64_t stride = fc_stride(D), P = fc_params(D);
    const int64_t s0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (s0 >= stride) return;
    const int id = order[e];
    float *dst = elite_new + (int64_t)e * stride;
    if (id == 0) {
        *reinterpret_cast<float4 *>(dst + s0) = *reinterpret_cast<const float4 *>(elite_prev + s0);
        return;
    }
    const int c = id - 1;
    const float sigma = *sigma_prev_dev;
    const float4 pv = *reinterpret_cast<const float4 *>(elite_prev + (int64_t)(c % E) * stride + s0);
    float z[4];
    slab_quad_normals(seed, (uint32_t)c, stream_hi_prev, s0, D, P, z);
    const float in[4] = {pv.x, pv.y, pv.z, pv.w};
    float out[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) out[i] = (s0 + i >= P) ? in[i] : in[i] + sigma * z[i];
    *reinterpret_cast<float4 *>(dst + s0) = make_float4(out[0], out[1], out[2], out[3]);
}

__global__ __launch_bounds__(256) void fc_gather_kernel(const float *src_slab, const int32_t *src_idx,
                                                         float *dst_slab, int dst_first, int64_t stride)
{
    const int c = blockIdx.y;
    const int64_t s0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (s0 >= stride) return;
    const float4 v = *reinterpret_cast<const float4 *>(src_slab + (int64_t)src_idx[c] * stride + s0);
    *reinterpret_cast<float4 *>(dst_slab + (int64_t)(dst_first + c) * stride + s0) = v;
}

// ---- elites -> elite buffer, HoF FIFO push, best -> pop[0] for up to three roles in ONE launch (instead of five net
// copies per role).  A thread owns one 16-byte piece of every net it touches: it reads its piece of all sources first,
// so the in-place HoF shift and a best individual that already sits in pop[0] need no second buffer.
struct GaPromoteArgs {
    coevo_ga_promote_role role[3];
    int E, hof;
    // rebuild mode (coevo_ga_promote_rebuild): the new elites are not in `elite` yet - elite[k] is individual order[k] of the
    // generation just evaluated, i.e. the unchanged best (id 0 = old elite 0) or child c = id - 1 = old elite[c % E] +
    // sigma * noise(c), regenerated here from the OLD elites in place
    const float *sigma;          // [3] device: what that generation's children were bred with; NULL: not rebuild mode
    const int32_t *gen_dev;      // device generation counter g (the children were bred with streams of g - 1), or NULL
    uint64_t seed;
    uint32_t stream_hi_prev;     // + role index (+ 4 (g - 1) with gen_dev)
    int32_t *tick;               // NULL, or the generation counter this launch increments (nothing in it reads the counter
                                 // after the first instructions of its blocks: see ga_promote_kernel)
};
constexpr int PROMOTE_MAX_E = 8, PROMOTE_MAX_HOF = 16;

// A thread's 16-byte pieces are held in the frames of a compile-time recursion - every load on the way down, every store
// on the way back - not in arrays: as float4 e[8], h[16] filled by unrolled loops the compiler left both in scratch
// (400 bytes per lane).  Loads take clamped indices and stores are unconditional; a surplus level (index >= count)
// rewrites the last slot with the value it holds anyway or that the caller overwrites next (same thread, same address,
// program order).
template <int I>
__device__ __forceinline__ void promote_hof_shift(float *hof, int64_t stride, int64_t s0, int n)
{
    if constexpr (I < PROMOTE_MAX_HOF) {
        const float4 v = *reinterpret_cast<const float4 *>(hof + (int64_t)(I < n ? I : n - 1) * stride + s0);
        promote_hof_shift<I + 1>(hof, stride, s0, n);
        *reinterpret_cast<float4 *>(hof + (int64_t)(I < n ? I - 1 : n - 1) * stride + s0) = v;
    }
}

// elite[k] = src[idx(k)] for k < E (k descending on the way back); returns src[idx(0)]
template <int K>
__device__ __forceinline__ float4 promote_elites(const float *src, const int32_t *order, float *elite, bool store,
                                                 int64_t stride, int64_t s0, int E)
{
    if constexpr (K < PROMOTE_MAX_E) {
        const int kc = K < E ? K : E - 1;
        const float4 v = *reinterpret_cast<const float4 *>(src + (int64_t)(order ? order[kc] : kc) * stride + s0);
        promote_elites<K + 1>(src, order, elite, store, stride, s0, E);
        if (store) *reinterpret_cast<float4 *>(elite + (int64_t)kc * stride + s0) = v;
        return v;
    } else {
        return make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// rebuild mode: every OLD elite piece a new elite needs is loaded on the way down, the new pieces are stored on the way back
// (each slab position depends only on the old elites' values at the same position: in place, no second elite buffer)
template <int K>
__device__ __forceinline__ float4 promote_rebuild(float *elite, const int32_t *order, int64_t stride, int64_t s0, int E, int D,
                                                  float sigma, uint64_t seed, uint32_t shi)
{
    if constexpr (K < PROMOTE_MAX_E) {
        const int kc = K < E ? K : E - 1;
        const int id = order[kc];
        const int c = id > 0 ? id - 1 : 0;
        const float4 pv = *reinterpret_cast<const float4 *>(elite + (int64_t)(id > 0 ? c % E : 0) * stride + s0);
        promote_rebuild<K + 1>(elite, order, stride, s0, E, D, sigma, seed, shi);
        float4 nv = pv;
        if (id > 0 && K < E) {   // (a surplus level re-stores slot E - 1 with the piece it loaded; level E - 1 overwrites it after)
            const int64_t P = fc_params(D);
            float z[4];
            slab_quad_normals(seed, (uint32_t)c, shi, s0, D, P, z);
            const float in[4] = {pv.x, pv.y, pv.z, pv.w};
            float out[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) out[i] = (s0 + i >= P) ? in[i] : in[i] + sigma * z[i];   // as fc_rebuild_elites_kernel
            nv = make_float4(out[0], out[1], out[2], out[3]);
        }
        *reinterpret_cast<float4 *>(elite + (int64_t)kc * stride + s0) = nv;
        return nv;
    } else {
        return make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

__global__ __launch_bounds__(256) void ga_promote_kernel(GaPromoteArgs a)
{
    // (field-wise scalar selects: indexing the by-value argument array dynamically copies it to scratch)
    const unsigned y = blockIdx.y;
#define PROMOTE_SEL(f) (y == 0 ? a.role[0].f : (y == 1 ? a.role[1].f : a.role[2].f))
    float *pop = PROMOTE_SEL(pop), *hof = PROMOTE_SEL(hof), *elite = PROMOTE_SEL(elite);
    const int32_t *order = PROMOTE_SEL(order);
    const int D = PROMOTE_SEL(D), from_pop = PROMOTE_SEL(elites_from_pop), to_pop0 = PROMOTE_SEL(best_to_pop0);
#undef PROMOTE_SEL
    const int64_t stride = fc_stride(D);
    const int64_t s0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (s0 >= stride) return;
    // every source piece is read before the first store that could alias it (a best individual that already sits in
    // pop[0], the in-place HoF shift)
    float4 e0;
    if (a.sigma && !from_pop) {
        uint32_t shi = a.stream_hi_prev + y;
        if (a.gen_dev) shi += 4u * (uint32_t)(*a.gen_dev - 1);
        e0 = promote_rebuild<0>(elite, order, stride, s0, a.E, D, a.sigma[y], a.seed, shi);
    } else {
        e0 = from_pop ? promote_elites<0>(pop, order, elite, true, stride, s0, a.E)
                      : *reinterpret_cast<const float4 *>(elite + s0);
    }
    promote_hof_shift<1>(hof, stride, s0, a.hof);
    *reinterpret_cast<float4 *>(hof + (int64_t)(a.hof - 1) * stride + s0) = e0;
    if (to_pop0) *reinterpret_cast<float4 *>(pop + s0) = e0;
    // the generation counter's tick (coevo_ga_promote_tick: only without gen_dev - no block of this launch reads the counter)
    if (a.tick && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *a.tick += 1;
}

// theta[p] += lr/(n*sigma) * sum_i fitness[i] * (pert_i[p] - theta[p]), i ascending, one fmaf per term.
// The perturbation is read back from the materialised perturbed nets (one coalesced streaming pass over n*4P bytes)
// instead of being regenerated: 0.3 ms instead of 2 ms per role at n = 1000.  LayerNorm entries are never perturbed
// (their difference is exactly 0) and are skipped.
__global__ __launch_bounds__(256) void es_update_kernel(float *theta, const float *pert_slab, int D,
                                                         const float *fitness, int n, const float *sigma_dev, float lr)
{
    const int64_t stride = fc_stride(D), P = fc_params(D);
    const int64_t s0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (s0 >= stride) return;
    const float sigma = *sigma_dev;
    const float scale = lr / ((float)n * sigma);
    const float4 tv = *reinterpret_cast<const float4 *>(theta + s0);
    const float th[4] = {tv.x, tv.y, tv.z, tv.w};
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const float *pp = pert_slab + s0;
    int i = 0;
    // the perturbed nets are read exactly once: non-temporal, and 16 rows in flight per lane (only stride/1024 = 137
    // workgroups exist - the sum over i is sequential by definition - so the memory-level parallelism comes from depth)
    typedef float f32x4_nt __attribute__((ext_vector_type(4)));
    constexpr int UE = 16;
    for (; i + UE <= n; i += UE) {
        float4 pv[UE];
#pragma unroll
        for (int u = 0; u < UE; ++u) {
            const f32x4_nt v = __builtin_nontemporal_load(reinterpret_cast<const f32x4_nt *>(pp + (int64_t)(i + u) * stride));
            pv[u] = make_float4(v[0], v[1], v[2], v[3]);
        }
#pragma unroll
        for (int u = 0; u < UE; ++u) {
            const float f = fitness[i + u];
            acc[0] = __builtin_fmaf(f, pv[u].x - th[0], acc[0]);
            acc[1] = __builtin_fmaf(f, pv[u].y - th[1], acc[1]);
            acc[2] = __builtin_fmaf(f, pv[u].z - th[2], acc[2]);
            acc[3] = __builtin_fmaf(f, pv[u].w - th[3], acc[3]);
        }
    }
    for (; i < n; ++i) {
        const float4 pv = *reinterpret_cast<const float4 *>(pp + (int64_t)i * stride);
        const float f = fitness[i];
        acc[0] = __builtin_fmaf(f, pv.x - th[0], acc[0]);
        acc[1] = __builtin_fmaf(f, pv.y - th[1], acc[1]);
        acc[2] = __builtin_fmaf(f, pv.z - th[2], acc[2]);
        acc[3] = __builtin_fmaf(f, pv.w - th[3], acc[3]);
    }
    float out[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int64_t s = s0 + c;
        out[c] = (s < P && !fc_slab_is_layernorm(s, D)) ? th[c] + scale * acc[c] : th[c];
    }
    *reinterpret_cast<float4 *>(theta + s0) = make_float4(out[0], out[1], out[2], out[3]);
}

// K5 in two steps.  Step 1: partial[c][p] = sum over the individuals i of chunk c (global chunk gc = chunk_first + c
// covers [gc*n/C, (gc+1)*n/C)), i ascending, of fitness[i] * (pert_i[p] - theta[p]), one fmaf per term starting from
// 0.  grid (stride/1024, n_chunks): C times the workgroups of the sequential form (137 of them fill half the chip), and
// a population shard computes exactly its own chunks.  Step 2 (es_apply_kernel): theta += scale * (((p_0 + p_1) + p_2)
// + ...).  chunks_total = 1 is the sequential sum of es_update_kernel, bit for bit.
__global__ __launch_bounds__(256) void es_partial_kernel(const float *theta, const float *pert_slab, int ind_first,
                                                          int64_t stride, const float *fitness_all, int n_total,
                                                          int chunks_total, int chunk_first, float *partial)
{
    const int64_t s0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (s0 >= stride) return;
    const int gc = chunk_first + blockIdx.y;
    const int i_lo = (int)((int64_t)gc * n_total / chunks_total), i_hi = (int)((int64_t)(gc + 1) * n_total / chunks_total);
    const float4 tv = *reinterpret_cast<const float4 *>(theta + s0);
    const float th[4] = {tv.x, tv.y, tv.z, tv.w};
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const float *pp = pert_slab + s0 - (int64_t)ind_first * stride;  // indexed by the GLOBAL individual
    typedef float f32x4_nt __attribute__((ext_vector_type(4)));
    constexpr int UE = 16;
    int i = i_lo;
    for (; i + UE <= i_hi; i += UE) {
        float4 pv[UE];
#pragma unroll
        for (int u = 0; u < UE; ++u) {
            const f32x4_nt v = __builtin_nontemporal_load(reinterpret_cast<const f32x4_nt *>(pp + (int64_t)(i + u) * stride));
            pv[u] = make_float4(v[0], v[1], v[2], v[3]);
        }
#pragma unroll
        for (int u = 0; u < UE; ++u) {
            const float f = fitness_all[i + u];
            acc[0] = __builtin_fmaf(f, pv[u].x - th[0], acc[0]);
            acc[1] = __builtin_fmaf(f, pv[u].y - th[1], acc[1]);
            acc[2] = __builtin_fmaf(f, pv[u].z - th[2], acc[2]);
            acc[3] = __builtin_fmaf(f, pv[u].w - th[3], acc[3]);
        }
    }
    for (; i < i_hi; ++i) {
        const float4 pv = *reinterpret_cast<const float4 *>(pp + (int64_t)i * stride);
        const float f = fitness_all[i];
        acc[0] = __builtin_fmaf(f, pv.x - th[0], acc[0]);
        acc[1] = __builtin_fmaf(f, pv.y - th[1], acc[1]);
        acc[2] = __builtin_fmaf(f, pv.z - th[2], acc[2]);
        acc[3] = __builtin_fmaf(f, pv.w - th[3], acc[3]);
    }
    *reinterpret_cast<float4 *>(partial + (int64_t)blockIdx.y * stride + s0) = make_float4(acc[0], acc[1], acc[2], acc[3]);
}

// partial of global chunk c lives at partials + (c / chunks_per_block) * block_stride + (c % chunks_per_block) * stride
// (one block per rank after the all-gather; a single block on one GPU)
// which slab positions an ES update leaves alone: padding (>= total) and up to three normalisation-affine ranges
struct SlabSkip {
    int64_t total, a0, b0, a1, b1, a2, b2;
    __host__ __device__ bool operator()(int64_t s) const
    {
        return s >= total || (s >= a0 && s < b0) || (s >= a1 && s < b1) || (s >= a2 && s < b2);
    }
};

__global__ __launch_bounds__(256) void es_apply_kernel(float *theta, const float *partials, int chunks_total,
                                                        int chunks_per_block, int64_t block_stride, int64_t stride,
                                                        SlabSkip skip, int n_total, const float *sigma_dev, float lr)
{
    const int64_t s0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (s0 >= stride) return;
    const float sigma = *sigma_dev;
    const float scale = lr / ((float)n_total * sigma);
    float4 tot = *reinterpret_cast<const float4 *>(partials + s0);
    for (int c = 1; c < chunks_total; ++c) {
        const float4 v = *reinterpret_cast<const float4 *>(partials + (int64_t)(c / chunks_per_block) * block_stride +
                                                           (int64_t)(c % chunks_per_block) * stride + s0);
        tot.x = tot.x + v.x; tot.y = tot.y + v.y; tot.z = tot.z + v.z; tot.w = tot.w + v.w;
    }
    const float4 tv = *reinterpret_cast<const float4 *>(theta + s0);
    const float th[4] = {tv.x, tv.y, tv.z, tv.w}, ac[4] = {tot.x, tot.y, tot.z, tot.w};
    float out[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int64_t s = s0 + c;
        out[c] = skip(s) ? th[c] : th[c] + scale * ac[c];
    }
    *reinterpret_cast<float4 *>(theta + s0) = make_float4(out[0], out[1], out[2], out[3]);
}

__global__ __launch_bounds__(256) void fc_pack_kernel(const float *flat, float *slab, int D, bool to_slab)
{
    const int net = blockIdx.y;
    const int64_t stride = fc_stride(D), P = fc_params(D);
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= stride) return;
    if (to_slab) {
        slab[(int64_t)net * stride + s] = (s < P) ? flat[(int64_t)net * P + fc_slab_to_flat(s, D)] : 0.0f;
    } else if (s < P) {
        const_cast<float *>(flat)[(int64_t)net * P + fc_slab_to_flat(s, D)] = slab[(int64_t)net * stride + s];
    }
}

}  // namespace coevo

using namespace coevo;

static bool fc_dim_ok(int D) { return D == 8 || D == 10; }

extern "C" int64_t coevo_fc_param_count(int D) { return fc_dim_ok(D) ? fc_params(D) : COEVO_ERR_ARG; }
extern "C" int64_t coevo_fc_slab_stride(int D) { return fc_dim_ok(D) ? fc_stride(D) : COEVO_ERR_ARG; }

extern "C" int coevo_fc_pack(const float *flat, float *slab, int n, int D, void *stream)
{
    if (!flat || !slab || n <= 0 || !fc_dim_ok(D)) return COEVO_ERR_ARG;
    const dim3 grid((unsigned)((fc_stride(D) + 255) / 256), (unsigned)n);
    hipLaunchKernelGGL(fc_pack_kernel, grid, dim3(256), 0, (hipStream_t)stream, flat, slab, D, true);
    COEVO_HIP_CHECK(hipGetLastError());
    return COEVO_OK;
}

extern "C" int coevo_fc_unpack(const float *slab, float *flat, int n, int D, void *stream)
{
    if (!flat || !slab || n <= 0 || !fc_dim_ok(D)) return COEVO_ERR_ARG;
    const dim3 grid((unsigned)((fc_stride(D) + 255) / 256), (unsigned)n);
    hipLaunchKernelGGL(fc_pack_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const float *)flat,
                       const_cast<float *>(slab), D, false);
    COEVO_HIP_CHECK(hipGetLastError());
    return COEVO_OK;
}

extern "C" int coevo_fc_perturb_dist(const float *parent_slab, const int32_t *parent_idx, float *child_slab,
                                     int child_first, int n_children, int D, const float *sigma_dev, uint64_t seed,
                                     uint32_t stream_lo_first, uint32_t stream_hi, int skip_layernorm,
                                     const int32_t *gen_dev, const float *dist_ref, double *dist_partial, void *stream);

extern "C" int coevo_fc_perturb(const float *parent_slab, const int32_t *parent_idx, float *child_slab,
                                int child_first, int n_children, int D, const float *sigma_dev, uint64_t seed,
                                uint32_t stream_lo_first, uint32_t stream_hi, int skip_layernorm, void *stream)
{
    return coevo_fc_perturb_dist(parent_slab, parent_idx, child_slab, child_first, n_children, D, sigma_dev, seed,
                                 stream_lo_first, stream_hi, skip_layernorm, nullptr, nullptr, nullptr, stream);
}

extern "C" int coevo_fc_perturb_gen(const float *parent_slab, const int32_t *parent_idx, float *child_slab,
                                    int child_first, int n_children, int D, const float *sigma_dev, uint64_t seed,
                                    uint32_t stream_lo_first, uint32_t stream_hi, int skip_layernorm,
                                    const int32_t *gen_dev, void *stream)
{
    return coevo_fc_perturb_dist(parent_slab, parent_idx, child_slab, child_first, n_children, D, sigma_dev, seed,
                                 stream_lo_first, stream_hi, skip_layernorm, gen_dev, nullptr, nullptr, stream);
}

extern "C" int coevo_fc_perturb_flags(const float *parent_slab, const int32_t *parent_idx, float *child_slab,
                                      int child_first, int n_children, int D, const float *sigma_dev, uint64_t seed,
                                      uint32_t stream_lo_first, uint32_t stream_hi, int flags, void *stream)
{
    if (flags < 0 || flags > 3) return COEVO_ERR_ARG;
    return coevo_fc_perturb_dist(parent_slab, parent_idx, child_slab, child_first, n_children, D, sigma_dev, seed,
                                 stream_lo_first, stream_hi, flags, nullptr, nullptr, nullptr, stream);
}

extern "C" int64_t coevo_fc_perturb_blocks(int D) { return fc_dim_ok(D) ? (fc_stride(D) / 4 + 255) / 256 : COEVO_ERR_ARG; }

extern "C" int coevo_fc_perturb_dist(const float *parent_slab, const int32_t *parent_idx, float *child_slab,
                                     int child_first, int n_children, int D, const float *sigma_dev, uint64_t seed,
                                     uint32_t stream_lo_first, uint32_t stream_hi, int skip_layernorm,
                                     const int32_t *gen_dev, const float *dist_ref, double *dist_partial, void *stream)
{
    if ((dist_ref == nullptr) != (dist_partial == nullptr)) return COEVO_ERR_ARG;
    if (!parent_slab || !parent_idx || !child_slab || !sigma_dev || !fc_dim_ok(D)) return COEVO_ERR_ARG;
    if (n_children < 0 || child_first < 0 || n_children > 65535 || skip_layernorm < 0 || skip_layernorm > 3)
        return COEVO_ERR_ARG;
    if (n_children == 0) return COEVO_OK;
    const dim3 grid((unsigned)((fc_stride(D) / 4 + 255) / 256), (unsigned)n_children);
    hipLaunchKernelGGL(fc_perturb_kernel, grid, dim3(256), 0, (hipStream_t)stream, parent_slab, parent_idx,
                       child_slab, child_first, D, sigma_dev, seed, stream_lo_first, stream_hi, skip_layernorm, gen_dev,
                       dist_ref, dist_partial);
    COEVO_HIP_CHECK(hipGetLastError());
    return COEVO_OK;
}

extern "C" int coevo_fc_perturb_dist_multi(const coevo_fc_perturb_job *jobs, int n_jobs, uint64_t seed, int skip_layernorm,
                                           const int32_t *gen_dev, void *stream)
{
    if (!jobs || n_jobs < 1 || n_jobs > COEVO_MAX_JOBS || skip_layernorm < 0 || skip_layernorm > 3) return COEVO_ERR_ARG;
    PerturbJobs pj{};
    unsigned gx = 0, gy = 0;
    for (int i = 0; i < n_jobs; ++i) {
        const coevo_fc_perturb_job &j = jobs[i];
        if ((j.dist_ref == nullptr) != (j.dist_partial == nullptr)) return COEVO_ERR_ARG;
        if (!j.parent_slab || !j.parent_idx || !j.child_slab || !j.sigma_dev || !fc_dim_ok(j.D)) return COEVO_ERR_ARG;
        if (j.n_children < 0 || j.child_first < 0 || j.n_children > 65535) return COEVO_ERR_ARG;
        pj.j[i] = j;
        const unsigned nb = (unsigned)((fc_stride(j.D) / 4 + 255) / 256);
        gx = nb > gx ? nb : gx;
        gy = (unsigned)j.n_children > gy ? (unsigned)j.n_children : gy;
    }
    if (gy == 0) return COEVO_OK;
    hipLaunchKernelGGL(fc_perturb_multi_kernel, dim3(gx, gy, (unsigned)n_jobs), dim3(256), 0, (hipStream_t)stream, pj, seed,
                       skip_layernorm, gen_dev);
    COEVO_HIP_CHECK(hipGetLastError());
    return COEVO_OK;
}

extern "C" int coevo_fc_rebuild_elites(const float *elite_prev, const int32_t *order, float *elite_new, int E, int D,
                                       const float *sigma_prev_dev, uint64_t seed, uint32_t stream_hi_prev,
                                       const int32_t *gen_dev, void *stream)
{
    if (!elite_prev || !order || !elite_new || !sigma_prev_dev || !fc_dim_ok(D) || E <= 0 || E > 65535)
        return COEVO_ERR_ARG;
    if (elite_prev == elite_new) return COEVO_ERR_ARG;  // elite e reads several previous elites: never in place
    const dim3 grid((unsigned)((fc_stride(D) / 4 + 255) / 256), (unsigned)E);
    hipLaunchKernelGGL(fc_rebuild_elites_kernel, grid, dim3(256), 0, (hipStream_t)stream, elite_prev, order, elite_new,
                       E, D, sigma_prev_dev, seed, stream_hi_prev, gen_dev);
    COEVO_HIP_CHECK(hipGetLastError());
    return COEVO_OK;
}

extern "C" int coevo_fc_gather(const float *src_slab, const int32_t *src_idx, float *dst_slab, int dst_first,
                               int n, int D, void *stream)
{
    if (!src_slab || !src_idx || !dst_slab || !fc_dim_ok(D) || n < 0 || dst_first < 0 || n > 65535)
        return COEVO_ERR_ARG;
    if (n == 0) return COEVO_OK;
    const dim3 grid((unsigned)((fc_stride(D) / 4 + 255) / 256), (unsigned)n);
    hipLaunchKernelGGL(fc_gather_kernel, grid, dim3(256), 0, (hipStream_t)stream, src_slab, src_idx, dst_slab,
                       dst_first, fc_stride(D));
    COEVO_HIP_CHECK(hipGetLastError());
    return COEVO_OK;
}

static int ga_promote_launch(const coevo_ga_promote_role *roles, int n_roles, int E, int hof, const float *sigma,
                             uint64_t seed, uint32_t stream_hi_prev, const int32_t *gen_dev, int32_t *tick, void *stream);

extern "C" int coevo_ga_promote(const coevo_ga_promote_role *roles, int n_roles, int E, int hof, void *stream)
{
    return ga_promote_launch(roles, n_roles, E, hof, nullptr, 0, 0, nullptr, nullptr, stream);
}

// ... with coevo_counter_add(counter, 1) in the same launch (the last launch of a device-resident generation's tail)
extern "C" int coevo_ga_promote_tick(const coevo_ga_promote_role *roles, int n_roles, int E, int hof, int32_t *counter,
                                     void *stream)
{
    if (!counter) return COEVO_ERR_ARG;
    return ga_promote_launch(roles, n_roles, E, hof, nullptr, 0, 0, nullptr, counter, stream);
}

// Promotion with the elites REBUILT in the same launch (a population-sharded run: a rank holds only its own individuals, so
// the new elites - individuals order[k] of the generation just evaluated - are regenerated from last generation's elites and
// the counter-based noise their children were bred with: genetic_algorithm.py:232-252 without a weight crossing xGMI).
// Replaces, per role, coevo_fc_gather (elite -> elite_prev) + coevo_fc_rebuild_elites + the coevo_ga_promote that followed:
// roles with elites_from_pop == 0 rebuild in place from `elite`, roles with elites_from_pop != 0 are promoted as usual.
// sigma: device [n_roles] floats (what the previous generation's children were bred with); noise stream of role r:
// stream_hi_prev + r (+ 4 (g - 1) when gen_dev is given).
extern "C" int coevo_ga_promote_rebuild(const coevo_ga_promote_role *roles, int n_roles, int E, int hof, const float *sigma,
                                        uint64_t seed, uint32_t stream_hi_prev, const int32_t *gen_dev, void *stream)
{
    if (!sigma) return COEVO_ERR_ARG;
    for (int r = 0; roles && r < n_roles && r < 3; ++r)
        if (!roles[r].elites_from_pop && !roles[r].order) return COEVO_ERR_ARG;
    return ga_promote_launch(roles, n_roles, E, hof, sigma, seed, stream_hi_prev, gen_dev, nullptr, stream);
}

static int ga_promote_launch(const coevo_ga_promote_role *roles, int n_roles, int E, int hof, const float *sigma,
                             uint64_t seed, uint32_t stream_hi_prev, const int32_t *gen_dev, int32_t *tick, void *stream)
{
    if (!roles || n_roles < 1 || n_roles > 3 || E < 1 || E > PROMOTE_MAX_E || hof < 1 || hof > PROMOTE_MAX_HOF)
        return COEVO_ERR_ARG;
    GaPromoteArgs a{};
    a.sigma = sigma; a.seed = seed; a.stream_hi_prev = stream_hi_prev; a.gen_dev = gen_dev; a.tick = tick;
    int64_t max_stride = 0;
    for (int r = 0; r < n_roles; ++r) {
        const coevo_ga_promote_role &R = roles[r];
        if (!R.pop || !R.hof || !R.elite || !fc_dim_ok(R.D) || (R.elites_from_pop && !R.order)) return COEVO_ERR_ARG;
        a.role[r] = R;
        if (fc_stride(R.D) > max_stride) max_stride = fc_stride(R.D);
    }
    a.E = E; a.hof = hof;
    const dim3 grid((unsigned)((max_stride / 4 + 255) / 256), (unsigned)n_roles);
    hipLaunchKernelGGL(ga_promote_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
    COEVO_HIP_CHECK(hipGetLastError());
    return COEVO_OK;
}

extern "C" int coevo_es_update(float *theta_slab_net, const float *pert_slab, int D, const float *fitness, int n,
                               const float *sigma_dev, float lr, void *stream)
{
    if (!theta_slab_net || !pert_slab || !fitness || !sigma_dev || !fc_dim_ok(D) || n <= 0) return COEVO_ERR_ARG;
    const dim3 grid((unsigned)((fc_stride(D) / 4 + 255) / 256));
    hipLaunchKernelGGL(es_update_kernel, grid, dim3(256), 0, (hipStream_t)stream, theta_slab_net, pert_slab, D,
                       fitness, n, sigma_dev, lr);
    COEVO_HIP_CHECK(hipGetLastError());
    return COEVO_OK;
}

static int es_partial_launch(const float *theta_net, const float *pert_slab_local, int ind_first, int64_t stride,
                             const float *fitness_all, int n_total, int chunks_total, int chunk_first, int n_chunks,
                             float *partial, void *stream)
{
    if (!theta_net || !pert_slab_local || !fitness_all || !partial) return COEVO_ERR_ARG;
    if (n_total <= 0 || chunks_total <= 0 || chunk_first < 0 || n_chunks <= 0 || chunk_first + n_chunks > chunks_total ||
        n_chunks > 65535 || ind_first < 0)
        return COEVO_ERR_ARG;
    // the caller's nets must start exactly where its first chunk starts
    if ((int64_t)chunk_first * n_total / chunks_total != ind_first) return COEVO_ERR_ARG;
    const dim3 grid((unsigned)((stride / 4 + 255) / 256), (unsigned)n_chunks);
    hipLaunchKernelGGL(es_partial_kernel, grid, dim3(256), 0, (hipStream_t)stream, theta_net, pert_slab_local, ind_first,
                       stride, fitness_all, n_total, chunks_total, chunk_first, partial);
    COEVO_HIP_CHECK(hipGetLastError());
    return COEVO_OK;
}

static int es_apply_launch(float *theta_net, const float *partials, int chunks_total, int chunks_per_block,
                           int64_t block_stride_floats, int64_t stride, SlabSkip skip, int n_total,
                           const float *sigma_dev, float lr, void *stream)
{
    if (!theta_net || !partials || !sigma_dev || n_total <= 0 || chunks_total <= 0 || chunks_per_block <= 0 ||
        block_stride_floats < 0 || (block_stride_floats & 3))
        return COEVO_ERR_ARG;
    const dim3 grid((unsigned)((stride / 4 + 255) / 256));
    hipLaunchKernelGGL(es_apply_kernel, grid, dim3(256), 0, (hipStream_t)stream, theta_net, partials, chunks_total,
                       chunks_per_block, block_stride_floats, stride, skip, n_total, sigma_dev, lr);
    COEVO_HIP_CHECK(hipGetLastError());
    return COEVO_OK;
}

extern "C" int coevo_es_partial(const float *theta_net, const float *pert_slab_local, int ind_first, int D,
                                const float *fitness_all, int n_total, int chunks_total, int chunk_first, int n_chunks,
                                float *partial, void *stream)
{
    if (!fc_dim_ok(D)) return COEVO_ERR_ARG;
    return es_partial_launch(theta_net, pert_slab_local, ind_first, fc_stride(D), fitness_all, n_total, chunks_total,
                             chunk_first, n_chunks, partial, stream);
}

extern "C" int coevo_es_apply(float *theta_net, const float *partials, int chunks_total, int chunks_per_block,
                              int64_t block_stride_floats, int D, int n_total, const float *sigma_dev, float lr,
                              void *stream)
{
    if (!fc_dim_ok(D)) return COEVO_ERR_ARG;
    const int64_t g1 = fc_off_b1(D) + H1, g2 = fc_off_b2(D) + H2;  // LayerNorm affine: never perturbed, never updated
    const SlabSkip skip{fc_params(D), g1, g1 + 2 * H1, g2, g2 + 2 * H2, 0, 0};
    return es_apply_launch(theta_net, partials, chunks_total, chunks_per_block, block_stride_floats, fc_stride(D), skip,
                           n_total, sigma_dev, lr, stream);
}

static bool dqn_ok(int C, int n) { return C >= 1 && C <= 6 && n >= 1 && n <= COEVO_DQN_LOGIT_STRIDE; }

extern "C" int coevo_dqn_es_partial(const float *theta_net, const float *pert_slab_local, int ind_first, int C,
                                    int n_actions, const float *fitness_all, int n_total, int chunks_total,
                                    int chunk_first, int n_chunks, float *partial, void *stream)
{
    if (!dqn_ok(C, n_actions)) return COEVO_ERR_ARG;
    return es_partial_launch(theta_net, pert_slab_local, ind_first, dqn_layout(C, n_actions).stride, fitness_all,
                             n_total, chunks_total, chunk_first, n_chunks, partial, stream);
}

extern "C" int coevo_dqn_es_apply(float *theta_net, const float *partials, int chunks_total, int chunks_per_block,
                                  int64_t block_stride_floats, int C, int n_actions, int n_total,
                                  const float *sigma_dev, float lr, void *stream)
{
    if (!dqn_ok(C, n_actions)) return COEVO_ERR_ARG;
    const DqnLayout L = dqn_layout(C, n_actions);  // BatchNorm affine is not perturbable (Atari/deepqn.py:158-171)
    const SlabSkip skip{L.total, L.b1 + 32, L.w2, L.b2 + 64, L.w3, L.b3 + 64, L.wf};
    return es_apply_launch(theta_net, partials, chunks_total, chunks_per_block, block_stride_floats, L.stride, skip,
                           n_total, sigma_dev, lr, stream);
}

/* dst[dst_first + i] = src[src_idx[i]] for nets of any layout (stride floats apart, a multiple of 4) */
extern "C" int coevo_net_gather(const float *src_slab, const int32_t *src_idx, float *dst_slab, int dst_first, int n,
                                int64_t stride_floats, void *stream)
{
    if (!src_slab || !src_idx || !dst_slab || n < 0 || dst_first < 0 || n > 65535 || stride_floats <= 0 ||
        (stride_floats & 3))
        return COEVO_ERR_ARG;
    if (n == 0) return COEVO_OK;
    const dim3 grid((unsigned)((stride_floats / 4 + 255) / 256), (unsigned)n);
    hipLaunchKernelGGL(fc_gather_kernel, grid, dim3(256), 0, (hipStream_t)stream, src_slab, src_idx, dst_slab,
                       dst_first, stride_floats);
    COEVO_HIP_CHECK(hipGetLastError());
    return COEVO_OK;
}


// ---- the noise contract, exposed: the round count and the generator itself (raw words and the Gaussians built from them),
// so that a binding / a resumed run can check what it is loading and the tests can hold the generator against published
// known-answer vectors (Random123's kat_vectors) and moment / correlation bounds, independently of the oracle.
namespace coevo {
__global__ void philox_raw_kernel(int rounds, const uint32_t *ctr_key, int n, uint32_t *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t *c = ctr_key + 6 * (size_t)i;
    u32x4 o;
    if (rounds == 7) o = philox4x32<7>(c[0], c[1], c[2], c[3], c[4], c[5]);
    else o = philox4x32<10>(c[0], c[1], c[2], c[3], c[4], c[5]);
#pragma unroll
    for (int k = 0; k < 4; ++k) out[4 * (size_t)i + k] = o.v[k];
}

__global__ void philox_normal_kernel(uint64_t seed, uint32_t stream_lo, uint32_t stream_hi, uint32_t q_first, int n_quads,
                                     float *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_quads) return;
    float z[4];
    philox_normal4(seed, stream_lo, stream_hi, q_first + (uint32_t)i, z);
#pragma unroll
    for (int k = 0; k < 4; ++k) out[4 * (size_t)i + k] = z[k];
}
}  // namespace coevo

extern "C" int coevo_noise_rounds(void) { return coevo::COEVO_NOISE_ROUNDS; }

extern "C" int coevo_philox4x32(int rounds, const uint32_t *ctr_key, int n, uint32_t *out, void *stream)
{
    if (!ctr_key || !out || n <= 0 || (rounds != 7 && rounds != 10)) return COEVO_ERR_ARG;
    hipLaunchKernelGGL(coevo::philox_raw_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, rounds, ctr_key, n,
                       out);
    COEVO_HIP_CHECK(hipGetLastError());
    return COEVO_OK;
}

extern "C" int coevo_philox_normals(uint64_t seed, uint32_t stream_lo, uint32_t stream_hi, uint32_t q_first, int n_quads,
                                    float *out, void *stream)
{
    if (!out || n_quads <= 0) return COEVO_ERR_ARG;
    hipLaunchKernelGGL(coevo::philox_normal_kernel, dim3((n_quads + 255) / 256), dim3(256), 0, (hipStream_t)stream, seed,
                       stream_lo, stream_hi, q_first, n_quads, out);
    COEVO_HIP_CHECK(hipGetLastError());
    return COEVO_OK;
}

COEVO_DEFINE_TU_FLAGS(offspring)
