// Probe: sustained issue rate of the f32 MFMAs on gfx950, whole chip, as a function of waves per SIMD, independent
// accumulators per wave and vector instructions interleaved per MFMA.  Prints SIMD cycles per MFMA at the clock the
// launch held (s_memrealtime 100 MHz against s_memtime shader cycles of one wave).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_rate_probe.hip -o /tmp/mfma_rate_probe && /tmp/mfma_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// KIND 0: 16x16x4, 1: 4x4x1, 2: 32x32x2.  NACC independent accumulators, NV dependent-free v_fma per MFMA
template <int KIND, int NACC, int NV>
__global__ void rate(float *out, unsigned long long *clk, int iters, float a, float b)
{
    f32x4 acc[NACC];
    f32x16 acc32[KIND == 2 ? NACC : 1];
    float v[NV > 0 ? NV : 1];
    for (int i = 0; i < NACC; ++i) acc[i] = {a, b, a, b};
    if constexpr (KIND == 2)
        for (int i = 0; i < NACC; ++i)
            for (int j = 0; j < 16; ++j) acc32[i][j] = a;
    for (int i = 0; i < (NV > 0 ? NV : 1); ++i) v[i] = a + i;
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            if constexpr (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
            if constexpr (KIND == 1) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
            if constexpr (KIND == 2) acc32[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc32[i], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                v[k] = __builtin_fmaf(v[k], b, a);
                asm volatile("" : "+v"(v[k]));
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
    if constexpr (KIND == 2)
        for (int i = 0; i < NACC; ++i) s += acc32[i][5];
    for (int i = 0; i < (NV > 0 ? NV : 1); ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

// MODE 0: NACC MFMAs, then NACC * NV independent v_fma in one batch; MODE 1: one ds_read_b32 per MFMA (value unused by the
// MFMA: issue cost only); MODE 2: one ds_read_u8 + cvt + 2 VALU (conv1's gather) per 2 MFMAs, the value feeding them
template <int MODE, int NACC, int NV>
__global__ void mixed(float *out, unsigned long long *clk, int iters, float a, float b)
{
    __shared__ float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = a * i;
    __syncthreads();
    f32x4 acc[NACC];
    constexpr int NVT = NACC * (NV > 0 ? NV : 1);
    float v[NVT];
    for (int i = 0; i < NACC; ++i) acc[i] = {a, b, a, b};
    for (int i = 0; i < NVT; ++i) v[i] = a + i;
    int off = (threadIdx.x * 17) & 4095;
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < NACC * NV; ++k) {
                v[k] = __builtin_fmaf(v[k], b, a);
                asm volatile("" : "+v"(v[k]));
            }
            __builtin_amdgcn_sched_barrier(0);
        } else if constexpr (MODE == 1) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
                v[i] = lds[(off + 64 * i + it) & 4095];
            }
#pragma unroll
            for (int i = 0; i < NACC; ++i) asm volatile("" : "+v"(v[i]));
        } else {
            float g[NACC / 2];
#pragma unroll
            for (int i = 0; i < NACC / 2; ++i) {
                const unsigned char u = reinterpret_cast<const unsigned char *>(lds)[(off + 64 * i + it) & 16383];
                const float x = (float)u;
                g[i] = __builtin_fmaf(x, a, x * b);
            }
#pragma unroll
            for (int i = 0; i < NACC / 2; ++i) {
                acc[2 * i] = __builtin_amdgcn_mfma_f32_16x16x4f32(g[i], b, acc[2 * i], 0, 0, 0);
                acc[2 * i + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(g[i], a, acc[2 * i + 1], 0, 0, 0);
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
    for (int i = 0; i < NVT; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int MODE, int NACC, int NV>
void run_mixed(int waves_per_simd, const char *name)
{
    const int iters = 20000 / NACC, threads = 256, blocks = 256 * waves_per_simd;
    float *out; unsigned long long *clk;
    hipMalloc(&out, (size_t)blocks * threads * 4); hipMalloc(&clk, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((mixed<MODE, NACC, NV>), dim3(blocks), dim3(threads), 0, 0, out, clk, iters, 1.0f, 0.5f);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c[2]; hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost);
    const double n = (double)iters * NACC * waves_per_simd;
    const double ghz = (double)c[0] / ((double)c[1] * 10.0);
    printf("%-28s acc %d  nv %d  waves/SIMD %d: %.3f ms  %.1f cycles per MFMA per SIMD (%.2f GHz, %.1f per own MFMA)\n",
           name, NACC, NV, waves_per_simd, ms, ms * 1e6 * ghz / n, ghz, (double)c[0] / (iters * NACC));
    hipFree(out); hipFree(clk);
}

template <int KIND, int NACC, int NV>
void run(int waves_per_simd, const char *name)
{
    const int iters = 20000 / NACC, threads = 256, blocks = 256 * waves_per_simd;   // 4 waves per workgroup: one per SIMD
    float *out; unsigned long long *clk;
    hipMalloc(&out, (size_t)blocks * threads * 4); hipMalloc(&clk, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((rate<KIND, NACC, NV>), dim3(blocks), dim3(threads), 0, 0, out, clk, iters, 1.0f, 0.5f);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c[2]; hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost);
    const double n = (double)iters * NACC * waves_per_simd;   // MFMAs per SIMD
    const double ghz = (double)c[0] / ((double)c[1] * 10.0);  // shader cycles per ns (100 MHz realtime counter)
    printf("%-9s acc %d  valu/mfma %d  waves/SIMD %d: %.3f ms  %.1f cycles per MFMA per SIMD (wave 0: %.2f GHz, %.1f cycles per own MFMA)\n",
           name, NACC, NV, waves_per_simd, ms, ms * 1e6 * ghz / n, ghz, (double)c[0] / (iters * NACC));
    hipFree(out); hipFree(clk);
}

int main(int argc, char **argv)
{
    if (argc > 1 && argv[1][0] == 'c') {   // "chain": what ONE dependent accumulator chain costs per step (round 5: the small-shard
        for (int w : {1, 2, 4}) {          // cycle kernel is one or two chains per wave)
            run<1, 1, 0>(w, "4x4x1");
            run<1, 2, 0>(w, "4x4x1");
            run<1, 3, 0>(w, "4x4x1");
            run<0, 1, 0>(w, "16x16x4");
            run<0, 2, 0>(w, "16x16x4");
        }
        return 0;
    }
    for (int w : {1, 2, 6}) {
        run<0, 1, 0>(w, "16x16x4");
        run<0, 2, 0>(w, "16x16x4");
        run<0, 4, 0>(w, "16x16x4");
        run<0, 8, 0>(w, "16x16x4");
    }
    for (int w : {1, 2, 6}) {
        run<0, 4, 1>(w, "16x16x4");
        run<0, 4, 2>(w, "16x16x4");
        run<0, 4, 4>(w, "16x16x4");
    }
    for (int w : {1, 2, 6}) {
        run_mixed<0, 4, 1>(w, "4 mfma then 4 valu");
        run_mixed<0, 4, 2>(w, "4 mfma then 8 valu");
        run_mixed<0, 8, 2>(w, "8 mfma then 16 valu");
        run_mixed<1, 4, 0>(w, "ds_read_b32 per mfma");
        run_mixed<1, 8, 0>(w, "ds_read_b32 per mfma");
        run_mixed<2, 4, 0>(w, "u8 gather+cvt per 2 mfma");
        run_mixed<2, 8, 0>(w, "u8 gather+cvt per 2 mfma");
    }
    for (int w : {1, 6}) {
        run<1, 4, 0>(w, "4x4x1");
        run<1, 8, 0>(w, "4x4x1");
        run<2, 2, 0>(w, "32x32x2");
        run<2, 4, 0>(w, "32x32x2");
    }
    return 0;
}
