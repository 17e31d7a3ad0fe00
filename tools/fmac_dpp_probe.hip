// Probe (round 5): issue rate of v_fmac_f32 with a DPP row_newbcast source - the per-row activation broadcast of the small-shard
// fc2 loop (one VGPR holds 16 consecutive activations of a row, lane % 16 = k; `row_newbcast:j` hands lane j's value to every
// lane of its 16-lane row inside the fmac itself) - against the plain v_fmac_f32, R independent accumulators per lane.
//   hipcc --offload-arch=gfx950 -O3 tools/fmac_dpp_probe.hip -o variants/fmac_dpp_probe && variants/fmac_dpp_probe
#include <hip/hip_runtime.h>
#include <cstdio>

template <int J>
__device__ __forceinline__ void fmac_bcast(float &acc, float xv, float w)
{
    asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(xv), "v"(w), "n"(J));
}
__device__ __forceinline__ void fmac_plain(float &acc, float xv, float w)
{
    asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(acc) : "v"(xv), "v"(w));
}

template <int R, bool DPP>
__global__ void rate(float *out, unsigned long long *clk, int iters, float a, float b)
{
    float acc[R], xv[R], w[4] = {a, b, a + 1, b + 1};
    for (int r = 0; r < R; ++r) { acc[r] = a; xv[r] = a + r + (threadIdx.x & 15); }
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if constexpr (DPP) {
                    if (e == 0) fmac_bcast<0>(acc[r], xv[r], w[e]);
                    if (e == 1) fmac_bcast<5>(acc[r], xv[r], w[e]);
                    if (e == 2) fmac_bcast<10>(acc[r], xv[r], w[e]);
                    if (e == 3) fmac_bcast<15>(acc[r], xv[r], w[e]);
                } else {
                    fmac_plain(acc[r], xv[r], w[e]);
                }
            }
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int r = 0; r < R; ++r) s += acc[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int R, bool DPP>
void run(int waves_per_simd)
{
    const int iters = 4000, threads = 256, blocks = 256 * waves_per_simd;
    float *out; unsigned long long *clk;
    if (hipMalloc(&out, (size_t)blocks * threads * 4) != hipSuccess || hipMalloc(&clk, 16) != hipSuccess) return;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((rate<R, DPP>), dim3(blocks), dim3(threads), 0, 0, out, clk, iters, 1.0f, 0.5f);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    }
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c[2]; (void)hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost);
    const double n = (double)iters * 4 * R;
    printf("%-6s R %d  waves/SIMD %d: %.3f ms, %.2f cycles per own fmac (wave 0), %.2f cycles per fmac per SIMD\n", DPP ? "dpp" : "plain", R,
           waves_per_simd, ms, (double)c[0] / n, ms * 1e6 * ((double)c[0] / ((double)c[1] * 10.0)) / (n * waves_per_simd));
    (void)hipFree(out); (void)hipFree(clk);
}

int main()
{
    for (int w : {1, 2, 4}) {
        run<5, false>(w); run<5, true>(w);
        run<8, false>(w); run<8, true>(w);
        run<2, true>(w);
    }
    return 0;
}
