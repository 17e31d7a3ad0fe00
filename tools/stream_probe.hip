// diagnostic: how fast can ONE workgroup (4 waves) stream its own contiguous block from HBM, as a function of the
// number of workgroups per CU, loads in flight per lane (U) and waves per workgroup?   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef float f32x4_nt __attribute__((ext_vector_type(4)));
template <int U, int THREADS, bool NT = false>
__global__ __launch_bounds__(THREADS) void stream_kernel(const float4 *base, size_t f4_per_wg, float *out)
{
    const float4 *p = base + (size_t)blockIdx.x * f4_per_wg + threadIdx.x;
    float acc = 0.f;
    const int iters = (int)(f4_per_wg / THREADS / U);
    for (int it = 0; it < iters; ++it) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if constexpr (NT) {
                const f32x4_nt t = __builtin_nontemporal_load(reinterpret_cast<const f32x4_nt *>(p + (size_t)(it * U + u) * THREADS));
                v[u] = make_float4(t[0], t[1], t[2], t[3]);
            } else {
                v[u] = p[(size_t)(it * U + u) * THREADS];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    if (acc == 12345.678f) out[blockIdx.x] = acc;
}

template <int U, int THREADS, bool NT = false>
int run(const float4 *buf, float *out, int n_wg, size_t bytes_per_wg, const char *tag)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const size_t f4 = bytes_per_wg / 16;
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((stream_kernel<U, THREADS, NT>), dim3(n_wg), dim3(THREADS), 0, 0, buf, f4, out);
    CK(hipEventRecord(a));
    const int reps = 5;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((stream_kernel<U, THREADS, NT>), dim3(n_wg), dim3(THREADS), 0, 0, buf, f4, out);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= reps;
    const double tot = (double)n_wg * bytes_per_wg;
    printf("%-10s U=%2d threads=%3d wgs=%5d  %8.1f us  total %7.0f GB/s  per-WG %6.1f GB/s\n", tag, U, THREADS, n_wg, ms * 1e3,
           tot / ms / 1e6, (double)bytes_per_wg / ms / 1e6);
    return 0;
}

int main()
{
    const size_t bytes_per_wg = 512 * 1024;
    const int max_wg = 2048;
    float4 *buf; float *out;
    CK(hipMalloc(&buf, (size_t)max_wg * bytes_per_wg));
    CK(hipMemset(buf, 0, (size_t)max_wg * bytes_per_wg));
    CK(hipMalloc(&out, max_wg * 4));
    for (int n : {1, 64, 128, 256, 512, 768, 1024, 2048}) if (run<16, 256>(buf, out, n, bytes_per_wg, "u16")) return 1;
    for (int n : {1, 256, 512, 1024}) if (run<8, 256>(buf, out, n, bytes_per_wg, "u8")) return 1;
    for (int n : {1, 256, 512}) if (run<32, 256>(buf, out, n, bytes_per_wg, "u32")) return 1;
    for (int n : {1, 256, 512}) if (run<16, 512>(buf, out, n, bytes_per_wg, "t512")) return 1;
    for (int n : {1, 256, 512}) if (run<16, 1024>(buf, out, n, bytes_per_wg, "t1024")) return 1;
    for (int n : {256, 512, 1024, 2048}) if (run<16, 256, true>(buf, out, n, bytes_per_wg, "u16 nt")) return 1;
    for (int n : {512, 1024}) if (run<8, 256, true>(buf, out, n, bytes_per_wg, "u8 nt")) return 1;
    return 0;
}
