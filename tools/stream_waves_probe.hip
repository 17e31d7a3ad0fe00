// Probe (round 5): how fast ONE workgroup per CU streams a 512 KiB net (the fc2 matrix of a per-individual FCNetwork) as a
// function of its wave count and of the 16-byte loads each lane keeps in flight - the stream leg of the small-launch cycle
// kernel (a rank of a sharded population: 225 workgroups on 256 CUs, the nets re-read every env-cycle out of the Infinity Cache).
//   hipcc --offload-arch=gfx950 -O3 tools/stream_waves_probe.hip -o variants/stream_waves_probe && variants/stream_waves_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int WAVES, int DEPTH, bool NT>
__global__ __launch_bounds__(WAVES * 64) void stream(const float4 *base, float *out, int pieces_per_wg)
{
    // wave w streams pieces [w * per, (w + 1) * per) of the workgroup's region, DEPTH wave-loads (1 KiB each) in flight
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int per = pieces_per_wg / WAVES;
    const float4 *p = base + (size_t)blockIdx.x * pieces_per_wg * 64 + (size_t)w * per * 64 + l;
    float4 buf[DEPTH];
    float acc = 0.f;
    typedef float f4 __attribute__((ext_vector_type(4)));
    auto ld = [&](int i) {
        if constexpr (NT) {
            const f4 v = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(p + (size_t)i * 64));
            return make_float4(v[0], v[1], v[2], v[3]);
        } else {
            return p[(size_t)i * 64];
        }
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) buf[d] = ld(d);
    for (int i = 0; i < per; i += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const float4 v = buf[d];
            if (i + DEPTH + d < per) buf[d] = ld(i + DEPTH + d);
            acc += v.x + v.y + v.z + v.w;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int WAVES, int DEPTH, bool NT>
void run(const float4 *buf, float *out, int wgs)
{
    const int pieces = 512 * 1024 / 1024;   // 512 one-KiB wave-loads per workgroup
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9f, sum = 0;
    for (int rep = 0; rep < 12; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((stream<WAVES, DEPTH, NT>), dim3(wgs), dim3(WAVES * 64), 0, 0, buf, out, pieces);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 2) { best = ms < best ? ms : best; sum += ms; }
    }
    printf("%2d waves x %2d loads in flight, %s, %3d workgroups: %.1f us (best %.1f) = %.1f GB/s per CU, %.2f TB/s\n", WAVES, DEPTH,
           NT ? "nt   " : "plain", wgs, 1e3 * sum / 10, 1e3 * best, 512.0 * 1024 / (1e3 * sum / 10) / 1e3,
           wgs * 512.0 * 1024 / (1e3 * sum / 10) / 1e6);
}

int main()
{
    const int wgs = 225;
    float4 *buf; float *out;
    if (hipMalloc(&buf, (size_t)wgs * 512 * 1024) != hipSuccess || hipMalloc(&out, (size_t)wgs * 1024 * 4) != hipSuccess) return 1;
    (void)hipMemset(buf, 0, (size_t)wgs * 512 * 1024);
    run<4, 8, false>(buf, out, wgs);  run<4, 16, false>(buf, out, wgs); run<4, 32, false>(buf, out, wgs);
    run<8, 8, false>(buf, out, wgs);  run<8, 16, false>(buf, out, wgs);
    run<16, 4, false>(buf, out, wgs); run<16, 8, false>(buf, out, wgs);
    run<4, 16, true>(buf, out, wgs);  run<8, 8, true>(buf, out, wgs);   run<8, 16, true>(buf, out, wgs); run<16, 8, true>(buf, out, wgs);
    run<4, 16, false>(buf, out, 75);  run<8, 8, false>(buf, out, 75);   run<16, 8, false>(buf, out, 75);
    return 0;
}
