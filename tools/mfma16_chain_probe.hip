// Probe: is v_mfma_f32_16x16x4_f32 with C-in = bias bit-identical to acc=bias; acc=fmaf(w_k,a_k,acc), k ascending?
// B operands of the four 16-column tiles of a wave are built from each lane's own 16-byte row piece with a 4x4
// (register x 16-lane row) transpose: two v_permlane32_swap + two v_permlane16_swap.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// A [16 rows][K] row-major, W [64 cols][K] row-major (four 16-col tiles), bias[64]; D[16][64]
__global__ void probe(const float *A, const float *W, const float *bias, float *D, int K)
{
    const int l = threadIdx.x, c = l & 15, g = l >> 4;
    f32x4 acc[4];
    for (int T = 0; T < 4; ++T)
        for (int i = 0; i < 4; ++i) acc[T][i] = bias[16 * T + c];
    for (int kq = 0; kq < K / 4; ++kq) {
        const float4 wv = *reinterpret_cast<const float4 *>(W + (size_t)l * K + 4 * kq);
        u32x2 s02 = __builtin_amdgcn_permlane32_swap(__float_as_uint(wv.x), __float_as_uint(wv.z), false, false);
        u32x2 s13 = __builtin_amdgcn_permlane32_swap(__float_as_uint(wv.y), __float_as_uint(wv.w), false, false);
        u32x2 y01 = __builtin_amdgcn_permlane16_swap(s02[0], s13[0], false, false);
        u32x2 y23 = __builtin_amdgcn_permlane16_swap(s02[1], s13[1], false, false);
        const float b[4] = {__uint_as_float(y01[0]), __uint_as_float(y01[1]), __uint_as_float(y23[0]), __uint_as_float(y23[1])};
        const float a = A[(size_t)c * K + 4 * kq + g];
        for (int T = 0; T < 4; ++T) acc[T] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[T], acc[T], 0, 0, 0);
    }
    for (int T = 0; T < 4; ++T)
        for (int r = 0; r < 4; ++r) D[(4 * g + r) * 64 + 16 * T + c] = acc[T][r];
}

int main()
{
    const int K = 512;
    std::vector<float> A(16 * K), W(64 * K), b(64), D(16 * 64), ref(16 * 64);
    srand(1);
    auto rnd = []() { return (float)rand() / RAND_MAX * 2.f - 1.f; };
    for (auto &v : A) v = rnd() > 0 ? rnd() : 0.f;
    for (auto &v : W) v = rnd() * 0.05f;
    for (auto &v : b) v = rnd() * 0.05f;
    A[5] = 1e-41f; W[7] = 3e-42f; A[9 * K + 100] = 3e38f; W[3 * K + 100] = 3e38f;  // denormals, an overflow to inf
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 64; ++j) {
            float acc = b[j];
            for (int k = 0; k < K; ++k) acc = fmaf(W[(size_t)j * K + k], A[(size_t)i * K + k], acc);
            ref[i * 64 + j] = acc;
        }
    float *dA, *dW, *db, *dD;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dW, W.size() * 4); hipMalloc(&db, 256); hipMalloc(&dD, D.size() * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(db, b.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dW, db, dD, K);
    hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0, nonfinite = 0;
    for (size_t i = 0; i < D.size(); ++i) {
        if (!std::isfinite(ref[i])) ++nonfinite;
        if (memcmp(&D[i], &ref[i], 4)) { if (bad < 8) printf("mismatch %zu: %.9g vs %.9g\n", i, D[i], ref[i]); ++bad; }
    }
    printf("mfma 16x16x4 chain vs fmaf chain: %d mismatches of %zu (%d non-finite references)\n", bad, D.size(), nonfinite);
    return bad != 0;
}
