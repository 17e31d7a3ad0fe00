// Probe: is v_mfma_f32_32x32x2_f32 with C-in = bias bit-identical to acc=bias; acc=fmaf(w_k,a_k,acc) k ascending?
// Also exercises __builtin_amdgcn_permlane32_swap to build B operands from 16-byte row pieces.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// A [32 rows][K] row-major, W [64 cols][K] row-major (two 32-col tiles), bias[64]; D[32][64]
__global__ void probe(const float *A, const float *W, const float *bias, float *D, int K)
{
    const int l = threadIdx.x;
    f32x16 acc0, acc1;
    for (int i = 0; i < 16; ++i) { acc0[i] = bias[l & 31]; acc1[i] = bias[32 + (l & 31)]; }
    for (int kq = 0; kq < K / 4; ++kq) {
        const float4 wv = *reinterpret_cast<const float4 *>(W + (size_t)l * K + 4 * kq);
        unsigned x = __float_as_uint(wv.x), y = __float_as_uint(wv.y), z = __float_as_uint(wv.z), w = __float_as_uint(wv.w);
        u32x2 s0 = __builtin_amdgcn_permlane32_swap(x, y, false, false);
        u32x2 s1 = __builtin_amdgcn_permlane32_swap(z, w, false, false);
        const float bA0 = __uint_as_float(s0[0]), bB0 = __uint_as_float(s0[1]);
        const float bA1 = __uint_as_float(s1[0]), bB1 = __uint_as_float(s1[1]);
        const float a0 = A[(size_t)(l & 31) * K + 4 * kq + (l >> 5)];
        const float a1 = A[(size_t)(l & 31) * K + 4 * kq + 2 + (l >> 5)];
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bA0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bB0, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bA1, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bB1, acc1, 0, 0, 0);
    }
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        D[row * 64 + (l & 31)] = acc0[r];
        D[row * 64 + 32 + (l & 31)] = acc1[r];
    }
}

int main()
{
    const int K = 512;
    std::vector<float> A(32 * K), W(64 * K), b(64), D(32 * 64), ref(32 * 64);
    srand(1);
    auto rnd = []() { return (float)rand() / RAND_MAX * 2.f - 1.f; };
    for (auto &v : A) v = rnd() > 0 ? rnd() : 0.f;
    for (auto &v : W) v = rnd() * 0.05f;
    for (auto &v : b) v = rnd() * 0.05f;
    A[5] = 1e-41f; W[7] = 3e-42f;
    for (int i = 0; i < 32; ++i)
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
    int bad = 0;
    for (size_t i = 0; i < D.size(); ++i)
        if (memcmp(&D[i], &ref[i], 4)) { if (bad < 5) printf("mismatch %zu: %.9g vs %.9g\n", i, D[i], ref[i]); ++bad; }
    printf("mfma chain vs fmaf chain: %d mismatches of %zu\n", bad, D.size());
    return bad != 0;
}
