// Probe: v_mfma_f32_4x4x1_16B_f32 (16 blocks of 4x4x1) as a sequential-k fma chain for <= 4 rows x 64 columns:
// lane l supplies B = W[column l][k] (its own streamed value) and A = x[row l%4][k]; accumulator register i = row i of
// column l.  Is it bit-identical to acc=bias; acc=fmaf(w_k,a_k,acc), k ascending?
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// A [8 rows][K], W [64 cols][K], bias[64]; D[8][64]
__global__ void probe(const float *A, const float *W, const float *bias, float *D, int K)
{
    const int l = threadIdx.x;
    f32x4 acc0, acc1;
    for (int i = 0; i < 4; ++i) { acc0[i] = bias[l]; acc1[i] = bias[l]; }
    for (int kq = 0; kq < K / 4; ++kq) {
        const float4 wv = *reinterpret_cast<const float4 *>(W + (size_t)l * K + 4 * kq);
        const float4 x0 = *reinterpret_cast<const float4 *>(A + (size_t)(l & 3) * K + 4 * kq);
        const float4 x1 = *reinterpret_cast<const float4 *>(A + (size_t)(4 + (l & 3)) * K + 4 * kq);
        acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x0.x, wv.x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(x1.x, wv.x, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x0.y, wv.y, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(x1.y, wv.y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x0.z, wv.z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(x1.z, wv.z, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x0.w, wv.w, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(x1.w, wv.w, acc1, 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) { D[i * 64 + l] = acc0[i]; D[(4 + i) * 64 + l] = acc1[i]; }
}

int main()
{
    const int K = 512;
    std::vector<float> A(8 * K), W(64 * K), b(64), D(8 * 64), ref(8 * 64);
    srand(1);
    auto rnd = []() { return (float)rand() / RAND_MAX * 2.f - 1.f; };
    for (auto &v : A) v = rnd() > 0 ? rnd() : 0.f;
    for (auto &v : W) v = rnd() * 0.05f;
    for (auto &v : b) v = rnd() * 0.05f;
    A[5] = 1e-41f; W[7] = 3e-42f; A[6 * K + 100] = 3e38f; W[3 * K + 100] = 3e38f;
    for (int i = 0; i < 8; ++i)
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
    printf("mfma 4x4x1 chain vs fmaf chain: %d mismatches of %zu (%d non-finite references)\n", bad, D.size(), nonfinite);
    return bad != 0;
}
