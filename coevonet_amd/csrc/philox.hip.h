// Counter-based Gaussian noise shared by the offspring kernels: Philox4x32-7 + Box-Muller with fmaf-only log / sincos
// polynomials (bit-identical with oracle/coevo_oracle.c).  Seven rounds for the offspring noise - the fewest that pass
// BigCrush (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11, table 2: Philox4x32 is Crush-resistant
// from 7 rounds on; 10 is the library default's safety margin).  The 64-bit multiplies are a third of the breeding
// kernels' issue cycles, and `device_philox` is the build's own noise definition (the reference's torch / numpy streams are
// the `host_reference` mode), so the three rounds saved come at no loss of contract.  The synthetic env keeps 10 rounds
// (its frames are keyed test data, not a noise source whose cost matters).
#pragma once
#include "coevo_common.hip.h"

namespace coevo {

struct u32x4 { uint32_t v[4]; };

constexpr int COEVO_NOISE_ROUNDS = 7;

template <int ROUNDS>
__device__ inline u32x4 philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return u32x4{{c0, c1, c2, c3}};
}

__device__ inline u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1)
{
    return philox4x32<10>(c0, c1, c2, c3, k0, k1);
}

// ln(x), x a normal float in (0,1): exponent by bit ops, cephes logf polynomial evaluated with fmaf only
__device__ inline float canon_logf(float x)
{
    const uint32_t b = __float_as_uint(x);
    int e = (int)((b >> 23) & 0xff) - 126;
    float m = __uint_as_float((b & 0x007fffffu) | 0x3f000000u);
    if (m < 0.707106781186547524f) { e -= 1; m = (m + m) - 1.0f; } else { m = m - 1.0f; }
    const float z = m * m;
    float y = 7.0376836292E-2f;
    y = __builtin_fmaf(y, m, -1.1514610310E-1f);
    y = __builtin_fmaf(y, m, 1.1676998740E-1f);
    y = __builtin_fmaf(y, m, -1.2420140846E-1f);
    y = __builtin_fmaf(y, m, 1.4249322787E-1f);
    y = __builtin_fmaf(y, m, -1.6668057665E-1f);
    y = __builtin_fmaf(y, m, 2.0000714765E-1f);
    y = __builtin_fmaf(y, m, -2.4999993993E-1f);
    y = __builtin_fmaf(y, m, 3.3333331174E-1f);
    y = (y * m) * z;
    const float fe = (float)e;
    y = __builtin_fmaf(-2.12194440e-4f, fe, y);
    y = __builtin_fmaf(-0.5f, z, y);
    float r = m + y;
    r = __builtin_fmaf(0.693359375f, fe, r);
    return r;
}

// (cos, sin)(2*pi*u), u = k/2^24: exact quadrant split, cephes polynomials on [0, pi/4], fmaf only
__device__ inline void canon_sincos2pi(float u, float &c_out, float &s_out)
{
    const float t = u * 4.0f;
    const float qf = __builtin_floorf(t);
    const int q = (int)qf;
    float f = t - qf;
    const bool swap = f > 0.5f;
    if (swap) f = 1.0f - f;
    const float x = f * 1.57079632679489661923f;
    const float z = x * x;
    float sp = -1.9515295891E-4f;
    sp = __builtin_fmaf(sp, z, 8.3321608736E-3f);
    sp = __builtin_fmaf(sp, z, -1.6666654611E-1f);
    float s = __builtin_fmaf(sp * z, x, x);
    float cp = 2.443315711809948E-005f;
    cp = __builtin_fmaf(cp, z, -1.388731625493765E-003f);
    cp = __builtin_fmaf(cp, z, 4.166664568298827E-002f);
    float c = __builtin_fmaf(cp * z, z, __builtin_fmaf(-0.5f, z, 1.0f));
    if (swap) { const float tmp = s; s = c; c = tmp; }
    switch (q & 3) {
    case 0: c_out = c; s_out = s; break;
    case 1: c_out = -s; s_out = c; break;
    case 2: c_out = -c; s_out = -s; break;
    default: c_out = s; s_out = -c; break;
    }
}

__device__ inline void box_muller(uint32_t a, uint32_t b, float &z0, float &z1)
{
    const float u1 = (float)(2u * (a >> 9) + 1u) * 5.9604644775390625e-08f;
    const float u2 = (float)(b >> 8) * 5.9604644775390625e-08f;
    const float r = __builtin_sqrtf(-2.0f * canon_logf(u1));
    float c, s;
    canon_sincos2pi(u2, c, s);
    z0 = r * c;
    z1 = r * s;
}

__device__ inline void philox_normal4(uint64_t seed, uint32_t stream_lo, uint32_t stream_hi, uint32_t q, float z[4])
{
    const u32x4 o = philox4x32<COEVO_NOISE_ROUNDS>(q, stream_lo, stream_hi, 0x636f6576u, (uint32_t)seed, (uint32_t)(seed >> 32));
    box_muller(o.v[0], o.v[1], z[0], z[1]);
    box_muller(o.v[2], o.v[3], z[2], z[3]);
}

}  // namespace coevo
