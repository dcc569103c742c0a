// cosf / sinf / atan2f as the reference's host libm evaluates them, for host and device alike.
//
// The reference builds its warp matrix with std::cos(float) / std::sin(float) and decomposes the measured transform with
// std::atan2(float, float) (/root/reference/src/Stabilizer.cpp:662, 902-908, 1689): glibc's cosf / sinf / atan2f.  The
// device's libm rounds some arguments the other way in the last place, and one ulp in a matrix entry moves the 1/1024-px
// coordinate of a whole pixel column across a rounding boundary - the only thing that stood between this library's frames
// and the oracle's, bit for bit.  These are restatements of glibc's algorithms as shipped from 2.28 on and verified against
// 2.35 on x86-64 with FMA (2.41 replaces atanf / atan2f with correctly rounded versions; x86 hosts without FMA run an unfused
// sincosf - tests/test_libm.py skips its host comparison there and says why):
//   cosf / sinf   sysdeps/ieee754/flt-32/s_cosf.c, s_sinf.c, sincosf.h (the ARM optimized-routines kernels: the argument in
//                 double, reduction by pi/2, one of two degree-8 / degree-7 polynomials); multiply-adds fused, which is what the
//                 x86-64 (FMA ifunc variants) and aarch64 builds execute;
//   atan2f        sysdeps/ieee754/flt-32/e_atan2f.c + s_atanf.c (fdlibm's float code: argument reduction to one of four
//                 intervals, an 11-term odd polynomial, separate float operations, no fusing).
// tests/test_libm.py holds them against the host's own libm: cosf and sinf over EVERY float, atanf over every float, atan2f
// over 2^31 pairs - no mismatch on glibc 2.35 / x86-64 -, and the device build against the same values (tests, -m gpu).
#ifndef VS_LIBM_H
#define VS_LIBM_H

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define VS_LIBM_HD __host__ __device__ __forceinline__
#else
#define VS_LIBM_HD inline
#endif

namespace vslibm {

VS_LIBM_HD uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
VS_LIBM_HD float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
VS_LIBM_HD uint32_t abstop12(float x) { return (f2u(x) >> 20) & 0x7ffu; }

// sin (n even) or cos (n odd) of the reduced argument; neg: the table entry whose cosine coefficients are negated
VS_LIBM_HD float sincos_poly(double x, double x2, bool neg, int n) {
    if ((n & 1) == 0) {
        const double s1c = -0x1.555545995a603p-3, s2c = 0x1.1107605230bc4p-7, s3c = -0x1.994eb3774cf24p-13;
        const double x3 = x * x2;
        const double s1 = __builtin_fma(x2, s3c, s2c);
        const double x7 = x3 * x2;
        const double s = __builtin_fma(x3, s1c, x);
        return (float)__builtin_fma(x7, s1, s);
    }
    const double sg = neg ? -1.0 : 1.0;
    const double c0 = sg * 0x1p0, c1c = sg * -0x1.ffffffd0c621cp-2, c2c = sg * 0x1.55553e1068f19p-5, c3 = sg * -0x1.6c087e89a359dp-10,
                 c4 = sg * 0x1.99343027bf8c3p-16;
    const double x4 = x2 * x2;
    const double c2 = __builtin_fma(x2, c4, c3);
    const double c1 = __builtin_fma(x2, c1c, c0);
    const double x6 = x4 * x2;
    const double c = __builtin_fma(x4, c2c, c1);
    return (float)__builtin_fma(x6, c2, c);
}

// |x| >= 120: the quadrant and the reduced argument from 192 bits of 4/pi (reduce_large)
VS_LIBM_HD double reduce_large(uint32_t xi, int* np) {
    const uint32_t inv_pio4[24] = {0xa2u,       0xa2f9u,     0xa2f983u,   0xa2f9836eu, 0xf9836e4eu, 0x836e4e44u, 0x6e4e4415u, 0x4e441529u,
                                   0x441529fcu, 0x1529fc27u, 0x29fc2757u, 0xfc2757d1u, 0x2757d1f5u, 0x57d1f534u, 0xd1f534ddu, 0xf534ddc0u,
                                   0x34ddc0dbu, 0xddc0db62u, 0xc0db6295u, 0xdb629599u, 0x6295993cu, 0x95993c43u, 0x993c4390u, 0x3c439041u};
    const uint32_t* arr = &inv_pio4[(xi >> 26) & 15];
    const int shift = (xi >> 23) & 7;
    uint64_t n, res0, res1, res2;
    xi = (xi & 0xffffffu) | 0x800000u;
    xi <<= shift;
    res0 = (uint32_t)(xi * arr[0]);
    res1 = (uint64_t)xi * arr[4];
    res2 = (uint64_t)xi * arr[8];
    res0 = (res2 >> 32) | (res0 << 32);
    res0 += res1;
    n = (res0 + (1ULL << 61)) >> 62;
    res0 -= n << 62;
    const double x = (double)(int64_t)res0;
    *np = (int)n;
    return x * 0x1.921FB54442D18p-62;
}

template <bool COS>
VS_LIBM_HD float sincosf_impl(float y) {
    double x = (double)y;
    const double sign4[4] = {1.0, -1.0, -1.0, 1.0};
    const uint32_t top = abstop12(y);
    if (top < abstop12(0x1.921FB6p-1f)) {                       // |y| < pi/4
        const double x2 = x * x;
        if (top < abstop12(0x1p-12f)) return COS ? 1.0f : y;
        return sincos_poly(x, x2, false, COS ? 1 : 0);
    }
    if (top < abstop12(120.0f)) {                                // reduce_fast
        const double r = x * 0x1.45F306DC9C883p+23;
        const int n = ((int32_t)r + 0x800000) >> 24;
        x = __builtin_fma(-(double)n, 0x1.921FB54442D18p0, x);
        const double s = sign4[n & 3];
        return sincos_poly(x * s, x * x, (n & 2) != 0, COS ? n ^ 1 : n);
    }
    if (top < abstop12(__builtin_inff())) {
        const uint32_t xi = f2u(y);
        const int sign = (int)(xi >> 31);
        int n;
        x = reduce_large(xi, &n);
        const double s = sign4[(n + sign) & 3];
        return sincos_poly(x * s, x * x, ((n + sign) & 2) != 0, COS ? n ^ 1 : n);
    }
    return y - y;                                                // inf, nan -> nan
}

VS_LIBM_HD float cosf_ref(float y) { return sincosf_impl<true>(y); }
VS_LIBM_HD float sinf_ref(float y) { return sincosf_impl<false>(y); }

VS_LIBM_HD float atanf_ref(float x) {
    const float atanhi[4] = {4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f};
    const float atanlo[4] = {5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f};
    const float aT[11] = {3.3333334327e-01f, -2.0000000298e-01f, 1.4285714924e-01f, -1.1111110449e-01f, 9.0908870101e-02f, -7.6918758452e-02f,
                          6.6610731184e-02f, -5.8335702866e-02f, 4.9768779427e-02f, -3.6531571299e-02f, 1.6285819933e-02f};
    const int32_t hx = (int32_t)f2u(x), ix = hx & 0x7fffffff;
    int id;
    if (ix >= 0x4c000000) {                                       // |x| >= 2^25
        if (ix > 0x7f800000) return x + x;
        return hx > 0 ? atanhi[3] + atanlo[3] : -atanhi[3] - atanlo[3];
    }
    if (ix < 0x3ee00000) {                                        // |x| < 0.4375
        if (ix < 0x31000000) return x;                            // |x| < 2^-29
        id = -1;
    } else {
        x = __builtin_fabsf(x);
        if (ix < 0x3f980000) {
            if (ix < 0x3f300000) { id = 0; x = (2.0f * x - 1.0f) / (2.0f + x); }
            else { id = 1; x = (x - 1.0f) / (x + 1.0f); }
        } else {
            if (ix < 0x401c0000) { id = 2; x = (x - 1.5f) / (1.0f + 1.5f * x); }
            else { id = 3; x = -1.0f / x; }
        }
    }
    const float z = x * x, w = z * z;
    const float s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
    const float s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
    if (id < 0) return x - x * (s1 + s2);
    const float r = atanhi[id] - ((x * (s1 + s2) - atanlo[id]) - x);
    return hx < 0 ? -r : r;
}

VS_LIBM_HD float atan2f_ref(float y, float x) {
    const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
    const int32_t hx = (int32_t)f2u(x), ix = hx & 0x7fffffff, hy = (int32_t)f2u(y), iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;
    if (hx == 0x3f800000) return atanf_ref(y);
    const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
    if (iy == 0) return m < 2 ? y : (m == 2 ? pi + tiny : -pi - tiny);
    if (ix == 0) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000) {
        if (iy == 0x7f800000) return m == 0 ? pi_o_4 + tiny : (m == 1 ? -pi_o_4 - tiny : (m == 2 ? 3.0f * pi_o_4 + tiny : -3.0f * pi_o_4 - tiny));
        return m == 0 ? 0.0f : (m == 1 ? -0.0f : (m == 2 ? pi + tiny : -pi - tiny));
    }
    if (iy == 0x7f800000) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    const int k = (iy - ix) >> 23;
    float z;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
    else if (hx < 0 && k < -60) z = 0.0f;
    else z = atanf_ref(__builtin_fabsf(y / x));
    if (m == 0) return z;
    if (m == 1) return u2f(f2u(z) ^ 0x80000000u);
    if (m == 2) return pi - (z - pi_lo);
    return (z - pi_lo) - pi;
}

}  // namespace vslibm

#endif
