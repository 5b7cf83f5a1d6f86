// glibc_log.hpp -- the reference's f64::ln on the device, bit for bit.
//
// Rust's f64::ln is the platform libm's log.  The core/accessory regression
// (jaccard.rs:105-142) amplifies the LAST BIT of ln J without bound when the fit is flat (the same
// bin-match count at every k-mer length: y_diff is then pure rounding noise and the core distance
// comes out 0 or 1), so "within 1e-6 of the reference on the same box" needs the same ln, not an
// accurate one.  Without a completeness correction ln J is a table over samebits built on the host
// with libm (capi.cpp ensure_ytab); with one, the argument depends on the pair's completeness
// values and the logarithm has to be taken on the device.  This header restates glibc 2.35's
// double-precision log (sysdeps/ieee754/dbl-64/e_log.c = ARM optimized-routines math/log.c, N = 128
// table, constants in glibc_log_data.inc) as the two forms x86-64 glibc dispatches between:
//
//   SKL_LOG_FMA   __ieee754_log_fma  (CPUs with FMA + AVX2): built with -mfma, so GCC contracted
//                 the source's a*b+c expressions; which ones is read off the disassembly of the
//                 shipped object (libm-2.35.a, e_log-fma.o) and reproduced call for call below;
//   SKL_LOG_SSE2  __ieee754_log_sse2 (everything else): the source expressions as written, no
//                 contraction, the tab2 {chi, clo} form of r.
//
// v_fma_f64 / v_mul_f64 / v_add_f64 are IEEE-754 correctly rounded with denormals, exactly like
// vfmadd*sd / mulsd / addsd, so the same sequence gives the same bits (translation units including
// this header are built with -ffp-contract=off).  Which form the running host uses is probed once
// per process against std::log (capi.cpp log_variant()); tests/test_glibc_log_cpu.py checks both
// forms against the glibc objects themselves and the probed one against the host libm.
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define SKL_LOG_HD __host__ __device__ __forceinline__
#else
#define SKL_LOG_HD inline
#endif

namespace skl {

enum LogVariant : int { SKL_LOG_FMA = 0, SKL_LOG_SSE2 = 1 };

constexpr int GLIBC_LOG_DATA_LEN = 530;

// ln2hi, ln2lo, A[5], B[11], tab[128] {invc, logc}, tab2[128] {chi, clo}
static const double glibc_log_data_host[GLIBC_LOG_DATA_LEN] = {
#include "glibc_log_data.inc"
};
#if defined(__HIPCC__)
static __device__ const double glibc_log_data_dev[GLIBC_LOG_DATA_LEN] = {
#include "glibc_log_data.inc"
};
#endif

SKL_LOG_HD const double *glibc_log_data()
{
#if defined(__HIP_DEVICE_COMPILE__)
    return glibc_log_data_dev;
#else
    return glibc_log_data_host;
#endif
}

SKL_LOG_HD uint64_t skl_as_u64(double x)
{
    union { double d; uint64_t u; } v;
    v.d = x;
    return v.u;
}

SKL_LOG_HD double skl_as_f64(uint64_t x)
{
    union { double d; uint64_t u; } v;
    v.u = x;
    return v.d;
}

// log(x) as glibc 2.35 computes it on x86-64; `variant` selects the dispatch target.
SKL_LOG_HD double glibc_log(double x, int variant)
{
    const double *D = glibc_log_data();
    const double Ln2hi = D[0], Ln2lo = D[1];
    const double *A = D + 2;    // poly
    const double *B = D + 7;    // poly1
    const double *T = D + 18;   // {invc, logc}
    const double *T2 = D + 18 + 256;   // {chi, clo}
    uint64_t ix = skl_as_u64(x);
    const uint32_t top = (uint32_t)(ix >> 48);
    const uint64_t LO = 0x3fee000000000000ull;   // asuint64(1.0 - 0x1p-4)
    const uint64_t HI = 0x3ff1090000000000ull;   // asuint64(1.0 + 0x1.09p-4)
    if (ix - LO < HI - LO) {
        // |x - 1| small: log1p polynomial on r = x - 1 with a double-double head
        if (ix == 0x3ff0000000000000ull) return 0.0;
        const double r = x - 1.0;
        const double r2 = r * r;
        const double r3 = r * r2;
        if (variant == SKL_LOG_FMA) {
            double t1 = __builtin_fma(r, B[2], B[1]);
            double t4 = __builtin_fma(r, B[5], B[4]);
            double t7 = __builtin_fma(r, B[8], B[7]);
            t1 = __builtin_fma(r2, B[3], t1);
            t4 = __builtin_fma(r2, B[6], t4);
            t7 = __builtin_fma(r2, B[9], t7);
            t7 = __builtin_fma(r3, B[10], t7);
            const double t = __builtin_fma(t7, r3, t4);
            const double p = __builtin_fma(t, r3, t1);
            const double w27 = __builtin_fma(r, 0x1p27, r);          // r + r * 0x1p27, one rounding
            const double rhi = __builtin_fma(-0x1p27, r, w27);       // ... - r * 0x1p27
            const double rhi2 = rhi * rhi;
            const double rlo = r - rhi;
            const double hi = __builtin_fma(rhi2, B[0], r);
            const double rmh = r - hi;
            const double rph = r + rhi;
            double lo = __builtin_fma(rhi2, B[0], rmh);
            const double brlo = B[0] * rlo;
            lo = __builtin_fma(brlo, rph, lo);
            const double y = __builtin_fma(p, r3, lo);
            return hi + y;
        }
        double y = r3 * (B[1] + r * B[2] + r2 * B[3] +
                         r3 * (B[4] + r * B[5] + r2 * B[6] + r3 * (B[7] + r * B[8] + r2 * B[9] + r3 * B[10])));
        double w = r * 0x1p27;
        const double rhi = r + w - w;
        const double rlo = r - rhi;
        w = rhi * rhi * B[0];
        const double hi = r + w;
        double lo = r - hi + w;
        lo += B[0] * rlo * (rhi + r);
        y += lo;
        y += hi;
        return y;
    }
    if (top - 0x0010u >= 0x7ff0u - 0x0010u) {
        // zero, negative, infinite, NaN or subnormal
        if (ix * 2 == 0) return -__builtin_huge_val();                  // __math_divzero(1)
        if (ix == 0x7ff0000000000000ull) return x;                      // log(inf) = inf
        if ((top & 0x8000u) || (top & 0x7ff0u) == 0x7ff0u) return (x - x) / (x - x);   // __math_invalid: NaN
        ix = skl_as_u64(x * 0x1p52);                                    // subnormal: normalise
        ix -= 52ull << 52;
    }
    // x = 2^k z, z in [OFF, 2 OFF), split into N = 128 subintervals: log(x) = k ln2 + log(c) + log1p(z/c - 1)
    const uint64_t tmp = ix - 0x3fe6000000000000ull;
    const int i = (int)((tmp >> 45) & 127u);
    const int k = (int)((int64_t)tmp >> 52);
    const uint64_t iz = ix - (tmp & (0xfffull << 52));
    const double invc = T[2 * i], logc = T[2 * i + 1];
    const double z = skl_as_f64(iz);
    const double kd = (double)k;
    if (variant == SKL_LOG_FMA) {
        const double r = __builtin_fma(z, invc, -1.0);
        const double w = __builtin_fma(kd, Ln2hi, logc);
        const double p12 = __builtin_fma(r, A[2], A[1]);
        const double hi = r + w;
        const double r2 = r * r;
        double lo = w - hi;
        lo = lo + r;
        lo = __builtin_fma(kd, Ln2lo, lo);
        const double r3 = r * r2;
        const double p34 = __builtin_fma(r, A[4], A[3]);
        lo = __builtin_fma(r2, A[0], lo);
        const double p = __builtin_fma(p34, r2, p12);
        const double y = __builtin_fma(r3, p, lo);
        return y + hi;
    }
    const double r = (z - T2[2 * i] - T2[2 * i + 1]) * invc;
    const double w = kd * Ln2hi + logc;
    const double hi = w + r;
    const double lo = w - hi + r + kd * Ln2lo;
    const double r2 = r * r;
    return lo + r2 * A[0] + r * r2 * (A[1] + r * A[2] + r2 * (A[3] + r * A[4])) + hi;
}

}  // namespace skl
