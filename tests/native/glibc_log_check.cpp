// glibc_log_check.cpp -- CPU check of csrc/glibc_log.hpp (test infrastructure).
//   glibc_log_check host N          : which variant equals std::log on this host for N arguments?
//   glibc_log_check objects N       : (built with -DWITH_GLIBC_OBJECTS, linked against libm-2.35.a)
//                                     both variants against __ieee754_log_fma / __ieee754_log_sse2
// Arguments cover what the distance path feeds the logarithm: Jaccard values in (0, 1], the
// |x - 1| < 1/16 branch, powers of two, subnormals, zero, and uniform random bit patterns.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../../sketchlib.rust_amd/csrc/glibc_log.hpp"

#ifdef WITH_GLIBC_OBJECTS
extern "C" double __ieee754_log_fma(double);
extern "C" double __ieee754_log_sse2(double);
// e_log.o also holds glibc's ifunc resolver for log, which reads the dynamic loader's CPU feature
// block; zeros make it pick the baseline form (it is not called by this program)
extern "C" { char _dl_x86_cpu_features[4096]; }
#endif

static bool same(double a, double b)
{
    if (std::isnan(a) && std::isnan(b)) return true;
    return skl::skl_as_u64(a) == skl::skl_as_u64(b);
}

int main(int argc, char **argv)
{
    const char *mode = argc > 1 ? argv[1] : "host";
    const size_t n = argc > 2 ? strtoull(argv[2], nullptr, 10) : 2000000;
    std::mt19937_64 rng(0x5EED);
    std::vector<double> xs;
    xs.reserve(n + 100000);
    std::uniform_real_distribution<double> u01(0.0, 1.0);
    for (size_t i = 0; i < n / 4; ++i) xs.push_back(u01(rng));                       // J in (0, 1)
    for (size_t i = 0; i < n / 4; ++i) xs.push_back(0.9375 + 0.13 * u01(rng));       // around 1
    for (size_t i = 0; i < n / 4; ++i) {                                             // J(samebits) / factor, as the path builds it
        const uint32_t ss64 = 1u + (uint32_t)(rng() % 200);
        const uint32_t maxnbits = ss64 * 64u, expected = maxnbits >> 14;
        const uint32_t sb = (uint32_t)(rng() % (maxnbits + 1));
        const uint32_t diff = sb > expected ? sb - expected : 0u;
        double j = ((double)diff * (double)maxnbits) / (double)(maxnbits - expected) / (double)(64u * ss64);
        const double c1 = 0.5 + 0.5 * u01(rng), c2 = 0.5 + 0.5 * u01(rng);
        j = j / (c1 * c2 / (c1 + c2 - c1 * c2));
        xs.push_back(j < 1.0 ? j : 1.0);
    }
    for (size_t i = 0; i < n / 4; ++i) xs.push_back(skl::skl_as_f64(rng()));          // any bit pattern
    for (int e = -1074; e <= 1023; ++e) xs.push_back(std::ldexp(1.0, e));
    const double specials[] = {0.0, -0.0, 1.0, -1.0, INFINITY, -INFINITY, NAN, 5e-324, 2.2250738585072014e-308,
                               0.9375, 1.064697265625, 0.93749999999999989, 1.0646972656250002};
    for (double s : specials) xs.push_back(s);

    if (!strcmp(mode, "host")) {
        size_t bad[2] = {0, 0};
        for (double x : xs) {
            const double ref = std::log(x);
            for (int v = 0; v < 2; ++v) bad[v] += !same(skl::glibc_log(x, v), ref);
        }
        printf("{\"arguments\": %zu, \"mismatch_fma\": %zu, \"mismatch_sse2\": %zu}\n", xs.size(), bad[0], bad[1]);
        return 0;
    }
#ifdef WITH_GLIBC_OBJECTS
    size_t bad[2] = {0, 0};
    for (double x : xs) {
        bad[0] += !same(skl::glibc_log(x, skl::SKL_LOG_FMA), __ieee754_log_fma(x));
        bad[1] += !same(skl::glibc_log(x, skl::SKL_LOG_SSE2), __ieee754_log_sse2(x));
    }
    printf("{\"arguments\": %zu, \"mismatch_fma\": %zu, \"mismatch_sse2\": %zu}\n", xs.size(), bad[0], bad[1]);
    return 0;
#else
    fprintf(stderr, "built without the glibc objects\n");
    return 2;
#endif
}
