// CPU check of physicl_amd/csrc/pcl_sincos.h against long-double libm (x87: 64-bit significand).
// Prints: n_checked max_ulp_sin max_ulp_cos worst_x_sin worst_x_cos.  Built and run by tests/test_sincos_cpu.py.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "pcl_sincos.h"

static double ulp_err(double got, long double want) {
    if (want == 0.0L) return got == 0.0 ? 0.0 : 1e30;
    int e;
    std::frexp((double)want, &e);                       // |want| in [2^(e-1), 2^e): ulp = 2^(e-53)
    const long double ulp = std::ldexp(1.0L, e - 53);
    return (double)(fabsl((long double)got - want) / ulp);
}

static uint64_t splitmix(uint64_t &s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

int main(int argc, char **argv) {
    const long n = argc > 1 ? atol(argv[1]) : 2000000;
    const double pi = 3.141592653589793;
    double ms = 0, mc = 0, xs = 0, xc = 0;
    long checked = 0;
    auto check = [&](double x) {
        double s, c;
        pcl_sincos_2pi(x, &s, &c);
        const double es = ulp_err(s, sinl((long double)x)), ec = ulp_err(c, cosl((long double)x));
        if (es > ms) { ms = es; xs = x; }
        if (ec > mc) { mc = ec; xc = x; }
        ++checked;
    };
    uint64_t st = 12345;
    for (long i = 0; i < n; ++i) {
        const double u = (double)(splitmix(st) >> 11) * (1.0 / 9007199254740992.0);   // the 53-bit grid of the draws
        check(u * 2 * pi);                                                               // rtheta, light.py:285
        check(u * pi);                                                                   // rphi
    }
    // the multiples of pi/2 and their neighbours, the ends of the range, tiny arguments
    for (int k = 0; k <= 4; ++k) {
        double x = k * (pi / 2);
        for (int d = -3; d <= 3; ++d) {
            double xx = x;
            for (int j = 0; j < (d < 0 ? -d : d); ++j) xx = std::nextafter(xx, d < 0 ? -1.0 : 10.0);
            if (xx >= 0 && xx <= PCL_SINCOS_XMAX) check(xx);
        }
    }
    for (int e = -1074; e <= 2; e += 7) check(std::ldexp(1.0, e));
    check(0.0);
    check(PCL_SINCOS_XMAX);
    for (int k = 1; k < 4096; ++k) check(k * (PCL_SINCOS_XMAX / 4096));                // includes the quadrant switch points
    for (int k = 0; k <= 8; ++k)                                                         // u on coarse binary fractions
        for (int d = 0; d < 64; ++d) check(((k / 8.0) + d * 0x1p-53) * 2 * pi);
    printf("%ld %.4f %.4f %.17g %.17g\n", checked, ms, mc, xs, xc);
    return 0;
}
