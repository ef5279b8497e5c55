// pcl_sincos.h -- sin and cos of an angle known to lie in [0, 2*pi], fp64, < 1 ulp each.
//
// The scatter step evaluates sin/cos of rtheta = u*2*pi and rphi = u*pi only (physicl/light.py:285, 309-311),
// so the general-purpose argument reduction of a libm sincos (huge arguments, Payne-Hanek) is dead weight and
// -- once K steps run per pass over the store -- the two sincos calls are the largest single item of a hit.
// This version reduces with a two-term Cody-Waite split of pi/2 (exact first step: the multiple of pi/2 is at
// most 4), evaluates the Taylor polynomials of sin (to y^17) and cos (to y^16) on |y| <= pi/4 with explicit
// FMAs and a compensated tail, and swaps/negates by quadrant.  ~45 instructions instead of ~110.
// Pure C++ with compiler builtins: the same text is compiled by hipcc / hipRTC for the device and by g++ in
// tests/native/sincos_check.cpp, which measures the error against long-double libm on the CPU.
#ifndef PCL_SINCOS_H
#define PCL_SINCOS_H

#ifndef PCL_SC_FN
#define PCL_SC_FN static inline
#endif

// valid for 0 <= x <= PCL_SINCOS_XMAX (callers fall back to the library sincos outside it, NaN included)
#define PCL_SINCOS_XMAX 6.5

PCL_SC_FN void pcl_sincos_2pi(double x, double *s_out, double *c_out) {
    // x = fn * pi/2 + (y + yl),  fn in {0..4},  |y| <= pi/4
    const double fn = __builtin_rint(x * 0x1.45f306dc9c883p-1);      // 2/pi
    const double r0 = __builtin_fma(-fn, 0x1.921fb54442d18p+0, x);   // fl(pi/2): exact (both are multiples of 2^-53, |r0| < 1)
    const double t = fn * 0x1.1a62633145c07p-54;                     // pi/2 - fl(pi/2)
    const double y = r0 - t;
    const double yl = (r0 - y) - t;
    const double z = y * y;
    // sin(y + yl) = y + (y^3 * S(z) + yl * (1 - z/2))
    double ps = 2.8114572543455206e-15;                              //  1/17!
    ps = __builtin_fma(ps, z, -7.647163731819816e-13);               // -1/15!
    ps = __builtin_fma(ps, z, 1.6059043836821613e-10);               //  1/13!
    ps = __builtin_fma(ps, z, -2.505210838544172e-08);               // -1/11!
    ps = __builtin_fma(ps, z, 2.7557319223985893e-06);               //  1/9!
    ps = __builtin_fma(ps, z, -0.0001984126984126984);               // -1/7!
    ps = __builtin_fma(ps, z, 0.008333333333333333);                 //  1/5!
    ps = __builtin_fma(ps, z, -0.16666666666666666);                 // -1/3!
    const double hz = 0.5 * z;
    const double sn = y + __builtin_fma(y * z, ps, __builtin_fma(-hz, yl, yl));
    // cos(y + yl) = (1 - z/2) + (z^2 * C(z) - y * yl), the leading difference compensated
    double pc = 4.779477332387385e-14;                               //  1/16!
    pc = __builtin_fma(pc, z, -1.1470745597729725e-11);              // -1/14!
    pc = __builtin_fma(pc, z, 2.08767569878681e-09);                 //  1/12!
    pc = __builtin_fma(pc, z, -2.755731922398589e-07);               // -1/10!
    pc = __builtin_fma(pc, z, 2.48015873015873e-05);                 //  1/8!
    pc = __builtin_fma(pc, z, -0.001388888888888889);                // -1/6!
    pc = __builtin_fma(pc, z, 0.041666666666666664);                 //  1/4!
    const double w = 1.0 - hz;
    const double cs = w + (((1.0 - w) - hz) + __builtin_fma(z * z, pc, -(y * yl)));
    // quadrant
    const int q = (int)fn & 3;
    const double s1 = (q & 1) ? cs : sn, c1 = (q & 1) ? sn : cs;
    *s_out = (q & 2) ? -s1 : s1;
    *c_out = ((q + 1) & 2) ? -c1 : c1;
}

#endif // PCL_SINCOS_H
