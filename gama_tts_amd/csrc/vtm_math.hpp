// Short fp64 elementary functions for the per-step parameter conversions.
//
// The device math library's exp2/exp10/tan/cos are general (all ranges, all special values) and
// cost 150-200 instructions each; the conversions of the vocal-tract model only ever see
//   2^x   with |x| < 1000            (pitch -> Hz, dB -> amplitude)
//   tan t with 0 <= t < pi/2         (band-pass bandwidth)
//   cos t with 0 <= t < pi           (band-pass centre frequency)
// so they are evaluated here with plain Cody-Waite reductions and Taylor/Horner kernels, ~1 ulp
// (tests/test_capi_cpu.py::test_short_math_accuracy checks < 4e-16 relative against libm).
// Arguments outside those ranges fall back to the library functions.
#pragma once

#include <cmath>

#if defined(__HIPCC__) || defined(__CUDACC__)
#define GVTM_HD __host__ __device__ __forceinline__
#else
#define GVTM_HD inline
#endif

namespace gvtm {
namespace vmath {

// exp(t) for |t| <= 0.35: Taylor series to t^13 (remainder < 5e-18)
GVTM_HD double exp_small(double t)
{
	double p = 1.0 / 6227020800.0;
	p = std::fma(p, t, 1.0 / 479001600.0);
	p = std::fma(p, t, 1.0 / 39916800.0);
	p = std::fma(p, t, 1.0 / 3628800.0);
	p = std::fma(p, t, 1.0 / 362880.0);
	p = std::fma(p, t, 1.0 / 40320.0);
	p = std::fma(p, t, 1.0 / 5040.0);
	p = std::fma(p, t, 1.0 / 720.0);
	p = std::fma(p, t, 1.0 / 120.0);
	p = std::fma(p, t, 1.0 / 24.0);
	p = std::fma(p, t, 1.0 / 6.0);
	p = std::fma(p, t, 0.5);
	p = std::fma(p, t, 1.0);
	p = std::fma(p, t, 1.0);
	return p;
}

// 2^(x + x_lo), x_lo a small correction term (|x| < 1000)
GVTM_HD double exp2_split(double x, double x_lo)
{
	const double n = std::rint(x);
	const double f = (x - n) + x_lo; // [-0.5, 0.5]
	constexpr double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
	const double t = std::fma(f, ln2_hi, f * ln2_lo);
	return std::ldexp(exp_small(t), static_cast<int>(n));
}

GVTM_HD double exp2_short(double x)
{
	if (!(std::fabs(x) < 1000.0)) return std::exp2(x);
	return exp2_split(x, 0.0);
}

// 10^y = 2^(y * log2(10)) with the product carried in two parts
GVTM_HD double exp10_short(double y)
{
	if (!(std::fabs(y) < 300.0)) return std::pow(10.0, y);
	constexpr double l_hi = 3.32192809488736218171e+00, l_lo = 1.66166637431171660544e-16;
	const double x = y * l_hi;
	const double x_lo = std::fma(y, l_hi, -x) + y * l_lo;
	return exp2_split(x, x_lo);
}

// sin r and cos r for |r| <= pi/4 (Taylor, remainders < 1e-17)
GVTM_HD double sin_kernel(double r)
{
	const double z = r * r;
	double p = 1.0 / 6402373705728000.0; // 1/18!  (sign handled below)
	p = -1.0 / 355687428096000.0 + z * p;      // -1/17!
	p = 1.0 / 1307674368000.0 + z * p;         // 1/15!
	p = -1.0 / 6227020800.0 + z * p;           // -1/13!
	p = 1.0 / 39916800.0 + z * p;              // 1/11!
	p = -1.0 / 362880.0 + z * p;               // -1/9!
	p = 1.0 / 5040.0 + z * p;                  // 1/7!
	p = -1.0 / 120.0 + z * p;                  // -1/5!
	p = 1.0 / 6.0 + z * p;                     // 1/3!   (applied with a minus sign)
	return std::fma(-(r * z), p, r);
}

GVTM_HD double cos_kernel(double r)
{
	const double z = r * r;
	double p = -1.0 / 6402373705728000.0;      // -1/18!
	p = 1.0 / 20922789888000.0 + z * p;        // 1/16!
	p = -1.0 / 87178291200.0 + z * p;          // -1/14!
	p = 1.0 / 479001600.0 + z * p;             // 1/12!
	p = -1.0 / 3628800.0 + z * p;              // -1/10!
	p = 1.0 / 40320.0 + z * p;                 // 1/8!
	p = -1.0 / 720.0 + z * p;                  // -1/6!
	p = 1.0 / 24.0 + z * p;                    // 1/4!
	const double hz = 0.5 * z;
	const double w = 1.0 - hz;
	// 1 - z/2 + z^2 * p, with the rounding error of (1 - z/2) fed back
	return w + (((1.0 - w) - hz) + z * z * p);
}

// r = t - k * pi/2 in two steps (k in {0, 1, 2})
GVTM_HD double reduce_half_pi(double t, double k)
{
	constexpr double p1 = 1.57079632673412561417e+00, p2 = 6.07710050650619224932e-11, p3 = 2.02226624879595063154e-21;
	double r = std::fma(-k, p1, t);
	r = std::fma(-k, p2, r);
	return std::fma(-k, p3, r);
}

GVTM_HD double cos_short(double t)
{
	if (!(t >= 0.0 && t < 3.2)) return std::cos(t);
	const double k = std::rint(t * 0.63661977236758134308); // 2/pi
	const double r = reduce_half_pi(t, k);
	if (k == 0.0) return cos_kernel(r);
	if (k == 1.0) return -sin_kernel(r);
	return -cos_kernel(r);
}

GVTM_HD double tan_short(double t)
{
	if (!(t >= 0.0 && t < 1.5)) return std::tan(t);
	if (t <= 0.78539816339744830962) return sin_kernel(t) / cos_kernel(t);
	const double r = -reduce_half_pi(t, 1.0); // pi/2 - t in (0, pi/4)
	return cos_kernel(r) / sin_kernel(r);
}

} // namespace vmath
} // namespace gvtm
