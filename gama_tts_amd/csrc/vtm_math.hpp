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
#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
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

// ---------------------------------------------------------------------------------------------
// powf(2.0f, x) and powf(10.0f, y) AS glibc 2.35 COMPUTES THEM (sysdeps/ieee754/flt-32/e_powf.c,
// e_exp2f_data.c — the ARM optimized-routines algorithm), for the all-float path.
//
// With TFloat = float the reference model calls std::pow(float, float) = powf once per step for
// the pitch and three times for amplitudes.  The oscillator phase is a float running sum of the
// pitch increments, so a single last-bit difference in one powf result shifts the phase of
// everything after it: "correctly rounded" is not good enough (glibc's powf errs by up to 0.82 ulp),
// the same function has to be computed.  powf(x, y) = exp2_inline(y * log2_inline(x)) in double:
//   * log2_inline(2.0f) is exactly 1, log2_inline(10.0f) is the constant below (its table/polynomial
//     value, 1.7e-12 above the true log2 10);
//   * exp2_inline: k/32 + r split with the 0x1.8p52/32 shift, 2^(k/32) from a 32-entry table,
//     2^r by a cubic.
// tests/test_capi_cpu.py compares both against this machine's libm over every float in the ranges
// the model produces (bit-identical, fused or not: the float rounding absorbs the difference).
GVTM_HD float powf_exp2_core(double xd)
{
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
	// tab[i] = bits(2^(i/32)) - (i << 47)
	constexpr uint64_t tab[32] = {
		0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
		0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
		0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
		0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
		0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
		0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
		0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
		0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};
	constexpr double c0 = 0x1.c6af84b912394p-5, c1 = 0x1.ebfce50fac4f3p-3, c2 = 0x1.62e42ff0c52d6p-1;
	constexpr double shift = 0x1.8p+52 / 32;
	double kd = xd + shift; // rounds xd to a multiple of 1/32; the integer lands in the low mantissa bits
	uint64_t ki;
	std::memcpy(&ki, &kd, sizeof(ki));
	kd -= shift;
	const double r = xd - kd;
	const uint64_t t = tab[ki & 31u] + (ki << 47);
	double s;
	std::memcpy(&s, &t, sizeof(s));
	const double z = c0 * r + c1;
	const double r2 = r * r;
	double y = c2 * r + 1.0;
	y = z * r2 + y;
	y = y * s;
	return static_cast<float>(y);
}

// powf(2.0f, x), |x| < 100 (the model's pitch range gives |x| < 10)
GVTM_HD float powf_base2(float x)
{
	if (!(std::fabs(x) < 100.0f)) return static_cast<float>(std::exp2(static_cast<double>(x)));
	return powf_exp2_core(static_cast<double>(x));
}

// powf(10.0f, y), |y| < 30 (dB conversions give -3 <= y < 0)
GVTM_HD float powf_base10(float y)
{
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
	if (!(std::fabs(y) < 30.0f)) return static_cast<float>(std::pow(10.0, static_cast<double>(y)));
	constexpr double log2_10_glibc = 0x1.a934f0979b22dp+1; // log2_inline(10.0f)
	const double ylogx = static_cast<double>(y) * log2_10_glibc;
	return powf_exp2_core(ylogx);
}

// ---------------------------------------------------------------------------------------------
// cosf and tanf AS glibc 2.35 COMPUTES THEM, for the band-pass design of the all-float path
// (BandpassFilter::update calls std::cos / std::tan on floats once per step; a last-bit difference in
// a coefficient perturbs the frication noise, which in quiet passages is all there is).
//   cosf: sysdeps/ieee754/flt-32/s_cosf.c, s_sincosf.h — double arithmetic, fast quadrant reduction,
//         degree-8 / degree-7 polynomials;
//   tanf: s_tanf.c, k_tanf.c, e_rem_pio2f.c — the fdlibm float kernel (odd polynomial to x^27, the
//         pi/4 - x reflection above 0.6744, -1/tan for the second octant).
// Compared with this machine's libm over every float of the ranges in use (tests/test_capi_cpu.py):
// cos on [0, 3.2], tan on [0, 1.38] (bandwidth < 0.44 fs) are bit-identical.
GVTM_HD uint32_t float_bits(float f)
{
	uint32_t u;
	std::memcpy(&u, &f, sizeof(u));
	return u;
}
GVTM_HD float bits_float(uint32_t u)
{
	float f;
	std::memcpy(&f, &u, sizeof(f));
	return f;
}

GVTM_HD float cosf_glibc(float y)
{
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
	const uint32_t top = (float_bits(y) >> 20) & 0x7ffu;
	if (!(top < ((float_bits(120.0f) >> 20) & 0x7ffu))) return static_cast<float>(std::cos(static_cast<double>(y)));
	constexpr double hpi_inv = 0x1.45F306DC9C883p+23, hpi = 0x1.921FB54442D18p0;
	constexpr double c0 = 0x1p0, c1 = -0x1.ffffffd0c621cp-2, c2 = 0x1.55553e1068f19p-5, c3 = -0x1.6c087e89a359dp-10,
			c4 = 0x1.99343027bf8c3p-16;
	constexpr double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
	double x = y;
	int n = 1;       // polynomial selector: odd = cosine polynomial, even = sine polynomial
	double flip = 1.0; // sign of the cosine polynomial (second table of the reference)
	if (top < ((float_bits(0x1.921FB6p-1f) >> 20) & 0x7ffu)) {
		if (top < ((float_bits(0x1p-12f) >> 20) & 0x7ffu)) return 1.0f;
	} else {
		const double r = x * hpi_inv;
		const int q = (static_cast<int32_t>(r) + 0x800000) >> 24;
		x = x - q * hpi;
		const double sgn = ((q & 3) == 1 || (q & 3) == 2) ? -1.0 : 1.0; // sign[] = {1, -1, -1, 1}
		if (q & 2) flip = -1.0;
		n = q ^ 1;
		const double x2r = x * x;
		x = x * sgn;
		if ((n & 1) == 0) {
			const double x3 = x * x2r;
			const double t1 = s2 + x2r * s3;
			const double x7 = x3 * x2r;
			const double t = x + x3 * s1;
			return static_cast<float>(t + x7 * t1);
		}
		const double x4 = x2r * x2r;
		const double d2 = flip * c3 + x2r * (flip * c4);
		const double d1 = flip * c0 + x2r * (flip * c1);
		const double x6 = x4 * x2r;
		const double d = d1 + x4 * (flip * c2);
		return static_cast<float>(d + x6 * d2);
	}
	const double x2 = x * x;
	const double x4 = x2 * x2;
	const double d2 = c3 + x2 * c4;
	const double d1 = c0 + x2 * c1;
	const double x6 = x4 * x2;
	const double d = d1 + x4 * c2;
	return static_cast<float>(d + x6 * d2);
}

// __kernel_tanf(x, y, iy) for x >= 0
GVTM_HD float tanf_kernel_glibc(float x, float y, int iy)
{
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
	constexpr float T0 = 3.3333334327e-01f, T1 = 1.3333334029e-01f, T2 = 5.3968254477e-02f, T3 = 2.1869488060e-02f,
			T4 = 8.8632395491e-03f, T5 = 3.5920790397e-03f, T6 = 1.4562094584e-03f, T7 = 5.8804126456e-04f,
			T8 = 2.4646313977e-04f, T9 = 7.8179444245e-05f, T10 = 7.1407252108e-05f, T11 = -1.8558637748e-05f,
			T12 = 2.5907305826e-05f;
	constexpr float pio4 = 7.8539812565e-01f, pio4lo = 3.7748947079e-08f;
	const uint32_t ix = float_bits(x) & 0x7fffffffu;
	if (ix < 0x39000000u) { // |x| < 2^-13
		if (static_cast<int>(x) == 0) {
			if (iy == 1) return x;
			return -1.0f / x;
		}
	}
	const bool big = ix >= 0x3f2ca140u; // |x| >= 0.6744
	float z, r, v, w, s;
	if (big) {
		z = pio4 - x;
		w = pio4lo - y;
		x = z + w;
		y = 0.0f;
		if (std::fabs(x) < 0x1p-13f) return static_cast<float>(iy) * (1.0f - 2.0f * static_cast<float>(iy) * x);
	}
	z = x * x;
	w = z * z;
	r = T1 + w * (T3 + w * (T5 + w * (T7 + w * (T9 + w * T11))));
	v = z * (T2 + w * (T4 + w * (T6 + w * (T8 + w * (T10 + w * T12)))));
	s = z * x;
	r = y + z * (s * (r + v) + y);
	r += T0 * s;
	w = x + r;
	if (big) {
		v = static_cast<float>(iy);
		return v - 2.0f * (x - (w * w / (w + v) - r));
	}
	if (iy == 1) return w;
	// -1 / (x + r), accurately
	z = bits_float(float_bits(w) & 0xfffff000u);
	v = r - (z - x);
	const float a = -1.0f / w;
	const float t = bits_float(float_bits(a) & 0xfffff000u);
	s = 1.0f + t * z;
	return t + a * (s + t * v);
}

GVTM_HD float tanf_glibc(float x)
{
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
	if (!(x >= 0.0f && x <= 1.38f)) return static_cast<float>(std::tan(static_cast<double>(x)));
	const uint32_t ix = float_bits(x);
	if (ix <= 0x3f490fdau) return tanf_kernel_glibc(x, 0.0f, 1);
	// __ieee754_rem_pio2f for pi/4 < x < 3pi/4: x - pi/2 in two floats
	constexpr float pio2_1 = 1.5707855225e+00f, pio2_1t = 1.0804334124e-05f, pio2_2 = 1.0804273188e-05f,
			pio2_2t = 6.0770999344e-11f;
	float z = x - pio2_1, y0, y1;
	if ((ix & 0xfffffff0u) != 0x3fc90fd0u) {
		y0 = z - pio2_1t;
		y1 = (z - y0) - pio2_1t;
	} else {
		z -= pio2_2;
		y0 = z - pio2_2t;
		y1 = (z - y0) - pio2_2t;
	}
	// y0 < 0 here (x < pi/2): the kernel works on |y0| and the sign is restored by symmetry
	const float m = tanf_kernel_glibc(-y0, -y1, -1);
	return -m;
}

} // namespace vmath
} // namespace gvtm
