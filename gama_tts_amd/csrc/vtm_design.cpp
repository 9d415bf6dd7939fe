// Design-time math for the batched vocal-tract model (host; fp64, or float for GVTM_PRECISION_F32).
//
// Each routine states which reference routine defines the numbers it must
// reproduce; tests/test_capi_cpu.py compares every table with the oracle
// bit for bit through gvtm_plan_table().
#include "vtm_design.hpp"

#include <algorithm>
#include <cmath>
#include <sstream>

namespace gvtm {

namespace {

constexpr double kPi = 3.14159265358979323846;

// The reference's classes are templates over TFloat and write their literals as TFloat
// (1.0f, TFloat{...}), so with TFloat = float every operation below is a float operation and
// std::pow/cos/sin/sqrt/rint/abs resolve to the float overloads.  T plays TFloat here.

// --- glottal-source anti-alias FIR -------------------------------------------------
// Numbers defined by WavetableGlottalSourceFIRFilter::maximallyFlat / rationalApproximation /
// trim (vtm/WavetableGlottalSourceFIRFilter.h:137-215, :316-361, :227-235) with
// beta = 0.2, gamma = 0.1, cutoff = 1e-8 (vtm/WavetableGlottalSource.h:94-96).
struct Rational { int numerator, denominator, order; };

template <typename T>
Rational best_rational(T value, int order)
{
	constexpr int kLimit = 200;
	Rational r{0, 0, -1};
	if (order <= 0) return r;
	const T frac = std::abs(value - static_cast<int>(value));
	const int max_den = std::min(2 * order, kLimit);
	T best = 1.0;
	int best_num = 0;
	for (int den = order; den <= max_den; ++den) {
		const T scaled = den * frac;
		const int nearest = static_cast<int>(scaled + T(0.5));
		const T err = std::abs((scaled - static_cast<double>(nearest)) / den); // evaluated in double (:339)
		if (err < best) {
			best = err;
			best_num = nearest;
			r.denominator = den;
		}
	}
	r.numerator = static_cast<int>(std::abs(value)) * r.denominator + best_num;
	if (value < T(0)) r.numerator = -r.numerator;
	r.order = r.denominator - 1;
	if (r.numerator == r.denominator) {
		r.denominator = max_den;
		r.order = r.numerator = r.denominator - 1;
	}
	return r;
}

template <typename T>
std::vector<T> design_glottal_fir()
{
	const T beta = 0.2, gamma = 0.1, cutoff = 0.00000001;
	constexpr int kLimit = 200;
	const int nt0 = static_cast<int>(T(1) / (T(4) * gamma * gamma));
	const T ac = (T(1) + std::cos((T(2) * static_cast<T>(kPi)) * beta)) / T(2);
	const Rational q = best_rational<T>(ac, nt0);
	const int np = q.denominator;
	const int nt = q.order;
	const int numer = q.numerator == 0 ? 1 : q.numerator;
	const int n = 2 * np - 1;

	std::vector<T> mag(kLimit + 2, T(0)), cosv(kLimit + 2, T(0)), half(kLimit + 2, T(0));
	mag[1] = cosv[1] = 1.0;
	const int terms = nt - numer;
	for (int i = 2; i <= np; ++i) {
		cosv[i] = std::cos((T(2) * static_cast<T>(kPi)) * (static_cast<T>(i - 1) / n));
		const T x = (T(1) - cosv[i]) / T(2);
		if (numer == nt) continue;
		T y = x, sum = 1.0;
		for (int j = 1; j <= terms; ++j) {
			T z = y;
			for (int jj = 1; jj <= numer - 1; ++jj) z *= T(1) + (static_cast<T>(j) / jj);
			y *= x;
			sum += z;
		}
		// std::pow(TFloat, int) promotes to double, and so does the product
		mag[i] = static_cast<T>(sum * std::pow(static_cast<double>(T(1) - x), static_cast<double>(numer)));
	}
	// N-point inverse DFT of the symmetric magnitude response
	for (int i = 1; i <= np; ++i) {
		T acc = mag[1] / T(2);
		for (int j = 2; j <= np; ++j) {
			int m = ((i - 1) * (j - 1)) % n;
			if (m > nt) m = n - m;
			acc += cosv[m + 1] * mag[j];
		}
		half[i] = acc * (T(2) / static_cast<T>(n));
	}
	int keep = np;
	for (int i = np; i > 0; --i) {
		if (std::abs(half[i]) >= std::abs(cutoff)) { keep = i; break; }
	}
	// mirror the half response into a linear-phase filter: h[keep] ... h[1] ... h[keep]
	std::vector<T> taps;
	taps.reserve(2 * keep - 1);
	for (int i = keep; i >= 1; --i) taps.push_back(half[i]);
	for (int i = 2; i <= keep; ++i) taps.push_back(half[i]);
	return taps;
}

// --- sample-rate-converter prototype ------------------------------------------------
// Kaiser-windowed sinc of SampleRateConverter::initializeFilter / Izero
// (vtm/SampleRateConverter.h:230-255, :175-194).
template <typename T>
T bessel_i0(T x)
{
	T sum = 1, term = 1;
	const T half = x / T(2);
	const T eps = 1E-21;
	int n = 1;
	do {
		T t = half / n;
		n += 1;
		t *= t;
		term *= t;
		sum += term;
	} while (term >= eps * sum);
	return sum;
}

template <typename T>
void design_src_filter(std::vector<T>& h, std::vector<T>& dh)
{
	const T beta = 5.658;
	const T cutoff = 11.0 / 13.0;
	h.assign(kSrcFilterLength, T(0));
	dh.assign(kSrcFilterLength, T(0));
	h[0] = cutoff;
	const T dx = kPi / kSrcPhases;
	for (unsigned i = 1; i < static_cast<unsigned>(kSrcFilterLength); ++i) {
		const T y = i * dx;
		h[i] = std::sin(y * cutoff) / y;
	}
	const T inv_i0 = T(1) / bessel_i0<T>(beta);
	for (unsigned i = 0; i < static_cast<unsigned>(kSrcFilterLength); ++i) {
		const T t = static_cast<T>(i) / kSrcFilterLength;
		h[i] *= bessel_i0<T>(beta * std::sqrt(T(1) - (t * t))) * inv_i0;
	}
	for (int i = 0; i + 1 < kSrcFilterLength; ++i) dh[i] = h[i + 1] - h[i];
	dh[kSrcFilterLength - 1] = T(0) - h[kSrcFilterLength - 1];
}

// --- glottal wavetable ----------------------------------------------------------------
// WavetableGlottalSource constructor (vtm/WavetableGlottalSource.h:90-141).
template <typename T>
void design_wavetable(const gvtm_config& c, DeviceConstants& k, std::vector<T>& table)
{
	const unsigned len = kWavetableLength;
	const T tp = static_cast<T>(c.glottal_pulse_tp), tn_min = static_cast<T>(c.glottal_pulse_tn_min),
			tn_max = static_cast<T>(c.glottal_pulse_tn_max);
	k.table_div1 = static_cast<unsigned>(std::rint(len * (tp / T(100))));
	k.table_div2 = static_cast<unsigned>(std::rint(len * ((tp + tn_max) / T(100))));
	k.tn_delta = std::rint(len * ((tn_max - tn_min) / T(100)));
	k.basic_increment = len / static_cast<T>(k.sample_rate);
	table.assign(len, T(0));
	if (c.waveform == 0) {
		const T fall = k.table_div2 - k.table_div1;
		for (unsigned i = 0; i < k.table_div1; ++i) {
			const T x = static_cast<T>(i) / k.table_div1;
			const T x2 = x * x;
			const T x3 = x2 * x;
			table[i] = (T(3) * x2) - (T(2) * x3);
		}
		for (unsigned i = k.table_div1, j = 0; i < k.table_div2 && i < len; ++i, ++j) {
			const T x = static_cast<T>(j) / fall;
			table[i] = T(1) - (x * x);
		}
	} else {
		for (unsigned i = 0; i < len; ++i) {
			table[i] = std::sin((static_cast<T>(i) / len) * T(2) * static_cast<T>(kPi));
		}
	}
}

template <typename T>
T junction(T left_radius, T right_radius)
{
	const T a = left_radius * left_radius;
	const T b = right_radius * right_radius;
	return (a - b) / (a + b);
}

// Util::amplitude60dB (vtm/VTMUtil.h:48-67)
template <typename T>
T amplitude_60db_t(T db)
{
	if (db <= T(0)) return T(0);
	if (db == T(60)) return T(1);
	return std::pow(T(10), (db - T(60)) * static_cast<T>(1.0 / 20.0));
}

template <typename T>
struct Tables {
	std::vector<T> fir, src_h, src_dh, wavetable;
};

// Everything derived from the configuration, computed in T as the reference's TFloat does.
template <typename T>
std::string design_numbers(const gvtm_config& c, double control_rate, DeviceConstants& k, Tables<T>& tb)
{
	std::ostringstream err;
	auto finite_pos = [](double v) { return std::isfinite(v) && v > 0.0; };

	// loadConfiguration (vtm/VocalTractModel0.h:266-305): length clamp, radius scaling
	T length = static_cast<T>(c.vocal_tract_length_offset) + static_cast<T>(c.vocal_tract_length);
	length = std::min(std::max(length, T(3)), T(30));
	const T global_radius = static_cast<T>(c.global_radius_coef);
	const T global_nasal = static_cast<T>(c.global_nasal_radius_coef);
	const T aperture = static_cast<T>(c.aperture_radius) * global_radius;
	T nasal[6] = {T(0)};
	for (int i = 0; i < 5; ++i) nasal[i + 1] = static_cast<T>(c.nasal_radius[i]) * global_nasal;
	for (int i = 0; i < 8; ++i) k.radius_coef[i] = static_cast<T>(c.radius_coef[i]) * global_radius;
	for (int i = 1; i < 6; ++i) {
		if (!finite_pos(nasal[i])) return "nasal radii must be > 0";
	}
	if (!finite_pos(aperture)) return "aperture_radius must be > 0";

	// initializeSynthesizer (vtm/VocalTractModel0.h:338-392, VocalTractModel2.h:413-467)
	const T speed = T(331.4) + (T(0.6) * static_cast<T>(c.temperature));
	const int sections = c.tube_layout == GVTM_TUBE_30_18 ? 30 : 10; // TOTAL_SECTIONS (VocalTractModel4.h:194 / VocalTractModel0.h:128)
	k.sample_rate = static_cast<int>((speed * (sections * c.section_delay) * 100.0f) / length);
	if (k.sample_rate < 2000) return "internal sample rate too low";
	const T nyquist = k.sample_rate / 2.0f; // int / float -> float arithmetic, as in the reference
	k.breathiness = static_cast<T>(c.breathiness) / T(100);
	const T mix_amp = amplitude_60db_t<T>(static_cast<T>(c.mix_offset));
	if (!(mix_amp > T(0))) return "mix_offset must be > 0 dB";
	k.crossmix_factor = T(1) / mix_amp;
	k.damping = T(1) - (static_cast<T>(c.loss_factor) / T(100));

	const T mouth_ap = (nyquist - static_cast<T>(c.mouth_coefficient)) / nyquist;
	k.mouth_b0_refl = T(1) - std::abs(mouth_ap); // ReflectionFilter.h:55-60
	k.mouth_a1_refl = -mouth_ap;
	k.mouth_a_rad = mouth_ap;                    // RadiationFilter.h:54-61: b0 = a, b1 = a1 = -a
	const T nose_ap = (nyquist - static_cast<T>(c.nose_coefficient)) / nyquist;
	k.nose_b0_refl = T(1) - std::abs(nose_ap);
	k.nose_a1_refl = -nose_ap;
	k.nose_a_rad = nose_ap;

	// initializeNasalCavity (vtm/VocalTractModel0.h:457-470)
	k.nasal_k[0] = 0.0;
	for (int i = 1; i < 5; ++i) k.nasal_k[i] = junction<T>(nasal[i], nasal[i + 1]);
	k.nasal_k[5] = junction<T>(nasal[5], aperture);
	k.aperture_radius2 = aperture * aperture;
	k.nasal_r2_sq = nasal[1] * nasal[1];

	// Throat (vtm/Throat.h:52-60)
	const T fs = static_cast<T>(k.sample_rate);
	const T throat_b0 = (static_cast<T>(c.throat_cutoff) * T(2)) / fs;
	k.throat_b0 = throat_b0;
	k.throat_a1 = throat_b0 - T(1);
	k.throat_gain = amplitude_60db_t<T>(static_cast<T>(c.throat_volume));
	k.bp_T = T(1) / fs; // BandpassFilter::update (BandpassFilter.h:104)

	// Controller::synthesize (vtm_control_model/Controller.cpp:286-287)
	k.control_steps = static_cast<unsigned>(std::rint(static_cast<double>(k.sample_rate) / control_rate));
	if (k.control_steps == 0) return "control_rate above the internal sample rate";
	k.interp_coef = 1.0f / k.control_steps;

	// SampleRateConverter::initializeConversion (vtm/SampleRateConverter.h:136-164)
	const T ratio = static_cast<T>(c.output_rate) / fs;
	k.src_ratio = ratio;
	k.time_inc = static_cast<unsigned>(std::rint(std::pow(2.0, 16) / ratio)); // double arithmetic (:145)
	if (k.time_inc == 0) return "output_rate too high for the 16.16 time register";
	const T rounded_ratio = std::pow(2.0, 16) / k.time_inc;
	k.upsampling = ratio >= T(1);
	if (k.upsampling) {
		k.phase_inc = 0;
		k.pad = kSrcZeroCrossings;
	} else {
		k.phase_inc = static_cast<unsigned>(std::rint(ratio * 65536));
		k.pad = static_cast<int>(kSrcZeroCrossings / rounded_ratio) + 1;
	}
	if (k.pad > kMaxPad || (k.phase_inc == 0 && !k.upsampling)) {
		err << "output_rate / internal rate = " << static_cast<double>(ratio) << " is below the supported down-sampling range";
		return err.str();
	}

	design_wavetable<T>(c, k, tb.wavetable);
	tb.fir = design_glottal_fir<T>();
	k.fir_taps = static_cast<int>(tb.fir.size());
	if (k.fir_taps > kMaxFirTaps || k.fir_taps > 49) return "glottal FIR longer than the device pre-roll (49 taps)";
	design_src_filter<T>(tb.src_h, tb.src_dh);
	return "";
}

} // namespace

double amplitude_60db(double db)
{
	return amplitude_60db_t<double>(db);
}

std::string design_plan(const gvtm_config& c, double control_rate, Design& out)
{
	std::ostringstream err;
	auto finite_pos = [](double v) { return std::isfinite(v) && v > 0.0; };
	if (!finite_pos(c.output_rate)) return "output_rate must be > 0";
	if (!finite_pos(control_rate)) return "control_rate must be > 0";
	if (c.section_delay < 1 || c.section_delay > kMaxSectionDelay) {
		err << "section_delay must be in 1.." << kMaxSectionDelay;
		return err.str();
	}
	if (c.waveform != 0 && c.waveform != 1) return "waveform must be 0 (pulse) or 1 (sine)";
	if (c.tube_layout != GVTM_TUBE_10_6 && c.tube_layout != GVTM_TUBE_30_18) return "unknown tube_layout";
	if (c.tube_layout == GVTM_TUBE_30_18 && c.section_delay != 1) return "the 30+18-section tube (VocalTractModel4) runs with section_delay 1";
	if (c.precision != GVTM_PRECISION_F64 && c.precision != GVTM_PRECISION_MIXED && c.precision != GVTM_PRECISION_F32) return "unknown precision";
	if (!(c.glottal_pulse_tp > 0.0) || c.glottal_pulse_tn_min < 0.0 || c.glottal_pulse_tn_max < c.glottal_pulse_tn_min ||
			c.glottal_pulse_tp + c.glottal_pulse_tn_max > 100.0) {
		return "glottal pulse shape needs tp > 0, 0 <= tn_min <= tn_max, tp + tn_max <= 100";
	}
	if (!(c.temperature > -273.0) || !std::isfinite(c.temperature)) return "temperature out of range";

	out.config = c;
	out.control_rate = control_rate;
	DeviceConstants& k = out.k;
	k = DeviceConstants{};
	k.section_delay = c.section_delay;
	k.layout = c.tube_layout;
	k.waveform = c.waveform;
	k.modulation = c.noise_modulation != 0;
	out.f32 = c.precision == GVTM_PRECISION_F32;
	if (out.f32) {
		// TFloat = float: every derived number and table is computed in float, as VocalTractModel0<float> does;
		// the constants are carried in DeviceConstants' double fields (exact) and narrowed back on the device
		Tables<float> tb;
		const std::string msg = design_numbers<float>(c, control_rate, k, tb);
		if (!msg.empty()) return msg;
		out.fir_f = tb.fir; out.src_h_f = tb.src_h; out.src_dh_f = tb.src_dh; out.wavetable_f = tb.wavetable;
		out.fir.assign(tb.fir.begin(), tb.fir.end());
		out.src_h.assign(tb.src_h.begin(), tb.src_h.end());
		out.src_dh.assign(tb.src_dh.begin(), tb.src_dh.end());
		out.wavetable.assign(tb.wavetable.begin(), tb.wavetable.end());
		return "";
	}
	Tables<double> tb;
	const std::string msg = design_numbers<double>(c, control_rate, k, tb);
	if (!msg.empty()) return msg;
	out.fir = tb.fir; out.src_h = tb.src_h; out.src_dh = tb.src_dh; out.wavetable = tb.wavetable;
	return "";
}

// NoiseSource::getSample + NoiseFilter::filter (vtm/NoiseSource.h:40-44, vtm/NoiseFilter.h:63-68) for steps [0, n): the
// generator is double whatever TFloat is; the one-zero low-pass adds in TFloat.
void design_noise_table(size_t n, bool as_float, void* out)
{
	// (this file is compiled with -ffp-contract=off: the product is rounded before its floor is subtracted, as in the
	// reference and in the kernel's noise_advance)
	double seed = 0.7892347; // NoiseSource.h:32-34
	if (as_float) {
		float* o = static_cast<float*>(out);
		float prev = 0.0f;
		for (size_t i = 0; i < n; ++i) {
			const double product = seed * 377.0;
			seed = product - std::floor(product);
			const float white = static_cast<float>(seed - 0.5);
			o[i] = white + prev;
			prev = white;
		}
	} else {
		double* o = static_cast<double*>(out);
		double prev = 0.0;
		for (size_t i = 0; i < n; ++i) {
			const double product = seed * 377.0;
			seed = product - std::floor(product);
			const double white = seed - 0.5;
			o[i] = white + prev;
			prev = white;
		}
	}
}

void radiation_impedance(double radius, double period, double out[6])
{
	const double transition = 0.5e-2;
	const double rr = radius < transition ? transition : radius;
	const double trans_freq = 62.3371 / rr + 320.204;
	const double cos_wt = std::cos((2.0 * kPi) * trans_freq * period);
	const double qa = 2.0 * cos_wt;
	const double qb = -2.0 * (cos_wt + 1.0);
	const double qc = cos_wt + 1.0;
	const double delta = qb * qb - 4.0 * qa * qc;
	double a = (-qb - std::sqrt(delta)) / (2.0 * qa);
	const double b = 2.0 * a - 1.0;
	if (radius < transition) a *= 40391.2 * (radius * radius);
	const double coef = 1.0 / (a + 1.0);
	const double a_plus_b = a + b;
	out[0] = a_plus_b * coef;
	out[1] = 2.0 * coef;
	out[2] = -2.0 * b * coef;
	out[3] = a_plus_b * coef;
	out[4] = (a - 1.0) * coef;
	out[5] = (b - a) * coef;
}

// VocalTractModel5: loadConfiguration (vtm/VocalTractModel5.h:375-421) and initializeSynthesizer (:455-521)
// with TFloat = double, the checks its constructors make included.
std::string design_plan5(const gvtm5_config& c, double control_rate, Design& out)
{
	std::ostringstream err;
	auto finite_pos = [](double v) { return std::isfinite(v) && v > 0.0; };
	if (!finite_pos(c.output_rate)) return "output_rate must be > 0";
	if (!finite_pos(control_rate)) return "control_rate must be > 0";
	if (c.precision != GVTM_PRECISION_F64) return "model 5 computes in fp64 only (VocalTractModel5<double,1>)";
	if (c.reserved_ != 0) return "reserved_ must be 0";
	if (c.waveform != 0 && c.waveform != 1) return "waveform must be 0 (pulse) or 1 (sine)";
	if (!(c.temperature > -273.0) || !std::isfinite(c.temperature)) return "temperature out of range";

	out.model5 = true;
	out.config5 = c;
	out.config = gvtm_config{};
	out.config.output_rate = c.output_rate;
	out.control_rate = control_rate;
	out.f32 = false;
	DeviceConstants& k = out.k;
	Model5Constants& m = out.k5;
	k = DeviceConstants{};
	m = Model5Constants{};
	k.section_delay = 1;
	k.layout = 2;
	k.waveform = c.waveform;
	k.modulation = c.noise_modulation != 0;
	m.bypass = c.bypass == 1;
	m.constant_mouth = c.constant_radius_mouth_impedance != 0;
	m.output_rate = c.output_rate;

	double length = c.vocal_tract_length_offset + c.vocal_tract_length;
	length = std::min(std::max(length, 3.0), 30.0);
	double nasal[7] = {0.0};
	for (int i = 0; i < 6; ++i) {
		nasal[i + 1] = c.nasal_radius[i] * c.global_nasal_radius_coef;
		if (!finite_pos(nasal[i + 1])) return "nasal radii must be > 0";
	}
	for (int i = 0; i < 8; ++i) k.radius_coef[i] = c.radius_coef[i] * c.global_radius_coef;

	const double speed = 331.4 + (0.6 * c.temperature);
	m.sample_rate = (speed * (30 * 1) * 100.0f) / length;
	// PoleZeroRadiationImpedance's constructor (vtm/PoleZeroRadiationImpedance.h:116-119)
	if (!(m.sample_rate >= 50000.0)) return "model 5 needs an internal rate of at least 50 kHz (vocal tract too long / too cold)";
	k.sample_rate = static_cast<int>(m.sample_rate);
	k.breathiness = c.breathiness / 100.0f;
	const double mix_amp = amplitude_60db_t<double>(c.mix_offset);
	if (!(mix_amp > 0.0)) return "mix_offset must be > 0 dB";
	k.crossmix_factor = 1.0f / mix_amp;
	k.damping = 1.0f - (c.loss_factor / 100.0f);

	// RosenbergBGlottalSource's constructor (vtm/RosenbergBGlottalSource.h:66-96)
	m.rb_tn_min = c.glottal_pulse_tn_min / 100.0f;
	m.rb_tn_max = c.glottal_pulse_tn_max / 100.0f;
	m.rb_t1 = c.glottal_pulse_tp / 100.0f;
	if (!(m.rb_t1 >= 1.0e-2) || !(m.rb_tn_min >= 1.0e-2) || !(m.rb_tn_max >= 1.0e-2) || m.rb_tn_min > m.rb_tn_max ||
			m.rb_t1 + m.rb_tn_max > 1.0) {
		return "glottal pulse shape needs tp, tn_min, tn_max >= 1 %, tn_min <= tn_max, tp + tn_max <= 100";
	}

	m.period = 1.0f / m.sample_rate;
	if (m.constant_mouth) radiation_impedance(c.mouth_impedance_radius * 1.0e-2f, m.period, m.mouth_c);
	// initializeNasalCavity (vtm/VocalTractModel5.h:593-603)
	for (int i = 1; i < 6; ++i) m.nasal_k[i] = junction<double>(nasal[i], nasal[i + 1]);
	radiation_impedance(std::sqrt(0.5f * nasal[6] * nasal[6]) * 1.0e-2f, m.period, m.nose_c);
	m.nasal_r1_sq = nasal[1] * nasal[1];

	// Butterworth filters (their update() range checks included)
	auto butter_ok = [&](double cutoff) { return cutoff >= 1.0 && cutoff <= m.sample_rate * 0.48; };
	if (!butter_ok(c.glottal_noise_cutoff) || !butter_ok(c.frication_noise_cutoff) || !butter_ok(c.glottal_lowpass_cutoff)) {
		return "Butterworth cutoffs must lie between 1 Hz and 0.48 of the internal rate";
	}
	auto butter1 = [&](double cutoff, double& b0, double& a1) {
		const double wcT = 2.0 * std::tan(kPi * cutoff / m.sample_rate);
		const double c1 = 1.0 / (wcT + 2.0);
		b0 = c1 * wcT;
		a1 = c1 * (wcT - 2.0);
	};
	butter1(c.glottal_noise_cutoff, m.gn_b0, m.gn_a1);
	butter1(c.glottal_lowpass_cutoff, m.gp_b0, m.gp_a1);
	{
		const double wcT = 2.0 * std::tan(kPi * c.frication_noise_cutoff / m.sample_rate);
		const double wc2T2 = wcT * wcT;
		const double c1 = 2.0 * std::sqrt(2.0) * wcT;
		const double c2 = 1.0 / (wc2T2 + c1 + 4.0);
		m.fn_b0 = c2 * wc2T2;
		m.fn_b1 = 2.0 * m.fn_b0;
		m.fn_a1 = c2 * (2.0 * wc2T2 - 8.0);
		m.fn_a2 = c2 * (wc2T2 - c1 + 4.0);
	}
	m.frication_factor = c.frication_factor;
	m.min_loss = c.min_glottal_loss / 100.0f;
	m.max_loss = c.max_glottal_loss / 100.0f;
	k.bp_T = 1.0 / m.sample_rate; // BandpassFilter::update (BandpassFilter.h:104)

	// Controller::synthesize (vtm_control_model/Controller.cpp:286-287)
	k.control_steps = static_cast<unsigned>(std::rint(m.sample_rate / control_rate));
	if (k.control_steps == 0) return "control_rate above the internal sample rate";
	k.interp_coef = 1.0f / k.control_steps;

	// SampleRateConverter::initializeConversion (vtm/SampleRateConverter.h:136-164)
	const double ratio = c.output_rate / m.sample_rate;
	k.src_ratio = ratio;
	k.time_inc = static_cast<unsigned>(std::rint(std::pow(2.0, 16) / ratio));
	if (k.time_inc == 0) return "output_rate too high for the 16.16 time register";
	const double rounded_ratio = std::pow(2.0, 16) / k.time_inc;
	k.upsampling = ratio >= 1.0;
	if (k.upsampling) {
		k.phase_inc = 0;
		k.pad = kSrcZeroCrossings;
	} else {
		k.phase_inc = static_cast<unsigned>(std::rint(ratio * 65536));
		k.pad = static_cast<int>(kSrcZeroCrossings / rounded_ratio) + 1;
	}
	if (k.pad > kMaxPad || (k.phase_inc == 0 && !k.upsampling)) {
		err << "output_rate / internal rate = " << ratio << " is below the supported down-sampling range";
		return err.str();
	}
	// the model 5 kernel parks converted samples in a 512-entry ring until the difference filter has emitted them in
	// aligned blocks: two chunks of 60 steps must fit beside the held-back block
	if (ratio > 3.0) return "output_rate above 3x the internal rate is not supported by the model 5 path";
	design_src_filter<double>(out.src_h, out.src_dh);
	out.fir.clear();
	out.wavetable.clear();
	return "";
}

// --- parameter-track generation (vtm_tracks.hip) -----------------------------------------------------

const char* design_tracks(const gvtm_track_config& c, TrackConstants& k)
{
	if (c.control_period_ms < 1 || c.control_period_ms > 1000) return "control_period_ms out of range";
	if (c.reserved_ != 0) return "reserved_ must be 0";
	k = TrackConstants{};
	k.control_period = c.control_period_ms;
	k.macro_intonation = c.macro_intonation != 0;
	k.micro_intonation = c.micro_intonation != 0;
	k.intonation_drift = c.intonation_drift != 0;
	k.smooth_intonation = c.smooth_intonation != 0;
	k.initial_pitch = c.initial_pitch;
	k.mean_pitch = c.mean_pitch;
	if (k.intonation_drift) {
		// DriftGenerator::setUp (DriftGenerator.cpp:49-56) and Butterworth2LowPassFilter<double>::update
		// (vtm/Butterworth2LowpassFilter.h:88-107, including its range check)
		if (!(c.drift_sample_rate > 0.0) || c.drift_lowpass_cutoff < 1.0 || c.drift_lowpass_cutoff > c.drift_sample_rate * 0.48) {
			return "drift_lowpass_cutoff must lie between 1 Hz and 0.48 of drift_sample_rate";
		}
		k.pitch_deviation = c.drift_deviation * 2.0;
		k.pitch_offset = c.drift_deviation;
		const double wcT = 2.0 * std::tan(kPi * c.drift_lowpass_cutoff / c.drift_sample_rate);
		const double wc2T2 = wcT * wcT;
		const double c1 = 2.0 * std::sqrt(2.0) * wcT;
		const double c2 = 1.0 / (wc2T2 + c1 + 4.0);
		k.b0 = c2 * wc2T2;
		k.b1 = 2.0 * k.b0;
		k.a1 = c2 * (2.0 * wc2T2 - 8.0);
		k.a2 = c2 * (wc2T2 - c1 + 4.0);
	}
	return "";
}

size_t tracks_frame_count(int control_period, const gvtm_event* events, size_t n_events)
{
	// the control-period loop of EventList::generateOutput (EventList.cpp:985-1032) without the arithmetic
	if (n_events < 2) return 0;
	size_t target = 1, n = 0;
	long long now = 0;
	while (target < n_events) {
		++n;
		now += control_period;
		if (now >= events[target].time_ms) {
			if (++target == n_events) break;
		}
	}
	return n;
}

} // namespace gvtm
