// Design-time math for the batched vocal-tract model (host, fp64).
//
// Each routine states which reference routine defines the numbers it must
// reproduce; tests/test_design_tables.py compares every table with the oracle
// bit for bit through gvtm_plan_table().
#include "vtm_design.hpp"

#include <algorithm>
#include <cmath>
#include <sstream>

namespace gvtm {

namespace {

constexpr double kPi = 3.14159265358979323846;

// --- glottal-source anti-alias FIR -------------------------------------------------
// Numbers defined by WavetableGlottalSourceFIRFilter::maximallyFlat / rationalApproximation /
// trim (vtm/WavetableGlottalSourceFIRFilter.h:137-215, :316-361, :227-235) with
// beta = 0.2, gamma = 0.1, cutoff = 1e-8 (vtm/WavetableGlottalSource.h:94-96).
struct Rational { int numerator, denominator, order; };

Rational best_rational(double value, int order)
{
	constexpr int kLimit = 200;
	Rational r{0, 0, -1};
	if (order <= 0) return r;
	const double frac = std::fabs(value - static_cast<int>(value));
	const int max_den = std::min(2 * order, kLimit);
	double best = 1.0;
	int best_num = 0;
	for (int den = order; den <= max_den; ++den) {
		const double scaled = den * frac;
		const int nearest = static_cast<int>(scaled + 0.5);
		const double err = std::fabs((scaled - static_cast<double>(nearest)) / den);
		if (err < best) {
			best = err;
			best_num = nearest;
			r.denominator = den;
		}
	}
	r.numerator = static_cast<int>(std::fabs(value)) * r.denominator + best_num;
	if (value < 0.0) r.numerator = -r.numerator;
	r.order = r.denominator - 1;
	if (r.numerator == r.denominator) {
		r.denominator = max_den;
		r.order = r.numerator = r.denominator - 1;
	}
	return r;
}

std::vector<double> design_glottal_fir()
{
	constexpr double beta = 0.2, gamma = 0.1, cutoff = 0.00000001;
	constexpr int kLimit = 200;
	const int nt0 = static_cast<int>(1.0 / (4.0 * gamma * gamma));
	const double ac = (1.0 + std::cos((2.0 * kPi) * beta)) / 2.0;
	const Rational q = best_rational(ac, nt0);
	const int np = q.denominator;
	const int nt = q.order;
	const int numer = q.numerator == 0 ? 1 : q.numerator;
	const int n = 2 * np - 1;

	std::vector<double> mag(kLimit + 2, 0.0), cosv(kLimit + 2, 0.0), half(kLimit + 2, 0.0);
	mag[1] = cosv[1] = 1.0;
	const int terms = nt - numer;
	for (int i = 2; i <= np; ++i) {
		cosv[i] = std::cos((2.0 * kPi) * (static_cast<double>(i - 1) / n));
		const double x = (1.0 - cosv[i]) / 2.0;
		if (numer == nt) continue;
		double y = x, sum = 1.0;
		for (int j = 1; j <= terms; ++j) {
			double z = y;
			for (int jj = 1; jj <= numer - 1; ++jj) z *= 1.0 + (static_cast<double>(j) / jj);
			y *= x;
			sum += z;
		}
		mag[i] = sum * std::pow(1.0 - x, numer);
	}
	// N-point inverse DFT of the symmetric magnitude response
	for (int i = 1; i <= np; ++i) {
		double acc = mag[1] / 2.0;
		for (int j = 2; j <= np; ++j) {
			int m = ((i - 1) * (j - 1)) % n;
			if (m > nt) m = n - m;
			acc += cosv[m + 1] * mag[j];
		}
		half[i] = acc * (2.0 / static_cast<double>(n));
	}
	int keep = np;
	for (int i = np; i > 0; --i) {
		if (std::fabs(half[i]) >= std::fabs(cutoff)) { keep = i; break; }
	}
	// mirror the half response into a linear-phase filter: h[keep] ... h[1] ... h[keep]
	std::vector<double> taps;
	taps.reserve(2 * keep - 1);
	for (int i = keep; i >= 1; --i) taps.push_back(half[i]);
	for (int i = 2; i <= keep; ++i) taps.push_back(half[i]);
	return taps;
}

// --- sample-rate-converter prototype ------------------------------------------------
// Kaiser-windowed sinc of SampleRateConverter::initializeFilter / Izero
// (vtm/SampleRateConverter.h:230-255, :175-194).
double bessel_i0(double x)
{
	double sum = 1.0, term = 1.0;
	const double half = x / 2.0;
	int n = 1;
	do {
		double t = half / n;
		n += 1;
		t *= t;
		term *= t;
		sum += term;
	} while (term >= 1E-21 * sum);
	return sum;
}

void design_src_filter(std::vector<double>& h, std::vector<double>& dh)
{
	const double beta = 5.658;
	const double cutoff = 11.0 / 13.0;
	h.assign(kSrcFilterLength, 0.0);
	dh.assign(kSrcFilterLength, 0.0);
	h[0] = cutoff;
	const double dx = kPi / kSrcPhases;
	for (unsigned i = 1; i < static_cast<unsigned>(kSrcFilterLength); ++i) {
		const double y = i * dx;
		h[i] = std::sin(y * cutoff) / y;
	}
	const double inv_i0 = 1.0 / bessel_i0(beta);
	for (unsigned i = 0; i < static_cast<unsigned>(kSrcFilterLength); ++i) {
		const double t = static_cast<double>(i) / kSrcFilterLength;
		h[i] *= bessel_i0(beta * std::sqrt(1.0 - (t * t))) * inv_i0;
	}
	for (int i = 0; i + 1 < kSrcFilterLength; ++i) dh[i] = h[i + 1] - h[i];
	dh[kSrcFilterLength - 1] = 0.0 - h[kSrcFilterLength - 1];
}

// --- glottal wavetable ----------------------------------------------------------------
// WavetableGlottalSource constructor (vtm/WavetableGlottalSource.h:90-141).
void design_wavetable(const gvtm_config& c, DeviceConstants& k, std::vector<double>& table)
{
	const unsigned len = kWavetableLength;
	k.table_div1 = static_cast<unsigned>(std::rint(len * (c.glottal_pulse_tp / 100.0)));
	k.table_div2 = static_cast<unsigned>(std::rint(len * ((c.glottal_pulse_tp + c.glottal_pulse_tn_max) / 100.0)));
	k.tn_delta = std::rint(len * ((c.glottal_pulse_tn_max - c.glottal_pulse_tn_min) / 100.0));
	k.basic_increment = len / static_cast<double>(k.sample_rate);
	table.assign(len, 0.0);
	if (c.waveform == 0) {
		const double fall = k.table_div2 - k.table_div1;
		for (unsigned i = 0; i < k.table_div1; ++i) {
			const double x = static_cast<double>(i) / k.table_div1;
			const double x2 = x * x;
			table[i] = (3.0 * x2) - (2.0 * (x2 * x));
		}
		for (unsigned i = k.table_div1, j = 0; i < k.table_div2 && i < len; ++i, ++j) {
			const double x = static_cast<double>(j) / fall;
			table[i] = 1.0 - (x * x);
		}
	} else {
		for (unsigned i = 0; i < len; ++i) {
			table[i] = std::sin((static_cast<double>(i) / len) * 2.0 * kPi);
		}
	}
}

double junction(double left_radius, double right_radius)
{
	const double a = left_radius * left_radius;
	const double b = right_radius * right_radius;
	return (a - b) / (a + b);
}

} // namespace

double amplitude_60db(double db)
{
	if (db <= 0.0) return 0.0;
	if (db == 60.0) return 1.0;
	return std::pow(10.0, (db - 60.0) * (1.0 / 20.0));
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
	if (c.precision != GVTM_PRECISION_F64 && c.precision != GVTM_PRECISION_MIXED) return "unknown precision";
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

	// loadConfiguration (vtm/VocalTractModel0.h:266-305): length clamp, radius scaling
	double length = c.vocal_tract_length_offset + c.vocal_tract_length;
	length = std::min(std::max(length, 3.0), 30.0);
	const double aperture = c.aperture_radius * c.global_radius_coef;
	double nasal[6] = {0.0};
	for (int i = 0; i < 5; ++i) nasal[i + 1] = c.nasal_radius[i] * c.global_nasal_radius_coef;
	for (int i = 0; i < 8; ++i) k.radius_coef[i] = c.radius_coef[i] * c.global_radius_coef;
	for (int i = 1; i < 6; ++i) {
		if (!finite_pos(nasal[i])) return "nasal radii must be > 0";
	}
	if (!finite_pos(aperture)) return "aperture_radius must be > 0";

	// initializeSynthesizer (vtm/VocalTractModel0.h:338-392, VocalTractModel2.h:413-467)
	const double speed = 331.4 + (0.6 * c.temperature);
	const int sections = c.tube_layout == GVTM_TUBE_30_18 ? 30 : 10; // TOTAL_SECTIONS (VocalTractModel4.h:194 / VocalTractModel0.h:128)
	k.sample_rate = static_cast<int>((speed * (sections * c.section_delay) * 100.0) / length);
	if (k.sample_rate < 2000) return "internal sample rate too low";
	const double nyquist = static_cast<float>(k.sample_rate) / 2.0f; // float arithmetic, as in the reference
	k.breathiness = c.breathiness / 100.0;
	const double mix_amp = amplitude_60db(c.mix_offset);
	if (!(mix_amp > 0.0)) return "mix_offset must be > 0 dB";
	k.crossmix_factor = 1.0 / mix_amp;
	k.damping = 1.0 - (c.loss_factor / 100.0);

	const double mouth_ap = (nyquist - c.mouth_coefficient) / nyquist;
	k.mouth_b0_refl = 1.0 - std::fabs(mouth_ap); // ReflectionFilter.h:55-60
	k.mouth_a1_refl = -mouth_ap;
	k.mouth_a_rad = mouth_ap;                    // RadiationFilter.h:54-61: b0 = a, b1 = a1 = -a
	const double nose_ap = (nyquist - c.nose_coefficient) / nyquist;
	k.nose_b0_refl = 1.0 - std::fabs(nose_ap);
	k.nose_a1_refl = -nose_ap;
	k.nose_a_rad = nose_ap;

	// initializeNasalCavity (vtm/VocalTractModel0.h:457-470)
	k.nasal_k[0] = 0.0;
	for (int i = 1; i < 5; ++i) k.nasal_k[i] = junction(nasal[i], nasal[i + 1]);
	k.nasal_k[5] = junction(nasal[5], aperture);
	k.aperture_radius2 = aperture * aperture;
	k.nasal_r2_sq = nasal[1] * nasal[1];

	// Throat (vtm/Throat.h:52-60)
	k.throat_b0 = (c.throat_cutoff * 2.0) / static_cast<double>(k.sample_rate);
	k.throat_a1 = k.throat_b0 - 1.0;
	k.throat_gain = amplitude_60db(c.throat_volume);
	k.bp_T = 1.0 / static_cast<double>(k.sample_rate);

	// Controller::synthesize (vtm_control_model/Controller.cpp:286-287)
	k.control_steps = static_cast<unsigned>(std::rint(static_cast<double>(k.sample_rate) / control_rate));
	if (k.control_steps == 0) return "control_rate above the internal sample rate";
	k.interp_coef = 1.0f / k.control_steps;

	// SampleRateConverter::initializeConversion (vtm/SampleRateConverter.h:136-164)
	k.src_ratio = c.output_rate / static_cast<double>(k.sample_rate);
	k.time_inc = static_cast<unsigned>(std::rint(std::pow(2.0, 16) / k.src_ratio));
	if (k.time_inc == 0) return "output_rate too high for the 16.16 time register";
	const double rounded_ratio = std::pow(2.0, 16) / k.time_inc;
	k.upsampling = k.src_ratio >= 1.0;
	if (k.upsampling) {
		k.phase_inc = 0;
		k.pad = kSrcZeroCrossings;
	} else {
		k.phase_inc = static_cast<unsigned>(std::rint(k.src_ratio * 65536));
		k.pad = static_cast<int>(kSrcZeroCrossings / rounded_ratio) + 1;
	}
	if (k.pad > kMaxPad || (k.phase_inc == 0 && !k.upsampling)) {
		err << "output_rate / internal rate = " << k.src_ratio << " is below the supported down-sampling range";
		return err.str();
	}

	design_wavetable(c, k, out.wavetable);
	out.fir = design_glottal_fir();
	k.fir_taps = static_cast<int>(out.fir.size());
	if (k.fir_taps > kMaxFirTaps || k.fir_taps > 49) return "glottal FIR longer than the device pre-roll (49 taps)";
	design_src_filter(out.src_h, out.src_dh);
	return "";
}

bool output_count_for_steps(const DeviceConstants& k, uint64_t steps, uint64_t& n_out)
{
	// An output sample k is emitted while its integer read position
	// P_k = floor(k * time_inc / 2^16) lies before the end pointer, and the final end pointer
	// (after flushBuffer()'s 2*pad zero fills, SampleRateConverter.h:462-471) is steps + 2*pad.
	const uint64_t fills = steps + 2ull * static_cast<uint64_t>(k.pad);
	n_out = ((fills << 16) + k.time_inc - 1) / k.time_inc;
	if (k.upsampling) return true;
	// Down-sampling: the read position advances by more than one input sample per output, so
	// an automatic dataEmpty() (every fill_size fills) can leave emptyPtr beyond its end
	// pointer.  If fewer fills than that overshoot follow before flushBuffer()'s explicit
	// dataEmpty(), it sees endPtr < emptyPtr, adds the ring size and converts ~1024 stale ring
	// samples (SampleRateConverter.h:298-308).  That output depends on ring leftovers; it is
	// detected here and refused rather than reproduced.
	const uint64_t fill_size = static_cast<uint64_t>(kSrcRing - 2 * k.pad);
	const uint64_t last_auto_end = (fills / fill_size) * fill_size;
	if (last_auto_end == 0) return true;
	const uint64_t k_star = ((last_auto_end << 16) + k.time_inc - 1) / k.time_inc; // first output at or past it
	const uint64_t p_star = (k_star * static_cast<uint64_t>(k.time_inc)) >> 16;
	return p_star <= fills;
}

} // namespace gvtm
