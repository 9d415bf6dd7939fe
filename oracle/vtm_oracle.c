/*
 * TEST INFRASTRUCTURE — NOT PART OF THE PRODUCT.  See vtm_oracle.h.
 *
 * Scalar fp64 restatement of the GamaTTS vocal-tract-model hot path.  Written
 * from the algorithm, as stage objects over plain structs; every function
 * names the reference lines whose arithmetic (including evaluation order, so
 * that the result is bit-identical when compiled with -ffp-contract=off) it
 * follows.  Paths are relative to /root/reference/gama_tts/src/.
 */
#include "vtm_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------------- */
/* vtm/VTMUtil.h                                                              */

/* Util::amplitude60dB, vtm/VTMUtil.h:48-67 */
static double amplitude60dB(double db)
{
	if (db <= 0.0) return 0.0;
	if (db == 60.0) return 1.0;
	db -= 60.0;
	return pow(10.0, db * (1.0 / 20.0));
}

/* Util::frequency, vtm/VTMUtil.h:74-84 */
static double frequency(double pitch)
{
	return 220.0 * pow(2.0, (pitch + 3.0) * (1.0 / 12.0));
}

/* Util::calculateOutputScale, vtm/VTMUtil.cpp:48-67 + maximumAbsoluteValue, VTMUtil.h:115-127 */
float vtmo_output_scale(const float* x, size_t n)
{
	float maxv = 0.0f;
	for (size_t i = 0; i < n; ++i) {
		const float a = fabsf(x[i]);
		if (a > maxv) maxv = a;
	}
	if (maxv < 1.0e-30f) return 0.0f;
	return 0.95f / maxv;
}

/* ------------------------------------------------------------------------- */
/* vtm/WavetableGlottalSourceFIRFilter.h                                      */

#define FIR_LIMIT 200

/* rationalApproximation, WavetableGlottalSourceFIRFilter.h:316-361 */
static void rational_approximation(double number, int* order, int* numerator, int* denominator)
{
	if (*order <= 0) {
		*numerator = 0;
		*denominator = 0;
		*order = -1;
		return;
	}
	const double fractional = fabs(number - (int) number);
	int order_max = 2 * (*order);
	if (order_max > FIR_LIMIT) order_max = FIR_LIMIT;

	double min_error = 1.0;
	int modulus = 0;
	for (int i = *order; i <= order_max; ++i) {
		const double ps = i * fractional;
		const int ip = (int) (ps + 0.5);
		const double error = fabs((ps - (double) ip) / i);
		if (error < min_error) {
			min_error = error;
			modulus = ip;
			*denominator = i;
		}
	}
	*numerator = (int) fabs(number) * (*denominator) + modulus;
	if (number < 0.0) *numerator *= -1;
	*order = *denominator - 1;
	if (*numerator == *denominator) {
		*denominator = order_max;
		*order = *numerator = *denominator - 1;
	}
}

/* maximallyFlat, WavetableGlottalSourceFIRFilter.h:137-215 (beta/gamma range
 * checks omitted: the caller passes the fixed 0.2 / 0.1 of WavetableGlottalSource.h:94-95) */
static void maximally_flat(double beta, double gamma, int* np, double* coefficient)
{
	double a[FIR_LIMIT + 1], c[FIR_LIMIT + 1];
	int numerator;

	memset(a, 0, sizeof(a));
	memset(c, 0, sizeof(c));
	*np = 0;
	int nt = (int) (1.0 / (4.0 * gamma * gamma));
	const double ac = (1.0 + cos((2.0 * M_PI) * beta)) / 2.0;
	rational_approximation(ac, &nt, &numerator, np);

	const int n = (2 * (*np)) - 1;
	if (numerator == 0) numerator = 1;

	c[1] = a[1] = 1.0;
	const int ll = nt - numerator;
	for (int i = 2; i <= *np; ++i) {
		c[i] = cos((2.0 * M_PI) * ((double) (i - 1) / n));
		const double x = (1.0 - c[i]) / 2.0;
		double y = x;
		if (numerator == nt) continue;
		double sum = 1.0;
		for (int j = 1; j <= ll; ++j) {
			double z = y;
			if (numerator != 1) {
				for (int jj = 1; jj <= (numerator - 1); ++jj) {
					z *= 1.0 + ((double) j / jj);
				}
			}
			y *= x;
			sum += z;
		}
		a[i] = sum * pow((1.0 - x), numerator);
	}
	for (int i = 1; i <= *np; ++i) {
		coefficient[i] = a[1] / 2.0;
		for (int j = 2; j <= *np; ++j) {
			int m = ((i - 1) * (j - 1)) % n;
			if (m > nt) m = n - m;
			coefficient[i] += c[m + 1] * a[j];
		}
		coefficient[i] *= 2.0 / (double) n;
	}
}

/* ctor, WavetableGlottalSourceFIRFilter.h:74-114 with trim (:227-235);
 * beta 0.2, gamma 0.1, cutoff 1e-8 from WavetableGlottalSource.h:94-96 */
int vtmo_fir_coefficients(double* coef)
{
	int nc;
	double coefficient[FIR_LIMIT + 1];
	maximally_flat(0.2, 0.1, &nc, coefficient);
	for (int i = nc; i > 0; --i) {
		if (fabs(coefficient[i]) >= fabs(0.00000001)) {
			nc = i;
			break;
		}
	}
	const int taps = (nc * 2) - 1;
	int increment = -1;
	int pointer = nc;
	for (int i = 0; i < taps; ++i) {
		coef[i] = coefficient[pointer];
		pointer += increment;
		if (pointer <= 0) {
			pointer = 2;
			increment = 1;
		}
	}
	return taps;
}

/* ------------------------------------------------------------------------- */
/* vtm/WavetableGlottalSource.h                                               */

typedef struct {
	unsigned div1, div2;
	double tn_length, tn_delta, basic_increment, position, prev_amplitude;
	double table[VTMO_WAVETABLE_LENGTH];
	int taps, ptr;
	double coef[VTMO_FIR_MAX_TAPS];
	double data[VTMO_FIR_MAX_TAPS];
	double* tap_pos; /* optional debug tap */
} glottal_source;

/* ctor, WavetableGlottalSource.h:90-141 */
static void glottal_init(glottal_source* g, int waveform, double sample_rate, double tp, double tn_min, double tn_max)
{
	const unsigned len = VTMO_WAVETABLE_LENGTH;
	g->div1 = (unsigned) rint(len * (tp / 100.0));
	g->div2 = (unsigned) rint(len * ((tp + tn_max) / 100.0));
	g->tn_length = g->div2 - g->div1;
	g->tn_delta = rint(len * ((tn_max - tn_min) / 100.0));
	g->basic_increment = len / sample_rate;
	g->position = 0.0;
	g->prev_amplitude = -1.0;
	if (waveform == 0) {
		for (unsigned i = 0; i < g->div1; ++i) {
			const double x = (double) i / g->div1;
			const double x2 = x * x;
			const double x3 = x2 * x;
			g->table[i] = (3.0 * x2) - (2.0 * x3);
		}
		for (unsigned i = g->div1, j = 0; i < g->div2; ++i, ++j) {
			const double x = (double) j / g->tn_length;
			g->table[i] = 1.0 - (x * x);
		}
		for (unsigned i = g->div2; i < len; ++i) g->table[i] = 0.0;
	} else {
		for (unsigned i = 0; i < len; ++i) {
			g->table[i] = sin(((double) i / len) * 2.0 * M_PI);
		}
	}
	g->taps = vtmo_fir_coefficients(g->coef);
	memset(g->data, 0, sizeof(g->data));
	g->ptr = 0;
}

void vtmo_wavetable(const vtmo_config* cfg, int sample_rate, double* table)
{
	glottal_source g;
	glottal_init(&g, cfg->waveform, sample_rate, cfg->glottal_pulse_tp, cfg->glottal_pulse_tn_min, cfg->glottal_pulse_tn_max);
	memcpy(table, g.table, sizeof(g.table));
}

/* setup, WavetableGlottalSource.h:162-184 */
static void glottal_setup(glottal_source* g, double amplitude)
{
	if (g->tn_delta == 0.0 || amplitude == g->prev_amplitude) return;
	g->prev_amplitude = amplitude;
	double new_div2 = g->div2 - rint(amplitude * g->tn_delta);
	if (new_div2 < 0.0) new_div2 = 0.0;
	const double inv = 1.0 / (new_div2 - g->div1);
	double x = 0.0;
	for (unsigned i = g->div1, end = (unsigned) new_div2; i < end; ++i, x += inv) {
		g->table[i] = 1.0 - (x * x);
	}
	for (unsigned i = (unsigned) new_div2; i < g->div2; ++i) g->table[i] = 0.0;
}

/* mod0, WavetableGlottalSource.h:265-272 */
static double mod0(double v)
{
	if (v > (double) (VTMO_WAVETABLE_LENGTH - 1)) v -= (double) VTMO_WAVETABLE_LENGTH;
	return v;
}

/* filter, WavetableGlottalSourceFIRFilter.h:276-304 */
static double fir_filter(glottal_source* g, double input, int need_output)
{
	if (need_output) {
		double output = 0.0;
		g->data[g->ptr] = input;
		for (int i = 0; i < g->taps; ++i) {
			output += g->data[g->ptr] * g->coef[i];
			if (++g->ptr >= g->taps) g->ptr = 0;
		}
		if (--g->ptr < 0) g->ptr = g->taps - 1;
		return output;
	}
	g->data[g->ptr] = input;
	if (--g->ptr < 0) g->ptr = g->taps - 1;
	return 0.0;
}

/* getSample (2x oversampling oscillator), WavetableGlottalSource.h:212-235.
 * The cast of a position in (-1, 0) to unsigned goes through a signed
 * truncation (what x86-64 g++ emits for double -> unsigned int), i.e. 0. */
static double glottal_sample(glottal_source* g, double freq)
{
	double output = 0.0;
	for (int i = 0; i < 2; ++i) {
		g->position = mod0(g->position + ((freq / 2.0) * g->basic_increment));
		if (g->tap_pos) g->tap_pos[i] = g->position;
		const unsigned lower = (unsigned) (long long) g->position;
		const unsigned upper = (unsigned) (long long) mod0((double) (lower + 1));
		const double v = g->table[lower] + ((g->position - lower) * (g->table[upper] - g->table[lower]));
		output = fir_filter(g, v, i);
	}
	return output;
}

/* ------------------------------------------------------------------------- */
/* vtm/SampleRateConverter.h                                                  */

enum {
	SRC_BUFFER_SIZE = 1024,
	SRC_L_BITS = 8,
	SRC_L_RANGE = 256,
	SRC_M_BITS = 8,
	SRC_M_RANGE = 256,
	SRC_ZERO_CROSSINGS = 13,
	SRC_FILTER_LENGTH = VTMO_SRC_FILTER_LENGTH,
	SRC_FRACTION_BITS = 16,
	SRC_FRACTION_RANGE = 65536,
	SRC_FILTER_LIMIT = VTMO_SRC_FILTER_LENGTH - 1
};
#define SRC_M_MASK 0x000000FFu
#define SRC_L_MASK 0x0000FF00u
#define SRC_FRACTION_MASK 0x0000FFFFu
#define SRC_N_MASK 0xFFFF0000u

typedef struct {
	double ratio;
	int fill_ptr, empty_ptr, pad_size, fill_size;
	unsigned time_register_increment, filter_increment, phase_increment, time_register;
	int fill_counter;
	int count_only;
	double h[SRC_FILTER_LENGTH], delta_h[SRC_FILTER_LENGTH], buffer[SRC_BUFFER_SIZE];
	float* out;
	size_t out_cap, out_n;
} src_state;

/* Izero, SampleRateConverter.h:175-194 */
static double izero(double x)
{
	double sum, u, halfx, temp;
	int n;
	sum = u = n = 1;
	halfx = x / 2.0;
	do {
		temp = halfx / n;
		n += 1;
		temp *= temp;
		u *= temp;
		sum += u;
	} while (u >= 1E-21 * sum);
	return sum;
}

/* initializeFilter, SampleRateConverter.h:230-255 */
void vtmo_src_filter(double* h, double* delta_h)
{
	const double beta = 5.658;
	const double lp_cutoff = 11.0 / 13.0;
	h[0] = lp_cutoff;
	const double x = M_PI / SRC_L_RANGE;
	for (unsigned i = 1; i < SRC_FILTER_LENGTH; ++i) {
		const double y = i * x;
		h[i] = sin(y * lp_cutoff) / y;
	}
	const double ibeta = 1.0 / izero(beta);
	for (unsigned i = 0; i < SRC_FILTER_LENGTH; ++i) {
		const double temp = (double) i / SRC_FILTER_LENGTH;
		h[i] *= izero(beta * sqrt(1.0 - (temp * temp))) * ibeta;
	}
	for (unsigned i = 0; i < SRC_FILTER_LIMIT; ++i) delta_h[i] = h[i + 1] - h[i];
	delta_h[SRC_FILTER_LIMIT] = 0.0 - h[SRC_FILTER_LIMIT];
}

/* initializeConversion + initializeBuffer + reset, SampleRateConverter.h:136-164, :206-218, :119-125 */
static void src_init(src_state* s, double input_rate, double output_rate, int count_only)
{
	memset(s, 0, sizeof(*s));
	s->count_only = count_only;
	if (!count_only) vtmo_src_filter(s->h, s->delta_h);
	s->ratio = output_rate / input_rate;
	s->time_register_increment = (unsigned) rint(pow(2.0, SRC_FRACTION_BITS) / s->ratio);
	const double rounded_ratio = pow(2.0, SRC_FRACTION_BITS) / s->time_register_increment;
	if (s->ratio >= 1.0) {
		s->filter_increment = SRC_L_RANGE;
	} else {
		s->phase_increment = (unsigned) rint(s->ratio * SRC_FRACTION_RANGE);
	}
	s->pad_size = (s->ratio >= 1.0) ? (int) SRC_ZERO_CROSSINGS : (int) (SRC_ZERO_CROSSINGS / rounded_ratio) + 1;
	s->fill_ptr = s->pad_size;
	s->fill_size = SRC_BUFFER_SIZE - (2 * s->pad_size);
}

static void src_emit(src_state* s, double v)
{
	if (s->out_n < s->out_cap) s->out[s->out_n] = (float) v;
	s->out_n++;
}

static void sr_inc(int* p) { if (++(*p) >= SRC_BUFFER_SIZE) (*p) -= SRC_BUFFER_SIZE; }
static void sr_dec(int* p) { if (--(*p) < 0) (*p) += SRC_BUFFER_SIZE; }

/* dataEmpty, SampleRateConverter.h:295-416 */
static void src_data_empty(src_state* s)
{
	int end_ptr = s->fill_ptr - s->pad_size;
	if (end_ptr < 0) end_ptr += SRC_BUFFER_SIZE;
	if (end_ptr < s->empty_ptr) end_ptr += SRC_BUFFER_SIZE;

	if (s->ratio >= 1.0) {
		while (s->empty_ptr < end_ptr) {
			double output = 0.0;
			if (!s->count_only) {
				double interpolation = (double) (s->time_register & SRC_M_MASK) / SRC_M_RANGE;
				int index = s->empty_ptr;
				for (unsigned fi = (s->time_register & SRC_L_MASK) >> SRC_M_BITS; fi < SRC_FILTER_LENGTH;
						sr_dec(&index), fi += s->filter_increment) {
					output += (s->buffer[index] * (s->h[fi] + (s->delta_h[fi] * interpolation)));
				}
				s->time_register = ~s->time_register;
				interpolation = (double) (s->time_register & SRC_M_MASK) / SRC_M_RANGE;
				index = s->empty_ptr;
				sr_inc(&index);
				for (unsigned fi = (s->time_register & SRC_L_MASK) >> SRC_M_BITS; fi < SRC_FILTER_LENGTH;
						sr_inc(&index), fi += s->filter_increment) {
					output += (s->buffer[index] * (s->h[fi] + (s->delta_h[fi] * interpolation)));
				}
				s->time_register = ~s->time_register;
			}
			src_emit(s, output);
			s->time_register += s->time_register_increment;
			s->empty_ptr += (int) ((s->time_register & SRC_N_MASK) >> SRC_FRACTION_BITS);
			if (s->empty_ptr >= SRC_BUFFER_SIZE) {
				s->empty_ptr -= SRC_BUFFER_SIZE;
				end_ptr -= SRC_BUFFER_SIZE;
			}
			s->time_register &= (~SRC_N_MASK);
		}
	} else {
		while (s->empty_ptr < end_ptr) {
			double output = 0.0;
			if (!s->count_only) {
				unsigned phase_index = (unsigned) rint((s->time_register & SRC_FRACTION_MASK) * s->ratio);
				int index = s->empty_ptr;
				unsigned impulse_index;
				while ((impulse_index = (phase_index >> SRC_M_BITS)) < SRC_FILTER_LENGTH) {
					const double impulse = s->h[impulse_index] + (s->delta_h[impulse_index] *
							((double) (phase_index & SRC_M_MASK) / SRC_M_RANGE));
					output += s->buffer[index] * impulse;
					sr_dec(&index);
					phase_index += s->phase_increment;
				}
				phase_index = (unsigned) rint((double) ((~s->time_register) & SRC_FRACTION_MASK) * s->ratio);
				index = s->empty_ptr;
				sr_inc(&index);
				while ((impulse_index = (phase_index >> SRC_M_BITS)) < SRC_FILTER_LENGTH) {
					const double impulse = s->h[impulse_index] + (s->delta_h[impulse_index] *
							((double) (phase_index & SRC_M_MASK) / SRC_M_RANGE));
					output += s->buffer[index] * impulse;
					sr_inc(&index);
					phase_index += s->phase_increment;
				}
			}
			src_emit(s, output);
			s->time_register += s->time_register_increment;
			s->empty_ptr += (int) ((s->time_register & SRC_N_MASK) >> SRC_FRACTION_BITS);
			if (s->empty_ptr >= SRC_BUFFER_SIZE) {
				s->empty_ptr -= SRC_BUFFER_SIZE;
				end_ptr -= SRC_BUFFER_SIZE;
			}
			s->time_register &= (~SRC_N_MASK);
		}
	}
}

/* dataFill, SampleRateConverter.h:268-282 */
static void src_data_fill(src_state* s, double v)
{
	s->buffer[s->fill_ptr] = v;
	sr_inc(&s->fill_ptr);
	if (++s->fill_counter >= s->fill_size) {
		src_data_empty(s);
		s->fill_counter = 0;
	}
}

/* flushBuffer, SampleRateConverter.h:462-471 */
static void src_flush(src_state* s)
{
	for (int i = 0; i < s->pad_size * 2; ++i) src_data_fill(s, 0.0);
	src_data_empty(s);
}

/* ------------------------------------------------------------------------- */
/* one-pole / two-pole helpers                                                */

typedef struct { double b0, b1, a1, x1, y1; } radiation_filter;   /* vtm/RadiationFilter.h:54-79 */
typedef struct { double b0, a1, y1; } reflection_filter;          /* vtm/ReflectionFilter.h:55-76 */
typedef struct { double b0, a1, gain, y1; } throat_filter;        /* vtm/Throat.h:52-85 */
typedef struct {                                                  /* vtm/BandpassFilter.h */
	double b0, a2, a1, x1, x2, y1, y2, prev_rate, prev_bw, prev_cf;
} bandpass_filter;

static void radiation_init(radiation_filter* f, double a) { f->b0 = a; f->b1 = -f->b0; f->a1 = -f->b0; f->x1 = f->y1 = 0.0; }
static double radiation_run(radiation_filter* f, double x)
{
	const double y = f->b0 * x + f->b1 * f->x1 - f->a1 * f->y1;
	f->x1 = x;
	f->y1 = y;
	return y;
}
static void reflection_init(reflection_filter* f, double a) { f->b0 = 1.0 - fabs(a); f->a1 = -a; f->y1 = 0.0; }
static double reflection_run(reflection_filter* f, double x)
{
	const double y = f->b0 * x - f->a1 * f->y1;
	f->y1 = y;
	return y;
}
static void throat_init(throat_filter* f, double rate, double cutoff, double gain)
{
	f->b0 = (cutoff * 2.0) / rate;
	f->a1 = f->b0 - 1.0;
	f->gain = gain;
	f->y1 = 0.0;
}
static double throat_run(throat_filter* f, double x)
{
	const double y = f->b0 * x - f->a1 * f->y1;
	f->y1 = y;
	return y * f->gain;
}
/* BandpassFilter::update, vtm/BandpassFilter.h:91-110 */
static void bandpass_update(bandpass_filter* f, double rate, double bw, double cf)
{
	if (rate == f->prev_rate && bw == f->prev_bw && cf == f->prev_cf) return;
	f->prev_rate = rate;
	f->prev_bw = bw;
	f->prev_cf = cf;
	const double pi = M_PI;
	const double T = 1.0 / rate;
	const double tan_value = tan(pi * bw * T);
	const double cos_value = cos(2.0 * pi * cf * T);
	f->a2 = (1.0 - tan_value) / (1.0 + tan_value);
	f->a1 = -(1.0 + f->a2) * cos_value;
	f->b0 = 0.5 - 0.5 * f->a2;
}
/* BandpassFilter::filter, vtm/BandpassFilter.h:114-122 */
static double bandpass_run(bandpass_filter* f, double x)
{
	const double y = f->b0 * (x - f->x2) - f->a1 * f->y1 - f->a2 * f->y2;
	f->x2 = f->x1;
	f->x1 = x;
	f->y2 = f->y1;
	f->y1 = y;
	return y;
}

/* ------------------------------------------------------------------------- */
/* the tube model                                                             */

#define MAX_DELAY 8

enum { S1, S2, S3, S4, S5, S6, S7, S8, S9, S10, N_SECTIONS };
enum { N1, N2, N3, N4, N5, N6, N_NASAL };
#define MAX_SECTIONS 30 /* VocalTractModel4: 30 oropharynx + 18 nasal sections */
#define MAX_NASAL 18
enum { P_PITCH, P_GLOT_VOL, P_ASP_VOL, P_FRIC_VOL, P_FRIC_POS, P_FRIC_CF, P_FRIC_BW,
       P_R1, P_R2, P_R3, P_R4, P_R5, P_R6, P_R7, P_R8, P_VELUM };

typedef struct { double top[MAX_DELAY + 1], bottom[MAX_DELAY + 1]; } section;

typedef struct {
	/* configuration (loadConfiguration, VocalTractModel0.h:266-305) */
	double output_rate, tp, tn_min, tn_max, breathiness, length, temperature, loss_factor;
	double aperture_radius, mouth_coef, nose_coef, nasal_radius[N_NASAL], throat_cutoff, throat_vol, mix_offset;
	double radius_coef[8];
	int waveform, modulation, delay, layout;
	/* derived (initializeSynthesizer, VocalTractModel0.h:338-392) */
	int sample_rate;
	double damping, crossmix_factor, breathiness_factor;
	/* state */
	section oro[MAX_SECTIONS], nasal[MAX_NASAL];
	unsigned in_ptr, out_ptr;
	double oro_k[8], nasal_k[N_NASAL], alpha_l, alpha_r, alpha_u, tap[8];
	double cur[VTMO_N_PARAM];
	radiation_filter mouth_rad, nose_rad;
	reflection_filter mouth_refl, nose_refl;
	throat_filter throat;
	bandpass_filter bandpass;
	glottal_source glottal;
	double noise_seed, noise_x1;
	src_state src;
	double* taps; /* optional debug taps for the current step: u, sig, thr, fir, lpnoise, pos0, pos1, x */
} vtm_model;

static double junction2(double left_radius, double right_radius)
{
	/* Junction2::configure, VocalTractModel2.h:214-218 == VocalTractModel0.h:488-490 */
	const double r0 = left_radius * left_radius;
	const double r1 = right_radius * right_radius;
	return (r0 - r1) / (r0 + r1);
}

/* loadConfiguration + initializeSynthesizer + initializeNasalCavity,
 * VocalTractModel0.h:266-305, :338-392, :457-470 (VocalTractModel2.h:335-376, :413-467, :534-543) */
static int model_init(vtm_model* m, const vtmo_config* c, int count_only)
{
	memset(m, 0, sizeof(*m));
	if (c->section_delay < 1 || c->section_delay > MAX_DELAY) return -1;
	m->delay = c->section_delay;
	m->layout = c->layout;
	if (m->layout != 0 && m->layout != 1) return -1;
	m->output_rate = c->output_rate;
	m->waveform = c->waveform;
	m->tp = c->glottal_pulse_tp;
	m->tn_min = c->glottal_pulse_tn_min;
	m->tn_max = c->glottal_pulse_tn_max;
	m->breathiness = c->breathiness;
	m->length = c->vocal_tract_length_offset + c->vocal_tract_length;
	if (m->length < 3.0) m->length = 3.0;
	else if (m->length > 30.0) m->length = 30.0;
	m->temperature = c->temperature;
	m->loss_factor = c->loss_factor;
	m->mouth_coef = c->mouth_coefficient;
	m->nose_coef = c->nose_coefficient;
	m->throat_cutoff = c->throat_cutoff;
	m->throat_vol = c->throat_volume;
	m->modulation = c->noise_modulation;
	m->mix_offset = c->mix_offset;
	m->aperture_radius = c->aperture_radius * c->global_radius_coef;
	m->nasal_radius[0] = 0.0;
	for (int i = 0; i < 5; ++i) m->nasal_radius[i + 1] = c->nasal_radius[i] * c->global_nasal_radius_coef;
	for (int i = 0; i < 8; ++i) m->radius_coef[i] = c->radius_coef[i] * c->global_radius_coef;

	/* reset(): VocalTractModel2.h:380-404 (inPtr 0, outPtr 1; D=1 is VocalTractModel0's cur/prev pair) */
	m->in_ptr = 0;
	m->out_ptr = 1;

	const double speed = 331.4 + (0.6 * m->temperature); /* Util::speedOfSound, VTMUtil.h:104-110 */
	/* TOTAL_SECTIONS is 10 for VocalTractModel0/2 and 30 for VocalTractModel4 (VocalTractModel4.h:467) */
	m->sample_rate = (int) ((speed * ((m->layout ? MAX_SECTIONS : N_SECTIONS) * m->delay) * 100.0) / m->length);
	const double nyquist = (float) m->sample_rate / 2.0f; /* int / float -> float arithmetic, VocalTractModel0.h:345 */
	m->breathiness_factor = m->breathiness / 100.0;
	m->crossmix_factor = 1.0 / amplitude60dB(m->mix_offset);
	m->damping = 1.0 - (m->loss_factor / 100.0);

	if (!count_only) {
		glottal_init(&m->glottal, m->waveform, m->sample_rate, m->tp, m->tn_min, m->tn_max);
	}
	const double mouth_ap = (nyquist - m->mouth_coef) / nyquist;
	radiation_init(&m->mouth_rad, mouth_ap);
	reflection_init(&m->mouth_refl, mouth_ap);
	const double nose_ap = (nyquist - m->nose_coef) / nyquist;
	radiation_init(&m->nose_rad, nose_ap);
	reflection_init(&m->nose_refl, nose_ap);

	for (int i = N2; i < N6; ++i) m->nasal_k[i] = junction2(m->nasal_radius[i], m->nasal_radius[i + 1]);
	m->nasal_k[N6] = junction2(m->nasal_radius[N6], m->aperture_radius);

	throat_init(&m->throat, m->sample_rate, m->throat_cutoff, amplitude60dB(m->throat_vol));
	src_init(&m->src, m->sample_rate, m->output_rate, count_only);
	m->bandpass.prev_rate = m->bandpass.prev_bw = m->bandpass.prev_cf = -1.0;
	m->noise_seed = 0.7892347; /* NoiseSource.h:32-34 */
	return 0;
}

/* setAllParameters, VocalTractModel0.h:698-716 */
static void model_set_parameters(vtm_model* m, const float* p)
{
	for (int i = P_PITCH; i <= P_FRIC_BW; ++i) m->cur[i] = p[i];
	for (int i = P_R1; i <= P_R8; ++i) {
		const double r = p[i] * m->radius_coef[i - P_R1];
		m->cur[i] = (r < 0.01) ? 0.01 : r; /* std::max(r, 0.01) */
	}
	m->cur[P_VELUM] = p[P_VELUM];
}

/* calculateTubeCoefficients, VocalTractModel0.h:484-512 */
static void model_tube_coefficients(vtm_model* m)
{
	for (int i = 0; i < 7; ++i) m->oro_k[i] = junction2(m->cur[P_R1 + i], m->cur[P_R1 + i + 1]);
	m->oro_k[7] = junction2(m->cur[P_R8], m->aperture_radius);
	const double r1_2 = m->cur[P_R4] * m->cur[P_R4];
	const double r0_2 = r1_2;
	const double r2_2 = m->cur[P_VELUM] * m->cur[P_VELUM];
	const double sum = 2.0 / (r0_2 + r1_2 + r2_2);
	m->alpha_l = sum * r0_2;
	m->alpha_r = sum * r1_2;
	m->alpha_u = sum * r2_2;
	const double rb = m->nasal_radius[N2] * m->nasal_radius[N2];
	m->nasal_k[N1] = (r2_2 - rb) / (r2_2 + rb);
}

/* setFricationTaps, VocalTractModel0.h:524-552 */
static void model_frication_taps(vtm_model* m)
{
	const double amp = amplitude60dB(m->cur[P_FRIC_VOL]);
	const int ip = (int) m->cur[P_FRIC_POS];
	const double complement = m->cur[P_FRIC_POS] - ip;
	const double remainder = 1.0 - complement;
	for (int i = 0; i < 8; ++i) {
		if (i == ip) {
			m->tap[i] = remainder * amp;
			if ((i + 1) < 8) m->tap[++i] = complement * amp;
		} else {
			m->tap[i] = 0.0;
		}
	}
}

static void propagate_junction(vtm_model* m, section* l, double k, section* r, double fric)
{
	/* propagateJunction(Junction2), VocalTractModel2.h:255-259 == VocalTractModel0.h:576-592 */
	const double delta = k * (l->top[m->out_ptr] - r->bottom[m->out_ptr]);
	r->top[m->in_ptr] = (l->top[m->out_ptr] + delta) * m->damping + fric;
	l->bottom[m->in_ptr] = (r->bottom[m->out_ptr] + delta) * m->damping;
}

/* vocalTract, VocalTractModel0.h:565-661 / VocalTractModel2.h:626-669 */
static double model_vocal_tract(vtm_model* m, double input, double frication)
{
	/* Section::movePointers, VocalTractModel2.h:241-248 */
	m->in_ptr = m->out_ptr;
	m->out_ptr = (m->out_ptr == (unsigned) m->delay) ? 0 : m->out_ptr + 1;
	const unsigned in = m->in_ptr, out = m->out_ptr;
	section* o = m->oro;
	section* n = m->nasal;
	const double d = m->damping;

	o[S1].top[in] = o[S1].bottom[out] * d + input;
	/* S1-S2: VocalTractModel0 adds no frication term here; x + 0.0 is exact */
	{
		const double delta = m->oro_k[0] * (o[S1].top[out] - o[S2].bottom[out]);
		o[S2].top[in] = (o[S1].top[out] + delta) * d;
		o[S1].bottom[in] = (o[S2].bottom[out] + delta) * d;
	}
	for (int i = S2, j = 1, k = 0; i < S4; ++i, ++j, ++k) {
		propagate_junction(m, &o[i], m->oro_k[j], &o[i + 1], m->tap[k] * frication);
	}
	{ /* 3-way junction, VocalTractModel0.h:595-604 */
		const double jp = (m->alpha_l * o[S4].top[out]) + (m->alpha_r * o[S5].bottom[out]) + (m->alpha_u * n[N1].bottom[out]);
		o[S4].bottom[in] = (jp - o[S4].top[out]) * d;
		o[S5].top[in] = ((jp - o[S5].bottom[out]) * d) + (m->tap[2] * frication);
		n[N1].top[in] = (jp - n[N1].bottom[out]) * d;
	}
	propagate_junction(m, &o[S5], m->oro_k[3], &o[S6], m->tap[3] * frication);
	/* pure delay with damping, VocalTractModel0.h:616-620 */
	o[S7].top[in] = (o[S6].top[out] * d) + (m->tap[4] * frication);
	o[S6].bottom[in] = o[S7].bottom[out] * d;
	for (int i = S7, j = 4, k = 5; i < S10; ++i, ++j, ++k) {
		propagate_junction(m, &o[i], m->oro_k[j], &o[i + 1], m->tap[k] * frication);
	}
	o[S10].bottom[in] = d * reflection_run(&m->mouth_refl, m->oro_k[7] * o[S10].top[out]);
	double output = radiation_run(&m->mouth_rad, (1.0 + m->oro_k[7]) * o[S10].top[out]);

	for (int i = N1; i < N6; ++i) {
		const double delta = m->nasal_k[i] * (n[i].top[out] - n[i + 1].bottom[out]);
		n[i + 1].top[in] = (n[i].top[out] + delta) * d;
		n[i].bottom[in] = (n[i + 1].bottom[out] + delta) * d;
	}
	n[N6].bottom[in] = d * reflection_run(&m->nose_refl, m->nasal_k[N6] * n[N6].top[out]);
	output += radiation_run(&m->nose_rad, (1.0 + m->nasal_k[N6]) * n[N6].top[out]);
	return output;
}

/* Simple copy between sections of one region, VocalTractModel4.h:296-300 */
static void propagate_copy(vtm_model* m, section* l, section* r)
{
	r->top[m->in_ptr] = l->top[m->out_ptr];
	l->bottom[m->in_ptr] = r->bottom[m->out_ptr];
}

/* Delay with damping and frication, VocalTractModel4.h:301-305 (== VocalTractModel2.h:251-254) */
static void propagate_damped(vtm_model* m, section* l, section* r, double fric)
{
	r->top[m->in_ptr] = l->top[m->out_ptr] * m->damping + fric;
	l->bottom[m->in_ptr] = r->bottom[m->out_ptr] * m->damping;
}

/* vocalTract of VocalTractModel4 (VocalTractModel4.h:671-745): 30 oropharynx + 18 nasal sections,
 * scattering only at the region boundaries, plain copies inside a region. */
static double model_vocal_tract4(vtm_model* m, double input, double frication)
{
	m->in_ptr = m->out_ptr;
	m->out_ptr = (m->out_ptr == (unsigned) m->delay) ? 0 : m->out_ptr + 1;
	const unsigned in = m->in_ptr, out = m->out_ptr;
	section* o = m->oro; /* o[i] = S(i+1) */
	section* n = m->nasal;
	const double d = m->damping;
	const double* k = m->oro_k; /* J1..J8 */
	const double* t = m->tap;   /* FC1..FC8 */

	o[0].top[in] = o[0].bottom[out] * d + input;
	propagate_copy(m, &o[0], &o[1]);
	propagate_copy(m, &o[1], &o[2]);
	propagate_junction(m, &o[2], k[0], &o[3], 0.0);
	propagate_copy(m, &o[3], &o[4]);
	propagate_junction(m, &o[4], k[1], &o[5], t[0] * frication);
	propagate_copy(m, &o[5], &o[6]);
	propagate_copy(m, &o[6], &o[7]);
	propagate_copy(m, &o[7], &o[8]);
	propagate_junction(m, &o[8], k[2], &o[9], t[1] * frication);
	propagate_copy(m, &o[9], &o[10]);
	propagate_copy(m, &o[10], &o[11]);
	{ /* 3-way junction S12 / S13 / N1 */
		const double jp = m->alpha_l * o[11].top[out] + m->alpha_r * o[12].bottom[out] + m->alpha_u * n[0].bottom[out];
		o[11].bottom[in] = (jp - o[11].top[out]) * d;
		o[12].top[in] = (jp - o[12].bottom[out]) * d + t[2] * frication;
		n[0].top[in] = (jp - n[0].bottom[out]) * d;
	}
	propagate_copy(m, &o[12], &o[13]);
	propagate_copy(m, &o[13], &o[14]);
	propagate_junction(m, &o[14], k[3], &o[15], t[3] * frication);
	propagate_copy(m, &o[15], &o[16]);
	propagate_copy(m, &o[16], &o[17]);
	propagate_damped(m, &o[17], &o[18], t[4] * frication);
	propagate_copy(m, &o[18], &o[19]);
	propagate_copy(m, &o[19], &o[20]);
	propagate_junction(m, &o[20], k[4], &o[21], t[5] * frication);
	propagate_copy(m, &o[21], &o[22]);
	propagate_copy(m, &o[22], &o[23]);
	propagate_copy(m, &o[23], &o[24]);
	propagate_junction(m, &o[24], k[5], &o[25], t[6] * frication);
	propagate_copy(m, &o[25], &o[26]);
	propagate_junction(m, &o[26], k[6], &o[27], t[7] * frication);
	propagate_copy(m, &o[27], &o[28]);
	propagate_copy(m, &o[28], &o[29]);
	o[29].bottom[in] = d * reflection_run(&m->mouth_refl, k[7] * o[29].top[out]);
	double output = radiation_run(&m->mouth_rad, (1.0 + k[7]) * o[29].top[out]);

	for (int g = 0; g < 6; ++g) { /* nasal regions of three sections, junction after each but the last */
		propagate_copy(m, &n[3 * g], &n[3 * g + 1]);
		propagate_copy(m, &n[3 * g + 1], &n[3 * g + 2]);
		if (g < 5) propagate_junction(m, &n[3 * g + 2], m->nasal_k[g], &n[3 * g + 3], 0.0);
	}
	n[17].bottom[in] = d * reflection_run(&m->nose_refl, m->nasal_k[5] * n[17].top[out]);
	output += radiation_run(&m->nose_rad, (1.0 + m->nasal_k[5]) * n[17].top[out]);
	return output;
}

/* execSynthesisStep, VocalTractModel0.h:396-445; returns the sample handed to the SRC */
static double model_step(vtm_model* m)
{
	const double f0 = frequency(m->cur[P_PITCH]);
	const double ax = amplitude60dB(m->cur[P_GLOT_VOL]);
	const double ah1 = amplitude60dB(m->cur[P_ASP_VOL]);
	model_tube_coefficients(m);
	model_frication_taps(m);
	bandpass_update(&m->bandpass, m->sample_rate, m->cur[P_FRIC_BW], m->cur[P_FRIC_CF]);

	/* NoiseSource::getSample NoiseSource.h:40-44; NoiseFilter::filter NoiseFilter.h:63-68 */
	const double product = m->noise_seed * 377.0;
	m->noise_seed = product - (int) product;
	const double white = m->noise_seed - 0.5;
	const double lp_noise = white + m->noise_x1;
	m->noise_x1 = white;

	if (m->waveform == 0) glottal_setup(&m->glottal, ax);
	m->glottal.tap_pos = m->taps ? m->taps + 5 : NULL;
	double pulse = glottal_sample(&m->glottal, f0);
	if (m->taps) { m->taps[3] = pulse; m->taps[4] = lp_noise; }
	const double pulsed_noise = lp_noise * pulse;
	pulse = ax * ((pulse * (1.0 - m->breathiness_factor)) + (pulsed_noise * m->breathiness_factor));

	double signal;
	if (m->modulation) {
		double crossmix = ax * m->crossmix_factor;
		crossmix = (crossmix < 1.0) ? crossmix : 1.0;
		signal = (pulsed_noise * crossmix) + (lp_noise * (1.0 - crossmix));
	} else {
		signal = lp_noise;
	}
	const double fric = bandpass_run(&m->bandpass, signal);
	if (m->taps) { m->taps[0] = (pulse + (ah1 * signal)) * 0.125; m->taps[1] = signal; m->taps[2] = pulse * 0.125; }
	signal = m->layout ? model_vocal_tract4(m, ((pulse + (ah1 * signal)) * 0.125), fric)
	                   : model_vocal_tract(m, ((pulse + (ah1 * signal)) * 0.125), fric);
	signal += throat_run(&m->throat, pulse * 0.125);
	if (m->taps) m->taps[7] = signal;
	return signal;
}

/* ------------------------------------------------------------------------- */
/* public API                                                                 */

int vtmo_derive(const vtmo_config* cfg, double control_rate, vtmo_derived* out)
{
	vtm_model* m = (vtm_model*) malloc(sizeof(vtm_model));
	if (!m) return -1;
	if (model_init(m, cfg, 0) != 0) { free(m); return -1; }
	out->sample_rate = m->sample_rate;
	out->control_steps = (unsigned) rint((double) m->sample_rate / control_rate);
	out->fir_taps = m->glottal.taps;
	out->table_div1 = m->glottal.div1;
	out->table_div2 = m->glottal.div2;
	out->tn_delta = m->glottal.tn_delta;
	out->time_register_increment = m->src.time_register_increment;
	out->phase_increment = m->src.phase_increment;
	out->pad_size = m->src.pad_size;
	out->upsampling = m->src.ratio >= 1.0;
	free(m);
	return 0;
}

void vtmo_noise_sequence(double* lp_noise, size_t count)
{
	double seed = 0.7892347, x1 = 0.0;
	for (size_t i = 0; i < count; ++i) {
		const double product = seed * 377.0;
		seed = product - (int) product;
		const double white = seed - 0.5;
		lp_noise[i] = white + x1;
		x1 = white;
	}
}

static size_t run(const vtmo_config* cfg, double control_rate, const float* params, size_t n_frames,
		float* out, size_t cap, double* internal_signal, int count_only, double* taps)
{
	vtm_model* m = (vtm_model*) malloc(sizeof(vtm_model));
	if (!m) return (size_t) -1;
	if (model_init(m, cfg, count_only) != 0) { free(m); return (size_t) -1; }
	m->src.out = out;
	m->src.out_cap = cap;

	/* Controller::synthesize, vtm_control_model/Controller.cpp:277-313.  The
	 * appended copy of the last frame (:283) is expressed by clamping the index. */
	if (n_frames > 0) {
		const unsigned control_steps = (unsigned) rint((double) m->sample_rate / control_rate);
		const float coef = 1.0f / control_steps;
		float cur[VTMO_N_PARAM], delta[VTMO_N_PARAM];
		size_t step = 0;
		for (size_t i = 1; i <= n_frames; ++i) {
			const float* p0 = params + (i - 1) * VTMO_N_PARAM;
			const float* p1 = params + ((i < n_frames) ? i : n_frames - 1) * VTMO_N_PARAM;
			for (int j = 0; j < VTMO_N_PARAM && !count_only; ++j) {
				cur[j] = p0[j];
				delta[j] = (p1[j] - cur[j]) * coef;
			}
			for (unsigned j = 0; j < control_steps; ++j, ++step) {
				if (count_only) {
					src_data_fill(&m->src, 0.0);
				} else {
					model_set_parameters(m, cur);
					m->taps = taps ? taps + step * 8 : NULL;
					const double s = model_step(m);
					if (internal_signal) internal_signal[step] = s;
					src_data_fill(&m->src, s);
				}
				for (int k = 0; k < VTMO_N_PARAM && !count_only; ++k) cur[k] += delta[k];
			}
		}
	}
	src_flush(&m->src); /* finishSynthesis, VocalTractModel0.h:720-723 */
	const size_t n = m->src.out_n;
	free(m);
	return n;
}

size_t vtmo_output_count(const vtmo_config* cfg, double control_rate, size_t n_frames)
{
	return run(cfg, control_rate, NULL, n_frames, NULL, 0, NULL, 1, NULL);
}

size_t vtmo_synthesize(const vtmo_config* cfg, double control_rate, const float* params, size_t n_frames,
		float* out, size_t out_capacity, double* internal_signal)
{
	return run(cfg, control_rate, params, n_frames, out, out_capacity, internal_signal, 0, NULL);
}

size_t vtmo_synthesize_batch(const vtmo_config* cfg, double control_rate, const float* params,
		size_t batch, size_t n_frames, float* out, size_t out_stride)
{
	size_t n = 0;
	for (size_t b = 0; b < batch; ++b) {
		n = run(cfg, control_rate, params + b * n_frames * VTMO_N_PARAM, n_frames,
				out + b * out_stride, out_stride, NULL, 0, NULL);
		if (n == (size_t) -1) return n;
	}
	return n;
}

/* Debug: per-step taps[n_steps][8] = tube input, band-pass input, throat input, FIR output,
 * low-passed noise, oscillator position after each half step (2), sample handed to the SRC. */
size_t vtmo_synthesize_debug(const vtmo_config* cfg, double control_rate, const float* params, size_t n_frames,
		float* out, size_t out_capacity, double* taps)
{
	return run(cfg, control_rate, params, n_frames, out, out_capacity, NULL, 0, taps);
}
