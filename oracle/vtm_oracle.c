/*
 * TEST INFRASTRUCTURE — NOT PART OF THE PRODUCT.  See vtm_oracle.h.
 *
 * Scalar restatement of the GamaTTS vocal-tract-model hot path.  The algorithm
 * lives in vtm_oracle_body.inc, written once over `real` and compiled twice
 * (vtm_oracle_f64.c: TFloat = double, vtm_oracle_f32.c: TFloat = float); every
 * function there names the reference lines whose arithmetic (including
 * evaluation order, so that the result is bit-identical when compiled with
 * -ffp-contract=off) it follows.  This file holds what does not depend on
 * TFloat and dispatches on vtmo_config::float_model.
 */
#include "vtm_oracle.h"

#include <math.h>
#include <stdlib.h>

size_t vtmo_run_f64(const vtmo_config*, double, const float*, size_t, float*, size_t, double*, int, double*);
size_t vtmo_run_f32(const vtmo_config*, double, const float*, size_t, float*, size_t, double*, int, double*);
int vtmo_derive_f64(const vtmo_config*, double, vtmo_derived*);
int vtmo_derive_f32(const vtmo_config*, double, vtmo_derived*);

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

/* NoiseSource::getSample (NoiseSource.h:40-44) through NoiseFilter<double> (NoiseFilter.h:63-68) */
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

/* This machine's libm, vectorised: lets the tests pin the product's restatement of glibc's powf
 * (gama_tts_amd/csrc/vtm_math.hpp) over whole argument ranges. */
void vtmo_libm_powf(float base, const float* x, size_t n, float* out)
{
	for (size_t i = 0; i < n; ++i) out[i] = powf(base, x[i]);
}
void vtmo_libm_tanf_cosf(int which, const float* x, size_t n, float* out)
{
	for (size_t i = 0; i < n; ++i) out[i] = which ? cosf(x[i]) : tanf(x[i]);
}

int vtmo_fir_coefficients(double* coef) { return vtmo_fir_coefficients_f64(coef); }
void vtmo_src_filter(double* h, double* delta_h) { vtmo_src_filter_f64(h, delta_h); }
void vtmo_wavetable(const vtmo_config* cfg, int sample_rate, double* table) { vtmo_wavetable_f64(cfg, sample_rate, table); }

static size_t run(const vtmo_config* cfg, double control_rate, const float* params, size_t n_frames,
		float* out, size_t cap, double* internal_signal, int count_only, double* taps)
{
	return cfg->float_model ? vtmo_run_f32(cfg, control_rate, params, n_frames, out, cap, internal_signal, count_only, taps)
	                        : vtmo_run_f64(cfg, control_rate, params, n_frames, out, cap, internal_signal, count_only, taps);
}

size_t vtmo5_run_f64(const vtmo5_config*, double, const float*, size_t, float*, size_t, int*);
size_t vtmo5_run_f32(const vtmo5_config*, double, const float*, size_t, float*, size_t, int*);

size_t vtmo5_synthesize(const vtmo5_config* cfg, double control_rate, const float* params, size_t n_frames,
		float* out, size_t out_capacity, int* sample_rate_milli)
{
	return cfg->float_model ? vtmo5_run_f32(cfg, control_rate, params, n_frames, out, out_capacity, sample_rate_milli)
	                        : vtmo5_run_f64(cfg, control_rate, params, n_frames, out, out_capacity, sample_rate_milli);
}

int vtmo_derive(const vtmo_config* cfg, double control_rate, vtmo_derived* out)
{
	return cfg->float_model ? vtmo_derive_f32(cfg, control_rate, out) : vtmo_derive_f64(cfg, control_rate, out);
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
