/*
 * TEST INFRASTRUCTURE — NOT PART OF THE PRODUCT.
 *
 * CPU restatement (plain C, scalar; TFloat = double and TFloat = float) of GamaTTS's
 * vocal-tract-model hot path, used only as the parity checker by tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg.  Nothing under gama_tts_amd/ may include, link
 * or call it.
 *
 * Pinned: bit-identical to the real reference (oracle/_ref/ref_vtm, compiled
 * from /root/reference with -O2 -ffp-contract=off) on every golden vector in
 * tests/golden/ (tests/test_oracle_vs_golden.py), and — in the build container,
 * where /root/reference exists — on freshly generated random tracks
 * (tests/test_oracle_vs_reference.py).
 *
 * What it restates (reference file:line in vtm_oracle.c next to each function):
 *   Controller::synthesize                      vtm_control_model/Controller.cpp:277-313
 *   VocalTractModel0<TFloat> / VocalTractModel2<TFloat,D> / VocalTractModel4<TFloat,1>, TFloat = double | float
 *                                               vtm/VocalTractModel0.h, vtm/VocalTractModel2.h, vtm/VocalTractModel4.h
 *   WavetableGlottalSource (+FIR), SampleRateConverter, BandpassFilter,
 *   NoiseSource/NoiseFilter, Radiation/ReflectionFilter, Throat, VTMUtil
 */
#ifndef VTM_ORACLE_H_
#define VTM_ORACLE_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VTMO_N_PARAM 16
#define VTMO_SRC_FILTER_LENGTH 3328 /* 13 zero crossings * 256 phases */
#define VTMO_WAVETABLE_LENGTH 512
#define VTMO_FIR_MAX_TAPS 401

/* The scalar keys VocalTractModel0/2::loadConfiguration reads
 * (VocalTractModel0.h:266-305), already converted to numbers. */
typedef struct vtmo_config {
	double output_rate;               /* output_rate */
	int    waveform;                  /* waveform: 0 pulse, 1 sine */
	double glottal_pulse_tp;          /* glottal_pulse_tp (%) */
	double glottal_pulse_tn_min;      /* glottal_pulse_tn_min (%) */
	double glottal_pulse_tn_max;      /* glottal_pulse_tn_max (%) */
	double breathiness;               /* breathiness (%) */
	double vocal_tract_length_offset; /* vocal_tract_length_offset (cm) */
	double vocal_tract_length;        /* vocal_tract_length (cm) */
	double temperature;               /* temperature (deg C) */
	double loss_factor;               /* loss_factor (%) */
	double mouth_coefficient;         /* mouth_coefficient */
	double nose_coefficient;          /* nose_coefficient */
	double throat_cutoff;             /* throat_cutoff (Hz) */
	double throat_volume;             /* throat_volume (dB) */
	int    noise_modulation;          /* noise_modulation */
	double mix_offset;                /* mix_offset (dB) */
	double global_radius_coef;        /* global_radius_coef */
	double global_nasal_radius_coef;  /* global_nasal_radius_coef */
	double aperture_radius;           /* aperture_radius (cm) */
	double nasal_radius[5];           /* nasal_radius_1..5 (cm) */
	double radius_coef[8];            /* radius_1_coef..radius_8_coef */
	int    section_delay;             /* VocalTractModel2's SectionDelay template argument; 1 == VocalTractModel0 */
	int    layout;                    /* 0: 10 + 6 sections (VocalTractModel0/2); 1: 30 + 18 sections (VocalTractModel4) */
	int    float_model;               /* 0: TFloat = double (models 0, 2, 3, 4); 1: TFloat = float (model 1 = VocalTractModel0<float>,
	                                     VocalTractModel2<float,D>, VocalTractModel4<float,1>) */
} vtmo_config;

/* VocalTractModel5::loadConfiguration (vtm/VocalTractModel5.h:375-421), as numbers. */
typedef struct vtmo5_config {
	double output_rate;
	int    waveform, noise_modulation, bypass, constant_radius_mouth_impedance;
	double glottal_pulse_tp, glottal_pulse_tn_min, glottal_pulse_tn_max, breathiness;
	double vocal_tract_length_offset, vocal_tract_length, temperature, loss_factor, mix_offset;
	double global_radius_coef, global_nasal_radius_coef;
	double nasal_radius[6];           /* nasal_radius_2 .. nasal_radius_7 */
	double radius_coef[8];
	double glottal_noise_cutoff, frication_noise_cutoff, frication_factor, min_glottal_loss, max_glottal_loss,
	       glottal_lowpass_cutoff, mouth_impedance_radius;
	int    float_model;
} vtmo5_config;

/* VocalTractModel5<TFloat,1> (reference model 5) driven by Controller::synthesize: returns the number of
 * output samples (only out_capacity are stored).  ORACLE ONLY this round: the device path does not serve
 * model 5 yet.  sample_rate_milli, if non-NULL, receives the internal rate in mHz (it is not an integer). */
size_t vtmo5_synthesize(const vtmo5_config* cfg, double control_rate, const float* params, size_t n_frames,
		float* out, size_t out_capacity, int* sample_rate_milli);

/* Design-time quantities derived from the configuration. */
typedef struct vtmo_derived {
	int      sample_rate;             /* internal rate (VocalTractModel0.h:344, VocalTractModel2.h:419) */
	unsigned control_steps;           /* Controller.cpp:286 */
	int      fir_taps;                /* WavetableGlottalSourceFIRFilter.h:86 */
	unsigned table_div1, table_div2;  /* WavetableGlottalSource.h:105-106 */
	double   tn_delta;                /* WavetableGlottalSource.h:108 */
	unsigned time_register_increment; /* SampleRateConverter.h:145 */
	unsigned phase_increment;         /* SampleRateConverter.h:154 (down-sampling only) */
	int      pad_size;                /* SampleRateConverter.h:158-160 */
	int      upsampling;              /* sampleRateRatio_ >= 1 */
} vtmo_derived;

int vtmo_derive(const vtmo_config* cfg, double control_rate, vtmo_derived* out);

/* Design tables (exposed so that tests can compare the product's host-side
 * table design with the restatement). */
int  vtmo_fir_coefficients(double* coef /* [VTMO_FIR_MAX_TAPS] */);
void vtmo_src_filter(double* h /* [3328] */, double* delta_h /* [3328] */);
void vtmo_wavetable(const vtmo_config* cfg, int sample_rate, double* table /* [512] */);
/* the same tables as the TFloat = float models design them (float arithmetic throughout; 47 FIR taps) */
int  vtmo_fir_coefficients_f32(float* coef /* [VTMO_FIR_MAX_TAPS] */);
void vtmo_src_filter_f32(float* h /* [3328] */, float* delta_h /* [3328] */);
void vtmo_wavetable_f32(const vtmo_config* cfg, int sample_rate, float* table /* [512] */);
int  vtmo_fir_coefficients_f64(double* coef);
void vtmo_src_filter_f64(double* h, double* delta_h);
void vtmo_wavetable_f64(const vtmo_config* cfg, int sample_rate, double* table);
/* lpNoise[n] of VocalTractModel0.h:408 for n = 0..count-1 (same for every utterance). */
void vtmo_noise_sequence(double* lp_noise, size_t count);

/* Number of output samples finishSynthesis() leaves in outputBuffer() for
 * n_frames input frames (0 frames -> flush only). */
size_t vtmo_output_count(const vtmo_config* cfg, double control_rate, size_t n_frames);

/* One utterance: params[n_frames][16] float32 -> out[<= out_capacity] float32.
 * Returns the number of samples produced (may exceed out_capacity, in which
 * case only out_capacity samples were stored), or (size_t)-1 on bad config.
 * internal_signal, if non-NULL, receives the n_frames*control_steps
 * internal-rate samples handed to the sample-rate converter (debug tap). */
size_t vtmo_synthesize(const vtmo_config* cfg, double control_rate,
		const float* params, size_t n_frames,
		float* out, size_t out_capacity, double* internal_signal);

/* Debug variant: taps[n_frames*control_steps][8] receives, per internal step: tube input,
 * band-pass input, throat input, glottal FIR output, low-passed noise, oscillator position
 * after each of the two half steps, sample handed to the SRC. */
size_t vtmo_synthesize_debug(const vtmo_config* cfg, double control_rate,
		const float* params, size_t n_frames, float* out, size_t out_capacity, double* taps);

/* Batch of equal-length utterances, params[batch][n_frames][16],
 * out[batch][out_stride]; returns samples per utterance. */
size_t vtmo_synthesize_batch(const vtmo_config* cfg, double control_rate,
		const float* params, size_t batch, size_t n_frames,
		float* out, size_t out_stride);

/* libm of this machine over an array (powf(base, x[i]); tanf / cosf for which = 0 / 1) */
void vtmo_libm_powf(float base, const float* x, size_t n, float* out);
void vtmo_libm_tanf_cosf(int which, const float* x, size_t n, float* out);

/* Util::calculateOutputScale (VTMUtil.cpp:48-67): 0.95/max|x|, 0 if max < 1e-30. */
float vtmo_output_scale(const float* x, size_t n);

#ifdef __cplusplus
}
#endif

#endif /* VTM_ORACLE_H_ */
