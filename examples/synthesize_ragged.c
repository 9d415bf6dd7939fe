/* Plain-C use of include/gama_vtm.h: a RAGGED batch (utterances of different lengths) on a down-sampling plan, the case
 * in which the row stride must come from gvtm_output_capacity().  Reference model 3 = VocalTractModel2<double,3>
 * (60 102 Hz internal) converted down to 44.1 kHz: at a few lengths the reference's SampleRateConverter runs into its
 * flush overrun (SampleRateConverter.h:298-308 with :462-471) and returns ~750 samples more than a slightly LONGER
 * utterance -- e.g. 2334 frames -> 411 798 samples against 411 223 for 2335 frames.  A stride of
 * gvtm_output_count(plan, max_frames) would cut that row short.
 *
 * Build (from the repo root):
 *   gcc -std=c99 -Wall -O2 -Iinclude examples/synthesize_ragged.c -Lgama_tts_amd/lib -lgama_vtm \
 *       -Wl,-rpath,$PWD/gama_tts_amd/lib -o /tmp/synthesize_ragged
 * Without an MI355X the program prints the sizing (design-only plan) and stops at the synthesis call with
 * GVTM_ERR_NO_DEVICE (there is no CPU path). */
#include <stdio.h>
#include <stdlib.h>

#include "gama_vtm.h"

static gvtm_config male_voice_model3(void)
{
	/* data/voice/english/0_male: vtm.txt + variant/male.txt, model = 3 */
	gvtm_config c = {0};
	const double nasal[5] = {1.35, 1.96, 1.91, 1.3, 0.73};
	int i;
	c.output_rate = 44100.0;
	c.waveform = 0;
	c.noise_modulation = 1;
	c.glottal_pulse_tp = 40.0;
	c.glottal_pulse_tn_min = 24.0;
	c.glottal_pulse_tn_max = 24.0;
	c.breathiness = 0.5;
	c.vocal_tract_length_offset = 0.0;
	c.vocal_tract_length = 17.5;
	c.temperature = 32.0;
	c.loss_factor = 0.8;
	c.mouth_coefficient = 5000.0;
	c.nose_coefficient = 5000.0;
	c.throat_cutoff = 1500.0;
	c.throat_volume = 6.0;
	c.mix_offset = 48.0;
	c.global_radius_coef = 1.0;
	c.global_nasal_radius_coef = 1.0;
	c.aperture_radius = 3.05;
	for (i = 0; i < 5; ++i) c.nasal_radius[i] = nasal[i];
	for (i = 0; i < 8; ++i) c.radius_coef[i] = 1.0;
	c.section_delay = 3;                 /* model 3 */
	c.precision = GVTM_PRECISION_F64;    /* the factory's TFloat = double */
	c.tube_layout = GVTM_TUBE_10_6;
	return c;
}

int main(void)
{
	/* 2334 frames is an overrun length of this plan; its neighbour is one frame longer and 575 samples SHORTER */
	const int32_t frame_counts[3] = {2334, 2335, 40};
	const size_t batch = 3, max_frames = 2335;
	const float frame[GVTM_N_PARAM] = {-12.0f, 60.0f, 0.0f, 0.0f, 5.5f, 2500.0f, 500.0f,
	                                   0.8f, 0.89f, 0.99f, 0.81f, 0.76f, 1.05f, 1.23f, 0.01f, 0.1f};
	gvtm_config cfg = male_voice_model3();
	gvtm_plan* plan = NULL;
	size_t stride, b, f, longest;
	int k, rc, device = gvtm_device_count() > 0 ? 0 : GVTM_DEVICE_NONE;

	rc = gvtm_plan_create(&cfg, 250.0, device, &plan);
	if (rc != GVTM_OK) {
		fprintf(stderr, "plan: %s (%s)\n", gvtm_status_string(rc), gvtm_last_error());
		return 1;
	}
	stride = gvtm_output_capacity(plan, max_frames); /* holds every utterance of up to max_frames frames */
	longest = gvtm_output_count(plan, max_frames);
	for (b = 0; b < batch; ++b) {
		printf("utterance %zu: %d frames -> %zu samples\n", b, (int) frame_counts[b], gvtm_output_count(plan, (size_t) frame_counts[b]));
	}
	printf("gvtm_output_count(max_frames) = %zu, gvtm_output_capacity(max_frames) = %zu\n", longest, stride);
	if (stride < gvtm_output_count(plan, 2334)) return 2; /* (the property this example is about) */
	{
		float* params = calloc(batch * max_frames * GVTM_N_PARAM, sizeof(float));
		float* audio = malloc(sizeof(float) * batch * stride);
		int64_t counts[3];
		float peaks[3];
		if (!params || !audio) return 1;
		for (b = 0; b < batch; ++b) {
			for (f = 0; f < (size_t) frame_counts[b]; ++f) {
				for (k = 0; k < GVTM_N_PARAM; ++k) params[(b * max_frames + f) * GVTM_N_PARAM + k] = frame[k];
			}
		}
		rc = gvtm_synthesize_batch_host(plan, params, frame_counts, batch, max_frames, audio, stride, counts, peaks);
		if (rc != GVTM_OK) {
			printf("synthesis: %s (%s)\n", gvtm_status_string(rc), gvtm_last_error());
		} else {
			for (b = 0; b < batch; ++b) {
				printf("utterance %zu: %lld samples in a row of %zu, peak %g\n", b, (long long) counts[b], stride, peaks[b]);
				if ((size_t) counts[b] > stride) rc = GVTM_ERR_INVALID_ARGUMENT;
			}
		}
		free(params); free(audio);
	}
	gvtm_plan_destroy(plan);
	return rc == GVTM_OK || rc == GVTM_ERR_NO_DEVICE ? 0 : 1;
}
