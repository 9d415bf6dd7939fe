/* Plain-C use of include/gama_vtm.h: a batch of identical two-second utterances through reference model 0's
 * semantics, peaks and sample counts printed.  Build (from the repo root):
 *   gcc -std=c99 -Wall -O2 -Iinclude examples/synthesize_batch.c -Lgama_tts_amd/lib -lgama_vtm \
 *       -Wl,-rpath,$PWD/gama_tts_amd/lib -o /tmp/synthesize_batch
 * Without an MI355X the program stops at the first synthesis call with GVTM_ERR_NO_DEVICE (there is no CPU path);
 * the design-only part (rates, output length) still runs. */
#include <stdio.h>
#include <stdlib.h>

#include "gama_vtm.h"

static gvtm_config male_voice(void)
{
	/* data/voice/english/0_male: vtm.txt + variant/male.txt */
	gvtm_config c = {0};
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
	{
		const double nasal[5] = {1.35, 1.96, 1.91, 1.3, 0.73};
		int i;
		for (i = 0; i < 5; ++i) c.nasal_radius[i] = nasal[i];
		for (i = 0; i < 8; ++i) c.radius_coef[i] = 1.0;
	}
	c.section_delay = 1;
	c.precision = GVTM_PRECISION_F32; /* what `model = 1` computes */
	c.tube_layout = GVTM_TUBE_10_6;
	return c;
}

int main(int argc, char** argv)
{
	const size_t batch = argc > 1 ? (size_t) atoi(argv[1]) : 4, frames = 500;
	const float frame[GVTM_N_PARAM] = {-12.0f, 60.0f, 0.0f, 0.0f, 5.5f, 2500.0f, 500.0f,
	                                   0.8f, 0.89f, 0.99f, 0.81f, 0.76f, 1.05f, 1.23f, 0.01f, 0.1f}; /* an "aa"-like posture */
	gvtm_config cfg = male_voice();
	gvtm_plan* plan = NULL;
	gvtm_info info;
	size_t n, b, f;
	int rc, device = gvtm_device_count() > 0 ? 0 : GVTM_DEVICE_NONE;

	rc = gvtm_plan_create(&cfg, 250.0, device, &plan);
	if (rc != GVTM_OK) {
		fprintf(stderr, "plan: %s (%s)\n", gvtm_status_string(rc), gvtm_last_error());
		return 1;
	}
	gvtm_plan_info(plan, &info);
	n = gvtm_output_count(plan, frames);
	printf("internal rate %d Hz, %u steps per frame, %zu samples per utterance of %zu frames\n", info.internal_sample_rate,
			info.control_steps, n, frames);
	{
		float* params = malloc(sizeof(float) * batch * frames * GVTM_N_PARAM);
		float* audio = malloc(sizeof(float) * batch * n);
		int64_t* counts = malloc(sizeof(int64_t) * batch);
		float* peaks = malloc(sizeof(float) * batch);
		if (!params || !audio || !counts || !peaks) return 1;
		for (b = 0; b < batch; ++b) {
			for (f = 0; f < frames; ++f) {
				int k;
				for (k = 0; k < GVTM_N_PARAM; ++k) params[(b * frames + f) * GVTM_N_PARAM + k] = frame[k];
			}
		}
		rc = gvtm_synthesize_batch_host(plan, params, NULL, batch, frames, audio, n, counts, peaks);
		if (rc != GVTM_OK) {
			printf("synthesis: %s (%s)\n", gvtm_status_string(rc), gvtm_last_error());
		} else {
			for (b = 0; b < batch; ++b) printf("utterance %zu: %lld samples, peak %g, scale %g\n", b, (long long) counts[b], peaks[b], 0.95 / peaks[b]);
		}
		free(params); free(audio); free(counts); free(peaks);
	}
	gvtm_plan_destroy(plan);
	return rc == GVTM_OK || rc == GVTM_ERR_NO_DEVICE ? 0 : 1;
}
