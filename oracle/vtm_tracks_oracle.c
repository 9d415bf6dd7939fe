/*
 * TEST INFRASTRUCTURE — NOT PART OF THE PRODUCT.  See vtm_tracks_oracle.h.
 * Paths are relative to /root/reference/gama_tts/src/.
 */
#include "vtm_tracks_oracle.h"

#include <math.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

enum { EV_TIME = 0, EV_HAS_INTERP = 1, EV_A = 2, EV_B = 3, EV_C = 4, EV_D = 5, EV_PARAM = 6, EV_SPECIAL = 22, N_PARAM = 16 };

typedef struct { double b0, b1, a1, a2; } butter2;

/* Butterworth2LowPassFilter<double>::update, vtm/Butterworth2LowpassFilter.h:88-107 */
static void butter2_design(butter2* f, double sample_rate, double cutoff)
{
	const double pi = M_PI;
	const double wcT = 2.0f * tan(pi * cutoff / sample_rate);
	const double wc2T2 = wcT * wcT;
	const double c1 = 2.0f * sqrt(2.0) * wcT;
	const double c2 = 1.0f / (wc2T2 + c1 + 4.0f);
	f->b0 = c2 * wc2T2;
	f->b1 = 2.0f * f->b0;
	f->a1 = c2 * (2.0f * wc2T2 - 8.0f);
	f->a2 = c2 * (wc2T2 - c1 + 4.0f);
}

/* DriftGenerator::drift, vtm_control_model/DriftGenerator.cpp:72-84 with
 * Butterworth2LowPassFilter<double>::filter, vtm/Butterworth2LowpassFilter.h:111-120 */
static double drift_sample(vtmo_drift_state* s, const butter2* f, double pitch_deviation, double pitch_offset)
{
	const double temp = s->seed * 377.0;
	s->seed = temp - (int) temp;
	const double x = (s->seed * pitch_deviation) - pitch_offset;
	const double y = f->b0 * (x + s->x2) + f->b1 * s->x1 - f->a1 * s->y1 - f->a2 * s->y2;
	s->x2 = s->x1;
	s->x1 = x;
	s->y2 = s->y1;
	s->y1 = y;
	return y;
}

#define EVENT(i) (events + (size_t) (i) * VTMO_EVENT_DOUBLES)
#define IS_EMPTY(v) (isinf(v) && (v) > 0) /* Event::EMPTY_PARAMETER = +infinity, EventList.cpp:38 */

/* EventList::generateOutput, vtm_control_model/EventList.cpp:930-1091 */
size_t vtmo_tracks_generate(const vtmo_track_config* cfg, const double* events, size_t n_events,
		vtmo_drift_state* drift, float* frames, size_t capacity)
{
	if (n_events < 2) return 0;
	const int cp = cfg->control_period;
	double cur[N_PARAM], delta[N_PARAM], scur[N_PARAM], sdelta[N_PARAM];
	for (int i = 0; i < N_PARAM; ++i) cur[i] = delta[i] = scur[i] = sdelta[i] = 0.0;

	/* DriftGenerator::setUp, DriftGenerator.cpp:49-56 */
	butter2 filt;
	butter2_design(&filt, cfg->drift_sample_rate, cfg->drift_lowpass_cutoff);
	const double pitch_deviation = cfg->drift_deviation * 2.0;
	const double pitch_offset = cfg->drift_deviation;

	for (int i = 0; i < N_PARAM; ++i) { /* :944-954 */
		cur[i] = EVENT(0)[EV_PARAM + i];
		size_t j = 1;
		double value;
		while (IS_EMPTY(value = EVENT(j)[EV_PARAM + i])) {
			if (++j >= n_events) break;
		}
		if (j < n_events) delta[i] = ((value - cur[i]) / (int) EVENT(j)[EV_TIME]) * cp;
	}

	double pa = 0.0, pb = 0.0, pc = 0.0, pd = 0.0;
	if (cfg->macro_intonation) { /* :959-981 */
		size_t j = 0;
		for (; j < n_events; ++j) {
			if (EVENT(j)[EV_HAS_INTERP] != 0.0) break;
		}
		if (j < n_events) {
			const double y1 = cfg->initial_pitch;
			const double x2 = (int) EVENT(j)[EV_TIME];
			const double* d = EVENT(j);
			if (cfg->smooth_intonation) {
				const double y2 = x2 * (x2 * (x2 * d[EV_A] + d[EV_B]) + d[EV_C]) + d[EV_D];
				pc = (y2 - y1) / x2;
				pd = y1;
			} else {
				const double y2 = x2 * d[EV_A] + d[EV_B];
				pa = (y2 - y1) / x2;
				pb = y1;
			}
		}
	}

	size_t target = 1;
	int target_time = (int) EVENT(target)[EV_TIME];
	int now = 0;
	size_t n = 0;
	while (target < n_events) { /* :988-1086 */
		float param[N_PARAM];
		for (int j = 0; j < N_PARAM; ++j) param[j] = (float) (cur[j] + scur[j]);
		if (!cfg->micro_intonation) param[0] = 0.0;
		if (cfg->intonation_drift) param[0] += (float) drift_sample(drift, &filt, pitch_deviation, pitch_offset);
		if (cfg->macro_intonation) {
			const double x = now;
			const double intonation = cfg->smooth_intonation ? x * (x * (x * pa + pb) + pc) + pd : x * pa + pb;
			param[0] += (float) intonation;
		}
		param[0] += (float) cfg->mean_pitch;
		if (n < capacity) {
			for (int j = 0; j < N_PARAM; ++j) frames[n * N_PARAM + j] = param[j];
		}
		++n;

		for (int j = 0; j < N_PARAM; ++j) {
			if (delta[j]) cur[j] += delta[j];
		}
		for (int j = 0; j < N_PARAM; ++j) {
			if (sdelta[j]) scur[j] += sdelta[j];
		}
		now += cp;
		if (now >= target_time) {
			if (++target == n_events) break;
			target_time = (int) EVENT(target)[EV_TIME];
			for (int special = 0; special < 2; ++special) { /* :1035-1070 */
				const int base = special ? EV_SPECIAL : EV_PARAM;
				double* c = special ? scur : cur;
				double* dl = special ? sdelta : delta;
				for (int j = 0; j < N_PARAM; ++j) {
					if (!IS_EMPTY(EVENT(target - 1)[base + j])) {
						size_t k = target;
						double value;
						while (IS_EMPTY(value = EVENT(k)[base + j])) {
							if (++k >= n_events) break;
						}
						if (!IS_EMPTY(value)) {
							dl[j] = ((value - c[j]) / ((int) EVENT(k)[EV_TIME] - now)) * cp;
						} else {
							dl[j] = 0.0;
						}
					}
				}
			}
			if (cfg->macro_intonation && EVENT(target - 1)[EV_HAS_INTERP] != 0.0) { /* :1072-1084 */
				const double* d = EVENT(target - 1);
				pa = d[EV_A];
				pb = d[EV_B];
				if (cfg->smooth_intonation) {
					pc = d[EV_C];
					pd = d[EV_D];
				}
			}
		}
	}
	return n;
}
