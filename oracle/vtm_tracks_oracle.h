/*
 * TEST INFRASTRUCTURE — NOT PART OF THE PRODUCT.
 *
 * CPU restatement (plain C) of the step that feeds the vocal-tract hot path:
 *   GS::VTMControlModel::EventList::generateOutput()   vtm_control_model/EventList.cpp:930-1091
 *   GS::VTMControlModel::DriftGenerator                vtm_control_model/DriftGenerator.cpp:72-84, :49-56
 *   GS::VTM::Butterworth2LowPassFilter<double>         vtm/Butterworth2LowpassFilter.h
 * i.e. event list (posture targets per parameter, special-parameter targets, macro-intonation
 * polynomials) -> one float32[16] parameter frame per control period.
 *
 * Pinned: bit-identical to the real reference on the captured fixtures in
 * tests/golden/tracks_golden.npz (oracle/_ref/ref_tracks_capture: the reference's own text parser,
 * rules and EventList, six generateOutput() calls per text with different intonation settings).
 */
#ifndef VTM_TRACKS_ORACLE_H_
#define VTM_TRACKS_ORACLE_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VTMO_EVENT_DOUBLES 38 /* time, has_interp, a, b, c, d, parameters[16], specialParameters[16]; empty = +inf */

typedef struct vtmo_track_config {
	int    control_period;      /* ms */
	int    macro_intonation, micro_intonation, intonation_drift, smooth_intonation;
	double initial_pitch, mean_pitch;
	double drift_deviation, drift_sample_rate, drift_lowpass_cutoff; /* DriftGenerator::setUp arguments */
} vtmo_track_config;

/* DriftGenerator state: noise seed and the Butterworth filter's memory.  A fresh generator is
 * {0.7892347, 0, 0, 0, 0}; the state carries over between generateOutput() calls of one Controller. */
typedef struct vtmo_drift_state { double seed, x1, x2, y1, y2; } vtmo_drift_state;

/* events[n_events][VTMO_EVENT_DOUBLES] -> frames[<= capacity][16]; returns the number of frames
 * generateOutput() pushes (may exceed capacity; only capacity frames are stored).  drift is in/out. */
size_t vtmo_tracks_generate(const vtmo_track_config* cfg, const double* events, size_t n_events,
		vtmo_drift_state* drift, float* frames, size_t capacity);

#ifdef __cplusplus
}
#endif

#endif
