// EventList::generateOutput on the device (vtm_control_model/EventList.cpp:930-1091).
//
// One 16-lane row per utterance, one lane per parameter (the reference's inner `for j < numParam`
// loops become the lanes), four utterances per wavefront.  Every lane walks the control periods of its
// utterance with its own running value + delta (double, as in the reference) and, at an event boundary,
// its own forward search for the next event that carries a value for its parameter; lane 0 also runs
// the pitch extras (drift generator, macro-intonation polynomial, mean pitch).  The 16 lanes of a row
// read one event's parameters[16] (128 contiguous bytes) and write one frame (64 contiguous bytes).
// Bit parity with the reference: same double operations in the same order, no FMA contraction.
#include "vtm_tracks.hpp"

#include <cmath>

namespace gvtm {

namespace {

__device__ __forceinline__ bool is_empty(double v)
{
	return v == HUGE_VAL; // Event::EMPTY_PARAMETER = +infinity (EventList.cpp:38)
}

} // namespace

__global__ __launch_bounds__(64) void vtm_tracks_kernel(const TrackArgs a)
{
#pragma clang fp contract(off)
	const int lane = threadIdx.x & 63;
	const int j = lane & 15; // parameter
	const size_t utt = static_cast<size_t>(blockIdx.x) * 4 + (lane >> 4);
	if (utt >= a.batch) return;
	const TrackConstants& k = a.k;
	const gvtm_event* ev = a.events + a.event_offsets[utt];
	const int64_t n_events = a.event_offsets[utt + 1] - a.event_offsets[utt];
	float* out = a.params + utt * a.max_frames * 16;
	if (n_events < 2) { // EventList.cpp:932-934
		if (j == 0 && a.frame_counts) a.frame_counts[utt] = 0;
		return;
	}
	const int cp = k.control_period;

	// current values and deltas of my parameter (:944-954); the special parameters start at 0
	double cur = ev[0].param[j], delta = 0.0, scur = 0.0, sdelta = 0.0;
	{
		int64_t q = 1;
		double value;
		while (is_empty(value = ev[q].param[j])) {
			if (++q >= n_events) break;
		}
		if (q < n_events) delta = ((value - cur) / ev[q].time_ms) * cp;
	}

	// lane 0: macro intonation polynomial (:959-981) and the drift generator's state
	double pa = 0.0, pb = 0.0, pc = 0.0, pd = 0.0;
	gvtm_drift_state ds = {0.7892347, 0.0, 0.0, 0.0, 0.0}; // DriftGenerator.cpp:28, :40
	if (j == 0) {
		if (a.drift) ds = a.drift[utt];
		if (k.macro_intonation) {
			int64_t q = 0;
			for (; q < n_events; ++q) {
				if (ev[q].has_interp) break;
			}
			if (q < n_events) {
				const double y1 = k.initial_pitch;
				const double x2 = ev[q].time_ms;
				const double* d = ev[q].interp;
				if (k.smooth_intonation) {
					const double y2 = x2 * (x2 * (x2 * d[0] + d[1]) + d[2]) + d[3];
					pc = (y2 - y1) / x2;
					pd = y1;
				} else {
					const double y2 = x2 * d[0] + d[1];
					pa = (y2 - y1) / x2;
					pb = y1;
				}
			}
		}
	}

	int64_t target = 1;
	int target_time = ev[target].time_ms;
	int now = 0;
	size_t n = 0;
	while (target < n_events) { // :988-1086
		float p = static_cast<float>(cur + scur);
		if (j == 0) {
			if (!k.micro_intonation) p = 0.0f;
			if (k.intonation_drift) {
				// DriftGenerator::drift (DriftGenerator.cpp:72-84) through Butterworth2LowPassFilter::filter
				const double temp = ds.seed * 377.0;
				ds.seed = temp - static_cast<int>(temp);
				const double x = (ds.seed * k.pitch_deviation) - k.pitch_offset;
				const double y = k.b0 * (x + ds.x2) + k.b1 * ds.x1 - k.a1 * ds.y1 - k.a2 * ds.y2;
				ds.x2 = ds.x1;
				ds.x1 = x;
				ds.y2 = ds.y1;
				ds.y1 = y;
				p += static_cast<float>(y);
			}
			if (k.macro_intonation) {
				const double x = now;
				const double intonation = k.smooth_intonation ? x * (x * (x * pa + pb) + pc) + pd : x * pa + pb;
				p += static_cast<float>(intonation);
			}
			p += static_cast<float>(k.mean_pitch);
		}
		if (n < a.max_frames) out[n * 16 + j] = p;
		++n;

		if (delta != 0.0) cur += delta;
		if (sdelta != 0.0) scur += sdelta;
		now += cp;
		if (now >= target_time) {
			if (++target == n_events) break;
			target_time = ev[target].time_ms;
			if (!is_empty(ev[target - 1].param[j])) { // :1035-1052
				int64_t q = target;
				double value;
				while (is_empty(value = ev[q].param[j])) {
					if (++q >= n_events) break;
				}
				delta = is_empty(value) ? 0.0 : ((value - cur) / (ev[q].time_ms - now)) * cp;
			}
			if (!is_empty(ev[target - 1].special[j])) { // :1053-1070
				int64_t q = target;
				double value;
				while (is_empty(value = ev[q].special[j])) {
					if (++q >= n_events) break;
				}
				sdelta = is_empty(value) ? 0.0 : ((value - scur) / (ev[q].time_ms - now)) * cp;
			}
			if (j == 0 && k.macro_intonation && ev[target - 1].has_interp) { // :1072-1084
				const double* d = ev[target - 1].interp;
				pa = d[0];
				pb = d[1];
				if (k.smooth_intonation) {
					pc = d[2];
					pd = d[3];
				}
			}
		}
	}
	if (j == 0) {
		if (a.frame_counts) a.frame_counts[utt] = static_cast<int32_t>(n);
		if (a.drift) a.drift[utt] = ds;
	}
}

hipError_t launch_tracks(const TrackArgs& args, hipStream_t stream)
{
	const unsigned groups = static_cast<unsigned>((args.batch + 3) / 4);
	if (groups == 0) return hipSuccess;
	hipLaunchKernelGGL(vtm_tracks_kernel, dim3(groups), dim3(64), 0, stream, args);
	return hipGetLastError();
}

} // namespace gvtm
