// EventList::generateOutput on the device (vtm_control_model/EventList.cpp:930-1091).
//
// One utterance per wavefront, one lane per parameter (the reference's inner `for j < numParam` loops
// become 16 lanes; the other 48 stay idle on purpose, see below).  Every lane walks the control periods of its
// utterance with its own running value + delta (double, as in the reference) and, at an event boundary,
// its own forward search for the next event that carries a value for its parameter; lane 0 also runs
// the pitch extras (drift generator, macro-intonation polynomial, mean pitch).  The 16 lanes read one
// event's parameters[16] (128 contiguous bytes) and write one frame (64 contiguous bytes).
// The kernel is latency-bound, not throughput-bound: an event boundary is a round trip to memory (~1-2 us)
// and a wavefront pays it for every boundary of every utterance it hosts, one at a time (the rows diverge).
// Four utterances per wavefront measured 1.14 ms on batch 4096 x 80 events (13.5 us per event); one per
// wavefront takes the same boundaries in parallel across four times as many wavefronts.
// Bit parity with the reference: same double operations in the same order, no FMA contraction.
#include "vtm_tracks.hpp"

#include <cmath>

namespace gvtm {

namespace {

__device__ __forceinline__ bool is_empty(double v)
{
	return v == HUGE_VAL; // Event::EMPTY_PARAMETER = +infinity (EventList.cpp:38)
}

// A value that came from memory, re-issued from the vector ALU.  gfx9 counts loads AND stores in one
// in-order counter (vmcnt): a register the compiler believes may still be in flight at the top of the
// per-frame loop costs an s_waitcnt vmcnt(0) there, which also waits for the previous frame's STORE to be
// acknowledged (~2 us per frame, measured 1.19 ms per launch).  Every loop-carried value that is loaded
// (start values, the next event's time, the intonation cubic) is therefore settled where it is loaded, at an
// event boundary, and the per-frame path carries ALU results only.
__device__ __forceinline__ int settle(int v)
{
	int r;
	asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(v));
	return r;
}
__device__ __forceinline__ double settle(double v)
{
	return __hiloint2double(settle(__double2hiint(v)), settle(__double2loint(v)));
}

} // namespace

__global__ __launch_bounds__(16) void vtm_tracks_kernel(const TrackArgs a)
{
#pragma clang fp contract(off)
	const int j = threadIdx.x & 15; // parameter
	const size_t utt = blockIdx.x;
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
	double cur = settle(ev[0].param[j]), delta = 0.0, scur = 0.0, sdelta = 0.0;
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
		if (a.drift) {
			const gvtm_drift_state in = a.drift[utt];
			ds.seed = settle(in.seed); ds.x1 = settle(in.x1); ds.x2 = settle(in.x2); ds.y1 = settle(in.y1); ds.y2 = settle(in.y2);
		}
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
	int target_time = settle(ev[target].time_ms);
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
			// The reference walks forward from `target` until it meets an event that sets my parameter: a chain of
			// dependent reads.  Here the next three events are fetched at once, unconditionally (their addresses
			// are known), and scanned in registers; only a longer gap falls back to the one-by-one walk.
			const int64_t q1 = target + 1 < n_events ? target + 1 : n_events - 1;
			const int64_t q2 = target + 2 < n_events ? target + 2 : n_events - 1;
			const double prev_p = ev[target - 1].param[j], prev_s = ev[target - 1].special[j];
			const double p0 = ev[target].param[j], p1 = ev[q1].param[j], p2 = ev[q2].param[j];
			const double s0 = ev[target].special[j], s1 = ev[q1].special[j], s2 = ev[q2].special[j];
			const int t0 = ev[target].time_ms, t1 = ev[q1].time_ms, t2 = ev[q2].time_ms;
			// the event just passed may carry the next macro-intonation polynomial: fetched in the same round trip
			const int passed_interp = ev[target - 1].has_interp;
			const double i0 = ev[target - 1].interp[0], i1 = ev[target - 1].interp[1], i2 = ev[target - 1].interp[2],
					i3 = ev[target - 1].interp[3];
			target_time = settle(t0);
			auto next_value = [&](bool special, double v0, double v1, double v2, double& value, int& time) {
				// first event at or after `target` that sets the parameter; value stays +inf when there is none
				if (!is_empty(v0)) { value = v0; time = t0; return; }
				if (!is_empty(v1) || q1 != target + 1) { value = q1 == target + 1 ? v1 : HUGE_VAL; time = t1; return; }
				if (!is_empty(v2) || q2 != target + 2) { value = q2 == target + 2 ? v2 : HUGE_VAL; time = t2; return; }
				int64_t q = target + 3;
				value = HUGE_VAL;
				while (q < n_events) {
					value = special ? ev[q].special[j] : ev[q].param[j];
					if (!is_empty(value)) { time = ev[q].time_ms; return; }
					++q;
				}
			};
			if (!is_empty(prev_p)) { // :1035-1052
				double value;
				int time = 0;
				next_value(false, p0, p1, p2, value, time);
				delta = is_empty(value) ? 0.0 : ((value - cur) / (time - now)) * cp;
			}
			if (!is_empty(prev_s)) { // :1053-1070
				double value;
				int time = 0;
				next_value(true, s0, s1, s2, value, time);
				sdelta = is_empty(value) ? 0.0 : ((value - scur) / (time - now)) * cp;
			}
			if (j == 0 && k.macro_intonation && passed_interp) { // :1072-1084
				pa = settle(i0);
				pb = settle(i1);
				if (k.smooth_intonation) {
					pc = settle(i2);
					pd = settle(i3);
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
	if (args.batch == 0) return hipSuccess;
	hipLaunchKernelGGL(vtm_tracks_kernel, dim3(static_cast<unsigned>(args.batch)), dim3(16), 0, stream, args);
	return hipGetLastError();
}

} // namespace gvtm
