// EventList::generateOutput on the device (vtm_control_model/EventList.cpp:930-1091).
//
// Two utterances per workgroup of one wavefront, 32 lanes each; lanes 0..15 of a row walk (one lane per parameter: the
// reference's inner `for j < numParam` loops), each with its own running value + delta (double, as in the reference); lane 0
// also runs the pitch extras (drift generator, macro-intonation polynomial, mean pitch).
// The walk is a chain of dependent steps, and in rounds 1 and 2 (0.57 ms per 4096 x 80 events) most of its length was
// memory round trips at the event boundaries: the reference's forward search for the next event that sets a parameter
// is a chain of dependent reads (up to a dozen for the sparse special parameters), and on gfx9 loads and stores share one
// in-order counter (vmcnt), so every boundary also waited for the frames stored before it.  Now:
//  - a table in LDS says, per event and column, how far ahead the next event that sets the column is (built once per
//    utterance by lanes 16..47, one per column, from reads that do not depend on each other), so a search is one LDS read
//    and one read of the event found;
//  - what a boundary needs (the event just passed, the next event's time, the values found) is requested at the boundary
//    BEFORE it and looked at a whole inter-event gap later;
//  - frames collect in LDS and leave 32 at a time as one contiguous 2 KB store by all 64 lanes (one store instruction
//    in 32 frames' time instead of 32);
//  - 9.7 KB of LDS per utterance and ~90 registers: a batch of 4096 is resident at once.
// 0.31 ms per 4096 x 80 events (0.37 with one utterance per wavefront).  (Also measured: a separate writer wavefront per utterance, 0.51 ms -- two wavefronts per
// utterance halve the utterances in flight; the events themselves staged in LDS, 1.29 ms -- 24-32 KB per utterance leave 4-6
// workgroups per compute unit; four utterances per wavefront, 1.14 ms -- the rows diverge at their boundaries.)
// Bit parity with the reference: same double operations in the same order, no FMA contraction.
#include "vtm_tracks.hpp"

#include <cmath>
#include <cstddef>

namespace gvtm {

namespace {

__device__ __forceinline__ bool is_empty(double v)
{
	return v == HUGE_VAL; // Event::EMPTY_PARAMETER = +infinity (EventList.cpp:38)
}

} // namespace

// column c of an event: parameter c (c < 16) or special parameter c - 16 -- the two arrays follow each other in gvtm_event
__device__ __forceinline__ double column(const gvtm_event* e, int c)
{
	static_assert(offsetof(gvtm_event, special) == offsetof(gvtm_event, param) + 16 * sizeof(double), "param[16] and special[16] are contiguous");
	return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(e) + offsetof(gvtm_event, param) + sizeof(double) * c);
}

constexpr int kRingFrames = 32;   // frames collected in LDS before they leave as one 2 KB store
constexpr int kTableEvents = 240; // event lists up to this length get a "next event that sets column c" table in LDS
constexpr int kFar = 255;         // table entry: no event within 254 sets the column

// What the boundary at which `target` becomes T looks at, requested one boundary earlier (nothing in the walker wavefront
// stores to memory, so these loads are waited for exactly where they are used -- a whole inter-event gap later).
struct Staged {
	double prev_p, prev_s;        // event T - 1: my parameter / special parameter (set or EMPTY)
	int prev_has_interp;
	double i0, i1, i2, i3;        // event T - 1: macro-intonation polynomial
	int time_t;                   // event T: time
	double next_p, next_s;        // first event >= T that sets my parameter / special parameter: its value (EMPTY: none) ...
	int next_p_time, next_s_time; // ... and its time
};

// One utterance, walked by the lanes `l` of its row of the wavefront: lanes 0..15 one parameter each, the rest (if any) mirror
// them; all of them build the table and carry the frames out.  ring / ahead: the row's LDS.
template <int LW> // lanes of the row: 64, 32 or 16
__device__ __forceinline__ void tracks_row(const TrackArgs& a, size_t utt, int l, float (*ring)[16], unsigned char (*ahead)[32])
{
#pragma clang fp contract(off)
	const int j = l & 15; // parameter
	const TrackConstants& k = a.k;
	const gvtm_event* ev = a.events + a.event_offsets[utt];
	const int64_t n_events = a.event_offsets[utt + 1] - a.event_offsets[utt];
	float* out = a.params + utt * a.max_frames * 16;
	if (n_events < 2) { // EventList.cpp:932-934
		if (l == 0 && a.frame_counts) a.frame_counts[utt] = 0;
		return;
	}
	const bool tabled = n_events <= kTableEvents;
	// The table: lanes 0..31, one per column, walk the events backwards (the reads do not depend on each other: param[16] and
	// special[16] are 32 consecutive doubles of an event).
	if (tabled) {
		const int ne = static_cast<int>(n_events);
		for (int c = l; c < 32; c += LW) {
			int last = ne + kFar; // none so far
			ahead[ne][c] = kFar;
#pragma unroll 8
			for (int q = ne - 1; q >= 0; --q) {
				if (!is_empty(column(ev + q, c))) last = q;
				const int d = last - q;
				ahead[q][c] = static_cast<unsigned char>(d < kFar ? d : kFar);
			}
		}
	}
	// ---- the walk (the whole row runs it: lanes 16.. mirror lanes 0..15 and never write a frame)
	const bool walker = l < 16;
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
	// the table is complete: one wavefront, whose LDS operations execute in order -- the compiler only has to keep them so
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");

	// first event >= q that sets column c of my lane (c = j: parameter, 16 + j: special), n_events if none; the reference
	// walks there event by event (:1037-1046, :1055-1064), and so does this for lists too long for the table
	auto first_set = [&](int64_t q, int c) -> int64_t {
		if (tabled) {
			const int d = ahead[q][c];
			if (d < kFar) return q + d;
			q += kFar - 1; // nothing in [q, q + 254): on from there
			if (q >= n_events) return n_events;
		}
		while (q < n_events && is_empty(column(ev + q, c))) ++q;
		return q;
	};
	auto stage = [&](int64_t T) { // T <= n_events
		Staged st;
		const gvtm_event* pe = ev + (T - 1);
		st.prev_p = pe->param[j];
		st.prev_s = pe->special[j];
		st.prev_has_interp = pe->has_interp;
		st.i0 = pe->interp[0]; st.i1 = pe->interp[1]; st.i2 = pe->interp[2]; st.i3 = pe->interp[3];
		const int64_t Tc = T < n_events ? T : n_events - 1;
		st.time_t = ev[Tc].time_ms;
		const int64_t qp = first_set(Tc, j), qs = first_set(Tc, 16 + j);
		st.next_p = HUGE_VAL; st.next_s = HUGE_VAL; st.next_p_time = 0; st.next_s_time = 0;
		if (qp < n_events) { st.next_p = ev[qp].param[j]; st.next_p_time = ev[qp].time_ms; }
		if (qs < n_events) { st.next_s = ev[qs].special[j]; st.next_s_time = ev[qs].time_ms; }
		return st;
	};

	// the frames [first, end) of the ring leave: 128 float4 over the row's lanes (frames beyond the rows' length are dropped)
	auto flush = [&](size_t first, size_t end) {
		if (end > a.max_frames) end = a.max_frames;
		const float4* src = reinterpret_cast<const float4*>(&ring[0][0]);
		float4* dst = reinterpret_cast<float4*>(out + first * 16);
		for (int q = l; q < kRingFrames * 4; q += LW) {
			if (first + static_cast<size_t>(q >> 2) < end) dst[q] = src[q];
		}
	};

	int64_t target = 1;
	int target_time = ev[1].time_ms;
	Staged st = stage(2); // the first boundary makes target 2
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
		if (walker) ring[n % kRingFrames][j] = p;
		++n;
		if (n % kRingFrames == 0) flush(n - kRingFrames, n); // 32 frames x 16 floats = 2 KB contiguous in the output

		if (delta != 0.0) cur += delta;
		if (sdelta != 0.0) scur += sdelta;
		now += cp;
		if (now >= target_time) {
			if (++target == n_events) break;
			// what the reference finds by walking forward from `target` was requested at the previous boundary
			const Staged s0 = st;
			st = stage(target + 1); // for the next boundary
			target_time = s0.time_t;
			if (!is_empty(s0.prev_p)) { // :1035-1052
				delta = is_empty(s0.next_p) ? 0.0 : ((s0.next_p - cur) / (s0.next_p_time - now)) * cp;
			}
			if (!is_empty(s0.prev_s)) { // :1053-1070
				sdelta = is_empty(s0.next_s) ? 0.0 : ((s0.next_s - scur) / (s0.next_s_time - now)) * cp;
			}
			if (j == 0 && k.macro_intonation && s0.prev_has_interp) { // :1072-1084: the event just passed carries the next polynomial
				pa = s0.i0;
				pb = s0.i1;
				if (k.smooth_intonation) {
					pc = s0.i2;
					pd = s0.i3;
				}
			}
		}
	}
	if (n % kRingFrames != 0) flush(n - n % kRingFrames, n);
	if (l == 0) {
		if (a.frame_counts) a.frame_counts[utt] = static_cast<int32_t>(n);
		if (a.drift) a.drift[utt] = ds;
	}
}

// ROWS utterances per workgroup of one wavefront (64 / ROWS lanes each)
template <int ROWS>
__global__ __launch_bounds__(64) void vtm_tracks_kernel(const TrackArgs a)
{
	__shared__ __attribute__((aligned(16))) float ring[ROWS][kRingFrames][16];
	// ahead[q][c]: how many events after q the first one >= q that sets column c is (0..15 parameters, 16..31 special
	// parameters); kFar: none within reach.  7.7 KB + the 2 KB above per utterance.
	__shared__ unsigned char ahead[ROWS][kTableEvents + 1][32];
	const int tid = threadIdx.x;
	const int row = tid / (64 / ROWS), l = tid % (64 / ROWS);
	const size_t utt = static_cast<size_t>(blockIdx.x) * ROWS + row;
	if (utt < a.batch) tracks_row<64 / ROWS>(a, utt, l, ring[row], ahead[row]);
}

hipError_t launch_tracks(const TrackArgs& args, hipStream_t stream)
{
	if (args.batch == 0) return hipSuccess;
	// Utterances per wavefront.  Two: the per-frame instructions serve two utterances (the walk fills 16 lanes, the wavefront
	// has 64), a boundary's instructions run when either row is at one; half the wavefronts in flight (19.4 KB of LDS per
	// workgroup: eight per compute unit).  4096 x 80 events: one / two / four rows 0.37-0.40 / 0.31 / 0.33 ms.
#ifndef GVTM_TRACK_ROWS
#define GVTM_TRACK_ROWS 2
#endif
	constexpr int kRows = GVTM_TRACK_ROWS;
	hipLaunchKernelGGL(vtm_tracks_kernel<kRows>, dim3(static_cast<unsigned>((args.batch + kRows - 1) / kRows)), dim3(64), 0, stream, args);
	return hipGetLastError();
}

} // namespace gvtm
