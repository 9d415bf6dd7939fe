// CDNA4 (gfx950) kernels of the batched vocal-tract model.
//
// One workgroup (256 threads = 4 wavefronts) owns one utterance and walks it in
// chunks of kChunk internal-rate steps.  Inside a chunk the reference's per-sample
// object graph (VocalTractModel0::execSynthesisStep, vtm/VocalTractModel0.h:396-445)
// is re-cut by *dependency structure* instead of by class:
//
//   P1  interpolate   16 lanes, one per parameter: the float32 running sum of
//                     Controller::synthesize (vtm_control_model/Controller.cpp:294-311)
//   P2  convert       one lane per step: pitch->phase increment, dB->amplitude,
//                     radii->junction coefficients, frication taps, band-pass design
//                     (VocalTractModel0.h:399-404, :484-552; BandpassFilter.h:91-110)
//   P3  scan          one lane: the two truly serial scalar recurrences, oscillator
//                     phase (WavetableGlottalSource.h:196-199, :265-272) and the noise
//                     generator (NoiseSource.h:40-44, NoiseFilter.h:63-68)
//   P4  source        one lane per half-step wavetable lookup, then one lane per step:
//                     49-tap FIR, breathiness / cross-mix (WavetableGlottalSource.h:212-235,
//                     WavetableGlottalSourceFIRFilter.h:276-304, VocalTractModel0.h:416-438)
//   P5  tube          one wavefront, ONE LANE PER TUBE SECTION (10 oropharynx + 6 nasal =
//                     one 16-lane DPP row): scattering junctions with row_shr/row_shl
//                     neighbour exchange, mouth/nose reflection+radiation filters on the
//                     two end lanes, band-pass and throat IIRs (VocalTractModel0.h:565-661)
//   P6  resample      one lane per OUTPUT sample: the Kaiser-sinc polyphase converter
//                     evaluated feed-forward from the 16.16 time register
//                     (SampleRateConverter.h:295-416), coalesced float32 stores
//
// Everything between the parameter frames (HBM in) and the audio samples (HBM out)
// lives in LDS; there is no intermediate global traffic.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "vtm_design.hpp"
#include "vtm_kernels.hpp"

namespace gvtm {

namespace {

// ----------------------------------------------------------------------------------
// cross-lane helpers (one tube section per lane; a DPP row is 16 lanes)

constexpr int kDppRowShr1 = 0x111; // lane i <- lane i-1 within the row, 0 shifted in
constexpr int kDppRowShl1 = 0x101; // lane i <- lane i+1 within the row, 0 shifted in

__device__ __forceinline__ float from_left(float v)
{
	return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), kDppRowShr1, 0xF, 0xF, true));
}
__device__ __forceinline__ float from_right(float v)
{
	return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), kDppRowShl1, 0xF, 0xF, true));
}
__device__ __forceinline__ double from_left(double v)
{
	const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), kDppRowShr1, 0xF, 0xF, true);
	const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), kDppRowShr1, 0xF, 0xF, true);
	return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double from_right(double v)
{
	const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), kDppRowShl1, 0xF, 0xF, true);
	const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), kDppRowShl1, 0xF, 0xF, true);
	return __hiloint2double(hi, lo);
}
template <int LANE>
__device__ __forceinline__ float lane_value(float v)
{
	return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), LANE));
}
template <int LANE>
__device__ __forceinline__ double lane_value(double v)
{
	const int lo = __builtin_amdgcn_readlane(__double2loint(v), LANE);
	const int hi = __builtin_amdgcn_readlane(__double2hiint(v), LANE);
	return __hiloint2double(hi, lo);
}

// Util::amplitude60dB (vtm/VTMUtil.h:48-67)
__device__ __forceinline__ double amplitude_60db_dev(double db)
{
	if (db <= 0.0) return 0.0;
	if (db == 60.0) return 1.0;
	return exp10((db - 60.0) * (1.0 / 20.0));
}

// NoiseSource::getSample (vtm/NoiseSource.h:40-44): seed = frac(seed * 377).  The sequence is
// chaotic, so the product has to be rounded to double before the integer part is removed,
// exactly as the reference does — a contracted fma(seed, 377, -floor(p)) decorrelates the
// sequence within a few steps (SURVEY.md hard part 5).
__device__ __forceinline__ double noise_advance(double seed)
{
#pragma clang fp contract(off)
	const double product = seed * 377.0;
	return product - floor(product);
}

// One step of Controller::synthesize's float32 running sum (Controller.cpp:308-310); kept
// out of FMA contraction so that the per-step parameter values are bit-exact.
__device__ __forceinline__ float interp_delta(float next, float cur, float coef)
{
#pragma clang fp contract(off)
	const float diff = next - cur;
	return diff * coef;
}
__device__ __forceinline__ float interp_advance(float cur, float delta)
{
#pragma clang fp contract(off)
	return cur + delta;
}

template <typename TT>
struct SrcTap {
	TT h, dh;
};

// Record of per-step quantities handed from the parallel phases to the tube wavefront.
//   [0..15]   coefficient of the junction on the LEFT of section lane L
//             (slot 10 carries the mouth coefficient so that lane 9 receives it from its right)
//   [16..23]  frication taps FC1..FC8 (sections S3..S10 = lanes 2..9)
//   [24..26]  tube input, band-pass input, throat input   ([24],[25] hold ax, ah1 until P4)
//   [27..29]  band-pass b0, a1, a2
//   [30..31]  3-way junction alpha (left == right), alpha upper
constexpr int kRec = 32;
constexpr int kRecTap = 16;
constexpr int kRecU = 24, kRecSig = 25, kRecThr = 26, kRecB0 = 27, kRecA1 = 28, kRecA2 = 29, kRecAlr = 30, kRecAu = 31;

template <typename TT, typename ST>
struct Smem {
	double* wavetable; // [512]
	double* fir;       // [kMaxFirTaps]
	SrcTap<ST>* src;   // [3328]
	float* prm;        // [kChunk][16]
	TT* rec;           // [kChunk][kRec]
	double* inc;       // [kChunk]
	double* pos;       // [2*kChunk]
	double* noise;     // [kChunk]
	TT* w;             // [kMaxFirTaps + 2*kChunk]
	ST* x;             // [kXCap]
	float* red;        // [kBlock] reduction scratch
};

template <typename TT, typename ST>
__device__ __forceinline__ Smem<TT, ST> carve(unsigned char* base)
{
	Smem<TT, ST> s;
	size_t off = 0;
	auto take = [&](size_t bytes) {
		unsigned char* p = base + off;
		off += (bytes + 15) & ~size_t(15);
		return p;
	};
	s.wavetable = reinterpret_cast<double*>(take(sizeof(double) * kWavetableLength));
	s.fir = reinterpret_cast<double*>(take(sizeof(double) * kMaxFirTaps));
	s.src = reinterpret_cast<SrcTap<ST>*>(take(sizeof(SrcTap<ST>) * kSrcFilterLength));
	s.prm = reinterpret_cast<float*>(take(sizeof(float) * kChunk * 16));
	s.rec = reinterpret_cast<TT*>(take(sizeof(TT) * kChunk * kRec));
	s.inc = reinterpret_cast<double*>(take(sizeof(double) * kChunk));
	s.pos = reinterpret_cast<double*>(take(sizeof(double) * 2 * kChunk));
	s.noise = reinterpret_cast<double*>(take(sizeof(double) * kChunk));
	s.w = reinterpret_cast<TT*>(take(sizeof(TT) * (kMaxFirTaps + 2 * kChunk)));
	s.x = reinterpret_cast<ST*>(take(sizeof(ST) * kXCap));
	s.red = reinterpret_cast<float*>(take(sizeof(float) * kBlock));
	return s;
}

// Glottal wavetable entry i for the current source amplitude.  With tn_delta == 0 (every
// shipped voice) the table is static; otherwise WavetableGlottalSource::setup
// (vtm/WavetableGlottalSource.h:162-184) rewrites the falling part as a pure function of
// the amplitude, which is evaluated here in closed form.
__device__ __forceinline__ double wavetable_entry(const double* table, const DeviceConstants& k, unsigned i, double ax)
{
	if (k.tn_delta == 0.0 || k.waveform != 0 || i < k.table_div1 || i >= k.table_div2) return table[i];
	double new_div2 = static_cast<double>(k.table_div2) - rint(ax * k.tn_delta);
	new_div2 = new_div2 > 0.0 ? new_div2 : 0.0;
	if (i >= static_cast<unsigned>(new_div2)) return 0.0;
	const double inv = 1.0 / (new_div2 - static_cast<double>(k.table_div1));
	const double x = static_cast<double>(i - k.table_div1) * inv;
	return 1.0 - (x * x);
}

} // namespace

size_t synth_lds_bytes(bool mixed)
{
	auto al = [](size_t b) { return (b + 15) & ~size_t(15); };
	const size_t tt = sizeof(double);
	const size_t st = mixed ? sizeof(float) : sizeof(double);
	size_t n = 0;
	n += al(sizeof(double) * kWavetableLength);
	n += al(sizeof(double) * kMaxFirTaps);
	n += al(2 * st * kSrcFilterLength);
	n += al(sizeof(float) * kChunk * 16);
	n += al(tt * kChunk * kRec);
	n += al(sizeof(double) * kChunk);
	n += al(sizeof(double) * 2 * kChunk);
	n += al(sizeof(double) * kChunk);
	n += al(tt * (kMaxFirTaps + 2 * kChunk));
	n += al(st * kXCap);
	n += al(sizeof(float) * kBlock);
	return n;
}

template <typename TT, typename ST, int D>
__global__ __launch_bounds__(kBlock) void vtm_synth_kernel(const SynthArgs a)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
	const Smem<TT, ST> sm = carve<TT, ST>(smem_raw);
	const DeviceConstants& k = a.k;

	const int tid = threadIdx.x;
	const int lane = tid & 63;
	const int wave = tid >> 6;
	const size_t utt = blockIdx.x;

	const int frames = a.frame_counts ? a.frame_counts[utt] : static_cast<int>(a.max_frames);
	const uint64_t steps = static_cast<uint64_t>(frames > 0 ? frames : 0) * k.control_steps;
	const float* __restrict__ P = a.params + utt * a.max_frames * 16;
	float* __restrict__ out = a.audio + utt * a.audio_stride;
	const int taps = k.fir_taps;
	const int hist_w = taps - 1;
	const int hist_x = 2 * k.pad - 1;

	// ---- stage the design tables in LDS, clear the histories
	for (int i = tid; i < kWavetableLength; i += kBlock) sm.wavetable[i] = a.wavetable[i];
	for (int i = tid; i < kMaxFirTaps; i += kBlock) sm.fir[i] = i < taps ? a.fir[i] : 0.0;
	for (int i = tid; i < kSrcFilterLength; i += kBlock) {
		sm.src[i].h = static_cast<ST>(a.src_h[i]);
		sm.src[i].dh = static_cast<ST>(a.src_dh[i]);
	}
	for (int i = tid; i < kMaxFirTaps + 2 * kChunk; i += kBlock) sm.w[i] = TT(0);
	for (int i = tid; i < kXCap; i += kBlock) sm.x[i] = ST(0);

	// ---- persistent per-role state
	// P1: lanes 0..15 of wave 0 interpolate parameter `tid`
	float ip_cur = 0.f, ip_delta = 0.f, ip_this = 0.f, ip_next = 0.f, ip_next2 = 0.f;
	unsigned ip_j = 0;
	int ip_frame = 0;
	if (tid < 16 && frames > 0) {
		ip_this = P[tid];
		ip_next = P[static_cast<size_t>(frames > 1 ? 1 : 0) * 16 + tid];
		ip_next2 = P[static_cast<size_t>(frames > 2 ? 2 : frames - 1) * 16 + tid];
	}
	// P3: oscillator phase and noise generator
	double sc_pos = 0.0, sc_seed = 0.7892347, sc_prev = 0.0;
	// P5: tube state, one section per lane (all four DPP rows of the wavefront mirror row 0)
	const int L = lane & 15;
	TT top[D], bot[D];
#pragma unroll
	for (int i = 0; i < D; ++i) top[i] = bot[i] = TT(0);
	TT bp_x1 = 0, bp_x2 = 0, bp_y1 = 0, bp_y2 = 0; // BandpassFilter state
	TT refl_y1 = 0, rad_x1 = 0, rad_y1 = 0;        // end-lane filters
	TT thr_y1 = 0;
	const bool lane_first = (L == 0);
	const bool lane_3way_right = (L == 3);              // S4: its right side is the 3-way junction
	const bool lane_3way_left = (L == 4) || (L == 10);  // S5 and N1: their left side is the 3-way junction
	const bool lane_end = (L == 9) || (L == 15);        // S10 (mouth) and N6 (nose)
	const int tap_slot = (L >= 2 && L <= 9) ? (kRecTap + L - 2) : 4; // slot 4 is always zero
	const TT damping = static_cast<TT>(k.damping);
	const TT refl_b0 = static_cast<TT>(L == 9 ? k.mouth_b0_refl : (L == 15 ? k.nose_b0_refl : 0.0));
	const TT refl_a1 = static_cast<TT>(L == 9 ? k.mouth_a1_refl : (L == 15 ? k.nose_a1_refl : 0.0));
	const TT rad_a = static_cast<TT>(L == 9 ? k.mouth_a_rad : (L == 15 ? k.nose_a_rad : 0.0));
	const TT nose_k = static_cast<TT>(k.nasal_k[5]);
	const TT thr_b0 = static_cast<TT>(k.throat_b0), thr_a1 = static_cast<TT>(k.throat_a1), thr_gain = static_cast<TT>(k.throat_gain);
	// P6
	uint64_t k_next = 0;
	float my_max = 0.f;

	const uint64_t n_chunks = steps == 0 ? 1 : (steps + kChunk - 1) / kChunk;
	__syncthreads();

	// diagnostics: shader cycles per phase, accumulated by thread 0 right after each barrier
	unsigned long long ph_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	unsigned long long ph_prev = 0;
	const bool stamping = a.phase_cycles != nullptr && tid == 0;
	if (stamping) ph_prev = clock64();
#define GVTM_STAMP(i) \
	if (stamping) { const unsigned long long now_ = clock64(); ph_acc[i] += now_ - ph_prev; ph_prev = now_; }

	for (uint64_t c = 0; c < n_chunks; ++c) {
		const uint64_t n0 = c * kChunk;
		const int valid = static_cast<int>((steps - n0) < static_cast<uint64_t>(kChunk) ? (steps - n0) : kChunk);
		const bool last = (c + 1 == n_chunks);

		// ---------------- P1: parameter interpolation (float32 running sum) ----------------
		if (tid < 16) {
			for (int s = 0; s < valid; ++s) {
				if (ip_j == 0) {
					// Controller.cpp:297-300: restart from the frame value, delta towards the next frame
					ip_cur = ip_this;
					ip_delta = interp_delta(ip_next, ip_cur, k.interp_coef);
				}
				sm.prm[s * 16 + tid] = ip_cur;
				ip_cur = interp_advance(ip_cur, ip_delta); // Controller.cpp:308-310
				if (++ip_j == k.control_steps) {
					ip_j = 0;
					++ip_frame;
					ip_this = ip_next;
					ip_next = ip_next2;
					const int f = ip_frame + 2;
					ip_next2 = P[static_cast<size_t>(f < frames ? f : frames - 1) * 16 + tid];
				}
			}
		}
		__syncthreads();

		GVTM_STAMP(0)

		// ---------------- P2: per-step parameter conversion ----------------
		if (tid < valid) {
			const float* pr = sm.prm + tid * 16;
			TT* rec = sm.rec + tid * kRec;
			// Util::frequency (VTMUtil.h:74-84) -> oscillator increment (WavetableGlottalSource.h:198, :218)
			const double f0 = 220.0 * exp2((static_cast<double>(pr[0]) + 3.0) * (1.0 / 12.0));
			sm.inc[tid] = (f0 / 2.0) * k.basic_increment;
			rec[kRecU] = static_cast<TT>(amplitude_60db_dev(pr[1]));   // ax, consumed by P4
			rec[kRecSig] = static_cast<TT>(amplitude_60db_dev(pr[2])); // ah1, consumed by P4
			// setAllParameters (VocalTractModel0.h:698-716): radius scaling and floor
			double r2[8];
#pragma unroll
			for (int i = 0; i < 8; ++i) {
				double r = static_cast<double>(pr[7 + i]) * k.radius_coef[i];
				r = r > 0.01 ? r : 0.01;
				r2[i] = r * r;
			}
			const double velum2 = static_cast<double>(pr[15]) * static_cast<double>(pr[15]);
			// calculateTubeCoefficients (VocalTractModel0.h:484-512)
			double kj[8];
#pragma unroll
			for (int i = 0; i < 7; ++i) kj[i] = (r2[i] - r2[i + 1]) / (r2[i] + r2[i + 1]);
			kj[7] = (r2[7] - k.aperture_radius2) / (r2[7] + k.aperture_radius2);
			const double asum = 2.0 / (r2[3] + r2[3] + velum2);
			const double nk1 = (velum2 - k.nasal_r2_sq) / (velum2 + k.nasal_r2_sq);
			rec[0] = TT(-1);                       // S1: top = bottom * damping + input
			rec[1] = static_cast<TT>(kj[0]);       // S1|S2
			rec[2] = static_cast<TT>(kj[1]);       // S2|S3
			rec[3] = static_cast<TT>(kj[2]);       // S3|S4
			rec[4] = TT(0);                        // S4|S5 is the 3-way junction (slot doubles as the zero tap)
			rec[5] = static_cast<TT>(kj[3]);       // S5|S6
			rec[6] = TT(0);                        // S6|S7 lie in one region: pure delay
			rec[7] = static_cast<TT>(kj[4]);       // S7|S8
			rec[8] = static_cast<TT>(kj[5]);       // S8|S9
			rec[9] = static_cast<TT>(kj[6]);       // S9|S10
			rec[10] = static_cast<TT>(kj[7]);      // S10|air, fetched by lane 9 from its right
			rec[11] = static_cast<TT>(nk1);        // N1|N2
			rec[12] = static_cast<TT>(k.nasal_k[1]);
			rec[13] = static_cast<TT>(k.nasal_k[2]);
			rec[14] = static_cast<TT>(k.nasal_k[3]);
			rec[15] = static_cast<TT>(k.nasal_k[4]);
			rec[kRecAlr] = static_cast<TT>(asum * r2[3]);
			rec[kRecAu] = static_cast<TT>(asum * velum2);
			// setFricationTaps (VocalTractModel0.h:524-552)
			{
				const double amp = amplitude_60db_dev(pr[3]);
				const double fpos = pr[4];
				const int ipart = static_cast<int>(fpos);
				const double complement = fpos - ipart;
				const double remainder = 1.0 - complement;
#pragma unroll
				for (int i = 0; i < 8; ++i) {
					double t = 0.0;
					if (i == ipart) t = remainder * amp;
					else if (i == ipart + 1) t = complement * amp;
					rec[kRecTap + i] = static_cast<TT>(t);
				}
			}
			// BandpassFilter::update (BandpassFilter.h:91-110)
			{
				const double pi = 3.14159265358979323846;
				const double tan_v = tan(pi * static_cast<double>(pr[6]) * k.bp_T);
				const double cos_v = cos(2.0 * pi * static_cast<double>(pr[5]) * k.bp_T);
				const double a2 = (1.0 - tan_v) / (1.0 + tan_v);
				const double a1 = -(1.0 + a2) * cos_v;
				const double b0 = 0.5 - 0.5 * a2;
				rec[kRecB0] = static_cast<TT>(b0);
				rec[kRecA1] = static_cast<TT>(a1);
				rec[kRecA2] = static_cast<TT>(a2);
			}
		}
		__syncthreads();

		GVTM_STAMP(1)

		// ---------------- P3: serial scalar recurrences ----------------
		if (tid == 64) {
			for (int s = 0; s < valid; ++s) {
				const double inc = sm.inc[s];
#pragma unroll
				for (int h = 0; h < 2; ++h) {
					double p = sc_pos + inc;
					if (p > 511.0) p -= 512.0; // mod0, WavetableGlottalSource.h:265-272
					sc_pos = p;
					sm.pos[2 * s + h] = p;
				}
				sc_seed = noise_advance(sc_seed);
				const double white = sc_seed - 0.5;
				sm.noise[s] = white + sc_prev; // NoiseFilter::filter
				sc_prev = white;
			}
		}
		__syncthreads();

		GVTM_STAMP(2)

		// ---------------- P4a: wavetable lookups at the 2x oversampled rate ----------------
		for (int h = tid; h < 2 * valid; h += kBlock) {
			const double p = sm.pos[h];
			// a position in (-1, 0) truncates towards zero, as the reference's cast does
			const unsigned lower = static_cast<unsigned>(static_cast<int>(p));
			const unsigned upper = (lower + 1 > 511u) ? lower + 1 - 512u : lower + 1;
			const double ax = static_cast<double>(sm.rec[(h >> 1) * kRec + kRecU]);
			const double wl = wavetable_entry(sm.wavetable, k, lower, ax);
			const double wu = wavetable_entry(sm.wavetable, k, upper, ax);
			sm.w[hist_w + h] = static_cast<TT>(wl + ((p - static_cast<double>(lower)) * (wu - wl)));
		}
		__syncthreads();

		GVTM_STAMP(3)

		// ---------------- P4b: FIR decimator + source mixing ----------------
		if (tid < valid) {
			const TT* wp = sm.w + hist_w + 2 * tid + 1; // newest sample of this step
			TT acc = 0;
			for (int i = 0; i < taps; ++i) acc += wp[-i] * static_cast<TT>(sm.fir[i]);
			TT* rec = sm.rec + tid * kRec;
			const double ax = rec[kRecU], ah1 = rec[kRecSig];
			const double lp = sm.noise[tid];
			double pulse = acc;
			const double pulsed_noise = lp * pulse;
			pulse = ax * ((pulse * (1.0 - k.breathiness)) + (pulsed_noise * k.breathiness));
			double signal;
			if (k.modulation) {
				double cm = ax * k.crossmix_factor;
				cm = cm < 1.0 ? cm : 1.0;
				signal = (pulsed_noise * cm) + (lp * (1.0 - cm));
			} else {
				signal = lp;
			}
			rec[kRecU] = static_cast<TT>((pulse + (ah1 * signal)) * 0.125);
			rec[kRecSig] = static_cast<TT>(signal);
			rec[kRecThr] = static_cast<TT>(pulse * 0.125);
			if (a.debug_taps) {
				double* t = a.debug_taps + (utt * a.max_frames * k.control_steps + n0 + tid) * 8;
				t[0] = rec[kRecU]; t[1] = rec[kRecSig]; t[2] = rec[kRecThr]; t[3] = acc; t[4] = lp;
				t[5] = sm.pos[2 * tid]; t[6] = sm.pos[2 * tid + 1];
			}
		}
		__syncthreads();

		GVTM_STAMP(4)

		// ---------------- P5: the waveguide, one section per lane ----------------
		if (wave == 0) {
			static_assert(kChunk % D == 0, "chunk must be a multiple of the section delay");
			for (int s0 = 0; s0 < valid; s0 += D) {
#pragma unroll
				for (int ph = 0; ph < D; ++ph) {
					const int s = s0 + ph;
					if (s < valid) {
						const TT* rec = sm.rec + s * kRec;
						const TT kl = rec[L];
						const TT tapc = rec[tap_slot];
						const TT u = rec[kRecU], sig = rec[kRecSig], thr = rec[kRecThr];
						const TT b0 = rec[kRecB0], a1 = rec[kRecA1], a2 = rec[kRecA2];
						const TT alr = rec[kRecAlr], au = rec[kRecAu];
						// BandpassFilter::filter (BandpassFilter.h:114-122)
						const TT fric = b0 * (sig - bp_x2) - a1 * bp_y1 - a2 * bp_y2;
						bp_x2 = bp_x1; bp_x1 = sig; bp_y2 = bp_y1; bp_y1 = fric;

						const TT T = top[ph], B = bot[ph]; // values written D steps ago
						const TT Tl = from_left(T);        // top of the section on my left
						const TT Br = from_right(B);       // bottom of the section on my right
						TT kr = from_right(kl);
						kr = (L == 15) ? nose_k : kr;
						// 3-way junction S4 / S5 / N1 (VocalTractModel0.h:595-604)
						const TT jp = (alr * lane_value<3>(T)) + (alr * lane_value<4>(B)) + (au * lane_value<10>(B));
						const TT inj = lane_first ? u : tapc * fric;
						// my left junction produces my new top
						const TT dl = kl * (Tl - B);
						TT tn = ((Tl + dl) * damping) + inj;
						const TT tn3 = ((jp - B) * damping) + inj;
						tn = lane_3way_left ? tn3 : tn;
						// my right junction produces my new bottom
						const TT dr = kr * (T - Br);
						TT bn = (Br + dr) * damping;
						const TT bn3 = (jp - T) * damping;
						bn = lane_3way_right ? bn3 : bn;
						// open ends: reflection (ReflectionFilter.h:66-76) and radiation (RadiationFilter.h:68-79)
						const TT ry = refl_b0 * (kr * T) - refl_a1 * refl_y1;
						refl_y1 = ry;
						bn = lane_end ? damping * ry : bn;
						const TT ox = (TT(1) + kr) * T;
						const TT oy = rad_a * ox + (-rad_a) * rad_x1 - (-rad_a) * rad_y1;
						rad_x1 = ox; rad_y1 = oy;
						TT sample = lane_value<9>(oy) + lane_value<15>(oy);
						// Throat::process (Throat.h:76-85)
						const TT ty = thr_b0 * thr - thr_a1 * thr_y1;
						thr_y1 = ty;
						sample += ty * thr_gain;
						top[ph] = tn;
						bot[ph] = bn;
						if (lane == 0) {
							sm.x[hist_x + s] = static_cast<ST>(sample);
							if (a.debug_taps) a.debug_taps[(utt * a.max_frames * k.control_steps + n0 + s) * 8 + 7] = sample;
						}
					}
				}
			}
			if (last) {
				// flushBuffer (SampleRateConverter.h:462-471): 2*pad zero samples follow the utterance
				for (int i = lane; i < 2 * k.pad; i += 64) sm.x[hist_x + valid + i] = ST(0);
			}
		}
		__syncthreads();

		GVTM_STAMP(5)

		// ---------------- P6: sample-rate conversion, one lane per output sample ----------------
		{
			const uint64_t filled = last ? steps + 2ull * k.pad : n0 + kChunk;
			const uint64_t k_end = ((filled << 16) + k.time_inc - 1) / k.time_inc; // outputs with P_k < filled
			const ST* xb = sm.x + hist_x - static_cast<int64_t>(n0); // xb[n] = internal sample n
			for (uint64_t ko = k_next + tid; ko < k_end; ko += kBlock) {
				const uint64_t t = ko * k.time_inc;
				const int64_t Pk = static_cast<int64_t>(t >> 16);
				const unsigned frac = static_cast<unsigned>(t & 0xFFFFu);
				// ring index p of the reference holds internal sample p - pad
				const ST* v = xb + (Pk - k.pad);
				ST acc = 0;
				if (k.upsampling) {
					const unsigned l = frac >> 8, m = frac & 0xFFu;
					const ST interp = static_cast<ST>(m) / ST(256);
#pragma unroll
					for (int j = 0; j < kSrcZeroCrossings; ++j) {
						const SrcTap<ST> c = sm.src[l + 256 * j];
						acc += v[-j] * (c.h + (c.dh * interp));
					}
					const unsigned nfrac = (~frac) & 0xFFFFu;
					const unsigned l2 = nfrac >> 8, m2 = nfrac & 0xFFu;
					const ST interp2 = static_cast<ST>(m2) / ST(256);
#pragma unroll
					for (int j = 0; j < kSrcZeroCrossings; ++j) {
						const SrcTap<ST> c = sm.src[l2 + 256 * j];
						acc += v[1 + j] * (c.h + (c.dh * interp2));
					}
				} else {
					unsigned ph = static_cast<unsigned>(rint(static_cast<double>(frac) * k.src_ratio));
					unsigned ii;
					int j = 0;
					while ((ii = (ph >> 8)) < static_cast<unsigned>(kSrcFilterLength)) {
						const SrcTap<ST> c = sm.src[ii];
						acc += v[-j] * (c.h + (c.dh * (static_cast<ST>(ph & 0xFFu) / ST(256))));
						++j;
						ph += k.phase_inc;
					}
					ph = static_cast<unsigned>(rint(static_cast<double>((~frac) & 0xFFFFu) * k.src_ratio));
					j = 0;
					while ((ii = (ph >> 8)) < static_cast<unsigned>(kSrcFilterLength)) {
						const SrcTap<ST> c = sm.src[ii];
						acc += v[1 + j] * (c.h + (c.dh * (static_cast<ST>(ph & 0xFFu) / ST(256))));
						++j;
						ph += k.phase_inc;
					}
				}
				const float y = static_cast<float>(acc);
				if (ko < a.audio_stride) out[ko] = y;
				my_max = fmaxf(my_max, fabsf(y));
			}
			k_next = k_end;
		}
		__syncthreads();

		GVTM_STAMP(6)

		// ---------------- carry histories into the next chunk ----------------
		if (!last) {
			for (int i = tid; i < hist_x; i += kBlock) sm.x[i] = sm.x[kChunk + i];
			// FIR history: the taps-1 newest half-step samples move to the front
			TT keep = 0;
			if (tid < hist_w) keep = sm.w[2 * kChunk + tid];
			__syncthreads();
			if (tid < hist_w) sm.w[tid] = keep;
		}
	}

	GVTM_STAMP(7)
#undef GVTM_STAMP
	if (stamping) {
		for (int i = 0; i < 8; ++i) a.phase_cycles[utt * 8 + i] = ph_acc[i];
	}

	// ---- per-utterance peak (for Util::calculateOutputScale) and sample count
	sm.red[tid] = my_max;
	__syncthreads();
	for (int stride = kBlock / 2; stride > 0; stride >>= 1) {
		if (tid < stride) sm.red[tid] = fmaxf(sm.red[tid], sm.red[tid + stride]);
		__syncthreads();
	}
	if (tid == 0) {
		if (a.maxabs) a.maxabs[utt] = sm.red[0];
		if (a.out_counts) a.out_counts[utt] = static_cast<int64_t>(k_next);
	}
}

// Controller::writeOutputToBuffer / writeOutputToFile (Controller.cpp:315-340): scale by
// 0.95 / max|x| (Util::calculateOutputScale, VTMUtil.cpp:48-67); the int16 form rounds as
// WAVEFileWriter::writeSample does (WAVEFileWriter.cpp:122-125).  HBM-bound elementwise pass.
__global__ __launch_bounds__(256) void vtm_normalize_kernel(const NormalizeArgs a)
{
	const size_t utt = blockIdx.y;
	const int64_t n = a.counts ? a.counts[utt] : static_cast<int64_t>(a.audio_stride);
	const float peak = a.maxabs[utt];
	const float scale = (peak < 1.0e-30f) ? 0.0f : 0.95f / peak;
	if (a.scales && blockIdx.x == 0 && threadIdx.x == 0) a.scales[utt] = scale;
	const float* __restrict__ in = a.audio + utt * a.audio_stride;
	const int64_t limit = n < static_cast<int64_t>(a.audio_stride) ? n : static_cast<int64_t>(a.audio_stride);
	for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < limit;
			i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
		const float v = in[i] * scale;
		if (a.out_f32) a.out_f32[utt * a.audio_stride + i] = v;
		if (a.out_i16) a.out_i16[utt * a.audio_stride + i] = static_cast<int16_t>(static_cast<int>(roundf(v * 32767.0f)));
	}
}

template <typename TT, typename ST, int D>
static hipError_t launch_one(const SynthArgs& args, size_t batch, size_t lds, hipStream_t stream)
{
	auto fn = vtm_synth_kernel<TT, ST, D>;
	hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
			static_cast<int>(lds));
	if (e != hipSuccess) return e;
	hipLaunchKernelGGL(fn, dim3(static_cast<unsigned>(batch)), dim3(kBlock), lds, stream, args);
	return hipGetLastError();
}

hipError_t launch_synth(const SynthArgs& args, size_t batch, bool mixed, hipStream_t stream)
{
	const size_t lds = synth_lds_bytes(mixed);
	const int d = args.k.section_delay;
	if (mixed) {
		switch (d) {
		case 1: return launch_one<double, float, 1>(args, batch, lds, stream);
		case 2: return launch_one<double, float, 2>(args, batch, lds, stream);
		case 3: return launch_one<double, float, 3>(args, batch, lds, stream);
		case 4: return launch_one<double, float, 4>(args, batch, lds, stream);
		}
	} else {
		switch (d) {
		case 1: return launch_one<double, double, 1>(args, batch, lds, stream);
		case 2: return launch_one<double, double, 2>(args, batch, lds, stream);
		case 3: return launch_one<double, double, 3>(args, batch, lds, stream);
		case 4: return launch_one<double, double, 4>(args, batch, lds, stream);
		}
	}
	return hipErrorInvalidValue;
}

hipError_t launch_normalize(const NormalizeArgs& args, size_t batch, hipStream_t stream)
{
	const unsigned bx = static_cast<unsigned>((args.audio_stride + 256 * 8 - 1) / (256 * 8));
	hipLaunchKernelGGL(vtm_normalize_kernel, dim3(bx > 0 ? bx : 1, static_cast<unsigned>(batch)), dim3(256), 0, stream, args);
	return hipGetLastError();
}

} // namespace gvtm
