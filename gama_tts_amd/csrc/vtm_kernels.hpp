// Launch interface between the C ABI (vtm_capi.cpp) and the kernels (vtm_kernels.hip).
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "vtm_design.hpp"

namespace gvtm {

// ---- streams (gvtm_stream_*): what an utterance carries from one launch to the next, in device memory ----
//
//   StreamHeader | scalars CT[kStreamScalars] | tube CT[lanes][2 * SectionDelay + 1] | decimator pre-roll CT[48] | ring ST[xr]
//
// i.e. exactly the reference model's members that survive a step (vtm/VocalTractModel0.h:221-252): the section delay
// lines, the filter memories, the oscillator phase, the noise seed, the decimator's and the converter's buffers with
// the converter's position (time register and pointers follow from the number of steps, SampleRateConverter.h:268-282).
// A fresh utterance (create / reset) is all zeros except the noise seed (NoiseSource.h:32-34).
struct StreamHeader {
	unsigned long long step_base; // internal steps synthesized so far
	double seed;                  // NoiseSource::seed_
	unsigned peak_bits;           // running max |sample| as float bits
	unsigned reserved_;
	unsigned long long pad_;
};
static_assert(sizeof(StreamHeader) == 32, "stream header layout");
constexpr int kStreamScalars = 12;
constexpr int kSsPos = 0, kSsPrev = 1;           // oscillator position, previous white-noise sample
constexpr int kSsBp = 2, kSsThr = 6;             // band-pass x1 x2 y1 y2, throat y1
constexpr int kSsRadMouth = 7, kSsRadNose = 9;   // radiation filters {x1, y1}
constexpr int kStreamPreRoll = 48;               // = kWPre of the kernel
enum StreamMode : int { kStreamNone = 0, kStreamPush = 1, kStreamFinish = 2 };

template <typename CT, typename ST>
struct StreamLayout {
	static constexpr size_t scalars() { return sizeof(StreamHeader); }
	static constexpr size_t tube() { return scalars() + sizeof(CT) * kStreamScalars; }
	static constexpr size_t wpre(int lanes, int words) { return tube() + sizeof(CT) * static_cast<size_t>(lanes) * words; }
	static constexpr size_t ring(int lanes, int words) { return wpre(lanes, words) + sizeof(CT) * kStreamPreRoll; }
	static constexpr size_t bytes(int lanes, int words, int xr) { return (ring(lanes, words) + sizeof(ST) * static_cast<size_t>(xr) + 15) & ~size_t(15); }
};

// Reference model 5 (VocalTractModel5, vtm/VocalTractModel5.h:523-579): what its members keep between steps -- the scans'
// phases, noise seed and low-pass memories, the recursive filter halves' memories, the last inputs of their feed-forward
// halves, every section's top / bottom flow and radiation-impedance memories, the converter's 1024-sample ring and the
// ring of converted samples the difference filter looks back into.  A fresh utterance is zeros except the Rosenberg
// source's t2 (RosenbergBGlottalSource::reset) and the noise seed.
constexpr int kS5Scan = 0;     // t, t2, seed, glottal-noise x1 y1, frication-noise x1 x2 y1 y2
constexpr int kS5Gp = 9;       // glottal low-pass y1
constexpr int kS5Bp = 10;      // band-pass y1, y2
constexpr int kS5Fo = 12;      // transmitted flow y1: mouth, nose
constexpr int kS5ValLast = 14; // last pulse value (the low-pass's x1)
constexpr int kS5FnmLast = 15; // last two band-pass inputs (x2, x1)
constexpr int kS5Carry = 17;   // last flow into mouth, nose
constexpr int kStream5Scalars = 24;
struct Stream5Layout {
	static constexpr size_t scalars() { return sizeof(StreamHeader); }
	static constexpr size_t tube() { return scalars() + sizeof(double) * kStream5Scalars; }
	static constexpr size_t ring() { return tube() + sizeof(double) * 51 * 4; }
	static constexpr size_t yring() { return ring() + sizeof(double) * kSrcRing; }
	static constexpr size_t bytes() { return (yring() + sizeof(float) * 512 + 15) & ~size_t(15); }
};

struct SynthArgs {
	DeviceConstants k;              // by value (host-side launch decisions)
	const DeviceConstants* kconst;  // the same constants in device memory (the kernel stages them in LDS)
	const float* params;         // [batch][max_frames][16]
	const int32_t* frame_counts; // [batch] or null
	float* audio;                // [batch][audio_stride]
	int64_t* out_counts;         // [batch] or null
	float* maxabs;               // [batch] or null
	const void* wavetable;       // [512]      design tables: double, or float for GVTM_PRECISION_F32
	const void* fir;             // [fir_taps]
	// the same coefficients by value: kernel arguments are read with scalar loads, so the decimator's unrolled tap loop
	// takes them from SGPRs (d: the double design, f: the float design of GVTM_PRECISION_F32)
	union FirByValue { double d[49]; float f[64]; } fir_k;
	const void* src_h;           // [3328]
	const void* src_dh;          // [3328]
	const void* noise_lp = nullptr;  // [noise_len] the noise source's low-passed samples by internal step (one-shot launches), or
	                                 // null: the scan wavefront generates them (streams, whose step count has no bound)
	unsigned long long noise_len = 0;
	size_t max_frames;
	size_t audio_stride;
	size_t batch;
	int xr;                      // internal-rate ring length per utterance (a power of two; synth_ring_length())
	unsigned char* stream = nullptr; // null, or [batch] stream states of stream_stride bytes each (layout above)
	size_t stream_stride = 0;
	int stream_mode = kStreamNone;   // StreamMode
	double* debug_taps;          // null, or [batch][max_frames*control_steps][8] per-step taps (tests only)
	unsigned long long* phase_cycles; // null, or [workgroups][16] shader cycles per role wavefront and helper stage (diagnostics only)
	const Model5Constants* k5const = nullptr; // model 5 only: its constants in device memory
};

struct NormalizeArgs {
	const float* audio;
	const int64_t* counts; // or null
	const float* maxabs;
	float* out_f32;        // or null
	int16_t* out_i16;      // or null
	float* scales;         // or null
	size_t audio_stride;
};

// rows: utterances per workgroup (1, 2, 4 or 8); synth_rows() picks it from the batch size unless `requested` names one
// precision: gvtm_precision
int synth_rows(int precision, size_t batch, int requested, int section_delay = 1);
// internal-rate ring length for a plan: the reference's BUFFER_SIZE (1024) when down-sampling, so that the flush
// overrun aliases as the reference's ring does; otherwise the smallest power of two holding two chunks, the
// resampler's history and the flush zeros
int synth_ring_length(const DeviceConstants& k, int precision, int rows);
size_t synth_lds_bytes(const DeviceConstants& k, int precision, int rows, int xr = 0 /* 0: the shape's own ring length */);
hipError_t launch_synth(const SynthArgs& args, size_t batch, int precision, int rows, hipStream_t stream);
// bytes of one utterance's stream state for a plan (ring length of the one-row shape: streams whose utterances are not in
// lockstep run one utterance per workgroup, and every shape of a stream uses that ring length)
size_t stream_state_bytes(const DeviceConstants& k, int precision, int xr);
// reference model 5 (VocalTractModel5<double,1>), fp64: rows = utterances per workgroup, 1 or 2
size_t synth5_lds_bytes(int rows = 1);
hipError_t launch_synth5(const SynthArgs& args, size_t batch, int rows, hipStream_t stream);
constexpr int kDppSelftestInts = 640;
hipError_t launch_dpp_selftest(int* d_out /* [kDppSelftestInts] */, hipStream_t stream);
hipError_t launch_float_math_probe(int kind, const float* d_x, size_t n, float* d_out, hipStream_t stream);
hipError_t launch_normalize(const NormalizeArgs& args, size_t batch, hipStream_t stream);

} // namespace gvtm
