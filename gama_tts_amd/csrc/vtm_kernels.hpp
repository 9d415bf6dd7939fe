// Launch interface between the C ABI (vtm_capi.cpp) and the kernels (vtm_kernels.hip).
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "vtm_design.hpp"

namespace gvtm {

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
	const void* src_h;           // [3328]
	const void* src_dh;          // [3328]
	size_t max_frames;
	size_t audio_stride;
	size_t batch;
	int xr;                      // internal-rate ring length per utterance (a power of two; synth_ring_length())
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
size_t synth_lds_bytes(const DeviceConstants& k, int precision, int rows);
hipError_t launch_synth(const SynthArgs& args, size_t batch, int precision, int rows, hipStream_t stream);
// reference model 5 (VocalTractModel5<double,1>): one utterance per workgroup, fp64
size_t synth5_lds_bytes();
hipError_t launch_synth5(const SynthArgs& args, size_t batch, hipStream_t stream);
constexpr int kDppSelftestInts = 640;
hipError_t launch_dpp_selftest(int* d_out /* [kDppSelftestInts] */, hipStream_t stream);
hipError_t launch_float_math_probe(int kind, const float* d_x, size_t n, float* d_out, hipStream_t stream);
hipError_t launch_normalize(const NormalizeArgs& args, size_t batch, hipStream_t stream);

} // namespace gvtm
