// CDNA4 (gfx950) kernels of the batched vocal-tract model.
//
//   vtm_synth_kernel   (vtm_kernel_v2.inc)  VocalTractModel0 / 2 / 4 semantics: a workgroup owns 1, 2, 4 or 8 utterances and
//                      5 + NH wavefronts with fixed roles (tube, scans, pre-/post-tube filters, interpolation, helpers),
//                      software-pipelined over chunks of internal-rate steps with one barrier per tick; see that file's header
//   vtm5_synth_kernel  (vtm_kernel_m5.inc)  VocalTractModel5 semantics, same organisation
//   vtm_normalize_kernel                    output scaling of Controller::writeOutputToBuffer / writeOutputToFile
//
// Everything between the parameter frames (HBM in) and the audio samples (HBM out) lives in LDS; there is no
// intermediate global traffic.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "vtm_design.hpp"
#include "vtm_kernels.hpp"
#include "vtm_math.hpp"

namespace gvtm {

namespace {
#include "vtm_device_common.inc"
} // namespace

#include "vtm_kernel_v2.inc"
#include "vtm_kernel_m5.inc"

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

// kernel geometry: utterances per workgroup (DPP rows), chunk length, helper wavefronts,
// internal-rate ring length
template <typename CT, typename ST, int U_, int D_ = 1>
struct V2Shape {
	static constexpr int U = U_;
	// Chunk length C: a multiple of 4 (scan blocks) and of the tube unroll (2 for SectionDelay 1, D for
	// even D, 6 for 3), i.e. of 12.
	//   U = 1: the tick time is the tube wavefront's plus what a tick costs besides (barrier, stage prologues): measured
	//          on batch 256 in float, C = 60 / 96 / 120 / 144 / 168 / 192 -> 3.73 / 3.58 / 3.53 / 3.48 / 3.52 / 3.59 ms (108 and
	//          180, whose last 64-item helper pass is half empty: 3.73 / 3.70); mixed 60 / 96 / 120 -> 4.19 / 4.04 / 4.15;
	//          fp64 60 / 72 / 84 -> 4.14 / 4.65 / 4.07 (LDS ends there).  No power-of-two lengths (LDS strides).
	//   U > 1: every tick has fixed costs (barrier, stage prologues, ticket traffic, partly filled 64-item
	//          passes), so the longest chunk LDS allows wins: measured on batch 4096 in float, C = 24 / 36 / 48
	//          -> 24.8 / 21.6 / 18.4 ms; batch 512, C = 24 / 48 / 96 -> 4.71 / 4.64 / 4.38 ms.
	// LDS per workgroup: tests/tools/lds_sizes.py (float 1x144 120 KB, 2x96 152, 4x48 154; mixed 4x32 154; fp64 4x32 159).
	// (fp64 with several rows: resampler table without its delta half)
	static constexpr bool kAllFloat = sizeof(CT) == 4;
	static constexpr bool kMixed = sizeof(CT) == 8 && sizeof(ST) == 4;
#ifdef GVTM_TUNE_C1
	static constexpr int C = (U_ == 1) ? GVTM_TUNE_C1 : (U_ == 2 ? GVTM_TUNE_C2 : GVTM_TUNE_C4);
#else
	// four rows in double: 32 steps = two FULL 64-item passes per stage (24 steps left the second pass half empty and
	// made every pass a large share of a tick); SectionDelay 3 unrolls the tube by 6 and keeps 24.  (28 steps for
	// SectionDelay 1, so that the resampler's 61.6 outputs per chunk fill ONE pass, measured slower: 12.7 vs 13.3 G.)
	static constexpr int C = (U_ == 1) ? (kAllFloat ? 144 : (kMixed ? 96 : 84))
	                                   : (U_ == 2 ? (kAllFloat ? 96 : 48) : (U_ == 8 ? 24 : (kAllFloat ? (D_ == 3 ? 36 : 48) : (D_ == 3 ? 24 : 32))));
	// (float, four rows, SectionDelay 3: 36 -- reference model 3 down-samples to 44.1 kHz, so its rings are 1024 samples each,
	// and with the lane-indexed tube record 48 steps no longer fit)
#endif
#ifndef GVTM_TUNE_NH_MULTI
#define GVTM_TUNE_NH_MULTI 7
#endif
#ifndef GVTM_TUNE_NH_SINGLE
#define GVTM_TUNE_NH_SINGLE 3
#endif
#ifndef GVTM_TUNE_NH_OCTO
#define GVTM_TUNE_NH_OCTO 6
#endif
	// 8 resp. 12 wavefronts per workgroup (16 in float with four rows, below); eight rows: two wavefronts per serial role + 6 helpers = 16
	// Float, four utterances per workgroup: ELEVEN helpers = 16 wavefronts, the most a workgroup can have (128 registers
	// each).  With the tube record in blocks of four steps the tube and pre-tube filter wavefronts need 177 / 150 cycles per
	// step and the helper pool is the tick: same box, 4096 utterances, SectionDelay 2 x 2000 frames: 7 / 8 / 9 / 11 helpers
	// -> 107.6 / 106.2 / 103.6 / 100.0 ms; SectionDelay 1 x 500 frames: 14.24 / 14.00 / 13.68 / 13.11 ms.  (Round 2, when
	// the tube was the tick: 14 / 16 wavefronts 63.0 -> 62.6 ms.  The double models spill at 128 registers: fp64 22.8 -> 24.5 ms.)
#ifndef GVTM_TUNE_NH_F32_4
#define GVTM_TUNE_NH_F32_4 11
#endif
	static constexpr int NH = (U_ == 1) ? GVTM_TUNE_NH_SINGLE : (U_ == 8 ? GVTM_TUNE_NH_OCTO : ((U_ == 4 && kAllFloat) ? GVTM_TUNE_NH_F32_4 : GVTM_TUNE_NH_MULTI));
	static constexpr int kWaves = 5 * ((U_ + 3) / 4) + NH;
};

// internal-rate ring of a workgroup row: a power of two holding two chunks, the resampler's history and the
// flush zeros; the reference's own BUFFER_SIZE when down-sampling (the flush overrun reads the ring's leftovers
// modulo that length, vtm_kernel_v2.inc's epilogue)
static int ring_length(const DeviceConstants& k, int chunk)
{
	if (!k.upsampling) return kSrcRing;
	int xr = 128;
	// (+ 64: the resampler emits on a 64-aligned grid of outputs, so up to 63 outputs = at most 64 inputs wait a chunk longer)
	while (xr < 2 * chunk + 4 * k.pad + 64) xr *= 2;
	return xr;
}

// helper wavefronts of the 48-section tube's workgroups: U tube wavefronts + 4 other serial ones + helpers = 12
// (three wavefronts per SIMD: 168 registers each)
#ifndef GVTM_TUNE_WIDE_NH2
#define GVTM_TUNE_WIDE_NH2 6
#endif
#ifndef GVTM_TUNE_WIDE_NH4
#define GVTM_TUNE_WIDE_NH4 4
#endif
template <int U>
constexpr int wide_helpers()
{
	return U == 1 ? GVTM_TUNE_NH_SINGLE : (U == 2 ? GVTM_TUNE_WIDE_NH2 : GVTM_TUNE_WIDE_NH4);
}

template <typename CT, typename ST, int D, int U, int LAYOUT = 0>
static hipError_t launch_v2(const SynthArgs& args, size_t batch, hipStream_t stream)
{
	using S = V2Shape<CT, ST, U, D>;
	constexpr int NH = LAYOUT == 1 ? wide_helpers<U>() : S::NH;
	constexpr int kWaves = v2::serial_waves<U, LAYOUT>() + NH;
	auto fn = v2::vtm_synth_kernel<CT, ST, D, S::U, S::C, NH, LAYOUT>;
	// (a stream keeps ONE ring length for all shapes, the one-row shape's: longer than this shape needs, never shorter)
	if (args.xr < ring_length(args.k, S::C) || (args.xr & (args.xr - 1)) != 0 || 2 * S::C + 4 * args.k.pad + 64 > args.xr) return hipErrorInvalidValue;
	if (!args.k.upsampling && args.xr != kSrcRing) return hipErrorInvalidValue;
	const size_t lds = v2::smem_bytes<CT, ST, S::U, S::C, v2::lane_rec<CT, LAYOUT>()>(args.xr);
	hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
			static_cast<int>(lds));
	if (e != hipSuccess) return e;
	const unsigned groups = static_cast<unsigned>((batch + S::U - 1) / S::U);
	hipLaunchKernelGGL(fn, dim3(groups), dim3(kWaves * 64), lds, stream, args);
	return hipGetLastError();
}

int synth_rows(int precision, size_t batch, int requested, int section_delay)
{
	// utterances per workgroup = DPP rows used by the serial wavefronts.  One row keeps the most
	// workgroups in flight (best latency for small batches); more rows amortise the serial
	// instruction streams once there are more utterances than compute units.
	// eight rows (two wavefronts per serial role) fit in float only
	const int max_rows = precision == GVTM_PRECISION_F32 ? 8 : 4;
	int rows = requested;
	if (rows != 1 && rows != 2 && rows != 4 && rows != 8) {
		rows = batch > 512 ? 4 : (batch > 256 ? 2 : 1);
		// (fp64 used to stop at two rows: with the compact records four fit with a chunk of 32 and run faster)
		// mixed with SectionDelay 3 or 4 (deeper fp64 delay lines per lane): measured 3.36 vs 2.63 and 2.07 vs 1.65 G samples/s
		if (precision != GVTM_PRECISION_F32 && section_delay >= 3 && rows > 2) rows = 2;
	}
	return rows > max_rows ? max_rows : rows;
}

// what depends on the workgroup shape V2Shape picks: chunk length -> ring length -> LDS bytes
struct ShapeNumbers { int chunk; size_t lds_fixed; size_t ring_elem; };

template <typename CT, typename ST, int U, int D>
static ShapeNumbers v2_numbers(int layout)
{
	using S = V2Shape<CT, ST, U, D>;
	const size_t lds = layout == 1 ? v2::smem_bytes<CT, ST, S::U, S::C, v2::lane_rec<CT, 1>()>(0) : v2::smem_bytes<CT, ST, S::U, S::C, v2::lane_rec<CT, 0>()>(0);
	return ShapeNumbers{S::C, lds, sizeof(ST) * S::U};
}

template <typename CT, typename ST, int D>
static ShapeNumbers v2_numbers_rows(int rows, int layout)
{
	if constexpr (sizeof(CT) == 4) {
		if (rows == 8) return v2_numbers<CT, ST, 8, D>(layout);
	}
	return rows == 4 ? v2_numbers<CT, ST, 4, D>(layout) : (rows == 2 ? v2_numbers<CT, ST, 2, D>(layout) : v2_numbers<CT, ST, 1, D>(layout));
}

static ShapeNumbers shape_numbers(int precision, int rows, int delay, int layout)
{
	// the chunk length depends on the SectionDelay through "1, 3 or another"
	if (delay == 3) {
		if (precision == GVTM_PRECISION_F32) return v2_numbers_rows<float, float, 3>(rows, layout);
		if (precision == GVTM_PRECISION_MIXED) return v2_numbers_rows<double, float, 3>(rows, layout);
		return v2_numbers_rows<double, double, 3>(rows, layout);
	}
	if (delay != 1) {
		if (precision == GVTM_PRECISION_F32) return v2_numbers_rows<float, float, 2>(rows, layout);
		if (precision == GVTM_PRECISION_MIXED) return v2_numbers_rows<double, float, 2>(rows, layout);
		return v2_numbers_rows<double, double, 2>(rows, layout);
	}
	if (precision == GVTM_PRECISION_F32) return v2_numbers_rows<float, float, 1>(rows, layout);
	if (precision == GVTM_PRECISION_MIXED) return v2_numbers_rows<double, float, 1>(rows, layout);
	return v2_numbers_rows<double, double, 1>(rows, layout);
}

int synth_ring_length(const DeviceConstants& k, int precision, int rows)
{
	return ring_length(k, shape_numbers(precision, rows, k.section_delay, k.layout).chunk);
}

size_t stream_state_bytes(const DeviceConstants& k, int precision, int xr)
{
	const int lanes = k.layout == 1 ? 48 : 16, words = 2 * k.section_delay + 1;
	if (precision == GVTM_PRECISION_F32) return StreamLayout<float, float>::bytes(lanes, words, xr);
	if (precision == GVTM_PRECISION_MIXED) return StreamLayout<double, float>::bytes(lanes, words, xr);
	return StreamLayout<double, double>::bytes(lanes, words, xr);
}

size_t synth_lds_bytes(const DeviceConstants& k, int precision, int rows, int xr)
{
	const ShapeNumbers n = shape_numbers(precision, rows, k.section_delay, k.layout);
	return n.lds_fixed + ((n.ring_elem * static_cast<size_t>(xr > 0 ? xr : ring_length(k, n.chunk)) + 15) & ~size_t(15));
}

template <typename CT, typename ST, int U>
static hipError_t launch_v2_d(const SynthArgs& args, size_t batch, hipStream_t stream)
{
	if (args.k.layout == 1) {
		// VocalTractModel4: 48 section lanes = one utterance per tube wavefront (up to four of them per
		// workgroup), SectionDelay 1 only
		if (args.k.section_delay != 1) return hipErrorInvalidValue;
		if constexpr (U >= 4) return launch_v2<CT, ST, 1, 4, 1>(args, batch, stream);
		else return launch_v2<CT, ST, 1, U, 1>(args, batch, stream);
	}
	switch (args.k.section_delay) {
	case 1: return launch_v2<CT, ST, 1, U>(args, batch, stream);
	case 2: return launch_v2<CT, ST, 2, U>(args, batch, stream);
	case 3: return launch_v2<CT, ST, 3, U>(args, batch, stream);
	case 4: return launch_v2<CT, ST, 4, U>(args, batch, stream);
	}
	return hipErrorInvalidValue;
}

template <typename CT, typename ST>
static hipError_t launch_v2_rows(const SynthArgs& args, size_t batch, int rows, hipStream_t stream)
{
	if constexpr (sizeof(CT) == 4) {
		if (rows == 8) return launch_v2_d<CT, ST, 8>(args, batch, stream);
	}
	if (rows == 4) return launch_v2_d<CT, ST, 4>(args, batch, stream);
	if (rows == 2) return launch_v2_d<CT, ST, 2>(args, batch, stream);
	return launch_v2_d<CT, ST, 1>(args, batch, stream);
}

hipError_t launch_synth(const SynthArgs& args, size_t batch, int precision, int rows, hipStream_t stream)
{
	if (precision == GVTM_PRECISION_F32) return launch_v2_rows<float, float>(args, batch, rows, stream);
	if (precision == GVTM_PRECISION_MIXED) return launch_v2_rows<double, float>(args, batch, rows, stream);
	return launch_v2_rows<double, double>(args, batch, rows, stream);
}

// reference model 5.  One utterance per workgroup: chunk of 60 steps (one 64-lane pass per per-step stage), three helper
// wavefronts.  Two utterances per workgroup (batches beyond one workgroup per compute unit): two tube wavefronts, chunk of
// 24 steps (2 x 24 items per per-step pass; what LDS holds with two 62-entry tube records per step), five helpers.
constexpr int kM5Ring = kSrcRing; // the reference's BUFFER_SIZE: see the flush-overrun epilogue
constexpr int kM5Chunk1 = 60, kM5Helpers1 = 3;
#ifndef GVTM_TUNE_M5_NH2
#define GVTM_TUNE_M5_NH2 5
#endif
constexpr int kM5Chunk2 = 24, kM5Helpers2 = GVTM_TUNE_M5_NH2;

size_t synth5_lds_bytes(int rows)
{
	return rows == 2 ? m5::Offsets<kM5Chunk2, kM5Ring, 2>().total : m5::Offsets<kM5Chunk1, kM5Ring, 1>().total;
}

template <int C, int NH, int U>
static hipError_t launch_synth5_shape(const SynthArgs& args, size_t batch, hipStream_t stream)
{
	auto fn = m5::vtm5_synth_kernel<C, NH, kM5Ring, U>;
	const size_t lds = m5::Offsets<C, kM5Ring, U>().total;
	hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
	if (e != hipSuccess) return e;
	hipLaunchKernelGGL(fn, dim3(static_cast<unsigned>((batch + U - 1) / U)), dim3((3 + U + NH) * 64), lds, stream, args);
	return hipGetLastError();
}

hipError_t launch_synth5(const SynthArgs& args, size_t batch, int rows, hipStream_t stream)
{
	if (!args.k5const) return hipErrorInvalidValue;
	if (rows == 2) return launch_synth5_shape<kM5Chunk2, kM5Helpers2, 2>(args, batch, stream);
	return launch_synth5_shape<kM5Chunk1, kM5Helpers1, 1>(args, batch, stream);
}

hipError_t launch_normalize(const NormalizeArgs& args, size_t batch, hipStream_t stream)
{
	const unsigned bx = static_cast<unsigned>((args.audio_stride + 256 * 8 - 1) / (256 * 8));
	hipLaunchKernelGGL(vtm_normalize_kernel, dim3(bx > 0 ? bx : 1, static_cast<unsigned>(batch)), dim3(256), 0, stream, args);
	return hipGetLastError();
}

} // namespace gvtm
