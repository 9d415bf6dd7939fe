// Diagnostics-only kernels: compiled into libgama_vtm_diag.so (tests and tools), never into the product library.
//
//   dpp_selftest_kernel      every cross-lane primitive the tube wavefront relies on, lane by lane
//   float_math_probe_kernel  the all-float path's per-step conversions evaluated on the device
#include <hip/hip_runtime.h>

#include <cstdint>

#include "vtm_design.hpp"
#include "vtm_kernels.hpp"
#include "vtm_math.hpp"

namespace gvtm {

namespace {
#include "vtm_device_common.inc"
} // namespace

// out[lane] = value received from `lane id` through each cross-lane primitive the tube relies on
__global__ void dpp_selftest_kernel(int* out)
{
	const int lane = threadIdx.x & 63;
	const double v = static_cast<double>(lane);
	out[lane] = static_cast<int>(from_left(v));
	out[64 + lane] = static_cast<int>(from_right(v));
	out[128 + lane] = static_cast<int>(row_rotate<6>(v));
	out[192 + lane] = static_cast<int>(row_rotate<10>(v));
	out[256 + lane] = static_cast<int>(from_left_wave(v));
	out[320 + lane] = static_cast<int>(from_right_wave(v));
	out[384 + lane] = static_cast<int>(row_bcast<3>(v));                      // 64-bit: v_mov_b64_dpp row_newbcast
	out[448 + lane] = static_cast<int>(row_bcast<10>(static_cast<float>(lane))); // 32-bit
	out[512 + lane] = static_cast<int>(from_left(static_cast<float>(lane)));
	out[576 + lane] = static_cast<int>(row_rotate<10>(static_cast<float>(lane)));
}


hipError_t launch_dpp_selftest(int* d_out, hipStream_t stream)
{
	hipLaunchKernelGGL(dpp_selftest_kernel, dim3(1), dim3(64), 0, stream, d_out);
	return hipGetLastError();
}

// Test hook: the all-float path's per-step conversions evaluated ON THE DEVICE
// (kind 0 = Util::frequency, 1 = Util::amplitude60dB, 2 = tanf stand-in, 3 = cosf stand-in)
__global__ void float_math_probe_kernel(int kind, const float* x, size_t n, float* out)
{
	const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const float v = x[i];
	float r = 0.0f;
	switch (kind) {
	case 0: r = frequency_dev(v); break;
	case 1: r = amplitude_60db_dev(v); break;
	case 2: r = tan_dev(v); break;
	case 3: r = cos_dev(v); break;
	case 4: r = fdiv_n(v, x[i ^ 1]); break; // (n even)
	}
	out[i] = r;
}

hipError_t launch_float_math_probe(int kind, const float* d_x, size_t n, float* d_out, hipStream_t stream)
{
	hipLaunchKernelGGL(float_math_probe_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, stream, kind, d_x, n, d_out);
	return hipGetLastError();
}

} // namespace gvtm
