// Diagnostics only: does a vector instruction cost a SIMD less when only the first 32 / 16 / 4 lanes of the wavefront are
// enabled?  (The serial roles of the synthesis kernel own 4, 8, 36 or 64 of their 64 lanes.)
#include <hip/hip_runtime.h>
#include <cstdio>

#define N 4096

template <int MODE>
__global__ void masked_kernel(float* out, unsigned long long* cyc, float a, float b, int active)
{
	float x[8];
	double d[8];
	for (int i = 0; i < 8; ++i) { x[i] = a + threadIdx.x + i; d[i] = a + i; }
	unsigned long long t0 = 0, t1 = 0;
	if (static_cast<int>(threadIdx.x & 63) < active) {
		t0 = clock64();
		for (int it = 0; it < N; ++it) {
#pragma unroll
			for (int i = 0; i < 8; ++i) {
				if (MODE == 0) x[i] = __builtin_fmaf(x[i], a, b);
				if (MODE == 1) d[i] = __builtin_fma(d[i], (double) a, (double) b);
				if (MODE == 2) x[i] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x[i]), 0x111, 0xF, 0xF, true));
			}
		}
		t1 = clock64();
	}
	float s = 0;
	for (int i = 0; i < 8; ++i) s += x[i] + (float) d[i];
	out[threadIdx.x] = s;
	if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

template <int MODE>
static double run(int waves, int active, float* out, unsigned long long* cyc)
{
	hipMemset(cyc, 0, 16 * 8);
	masked_kernel<MODE><<<1, waves * 64>>>(out, cyc, 0.9999f, 0.0001f, active);
	hipDeviceSynchronize();
	unsigned long long h[16];
	hipMemcpy(h, cyc, 16 * 8, hipMemcpyDeviceToHost);
	double worst = 0;
	for (int w = 0; w < waves; ++w) worst = h[w] > worst ? (double) h[w] : worst;
	return worst / (N * 8.0);
}

int main()
{
	float* out; unsigned long long* cyc;
	hipMalloc(&out, 1024 * 4); hipMalloc(&cyc, 16 * 8);
	const char* names[3] = {"v_fma_f32", "v_fma_f64", "v_mov_b32_dpp"};
	for (int waves : {4, 12}) { // one / three wavefronts per SIMD
		for (int mode = 0; mode < 3; ++mode) {
			printf("%2d wavefronts, %-14s cycles per instruction with 64 / 32 / 16 / 4 lanes enabled:", waves, names[mode]);
			for (int active : {64, 32, 16, 4}) {
				const double c = mode == 0 ? run<0>(waves, active, out, cyc) : (mode == 1 ? run<1>(waves, active, out, cyc) : run<2>(waves, active, out, cyc));
				printf(" %.2f", c);
			}
			printf("\n");
		}
	}
	return 0;
}
