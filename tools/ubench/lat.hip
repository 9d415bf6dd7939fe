// Lone-wavefront latency/issue microbenchmarks for the serial roles (diagnostics only).
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 4096
template <int MODE>
__global__ void k(double* out, unsigned long long* cyc, double a, double b)
{
	__shared__ double lds[256];
	lds[threadIdx.x] = a * threadIdx.x;
	__syncthreads();
	double x0 = a + threadIdx.x, x1 = b, x2 = a * 2, x3 = b * 3;
	int idx = threadIdx.x & 63;
	unsigned long long t0 = clock64();
	for (int i = 0; i < N; ++i) {
		if (MODE == 0) { x0 = __builtin_fma(x0, a, b); }                       // dependent fma f64
		if (MODE == 1) { x0 = __builtin_fma(x0, a, b); x1 = __builtin_fma(x1, a, b); x2 = __builtin_fma(x2, a, b); x3 = __builtin_fma(x3, a, b); } // 4 independent
		if (MODE == 2) { x0 = x0 + b; x0 = (x0 > 511.0) ? x0 - 512.0 : x0; } // add, cmp, cndmask chain
		if (MODE == 3) { // dependent DPP shift of a double + add
			int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x0), 0x111, 0xF, 0xF, true);
			int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x0), 0x111, 0xF, 0xF, true);
			x0 = __hiloint2double(hi, lo) + b;
		}
		if (MODE == 4) { // readlane -> fma dependent
			int lo = __builtin_amdgcn_readlane(__double2loint(x0), 3);
			int hi = __builtin_amdgcn_readlane(__double2hiint(x0), 3);
			x0 = __builtin_fma(__hiloint2double(hi, lo), a, b);
		}
		if (MODE == 5) { // dependent LDS read (pointer chase through index)
			x0 = lds[idx]; idx = (static_cast<int>(x0) + i) & 63;
		}
		if (MODE == 6) { float f = static_cast<float>(x0); f = __builtin_fmaf(f, 1.0001f, 0.5f); x0 = f; } // cvt chain
		if (MODE == 7) { x0 = x0 * a; x0 = x0 - floor(x0); }                   // noise chain
	}
	unsigned long long t1 = clock64();
	out[threadIdx.x] = x0 + x1 + x2 + x3 + idx;
	if (threadIdx.x == 0) cyc[MODE] = t1 - t0;
}
int main()
{
	double* out; unsigned long long* cyc;
	hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 16 * 8);
	hipMemset(cyc, 0, 16 * 8);
	k<0><<<1, 64>>>(out, cyc, 0.999, 0.001); k<1><<<1, 64>>>(out, cyc, 0.999, 0.001); k<2><<<1, 64>>>(out, cyc, 0.999, 3.1);
	k<3><<<1, 64>>>(out, cyc, 0.999, 0.001); k<4><<<1, 64>>>(out, cyc, 0.999, 0.001); k<5><<<1, 64>>>(out, cyc, 0.999, 0.001);
	k<6><<<1, 64>>>(out, cyc, 0.999, 0.001); k<7><<<1, 64>>>(out, cyc, 377.0, 0.001);
	hipDeviceSynchronize();
	unsigned long long h[16]; hipMemcpy(h, cyc, 16 * 8, hipMemcpyDeviceToHost);
	const char* names[] = {"dependent v_fma_f64", "4 independent v_fma_f64 (per group)", "add+cmp+cndmask chain (phase wrap)", "DPP(double)+add chain", "readlane(double)+fma chain", "dependent ds_read_b64 (+cvt,and)", "cvt f64->f32, fmaf, cvt back", "mul+floor+sub chain (noise)"};
	for (int i = 0; i < 8; ++i) printf("%-40s %.1f cycles/iter\n", names[i], (double) h[i] / N);
	return 0;
}
