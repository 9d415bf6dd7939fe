// Diagnostics only: (1) which SIMD each wavefront of a workgroup lands on, (2) what a vector instruction costs a SIMD when
// one, two or three wavefronts on it want to issue all the time (the serial-role wavefronts share their SIMD with helpers).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void where_kernel(unsigned* out)
{
	unsigned id;
	asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
	if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = id;
}

#define N 2048
using F2 = float __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void issue_kernel(float* out, unsigned long long* cyc, float a, float b)
{
	// eight independent chains per lane: latency never limits, only issue / pipe occupancy
	float x[8];
	double d[8];
	F2 p[4];
	for (int i = 0; i < 8; ++i) { x[i] = a + threadIdx.x + i; d[i] = a + i; }
	for (int i = 0; i < 4; ++i) p[i] = F2{a + i, b - i};
	__syncthreads();
	const unsigned long long t0 = clock64();
	for (int it = 0; it < N; ++it) {
#pragma unroll
		for (int i = 0; i < 8; ++i) {
			if (MODE == 0) x[i] = __builtin_fmaf(x[i], a, b);
			if (MODE == 1) x[i] = x[i] * a;
			if (MODE == 2) d[i] = __builtin_fma(d[i], (double) a, (double) b);
			if (MODE == 3) d[i] = d[i] + (double) b;
			if (MODE == 4) d[i] = d[i] * (double) a;
			if (MODE == 5) x[i] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x[i]), 0x111, 0xF, 0xF, true));
			if (MODE == 8) x[i] = (x[i] > b) ? x[i] - a : x[i];
		}
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			if (MODE == 6) { asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(F2{a, a})); asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(F2{b, b})); }
			if (MODE == 7) { asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(F2{a, a})); asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(F2{b, b})); }
			if (MODE == 9) { asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(F2{a, a})); asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(F2{b, b})); }
		}
	}
	const unsigned long long t1 = clock64();
	float s = 0;
	for (int i = 0; i < 8; ++i) s += x[i] + (float) d[i];
	for (int i = 0; i < 4; ++i) s += p[i].x + p[i].y;
	out[threadIdx.x] = s;
	if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

template <int MODE>
double run(int waves, float* out, unsigned long long* cyc)
{
	hipMemset(cyc, 0, 16 * 8);
	issue_kernel<MODE><<<1, waves * 64>>>(out, cyc, 0.9999f, 0.0001f);
	hipDeviceSynchronize();
	unsigned long long h[16];
	hipMemcpy(h, cyc, 16 * 8, hipMemcpyDeviceToHost);
	double worst = 0;
	for (int w = 0; w < waves; ++w) worst = h[w] > worst ? (double) h[w] : worst;
	return worst / (N * 8.0);
}

int main()
{
	unsigned* where; float* out; unsigned long long* cyc;
	hipMalloc(&where, 64 * 16 * 4); hipMalloc(&out, 1024 * 4); hipMalloc(&cyc, 16 * 8);
	for (int waves : {8, 12, 16}) {
		hipMemset(where, 0xff, 64 * 16 * 4);
		where_kernel<<<64, waves * 64>>>(where);
		hipDeviceSynchronize();
		std::vector<unsigned> h(64 * 16);
		hipMemcpy(h.data(), where, h.size() * 4, hipMemcpyDeviceToHost);
		for (int b = 0; b < 3; ++b) {
			printf("workgroup %d of %d wavefronts: SIMD of wavefront 0..%d:", b, waves, waves - 1);
			for (int w = 0; w < waves; ++w) printf(" %u", (h[b * 16 + w] >> 4) & 3);
			printf("   (CU %u SE %u)\n", (h[b * 16] >> 8) & 15, (h[b * 16] >> 13) & 7);
		}
	}
	const char* names[] = {"v_fma_f32", "v_mul_f32", "v_fma_f64", "v_add_f64", "v_mul_f64", "v_mov_b32 dpp row_shr", "v_pk_mul_f32", "v_pk_add_f32", "v_cmp+v_cndmask+v_sub (3 instr)", "v_pk_fma_f32"};
	printf("cycles per wave-instruction as seen by each wavefront, 8 independent chains; wavefronts per workgroup = 1, 4, 8, 12 (SIMD shared by 1, 1, 2, 3)\n");
	for (int m = 0; m < 10; ++m) {
		double r[4];
		int ws[4] = {1, 4, 8, 12};
		for (int j = 0; j < 4; ++j) {
			switch (m) {
			case 0: r[j] = run<0>(ws[j], out, cyc); break; case 1: r[j] = run<1>(ws[j], out, cyc); break;
			case 2: r[j] = run<2>(ws[j], out, cyc); break; case 3: r[j] = run<3>(ws[j], out, cyc); break;
			case 4: r[j] = run<4>(ws[j], out, cyc); break; case 5: r[j] = run<5>(ws[j], out, cyc); break;
			case 6: r[j] = run<6>(ws[j], out, cyc); break; case 7: r[j] = run<7>(ws[j], out, cyc); break;
			case 8: r[j] = run<8>(ws[j], out, cyc); break; case 9: r[j] = run<9>(ws[j], out, cyc); break;
			}
		}
		printf("%-34s %6.2f %6.2f %6.2f %6.2f\n", names[m], r[0], r[1], r[2], r[3]);
	}
	return 0;
}
