// lab: v_mfma_f64_16x16x4_f64 on gfx950 -- operand / result lane maps checked with exact integer data, and the issue rate of
// dependent chains (1, 2, 4 independent accumulators) at 1, 2 and 4 wavefronts per SIMD.  Feeds the design of the LinPSF fit
// (csrc/linpsf_mfma.hip): D[pixel][cadence] = K^T[pixel][monomial] * M[monomial][cadence].
//   hipcc --offload-arch=gfx950 -O3 -o mfma_f64 mfma_f64.hip && ./mfma_f64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double f64x4 __attribute__((ext_vector_type(4)));

__global__ void layout_kernel(const double* A, const double* B, double* D)
{
	// A [16][4] row-major (row = output row, k), B [4][16] (k, output column); lane l: A[l & 15][l >> 4], B[l >> 4][l & 15]
	const int l = threadIdx.x;
	const double a = A[(l & 15) * 4 + (l >> 4)], b = B[(l >> 4) * 16 + (l & 15)];
	f64x4 c = {0.0, 0.0, 0.0, 0.0};
	c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
	// expected: D[row = (l >> 4) + 4 r][col = l & 15] in register r
	for (int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = c[r];
}

template <int CHAINS>
__global__ __launch_bounds__(256) void rate_kernel(double* out, int iters)
{
	f64x4 acc[CHAINS];
	for (int i = 0; i < CHAINS; ++i) acc[i] = f64x4{0.0, 0.0, 0.0, 0.0};
	double a = (double)(threadIdx.x & 7) * 0.25, b = (double)(threadIdx.x & 3) * 0.5;
	for (int it = 0; it < iters; ++it) {
#pragma unroll
		for (int u = 0; u < 8; ++u) {
#pragma unroll
			for (int i = 0; i < CHAINS; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
		}
	}
	double s = 0.0;
	for (int i = 0; i < CHAINS; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
	if (s == 12345.678) out[0] = s;
}

// the same chains with 10 independent v_fma_f64 per MFMA in the same wavefront: does the vector pipe run beside the matrix pipe?
template <int CHAINS, int NV>
__global__ __launch_bounds__(256) void mix_kernel(double* out, int iters)
{
	f64x4 acc[CHAINS];
	for (int i = 0; i < CHAINS; ++i) acc[i] = f64x4{0.0, 0.0, 0.0, 0.0};
	double a = (double)(threadIdx.x & 7) * 0.25, b = (double)(threadIdx.x & 3) * 0.5;
	double v[8];
	for (int i = 0; i < 8; ++i) v[i] = 1.0 + threadIdx.x * 1e-9 * i;
	for (int it = 0; it < iters; ++it) {
#pragma unroll
		for (int u = 0; u < 8; ++u) {
#pragma unroll
			for (int i = 0; i < CHAINS; ++i) {
				acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
				for (int q = 0; q < NV; ++q) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(v[q & 7]) : "v"(v[(q + 1) & 7]));
			}
		}
	}
	double s = 0.0;
	for (int i = 0; i < CHAINS; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
	for (int i = 0; i < 8; ++i) s += v[i];
	if (s == 12345.678) out[0] = s;
}

// other instruction kinds beside the MFMAs: 1 v_fma_f32, 2 v_add_u32, 3 v_cvt_f64_f32, 4 v_cndmask_b32, 5 v_mul_f64, 6 v_mov_b32
template <int KIND, int NV>
__global__ __launch_bounds__(256) void mixk_kernel(double* out, int iters)
{
	f64x4 acc[2];
	for (int i = 0; i < 2; ++i) acc[i] = f64x4{0.0, 0.0, 0.0, 0.0};
	double a = (double)(threadIdx.x & 7) * 0.25, b = (double)(threadIdx.x & 3) * 0.5;
	float f[8]; unsigned u32[8]; double d[8];
	for (int i = 0; i < 8; ++i) { f[i] = 1.f + threadIdx.x * 1e-6f * i; u32[i] = threadIdx.x + i; d[i] = 1.0 + threadIdx.x * 1e-9 * i; }
	for (int it = 0; it < iters; ++it) {
#pragma unroll
		for (int u = 0; u < 8; ++u) {
#pragma unroll
			for (int i = 0; i < 2; ++i) {
				acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
				for (int q = 0; q < NV; ++q) {
					if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[q & 7]) : "v"(f[(q + 1) & 7]));
					if (KIND == 2) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u32[q & 7]) : "v"(u32[(q + 1) & 7]));
					if (KIND == 3) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[q & 7]) : "v"(f[(q + 1) & 7]));
					if (KIND == 4) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u32[q & 7]) : "v"(u32[(q + 1) & 7]) : );
					if (KIND == 5) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[q & 7]) : "v"(d[(q + 1) & 7]));
					if (KIND == 6) asm volatile("v_mov_b32 %0, %1" : "=v"(u32[q & 7]) : "v"(u32[(q + 1) & 7]));
				}
			}
		}
	}
	double s = 0.0;
	for (int i = 0; i < 2; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
	for (int i = 0; i < 8; ++i) s += f[i] + u32[i] + d[i];
	if (s == 12345.678) out[0] = s;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <typename K>
static double time_kernel(K kern, int blocks, int iters, double* d_out)
{
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_out, iters);
	hipDeviceSynchronize();
	hipEventRecord(e0);
	hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_out, iters);
	hipEventRecord(e1);
	hipEventSynchronize(e1);
	float ms = 0.f;
	hipEventElapsedTime(&ms, e0, e1);
	return ms;
}

int main()
{
	// layout
	std::vector<double> A(64), B(64), D(256), R(256, 0.0);
	for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) A[i * 4 + k] = (double)(1 + i * 7 + k * 3);
	for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) B[k * 16 + j] = (double)(2 + k * 11 + j * 5 + (j * j) % 7);
	for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 4; ++k) R[i * 16 + j] += A[i * 4 + k] * B[k * 16 + j];
	double *dA, *dB, *dD;
	CK(hipMalloc(&dA, 64 * 8)); CK(hipMalloc(&dB, 64 * 8)); CK(hipMalloc(&dD, 256 * 8));
	CK(hipMemcpy(dA, A.data(), 64 * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 64 * 8, hipMemcpyHostToDevice));
	hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dD);
	CK(hipMemcpy(D.data(), dD, 256 * 8, hipMemcpyDeviceToHost));
	int bad = 0;
	for (int i = 0; i < 256; ++i) bad += (D[i] != R[i]);
	printf("layout: %d of 256 elements differ (A[l&15][l>>4], B[l>>4][l&15], D[(l>>4)+4r][l&15])\n", bad);

	double* d_out;
	CK(hipMalloc(&d_out, 8));
	const int iters = 20000;
	const double flop_per_mfma = 2.0 * 16 * 16 * 4;
	for (int wps = 1; wps <= 4; wps *= 2) {
		const int blocks = 256 * wps;   // one 256-thread workgroup = one wavefront per SIMD of a CU
		struct { const char* name; double ms; int chains; } rows[3] = {
			{"1 chain ", time_kernel(rate_kernel<1>, blocks, iters, d_out), 1},
			{"2 chains", time_kernel(rate_kernel<2>, blocks, iters, d_out), 2},
			{"4 chains", time_kernel(rate_kernel<4>, blocks, iters, d_out), 4}};
		for (auto& r : rows) {
			const double n = (double)blocks * 4 * iters * 8 * r.chains;
			printf("waves/SIMD %d  %s  %.3f ms  %.1f TFLOP/s  (%.1f cycles per MFMA per SIMD at 2.4 GHz)\n", wps, r.name, r.ms,
				n * flop_per_mfma / (r.ms * 1e-3) / 1e12, r.ms * 1e-3 * 2.4e9 / (iters * 8.0 * r.chains * wps));
		}
	}
	for (int wps = 1; wps <= 4; wps *= 2) {
		const int blocks = 256 * wps;
		const double m0 = time_kernel(mix_kernel<2, 0>, blocks, iters, d_out), m4 = time_kernel(mix_kernel<2, 4>, blocks, iters, d_out),
			m8 = time_kernel(mix_kernel<2, 8>, blocks, iters, d_out), m16 = time_kernel(mix_kernel<2, 16>, blocks, iters, d_out);
		printf("waves/SIMD %d  2 chains + {0, 4, 8, 16} v_fma_f64 per MFMA: %.3f %.3f %.3f %.3f ms\n", wps, m0, m4, m8, m16);
	}
	{
		const int blocks = 256 * 2;
		printf("2 waves/SIMD, 2 chains + {0, 4, 8, 16} instructions of another kind per MFMA (ms):\n");
#define ROW(K, NAME) printf("  %-14s %.3f %.3f %.3f %.3f\n", NAME, time_kernel(mixk_kernel<K, 0>, blocks, iters, d_out), time_kernel(mixk_kernel<K, 4>, blocks, iters, d_out), \
			time_kernel(mixk_kernel<K, 8>, blocks, iters, d_out), time_kernel(mixk_kernel<K, 16>, blocks, iters, d_out))
		ROW(1, "v_fma_f32"); ROW(2, "v_add_u32"); ROW(3, "v_cvt_f64_f32"); ROW(4, "v_cndmask_b32"); ROW(5, "v_mul_f64"); ROW(6, "v_mov_b32");
#undef ROW
	}
	return bad != 0;
}
