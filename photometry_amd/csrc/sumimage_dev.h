// sumimage_dev.h -- device code of A1 shared by the stand-alone sum-image kernel (sumimage.hip) and the fused
// per-target kernel (fused.hip): one wavefront reduces pixel rows (time series) to their good-cadence means.
#pragma once
#include "common.h"
#include <cmath>

namespace tp_sum {

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
	return v;
}
__device__ __forceinline__ int wave_sum_i32(int v) {
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
	return v;
}

__device__ __forceinline__ void acc1(float v, unsigned g, double& s, int& n) {
	// isfinite(v) && good  (BasePhotometry.py:1011-1015)
	bool ok = (g != 0u) && (fabsf(v) <= 3.402823466e+38f);
	s += ok ? (double)v : 0.0;
	n += ok ? 1 : 0;
}

// good-quality flags of the cadences, one byte each, padded to a multiple of 4 (BasePhotometry.py:1010)
__device__ __forceinline__ void stage_good(unsigned char* good, const int32_t* q, uint32_t bitmask, int n_cad, int tid, int nthreads) {
	const int n_cad4 = (n_cad + 3) & ~3;
	for (int k = tid; k < n_cad4; k += nthreads)
		good[k] = (k < n_cad && ((uint32_t)q[k] & bitmask) == 0u) ? 1 : 0;
}

// R pixel rows at once by one wavefront (R independent 1 KiB loads in flight per lane and step).  Every lane adds
// its cadences in increasing order and the 64 partial sums go through the same shuffle tree whatever R is, so the
// result does not depend on R.  Lane 0 returns the means (NaN where no good finite cadence exists).
template <int R>
__device__ __forceinline__ void rows_mean_vec4(const float* base, int64_t t_pitch, int p0, const float* sub, const unsigned char* good,
	int n_cad, int lane, double (&mean)[R])
{
	const int nq = ((n_cad + 3) & ~3) >> 2; // quads (the tail quad reads into the row padding: pitch % 4 == 0)
	const float4* sub4 = reinterpret_cast<const float4*>(sub);
	const uint32_t* good4 = reinterpret_cast<const uint32_t*>(good);
	const float4* row4[R];
	double s[R];
	int n[R];
#pragma unroll
	for (int j = 0; j < R; j++) { row4[j] = reinterpret_cast<const float4*>(base + (int64_t)(p0 + j) * t_pitch); s[j] = 0.0; n[j] = 0; }
	for (int qd = lane; qd < nq; qd += 64) {
		float4 a[R];
#pragma unroll
		for (int j = 0; j < R; j++) a[j] = row4[j][qd];
		const uint32_t g = good4[qd];
		if (sub) {
			const float4 sa = sub4[qd];
#pragma unroll
			for (int j = 0; j < R; j++) { a[j].x -= sa.x; a[j].y -= sa.y; a[j].z -= sa.z; a[j].w -= sa.w; }
		}
#pragma unroll
		for (int j = 0; j < R; j++) {
			acc1(a[j].x, g & 0xffu, s[j], n[j]); acc1(a[j].y, g & 0xff00u, s[j], n[j]);
			acc1(a[j].z, g & 0xff0000u, s[j], n[j]); acc1(a[j].w, g & 0xff000000u, s[j], n[j]);
		}
	}
#pragma unroll
	for (int j = 0; j < R; j++) {
		const double ss = wave_sum_f64(s[j]);
		const int nn = wave_sum_i32(n[j]);
		mean[j] = (nn > 0) ? ss / (double)nn : __builtin_nan("");
	}
}

__device__ __forceinline__ double row_mean_scalar(const float* row, const float* sub, const unsigned char* good, int n_cad, int lane)
{
	double s = 0.0;
	int n = 0;
	for (int k = lane; k < n_cad; k += 64) acc1(sub ? (row[k] - sub[k]) : row[k], good[k], s, n);
	s = wave_sum_f64(s);
	n = wave_sum_i32(n);
	return (n > 0) ? s / (double)n : __builtin_nan("");
}

} // namespace tp_sum
