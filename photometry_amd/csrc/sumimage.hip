// sumimage.hip -- A1: the sum image (mean over good-quality cadences, NaN excluded).
//
// Replaces BasePhotometry.sumimage (photometry/BasePhotometry.py:1008-1019) and the FFI
// accumulation of photometry/prepare.py:450-453,459.
//
// Mapping (gfx950): one 256-thread workgroup (4 wavefronts) per target.  A pixel's time series
// is contiguous in HBM (time-fastest cube), so each wavefront owns a pixel row at a time and
// streams it with 128-bit loads, 64 lanes x 16 B = 1 KiB per instruction, fully coalesced.
// The good-quality flags are staged once per workgroup in LDS (one byte per cadence).  Each lane
// keeps a float64 partial sum and an int count; one 64-lane DPP/shuffle reduction per pixel.
// HBM-bound: algorithmic bytes per target = P*T*4 (images) + T*4 (quality) + P*8 (output).
#include "common.h"
#include <cmath>

namespace {

constexpr int kBlock = 256;
constexpr int kWaves = kBlock / 64;

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

template <bool VEC4>
__global__ __launch_bounds__(kBlock) void tp_sumimage_kernel(
	const float* __restrict__ images, const int32_t* __restrict__ quality, int64_t quality_stride,
	uint32_t bitmask, double* __restrict__ out, int n_cad, int n_pix, int64_t t_pitch,
	const float* __restrict__ subtract, int64_t subtract_pitch)
{
	extern __shared__ __align__(16) unsigned char good[]; // [round_up(n_cad, 4)]
	const int target = blockIdx.x;
	const int tid = threadIdx.x;
	const int lane = tid & 63;
	const int wave = tid >> 6;
	const int32_t* q = quality + (int64_t)target * quality_stride;
	const int n_cad4 = (n_cad + 3) & ~3;
	for (int k = tid; k < n_cad4; k += kBlock)
		good[k] = (k < n_cad && ((uint32_t)q[k] & bitmask) == 0u) ? 1 : 0;
	__syncthreads();

	const float* base = images + (int64_t)target * n_pix * t_pitch;
	double* o = out + (int64_t)target * n_pix;
	// optional on-the-fly background subtraction (prepare.py:419-420: float32 image - float32 background)
	const float* sub = subtract ? (subtract + (int64_t)target * subtract_pitch) : nullptr;
	const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);

	for (int p = wave; p < n_pix; p += kWaves) {
		const float* row = base + (int64_t)p * t_pitch;
		double s = 0.0;
		int n = 0;
		if (VEC4) {
			const int nq = n_cad4 >> 2; // quads (the tail quad reads into the row padding: pitch % 4 == 0)
			const float4* row4 = reinterpret_cast<const float4*>(row);
			const float4* sub4 = reinterpret_cast<const float4*>(sub);
			const uint32_t* good4 = reinterpret_cast<const uint32_t*>(good);
			int qd = lane;
			// two independent 1 KiB loads in flight per wavefront per iteration
			for (; qd + 64 < nq; qd += 128) {
				float4 a = row4[qd];
				float4 b = row4[qd + 64];
				const float4 sa = sub ? sub4[qd] : zero4;
				const float4 sb = sub ? sub4[qd + 64] : zero4;
				if (sub) { a.x -= sa.x; a.y -= sa.y; a.z -= sa.z; a.w -= sa.w; b.x -= sb.x; b.y -= sb.y; b.z -= sb.z; b.w -= sb.w; }
				uint32_t ga = good4[qd];
				uint32_t gb = good4[qd + 64];
				acc1(a.x, ga & 0xffu, s, n); acc1(a.y, ga & 0xff00u, s, n);
				acc1(a.z, ga & 0xff0000u, s, n); acc1(a.w, ga & 0xff000000u, s, n);
				acc1(b.x, gb & 0xffu, s, n); acc1(b.y, gb & 0xff00u, s, n);
				acc1(b.z, gb & 0xff0000u, s, n); acc1(b.w, gb & 0xff000000u, s, n);
			}
			for (; qd < nq; qd += 64) {
				float4 a = row4[qd];
				if (sub) { const float4 sa = sub4[qd]; a.x -= sa.x; a.y -= sa.y; a.z -= sa.z; a.w -= sa.w; }
				uint32_t ga = good4[qd];
				acc1(a.x, ga & 0xffu, s, n); acc1(a.y, ga & 0xff00u, s, n);
				acc1(a.z, ga & 0xff0000u, s, n); acc1(a.w, ga & 0xff000000u, s, n);
			}
		} else {
			for (int k = lane; k < n_cad; k += 64) acc1(sub ? (row[k] - sub[k]) : row[k], good[k], s, n);
		}
		s = wave_sum_f64(s);
		n = wave_sum_i32(n);
		if (lane == 0) o[p] = (n > 0) ? s / (double)n : __builtin_nan("");
	}
}

} // namespace

extern "C" int tp_sumimage(tp_ctx* ctx, const tp_cube_desc* desc, const float* d_images,
	const int32_t* d_quality, int64_t quality_target_stride, uint32_t bitmask,
	const float* d_subtract, int64_t subtract_pitch, double* d_sumimage)
{
	TP_CHECK_CTX(ctx);
	TP_API_BEGIN
	TP_REQUIRE(ctx, tp_desc_ok(desc), "tp_sumimage: bad cube descriptor");
	TP_REQUIRE(ctx, d_images && d_quality && d_sumimage, "tp_sumimage: null pointer");
	TP_REQUIRE(ctx, quality_target_stride == 0 || quality_target_stride >= desc->n_cad, "tp_sumimage: bad quality stride");
	if (desc->n_targets == 0) return TP_OK;
	const int n_pix = desc->height * desc->width;
	const size_t shmem = (size_t)((desc->n_cad + 3) & ~3) + 16;
	TP_REQUIRE(ctx, shmem <= 160 * 1024, "tp_sumimage: n_cad too large for the LDS quality table");
	// VEC4 reads the row padding of the last quad: needs pitch % 4 == 0 (so the quad is inside the pitch)
	TP_REQUIRE(ctx, d_subtract == nullptr || subtract_pitch >= desc->n_cad, "tp_sumimage: bad subtract pitch");
	const bool vec4 = tp_vec4_ok(d_images, desc->t_pitch) && (d_subtract == nullptr || (tp_vec4_ok(d_subtract, subtract_pitch)));
	dim3 grid((unsigned)desc->n_targets), block(kBlock);
	if (vec4) {
		TP_LAUNCH(ctx, TPK_SUMIMAGE, tp_sumimage_kernel<true>, grid, block, shmem,
			d_images, d_quality, quality_target_stride, bitmask, d_sumimage, desc->n_cad, n_pix, desc->t_pitch, d_subtract, subtract_pitch);
	} else {
		TP_LAUNCH(ctx, TPK_SUMIMAGE, tp_sumimage_kernel<false>, grid, block, shmem,
			d_images, d_quality, quality_target_stride, bitmask, d_sumimage, desc->n_cad, n_pix, desc->t_pitch, d_subtract, subtract_pitch);
	}
	TP_LAUNCH_CHECK(ctx, "tp_sumimage_kernel");
	return TP_OK;
	TP_API_END(ctx)
}
