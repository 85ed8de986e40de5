// background.hip -- B* (stamp-level background per cadence), B2 (time smoothing), B3 (subtraction).
//
// B2 / B3 replace photometry/prepare.py:317-335 (nanmean over a +-w cadence window, float32) and
// :419-425 (image -= background; manual-exclude pixels -> NaN).
//
// B* is the build-defined stamp analogue of backgrounds.fit_background (photometry/backgrounds.py:
// 52-211; Background2D on 64x64 tiles cannot be applied to a 15x15 stamp): pixel mask of
// backgrounds.py:89-94, one mesh cell = the stamp, SigmaClip(3 sigma, 5 iterations, median/std),
// SExtractor mode estimator, "more than 50 % masked -> no estimate".  See oracle/backgrounds.py.
//
// Mapping (gfx950).  The cube is time-fastest, so consecutive lanes read consecutive cadences of a
// pixel's time series (coalesced) and NO LDS transposition is needed to get a frame into registers.
// A frame (one cadence of one target, <= 256 pixels) is owned by a QUAD of lanes, 64 pixel values per
// lane in VGPRs; a wavefront holds 16 frames, a 256-thread workgroup 64 consecutive cadences.
//   1. each lane sorts its 64 values with a fully unrolled bitonic network (v_min_f32 / v_max_f32 on compile-time
//      register indices: no divergence, no memory traffic);
//   2. the four sorted runs are merged across the quad: a MIRROR stage (register j against register 63-j of the partner
//      lane, DPP quad_perm) turns two ascending runs into two bitonic halves, all further stages are ascending merges;
//   3. one predicated pass over the registers gives the float64 sums of the frame; the sorted values are staged in LDS and
//      indexed by rank for the clipping passes (median = two reads, bounds by binary search, sums updated by the ranks
//      that leave the kept range).
// Measured alternatives (C3 cube): one thread per frame with 256 registers 33.6 ms (instruction-cache bound); clipping as
// two full register passes per iteration 14.1 ms; a register prefetch buffer 15.0 ms (one wave per SIMD); loads through a
// coalescing LDS transpose tile (256-byte segments, two workgroup barriers) 14.2 ms; this version 10.8 ms.
#include "common.h"
#include <cmath>

namespace {

// v_min_f32 / v_max_f32 without the NaN-quieting canonicalisation fminf / fmaxf carry (no value is ever NaN here: masked
// pixels are +inf sentinels); plain asm so that the scheduler is still free to reorder
__device__ __forceinline__ float tp_min(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float tp_max(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

// Compile-time bitonic network on a register array.  One template instantiation per (size, stride)
// stage keeps every unrolled body small enough for the optimiser's full-unroll budget.
template <int N, int SIZE, int STRIDE>
struct BitonicStage {
	static __device__ __forceinline__ void run(float (&v)[N]) {
#pragma unroll
		for (int t = 0; t < N / 2; ++t) {
			const int lo = (t / STRIDE) * (STRIDE * 2) + (t % STRIDE);
			const int hi = lo + STRIDE;
			const bool up = ((lo & SIZE) == 0);
			const float a = v[lo], b = v[hi];
			const float mn = tp_min(a, b), mx = tp_max(a, b);
			v[lo] = up ? mn : mx;
			v[hi] = up ? mx : mn;
		}
		BitonicStage<N, SIZE, STRIDE / 2>::run(v);
	}
};
template <int N, int SIZE>
struct BitonicStage<N, SIZE, 0> {
	static __device__ __forceinline__ void run(float (&)[N]) {}
};
template <int N, int SIZE>
struct BitonicLevel {
	static __device__ __forceinline__ void run(float (&v)[N]) {
		BitonicLevel<N, SIZE / 2>::run(v);
		BitonicStage<N, SIZE, SIZE / 2>::run(v);
	}
};
template <int N>
struct BitonicLevel<N, 1> {
	static __device__ __forceinline__ void run(float (&)[N]) {}
};

struct BkgArgs {
	const float* raw; float* out; int n_cad; int n_pix; int64_t t_pitch; int64_t out_pitch;
	float flux_cutoff; float exclude_fraction;
};

// SExtractorBackground (photutils 1.3.0) on the clipped statistics
__device__ __forceinline__ float sextractor_mode(double med, double mean, double sd) {
	double bkg;
	if (sd == 0.0) bkg = mean;
	else if (fabs(mean - med) / sd < 0.3) bkg = 2.5 * med - 1.5 * mean;
	else bkg = med;
	return (float)bkg;
}

// DPP quad permutes: value of lane^1 / lane^2 / lane^3 within the quad
__device__ __forceinline__ float quad_xor1(float x) {
	return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xF, 0xF, false)); // quad_perm [1,0,3,2]
}
__device__ __forceinline__ float quad_xor2(float x) {
	return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xF, 0xF, false)); // quad_perm [2,3,0,1]
}
__device__ __forceinline__ float quad_rev(float x) {
	return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x1B, 0xF, 0xF, false)); // quad_perm [3,2,1,0]
}
// compare-exchange of register j with register j of lane^1: the lower lane keeps the minimum
__device__ __forceinline__ void cross_stage_xor1(float (&v)[64], bool keepmin) {
#pragma unroll
	for (int j = 0; j < 64; ++j) {
		const float p = quad_xor1(v[j]);
		v[j] = keepmin ? tp_min(v[j], p) : tp_max(v[j], p);
	}
}
// MIRROR stage of a merge: register j against register 63-j of the partner lane (lane^1 for runs of 64, lane^3 for runs of
// 128).  Two ascending runs become two bitonic halves whose merges are all ascending -- no direction flags, no sign flips.
template <bool ACROSS_QUAD>
__device__ __forceinline__ void mirror_stage(float (&v)[64], bool keepmin) {
	float w[64];
#pragma unroll
	for (int j = 0; j < 64; ++j) {
		const float p = ACROSS_QUAD ? quad_rev(v[63 - j]) : quad_xor1(v[63 - j]);
		w[j] = keepmin ? tp_min(v[j], p) : tp_max(v[j], p);
	}
#pragma unroll
	for (int j = 0; j < 64; ++j) v[j] = w[j];
}
// ascending bitonic MERGE of a lane's 64 values (strides 32..1)
__device__ __forceinline__ void local_merge(float (&v)[64]) { BitonicStage<64, 64, 32>::run(v); }

constexpr int kFramesPerWave = 16;
constexpr int kBkgThreads = 128;
constexpr int kFramesPerBlock = kFramesPerWave * (kBkgThreads / 64);

__device__ __forceinline__ double quad_sum(double x) { x += __shfl_xor(x, 1, 64); x += __shfl_xor(x, 2, 64); return x; }
__device__ __forceinline__ float quad_sum(float x) { x += quad_xor1(x); x += quad_xor2(x); return x; }
__device__ __forceinline__ int quad_sum(int x) { x += __shfl_xor(x, 1, 64); x += __shfl_xor(x, 2, 64); return x; }

__global__ __launch_bounds__(kBkgThreads) void tp_bkg_stamp_kernel(BkgArgs a, int frame_stride)
{
	extern __shared__ __align__(16) float s_sorted[]; // [waves][kFramesPerWave][frame_stride]
	const int target = blockIdx.x;
	const int tid = threadIdx.x;
	const int wave = tid >> 6, lane = tid & 63;
	const int f = lane >> 2, q = lane & 3;   // frame within the wavefront, quarter of the frame
	const int k = blockIdx.y * kFramesPerBlock + wave * kFramesPerWave + f;
	const bool active = k < a.n_cad;
	const float* base = a.raw + (int64_t)target * a.n_pix * a.t_pitch + (active ? k : 0);
	const float inf = __builtin_inff();
	float* fr = s_sorted + (size_t)(wave * kFramesPerWave + f) * frame_stride;
	float v[64];
	int n = 0;
#pragma unroll
	for (int j = 0; j < 64; ++j) {
		const int i = q * 64 + j;
		float x = inf;
		if (i < a.n_pix) x = base[(int64_t)i * a.t_pitch];
		// backgrounds.py:91-94: mask = ~isfinite | > flux_cutoff | < 0   (+inf padding is "masked" too)
		const bool ok = (fabsf(x) <= 3.402823466e+38f) && !(x > a.flux_cutoff) && !(x < 0.f);
		n += ok ? 1 : 0;
		v[j] = ok ? x : inf;
	}
	n = quad_sum(n);

	// --- distributed bitonic sort of the 256 values of the quad: global index (= rank) g = q*64 + j ---
	BitonicLevel<64, 32>::run(v);                       // sizes 2..32: directions fixed by the local index
	local_merge(v);                                     // size 64: every lane ascending
	mirror_stage<false>(v, (q & 1) == 0);               // size 128: mirror against lane^1 ...
	local_merge(v);                                     //           ... then ascending merges: lanes (0,1) and (2,3) sorted
	mirror_stage<true>(v, (q & 2) == 0);                // size 256: mirror against lane^3,
	cross_stage_xor1(v, (q & 1) == 0);                  //           stride 64 against lane^1,
	local_merge(v);                                     //           strides 32..1

	// --- sigma clipping.  The kept set is always a contiguous rank range [lo_i, hi_i) of the sorted values.  One
	// predicated pass over the lane's 64 registers gives the float64 sums of the whole frame; the sorted values are
	// then staged in LDS (1 KiB per frame) where they can be indexed by rank: the median is two reads, the new bounds
	// are two binary searches for the float32-exact thresholds, and the sums are UPDATED by subtracting only the
	// ranks that leave the range (shared by the four lanes of the quad) -- a few dozen LDS reads per clipping pass
	// instead of two 64-register passes of float64 arithmetic.  All four lanes of a quad carry the same scalars.
	// A wavefront stages and reads only its own 16 frames and its LDS operations execute in order: no workgroup
	// barrier.  Only the ranks below frame_stride are staged (ranks >= n_pix are +inf sentinels nobody reads).
#pragma unroll
	for (int j = 0; j < 64; j += 4)
		if (q * 64 + j < frame_stride) *reinterpret_cast<float4*>(fr + q * 64 + j) = make_float4(v[j], v[j + 1], v[j + 2], v[j + 3]);
	const int nmasked = a.n_pix - n;
	const bool usable = active && (n > 0) && !((float)nmasked > a.exclude_fraction * (float)a.n_pix);
	const int rbase = q * 64;
	double s1 = 0.0, s2 = 0.0;
#pragma unroll
	for (int j = 0; j < 64; ++j) {
		const bool in = (rbase + j) < n;              // ranks >= n are the +inf sentinels
		float xf = in ? v[j] : 0.f;
		asm volatile("" : "+v"(xf));   // keep the conversion in the loop (else 64 doubles stay live)
		const double x = (double)xf;
		s1 += x;
		s2 = __builtin_fma(x, x, s2);
	}
	s1 = quad_sum(s1); s2 = quad_sum(s2);
	__builtin_amdgcn_wave_barrier();
	float result = __builtin_nanf("");
	if (usable) {
		int lo_i = 0, hi_i = n;
		double med = 0.0, mean = 0.0, sd = 0.0;
#pragma unroll 1
		for (int it = 0; it <= 5; ++it) {
			const int m = hi_i - lo_i;
			const int m1 = lo_i + (m >> 1);          // upper middle rank
			const int m0 = (m & 1) ? m1 : (m1 - 1);  // lower middle rank
			med = ((double)fr[m0] + (double)fr[m1]) / 2.0;
			mean = s1 / (double)m;
			double var = s2 / (double)m - mean * mean;
			if (var < 0.0) var = 0.0;
			sd = sqrt(var);
			if (it == 5) break;                        // maxiters = 5 clipping passes, then the final statistics
			// exact float32 thresholds: for float x, (double)x < lo <=> x < round_up(lo); (double)x > hi <=> x > round_down(hi)
			const float lo_f = __double2float_ru(med - 3.0 * sd);
			const float hi_f = __double2float_rd(med + 3.0 * sd);
			// new_lo = first rank in range whose value is >= lo_f, new_hi = first rank whose value is > hi_f
			int a0 = lo_i, b0 = hi_i, a1 = lo_i, b1 = hi_i;
#pragma unroll
			for (int step = 0; step < 8; ++step) {     // the range holds at most 256 ranks
				const int mid0 = (a0 + b0) >> 1, mid1 = (a1 + b1) >> 1;
				const float x0 = fr[(a0 < b0) ? mid0 : lo_i], x1 = fr[(a1 < b1) ? mid1 : lo_i];
				if (a0 < b0) { if (x0 < lo_f) a0 = mid0 + 1; else b0 = mid0; }
				if (a1 < b1) { if (x1 > hi_f) b1 = mid1; else a1 = mid1 + 1; }
			}
			if (a0 < b0) { if (fr[(a0 + b0) >> 1] < lo_f) a0 = ((a0 + b0) >> 1) + 1; else b0 = (a0 + b0) >> 1; }
			if (a1 < b1) { if (fr[(a1 + b1) >> 1] > hi_f) b1 = (a1 + b1) >> 1; else a1 = ((a1 + b1) >> 1) + 1; }
			const int new_lo = a0, new_hi = a1;
			if (new_lo == lo_i && new_hi == hi_i) break; // nchanged == 0: the statistics of this range are final
			// subtract the ranks that leave the range: [lo_i, new_lo) and [new_hi, hi_i), one of every four per lane
			double r1 = 0.0, r2 = 0.0;
			for (int r = lo_i + q; r < new_lo; r += 4) { const double x = (double)fr[r]; r1 += x; r2 = __builtin_fma(x, x, r2); }
			for (int r = new_hi + q; r < hi_i; r += 4) { const double x = (double)fr[r]; r1 += x; r2 = __builtin_fma(x, x, r2); }
			r1 = quad_sum(r1); r2 = quad_sum(r2);
			s1 -= r1; s2 -= r2;
			lo_i = new_lo;
			hi_i = new_hi;
		}
		result = sextractor_mode(med, mean, sd);
	}
	if (q == 0 && active) a.out[(int64_t)target * a.out_pitch + k] = result;
}

// Generic fallback for stamps with more than 256 pixels: one wavefront per (target, cadence), values
// sorted in LDS.  Correct for any size that fits LDS; not tuned (large stamps are the bright-star tail).
__global__ __launch_bounds__(64) void tp_bkg_stamp_generic_kernel(BkgArgs a, int np2)
{
	extern __shared__ float sv[];
	const int target = blockIdx.x;
	const int k = blockIdx.y * 1 + blockIdx.z * 65535;
	const int lane = threadIdx.x;
	if (k >= a.n_cad) return;
	const float* base = a.raw + (int64_t)target * a.n_pix * a.t_pitch + k;
	const float inf = __builtin_inff();
	__shared__ int s_n;
	if (lane == 0) s_n = 0;
	__syncthreads();
	int cnt = 0;
	for (int i = lane; i < np2; i += 64) {
		float x = inf;
		if (i < a.n_pix) {
			x = base[(int64_t)i * a.t_pitch];
			const bool ok = (fabsf(x) <= 3.402823466e+38f) && !(x > a.flux_cutoff) && !(x < 0.f);
			cnt += ok ? 1 : 0;
			x = ok ? x : inf;
		}
		sv[i] = x;
	}
	atomicAdd(&s_n, cnt);
	__syncthreads();
	for (int size = 2; size <= np2; size <<= 1) {
		for (int stride = size >> 1; stride > 0; stride >>= 1) {
			for (int t = lane; t < np2 / 2; t += 64) {
				const int lo = ((t & ~(stride - 1)) << 1) | (t & (stride - 1)); // stride is a power of two
				const int hi = lo + stride;
				const bool up = ((lo & size) == 0);
				const float x = sv[lo], y = sv[hi];
				if ((x > y) == up) { sv[lo] = y; sv[hi] = x; }
			}
			__syncthreads();
		}
	}
	if (lane != 0) return;
	const int n = s_n;
	float result = __builtin_nanf("");
	const int nmasked = a.n_pix - n;
	if (n > 0 && !((float)nmasked > a.exclude_fraction * (float)a.n_pix)) {
		int lo_i = 0, hi_i = n;
		double med = 0.0, mean = 0.0, sd = 0.0;
		for (int it = 0; it <= 5; ++it) {
			const int m = hi_i - lo_i;
			const int m1 = lo_i + (m >> 1);
			const int m0 = (m & 1) ? m1 : (m1 - 1);
			double s1 = 0.0, s2 = 0.0;
			for (int i = lo_i; i < hi_i; ++i) { const double x = (double)sv[i]; s1 += x; s2 += x * x; }
			med = ((double)sv[m0] + (double)sv[m1]) / 2.0;
			mean = s1 / (double)m;
			double var = s2 / (double)m - mean * mean;
			if (var < 0.0) var = 0.0;
			sd = sqrt(var);
			if (it == 5) break;
			const double lo = med - 3.0 * sd, hi = med + 3.0 * sd;
			int below = 0, above = 0;
			for (int i = lo_i; i < hi_i; ++i) { const double x = (double)sv[i]; below += (x < lo); above += (x > hi); }
			if (below == 0 && above == 0) break;
			lo_i += below;
			hi_i -= above;
		}
		result = sextractor_mode(med, mean, sd);
	}
	a.out[(int64_t)target * a.out_pitch + k] = result;
}

// B2: bottleneck.nanmean over the window [k-w, k+w] clipped to the series (float32 accumulate)
__global__ __launch_bounds__(256) void tp_bkg_smooth_kernel(const float* __restrict__ in, float* __restrict__ out,
	int n_cad, int64_t pitch, int w)
{
	const int target = blockIdx.x;
	const int k = blockIdx.y * blockDim.x + threadIdx.x;
	if (k >= n_cad) return;
	const float* x = in + (int64_t)target * pitch;
	const int i1 = (k - w > 0) ? (k - w) : 0;
	const int i2 = (k + w + 1 < n_cad) ? (k + w + 1) : n_cad;
	float asum = 0.f;
	int cnt = 0;
	for (int n = i1; n < i2; ++n) {
		const float v = x[n];
		if (v == v) { asum += v; cnt++; }
	}
	out[(int64_t)target * pitch + k] = (cnt > 0) ? (asum / (float)cnt) : __builtin_nanf("");
}

// B3: images = raw - bkg[k]; optional manual-exclude flags (PixelQualityFlags.ManualExclude = 2) -> NaN
template <bool VEC4>
__global__ __launch_bounds__(256) void tp_bkg_subtract_kernel(const float* __restrict__ raw, const float* __restrict__ raw_err,
	const float* __restrict__ bkg, int64_t bkg_pitch, const uint8_t* __restrict__ flags, uint32_t flag_mask,
	float* __restrict__ img, float* __restrict__ err, int n_cad, int n_pix, int64_t t_pitch)
{
	const int target = blockIdx.x;
	const int64_t nq = VEC4 ? (t_pitch >> 2) : t_pitch;
	const int64_t idx = (int64_t)blockIdx.y * blockDim.x + threadIdx.x;
	if (idx >= (int64_t)n_pix * nq) return;
	const int p = (int)(idx / nq);
	const int64_t q = idx - (int64_t)p * nq;
	const int64_t off = ((int64_t)target * n_pix + p) * t_pitch + (VEC4 ? q * 4 : q);
	const float* b = bkg + (int64_t)target * bkg_pitch + (VEC4 ? q * 4 : q);
	constexpr int V = VEC4 ? 4 : 1;
	float x[V], e[V], bb[V];
	if (VEC4) {
		const float4 t = *reinterpret_cast<const float4*>(raw + off); x[0] = t.x; x[1 % V] = t.y; x[2 % V] = t.z; x[3 % V] = t.w;
		const float4 u = *reinterpret_cast<const float4*>(b); bb[0] = u.x; bb[1 % V] = u.y; bb[2 % V] = u.z; bb[3 % V] = u.w;
		if (raw_err) { const float4 s = *reinterpret_cast<const float4*>(raw_err + off); e[0] = s.x; e[1 % V] = s.y; e[2 % V] = s.z; e[3 % V] = s.w; }
	} else {
		x[0] = raw[off]; bb[0] = b[0];
		if (raw_err) e[0] = raw_err[off];
	}
#pragma unroll
	for (int j = 0; j < V; ++j) {
		const int64_t kk = (VEC4 ? q * 4 : q) + j;
		float r = x[j] - bb[j];
		bool excl = false;
		if (flags && kk < n_cad) excl = (flags[((int64_t)target * n_pix + p) * n_cad + kk] & flag_mask) != 0;
		x[j] = excl ? __builtin_nanf("") : r;
		if (raw_err) e[j] = excl ? __builtin_nanf("") : e[j];
	}
	if (VEC4) {
		*reinterpret_cast<float4*>(img + off) = make_float4(x[0], x[1 % V], x[2 % V], x[3 % V]);
		if (raw_err && err) *reinterpret_cast<float4*>(err + off) = make_float4(e[0], e[1 % V], e[2 % V], e[3 % V]);
	} else {
		img[off] = x[0];
		if (raw_err && err) err[off] = e[0];
	}
}

} // namespace

extern "C" int tp_background_stamp(tp_ctx* ctx, const tp_cube_desc* desc, const float* d_raw,
	double flux_cutoff, double exclude_percentile, float* d_bkg, int64_t bkg_pitch)
{
	TP_CHECK_CTX(ctx);
	TP_API_BEGIN
	TP_REQUIRE(ctx, tp_desc_ok(desc), "tp_background_stamp: bad cube descriptor");
	TP_REQUIRE(ctx, d_raw && d_bkg, "tp_background_stamp: null pointer");
	TP_REQUIRE(ctx, bkg_pitch >= desc->n_cad, "tp_background_stamp: bkg_pitch < n_cad");
	if (desc->n_targets == 0 || desc->n_cad == 0) return TP_OK;
	BkgArgs a;
	a.raw = d_raw; a.out = d_bkg; a.n_cad = desc->n_cad; a.n_pix = desc->height * desc->width;
	a.t_pitch = desc->t_pitch; a.out_pitch = bkg_pitch;
	a.flux_cutoff = (float)flux_cutoff; a.exclude_fraction = (float)(exclude_percentile / 100.0);
	if (a.n_pix <= 256) {
		// staged frame: the ranks that can hold a value, 16-byte aligned rows, +4 floats so that consecutive frames
		// start 4 banks apart
		const int frame_stride = ((a.n_pix + 3) & ~3) + 4;
		const size_t shmem = (size_t)kFramesPerBlock * frame_stride * sizeof(float);
		dim3 block(kBkgThreads), grid((unsigned)desc->n_targets, (unsigned)((desc->n_cad + kFramesPerBlock - 1) / kFramesPerBlock));
		TP_LAUNCH(ctx, TPK_BKG_STAMP, tp_bkg_stamp_kernel, grid, block, shmem, a, frame_stride);
	} else {
		int np2 = 1;
		while (np2 < a.n_pix) np2 <<= 1;
		const size_t shmem = (size_t)np2 * sizeof(float);
		TP_REQUIRE(ctx, shmem <= 150 * 1024, "tp_background_stamp: stamp too large");
		TP_REQUIRE(ctx, desc->n_cad <= 2147483647, "tp_background_stamp: too many cadences");
		if (shmem > 64 * 1024)
			TP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(tp_bkg_stamp_generic_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
		const unsigned gy = (unsigned)((desc->n_cad < 65535) ? desc->n_cad : 65535);
		const unsigned gz = (unsigned)((desc->n_cad + 65534) / 65535);
		dim3 block(64), grid((unsigned)desc->n_targets, gy, gz);
		TP_LAUNCH(ctx, TPK_BKG_STAMP, tp_bkg_stamp_generic_kernel, grid, block, shmem, a, np2);
	}
	TP_LAUNCH_CHECK(ctx, "tp_bkg_stamp_kernel");
	return TP_OK;
	TP_API_END(ctx)
}

extern "C" int tp_smooth_time(tp_ctx* ctx, int32_t n_targets, int32_t n_cad, int64_t pitch, int32_t time_smooth,
	const float* d_in, float* d_out)
{
	TP_CHECK_CTX(ctx);
	TP_API_BEGIN
	TP_REQUIRE(ctx, n_targets >= 0 && n_cad >= 0 && pitch >= n_cad && time_smooth >= 1, "tp_smooth_time: bad geometry");
	TP_REQUIRE(ctx, d_in && d_out && d_in != d_out, "tp_smooth_time: null or aliased pointers");
	if (n_targets == 0 || n_cad == 0) return TP_OK;
	dim3 block(256), grid((unsigned)n_targets, (unsigned)((n_cad + 255) / 256));
	TP_LAUNCH(ctx, TPK_BKG_SMOOTH, tp_bkg_smooth_kernel, grid, block, 0, d_in, d_out, (int)n_cad, pitch, (int)(time_smooth / 2));
	TP_LAUNCH_CHECK(ctx, "tp_bkg_smooth_kernel");
	return TP_OK;
	TP_API_END(ctx)
}

extern "C" int tp_subtract_background(tp_ctx* ctx, const tp_cube_desc* desc, const float* d_raw, const float* d_raw_err,
	const float* d_bkg, int64_t bkg_pitch, const uint8_t* d_pixel_flags, uint32_t flag_mask,
	float* d_images, float* d_images_err)
{
	TP_CHECK_CTX(ctx);
	TP_API_BEGIN
	TP_REQUIRE(ctx, tp_desc_ok(desc), "tp_subtract_background: bad cube descriptor");
	TP_REQUIRE(ctx, d_raw && d_bkg && d_images, "tp_subtract_background: null pointer");
	TP_REQUIRE(ctx, bkg_pitch >= desc->t_pitch, "tp_subtract_background: bkg_pitch must be >= the cube's t_pitch");
	if (desc->n_targets == 0 || desc->n_cad == 0) return TP_OK;
	const int n_pix = desc->height * desc->width;
	bool vec4 = tp_vec4_ok(d_raw, desc->t_pitch) && tp_vec4_ok(d_images, desc->t_pitch) && tp_vec4_ok(d_bkg, bkg_pitch)
;
	if (d_raw_err) vec4 = vec4 && tp_vec4_ok(d_raw_err, desc->t_pitch) && (!d_images_err || tp_vec4_ok(d_images_err, desc->t_pitch));
	const int64_t nq = vec4 ? (desc->t_pitch / 4) : desc->t_pitch;
	const int64_t per_target = (int64_t)n_pix * nq;
	TP_REQUIRE(ctx, (per_target + 255) / 256 <= 65535, "tp_subtract_background: stamp cube too large");
	dim3 block(256), grid((unsigned)desc->n_targets, (unsigned)((per_target + 255) / 256));
	if (vec4) {
		TP_LAUNCH(ctx, TPK_BKG_SUBTRACT, tp_bkg_subtract_kernel<true>, grid, block, 0, d_raw, d_raw_err, d_bkg, bkg_pitch,
			d_pixel_flags, flag_mask, d_images, d_images_err, desc->n_cad, n_pix, desc->t_pitch);
	} else {
		TP_LAUNCH(ctx, TPK_BKG_SUBTRACT, tp_bkg_subtract_kernel<false>, grid, block, 0, d_raw, d_raw_err, d_bkg, bkg_pitch,
			d_pixel_flags, flag_mask, d_images, d_images_err, desc->n_cad, n_pix, desc->t_pitch);
	}
	TP_LAUNCH_CHECK(ctx, "tp_bkg_subtract_kernel");
	return TP_OK;
	TP_API_END(ctx)
}
