// fullframe.hip -- the full-frame (prepare stage) background path on a frame stack [T][R][C] resident in HBM.
//
// B1  tp_background_mesh + tp_background_zoom: the square component of backgrounds.fit_background (photometry/backgrounds.py:52-211):
//     pixel mask (:89-97), photutils Background2D on 64 x 64 boxes with SigmaClip(3, maxiters = 5) and the SExtractor estimator
//     (:200-206), cubic-spline zoom of the filtered mesh.  Alone it is the plain-image branch (bkgiters = 1, :156-157); for TESS
//     frames it alternates with the radial component of csrc/radial.hip, whose image comes in as d_subtract.
//     The mesh statistics (every pixel of every frame: the heavy part), the finishing of the 32 x 32 mesh (exclusion of mostly-masked
//     boxes -- counted AFTER the sigma clip, as photutils does --, IDW fill, 3 x 3 median filter, spline prefilter:
//     tp_mesh_finish_kernel) and the zoom back to full resolution all run here.  For TESS frames the radial component is not read
//     from a stored image but evaluated in the mesh kernel from its ring profile (tp_background_mesh_radial, fullframe_dev.h).
// B2  tp_frames_smooth_time: prepare.py:317-335 for per-pixel background images (nanmean over +-w frames, float32).
// B3  tp_frames_subtract:    prepare.py:419-425 (image -= background, manual-exclude pixels -> NaN).
// A1  tp_frames_sumimage:    prepare.py:450-453, 459 (mean over the good-quality frames, NaN pixels skipped).
//
// tp_background_mesh, gfx950: one 256-thread workgroup per (box, frame).  The box (64 rows of 256 contiguous bytes) is loaded
// coalesced, masked pixels become +inf, and the 4096 values are sorted in LDS by a bitonic network in which every thread keeps
// 16 consecutive keys: the strides 8..1 of every merge level run in registers (v_min / v_max on compile-time indices), only the
// 36 stages with stride >= 16 exchange through LDS (36 + 9 barriers instead of 78).  On the sorted keys the kept set of the
// sigma clipping is a rank range: median by index, float64 two-pass mean / std by block reductions, bounds by counting.
// LDS-exchange bound (not HBM): the median of 4096 keys needs their order.
#include "common.h"
#include "fullframe_dev.h"
#include <cmath>

namespace {

constexpr int kMeshThreads = 256;
constexpr int kKeysPerThread = 16;
constexpr int kMeshKeys = kMeshThreads * kKeysPerThread;   // 4096 = 64 x 64
constexpr int kMaxFusedKnots = 72;                         // knots + coefficients + reciprocals of a frame's ring profile staged in LDS by the mesh kernel (64 rings + 4 + pad: 4.6 KB)

__device__ __forceinline__ double mesh_block_sum(double v, double* red) {
	const int tid = threadIdx.x;
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
	__syncthreads();
	if ((tid & 63) == 0) red[tid >> 6] = v;
	__syncthreads();
	return (red[0] + red[1]) + (red[2] + red[3]);
}
// two sums at once (red: 8 doubles)
__device__ __forceinline__ void mesh_block_sum2(double& x, double& y, double* red) {
	const int tid = threadIdx.x;
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) { x += __shfl_down(x, off, 64); y += __shfl_down(y, off, 64); }
	__syncthreads();
	if ((tid & 63) == 0) { red[tid >> 6] = x; red[4 + (tid >> 6)] = y; }
	__syncthreads();
	x = (red[0] + red[1]) + (red[2] + red[3]);
	y = (red[4] + red[5]) + (red[6] + red[7]);
}

// ascending / descending compare-exchange network on 16 registers for strides 8, 4, 2, 1 (one merge tail)
template <int STRIDE>
__device__ __forceinline__ void local_tail(float (&v)[kKeysPerThread], bool up) {
#pragma unroll
	for (int t = 0; t < kKeysPerThread / 2; ++t) {
		const int lo = (t / STRIDE) * (STRIDE * 2) + (t % STRIDE), hi = lo + STRIDE;
		const float a = v[lo], b = v[hi];
		const float mn = fminf(a, b), mx = fmaxf(a, b);
		v[lo] = up ? mn : mx;
		v[hi] = up ? mx : mn;
	}
	if constexpr (STRIDE > 1) local_tail<STRIDE / 2>(v, up);
}

struct MeshArgs {
	const float* frames; int n_rows, n_cols; int64_t row_pitch, frame_stride;
	const uint8_t* exclude; int64_t exclude_frame_stride;     // optional manual-exclude image(s), [R][C] per frame (stride 0 = shared)
	const float* subtract; int64_t subtract_frame_stride;     // optional image(s) [R][C] taken off the values AFTER the masking (the radial component)
	bool radial_on; RadialSpline radial;                      // ... or the radial component evaluated here from its ring profile (max_knots <= kMaxFusedKnots)
	float flux_cutoff; int box; int nx, ny;
	double* mesh; int32_t* nmasked;
};

__global__ __launch_bounds__(kMeshThreads) void tp_bkg_mesh_kernel(MeshArgs a)
{
	__shared__ __align__(16) float keys[kMeshKeys];
	__shared__ double red[8];
	__shared__ double sp[kRadialStageDoubles(kMaxFusedKnots)];
	const int bx = blockIdx.x, by = blockIdx.y, frame = blockIdx.z;
	const int tid = threadIdx.x;
	const int box = a.box, npix = box * box;
	int rn = 0;
	double rzp = 0.0;
	if (a.radial_on) {
		rn = a.radial.n_knots[frame];
		rzp = a.radial.zeropoint[frame];
		stage_radial(sp, kMaxFusedKnots, a.radial, frame, rn, tid, kMeshThreads);
	}
	const float inf = __builtin_inff();
	const float* img = a.frames + (int64_t)frame * a.frame_stride;
	const uint8_t* excl = a.exclude ? (a.exclude + (int64_t)frame * a.exclude_frame_stride) : nullptr;
	const float* minus = a.subtract ? (a.subtract + (int64_t)frame * a.subtract_frame_stride) : nullptr;
	// ---- load + mask (backgrounds.py:89-97); pixels beyond the frame (padding of the last boxes) are masked
	int nvalid = 0;
	for (int e = tid; e < kMeshKeys; e += kMeshThreads) {
		float x = inf;
		if (e < npix) {
			const int r = by * box + e / box, c = bx * box + e % box;
			if (r < a.n_rows && c < a.n_cols) {
				x = img[(int64_t)r * a.row_pitch + c];
				bool ok = (x >= 0.f) && (x <= a.flux_cutoff);
				if (excl && excl[(int64_t)r * a.n_cols + c]) ok = false;
				nvalid += ok ? 1 : 0;
				if (minus) x = (float)((double)x - (double)minus[(int64_t)r * a.n_cols + c]);   // backgrounds.py:200 (img0 - img_bkg_radial)
				else if (a.radial_on)   // the same float32 image, evaluated here: (float)(10**s(r) - zeropoint), see tp_radial_evaluate
					x = (float)((double)x - (double)(float)radial_value(sp, kMaxFusedKnots, rn, rzp, a.radial.col_offset, a.radial.xcen, a.radial.ycen, r, c));
				x = ok ? x : inf;
			}
		}
		keys[e] = x;
	}
	const int n = (int)mesh_block_sum((double)nvalid, red);
	__syncthreads();
	// ---- bitonic sort of the 4096 keys, ascending; thread t owns keys[16 t .. 16 t + 15] in registers.  A compare-exchange stage
	// of stride s pairs index i with i ^ s: s < 16 is another register of the thread, 16 <= s < 1024 the same register of lane
	// ^ (s / 16) of the wavefront (DPP for lane ^ 1, ^ 2; ds_swizzle for ^ 4, ^ 8, ^ 16; ds_bpermute for ^ 32: no memory, no barrier),
	// and only s = 1024, 2048 (another wavefront: 3 of the 78 stages) go through LDS.  Round 4 ran the 36 stages with s >= 16 through
	// LDS with a barrier each (~135 instructions per stage and thread); a cross-lane stage is 16 x (move + v_med3).
	float v[kKeysPerThread];
	{
		const float4* k4 = reinterpret_cast<const float4*>(keys + tid * kKeysPerThread);
#pragma unroll
		for (int q = 0; q < 4; ++q) { const float4 t = k4[q]; v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w; }
		// sizes 2, 4, 8 inside the thread: direction from the local index
#pragma unroll
		for (int size = 2; size <= 8; size <<= 1) {
#pragma unroll
			for (int stride = size >> 1; stride > 0; stride >>= 1) {
#pragma unroll
				for (int t = 0; t < kKeysPerThread / 2; ++t) {
					const int lo = (t / stride) * (stride * 2) + (t % stride), hi = lo + stride;
					const bool up = ((lo & size) == 0);
					const float x = v[lo], y = v[hi];
					const float mn = fminf(x, y), mx = fmaxf(x, y);
					v[lo] = up ? mn : mx;
					v[hi] = up ? mx : mn;
				}
			}
		}
	}
	__syncthreads();   // (every thread has its keys in registers: the LDS array is free for the cross-wavefront stages)
	const int lane = tid & 63;
	const int partner32 = (lane ^ 32) << 2;
#pragma unroll 1
	for (int size = 16; size <= kMeshKeys; size <<= 1) {
		// ascending where bit `size` of the global index is clear; the last merge is ascending everywhere
		const bool up = (size == kMeshKeys) || (((tid * kKeysPerThread) & size) == 0);
		// ---- strides 2048, 1024: the partner is the same lane of another wavefront -- through LDS
		for (int stride = size >> 1; stride >= 1024; stride >>= 1) {
			float4* o4 = reinterpret_cast<float4*>(keys + tid * kKeysPerThread);
#pragma unroll
			for (int q = 0; q < 4; ++q) o4[q] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
			__syncthreads();
			const int ptid = tid ^ (stride / kKeysPerThread);
			const float sel = ((((tid * kKeysPerThread) & stride) == 0) == up) ? -inf : inf;   // the lower index keeps the minimum when ascending
			const float4* p4 = reinterpret_cast<const float4*>(keys + ptid * kKeysPerThread);
#pragma unroll
			for (int q = 0; q < 4; ++q) {
				const float4 t = p4[q];
				v[4 * q] = __builtin_amdgcn_fmed3f(v[4 * q], t.x, sel); v[4 * q + 1] = __builtin_amdgcn_fmed3f(v[4 * q + 1], t.y, sel);
				v[4 * q + 2] = __builtin_amdgcn_fmed3f(v[4 * q + 2], t.z, sel); v[4 * q + 3] = __builtin_amdgcn_fmed3f(v[4 * q + 3], t.w, sel);
			}
			__syncthreads();
		}
		// ---- strides 512 .. 16: the same register of lane ^ 32 .. lane ^ 1
		auto cross = [&](auto fetch, int lane_bit) {
			const float sel = (((lane & lane_bit) == 0) == up) ? -inf : inf;
#pragma unroll
			for (int j = 0; j < kKeysPerThread; ++j) v[j] = __builtin_amdgcn_fmed3f(v[j], fetch(v[j]), sel);
		};
		if (size >= 1024) cross([&](float x) { return __int_as_float(__builtin_amdgcn_ds_bpermute(partner32, __float_as_int(x))); }, 32);
		if (size >= 512) cross([](float x) { return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(x), 0x401F)); }, 16);
		if (size >= 256) cross([](float x) { return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(x), 0x201F)); }, 8);
		if (size >= 128) cross([](float x) { return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(x), 0x101F)); }, 4);
		if (size >= 64) cross([](float x) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0x4E, 0xF, 0xF, true)); }, 2);
		if (size >= 32) cross([](float x) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0xB1, 0xF, 0xF, true)); }, 1);
		// ---- strides 8 .. 1 inside the thread
		local_tail<8>(v, up);
	}
	{
		float4* o4 = reinterpret_cast<float4*>(keys + tid * kKeysPerThread);
#pragma unroll
		for (int q = 0; q < 4; ++q) o4[q] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
	}
	__syncthreads();
	// ---- SigmaClip(3, maxiters = 5) on the rank range [lo, hi), then the SExtractor estimate.  The keys are sorted, so what a pass
	// clips is a piece at each end of the range: found by two binary searches, and the sums of the kept set follow by taking the
	// clipped keys' contributions off running sums (round 4 made three strided passes over the whole range and four workgroup
	// reductions per iteration).  The sums run over x - c, c = the first median: the one-pass variance then has nothing to cancel.
	double result = __builtin_nan("");
	int nkept = 0;
	if (n > 0) {
		int lo = 0, hi = n;
		const double c0 = ((double)keys[(n - 1) >> 1] + (double)keys[n >> 1]) / 2.0;
		double s1 = 0.0, s2 = 0.0;
		for (int i = tid; i < n; i += kMeshThreads) { const double d = (double)keys[i] - c0; s1 += d; s2 += d * d; }
		mesh_block_sum2(s1, s2, red);
		double med = 0.0, mean = 0.0, sd = 0.0;
		for (int it = 0; it <= 5; ++it) {
			const int m = hi - lo;
			const int m1 = lo + (m >> 1), m0 = (m & 1) ? m1 : (m1 - 1);
			med = ((double)keys[m0] + (double)keys[m1]) / 2.0;
			const double mc = s1 / (double)m;
			mean = c0 + mc;
			double var = s2 / (double)m - mc * mc;     // np.std: population
			if (!(var > 0.0)) var = 0.0;
			sd = sqrt(var);
			if (it == 5) break;
			const double tlo = med - 3.0 * sd, thi = med + 3.0 * sd;
			// first key >= tlo, first key > thi (every thread the same search: broadcast reads)
			int a0 = lo, a1 = hi;
			while (a0 < a1) { const int mid = (a0 + a1) >> 1; if ((double)keys[mid] < tlo) a0 = mid + 1; else a1 = mid; }
			const int nb = a0 - lo;
			int b0 = lo, b1 = hi;
			while (b0 < b1) { const int mid = (b0 + b1) >> 1; if ((double)keys[mid] > thi) b1 = mid; else b0 = mid + 1; }
			const int na = hi - b0;
			if (nb == 0 && na == 0) break;
			double r1 = 0.0, r2 = 0.0;
			for (int i = tid; i < nb + na; i += kMeshThreads) {
				const double d = (double)keys[(i < nb) ? (lo + i) : (hi - na + (i - nb))] - c0;
				r1 += d; r2 += d * d;
			}
			mesh_block_sum2(r1, r2, red);
			s1 -= r1; s2 -= r2;
			lo += nb; hi -= na;
			if (hi <= lo) break;       // (cannot happen: the median is never clipped)
		}
		if (sd == 0.0) result = mean;
		else if (fabs(mean - med) / sd < 0.3) result = 2.5 * med - 1.5 * mean;
		else result = med;
		nkept = hi - lo;
	}
	if (tid == 0) {
		const int64_t o = ((int64_t)frame * a.ny + by) * a.nx + bx;
		a.mesh[o] = result;
		// photutils selects the cells a SECOND time on the sigma-clipped data (Background2D._calc_bkg_bkgrms; mesh_nmasked):
		// masked, padded and clipped pixels all count against exclude_percentile
		a.nmasked[o] = npix - nkept;
	}
}

// cubic B-spline zoom of the prefiltered mesh coefficients (scipy.ndimage.zoom order 3, mode 'reflect', grid_mode = True) with
// the clipping of photutils' BkgZoomInterpolator; one thread per output pixel.  The 16 coefficients a pixel needs are read
// straight from memory: the 64 pixels of a mesh cell along a row share them (one broadcast line), and a row segment of 256 pixels
// touches 4 x 8 of them -- staging the whole mesh in LDS per workgroup (first version) moved 8 KB and a barrier per 256 pixels
// and, with two integer modulo reflections per index, took 0.10 ms per 2048 x 2048 frame.
constexpr int kZoomRows = 16;
__global__ __launch_bounds__(256) void tp_bkg_zoom_kernel(const double* __restrict__ coef, const double* __restrict__ vmin, const double* __restrict__ vmax,
	int ny, int nx, int box, int n_rows, int n_cols, int64_t out_row_pitch, int64_t out_frame_stride, float* __restrict__ out)
{
	const int frame = blockIdx.z;
	const int col = blockIdx.x * blockDim.x + threadIdx.x;
	if (col >= n_cols) return;
	// a thread walks down its column of a strip of kZoomRows rows: column weights once, row sums once per mesh cell
	const ZoomImage z{coef, vmin, vmax, ny, nx, box, n_cols};
	const int r0 = blockIdx.y * kZoomRows, r1 = (r0 + kZoomRows < n_rows) ? (r0 + kZoomRows) : n_rows;
	ZoomAxis ax, ay;
	zoom_axis(col, box, nx, ax);
	const double lo = vmin[frame], hi = vmax[frame];
	double T[4];
	int have = -0x7fffffff;
	for (int row = r0; row < r1; ++row) {
		zoom_axis(row, box, ny, ay);
		if (ay.start != have) { zoom_sums(z, frame, ay, ax, T); have = ay.start; }
		out[(int64_t)frame * out_frame_stride + (int64_t)row * out_row_pitch + col] = zoom_from_sums(ay, T, lo, hi);
	}
}

// B2 on images: out[k][p] = nanmean(in[k-w .. k+w][p]), sequential float32 accumulation like bottleneck.nanmean
__global__ __launch_bounds__(256) void tp_frames_smooth_kernel(const float* __restrict__ in, float* __restrict__ out, int n_frames, int64_t n_pix,
	int64_t frame_stride, int w)
{
	const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const int k = blockIdx.y;
	if (p >= n_pix) return;
	const int i1 = (k - w > 0) ? (k - w) : 0, i2 = (k + w + 1 < n_frames) ? (k + w + 1) : n_frames;
	float asum = 0.f;
	int cnt = 0;
	for (int i = i1; i < i2; ++i) { const float v = in[(int64_t)i * frame_stride + p]; if (v == v) { asum += v; cnt++; } }
	out[(int64_t)k * frame_stride + p] = (cnt > 0) ? (asum / (float)cnt) : __builtin_nanf("");
}

// B3 on images (prepare.py:419-425)
__global__ __launch_bounds__(256) void tp_frames_subtract_kernel(const float* __restrict__ raw, const float* __restrict__ raw_err, const float* __restrict__ bkg,
	const uint8_t* __restrict__ flags, uint32_t flag_mask, float* __restrict__ img, float* __restrict__ err, int64_t n)
{
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const bool excl = flags && (flags[i] & flag_mask) != 0;
	const float r = raw[i] - bkg[i];
	img[i] = excl ? __builtin_nanf("") : r;
	if (raw_err && err) err[i] = excl ? __builtin_nanf("") : raw_err[i];
}

// A1 on images (prepare.py:450-453, 459): SumImage = sum over good frames of the finite pixels / their count (0 / 0 = NaN)
__global__ __launch_bounds__(256) void tp_frames_sumimage_kernel(const float* __restrict__ images, const int32_t* __restrict__ quality, uint32_t bitmask,
	int n_frames, int64_t n_pix, int64_t frame_stride, double* __restrict__ out)
{
	const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (p >= n_pix) return;
	double s = 0.0;
	int cnt = 0;
	for (int k = 0; k < n_frames; ++k) {
		if ((quality[k] & bitmask) != 0) continue;
		const float v = images[(int64_t)k * frame_stride + p];
		if (fabsf(v) <= 3.402823466e+38f) { s += (double)v; cnt++; }
	}
	out[p] = (cnt > 0) ? s / (double)cnt : __builtin_nan("");
}

// pixel_flags.py:54-56: is the whole frame zero?  flags[k] starts at 1 and any block that sees a non-zero (or NaN) pixel clears it
__global__ __launch_bounds__(256) void tp_frames_all_zero_kernel(const float* __restrict__ frames, int64_t n_pix, int64_t frame_stride, int32_t* __restrict__ all_zero)
{
	const int frame = blockIdx.y;
	const float* img = frames + (int64_t)frame * frame_stride;
	bool nonzero = false;
	for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < n_pix; p += (int64_t)gridDim.x * 256) nonzero |= !(img[p] == 0.f);
	if (__any(nonzero) && (threadIdx.x & 63) == 0) all_zero[frame] = 0;
}

// prepare.py:296-297, 406-408: NotUsedForBackground where fit_background masked the pixel (backgrounds.py:89-97: not finite,
// above the cut-off, negative, manually excluded), ManualExclude where pixel_manual_exclude did: columns >= first_excluded[k],
// or the whole frame when it is zero everywhere and zero_is_excluded (TESS data)
__global__ __launch_bounds__(256) void tp_frames_pixel_flags_kernel(const float* __restrict__ frames, int n_rows, int n_cols, int64_t row_pitch,
	int64_t frame_stride, const int32_t* __restrict__ first_excluded, const int32_t* __restrict__ all_zero, int zero_is_excluded,
	float flux_cutoff, uint8_t bit_background, uint8_t bit_manual, uint8_t* __restrict__ flags)
{
	const int frame = blockIdx.z, row = blockIdx.y, col = blockIdx.x * 256 + threadIdx.x;
	if (col >= n_cols) return;
	const float x = frames[(int64_t)frame * frame_stride + (int64_t)row * row_pitch + col];
	int first = first_excluded ? first_excluded[frame] : n_cols;
	if (zero_is_excluded && all_zero && all_zero[frame]) first = 0;
	const bool manual = col >= first;
	const bool masked = !((x >= 0.f) && (x <= flux_cutoff)) || manual;
	flags[((int64_t)frame * n_rows + row) * n_cols + col] = (uint8_t)((masked ? bit_background : 0) | (manual ? bit_manual : 0));
}

// prepare.py:435, 464-466: used[p] = (number of frames in which bit_background is clear) / n_frames > threshold
__global__ __launch_bounds__(256) void tp_frames_used_kernel(const uint8_t* __restrict__ flags, int n_frames, int64_t n_pix, uint8_t bit_background,
	double threshold, uint8_t* __restrict__ used)
{
	const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
	if (p >= n_pix) return;
	int cnt = 0;
	for (int k = 0; k < n_frames; ++k) cnt += ((flags[(int64_t)k * n_pix + p] & bit_background) == 0) ? 1 : 0;
	used[p] = ((double)cnt / (double)n_frames > threshold) ? 1 : 0;
}


//--------------------------------------------------------------------------------------------------
// The low-resolution part of photutils Background2D (1.3.0) after the per-cell statistics, and the cubic-spline prefilter of
// the zoom -- for a few KB per frame, one workgroup per frame with the mesh in LDS (the host did this with numpy / scipy):
//   1. cells with more than exclude_percentile % masked pixels (or a non-finite statistic) are replaced by the inverse-distance
//      weighted mean of the 10 nearest kept cells (ties in distance: row-major order of the kept cells, like the stable argsort);
//   2. 3 x 3 median filter ignoring NaN, windows running off the mesh padded with NaN (generic_filter(nanmedian, mode='constant'));
//   3. min / max of the filtered mesh (the zoom clips to them) and the B-spline coefficients scipy.ndimage.zoom(order = 3,
//      mode = 'reflect') interpolates from: spline_filter1d along both axes (pole sqrt(3) - 2, scipy's ni_splines.c
//      initialisation for the half-sample symmetric boundary).
// A frame without any kept cell comes back NaN.
//--------------------------------------------------------------------------------------------------
constexpr int kMaxMeshCells = 8192;   // the kernel's work arrays live in LDS: 18 bytes per cell, 144 KB of the CU's 160 at the limit
                                      // (a 2048 x 2048 frame in 64-pixel boxes has 1024 cells; 4096 x 4096, or boxes of 32 pixels, 4096)

__device__ inline void prefilter_reflect(double* c, int n, int stride) {
	if (n < 2) return;
	const double z = -0.26794919243112270647;   // sqrt(3) - 2
	const double gain = (1.0 - z) * (1.0 - 1.0 / z);
	for (int i = 0; i < n; ++i) c[i * stride] *= gain;
	double z_i = z;
	double z_n = 1.0;   // z^n (scipy: pow(z, n)); by multiplication: a handful of factors, the same to an ulp or two
	for (int i = 0; i < n; ++i) z_n *= z;
	const double c0 = c[0];
	double acc = c[0] + z_n * c[(n - 1) * stride];
	// scipy accumulates IN PLACE in c[0]: the last term (i = n - 1) reads the running sum, not the original first sample
	for (int i = 1; i < n; ++i) { acc += z_i * (c[i * stride] + z_n * ((i == n - 1) ? acc : c[(n - 1 - i) * stride])); z_i *= z; }
	acc *= z / (1.0 - z_n * z_n);
	c[0] = acc + c0;
	for (int i = 1; i < n; ++i) c[i * stride] += z * c[(i - 1) * stride];
	c[(n - 1) * stride] *= z / (z - 1.0);
	for (int i = n - 2; i >= 0; --i) c[i * stride] = z * (c[(i + 1) * stride] - c[i * stride]);
}

__global__ __launch_bounds__(256) void tp_mesh_finish_kernel(const double* __restrict__ mesh, const int32_t* __restrict__ nmasked, int ny, int nx,
	double max_masked, int filter_size, double* __restrict__ coef, double* __restrict__ vmin, double* __restrict__ vmax, double* __restrict__ filtered)
{
	extern __shared__ __align__(16) double s_mesh_work[];          // [2][ncp] doubles + [ncp] unsigned shorts, ncp = cells rounded up to 4
	const int ncp = (ny * nx + 3) & ~3;
	double* a = s_mesh_work;
	double* b = a + ncp;
	unsigned short* kept = reinterpret_cast<unsigned short*>(b + ncp);
	__shared__ int n_kept;
	__shared__ double red[8];
	const int frame = blockIdx.x, tid = threadIdx.x, nc = ny * nx;
	const double* m = mesh + (int64_t)frame * nc;
	const int32_t* nm = nmasked + (int64_t)frame * nc;
	for (int i = tid; i < nc; i += 256) a[i] = m[i];
	{
		// the kept cells in row-major order (np.nonzero): an ordered compaction, 256 cells per step (ballot / popcount inside a
		// wavefront, four counts across them) -- one thread walking the cells was most of this kernel's time
		__shared__ int wcount[4];
		const int lane = tid & 63, wave = tid >> 6;
		int base = 0;
		for (int i0 = 0; i0 < nc; i0 += 256) {
			const int i = i0 + tid;
			bool keep = false;
			if (i < nc) { const double v = m[i]; keep = ((double)nm[i] <= max_masked) && (v - v == 0.0); }
			const unsigned long long bal = __ballot(keep);
			if (lane == 0) wcount[wave] = __popcll(bal);
			__syncthreads();
			int before = 0, total = 0;
			for (int w = 0; w < 4; ++w) { if (w < wave) before += wcount[w]; total += wcount[w]; }
			if (keep) kept[base + before + __popcll(bal & ((1ull << lane) - 1ull))] = (unsigned short)i;
			base += total;
			__syncthreads();
		}
		if (tid == 0) n_kept = base;
	}
	__syncthreads();
	const int nk = n_kept;
	double* outc = coef + (int64_t)frame * nc;
	if (nk == 0) {
		for (int i = tid; i < nc; i += 256) { outc[i] = __builtin_nan(""); if (filtered) filtered[(int64_t)frame * nc + i] = __builtin_nan(""); }
		if (tid == 0) { vmin[frame] = __builtin_nan(""); vmax[frame] = __builtin_nan(""); }
		return;
	}
	// ---- 1. inverse-distance fill of the excluded cells from the 10 nearest kept ones
	for (int i = tid; i < nc; i += 256) {
		const double v = a[i];
		double r = v;
		if (!((double)nm[i] <= max_masked && v - v == 0.0)) {
			const int y = i / nx, x = i - y * nx;
			double dsel[10], vsel[10];
			int nsel = 0;
			double last_d = -1.0; int last_k = -1;
			for (int s = 0; s < 10 && s < nk; ++s) {
				// the next (distance, position in the kept list) in lexicographic order
				double bd = __builtin_inf(); int bk = -1;
				for (int k = 0; k < nk; ++k) {
					const int c = kept[k];
					const int cy = c / nx, cx = c - cy * nx;
					const double d = hypot((double)(cy - y), (double)(cx - x));
					if ((d > last_d || (d == last_d && k > last_k)) && d < bd) { bd = d; bk = k; }
				}
				if (bk < 0) break;
				dsel[nsel] = bd; vsel[nsel] = a[kept[bk]]; ++nsel;
				last_d = bd; last_k = bk;
			}
			// np.sum over 10 (or fewer) terms: numpy's pairwise reduction -- eight running sums, then the tail
			double w[10], wv[10];
			for (int s = 0; s < nsel; ++s) { w[s] = 1.0 / dsel[s]; wv[s] = w[s] * vsel[s]; }
			auto npsum = [&](const double* q, int n) {
				if (n < 8) { double t = 0.0; for (int s = 0; s < n; ++s) t += q[s]; return t; }
				double t = ((q[0] + q[1]) + (q[2] + q[3])) + ((q[4] + q[5]) + (q[6] + q[7]));
				for (int s = 8; s < n; ++s) t += q[s];
				return t;
			};
			r = npsum(wv, nsel) / npsum(w, nsel);
		}
		b[i] = r;
	}
	__syncthreads();
	// ---- 2. nan-median filter
	if (filter_size > 1) {
		const int h = filter_size / 2;
		for (int i = tid; i < nc; i += 256) {
			const int y = i / nx, x = i - y * nx;
			double w[25];
			int n = 0;
			for (int dy = -h; dy < filter_size - h; ++dy)
				for (int dx = -h; dx < filter_size - h; ++dx) {
					const int yy = y + dy, xx = x + dx;
					if (yy < 0 || yy >= ny || xx < 0 || xx >= nx) continue;
					const double v = b[yy * nx + xx];
					if (v == v && n < 25) {
						int j = n++;
						while (j > 0 && w[j - 1] > v) { w[j] = w[j - 1]; --j; }
						w[j] = v;
					}
				}
			a[i] = (n == 0) ? __builtin_nan("") : ((n & 1) ? w[n >> 1] : 0.5 * (w[(n >> 1) - 1] + w[n >> 1]));
		}
	} else {
		for (int i = tid; i < nc; i += 256) a[i] = b[i];
	}
	__syncthreads();
	// ---- 3. range of the mesh (np.min / np.max propagate NaN) and the spline coefficients
	double lo = __builtin_inf(), hi = -__builtin_inf();
	bool anynan = false;
	for (int i = tid; i < nc; i += 256) { const double v = a[i]; anynan = anynan || (v != v); lo = fmin(lo, v); hi = fmax(hi, v); }
	for (int off = 32; off > 0; off >>= 1) { lo = fmin(lo, __shfl_xor(lo, off, 64)); hi = fmax(hi, __shfl_xor(hi, off, 64)); }
	const bool wavenan = __any(anynan) != 0;
	if ((tid & 63) == 0) { red[tid >> 6] = wavenan ? __builtin_nan("") : lo; red[4 + (tid >> 6)] = wavenan ? __builtin_nan("") : hi; }
	if (filtered) for (int i = tid; i < nc; i += 256) filtered[(int64_t)frame * nc + i] = a[i];
	__syncthreads();
	if (tid == 0) {
		double l = red[0], hmax = red[4];
		for (int w = 1; w < 4; ++w) { l = (red[w] != red[w] || l != l) ? __builtin_nan("") : fmin(l, red[w]); hmax = (red[4 + w] != red[4 + w] || hmax != hmax) ? __builtin_nan("") : fmax(hmax, red[4 + w]); }
		vmin[frame] = l; vmax[frame] = hmax;
	}
	// spline_filter1d(axis = 1 of (T, ny, nx)) = along the rows' index y, for every column; then along x for every row.  An axis
	// of length one is left alone (scipy skips it; the zoom is constant along it): a single row or column of cells is the 1-D case.
	for (int x = tid; x < nx; x += 256) prefilter_reflect(a + x, ny, nx);
	__syncthreads();
	for (int y = tid; y < ny; y += 256) prefilter_reflect(a + y * nx, nx, 1);
	__syncthreads();
	for (int i = tid; i < nc; i += 256) outc[i] = a[i];
}

} // namespace

extern "C" int tp_frames_pixel_flags(tp_ctx* ctx, const float* d_frames, int32_t n_frames, int32_t frame_rows, int32_t frame_cols,
	int64_t row_pitch, int64_t frame_stride, const int32_t* d_first_excluded_column, int32_t zero_is_excluded, double flux_cutoff,
	uint32_t bit_background, uint32_t bit_manual, int32_t* d_all_zero, uint8_t* d_pixel_flags)
{
	TP_CHECK_CTX(ctx);
	TP_API_BEGIN
	TP_REQUIRE(ctx, d_frames && d_all_zero && d_pixel_flags, "tp_frames_pixel_flags: null pointer");
	TP_REQUIRE(ctx, n_frames >= 0 && n_frames <= 65535 && frame_rows > 0 && frame_rows <= 65535 && frame_cols > 0 && row_pitch >= frame_cols
		&& frame_stride >= (int64_t)frame_rows * row_pitch, "tp_frames_pixel_flags: bad frame geometry");
	TP_REQUIRE(ctx, bit_background < 256 && bit_manual < 256, "tp_frames_pixel_flags: the flags are uint8");
	TP_REQUIRE(ctx, row_pitch == frame_cols, "tp_frames_pixel_flags: frames must be contiguous images");
	if (n_frames == 0) return TP_OK;
	std::vector<int32_t> ones((size_t)n_frames, 1);
	TP_HIP(ctx, hipMemcpyAsync(d_all_zero, ones.data(), ones.size() * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
	TP_HIP(ctx, hipStreamSynchronize(ctx->stream));   // the host vector goes out of scope
	const int64_t n_pix = (int64_t)frame_rows * frame_cols;
	TP_LAUNCH(ctx, TPK_BKG_SUBTRACT, tp_frames_all_zero_kernel, dim3(256, (unsigned)n_frames), dim3(256), 0, d_frames, n_pix, frame_stride, d_all_zero);
	TP_LAUNCH(ctx, TPK_BKG_SUBTRACT, tp_frames_pixel_flags_kernel, dim3((unsigned)((frame_cols + 255) / 256), (unsigned)frame_rows, (unsigned)n_frames), dim3(256), 0,
		d_frames, (int)frame_rows, (int)frame_cols, row_pitch, frame_stride, d_first_excluded_column, (const int32_t*)d_all_zero, (int)zero_is_excluded,
		(float)flux_cutoff, (uint8_t)bit_background, (uint8_t)bit_manual, d_pixel_flags);
	TP_LAUNCH_CHECK(ctx, "tp_frames_pixel_flags kernels");
	return TP_OK;
	TP_API_END(ctx)
}

extern "C" int tp_frames_used_in_background(tp_ctx* ctx, const uint8_t* d_pixel_flags, int32_t n_frames, int64_t n_pixels, uint32_t bit_background,
	double threshold, uint8_t* d_used)
{
	TP_CHECK_CTX(ctx);
	TP_API_BEGIN
	TP_REQUIRE(ctx, d_pixel_flags && d_used, "tp_frames_used_in_background: null pointer");
	TP_REQUIRE(ctx, n_frames > 0 && n_pixels > 0 && bit_background < 256, "tp_frames_used_in_background: bad arguments");
	TP_LAUNCH(ctx, TPK_BKG_SUBTRACT, tp_frames_used_kernel, dim3((unsigned)((n_pixels + 255) / 256)), dim3(256), 0, d_pixel_flags, (int)n_frames, n_pixels,
		(uint8_t)bit_background, threshold, d_used);
	TP_LAUNCH_CHECK(ctx, "tp_frames_used_kernel");
	return TP_OK;
	TP_API_END(ctx)
}

static int background_mesh_launch(tp_ctx* ctx, const char* who, const float* d_frames, int32_t n_frames, int32_t frame_rows, int32_t frame_cols,
	int64_t row_pitch, int64_t frame_stride, const uint8_t* d_exclude, int64_t exclude_frame_stride,
	const float* d_subtract, int64_t subtract_frame_stride, const tp_radial_image* radial,
	double flux_cutoff, int32_t box_size, double* d_mesh, int32_t* d_nmasked)
{
	TP_REQUIRE(ctx, d_frames && d_mesh && d_nmasked, "tp_background_mesh: null pointer");
	TP_REQUIRE(ctx, n_frames >= 0 && frame_rows > 0 && frame_cols > 0 && row_pitch >= frame_cols && frame_stride >= (int64_t)frame_rows * row_pitch, "tp_background_mesh: bad frame geometry");
	TP_REQUIRE(ctx, box_size >= 1 && box_size * box_size <= kMeshKeys, "tp_background_mesh: box_size must be 1..64");
	if (n_frames == 0) return TP_OK;
	MeshArgs a;
	a.frames = d_frames; a.n_rows = frame_rows; a.n_cols = frame_cols; a.row_pitch = row_pitch; a.frame_stride = frame_stride;
	a.exclude = d_exclude; a.exclude_frame_stride = exclude_frame_stride;
	a.subtract = d_subtract; a.subtract_frame_stride = subtract_frame_stride;
	a.radial_on = false;
	a.radial = RadialSpline{};
	if (radial) {
		TP_REQUIRE(ctx, d_subtract == nullptr, "tp_background_mesh_radial: a stored image and a ring profile to subtract");
		TP_REQUIRE(ctx, radial->d_knots && radial->d_coefs && radial->d_n_knots && radial->d_zeropoint && radial->max_knots >= 8 && radial->max_knots <= kMaxFusedKnots,
			"tp_background_mesh_radial: bad ring profile (max_knots 8..72)");
		a.radial_on = true;
		a.radial = RadialSpline{radial->col_offset, radial->xcen, radial->ycen, radial->d_knots, radial->d_coefs, radial->d_n_knots, radial->max_knots, radial->d_zeropoint};
	}
	a.flux_cutoff = (float)flux_cutoff; a.box = box_size;
	a.nx = (frame_cols + box_size - 1) / box_size; a.ny = (frame_rows + box_size - 1) / box_size;
	a.mesh = d_mesh; a.nmasked = d_nmasked;
	TP_REQUIRE(ctx, a.ny <= 65535 && n_frames <= 65535, "tp_background_mesh: too many boxes / frames for one launch");
	TP_LAUNCH(ctx, TPK_BKG_MESH, tp_bkg_mesh_kernel, dim3((unsigned)a.nx, (unsigned)a.ny, (unsigned)n_frames), dim3(kMeshThreads), 0, a);
	TP_LAUNCH_CHECK(ctx, who);
	return TP_OK;
}

extern "C" int tp_background_mesh(tp_ctx* ctx, const float* d_frames, int32_t n_frames, int32_t frame_rows, int32_t frame_cols,
	int64_t row_pitch, int64_t frame_stride, const uint8_t* d_exclude, int64_t exclude_frame_stride,
	const float* d_subtract, int64_t subtract_frame_stride,
	double flux_cutoff, int32_t box_size, double* d_mesh, int32_t* d_nmasked)
{
	TP_CHECK_CTX(ctx);
	TP_API_BEGIN
	return background_mesh_launch(ctx, "tp_bkg_mesh_kernel", d_frames, n_frames, frame_rows, frame_cols, row_pitch, frame_stride, d_exclude, exclude_frame_stride,
		d_subtract, subtract_frame_stride, nullptr, flux_cutoff, box_size, d_mesh, d_nmasked);
	TP_API_END(ctx)
}

extern "C" int tp_background_mesh_radial(tp_ctx* ctx, const float* d_frames, int32_t n_frames, int32_t frame_rows, int32_t frame_cols,
	int64_t row_pitch, int64_t frame_stride, const uint8_t* d_exclude, int64_t exclude_frame_stride,
	const tp_radial_image* radial, double flux_cutoff, int32_t box_size, double* d_mesh, int32_t* d_nmasked)
{
	TP_CHECK_CTX(ctx);
	TP_API_BEGIN
	TP_REQUIRE(ctx, radial != nullptr, "tp_background_mesh_radial: null ring profile");
	return background_mesh_launch(ctx, "tp_bkg_mesh_kernel (radial)", d_frames, n_frames, frame_rows, frame_cols, row_pitch, frame_stride, d_exclude, exclude_frame_stride,
		nullptr, 0, radial, flux_cutoff, box_size, d_mesh, d_nmasked);
	TP_API_END(ctx)
}

extern "C" int tp_background_mesh_finish(tp_ctx* ctx, const double* d_mesh, const int32_t* d_nmasked, int32_t n_frames, int32_t mesh_rows,
	int32_t mesh_cols, int32_t box_size, double exclude_percentile, int32_t filter_size, double* d_coef, double* d_vmin, double* d_vmax,
	double* d_filtered)
{
	TP_CHECK_CTX(ctx);
	TP_API_BEGIN
	TP_REQUIRE(ctx, d_mesh && d_nmasked && d_coef && d_vmin && d_vmax, "tp_background_mesh_finish: null pointer");
	TP_REQUIRE(ctx, n_frames >= 0 && mesh_rows > 0 && mesh_cols > 0 && mesh_rows * mesh_cols <= kMaxMeshCells, "tp_background_mesh_finish: at most 8192 cells per frame");
	TP_REQUIRE(ctx, box_size > 0 && filter_size >= 1 && filter_size <= 5 && (filter_size & 1), "tp_background_mesh_finish: filter_size must be 1, 3 or 5");
	if (n_frames == 0) return TP_OK;
	const size_t ncp = ((size_t)mesh_rows * mesh_cols + 3) & ~(size_t)3;
	const size_t shmem = ncp * (2 * sizeof(double) + sizeof(unsigned short));
	if (shmem > 48 * 1024)
		TP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(tp_mesh_finish_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
	TP_LAUNCH(ctx, TPK_BKG_MESH, tp_mesh_finish_kernel, dim3((unsigned)n_frames), dim3(256), shmem, d_mesh, d_nmasked, (int)mesh_rows, (int)mesh_cols,
		exclude_percentile / 100.0 * (double)box_size * (double)box_size, (int)filter_size, d_coef, d_vmin, d_vmax, d_filtered);
	TP_LAUNCH_CHECK(ctx, "tp_mesh_finish_kernel");
	return TP_OK;
	TP_API_END(ctx)
}

extern "C" int tp_background_zoom(tp_ctx* ctx, const double* d_coef, const double* d_vmin, const double* d_vmax, int32_t n_frames,
	int32_t mesh_rows, int32_t mesh_cols, int32_t box_size, int32_t frame_rows, int32_t frame_cols, int64_t row_pitch, int64_t frame_stride,
	float* d_background)
{
	TP_CHECK_CTX(ctx);
	TP_API_BEGIN
	TP_REQUIRE(ctx, d_coef && d_vmin && d_vmax && d_background, "tp_background_zoom: null pointer");
	TP_REQUIRE(ctx, mesh_rows > 0 && mesh_cols > 0 && box_size > 0 && frame_rows > 0 && frame_cols > 0 && row_pitch >= frame_cols, "tp_background_zoom: bad geometry");
	TP_REQUIRE(ctx, frame_rows <= 65535 && n_frames <= 65535, "tp_background_zoom: too many rows / frames for one launch");
	if (n_frames == 0) return TP_OK;
	dim3 grid((unsigned)((frame_cols + 255) / 256), (unsigned)((frame_rows + kZoomRows - 1) / kZoomRows), (unsigned)n_frames);
	TP_LAUNCH(ctx, TPK_BKG_ZOOM, tp_bkg_zoom_kernel, grid, dim3(256), 0, d_coef, d_vmin, d_vmax,
		(int)mesh_rows, (int)mesh_cols, (int)box_size, (int)frame_rows, (int)frame_cols, row_pitch, frame_stride, d_background);
	TP_LAUNCH_CHECK(ctx, "tp_bkg_zoom_kernel");
	return TP_OK;
	TP_API_END(ctx)
}

extern "C" int tp_frames_smooth_time(tp_ctx* ctx, int32_t n_frames, int64_t n_pixels, int64_t frame_stride, int32_t time_smooth,
	const float* d_in, float* d_out)
{
	TP_CHECK_CTX(ctx);
	TP_API_BEGIN
	TP_REQUIRE(ctx, d_in && d_out && d_in != d_out, "tp_frames_smooth_time: null or aliased pointers");
	TP_REQUIRE(ctx, n_frames >= 0 && n_frames <= 65535 && n_pixels >= 0 && frame_stride >= n_pixels && time_smooth >= 1, "tp_frames_smooth_time: bad geometry");
	if (n_frames == 0 || n_pixels == 0) return TP_OK;
	dim3 grid((unsigned)((n_pixels + 255) / 256), (unsigned)n_frames);
	TP_LAUNCH(ctx, TPK_BKG_SMOOTH, tp_frames_smooth_kernel, grid, dim3(256), 0, d_in, d_out, (int)n_frames, n_pixels, frame_stride, (int)(time_smooth / 2));
	TP_LAUNCH_CHECK(ctx, "tp_frames_smooth_kernel");
	return TP_OK;
	TP_API_END(ctx)
}

extern "C" int tp_frames_subtract(tp_ctx* ctx, int64_t n_values, const float* d_raw, const float* d_raw_err, const float* d_bkg,
	const uint8_t* d_pixel_flags, uint32_t flag_mask, float* d_images, float* d_images_err)
{
	TP_CHECK_CTX(ctx);
	TP_API_BEGIN
	TP_REQUIRE(ctx, d_raw && d_bkg && d_images, "tp_frames_subtract: null pointer");
	if (n_values <= 0) return TP_OK;
	TP_REQUIRE(ctx, (n_values + 255) / 256 <= 2147483647ll, "tp_frames_subtract: too many values for one launch");
	TP_LAUNCH(ctx, TPK_BKG_SUBTRACT, tp_frames_subtract_kernel, dim3((unsigned)((n_values + 255) / 256)), dim3(256), 0, d_raw, d_raw_err, d_bkg,
		d_pixel_flags, flag_mask, d_images, d_images_err, n_values);
	TP_LAUNCH_CHECK(ctx, "tp_frames_subtract_kernel");
	return TP_OK;
	TP_API_END(ctx)
}

extern "C" int tp_frames_sumimage(tp_ctx* ctx, int32_t n_frames, int64_t n_pixels, int64_t frame_stride, const float* d_images,
	const int32_t* d_quality, uint32_t bitmask, double* d_sumimage)
{
	TP_CHECK_CTX(ctx);
	TP_API_BEGIN
	TP_REQUIRE(ctx, d_images && d_quality && d_sumimage, "tp_frames_sumimage: null pointer");
	TP_REQUIRE(ctx, n_frames >= 0 && n_pixels >= 0 && frame_stride >= n_pixels, "tp_frames_sumimage: bad geometry");
	if (n_pixels == 0) return TP_OK;
	TP_LAUNCH(ctx, TPK_SUMIMAGE, tp_frames_sumimage_kernel, dim3((unsigned)((n_pixels + 255) / 256)), dim3(256), 0, d_images, d_quality, bitmask,
		(int)n_frames, n_pixels, frame_stride, d_sumimage);
	TP_LAUNCH_CHECK(ctx, "tp_frames_sumimage_kernel");
	return TP_OK;
	TP_API_END(ctx)
}
