// radial.hip -- the radial ("corner glow") component of backgrounds.fit_background for TESS full-frame images
// (photometry/backgrounds.py:104-197) on a frame stack resident in HBM.  Per iteration of its bkgiters loop:
//
//   tp_radial_zeropoint   zeropoint = -min(img - square over the unmasked pixels) + 1                       (:166-168)
//   tp_radial_ring_modes  per ring of 15 pixels beyond 2400 pixels from the camera centre: the mode of log10(img - square +
//                         zeropoint) -- argmax of statsmodels' FFT Gaussian KDE on 2048 grid points with the
//                         normal-reference bandwidth (_reduce_mode :20-32, binned_statistic :171-176)
//   tp_radial_profiles    3-point median of the ~40 ring modes and the interpolating cubic spline through them              (:179-187)
//   tp_radial_evaluate    img_bkg_radial = 10**spline(r) - zeropoint for every pixel (ext = 3: constant beyond the end knots), :188
//   The *_zoom entries take the square component (the previous iteration's zoomed mesh) as spline coefficients and evaluate it where
//   it is read (fullframe_dev.h): inside the alternation no frame-sized image is stored.
//
// The geometry (distance of every pixel from the camera centre, ring membership) does not depend on the frame: the host
// lists the pixels of every ring once (row-major inside a ring, like r[~mask] in the reference) and the ring kernel walks
// its list.  One 256-thread workgroup per (ring, frame):
//   1. gather the unmasked pixels of the ring into an HBM scratch row (<= ~25 000 float64) by an ORDERED compaction (the
//      order of the pixel list), with their count, sum, min, max: nothing in the kernel depends on the order of arrival;
//   2. standard deviation (two-pass, ddof = 1), quartiles by radix selection on the order-preserving 64-bit keys
//      (scipy.stats.scoreatpercentile's linear interpolation between the two neighbouring order statistics);
//   3. bandwidth C min(std, IQR / 1.349) n^-1/5; linear binning on 2048 grid points over [min - 3 bw, max + 3 bw], the
//      weights accumulated as 40-bit fixed-point integers (LDS integer atomics: independent of the order of arrival);
//   4. forward FFT (radix 2, LDS, twiddles from sincospi), multiplication by Silverman's transform of the Gaussian, second
//      forward FFT of the conjugate = the inverse transform of a Hermitian spectrum; first index of the maximum -> grid value.
// Bound by LDS/latency (18 passes over an L2-resident row, two 2048-point FFTs); a frame has ~40 rings, so the whole
// component is a few hundred microseconds per frame and iteration next to the 64 x 64 mesh statistics.
#include "common.h"
#include "fullframe_dev.h"
#include <cmath>

namespace {

constexpr int kRadThreads = 256;
constexpr int kKdeGrid = 2048;
constexpr int kKdeLog2 = 11;
constexpr int kRadBatch = 8;    // loads of the ring's scratch row in flight per thread

struct RadialImage {
	const float* frames; int64_t frame_stride;
	const float* square; int64_t square_stride;           // previous iteration's mesh background, or null (first iteration) ...
	bool zoom_on; ZoomImage zoom;                         // ... or the same image evaluated here from the mesh's spline coefficients
	const uint8_t* exclude; int64_t exclude_stride;       // manual-exclude image(s), stride 0 = shared
	float flux_cutoff;
};

// backgrounds.py:89-97 on the RAW image; value = img - square (float64 once a square component exists, float32 before)
// (x, excluded: the pixel's value and manual-exclude flag, loaded by the caller -- the ring kernel keeps several in flight)
__device__ __forceinline__ bool radial_pixel_loaded(const RadialImage& a, int frame, int64_t p, float x, bool excluded, double& value) {
	bool ok = (x >= 0.f) && (x <= a.flux_cutoff) && !excluded;
	if (a.square) value = (double)x - (double)a.square[(int64_t)frame * a.square_stride + p];
	else if (a.zoom_on) { const uint32_t q = (uint32_t)p, row = q / (uint32_t)a.zoom.n_cols;   // (pixel indices fit 31 bits: radial_image_ok)
		value = (double)x - (double)zoom_value(a.zoom, frame, (int)row, (int)(q - row * (uint32_t)a.zoom.n_cols)); }
	else value = (double)x;
	return ok;
}
__device__ __forceinline__ bool radial_pixel(const RadialImage& a, int frame, int64_t p, double& value) {
	const float x = a.frames[(int64_t)frame * a.frame_stride + p];
	const bool excluded = a.exclude && a.exclude[(int64_t)frame * a.exclude_stride + p];
	return radial_pixel_loaded(a, frame, p, x, excluded, value);
}

__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_xor(v, off, 64));
	return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
	return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
	return v;
}

// block-wide reductions through a 4-entry LDS array; every thread gets the result
template <int OP>
__device__ __forceinline__ double block_reduce(double v, double* red) {
	v = (OP == 0) ? wave_sum(v) : (OP == 1) ? wave_min(v) : wave_max(v);
	__syncthreads();
	if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
	__syncthreads();
	if (OP == 0) return (red[0] + red[1]) + (red[2] + red[3]);
	if (OP == 1) return fmin(fmin(red[0], red[1]), fmin(red[2], red[3]));
	return fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
}

__global__ __launch_bounds__(kRadThreads) void tp_radial_min_partial_kernel(RadialImage a, int64_t n_pix, double* __restrict__ partial)
{
	__shared__ double red[4];
	const int frame = blockIdx.y;
	double mn = __builtin_inf();
	for (int64_t p = (int64_t)blockIdx.x * kRadThreads + threadIdx.x; p < n_pix; p += (int64_t)gridDim.x * kRadThreads) {
		double v;
		if (radial_pixel(a, frame, p, v)) mn = fmin(mn, v);
	}
	mn = block_reduce<1>(mn, red);
	if (threadIdx.x == 0) partial[(int64_t)frame * gridDim.x + blockIdx.x] = mn;
}

// The same with the square component evaluated from the mesh's spline coefficients: a thread walks down one column of a band of
// rows, so that the column's weights are formed once and the row sums of the coefficients once per mesh cell (zoom_sums).
// grid (column blocks, row bands, frames); partial [frame][gridDim.x * gridDim.y]
__global__ __launch_bounds__(kRadThreads) void tp_radial_min_zoom_kernel(RadialImage a, int n_rows, int n_cols, int rows_per_band, double* __restrict__ partial)
{
	__shared__ double red[4];
	const int frame = blockIdx.z;
	const int col = blockIdx.x * kRadThreads + threadIdx.x;
	const int r0 = blockIdx.y * rows_per_band, r1 = (r0 + rows_per_band < n_rows) ? (r0 + rows_per_band) : n_rows;
	double mn = __builtin_inf();
	if (col < n_cols) {
		ZoomAxis ax, ay;
		zoom_axis(col, a.zoom.box, a.zoom.nx, ax);
		const double lo = a.zoom.vmin[frame], hi = a.zoom.vmax[frame];
		double T[4];
		int have = -0x7fffffff;
		for (int r = r0; r < r1; ++r) {
			const int64_t p = (int64_t)r * n_cols + col;
			const float x = a.frames[(int64_t)frame * a.frame_stride + p];
			bool ok = (x >= 0.f) && (x <= a.flux_cutoff);
			if (a.exclude && a.exclude[(int64_t)frame * a.exclude_stride + p]) ok = false;
			zoom_axis(r, a.zoom.box, a.zoom.ny, ay);
			if (ay.start != have) { zoom_sums(a.zoom, frame, ay, ax, T); have = ay.start; }
			const double v = (double)x - (double)zoom_from_sums(ay, T, lo, hi);
			if (ok) mn = fmin(mn, v);
		}
	}
	mn = block_reduce<1>(mn, red);
	if (threadIdx.x == 0) partial[((int64_t)frame * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = mn;
}

// zeropoint = -min + 1.0 (float64; with numpy 1.21 the float32 minimum of the first iteration is promoted by the Python
// float); +inf partials (nothing unmasked) -> NaN
__global__ __launch_bounds__(kRadThreads) void tp_radial_min_final_kernel(const double* __restrict__ partial, int n_partial, double* __restrict__ zeropoint)
{
	__shared__ double red[4];
	const int frame = blockIdx.x;
	double mn = __builtin_inf();
	for (int i = threadIdx.x; i < n_partial; i += kRadThreads) mn = fmin(mn, partial[(int64_t)frame * n_partial + i]);
	mn = block_reduce<1>(mn, red);
	if (threadIdx.x == 0) zeropoint[frame] = (mn == __builtin_inf()) ? __builtin_nan("") : (-mn + 1.0);
}

__device__ __forceinline__ uint64_t order_key(double v) {
	const uint64_t b = (uint64_t)__double_as_longlong(v);
	return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double key_value(uint64_t k) {
	const uint64_t b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
	return __longlong_as_double((long long)b);
}

// k-th and (k+1)-th smallest of vals[0..n) (k + 1 clamped to n - 1): MSB-first radix selection, 8 bits per pass
__device__ void select_pair(const double* __restrict__ vals, int n, int k, int* hist, int* sh, double* red, double& v0, double& v1)
{
	const int tid = threadIdx.x;
	uint64_t prefix = 0, pmask = 0;
	int rank = k;
	for (int shift = 56; shift >= 0; shift -= 8) {
		hist[tid] = 0;
		__syncthreads();
		// (eight loads in flight per thread: the row lives in L2, and with one ring per workgroup and two or three workgroups per CU
		// nothing else hides a load's latency -- a load per iteration was most of this kernel's time)
		for (int i0 = tid; i0 < n; i0 += kRadBatch * kRadThreads) {
			double x[kRadBatch];
#pragma unroll
			for (int u = 0; u < kRadBatch; ++u) { const int i = i0 + u * kRadThreads; x[u] = vals[(i < n) ? i : (n - 1)]; }
#pragma unroll
			for (int u = 0; u < kRadBatch; ++u) {
				const uint64_t key = order_key(x[u]);
				if (i0 + u * kRadThreads < n && (key & pmask) == prefix) atomicAdd(&hist[(int)((key >> shift) & 255u)], 1);
			}
		}
		__syncthreads();
		{
			// the digit whose bin holds the rank: the first d with hist[0] + ... + hist[d] > rank -- one bin per thread, an inclusive
			// scan by shuffles inside the wavefronts and over their four totals (a single thread walking the 256 bins through LDS was
			// a seventh of the kernel's time: 16 such walks per ring)
			const int c = hist[tid];
			int x = c;
#pragma unroll
			for (int off = 1; off < 64; off <<= 1) { const int y = __shfl_up(x, off, 64); if ((tid & 63) >= off) x += y; }
			int* wtot = reinterpret_cast<int*>(red);           // four wave totals (the reductions that use `red` come later)
			if ((tid & 63) == 63) wtot[tid >> 6] = x;
			__syncthreads();
			int before = 0;
			for (int w = 0; w < (tid >> 6); ++w) before += wtot[w];
			const int incl = x + before, excl = incl - c;
			if (excl <= rank && rank < incl) { sh[0] = tid; sh[1] = rank - excl; }
		}
		__syncthreads();
		prefix |= (uint64_t)sh[0] << shift;
		pmask |= 255ull << shift;
		rank = sh[1];
		__syncthreads();
	}
	v0 = key_value(prefix);
	// the next order statistic: the same value if enough copies of it exist, else the smallest larger one
	int le = 0;
	double nxt = __builtin_inf();
	for (int i0 = tid; i0 < n; i0 += kRadBatch * kRadThreads) {
		double x[kRadBatch];
#pragma unroll
		for (int u = 0; u < kRadBatch; ++u) { const int i = i0 + u * kRadThreads; x[u] = vals[(i < n) ? i : (n - 1)]; }
#pragma unroll
		for (int u = 0; u < kRadBatch; ++u) {
			if (i0 + u * kRadThreads >= n) continue;
			le += (x[u] <= v0) ? 1 : 0;
			if (x[u] > v0) nxt = fmin(nxt, x[u]);
		}
	}
	const int n_le = (int)block_reduce<0>((double)le, red);
	nxt = block_reduce<1>(nxt, red);
	v1 = (n_le > k + 1 || nxt == __builtin_inf()) ? v0 : nxt;
}

// scipy.stats.scoreatpercentile(x, per) (interpolation_method = 'fraction') from the two neighbouring order statistics
__device__ __forceinline__ double percentile_from_pair(int n, double per, double s0, double s1) {
	const double idx = per / 100.0 * (double)(n - 1);
	const int i = (int)idx;
	if ((double)i == idx) return s0;
	const double w0 = (double)(i + 1) - idx, w1 = idx - (double)i;
	return (s0 * w0 + s1 * w1) / (w0 + w1);
}

struct RingArgs {
	RadialImage img;
	const double* zeropoint;
	const int32_t* ring_pixels; const int32_t* ring_offsets; int n_rings;
	double* scratch; int64_t scratch_stride;     // per frame: one row of ring_offsets[n_rings] float64
	double bw_constant;                          // the kernel's normal-reference constant 2 (1/24)^(1/5)
	double* modes; int32_t* counts;              // [frame][ring]
};

__global__ __launch_bounds__(kRadThreads) void tp_radial_ring_kernel(RingArgs a)
{
	__shared__ double re[kKdeGrid], im[kKdeGrid];
	__shared__ double tw_re[kKdeGrid / 2], tw_im[kKdeGrid / 2];
	__shared__ double red[4];
	__shared__ int hist[256];
	__shared__ int sh[2];
	const int ring = blockIdx.x, frame = blockIdx.y, tid = threadIdx.x;
	const int p0 = a.ring_offsets[ring], p1 = a.ring_offsets[ring + 1];
	double* vals = a.scratch + (int64_t)frame * a.scratch_stride + p0;
	const double zp = a.zeropoint[frame];
	const bool single = (a.img.square == nullptr) && !a.img.zoom_on;   // first iteration: float32 arithmetic (numpy: float32 array + scalar, log10 of float32)
	const float zp32 = (float)zp;
	double* mode_out = a.modes + (int64_t)frame * a.n_rings + ring;

	// ---- 1. gather, in the order of the ring's pixel list (row-major, like values[binnumber == j] in the reference): an
	// ordered compaction -- per block of 256 list entries the kept ones are ranked by ballot / popcount inside a wavefront
	// and by a four-entry count across the wavefronts.  Same order, same sums, same result in every run.
	const int lane = tid & 63, wave = tid >> 6;
	int base = 0;
	double sum = 0.0, mn = __builtin_inf(), mx = -__builtin_inf();
	for (int ib = p0; ib < p1; ib += kRadBatch * kRadThreads) {
	// the pixel indices, then the pixels, of the next kRadBatch steps: two dependent loads per step would otherwise be waited
	// for one step at a time
	int pix[kRadBatch];
	float xin[kRadBatch];
	bool exc[kRadBatch];
#pragma unroll
	for (int u = 0; u < kRadBatch; ++u) { const int i = ib + u * kRadThreads + tid; pix[u] = a.ring_pixels[(i < p1) ? i : (p1 - 1)]; }
#pragma unroll
	for (int u = 0; u < kRadBatch; ++u) {
		xin[u] = a.img.frames[(int64_t)frame * a.img.frame_stride + pix[u]];
		exc[u] = a.img.exclude && a.img.exclude[(int64_t)frame * a.img.exclude_stride + pix[u]];
	}
#pragma unroll
	for (int u = 0; u < kRadBatch; ++u) {
		const int i0 = ib + u * kRadThreads;
		if (i0 >= p1) break;                        // uniform
		const int i = i0 + tid;
		double v = 0.0, lg = 0.0;
		const bool keep = (i < p1) && radial_pixel_loaded(a.img, frame, pix[u], xin[u], exc[u], v);
		if (keep) {
			if (single) lg = (double)(float)log10((double)((float)v + zp32));
			else lg = log10(v + zp);
		}
		const unsigned long long bal = __ballot(keep);
		if (lane == 0) hist[wave] = __popcll(bal);
		__syncthreads();
		int before = 0, total = 0;
#pragma unroll
		for (int w = 0; w < 4; ++w) { const int c = hist[w]; before += (w < wave) ? c : 0; total += c; }
		if (keep) {
			vals[base + before + __popcll(bal & ((1ull << lane) - 1ull))] = lg;
			sum += lg; mn = fmin(mn, lg); mx = fmax(mx, lg);
		}
		base += total;
		__syncthreads();
	}
	}
	const int n = base;
	__threadfence_block();
	if (a.counts && tid == 0) a.counts[(int64_t)frame * a.n_rings + ring] = n;
	if (n < 2 || !(zp == zp)) {
		// an empty ring is NaN (_reduce_mode :21-22); one value gives a NaN bandwidth (std with ddof = 1), a NaN grid and support[0] = NaN
		if (tid == 0) *mode_out = __builtin_nan("");
		return;
	}
	sum = block_reduce<0>(sum, red);
	mn = block_reduce<1>(mn, red);
	mx = block_reduce<2>(mx, red);
	__syncthreads();   // the scratch row is complete and visible

	// ---- 2. spread: std (ddof = 1) and inter-quartile range
	const double mean = sum / (double)n;
	double ss = 0.0;
	for (int i0 = tid; i0 < n; i0 += kRadBatch * kRadThreads) {
		double x[kRadBatch];
#pragma unroll
		for (int u = 0; u < kRadBatch; ++u) { const int i = i0 + u * kRadThreads; x[u] = vals[(i < n) ? i : (n - 1)]; }
#pragma unroll
		for (int u = 0; u < kRadBatch; ++u) if (i0 + u * kRadThreads < n) { const double d = x[u] - mean; ss += d * d; }   // (the order of round 4: i ascending)
	}
	ss = block_reduce<0>(ss, red);
	const double sd = sqrt(ss / (double)(n - 1));
	double a0, a1, b0, b1;
	select_pair(vals, n, (int)(0.25 * (double)(n - 1)), hist, sh, red, a0, a1);
	select_pair(vals, n, (int)(0.75 * (double)(n - 1)), hist, sh, red, b0, b1);
	const double q25 = percentile_from_pair(n, 25.0, a0, a1), q75 = percentile_from_pair(n, 75.0, b0, b1);
	const double iqr = (q75 - q25) / 1.349;
	const double sigma = (iqr > 0.0) ? fmin(sd, iqr) : sd;
	const double bw = a.bw_constant * sigma * pow((double)n, -0.2);
	if (bw == 0.0) {
		// "Selected KDE bandwidth is 0" -> the median (:27-31)
		double m0, m1;
		select_pair(vals, n, (n - 1) / 2, hist, sh, red, m0, m1);
		if (tid == 0) *mode_out = (n & 1) ? m0 : 0.5 * (m0 + m1);
		return;
	}

	// ---- 3. linear binning (statsmodels fast_linbin, with its "li > 1" guard)
	const double lo = mn - 3.0 * bw, hi = mx + 3.0 * bw;
	const double delta = (hi - lo) / (double)(kKdeGrid - 1);
	// The weights (1 - rem, rem) are accumulated as 40-bit fixed-point integers with LDS integer atomics: the sum of integers
	// does not depend on the order of arrival, so the binned grid is the same in every run (float atomics are not); the
	// quantisation (2^-41 per sample) is far below the rounding of the float32 logarithm the samples come from.
	unsigned long long* bins = reinterpret_cast<unsigned long long*>(re);
	for (int i = tid; i < kKdeGrid; i += kRadThreads) { bins[i] = 0ull; im[i] = 0.0; }
	for (int i = tid; i < kKdeGrid / 2; i += kRadThreads) {
		double s, c;
		sincospi(-2.0 * (double)i / (double)kKdeGrid, &s, &c);
		tw_re[i] = c; tw_im[i] = s;
	}
	__syncthreads();
	constexpr double kFix = 1099511627776.0;   // 2^40
	for (int i0 = tid; i0 < n; i0 += kRadBatch * kRadThreads) {
		double xin[kRadBatch];
#pragma unroll
		for (int u = 0; u < kRadBatch; ++u) { const int i = i0 + u * kRadThreads; xin[u] = vals[(i < n) ? i : (n - 1)]; }
#pragma unroll
		for (int u = 0; u < kRadBatch; ++u) {
		if (i0 + u * kRadThreads >= n) continue;
		const double lx = (xin[u] - lo) / delta;
		const int li = (int)lx;
		const double rem = lx - (double)li;
		if (li > 1 && li < kKdeGrid) {
			const unsigned long long q = (unsigned long long)rint(rem * kFix);
			// bit-reversed positions: the FFT below is decimation in time
			atomicAdd(&bins[__brev((unsigned)li) >> (32 - kKdeLog2)], (unsigned long long)kFix - q);
			if (li + 1 < kKdeGrid) atomicAdd(&bins[__brev((unsigned)(li + 1)) >> (32 - kKdeLog2)], q);
		}
		}
	}
	__syncthreads();
	for (int i = tid; i < kKdeGrid; i += kRadThreads) re[i] = (double)bins[i] * (1.0 / kFix);
	__syncthreads();

	// ---- 4. density = IFFT(FFT(binned) * Silverman transform); only its argmax is used (positive scale factors dropped)
	const double range = hi - lo;
	const double fac1 = 2.0 * (M_PI * bw / range) * (M_PI * bw / range);
	for (int pass = 0; pass < 2; ++pass) {
		for (int s = 1; s <= kKdeLog2; ++s) {
			const int half = 1 << (s - 1), tstep = kKdeGrid >> s;
			for (int b = tid; b < kKdeGrid / 2; b += kRadThreads) {
				const int j = b & (half - 1), base = ((b >> (s - 1)) << s) + j;
				const double wr = tw_re[j * tstep], wi = tw_im[j * tstep];
				const double xr = re[base + half], xi = im[base + half];
				const double tr = xr * wr - xi * wi, ti = xr * wi + xi * wr;
				const double ur = re[base], ui = im[base];
				re[base] = ur + tr; im[base] = ui + ti;
				re[base + half] = ur - tr; im[base + half] = ui - ti;
			}
			__syncthreads();
		}
		if (pass == 0) {
			// multiply by the kernel transform, conjugate, and put back in bit-reversed order for the second transform
			double zr[kKdeGrid / kRadThreads], zi[kKdeGrid / kRadThreads];
#pragma unroll
			for (int q = 0; q < kKdeGrid / kRadThreads; ++q) {
				const int J = tid + q * kRadThreads;
				const double jj = (double)((J <= kKdeGrid / 2) ? J : (kKdeGrid - J));
				const double t = jj / (double)kKdeGrid * M_PI;
				const double fac = exp(-(jj * jj) * fac1) / (1.0 - 1.0 / 3.0 * (t * t));
				zr[q] = re[J] * fac; zi[q] = -im[J] * fac;
			}
			__syncthreads();
#pragma unroll
			for (int q = 0; q < kKdeGrid / kRadThreads; ++q) {
				const int J = tid + q * kRadThreads;
				const int r = (int)(__brev((unsigned)J) >> (32 - kKdeLog2));
				re[r] = zr[q]; im[r] = zi[q];
			}
			__syncthreads();
		}
	}
	// first index of the maximum (np.argmax)
	double best = -__builtin_inf();
	int best_i = kKdeGrid;
	for (int i = tid; i < kKdeGrid; i += kRadThreads) {
		const double f = re[i];
		if (f > best) { best = f; best_i = i; }
	}
	const double wmax = block_reduce<2>(best, red);
	const int cand = (best == wmax) ? best_i : kKdeGrid;
	const int arg = (int)block_reduce<1>((double)cand, red);
	if (tid == 0) {
		// np.linspace(lo, hi, 2048): arange * step + lo, last point = hi
		*mode_out = (arg == kKdeGrid - 1) ? hi : ((double)arg * delta + lo);
	}
}


//--------------------------------------------------------------------------------------------------
// The ring profile of a frame: 3-point moving median of the ring modes (utilities.move_median_central, utilities.py:52-62,
// backgrounds.py:178-179) and the interpolating cubic spline through the rings that have a mode
// (InterpolatedUnivariateSpline(k = 3), backgrounds.py:186: FITPACK's interpolating spline puts its interior knots on the data
// points x[2 .. m-3] -- "not a knot" at the second and the second-to-last point -- and solves the m x m collocation system).
// A few dozen points per frame: one lane per frame does it, the collocation matrix (four non-zeros per row) in LDS.
//--------------------------------------------------------------------------------------------------
constexpr int kMaxRings = 64;

__device__ inline double nan_median(const double* v, int n) {
	double w[8];
	int m = 0;
	for (int i = 0; i < n && i < 8; ++i) if (v[i] == v[i]) w[m++] = v[i];
	if (m == 0) return __builtin_nan("");
	for (int i = 1; i < m; ++i) { const double x = w[i]; int j = i; while (j > 0 && w[j - 1] > x) { w[j] = w[j - 1]; --j; } w[j] = x; }
	return (m & 1) ? w[m >> 1] : 0.5 * (w[(m >> 1) - 1] + w[m >> 1]);
}

__global__ __launch_bounds__(64) void tp_radial_profile_kernel(const double* __restrict__ modes, const double* __restrict__ bin_center, int n_rings,
	int width, int n_frames, int max_knots, double* __restrict__ knots, double* __restrict__ coefs, int32_t* __restrict__ n_knots)
{
	// One wavefront per frame.  Round 4 let ONE lane do everything (182 us per launch: a chain of ~3 000 LDS round trips); now the
	// lanes share what is independent -- the moving medians, the collocation rows, and inside every pivot step of the elimination the
	// (row, column) entries of the up to four rows it touches -- with every value formed by the same operations in the same order.
	__shared__ double A[kMaxRings][8];      // banded rows after elimination: columns j0 .. j0 + 6 of the row
	__shared__ double s2l[kMaxRings], prof[kMaxRings], x[kMaxRings], y[kMaxRings], trail[kMaxRings], t[kMaxRings + 8];
	__shared__ int j0[kMaxRings];
	const int frame = blockIdx.x, lane = threadIdx.x;
	if (frame >= n_frames) return;
	double* kt = knots + (int64_t)frame * max_knots;
	double* kc = coefs + (int64_t)frame * max_knots;
	const int n = n_rings;
	if (lane < n) s2l[lane] = modes[(int64_t)frame * n_rings + lane];
	__syncthreads();
	// ---- move_median_central: trailing nan-median over `width` points (bottleneck.move_median, min_count = 1), shifted to the
	// centre, the ends redone over the first / last k + 2 points
	if (width > 1) {
		if (lane < n) {
			const int a0 = (lane - width + 1 > 0) ? (lane - width + 1) : 0;
			trail[lane] = nan_median(s2l + a0, lane + 1 - a0);
		}
		__syncthreads();
		// np.roll(trail, -width // 2 + 1): Python floor division of the NEGATED width
		const int shift = -((width + 1) / 2) + 1;
		if (lane < n) prof[lane] = trail[((lane - shift) % n + n) % n];
		__syncthreads();
		if (lane == 0) {   // (a handful of points; later ones overwrite earlier ones when the series is very short: in order)
			for (int k = 0; k < width / 2 + 1 && k < n; ++k) {
				const int c0 = (k + 2 < n) ? (k + 2) : n;
				prof[k] = nan_median(s2l, c0);
				prof[n - 1 - k] = nan_median(s2l + n - c0, c0);
			}
		}
	} else if (lane < n) {
		prof[lane] = s2l[lane];
	}
	__syncthreads();
	// ---- the rings that have a mode, in order
	const bool has = (lane < n) && (prof[lane] == prof[lane]);
	const unsigned long long bal = __ballot(has);
	const int m = __popcll(bal);
	if (has) { const int pos = __popcll(bal & ((1ull << lane) - 1ull)); x[pos] = bin_center[lane]; y[pos] = prof[lane]; }
	__syncthreads();
	// fewer than 3 points: "The required number of points for qubic spline" (:183); exactly 3: FITPACK refuses (m > k)
	if (m < 4 || m + 4 > max_knots) { if (lane == 0) n_knots[frame] = 0; return; }
	// ---- knots
	if (lane < 4) { t[lane] = x[0]; t[m + lane] = x[m - 1]; }
	if (lane >= 4 && lane < m) t[lane] = x[lane - 2];
	__syncthreads();
	// ---- collocation rows: the four cubic B-splines that are non-zero at x[i] (de Boor's recurrence); lane i builds row i
	if (lane < m) {
		const int i = lane;
		int l = 3;
		while (l < m - 1 && x[i] >= t[l + 1]) ++l;     // t[l] <= x < t[l + 1]; the last point stays in the last interval
		double h[4] = {1.0, 0.0, 0.0, 0.0}, hh[4];
		for (int j = 1; j <= 3; ++j) {
			for (int q = 0; q < j; ++q) hh[q] = h[q];
			h[0] = 0.0;
			for (int q = 0; q < j; ++q) {
				const int li = l + q + 1, lj = li - j;
				const double f = hh[q] / (t[li] - t[lj]);
				h[q] += f * (t[li] - x[i]);
				h[q + 1] = f * (x[i] - t[lj]);
			}
		}
		// row i: columns l - 3 .. l; stored relative to the first column the elimination can still touch (i - 3 clipped)
		const int jj = (i - 3 > 0) ? (i - 3) : 0;
		j0[i] = jj;
		for (int q = 0; q < 8; ++q) A[i][q] = 0.0;
		for (int q = 0; q < 4; ++q) { const int col = l - 3 + q - jj; if (col >= 0 && col < 8) A[i][col] = h[q]; }
	}
	__syncthreads();
	// ---- Gaussian elimination with partial pivoting inside the band (row i has non-zeros in columns i - 3 .. i + 3 at most).
	// Lane (rr, q) = (lane >> 3, lane & 7) of the first 32 holds entry q of row c + rr.
	const int rr = lane >> 3, q = lane & 7;
	for (int c = 0; c < m; ++c) {
		const int r = c + rr;
		const bool mine = (rr < 4) && (r < m);
		// the rows that can have an entry in column c are c .. c + 3; left of column c they are already zero: re-base them to c
		double val = 0.0;
		if (mine) {
			const int sh = c - j0[r];
			val = (sh > 0) ? ((q + sh < 8) ? A[r][q + sh] : 0.0) : A[r][q];
		}
		__syncthreads();
		if (mine) { A[r][q] = val; if (q == 0) j0[r] = c; }
		__syncthreads();
		int piv = c;
		double best = 0.0;
		for (int k = c; k < m && k <= c + 3; ++k) {
			const double v = fabs(A[k][0]);
			if (v > best) { best = v; piv = k; }
		}
		if (piv != c) {      // (uniform: every lane found the same pivot)
			double u0 = 0.0, u1 = 0.0;
			if (rr == 0) { u0 = A[c][q]; u1 = A[piv][q]; }
			__syncthreads();
			if (rr == 0) { A[c][q] = u1; A[piv][q] = u0; }
			if (lane == 0) { const double ty = y[c]; y[c] = y[piv]; y[piv] = ty; }
			__syncthreads();
		}
		const double d = A[c][0];
		double neu = 0.0, f = 0.0;
		bool upd = false;
		if (mine && rr >= 1) {
			const double a0 = A[r][0];
			if (a0 != 0.0) {
				upd = true;
				f = a0 / d;
				if (q < 7) neu = A[r][q] - f * A[c][q];   // after the exchanges a row reaches at most column c + 6
			}
		}
		__syncthreads();
		if (upd) {
			if (q < 7) A[r][q] = neu;
			else y[r] -= f * y[c];
		}
		__syncthreads();
	}
	if (lane == 0) {
		for (int c = m - 1; c >= 0; --c) {   // row c is stored from column c on
			double ar[7], yr[7];
#pragma unroll
			for (int k = 0; k < 7; ++k) { ar[k] = A[c][k]; yr[k] = (k >= 1 && c + k < m) ? y[c + k] : 0.0; }
			double acc = y[c];
#pragma unroll
			for (int k = 1; k < 7; ++k) if (c + k < m) acc -= ar[k] * yr[k];
			y[c] = acc / ar[0];
		}
	}
	__syncthreads();
	for (int i = lane; i < m + 4; i += 64) kt[i] = t[i];
	for (int i = lane; i < m; i += 64) kc[i] = y[i];
	if (lane == 0) n_knots[frame] = m + 4;
}

struct EvalArgs {
	int n_rows, n_cols; int64_t frame_stride;
	RadialSpline sp;
	const float* add; int64_t add_stride;
	bool zoom_on; ZoomImage zoom;         // the square component evaluated here instead of read from `add`
	float* out;
};

// out = float32(10**spline(r) - zeropoint [+ add]); a frame with n_knots == 0 has no radial component (out = add or 0).
// A workgroup takes 256 columns of a strip of kEvalRows rows; a thread walks down its column (the square component from the mesh
// coefficients: column weights once, row sums once per mesh cell).
constexpr int kEvalRows = 16;
__global__ __launch_bounds__(256) void tp_radial_eval_kernel(EvalArgs a)
{
	extern __shared__ double sp[];   // knots, coefficients, reciprocal knot differences (stage_radial)
	const int frame = blockIdx.z;
	const int n = a.sp.n_knots[frame];
	stage_radial(sp, a.sp.max_knots, a.sp, frame, n, threadIdx.x, 256);
	const int col = blockIdx.x * 256 + threadIdx.x;
	if (col >= a.n_cols) return;
	const int r0 = blockIdx.y * kEvalRows, r1 = (r0 + kEvalRows < a.n_rows) ? (r0 + kEvalRows) : a.n_rows;
	const double zp = a.sp.zeropoint[frame];
	ZoomAxis ax, ay;
	double T[4], lo = 0.0, hi = 0.0;
	int have = -0x7fffffff;
	if (a.zoom_on) { zoom_axis(col, a.zoom.box, a.zoom.nx, ax); lo = a.zoom.vmin[frame]; hi = a.zoom.vmax[frame]; }
	for (int row = r0; row < r1; ++row) {
		const int64_t p = (int64_t)row * a.n_cols + col;
		const double radial = radial_value(sp, a.sp.max_knots, n, zp, a.sp.col_offset, a.sp.xcen, a.sp.ycen, row, col);
		double base = 0.0;
		if (a.add) base = (double)a.add[(int64_t)frame * a.add_stride + p];
		else if (a.zoom_on) {
			zoom_axis(row, a.zoom.box, a.zoom.ny, ay);
			if (ay.start != have) { zoom_sums(a.zoom, frame, ay, ax, T); have = ay.start; }
			base = (double)zoom_from_sums(ay, T, lo, hi);
		}
		a.out[(int64_t)frame * a.frame_stride + p] = (float)(radial + base);
	}
}

} // namespace

static bool radial_image_ok(int32_t n_frames, int64_t n_pixels, int64_t frame_stride) {
	return n_frames >= 0 && n_frames <= 65535 && n_pixels > 0 && n_pixels <= 0x7fffffff && frame_stride >= n_pixels;
}

static bool zoom_image_ok(const tp_zoom_image* z) {
	return z && z->d_coef && z->d_vmin && z->d_vmax && z->mesh_rows > 0 && z->mesh_cols > 0 && z->box_size > 0 && z->frame_cols > 0;
}
static ZoomImage zoom_of(const tp_zoom_image* z) {
	return ZoomImage{z->d_coef, z->d_vmin, z->d_vmax, z->mesh_rows, z->mesh_cols, z->box_size, z->frame_cols};
}

static int radial_zeropoint_launch(tp_ctx* ctx, const float* d_frames, int32_t n_frames, int64_t n_pixels, int64_t frame_stride,
	const float* d_square, int64_t square_frame_stride, const tp_zoom_image* zoom, const uint8_t* d_exclude, int64_t exclude_frame_stride, double flux_cutoff,
	double* d_partial, int32_t n_partial, double* d_zeropoint)
{
	TP_REQUIRE(ctx, d_frames && d_partial && d_zeropoint, "tp_radial_zeropoint: null pointer");
	TP_REQUIRE(ctx, radial_image_ok(n_frames, n_pixels, frame_stride), "tp_radial_zeropoint: bad frame geometry");
	TP_REQUIRE(ctx, n_partial >= 1 && n_partial <= 65535, "tp_radial_zeropoint: n_partial must be 1..65535");
	TP_REQUIRE(ctx, zoom == nullptr || (zoom_image_ok(zoom) && d_square == nullptr && n_pixels % zoom->frame_cols == 0), "tp_radial_zeropoint_zoom: bad mesh image");
	if (n_frames == 0) return TP_OK;
	RadialImage img{d_frames, frame_stride, d_square, square_frame_stride, zoom != nullptr, zoom ? zoom_of(zoom) : ZoomImage{}, d_exclude, exclude_frame_stride, (float)flux_cutoff};
	int n_used = n_partial;
	const int n_cols = zoom ? zoom->frame_cols : 0;
	const int col_blocks = zoom ? (n_cols + kRadThreads - 1) / kRadThreads : 0;
	if (zoom && col_blocks <= n_partial) {
		// column walk: as many row bands as the partial array holds (a minimum does not depend on how the pixels are divided)
		const int n_rows = (int)(n_pixels / n_cols);
		int bands = n_partial / col_blocks;
		if (bands > n_rows) bands = n_rows;
		const int rows_per_band = (n_rows + bands - 1) / bands;
		bands = (n_rows + rows_per_band - 1) / rows_per_band;
		n_used = bands * col_blocks;
		TP_LAUNCH(ctx, TPK_BKG_RADIAL, tp_radial_min_zoom_kernel, dim3((unsigned)col_blocks, (unsigned)bands, (unsigned)n_frames), dim3(kRadThreads), 0,
			img, n_rows, n_cols, rows_per_band, d_partial);
	} else {
		TP_LAUNCH(ctx, TPK_BKG_RADIAL, tp_radial_min_partial_kernel, dim3((unsigned)n_partial, (unsigned)n_frames), dim3(kRadThreads), 0, img, n_pixels, d_partial);
	}
	TP_LAUNCH(ctx, TPK_BKG_RADIAL, tp_radial_min_final_kernel, dim3((unsigned)n_frames), dim3(kRadThreads), 0, d_partial, n_used, d_zeropoint);
	TP_LAUNCH_CHECK(ctx, "tp_radial_min kernels");
	return TP_OK;
}

extern "C" int tp_radial_zeropoint(tp_ctx* ctx, const float* d_frames, int32_t n_frames, int64_t n_pixels, int64_t frame_stride,
	const float* d_square, int64_t square_frame_stride, const uint8_t* d_exclude, int64_t exclude_frame_stride, double flux_cutoff,
	double* d_partial, int32_t n_partial, double* d_zeropoint)
{
	TP_CHECK_CTX(ctx);
	TP_API_BEGIN
	return radial_zeropoint_launch(ctx, d_frames, n_frames, n_pixels, frame_stride, d_square, square_frame_stride, nullptr, d_exclude, exclude_frame_stride,
		flux_cutoff, d_partial, n_partial, d_zeropoint);
	TP_API_END(ctx)
}

extern "C" int tp_radial_zeropoint_zoom(tp_ctx* ctx, const float* d_frames, int32_t n_frames, int64_t n_pixels, int64_t frame_stride,
	const tp_zoom_image* square, const uint8_t* d_exclude, int64_t exclude_frame_stride, double flux_cutoff,
	double* d_partial, int32_t n_partial, double* d_zeropoint)
{
	TP_CHECK_CTX(ctx);
	TP_API_BEGIN
	TP_REQUIRE(ctx, square != nullptr, "tp_radial_zeropoint_zoom: null mesh image");
	return radial_zeropoint_launch(ctx, d_frames, n_frames, n_pixels, frame_stride, nullptr, 0, square, d_exclude, exclude_frame_stride,
		flux_cutoff, d_partial, n_partial, d_zeropoint);
	TP_API_END(ctx)
}

static int radial_ring_launch(tp_ctx* ctx, const float* d_frames, int32_t n_frames, int64_t n_pixels, int64_t frame_stride,
	const float* d_square, int64_t square_frame_stride, const tp_zoom_image* zoom, const uint8_t* d_exclude, int64_t exclude_frame_stride, double flux_cutoff,
	const double* d_zeropoint, const int32_t* d_ring_pixels, const int32_t* d_ring_offsets, int32_t n_rings, int32_t n_ring_pixels,
	double bandwidth_constant, double* d_scratch, double* d_modes, int32_t* d_counts)
{
	TP_REQUIRE(ctx, d_frames && d_zeropoint && d_ring_pixels && d_ring_offsets && d_scratch && d_modes, "tp_radial_ring_modes: null pointer");
	TP_REQUIRE(ctx, radial_image_ok(n_frames, n_pixels, frame_stride), "tp_radial_ring_modes: bad frame geometry");
	TP_REQUIRE(ctx, n_rings >= 0 && n_rings <= 65535 && n_ring_pixels >= 0, "tp_radial_ring_modes: bad ring list");
	TP_REQUIRE(ctx, zoom == nullptr || (zoom_image_ok(zoom) && d_square == nullptr && n_pixels % zoom->frame_cols == 0), "tp_radial_ring_modes_zoom: bad mesh image");
	if (n_frames == 0 || n_rings == 0) return TP_OK;
	RingArgs a;
	a.img = RadialImage{d_frames, frame_stride, d_square, square_frame_stride, zoom != nullptr, zoom ? zoom_of(zoom) : ZoomImage{}, d_exclude, exclude_frame_stride, (float)flux_cutoff};
	a.zeropoint = d_zeropoint;
	a.ring_pixels = d_ring_pixels; a.ring_offsets = d_ring_offsets; a.n_rings = n_rings;
	a.scratch = d_scratch; a.scratch_stride = n_ring_pixels;
	a.bw_constant = bandwidth_constant;
	a.modes = d_modes; a.counts = d_counts;
	TP_LAUNCH(ctx, TPK_BKG_RADIAL, tp_radial_ring_kernel, dim3((unsigned)n_rings, (unsigned)n_frames), dim3(kRadThreads), 0, a);
	TP_LAUNCH_CHECK(ctx, "tp_radial_ring_kernel");
	return TP_OK;
}

extern "C" int tp_radial_ring_modes(tp_ctx* ctx, const float* d_frames, int32_t n_frames, int64_t n_pixels, int64_t frame_stride,
	const float* d_square, int64_t square_frame_stride, const uint8_t* d_exclude, int64_t exclude_frame_stride, double flux_cutoff,
	const double* d_zeropoint, const int32_t* d_ring_pixels, const int32_t* d_ring_offsets, int32_t n_rings, int32_t n_ring_pixels,
	double bandwidth_constant, double* d_scratch, double* d_modes, int32_t* d_counts)
{
	TP_CHECK_CTX(ctx);
	TP_API_BEGIN
	return radial_ring_launch(ctx, d_frames, n_frames, n_pixels, frame_stride, d_square, square_frame_stride, nullptr, d_exclude, exclude_frame_stride, flux_cutoff,
		d_zeropoint, d_ring_pixels, d_ring_offsets, n_rings, n_ring_pixels, bandwidth_constant, d_scratch, d_modes, d_counts);
	TP_API_END(ctx)
}

extern "C" int tp_radial_ring_modes_zoom(tp_ctx* ctx, const float* d_frames, int32_t n_frames, int64_t n_pixels, int64_t frame_stride,
	const tp_zoom_image* square, const uint8_t* d_exclude, int64_t exclude_frame_stride, double flux_cutoff,
	const double* d_zeropoint, const int32_t* d_ring_pixels, const int32_t* d_ring_offsets, int32_t n_rings, int32_t n_ring_pixels,
	double bandwidth_constant, double* d_scratch, double* d_modes, int32_t* d_counts)
{
	TP_CHECK_CTX(ctx);
	TP_API_BEGIN
	TP_REQUIRE(ctx, square != nullptr, "tp_radial_ring_modes_zoom: null mesh image");
	return radial_ring_launch(ctx, d_frames, n_frames, n_pixels, frame_stride, nullptr, 0, square, d_exclude, exclude_frame_stride, flux_cutoff,
		d_zeropoint, d_ring_pixels, d_ring_offsets, n_rings, n_ring_pixels, bandwidth_constant, d_scratch, d_modes, d_counts);
	TP_API_END(ctx)
}

static int radial_evaluate_launch(tp_ctx* ctx, int32_t n_frames, int32_t frame_rows, int32_t frame_cols, int64_t frame_stride,
	double col_offset, double xcen, double ycen, const double* d_knots, const double* d_coefs, const int32_t* d_n_knots, int32_t max_knots,
	const double* d_zeropoint, const float* d_add, int64_t add_frame_stride, const tp_zoom_image* zoom, float* d_out)
{
	TP_REQUIRE(ctx, d_knots && d_coefs && d_n_knots && d_zeropoint && d_out, "tp_radial_evaluate: null pointer");
	TP_REQUIRE(ctx, n_frames >= 0 && n_frames <= 65535 && frame_rows > 0 && frame_rows <= 65535 && frame_cols > 0
		&& frame_stride >= (int64_t)frame_rows * frame_cols, "tp_radial_evaluate: bad frame geometry");
	TP_REQUIRE(ctx, max_knots >= 8 && max_knots <= 1024, "tp_radial_evaluate: max_knots must be 8..1024");
	TP_REQUIRE(ctx, zoom == nullptr || (zoom_image_ok(zoom) && d_add == nullptr && zoom->frame_cols == frame_cols), "tp_radial_evaluate_zoom: bad mesh image");
	if (n_frames == 0) return TP_OK;
	EvalArgs a;
	a.n_rows = frame_rows; a.n_cols = frame_cols; a.frame_stride = frame_stride;
	a.sp = RadialSpline{col_offset, xcen, ycen, d_knots, d_coefs, d_n_knots, max_knots, d_zeropoint};
	a.add = d_add; a.add_stride = add_frame_stride;
	a.zoom_on = zoom != nullptr; a.zoom = zoom ? zoom_of(zoom) : ZoomImage{};
	a.out = d_out;
	dim3 grid((unsigned)((frame_cols + 255) / 256), (unsigned)((frame_rows + kEvalRows - 1) / kEvalRows), (unsigned)n_frames);
	TP_LAUNCH(ctx, TPK_BKG_RADIAL, tp_radial_eval_kernel, grid, dim3(256), (size_t)kRadialStageDoubles(max_knots) * sizeof(double), a);
	TP_LAUNCH_CHECK(ctx, "tp_radial_eval_kernel");
	return TP_OK;
}

extern "C" int tp_radial_evaluate(tp_ctx* ctx, int32_t n_frames, int32_t frame_rows, int32_t frame_cols, int64_t frame_stride,
	double col_offset, double xcen, double ycen, const double* d_knots, const double* d_coefs, const int32_t* d_n_knots, int32_t max_knots,
	const double* d_zeropoint, const float* d_add, int64_t add_frame_stride, float* d_out)
{
	TP_CHECK_CTX(ctx);
	TP_API_BEGIN
	return radial_evaluate_launch(ctx, n_frames, frame_rows, frame_cols, frame_stride, col_offset, xcen, ycen, d_knots, d_coefs, d_n_knots, max_knots,
		d_zeropoint, d_add, add_frame_stride, nullptr, d_out);
	TP_API_END(ctx)
}

extern "C" int tp_radial_evaluate_zoom(tp_ctx* ctx, int32_t n_frames, int32_t frame_rows, int32_t frame_cols, int64_t frame_stride,
	const tp_radial_image* radial, const tp_zoom_image* add, float* d_out)
{
	TP_CHECK_CTX(ctx);
	TP_API_BEGIN
	TP_REQUIRE(ctx, radial != nullptr && add != nullptr, "tp_radial_evaluate_zoom: null ring profile / mesh image");
	return radial_evaluate_launch(ctx, n_frames, frame_rows, frame_cols, frame_stride, radial->col_offset, radial->xcen, radial->ycen,
		radial->d_knots, radial->d_coefs, radial->d_n_knots, radial->max_knots, radial->d_zeropoint, nullptr, 0, add, d_out);
	TP_API_END(ctx)
}

extern "C" int tp_radial_profiles(tp_ctx* ctx, int32_t n_frames, int32_t n_rings, const double* d_modes, const double* d_bin_center,
	int32_t radial_smooth, int32_t max_knots, double* d_knots, double* d_coefs, int32_t* d_n_knots)
{
	TP_CHECK_CTX(ctx);
	TP_API_BEGIN
	TP_REQUIRE(ctx, d_modes && d_bin_center && d_knots && d_coefs && d_n_knots, "tp_radial_profiles: null pointer");
	TP_REQUIRE(ctx, n_frames >= 0 && n_rings >= 1 && n_rings <= kMaxRings, "tp_radial_profiles: at most 64 rings");
	TP_REQUIRE(ctx, radial_smooth >= 0 && radial_smooth <= 8 && max_knots >= n_rings + 4, "tp_radial_profiles: radial_smooth must be 0..8, max_knots >= n_rings + 4");
	if (n_frames == 0) return TP_OK;
	TP_LAUNCH(ctx, TPK_BKG_RADIAL, tp_radial_profile_kernel, dim3((unsigned)n_frames), dim3(64), 0, d_modes, d_bin_center, (int)n_rings,
		(int)radial_smooth, (int)n_frames, (int)max_knots, d_knots, d_coefs, d_n_knots);
	TP_LAUNCH_CHECK(ctx, "tp_radial_profile_kernel");
	return TP_OK;
	TP_API_END(ctx)
}
