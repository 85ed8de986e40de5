// diagnostics.hip -- light-curve diagnostics of a batch (the reductions that run on every
// OK / WARNING target right after the hot path and feed the scheduler's `diagnostics` table).
//
// Replaces the block of BasePhotometry.photometry (photometry/BasePhotometry.py:1343-1407) and
// utilities.rms_timescale (photometry/utilities.py:227-264):
//   mean_flux   = nanmedian(flux[good])                                                  (:1357)
//   rel = flux/mean_flux - 1, rel_err = |1/mean_flux| * flux_err                         (:1360-1361)
//   variance    = nanvar(rel, ddof=1)                                                    (:1364)
//   rms_hour    = 1.4826 * nanmedian(|b - nanmedian(b)|), b = nanmean of rel per one-hour time bin (:1365)
//   ptp         = nanmedian(|diff(rel)|)                                                 (:1366)
//   pos_centroid= nanmedian(pos_centroid[good], axis=0)                                  (:1369)
//   variability = nanstd(rel - cubic weighted polyfit) / nanmedian(rel_err)              (:1372-1393)
//   mask_size, edge_flux = sum(mask), nansum(sumimage[mask & stamp edge])                (:1394-1403)
// "good" = cadences whose quality passes the TESS default bitmask (:1353).
//
// Mapping (gfx950): one 256-thread workgroup per target, the good-cadence series in LDS; every median is a radix SELECT
// on order-preserving 64-bit keys (block_median: one 256-bin LDS histogram per digit below the keys' common prefix; the
// first version sorted bitonically, 66 barrier-separated stages per median), sums are fixed-shape tree reductions (deterministic).
// The cubic fit is done on the time axis mapped to [-1, 1] (the fitted polynomial is invariant under an affine change
// of variable, the normal equations then have a condition number of ~1e3 instead of ~1e12).
// Bytes: 4 series x T x 8 per target in, 80 B out -- negligible next to the cubes; latency-bound.
#include "common.h"
#include <cmath>

void* tp_ctx_scratch(tp_ctx* ctx, size_t bytes); // aperture.hip

namespace {

constexpr int kThreads = 256;

struct DiagArgs {
	const double* flux; const double* flux_err; const double* ccol; const double* crow; int64_t lc_pitch;
	const double* time; const int32_t* quality; int64_t quality_stride; uint32_t bitmask;
	const int32_t* status; const double* sumimage; const uint8_t* mask; int height, width;
	int n_cad; int tp2; double timescale; double* out;
	unsigned char* gscratch; size_t gscratch_per_target; // series arrays in HBM when they do not fit the LDS (long light curves)
};

enum { F_ALLNAN_FLUX = 1, F_ALLNAN_ERR = 2, F_BAD_TIME = 4, F_NO_DETREND = 8, F_TOO_MANY_BINS = 16 };

__device__ __forceinline__ bool is_nan(double x) { return x != x; }
__device__ __forceinline__ bool is_finite(double x) { return fabs(x) <= 1.7976931348623157e308; }

// fixed-shape reductions over the workgroup (every thread gets the result): a shuffle tree inside each wavefront,
// then the four partial results in wavefront order -- deterministic, two barriers
template <class OP>
__device__ __forceinline__ double block_reduce(double v, double* red, OP op) {
	const int tid = threadIdx.x;
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) v = op(v, __shfl_down(v, off, 64));
	__syncthreads(); // the previous result in red[] has been consumed
	if ((tid & 63) == 0) red[tid >> 6] = v;
	__syncthreads();
	double r = red[0];
#pragma unroll
	for (int w = 1; w < kThreads / 64; ++w) r = op(r, red[w]);
	return r;
}
__device__ double block_sum(double v, double* red) { return block_reduce(v, red, [](double x, double y) { return x + y; }); }
__device__ double block_min(double v, double* red) { return block_reduce(v, red, [](double x, double y) { return (y < x) ? y : x; }); }
__device__ double block_max(double v, double* red) { return block_reduce(v, red, [](double x, double y) { return (y > x) ? y : x; }); }

// order-preserving map double -> uint64 (negative values: all bits flipped, others: sign bit set)
__device__ __forceinline__ unsigned long long sort_key(double x) {
	const unsigned long long u = (unsigned long long)__double_as_longlong(x);
	return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}

// nanmedian of v[0..m) (LDS, not modified) by an 8-pass radix SELECT on the 64-bit keys: per pass one 256-bin LDS
// histogram of the digit among the candidates that share the prefix found so far, one wavefront picks the bin that
// holds the wanted rank.  A few hundred cycles per pass instead of the 66 barrier-separated stages of a bitonic sort.
// `hist` is a 260-entry LDS scratch.  Returns NaN when every entry is NaN.
__device__ double block_median(const double* v, int m, double* red, unsigned int* hist) {
	const int tid = threadIdx.x;
	int c = 0;
	for (int i = tid; i < m; i += kThreads) c += !is_nan(v[i]);
	const int n = (int)block_sum((double)c, red);
	if (n == 0) return __builtin_nan("");
	const int k_lo = (n - 1) >> 1, k_hi = n >> 1;
	// The values of a light curve share their sign, exponent and leading mantissa bits: in the top digits every
	// candidate falls into ONE histogram bin, i.e. every lane of every wavefront does an atomic on the same LDS word.
	// Those digits are known from the extreme keys: start below the common prefix of the smallest and the largest key.
	double vmn = __builtin_inf(), vmx = -__builtin_inf();
	for (int i = tid; i < m; i += kThreads) { const double x = v[i]; if (!is_nan(x)) { if (x < vmn) vmn = x; if (x > vmx) vmx = x; } }
	const unsigned long long kmin = sort_key(block_min(vmn, red)), kmax = sort_key(block_max(vmx, red));
	const unsigned long long diff = kmin ^ kmax;
	int top = 7;
	while (top >= 0 && ((diff >> (top * 8)) & 255ull) == 0ull) --top; // digits above `top` are common to all keys
	unsigned long long prefix = 0ull, pmask = 0ull;
	if (top < 7) { pmask = ~0ull << ((top + 1) * 8); prefix = kmin & pmask; }
	int k = k_lo;
	for (int pass = top; pass >= 0; --pass) {
		const int shift = pass * 8;
		hist[tid] = 0u;
		__syncthreads();
		for (int i = tid; i < m; i += kThreads) {
			const double x = v[i];
			if (is_nan(x)) continue;
			const unsigned long long key = sort_key(x);
			if ((key & pmask) == prefix) atomicAdd(&hist[(unsigned)(key >> shift) & 255u], 1u);
		}
		__syncthreads();
		if (tid < 64) { // one wavefront: lane l owns bins 4l..4l+3
			const unsigned c0 = hist[4 * tid], c1 = hist[4 * tid + 1], c2 = hist[4 * tid + 2], c3 = hist[4 * tid + 3];
			const unsigned tot = c0 + c1 + c2 + c3;
			unsigned inc = tot;
#pragma unroll
			for (int off = 1; off < 64; off <<= 1) { const unsigned o = __shfl_up(inc, off, 64); if (tid >= off) inc += o; }
			const unsigned exc = inc - tot;
			if ((unsigned)k >= exc && (unsigned)k < inc) {
				unsigned r = (unsigned)k - exc, d = 0;
				if (r >= c0) { r -= c0; d = 1; if (r >= c1) { r -= c1; d = 2; if (r >= c2) { r -= c2; d = 3; } } }
				hist[256] = 4u * tid + d;
				hist[257] = r;
			}
		}
		__syncthreads();
		prefix |= ((unsigned long long)hist[256]) << shift;
		pmask |= 255ull << shift;
		k = (int)hist[257];
		__syncthreads();
	}
	// prefix is the key of the k_lo-th value: recover the double
	const unsigned long long u = (prefix >> 63) ? (prefix & 0x7fffffffffffffffull) : ~prefix;
	const double v_lo = __longlong_as_double((long long)u);
	if (k_hi == k_lo) return v_lo;
	// the next order statistic: v_lo again if enough values are <= v_lo, else the smallest value above it
	int cle = 0;
	double above = __builtin_inf();
	for (int i = tid; i < m; i += kThreads) {
		const double x = v[i];
		if (is_nan(x)) continue;
		if (x <= v_lo) cle++; else if (x < above) above = x;
	}
	const int nle = (int)block_sum((double)cle, red);
	const double amin = block_min(above, red);
	const double v_hi = (nle > k_hi) ? v_lo : amin;
	return (v_lo + v_hi) / 2.0;
}

// numpy's pairwise add.reduce on n <= 128 contiguous doubles (loops_utils.h.src)
__device__ double pairwise_leaf(const double* a, int n) {
	if (n < 8) { double r = 0.0; for (int i = 0; i < n; ++i) r += a[i]; return 0.0 + r; }
	double r[8];
	for (int j = 0; j < 8; ++j) r[j] = a[j];
	int i = 8;
	for (; i < n - (n % 8); i += 8) for (int j = 0; j < 8; ++j) r[j] += a[i + j];
	double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
	for (; i < n; ++i) res += a[i];
	return res;
}

// numpy's pairwise add.reduce for any n: pw(n) = leaf if n <= 128, else pw(n2) + pw(n - n2) with n2 = n / 2 rounded down to a
// multiple of 8.  Run by one thread; the recursion is an explicit stack in the workgroup's scratch (left sums in dstack,
// offset / length / state triples in istack; 16 levels cover 128 * 2^16 values).
__device__ double pairwise_sum(const double* a, int n, double* dstack, int* istack) {
	int sp = 0;
	istack[0] = 0; istack[1] = n; istack[2] = 0;
	double ret = 0.0;
	while (true) {
		int* f = istack + 3 * sp;
		const int off = f[0], len = f[1];
		int n2 = len / 2; n2 -= n2 % 8;
		if (f[2] == 0) {
			if (len > 128 && sp < 15) { f[2] = 1; ++sp; istack[3 * sp] = off; istack[3 * sp + 1] = n2; istack[3 * sp + 2] = 0; continue; }
			ret = pairwise_leaf(a + off, len);
		} else if (f[2] == 1) {
			dstack[sp] = ret; f[2] = 2; ++sp; istack[3 * sp] = off + n2; istack[3 * sp + 1] = len - n2; istack[3 * sp + 2] = 0; continue;
		} else ret = dstack[sp] + ret;
		if (sp == 0) break;
		--sp;
	}
	return ret;
}

__global__ __launch_bounds__(kThreads) void tp_diagnostics_kernel(DiagArgs a)
{
	extern __shared__ __align__(16) double lds[];
	const int target = blockIdx.x;
	const int tid = threadIdx.x;
	const int T = a.n_cad, TP2 = a.tp2;
	// small reduction scratch always in LDS; the series arrays in LDS when they fit, else in a per-target slice of HBM
	// (2-minute-cadence light curves: ~20 000 cadences; same code, the reductions then run out of L2)
	double* red = lds;                 // [kThreads]
	int* ired = reinterpret_cast<int*>(red + kThreads); // [kThreads + 1]
	unsigned int* hist = reinterpret_cast<unsigned int*>(ired + kThreads + 1); // [260] radix-select scratch
	double* series = a.gscratch ? reinterpret_cast<double*>(a.gscratch + (size_t)blockIdx.x * a.gscratch_per_target)
		: reinterpret_cast<double*>(hist + 260 + ((kThreads + 1 + 260) & 1)); // keep 8-byte alignment
	double* srt = series;              // [TP2] scratch series (TP2 = max(T, 256))
	double* fb = srt;                  // the binned flux of the rms section lives in the same scratch (its deviations replace it in place)
	double* gflux = srt + TP2;         // [T] relative flux of the good cadences
	// the relative error and the time of a good cadence are not kept in the series scratch: they are one multiplication / one
	// load away from arrays that stay in L2 (with the bin array sharing the scratch series: 31 instead of 62 KB of LDS per 1300-cadence target, four workgroups per CU)
	int* gk = reinterpret_cast<int*>(gflux + T); // [T] original index of the g-th good cadence
	int* bt = gk + T;                  // [T] time bin of every good cadence
	double* o = a.out + (int64_t)target * 10;
	const double nan = __builtin_nan("");

	const int st = a.status ? a.status[target] : TP_STATUS_OK;
	if (st != TP_STATUS_OK && st != TP_STATUS_WARNING) { // :1343: only OK / WARNING targets get diagnostics
		if (tid < 10) o[tid] = nan;
		return;
	}
	const double* flux = a.flux + (int64_t)target * a.lc_pitch;
	const double* ferr = a.flux_err + (int64_t)target * a.lc_pitch;
	const double* ccol = a.ccol + (int64_t)target * a.lc_pitch;
	const double* crow = a.crow + (int64_t)target * a.lc_pitch;
	const int32_t* q = a.quality + (int64_t)target * a.quality_stride;
	int flags = 0;

	// ---- mask size and flux on the stamp edge (:1394-1403)
	double mask_size = nan, edge_flux = nan;
	if (a.mask && a.sumimage) {
		const int H = a.height, W = a.width, P = H * W;
		const uint8_t* m = a.mask + (int64_t)target * P;
		const double* S = a.sumimage + (int64_t)target * P;
		int c = 0;
		for (int p = tid; p < P; p += kThreads) c += m[p] ? 1 : 0;
		mask_size = block_sum((double)c, red);
		// the selected pixels in raster order (ballot compaction, wavefront after wavefront), NaN -> 0
		if (tid == 0) ired[0] = 0;
		__syncthreads();
		for (int p0 = 0; p0 < P; p0 += kThreads) {
			const int p = p0 + tid;
			bool sel = false;
			double val = 0.0;
			if (p < P) {
				const int r = p / W, cc = p - r * W;
				sel = m[p] && (r == 0 || r == H - 1 || cc == 0 || cc == W - 1);
				if (sel) { val = S[p]; if (is_nan(val)) val = 0.0; }
			}
			const unsigned long long bal = __ballot(sel);
			const int lane = tid & 63, wv = tid >> 6;
			for (int w = 0; w < kThreads / 64; ++w) { // waves in order
				if (wv == w) {
					const int base = ired[0];
					if (sel) srt[base + __popcll(bal & ((1ull << lane) - 1ull))] = val;
					if (lane == 0) ired[0] = base + (int)__popcll(bal);
				}
				__syncthreads();
			}
		}
		if (tid == 0) { // numpy pairwise sum of the gathered values
			const int n = ired[0];
			red[0] = pairwise_sum(srt, n, red + 16, ired + 16);
		}
		__syncthreads();
		edge_flux = red[0];
		__syncthreads();
	}

	// ---- all-NaN checks over the whole light curve (:1346-1349)
	{
		int anyf = 0, anye = 0;
		for (int k = tid; k < T; k += kThreads) { anyf |= !is_nan(flux[k]); anye |= !is_nan(ferr[k]); }
		const double sf = block_sum((double)anyf, red), se = block_sum((double)anye, red);
		if (sf == 0.0) flags |= F_ALLNAN_FLUX;
		else if (se == 0.0) flags |= F_ALLNAN_ERR;
	}
	if (flags) {
		if (tid == 0) { for (int i = 0; i < 7; ++i) o[i] = nan; o[7] = mask_size; o[8] = edge_flux; o[9] = (double)flags; }
		return;
	}

	// ---- ordered compaction of the good cadences (:1353)
	const int chunk = (T + kThreads - 1) / kThreads;
	{
		int c = 0;
		for (int k = tid * chunk; k < (tid + 1) * chunk && k < T; ++k) c += (((uint32_t)q[k] & a.bitmask) == 0u) ? 1 : 0;
		ired[tid] = c;
		__syncthreads();
		if (tid == 0) { int run = 0; for (int t = 0; t < kThreads; ++t) { const int v = ired[t]; ired[t] = run; run += v; } ired[kThreads] = run; }
		__syncthreads();
		int pos = ired[tid];
		for (int k = tid * chunk; k < (tid + 1) * chunk && k < T; ++k) if (((uint32_t)q[k] & a.bitmask) == 0u) gk[pos++] = k;
		__syncthreads();
	}
	const int Ng = ired[kThreads];

	// ---- mean flux, relative flux and error (:1357-1361)
	for (int g = tid; g < Ng; g += kThreads) srt[g] = flux[gk[g]];
	const double mean_flux = block_median(srt, Ng, red, hist);
	const double ainv = fabs(1.0 / mean_flux);
	for (int g = tid; g < Ng; g += kThreads) gflux[g] = (flux[gk[g]] / mean_flux) - 1.0;
	__syncthreads();
	const double* tptr = a.time;
	auto gtime_at = [&](int g) { return tptr[gk[g]]; };
	auto gerr_at = [&](int g) { return ainv * ferr[gk[g]]; };

	// ---- variance = nanvar(rel, ddof=1) (:1364): two passes
	double variance;
	{
		double s = 0.0; int c = 0;
		for (int g = tid; g < Ng; g += kThreads) { const double v = gflux[g]; if (!is_nan(v)) { s += v; c++; } }
		const double tot = block_sum(s, red), cnt = block_sum((double)c, red);
		const double avg = tot / cnt;
		double s2 = 0.0;
		for (int g = tid; g < Ng; g += kThreads) { const double v = gflux[g]; if (!is_nan(v)) { const double d = v - avg; s2 += d * d; } }
		const double tot2 = block_sum(s2, red);
		variance = (cnt - 1.0 > 0.0) ? tot2 / (cnt - 1.0) : nan;
		if (cnt == 0.0) variance = nan;
	}

	// ---- rms on the one-hour time scale (utilities.py:227-264)
	double rms_hour = nan;
	{
		int cf = 0, ct = 0;
		double tmn = __builtin_inf(), tmx = -__builtin_inf();
		for (int g = tid; g < Ng; g += kThreads) {
			cf += !is_nan(gflux[g]);
			const double t = gtime_at(g);
			if (!is_nan(t)) { ct++; if (t < tmn) tmn = t; if (t > tmx) tmx = t; }
		}
		const double nfl = block_sum((double)cf, red), nt = block_sum((double)ct, red);
		const double tmin = block_min(tmn, red), tmax = block_max(tmx, red);
		if (Ng > 0 && nfl > 0.0) {
			if (nt == 0.0 || !is_finite(tmin) || !is_finite(tmax) || !(tmax - tmin > 0.0)) flags |= F_BAD_TIME;
			else {
				const double ts = a.timescale;
				const double nbd = ceil((tmax - tmin) / ts); // len(np.arange(tmin, tmax, ts))
				// more bins than the bin array holds (a sparse series: 100 points over 27 days in the reference's own test): the
				// empty bins are NaN and nanmedian ignores them, so for a time-ordered series the means of the non-empty bins are
				// kept at the position of the first sample of their run instead
				const bool many = !(nbd <= (double)TP2);
				if (many && !(nbd <= 2.0e9)) flags |= F_TOO_MANY_BINS;
				else {
					const int nb = (int)nbd;
					// numpy's arange fills start + i*delta with delta = (start + step) - start, rounded: NOT i*step
					const double delta = (tmin + ts) - tmin;
					// bin of every finite sample: searchsorted(right) on the edges tmin + i*delta (i < nb), tmax; the last bin is closed
					int mono = 1;
					for (int g = tid; g < Ng; g += kThreads) {
						int b = -1;
						const double x = gtime_at(g);
						if (!is_nan(x)) {
							b = (int)floor((x - tmin) / delta);
							if (b < 0) b = 0;
							if (b > nb - 1) b = nb - 1;
							while (b + 1 <= nb - 1 && (tmin + (double)(b + 1) * delta) <= x) ++b;
							while (b > 0 && (tmin + (double)b * delta) > x) --b;
						}
						bt[g] = b;
					}
					__syncthreads();
					for (int g = tid; g + 1 < Ng; g += kThreads) if (bt[g] < 0 || bt[g + 1] < bt[g]) mono = 0;
					if (Ng > 0 && tid == 0 && bt[Ng - 1] < 0) mono = 0;
					const bool sorted = block_min((double)mono, red) > 0.0;
					if (many) {
						if (!sorted) flags |= F_TOO_MANY_BINS;
						else {
							for (int g = tid; g < Ng; g += kThreads) {
								double v = nan;
								if (g == 0 || bt[g] != bt[g - 1]) {
									const int b = bt[g];
									double sacc = 0.0; int c = 0;
									for (int q = g; q < Ng && bt[q] == b; ++q) { const double y = gflux[q]; if (!is_nan(y) && is_finite(y)) { sacc += y; c++; } }
									v = c ? (0.0 + sacc) / (double)c : nan;
								}
								fb[g] = v;
							}
							__syncthreads();
							const double med1 = block_median(fb, Ng, red, hist);
							for (int g = tid; g < Ng; g += kThreads) srt[g] = fabs(fb[g] - med1);   // in place: fb IS srt
							const double med2 = block_median(srt, Ng, red, hist);
							rms_hour = 1.482602218505602 * med2;
						}
					} else {
					// nanmean per bin, samples added in time-series order (numpy's order inside binned_statistic).  Time-ordered
					// series (the normal case): the samples of a bin are a contiguous run found by two binary searches.
					for (int b = tid; b < nb; b += kThreads) {
						int g0 = 0, g1 = Ng;
						if (sorted) {
							int lo = 0, hi = Ng;
							while (lo < hi) { const int mid = (lo + hi) >> 1; if (bt[mid] < b) lo = mid + 1; else hi = mid; }
							g0 = lo; hi = Ng;
							while (lo < hi) { const int mid = (lo + hi) >> 1; if (bt[mid] <= b) lo = mid + 1; else hi = mid; }
							g1 = lo;
						}
						double sacc = 0.0; int c = 0;
						for (int g = g0; g < g1; ++g) { const double y = gflux[g]; if (bt[g] == b && !is_nan(y) && is_finite(y)) { sacc += y; c++; } }
						fb[b] = c ? (0.0 + sacc) / (double)c : nan;
					}
					__syncthreads();
					const double med1 = block_median(fb, nb, red, hist);
					for (int b = tid; b < nb; b += kThreads) srt[b] = fabs(fb[b] - med1);   // in place: fb IS srt
					const double med2 = block_median(srt, nb, red, hist);
					rms_hour = 1.482602218505602 * med2; // utilities.mad_to_sigma (:25)
					}
				}
			}
		}
	}

	// ---- point-to-point scatter (:1366)
	for (int g = tid; g + 1 < Ng; g += kThreads) srt[g] = fabs(gflux[g + 1] - gflux[g]);
	const double ptp = block_median(srt, (Ng > 0) ? (Ng - 1) : 0, red, hist);

	// ---- median centroid (:1369)
	for (int g = tid; g < Ng; g += kThreads) srt[g] = ccol[gk[g]];
	const double cen_col = block_median(srt, Ng, red, hist);
	for (int g = tid; g < Ng; g += kThreads) srt[g] = crow[gk[g]];
	const double cen_row = block_median(srt, Ng, red, hist);

	// ---- variability (:1372-1393): weighted cubic fit, standard deviation of the residuals / median error
	double variability;
	{
		int c = 0;
		double tmn = __builtin_inf(), tmx = -__builtin_inf();
		for (int g = tid; g < Ng; g += kThreads) {
			const double t = gtime_at(g);
			const bool ok = is_finite(t) && is_finite(gflux[g]) && is_finite(gerr_at(g));
			if (ok) { c++; if (t < tmn) tmn = t; if (t > tmx) tmx = t; }
		}
		const double nfit = block_sum((double)c, red);
		const double t0 = block_min(tmn, red), t1 = block_max(tmx, red);
		double pc[4] = {0.0, 0.0, 0.0, 0.0}; // polynomial in u = (t - mid) / half, powers 0..3
		bool have_fit = false;
		const double mid = 0.5 * (t0 + t1), half = (t1 - t0 > 0.0) ? 0.5 * (t1 - t0) : 1.0;
		if (nfit > 0.0) {
			// moments of the weighted normal equations: M[k] = sum w^2 u^k (k = 0..6), R[k] = sum w^2 u^k y (k = 0..3)
			double mloc[7] = {0, 0, 0, 0, 0, 0, 0}, rloc[4] = {0, 0, 0, 0};
			for (int g = tid; g < Ng; g += kThreads) {
				const double t = gtime_at(g), ge = gerr_at(g);
				const bool ok = is_finite(t) && is_finite(gflux[g]) && is_finite(ge);
				if (!ok) continue;
				const double u = (t - mid) / half, w = 1.0 / ge, w2 = w * w, y = gflux[g];
				double p = w2;
				for (int k = 0; k < 7; ++k) { mloc[k] += p; if (k < 4) rloc[k] += p * y; p *= u; }
			}
			double M[7], R[4];
			for (int k = 0; k < 7; ++k) M[k] = block_sum(mloc[k], red);
			for (int k = 0; k < 4; ++k) R[k] = block_sum(rloc[k], red);
			// Cholesky of the 4x4 Hankel matrix A[i][j] = M[i+j]; a vanishing pivot = rank deficiency (np.RankWarning)
			double L[4][4] = {{0}};
			bool ok = true;
			for (int i = 0; i < 4 && ok; ++i) {
				for (int j = 0; j <= i; ++j) {
					double s = M[i + j];
					for (int k = 0; k < j; ++k) s -= L[i][k] * L[j][k];
					if (i == j) { if (!(s > 1e-13 * M[2 * i]) || !is_finite(s)) { ok = false; break; } L[i][i] = sqrt(s); }
					else L[i][j] = s / L[j][j];
				}
			}
			if (ok) {
				double z[4];
				for (int i = 0; i < 4; ++i) { double s = R[i]; for (int k = 0; k < i; ++k) s -= L[i][k] * z[k]; z[i] = s / L[i][i]; }
				for (int i = 3; i >= 0; --i) { double s = z[i]; for (int k = i + 1; k < 4; ++k) s -= L[k][i] * pc[k]; pc[i] = s / L[i][i]; }
				have_fit = true;
			}
		}
		if (!have_fit) flags |= F_NO_DETREND; // "Could not detrend lightcurve for variability calculation." -> detrend = 0
		// nanstd(rel - detrend), ddof = 0
		double s = 0.0; int cn = 0;
		for (int g = tid; g < Ng; g += kThreads) {
			double d = gflux[g];
			if (have_fit) { const double u = (gtime_at(g) - mid) / half; d -= ((pc[3] * u + pc[2]) * u + pc[1]) * u + pc[0]; }
			srt[g] = d;
			if (!is_nan(d)) { s += d; cn++; }
		}
		const double tot = block_sum(s, red), cnt = block_sum((double)cn, red);
		const double avg = tot / cnt;
		double s2 = 0.0;
		for (int g = tid; g < Ng; g += kThreads) { const double d = srt[g]; if (!is_nan(d)) { const double e = d - avg; s2 += e * e; } }
		const double tot2 = block_sum(s2, red);
		const double sd = (cnt > 0.0) ? sqrt(tot2 / cnt) : nan;
		for (int g = tid; g < Ng; g += kThreads) srt[g] = gerr_at(g);   // the residuals in srt have been consumed (barriers of the sums above)
		const double med_err = block_median(srt, Ng, red, hist);
		variability = sd / med_err;
	}

	if (tid == 0) {
		o[0] = mean_flux; o[1] = variance; o[2] = rms_hour; o[3] = ptp; o[4] = cen_col; o[5] = cen_row;
		o[6] = variability; o[7] = mask_size; o[8] = edge_flux; o[9] = (double)flags;
	}
}

} // namespace

extern "C" int tp_lightcurve_diagnostics(tp_ctx* ctx, int32_t n_targets, int32_t n_cad,
	const double* d_flux, const double* d_flux_err, const double* d_centroid_col, const double* d_centroid_row, int64_t lc_pitch,
	const double* d_time, const int32_t* d_quality, int64_t quality_target_stride, uint32_t bitmask,
	const int32_t* d_status, const double* d_sumimage, const uint8_t* d_mask, int32_t height, int32_t width,
	double timescale_days, double* d_diag)
{
	TP_CHECK_CTX(ctx);
	TP_API_BEGIN
	TP_REQUIRE(ctx, n_targets >= 0 && n_cad > 0, "tp_lightcurve_diagnostics: bad sizes");
	TP_REQUIRE(ctx, d_flux && d_flux_err && d_centroid_col && d_centroid_row && d_time && d_quality && d_diag, "tp_lightcurve_diagnostics: null pointer");
	TP_REQUIRE(ctx, lc_pitch >= n_cad, "tp_lightcurve_diagnostics: lc_pitch < n_cad");
	TP_REQUIRE(ctx, quality_target_stride == 0 || quality_target_stride >= n_cad, "tp_lightcurve_diagnostics: bad quality stride");
	TP_REQUIRE(ctx, (d_mask == nullptr) == (d_sumimage == nullptr), "tp_lightcurve_diagnostics: mask and sum image go together");
	TP_REQUIRE(ctx, d_mask == nullptr || (height > 0 && width > 0 && (int64_t)height * width <= 2147483647ll), "tp_lightcurve_diagnostics: bad stamp geometry");
	TP_REQUIRE(ctx, timescale_days > 0, "tp_lightcurve_diagnostics: timescale must be positive");
	if (n_targets == 0) return TP_OK;
	// scratch length: the series, the time bins (capped) and the in-mask pixels of the stamp edge (at most its perimeter)
	int tp2 = (n_cad > 256) ? n_cad : 256;
	if (d_mask && 2 * (height + width) > tp2) tp2 = 2 * (height + width);
	const size_t small_bytes = kThreads * sizeof(double) + ((size_t)kThreads + 1 + 260 + 1) * sizeof(int);
	const size_t series_bytes = (((size_t)tp2 + (size_t)n_cad) * sizeof(double) + 2 * (size_t)n_cad * sizeof(int) + 15) & ~(size_t)15;
	size_t shmem = small_bytes + series_bytes + 16;
	unsigned char* gscratch = nullptr;
	if (shmem > 160 * 1024) { // long light curves: the series arrays move to HBM scratch, one slice per target
		gscratch = static_cast<unsigned char*>(tp_ctx_scratch(ctx, series_bytes * (size_t)n_targets));
		TP_REQUIRE(ctx, gscratch != nullptr, "tp_lightcurve_diagnostics: out of device memory for the series scratch");
		shmem = small_bytes + 16;
	}
	DiagArgs a;
	a.flux = d_flux; a.flux_err = d_flux_err; a.ccol = d_centroid_col; a.crow = d_centroid_row; a.lc_pitch = lc_pitch;
	a.time = d_time; a.quality = d_quality; a.quality_stride = quality_target_stride; a.bitmask = bitmask;
	a.status = d_status; a.sumimage = d_sumimage; a.mask = d_mask; a.height = height; a.width = width;
	a.n_cad = n_cad; a.tp2 = tp2; a.timescale = timescale_days; a.out = d_diag;
	a.gscratch = gscratch; a.gscratch_per_target = series_bytes;
	if (shmem > 64 * 1024)
		TP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(tp_diagnostics_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
	TP_LAUNCH(ctx, TPK_DIAGNOSTICS, tp_diagnostics_kernel, dim3((unsigned)n_targets), dim3(kThreads), shmem, a);
	TP_LAUNCH_CHECK(ctx, "tp_diagnostics_kernel");
	return TP_OK;
	TP_API_END(ctx)
}
