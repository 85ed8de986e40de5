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
// Mapping (gfx950): one 256-thread workgroup per target, the good-cadence series in LDS; every median is a bitonic
// sort in LDS (NaN -> +inf sentinels, element picked by rank), sums are fixed-shape tree reductions (deterministic).
// The cubic fit is done on the time axis mapped to [-1, 1] (the fitted polynomial is invariant under an affine change
// of variable, the normal equations then have a condition number of ~1e3 instead of ~1e12).
// Bytes: 4 series x T x 8 per target in, 80 B out -- negligible next to the cubes; latency-bound.
#include "common.h"
#include <cmath>

namespace {

constexpr int kThreads = 256;

struct DiagArgs {
	const double* flux; const double* flux_err; const double* ccol; const double* crow; int64_t lc_pitch;
	const double* time; const int32_t* quality; int64_t quality_stride; uint32_t bitmask;
	const int32_t* status; const double* sumimage; const uint8_t* mask; int height, width;
	int n_cad; int tp2; double timescale; double* out;
};

enum { F_ALLNAN_FLUX = 1, F_ALLNAN_ERR = 2, F_BAD_TIME = 4, F_NO_DETREND = 8, F_TOO_MANY_BINS = 16 };

__device__ __forceinline__ bool is_nan(double x) { return x != x; }
__device__ __forceinline__ bool is_finite(double x) { return fabs(x) <= 1.7976931348623157e308; }

// fixed-shape tree reductions over the workgroup (every thread gets the result)
__device__ double block_sum(double v, double* red) {
	const int tid = threadIdx.x;
	__syncthreads();
	red[tid] = v;
	__syncthreads();
	for (int s = kThreads / 2; s > 0; s >>= 1) { if (tid < s) red[tid] += red[tid + s]; __syncthreads(); }
	return red[0];
}
__device__ double block_min(double v, double* red) {
	const int tid = threadIdx.x;
	__syncthreads();
	red[tid] = v;
	__syncthreads();
	for (int s = kThreads / 2; s > 0; s >>= 1) { if (tid < s) { const double o = red[tid + s]; if (o < red[tid]) red[tid] = o; } __syncthreads(); }
	return red[0];
}
__device__ double block_max(double v, double* red) {
	const int tid = threadIdx.x;
	__syncthreads();
	red[tid] = v;
	__syncthreads();
	for (int s = kThreads / 2; s > 0; s >>= 1) { if (tid < s) { const double o = red[tid + s]; if (o > red[tid]) red[tid] = o; } __syncthreads(); }
	return red[0];
}

// nanmedian of srt[0..m): NaN entries must already be +inf; returns NaN when no value is left.
__device__ double block_median(double* srt, int m, double* red) {
	const int tid = threadIdx.x;
	__syncthreads();
	int p2 = 1;
	while (p2 < m) p2 <<= 1;
	int cnt = 0;
	for (int i = tid; i < p2; i += kThreads) {
		if (i >= m) srt[i] = __builtin_inf();
		else if (!is_nan(srt[i])) cnt++;
		if (i < m && is_nan(srt[i])) srt[i] = __builtin_inf();
	}
	const int n = (int)block_sum((double)cnt, red);
	for (int size = 2; size <= p2; size <<= 1) {
		for (int stride = size >> 1; stride > 0; stride >>= 1) {
			for (int t = tid; t < p2 / 2; t += kThreads) {
				const int lo = (t / stride) * (stride * 2) + (t % stride);
				const int hi = lo + stride;
				const bool up = ((lo & size) == 0);
				const double x = srt[lo], y = srt[hi];
				if ((x > y) == up) { srt[lo] = y; srt[hi] = x; }
			}
			__syncthreads();
		}
	}
	double med = __builtin_nan("");
	if (n > 0) med = (n & 1) ? srt[n >> 1] : (srt[(n >> 1) - 1] + srt[n >> 1]) / 2.0;
	__syncthreads();
	return med;
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

__global__ __launch_bounds__(kThreads) void tp_diagnostics_kernel(DiagArgs a)
{
	extern __shared__ __align__(16) double lds[];
	const int target = blockIdx.x;
	const int tid = threadIdx.x;
	const int T = a.n_cad, TP2 = a.tp2;
	double* srt = lds;                 // [TP2] sort buffer
	double* fb = srt + TP2;            // [TP2] binned flux
	double* gflux = fb + TP2;          // [T] relative flux of the good cadences
	double* gerr = gflux + T;          // [T]
	double* gtime = gerr + T;          // [T]
	double* red = gtime + T;           // [kThreads]
	int* gk = reinterpret_cast<int*>(red + kThreads); // [T] original index of the g-th good cadence
	int* ired = gk + T;                // [kThreads + 1]
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
		if (tid == 0) { // the selected pixels in raster order, NaN -> 0, numpy pairwise sum
			int n = 0;
			for (int p = 0; p < P; ++p) {
				const int r = p / W, cc = p - r * W;
				if (m[p] && (r == 0 || r == H - 1 || cc == 0 || cc == W - 1)) { const double v = S[p]; srt[n++] = is_nan(v) ? 0.0 : v; }
			}
			double tot;
			if (n <= 128) tot = pairwise_leaf(srt, n);
			else { int n2 = n / 2; n2 -= n2 % 8; tot = pairwise_leaf(srt, n2) + pairwise_leaf(srt + n2, n - n2); }
			red[0] = tot;
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
	const double mean_flux = block_median(srt, Ng, red);
	const double ainv = fabs(1.0 / mean_flux);
	for (int g = tid; g < Ng; g += kThreads) {
		const int k = gk[g];
		gflux[g] = (flux[k] / mean_flux) - 1.0;
		gerr[g] = ainv * ferr[k];
		gtime[g] = a.time[k];
	}
	__syncthreads();

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
			const double t = gtime[g];
			if (!is_nan(t)) { ct++; if (t < tmn) tmn = t; if (t > tmx) tmx = t; }
		}
		const double nfl = block_sum((double)cf, red), nt = block_sum((double)ct, red);
		const double tmin = block_min(tmn, red), tmax = block_max(tmx, red);
		if (Ng > 0 && nfl > 0.0) {
			if (nt == 0.0 || !is_finite(tmin) || !is_finite(tmax) || !(tmax - tmin > 0.0)) flags |= F_BAD_TIME;
			else {
				const double ts = a.timescale;
				const double nbd = ceil((tmax - tmin) / ts); // len(np.arange(tmin, tmax, ts))
				if (!(nbd <= (double)TP2)) flags |= F_TOO_MANY_BINS;
				else {
					const int nb = (int)nbd;
					// numpy's arange fills start + i*delta with delta = (start + step) - start, rounded: NOT i*step
					const double delta = (tmin + ts) - tmin;
					// bin of every finite sample: searchsorted(right) on the edges tmin + i*delta (i < nb), tmax; the last bin is closed
					int* bidx = reinterpret_cast<int*>(srt); // [T] ints in the sort buffer (TP2 doubles >= T ints)
					for (int g = tid; g < Ng; g += kThreads) {
						int b = -1;
						const double x = gtime[g];
						if (!is_nan(gflux[g]) && is_finite(gflux[g]) && !is_nan(x)) {
							b = (int)floor((x - tmin) / delta);
							if (b < 0) b = 0;
							if (b > nb - 1) b = nb - 1;
							while (b + 1 <= nb - 1 && (tmin + (double)(b + 1) * delta) <= x) ++b;
							while (b > 0 && (tmin + (double)b * delta) > x) --b;
						}
						bidx[g] = b;
					}
					__syncthreads();
					// nanmean per bin, samples added in time-series order (numpy's order inside binned_statistic)
					for (int b = tid; b < nb; b += kThreads) {
						double s = 0.0; int c = 0;
						for (int g = 0; g < Ng; ++g) if (bidx[g] == b) { s += gflux[g]; c++; }
						fb[b] = c ? (0.0 + s) / (double)c : nan;
					}
					__syncthreads();
					for (int b = tid; b < nb; b += kThreads) srt[b] = fb[b];
					const double med1 = block_median(srt, nb, red);
					for (int b = tid; b < nb; b += kThreads) srt[b] = fabs(fb[b] - med1);
					const double med2 = block_median(srt, nb, red);
					rms_hour = 1.482602218505602 * med2; // utilities.mad_to_sigma (:25)
				}
			}
		}
	}

	// ---- point-to-point scatter (:1366)
	for (int g = tid; g + 1 < Ng; g += kThreads) srt[g] = fabs(gflux[g + 1] - gflux[g]);
	const double ptp = block_median(srt, (Ng > 0) ? (Ng - 1) : 0, red);

	// ---- median centroid (:1369)
	for (int g = tid; g < Ng; g += kThreads) srt[g] = ccol[gk[g]];
	const double cen_col = block_median(srt, Ng, red);
	for (int g = tid; g < Ng; g += kThreads) srt[g] = crow[gk[g]];
	const double cen_row = block_median(srt, Ng, red);

	// ---- variability (:1372-1393): weighted cubic fit, standard deviation of the residuals / median error
	double variability;
	{
		int c = 0;
		double tmn = __builtin_inf(), tmx = -__builtin_inf();
		for (int g = tid; g < Ng; g += kThreads) {
			const bool ok = is_finite(gtime[g]) && is_finite(gflux[g]) && is_finite(gerr[g]);
			if (ok) { c++; const double t = gtime[g]; if (t < tmn) tmn = t; if (t > tmx) tmx = t; }
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
				const bool ok = is_finite(gtime[g]) && is_finite(gflux[g]) && is_finite(gerr[g]);
				if (!ok) continue;
				const double u = (gtime[g] - mid) / half, w = 1.0 / gerr[g], w2 = w * w, y = gflux[g];
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
			if (have_fit) { const double u = (gtime[g] - mid) / half; d -= ((pc[3] * u + pc[2]) * u + pc[1]) * u + pc[0]; }
			srt[g] = d;
			if (!is_nan(d)) { s += d; cn++; }
		}
		const double tot = block_sum(s, red), cnt = block_sum((double)cn, red);
		const double avg = tot / cnt;
		double s2 = 0.0;
		for (int g = tid; g < Ng; g += kThreads) { const double d = srt[g]; if (!is_nan(d)) { const double e = d - avg; s2 += e * e; } }
		const double tot2 = block_sum(s2, red);
		const double sd = (cnt > 0.0) ? sqrt(tot2 / cnt) : nan;
		for (int g = tid; g < Ng; g += kThreads) srt[g] = gerr[g];
		const double med_err = block_median(srt, Ng, red);
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
	TP_REQUIRE(ctx, d_mask == nullptr || (height > 0 && width > 0 && 2 * (height + width) <= 256), "tp_lightcurve_diagnostics: bad stamp geometry");
	TP_REQUIRE(ctx, timescale_days > 0, "tp_lightcurve_diagnostics: timescale must be positive");
	if (n_targets == 0) return TP_OK;
	int tp2 = 256;
	while (tp2 < n_cad) tp2 <<= 1;
	const size_t shmem = ((size_t)2 * tp2 + 3 * (size_t)n_cad + kThreads) * sizeof(double) + ((size_t)n_cad + kThreads + 1) * sizeof(int) + 16;
	TP_REQUIRE(ctx, shmem <= 160 * 1024, "tp_lightcurve_diagnostics: light curve too long for the LDS-resident reductions (about 4000 cadences)");
	DiagArgs a;
	a.flux = d_flux; a.flux_err = d_flux_err; a.ccol = d_centroid_col; a.crow = d_centroid_row; a.lc_pitch = lc_pitch;
	a.time = d_time; a.quality = d_quality; a.quality_stride = quality_target_stride; a.bitmask = bitmask;
	a.status = d_status; a.sumimage = d_sumimage; a.mask = d_mask; a.height = height; a.width = width;
	a.n_cad = n_cad; a.tp2 = tp2; a.timescale = timescale_days; a.out = d_diag;
	if (shmem > 64 * 1024)
		TP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(tp_diagnostics_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
	TP_LAUNCH(ctx, TPK_DIAGNOSTICS, tp_diagnostics_kernel, dim3((unsigned)n_targets), dim3(kThreads), shmem, a);
	TP_LAUNCH_CHECK(ctx, "tp_diagnostics_kernel");
	return TP_OK;
	TP_API_END(ctx)
}
