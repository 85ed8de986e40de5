// k2p2_core.h -- A2..A5b + A7: K2P2 pixel-mask creation for ONE target by ONE wavefront.
//
// Replaces k2p2FixFromSum and its helpers (photometry/AperturePhotometry/k2p2v2.py:63-86, 89-288,
// 291-341, 344-623) plus the mask selection / minimum aperture / contamination logic of
// AperturePhotometry.do_photometry (photometry/AperturePhotometry/photometry.py:31-41, 93-131,
// 220-254).  Third-party algorithms are restated from their published form:
//   statsmodels 0.13.2  bw_scott/_select_sigma, kdensityfft (linear binning + Silverman transform),
//                       KDEUnivariate.evaluate            (k2p2v2.py:410-420)
//   scipy 1.7.3         stats.trim1, optimize bracket/Brent/_minimize_powell (k2p2v2.py:402,421),
//                       ndimage.gaussian_filter(sigma=.5) (5 taps, reflect), ndimage.label
//   scikit-learn 1.0.2  DBSCAN(eps=sqrt2+eps, min_samples=4) on a pixel grid
//   scikit-image 0.19.2 peak_local_max, watershed (priority flood, label at push)
//
// Execution model.  The code is written in SPMD "phase" style: all cross-lane communication goes
// through the per-target shared arrays (LDS), phases are separated by TP_SYNC(), parallel loops
// use TP_PAR_FOR (independent iterations), inherently sequential parts (priority-flood watershed)
// run under TP_SERIAL on lane 0, and scalar control flow (Brent/Powell iterations) is executed
// redundantly and identically by all 64 lanes.  Reductions are "per-lane partials, then every lane
// sums the 64 partials in lane order" so that the arithmetic is order-deterministic.
//
// The CPU unit tests compile the same phases with loops over the lanes (tests/hostsim) to check the logic where no GPU
// is available; the product never runs that build.
// Built with -ffp-contract=off on both sides; exp() is an own implementation, so host-sim and
// device results are bit-identical.
#pragma once
#include <stdint.h>
#include <math.h>

// The lane layer (how a "phase" runs over the 64 lanes, the reductions) comes from a separate header: k2p2_lanes.h for the
// device; the CPU unit tests of the logic substitute their own (tests/hostsim), the product never does.
#ifndef K2P2_LANES_HEADER
#define K2P2_LANES_HEADER "k2p2_lanes.h"
#endif
#define K2P2_LANES_SECTION 1
#include K2P2_LANES_HEADER
#undef K2P2_LANES_SECTION
#ifndef TP_NO_UNROLL
#define TP_NO_UNROLL
#endif
#ifndef TP_ALWAYS_INLINE
#define TP_ALWAYS_INLINE inline
#endif

// lab hook: a scratch build (-DTP_LAB_K2P2_CLOCK, k2p2.hip) accumulates the cycles between consecutive hooks per phase; nothing in the product
#ifndef TP_K2P2_CLOCK
#define TP_K2P2_CLOCK(k, i)
#endif

namespace k2p2 {

constexpr int kGrid = 128;          // KDE FFT grid: gridsize=100 -> next power of two (kde.py kdensityfft)
static_assert(kGrid == 128, "the bit reversal of the KDE transform is written for seven bits");
constexpr double kMadToSigma = 1.482602218505602;   // photometry/utilities.py:25
constexpr double kPi = 3.141592653589793;

// status / flag encodings (per target)
enum : int32_t {
	FLAG_MIN_APERTURE = 1,      // photometry.py:99-112 (status WARNING)
	FLAG_EDGE_DOWN = 2, FLAG_EDGE_UP = 4, FLAG_EDGE_LEFT = 8, FLAG_EDGE_RIGHT = 16,   // photometry.py:123-131
	FLAG_NOSTARS = 32,          // K2P2NoStars (k2p2v2.py:454-455)
	FLAG_NOMASKS = 64,          // k2p2FixFromSum returned None (k2p2v2.py:530-531)
	ERR_SHIFT = 8,
	ERR_NOFLUX = 1,             // K2P2NoFlux, uncaught by the plugin (k2p2v2.py:398-399)
	ERR_BANDWIDTH_ZERO = 2,     // statsmodels RuntimeError "Selected KDE bandwidth is 0"
	ERR_NO_PEAKS = 3,           // np.argmin of an empty distance array (k2p2v2.py:146)
	ERR_TARGET_OUTSIDE = 4,     // IndexError at photometry.py:107
	ERR_TOO_MANY_MASKS = 5,     // photometry.py:114-116
	ERR_NO_TARGETS_IN_MASK = 6, // photometry.py:227-230
};

struct Params {
	double thresh;                 // 0.8   (photometry.py:55)
	int32_t min_no_pixels_in_mask; // 4
	int32_t min_for_cluster;       // 4
	int32_t extend_overflow;       // 1
	int32_t reserved;
	double ws_thres;               // 0
	double saturation_limit;       // 7.0 (k2p2v2.py:49)
	double gauss_w0, gauss_w1, gauss_w2; // ndimage.gaussian_filter(sigma=0.5) normalised taps
};

// Per-target inputs / outputs (global memory)
struct Target {
	const double* S;          // sum image [H*W]
	int H, W;
	int ncat;
	const float* cat_col;     // column_stamp
	const float* cat_row;     // row_stamp
	const float* cat_tmag;
	const float* cat_ccd_col; // column (CCD)
	const float* cat_ccd_row; // row (CCD)
	const int64_t* cat_starid;
	double tpos_row, tpos_col;   // target CCD position (target_pos_row/column)
	int stamp_row0, stamp_col0;  // stamp[0], stamp[2]
	double target_tmag;
	int64_t target_starid;
	const int32_t* aperture;     // [H*W]
	const double* cut_override;  // optional: replace CUT (tests of the integer pipeline)
	// outputs
	uint8_t* mask;            // [H*W] final_phot_mask
	int32_t* status;          // STATUS integer
	int32_t* flags;
	double* contamination;
	double* diag;             // [8]: CUT, MODE, MAD1, bandwidth, max_guess, nflux, margin, nmasks
	uint8_t* cat_in_mask;     // [ncat] 1 if the catalog star falls in the final mask (skip_targets source)
};

// Labels, pixel indices and counters bounded by P (<= 54*54, the LDS limit) are 16-bit: LDS footprint decides how many targets
// (wavefronts) a CU holds at once.
typedef int16_t lab_t;

// Shared (LDS) work arrays of one target.  Sizes in elements; P = H*W, Pp = pow2 >= P.
struct Shared {
#ifdef TP_LAB_K2P2_CLOCK
	unsigned long long clk0, clk[16];
#endif
	int lane;
	int P, Pp, H, W;
	uint32_t wmagic;  // ceil(2^32 / W): row of a pixel index by one multiplication (exact for indices and widths below 2^16)
	double* S;        // [P]
	double* srt;      // [Pp] sorted positive fluxes (+inf padding)
	double* Z;        // [P]
	double* dist;     // [P]
	double* tmp;      // [P]
	double* hval;     // [max(64, P/2 + 1)] per-lane scratch / the watershed's rank -> pixel table (int32)
	double* red;      // [64]
	double* grid;     // [kGrid + 132]: real parts [kGrid] (the binned counts, in the end the density), imaginary parts [kGrid] of the KDE's transforms
	lab_t* lab;       // [P] DBSCAN labels
	lab_t* lab2;      // [P] labels after watershed
	lab_t* mark;      // [P] markers / component labels
	lab_t* wsout;     // [P]
	lab_t* hage;      // [P]
	lab_t* hpix;      // [P]
	int32_t* ired;    // [64]
	int32_t* scal;    // [32] uniform scalars
	uint8_t* idx;     // [P]
	uint8_t* core;    // [P]
	uint8_t* lmax;    // [P]
	uint8_t* msk;     // [P] current mask
	uint8_t* sat;     // [P] saturated additions for the current mask
	uint8_t* res;     // [P] result mask
	const double* twid; // [2*kGrid] cos, sin of 2*pi*j/kGrid (global / constant)
	double* twl;      // [2*kGrid] LDS copy of twid, valid during the threshold phase
};

// LDS plan.  The arrays of the threshold phase (A2: sorted fluxes + the KDE grids) are dead once CUT is known and the
// arrays of the clustering / watershed / assembly phases (A3..A5) are not touched before, so the two sets share one
// region; with 16-bit labels and a half-size scratch that is 13.1 KB instead of 22.9 KB per 15x15 target,
// i.e. 12 instead of 7 resident wavefronts per CU.
struct SharedLayout {
	int Pa, Pp, Pp_sort, hval_len;
	size_t off_region, region_bytes, off_ints, off_bytes, total;
};

inline TP_HD SharedLayout shared_layout(int P) {
	SharedLayout L;
	L.Pa = (P < 64) ? 64 : P; // several P-sized arrays double as 64-entry per-lane scratch
	L.Pp = 1;
	while (L.Pp < L.Pa) L.Pp <<= 1;
	L.Pp_sort = 1;
	while (L.Pp_sort < P) L.Pp_sort <<= 1;
	const size_t Pa = (size_t)L.Pa;
	L.off_region = (2 * Pa + 64) * 8;                                   // S, tmp, red
	const size_t phase1 = ((size_t)L.Pp + 3 * kGrid + 132) * 8;        // srt, grid (real and imaginary parts of the KDE's in-place transforms), twiddle copy [2*kGrid]
	L.hval_len = (Pa / 2 + 1 > 64) ? (Pa / 2 + 1) : 64;
	const size_t phase2 = (2 * Pa + (size_t)L.hval_len) * 8 + 4 * Pa * sizeof(lab_t); // Z, dist, hval | lab, lab2, mark, wsout
	L.region_bytes = ((phase1 > phase2 ? phase1 : phase2) + 15) & ~(size_t)15;
	L.off_ints = L.off_region + L.region_bytes;                         // hage, hpix, ired, scal
	L.off_bytes = L.off_ints + (2 * ((Pa + 1) & ~(size_t)1) * sizeof(lab_t)) + (64 + 32) * 4; // idx, core, lmax, msk, sat, res
	L.total = L.off_bytes + ((6 * Pa + 15) & ~(size_t)15) + 64;
	return L;
}

inline TP_HD size_t shared_bytes(int P) { return shared_layout(P).total; }

inline TP_DEV void shared_carve(Shared& k, void* base, int H, int W, int lane, const double* twid) {
	const int P = H * W;
	const SharedLayout L = shared_layout(P);
	const int Pa = L.Pa;
	k.lane = lane; k.P = P; k.Pp = L.Pp_sort; k.H = H; k.W = W; k.twid = twid;
	k.wmagic = (W > 1) ? (uint32_t)((0x100000000ull + (uint64_t)W - 1) / (uint64_t)W) : 0u;
	unsigned char* b0 = (unsigned char*)base;
	double* d = (double*)b0;
	k.S = d; d += Pa;
	k.tmp = d; d += Pa;
	k.red = d; d += 64;
	// phase 1 view of the region
	double* r1 = (double*)(b0 + L.off_region);
	k.srt = r1; r1 += L.Pp;
	k.grid = r1; r1 += kGrid + 132;
	k.twl = r1;
	// phase 2 view of the region
	double* r2 = (double*)(b0 + L.off_region);
	k.Z = r2; r2 += Pa;
	k.dist = r2; r2 += Pa;
	k.hval = r2; r2 += L.hval_len;
	lab_t* ri = (lab_t*)r2;
	k.lab = ri; ri += Pa;
	k.lab2 = ri; ri += Pa;
	k.mark = ri; ri += Pa;
	k.wsout = ri;
	// ired / scal first: 4-byte aligned whatever Pa is; hage doubles as a uint32 bit set (watershed) and follows them
	int32_t* i = (int32_t*)(b0 + L.off_ints);
	k.ired = i; i += 64;
	k.scal = i; i += 32;
	lab_t* hi = (lab_t*)i;
	k.hage = hi; hi += (Pa + 1) & ~1;
	k.hpix = hi;
	uint8_t* b = b0 + L.off_bytes;
	k.idx = b; b += Pa;
	k.core = b; b += Pa;
	k.lmax = b; b += Pa;
	k.msk = b; b += Pa;
	k.sat = b; b += Pa;
	k.res = b;
}

//--------------------------------------------------------------------------------------------------
// deterministic math
//--------------------------------------------------------------------------------------------------
inline TP_DEV double tp_inf() { return __builtin_inf(); }
inline TP_DEV double tp_nan() { return __builtin_nan(""); }
inline TP_DEV bool tp_isnan(double x) { return x != x; }
// p / W for 0 <= p < 2^16 (stamps hold at most 65 535 pixels): floor(p / W) = (p * ceil(2^32 / W)) >> 32 exactly, because the
// error term p * (ceil(2^32 / W) * W - 2^32) stays below 2^32.  (A division by a run-time W is ~30 instructions, and every phase
// of the builder converts pixel indices to rows and columns.)
inline TP_DEV int row_of(const Shared& k, int p) { return (k.W > 1) ? (int)(((uint64_t)(uint32_t)p * (uint64_t)k.wmagic) >> 32) : p; }

// A WINDOW of the stamp: rows r0 .. r0 + h - 1, columns c0 .. c0 + w - 1.  The per-cluster passes of A4 (blur, peaks, marker labels,
// the saturated-column test) touch a cluster's neighbourhood only: everywhere else their results are constants (zero) that a
// plain store pass writes.  A pass over a window computes, for its pixels, exactly what the pass over the whole stamp computed:
// the pixel index, its neighbours and the reflections at the STAMP's edges are those of the stamp.
struct Win { int r0, c0, h, w, n; uint32_t magic; bool full; };
inline TP_DEV Win win_make(const Shared& k, int r0, int r1, int c0, int c1) {   // inclusive limits, clipped to the stamp
	Win v;
	if (r0 < 0) r0 = 0;
	if (c0 < 0) c0 = 0;
	if (r1 > k.H - 1) r1 = k.H - 1;
	if (c1 > k.W - 1) c1 = k.W - 1;
	v.r0 = r0; v.c0 = c0; v.h = (r1 >= r0) ? (r1 - r0 + 1) : 0; v.w = (c1 >= c0) ? (c1 - c0 + 1) : 0;
	v.n = v.h * v.w;
	v.full = (v.n == k.P);
	// ceil(2^32 / w): exact below 2^16 (the whole stamp needs none: its q-th pixel is pixel q)
	v.magic = (v.w > 1 && !v.full) ? (uint32_t)((0x100000000ull + (uint64_t)v.w - 1ull) / (uint64_t)v.w) : 0u;
	return v;
}
inline TP_DEV Win win_full(const Shared& k) { return win_make(k, 0, k.H - 1, 0, k.W - 1); }
inline TP_DEV bool win_is_full(const Shared& k, const Win& v) { (void)k; return v.full; }
// stamp pixel index of the q-th pixel of the window (row-major)
inline TP_DEV int win_pix(const Shared& k, const Win& v, int q) {
	if (v.full) return q;      // (uniform: the passes over the whole stamp pay one scalar branch, not the index arithmetic)
	const int rr = (v.w > 1) ? (int)(((uint64_t)(uint32_t)q * (uint64_t)v.magic) >> 32) : q;
	return (v.r0 + rr) * k.W + v.c0 + (q - rr * v.w);
}

inline TP_DEV double tp_pow2(int e) { // 2^e for -1022 <= e <= 1023
	union { uint64_t u; double d; } c;
	c.u = (uint64_t)(e + 1023) << 52;
	return c.d;
}

// exp(x): range reduction x = k ln2 + r, 13-term Taylor in Horner form (|r| <= 0.3466 -> < 1 ulp),
// plain IEEE mul/add only so that host-sim and device agree bit for bit.
inline TP_DEV double tp_exp(double x) {
	if (x != x) return x;
	if (x > 709.782712893384) return tp_inf();
	if (x < -745.2) return 0.0;
	const double LOG2E = 1.4426950408889634074;
	const double LN2_HI = 6.93147180369123816490e-01;
	const double LN2_LO = 1.90821492927058770002e-10;
	double kf = __builtin_floor(x * LOG2E + 0.5);
	int ki = (int)kf;
	double r = (x - kf * LN2_HI) - kf * LN2_LO;
	double p = 1.0 / 6227020800.0;            // 1/13!
	p = p * r + 1.0 / 479001600.0;            // 1/12!
	p = p * r + 1.0 / 39916800.0;
	p = p * r + 1.0 / 3628800.0;
	p = p * r + 1.0 / 362880.0;
	p = p * r + 1.0 / 40320.0;
	p = p * r + 1.0 / 5040.0;
	p = p * r + 1.0 / 720.0;
	p = p * r + 1.0 / 120.0;
	p = p * r + 1.0 / 24.0;
	p = p * r + 1.0 / 6.0;
	p = p * r + 0.5;
	p = p * r + 1.0;
	p = p * r + 1.0;
	int k1 = ki / 2, k2 = ki - k1;
	return (p * tp_pow2(k1)) * tp_pow2(k2);
}

// Reductions over the 64 per-lane partials in k.red / k.ired (written by a TP_LANE_LOOP, followed by
// TP_SYNC).  Fixed binary-tree association: a[l] (op)= a[l+32], then +16, ... so that the host
// simulation and the device (DPP / permute shuffles, result broadcast from lane 0) agree bit for bit.
#define K2P2_LANES_SECTION 2
#include K2P2_LANES_HEADER
#undef K2P2_LANES_SECTION

//--------------------------------------------------------------------------------------------------
// A2: threshold
//--------------------------------------------------------------------------------------------------
// Bitonic sort of k.srt[0..Pp) ascending (NaN-free input; +inf padding).
inline TP_DEV void bitonic_sort(Shared& k) {
#ifdef TP_HAVE_WAVE_SORT
	if (k.Pp <= 256) { wave_sort_regs<4>(k); return; }     // the device's lane layer sorts in registers: four keys per lane,
	if (k.Pp <= 1024) { wave_sort_regs<16>(k); return; }   // sixteen for stamps up to 32 x 32
#endif
	const int n = k.Pp;
	for (int size = 2; size <= n; size <<= 1) {
		for (int stride = size >> 1; stride > 0; stride >>= 1) {
			TP_PAR_FOR(t, n >> 1) {
				const int lo = ((t & ~(stride - 1)) << 1) | (t & (stride - 1)); // stride is a power of two: no division
				const int hi = lo + stride;
				const bool up = ((lo & size) == 0);
				const double a = k.srt[lo], b = k.srt[hi];
				if ((a > b) == up) { k.srt[lo] = b; k.srt[hi] = a; }
			}
			TP_SYNC();
		}
	}
}

// scipy.stats.scoreatpercentile(x, per) on sorted data (interpolation 'fraction')
inline TP_DEV double score_at_percentile(const double* sorted, int n, double per) {
	const double idxf = per / 100.0 * (double)(n - 1);
	const int i = (int)idxf;
	if ((double)i == idxf) return sorted[i];
	const double w0 = (double)(i + 1) - idxf, w1 = idxf - (double)i;
	const double sumval = w0 + w1;
	return (sorted[i] * w0 + sorted[i + 1] * w1) / sumval;
}


// -KDE(x): direct Gaussian sum over the nc values k.srt[0..nc) with bandwidth h
// (statsmodels kernels.Gaussian: 0.3989422804014327*exp(-z**2/2); density = 1/(h n) * sum)
inline TP_DEV double neg_kde(Shared& k, int nc, double h, double x) {
	const double* srt = k.srt;
	const double tot = wave_sum_f(k, [=](int l) {
		double s = 0.0;
		for (int i = l; i < nc; i += 64) {
			const double z = (srt[i] - x) / h;
			s += 0.3989422804014327 * tp_exp(-(z * z) / 2.0);
		}
		return s;
	});
	return -1.0 * ((1.0 / (h * (double)nc)) * tot);
}

struct KdeFn {
	Shared* k; int nc; double h; double p; double xi; int* ncalls;
	inline TP_DEV double operator()(double alpha) const { ++(*ncalls); return neg_kde(*k, nc, h, p + alpha * xi); }
};

// scipy.optimize.bracket(func, xa=0, xb=1) -- optimize.py (mnbrak)
template <class F>
inline TP_DEV void sp_bracket(const F& func, double& xa, double& xb, double& xc, double& fa, double& fb, double& fc) {
	const double gold = 1.618034, verysmall = 1e-21, grow_limit = 110.0;
	xa = 0.0; xb = 1.0;
	fa = func(xa);
	fb = func(xb);
	if (fa < fb) { double t = xa; xa = xb; xb = t; t = fa; fa = fb; fb = t; }
	xc = xb + gold * (xb - xa);
	fc = func(xc);
	int iter = 0;
	while (fc < fb) {
		const double tmp1 = (xb - xa) * (fb - fc);
		const double tmp2 = (xb - xc) * (fb - fa);
		const double val = tmp2 - tmp1;
		double denom;
		if (fabs(val) < verysmall) denom = 2.0 * verysmall;
		else denom = 2.0 * val;
		double w = xb - ((xb - xc) * tmp2 - (xb - xa) * tmp1) / denom;
		const double wlim = xb + grow_limit * (xc - xb);
		if (iter > 1000) break; // scipy raises RuntimeError here
		iter += 1;
		double fw;
		if ((w - xc) * (xb - w) > 0.0) {
			fw = func(w);
			if (fw < fc) { xa = xb; xb = w; fa = fb; fb = fw; break; }
			else if (fw > fb) { xc = w; fc = fw; break; }
			w = xc + gold * (xc - xb);
			fw = func(w);
		} else if ((w - wlim) * (wlim - xc) >= 0.0) {
			w = wlim;
			fw = func(w);
		} else if ((w - wlim) * (xc - w) > 0.0) {
			fw = func(w);
			if (fw < fc) {
				xb = xc; xc = w; w = xc + gold * (xc - xb);
				fb = fc; fc = fw; fw = func(w);
			}
		} else {
			w = xc + gold * (xc - xb);
			fw = func(w);
		}
		xa = xb; xb = xc; xc = w;
		fa = fb; fb = fc; fc = fw;
	}
}

// scipy.optimize.Brent(func, tol).optimize() with brack=None (scipy 1.7.3: no bracket validation)
template <class F>
inline TP_DEV void sp_brent(const F& func, double tol, double& xmin, double& fval) {
	const double mintol = 1.0e-11, cg = 0.3819660;
	double xa, xb, xc, fa, fb, fc;
	sp_bracket(func, xa, xb, xc, fa, fb, fc);
	double x = xb, w = xb, v = xb;
	double fw = fb, fv = fb, fx = fb;
	double a, b;
	if (xa < xc) { a = xa; b = xc; } else { a = xc; b = xa; }
	double deltax = 0.0, rat = 0.0;
	int iter = 0;
	while (iter < 500) {
		const double tol1 = tol * fabs(x) + mintol;
		const double tol2 = 2.0 * tol1;
		const double xmid = 0.5 * (a + b);
		if (fabs(x - xmid) < (tol2 - 0.5 * (b - a))) break;
		if (fabs(deltax) <= tol1) {
			if (x >= xmid) deltax = a - x; else deltax = b - x;
			rat = cg * deltax;
		} else {
			double tmp1 = (x - w) * (fx - fv);
			double tmp2 = (x - v) * (fx - fw);
			double p = (x - v) * tmp2 - (x - w) * tmp1;
			tmp2 = 2.0 * (tmp2 - tmp1);
			if (tmp2 > 0.0) p = -p;
			tmp2 = fabs(tmp2);
			const double dx_temp = deltax;
			deltax = rat;
			if ((p > tmp2 * (a - x)) && (p < tmp2 * (b - x)) && (fabs(p) < fabs(0.5 * tmp2 * dx_temp))) {
				rat = p * 1.0 / tmp2;
				const double u = x + rat;
				if ((u - a) < tol2 || (b - u) < tol2) {
					if (xmid - x >= 0) rat = tol1; else rat = -tol1;
				}
			} else {
				if (x >= xmid) deltax = a - x; else deltax = b - x;
				rat = cg * deltax;
			}
		}
		double u;
		if (fabs(rat) < tol1) {
			if (rat >= 0) u = x + tol1; else u = x - tol1;
		} else {
			u = x + rat;
		}
		const double fu = func(u);
		if (fu > fx) {
			if (u < x) a = u; else b = u;
			if ((fu <= fw) || (w == x)) { v = w; w = u; fv = fw; fw = fu; }
			else if ((fu <= fv) || (v == x) || (v == w)) { v = u; fv = fu; }
		} else {
			if (u >= x) a = x; else b = x;
			v = w; w = x; x = u;
			fv = fw; fw = fx; fx = fu;
		}
		iter += 1;
	}
	xmin = x;
	fval = fx;
}

// scipy.optimize.minimize(f, x0, method='Powell').x for one parameter (xtol = ftol = 1e-4)
inline TP_DEV double powell_mode(Shared& k, int nc, double h, double x0) {
	const double xtol = 1e-4, ftol = 1e-4;
	int ncalls = 0;
	double x = x0;
	double direc = 1.0;
	double fval = neg_kde(k, nc, h, x); ncalls++;
	double x1 = x;
	int iter = 0;
	while (true) {
		const double fx = fval;
		double delta = 0.0;
		double direc1 = direc;
		double fx2 = fval;
		if (direc1 != 0.0) { // _linesearch_powell
			KdeFn fn{&k, nc, h, x, direc1, &ncalls};
			double alpha_min, fret;
			sp_brent(fn, xtol * 100, alpha_min, fret);
			direc1 = alpha_min * direc1;
			x = x + direc1;
			fval = fret;
		}
		if ((fx2 - fval) > delta) delta = fx2 - fval;
		iter += 1;
		const double bnd = ftol * (fabs(fx) + fabs(fval)) + 1e-20;
		if (2.0 * (fx - fval) <= bnd) break;
		if (ncalls >= 1000) break;
		if (iter >= 1000) break;
		if (tp_isnan(fx) && tp_isnan(fval)) break;
		direc1 = x - x1;
		x1 = x;
		const double x2 = x + 1 * direc1;
		fx2 = neg_kde(k, nc, h, x2); ncalls++;
		if (fx > fx2) {
			double t = 2.0 * (fx + fx2 - 2.0 * fval);
			double temp = (fx - fval - delta);
			t *= temp * temp;
			temp = fx - fx2;
			t -= delta * temp * temp;
			if (t < 0.0) {
				if (direc1 != 0.0) {
					KdeFn fn{&k, nc, h, x, direc1, &ncalls};
					double alpha_min, fret;
					sp_brent(fn, xtol * 100, alpha_min, fret);
					direc1 = alpha_min * direc1;
					x = x + direc1;
					fval = fret;
				}
				if (direc1 != 0.0) direc = direc1;
			}
		}
	}
	return x;
}

// Returns 0 ok, or an ERR_* code; fills diag[0..5] and leaves CUT in *cut (may be NaN).
inline TP_DEV int threshold(Shared& k, const Params& prm, const Target& t, double* cut) {
	const int P = k.P;

	// Flux = S[~isnan(S)]; Flux = Flux[Flux > 0]   (k2p2v2.py:394-395) -> compacted in raster order
	TP_LANE_LOOP(l) {
		int c = 0;
		for (int p = l; p < P; p += 64) c += (k.S[p] > 0.0) ? 1 : 0;
		k.ired[l] = c;
	}
	TP_SYNC();
	const int nflux = sum_ired(k);
	TP_SYNC();
	if (nflux == 0) return ERR_NOFLUX;
	// fill the sort buffer (order is irrelevant before sorting)
	TP_PAR_FOR(p, k.Pp) k.srt[p] = (p < P && k.S[p] > 0.0) ? k.S[p] : tp_inf();
	TP_SYNC();
	bitonic_sort(k);
	TP_K2P2_CLOCK(k, 12);
	// count of finite entries == nflux unless some flux is +inf (kept, as numpy would)
	// trim1(sorted, 0.15): keep the n - int(0.15 n) smallest (scipy/stats trim1, tail='right')
	int nc = nflux - (int)(0.15 * (double)nflux);
	// flux_cut = flux_cut[flux_cut < 70000]  (k2p2v2.py:407): a prefix of the sorted array
	TP_LANE_LOOP(l) {
		int c = 0;
		for (int i = l; i < nc; i += 64) c += (k.srt[i] < 70000.0) ? 1 : 0;
		k.ired[l] = c;
	}
	TP_SYNC();
	nc = sum_ired(k);
	TP_SYNC();
	if (t.diag) { TP_SERIAL { t.diag[5] = (double)nflux; } }

	// --- bw_scott: 1.059 * min(std(ddof=1), IQR/1.349) * n^-0.2   (bandwidths.py)
	double bw;
	{
		TP_LANE_LOOP(l) { double s = 0.0; for (int i = l; i < nc; i += 64) s += k.srt[i]; k.red[l] = s; }
		TP_SYNC();
		const double mean = sum_red(k) / (double)nc;
		TP_SYNC();
		TP_LANE_LOOP(l) { double s = 0.0; for (int i = l; i < nc; i += 64) { const double d = k.srt[i] - mean; s += d * d; } k.red[l] = s; }
		TP_SYNC();
		const double var = sum_red(k) / (double)(nc - 1); // nc == 1 -> 0/0 = NaN like numpy
		TP_SYNC();
		const double std_dev = sqrt(var);
		double A = std_dev;
		if (nc > 0) {
			const double IQR = (score_at_percentile(k.srt, nc, 75.0) - score_at_percentile(k.srt, nc, 25.0)) / 1.349;
			if (IQR > 0) A = (std_dev < IQR || tp_isnan(std_dev)) ? std_dev : IQR; // np.minimum propagates NaN
		}
		// n ** (-0.2) via exp/log would not be bit-stable; use pow from the toolchain only here
		bw = 1.059 * A * pow((double)nc, -0.2);
	}
	if (nc == 0) bw = tp_nan();
	TP_K2P2_CLOCK(k, 13);
	if (bw == 0.0) return ERR_BANDWIDTH_ZERO;

	// --- kdensityfft: linear binning on a 128-point grid, Silverman transform, inverse transform
	double max_guess;
	{
		const int M = kGrid;
		const double a = k.srt[0] - 3.0 * bw;
		const double b = ((nc > 0) ? k.srt[nc - 1] : tp_nan()) + 3.0 * bw;
		const double delta = (b - a) / (double)(M - 1);   // np.linspace retstep
		const double RANGE = b - a;
		double* binned = k.grid; double* dens = k.grid; double* fre = k.grid; double* fim = k.grid + M; // the transforms work in place: dens overwrites binned
		// the DFT twiddles next to the grids (LDS) for the two transforms below
		TP_PAR_FOR(j, 2 * M) k.twl[j] = k.twid[j];
		// fast_linbin: bin m accumulates, in data order, (1 - rem) from points with li == m and rem from li == m-1.
		// (li, rem) of every point once (k.hage / k.tmp), then each lane owns two bins
		TP_PAR_FOR(i, nc) {
			const double lxi = (k.srt[i] - a) / delta;
			int li = (int)lxi;
			if (!(lxi > -1.0e9 && lxi < 1.0e9)) li = -1; // NaN / absurd: the point is dropped like in the reference (li > 1 fails)
			k.hage[i] = li;
			k.tmp[i] = lxi - (double)li;
		}
		TP_SYNC();
		// The points are sorted, so li never decreases: the contributors of bin m (li == m-1 adding rem, then li == m adding
		// 1 - rem, in data order like the reference's loop) are two adjacent runs.  A lane owns bins 2l and 2l+1; the four
		// run boundaries come from four interleaved binary searches, the sums are plain loops over known ranges (no
		// data-dependent exit: the LDS reads pipeline).  Most points sit in a few bins (the sky level): without this a
		// handful of lanes walked 100+ dependent iterations while the others idled.
		TP_LANE_LOOP(l) {
			int lo[4] = {0, 0, 0, 0}, hi[4] = {nc, nc, nc, nc};
			// (enough halvings for nc samples: nine sufficed for the 256 pixels of a 16 x 16 stamp and were what rounds 1-5 made for
			// EVERY stamp -- above 512 samples the run boundaries came out up to two short, the binned counts and with them the
			// KDE's argmax a grid step off on large stamps; the Powell search that starts there usually ended in the same mode,
			// which is how it went unseen.  Found and fixed in round 6: tests/test_k2p2_hostsim.py::test_hostsim_kde_argmax_on_large_stamps)
			int nsteps = 1;
			while ((1 << nsteps) <= nc) ++nsteps;
			for (int step = 0; step <= nsteps; ++step) {
				for (int e = 0; e < 4; ++e) {
					if (lo[e] < hi[e]) {
						const int mid = (lo[e] + hi[e]) >> 1;
						if (k.hage[mid] < 2 * l - 1 + e) lo[e] = mid + 1; else hi[e] = mid;
					}
				}
			}
			for (int e = 0; e < 2; ++e) {
				const int m = 2 * l + e;
				double g = 0.0;
				if (m - 1 > 1 && m - 1 < M) for (int i = lo[e]; i < lo[e + 1]; ++i) g = g + k.tmp[i];
				if (m > 1 && m < M) for (int i = lo[e + 1]; i < lo[e + 2]; ++i) g = g + 1 - k.tmp[i];
				binned[m] = g / (delta * (double)nc);
			}
		}
		TP_SYNC();
		// zstar = silverman_transform * forrt(binned), density = revrt(zstar): statsmodels' kdensityfft takes numpy's rfft / irfft.
		// Here: a radix-2 complex FFT of the M real counts in place -- decimation in frequency forward (natural order in,
		// bit-reversed out), the Silverman factors applied where the coefficients lie, decimation in time back (bit-reversed in,
		// natural order out): no permutation pass, 2 x 7 stages of 64 butterflies, one per lane.  (The two O(M^2) sums this
		// replaces were a sixth of the mask builder's time, tools/k2p2_timing.py.)  Only the argmax of the density is read
		// (k2p2v2.py:420), so the summation order is free.
		TP_PAR_FOR(j, M) fim[j] = 0.0;
		TP_SYNC();
		TP_NO_UNROLL   // (unrolled seven times the stages cost 476 bytes of scratch per lane and the fused kernel its third wavefront per SIMD)
		for (int lg = 6; lg >= 0; --lg) {   // half-length h = 2^lg of the butterflies: 64 .. 1
			const int h = 1 << lg;
			TP_PAR_FOR(bf, M / 2) {
				const int i = ((bf & ~(h - 1)) << 1) | (bf & (h - 1));
				const int tw = (bf & (h - 1)) << (6 - lg);
				const double ar = fre[i], ai = fim[i], br = fre[i + h], bi = fim[i + h];
				const double wr = k.twl[tw], wi = -k.twl[M + tw];          // e^{-2 pi i tw / M}
				const double dr = ar - br, di = ai - bi;
				fre[i] = ar + br; fim[i] = ai + bi;
				fre[i + h] = dr * wr - di * wi; fim[i + h] = dr * wi + di * wr;
			}
			TP_SYNC();
		}
		// FAC[k] * Y[k] / M at the bit-reversed position of k; the spectrum of real data is Hermitian (k and M - k), and the
		// imaginary parts of k = 0 and k = M / 2 do not enter a real inverse transform
		TP_PAR_FOR(pos, M) {
			int kk = 0;
			for (int q = 0; q < 7; ++q) kk = (kk << 1) | ((pos >> q) & 1);   // M = 128: seven bits
			const int jj = (kk <= M / 2) ? kk : (M - kk);
			const double FAC1 = 2.0 * ((kPi * bw / RANGE) * (kPi * bw / RANGE));
			const double J = (double)jj;
			const double BC = 1.0 - 1.0 / 3.0 * ((J * 1.0 / (double)M * kPi) * (J * 1.0 / (double)M * kPi));
			const double FAC = tp_exp(-(J * J * FAC1)) / BC;
			fre[pos] = FAC * (fre[pos] / (double)M);
			fim[pos] = (kk == 0 || kk == M / 2) ? 0.0 : FAC * (fim[pos] / (double)M);
		}
		TP_SYNC();
		TP_NO_UNROLL
		for (int lg = 0; lg <= 6; ++lg) {
			const int h = 1 << lg;
			TP_PAR_FOR(bf, M / 2) {
				const int i = ((bf & ~(h - 1)) << 1) | (bf & (h - 1));
				const int tw = (bf & (h - 1)) << (6 - lg);
				const double wr = k.twl[tw], wi = k.twl[M + tw];           // e^{+2 pi i tw / M}
				const double br = fre[i + h], bi = fim[i + h];
				const double tr = br * wr - bi * wi, ti = br * wi + bi * wr;
				const double ar = fre[i], ai = fim[i];
				fre[i] = ar + tr; fim[i] = ai + ti;
				fre[i + h] = ar - tr; fim[i + h] = ai - ti;
			}
			TP_SYNC();
		}
		// (dens = fre: the real part, in natural order)
		// support[argmax(density)]: np.argmax returns the first maximum, NaN counts as maximum
		// (two grid points per lane, then the tree picks the larger value / the smaller index on ties)
		TP_LANE_LOOP(l) {
			const double d0 = dens[l], d1 = dens[l + 64];
			// key: NaN beats everything (earliest NaN wins), else larger value, ties -> smaller index
			int idx; double val;
			if (tp_isnan(d0)) { idx = l; val = tp_inf(); }
			else if (tp_isnan(d1)) { idx = l + 64; val = tp_inf(); }
			else if (d1 > d0) { idx = l + 64; val = d1; }
			else { idx = l; val = d0; }
			k.red[l] = val; k.ired[l] = idx;
		}
		TP_SYNC();
		const double vmax = max_arr(k, k.red);
		TP_SYNC();
		// among the lanes holding the maximum, the smallest grid index (NaN candidates carry +inf and their index)
		TP_LANE_LOOP(l) { k.ired[l] = (k.red[l] == vmax) ? -k.ired[l] : -(1 << 30); }
		TP_SYNC();
		const int am = -max_ired(k);
		TP_SYNC();
		// np.linspace(a, b, M)[am] = a + am*step, last point set to b exactly
		max_guess = (am == M - 1) ? b : (a + (double)am * delta);
		TP_SYNC();
	}
	TP_K2P2_CLOCK(k, 14);
	const double MODE = powell_mode(k, nc, bw, max_guess);
	TP_K2P2_CLOCK(k, 15);

	// MAD1 = mad_to_sigma * nanmedian(|Flux[Flux < MODE] - MODE|)   (k2p2v2.py:424)
	// Flux sorted ascending: the selection is the prefix [0, c)
	TP_LANE_LOOP(l) {
		int c = 0;
		for (int i = l; i < nflux; i += 64) c += (k.srt[i] < MODE) ? 1 : 0;
		k.ired[l] = c;
	}
	TP_SYNC();
	const int c = sum_ired(k);
	TP_SYNC();
	double med;
	if (c == 0) med = tp_nan();
	else if (c & 1) med = fabs(k.srt[c / 2] - MODE);
	else {
		const double d0 = fabs(k.srt[c / 2] - MODE), d1 = fabs(k.srt[c / 2 - 1] - MODE);
		med = (d0 + d1) / 2.0; // np.median: mean of the two middle values (ascending order: d0 <= d1)
	}
	const double MAD1 = kMadToSigma * med;
	const double CUT = MODE + prm.thresh * MAD1;
	*cut = CUT;
	if (t.diag) {
		TP_SERIAL { t.diag[0] = CUT; t.diag[1] = MODE; t.diag[2] = MAD1; t.diag[3] = bw; t.diag[4] = max_guess; }
	}
	return 0;
}

//--------------------------------------------------------------------------------------------------
// small image helpers
//--------------------------------------------------------------------------------------------------
// Connected-component labelling of the non-zero pixels of `in` by min-index propagation;
// conn8: 8- or 4-connectivity.  out[p] = 1-based component number in raster order of first pixel
// (scipy.ndimage.label numbering), 0 for background.  Returns the number of components.
// `win`: a window that holds every set pixel of `in` -- the propagation sweeps run over it alone (the other passes are single
// reads or stores per pixel and stay on the whole stamp)
inline TP_DEV int label_components_sweeps(Shared& k, const uint8_t* in, lab_t* out, bool conn8, const Win& win) {
	const int P = k.P, H = k.H, W = k.W;
	TP_PAR_FOR(p, P) out[p] = in[p] ? p : -1;
	TP_SYNC();
	while (true) {
		TP_LANE_LOOP(l) {
			int changed = 0;
			for (int q = l; q < win.n; q += 64) {
				const int p = win_pix(k, win, q);
				int cur = out[p];
				if (cur < 0) continue;
				const int r = row_of(k, p), c = p - r * W;
				int best = cur;
				for (int dr = -1; dr <= 1; ++dr) {
					for (int dc = -1; dc <= 1; ++dc) {
						if (dr == 0 && dc == 0) continue;
						if (!conn8 && dr != 0 && dc != 0) continue;
						const int rr = r + dr, cc = c + dc;
						if (rr < 0 || rr >= H || cc < 0 || cc >= W) continue;
						const int v = out[rr * W + cc];
						if (v >= 0 && v < best) best = v;
					}
				}
				if (best < cur) { out[p] = best; changed = 1; }
			}
			k.ired[l] = changed;
		}
		TP_SYNC();
		const int any = or_ired(k);
		TP_SYNC();
		if (!any) break;
		// pointer jumping: every pixel adopts the label of the pixel it points to (labels only decrease and
		// stay inside the component, so the fixed point -- the component's smallest index -- is unchanged)
		TP_PAR_FOR(q, win.n) { const int p = win_pix(k, win, q); const int cur = out[p]; if (cur >= 0) { const int nxt = out[cur]; if (nxt < cur) out[p] = nxt; } }
		TP_SYNC();
	}
	// roots -> consecutive numbers in raster order: every lane owns a contiguous chunk of pixels, counts its
	// roots, an exclusive scan over the 64 per-lane counts gives the chunk's first number
	const int chunk = (P + 63) / 64;
	TP_LANE_LOOP(l) {
		int c = 0;
		for (int p = l * chunk; p < (l + 1) * chunk && p < P; ++p) c += (out[p] == p) ? 1 : 0;
		k.ired[l] = c;
	}
	TP_SYNC();
	TP_LANE_LOOP(l) {
		int base = 0;
		for (int m = 0; m < l; ++m) base += k.ired[m];
		for (int p = l * chunk; p < (l + 1) * chunk && p < P; ++p) if (out[p] == p) k.hpix[p] = ++base;
	}
	TP_SYNC();
	const int n = sum_ired(k);
	TP_PAR_FOR(p, P) { const int root = out[p]; k.hage[p] = (root >= 0) ? k.hpix[root] : 0; }
	TP_SYNC();
	TP_PAR_FOR(p, P) out[p] = k.hage[p];
	TP_SYNC();
	return n;
}
// The same labelling for stamps up to 64 x 64 with the ROWS AS BIT MASKS: lane r holds row r of `in` as a 64-bit word, a component is
// flooded from its first pixel in raster order -- within a row a seed fills its whole run of set pixels by a parallel-prefix carry
// (six shifts each way), between rows by the neighbours' words (for 8-connectivity widened by one bit each way) -- until no row changes.
// One LDS word per row and step where the sweeps read nine labels per pixel and sweep; the numbering (components in raster order of
// their first pixel, scipy.ndimage.label's) is the order the floods are started in.  Scratch: k.tmp (the rows), k.red (what is left),
// k.hval and k.dist (the flood, double-buffered): 64 words each; callers for which k.dist is live take the sweeps.
inline TP_DEV uint64_t fill_runs(uint64_t m, uint64_t s) {
	// every run of consecutive set bits of m that holds a bit of s, entirely (Kogge-Stone carry towards both ends)
	uint64_t g = s & m, p = m;
	g |= p & (g << 1); p &= (p << 1);
	g |= p & (g << 2); p &= (p << 2);
	g |= p & (g << 4); p &= (p << 4);
	g |= p & (g << 8); p &= (p << 8);
	g |= p & (g << 16); p &= (p << 16);
	g |= p & (g << 32);
	uint64_t h = g; p = m;
	h |= p & (h >> 1); p &= (p >> 1);
	h |= p & (h >> 2); p &= (p >> 2);
	h |= p & (h >> 4); p &= (p >> 4);
	h |= p & (h >> 8); p &= (p >> 8);
	h |= p & (h >> 16); p &= (p >> 16);
	h |= p & (h >> 32);
	return h;
}
inline TP_DEV int label_components_rows(Shared& k, const uint8_t* in, lab_t* out, bool conn8) {
	const int P = k.P, H = k.H, W = k.W;
	uint64_t* M = (uint64_t*)k.tmp;     // the rows of `in`
	uint64_t* R = (uint64_t*)k.red;     // ... of the pixels not yet in a component
	uint64_t* C = (uint64_t*)k.hval;    // the component being flooded
	uint64_t* D = (uint64_t*)k.dist;    // ... after the step in work
	TP_PAR_FOR(p, P) out[p] = 0;
	TP_LANE_LOOP(l) {
		uint64_t m = 0;
		if (l < H) for (int c = 0; c < W; ++c) m |= (uint64_t)(in[l * W + c] ? 1 : 0) << c;
		M[l] = m; R[l] = m;
	}
	TP_SYNC();
	int n = 0;
	while (true) {
		// the first pixel, in raster order, that is in no component yet: the lowest row with one, its lowest column
		TP_LANE_LOOP(l) { k.ired[l] = (l < H && R[l] != 0) ? (64 - l) : 0; }
		TP_SYNC();
		const int top = max_ired(k);
		TP_SYNC();
		if (top == 0) break;
		const int r0 = 64 - top;
		++n;
		TP_LANE_LOOP(l) { const uint64_t r = R[l]; C[l] = (l == r0) ? fill_runs(M[l], r & (0 - r)) : 0; }
		TP_SYNC();
		while (true) {
			TP_LANE_LOOP(l) {
				int changed = 0;
				uint64_t s = C[l];
				if (l < H) {
					uint64_t v = ((l > 0) ? C[l - 1] : 0) | ((l + 1 < H) ? C[l + 1] : 0);
					if (conn8) v |= (v << 1) | (v >> 1);
					const uint64_t m = M[l];
					const uint64_t grown = s | (v & m);
					if (grown != s) { s = fill_runs(m, grown); changed = 1; }
				}
				D[l] = s;
				k.ired[l] = changed;
			}
			TP_SYNC();
			const int any = or_ired(k);
			TP_SYNC();
			{ uint64_t* x = C; C = D; D = x; }
			if (!any) break;
		}
		TP_LANE_LOOP(l) {
			uint64_t c = C[l];
			R[l] &= ~c;
			while (c) { const int b = __builtin_ctzll(c); out[l * W + b] = (lab_t)n; c &= c - 1; }
		}
		TP_SYNC();
	}
	return n;
}
// `scratch_free`: k.dist holds nothing the caller still needs (the rows version uses its first 64 words)
inline TP_DEV int label_components(Shared& k, const uint8_t* in, lab_t* out, bool conn8, const Win& win, bool scratch_free = true) {
	if (scratch_free && k.H <= 64 && k.W <= 64) return label_components_rows(k, in, out, conn8);
	return label_components_sweeps(k, in, out, conn8, win);
}
inline TP_DEV int label_components(Shared& k, const uint8_t* in, lab_t* out, bool conn8) { return label_components(k, in, out, conn8, win_full(k)); }

// bottleneck.nanmedian of v[0..n) (n small); returns NaN for no valid value.  Serial.
inline TP_DEV double nanmedian_small(const double* v, int n, double* scratch) {
	int m = 0;
	for (int i = 0; i < n; ++i) {
		const double x = v[i];
		if (tp_isnan(x)) continue;
		int j = m++;
		while (j > 0 && scratch[j - 1] > x) { scratch[j] = scratch[j - 1]; --j; }
		scratch[j] = x;
	}
	if (m == 0) return tp_nan();
	if (m & 1) return scratch[m / 2];
	return (scratch[m / 2 - 1] + scratch[m / 2]) / 2.0;
}

// k2p2_saturated for ONE mask (k2p2v2.py:291-341): mask in k.msk -> additions in k.sat.
// Uses k.tmp / k.dist as column scratch (dist holds no live data at either call site: before the cluster's blur, after the
// watershed).  Returns (uniform) number of pixels set in k.sat.
// `win`: a window that holds every pixel of the mask (the columns and rows outside it have no mask pixel: nothing to test there).
inline TP_DEV int saturated_one(Shared& k, const Win& win) {
	const int P = k.P, H = k.H, W = k.W;
	TP_PAR_FOR(p, P) k.sat[p] = 0;
	// mask_max = nanmax(S[mask])
	TP_LANE_LOOP(l) {
		double m = -tp_inf(); int any = 0;
		for (int q = l; q < win.n; q += 64) { const int p = win_pix(k, win, q); if (k.msk[p] && !tp_isnan(k.S[p])) { if (!any || k.S[p] > m) m = k.S[p]; any = 1; } }
		k.red[l] = m; k.ired[l] = any;
	}
	TP_SYNC();
	double mask_max = tp_nan();
	{
		const int any = or_ired(k);
		const double m = max_arr(k, k.red);
		if (any) mask_max = m;
	}
	TP_SYNC();
	// one column per lane (columns are independent: k2p2v2.py:312-339)
	TP_PAR_FOR(ci, win.w) {
		const int c = win.c0 + ci;
		double* pix = k.tmp + (size_t)c * H;      // [H] per column (W*H = P doubles)
		double* scr = k.dist + (size_t)c * H;
		int n = 0;
		for (int r = win.r0; r < win.r0 + win.h; ++r) if (k.msk[r * W + c]) pix[n++] = k.S[r * W + c];
		if (n == 0) continue;
		// ratio = |nanmedian(diff(pixels))| / nanmax(pixels)
		double pmax = tp_nan(); { int any = 0; for (int i = 0; i < n; ++i) if (!tp_isnan(pix[i])) { if (!any || pix[i] > pmax) pmax = pix[i]; any = 1; } }
		// necessary condition of k2p2v2.py:321 (median(pixels) >= mask_max/2) fails whenever the column maximum
		// is below mask_max/2 (or everything is NaN): skip the two medians
		if (!(pmax >= mask_max / 2)) continue;
		// sharper necessary condition for the same test: a median >= t needs at least ceil(m / 2) of the m non-NaN values
		// >= t (odd m: the middle one; even m: the upper middle one).  Ordinary stars have one or two such pixels per
		// column, so the two insertion-sort medians below are skipped for them.
		{
			int m = 0, cge = 0;
			for (int i = 0; i < n; ++i) if (!tp_isnan(pix[i])) { ++m; cge += (pix[i] >= mask_max / 2) ? 1 : 0; }
			if (cge < (m + 1) / 2) continue;
		}
		const double medpix = nanmedian_small(pix, n, scr);
		// diff in place (pix no longer needed afterwards except through medpix/pmax)
		for (int i = 0; i + 1 < n; ++i) pix[i] = pix[i + 1] - pix[i];
		const double meddiff = nanmedian_small(pix, n - 1, scr);
		const double ratio = fabs(meddiff) / pmax;
		if (ratio < 0.01 && medpix >= mask_max / 2) {
			// imax = nanargmax(S * mask * column_mask) over the whole image (first maximum in raster order)
			int imax = -1; double best = 0.0;
			for (int p = 0; p < P; ++p) {
				const int pr = row_of(k, p), pc = p - pr * W;
				const double v = k.S[p] * (double)(k.msk[p] ? 1 : 0) * (double)(pc == c ? 1 : 0);
				if (tp_isnan(v)) continue;
				if (imax < 0 || v > best) { best = v; imax = p; }
			}
			if (imax >= 0) {
				const int ir = row_of(k, imax), ic = imax - ir * W;
				// add_to_mask = idx & column; keep the 4-connected (vertical) run containing imax
				if (ic == c && k.idx[imax]) {
					int r0 = ir, r1 = ir;
					while (r0 > 0 && k.idx[(r0 - 1) * W + c]) --r0;
					while (r1 + 1 < H && k.idx[(r1 + 1) * W + c]) ++r1;
					for (int r = r0; r <= r1; ++r) k.sat[r * W + c] = 1;
				}
			}
		}
	}
	TP_SYNC();
	TP_LANE_LOOP(l) { int c = 0; for (int p = l; p < P; p += 64) c += k.sat[p]; k.ired[l] = c; }
	TP_SYNC();
	const int n = sum_ired(k);
	TP_SYNC();
	return n;
}
inline TP_DEV int saturated_one(Shared& k) { return saturated_one(k, win_full(k)); }

// ndimage.gaussian_filter(Z, 0.5): radius 2, correlate1d along axis 0 then axis 1, mode 'reflect',
// symmetric-kernel summation order of ni_filters.c: c*w0 + (l1+r1)*w1 + (l2+r2)*w2
inline TP_DEV int reflect_idx(int i, int n) {
	// scipy 'reflect' (d c b a | a b c d | d c b a), valid for any offset
	if (n == 1) return 0;
	const int n2 = 2 * n;
	if (i < 0) { i = -i - 1; }
	i = i % n2;
	if (i >= n) i = n2 - 1 - i;
	return i;
}
// `box`: a window outside which `in` is +0.0 (the cluster's bounding box).  The filter reaches two pixels: the row pass is non-zero
// on the box grown by two ROWS only, the column pass on the box grown by two pixels all round; everywhere else both passes of the
// whole-stamp filter give 0 * w0 + (0 + 0) * w1 + (0 + 0) * w2 = +0.0 -- written by a store pass instead of being computed.
inline TP_DEV void gaussian_blur(Shared& k, const Params& prm, const double* in, double* out, const Win& box) {
	const int P = k.P, H = k.H, W = k.W;
	const double w0 = prm.gauss_w0, w1 = prm.gauss_w1, w2 = prm.gauss_w2;
	const bool whole = win_is_full(k, box);
	if (!whole) {
		TP_PAR_FOR(p, P) { k.tmp[p] = 0.0; out[p] = 0.0; }
		TP_SYNC();
	}
	// axis 0 (rows) -> tmp
	const Win rows = whole ? box : win_make(k, box.r0 - 2, box.r0 + box.h + 1, box.c0, box.c0 + box.w - 1);
	TP_PAR_FOR(q, rows.n) {
		const int p = win_pix(k, rows, q);
		const int r = row_of(k, p), c = p - r * W;
		double v = in[p] * w0;
		v += (in[reflect_idx(r - 1, H) * W + c] + in[reflect_idx(r + 1, H) * W + c]) * w1;
		v += (in[reflect_idx(r - 2, H) * W + c] + in[reflect_idx(r + 2, H) * W + c]) * w2;
		k.tmp[p] = v;
	}
	TP_SYNC();
	const Win both = whole ? box : win_make(k, box.r0 - 2, box.r0 + box.h + 1, box.c0 - 2, box.c0 + box.w + 1);
	TP_PAR_FOR(q, both.n) {
		const int p = win_pix(k, both, q);
		const int r = row_of(k, p), c = p - r * W;
		double v = k.tmp[p] * w0;
		v += (k.tmp[r * W + reflect_idx(c - 1, W)] + k.tmp[r * W + reflect_idx(c + 1, W)]) * w1;
		v += (k.tmp[r * W + reflect_idx(c - 2, W)] + k.tmp[r * W + reflect_idx(c + 2, W)]) * w2;
		out[p] = v;
	}
	TP_SYNC();
}
inline TP_DEV void gaussian_blur(Shared& k, const Params& prm, const double* in, double* out) { gaussian_blur(k, prm, in, out, win_full(k)); }

// skimage.segmentation.watershed(-Z, markers, mask=Z) (connectivity 1, no compactness / lines):
// priority flood, a neighbour is labelled when it is pushed, pops in (value, age) order.
// in: k.mark (markers); out: k.wsout.  The image values -Z[p] of a cluster are distinct floats, so the
// pop order is the order of the values: every in-mask pixel gets its RANK (brightest = 0; computed in
// parallel, exact ties broken by pixel index instead of push age), and the heap becomes a bit set of
// ranks with a two-level find-first-set -- O(1) per push / pop instead of LDS heap sifts.
inline TP_DEV void watershed(Shared& k, int nmark) {
	const int P = k.P, H = k.H, W = k.W;
	// How many marker labels survive `markers *= mask`?  With a single one the flood is simply "every in-mask
	// pixel 4-connected to the marker gets its label" -- a parallel connected-component labelling, no queue.
	TP_PAR_FOR(m, nmark + 1) k.hage[m] = 0;
	TP_SYNC();
	TP_PAR_FOR(p, P) if (k.Z[p] != 0.0 && k.mark[p] != 0) k.hage[k.mark[p]] = 1; // benign same-value stores
	TP_SYNC();
	TP_LANE_LOOP(l) { int c = 0, v = 0; for (int m = 1 + l; m <= nmark; m += 64) if (k.hage[m]) { c++; v = m; } k.ired[l] = c; k.red[l] = (double)v; }
	TP_SYNC();
	const int nm_in = sum_ired(k);
	const int the_label = (int)max_arr(k, k.red);
	TP_SYNC();
	if (nm_in <= 1) {
		TP_PAR_FOR(p, P) k.msk[p] = (k.Z[p] != 0.0) ? 1 : 0;
		TP_SYNC();
		const int ncomp = label_components(k, k.msk, k.wsout, false);
		TP_PAR_FOR(m, ncomp + 1) k.hage[m] = 0;
		TP_SYNC();
		TP_PAR_FOR(p, P) if (k.msk[p] && k.mark[p] != 0) k.hage[k.wsout[p]] = 1;
		TP_SYNC();
		TP_PAR_FOR(p, P) { const int c = k.wsout[p]; k.wsout[p] = (nm_in == 1 && c > 0 && k.hage[c]) ? the_label : 0; }
		TP_SYNC();
		return;
	}
	lab_t* rank = k.hpix;            // [P] rank of pixel (only where Z != 0)
	int32_t* ord = (int32_t*)k.hval; // [P] pixel of rank r (hval is free here; 2 int32 per double slot)
	uint32_t* words = (uint32_t*)k.hage; // [ceil(P/32)] bit set of pushed ranks
	TP_PAR_FOR(p, P) { k.wsout[p] = (k.Z[p] != 0.0) ? k.mark[p] : 0; rank[p] = -1; }
	// rank of every in-mask pixel by decreasing Z (ties: raster order).  The in-mask pixels are first compacted in raster
	// order (values in k.dist, pixel indices in k.tmp: both are free during the flood), so a pixel is compared with the
	// few dozen pixels of its cluster instead of with the whole stamp.
	double* cz = k.dist;
	int32_t* cp = (int32_t*)k.tmp;
	const int chunk = (P + 63) / 64;
	TP_LANE_LOOP(l) {
		int c = 0;
		for (int p = l * chunk; p < (l + 1) * chunk && p < P; ++p) c += (k.Z[p] != 0.0) ? 1 : 0;
		k.ired[l] = c;
	}
	TP_SYNC();
	TP_LANE_LOOP(l) {
		int base = 0;
		for (int m = 0; m < l; ++m) base += k.ired[m];
		for (int p = l * chunk; p < (l + 1) * chunk && p < P; ++p) if (k.Z[p] != 0.0) { cz[base] = k.Z[p]; cp[base] = p; ++base; }
	}
	TP_SYNC();
	const int nz = sum_ired(k);
	TP_SYNC();
	TP_PAR_FOR(c, nz) {
		const double zc = cz[c];
		int r = 0;
		for (int d = 0; d < nz; ++d) {
			const double zd = cz[d];
			r += (zd > zc || (zd == zc && d < c)) ? 1 : 0;
		}
		rank[cp[c]] = r;
		ord[r] = cp[c];
	}
	const int nwords = (P + 31) / 32;
	TP_PAR_FOR(w, nwords) words[w] = 0u;
	TP_SYNC();
	// seeds: the ranks of the marker pixels (parallel, the bit set is an OR)
	TP_PAR_FOR(p, P) if (k.wsout[p] != 0) { const int r = rank[p]; TP_ATOMIC_OR(&words[r >> 5], 1u << (r & 31)); }
	TP_SYNC();
	TP_SERIAL {
		// one pop of the flood: pixel of rank 32 w + b leaves the set, its unlabelled in-mask neighbours get its label and enter;
		// pw[q]: the word of the bit set that neighbour q's rank went into, or -1 (the caller keeps its summary of non-zero words in
		// plain variables: captured by reference in a callback they were moved to scratch memory)
		auto pop = [&](int w, int b, int (&pw)[4]) {
			const int px = ord[w * 32 + b];
			const int lbl = k.wsout[px];
			const int r = row_of(k, px), c = px - r * W;
			// the four neighbours: every LDS read first (independent, one latency), then the decisions and the writes.
			// Two neighbours are never the same pixel, so reading ahead does not change the sequential semantics.
			const int nbr[4] = {r - 1, r, r, r + 1};
			const int nbc[4] = {c, c - 1, c + 1, c};
			int nb[4], wl[4], rn[4];
			double zn[4];
			for (int q = 0; q < 4; ++q) {
				const bool in = !(nbr[q] < 0 || nbr[q] >= H || nbc[q] < 0 || nbc[q] >= W);
				nb[q] = in ? (nbr[q] * W + nbc[q]) : px;       // out of the image: re-read the pixel itself (labelled: skipped)
				zn[q] = k.Z[nb[q]];
				wl[q] = k.wsout[nb[q]];
				rn[q] = rank[nb[q]];
			}
			for (int q = 0; q < 4; ++q) {
				pw[q] = -1;
				if (zn[q] == 0.0) continue;      // not in mask
				if (wl[q] != 0) continue;        // already labelled (or the out-of-image stand-in)
				k.wsout[nb[q]] = lbl;
				const int wn = rn[q] >> 5;
				words[wn] |= (1u << (rn[q] & 31));
				pw[q] = wn;
			}
		};
		const int nzw = (nz + 31) >> 5;      // words that can hold a rank of this cluster
		if (nzw <= 96) {
			// summary: bit w of sum[w / 32] set <=> words[w] != 0.  Ranks count the in-mask pixels of ONE cluster: 96 words cover the
			// largest LDS-resident stamp (54 x 54) even if a single cluster filled it.
			uint32_t sum0 = 0u, sum1 = 0u, sum2 = 0u;
			for (int w = 0; w < nzw; ++w) if (words[w] != 0u) { if (w < 32) sum0 |= (1u << w); else if (w < 64) sum1 |= (1u << (w - 32)); else sum2 |= (1u << (w - 64)); }
			while (sum0 | sum1 | sum2) {
				const int w = sum0 ? __builtin_ctz(sum0) : (sum1 ? (32 + __builtin_ctz(sum1)) : (64 + __builtin_ctz(sum2)));
				uint32_t bits = words[w];
				const int b = __builtin_ctz(bits);
				bits &= bits - 1u;
				words[w] = bits;
				if (bits == 0u) { if (w < 32) sum0 &= ~(1u << w); else if (w < 64) sum1 &= ~(1u << (w - 32)); else sum2 &= ~(1u << (w - 64)); }
				int pw[4];
				pop(w, b, pw);
				for (int q = 0; q < 4; ++q) {
					const int wn = pw[q];
					if (wn < 0) continue;
					if (wn < 32) sum0 |= (1u << wn); else if (wn < 64) sum1 |= (1u << (wn - 32)); else sum2 |= (1u << (wn - 64));
				}
			}
		} else {
			// a cluster of more than 3 072 pixels (only in the stamps of the brightest stars, whose work arrays live in HBM): the
			// three summary words do not reach -- rounds 1-5 shifted past them.  The lowest non-zero word is found by a scan from
			// the lowest word that can be non-zero (a push below it moves it back).
			int wscan = 0;
			while (true) {
				while (wscan < nzw && words[wscan] == 0u) ++wscan;
				if (wscan >= nzw) break;
				const int w = wscan;
				uint32_t bits = words[w];
				const int b = __builtin_ctz(bits);
				bits &= bits - 1u;
				words[w] = bits;
				int pw[4];
				pop(w, b, pw);
				for (int q = 0; q < 4; ++q) if (pw[q] >= 0 && pw[q] < wscan) wscan = pw[q];
			}
		}
	}
	TP_SYNC();
}

//--------------------------------------------------------------------------------------------------
// the whole per-target pipeline
//--------------------------------------------------------------------------------------------------
inline TP_DEV float mags_total_f32(const float* tmag, const uint8_t* sel, int n) {
	// -2.5*log10(nansum(10**(-0.4*mags)))  in float32 like numpy on a float32 column
	float s = 0.f;
	for (int i = 0; i < n; ++i) {
		if (!sel[i]) continue;
		const float v = powf(10.0f, -0.4f * tmag[i]);
		if (v == v) s += v;
	}
	return -2.5f * log10f(s);
}

// Returns the STATUS integer; on return k.res holds the final mask (what was written to t.mask).
// (always inlined into its kernels: as a call -- which the inliner chooses once the body passes its size threshold -- it costs a
// stack frame in scratch memory and the fused kernel its register allocation)
TP_ALWAYS_INLINE TP_DEV int run_target(Shared& k, const Params& prm, const Target& t) {
	const int P = k.P, H = k.H, W = k.W;
	const bool wide = P > 256;    // per-cluster windows and the early rejection of clusters: for stamps above 16 x 16 (see A4)
	TP_PAR_FOR(p, P) { k.S[p] = t.S[p]; k.res[p] = 0; }
	TP_SYNC();
	int flags = 0;
	int status = 1; // STATUS.OK
	int err = 0;
	bool have_masks = false;
	int hits = 0;

	// ---------------- A2 ----------------
	double CUT = tp_nan();
	if (t.cut_override) { CUT = *t.cut_override; if (t.diag) { TP_SERIAL { t.diag[0] = CUT; } } }
	else err = threshold(k, prm, t, &CUT);
	TP_K2P2_CLOCK(k, 0);

	// target pixel (photometry.py:107): Python round() = round-half-even; negative indices wrap
	int tr = (int)rint(t.tpos_row - (double)t.stamp_row0);
	int tc = (int)rint(t.tpos_col - (double)t.stamp_col0);
	bool target_inside = true;
	if (tr < 0) tr += H;
	if (tc < 0) tc += W;
	if (tr < 0 || tr >= H || tc < 0 || tc >= W) target_inside = false;

	int nmasks_total = 0;
	if (!err) {
		// idx = S > CUT (NaN -> False)  (k2p2v2.py:449-450)
		TP_LANE_LOOP(l) {
			int c = 0; double margin = tp_inf();
			for (int p = l; p < P; p += 64) {
				const double s = k.S[p];
				const uint8_t v = (s > CUT) ? 1 : 0;
				k.idx[p] = v; c += v;
				if (!tp_isnan(s)) { const double d = fabs(s - CUT); if (d < margin) margin = d; }
			}
			k.ired[l] = c; k.red[l] = margin;
		}
		TP_SYNC();
		const int nidx = sum_ired(k);
		const double margin = min_arr(k, k.red);
		TP_SYNC();
		if (t.diag) { TP_SERIAL { t.diag[6] = margin; } }
		if (nidx == 0) {
			flags |= FLAG_NOSTARS; // K2P2NoStars -> minimum aperture
		} else {
			// ---------------- A3: DBSCAN on the grid ----------------
			TP_PAR_FOR(p, P) {
				const int r = row_of(k, p), c = p - r * W;
				int cnt = 0;
				for (int dr = -1; dr <= 1; ++dr) for (int dc = -1; dc <= 1; ++dc) {
					const int rr = r + dr, cc = c + dc;
					if (rr >= 0 && rr < H && cc >= 0 && cc < W) cnt += k.idx[rr * W + cc];
				}
				k.core[p] = (k.idx[p] && cnt >= prm.min_for_cluster) ? 1 : 0;
			}
			TP_SYNC();
			const int nclusters = label_components(k, k.core, k.mark, true);
			// lab: -2 outside idx, -1 noise, cluster id (0-based) for core; border = min neighbouring cluster
			TP_PAR_FOR(p, P) {
				int v = -2;
				if (k.idx[p]) {
					v = -1;
					if (k.core[p]) v = k.mark[p] - 1;
					else {
						const int r = row_of(k, p), c = p - r * W;
						int best = 0x7fffffff;
						for (int dr = -1; dr <= 1; ++dr) for (int dc = -1; dc <= 1; ++dc) {
							const int rr = r + dr, cc = c + dc;
							if (rr >= 0 && rr < H && cc >= 0 && cc < W && k.core[rr * W + cc]) {
								const int q = k.mark[rr * W + cc] - 1;
								if (q < best) best = q;
							}
						}
						if (best != 0x7fffffff) v = best;
					}
				}
				k.lab[p] = v;
			}
			TP_SYNC();
			TP_K2P2_CLOCK(k, 1);
			// ---------------- A4: watershed per cluster (segmentation=True, any cluster) ----------------
			// Labels after k2p2WS: non-core -> noise (k2p2v2.py:112)
			TP_PAR_FOR(p, P) k.lab2[p] = (k.idx[p]) ? ((k.core[p]) ? k.lab[p] : -1) : -2;
			TP_SYNC();
			// A cluster NO catalogue star can reach is rejected here, before anything is computed for it.  k2p2WS keeps a cluster only
			// if a peak of its blurred image is matched by a star: a star takes its nearest peak if that lies within 5 sqrt 2 pixels
			// (2 sqrt 2 for a star fainter than the saturation limit; k2p2v2.py:144-153), else none, and a cluster without a matched
			// peak has no markers and is dropped (:218-223).  Peaks are pixels where the blurred image exceeds the threshold
			// max(min, ws_thres * max) >= 0, i.e. pixels of the cluster's bounding box grown by the filter's two pixels: a star whose
			// distance to that rectangle -- the same expression sqrt(dx^2 + dy^2) on the nearest point of the rectangle; every
			// rounding in it is monotone, so it bounds the distance to every pixel of the rectangle from below -- is not below its
			// limit matches nothing.  What the skipped code could still do is raise "no peaks" (a cluster whose blurred image has no
			// pixel above the threshold: the error of k2p2v2.py:146): impossible when the fluxes above the threshold are positive and
			// finite (CUT >= 0), ws_thres is 0 and the grown box is not the whole stamp (there are zeros: the maximum is a peak above
			// the minimum) -- otherwise the cluster takes the long way.  On a resized stamp of a crowded field most clusters are noise
			// far from any star: 7 of 10 on 25 x 25 stamps, and they were 40 % of the mask builder's time there.
			// All clusters at once: the bounding boxes by one sweep with LDS atomics (four ints per cluster in k.tmp, free until the
			// loop below), the reach of the stars one cluster per lane, the flags in k.core (free from here on), ONE relabelling sweep.
			{
				int32_t* bb = (int32_t*)k.tmp;                 // [nclusters][4]: first row, last row, first column, last column; then [nclusters] reach flags
				int32_t* reach = bb + 4 * nclusters;
				const bool room = 5 * nclusters <= 2 * ((P < 64) ? 64 : P);
				// (a stamp with ONE cluster -- the target star's -- skips all of this: the sweeps would cost a 15 x 15 target of the step a
				// tenth of its builder time, measured)
				// (stamps up to 16 x 16 take neither this nor the windows below: with 1.6 clusters per target, all within reach of the
				// target star, the sweeps that find boxes cost more than they save -- 3 % of the step's mask + extraction launch, same-box A/B)
				bool guard = wide && room && nclusters > 1 && t.ncat > 0 && CUT >= 0.0 && prm.ws_thres == 0.0;
				if (guard) {
					TP_LANE_LOOP(l) {
						double smax = 0.0;
						for (int p = l; p < P; p += 64) if (k.idx[p]) { const double sv = k.S[p]; smax = (sv > smax) ? sv : smax; }   // (S > CUT there: never NaN)
						k.red[l] = smax;
					}
					TP_SYNC();
					guard = max_arr(k, k.red) < 1e300;
					TP_SYNC();
				}
				if (guard) {
					TP_PAR_FOR(c, nclusters) { bb[4 * c] = H; bb[4 * c + 1] = -1; bb[4 * c + 2] = W; bb[4 * c + 3] = -1; reach[c] = 0; }
					TP_SYNC();
					TP_PAR_FOR(p, P) {
						const int c = k.lab[p];
						if (c >= 0) {
							const int r = row_of(k, p), cc = p - r * W;
							TP_ATOMIC_MIN(&bb[4 * c], r); TP_ATOMIC_MAX(&bb[4 * c + 1], r); TP_ATOMIC_MIN(&bb[4 * c + 2], cc); TP_ATOMIC_MAX(&bb[4 * c + 3], cc);
						}
					}
					TP_SYNC();
					// one (cluster, star) pair per lane and step: the stars' catalogue rows are read side by side, not one after the other
					const int npairs = nclusters * t.ncat;
					TP_PAR_FOR(q, npairs) {
						const int c = q / t.ncat, sidx = q - c * t.ncat;
						const Win grown = win_make(k, bb[4 * c] - 2, bb[4 * c + 1] + 2, bb[4 * c + 2] - 2, bb[4 * c + 3] + 2);
						int hit = 1;
						if (!win_is_full(k, grown)) {
							const double xlo = (double)grown.c0, xhi = (double)(grown.c0 + grown.w - 1), ylo = (double)grown.r0, yhi = (double)(grown.r0 + grown.h - 1);
							const double c0 = (double)t.cat_col[sidx], c1 = (double)t.cat_row[sidx];
							if ((c0 == c0) && (c1 == c1)) {                                // (a NaN position: the long way decides)
								double dx = 0.0, dy = 0.0;
								if (c0 < xlo) dx = xlo - c0; else if (c0 > xhi) dx = xhi - c0;
								if (c1 < ylo) dy = ylo - c1; else if (c1 > yhi) dy = yhi - c1;
								const double d = sqrt(dx * dx + dy * dy);
								const double dist_factor = ((double)t.cat_tmag[sidx] > prm.saturation_limit) ? 2.0 : 5.0;
								if (d >= dist_factor * 1.4142135623730951) hit = 0;
							}
						}
						if (hit) TP_ATOMIC_OR(&reach[c], 1);
					}
					TP_SYNC();
				}
				TP_PAR_FOR(c, nclusters) k.core[c] = (guard && reach[c] == 0) ? 1 : 0;
				TP_SYNC();
				if (guard) {
					TP_PAR_FOR(p, P) { const int c = k.lab2[p]; if (c >= 0 && k.core[c]) k.lab2[p] = -1; }
					TP_SYNC();
				}
			}
			int max_label = nclusters - 1;
#ifdef TP_LAB_K2P2_CLOCK
			k.clk[11] += (unsigned long long)nclusters;
#endif
			for (int lab = 0; lab < nclusters && !err; ++lab) {
				// a cluster no catalogue star can reach was rejected before the loop (below: `far`)
				if (k.core[lab]) { TP_K2P2_CLOCK(k, 8); continue; }   // (uniform)
				// pre-pass saturated mask of the un-split cluster incl. border points (k2p2v2.py:465-492)
				// ... and the cluster's bounding box (core and border pixels): the passes below that cost more than a store per pixel
				// run over windows around it (see struct Win), not over the stamp once per cluster -- on a 25 x 25 stamp of a crowded
				// field, with a dozen clusters, they were half of the mask builder's time
				// (per-lane partials in k.ired / k.red / k.tmp, which lie outside the region the two phases of the builder share: the
				// KDE grid of A2 overlaps A4's images and k.hval there.  Four extrema, three arrays: two sweeps.)
				int br0 = 0, br1 = H - 1, bc0 = 0, bc1 = W - 1;
				if (wide) {
					TP_LANE_LOOP(l) {
						int r0 = H, r1 = -1, c1 = -1;
						for (int p = l; p < P; p += 64) {
							const uint8_t m = (k.lab[p] == lab) ? 1 : 0;
							k.msk[p] = m;
							if (m) {
								const int r = row_of(k, p), c = p - r * W;
								r0 = (r < r0) ? r : r0; r1 = (r > r1) ? r : r1; c1 = (c > c1) ? c : c1;
							}
						}
						k.ired[l] = r1; k.red[l] = (double)(-r0); k.tmp[l] = (double)c1;
					}
					TP_SYNC();
					br1 = max_ired(k); br0 = -(int)max_arr(k, k.red); bc1 = (int)max_arr(k, k.tmp);
					TP_SYNC();
					TP_LANE_LOOP(l) {
						int c0 = W;
						for (int p = l; p < P; p += 64) if (k.msk[p]) { const int r = row_of(k, p), c = p - r * W; c0 = (c < c0) ? c : c0; }
						k.ired[l] = -c0;
					}
					TP_SYNC();
					bc0 = -max_ired(k);
					TP_SYNC();
				} else {
					TP_PAR_FOR(p, P) k.msk[p] = (k.lab[p] == lab) ? 1 : 0;
					TP_SYNC();
				}
				const Win box = win_make(k, br0, br1, bc0, bc1);
				const int nsat = saturated_one(k, box);
				TP_K2P2_CLOCK(k, 2);
				// Z = flux on the core pixels of this cluster
				TP_PAR_FOR(p, P) k.Z[p] = (k.lab2[p] == lab) ? k.S[p] : 0.0;
				TP_SYNC();
				gaussian_blur(k, prm, k.Z, k.dist, box);
				// peak_local_max(distance, exclude_border=False, threshold_rel=ws_thres, footprint=ones(3,3))
				// the blurred image is zero outside the box grown by two pixels (gaussian_blur): its extrema are those of that window
				// and, unless the window is the stamp, zero
				const Win near2 = win_make(k, br0 - 2, br1 + 2, bc0 - 2, bc1 + 2);
				TP_LANE_LOOP(l) {
					double mn = tp_inf(), mx = -tp_inf();
					if (l == 0 && !win_is_full(k, near2)) { mn = 0.0; mx = 0.0; }
					for (int q = l; q < near2.n; q += 64) { const double d = k.dist[win_pix(k, near2, q)]; if (d < mn) mn = d; if (d > mx) mx = d; }
					k.red[l] = mn; k.hval[l] = mx;
				}
				TP_SYNC();
				const double dmin = min_arr(k, k.red), dmax = max_arr(k, k.hval);
				TP_SYNC();
				double pk_thr = dmin;
				{ const double rel = prm.ws_thres * dmax; if (rel > pk_thr) pk_thr = rel; } // max(min, rel*max)
				// Candidate peaks: a pixel that equals the maximum of its 3 x 3 neighbourhood.  Three pixels or more from the box the
				// neighbourhood is all zero: such a pixel is a candidate (0 == 0) that the threshold below drops as long as it is not
				// negative (0 > pk_thr is false) -- so the test runs over the box grown by three, and `trivial` (every pixel of the
				// STAMP is a candidate) is decided there.  A negative (or NaN) threshold -- only with negative fluxes above a negative
				// CUT, or a negative ws_thres -- takes the pass over the whole stamp.
				const Win near3 = (pk_thr >= 0.0) ? win_make(k, br0 - 3, br1 + 3, bc0 - 3, bc1 + 3) : win_full(k);
				if (!win_is_full(k, near3)) { TP_PAR_FOR(p, P) k.lmax[p] = 0; }
				TP_SYNC();
				TP_LANE_LOOP(l) {
					int allpk = 1;
					for (int q = l; q < near3.n; q += 64) {
						const int p = win_pix(k, near3, q);
						const int r = row_of(k, p), c = p - r * W;
						double m = 0.0; // mode='constant', cval=0: out-of-image neighbours count as 0
						bool first = true;
						for (int dr = -1; dr <= 1; ++dr) for (int dc = -1; dc <= 1; ++dc) {
							const int rr = r + dr, cc = c + dc;
							const double v = (rr >= 0 && rr < H && cc >= 0 && cc < W) ? k.dist[rr * W + cc] : 0.0;
							if (first || v > m) { m = v; first = false; }
						}
						const uint8_t pk = (k.dist[p] == m) ? 1 : 0;
						k.lmax[p] = pk; // candidate peaks
						if (!pk) allpk = 0;
					}
					k.ired[l] = allpk;
				}
				TP_SYNC();
				const int trivial = and_ired(k);
				TP_SYNC();
				TP_LANE_LOOP(l) {
					int c = 0;
					for (int q = l; q < near3.n; q += 64) {
						const int p = win_pix(k, near3, q);
						const uint8_t pk = (!trivial && k.lmax[p] && k.dist[p] > pk_thr) ? 1 : 0;
						k.lmax[p] = pk; c += pk;
					}
					k.ired[l] = c;
				}
				TP_SYNC();
				const int npeaks = sum_ired(k);
				TP_SYNC();
				TP_K2P2_CLOCK(k, 3);
				// peaks matched to catalog stars (k2p2v2.py:144-153); candidates stay in lmax, selection in sat? no:
				// selection goes to k.core-independent temp: reuse wsout as "selected" flags
				TP_PAR_FOR(p, P) k.wsout[p] = 0;
				TP_SYNC();
				if (t.ncat > 0 && npeaks == 0) { err = ERR_NO_PEAKS; break; }
				// compact list of the peak pixels (order irrelevant: the selection below is a total order)
				TP_SERIAL { k.scal[3] = 0; }
				TP_SYNC();
				TP_PAR_FOR(p, P) if (k.lmax[p]) { const int slot = TP_ATOMIC_INC(&k.scal[3]); k.hpix[slot] = p; }
				TP_SYNC();
				TP_PAR_FOR(s, t.ncat) {
					const double c0 = (double)t.cat_col[s], c1 = (double)t.cat_row[s];
					int bi = -1; double bd = 0.0, bint = 0.0;
					for (int e = 0; e < npeaks; ++e) {
						const int p = k.hpix[e];
						const int r = row_of(k, p), c = p - r * W;
						const double dx = (double)c - c0, dy = (double)r - c1;
						const double d = sqrt(dx * dx + dy * dy);
						// np.argmin over the peaks sorted by decreasing intensity (stable: raster order among equal
						// intensities): first minimum of d; np.argmin treats NaN as the minimum
						const double inten = k.dist[p];
						bool better;
						if (bi < 0) better = true;
						else {
							const bool dn = tp_isnan(d), bn = tp_isnan(bd);
							const bool earlier = (inten > bint) || (inten == bint && p < bi); // position in the sorted peak list
							if (dn && bn) better = earlier;
							else if (bn) better = false;
							else if (dn) better = true;
							else if (d < bd) better = true;
							else if (d == bd) better = earlier;
							else better = false;
						}
						if (better) { bi = p; bd = d; bint = inten; }
					}
					if (bi >= 0) {
						const double dist_factor = ((double)t.cat_tmag[s] > prm.saturation_limit) ? 2.0 : 5.0;
						if (bd < dist_factor * 1.4142135623730951) k.wsout[bi] = 1; // benign same-value race
					}
				}
				TP_SYNC();
				TP_LANE_LOOP(l) { // local_maxi
					int c = 0;
					for (int q = l; q < near3.n; q += 64) { const int p = win_pix(k, near3, q); const uint8_t v = k.wsout[p] ? 1 : 0; k.lmax[p] = v; c += v; }
					k.ired[l] = c;
				}
				TP_SYNC();
				const int nselected = sum_ired(k);
				TP_SYNC();
				TP_K2P2_CLOCK(k, 4);
				// de-duplicate maxima inside saturated patches (k2p2v2.py:193-212): only a patch that holds two selected maxima changes
				if (nsat > 0 && nselected > 1) {
					const int ncomp = label_components(k, k.sat, k.mark, false, win_full(k), false);   // (k.dist is read below: the sweeps)
					for (int cc = 1; cc <= ncomp; ++cc) {
						TP_LANE_LOOP(l) {
							int c = 0;
							for (int p = l; p < P; p += 64) c += (k.lmax[p] && k.mark[p] == cc) ? 1 : 0;
							k.ired[l] = c;
						}
						TP_SYNC();
						const int nin = sum_ired(k);
						TP_SYNC();
						if (nin > 1) {
							TP_SERIAL {
								// imax = nanargmax(distance * local_maxi * sp): first maximum in raster order
								int imax = -1; double best = 0.0;
								for (int p = 0; p < P; ++p) {
									const double v = k.dist[p] * (double)k.lmax[p] * (double)((k.mark[p] == cc) ? 1 : 0);
									if (tp_isnan(v)) continue;
									if (imax < 0 || v > best) { best = v; imax = p; }
								}
								for (int p = 0; p < P; ++p) if (k.mark[p] == cc) k.lmax[p] = 0;
								if (imax >= 0) k.lmax[imax] = 1;
							}
							TP_SYNC();
						}
					}
				}
				TP_K2P2_CLOCK(k, 5);
				// markers = ndimage.label(local_maxi) (4-connectivity)
				// (no peak selected: no markers -- ndimage.label of an empty image -- without the labelling passes)
				const int nmark = (nselected > 0) ? label_components(k, k.lmax, k.mark, false, near3) : 0;   // (the selected peaks lie inside the window of the candidates)
				TP_K2P2_CLOCK(k, 6);
				if (nmark == 0) {
					// "No maxima were found": the cluster is rejected (k2p2v2.py:218-223)
					TP_PAR_FOR(p, P) if (k.lab2[p] == lab) k.lab2[p] = -1;
					TP_SYNC();
				} else {
					watershed(k, nmark); // k.mark -> k.wsout
					TP_K2P2_CLOCK(k, 7);
					// no_labels = number of distinct values in labels_ws, zero included (k2p2v2.py:230)
					TP_PAR_FOR(m, nmark + 1) k.hage[m] = 0;
					TP_SYNC();
					TP_PAR_FOR(p, P) k.hage[k.wsout[p]] = 1; // benign same-value stores
					TP_SYNC();
					TP_LANE_LOOP(l) { int c = 0; for (int m = l; m <= nmark; m += 64) c += k.hage[m]; k.ired[l] = c; }
					TP_SYNC();
					const int ndistinct = sum_ired(k);
					TP_SERIAL { k.scal[1] = ndistinct; }
					TP_SYNC();
					const int no_labels = k.scal[1];
					TP_SYNC();
					TP_PAR_FOR(p, P) {
						if (k.lab2[p] == lab) {
							int v = -1;
							const int wlab = k.wsout[p]; // Z != 0 here by construction
							if (wlab == 1) v = lab;
							else if (wlab >= 2 && (wlab - 2) < (no_labels - 2)) v = max_label + (wlab - 1);
							k.lab2[p] = v;
						}
					}
					TP_SYNC();
					if (no_labels - 2 > 0) max_label += (no_labels - 2);
				}
				TP_K2P2_CLOCK(k, 8);
			}

			// ---------------- A5: mask assembly, one candidate mask at a time ----------------
			if (!err) {
				for (int lab = 0; lab <= max_label && !err; ++lab) {
					TP_LANE_LOOP(l) { int c = 0; for (int p = l; p < P; p += 64) c += (k.lab2[p] == lab) ? 1 : 0; k.ired[l] = c; }
					TP_SYNC();
					const int npx = sum_ired(k);
					TP_SYNC();
					if (npx < prm.min_no_pixels_in_mask) continue;
					nmasks_total++;
					have_masks = true;
					TP_PAR_FOR(p, P) k.msk[p] = (k.lab2[p] == lab) ? 1 : 0;
					TP_SYNC();
					// fill holes: not in mask and all four neighbours in mask (k2p2v2.py:549-554)
					TP_PAR_FOR(p, P) {
						const int r = row_of(k, p), c = p - r * W;
						uint8_t fill = 0;
						if (!k.msk[p] && r > 0 && r < H - 1 && c > 0 && c < W - 1)
							fill = (k.msk[p - W] && k.msk[p + W] && k.msk[p - 1] && k.msk[p + 1]) ? 1 : 0;
						k.lmax[p] = fill;
					}
					TP_SYNC();
					TP_PAR_FOR(p, P) if (k.lmax[p]) k.msk[p] = 1;
					TP_SYNC();
					// extend overflow columns (k2p2v2.py:579-623)
					if (prm.extend_overflow) {
						const int nsat = saturated_one(k);
						if (nsat > 0) {
							// stars inside the (hole-filled) mask decide whether the extension is allowed
							TP_SERIAL {
								float s = 0.f; int nst = 0;
								for (int i = 0; i < t.ncat; ++i) {
									const int c = (int)rintf(t.cat_col[i]), r = (int)rintf(t.cat_row[i]);
									if (c < 0 || c >= W || r < 0 || r >= H) continue;
									if (!k.msk[r * W + c]) continue;
									const float v = powf(10.0f, -0.4f * t.cat_tmag[i]);
									if (v == v) s += v;
									nst++;
								}
								int allow = 0;
								if (nst > 0) {
									const float mt = -2.5f * log10f(s);
									allow = !((double)mt > prm.saturation_limit);
								}
								k.scal[2] = allow;
							}
							TP_SYNC();
							if (k.scal[2]) { TP_PAR_FOR(p, P) if (k.sat[p]) k.msk[p] = 1; }
							TP_SYNC();
						}
					}
					// does this mask contain the target pixel?  (photometry.py:107)
					if (!target_inside) { err = ERR_TARGET_OUTSIDE; break; }
					if (k.msk[tr * W + tc]) {
						hits++;
						TP_PAR_FOR(p, P) k.res[p] = k.msk[p];
					}
					TP_SYNC();
				}
			}
		}
	}
	TP_K2P2_CLOCK(k, 9);
	bool using_min = false;
	if (!err) {
		if (!have_masks) { using_min = true; if (!(flags & FLAG_NOSTARS)) flags |= FLAG_NOMASKS; }
		else if (hits == 0) using_min = true;
		else if (hits > 1) err = ERR_TOO_MANY_MASKS;
	}
	if (!err && using_min) {
		// _minimum_aperture (photometry.py:31-41)
		TP_PAR_FOR(p, P) {
			const int r = row_of(k, p), c = p - r * W;
			const double dc = ((double)(t.stamp_col0 + c + 1) - t.tpos_col) - 1.0;
			const double dr = ((double)(t.stamp_row0 + r + 1) - t.tpos_row) - 1.0;
			// bit 1 of BasePhotometry.aperture is "pixel collected" = finite sum image (BasePhotometry.py:1043): enforced here as well,
			// so that a caller without the sum image (it is computed in this very launch) may pass an all-ones aperture
			const double sv = t.S[p];
			const bool collected = (t.aperture[p] & 1) != 0 && (fabs(sv) <= 1.7976931348623157e308);
			k.res[p] = (fabs(dc) <= 1.0 && fabs(dr) <= 1.0 && collected) ? 1 : 0;
		}
		TP_SYNC();
		flags |= FLAG_MIN_APERTURE;
	}

	double contamination = tp_nan();
	if (!err) {
		// edges (photometry.py:123-131)
		TP_LANE_LOOP(l) {
			int e = 0;
			for (int p = l; p < P; p += 64) if (k.res[p]) {
				const int r = row_of(k, p), c = p - r * W;
				if (r == 0) e |= FLAG_EDGE_DOWN;
				if (r == H - 1) e |= FLAG_EDGE_UP;
				if (c == 0) e |= FLAG_EDGE_LEFT;
				if (c == W - 1) e |= FLAG_EDGE_RIGHT;
			}
			k.ired[l] = e;
		}
		TP_SYNC();
		flags |= or_ired(k);
		TP_SYNC();
		// ---------------- A7: contamination (photometry.py:220-238) ----------------
		// Which catalogue stars lie in the mask, and 10^(-0.4 Tmag) of those, 64 stars at a time with a lane per star (the catalogue
		// rows come from global memory and powf is a hundred instructions: star after star, by every lane, this was an eighth of the
		// builder's time on a 15 x 15 stamp); the float32 sum of photometry.py:233 is then taken in catalogue order from LDS.
		int nin = 0, only = -1;
		float ssum = 0.f;
		for (int base = 0; base < t.ncat; base += 64) {
			TP_LANE_LOOP(l) {
				const int s2 = base + l;
				int in = 0;
				double v = 0.0;
				if (s2 < t.ncat) {
					// rows == np.round(t['row'])+1 with 1-based grid rows: stamp index = round(row) - stamp_row0
					const int r = (int)rintf(t.cat_ccd_row[s2]) - t.stamp_row0;
					const int c = (int)rintf(t.cat_ccd_col[s2]) - t.stamp_col0;
					if (r >= 0 && r < H && c >= 0 && c < W) in = k.res[r * W + c] ? 1 : 0;
					if (t.cat_in_mask) t.cat_in_mask[s2] = (uint8_t)in;
					if (in) v = (double)powf(10.0f, -0.4f * t.cat_tmag[s2]);   // (a float widened: exact both ways, NaN included)
				}
				k.ired[l] = in; k.red[l] = v;
			}
			TP_SYNC();
			const int cnt = (t.ncat - base < 64) ? (t.ncat - base) : 64;
			for (int j = 0; j < cnt; ++j) {       // every lane, the same (uniform)
				if (!k.ired[j]) continue;
				nin++; only = base + j;
				const float v = (float)k.red[j];
				if (v == v) ssum += v;
			}
			TP_SYNC();
		}
		if (nin == 0) { err = ERR_NO_TARGETS_IN_MASK; }
		else if (nin == 1 && t.cat_starid[only] == t.target_starid) contamination = 0.0;
		else {
			// numpy 1.21 scalar promotion: float32 mags_total widened to float64 before the subtraction
			const double mags_total = -2.5 * (double)log10f(ssum);
			double cont = 1.0 - pow(10.0, 0.4 * (mags_total - t.target_tmag));
			if (cont < 0.0) cont = 0.0;
			contamination = cont;
		}
	}
	if (err) {
		status = 2; // STATUS.ERROR
		if (err == ERR_NO_TARGETS_IN_MASK && (flags & FLAG_MIN_APERTURE)) status = 3; // photometry.py:253-254 overrides
	} else if (flags & FLAG_MIN_APERTURE) status = 3; // STATUS.WARNING
	flags |= (err << ERR_SHIFT);

	const bool keep_mask = (status != 2) || (err == ERR_NO_TARGETS_IN_MASK); // photometry.py:204 ran before :227
	TP_PAR_FOR(p, P) { const uint8_t v = keep_mask ? k.res[p] : 0; k.res[p] = v; t.mask[p] = v; }
	TP_SERIAL {
		*t.status = status;
		*t.flags = flags;
		*t.contamination = contamination;
		if (t.diag) t.diag[7] = (double)nmasks_total;
	}
	TP_SYNC();
	TP_K2P2_CLOCK(k, 10);
	return status;
}

} // namespace k2p2
