// linpsf.hip -- P1..P4: linear PSF photometry (fixed centroids, simultaneous linear least squares).
//
// Replaces photometry/psf.py (PSF.__init__ :35-119, PSF.integrate_to_image :122-148) and
// photometry/linpsf_photometry.py (lsfit :22-34, LinPSFPhotometry.do_photometry :79-219).
//
// P1 (tp_linpsf_prf).  The reference builds, per target, PRF = sum_i PRF_i / dist_i (inverse-distance
// blend of the 25 SPOC PRF samples to the stamp centre, psf.py:101-113), normalises it (:116) and
// fits an interpolating bicubic spline (:119).  The spline fit is LINEAR in the data, so the
// coefficient table of the blend is the same blend of the 25 per-sample coefficient tables, which the
// host fits once per (camera, CCD) with the same scipy call the reference uses.  The kernel is a
// register-stationary AXPY: each thread keeps one coefficient of all samples in VGPRs and streams
// over the targets, so the base tables are read once and only the per-target table is written.
//
// P2..P4 (tp_linpsf_fit).  One THREAD per cadence of a target.  The FITPACK box integral of
// the bicubic spline over a pixel is separable (psf.py:146 -> dblint/fpintb):
//     integral = sum_ab wx[a] C[a][b] wy[b],  w = integrals of the B-spline basis over the pixel edge.
// The PRF grid is uniform (9 samples per pixel) and a pixel is exactly 9 knot intervals wide, so
// for a pixel whose lower edge sits at fraction phi of knot interval l the 13 non-zero weights are
//     [1-M(phi+3), 1-M(phi+2), 1-M(phi+1), 1-M(phi), 1, 1, 1, 1, 1, M(phi+3), M(phi+2), M(phi+1), M(phi)] * h
// with M the cumulative cardinal cubic B-spline: 4 numbers per axis per star, the same for every pixel
// of the stamp (pixels are whole multiples of 9 knots apart).  The general kernel evaluates that 13 x 13 contraction
// per star, pixel and cadence from the table in LDS; the polynomial path (below) turns it into a biquartic in the
// two phases whose 25 coefficients are shared by all cadences with the same knot intervals.
// The normal equations (A^T A, A^T b) are accumulated on the fly; x = pinv(A^T A) A^T b via a cyclic
// Jacobi eigen-decomposition with numpy's pinv cutoff (rcond = 1e-15 * largest singular value).
//
// Roofline: the FP64 pipe, not HBM: the image cube is read once (P*T*4 bytes per target).  Since round 3 the fit of a target
// with up to 4 stars runs on the matrix cores (linpsf_mfma.hip: ONE quartic spline per star and pixel over the knot intervals the
// star visits, cadences in natural order); the kernels of this file plan it (tp_linpsf_plan_kernel), build its coefficients
// (tp_linpsf_coef_kernel) and finalise it (tp_linpsf_finalize_m_kernel), and fit the targets that do not qualify on the vector
// ALUs (tp_linpsf_fit2_kernel: a biquartic per pixel and table origin, cadences sorted by origin; tp_linpsf_fit_direct_kernel;
// tp_linpsf_fit_many_kernel).
#include "linpsf_common.h"

void* tp_ctx_scratch(tp_ctx* ctx, size_t bytes); // aperture.hip

namespace {

using namespace tp_prf;
using namespace tp_linpsf;

constexpr int kMaxSamples = 32;

//--------------------------------------------------------------------------------------------------
// P1: per-target blend of the per-sample coefficient tables
//--------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tp_linpsf_prf_kernel(const double* __restrict__ base, int n_samples, int n_coef,
	const double* __restrict__ weights, int n_targets, double* __restrict__ out)
{
	const int c = blockIdx.x * blockDim.x + threadIdx.x;
	if (c >= n_coef) return;
	double b[kMaxSamples];
#pragma unroll
	for (int s = 0; s < kMaxSamples; ++s) b[s] = (s < n_samples) ? base[(int64_t)s * n_coef + c] : 0.0;
	for (int t = blockIdx.y; t < n_targets; t += gridDim.y) {
		const double* w = weights + (int64_t)t * n_samples;
		double acc = 0.0;
#pragma unroll
		for (int s = 0; s < kMaxSamples; ++s) if (s < n_samples) acc += w[s] * b[s];
		out[(int64_t)t * n_coef + c] = acc;
	}
}

// The same for exactly NS samples (the SPOC PRF files hold 25 per CCD): with the count known the weights of a target are ONE
// batch of scalar loads (the run-time count above turns them into 25 dependent round trips: 0.67 ms for 10 000 targets against
// the 0.3 ms the 1.1 GB of tables take to write).  Same sums in the same order.
template <int NS>
__global__ __launch_bounds__(256) void tp_linpsf_prf_fixed_kernel(const double* __restrict__ base, int n_coef,
	const double* __restrict__ weights, int n_targets, double* __restrict__ out)
{
	const int c = blockIdx.x * blockDim.x + threadIdx.x;
	if (c >= n_coef) return;
	double b[NS];
#pragma unroll
	for (int s = 0; s < NS; ++s) b[s] = base[(int64_t)s * n_coef + c];
	for (int t = blockIdx.y; t < n_targets; t += gridDim.y) {
		const double* w = weights + (int64_t)t * NS;
		double wv[NS];
#pragma unroll
		for (int s = 0; s < NS; ++s) wv[s] = w[s];
		double acc = 0.0;
#pragma unroll
		for (int s = 0; s < NS; ++s) acc += wv[s] * b[s];
		out[(int64_t)t * n_coef + c] = acc;
	}
}

// General path: direct evaluation of the 13x13 contraction per star, pixel and cadence.  Runs only for the
// targets that the polynomial path could not take (`todo` flag set, or todo == nullptr).
template <int S, int SLO>
__global__ __launch_bounds__(512) void tp_linpsf_fit_direct_kernel(FitArgs a, const int32_t* __restrict__ todo)
{
	extern __shared__ __align__(16) double lds[]; // [n*n] coefficient table + 2 x [n+4] knots
	const int target = blockIdx.x;
	if (todo && todo[target] != kPathDirect) return;
	{ const int nst = (int)(a.star_offsets[target + 1] - a.star_offsets[target]); if (nst < SLO || nst > S) return; } // another instantiation's targets
	const int tid = threadIdx.x;
	const int n = a.n;
	double* C = lds;
	double* kn = lds + (size_t)n * n;
	double* kny = kn + n + 4;
	const double* cg = a.coef + (int64_t)target * n * n;
	for (int i = tid; i < n * n; i += blockDim.x) C[i] = cg[i];
	for (int i = tid; i < n + 4; i += blockDim.x) { kn[i] = a.knots_x[i]; kny[i] = a.knots_y[i]; }
	__syncthreads();

	const int k = blockIdx.y * blockDim.x + tid;
	if (k >= a.n_cad) return;
	const int64_t s0 = a.star_offsets[target];
	int ns = (int)(a.star_offsets[target + 1] - s0);
	if (ns > S) ns = S; // host guarantees ns <= S for this instantiation
	const int H = a.height, W = a.width;
	const double h = kn[5] - kn[4], hy = kny[5] - kny[4];
	const double cutoff = a.cutoff;

	// per star: edge weights (same for every pixel) and table origin of pixel 0
	double mx[S][4], my[S][4], srow[S], scol[S];
	int ax0[S], by0[S];
#pragma unroll
	for (int s = 0; s < S; ++s) {
		if (s < ns) {
			srow[s] = a.pos_row[(s0 + s) * a.pos_pitch + k];
			scol[s] = a.pos_col[(s0 + s) * a.pos_pitch + k];
			// x <-> column (first spline axis), y <-> row  (psf.py:146)
			axis_weights(kn, n, scol[s], h, mx[s], ax0[s]);
			axis_weights(kny, n, srow[s], hy, my[s], by0[s]);
		} else {
			srow[s] = scol[s] = 0.0; ax0[s] = by0[s] = 4;
#pragma unroll
			for (int q = 0; q < 4; ++q) { mx[s][q] = 0.0; my[s][q] = 0.0; }
		}
	}

	double G[S][S], g[S];
#pragma unroll
	for (int i = 0; i < S; ++i) { g[i] = 0.0;
#pragma unroll
		for (int j = 0; j < S; ++j) G[i][j] = 0.0; }

	const float* img = a.images + (int64_t)target * H * W * a.t_pitch + k;
	const float sub = a.subtract ? a.subtract[(int64_t)target * a.subtract_pitch + k] : 0.f;
	const double h2 = h * hy;
	for (int i = 0; i < H; ++i) {
		for (int j = 0; j < W; ++j) {
			float bf = img[(int64_t)(i * W + j) * a.t_pitch];
			if (a.subtract) bf = bf - sub;
			if (!(fabsf(bf) <= 3.402823466e+38f)) continue; // good_pixels = isfinite(img) (linpsf_photometry.py:123)
			const double b = (double)bf;
			double av[S];
#pragma unroll
			for (int s = 0; s < S; ++s) {
				av[s] = 0.0;
				if (s < ns) {
					const double dc = (double)j - scol[s], dr = (double)i - srow[s];
					// psf.py:142  sqrt((j-col)^2 + (i-row)^2) < cutoff_radius  (a NaN position is never inside: zero column)
					const bool inside = sqrt(dc * dc + dr * dr) < cutoff;
					if (inside) {
						int ax = ax0[s] + 9 * j, by = by0[s] + 9 * i;
						ax = ax < 0 ? 0 : (ax > n - 13 ? n - 13 : ax);
						by = by < 0 ? 0 : (by > n - 13 ? n - 13 : by);
						av[s] = h2 * prf_pixel(C, n, ax, by, mx[s], my[s]);
					}
				}
			}
#pragma unroll
			for (int s = 0; s < S; ++s) {
				g[s] += av[s] * b;
#pragma unroll
				for (int u = 0; u < S; ++u) if (u >= s) G[s][u] += av[s] * av[u];
			}
		}
	}
#pragma unroll
	for (int s = 0; s < S; ++s)
#pragma unroll
		for (int u = 0; u < S; ++u) if (u < s) G[s][u] = G[u][s];

	double x[S];
	pinv_solve<S>(G, g, ns, x);
	const int ti = a.target_index[target];
	double tf = __builtin_nan("");
#pragma unroll
	for (int s = 0; s < S; ++s) {
		if (s < ns) {
			a.fluxes_all[(s0 + s) * a.out_pitch + k] = x[s];
			if (s == ti) tf = x[s];
		}
	}
	a.flux[(int64_t)target * a.out_pitch + k] = tf;
	a.flux_err[(int64_t)target * a.out_pitch + k] = __builtin_nan("");
}

//--------------------------------------------------------------------------------------------------
// Polynomial path.  For a fixed table origin (ax0, by0) -- i.e. fixed knot intervals of the star's sub-pixel phase --
// the pixel-integrated PRF of a pixel is a BIQUARTIC polynomial of the two phases (phi_x, phi_y): the 13 edge
// weights of an axis are the quartics below.  So
//   A. per fitted star the rectangle of origins (ax0, by0) its cadences visit is found (jitter spans a few knot intervals)
//      and the coefficient table is contracted into the 25 polynomial coefficients K[a][b] of every (star, origin, pixel)
//      item (separable: 13x13 + 5x13 FMAs per column b);
//   B. every cadence evaluates its stars' PRF values by Horner (24 FMAs per star and pixel instead of 169 table reads and
//      ~230 flops) and accumulates the normal equations.
// The arithmetic differs from the direct contraction only by rounding (1e-15 relative).  A target whose stars visit more
// origins than max_origins (pointing excursions) is flagged for the general kernel.
//--------------------------------------------------------------------------------------------------
// (kEdgePoly and axis_phase live in linpsf_dev.h: the non-linear PSF kernel uses the same polynomial form)

struct StarBox { int axmin, axmax, bymin, bymax, jmin, jmax, imin, imax; };

//--------------------------------------------------------------------------------------------------
// Three kernels (round 2; the round-1 kernel kept the 110 KB table in LDS, one 768-thread workgroup per CU, and read the 25
// coefficients of every Horner evaluation from LDS in every lane: 4 SIMDs share one LDS, so the coefficient reads, not the
// FMAs, set its pace -- 24 ms for the C3 batch):
//   plan  per target the boxes of its stars over ALL cadences (table origins visited, pixels that can be inside the
//         cut-off), the number of (pixel, origin) items and their place in the coefficient store (one atomic per target),
//         and the ORDER in which the fit walks the cadences: sorted by the origins of all stars (bitonic sort in LDS), because
//         the jitter straddles a knot boundary in most targets and a wavefront should see one origin per star;
//   coef  the 25 biquartic coefficients of every item, contracted from the target's table staged in LDS (one thread per
//         item, the 13 x 13 patch read once), written to the store: item = (pixel of the star's box, origin);
//   fit   one thread per cadence in 256-thread workgroups that use NO LDS: the coefficients of a (star, pixel, origin) are
//         the same for every cadence of a wavefront that sees that origin, so they are fetched by SCALAR loads from a
//         wave-uniform address (3 x s_load_dwordx16 + 1) and enter the Horner FMAs as SGPR addends (v_fma_f64 with a scalar
//         source) -- no LDS read, no barrier, and the occupancy is set by the registers.  Lanes of a wavefront that still
//         differ in origin are served in turn (ballot loop).  One instantiation per star count (1, 2, 3, 4, 5-8).
// The coefficient arithmetic and the accumulation order (pixels row-major) are those of the round-1 kernel: same results.
// Measured (C3: 10 000 targets, 18 057 fitted stars): plan 0.65 ms, coef 1.23 ms, fit 10.9 ms (24.2 ms in round 1's kernel).
// Also measured: the coefficients by per-lane vector loads of one address instead of scalar loads (17.7 ms: the texture
// addresser handles 64 lanes whatever they read); natural cadence order (18.6 ms: three origins per wavefront on average);
// the cadence's pixels fetched a row ahead through LDS (13.6 against 12.5: the loop is not waiting for its pixels); two or four
// cadences per lane sharing the scalar loads and the uniform tests (10.9 - 11.9 ms for the combinations tried: no gain, the
// extra registers cost what the shared work saves).  Round 3: the cadences sorted by origin only inside windows of 256 / 512
// consecutive cadences (a 128-byte line of a pixel's series is then touched by one workgroup: the PMC passes show 34 GB of
// line fills per step against 12 GB of necessary bytes for the global sort): 14.2 / 12.6 ms against 10.9 -- the extra origins
// per wavefront cost more than the re-fetched lines.
//--------------------------------------------------------------------------------------------------
// a * b + c with c in scalar registers: one VOP3 instruction (left alone the compiler copies a uniform addend into vector
// registers and accumulates with v_fmac)
__device__ __forceinline__ double fma_sgpr_addend(double a, double b, double c) {
	double r;
	asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c));
	return r;
}

// inverse of the reflected Gray code n ^ (n >> 1) on 4 bits: the place of a membership pattern in the order 1,3,2,6,7,5,4,12,...
__device__ __forceinline__ unsigned gray_rank4(unsigned g) { g ^= g >> 2; g ^= g >> 1; return g & 15u; }

// one segment of a target's series: the knot intervals every star visits in it (lo / hi per star and axis; hi < lo: never valid)
__device__ __forceinline__ void emit_segment(SegPlan& g, int target, int t0, int t1, const int (&lo)[kMfmaStars][2], const int (&hi)[kMfmaStars][2],
	int ns, const StarPlan* spl)
{
	g.target = target; g.tile0 = t0; g.tile1 = t1; g.kdoubles = 0; g.koff = 0;
	for (int s = 0; s < kMfmaStars; ++s) {
		const bool any = (s < ns) && (hi[s][0] >= lo[s][0]) && (hi[s][1] >= lo[s][1]) && (spl[s].nc > 0);
		g.axmin[s] = any ? lo[s][0] : 0; g.bymin[s] = any ? lo[s][1] : 0;
		g.na[s] = (uint8_t)(any ? (hi[s][0] - lo[s][0] + 1) : 0); g.nb[s] = (uint8_t)(any ? (hi[s][1] - lo[s][1] + 1) : 0);
		g.ksub[s] = 0;
	}
}

// Decides on the device (the knots live there) whether the uniform-grid forms apply: totals[kTotGeneral] = 1 if not.  The plan
// kernel then does nothing and the host, which reads the totals anyway, sends every target to the general kernels.
__global__ __launch_bounds__(64) void tp_linpsf_grid_kernel(const double* __restrict__ tx, const double* __restrict__ ty, int n, double cutoff, int force,
	unsigned long long* __restrict__ totals)
{
	if (threadIdx.x != 0 || blockIdx.x != 0) return;
	const bool ok = !force && (cutoff <= 5.25) && uniform_grid_ok(tx, n, cutoff) && uniform_grid_ok(ty, n, cutoff);
	if (!ok) totals[kTotGeneral] = 1ull;
}

// totals: kTotPolyItems items (25 doubles each) of the polynomial store; kTotKDoubles doubles of the matrix-core store (laid behind
// it); kTotPolyTargets targets left to the vector-ALU fit; kTotClass0 + c targets of class c of the matrix-core fit (class_lists[c][..])
__global__ __launch_bounds__(256) void tp_linpsf_plan_kernel(FitArgs a, StarPlan* __restrict__ plans, int32_t* __restrict__ todo,
	unsigned long long* __restrict__ totals, int max_origins, int32_t* __restrict__ order, int sort_n,
	MPlan* __restrict__ mplans, uint16_t* __restrict__ ulist, uint8_t* __restrict__ usig, int use_mfma, int32_t* __restrict__ class_lists, int n_targets,
	SegPlan* __restrict__ segs, int32_t* __restrict__ seg_lists)
{
	extern __shared__ unsigned long long skeys[];   // [sort_n] (key of the cadence's origins) * 8192 + cadence, or nothing
	// matrix-core path: the knot intervals every star visits per 16-cadence tile (x lowest / highest, y lowest / highest; the
	// sentinel 32767 / -32768: no valid position in the tile), and the segments the series is cut into
	__shared__ __align__(8) short s_tr[kMfmaStars][kMfmaCadTiles][4];
	__shared__ SegPlan s_seg[kMfmaSegs];
	__shared__ int s_nseg, s_walk, s_too_many;
	__shared__ StarBox sbox[kMaxStars];
	__shared__ StarPlan spl[kMaxStars];
	__shared__ double spos[4][kMfmaStars][4];      // per wavefront and star: min / max of the row and column position
	__shared__ double srange[kMfmaStars][4];
	__shared__ unsigned pkeys[kMfmaPixels];
	__shared__ unsigned s_tiles[kMfmaStars], s_etiles[kMfmaStars];
	__shared__ int s_ok, s_nkeys, s_path;
	__shared__ double kn[160], kny[160];
	const int target = blockIdx.x, tid = threadIdx.x;
	const int n = a.n;
	const int64_t s0 = a.star_offsets[target];
	const int ns = (int)(a.star_offsets[target + 1] - s0);
	int32_t* ord = order + (int64_t)target * a.n_cad;
	if (totals[kTotGeneral] != 0) return;   // tp_linpsf_grid_kernel found a grid / cut-off the uniform forms cannot take: the general kernels fit every target
	if (ns > kMaxStars) return;   // the many-star kernel's targets
	if (tid == 0) { s_ok = 0; s_nkeys = 0; s_path = kPathPoly; s_nseg = 0; }
	const bool want_segments = use_mfma && ns >= 1 && ns <= kMfmaStars;
	for (int i = tid; i < n + 4; i += 256) { kn[i] = a.knots_x[i]; kny[i] = a.knots_y[i]; }
	if (tid < kMaxStars) {
		sbox[tid].axmin = sbox[tid].bymin = sbox[tid].jmin = sbox[tid].imin = 0x7fffffff;
		sbox[tid].axmax = sbox[tid].bymax = sbox[tid].jmax = sbox[tid].imax = -0x7fffffff;
	}
	if (tid < kMfmaStars) { s_tiles[tid] = 0u; s_etiles[tid] = 0u; }
	__syncthreads();
	const double h = kn[5] - kn[4], hy = kny[5] - kny[4];
	const double cutoff = a.cutoff;
	for (int s = 0; s < ns; ++s) {
		const int big = 0x7fffffff;
		int lo[4] = {big, big, big, big}, hi[4] = {-big, -big, -big, -big};
		double pr[4] = {1e300, -1e300, 1e300, -1e300};   // row min, row max, column min, column max over the valid cadences
		// (whole rounds of 256 cadences, so that the 16 lanes of a tile of cadences reduce together)
		for (int k = tid; k < ((a.n_cad + 255) & ~255); k += 256) {
			const bool in_series = k < a.n_cad;
			const double srow = in_series ? a.pos_row[(s0 + s) * a.pos_pitch + k] : __builtin_nan(""), scol = in_series ? a.pos_col[(s0 + s) * a.pos_pitch + k] : __builtin_nan("");
			double phx, phy; int ax0, by0;
			const bool vx = axis_phase(kn, n, scol, h, phx, ax0);
			const bool vy = axis_phase(kny, n, srow, hy, phy, by0);
			if (want_segments && s < kMfmaStars) {
				// consecutive lanes hold consecutive cadences: 16 of them are one tile
				int t0 = (vx && vy) ? ax0 : 32767, t1 = (vx && vy) ? ax0 : -32768, t2 = (vx && vy) ? by0 : 32767, t3 = (vx && vy) ? by0 : -32768;
#pragma unroll
				for (int off = 1; off < 16; off <<= 1) {
					const int o0 = __shfl_xor(t0, off, 64), o1 = __shfl_xor(t1, off, 64), o2 = __shfl_xor(t2, off, 64), o3 = __shfl_xor(t3, off, 64);
					t0 = (o0 < t0) ? o0 : t0; t1 = (o1 > t1) ? o1 : t1; t2 = (o2 < t2) ? o2 : t2; t3 = (o3 > t3) ? o3 : t3;
				}
				if ((tid & 15) == 0 && in_series && (k >> 4) < kMfmaCadTiles) {
					// (an interval index beyond 16 bits -- a position thousands of pixels off -- can only come with others that are
					// not: the span test below then refuses the target; clamping keeps the order)
					auto cl = [](int v) { return (short)((v < -32767) ? -32767 : ((v > 32766) ? 32766 : v)); };
					const bool any = t1 >= t0;
					s_tr[s][k >> 4][0] = any ? cl(t0) : (short)32767; s_tr[s][k >> 4][1] = any ? cl(t1) : (short)-32768;
					s_tr[s][k >> 4][2] = any ? cl(t2) : (short)32767; s_tr[s][k >> 4][3] = any ? cl(t3) : (short)-32768;
				}
			}
			if (vx && vy) {
				const int v0[4] = {ax0, by0, (int)floor(scol - cutoff), (int)floor(srow - cutoff)};
				const int v1[4] = {ax0, by0, (int)ceil(scol + cutoff), (int)ceil(srow + cutoff)};
#pragma unroll
				for (int e = 0; e < 4; ++e) { lo[e] = (v0[e] < lo[e]) ? v0[e] : lo[e]; hi[e] = (v1[e] > hi[e]) ? v1[e] : hi[e]; }
				pr[0] = fmin(pr[0], srow); pr[1] = fmax(pr[1], srow); pr[2] = fmin(pr[2], scol); pr[3] = fmax(pr[3], scol);
			}
		}
#pragma unroll
		for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
			for (int e = 0; e < 4; ++e) {
				const int l2 = __shfl_xor(lo[e], off, 64), h2 = __shfl_xor(hi[e], off, 64);
				lo[e] = (l2 < lo[e]) ? l2 : lo[e];
				hi[e] = (h2 > hi[e]) ? h2 : hi[e];
			}
			pr[0] = fmin(pr[0], __shfl_xor(pr[0], off, 64)); pr[1] = fmax(pr[1], __shfl_xor(pr[1], off, 64));
			pr[2] = fmin(pr[2], __shfl_xor(pr[2], off, 64)); pr[3] = fmax(pr[3], __shfl_xor(pr[3], off, 64));
		}
		if ((tid & 63) == 0) {
			if (hi[0] >= lo[0]) {
				atomicMin(&sbox[s].axmin, lo[0]); atomicMax(&sbox[s].axmax, hi[0]);
				atomicMin(&sbox[s].bymin, lo[1]); atomicMax(&sbox[s].bymax, hi[1]);
				atomicMin(&sbox[s].jmin, lo[2]); atomicMax(&sbox[s].jmax, hi[2]);
				atomicMin(&sbox[s].imin, lo[3]); atomicMax(&sbox[s].imax, hi[3]);
			}
			if (s < kMfmaStars) {
#pragma unroll
				for (int e = 0; e < 4; ++e) spos[tid >> 6][s][e] = pr[e];
			}
		}
	}
	__syncthreads();
	if (tid == 0) {
		bool too_many = false;
		for (int s = 0; s < ns; ++s) {
			StarBox b = sbox[s];
			StarPlan& q = spl[s];
			q.axmin = q.bymin = 0; q.nby = 1; q.nc = 0; q.jmin = q.imin = 0; q.jmax = q.imax = -1; q.item_off = 0;
			if (b.axmax < b.axmin) continue;   // never a valid position: an all-zero column
			if (b.jmin < 0) b.jmin = 0;
			if (b.jmax > a.width - 1) b.jmax = a.width - 1;
			if (b.imin < 0) b.imin = 0;
			if (b.imax > a.height - 1) b.imax = a.height - 1;
			if (b.jmax < b.jmin || b.imax < b.imin) continue;   // never on the stamp
			q.axmin = b.axmin; q.bymin = b.bymin; q.nby = b.bymax - b.bymin + 1; q.nc = (b.axmax - b.axmin + 1) * q.nby;
			q.jmin = b.jmin; q.jmax = b.jmax; q.imin = b.imin; q.imax = b.imax;
			if (q.nc > max_origins) too_many = true;
		}
		for (int s = 0; s < ns && s < kMfmaStars; ++s) {
			srange[s][0] = fmin(fmin(spos[0][s][0], spos[1][s][0]), fmin(spos[2][s][0], spos[3][s][0]));
			srange[s][1] = fmax(fmax(spos[0][s][1], spos[1][s][1]), fmax(spos[2][s][1], spos[3][s][1]));
			srange[s][2] = fmin(fmin(spos[0][s][2], spos[1][s][2]), fmin(spos[2][s][2], spos[3][s][2]));
			srange[s][3] = fmax(fmax(spos[0][s][3], spos[1][s][3]), fmax(spos[2][s][3], spos[3][s][3]));
		}
		// The matrix-core path: the series cut into segments of 16-cadence tiles inside which no star visits more than kMfmaSpan knot
		// intervals per axis (greedy: a segment ends before the tile that would take a star beyond that).  A star that does so inside
		// ONE tile (jitter of a third of a pixel within 16 cadences), more than kMfmaSegs segments, or a series beyond
		// 16 * kMfmaCadTiles cadences leave the target to the vector-ALU kernels.
		// (a target without a fitted star -- its own catalogue entry dropped for a NaN magnitude or position -- has no class list:
		// the polynomial path finalises it as 'All target flux values are NaN')
		bool seg_ok = want_segments && a.height * a.width <= 65535 && a.n_cad <= 16 * kMfmaCadTiles;
		int nseg = 0;
		s_walk = 0;
		if (seg_ok) {
			// the common case needs no walk: no star leaves its three intervals during the whole series (no drift) -- one segment with
			// the boxes found above
			int lo[kMfmaStars][2], hi[kMfmaStars][2];
			for (int s = 0; s < kMfmaStars; ++s) { lo[s][0] = lo[s][1] = 32767; hi[s][0] = hi[s][1] = -32768; }
			bool whole = true;
			for (int s = 0; s < ns; ++s) {
				const StarBox b = sbox[s];
				if (b.axmax < b.axmin) continue;
				if (b.axmax - b.axmin + 1 > kMfmaSpan || b.bymax - b.bymin + 1 > kMfmaSpan || b.axmin < -32000 || b.axmax > 32000 || b.bymin < -32000 || b.bymax > 32000) whole = false;
				lo[s][0] = b.axmin; hi[s][0] = b.axmax; lo[s][1] = b.bymin; hi[s][1] = b.bymax;
			}
			if (whole) { emit_segment(s_seg[0], target, 0, (a.n_cad + 15) >> 4, lo, hi, ns, spl); nseg = 1; }
			else s_walk = 1;   // the first wavefront walks the tiles (below)
		}
		s_nseg = seg_ok ? nseg : 0;
		s_too_many = too_many ? 1 : 0;
	}
	__syncthreads();
	if (s_walk && tid < 64) {
		// Greedy segmentation by one wavefront, a window of 64 tiles at a time: lane j holds the knot intervals of tile pos + j,
		// an inclusive min / max scan gives every lane the range of [segment start, its tile], the first lane whose range goes
		// beyond the span ends the segment before its tile.  (One thread walking the tiles through LDS took 40 us per target.)
		const int ntile = (a.n_cad + 15) >> 4;
		const int lane = tid;
		int nseg = 0, seg_start = 0, pos = 0;
		bool ok = true;
		int clo[kMfmaStars][2], chi[kMfmaStars][2];   // the range of the open segment up to the window (wave-uniform)
#pragma unroll
		for (int s = 0; s < kMfmaStars; ++s) { clo[s][0] = clo[s][1] = 32767; chi[s][0] = chi[s][1] = -32768; }
		while (pos < ntile && ok) {
			const int t = pos + lane;
			const bool valid = t < ntile;
			int pl[kMfmaStars][2], pu[kMfmaStars][2];
			bool fits = true;
#pragma unroll
			for (int s = 0; s < kMfmaStars; ++s) {
#pragma unroll
				for (int e = 0; e < 2; ++e) {
					int l = (valid && s < ns) ? (int)s_tr[s][valid ? t : 0][2 * e] : 32767, u = (valid && s < ns) ? (int)s_tr[s][valid ? t : 0][2 * e + 1] : -32768;
					if (lane == 0) { l = (clo[s][e] < l) ? clo[s][e] : l; u = (chi[s][e] > u) ? chi[s][e] : u; }
#pragma unroll
					for (int off = 1; off < 64; off <<= 1) {
						const int ol = __shfl_up(l, off, 64), ou = __shfl_up(u, off, 64);
						if (lane >= off) { l = (ol < l) ? ol : l; u = (ou > u) ? ou : u; }
					}
					pl[s][e] = l; pu[s][e] = u;
					if (u >= l && u - l + 1 > kMfmaSpan) fits = false;
				}
			}
			const unsigned long long bad = __ballot(valid && !fits);
			if (bad == 0ull) {
				const int lastl = (ntile - 1 - pos < 63) ? (ntile - 1 - pos) : 63;
#pragma unroll
				for (int s = 0; s < kMfmaStars; ++s)
#pragma unroll
					for (int e = 0; e < 2; ++e) { clo[s][e] = __shfl(pl[s][e], lastl, 64); chi[s][e] = __shfl(pu[s][e], lastl, 64); }
				pos += 64;
				continue;
			}
			const int c = __builtin_ctzll(bad);
			if (c == 0 && pos == seg_start) { ok = false; break; }   // one tile of cadences alone goes beyond the span
			int lo[kMfmaStars][2], hi[kMfmaStars][2];
#pragma unroll
			for (int s = 0; s < kMfmaStars; ++s)
#pragma unroll
				for (int e = 0; e < 2; ++e) {
					const int sl = __shfl(pl[s][e], (c > 0) ? (c - 1) : 0, 64), su = __shfl(pu[s][e], (c > 0) ? (c - 1) : 0, 64);
					lo[s][e] = (c > 0) ? sl : clo[s][e]; hi[s][e] = (c > 0) ? su : chi[s][e];
					clo[s][e] = 32767; chi[s][e] = -32768;
				}
			if (nseg >= kMfmaSegs) { ok = false; break; }
			if (lane == 0) emit_segment(s_seg[nseg], target, seg_start, pos + c, lo, hi, ns, spl);
			++nseg;
			seg_start = pos = pos + c;
		}
		if (ok) {
			if (nseg >= kMfmaSegs) ok = false;
			else { if (lane == 0) emit_segment(s_seg[nseg], target, seg_start, ntile, clo, chi, ns, spl); ++nseg; }
		}
		if (lane == 0) s_nseg = ok ? nseg : 0;
	}
	__syncthreads();
	if (tid == 0) {
		if (s_nseg > 0) s_path = kPathMfma;
		else if (s_too_many) s_path = kPathDirect;   // pointing excursions over many knots: the general kernel
	}
	__syncthreads();
	if (s_path == kPathMfma) {
		// the pixels some star can reach at some cadence: nearer than the cut-off to the rectangle its position sweeps
		const int npix = a.height * a.width;
		const double reach = (cutoff + 1e-6) * (cutoff + 1e-6), always = (cutoff - 1e-6) * (cutoff - 1e-6);
		for (int p = tid; p < npix; p += 256) {
			const int i = p / a.width, j = p - i * a.width;
			unsigned sig = 0u, edge = 0u;
			for (int s = 0; s < ns; ++s) {
				if (spl[s].nc <= 0) continue;
				const double dr = fmax(0.0, fmax(srange[s][0] - (double)i, (double)i - srange[s][1]));
				const double dc = fmax(0.0, fmax(srange[s][2] - (double)j, (double)j - srange[s][3]));
				if (dr * dr + dc * dc < reach) {
					sig |= 1u << s;
					// an "edge" pixel is inside the cut-off at some positions of the star and outside at others: the farthest
					// corner of the rectangle the position sweeps is not inside
					const double fr = fmax(fabs((double)i - srange[s][0]), fabs((double)i - srange[s][1]));
					const double fc = fmax(fabs((double)j - srange[s][2]), fabs((double)j - srange[s][3]));
					if (!(fr * fr + fc * fc < always)) edge |= 1u << s;
				}
			}
			if (sig) {
				const int idx = atomicAdd(&s_nkeys, 1);
				// order: membership pattern (Gray rank), interior pixels before edge pixels, raster
				if (idx < kMfmaPixels) pkeys[idx] = (gray_rank4(sig) << 25) | ((edge ? 1u : 0u) << 24) | (edge << 20) | (sig << 16) | (unsigned)p;
			}
		}
		__syncthreads();
		const int nk = s_nkeys;
		if (nk > kMfmaPixels) {   // a large stamp: the vector-ALU kernels
			if (tid == 0) { bool tm = false; for (int s = 0; s < ns; ++s) if (spl[s].nc > max_origins) tm = true; s_path = tm ? kPathDirect : kPathPoly; }
		}
		else {
			uint16_t* ul = ulist + (int64_t)target * kMfmaPixels;
			uint8_t* us = usig + (int64_t)target * kMfmaPixels;
			if (tid < nk) {
				const unsigned key = pkeys[tid];
				int r = 0;
				for (int q = 0; q < nk; ++q) r += (pkeys[q] < key) ? 1 : 0;
				const unsigned sig = (key >> 16) & 15u, edge = (key >> 20) & 15u;
				ul[r] = (uint16_t)(key & 0xffffu);
				us[r] = (uint8_t)(sig | (edge << 4));
				for (int s = 0; s < ns; ++s) {
					if (sig & (1u << s)) atomicOr(&s_tiles[s], 1u << (r >> 4));
					if (edge & (1u << s)) atomicOr(&s_etiles[s], 1u << (r >> 4));
				}
			} else if (tid < kMfmaPixels) { ul[tid] = (uint16_t)0xffffu; us[tid] = (uint8_t)0; }
		}
		__syncthreads();
	}
	if (tid == 0) {
		int path = s_path;
		if (path == kPathMfma) {
			// per segment one spline per star over the knot intervals it visits there (at most 3 x 3), the segment's coefficient image
			// within the LDS of its class; otherwise the vector-ALU kernels take the target
			MPlan mp;
			mp.n_pix = s_nkeys; mp.n_tiles = (s_nkeys + 15) >> 4; mp.n_seg = s_nseg;
			for (int s = 0; s < kMfmaStars; ++s) {
				mp.tiles[s] = (s < ns) ? s_tiles[s] : 0u;
				mp.edge_tiles[s] = (s < ns) ? s_etiles[s] : 0u;
			}
			bool fits = true;
			long long total = 0;
			for (int i = 0; i < s_nseg; ++i) {
				SegPlan& g = s_seg[i];
				long long blocks = 0;
				for (int s = 0; s < kMfmaStars; ++s) {
					g.ksub[s] = (uint16_t)blocks;
					if (g.na[s] > 0) blocks += (long long)__popc(mp.tiles[s]) * mfma_steps(g.na[s], g.nb[s]);
				}
				if (blocks * 512 > ((ns <= 1) ? kMfmaLdsSmall : kMfmaLdsLarge)) fits = false;
				g.kdoubles = (int32_t)(blocks * 64);
				g.koff = total;
				total += blocks * 64;
			}
			if (fits) {
				const long long base = (long long)atomicAdd(&totals[kTotKDoubles], (unsigned long long)total);
				mplans[target] = mp;
				const int cls = ns - 1;   // one launch per star count
				const unsigned long long sat = atomicAdd(&totals[kTotSeg0 + cls], (unsigned long long)s_nseg);
				for (int i = 0; i < s_nseg; ++i) {
					s_seg[i].koff += base;
					segs[(int64_t)target * kMfmaSegs + i] = s_seg[i];
					seg_lists[(int64_t)cls * n_targets * kMfmaSegs + (int64_t)sat + i] = target * kMfmaSegs + i;
				}
				for (int s = 0; s < ns; ++s) plans[(int64_t)target * kMaxStars + s] = spl[s];
				todo[target] = kPathMfma;
				const unsigned long long at = atomicAdd(&totals[kTotClass0 + cls], 1ull);
				class_lists[(int64_t)cls * n_targets + (int64_t)at] = target;
			} else {
				bool too_many = false;
				for (int s = 0; s < ns; ++s) if (spl[s].nc > max_origins) too_many = true;
				path = too_many ? kPathDirect : kPathPoly;
			}
		}
		if (path == kPathDirect) { todo[target] = kPathDirect; atomicAdd(&totals[kTotDirectTargets], 1ull); }
		else if (path == kPathPoly) {
			long long items = 0;
			for (int s = 0; s < ns; ++s) {
				StarPlan& q = spl[s];
				q.item_off = items;
				if (q.nc > 0) items += (long long)q.nc * (q.jmax - q.jmin + 1) * (q.imax - q.imin + 1);
			}
			const long long base = (long long)atomicAdd(&totals[kTotPolyItems], (unsigned long long)items);
			atomicAdd(&totals[kTotPolyTargets], 1ull);
			for (int s = 0; s < ns; ++s) { spl[s].item_off += base; plans[(int64_t)target * kMaxStars + s] = spl[s]; }
			s_ok = 1;
		}
	}
	__syncthreads();
	if (!s_ok) return;
	// the order in which the fit kernel walks the cadences: sorted by the origins of all stars, so that the 64 cadences of a
	// wavefront share their polynomial coefficients (the jitter straddles a knot boundary in most targets)
	if (sort_n <= 0) { for (int k = tid; k < a.n_cad; k += 256) ord[k] = k; return; }
	for (int k = tid; k < sort_n; k += 256) {
		unsigned long long key = ~0ull;
		if (k < a.n_cad) {
			key = 0;
			for (int s = 0; s < ns; ++s) {
				const StarPlan q = spl[s];
				double phx, phy; int ax0, by0;
				const bool vx = axis_phase(kn, n, a.pos_col[(s0 + s) * a.pos_pitch + k], h, phx, ax0);
				const bool vy = axis_phase(kny, n, a.pos_row[(s0 + s) * a.pos_pitch + k], hy, phy, by0);
				const int cc = (vx && vy && q.nc > 0) ? ((ax0 - q.axmin) * q.nby + (by0 - q.bymin)) : 0;
				key = key * (unsigned long long)(max_origins + 1) + (unsigned long long)cc;
			}
			key = key * 8192ull + (unsigned long long)k;
		}
		skeys[k] = key;
	}
	__syncthreads();
	for (int size = 2; size <= sort_n; size <<= 1) {
		for (int stride = size >> 1; stride > 0; stride >>= 1) {
			for (int t = tid; t < sort_n / 2; t += 256) {
				const int lo = ((t / stride) * (stride << 1)) + (t % stride), hi = lo + stride;
				const bool up = ((lo & size) == 0);
				const unsigned long long x = skeys[lo], y = skeys[hi];
				if ((x > y) == up) { skeys[lo] = y; skeys[hi] = x; }
			}
			__syncthreads();
		}
	}
	for (int k = tid; k < a.n_cad; k += 256) ord[k] = (int)(skeys[k] & 8191ull);
}

// the 25 coefficients (times h2) of the 13 x 13 table patch at (ax, by): kk[e][b], e = power of phi_x, b = power of phi_y
__device__ __forceinline__ void patch_coefficients(const double* __restrict__ C, int n, int ax, int by, double h2, double (&kk)[5][5])
{
#pragma unroll
	for (int e = 0; e < 5; ++e)
#pragma unroll
		for (int bcol = 0; bcol < 5; ++bcol) kk[e][bcol] = 0.0;
	const double* c0 = C + (int64_t)ax * n + by;
#pragma unroll 1
	for (int pp = 0; pp < 13; ++pp) {
		const double* r = c0 + pp * n;
		double rv[13];
#pragma unroll
		for (int q = 0; q < 13; ++q) rv[q] = r[q];
		const double e0 = kEdgePoly[pp][0], e1 = kEdgePoly[pp][1], e2 = kEdgePoly[pp][2], e3 = kEdgePoly[pp][3], e4 = kEdgePoly[pp][4];
#pragma unroll
		for (int bcol = 0; bcol < 5; ++bcol) {
			double t = 0.0;
#pragma unroll
			for (int q = 0; q < 13; ++q) t = __builtin_fma(kEdgePoly[q][bcol], rv[q], t);
			kk[0][bcol] = __builtin_fma(e0, t, kk[0][bcol]);
			kk[1][bcol] = __builtin_fma(e1, t, kk[1][bcol]);
			kk[2][bcol] = __builtin_fma(e2, t, kk[2][bcol]);
			kk[3][bcol] = __builtin_fma(e3, t, kk[3][bcol]);
			kk[4][bcol] = __builtin_fma(e4, t, kk[4][bcol]);
		}
	}
#pragma unroll
	for (int e = 0; e < 5; ++e)
#pragma unroll
		for (int bcol = 0; bcol < 5; ++bcol) kk[e][bcol] *= h2;
}

// The same contraction for the `na` consecutive intervals (ax, by), (ax + 1, by) .. along x at once: their 13 x 13 patches are
// 13 + na - 1 table rows, and the inner sums t = sum_q E[q][b] C[row][by + q] of a row serve every interval that holds the row
// (2 340 -> 1 560 multiply-adds for two intervals, 3 510 -> 2 100 for three).  Every kk[ca] gets exactly the operations
// patch_coefficients gives it, in the same order: bit-identical.  Rows ax .. ax + 12 + na - 1 must lie inside the table.
__device__ __forceinline__ void patch_coefficients_along_x(const double* __restrict__ C, int n, int ax, int by, double h2, int na, double (&kk)[3][5][5])
{
#pragma unroll
	for (int ca = 0; ca < 3; ++ca)
#pragma unroll
		for (int e = 0; e < 5; ++e)
#pragma unroll
			for (int bcol = 0; bcol < 5; ++bcol) kk[ca][e][bcol] = 0.0;
	const double* c0 = C + (int64_t)ax * n + by;
#pragma unroll 1
	for (int row = 0; row < 12 + na; ++row) {
		const double* r = c0 + row * n;
		double rv[13], t[5];
#pragma unroll
		for (int q = 0; q < 13; ++q) rv[q] = r[q];
#pragma unroll
		for (int bcol = 0; bcol < 5; ++bcol) {
			double v = 0.0;
#pragma unroll
			for (int q = 0; q < 13; ++q) v = __builtin_fma(kEdgePoly[q][bcol], rv[q], v);
			t[bcol] = v;
		}
#pragma unroll
		for (int ca = 0; ca < 3; ++ca) {
			const int pp = row - ca;
			if (ca < na && pp >= 0 && pp < 13) {   // uniform
				const double e0 = kEdgePoly[pp][0], e1 = kEdgePoly[pp][1], e2 = kEdgePoly[pp][2], e3 = kEdgePoly[pp][3], e4 = kEdgePoly[pp][4];
#pragma unroll
				for (int bcol = 0; bcol < 5; ++bcol) {
					kk[ca][0][bcol] = __builtin_fma(e0, t[bcol], kk[ca][0][bcol]);
					kk[ca][1][bcol] = __builtin_fma(e1, t[bcol], kk[ca][1][bcol]);
					kk[ca][2][bcol] = __builtin_fma(e2, t[bcol], kk[ca][2][bcol]);
					kk[ca][3][bcol] = __builtin_fma(e3, t[bcol], kk[ca][3][bcol]);
					kk[ca][4][bcol] = __builtin_fma(e4, t[bcol], kk[ca][4][bcol]);
				}
			}
		}
	}
#pragma unroll
	for (int ca = 0; ca < 3; ++ca)
#pragma unroll
		for (int e = 0; e < 5; ++e)
#pragma unroll
			for (int bcol = 0; bcol < 5; ++bcol) kk[ca][e][bcol] *= h2;
}

constexpr int kCoefThreads = 512;
__global__ __launch_bounds__(kCoefThreads) void tp_linpsf_coef_kernel(FitArgs a, const StarPlan* __restrict__ plans, const int32_t* __restrict__ todo,
	double* __restrict__ store, const MPlan* __restrict__ mplans, const uint16_t* __restrict__ ulist, const uint8_t* __restrict__ usig,
	double* __restrict__ kstore, const SegPlan* __restrict__ segs)
{
	extern __shared__ __align__(16) double ctab[];   // the target's coefficient table [n*n]: every patch is read ~5 times over
	const int target = blockIdx.x, tid = threadIdx.x;
	const int path = todo[target];
	if (path == kPathDirect) return;
	const int ns = (int)(a.star_offsets[target + 1] - a.star_offsets[target]);
	if (ns > kMaxStars) return;
	const int n = a.n;
	const double h2 = (a.knots_x[5] - a.knots_x[4]) * (a.knots_y[5] - a.knots_y[4]);
	{
		// the whole table in flight at once (up to 39 doubles per thread for the largest table admitted), then into LDS: one round
		// trip to memory instead of one per slice.  (A workgroup per CU that walks the targets with the next table on its way in
		// registers while this one's patches are contracted: 0.85 against 0.76 ms -- 78 more registers, and the targets' work differs.)
		const double* cg = a.coef + (int64_t)target * n * n;
		constexpr int kPer = (140 * 140 + kCoefThreads - 1) / kCoefThreads;
		double tmp[kPer];
#pragma unroll
		for (int u = 0; u < kPer; ++u) { const int i = u * kCoefThreads + tid; tmp[u] = (i < n * n) ? cg[i] : 0.0; }
#pragma unroll
		for (int u = 0; u < kPer; ++u) { const int i = u * kCoefThreads + tid; if (i < n * n) ctab[i] = tmp[u]; }
	}
	__syncthreads();
	const double* C = ctab;
	if (path == kPathMfma) {
		// matrix-core layout (linpsf_mfma.hip): per (star, tile of the star) the A operands of the MFMA steps, lane = (group g,
		// pixel u of the tile).  The coefficients are those of the tensor-product quartic spline over the na x nb knot intervals
		// the star visits, in the basis {1, X, X^2, X^3, X^4, (X-1)+^4, (X-2)+^4} x {the same in Y}: ce[e][d] (e, d <= 4) is the
		// biquartic of interval (0, 0); a quartic spline changes only its leading coefficient at a knot, so the coefficient of
		// (X-a)+^4 Y^d is K(a,0)[4][d] - K(a-1,0)[4][d], of X^e (Y-b)+^4 it is K(0,b)[e][4] - K(0,b-1)[e][4], and of (X-a)+^4 (Y-b)+^4
		// the second difference of K[4][4] -- every interval's 13 x 13 patch is contracted as for the vector-ALU path.
		// Pixels of the tile the star never reaches get zeros.
		const MPlan mp = mplans[target];
		const uint16_t* ul = ulist + (int64_t)target * kMfmaPixels;
		const uint8_t* us = usig + (int64_t)target * kMfmaPixels;
		// One thread per (pixel of a tile, knot interval): the 13 x 13 patch of that interval is contracted into its 25
		// coefficients; the interval (0, 0) writes the steps that hold C[e][d], e <= 4, d < 4, at once, every interval leaves its
		// K[4][0..4] and K[0..3][4] in LDS, and one thread per pixel then forms the differences and writes the remaining steps.
		// Jobs: one per (segment, star that is on the stamp in it).  A LANE takes one pixel of the star's tiles and one interval
		// along y, and all na intervals along x (patch_coefficients_along_x); the nb lanes of a pixel are neighbours, so the
		// differences across y come from the lane below by one shuffle and those across x are the lane's own -- nothing goes
		// through LDS, and after the table is staged no wavefront waits for another: each takes every (waves)-th unit of 64 / nb
		// pixels of the job list.  (One thread per (pixel, interval) with the differences formed through LDS between two barriers
		// per round of 512 threads, a round per star and segment: 1.09 ms per 10 000 targets, 4.4 ms on the drift scene.)
		struct Job { int na, nb, nt, axmin, bymin, s; unsigned tiles; long long dst; };
		__shared__ Job s_job[kMfmaSegs * kMfmaStars];
		__shared__ int s_njobs;
		if (tid == 0) {
			int nj = 0;
			for (int sgi = 0; sgi < mp.n_seg; ++sgi) {
				const SegPlan sg = segs[(int64_t)target * kMfmaSegs + sgi];
				for (int s = 0; s < ns; ++s) {
					if (sg.na[s] == 0) continue;
					Job j;
					j.na = sg.na[s]; j.nb = sg.nb[s]; j.nt = __popc(mp.tiles[s]); j.axmin = sg.axmin[s]; j.bymin = sg.bymin[s]; j.s = s;
					j.tiles = mp.tiles[s]; j.dst = sg.koff + (long long)sg.ksub[s] * 64;
					s_job[nj++] = j;
				}
			}
			s_njobs = nj;
		}
		__syncthreads();
		const int njobs = s_njobs;
		const int lane = tid & 63, wave = tid >> 6, nwaves = (int)blockDim.x >> 6;
		int unit = 0;                              // units of the job list passed so far (uniform)
		for (int jb = 0; jb < njobs; ++jb) {
			const Job jq = s_job[jb];
			const int na = jq.na, nb = jq.nb, nk = mfma_steps(na, nb);
			const int per = 64 / nb, nitems = jq.nt * 16;
			const int nunits = (nitems + per - 1) / per;
			for (int c = 0; c < nunits; ++c, ++unit) {
				if (unit % nwaves != wave) continue;
				const int li = lane / nb, cb = lane - li * nb;
				const int item = c * per + li;
				const bool mine = li < per && item < nitems;
				double kk[3][5][5];
				int r = 0, u = 0;
				bool reach = false;
				int ax = 0, by = 0;
				if (mine) {
					r = item >> 4; u = item & 15;
					unsigned m = jq.tiles;
					for (int q = 0; q < r; ++q) m &= m - 1;          // drop the r lowest set bits
					const int slot = (__ffs(m) - 1) * 16 + u;
					const unsigned pix = ul[slot];
					if (pix != 0xffffu && ((us[slot] >> jq.s) & 1)) {
						const int i = (int)pix / a.width, j = (int)pix - i * a.width;
						ax = jq.axmin + 9 * j; by = (jq.bymin + cb) + 9 * i;
						by = by < 0 ? 0 : (by > n - 13 ? n - 13 : by);
						reach = true;
					}
				}
				if (reach && ax >= 0 && ax + na - 1 <= n - 13) {
					patch_coefficients_along_x(C, n, ax, by, h2, na, kk);
				} else {
#pragma unroll
					for (int ca = 0; ca < 3; ++ca) {
						if (reach && ca < na) {                  // an interval beyond the table's edge: clamped one by one
							int axc = ax + ca;
							axc = axc < 0 ? 0 : (axc > n - 13 ? n - 13 : axc);
							patch_coefficients(C, n, axc, by, h2, kk[ca]);
						} else {
#pragma unroll
							for (int e = 0; e < 5; ++e)
#pragma unroll
								for (int d = 0; d < 5; ++d) kk[ca][e][d] = 0.0;
						}
					}
				}
				// what the lane below (same pixel, interval cb - 1) holds of K[0][0..3][4] and K[ca][4][4]; zero below interval 0
				double lo_e4[4], lo_44[3];
#pragma unroll
				for (int e = 0; e < 4; ++e) { const double v = __shfl_up(kk[0][e][4], 1, 64); lo_e4[e] = (cb > 0) ? v : 0.0; }
#pragma unroll
				for (int ca = 0; ca < 3; ++ca) { const double v = __shfl_up(kk[ca][4][4], 1, 64); lo_44[ca] = (cb > 0) ? v : 0.0; }
				if (!mine) continue;
				double* dst = kstore + jq.dst + (int64_t)r * nk * 64 + u;
				// ce[4 + a][d] (d < 4), ce[e][4 + b] (e < 4), ce[4 + a][4 + b]: first differences along the axis that leaves interval 0,
				// the second difference of K[4][4] off both axes (operations and their order as in the LDS version)
				auto corner = [&](int ca) -> double {
					const double here = kk[ca][4][4], left = (ca > 0) ? kk[ca > 0 ? ca - 1 : 0][4][4] : 0.0;
					const double below = lo_44[ca], diag = (ca > 0) ? lo_44[ca > 0 ? ca - 1 : 0] : 0.0;
					return ((here - left) - below) + diag;
				};
				if (cb == 0) {
#pragma unroll
					for (int e = 0; e < 5; ++e)
#pragma unroll
						for (int g = 0; g < 4; ++g) dst[e * 64 + g * 16] = kk[0][e][g];
				}
				if (mfma_is22(na, nb)) {
					// 9 steps: y basis 4 with x basis 0..3; {x basis 4, 5 with y basis 4, x basis 0, 1 with y basis 5}; x basis 5 with y
					// basis 0..3; x basis 2..5 with y basis 5
					if (cb == 0) {
#pragma unroll
						for (int g = 0; g < 4; ++g) {
							dst[5 * 64 + g * 16] = kk[0][g][4] - 0.0;
							dst[7 * 64 + g * 16] = kk[1][4][g] - kk[0][4][g];
						}
						dst[6 * 64 + 0 * 16] = corner(0);
						dst[6 * 64 + 1 * 16] = corner(1);
					} else {
						dst[6 * 64 + 2 * 16] = kk[0][0][4] - lo_e4[0];
						dst[6 * 64 + 3 * 16] = kk[0][1][4] - lo_e4[1];
						dst[8 * 64 + 0 * 16] = kk[0][2][4] - lo_e4[2];
						dst[8 * 64 + 1 * 16] = kk[0][3][4] - lo_e4[3];
						dst[8 * 64 + 2 * 16] = corner(0);
						dst[8 * 64 + 3 * 16] = corner(1);
					}
				} else {
					// steps: 5, 6 for y interval 0; 7 .. for the x basis functions 5, 6; then two per further y interval
					const int ystep = 5 + 2 * cb + ((cb >= 1) ? (na - 1) : 0);
#pragma unroll
					for (int g = 0; g < 4; ++g) {
						dst[ystep * 64 + g * 16] = kk[0][g][4] - lo_e4[g];
						double cv = 0.0;
						if (g == 0) cv = corner(0);
						else if (g == 1 && na > 1) cv = corner(1);
						else if (g == 2 && na > 2) cv = corner(2);
						dst[(ystep + 1) * 64 + g * 16] = cv;
					}
					if (cb == 0) {
#pragma unroll
						for (int ca = 1; ca < 3; ++ca) {
							if (ca < na) {
#pragma unroll
								for (int g = 0; g < 4; ++g) dst[(6 + ca) * 64 + g * 16] = kk[ca][4][g] - kk[ca - 1][4][g];
							}
						}
					}
				}
			}
		}
		return;
	}
	for (int s = 0; s < ns; ++s) {
		const StarPlan p = plans[(int64_t)target * kMaxStars + s];
		const int ncols = p.jmax - p.jmin + 1, nrows = p.imax - p.imin + 1;
		if (p.nc <= 0 || ncols <= 0 || nrows <= 0) continue;
		const int nitems = p.nc * ncols * nrows;
		// one thread per item: the 13 x 13 patch of the table is read once and contracted into all 25 coefficients (the sums
		// run over q inside, over p outside)
		for (int item = tid; item < nitems; item += kCoefThreads) {
			const int pix = item / p.nc, co = item - pix * p.nc;
			const int ii = pix / ncols, jj = pix - ii * ncols;
			const int cx = co / p.nby, cy = co - cx * p.nby;
			int ax = (p.axmin + cx) + 9 * (p.jmin + jj), by = (p.bymin + cy) + 9 * (p.imin + ii);
			ax = ax < 0 ? 0 : (ax > n - 13 ? n - 13 : ax);
			by = by < 0 ? 0 : (by > n - 13 ? n - 13 : by);
			double kk[5][5];
			patch_coefficients(C, n, ax, by, h2, kk);
			double* dst = store + (p.item_off + item) * 25;
#pragma unroll
			for (int e = 0; e < 5; ++e)
#pragma unroll
				for (int bcol = 0; bcol < 5; ++bcol) dst[e * 5 + bcol] = kk[e][bcol];
		}
	}
}

template <int S, int SLO>
__global__ __launch_bounds__(256) void tp_linpsf_fit2_kernel(FitArgs a, const StarPlan* __restrict__ plans, const int32_t* __restrict__ todo,
	const double* __restrict__ store, const int32_t* __restrict__ order)
{
	const int target = blockIdx.x;
	const int64_t s0 = a.star_offsets[target];
	int ns = (int)(a.star_offsets[target + 1] - s0);
	if (ns < SLO || ns > S) return;   // another instantiation's targets
	if (todo[target] != kPathPoly) return;   // the general / the matrix-core kernel's
	const int tid = threadIdx.x;
	const int slot = blockIdx.y * blockDim.x + tid;   // position in the origin-sorted order of the target's cadences
	const bool active = slot < a.n_cad;
	const int kreal = order[(int64_t)target * a.n_cad + (active ? slot : (a.n_cad - 1))];
	const int k = kreal;
	const int n = a.n;
	const int H = a.height, W = a.width;
	const double h = a.knots_x[5] - a.knots_x[4], hy = a.knots_y[5] - a.knots_y[4];
	const double cutoff = a.cutoff, c2 = cutoff * cutoff;

	double phx[S], phy[S], srow[S], scol[S];
	int cc[S];
	bool valid[S];
	int ncs[S], ncols[S], jmin[S], jmax[S], imin[S], imax[S];
	long long ioff[S];
	int ui0 = H, ui1 = -1, uj0 = W, uj1 = -1;
#pragma unroll
	for (int s = 0; s < S; ++s) {
		valid[s] = false; phx[s] = phy[s] = 0.0; srow[s] = scol[s] = 0.0; cc[s] = 0;
		ncs[s] = 0; ncols[s] = 0; jmin[s] = imin[s] = 0; jmax[s] = imax[s] = -1; ioff[s] = 0;
		if (s < ns) {
			const StarPlan p = plans[(int64_t)target * kMaxStars + s];
			ncs[s] = p.nc; ncols[s] = p.jmax - p.jmin + 1; jmin[s] = p.jmin; jmax[s] = p.jmax; imin[s] = p.imin; imax[s] = p.imax; ioff[s] = p.item_off;
			srow[s] = a.pos_row[(s0 + s) * a.pos_pitch + k];
			scol[s] = a.pos_col[(s0 + s) * a.pos_pitch + k];
			int ax0, by0;
			// x <-> column (first spline axis), y <-> row  (psf.py:146)
			const bool vx = axis_phase(a.knots_x, n, scol[s], h, phx[s], ax0);
			const bool vy = axis_phase(a.knots_y, n, srow[s], hy, phy[s], by0);
			valid[s] = vx && vy && (p.nc > 0);
			cc[s] = valid[s] ? ((ax0 - p.axmin) * p.nby + (by0 - p.bymin)) : 0;
			if (p.nc > 0 && p.jmax >= p.jmin && p.imax >= p.imin) {
				ui0 = (p.imin < ui0) ? p.imin : ui0; ui1 = (p.imax > ui1) ? p.imax : ui1;
				uj0 = (p.jmin < uj0) ? p.jmin : uj0; uj1 = (p.jmax > uj1) ? p.jmax : uj1;
			}
		}
	}
	double G[S][S], g[S];
#pragma unroll
	for (int s = 0; s < S; ++s) { g[s] = 0.0;
#pragma unroll
		for (int t = 0; t < S; ++t) G[s][t] = 0.0; }
	const float* img = a.images + (int64_t)target * H * W * a.t_pitch + k;
	const float sub = a.subtract ? a.subtract[(int64_t)target * a.subtract_pitch + k] : 0.f;

	// item indices relative to the target's first item: 32-bit scalar arithmetic in the pixel loop
	const double* __restrict__ tstore = store + ioff[0] * 25;
	int rel[S];
#pragma unroll
	for (int s = 0; s < S; ++s) rel[s] = (int)(ioff[s] - ioff[0]);
	for (int i = ui0; i <= ui1; ++i) {
		// per star the columns of this row that are inside the cut-off for SOME cadence of the wavefront (a float bound with a
		// margin; the exact test of psf.py:142 stays in the loop), and where the row's items start in the store
		int ja[S], jb[S], rowoff[S];
		double dr2[S];
		int jfirst = W, jend = 0;
#pragma unroll
		for (int s = 0; s < S; ++s) {
			ja[s] = 0; jb[s] = -1; rowoff[s] = 0;
			const double dr = (double)i - srow[s];
			dr2[s] = dr * dr;
			if (s < ns && i >= imin[s] && i <= imax[s] && jmax[s] >= jmin[s]) {   // uniform
				const float w2 = (float)(c2 - dr2[s]);
				int jl = 0x3fffffff, jh = -0x3fffffff;
				if (valid[s] && w2 > -1e-3f) {
					const float w = sqrtf(fmaxf(w2, 0.f)) * 1.0001f + 1e-3f;
					jl = (int)floorf((float)scol[s] - w);
					jh = (int)ceilf((float)scol[s] + w);
				}
#pragma unroll
				for (int off = 32; off > 0; off >>= 1) {
					const int l2 = __shfl_xor(jl, off, 64), h2 = __shfl_xor(jh, off, 64);
					jl = (l2 < jl) ? l2 : jl;
					jh = (h2 > jh) ? h2 : jh;
				}
				jl = __builtin_amdgcn_readfirstlane(jl); jh = __builtin_amdgcn_readfirstlane(jh);
				ja[s] = (jl > jmin[s]) ? jl : jmin[s];
				jb[s] = (jh < jmax[s]) ? jh : jmax[s];
				rowoff[s] = rel[s] + ((i - imin[s]) * ncols[s] - jmin[s]) * ncs[s];
				if (jb[s] >= ja[s]) { jfirst = (ja[s] < jfirst) ? ja[s] : jfirst; jend = (jb[s] + 1 > jend) ? (jb[s] + 1) : jend; }
			}
		}
		if (jend <= jfirst) continue;
		auto pix_load = [&](int j) { j = (j < jend) ? j : (jend - 1); return img[((int64_t)(i * W) + j) * a.t_pitch]; };
		float pnext = pix_load(jfirst);
#pragma unroll 1
		for (int j = jfirst; j < jend; ++j) {
			const float pv = pnext;
			pnext = pix_load(j + 1);
			float bf = pv;
			if (a.subtract) bf = bf - sub;
			const bool fin = active && (fabsf(bf) <= 3.402823466e+38f);   // good_pixels = isfinite(img) (linpsf_photometry.py:123)
			const double b = (double)bf;
			double av[S];
#pragma unroll
			for (int s = 0; s < S; ++s) {
				av[s] = 0.0;
				if (j >= ja[s] && j <= jb[s]) {   // uniform
					const double dc = (double)j - scol[s];
					// psf.py:142  sqrt((j-col)^2 + (i-row)^2) < cutoff_radius; the squares decide unless they are within
					// rounding of each other (then, for the whole wavefront, the reference's own expression does)
					const double d2 = dc * dc + dr2[s];
					bool inside = d2 < c2;
					if (__any(fabs(d2 - c2) <= 1e-9 * c2)) inside = (fabs(d2 - c2) > 1e-9 * c2) ? (d2 < c2) : (sqrt(d2) < cutoff);
					const bool want = fin && valid[s] && inside;
					const int ibase = rowoff[s] + j * ncs[s];
					unsigned long long mask = __ballot(want);
					while (mask) {
						const int leader = __builtin_ctzll(mask);
						const int ccu = __builtin_amdgcn_readlane(cc[s], leader);
						const bool mine = want && (cc[s] == ccu);
						// wave-uniform address: scalar loads; the coefficients are the SGPR addends of the Horner FMAs
						const double* __restrict__ kp = tstore + (unsigned)(ibase + ccu) * 25u;
						double kc[25];
#pragma unroll
						for (int q = 0; q < 25; ++q) kc[q] = kp[q];
						double val = 0.0;
#pragma unroll
						for (int e = 4; e >= 0; --e) {
							double inner = kc[e * 5 + 4];
#pragma unroll
							for (int d = 3; d >= 0; --d) inner = fma_sgpr_addend(inner, phy[s], kc[e * 5 + d]);
							val = __builtin_fma(val, phx[s], inner);
						}
						if (mine) av[s] = val;
						mask &= ~__ballot(mine);
					}
				}
			}
			if (fin) {
#pragma unroll
				for (int s = 0; s < S; ++s) {
					g[s] += av[s] * b;
#pragma unroll
					for (int t = 0; t < S; ++t) if (t >= s) G[s][t] += av[s] * av[t];
				}
			}
		}
	}
	if (!active) return;
#pragma unroll
	for (int s = 0; s < S; ++s)
#pragma unroll
		for (int t = 0; t < S; ++t) if (t < s) G[s][t] = G[t][s];
	double x[S];
	pinv_solve<S>(G, g, ns, x);
	const int ti = a.target_index[target];
	double tf = __builtin_nan("");
#pragma unroll
	for (int s = 0; s < S; ++s) {
		if (s < ns) {
			a.fluxes_all[(s0 + s) * a.out_pitch + kreal] = x[s];
			if (s == ti) tf = x[s];
		}
	}
	a.flux[(int64_t)target * a.out_pitch + kreal] = tf;
	a.flux_err[(int64_t)target * a.out_pitch + kreal] = __builtin_nan("");
}

// Finalise (linpsf_photometry.py:197-219): mean fitted fluxes over the cadences with a valid target
// flux, contamination from the design matrix of the LAST cadence, status.
struct FinArgs {
	FitArgs f;
	double* contamination; int32_t* status; double* fluxes_mean;
	const int32_t* todo;   // targets marked kPathMfma are finalised by tp_linpsf_finalize_m_kernel (nullptr: none are)
};

template <int S, int SLO>
__global__ __launch_bounds__(256) void tp_linpsf_finalize_kernel(FinArgs fa)
{
	{ const int nst = (int)(fa.f.star_offsets[blockIdx.x + 1] - fa.f.star_offsets[blockIdx.x]); if (nst < SLO || nst > S) return; } // another instantiation's targets
	if (fa.todo && fa.todo[blockIdx.x] == kPathMfma) return;
	extern __shared__ __align__(16) double lds[];
	const FitArgs& a = fa.f;
	const int target = blockIdx.x;
	const int tid = threadIdx.x;
	const int n = a.n;
	// Only the design matrix of the LAST cadence is needed (about 140 star-pixel values x 169 table entries): the table is
	// read straight from HBM / L2 instead of being staged (110 KB of LDS would allow a single workgroup per CU).
	const double* C = a.coef + (int64_t)target * n * n;
	double* kn = lds;
	double* kny = kn + n + 4;
	double* red = kny + n + 4;           // [256]
	for (int i = tid; i < n + 4; i += blockDim.x) { kn[i] = a.knots_x[i]; kny[i] = a.knots_y[i]; }
	__syncthreads();
	const int64_t s0 = a.star_offsets[target];
	int ns = (int)(a.star_offsets[target + 1] - s0);
	if (ns > S) ns = S;
	const int ti = a.target_index[target];
	const double* ftar = a.flux + (int64_t)target * a.out_pitch;

	// count of valid cadences and per-star flux sums (only over cadences whose fit succeeded): one pass, every thread its
	// cadences in order, then a fixed tree (lanes, then the wavefronts in order)
	double mean[S];
	double part[S + 1];
#pragma unroll
	for (int u = 0; u <= S; ++u) part[u] = 0.0;
	for (int k = tid; k < a.n_cad; k += blockDim.x) {
		const bool ok = ftar[k] == ftar[k];
		part[0] += ok ? 1.0 : 0.0;
#pragma unroll
		for (int s = 0; s < S; ++s) if (s < ns) { const double v = a.fluxes_all[(s0 + s) * a.out_pitch + k]; part[1 + s] += ok ? v : 0.0; }
	}
#pragma unroll
	for (int u = 0; u <= S; ++u) {
#pragma unroll
		for (int off = 32; off > 0; off >>= 1) part[u] += __shfl_xor(part[u], off, 64);
		if ((tid & 63) == 0) red[(tid >> 6) * (S + 1) + u] = part[u];
	}
	__syncthreads();
	double cntd = 0.0;
	{
		const int nw = (int)blockDim.x >> 6;
		for (int w = 0; w < nw; ++w) cntd += red[w * (S + 1)];
#pragma unroll
		for (int s = 0; s < S; ++s) {
			double tot = 0.0;
			for (int w = 0; w < nw; ++w) tot += red[w * (S + 1) + 1 + s];
			mean[s] = tot / cntd;
		}
	}
	__syncthreads();
	if (cntd == 0.0) { // allnan(flux) -> ERROR (linpsf_photometry.py:198-200)
		if (tid == 0) { fa.status[target] = TP_STATUS_ERROR; fa.contamination[target] = __builtin_nan(""); }
		return;
	}
	// contamination = sum_p (A[p, others] . mean[others]) * A[p, target] / mean[target], A of the last cadence
	const int k = a.n_cad - 1;
	const int H = a.height, W = a.width;
	const double h = kn[5] - kn[4], hy = kny[5] - kny[4], h2 = h * hy;
	double mx[S][4], my[S][4], srow[S], scol[S];
	int ax0[S], by0[S];
#pragma unroll
	for (int s = 0; s < S; ++s) {
		if (s < ns) {
			srow[s] = a.pos_row[(s0 + s) * a.pos_pitch + k];
			scol[s] = a.pos_col[(s0 + s) * a.pos_pitch + k];
			axis_weights(kn, n, scol[s], h, mx[s], ax0[s]);
			axis_weights(kny, n, srow[s], hy, my[s], by0[s]);
		} else { srow[s] = scol[s] = 0.0; ax0[s] = by0[s] = 4;
#pragma unroll
			for (int q = 0; q < 4; ++q) { mx[s][q] = 0.0; my[s][q] = 0.0; } }
	}
	const float* img = a.images + (int64_t)target * H * W * a.t_pitch + k;
	const float sub = a.subtract ? a.subtract[(int64_t)target * a.subtract_pitch + k] : 0.f;
	double acc = 0.0;
	for (int p = tid; p < H * W; p += blockDim.x) {
		const int i = p / W, j = p - i * W;
		float bf = img[(int64_t)p * a.t_pitch];
		if (a.subtract) bf = bf - sub;
		if (!(fabsf(bf) <= 3.402823466e+38f)) continue;
		double others = 0.0, at = 0.0;
#pragma unroll
		for (int s = 0; s < S; ++s) {
			if (s >= ns) continue;
			const double dc = (double)j - scol[s], dr = (double)i - srow[s];
			double v = 0.0;
			if (sqrt(dc * dc + dr * dr) < a.cutoff) {
				int ax = ax0[s] + 9 * j, by = by0[s] + 9 * i;
				ax = ax < 0 ? 0 : (ax > n - 13 ? n - 13 : ax);
				by = by < 0 ? 0 : (by > n - 13 ? n - 13 : by);
				v = h2 * prf_pixel(C, n, ax, by, mx[s], my[s]);
			}
			if (s == ti) at = v; else others += v * mean[s];
		}
		acc += others * at;
	}
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
	if ((tid & 63) == 0) red[tid >> 6] = acc;
	__syncthreads();
	if (tid == 0) {
		double tot = 0.0;
		for (int w = 0; w < ((int)blockDim.x >> 6); ++w) tot += red[w];
		double mt = 0.0;
#pragma unroll
		for (int u = 0; u < S; ++u) if (u == ti) mt = mean[u];
		const double cont = tot / mt;
		fa.contamination[target] = cont;
		fa.status[target] = (cont > 0.1) ? TP_STATUS_WARNING : TP_STATUS_OK; // :214-219
		if (fa.fluxes_mean) {
#pragma unroll
			for (int u = 0; u < S; ++u) if (u < ns) fa.fluxes_mean[s0 + u] = mean[u];
		}
	}
}


// Finalise for the targets of the matrix-core fit: the same rules, with the design matrix of the last cadence as the fit
// kernel left it (alast[target][star][pixel of the list U], zero outside the cut-off and where the pixel is not finite) instead
// of a second evaluation of the PRF.
template <int S>
__global__ __launch_bounds__(256) void tp_linpsf_finalize_m_kernel(FinArgs fa, const int32_t* __restrict__ targets, const MPlan* __restrict__ mplans,
	const double* __restrict__ alast)
{
	__shared__ double red[4 * (S + 1)];
	const FitArgs& a = fa.f;
	const int target = targets[blockIdx.x];
	const int tid = threadIdx.x;
	const int64_t s0 = a.star_offsets[target];
	const int ti = a.target_index[target];
	const double* ftar = a.flux + (int64_t)target * a.out_pitch;
	double mean[S], part[S + 1];
#pragma unroll
	for (int u = 0; u <= S; ++u) part[u] = 0.0;
	for (int k = tid; k < a.n_cad; k += 256) {
		const bool ok = ftar[k] == ftar[k];
		part[0] += ok ? 1.0 : 0.0;
#pragma unroll
		for (int s = 0; s < S; ++s) { const double v = a.fluxes_all[(s0 + s) * a.out_pitch + k]; part[1 + s] += ok ? v : 0.0; }
	}
#pragma unroll
	for (int u = 0; u <= S; ++u) {
#pragma unroll
		for (int off = 32; off > 0; off >>= 1) part[u] += __shfl_xor(part[u], off, 64);
		if ((tid & 63) == 0) red[(tid >> 6) * (S + 1) + u] = part[u];
	}
	__syncthreads();
	double cntd = 0.0;
	for (int w = 0; w < 4; ++w) cntd += red[w * (S + 1)];
#pragma unroll
	for (int s = 0; s < S; ++s) {
		double tot = 0.0;
		for (int w = 0; w < 4; ++w) tot += red[w * (S + 1) + 1 + s];
		mean[s] = tot / cntd;
	}
	__syncthreads();
	if (cntd == 0.0) { // allnan(flux) -> ERROR (linpsf_photometry.py:198-200)
		if (tid == 0) { fa.status[target] = TP_STATUS_ERROR; fa.contamination[target] = __builtin_nan(""); }
		return;
	}
	// contamination = sum_p (A[p, others] . mean[others]) * A[p, target] / mean[target]
	const int npix = mplans[target].n_tiles * 16;
	const double* al = alast + (int64_t)target * kMfmaStars * kMfmaPixels;
	double acc = 0.0;
	for (int u = tid; u < npix; u += 256) {
		double others = 0.0, at = 0.0;
#pragma unroll
		for (int s = 0; s < S; ++s) {
			const double v = al[s * kMfmaPixels + u];
			if (s == ti) at = v; else others += v * mean[s];
		}
		acc += others * at;
	}
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
	if ((tid & 63) == 0) red[tid >> 6] = acc;
	__syncthreads();
	if (tid == 0) {
		const double tot = ((red[0] + red[1]) + red[2]) + red[3];
		double mt = 0.0;
#pragma unroll
		for (int u = 0; u < S; ++u) if (u == ti) mt = mean[u];
		const double cont = tot / mt;
		fa.contamination[target] = cont;
		fa.status[target] = (cont > 0.1) ? TP_STATUS_WARNING : TP_STATUS_OK; // :214-219
		if (fa.fluxes_mean) {
#pragma unroll
			for (int u = 0; u < S; ++u) fa.fluxes_mean[s0 + u] = mean[u];
		}
	}
}

//--------------------------------------------------------------------------------------------------
// Any number of fitted stars.  select_stars (linpsf_photometry.py:93-104) has no upper limit: a crowded target can bring
// more stars than the register-resident kernels above are instantiated for (8).  Those targets (rare) are fitted here with
// run-time sized normal equations kept in a context-owned HBM scratch: one thread per cadence like the direct kernel,
// element e of a thread's arrays at scratch[e * n_threads + thread] (coalesced across the cadences of a wavefront).
// Same arithmetic (direct 13x13 contraction, cyclic Jacobi pseudo-inverse with numpy's cut-off), only slower.
//--------------------------------------------------------------------------------------------------
constexpr int kMaxManyStars = 64;

struct ManyScratch {
	double* base; int64_t n_threads; int64_t gt;
	__device__ __forceinline__ double& at(int e) const { return base[(int64_t)e * n_threads + gt]; }
};

// GENERAL: any knot vectors and any cut-off radius (prf_pixel_general: the FITPACK box integral itself); `big_targets` may be
// null (= every target, first_target + blockIdx.x) and the table stays in HBM when it does not fit the LDS (table_in_lds = 0).
template <bool GENERAL>
__global__ __launch_bounds__(256) void tp_linpsf_fit_many_kernel(FitArgs a, const int32_t* __restrict__ big_targets, int first_target, int smax, double* __restrict__ scratch,
	int table_in_lds)
{
	extern __shared__ __align__(16) double lds[]; // [n*ny] coefficient table (if it fits) + [n+4] + [ny+4] knots
	const int target = big_targets ? big_targets[blockIdx.x] : (first_target + (int)blockIdx.x);
	const int tid = threadIdx.x;
	const int n = a.n, ny = a.ny;   // (ny != n only in the GENERAL instantiation)
	const double* cg = a.coef + (int64_t)target * n * ny;
	double* Cl = lds;
	double* kn = lds + (table_in_lds ? (size_t)n * ny : 0);
	double* kny = kn + n + 4;
	if (table_in_lds) for (int i = tid; i < n * ny; i += blockDim.x) Cl[i] = cg[i];
	for (int i = tid; i < n + 4; i += blockDim.x) kn[i] = a.knots_x[i];
	for (int i = tid; i < ny + 4; i += blockDim.x) kny[i] = a.knots_y[i];
	__syncthreads();
	const double* C = table_in_lds ? Cl : cg;
	const int k = blockIdx.y * blockDim.x + tid;
	if (k >= a.n_cad) return;
	const int64_t s0 = a.star_offsets[target];
	const int ns = (int)(a.star_offsets[target + 1] - s0);
	const int H = a.height, W = a.width;
	const double h = kn[5] - kn[4], hy = kny[5] - kny[4], h2 = h * hy;
	// layout of a thread's scratch: G[S*S] V[S*S] g[S] x[S] av[S] srow[S] scol[S] mx[4S] my[4S] ax0[S] by0[S]
	const int S = smax, SS = S * S;
	ManyScratch m{scratch, (int64_t)gridDim.x * gridDim.y * blockDim.x, ((int64_t)blockIdx.x * gridDim.y + blockIdx.y) * blockDim.x + tid};
	const int oG = 0, oV = SS, og = 2 * SS, ox = og + S, oav = ox + S, orow = oav + S, ocol = orow + S, omx = ocol + S, omy = omx + 4 * S,
		oax = omy + 4 * S, oby = oax + S;
	for (int s = 0; s < ns; ++s) {
		const double r = a.pos_row[(s0 + s) * a.pos_pitch + k], c = a.pos_col[(s0 + s) * a.pos_pitch + k];
		m.at(orow + s) = r; m.at(ocol + s) = c;
		if (!GENERAL) {
			double wx[4], wy[4];
			int ax0, by0;
			axis_weights(kn, n, c, h, wx, ax0);
			axis_weights(kny, n, r, hy, wy, by0);
			for (int q = 0; q < 4; ++q) { m.at(omx + 4 * s + q) = wx[q]; m.at(omy + 4 * s + q) = wy[q]; }
			m.at(oax + s) = (double)ax0; m.at(oby + s) = (double)by0;
		}
		m.at(og + s) = 0.0;
		for (int u = 0; u < ns; ++u) m.at(oG + s * S + u) = 0.0;
	}
	const float* img = a.images + (int64_t)target * H * W * a.t_pitch + k;
	const float sub = a.subtract ? a.subtract[(int64_t)target * a.subtract_pitch + k] : 0.f;
	for (int i = 0; i < H; ++i) {
		for (int j = 0; j < W; ++j) {
			float bf = img[(int64_t)(i * W + j) * a.t_pitch];
			if (a.subtract) bf = bf - sub;
			if (!(fabsf(bf) <= 3.402823466e+38f)) continue;
			const double b = (double)bf;
			bool any = false;
			for (int s = 0; s < ns; ++s) {
				double v = 0.0;
				const double dc = (double)j - m.at(ocol + s), dr = (double)i - m.at(orow + s);
				if (sqrt(dc * dc + dr * dr) < a.cutoff) {
					if (GENERAL) {
						// psf.py:146  integral(column_cen - 0.5, column_cen + 0.5, row_cen - 0.5, row_cen + 0.5)
						v = prf_pixel_general(C, n, ny, kn, kny, dc - 0.5, dc + 0.5, dr - 0.5, dr + 0.5);
					} else {
						double wx[4], wy[4];
						for (int q = 0; q < 4; ++q) { wx[q] = m.at(omx + 4 * s + q); wy[q] = m.at(omy + 4 * s + q); }
						int ax = (int)m.at(oax + s) + 9 * j, by = (int)m.at(oby + s) + 9 * i;
						ax = ax < 0 ? 0 : (ax > n - 13 ? n - 13 : ax);
						by = by < 0 ? 0 : (by > n - 13 ? n - 13 : by);
						v = h2 * prf_pixel(C, n, ax, by, wx, wy);
					}
					any = true;
				}
				m.at(oav + s) = v;
			}
			if (!any) continue; // a pixel outside every cut-off disc adds nothing to A^T A or A^T b
			for (int s = 0; s < ns; ++s) {
				const double as = m.at(oav + s);
				if (as == 0.0) continue;
				m.at(og + s) += as * b;
				for (int u = s; u < ns; ++u) m.at(oG + s * S + u) += as * m.at(oav + u);
			}
		}
	}
	for (int s = 0; s < ns; ++s) {
		for (int u = 0; u < s; ++u) m.at(oG + s * S + u) = m.at(oG + u * S + s);
		for (int u = 0; u < ns; ++u) m.at(oV + s * S + u) = (s == u) ? 1.0 : 0.0;
	}
	// cyclic Jacobi (same sweep order and stopping rule as pinv_solve)
	for (int sweep = 0; sweep < 30; ++sweep) {
		double off = 0.0, d2 = 0.0;
		for (int p = 0; p < ns; ++p) {
			const double d = m.at(oG + p * S + p);
			d2 += d * d;
			for (int q = p + 1; q < ns; ++q) { const double o = m.at(oG + p * S + q); off += o * o; }
		}
		if (!(off > 1e-34 * d2)) break;
		for (int p = 0; p < ns; ++p) {
			for (int q = p + 1; q < ns; ++q) {
				const double apq = m.at(oG + p * S + q);
				if (apq == 0.0) continue;
				const double theta = (m.at(oG + q * S + q) - m.at(oG + p * S + p)) / (2.0 * apq);
				const double t = ((theta >= 0.0) ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
				const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
				for (int e = 0; e < ns; ++e) {
					const double gp = m.at(oG + e * S + p), gq = m.at(oG + e * S + q);
					m.at(oG + e * S + p) = c * gp - sn * gq;
					m.at(oG + e * S + q) = sn * gp + c * gq;
				}
				for (int e = 0; e < ns; ++e) {
					const double gp = m.at(oG + p * S + e), gq = m.at(oG + q * S + e);
					m.at(oG + p * S + e) = c * gp - sn * gq;
					m.at(oG + q * S + e) = sn * gp + c * gq;
				}
				for (int e = 0; e < ns; ++e) {
					const double vp = m.at(oV + e * S + p), vq = m.at(oV + e * S + q);
					m.at(oV + e * S + p) = c * vp - sn * vq;
					m.at(oV + e * S + q) = sn * vp + c * vq;
				}
			}
		}
	}
	double smx = 0.0;
	for (int i = 0; i < ns; ++i) { const double v = fabs(m.at(oG + i * S + i)); if (v > smx || v != v) smx = v; }
	const double cut = 1e-15 * smx;
	for (int i = 0; i < ns; ++i) m.at(ox + i) = 0.0;
	for (int e = 0; e < ns; ++e) {
		const double lam = m.at(oG + e * S + e);
		double proj = 0.0;
		for (int i = 0; i < ns; ++i) proj += m.at(oV + i * S + e) * m.at(og + i);
		const double inv = (fabs(lam) > cut) ? (1.0 / lam) : ((lam != lam) ? lam : 0.0);
		const double coef = proj * inv;
		for (int i = 0; i < ns; ++i) m.at(ox + i) += m.at(oV + i * S + e) * coef;
	}
	const int ti = a.target_index[target];
	for (int s = 0; s < ns; ++s) a.fluxes_all[(s0 + s) * a.out_pitch + k] = m.at(ox + s);
	a.flux[(int64_t)target * a.out_pitch + k] = (ti >= 0 && ti < ns) ? m.at(ox + ti) : __builtin_nan("");
	a.flux_err[(int64_t)target * a.out_pitch + k] = __builtin_nan("");
}

// finalise for the targets of the kernel above (same rules as tp_linpsf_finalize_kernel, star loops at run time)
template <bool GENERAL>
__global__ __launch_bounds__(256) void tp_linpsf_finalize_many_kernel(FinArgs fa, const int32_t* __restrict__ big_targets, int first_target)
{
	extern __shared__ __align__(16) double lds[];
	const FitArgs& a = fa.f;
	const int target = big_targets ? big_targets[blockIdx.x] : (first_target + (int)blockIdx.x);
	const int tid = threadIdx.x;
	const int n = a.n, ny = a.ny;
	const double* C = a.coef + (int64_t)target * n * ny;
	double* kn = lds;
	double* kny = kn + n + 4;
	double* red = kny + ny + 4;           // [256]
	double* mean = red + 256;             // [kMaxManyStars]
	for (int i = tid; i < n + 4; i += blockDim.x) kn[i] = a.knots_x[i];
	for (int i = tid; i < ny + 4; i += blockDim.x) kny[i] = a.knots_y[i];
	__syncthreads();
	const int64_t s0 = a.star_offsets[target];
	const int ns = (int)(a.star_offsets[target + 1] - s0);
	const int ti = a.target_index[target];
	const double* ftar = a.flux + (int64_t)target * a.out_pitch;
	double cntd = 0.0;
	for (int s = -1; s < ns; ++s) {
		double acc = 0.0;
		for (int k = tid; k < a.n_cad; k += blockDim.x) {
			const bool ok = ftar[k] == ftar[k];
			if (s < 0) acc += ok ? 1.0 : 0.0;
			else acc += ok ? a.fluxes_all[(s0 + s) * a.out_pitch + k] : 0.0;
		}
		red[tid] = acc;
		__syncthreads();
		double tot = 0.0;
		for (int l = 0; l < (int)blockDim.x; ++l) tot += red[l];
		__syncthreads();
		if (s < 0) cntd = tot;
		else if (tid == 0) mean[s] = tot / cntd;
	}
	__syncthreads();
	if (cntd == 0.0) {
		if (tid == 0) { fa.status[target] = TP_STATUS_ERROR; fa.contamination[target] = __builtin_nan(""); }
		return;
	}
	const int k = a.n_cad - 1;
	const int H = a.height, W = a.width;
	const double h = kn[5] - kn[4], hy = kny[5] - kny[4], h2 = h * hy;
	const float* img = a.images + (int64_t)target * H * W * a.t_pitch + k;
	const float sub = a.subtract ? a.subtract[(int64_t)target * a.subtract_pitch + k] : 0.f;
	double acc = 0.0;
	for (int p = tid; p < H * W; p += blockDim.x) {
		const int i = p / W, j = p - i * W;
		float bf = img[(int64_t)p * a.t_pitch];
		if (a.subtract) bf = bf - sub;
		if (!(fabsf(bf) <= 3.402823466e+38f)) continue;
		double others = 0.0, at = 0.0;
		for (int s = 0; s < ns; ++s) {
			const double srow = a.pos_row[(s0 + s) * a.pos_pitch + k], scol = a.pos_col[(s0 + s) * a.pos_pitch + k];
			const double dc = (double)j - scol, dr = (double)i - srow;
			double v = 0.0;
			if (sqrt(dc * dc + dr * dr) < a.cutoff) {
				if (GENERAL) {
					v = prf_pixel_general(C, n, ny, kn, kny, dc - 0.5, dc + 0.5, dr - 0.5, dr + 0.5);
				} else {
					double wx[4], wy[4];
					int ax0, by0;
					axis_weights(kn, n, scol, h, wx, ax0);
					axis_weights(kny, n, srow, hy, wy, by0);
					int ax = ax0 + 9 * j, by = by0 + 9 * i;
					ax = ax < 0 ? 0 : (ax > n - 13 ? n - 13 : ax);
					by = by < 0 ? 0 : (by > n - 13 ? n - 13 : by);
					v = h2 * prf_pixel(C, n, ax, by, wx, wy);
				}
			}
			if (s == ti) at = v; else others += v * mean[s];
		}
		acc += others * at;
	}
	red[tid] = acc;
	__syncthreads();
	if (tid == 0) {
		double tot = 0.0;
		for (int l = 0; l < (int)blockDim.x; ++l) tot += red[l];
		const double cont = tot / mean[ti];
		fa.contamination[target] = cont;
		fa.status[target] = (cont > 0.1) ? TP_STATUS_WARNING : TP_STATUS_OK;
		if (fa.fluxes_mean) for (int u = 0; u < ns; ++u) fa.fluxes_mean[s0 + u] = mean[u];
	}
}

} // namespace

extern "C" int tp_linpsf_prf(tp_ctx* ctx, int32_t n_targets, int32_t n_samples, int32_t n_coef,
	const double* d_base_coef, const double* d_weights, double* d_coef)
{
	TP_CHECK_CTX(ctx);
	TP_API_BEGIN
	TP_REQUIRE(ctx, n_targets >= 0 && n_samples > 0 && n_samples <= kMaxSamples && n_coef > 0, "tp_linpsf_prf: bad sizes (at most 32 PRF samples)");
	TP_REQUIRE(ctx, d_base_coef && d_weights && d_coef, "tp_linpsf_prf: null pointer");
	if (n_targets == 0) return TP_OK;
	const unsigned gy = (unsigned)((n_targets < 256) ? n_targets : 256);   // 54 x 256 workgroups: each thread's trip (scalar loads of the weights, 25 FMAs, one store) is a latency chain

	dim3 block(256), grid((unsigned)((n_coef + 255) / 256), gy);
	if (n_samples == 25) TP_LAUNCH(ctx, TPK_LINPSF_PRF, tp_linpsf_prf_fixed_kernel<25>, grid, block, 0, d_base_coef, (int)n_coef, d_weights, (int)n_targets, d_coef);
	else TP_LAUNCH(ctx, TPK_LINPSF_PRF, tp_linpsf_prf_kernel, grid, block, 0, d_base_coef, (int)n_samples, (int)n_coef, d_weights, (int)n_targets, d_coef);
	TP_LAUNCH_CHECK(ctx, "tp_linpsf_prf_kernel");
	return TP_OK;
	TP_API_END(ctx)
}

extern "C" int tp_linpsf_set_path(tp_ctx* ctx, int32_t path)
{
	TP_CHECK_CTX(ctx);
	TP_REQUIRE(ctx, path == 0 || path == 1, "tp_linpsf_set_path: 1 (matrix-core fit where a target qualifies) or 0 (vector-ALU kernels only)");
	ctx->linpsf_path = path;
	return TP_OK;
}

extern "C" int tp_linpsf_last_counts(tp_ctx* ctx, int64_t* counts, int32_t n)
{
	TP_CHECK_CTX(ctx);
	TP_REQUIRE(ctx, counts != nullptr && n >= 1 && n <= 16, "tp_linpsf_last_counts: 1..16 counters");
	for (int i = 0; i < n; ++i) counts[i] = ctx->linpsf_counts[i];
	return TP_OK;
}

static int linpsf_fit_impl(tp_ctx* ctx, const tp_cube_desc* desc, const float* d_images,
	const float* d_subtract, int64_t subtract_pitch,
	const double* d_coef, const double* d_knots_x, const double* d_knots_y, int32_t n_coef_axis, int32_t n_coef_axis_y, int32_t max_stars,
	const int64_t* d_star_offsets, const int32_t* d_target_index,
	const double* d_pos_row, const double* d_pos_col, int64_t pos_pitch, double cutoff_radius,
	double* d_flux, double* d_flux_err, double* d_fluxes_all, int64_t out_pitch,
	double* d_contamination, int32_t* d_status, double* d_fluxes_mean)
{
	TP_CHECK_CTX(ctx);
	TP_API_BEGIN
	TP_REQUIRE(ctx, tp_desc_ok(desc), "tp_linpsf_fit: bad cube descriptor");
	TP_REQUIRE(ctx, d_images && d_coef && d_knots_x && d_knots_y && d_star_offsets && d_target_index && d_pos_row && d_pos_col, "tp_linpsf_fit: null input pointer");
	TP_REQUIRE(ctx, d_flux && d_flux_err && d_fluxes_all && d_contamination && d_status, "tp_linpsf_fit: null output pointer");
	TP_REQUIRE(ctx, pos_pitch >= desc->n_cad && out_pitch >= desc->n_cad, "tp_linpsf_fit: pitch < n_cad");
	TP_REQUIRE(ctx, d_subtract == nullptr || subtract_pitch >= desc->n_cad, "tp_linpsf_fit: bad subtract pitch");
	TP_REQUIRE(ctx, n_coef_axis >= 4 && n_coef_axis <= 2048 && n_coef_axis_y >= 4 && n_coef_axis_y <= 2048, "tp_linpsf_fit: coefficient table must be 4..2048 per axis");
	TP_REQUIRE(ctx, max_stars >= 1 && max_stars <= kMaxManyStars, "tp_linpsf_fit: at most 64 stars fitted per target");
	TP_REQUIRE(ctx, cutoff_radius > 0, "tp_linpsf_fit: cutoff_radius must be positive (infinity = no cut-off, psf.py:142 `cutoff_radius is None`)");
	if (desc->n_targets == 0 || desc->n_cad == 0) return TP_OK;

	FitArgs a;
	a.images = d_images; a.subtract = d_subtract; a.subtract_pitch = subtract_pitch;
	a.n_cad = desc->n_cad; a.height = desc->height; a.width = desc->width; a.t_pitch = desc->t_pitch;
	a.coef = d_coef; a.knots_x = d_knots_x; a.knots_y = d_knots_y; a.n = n_coef_axis; a.ny = n_coef_axis_y;
	a.star_offsets = d_star_offsets; a.target_index = d_target_index;
	a.pos_row = d_pos_row; a.pos_col = d_pos_col; a.pos_pitch = pos_pitch; a.cutoff = cutoff_radius;
	a.flux = d_flux; a.flux_err = d_flux_err; a.fluxes_all = d_fluxes_all; a.out_pitch = out_pitch;

	const size_t shmem = ((size_t)n_coef_axis * n_coef_axis_y + (n_coef_axis + 4) + (n_coef_axis_y + 4)) * sizeof(double);
	const int nblk = (desc->n_cad + 511) / 512;
	int threads = (((desc->n_cad + nblk - 1) / nblk) + 63) / 64 * 64;
	dim3 grid((unsigned)desc->n_targets, (unsigned)nblk), block((unsigned)threads);
	const size_t shmem_fin = (((size_t)n_coef_axis + 4) + ((size_t)n_coef_axis_y + 4) + 256) * sizeof(double);
	FinArgs fa; fa.f = a; fa.contamination = d_contamination; fa.status = d_status; fa.fluxes_mean = d_fluxes_mean; fa.todo = nullptr;
	// polynomial path: plan (boxes, item counts) -> coefficient store -> fit; targets whose stars visit more table origins than
	// max_origins are flagged and redone by the general kernel
	const int max_origins = 36;
	const size_t todo_bytes = ((size_t)desc->n_targets * sizeof(int32_t) + 255) & ~(size_t)255;
	const size_t plan_bytes = ((size_t)desc->n_targets * kMaxStars * sizeof(StarPlan) + 255) & ~(size_t)255;
	const size_t order_bytes = ((size_t)desc->n_targets * desc->n_cad * sizeof(int32_t) + 255) & ~(size_t)255;
	const size_t mplan_bytes = ((size_t)desc->n_targets * sizeof(MPlan) + 255) & ~(size_t)255;
	const size_t ulist_bytes = ((size_t)desc->n_targets * kMfmaPixels * sizeof(uint16_t) + 255) & ~(size_t)255;
	const size_t usig_bytes = ((size_t)desc->n_targets * kMfmaPixels * sizeof(uint8_t) + 255) & ~(size_t)255;
	const size_t lists_bytes = ((size_t)desc->n_targets * kMfmaClasses * sizeof(int32_t) + 255) & ~(size_t)255;
	const size_t segs_bytes = ((size_t)desc->n_targets * kMfmaSegs * sizeof(SegPlan) + 255) & ~(size_t)255;
	const size_t seglists_bytes = ((size_t)desc->n_targets * kMfmaSegs * kMfmaClasses * sizeof(int32_t) + 255) & ~(size_t)255;
	const size_t alast_bytes = (ctx->linpsf_path == 1) ? (((size_t)desc->n_targets * kMfmaStars * kMfmaPixels * sizeof(double) + 255) & ~(size_t)255) : 0;
	const size_t head_bytes = todo_bytes + plan_bytes + 256 + order_bytes + mplan_bytes + ulist_bytes + usig_bytes + lists_bytes + segs_bytes + seglists_bytes + alast_bytes;
	static_assert(kTotCount * sizeof(unsigned long long) <= 256, "the counters fit their block");
	TP_REQUIRE(ctx, tp_ctx_scratch(ctx, head_bytes) != nullptr, "tp_linpsf_fit: out of device memory for the plan");
	char* sbase = static_cast<char*>(ctx->scratch);
	int32_t* d_todo = reinterpret_cast<int32_t*>(sbase);
	StarPlan* d_plans = reinterpret_cast<StarPlan*>(sbase + todo_bytes);
	unsigned long long* d_total = reinterpret_cast<unsigned long long*>(sbase + todo_bytes + plan_bytes);
	int32_t* d_order = reinterpret_cast<int32_t*>(sbase + todo_bytes + plan_bytes + 256);
	MPlan* d_mplans = reinterpret_cast<MPlan*>(sbase + todo_bytes + plan_bytes + 256 + order_bytes);
	uint16_t* d_ulist = reinterpret_cast<uint16_t*>(sbase + todo_bytes + plan_bytes + 256 + order_bytes + mplan_bytes);
	uint8_t* d_usig = reinterpret_cast<uint8_t*>(sbase + todo_bytes + plan_bytes + 256 + order_bytes + mplan_bytes + ulist_bytes);
	int32_t* d_lists = reinterpret_cast<int32_t*>(sbase + todo_bytes + plan_bytes + 256 + order_bytes + mplan_bytes + ulist_bytes + usig_bytes);
	SegPlan* d_segs = reinterpret_cast<SegPlan*>(sbase + todo_bytes + plan_bytes + 256 + order_bytes + mplan_bytes + ulist_bytes + usig_bytes + lists_bytes);
	int32_t* d_seglists = reinterpret_cast<int32_t*>(sbase + todo_bytes + plan_bytes + 256 + order_bytes + mplan_bytes + ulist_bytes + usig_bytes + lists_bytes + segs_bytes);
	double* d_alast = reinterpret_cast<double*>(sbase + todo_bytes + plan_bytes + 256 + order_bytes + mplan_bytes + ulist_bytes + usig_bytes + lists_bytes + segs_bytes + seglists_bytes);
	// cadences sorted by origin in LDS (8 bytes per slot, next power of two); beyond 8192 cadences the order stays natural
	int sort_n = 64;
	while (sort_n < desc->n_cad) sort_n <<= 1;
	if (sort_n > 8192) sort_n = 0;
	// the matrix-core path needs the table in LDS (beside the job list of its coefficient kernel), and 32-bit element offsets
	// into a target's cube
	const int use_mfma = (ctx->linpsf_path == 1 && n_coef_axis == n_coef_axis_y && (size_t)n_coef_axis * n_coef_axis * sizeof(double) + 2048 <= 160 * 1024   // (2 KB: the kernel's job table)
		&& (int64_t)desc->height * desc->width * desc->t_pitch < (1ll << 30)) ? 1 : 0;
	fa.todo = use_mfma ? d_todo : nullptr;
	TP_HIP(ctx, hipMemsetAsync(d_todo, 0, todo_bytes, ctx->stream));
	TP_HIP(ctx, hipMemsetAsync(d_total, 0, 256, ctx->stream));
	// the uniform-grid kernels (everything below up to the many-star kernel) need the SPOC layout of the PRF grid: 9 samples per
	// pixel, the table resident in LDS, the cut-off inside the evenly spaced part of the knots.  Whether that holds is decided
	// where the knots are; anything else is fitted by the general kernels with the FITPACK box integral itself
	{
		// (a table with axes of different lengths is never the SPOC layout: the general kernels, the only ones that read a.ny)
		const int force = (n_coef_axis != n_coef_axis_y || n_coef_axis < 32 || n_coef_axis > 140 || !(cutoff_radius <= 5.25)) ? 1 : 0;
		hipLaunchKernelGGL(tp_linpsf_grid_kernel, dim3(1), dim3(64), 0, ctx->stream, d_knots_x, d_knots_y, (int)n_coef_axis, cutoff_radius, force, d_total);
		TP_LAUNCH_CHECK(ctx, "tp_linpsf_grid_kernel");
	}
	if (sort_n > 4096) TP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(tp_linpsf_plan_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sort_n * sizeof(unsigned long long))));
	TP_LAUNCH(ctx, TPK_LINPSF_PLAN, tp_linpsf_plan_kernel, dim3((unsigned)desc->n_targets), dim3(256), (size_t)sort_n * sizeof(unsigned long long), a, d_plans, d_todo, d_total, max_origins, d_order, sort_n,
		d_mplans, d_ulist, d_usig, use_mfma, d_lists, (int)desc->n_targets, d_segs, d_seglists);
	TP_LAUNCH_CHECK(ctx, "tp_linpsf_plan_kernel");
	// items of the polynomial store, doubles of the matrix-core store behind it, targets and segments per class.  The host needs
	// them to size the store and the launches: one round trip in the middle of the call (measured: the plan kernel's 0.2 ms and
	// the launch of the coefficient kernel hide it -- the step's wall time equals the sum of its kernels to 0.05 ms)
	unsigned long long totals[kTotCount] = {};
	TP_HIP(ctx, hipMemcpyAsync(totals, d_total, sizeof(totals), hipMemcpyDeviceToHost, ctx->stream));
	TP_HIP(ctx, hipStreamSynchronize(ctx->stream));
	{
		int64_t* c = ctx->linpsf_counts;
		for (int i = 0; i < 16; ++i) c[i] = 0;
		for (int k = 0; k < kMfmaClasses; ++k) {
			c[0] += (int64_t)totals[kTotClass0 + k]; c[1] += (int64_t)totals[kTotSeg0 + k];
			c[5 + k] = (int64_t)totals[kTotClass0 + k]; c[9 + k] = (int64_t)totals[kTotSeg0 + k];
		}
		c[2] = (int64_t)totals[kTotPolyTargets]; c[3] = (int64_t)totals[kTotDirectTargets];
	}
	if (totals[kTotGeneral] != 0) {
		// ---- any grid, any cut-off: every target through the run-time sized kernel (normal equations in an HBM scratch) with the
		// FITPACK box integral, a few GiB of scratch at a time
		std::vector<int64_t> off((size_t)desc->n_targets + 1);
		TP_HIP(ctx, hipMemcpyAsync(off.data(), d_star_offsets, off.size() * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
		TP_HIP(ctx, hipStreamSynchronize(ctx->stream));
		int smax = 1;
		for (int t = 0; t < desc->n_targets; ++t) {
			const int ns = (int)(off[t + 1] - off[t]);
			TP_REQUIRE(ctx, ns >= 0 && ns <= kMaxManyStars, "tp_linpsf_fit: a target has more than 64 fitted stars");
			if (ns > smax) smax = ns;
		}
		const int table_in_lds = (shmem <= (size_t)160 * 1024) ? 1 : 0;
		const size_t shmem_g = table_in_lds ? shmem : ((size_t)(n_coef_axis + 4) + (size_t)(n_coef_axis_y + 4)) * sizeof(double);
		const int threads_m = 256, nblk_m = (desc->n_cad + threads_m - 1) / threads_m;
		const size_t per_target = (size_t)(2 * smax * smax + 15 * smax) * sizeof(double) * nblk_m * threads_m;
		int64_t chunk = (int64_t)(((size_t)4 << 30) / per_target);
		if (chunk < 1) chunk = 1;
		if (chunk > desc->n_targets) chunk = desc->n_targets;
		TP_REQUIRE(ctx, tp_ctx_scratch(ctx, head_bytes + per_target * (size_t)chunk + 256) != nullptr, "tp_linpsf_fit: out of device memory for the scratch of the general kernels");
		double* d_scr = reinterpret_cast<double*>(static_cast<char*>(ctx->scratch) + head_bytes);
		TP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(tp_linpsf_fit_many_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem_g));
		const size_t shmem_fin_m = shmem_fin + kMaxManyStars * sizeof(double);
		for (int64_t first = 0; first < desc->n_targets; first += chunk) {
			const int64_t cnt = (desc->n_targets - first < chunk) ? (desc->n_targets - first) : chunk;
			TP_LAUNCH(ctx, TPK_LINPSF_FIT_DIRECT, tp_linpsf_fit_many_kernel<true>, dim3((unsigned)cnt, (unsigned)nblk_m), dim3(threads_m), shmem_g, a, (const int32_t*)nullptr, (int)first, smax, d_scr, table_in_lds);
			TP_LAUNCH_CHECK(ctx, "tp_linpsf_fit_many_kernel (general)");
			TP_LAUNCH(ctx, TPK_LINPSF_FIN, tp_linpsf_finalize_many_kernel<true>, dim3((unsigned)cnt), dim3(256), shmem_fin_m, fa, (const int32_t*)nullptr, (int)first);
			TP_LAUNCH_CHECK(ctx, "tp_linpsf_finalize_many_kernel (general)");
		}
		ctx->linpsf_counts[13] = desc->n_targets;
		return TP_OK;
	}
	const size_t poly_doubles = ((size_t)totals[kTotPolyItems] * 25 + 32 + 63) & ~(size_t)63;
	const size_t store_need = (poly_doubles + (size_t)totals[kTotKDoubles] + 64) * sizeof(double);
	if (ctx->store_bytes < store_need) {
		if (ctx->store) (void)hipFree(ctx->store);
		ctx->store = nullptr; ctx->store_bytes = 0;
		TP_HIP(ctx, tp_device_alloc(ctx, &ctx->store, store_need));
		ctx->store_bytes = store_need;
	}
	double* d_store = static_cast<double*>(ctx->store);
	double* d_kstore = d_store + poly_doubles;
	const size_t coef_lds = (size_t)n_coef_axis * n_coef_axis * sizeof(double);
	TP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(tp_linpsf_coef_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)coef_lds));
	TP_LAUNCH(ctx, TPK_LINPSF_COEF, tp_linpsf_coef_kernel, dim3((unsigned)desc->n_targets), dim3(kCoefThreads), coef_lds, a, (const StarPlan*)d_plans, (const int32_t*)d_todo, d_store,
		(const MPlan*)d_mplans, (const uint16_t*)d_ulist, (const uint8_t*)d_usig, d_kstore, (const SegPlan*)d_segs);
	TP_LAUNCH_CHECK(ctx, "tp_linpsf_coef_kernel");
	// the matrix-core fit of the targets marked for it (up to 4 stars, up to 256 reachable pixels)
	if (use_mfma) {
		const int rc = fit_mfma_launch(ctx, a, desc->n_targets, totals + kTotSeg0, totals + kTotClass0, d_segs, d_seglists, d_mplans, d_ulist, d_usig, d_kstore, d_alast);
		if (rc != TP_OK) return rc;
#define TP_LINPSF_FINM(CLS, SS) do { \
			if (totals[kTotClass0 + CLS] > 0) { \
				TP_LAUNCH(ctx, TPK_LINPSF_FIN, (tp_linpsf_finalize_m_kernel<SS>), dim3((unsigned)totals[kTotClass0 + CLS]), dim3(256), 0, fa, (const int32_t*)(d_lists + (size_t)(CLS) * desc->n_targets), \
					(const MPlan*)d_mplans, (const double*)d_alast); \
				TP_LAUNCH_CHECK(ctx, "tp_linpsf_finalize_m_kernel"); \
			} \
		} while (0)
		TP_LINPSF_FINM(0, 1); TP_LINPSF_FINM(1, 2); TP_LINPSF_FINM(2, 3); TP_LINPSF_FINM(3, 4);
#undef TP_LINPSF_FINM
	}
	const int nblk2 = (desc->n_cad + 255) / 256;
	const int threads2 = (((desc->n_cad + nblk2 - 1) / nblk2) + 63) / 64 * 64;
#define TP_LINPSF_FIT2(SS, SL) do { \
		TP_LAUNCH(ctx, TPK_LINPSF_FIT, (tp_linpsf_fit2_kernel<SS, SL>), dim3((unsigned)desc->n_targets, (unsigned)nblk2), dim3((unsigned)threads2), 0, a, (const StarPlan*)d_plans, (const int32_t*)d_todo, (const double*)d_store, (const int32_t*)d_order); \
		TP_LAUNCH_CHECK(ctx, "tp_linpsf_fit2_kernel"); \
	} while (0)
#define TP_LINPSF_LAUNCH(SS, SL) do { \
		TP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(tp_linpsf_fit_direct_kernel<SS, SL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem)); \
		TP_LAUNCH(ctx, TPK_LINPSF_FIT_DIRECT, (tp_linpsf_fit_direct_kernel<SS, SL>), grid, block, shmem, a, (const int32_t*)d_todo); \
		TP_LAUNCH_CHECK(ctx, "tp_linpsf_fit_direct_kernel"); \
		TP_LAUNCH(ctx, TPK_LINPSF_FIN, (tp_linpsf_finalize_kernel<SS, SL>), dim3((unsigned)desc->n_targets), dim3(256), shmem_fin, fa); \
		TP_LAUNCH_CHECK(ctx, "tp_linpsf_finalize_kernel"); \
	} while (0)
	// one instantiation per star count (the normal equations and the registers of a 1-star target are not those of a 4-star
	// one); a workgroup whose target belongs to another class exits at once
	if (totals[kTotPolyTargets] > 0) {   // none when the matrix-core fit has taken every target
		TP_LINPSF_FIT2(1, 0);
		if (max_stars > 1) TP_LINPSF_FIT2(2, 2);
		if (max_stars > 2) TP_LINPSF_FIT2(3, 3);
		if (max_stars > 3) TP_LINPSF_FIT2(4, 4);
		if (max_stars > 4) TP_LINPSF_FIT2(8, 5);
	}
	// the general kernel (flagged targets) and the finalisation, by coarser classes
	if (totals[kTotPolyTargets] + totals[kTotDirectTargets] > 0) {
		TP_LINPSF_LAUNCH(2, 0);
		if (max_stars > 2) TP_LINPSF_LAUNCH(4, 3);
		if (max_stars > 4) TP_LINPSF_LAUNCH(8, 5);
	}
#undef TP_LINPSF_FIT2
#undef TP_LINPSF_LAUNCH
	if (max_stars > kMaxStars) {
		// targets with more than 8 fitted stars (rare: crowded fields): listed on the host from the star offsets, fitted by
		// the run-time sized kernel out of an HBM scratch
		std::vector<int64_t> off((size_t)desc->n_targets + 1);
		TP_HIP(ctx, hipMemcpyAsync(off.data(), d_star_offsets, off.size() * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
		TP_HIP(ctx, hipStreamSynchronize(ctx->stream));
		std::vector<int32_t> big;
		int smax = 0;
		for (int t = 0; t < desc->n_targets; ++t) {
			const int ns = (int)(off[t + 1] - off[t]);
			TP_REQUIRE(ctx, ns <= kMaxManyStars, "tp_linpsf_fit: a target has more than 64 fitted stars");
			if (ns > kMaxStars) { big.push_back(t); if (ns > smax) smax = ns; }
		}
		ctx->linpsf_counts[4] = (int64_t)big.size();
		if (!big.empty()) {
			const int threads = 256, nblk_m = (desc->n_cad + threads - 1) / threads;
			const size_t per_thread = (size_t)(2 * smax * smax + 15 * smax) * sizeof(double);
			const size_t list_bytes = (big.size() * sizeof(int32_t) + 255) & ~(size_t)255;
			const size_t head = head_bytes;
			const size_t need = head + list_bytes + per_thread * big.size() * nblk_m * threads + 256;
			// the scratch also holds d_todo at its start: grow it BEFORE the class kernels' flags could be lost -- they are done
			TP_HIP(ctx, hipStreamSynchronize(ctx->stream));
			TP_REQUIRE(ctx, tp_ctx_scratch(ctx, need) != nullptr, "tp_linpsf_fit: out of device memory for the many-star scratch");
			char* base = static_cast<char*>(ctx->scratch) + head;
			int32_t* d_big = reinterpret_cast<int32_t*>(base);
			double* d_scr = reinterpret_cast<double*>(base + list_bytes);
			TP_HIP(ctx, hipMemcpyAsync(d_big, big.data(), big.size() * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
			TP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(tp_linpsf_fit_many_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
			TP_LAUNCH(ctx, TPK_LINPSF_FIT_DIRECT, tp_linpsf_fit_many_kernel<false>, dim3((unsigned)big.size(), (unsigned)nblk_m), dim3(threads), shmem, a, (const int32_t*)d_big, 0, smax, d_scr, 1);
			TP_LAUNCH_CHECK(ctx, "tp_linpsf_fit_many_kernel");
			const size_t shmem_fin_m = shmem_fin + kMaxManyStars * sizeof(double);
			TP_LAUNCH(ctx, TPK_LINPSF_FIN, tp_linpsf_finalize_many_kernel<false>, dim3((unsigned)big.size()), dim3(256), shmem_fin_m, fa, (const int32_t*)d_big, 0);
			TP_LAUNCH_CHECK(ctx, "tp_linpsf_finalize_many_kernel");
			TP_HIP(ctx, hipStreamSynchronize(ctx->stream)); // `big` (host) must outlive the copy
		}
	}
	return TP_OK;
	TP_API_END(ctx)
}

extern "C" int tp_linpsf_fit(tp_ctx* ctx, const tp_cube_desc* desc, const float* d_images,
	const float* d_subtract, int64_t subtract_pitch,
	const double* d_coef, const double* d_knots_x, const double* d_knots_y, int32_t n_coef_axis, int32_t max_stars,
	const int64_t* d_star_offsets, const int32_t* d_target_index,
	const double* d_pos_row, const double* d_pos_col, int64_t pos_pitch, double cutoff_radius,
	double* d_flux, double* d_flux_err, double* d_fluxes_all, int64_t out_pitch,
	double* d_contamination, int32_t* d_status, double* d_fluxes_mean)
{
	return linpsf_fit_impl(ctx, desc, d_images, d_subtract, subtract_pitch, d_coef, d_knots_x, d_knots_y, n_coef_axis, n_coef_axis, max_stars,
		d_star_offsets, d_target_index, d_pos_row, d_pos_col, pos_pitch, cutoff_radius, d_flux, d_flux_err, d_fluxes_all, out_pitch,
		d_contamination, d_status, d_fluxes_mean);
}

// the same for a PRF spline whose two axes have different numbers of samples (psf.py:119 takes any RectBivariateSpline): d_coef
// [n_targets][n_coef_axis_x * n_coef_axis_y], d_knots_x [n_coef_axis_x + 4], d_knots_y [n_coef_axis_y + 4]; fitted by the any-grid kernels
extern "C" int tp_linpsf_fit_xy(tp_ctx* ctx, const tp_cube_desc* desc, const float* d_images,
	const float* d_subtract, int64_t subtract_pitch,
	const double* d_coef, const double* d_knots_x, const double* d_knots_y, int32_t n_coef_axis_x, int32_t n_coef_axis_y, int32_t max_stars,
	const int64_t* d_star_offsets, const int32_t* d_target_index,
	const double* d_pos_row, const double* d_pos_col, int64_t pos_pitch, double cutoff_radius,
	double* d_flux, double* d_flux_err, double* d_fluxes_all, int64_t out_pitch,
	double* d_contamination, int32_t* d_status, double* d_fluxes_mean)
{
	return linpsf_fit_impl(ctx, desc, d_images, d_subtract, subtract_pitch, d_coef, d_knots_x, d_knots_y, n_coef_axis_x, n_coef_axis_y, max_stars,
		d_star_offsets, d_target_index, d_pos_row, d_pos_col, pos_pitch, cutoff_radius, d_flux, d_flux_err, d_fluxes_all, out_pitch,
		d_contamination, d_status, d_fluxes_mean);
}

// positions of the fitted stars of a field that moves as a whole (see tessphot_hip.h)
namespace {
__global__ __launch_bounds__(256) void tp_star_positions_kernel(int64_t n_stars, int n_cad, const float* __restrict__ base, const float* __restrict__ shift,
	double* __restrict__ pos, int64_t pitch)
{
	const int k = blockIdx.x * blockDim.x + threadIdx.x;
	if (k >= n_cad) return;
	const float sh = shift[k];
	for (int64_t s = blockIdx.y; s < n_stars; s += gridDim.y) pos[s * pitch + k] = (double)(base[s] + sh);
}
} // namespace

extern "C" int tp_star_positions(tp_ctx* ctx, int64_t n_stars, int32_t n_cad, const float* d_base, const float* d_shift, double* d_pos, int64_t pos_pitch)
{
	TP_CHECK_CTX(ctx);
	TP_API_BEGIN
	TP_REQUIRE(ctx, n_stars >= 0 && n_cad >= 0 && pos_pitch >= n_cad, "tp_star_positions: bad sizes");
	if (n_stars == 0 || n_cad == 0) return TP_OK;
	TP_REQUIRE(ctx, d_base && d_shift && d_pos, "tp_star_positions: null pointer");
	const unsigned gy = (unsigned)(n_stars < 65535 ? n_stars : 65535);
	TP_LAUNCH(ctx, TPK_STAR_POSITIONS, tp_star_positions_kernel, dim3((unsigned)((n_cad + 255) / 256), gy), dim3(256), 0, n_stars, (int)n_cad, d_base, d_shift, d_pos, pos_pitch);
	TP_LAUNCH_CHECK(ctx, "tp_star_positions_kernel");
	return TP_OK;
	TP_API_END(ctx)
}
