// psfphot.hip -- non-linear PSF photometry: PSFPhotometry.do_photometry for a batch.
//
// Replaces photometry/psf_photometry.py:52-108 (_lhood, Gaussian_d statistic with background) and :143-196 (the per-cadence
// scipy Nelder-Mead fit of (row, column, flux) of up to five stars, warm-started from the previous cadence, plus the
// aperture correction on the residuals), on top of the pixel-integrated PRF of psf.py:122-148 (linpsf_dev.h).
//
// Mapping (gfx950).  The cadences of a target form a CHAIN (the fit of cadence k starts from the solution of k-1), so the
// parallelism is across targets and inside one likelihood evaluation: one 256-thread workgroup per target, the simplex and the
// cadence's image / weight map in LDS.  Every thread runs the same Nelder-Mead control flow on the LDS-resident simplex (uniform
// branches: all decisions are taken on values read back from LDS); the bookkeeping (centroid, ordering, convergence test) is
// spread over vertex x component threads.  A likelihood evaluation spreads the pixels over the threads, each sums
// flux_s * PRF_s over the stars whose cut-off disc holds the pixel, chi^2 is reduced by a fixed shuffle / LDS tree.
// PRF_s of a pixel is evaluated in the POLYNOMIAL form of linpsf.hip: for fixed knot intervals of the star's sub-pixel phases
// it is a biquartic in the two phases.  The 25 coefficients of the 11 x 11 pixels around a star are cached in LDS per pair of
// knot intervals (24 KB a set, a pool of six sets shared by the target's stars, oldest replaced); an evaluation whose star sits
// in intervals that are not cached builds that set from the coefficient table in HBM / L2 first (121 x 5 contractions of a
// 13 x 13 patch over the 256 threads).
// Late in a fit the simplex is far smaller than a knot interval (1/9 pixel) and nothing is rebuilt: 24 FMAs and 25 LDS reads
// per star and pixel instead of the 169 table reads + ~230 flops of the direct contraction (round 2's first version, which
// was LDS-bandwidth bound with the 110 KB table resident).
// Round 4, measured (tools/psf_time.py, 4 096 targets x 50 cadences; lab clocks of a workgroup's phases, counters: tools/lab/psf_counters.sh):
//  * the kernel is a set of CHAINS: a three-star target takes 26 000 iterations (the reference's maxiter is reached at nearly
//    every cadence: nine parameters do not converge in 500 Nelder-Mead steps), 917 such targets on 512 slots are two rounds of
//    that chain, and the chip idles around them (0.9 wavefronts per SIMD on average, vector ALUs 14 % busy): what counts is the
//    latency of ONE iteration of ONE workgroup.  Where that went for a one-star target: 26 % one thread per star deriving the
//    star parameters (two dependent reads of the knots from L2, the result published through LDS behind a barrier), 24 % set
//    rebuilds, 15 % the pixels, 9 % the reduction, 27 % simplex bookkeeping.
//  * now: every thread derives the star parameters itself from the parameter vector and the knots in LDS (registers, no LDS
//    record, no barrier unless a set has to be rebuilt), the partial sums of an evaluation alternate between two places (one
//    barrier per evaluation instead of four), one kernel instantiation per star count: 32 -> 29 ns per simplex iteration
//    chip-wide, one-star targets 23 -> 15, two-star 28 -> 25.
//  * a RACE this uncovered (and the old kernel had, hidden by the barrier that ended every evaluation): the thread that accepts
//    a point overwrites fsim[D] while slower wavefronts may still be comparing with it (fxcc < fsim[D]) -- those then shrink
//    instead of accepting and the workgroup falls apart.  The decisions now compare with copies taken before the evaluation
//    (tests/test_gpu_psfphot.py::test_psf_fit_is_reproducible; the old kernel gave one target 17 834 iterations where 5 902 are right).
//  * built and dropped: ONE WAVEFRONT per target (181 ns against 94 at 512 targets: the pixels of an evaluation and the 218 work
//    items of a rebuild take four times as long, and the LDS of the cached sets lets a CU hold three such targets, not eight);
//    168 registers for three workgroups per CU (one- and two-star classes alone 10 % faster, the mix 1.82 s against 1.61: the
//    three-star chains that decide the run share their CUs with more neighbours); wave priority for the three-star class (no
//    change); three cached sets for a one-star target (+2 %); the accepted point handed to the sort and the convergence test
//    sharing the centroid's barrier (two barriers fewer per iteration: no gain -- barriers are not what an iteration waits for).
//  * 11.6 % of the star evaluations miss their cached sets (0.33 rebuilds per simplex iteration: a simplex that straddles a knot
//    in both axes alternates between four interval pairs); a rebuild is four dependent rounds of table reads from L2.
// The simplex search is scipy 1.7.3's `_minimize_neldermead` step for step (non-adaptive coefficients 1, 2, 0.5, 0.5; initial
// simplex 5 % / 0.00025; termination xatol = fatol = 1e-4; `success` = finished before maxiter; stable ordering of ties).
#include "common.h"
#include "linpsf_dev.h"
#include <cmath>
#include <vector>
#include <algorithm>
#include <cstdlib>

void* tp_ctx_scratch(tp_ctx* ctx, size_t bytes); // aperture.hip

namespace {

using namespace tp_prf;

constexpr int kMaxPsfStars = 5;                  // psf_photometry.py:127-128
constexpr int kMaxDim = 3 * kMaxPsfStars;
constexpr int kThreads = 256;

struct PsfArgs {
	const float* images; const float* backgrounds;
	int n_cad, height, width; int64_t t_pitch;
	const double* coef; const double* knots_x; const double* knots_y; int n; int ny;   // ny != n: the general instantiation only (tp_psf_fit_xy)
	const int64_t* star_offsets; const double* params0; const uint8_t* mini_aperture;
	float var_floor; double cutoff; int maxiter_first, maxiter;
	double* flux; double* flux_err; double* cen_row; double* cen_col; int64_t out_pitch;
	double* params_out; int32_t* nit; int32_t* status;
	double* gstore;   // [gstore_targets][kStoreSlots][kItems * 25]: the sets a target has built, kept in HBM / L2 (nullptr: none)
	int gstore_targets;   // targets with an index below this have a store
};

constexpr int kHalfBox = 5;                      // pixels inside the cut-off (<= 5.25) lie within +-5 of the pixel nearest to the star
constexpr int kBox = 2 * kHalfBox + 1;
// The cached (pixel offset) items of a star, 25 coefficients each: of the 11 x 11 box only the offsets (di, dj) that can be inside the
// cut-off for SOME position of the star within half a pixel of the box centre -- (|di| - 1/2)+^2 + (|dj| - 1/2)+^2 < 5.25^2: rows
// of 7, 9, 11 ... 11, 9, 7 offsets, 109 in all.  (Round 3 cached all 121: 24.2 KB a set; 21.8 KB now, and with the image and weight
// map kept in float32 and the knots read from global memory a one- or two-star target needs 48 KB of LDS instead of 56: three
// workgroups per CU instead of two.)
// (row widths 2 * {3, 4, 5, 5, 5, 5, 5, 5, 5, 4, 3} + 1: row_half below)
__device__ constexpr int kRowStart[kBox] = {0, 7, 16, 27, 38, 49, 60, 71, 82, 93, 102};
constexpr int kItems = 109;

// A target's store of built sets in HBM (L2-resident while the target runs): a simplex that straddles a knot alternates between
// two to four pairs of knot intervals per star, and with ONE set per star in LDS (three-star targets) every such switch was a
// rebuild -- four dependent rounds of table reads and ~700 dependent multiply-adds per thread, 12 % of the star evaluations and
// the largest piece of an evaluation (round 6, in-kernel clocks).  A set that was built once is copied back instead: 2 725 doubles,
// eleven coalesced loads per thread, one round trip.  The same numbers either way: bit-identical results.
constexpr int kStoreSlots = 16;
constexpr int kPool = 6;                         // most cached coefficient sets (24 KB each) a target gets; it uses psf_pool(ns) of them, pool / ns per star

// Sets cached for a target with ns fitted stars: two for a single star, one per star otherwise.  The kernel is a chain of
// dependent steps (one workgroup per target), so what counts is how many workgroups a CU holds: with 2 sets (56 KB of LDS) two,
// and the fit of one- and two-star targets takes 20 / 36 ns per simplex iteration instead of 34 / 49 with the 6 sets (145 KB) of
// round 2 -- more rebuilds, fewer idle CUs (measured, tools/psf_time.py).
__host__ __device__ constexpr int psf_pool(int ns) { return ns < 2 ? 2 : (ns > kPool ? kPool : ns); }


// width and first item of row r = di + 5 of the cached item set (kRowHalf / kRowStart as arithmetic: a table indexed by a
// lane's own di is a load from memory in front of every pixel)
__device__ __forceinline__ int row_half(int di) { const int m = (di < 0 ? -di : di) - 3; return 5 - (m > 0 ? m : 0); }
__device__ __forceinline__ int row_start(int r) { return (r == 0) ? 0 : ((r == 1) ? 7 : ((r == 10) ? 102 : (16 + 11 * (r - 2)))); }

// the parameters of one star as an evaluation needs them: in REGISTERS of every thread (each thread works them out itself from
// the parameter vector and the knots in LDS -- round 3 had one thread per star do it and publish them through LDS behind a
// barrier: a quarter of an iteration, tools/lab lab clocks)
struct StarR { double row, col, flux, phx, phy; int jstar, istar, slot, valid; };

// everything an evaluation needs besides the parameter vector
struct EvalCtx {
	int ns, n, ny, H, W, pool; double h, hy, cutoff;
	const double* Cg;            // the target's coefficient table in HBM
	const double* kn; const double* kny;   // the knot vectors (LDS)
	const float* img; const float* wgt;    // float32 as psf_photometry.py:75-86 computes them
	double* Kc; double* red; int* keys;    // keys[kPool][2]: the knot intervals of every cached set; red[2][4]
	int* nxt; int* rbs;                    // per star: the set of its share that is replaced next; the set to rebuild now (-1: none)
	double* gset; int* gkeys; int* gsrc;   // the target's store in HBM; its tags [kStoreSlots][3] = (star, kx, ky) and next slot [1] (LDS); per star: slot to copy from (>= 0) or -(slot + 1) to fill
};

// Star parameters of x for every thread; a star whose knot intervals are in none of its cached sets gets the oldest one rebuilt
// (the only case with barriers: the cache state changes).  Same arithmetic, same replacement order as before.
// GEN: any knot vectors, any cut-off radius -- nothing is cached, model_pixel integrates the spline over the pixel itself
template <int NS, bool GEN>
__device__ __forceinline__ void prepare_stars(const double* x, const EvalCtx& c, StarR (&st)[NS])
{
	const int tid = threadIdx.x;
	if (GEN) {
#pragma unroll
		for (int s = 0; s < NS; ++s) {
			st[s].row = x[3 * s]; st[s].col = x[3 * s + 1]; st[s].flux = x[3 * s + 2];
			st[s].valid = ((fabs(st[s].row) < 1e6) && (fabs(st[s].col) < 1e6)) ? 1 : 0;   // (as axis_phase: a NaN position is never inside the cut-off)
			st[s].phx = st[s].phy = 0.0; st[s].jstar = st[s].istar = 0; st[s].slot = -1;
		}
		return;
	}
	const int per = c.pool / NS;
	bool miss = false;
	int kxs[NS], kys[NS];
#pragma unroll
	for (int s = 0; s < NS; ++s) {
		st[s].row = x[3 * s]; st[s].col = x[3 * s + 1]; st[s].flux = x[3 * s + 2];
		int ax0, by0;
		const bool vx = axis_phase(c.kn, c.n, st[s].col, c.h, st[s].phx, ax0);     // x <-> column (first spline axis), y <-> row (psf.py:146)
		const bool vy = axis_phase(c.kny, c.n, st[s].row, c.hy, st[s].phy, by0);
		st[s].valid = (vx && vy) ? 1 : 0;
		st[s].jstar = st[s].valid ? (int)rint(st[s].col) : 0;
		st[s].istar = st[s].valid ? (int)rint(st[s].row) : 0;
		// the knot intervals, free of the pixel the star sits in: first = (l - 3) - 9 * jstar
		kxs[s] = ax0 + 9 * st[s].jstar; kys[s] = by0 + 9 * st[s].istar;
		// the star's share of the pool: sets [s * per, (s + 1) * per); a hit anywhere in it, else the oldest is replaced
		int hit = -1;
		for (int e = 0; e < per; ++e) { const int* k2 = c.keys + 2 * (s * per + e); if (k2[0] == kxs[s] && k2[1] == kys[s]) hit = s * per + e; }
		st[s].slot = hit;
		miss = miss || (st[s].valid && hit < 0);
	}
	if (!miss) return;                         // uniform: every thread read the same simplex and the same keys
	__syncthreads();                           // ... and has read them
#pragma unroll
	for (int s = 0; s < NS; ++s) {
		if (tid == s) {
			int rb = -1;
			if (st[s].valid && st[s].slot < 0) {
				rb = s * per + c.nxt[s];
				c.nxt[s] = (c.nxt[s] + 1 == per) ? 0 : (c.nxt[s] + 1);
				c.keys[2 * rb] = kxs[s]; c.keys[2 * rb + 1] = kys[s];
			}
			c.rbs[s] = rb;
		}
	}
	if (c.gset) {
		// (one thread: the stars of an evaluation share the store's tags and its replacement pointer)
		if (tid == 0) {
#pragma unroll
			for (int s = 0; s < NS; ++s) {
				int src = 0x7fffffff;
				if (st[s].valid && st[s].slot < 0) {
					for (int q = 0; q < kStoreSlots; ++q) if (c.gkeys[3 * q] == s && c.gkeys[3 * q + 1] == kxs[s] && c.gkeys[3 * q + 2] == kys[s]) src = q;
					if (src == 0x7fffffff) {
						const int q = c.gkeys[3 * kStoreSlots];
						c.gkeys[3 * kStoreSlots] = (q + 1 == kStoreSlots) ? 0 : (q + 1);
						c.gkeys[3 * q] = s; c.gkeys[3 * q + 1] = kxs[s]; c.gkeys[3 * q + 2] = kys[s];
						src = -(q + 1);
					}
				}
				c.gsrc[s] = src;
			}
		}
	}
	__syncthreads();
	const double h2 = c.h * c.hy;
#pragma unroll
	for (int s = 0; s < NS; ++s) {
		const int rb = c.rbs[s];
		if (rb < 0) continue;   // uniform
		st[s].slot = rb;
		const int kx = kxs[s], ky = kys[s];
		double* K = c.Kc + (size_t)rb * kItems * 25;
		const int gsrc = c.gset ? c.gsrc[s] : 0x7fffffff;   // uniform
		if (gsrc >= 0 && gsrc != 0x7fffffff) {
			// built before: back from the store
			const double* G = c.gset + (size_t)gsrc * kItems * 25;
			for (int w = tid; w < kItems * 25; w += kThreads) K[w] = G[w];
			continue;
		}

		// two threads per item: one contracts columns 0..2 of the 25 coefficients, the other columns 3..4, each with one pass over
		// the item's 13 x 13 patch of the table (round 3: a thread per (item, column) = five passes per item; the table is read
		// from L2 and those reads, ~0.7 MB per rebuild, were what the kernel waited for)
		for (int w = tid; w < kItems * 2; w += kThreads) {
			const int item = w >> 1, half = w & 1;
			int r = 0;
#pragma unroll
			for (int q = 1; q < kBox; ++q) r += (item >= kRowStart[q]) ? 1 : 0;
			const int di = r - kHalfBox, dj = (item - row_start(r)) - row_half(di);
			int ax = kx + 9 * dj, by = ky + 9 * di;
			ax = ax < 0 ? 0 : (ax > c.n - 13 ? c.n - 13 : ax);
			by = by < 0 ? 0 : (by > c.n - 13 ? c.n - 13 : by);
			if (half == 0) {
				double col[3][5];
				poly_columns<3>(c.Cg, c.n, ax, by, 0, h2, col);
#pragma unroll
				for (int b = 0; b < 3; ++b)
#pragma unroll
					for (int e = 0; e < 5; ++e) K[item * 25 + e * 5 + b] = col[b][e];
			} else {
				double col[2][5];
				poly_columns<2>(c.Cg, c.n, ax, by, 3, h2, col);
#pragma unroll
				for (int b = 0; b < 2; ++b)
#pragma unroll
					for (int e = 0; e < 5; ++e) K[item * 25 + e * 5 + 3 + b] = col[b][e];
			}
		}
	}
	__syncthreads();
	if (c.gset) {
		// the sets built just now go to the target's store (a pass of its own: inside the contraction the store's address and
		// the branch cost the kernel 60 registers)
#pragma unroll
		for (int s = 0; s < NS; ++s) {
			const int gsrc = c.gsrc[s];
			if (c.rbs[s] < 0 || gsrc >= 0) continue;   // uniform
			const double* K = c.Kc + (size_t)c.rbs[s] * kItems * 25;
			double* G = c.gset + (size_t)(-gsrc - 1) * kItems * 25;
			for (int w = tid; w < kItems * 25; w += kThreads) G[w] = K[w];
		}
	}
}

// sum over the stars of flux * pixel-integrated PRF at pixel (i, j)
template <int NS, bool GEN>
__device__ __forceinline__ double model_pixel(int i, int j, const EvalCtx& c, const StarR (&st)[NS])
{
	double mdl = 0.0;
#pragma unroll
	for (int s = 0; s < NS; ++s) {
		if (!st[s].valid) continue;
		if (GEN) {
			const double dc = (double)j - st[s].col, dr = (double)i - st[s].row;
			if (sqrt(dc * dc + dr * dr) < c.cutoff)     // psf.py:142, :146
				mdl += st[s].flux * prf_pixel_general(c.Cg, c.n, c.ny, c.kn, c.kny, dc - 0.5, dc + 0.5, dr - 0.5, dr + 0.5);
			continue;
		}
		const int di = i - st[s].istar, dj = j - st[s].jstar;
		if (di < -kHalfBox || di > kHalfBox) continue;
		const int half = row_half(di);
		if (dj < -half || dj > half) continue;          // offsets outside the cached set are never inside the cut-off
		const double dc = (double)j - st[s].col, dr = (double)i - st[s].row;
		if (sqrt(dc * dc + dr * dr) < c.cutoff)     // psf.py:142 (a NaN position is never inside)
			mdl += st[s].flux * poly_eval(c.Kc + ((size_t)st[s].slot * kItems + row_start(di + kHalfBox) + (dj + half)) * 25, st[s].phx, st[s].phy);
	}
	return mdl;
}

// chi^2 of the parameter vector x (psf_photometry.py:52-90); all threads call it, all get the same value.  One barrier: the
// partial sums of the wavefronts alternate between two places (`flip`), so the next evaluation never writes what a slower
// wavefront still reads.
template <int NS, bool GEN>
__device__ double likelihood(const double* x, const EvalCtx& c, int& flip)
{
	const int tid = threadIdx.x;
	StarR st[NS];
	prepare_stars<NS, GEN>(x, c, st);
	double acc = 0.0;
	for (int p = tid; p < c.H * c.W; p += kThreads) {
		const int i = p / c.W, j = p - i * c.W;
		const double r = (double)c.img[p] - model_pixel<NS, GEN>(i, j, c, st);
		const double term = (double)c.wgt[p] * (r * r);
		if (term == term) acc += term;                  // nansum
	}
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
	double* red = c.red + 4 * flip;
	flip ^= 1;
	if ((tid & 63) == 0) red[tid >> 6] = acc;
	__syncthreads();
	const double tot = (red[0] + red[1]) + (red[2] + red[3]);
	return tot;
}

// `targets`: the targets of this launch (all with the same number of fitted stars, so that `pool` sets are what each needs)
// NS: the number of fitted stars of the launch's targets (0: the targets without one)
template <int NS, bool GEN>
__global__ __launch_bounds__(kThreads) void tp_psf_fit_kernel(PsfArgs a, const int32_t* __restrict__ targets, int pool)
{
	extern __shared__ __align__(16) double lds[];
	const int target = targets[blockIdx.x];
	const int tid = threadIdx.x;
	const int n = a.n, ny = a.ny, H = a.height, W = a.width, P = H * W;
	double* sim = lds;                        // [(D+1)][kMaxDim]
	double* fsim = sim + (kMaxDim + 1) * kMaxDim;   // [D+1]
	double* xt = fsim + (kMaxDim + 1);        // trial points: xbar, xr, xe / xc [3][kMaxDim]
	double* x0 = xt + 3 * kMaxDim;            // warm start [kMaxDim]
	double* red = x0 + kMaxDim;               // [2][4] partial sums of the evaluations, [8] the convergence test's
	double* kn = red + 16;                    // [n + 4] knots of the first spline axis (every evaluation reads a few: LDS, not L2)
	double* kny = kn + (n + 4);               // [ny + 4]
	double* Kc = kny + (ny + 4);              // [kPool][kItems][25] cached polynomial coefficients
	int* keys = reinterpret_cast<int*>(Kc + (size_t)pool * kItems * 25);   // [kPool][2]
	int* nxt = keys + 2 * kPool;              // [kMaxPsfStars]
	int* rbs = nxt + kMaxPsfStars;            // [kMaxPsfStars]
	int* gkeys = rbs + kMaxPsfStars + 1;      // [kStoreSlots][3] tags of the target's store + [1] the slot filled next
	int* gsrc = gkeys + 3 * kStoreSlots + 1;  // [kMaxPsfStars]
	float* img = reinterpret_cast<float*>(gsrc + kMaxPsfStars);  // [P]   (4-byte aligned is enough)
	float* wgt = img + P;                     // [P]
	for (int q = tid; q < n + 4; q += kThreads) kn[q] = a.knots_x[q];
	for (int q = tid; q < ny + 4; q += kThreads) kny[q] = a.knots_y[q];
	if (tid < kMaxPsfStars) { nxt[tid] = 0; rbs[tid] = -1; }
	if (tid < 2 * pool) keys[tid] = -0x7fffffff;
	if (tid < 3 * kStoreSlots) gkeys[tid] = -0x7fffffff;
	if (tid == 0) gkeys[3 * kStoreSlots] = 0;
	const int64_t s0 = a.star_offsets[target];
	constexpr int ns = NS;   // (the host lists a target with more than five stars with the five-star ones: the first five are fitted)
	constexpr int D = 3 * ns;
	if (tid < D) x0[tid] = a.params0[s0 * 3 + tid];
	__syncthreads();
	const double h = kn[5] - kn[4], hy = kny[5] - kny[4];
	const double nan = __builtin_nan("");
	const uint8_t* mini = a.mini_aperture + (int64_t)target * P;
	const int64_t ob = (int64_t)target * a.out_pitch;
	if (ns == 0) { // nothing to fit: every minimisation of an empty vector "succeeds" at once; flux = aperture correction of the image
		for (int k = tid; k < a.n_cad; k += kThreads) { a.flux[ob + k] = nan; a.flux_err[ob + k] = nan; a.cen_row[ob + k] = nan; a.cen_col[ob + k] = nan; }
		if (tid == 0) a.status[target] = TP_STATUS_ERROR;
		return;
	}
	EvalCtx ec;
	ec.ns = ns; ec.n = n; ec.ny = ny; ec.H = H; ec.W = W; ec.pool = pool; ec.h = h; ec.hy = hy; ec.cutoff = a.cutoff;
	ec.Cg = a.coef + (int64_t)target * n * ny; ec.kn = kn; ec.kny = kny; ec.img = img; ec.wgt = wgt; ec.Kc = Kc; ec.red = red; ec.keys = keys; ec.nxt = nxt; ec.rbs = rbs;
	ec.gset = (a.gstore && !GEN && target < a.gstore_targets) ? (a.gstore + (size_t)target * kStoreSlots * kItems * 25) : nullptr; ec.gkeys = gkeys; ec.gsrc = gsrc;
	int flip = 0;
#define EVAL(xp) likelihood<(NS > 0 ? NS : 1), GEN>((xp), ec, flip)
	for (int k = 0; k < a.n_cad; ++k) {
		// ---- the cadence's image and weight map (float32 arithmetic of psf_photometry.py:75-86)
		const float* ip = a.images + (int64_t)target * P * a.t_pitch + k;
		const float* bp = a.backgrounds ? (a.backgrounds + (int64_t)target * P * a.t_pitch + k) : nullptr;
		for (int p = tid; p < P; p += kThreads) {
			const float im = ip[(int64_t)p * a.t_pitch];
			const float bk = bp ? bp[(int64_t)p * a.t_pitch] : 0.f;
			float var = fabsf(im + bk) + a.var_floor;
			if (var < 1e-9f) var = 1e-9f;
			float w = 1.0f / var;
			if (w < 1e-9f) w = 1e-9f;
			img[p] = im;
			wgt[p] = w;
		}
		__syncthreads();
		// ---- Nelder-Mead (scipy _minimize_neldermead)
		const int maxiter = (k > 0) ? a.maxiter : a.maxiter_first;
		// the simplex bookkeeping below is spread over the threads (vertex v = tid / 16, component d = tid % 16): every component
		// is computed by the same expression, in the same order, as scipy's vectorised numpy statements (a serial thread spent
		// two thirds of an iteration walking the 16 x 15 simplex through LDS)
		const int tv = tid >> 4, td = tid & 15;
		if (tv <= D && td < D) {
			const double y = x0[td];
			sim[tv * kMaxDim + td] = (tv >= 1 && td == tv - 1) ? ((y != 0.0) ? (1.0 + 0.05) * y : 0.00025) : y;
		}
		__syncthreads();
		for (int v = 0; v <= D; ++v) {
			const double f = EVAL(sim + v * kMaxDim);
			if (tid == 0) fsim[v] = f;
		}
		__syncthreads();
		auto sort_simplex = [&]() { // stable sort by fsim (numpy argsort of <= 16 values is an insertion sort)
			bool anynan = false;
			for (int v = 0; v <= D; ++v) anynan = anynan || (fsim[v] != fsim[v]);
			if (anynan) {
				// NaN does not order: replay the insertion sort itself, one thread
				if (tid == 0) {
					// (insertion by adjacent swaps: the same final order as numpy's insertion sort, and no per-thread array)
					for (int i = 1; i <= D; ++i) {
						for (int j = i - 1; j >= 0 && fsim[j] > fsim[j + 1]; --j) {
							const double f0 = fsim[j]; fsim[j] = fsim[j + 1]; fsim[j + 1] = f0;
							for (int d = 0; d < D; ++d) { const double v0 = sim[j * kMaxDim + d]; sim[j * kMaxDim + d] = sim[(j + 1) * kMaxDim + d]; sim[(j + 1) * kMaxDim + d] = v0; }
						}
					}
				}
				__syncthreads();
				return;
			}
			// the stable order as ranks: vertex v goes to the number of vertices that sort before it
			int rank = 0;
			double mine = 0.0, fmine = 0.0;
			if (tv <= D) {
				fmine = fsim[tv];
				for (int u = 0; u <= D; ++u) { const double fu = fsim[u]; rank += (fu < fmine || (fu == fmine && u < tv)) ? 1 : 0; }
				if (td < D) mine = sim[tv * kMaxDim + td];
			}
			__syncthreads();
			if (tv <= D) {
				if (td < D) sim[rank * kMaxDim + td] = mine;
				if (td == 15) fsim[rank] = fmine;
			}
			__syncthreads();
		};
		sort_simplex();
		int iterations = 1;
		while (iterations < maxiter) {
			// max |sim[1:] - sim[0]| and max |fsim[0] - fsim[1:]| (a NaN makes the maximum NaN, as numpy's does)
			double dx = 0.0, df = 0.0;
			if (tv >= 1 && tv <= D) {
				if (td < D) dx = fabs(sim[tv * kMaxDim + td] - sim[td]);
				if (td == 15) df = fabs(fsim[0] - fsim[tv]);
			}
#pragma unroll
			for (int off = 32; off > 0; off >>= 1) {
				const double ox = __shfl_xor(dx, off, 64), of = __shfl_xor(df, off, 64);
				if (ox > dx || ox != ox) dx = ox;
				if (of > df || of != of) df = of;
			}
			// (its own eight words: the evaluations' partial sums live elsewhere, and between two tests lies at least one barrier)
			double* red2 = red + 8;
			if ((tid & 63) == 0) { red2[tid >> 6] = dx; red2[4 + (tid >> 6)] = df; }
			__syncthreads();
			dx = red2[0]; df = red2[4];
#pragma unroll
			for (int w = 1; w < 4; ++w) {
				const double ox = red2[w], of = red2[4 + w];
				if (ox > dx || ox != ox) dx = ox;
				if (of > df || of != of) df = of;
			}
			if (dx <= 1e-4 && df <= 1e-4) break;
			double* xbar = xt; double* xr = xt + kMaxDim; double* xn = xt + 2 * kMaxDim;
			if (tid < D) {
				double sacc = 0.0;
				for (int v = 0; v < D; ++v) sacc += sim[v * kMaxDim + tid];   // np.add.reduce(sim[:-1], 0)
				xbar[tid] = sacc / (double)D;
				xr[tid] = 2.0 * xbar[tid] - sim[D * kMaxDim + tid];
			}
			// the values the decisions below compare with are taken NOW: the thread that accepts a point overwrites fsim[D] while
			// slower wavefronts may still be deciding (fxcc < fsim[D] read after that write sends them into the shrink branch: a race
			// the barrier that used to end every evaluation hid but did not exclude)
			const double f_best = fsim[0], f_second_worst = fsim[D - 1], f_worst = fsim[D];
			__syncthreads();
			const double fxr = EVAL(xr);
			bool doshrink = false;
			const double* take = nullptr; double ftake = 0.0;
			if (fxr < f_best) {
				if (tid < D) xn[tid] = 3.0 * xbar[tid] - 2.0 * sim[D * kMaxDim + tid];
				__syncthreads();
				const double fxe = EVAL(xn);
				if (fxe < fxr) { take = xn; ftake = fxe; } else { take = xr; ftake = fxr; }
			} else if (fxr < f_second_worst) {
				take = xr; ftake = fxr;
			} else if (fxr < f_worst) {
				if (tid < D) xn[tid] = 1.5 * xbar[tid] - 0.5 * sim[D * kMaxDim + tid];
				__syncthreads();
				const double fxc = EVAL(xn);
				if (fxc <= fxr) { take = xn; ftake = fxc; } else doshrink = true;
			} else {
				if (tid < D) xn[tid] = 0.5 * xbar[tid] + 0.5 * sim[D * kMaxDim + tid];
				__syncthreads();
				const double fxcc = EVAL(xn);
				if (fxcc < f_worst) { take = xn; ftake = fxcc; } else doshrink = true;
			}
			if (doshrink) {
				if (tv >= 1 && tv <= D && td < D) sim[tv * kMaxDim + td] = sim[td] + 0.5 * (sim[tv * kMaxDim + td] - sim[td]);
				__syncthreads();
				for (int v = 1; v <= D; ++v) {
					const double f = EVAL(sim + v * kMaxDim);
					if (tid == 0) fsim[v] = f;
				}
				__syncthreads();
			} else {
				if (tid < D) sim[D * kMaxDim + tid] = take[tid];
				if (tid == 0) fsim[D] = ftake;
				__syncthreads();
			}
			sort_simplex();
			++iterations;
		}
		const bool success = iterations < maxiter;
		// ---- result of the cadence (psf_photometry.py:157-196)
		double flux_ap = 0.0;
		if (success) {
			// residuals in the mini aperture: one more model evaluation at the solution
			StarR st[NS > 0 ? NS : 1];
			prepare_stars<(NS > 0 ? NS : 1), GEN>(sim, ec, st);
			double acc = 0.0;
			for (int p = tid; p < P; p += kThreads) {
				if (!mini[p]) continue;
				const int i = p / W, j = p - i * W;
				const double r = (double)img[p] - model_pixel<(NS > 0 ? NS : 1), GEN>(i, j, ec, st);
				if (r == r) acc += r;
			}
#pragma unroll
			for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
			double* redf = red + 4 * flip;
			flip ^= 1;
			if ((tid & 63) == 0) redf[tid >> 6] = acc;
			__syncthreads();
			flux_ap = (redf[0] + redf[1]) + (redf[2] + redf[3]);
		}
		if (tid == 0) {
			a.flux[ob + k] = success ? (sim[2] + flux_ap) : nan;
			a.flux_err[ob + k] = nan;
			a.cen_row[ob + k] = success ? sim[0] : nan;     // pos_centroid[k] = result[0, 0:2] = (row_stamp, column_stamp), as upstream
			a.cen_col[ob + k] = success ? sim[1] : nan;
			if (a.nit) a.nit[(int64_t)target * a.out_pitch + k] = iterations;
			if (success) for (int d = 0; d < D; ++d) x0[d] = sim[d];     // the next cadence starts from this solution
		}
		if (a.params_out && tid < D) a.params_out[((s0 * 3) + tid) * a.out_pitch + k] = success ? sim[tid] : nan;
		__syncthreads();
	}
#undef EVAL
	if (tid == 0) a.status[target] = TP_STATUS_OK; // psf_photometry.py:196
}

} // namespace


static int psf_fit_impl(tp_ctx* ctx, const tp_cube_desc* desc, const float* d_images, const float* d_backgrounds,
	const double* d_coef, const double* d_knots_x, const double* d_knots_y, int32_t n_coef_axis, int32_t n_coef_axis_y,
	const int64_t* d_star_offsets, const double* d_params0, const uint8_t* d_mini_aperture,
	double variance_floor, double cutoff_radius, int32_t maxiter_first, int32_t maxiter,
	double* d_flux, double* d_flux_err, double* d_centroid_row, double* d_centroid_col, int64_t out_pitch,
	double* d_params_out, int32_t* d_nit, int32_t* d_status)
{
	TP_CHECK_CTX(ctx);
	TP_API_BEGIN
	TP_REQUIRE(ctx, tp_desc_ok(desc), "tp_psf_fit: bad cube descriptor");
	TP_REQUIRE(ctx, d_images && d_coef && d_knots_x && d_knots_y && d_star_offsets && d_params0 && d_mini_aperture, "tp_psf_fit: null input pointer");
	TP_REQUIRE(ctx, d_flux && d_flux_err && d_centroid_row && d_centroid_col && d_status, "tp_psf_fit: null output pointer");
	TP_REQUIRE(ctx, out_pitch >= desc->n_cad, "tp_psf_fit: out_pitch < n_cad");
	TP_REQUIRE(ctx, n_coef_axis >= 4 && n_coef_axis <= 2048 && n_coef_axis_y >= 4 && n_coef_axis_y <= 2048, "tp_psf_fit: coefficient table must be 4..2048 per axis");
	TP_REQUIRE(ctx, cutoff_radius > 0, "tp_psf_fit: cutoff_radius must be positive (infinity = no cut-off, psf.py:142 `cutoff_radius is None`)");
	TP_REQUIRE(ctx, maxiter_first >= 1 && maxiter >= 1, "tp_psf_fit: bad iteration limits");
	if (desc->n_targets == 0 || desc->n_cad == 0) return TP_OK;
	const size_t P = (size_t)desc->height * desc->width;
	auto lds_bytes = [&](int pool) {
		const size_t doubles = (kMaxDim + 1) * kMaxDim + (kMaxDim + 1) + 3 * kMaxDim + kMaxDim + 16 + ((size_t)n_coef_axis + 4) + ((size_t)n_coef_axis_y + 4) + (size_t)pool * kItems * 25;
		return doubles * sizeof(double) + (2 * kPool + 2 * kMaxPsfStars + 1 + 3 * kStoreSlots + 1 + kMaxPsfStars) * sizeof(int) + 2 * P * sizeof(float) + 16;
	};
	// the targets by their number of fitted stars (one launch each, see psf_pool): the star offsets come to the host once, and
	// with them the knots: the cached biquartics need the SPOC layout of the PRF grid (9 samples per pixel, the cut-off inside
	// the evenly spaced knots, <= 5.25 as the cached item set assumes); any other grid or radius takes the general instantiation
	std::vector<int64_t> off((size_t)desc->n_targets + 1);
	std::vector<double> hk(((size_t)n_coef_axis + 4) + ((size_t)n_coef_axis_y + 4));
	TP_HIP(ctx, hipMemcpyAsync(off.data(), d_star_offsets, off.size() * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
	TP_HIP(ctx, hipMemcpyAsync(hk.data(), d_knots_x, ((size_t)n_coef_axis + 4) * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
	TP_HIP(ctx, hipMemcpyAsync(hk.data() + n_coef_axis + 4, d_knots_y, ((size_t)n_coef_axis_y + 4) * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
	TP_HIP(ctx, hipStreamSynchronize(ctx->stream));
	const bool general = !(n_coef_axis == n_coef_axis_y && n_coef_axis <= 140 && cutoff_radius <= 5.25 && uniform_grid_ok(hk.data(), n_coef_axis, cutoff_radius)
		&& uniform_grid_ok(hk.data() + n_coef_axis + 4, n_coef_axis, cutoff_radius));
	TP_REQUIRE(ctx, lds_bytes(general ? 0 : kPool) <= 160 * 1024, "tp_psf_fit: stamp too large for the LDS-resident image and weight map");
	std::vector<int32_t> lists[kMaxPsfStars + 1];
	for (int t = 0; t < desc->n_targets; ++t) {
		int ns = (int)(off[(size_t)t + 1] - off[(size_t)t]);
		ns = ns < 0 ? 0 : (ns > kMaxPsfStars ? kMaxPsfStars : ns);
		lists[ns].push_back(t);
	}
	int32_t* d_lists = static_cast<int32_t*>(tp_ctx_scratch(ctx, (size_t)desc->n_targets * sizeof(int32_t)));
	TP_REQUIRE(ctx, d_lists != nullptr, "tp_psf_fit: out of device memory for the target lists");
	PsfArgs a;
	a.images = d_images; a.backgrounds = d_backgrounds; a.n_cad = desc->n_cad; a.height = desc->height; a.width = desc->width; a.t_pitch = desc->t_pitch;
	a.coef = d_coef; a.knots_x = d_knots_x; a.knots_y = d_knots_y; a.n = n_coef_axis; a.ny = n_coef_axis_y;
	a.star_offsets = d_star_offsets; a.params0 = d_params0; a.mini_aperture = d_mini_aperture;
	a.var_floor = (float)variance_floor; a.cutoff = cutoff_radius; a.maxiter_first = maxiter_first; a.maxiter = maxiter;
	a.flux = d_flux; a.flux_err = d_flux_err; a.cen_row = d_centroid_row; a.cen_col = d_centroid_col; a.out_pitch = out_pitch;
	a.params_out = d_params_out; a.nit = d_nit; a.status = d_status;
	// the store of built coefficient sets (kStoreSlots per target, 349 KB): as many targets as 8 GiB hold (23 000); a target beyond
	// that rebuilds every set it needs, as every target did before round 6
	a.gstore = nullptr; a.gstore_targets = 0;
	const char* env_store = std::getenv("TESSPHOT_PSF_STORE");
	if (!general && !(env_store && env_store[0] == '0')) {
		const size_t per_target = (size_t)kStoreSlots * kItems * 25 * sizeof(double);
		const size_t n_store = std::min((size_t)desc->n_targets, ((size_t)8 << 30) / per_target);
		const size_t need = n_store * per_target;
		if (ctx->store_bytes < need) {
			if (ctx->store) (void)hipFree(ctx->store);
			ctx->store = nullptr; ctx->store_bytes = 0;
			if (tp_device_alloc(ctx, &ctx->store, need) == hipSuccess) ctx->store_bytes = need;
			else (void)hipGetLastError();            // no store: the fit runs without it
		}
		if (ctx->store_bytes >= need && need > 0) { a.gstore = static_cast<double*>(ctx->store); a.gstore_targets = (int)n_store; }
	}
	// the launches are independent: the context's stream and two side streams in turn, so that their tails overlap (every launch
	// ends with a few long-running workgroups on an otherwise idle chip)
	hipStream_t streams[3] = {ctx->stream, nullptr, nullptr};
	for (int i = 0; i < 2; ++i) {
		if (!ctx->side[i]) TP_HIP(ctx, hipStreamCreateWithFlags(&ctx->side[i], hipStreamNonBlocking));
		streams[i + 1] = ctx->side[i];
	}
	size_t at = 0;
	for (int ns = 0; ns <= kMaxPsfStars; ++ns) {
		if (lists[ns].empty()) continue;
		// (a failed copy must not be followed by launches on half-written lists; the copies already queued read host vectors
		// that die with this frame: wait for them)
		const hipError_t ce = hipMemcpyAsync(d_lists + at, lists[ns].data(), lists[ns].size() * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream);
		if (ce != hipSuccess) { (void)hipStreamSynchronize(ctx->stream); return ctx->fail(TP_ERR_HIP, "tp_psf_fit: target list copy", ce); }
		at += lists[ns].size();
	}
	hipEvent_t before = ctx->get_event();
	hipError_t err = hipEventRecord(before, ctx->stream);
	at = 0;
	int used = 0;
	bool waited[3] = {true, false, false};
	for (int ns = kMaxPsfStars; ns >= 0 && err == hipSuccess; --ns) {   // the longest fits first
		if (lists[ns].empty()) continue;
		size_t first = 0;
		for (int m = 0; m < ns; ++m) first += lists[m].size();
		const int pool = general ? 0 : psf_pool(ns);
		const int si = used++ % 3;
		if (!waited[si]) { err = hipStreamWaitEvent(streams[si], before, 0); waited[si] = true; if (err != hipSuccess) break; }
#define TP_PSF_LAUNCH_G(NSV, GENV) do { \
			err = hipFuncSetAttribute(reinterpret_cast<const void*>(tp_psf_fit_kernel<NSV, GENV>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes(pool)); \
			if (err == hipSuccess) { \
				TP_LAUNCH_ON(ctx, streams[si], TPK_PSF_FIT, (tp_psf_fit_kernel<NSV, GENV>), dim3((unsigned)lists[ns].size()), dim3(kThreads), lds_bytes(pool), a, (const int32_t*)(d_lists + first), pool); \
				err = hipGetLastError(); \
			} \
		} while (0)
#define TP_PSF_LAUNCH(NSV) do { if (general) TP_PSF_LAUNCH_G(NSV, true); else TP_PSF_LAUNCH_G(NSV, false); } while (0)
		switch (ns) {
			case 0: TP_PSF_LAUNCH(0); break;
			case 1: TP_PSF_LAUNCH(1); break;
			case 2: TP_PSF_LAUNCH(2); break;
			case 3: TP_PSF_LAUNCH(3); break;
			case 4: TP_PSF_LAUNCH(4); break;
			default: TP_PSF_LAUNCH(5); break;
		}
#undef TP_PSF_LAUNCH_G
#undef TP_PSF_LAUNCH
	}
	// the lists (host vectors, device copy) must outlive the copies and the launches
	for (int i = 2; i >= 0; --i) if (waited[i]) (void)hipStreamSynchronize(streams[i]);
	ctx->pool.push_back(before);
	if (err != hipSuccess) return ctx->fail(TP_ERR_HIP, "tp_psf_fit_kernel", err);
	return TP_OK;
	TP_API_END(ctx)
}

extern "C" int tp_psf_fit(tp_ctx* ctx, const tp_cube_desc* desc, const float* d_images, const float* d_backgrounds,
	const double* d_coef, const double* d_knots_x, const double* d_knots_y, int32_t n_coef_axis,
	const int64_t* d_star_offsets, const double* d_params0, const uint8_t* d_mini_aperture,
	double variance_floor, double cutoff_radius, int32_t maxiter_first, int32_t maxiter,
	double* d_flux, double* d_flux_err, double* d_centroid_row, double* d_centroid_col, int64_t out_pitch,
	double* d_params_out, int32_t* d_nit, int32_t* d_status)
{
	return psf_fit_impl(ctx, desc, d_images, d_backgrounds, d_coef, d_knots_x, d_knots_y, n_coef_axis, n_coef_axis, d_star_offsets, d_params0,
		d_mini_aperture, variance_floor, cutoff_radius, maxiter_first, maxiter, d_flux, d_flux_err, d_centroid_row, d_centroid_col, out_pitch,
		d_params_out, d_nit, d_status);
}

// the same for a PRF spline whose two axes have different numbers of samples (psf.py:119 takes any RectBivariateSpline): d_coef
// [n_targets][n_coef_axis_x * n_coef_axis_y], d_knots_x [n_coef_axis_x + 4], d_knots_y [n_coef_axis_y + 4]; the general instantiation
extern "C" int tp_psf_fit_xy(tp_ctx* ctx, const tp_cube_desc* desc, const float* d_images, const float* d_backgrounds,
	const double* d_coef, const double* d_knots_x, const double* d_knots_y, int32_t n_coef_axis_x, int32_t n_coef_axis_y,
	const int64_t* d_star_offsets, const double* d_params0, const uint8_t* d_mini_aperture,
	double variance_floor, double cutoff_radius, int32_t maxiter_first, int32_t maxiter,
	double* d_flux, double* d_flux_err, double* d_centroid_row, double* d_centroid_col, int64_t out_pitch,
	double* d_params_out, int32_t* d_nit, int32_t* d_status)
{
	return psf_fit_impl(ctx, desc, d_images, d_backgrounds, d_coef, d_knots_x, d_knots_y, n_coef_axis_x, n_coef_axis_y, d_star_offsets, d_params0,
		d_mini_aperture, variance_floor, cutoff_radius, maxiter_first, maxiter, d_flux, d_flux_err, d_centroid_row, d_centroid_col, out_pitch,
		d_params_out, d_nit, d_status);
}
