// linpsf_common.h -- definitions shared by the LinPSF translation units (linpsf.hip: plan / coefficient store / vector-ALU
// fit kernels and the C entry; linpsf_mfma.hip: the matrix-core fit kernel).
#pragma once
#include "common.h"
#include "linpsf_dev.h"
#include <cmath>

namespace tp_linpsf {

using namespace tp_prf;

constexpr int kMaxStars = 8;      // register-resident vector-ALU kernels (fit2 / direct)
constexpr int kMfmaStars = 4;     // matrix-core kernel (linpsf_mfma.hip)
constexpr int kMfmaPixels = 256;  // pixels of a target's union list U (16 tiles of 16)

//--------------------------------------------------------------------------------------------------
// P2..P4
//--------------------------------------------------------------------------------------------------
struct FitArgs {
	const float* images; const float* subtract; int64_t subtract_pitch;
	int n_cad, height, width; int64_t t_pitch;
	const double* coef;          // [n_targets][n*n]
	const double* knots_x;       // [n+4] knots along the first spline axis (columns)
	const double* knots_y;       // [n+4] knots along the second spline axis (rows)
	int n;                       // coefficients per axis (117)
	int ny;                      // ... along the second axis (= n for every kernel but the general ones: tp_linpsf_fit_xy)
	const int64_t* star_offsets; // [n_targets+1] into the fitted-star arrays
	const int32_t* target_index; // [n_targets] index of the main target inside its fitted stars
	const double* pos_row;       // [n_fit_stars][pos_pitch] row_stamp per cadence
	const double* pos_col;       // [n_fit_stars][pos_pitch]
	int64_t pos_pitch;
	double cutoff;
	double* flux;                // [n_targets][out_pitch]  lightcurve flux (target star)
	double* flux_err;            // [n_targets][out_pitch]  NaN (linpsf_photometry.py:169)
	double* fluxes_all;          // [n_fit_stars][out_pitch] fitted flux of every star (for the mean fluxes)
	int64_t out_pitch;
};

// Cyclic Jacobi eigen-decomposition based pseudo-inverse solve:  x = pinv(G) g,  G symmetric S x S.
template <int S>
__device__ __forceinline__ void pinv_solve(double (&G)[S][S], const double (&g)[S], int ns, double (&x)[S])
{
	double V[S][S];
#pragma unroll
	for (int i = 0; i < S; ++i)
#pragma unroll
		for (int j = 0; j < S; ++j) V[i][j] = (i == j) ? 1.0 : 0.0;
	for (int sweep = 0; sweep < 30; ++sweep) {
		double off = 0.0;
#pragma unroll
		for (int p = 0; p < S; ++p)
#pragma unroll
			for (int q = p + 1; q < S; ++q) if (q < ns) off += G[p][q] * G[p][q];
		double d2 = 0.0;
#pragma unroll
		for (int p = 0; p < S; ++p) if (p < ns) d2 += G[p][p] * G[p][p];
		if (!(off > 1e-34 * d2)) break; // off-diagonal below 1e-17 relative: converged (or NaN)
#pragma unroll
		for (int p = 0; p < S; ++p) {
#pragma unroll
			for (int q = p + 1; q < S; ++q) {
				if (q >= ns) continue;
				const double apq = G[p][q];
				if (apq == 0.0) continue;
				const double theta = (G[q][q] - G[p][p]) / (2.0 * apq);
				const double t = ((theta >= 0.0) ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
				const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
				for (int k = 0; k < S; ++k) {
					const double gkp = G[k][p], gkq = G[k][q];
					G[k][p] = c * gkp - s * gkq;
					G[k][q] = s * gkp + c * gkq;
				}
#pragma unroll
				for (int k = 0; k < S; ++k) {
					const double gpk = G[p][k], gqk = G[q][k];
					G[p][k] = c * gpk - s * gqk;
					G[q][k] = s * gpk + c * gqk;
				}
#pragma unroll
				for (int k = 0; k < S; ++k) {
					const double vkp = V[k][p], vkq = V[k][q];
					V[k][p] = c * vkp - s * vkq;
					V[k][q] = s * vkp + c * vkq;
				}
			}
		}
	}
	// numpy.linalg.pinv: singular values (= |eigenvalues|) <= 1e-15 * max are treated as zero
	double smax = 0.0;
#pragma unroll
	for (int i = 0; i < S; ++i) if (i < ns) { const double a = fabs(G[i][i]); if (a > smax || a != a) smax = a; }
	const double cut = 1e-15 * smax;
#pragma unroll
	for (int i = 0; i < S; ++i) x[i] = 0.0;
#pragma unroll
	for (int e = 0; e < S; ++e) {
		if (e >= ns) continue;
		const double lam = G[e][e];
		double proj = 0.0;
#pragma unroll
		for (int k = 0; k < S; ++k) if (k < ns) proj += V[k][e] * g[k];
		const double inv = (fabs(lam) > cut) ? (1.0 / lam) : ((lam != lam) ? lam : 0.0);
		const double coef = proj * inv;
#pragma unroll
		for (int k = 0; k < S; ++k) if (k < ns) x[k] += V[k][e] * coef;
	}
}


// The same solve for the well-conditioned case: Cholesky.  Returns false (x untouched) when a pivot is not safely positive
// -- singular, nearly singular or NaN normal equations -- and the caller takes pinv_solve, whose cut-off then matters
// (numpy.linalg.pinv, linpsf_photometry.py:22-34).  Where it succeeds the two agree to rounding times the condition number
// (< 1e4 here: pivots below 1e-4 of their diagonal element are refused).
template <int S>
__device__ __forceinline__ bool chol_solve(const double (&G)[S][S], const double (&g)[S], double (&x)[S])
{
	double L[S][S], y[S];
	bool ok = true;
#pragma unroll
	for (int j = 0; j < S; ++j) {
		double d = G[j][j];
#pragma unroll
		for (int k = 0; k < j; ++k) d -= L[j][k] * L[j][k];
		ok = ok && (d > 1e-4 * G[j][j]);
		const double inv = 1.0 / sqrt(d);
		L[j][j] = inv;   // the reciprocal of the diagonal element
#pragma unroll
		for (int i = j + 1; i < S; ++i) {
			double v = G[i][j];
#pragma unroll
			for (int k = 0; k < j; ++k) v -= L[i][k] * L[j][k];
			L[i][j] = v * inv;
		}
	}
	if (!ok) return false;
#pragma unroll
	for (int i = 0; i < S; ++i) {
		double v = g[i];
#pragma unroll
		for (int k = 0; k < i; ++k) v -= L[i][k] * y[k];
		y[i] = v * L[i][i];
	}
#pragma unroll
	for (int i = S - 1; i >= 0; --i) {
		double v = y[i];
#pragma unroll
		for (int k = i + 1; k < S; ++k) v -= L[k][i] * x[k];
		x[i] = v * L[i][i];
	}
	return true;
}

// plan of one fitted star: the table origins its cadences visit and the pixels its cut-off circle can reach
struct StarPlan { int axmin, bymin, nby, nc, jmin, jmax, imin, imax; long long item_off; };

// matrix-core path, per target: the pixels inside the cut-off of ANY fitted star at ANY cadence form the list U (ordered by
// which stars reach them -- Gray-code order of the membership bits, then raster -- so that the pixels of one star are
// contiguous), cut into tiles of 16; star s touches the tiles of `tiles[s]`.
struct MPlan {
	int32_t n_pix, n_tiles;
	uint32_t tiles[kMfmaStars];
	uint32_t edge_tiles[kMfmaStars];   // tiles with a pixel that is inside the star's cut-off at some cadences only
	int32_t n_seg;                     // segments of the series (records target * kMfmaSegs .. + n_seg of the segment array)
};
// The series of a target is cut into SEGMENTS of consecutive 16-cadence tiles inside which every fitted star visits at most
// kMfmaSpan knot intervals per axis: a star that drifts across the pixel during the series (pointing drift, velocity aberration:
// half a pixel is 4.5 knot intervals) stays on the matrix cores, each stretch of the series with the spline of the intervals it
// visits THEN (round 3 sent such a target to the vector-ALU kernels).  Without drift the jitter gives one segment.
// Per (target, segment): the coefficients of ONE tensor-product quartic spline per star over its na x nb intervals
// (linpsf_mfma.hip), laid out as the A operands of the matrix instruction: [star][rank of the tile among its tiles][step][64
// lanes] doubles, the whole segment contiguous from `koff` (that image is copied to LDS as it is), star s from
// `koff + 64 * ksub[s]`, `mfma_steps(na, nb)` steps per tile.  One workgroup of the fit kernel per segment.
struct SegPlan {
	int32_t target;
	int32_t tile0, tile1;              // 16-cadence tiles [tile0, tile1) of the series
	int32_t kdoubles;                  // size of the segment's image (a multiple of 64)
	int64_t koff;                      // doubles from the start of the matrix-core store
	int32_t axmin[kMfmaStars], bymin[kMfmaStars];   // first knot interval the star visits in this segment, per axis
	uint16_t ksub[kMfmaStars];         // in blocks of 64 doubles
	uint8_t na[kMfmaStars], nb[kMfmaStars];   // knot intervals visited along x / y (1..3; 0: the star is never on the stamp)
};
constexpr int kMfmaSpan = 3;          // knot intervals per axis a star may visit inside a segment
constexpr int kMfmaSegs = 8;          // segments per target (more: the vector-ALU kernels take the target)
constexpr int kMfmaCadTiles = 256;    // 16-cadence tiles of a series the plan kernel can cut into segments (4096 cadences: a sector at 600 s)
// steps of v_mfma_f64_16x16x4_f64 per (star, pixel tile): (4 + na) basis functions of x times the first four of y, then two steps
// for each of the nb remaining basis functions of y -- except for the commonest case, 2 x 2 intervals (36 products), which is
// packed into 9 steps instead of 10: the half-empty second step of y basis function 4 also carries x basis functions 0, 1 of y
// basis function 5, and one more step the other four (mfma_is22)
__host__ __device__ constexpr bool mfma_is22(int na, int nb) { return na == 2 && nb == 2; }
__host__ __device__ constexpr int mfma_steps(int na, int nb) { return mfma_is22(na, nb) ? 9 : ((4 + na) + 2 * nb); }
// LDS bytes for the coefficient image of a segment: "small" leaves room for two workgroups per CU, "large" (three and four
// stars only: their kernels run one workgroup per CU anyway) takes the LDS of the CU
constexpr int kMfmaLdsSmall = 75776, kMfmaLdsLarge = 157696;
// the plan kernel lists the targets and the segments of the matrix-core path by their number of fitted stars: class = stars - 1
constexpr int kMfmaClasses = kMfmaStars;
// counters the plan kernel keeps (64-bit words of one 256-byte block): kTotClass0 + c targets, kTotSeg0 + c segments of class c
enum { kTotPolyItems = 0, kTotKDoubles = 1, kTotPolyTargets = 2, kTotDirectTargets = 3, kTotGeneral = 4, kTotClass0 = 8, kTotSeg0 = 16, kTotCount = 24 };

// `todo` flag of a target (written by the plan kernel): which kernel fits it
enum { kPathPoly = 0, kPathDirect = 1, kPathMfma = 2 };

// linpsf_mfma.hip
int fit_mfma_launch(tp_ctx* ctx, const FitArgs& a, int n_targets, const unsigned long long* seg_counts, const unsigned long long* class_counts, const SegPlan* d_segs,
	const int32_t* d_seg_lists, const MPlan* d_mplans, const uint16_t* d_ulist, const uint8_t* d_usig, const double* d_kstore, double* d_alast);

} // namespace tp_linpsf
