// fullframe_dev.h -- images of the full-frame background path that are EVALUATED WHERE THEY ARE READ instead of stored:
//   * the zoomed mesh (the square component: tp_background_zoom's output) from its cubic B-spline coefficients,
//   * the radial component (tp_radial_evaluate's output) from the ring profile's spline.
// The alternation of fit_background's TESS branch (backgrounds.py:162-206) reads each of them two or three times per
// iteration; materialised they cost a 16 MB write and as many 16 MB reads per 2048 x 2048 frame (round 4: 304 MB of traffic per
// frame against 117 MB of algorithmic bytes).  Both functions return exactly the float32 the materialising kernels store.
#pragma once
#include "common.h"

struct ZoomImage {
	const double* coef; const double* vmin; const double* vmax;   // [frame][ny][nx], [frame], [frame] (tp_background_mesh_finish)
	int ny, nx, box, n_cols;                                      // n_cols: columns of the frame (pixel index -> row, column)
};

// One axis of scipy.ndimage.zoom(order 3, mode 'reflect', grid_mode = True) at an output position: the four cubic B-spline
// weights and the (reflected) indices of the coefficients they multiply.  (d c b a | a b c d | d c b a): an index is at most two
// outside [0, n), so one reflection does unless the mesh has a single cell along the axis.
struct ZoomAxis { double w[4]; int idx[4]; int start; };

__device__ __forceinline__ void zoom_axis(int pos, int box, int n, ZoomAxis& a)
{
	const double x = ((double)pos + 0.5) / (double)box - 0.5;
	const double fl = floor(x);
	a.start = (int)fl - 1;
	const double y = x - fl, zz = 1.0 - y;
	a.w[1] = (y * y * (y - 2.0) * 3.0 + 4.0) / 6.0;
	a.w[2] = (zz * zz * (zz - 2.0) * 3.0 + 4.0) / 6.0;
	a.w[0] = zz * zz * zz / 6.0;
	a.w[3] = 1.0 - a.w[0] - a.w[1] - a.w[2];
#pragma unroll
	for (int i = 0; i < 4; ++i) {
		int k = a.start + i;
		if (n < 2) k = 0;
		else { k = (k < 0) ? (-k - 1) : k; k = (k >= n) ? (2 * n - 1 - k) : k; }
		a.idx[i] = k;
	}
}

// the contraction of one row of coefficients with the column weights
__device__ __forceinline__ double zoom_row_sum(const double* crow, const ZoomAxis& ax)
{
	double t = 0.0;
#pragma unroll
	for (int i = 0; i < 4; ++i) t += ax.w[i] * crow[ax.idx[i]];
	return t;
}

// the row sums of the four coefficient rows an output row touches: they depend on the column and on ay.start only, so a thread
// that walks down a column recomputes them once per mesh cell (zoom_from_sums gives the same value as zoom_value, operation for
// operation)
__device__ __forceinline__ void zoom_sums(const ZoomImage& z, int frame, const ZoomAxis& ay, const ZoomAxis& ax, double (&T)[4])
{
	const double* c = z.coef + (int64_t)frame * z.ny * z.nx;
#pragma unroll
	for (int j = 0; j < 4; ++j) T[j] = zoom_row_sum(c + ay.idx[j] * z.nx, ax);
}

__device__ __forceinline__ float zoom_from_sums(const ZoomAxis& ay, const double (&T)[4], double lo, double hi)
{
	double acc = 0.0;
#pragma unroll
	for (int j = 0; j < 4; ++j) acc += ay.w[j] * T[j];
	acc = (acc < lo) ? lo : ((acc > hi) ? hi : acc);
	return (float)acc;
}

// the zoomed mesh at one output pixel, clipped to the range of the mesh (photutils BkgZoomInterpolator), rounded to float32
__device__ __forceinline__ float zoom_value(const ZoomImage& z, int frame, int row, int col)
{
	ZoomAxis ay, ax;
	zoom_axis(row, z.box, z.ny, ay);
	zoom_axis(col, z.box, z.nx, ax);
	double T[4];
	zoom_sums(z, frame, ay, ax, T);
	return zoom_from_sums(ay, T, z.vmin[frame], z.vmax[frame]);
}

struct RadialSpline {
	double col_offset, xcen, ycen;
	const double* knots; const double* coefs; const int32_t* n_knots; int max_knots;   // per frame: FITPACK knots t[0..n), coefficients c[0..n-4)
	const double* zeropoint;
};

// The ring profile of a frame staged for evaluation: knots t[0..n), coefficients c[0..n-4) and, per knot interval l, the six
// reciprocals of the knot differences de Boor's recursion divides by (one division per interval and difference instead of six
// per pixel: the quotient and the product by the correctly rounded reciprocal differ by an ulp at most, 1e-16 in the exponent
// of a value that is stored as float32).  `stage` = kMaxKnotsStaged-strided arrays in LDS, filled by stage_radial().
constexpr int kRadialStageDoubles(int max_knots) { return 8 * max_knots; }

__device__ __forceinline__ void stage_radial(double* stage, int max_knots, const RadialSpline& sp, int frame, int n, int tid, int nthreads)
{
	double* t = stage;
	double* c = stage + max_knots;
	double* inv = stage + 2 * max_knots;      // [l][6]
	for (int i = tid; i < n; i += nthreads) {
		t[i] = sp.knots[(int64_t)frame * sp.max_knots + i];
		c[i] = sp.coefs[(int64_t)frame * sp.max_knots + i];
	}
	__syncthreads();
	for (int l = 3 + tid; l <= n - 5; l += nthreads) {
		inv[6 * l + 0] = 1.0 / (t[l + 3] - t[l]);
		inv[6 * l + 1] = 1.0 / (t[l + 2] - t[l - 1]);
		inv[6 * l + 2] = 1.0 / (t[l + 1] - t[l - 2]);
		inv[6 * l + 3] = 1.0 / (t[l + 2] - t[l]);
		inv[6 * l + 4] = 1.0 / (t[l + 1] - t[l - 1]);
		inv[6 * l + 5] = 1.0 / (t[l + 1] - t[l]);
	}
	__syncthreads();
}

// 10**spline(r) - zeropoint at one pixel (backgrounds.py:186-188; ext = 3: the boundary value outside the end knots) from the
// staged profile; n = the frame's number of knots (< 8: no radial component, 0)
__device__ __forceinline__ double radial_value(const double* stage, int max_knots, int n, double zeropoint, double col_offset, double xcen, double ycen,
	int row, int col)
{
	if (n < 8) return 0.0;
	const double* t = stage;
	const double* c = stage + max_knots;
	const double dx = ((double)col + col_offset) - xcen, dy = (double)row - ycen;
	double x = sqrt(dx * dx + dy * dy);
	x = fmin(fmax(x, t[3]), t[n - 4]);
	// interval t[l] <= x < t[l + 1], 3 <= l <= n - 5
	int l = 3, h = n - 4;
	while (h - l > 1) { const int m = (l + h) >> 1; if (x >= t[m]) l = m; else h = m; }
	const double* iv = stage + 2 * max_knots + 6 * l;
	// de Boor, cubic
	double d0 = c[l - 3], d1 = c[l - 2], d2 = c[l - 1], d3 = c[l];
	double al;
	al = (x - t[l]) * iv[0];     d3 = (1.0 - al) * d2 + al * d3;
	al = (x - t[l - 1]) * iv[1]; d2 = (1.0 - al) * d1 + al * d2;
	al = (x - t[l - 2]) * iv[2]; d1 = (1.0 - al) * d0 + al * d1;
	al = (x - t[l]) * iv[3];     d3 = (1.0 - al) * d2 + al * d3;
	al = (x - t[l - 1]) * iv[4]; d2 = (1.0 - al) * d1 + al * d2;
	al = (x - t[l]) * iv[5];     d3 = (1.0 - al) * d2 + al * d3;
	return exp10(d3) - zeropoint;
}
