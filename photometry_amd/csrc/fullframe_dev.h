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

// scipy.ndimage.zoom(order 3, mode 'reflect', grid_mode = True) of the prefiltered mesh at one output pixel, clipped to the
// range of the mesh (photutils BkgZoomInterpolator), rounded to float32
__device__ __forceinline__ float zoom_value(const ZoomImage& z, int frame, int row, int col)
{
	const double* c = z.coef + (int64_t)frame * z.ny * z.nx;
	auto weights = [](double x, double (&w)[4], int& start) {
		const double fl = floor(x);
		start = (int)fl - 1;
		const double y = x - fl, zz = 1.0 - y;
		w[1] = (y * y * (y - 2.0) * 3.0 + 4.0) / 6.0;
		w[2] = (zz * zz * (zz - 2.0) * 3.0 + 4.0) / 6.0;
		w[0] = zz * zz * zz / 6.0;
		w[3] = 1.0 - w[0] - w[1] - w[2];
	};
	// (d c b a | a b c d | d c b a): an index is at most two outside [0, n), so one reflection does unless the mesh has a
	// single cell along the axis
	auto reflect = [](int i, int n) {
		if (n < 2) return 0;
		i = (i < 0) ? (-i - 1) : i;
		return (i >= n) ? (2 * n - 1 - i) : i;
	};
	double wy[4], wx[4];
	int sy, sx;
	weights(((double)row + 0.5) / (double)z.box - 0.5, wy, sy);
	weights(((double)col + 0.5) / (double)z.box - 0.5, wx, sx);
	int cx[4];
#pragma unroll
	for (int i = 0; i < 4; ++i) cx[i] = reflect(sx + i, z.nx);
	double acc = 0.0;
#pragma unroll
	for (int j = 0; j < 4; ++j) {
		const double* r = c + reflect(sy + j, z.ny) * z.nx;
		double t = 0.0;
#pragma unroll
		for (int i = 0; i < 4; ++i) t += wx[i] * r[cx[i]];
		acc += wy[j] * t;
	}
	const double lo = z.vmin[frame], hi = z.vmax[frame];
	acc = (acc < lo) ? lo : ((acc > hi) ? hi : acc);
	return (float)acc;
}

struct RadialSpline {
	double col_offset, xcen, ycen;
	const double* knots; const double* coefs; const int32_t* n_knots; int max_knots;   // per frame: FITPACK knots t[0..n), coefficients c[0..n-4)
	const double* zeropoint;
};

// 10**spline(r) - zeropoint at one pixel (backgrounds.py:186-188; ext = 3: the boundary value outside the end knots); t / c: the
// frame's knots and coefficients (the callers stage them in LDS), n its number of knots (< 8: no radial component, 0)
__device__ __forceinline__ double radial_value(const double* t, const double* c, int n, double zeropoint, double col_offset, double xcen, double ycen,
	int row, int col)
{
	if (n < 8) return 0.0;
	const double dx = ((double)col + col_offset) - xcen, dy = (double)row - ycen;
	double x = sqrt(dx * dx + dy * dy);
	x = fmin(fmax(x, t[3]), t[n - 4]);
	// interval t[l] <= x < t[l + 1], 3 <= l <= n - 5
	int l = 3, h = n - 4;
	while (h - l > 1) { const int m = (l + h) >> 1; if (x >= t[m]) l = m; else h = m; }
	// de Boor, cubic
	double d0 = c[l - 3], d1 = c[l - 2], d2 = c[l - 1], d3 = c[l];
	double al;
	al = (x - t[l]) / (t[l + 3] - t[l]);         d3 = (1.0 - al) * d2 + al * d3;
	al = (x - t[l - 1]) / (t[l + 2] - t[l - 1]); d2 = (1.0 - al) * d1 + al * d2;
	al = (x - t[l - 2]) / (t[l + 1] - t[l - 2]); d1 = (1.0 - al) * d0 + al * d1;
	al = (x - t[l]) / (t[l + 2] - t[l]);         d3 = (1.0 - al) * d2 + al * d3;
	al = (x - t[l - 1]) / (t[l + 1] - t[l - 1]); d2 = (1.0 - al) * d1 + al * d2;
	al = (x - t[l]) / (t[l + 1] - t[l]);         d3 = (1.0 - al) * d2 + al * d3;
	return pow(10.0, d3) - zeropoint;
}
