// linpsf_dev.h -- the pixel-integrated PRF (P2: PSF.integrate_to_image, photometry/psf.py:122-148) as device functions shared by
// the linear (linpsf.hip) and the non-linear (psfphot.hip) PSF photometry kernels.  See linpsf.hip for the derivation: on the
// uniform 9-samples-per-pixel PRF grid the FITPACK box integral of a pixel is separable and its 13 non-zero basis integrals per
// axis are [1-M(phi+3..phi), 1,1,1,1,1, M(phi+3..phi)] * h with M the cumulative cardinal cubic B-spline.
#pragma once
#include <cmath>

namespace tp_prf {

// cumulative cardinal cubic B-spline M(z) = int_0^z N(t) dt, N supported on [0, 4]
__device__ __forceinline__ double cumspline01(double w) { const double w2 = w * w; return w2 * w2 / 24.0; }               // z in [0,1], w = z
__device__ __forceinline__ double cumspline12(double w) { return 1.0 / 24.0 + (w + 1.5 * w * w + w * w * w - 0.75 * (w * w) * (w * w)) / 6.0; } // z in [1,2], w = z-1

// pixel-edge weights of one axis for a star at stamp coordinate `pos`: m[k] = M(phi + k), k = 0..3,
// and `first` such that pixel j uses table rows first + 9*j .. first + 9*j + 12.
// phi is the same for every pixel (pixels are 9 knot intervals apart); it is measured at the pixel
// nearest to the star, whose lower edge is guaranteed to lie inside the uniform part of the knot vector.
__device__ __forceinline__ void axis_weights(const double* kn, int n, double pos, double h, double (&m)[4], int& first)
{
	if (!(fabs(pos) < 1e6)) { m[0] = m[1] = m[2] = m[3] = 0.0; first = 4; return; } // NaN / absurd position: never inside the cut-off (psf.py:142)
	const int jstar = (int)rint(pos);
	const double x0 = ((double)jstar - pos) - 0.5;   // lower edge of pixel jstar relative to the star, in [-1, 0]
	// knot interval l with kn[l] <= x0 < kn[l+1]  (uniform interior knots, spacing h)
	int l = 4 + (int)floor((x0 - kn[4]) / h);
	if (l < 4) l = 4;
	if (l > n - 2) l = n - 2;
	if (x0 < kn[l] && l > 4) --l;
	else if (x0 >= kn[l + 1] && l < n - 2) ++l;
	const double phi = (x0 - kn[l]) / (kn[l + 1] - kn[l]);
	m[0] = cumspline01(phi);
	m[1] = cumspline12(phi);
	m[2] = 1.0 - cumspline12(1.0 - phi);  // M(2+phi) = 1 - M(2-phi), 2-phi in (1,2]
	m[3] = 1.0 - cumspline01(1.0 - phi);  // M(3+phi) = 1 - M(1-phi)
	first = (l - 3) - 9 * jstar;           // table index of weight p = 0 for pixel 0
}

// value of the pixel-integrated unit PRF of one star at pixel (i, j): h^2 * sum_pq wx[p] wy[q] C[ax+p][by+q]
__device__ __forceinline__ double prf_pixel(const double* __restrict__ C, int n, int ax, int by,
	const double (&mx)[4], const double (&my)[4])
{
	const double* c0 = C + (int64_t)ax * n + by;
	double acc = 0.0;
#pragma unroll
	for (int p = 0; p < 13; ++p) {
		const double* r = c0 + p * n;
		// inner contraction over q with weights [1-m3, 1-m2, 1-m1, 1-m0, 1,1,1,1,1, m3, m2, m1, m0]
		double t = ((r[4] + r[5]) + (r[6] + r[7])) + r[8];
		t += (r[0] + my[3] * (r[9] - r[0]));
		t += (r[1] + my[2] * (r[10] - r[1]));
		t += (r[2] + my[1] * (r[11] - r[2]));
		t += (r[3] + my[0] * (r[12] - r[3]));
		double wx;
		if (p < 4) wx = 1.0 - mx[3 - p];
		else if (p < 9) wx = 1.0;
		else wx = mx[12 - p];
		acc += wx * t;
	}
	return acc;
}

// ---- polynomial form (see linpsf.hip): for fixed knot intervals of a star's sub-pixel phases the pixel-integrated PRF of a
// pixel is a biquartic in the two phases; these are the quartics of the 13 pixel-edge weights
// coefficients (powers 0..4 of phi) of the 13 pixel-edge weights [1-m3, 1-m2, 1-m1, 1-m0, 1,1,1,1,1, m3, m2, m1, m0]
static __constant__ double kEdgePoly[13][5] = {
	{1.0 / 24.0, -1.0 / 6.0, 0.25, -1.0 / 6.0, 1.0 / 24.0},
	{0.5, -2.0 / 3.0, 0.0, 1.0 / 3.0, -0.125},
	{23.0 / 24.0, -1.0 / 6.0, -0.25, -1.0 / 6.0, 0.125},
	{1.0, 0.0, 0.0, 0.0, -1.0 / 24.0},
	{1.0, 0.0, 0.0, 0.0, 0.0}, {1.0, 0.0, 0.0, 0.0, 0.0}, {1.0, 0.0, 0.0, 0.0, 0.0}, {1.0, 0.0, 0.0, 0.0, 0.0}, {1.0, 0.0, 0.0, 0.0, 0.0},
	{23.0 / 24.0, 1.0 / 6.0, -0.25, 1.0 / 6.0, -1.0 / 24.0},
	{0.5, 2.0 / 3.0, 0.0, -1.0 / 3.0, 0.125},
	{1.0 / 24.0, 1.0 / 6.0, 0.25, 1.0 / 6.0, -0.125},
	{0.0, 0.0, 0.0, 0.0, 1.0 / 24.0},
};

// phase and table origin of one axis (same arithmetic as axis_weights); false for a NaN / absurd position
__device__ __forceinline__ bool axis_phase(const double* kn, int n, double pos, double h, double& phi, int& first)
{
	phi = 0.0; first = 4;
	if (!(fabs(pos) < 1e6)) return false;
	const int jstar = (int)rint(pos);
	const double x0 = ((double)jstar - pos) - 0.5;
	int l = 4 + (int)floor((x0 - kn[4]) / h);
	if (l < 4) l = 4;
	if (l > n - 2) l = n - 2;
	if (x0 < kn[l] && l > 4) --l;
	else if (x0 >= kn[l + 1] && l < n - 2) ++l;
	phi = (x0 - kn[l]) / (kn[l + 1] - kn[l]);
	first = (l - 3) - 9 * jstar;
	return true;
}


// columns b0 .. b0 + NB - 1 of the same 25 coefficients with ONE pass over the patch (poly_column reads the 169 table values once per
// column: five passes for a full set; the non-linear PSF kernel rebuilds its cached sets from the table in L2 and was bound by
// exactly that traffic).  Same sums in the same order as poly_column.
template <int NB>
__device__ __forceinline__ void poly_columns(const double* __restrict__ C, int n, int ax, int by, int b0, double h2, double (&out)[NB][5])
{
	double kk[NB][5];
#pragma unroll
	for (int b = 0; b < NB; ++b)
#pragma unroll
		for (int e = 0; e < 5; ++e) kk[b][e] = 0.0;
	const double* c0 = C + (int64_t)ax * n + by;
#pragma unroll 1
	for (int p0 = 0; p0 < 13; p0 += 4) {
		double rv[4][13];
#pragma unroll
		for (int u = 0; u < 4; ++u) {
			const int pp = (p0 + u < 13) ? (p0 + u) : 12;
			const double* r = c0 + pp * n;
#pragma unroll
			for (int q = 0; q < 13; ++q) rv[u][q] = r[q];
		}
#pragma unroll
		for (int u = 0; u < 4; ++u) {
			if (p0 + u < 13) {
#pragma unroll
				for (int b = 0; b < NB; ++b) {
					double t = 0.0;
#pragma unroll
					for (int q = 0; q < 13; ++q) t = __builtin_fma(kEdgePoly[q][b0 + b], rv[u][q], t);
#pragma unroll
					for (int e = 0; e < 5; ++e) kk[b][e] = __builtin_fma(kEdgePoly[p0 + u][e], t, kk[b][e]);
				}
			}
		}
	}
#pragma unroll
	for (int b = 0; b < NB; ++b)
#pragma unroll
		for (int e = 0; e < 5; ++e) out[b][e] = h2 * kk[b][e];
}

// the 25 coefficients K[e][b] (e: power of phi_x, b: power of phi_y), scaled by h2, of the patch C[ax .. ax+12][by .. by+12];
// one call computes column b (the sums run over q inside, over p outside: the order of tp_linpsf_coef_kernel)
__device__ __forceinline__ void poly_column(const double* __restrict__ C, int n, int ax, int by, int bcol, double h2, double (&out)[5])
{
	double kk[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
	const double* c0 = C + (int64_t)ax * n + by;
	// four table rows (52 loads) in flight at a time: the table is read from L2 at low occupancy, a row at a time would be 13
	// dependent round trips
#pragma unroll 1
	for (int p0 = 0; p0 < 13; p0 += 4) {
		double rv[4][13];
#pragma unroll
		for (int u = 0; u < 4; ++u) {
			const int pp = (p0 + u < 13) ? (p0 + u) : 12;
			const double* r = c0 + pp * n;
#pragma unroll
			for (int q = 0; q < 13; ++q) rv[u][q] = r[q];
		}
#pragma unroll
		for (int u = 0; u < 4; ++u) {
			if (p0 + u < 13) {
				double t = 0.0;
#pragma unroll
				for (int q = 0; q < 13; ++q) t = __builtin_fma(kEdgePoly[q][bcol], rv[u][q], t);
#pragma unroll
				for (int e = 0; e < 5; ++e) kk[e] = __builtin_fma(kEdgePoly[p0 + u][e], t, kk[e]);
			}
		}
	}
#pragma unroll
	for (int e = 0; e < 5; ++e) out[e] = h2 * kk[e];
}

// Horner evaluation of the biquartic with coefficients kp[e * 5 + d]
__device__ __forceinline__ double poly_eval(const double* kp, double phx, double phy)
{
	double val = 0.0;
#pragma unroll
	for (int e = 4; e >= 0; --e) {
		double inner = kp[e * 5 + 4];
#pragma unroll
		for (int d = 3; d >= 0; --d) inner = __builtin_fma(inner, phy, kp[e * 5 + d]);
		val = __builtin_fma(val, phx, inner);
	}
	return val;
}


// ---- any knot vector, any cut-off radius (psf.py:122-148 takes whatever RectBivariateSpline was built on): the FITPACK box integral
// (dblint -> fpintb) written for the device.  integral over [a, b] of the cubic B-spline N_i on the knots t[i .. i + 4] is
// J_i(b) - J_i(a) with J_i(e) = (t[i + 4] - t[i]) / 4 * sum_{j >= i} N4_j(e), N4 the quartic B-splines on the same knots: at an
// edge e in [t[l], t[l + 1]) the sum is 1 for i <= l - 4, 0 for i > l, and a partial sum of the five non-zero quartics for
// i = l - 3 .. l.  Like fpintb the limits are cut to [t[3], t[n]] (n = number of coefficients of the axis).
struct EdgeInt { int l; double s[4]; };      // s[q]: the sum for i = l - 3 + q

__device__ inline void edge_integrals(const double* __restrict__ t, int n, double e, EdgeInt& out)
{
	// knot interval of e: the last l in [3, n - 1] with t[l] <= e (e is already inside [t[3], t[n]])
	int lo = 3, hi = n - 1;
	while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (t[mid] <= e) lo = mid; else hi = mid - 1; }
	const int l = lo;
	// de Boor's recurrence up to degree 4: b[0 .. 4] = N4_{l-4 .. l}(e); it reads t[l - 3 .. l + 4]
	double b[5] = {1.0, 0.0, 0.0, 0.0, 0.0};
	for (int j = 1; j <= 4; ++j) {
		double saved = 0.0;
		for (int r = 0; r < j; ++r) {
			const double tr = t[l + r + 1], tl = t[l + r + 1 - j];
			const double f = b[r] / (tr - tl);
			b[r] = saved + f * (tr - e);
			saved = f * (e - tl);
		}
		b[j] = saved;
	}
	out.l = l;
	out.s[3] = b[4];
	out.s[2] = b[3] + out.s[3];
	out.s[1] = b[2] + out.s[2];
	out.s[0] = b[1] + out.s[1];
}

__device__ __forceinline__ double edge_cumulative(const EdgeInt& g, int i)   // J_i(e) / w_i
{
	const int q = i - (g.l - 3);
	if (q < 0) return 1.0;
	if (q > 3) return 0.0;
	return (q == 0) ? g.s[0] : ((q == 1) ? g.s[1] : ((q == 2) ? g.s[2] : g.s[3]));
}

// integral of the unit PRF spline over the pixel [xa, xb] x [ya, yb] (x: first spline axis = column direction, psf.py:146)
// (n, ny: coefficients along the first / the second axis; the table is [n][ny], RectBivariateSpline's layout)
__device__ inline double prf_pixel_general(const double* __restrict__ C, int n, int ny, const double* __restrict__ tx, const double* __restrict__ ty,
	double xa, double xb, double ya, double yb)
{
	if (!(xa < xb) || !(ya < yb)) return 0.0;
	xa = fmax(xa, tx[3]); xb = fmin(xb, tx[n]); ya = fmax(ya, ty[3]); yb = fmin(yb, ty[ny]);
	if (!(xa < xb) || !(ya < yb)) return 0.0;     // the pixel lies outside the PRF grid
	EdgeInt Xa, Xb, Ya, Yb;
	edge_integrals(tx, n, xa, Xa); edge_integrals(tx, n, xb, Xb);
	edge_integrals(ty, ny, ya, Ya); edge_integrals(ty, ny, yb, Yb);
	double acc = 0.0;
	for (int i = Xa.l - 3; i <= Xb.l; ++i) {
		const double wx = (edge_cumulative(Xb, i) - edge_cumulative(Xa, i)) * ((tx[i + 4] - tx[i]) * 0.25);
		const double* r = C + (int64_t)i * ny;
		double inner = 0.0;
		for (int j = Ya.l - 3; j <= Yb.l; ++j) {
			const double wy = (edge_cumulative(Yb, j) - edge_cumulative(Ya, j)) * ((ty[j + 4] - ty[j]) * 0.25);
			inner += wy * r[j];
		}
		acc += wx * inner;
	}
	return acc;
}

// Can the uniform-grid forms above be used?  They need 9 knot intervals per pixel and every pixel edge within the cut-off
// (|e| < cutoff + 1/2) inside the part of the knot vector where all five knots of the four straddling B-splines are evenly spaced:
// the interpolating knots are t[0..3] = x[0], t[4 + q] = x[2 + q], so t[4 .. n - 1] is even and an edge in [t[7], t[n - 4]) sees
// only it.  (For the SPOC grids, 117 samples over 13 pixels, that is cutoff <= 5.38.)
__host__ __device__ inline bool uniform_grid_ok(const double* __restrict__ t, int n, double cutoff)
{
	if (n < 32) return false;
	const double h = t[5] - t[4];
	if (!(h > 0.0) || fabs(1.0 / h - 9.0) > 1e-6) return false;
	for (int i = 4; i < n - 1; ++i) if (fabs((t[i + 1] - t[i]) - h) > 1e-9 * h) return false;
	return (-t[7] >= cutoff + 0.5) && (t[n - 4] >= cutoff + 0.5);
}

} // namespace tp_prf
