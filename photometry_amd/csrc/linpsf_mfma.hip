// linpsf_mfma.hip -- P2..P4 on the matrix cores: the LinPSF fit of targets with up to 4 fitted stars.
//
// Replaces PSF.integrate_to_image (photometry/psf.py:122-148), lsfit (photometry/linpsf_photometry.py:22-34) and the cadence
// loop of LinPSFPhotometry.do_photometry (:114-172) for the targets the plan kernel (linpsf.hip) marks kPathMfma.
//
// For fixed knot intervals of a star's sub-pixel phases (a table "origin") the pixel-integrated PRF of a pixel is a biquartic
// in the two phases (linpsf.hip).  Written in the monomial basis that is a matrix product
//     A[pixel][cadence] = sum_q K[pixel][q] * M[q][cadence],    M[q][cadence] = phi_x^e(q) * phi_y^d(q),
// 25 monomials padded to 28 = 7 steps of v_mfma_f64_16x16x4_f64 per tile of 16 pixels x 16 cadences.  The vector-ALU kernel
// (tp_linpsf_fit2_kernel) spends ~1 000 cycles per (wavefront, star, pixel) item on delivering 25 coefficients to 24 FMAs; here
// a coefficient is ONE register of the A operand for 16 cadences, and the matrix instruction does the broadcast.
//
// Layout.  Result tile D: column = cadence (lane & 15), row = pixel ((lane >> 4) + 4 r in register r): a lane owns one
// cadence and a quarter of the pixels, so the normal equations G = A^T A, g = A^T b of a cadence are sums INSIDE a lane over
// registers, tiles and the loop -- plus one cross-lane sum over the four lane groups at the very end.  Pixels: the list U of
// the target (every pixel inside the cut-off of some star at some cadence, ordered so that the pixels of one star are
// contiguous; plan kernel), cut into tiles of 16, the same for all stars, so products A_s A_t meet in the same register.
// Cadences: a workgroup owns a window of 256 consecutive cadences, loads the window's pixel series coalesced (lane = cadence)
// and stages them in LDS one pixel tile at a time; the window is sorted by the origins of all stars (rank by counting in LDS)
// and a wavefront takes four tiles of 16 sorted cadences -- a tile that still mixes origins runs one masked pass per origin
// (M = 0 for the other cadences), which is exact.  Measured per star on the bench scene: 5.8 pixel tiles, 1.22 passes per
// tile after the window sort (2.4 in natural order, 1.0 for a sort over the whole series -- which would turn the pixel loads
// into gathers).
//
// FP64 matrix and FP64 vector instructions share one pipe on this chip (tools/lab/mfma_f64.hip: v_fma_f64 beside the MFMAs adds
// its full issue time), so everything that can is done in FP32 or integer: the cut-off test runs in FP32 with an exact FP64
// re-test for lanes within 1e-4 of the radius.
#include "linpsf_common.h"

namespace {

using namespace tp_prf;
using namespace tp_linpsf;

typedef double f64x4 __attribute__((ext_vector_type(4)));

constexpr int kKeyBase = 38;   // origins per star + 2 (the plan admits at most 36)

// global -> LDS without a trip through the registers: every lane names its own source, the destination is
// lds_base + lane * BYTES (global_load_lds_dword / _dwordx4); completion is counted in vmcnt
__device__ __forceinline__ void dma_to_lds4(const void* src, void* lds_base)
{
	__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)lds_base, 4, 0, 0);
}
__device__ __forceinline__ void dma_to_lds16(const void* src, void* lds_base)
{
	__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)lds_base, 16, 0, 0);
}

// S fitted stars (exactly).  A workgroup of 8 wavefronts owns a window of 256 consecutive cadences; a wavefront two tiles of
// 16 (sorted) cadences.
template <int S>
__global__ __launch_bounds__(512, (S == 1) ? 4 : 2) void tp_linpsf_fitm_kernel(FitArgs a, const StarPlan* __restrict__ plans,
	const int32_t* __restrict__ todo, const MPlan* __restrict__ mplans, const uint16_t* __restrict__ ulist, const double* __restrict__ kstore)
{
	constexpr int NT = 2;
	constexpr int WIN = 256;
	constexpr int BSTR = WIN + 16;          // a pixel row of the staged tile; + 16: lane groups 0/1 (2/3) fall on different banks
	constexpr int NACC = S + S * (S + 1) / 2;
	__shared__ __align__(16) float bst[2][16 * BSTR];  // the pixel tile, double-buffered: [pixel][cadence of the window]
	__shared__ double sphx[S][WIN], sphy[S][WIN];
	__shared__ double skn[2][160];
	__shared__ float spcol[S][WIN], sprow[S][WIN];
	__shared__ float ssub[WIN];                     // subtracted series; NaN for cadences past the end (-> pixel not finite)
	__shared__ float scrow[kMfmaPixels], sccol[kMfmaPixels];
	__shared__ __align__(16) unsigned skey[WIN];
	__shared__ uint16_t sperm[WIN];
	__shared__ uint16_t sU[kMfmaPixels];
	__shared__ uint8_t scc[S][WIN];                 // origin of the star at the cadence, 255: no valid position

	// ---- everything the workgroup needs from memory that does not depend on another load, in one round trip
	const int target = blockIdx.x;
	const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int g = lane >> 4;
	const int w0 = blockIdx.y * WIN;
	const int n = a.n;
	const int path = todo[target];
	const int64_t s0 = a.star_offsets[target], s1 = a.star_offsets[target + 1];
	const MPlan mp = mplans[target];
	const unsigned upix = (tid < kMfmaPixels) ? ulist[(int64_t)target * kMfmaPixels + tid] : 0xffffu;
	const double knx = (tid < n + 4) ? a.knots_x[tid] : 0.0, kny = (tid < n + 4) ? a.knots_y[tid] : 0.0;
	StarPlan pl[S];
#pragma unroll
	for (int s = 0; s < S; ++s) pl[s] = plans[(int64_t)target * kMaxStars + s];
	const int kc = w0 + (tid & (WIN - 1));
	const bool act = kc < a.n_cad;
	const float subv = (a.subtract && act) ? a.subtract[(int64_t)target * a.subtract_pitch + kc] : 0.f;
	if (path != kPathMfma) return;
	if ((int)(s1 - s0) != S) return;   // another instantiation's targets
	double prow[S], pcol[S];
#pragma unroll
	for (int s = 0; s < S; ++s) {
		prow[s] = act ? a.pos_row[(s0 + s) * a.pos_pitch + kc] : 0.0;
		pcol[s] = act ? a.pos_col[(s0 + s) * a.pos_pitch + kc] : 0.0;
	}
	const int H = a.height, W = a.width;
	const int ntiles = mp.n_tiles;
	const double cutoff = a.cutoff, c2 = cutoff * cutoff;
	const float c2f = (float)c2;

	if (tid < kMfmaPixels) {
		sU[tid] = (uint16_t)upix;
		const int pi = (int)upix / W, pj = (int)upix - pi * W;
		scrow[tid] = (upix != 0xffffu) ? (float)pi : 1e6f;
		sccol[tid] = (upix != 0xffffu) ? (float)pj : 1e6f;
		if (tid < n + 4) { skn[0][tid] = knx; skn[1][tid] = kny; }
	}
	__syncthreads();
	// ---- the window's cadences: phases, origins, sort key
	if (tid < WIN) {
		const double h = skn[0][5] - skn[0][4], hy = skn[1][5] - skn[1][4];
		unsigned key = 0u;
#pragma unroll
		for (int s = 0; s < S; ++s) {
			double phx = 0.0, phy = 0.0;
			int cc = 255;
			if (act) {
				int ax0, by0;
				// x <-> column (first spline axis), y <-> row  (psf.py:146)
				const bool vx = axis_phase(skn[0], n, pcol[s], h, phx, ax0);
				const bool vy = axis_phase(skn[1], n, prow[s], hy, phy, by0);
				if (vx && vy && pl[s].nc > 0) cc = (ax0 - pl[s].axmin) * pl[s].nby + (by0 - pl[s].bymin);
			}
			sphx[s][tid] = phx; sphy[s][tid] = phy;
			spcol[s][tid] = (float)pcol[s]; sprow[s][tid] = (float)prow[s];
			scc[s][tid] = (uint8_t)cc;
			key = key * (unsigned)kKeyBase + (unsigned)((cc == 255) ? 0 : (cc + 1));
		}
		if (!act) key = 0x3fffffu;   // cadences past the end of the series sort last
		skey[tid] = (key << 8) | (unsigned)tid;
		ssub[tid] = act ? subv : __builtin_nanf("");
	}
	__syncthreads();
	if (tid < WIN) {
		const unsigned key = skey[tid];
		int r = 0;
		const uint4* k4 = reinterpret_cast<const uint4*>(skey);
#pragma unroll 8
		for (int q = 0; q < WIN / 4; ++q) {
			const uint4 v = k4[q];
			r += ((v.x < key) ? 1 : 0) + ((v.y < key) ? 1 : 0) + ((v.z < key) ? 1 : 0) + ((v.w < key) ? 1 : 0);
		}
		sperm[r] = (uint16_t)tid;
	}
	__syncthreads();

	// ---- this lane's cadences (tile c of the wavefront, column lane & 15) and what does not change over the pixel tiles
	int kloc[NT];
	bool tile_on[NT];
	float sbv[NT];
	double phx[NT][S], phy[NT][S];
	float scf[NT][S], srf[NT][S];
	int ccv[NT][S];
#pragma unroll
	for (int c = 0; c < NT; ++c) {
		kloc[c] = sperm[(wave * NT + c) * 16 + (lane & 15)];
		tile_on[c] = __any(w0 + kloc[c] < a.n_cad) != 0;
		sbv[c] = ssub[kloc[c]];
#pragma unroll
		for (int s = 0; s < S; ++s) {
			phx[c][s] = sphx[s][kloc[c]]; phy[c][s] = sphy[s][kloc[c]];
			scf[c][s] = spcol[s][kloc[c]]; srf[c][s] = sprow[s][kloc[c]];
			ccv[c][s] = scc[s][kloc[c]];
		}
	}
	// the origin the wavefront keeps in registers per star: that of its first valid cadence (the window is sorted: almost always
	// the origin of all its cadences); a tile is "fast" for a star when every valid cadence has it
	int ccw[S];
	bool fast[NT][S], anyv[NT][S];
#pragma unroll
	for (int s = 0; s < S; ++s) {
		ccw[s] = -1;
#pragma unroll
		for (int c = NT - 1; c >= 0; --c) {
			const unsigned long long vm = __ballot(ccv[c][s] != 255);
			anyv[c][s] = vm != 0ull;
			if (vm) ccw[s] = __builtin_amdgcn_readlane(ccv[c][s], __builtin_ctzll(vm));
		}
#pragma unroll
		for (int c = 0; c < NT; ++c) fast[c][s] = __ballot(ccv[c][s] != 255 && ccv[c][s] != ccw[s]) == 0ull;
	}

	double acc[NT][NACC];   // per tile: g[0..S), then G[s][t], t >= s, row-major
#pragma unroll
	for (int c = 0; c < NT; ++c)
#pragma unroll
		for (int m = 0; m < NACC; ++m) acc[c][m] = 0.0;

	int nts[S];
	unsigned tl[S];
	int64_t koff[S];
#pragma unroll
	for (int s = 0; s < S; ++s) { tl[s] = mp.tiles[s]; nts[s] = __popc(tl[s]); koff[s] = mp.koff[s]; }
	const float* img = a.images + (int64_t)target * H * W * a.t_pitch;
	// 16-byte DMA needs 16-byte aligned sources: rows on 16-byte boundaries, and a whole 4-cadence piece inside the row
	const bool vec4 = ((reinterpret_cast<uintptr_t>(a.images) & 15u) == 0) && (a.t_pitch % 4 == 0) && (a.t_pitch >= 4);

	// stage pixel tile P (16 series of the window, coalesced along the cadences) by LDS DMA into buffer `buf`
	auto stage_tile = [&](int P, float* buf) {
		if (vec4) {   // one 16-byte DMA per lane: a wavefront moves the 256 cadences of a pixel
			for (int r = wave; r < 16; r += 8) {
				unsigned pix = sU[P * 16 + r];
				pix = (pix == 0xffffu) ? 0u : pix;   // a pad slot: any valid address (its coordinates put it outside every cut-off)
				int k = w0 + lane * 4;
				k = (k + 4 <= (int)a.t_pitch) ? k : ((int)a.t_pitch - 4);   // past the end of the row: any address inside it (masked by ssub)
				dma_to_lds16(img + (int64_t)pix * a.t_pitch + k, buf + r * BSTR);
			}
		} else {
			for (int o = wave; o < 64; o += 8) {
				const int r = o >> 2, ch = o & 3;
				unsigned pix = sU[P * 16 + r];
				pix = (pix == 0xffffu) ? 0u : pix;
				int k = w0 + ch * 64 + lane;
				k = (k < a.n_cad) ? k : (a.n_cad - 1);
				dma_to_lds4(img + (int64_t)pix * a.t_pitch + k, buf + r * BSTR + ch * 64);
			}
		}
	};

	if (ntiles > 0) stage_tile(0, bst[0]);
	for (int P = 0; P < ntiles; ++P) {
		// A operands of (star, the wavefront's origin, pixel tile P): registers, one round trip to L2 together with the DMA wait
		double Kr[S][7];
		bool has[S];
#pragma unroll
		for (int s = 0; s < S; ++s) {
			has[s] = ((tl[s] >> P) & 1u) != 0u;   // wave-uniform
#pragma unroll
			for (int j = 0; j < 7; ++j) Kr[s][j] = 0.0;
			if (has[s] && ccw[s] >= 0) {
				const int rk = __popc(tl[s] & ((1u << P) - 1u));
				const double* kp = kstore + koff[s] + ((int64_t)(ccw[s] * nts[s] + rk) * 7) * 64 + lane;
#pragma unroll
				for (int j = 0; j < 7; ++j) Kr[s][j] = kp[j * 64];
			}
		}
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__syncthreads();   // tile P has landed for every wavefront, and every wavefront is done with the other buffer
		if (P + 1 < ntiles) stage_tile(P + 1, bst[(P + 1) & 1]);
		const float* buf = bst[P & 1];

		float pr[4], pc[4];
#pragma unroll
		for (int r = 0; r < 4; ++r) { pr[r] = scrow[P * 16 + g + 4 * r]; pc[r] = sccol[P * 16 + g + 4 * r]; }
#pragma unroll
		for (int c = 0; c < NT; ++c) {
			if (!tile_on[c]) continue;
			const int kl = kloc[c];
			float bv[4];
#pragma unroll
			for (int r = 0; r < 4; ++r) bv[r] = buf[(g + 4 * r) * BSTR + kl] - sbv[c];
			f64x4 D[S];
#pragma unroll
			for (int s = 0; s < S; ++s) {
				D[s] = f64x4{0.0, 0.0, 0.0, 0.0};
				if (!has[s] || !anyv[c][s]) continue;
				const bool valid = ccv[c][s] != 255;
				const double x = phx[c][s], y = phy[c][s];
				const double x2 = x * x, y2 = y * y, x3 = x2 * x, y3 = y2 * y, x4 = x2 * x2, y4 = y2 * y2;
				const double pyg = (g == 0) ? 1.0 : ((g == 1) ? y : ((g == 2) ? y2 : y3));
				const double pxg = (g == 0) ? 1.0 : ((g == 1) ? x : ((g == 2) ? x2 : x3));
				// monomials of (step j, group g): j < 5: x^j y^g; j = 5: x^g y^4; j = 6: g = 0: x^4 y^4, else 0
				if (fast[c][s]) {
					const double pm = valid ? pyg : 0.0, y4m = valid ? y4 : 0.0;
					D[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(Kr[s][0], pm, D[s], 0, 0, 0);
					D[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(Kr[s][1], x * pm, D[s], 0, 0, 0);
					D[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(Kr[s][2], x2 * pm, D[s], 0, 0, 0);
					D[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(Kr[s][3], x3 * pm, D[s], 0, 0, 0);
					D[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(Kr[s][4], x4 * pm, D[s], 0, 0, 0);
					D[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(Kr[s][5], pxg * y4m, D[s], 0, 0, 0);
					D[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(Kr[s][6], (g == 0) ? (x4 * y4m) : 0.0, D[s], 0, 0, 0);
				} else {
					// the tile mixes origins: one masked pass per origin (the monomials of the other cadences are zero)
					unsigned long long rem = __ballot(valid) & 0xffffull;
					const int rk = __popc(tl[s] & ((1u << P) - 1u));
					while (rem) {
						const int ccu = __builtin_amdgcn_readlane(ccv[c][s], __builtin_ctzll(rem));
						const bool mine = valid && (ccv[c][s] == ccu);
						rem &= ~__ballot(mine);
						double ka[7];
						if (ccu == ccw[s]) {
#pragma unroll
							for (int j = 0; j < 7; ++j) ka[j] = Kr[s][j];
						} else {
							const double* kp = kstore + koff[s] + ((int64_t)(ccu * nts[s] + rk) * 7) * 64 + lane;
#pragma unroll
							for (int j = 0; j < 7; ++j) ka[j] = kp[j * 64];
						}
						const double pm = mine ? pyg : 0.0, y4m = mine ? y4 : 0.0;
						D[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(ka[0], pm, D[s], 0, 0, 0);
						D[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(ka[1], x * pm, D[s], 0, 0, 0);
						D[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(ka[2], x2 * pm, D[s], 0, 0, 0);
						D[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(ka[3], x3 * pm, D[s], 0, 0, 0);
						D[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(ka[4], x4 * pm, D[s], 0, 0, 0);
						D[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(ka[5], pxg * y4m, D[s], 0, 0, 0);
						D[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(ka[6], (g == 0) ? (x4 * y4m) : 0.0, D[s], 0, 0, 0);
					}
				}
			}
			// normal equations of the tile: the pixels of this lane group, inside the cut-off, finite (linpsf_photometry.py:123)
#pragma unroll
			for (int r = 0; r < 4; ++r) {
				const bool fin = fabsf(bv[r]) <= 3.402823466e+38f;
				const double b = fin ? (double)bv[r] : 0.0;
				double av[S];
#pragma unroll
				for (int s = 0; s < S; ++s) {
					av[s] = 0.0;
					if (!has[s]) continue;
					const float dcf = pc[r] - scf[c][s], drf = pr[r] - srf[c][s];
					const float d2f = dcf * dcf + drf * drf;
					bool inside = d2f < c2f;
					// psf.py:142  sqrt((j-col)^2 + (i-row)^2) < cutoff_radius: FP32 decides unless it is within 1e-4 of the radius
					// squared (its error is below 1e-5 for stamps up to 256 pixels wide); then the FP64 expression does, and
					// the reference's own square root when that too is within rounding
					const bool near = fin && (fabsf(d2f - c2f) <= 1e-4f * c2f);
					if (__any(near)) {
						if (near) {
							const int k = w0 + kl;
							const double dc = (double)pc[r] - a.pos_col[(s0 + s) * a.pos_pitch + k];
							const double dr = (double)pr[r] - a.pos_row[(s0 + s) * a.pos_pitch + k];
							const double dr2 = dr * dr;
							const double d2 = dc * dc + dr2;
							inside = (fabs(d2 - c2) > 1e-9 * c2) ? (d2 < c2) : (sqrt(d2) < cutoff);
						}
					}
					av[s] = (fin && inside) ? D[s][r] : 0.0;
				}
				int m = S;
#pragma unroll
				for (int s = 0; s < S; ++s) {
					acc[c][s] += av[s] * b;
#pragma unroll
					for (int t = s; t < S; ++t) { acc[c][m] += av[s] * av[t]; ++m; }
				}
			}
		}
	}

	// ---- sum over the four lane groups; lane group g then solves the cadences of tile g
	double G[S][S], gv[S];
	{
		double tot[NACC];
#pragma unroll
		for (int m = 0; m < NACC; ++m) tot[m] = 0.0;
#pragma unroll
		for (int c = 0; c < NT; ++c) {
#pragma unroll
			for (int m = 0; m < NACC; ++m) {
				double v = acc[c][m];
				v += __shfl_xor(v, 16, 64);
				v += __shfl_xor(v, 32, 64);
				if (g == c) tot[m] = v;
			}
		}
		int m = S;
#pragma unroll
		for (int s = 0; s < S; ++s) {
			gv[s] = tot[s];
#pragma unroll
			for (int t = s; t < S; ++t) { G[s][t] = tot[m]; G[t][s] = tot[m]; ++m; }
		}
	}
	int kl = 0;
#pragma unroll
	for (int c = 0; c < NT; ++c) if (g == c) kl = kloc[c];
	double x[S];
	pinv_solve<S>(G, gv, S, x);
	const int k = w0 + kl;
	if (g >= NT || k >= a.n_cad) return;
	const int ti = a.target_index[target];
	double tf = __builtin_nan("");
#pragma unroll
	for (int s = 0; s < S; ++s) {
		a.fluxes_all[(s0 + s) * a.out_pitch + k] = x[s];
		if (s == ti) tf = x[s];
	}
	a.flux[(int64_t)target * a.out_pitch + k] = tf;
	a.flux_err[(int64_t)target * a.out_pitch + k] = __builtin_nan("");
}

} // namespace

namespace tp_linpsf {

// launches the matrix-core fit for the star counts present (max_stars = the largest count of the batch); every workgroup
// whose target is not marked kPathMfma, or belongs to another star count, exits at once
int fit_mfma_launch(tp_ctx* ctx, const FitArgs& a, int n_targets, int max_stars, const StarPlan* d_plans, const int32_t* d_todo,
	const MPlan* d_mplans, const uint16_t* d_ulist, const double* d_kstore)
{
#define TP_FITM(SS) do { \
		TP_LAUNCH(ctx, TPK_LINPSF_FIT_MFMA, (tp_linpsf_fitm_kernel<SS>), dim3((unsigned)n_targets, (unsigned)((a.n_cad + 255) / 256)), dim3(512), 0, \
			a, d_plans, d_todo, d_mplans, d_ulist, d_kstore); \
		TP_LAUNCH_CHECK(ctx, "tp_linpsf_fitm_kernel"); \
	} while (0)
	TP_FITM(1);
	if (max_stars > 1) TP_FITM(2);
	if (max_stars > 2) TP_FITM(3);
	if (max_stars > 3) TP_FITM(4);
#undef TP_FITM
	return TP_OK;
}

} // namespace tp_linpsf
