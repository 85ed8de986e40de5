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

// S fitted stars (exactly).  A workgroup owns a window of WIN consecutive cadences, one wavefront per tile of 16 (sorted)
// cadences: WIN / 16 wavefronts.  NK (star, origin) coefficient blocks are staged per pixel tile (KDBL: double-buffered).
template <int S, int WIN, int NK, bool KDBL>
__global__ __launch_bounds__(WIN * 4, (S == 1) ? 5 : 4) void tp_linpsf_fitm_kernel(FitArgs a, const StarPlan* __restrict__ plans,
	const int32_t* __restrict__ todo, const MPlan* __restrict__ mplans, const uint16_t* __restrict__ ulist, const double* __restrict__ kstore)
{
	constexpr int NTHR = WIN * 4, NWAVE = WIN / 16;
	constexpr int BSTR = WIN + 16;          // a pixel row of the staged tile; + 16: lane groups 0/1 (2/3) fall on different banks
	constexpr int NACC = S + S * (S + 1) / 2;
	constexpr int NORIG = 40;               // the plan admits at most 36 origins per star
	__shared__ __align__(16) double sK[KDBL ? 2 : 1][NK * 448];   // A operands of the staged blocks: [block][step][lane]
	__shared__ __align__(16) float bst[2][16 * BSTR];             // the pixel tile, double-buffered: [pixel][cadence of the window]
	__shared__ double sphx[S][WIN], sphy[S][WIN];
	__shared__ float spcol[S][WIN], sprow[S][WIN];
	__shared__ float ssub[WIN];                     // subtracted series; NaN for cadences past the end (-> pixel not finite)
	__shared__ float scrow[kMfmaPixels], sccol[kMfmaPixels];
	__shared__ uint16_t sperm[WIN];
	__shared__ uint16_t sU[kMfmaPixels];
	__shared__ uint8_t scc[S][WIN];                 // origin of the star at the cadence, 255: no valid position
	__shared__ uint8_t sslot[S][NORIG];             // block of (star, origin), 255: not staged (read from the store)
	__shared__ uint8_t sblk_s[NK], sblk_cc[NK];
	__shared__ unsigned smask[S][2];                // origins the star visits inside the window
	__shared__ int s_nblk;
	// set-up only: aliased with the coefficient blocks
	double (*skn)[160] = reinterpret_cast<double (*)[160]>(&sK[0][0]);                         // [2][160] knots
	unsigned* skey = reinterpret_cast<unsigned*>(&sK[0][0] + 320);                            // [WIN]
	uint16_t (*spart)[WIN] = reinterpret_cast<uint16_t (*)[WIN]>(&sK[0][0] + 320 + WIN / 2);  // [4][WIN] partial ranks
	static_assert(NK * 448 >= 320 + WIN / 2 + WIN, "set-up arrays do not fit the block buffer");

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
	const int kc = w0 + (tid & (WIN - 1));
	const bool act = kc < a.n_cad;
	const float subv = (a.subtract && act) ? a.subtract[(int64_t)target * a.subtract_pitch + kc] : 0.f;
	if (path != kPathMfma) return;
	if ((int)(s1 - s0) != S) return;   // another instantiation's targets
	const int H = a.height, W = a.width;
	const int ntiles = mp.n_tiles;
	const double cutoff = a.cutoff, c2 = cutoff * cutoff;
	const float c2f = (float)c2;

	for (int t = tid; t < kMfmaPixels; t += NTHR) {
		const unsigned px = (NTHR >= kMfmaPixels) ? upix : (unsigned)ulist[(int64_t)target * kMfmaPixels + t];
		sU[t] = (uint16_t)px;
		const int pi = (int)px / W, pj = (int)px - pi * W;
		scrow[t] = (px != 0xffffu) ? (float)pi : 1e6f;
		sccol[t] = (px != 0xffffu) ? (float)pj : 1e6f;
	}
	if (tid < n + 4) { skn[0][tid] = knx; skn[1][tid] = kny; }
	if (tid < S * 2) (&smask[0][0])[tid] = 0u;
	for (int i = tid; i < S * NORIG; i += NTHR) (&sslot[0][0])[i] = (uint8_t)255;
	__syncthreads();
	// ---- the window's cadences: phases, origins, sort key
	if (tid < WIN) {
		const double h = skn[0][5] - skn[0][4], hy = skn[1][5] - skn[1][4];
		unsigned key = 0u;
#pragma unroll
		for (int s = 0; s < S; ++s) {
			const StarPlan p = plans[(int64_t)target * kMaxStars + s];
			double phx = 0.0, phy = 0.0, prow = 0.0, pcol = 0.0;
			int cc = 255;
			if (act) {
				prow = a.pos_row[(s0 + s) * a.pos_pitch + kc];
				pcol = a.pos_col[(s0 + s) * a.pos_pitch + kc];
				int ax0, by0;
				// x <-> column (first spline axis), y <-> row  (psf.py:146)
				const bool vx = axis_phase(skn[0], n, pcol, h, phx, ax0);
				const bool vy = axis_phase(skn[1], n, prow, hy, phy, by0);
				if (vx && vy && p.nc > 0) cc = (ax0 - p.axmin) * p.nby + (by0 - p.bymin);
			}
			sphx[s][tid] = phx; sphy[s][tid] = phy;
			spcol[s][tid] = (float)pcol; sprow[s][tid] = (float)prow;
			scc[s][tid] = (uint8_t)cc;
			// the origins present: OR over the wavefront, one LDS atomic per half
			unsigned mlo = (cc < 32) ? (1u << cc) : 0u, mhi = (cc >= 32 && cc < 64) ? (1u << (cc - 32)) : 0u;
#pragma unroll
			for (int off = 32; off > 0; off >>= 1) { mlo |= __shfl_xor(mlo, off, 64); mhi |= __shfl_xor(mhi, off, 64); }
			if (lane == 0) { if (mlo) atomicOr(&smask[s][0], mlo); if (mhi) atomicOr(&smask[s][1], mhi); }
			key = key * (unsigned)kKeyBase + (unsigned)((cc == 255) ? 0 : (cc + 1));
		}
		if (!act) key = 0x3fffffu;   // cadences past the end of the series sort last
		skey[tid] = (key << 8) | (unsigned)tid;
		ssub[tid] = act ? subv : __builtin_nanf("");
	}
	__syncthreads();
	{
		// rank by counting, the keys split over the four threads of a cadence
		const int cad = tid & (WIN - 1), part = tid / WIN;
		const unsigned key = skey[cad];
		int r = 0;
		const uint4* k4 = reinterpret_cast<const uint4*>(skey) + part * (WIN / 16);
#pragma unroll 4
		for (int q = 0; q < WIN / 16; ++q) {
			const uint4 v = k4[q];
			r += ((v.x < key) ? 1 : 0) + ((v.y < key) ? 1 : 0) + ((v.z < key) ? 1 : 0) + ((v.w < key) ? 1 : 0);
		}
		spart[part][cad] = (uint16_t)r;
	}
	if (tid == NTHR - 1) {
		// the (star, origin) blocks of the window, in order; the first NK are staged in LDS per pixel tile
		int nb = 0;
		for (int s = 0; s < S; ++s) {
			unsigned long long m = ((unsigned long long)smask[s][1] << 32) | smask[s][0];
			while (m) {
				const int cc = __builtin_ctzll(m);
				m &= m - 1;
				if (nb < NK && cc < NORIG) { sslot[s][cc] = (uint8_t)nb; sblk_s[nb] = (uint8_t)s; sblk_cc[nb] = (uint8_t)cc; ++nb; }
			}
		}
		s_nblk = nb;
	}
	__syncthreads();
	if (tid < WIN) sperm[spart[0][tid] + spart[1][tid] + spart[2][tid] + spart[3][tid]] = (uint16_t)tid;
	__syncthreads();   // from here on sK belongs to the coefficient blocks

	// ---- this lane's cadence (the wavefront's tile, column lane & 15) and what does not change over the pixel tiles
	const int kl = sperm[wave * 16 + (lane & 15)];
	const bool tile_on = __any(w0 + kl < a.n_cad) != 0;
	const float sb = ssub[kl];
	double phx[S], phy[S];
	float scf[S], srf[S];
	int ccv[S];
	bool fast[S], anyv[S];
	int ccw[S];
#pragma unroll
	for (int s = 0; s < S; ++s) {
		phx[s] = sphx[s][kl]; phy[s] = sphy[s][kl];
		scf[s] = spcol[s][kl]; srf[s] = sprow[s][kl];
		ccv[s] = scc[s][kl];
		const unsigned long long vm = __ballot(ccv[s] != 255);
		anyv[s] = vm != 0ull;
		ccw[s] = vm ? __builtin_amdgcn_readlane(ccv[s], __builtin_ctzll(vm)) : 0;
		fast[s] = __ballot(ccv[s] != 255 && ccv[s] != ccw[s]) == 0ull;   // one origin in the tile: a single pass
	}

	// monomials of (step j, lane group g): j < 5: x^j y^g; j = 5: x^g y^4; j = 6: g = 0: x^4 y^4, else 0 -- the B operands of
	// the seven MFMA steps; for a tile with one origin they do not change over the pixel tiles (mask: `ok`)
	auto monomials = [&](int s, bool ok, double (&B)[7]) {
		const double x = phx[s], y = phy[s];
		const double x2 = x * x, y2 = y * y, x3 = x2 * x, y3 = y2 * y, x4 = x2 * x2, y4 = y2 * y2;
		const double pyg = (g == 0) ? 1.0 : ((g == 1) ? y : ((g == 2) ? y2 : y3));
		const double pxg = (g == 0) ? 1.0 : ((g == 1) ? x : ((g == 2) ? x2 : x3));
		const double pm = ok ? pyg : 0.0, y4m = ok ? y4 : 0.0;
		B[0] = pm; B[1] = x * pm; B[2] = x2 * pm; B[3] = x3 * pm; B[4] = x4 * pm; B[5] = pxg * y4m; B[6] = (g == 0) ? (x4 * y4m) : 0.0;
	};
	constexpr bool HOIST = false;      // measured: keeping them costs more (registers) than the twelve multiplications per pass
	constexpr bool FASTACC = (S <= 2); // the short form of the normal equations (below); with three stars its registers spill
	double Bh[HOIST ? S : 1][7];
	if (HOIST) {
#pragma unroll
		for (int s = 0; s < S; ++s) monomials(s, ccv[s] != 255, Bh[HOIST ? s : 0]);
	}

	double acc[NACC];   // g[0..S), then G[s][t], t >= s, row-major
#pragma unroll
	for (int m = 0; m < NACC; ++m) acc[m] = 0.0;

	int nts[S];
	unsigned tl[S], etl[S];
	int64_t koff[S];
#pragma unroll
	for (int s = 0; s < S; ++s) { tl[s] = mp.tiles[s]; etl[s] = mp.edge_tiles[s]; nts[s] = __popc(tl[s]); koff[s] = mp.koff[s]; }
	const int nblk = s_nblk;
	const float* img = a.images + (int64_t)target * H * W * a.t_pitch;
	// 16-byte DMA needs 16-byte aligned sources: rows on 16-byte boundaries, and a whole 4-cadence piece inside the row
	const bool vec4 = ((reinterpret_cast<uintptr_t>(a.images) & 15u) == 0) && (a.t_pitch % 4 == 0) && (a.t_pitch >= 4);

	// stage pixel tile P by LDS DMA: the 16 series of the window (coalesced along the cadences) ...
	auto stage_tile = [&](int P, float* buf) {
		if (vec4) {   // 16 bytes per lane: WIN / 4 lanes move the window of one pixel
			for (int r = wave; r < 16; r += NWAVE) {
				unsigned pix = sU[P * 16 + r];
				pix = (pix == 0xffffu) ? 0u : pix;   // a pad slot: any valid address (its coordinates put it outside every cut-off)
				int k = w0 + lane * 4;
				k = (k + 4 <= (int)a.t_pitch) ? k : ((int)a.t_pitch - 4);   // past the end of the row: any address inside it (masked by ssub)
				if (lane * 4 < WIN) dma_to_lds16(img + (int64_t)pix * a.t_pitch + k, buf + r * BSTR);
			}
		} else {
			for (int o = wave; o < 16 * (WIN / 64); o += NWAVE) {
				const int r = o / (WIN / 64), ch = o - r * (WIN / 64);
				unsigned pix = sU[P * 16 + r];
				pix = (pix == 0xffffu) ? 0u : pix;
				int k = w0 + ch * 64 + lane;
				k = (k < a.n_cad) ? k : (a.n_cad - 1);
				dma_to_lds4(img + (int64_t)pix * a.t_pitch + k, buf + r * BSTR + ch * 64);
			}
		}
	};
	// ... and the coefficient blocks of the tile: 3.5 KB each, four DMA pieces
	auto stage_blocks = [&](int P, double* kb) {
		for (int o = wave; o < nblk * 4; o += NWAVE) {
			const int b = o >> 2, ch = o & 3;
			const int s = sblk_s[b], cc = sblk_cc[b];
			unsigned tls = 0u; int ntss = 0; int64_t ko = 0;
#pragma unroll
			for (int u = 0; u < S; ++u) if (u == s) { tls = tl[u]; ntss = nts[u]; ko = koff[u]; }
			if (!((tls >> P) & 1u)) continue;
			const int rk = __popc(tls & ((1u << P) - 1u));
			const double* src = kstore + ko + ((int64_t)(cc * ntss + rk) * 7) * 64 + ch * 128 + lane * 2;
			if (ch * 64 + lane < 224) dma_to_lds16(src, kb + b * 448 + ch * 128);
		}
	};

	if (ntiles > 0) { stage_tile(0, bst[0]); stage_blocks(0, sK[0]); }
	for (int P = 0; P < ntiles; ++P) {
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__syncthreads();   // tile P has landed for every wavefront, and every wavefront is done with the other buffers
		if (P + 1 < ntiles) { stage_tile(P + 1, bst[(P + 1) & 1]); if (KDBL) stage_blocks(P + 1, sK[(P + 1) & 1]); }
		const float* buf = bst[P & 1];
		const double* kb = sK[KDBL ? (P & 1) : 0];

		if (tile_on) {
			float pr[4], pc[4], bv[4];
#pragma unroll
			for (int r = 0; r < 4; ++r) {
				pr[r] = scrow[P * 16 + g + 4 * r]; pc[r] = sccol[P * 16 + g + 4 * r];
				bv[r] = buf[(g + 4 * r) * BSTR + kl] - sb;
			}
			f64x4 D[S];
			bool has[S];
#pragma unroll
			for (int s = 0; s < S; ++s) {
				D[s] = f64x4{0.0, 0.0, 0.0, 0.0};
				has[s] = ((tl[s] >> P) & 1u) != 0u;   // wave-uniform
				if (!has[s] || !anyv[s]) continue;
				const bool valid = ccv[s] != 255;
				// one masked pass per origin in the tile (the monomials of the other cadences are zero): almost always one
				unsigned long long rem = fast[s] ? 1ull : (__ballot(valid) & 0xffffull);
				while (rem) {
					const int ccu = fast[s] ? ccw[s] : __builtin_amdgcn_readlane(ccv[s], __builtin_ctzll(rem));
					const bool mine = valid && (ccv[s] == ccu);
					rem = fast[s] ? 0ull : (rem & ~__ballot(mine));
					const int blk = __builtin_amdgcn_readfirstlane((int)sslot[s][ccu]);
					double ka[7], B[7];
					if (blk != 255) {
#pragma unroll
						for (int j = 0; j < 7; ++j) ka[j] = kb[blk * 448 + j * 64 + lane];
					} else {   // more (star, origin) blocks in the window than are staged: straight from the store
						const int rk = __popc(tl[s] & ((1u << P) - 1u));
						const double* kp = kstore + koff[s] + ((int64_t)(ccu * nts[s] + rk) * 7) * 64 + lane;
#pragma unroll
						for (int j = 0; j < 7; ++j) ka[j] = kp[j * 64];
					}
					if (HOIST && fast[s]) {
#pragma unroll
						for (int j = 0; j < 7; ++j) B[j] = Bh[HOIST ? s : 0][j];
					} else monomials(s, mine, B);
#pragma unroll
					for (int j = 0; j < 7; ++j) D[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(ka[j], B[j], D[s], 0, 0, 0);
				}
			}
			// normal equations of the tile: the pixels of this lane group, inside the cut-off, finite (linpsf_photometry.py:123).
			// Inside the cut-off: the coefficients of a pixel a star never reaches are zero, and the plan has put the pixels
			// that are inside at some cadences only ("edge") at the end of the list -- the test runs only in tiles that hold one.
			bool anyedge = false;
#pragma unroll
			for (int s = 0; s < S; ++s) anyedge = anyedge || (has[s] && ((etl[s] >> P) & 1u));
			const bool fin4 = (fabsf(bv[0]) <= 3.402823466e+38f) && (fabsf(bv[1]) <= 3.402823466e+38f) && (fabsf(bv[2]) <= 3.402823466e+38f)
				&& (fabsf(bv[3]) <= 3.402823466e+38f);
			if (FASTACC && !anyedge && !__any(!fin4)) {
#pragma unroll
				for (int r = 0; r < 4; ++r) {
					const double b = (double)bv[r];
					int m = S;
#pragma unroll
					for (int s = 0; s < S; ++s) {
						acc[s] += D[s][r] * b;
#pragma unroll
						for (int t = s; t < S; ++t) { acc[m] += D[s][r] * D[t][r]; ++m; }
					}
				}
			} else {
#pragma unroll
				for (int r = 0; r < 4; ++r) {
					const bool fin = fabsf(bv[r]) <= 3.402823466e+38f;
					const double b = fin ? (double)bv[r] : 0.0;
					double av[S];
#pragma unroll
					for (int s = 0; s < S; ++s) {
						av[s] = 0.0;
						if (!has[s]) continue;
						const float dcf = pc[r] - scf[s], drf = pr[r] - srf[s];
						const float d2f = dcf * dcf + drf * drf;
						bool inside = d2f < c2f;
						// psf.py:142  sqrt((j-col)^2 + (i-row)^2) < cutoff_radius: FP32 decides unless it is within 1e-4 of the
						// radius squared (its error is below 1e-5 for stamps up to 256 pixels wide); then the FP64 expression
						// does, and the reference's own square root when that too is within rounding
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
						acc[s] += av[s] * b;
#pragma unroll
						for (int t = s; t < S; ++t) { acc[m] += av[s] * av[t]; ++m; }
					}
				}
			}
		}
		if (!KDBL && P + 1 < ntiles) {
			__syncthreads();   // single block buffer: every wavefront is done with it
			stage_blocks(P + 1, sK[0]);
		}
	}

	// ---- sum over the four lane groups; the first group solves the tile's 16 cadences
	double G[S][S], gv[S];
	{
#pragma unroll
		for (int m = 0; m < NACC; ++m) {
			acc[m] += __shfl_xor(acc[m], 16, 64);
			acc[m] += __shfl_xor(acc[m], 32, 64);
		}
		int m = S;
#pragma unroll
		for (int s = 0; s < S; ++s) {
			gv[s] = acc[s];
#pragma unroll
			for (int t = s; t < S; ++t) { G[s][t] = acc[m]; G[t][s] = acc[m]; ++m; }
		}
	}
	const int k = w0 + kl;
	if (g != 0 || k >= a.n_cad) return;
	double x[S];
	pinv_solve<S>(G, gv, S, x);
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
#define TP_FITM(SS, WW, NKK, KD) do { \
		TP_LAUNCH(ctx, TPK_LINPSF_FIT_MFMA, (tp_linpsf_fitm_kernel<SS, WW, NKK, KD>), dim3((unsigned)n_targets, (unsigned)((a.n_cad + WW - 1) / WW)), dim3(WW * 4), 0, \
			a, d_plans, d_todo, d_mplans, d_ulist, d_kstore); \
		TP_LAUNCH_CHECK(ctx, "tp_linpsf_fitm_kernel"); \
	} while (0)
	TP_FITM(1, 128, 4, true);
	if (max_stars > 1) TP_FITM(2, 128, 6, true);
	if (max_stars > 2) TP_FITM(3, 128, 12, false);
	if (max_stars > 3) TP_FITM(4, 128, 12, false);
#undef TP_FITM
	return TP_OK;
}

} // namespace tp_linpsf
