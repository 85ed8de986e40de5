// linpsf_mfma.hip -- P2..P4 on the matrix cores: the LinPSF fit of targets with up to 4 fitted stars.
//
// Replaces PSF.integrate_to_image (photometry/psf.py:122-148), lsfit (photometry/linpsf_photometry.py:22-34) and the cadence
// loop of LinPSFPhotometry.do_photometry (:114-172) for the targets the plan kernel (linpsf.hip) marks kPathMfma.
//
// The pixel-integrated PRF of a pixel, as a function of the star's position, is a tensor-product QUARTIC SPLINE in the two
// sub-pixel phases with its knots on the PRF sample grid (the integral of the bicubic spline psf.py:119 fits; a pixel is exactly
// nine knot intervals wide, so both edges of a pixel cross their knots together).  Inside one pair of knot intervals -- a table
// "origin" -- it is the biquartic the vector-ALU path evaluates from 25 coefficients per (pixel, origin).  Over the na x nb
// intervals a star visits during the series (the jitter straddles a knot in most targets: 1 - 3 per axis) the same function is
//     F(X, Y) = sum_ED C[E][D] m_E(X) m_D(Y),   m = {1, t, t^2, t^3, t^4, (t-1)+^4, (t-2)+^4},   X = interval + phase,
// because a quartic spline changes only its leading coefficient at a simple knot -- ONE coefficient set per (star, pixel) for all
// cadences (coefficient kernel, linpsf.hip), and so a plain matrix product
//     A[pixel][cadence] = sum_q C[pixel][q] * M[q][cadence],    M[q][cadence] = m_E(q)(X) * m_D(q)(Y),
// 25 - 49 basis products = 7 - 13 steps of v_mfma_f64_16x16x4_f64 per tile of 16 pixels x 16 cadences, the cadences in their
// natural order (round 3's first version sorted every window of cadences by origin and ran one masked pass per origin with a
// coefficient block per (star, origin, tile): it spent its time staging those blocks).
//
// Layout.  One workgroup per SEGMENT of a target's series (linpsf_common.h: a stretch of 16-cadence tiles inside which every star
// visits at most 3 x 3 knot intervals; one segment per target unless a star drifts); the segment's coefficient image (A operands,
// [star][tile][step][lane], 20 - 150 KB) is copied to LDS once.  A wavefront takes one tile of 16 consecutive cadences at a time
// (with the Cholesky solve a sixteen-lane solve costs less than the idle tail of larger units).  Result tile D: column = cadence
// (lane & 15), row = pixel ((lane >> 4) + 4 r in register r): a lane owns one cadence and a quarter of the pixels, so the normal
// equations G = A^T A, g = A^T b of a cadence are sums INSIDE a lane over registers and pixel tiles, plus one cross-lane sum
// over the four lane groups per tile of cadences.
// Pixels: the list U of the target (every pixel inside the cut-off of some star at some cadence, ordered so that the pixels of one
// star are contiguous; plan kernel), cut into tiles of 16, the same for all stars, so products A_s A_t meet in the same register.
// The B operands (basis products of the lane's cadence) are computed once per star and tile of cadences and stay in registers.
//
// FP64 matrix and FP64 vector instructions share one pipe on this chip (tools/lab/mfma_f64.hip: v_fma_f64 beside the MFMAs adds
// its full issue time -- and so does every other vector instruction, 2.3 - 5 cycles each beside the 70 of an MFMA), so the
// vector work per tile is kept small: the cut-off test runs in FP32 with an exact FP64 re-test for lanes within 1e-4 of the
// radius, and only for the stars and registers that hold a pixel which is inside at some cadences only; a NaN pixel only zeroes
// its row; the terms of the normal equations are formed only for the stars that reach the tile.
#include "linpsf_common.h"

namespace {

using namespace tp_prf;
using namespace tp_linpsf;

typedef double f64x4 __attribute__((ext_vector_type(4)));

// global -> LDS without a trip through the registers: every lane names its own source, the destination is
// lds_base + lane * 16 (global_load_lds_dwordx4); completion is counted in vmcnt
__device__ __forceinline__ void dma_to_lds16(const void* src, void* lds_base)
{
	__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)lds_base, 16, 0, 0);
}

// S fitted stars (exactly); one workgroup of up to NTHR / 64 wavefronts per segment of the class list, at least MINW wavefronts
// resident per SIMD; a wavefront takes a tile of 16 cadences at a time (its unit of work: the smaller, the more evenly the
// segment divides over the wavefronts; 16 lanes then solve).
template <int S, int NTHR, int MINW>
__global__ __launch_bounds__(NTHR, MINW) void tp_linpsf_fitm_kernel(FitArgs a, const SegPlan* __restrict__ segs,
	const int32_t* __restrict__ seg_list, const MPlan* __restrict__ mplans, const uint16_t* __restrict__ ulist, const uint8_t* __restrict__ usig,
	const double* __restrict__ kstore, double* __restrict__ alast)
{
	const int NWV = (int)blockDim.x >> 6;   // wavefronts of the workgroup (chosen by the host for the length of the series)
	constexpr int NACC = S + S * (S + 1) / 2;
	extern __shared__ __align__(16) double sK[];    // the target's coefficient image
	// per pixel of the list, in the order a lane reads them -- [tile][lane group g][register r] = pixel 16 tile + g + 4 r: the
	// byte offset of its series in the target's cube (one 16-byte read gives a lane its four), row and column
	__shared__ __align__(16) unsigned soff[kMfmaPixels];
	__shared__ __align__(16) float scrow[kMfmaPixels], sccol[kMfmaPixels];
	__shared__ unsigned semask[S][16];   // per star and tile: the pixels (bit u of the tile) that are inside the cut-off at some cadences only

	const SegPlan sg = segs[seg_list[blockIdx.x]];
	const int target = sg.target;
	const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int g = lane >> 4;
	const int64_t s0 = a.star_offsets[target];
	const MPlan mp = mplans[target];
	const int H = a.height, W = a.width;
	const int ntiles = mp.n_tiles;
	const double cutoff = a.cutoff, c2 = cutoff * cutoff;
	const float c2f = (float)c2;

	// ---- the coefficient image by LDS DMA (1 KB per wavefront and instruction), the pixel list, the edge masks
	{
		const double* ksrc = kstore + sg.koff;
		for (int off = wave * 128; off < sg.kdoubles; off += NWV * 128)
			if (off + lane * 2 < sg.kdoubles) dma_to_lds16(ksrc + off + lane * 2, sK + off);
	}
	if (tid < S * 16) (&semask[0][0])[tid] = 0u;
	__syncthreads();
	for (int t = tid; t < kMfmaPixels; t += (int)blockDim.x) {
		const unsigned edge = (unsigned)usig[(int64_t)target * kMfmaPixels + t] >> 4;
#pragma unroll
		for (int s = 0; s < S; ++s) if ((edge >> s) & 1u) atomicOr(&semask[s][t >> 4], 1u << (t & 15));
		const unsigned px = ulist[(int64_t)target * kMfmaPixels + t];
		const int slot = (t & ~15) | ((t & 3) << 2) | ((t >> 2) & 3);   // pixel t = 16 tile + g + 4 r  ->  [tile][g][r]
		soff[slot] = (px == 0xffffu) ? 0u : (px * (unsigned)a.t_pitch) << 2;    // bytes; a pad slot: any valid address (its coefficients are zero)
		const int pi = (int)px / W, pj = (int)px - pi * W;
		scrow[slot] = (px != 0xffffu) ? (float)pi : 1e6f;
		sccol[slot] = (px != 0xffffu) ? (float)pj : 1e6f;
	}
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__syncthreads();

	// the knots are uniform where a star's pixel edges can fall (tp_linpsf_fit checks the cut-off radius for that): the phase
	// measured from the first interior knot is one multiplication away
	const double kx4 = a.knots_x[4], ky4 = a.knots_y[4];
	const double rh = 1.0 / (a.knots_x[5] - kx4), rhy = 1.0 / (a.knots_y[5] - ky4);
	unsigned tl[S], etl[S];
	int na[S], nb[S], nk[S], kbase[S], axmin[S], bymin[S];
#pragma unroll
	for (int s = 0; s < S; ++s) {
		tl[s] = mp.tiles[s]; etl[s] = mp.edge_tiles[s];
		na[s] = sg.na[s]; nb[s] = sg.nb[s];
		nk[s] = mfma_steps(na[s], nb[s]);
		kbase[s] = (int)sg.ksub[s] * 64;
		axmin[s] = sg.axmin[s]; bymin[s] = sg.bymin[s];
	}
	const float* img = a.images + (int64_t)target * H * W * a.t_pitch;
	const int ti = a.target_index[target];   // loaded once: inside the loop its latency would stand in front of every solve
	constexpr int GCAD = 16;        // cadences per unit

	// positions and subtracted value of the NEXT tile of cadences are loaded a tile ahead (their latency would otherwise stand
	// in front of every tile: nothing else can start before the basis products)
	double nprow[S], npcol[S];
	float nsb = 0.f;
	auto load_cadence = [&](int k0) {
		int kq = k0 + (lane & 15);
		kq = (kq < a.n_cad) ? kq : (a.n_cad - 1);
#pragma unroll
		for (int s = 0; s < S; ++s) { nprow[s] = a.pos_row[(s0 + s) * a.pos_pitch + kq]; npcol[s] = a.pos_col[(s0 + s) * a.pos_pitch + kq]; }
		if (a.subtract) nsb = a.subtract[(int64_t)target * a.subtract_pitch + kq];
	};
	constexpr bool AHEAD = (S == 1);   // with more stars the registers are worth more than the latency (measured)
	if (AHEAD) load_cadence((sg.tile0 + wave) * GCAD);
	for (int gi = sg.tile0 + wave; gi < sg.tile1; gi += NWV) {
		double kept[NACC];   // the normal equations of the cadence this lane solves: g[0..S), then G[s][t], t >= s, row-major
#pragma unroll
		for (int m = 0; m < NACC; ++m) kept[m] = 0.0;
		{
			const int k0 = gi * GCAD;
			const int k = k0 + (lane & 15);
			const bool act = k < a.n_cad;
			const bool last_here = (a.n_cad - 1 >= k0) && (a.n_cad - 1 < k0 + 16);   // uniform
			const int kk = act ? k : (a.n_cad - 1);
			const unsigned kk4 = (unsigned)kk << 2;
			// the first pixel tiles of this tile of cadences are on their way while the basis products are computed
			auto load_tile = [&](int P, float (&bv)[4]) {
				// byte offsets (the host admits cubes below 2^30 elements per target only): one addition per load
				const uint4 o = *reinterpret_cast<const uint4*>(&soff[P * 16 + g * 4]);
				const char* ib = reinterpret_cast<const char*>(img);
				bv[0] = *reinterpret_cast<const float*>(ib + (o.x + kk4)); bv[1] = *reinterpret_cast<const float*>(ib + (o.y + kk4));
				bv[2] = *reinterpret_cast<const float*>(ib + (o.z + kk4)); bv[3] = *reinterpret_cast<const float*>(ib + (o.w + kk4));
			};
			float bv0[4], bv1[4] = {0.f, 0.f, 0.f, 0.f}, bv2[4] = {0.f, 0.f, 0.f, 0.f};
			constexpr int RING = (S == 1) ? 3 : 2;   // pixel tiles in flight (registers again)
			load_tile(0, bv0);
			if (RING == 3 && ntiles > 1) load_tile(1, bv1);
			if (!AHEAD) load_cadence(k0);
			const float sb = nsb;
			// the short vector phases (basis products here, reduction and solve below) run at raised priority: a wavefront in one of
			// them would otherwise queue every instruction behind the 70-cycle matrix instructions of the wavefronts in their pixel loops
			__builtin_amdgcn_s_setprio(3);

			// ---- B operands: the basis products of this lane's cadence.  Steps (the order of the coefficient image): x basis
			// 0..4 times y basis g; y basis 4 times x basis g, then 4 + g; x basis 5, 6 times y basis g; y basis 5, 6 like 4
			// (2 x 2 intervals: 9 steps, see mfma_is22)
			double B[S][13];
			float scf[S], srf[S];
#pragma unroll
			for (int s = 0; s < S; ++s) {
				const double prow = nprow[s], pcol = npcol[s];
				// x <-> column (first spline axis), y <-> row  (psf.py:146).  Position -> knot interval + phase as in axis_phase
				// (linpsf_dev.h), in one piece: the lower edge of the nearest pixel, in knot intervals from the first interior knot,
				// less the origin of the intervals the star visits.  A NaN / absurd position fits nothing (psf.py:142).
				const bool ok = act && (fabs(pcol) < 1e6) && (fabs(prow) < 1e6) && (na[s] > 0);
				scf[s] = (float)pcol; srf[s] = (float)prow;
				const double jx = rint(pcol), jy = rint(prow);
				const double Xr = (((jx - pcol) - 0.5) - kx4) * rh + ((1.0 - (double)axmin[s]) - 9.0 * jx);
				const double Yr = (((jy - prow) - 0.5) - ky4) * rhy + ((1.0 - (double)bymin[s]) - 9.0 * jy);
				const double X = ok ? Xr : 0.0, Y = ok ? Yr : 0.0;
				const double X2 = X * X, Y2 = Y * Y, X3 = X2 * X, Y3 = Y2 * Y, X4 = X2 * X2, Y4 = Y2 * Y2;
				double t = fmax(X - 1.0, 0.0); t *= t; const double X5 = t * t;
				t = fmax(X - 2.0, 0.0); t *= t; const double X6 = t * t;
				t = fmax(Y - 1.0, 0.0); t *= t; const double Y5 = t * t;
				t = fmax(Y - 2.0, 0.0); t *= t; const double Y6 = t * t;
				double yg = (g == 0) ? 1.0 : ((g == 1) ? Y : ((g == 2) ? Y2 : Y3));
				const double xlo = (g == 0) ? 1.0 : ((g == 1) ? X : ((g == 2) ? X2 : X3));
				const double xhi = (g == 0) ? X4 : ((g == 1) ? X5 : ((g == 2) ? X6 : 0.0));
				yg = ok ? yg : 0.0;
				const double y4 = ok ? Y4 : 0.0, y5 = ok ? Y5 : 0.0, y6 = ok ? Y6 : 0.0;
				B[s][0] = yg; B[s][1] = X * yg; B[s][2] = X2 * yg; B[s][3] = X3 * yg; B[s][4] = X4 * yg;
				// 2 x 2 intervals (uniform per star): step 6 = {X^4, (X-1)+^4} y4 and {1, X} y5, step 9 = {X^2, X^3, X^4, (X-1)+^4} y5
				const bool p22 = mfma_is22(na[s], nb[s]);
				const double x6 = p22 ? ((g == 2) ? 1.0 : ((g == 3) ? X : xhi)) : xhi;
				const double x9 = p22 ? ((g == 0) ? X2 : ((g == 1) ? X3 : ((g == 2) ? X4 : X5))) : xlo;
				B[s][5] = xlo * y4; B[s][6] = x6 * ((p22 && g >= 2) ? y5 : y4);
				B[s][7] = X5 * yg; B[s][8] = X6 * yg;
				B[s][9] = x9 * y5; B[s][10] = xhi * y5;
				B[s][11] = xlo * y6; B[s][12] = xhi * y6;
			}

			if (AHEAD) load_cadence((gi + NWV) * GCAD);
			__builtin_amdgcn_s_setprio(0);

			double acc[NACC];
#pragma unroll
			for (int m = 0; m < NACC; ++m) acc[m] = 0.0;

			auto process = [&](int P, const float (&bvin)[4]) {
				float bv[4];
#pragma unroll
				for (int r = 0; r < 4; ++r) bv[r] = bvin[r] - sb;
				f64x4 D[S];
				bool has[S];
				bool anyedge = false;
#pragma unroll
				for (int s = 0; s < S; ++s) {
					D[s] = f64x4{0.0, 0.0, 0.0, 0.0};
					has[s] = ((tl[s] >> P) & 1u) != 0u;   // wave-uniform
					if (!has[s]) continue;
					anyedge = anyedge || (((etl[s] >> P) & 1u) != 0u);
					const int rk = __popc(tl[s] & ((1u << P) - 1u));
					const double* kb = sK + kbase[s] + rk * nk[s] * 64 + lane;
#pragma unroll
					for (int j = 0; j < 7; ++j) D[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(kb[j * 64], B[s][j], D[s], 0, 0, 0);
					kb += 7 * 64;
					if (na[s] >= 2) { D[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(kb[0], B[s][7], D[s], 0, 0, 0); kb += 64; }
					if (na[s] >= 3) { D[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(kb[0], B[s][8], D[s], 0, 0, 0); kb += 64; }
					if (nb[s] >= 2) {
						D[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(kb[0], B[s][9], D[s], 0, 0, 0);
						if (!mfma_is22(na[s], nb[s])) D[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(kb[64], B[s][10], D[s], 0, 0, 0);
						kb += 128;
					}
					if (nb[s] >= 3) {
						D[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(kb[0], B[s][11], D[s], 0, 0, 0);
						D[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(kb[64], B[s][12], D[s], 0, 0, 0);
					}
				}
				// normal equations of the tile: the pixels of this lane group, inside the cut-off, finite (linpsf_photometry.py:123).
				// Inside the cut-off: the coefficients of a pixel a star never reaches are zero, and the plan has put the pixels
				// that are inside at some cadences only ("edge") at the end of the list -- the test runs only in tiles that hold one.
				// Vector instructions do not run beside the matrix instructions on this chip (tools/lab/mfma_f64.hip: every kind
				// adds its issue time), so only the terms of the stars that reach the tile are formed (uniform branches).
				// all four finite <=> their sum is (a NaN or an infinity survives any sum; finite values that overflow it only send
				// the tile through the exact per-pixel test below)
				const bool fin4 = fabsf((bv[0] + bv[1]) + (bv[2] + bv[3])) <= 3.402823466e+38f;
				if (__any(!fin4)) {   // a NaN pixel in the tile (uniform): its row of the design matrix and its value count as zero
#pragma unroll
					for (int r = 0; r < 4; ++r) {
						const bool fin = fabsf(bv[r]) <= 3.402823466e+38f;
						bv[r] = fin ? bv[r] : 0.f;
#pragma unroll
						for (int s = 0; s < S; ++s) if (has[s]) D[s][r] = fin ? D[s][r] : 0.0;
					}
				}
				if (anyedge) {
					const float4 pr = *reinterpret_cast<const float4*>(&scrow[P * 16 + g * 4]), pc = *reinterpret_cast<const float4*>(&sccol[P * 16 + g * 4]);
					const float prr[4] = {pr.x, pr.y, pr.z, pr.w}, pcr[4] = {pc.x, pc.y, pc.z, pc.w};
#pragma unroll
					for (int s = 0; s < S; ++s) {
						if (!has[s] || !((etl[s] >> P) & 1u)) continue;
						// the pixels of this tile that are inside the star's cut-off at some cadences only: bit u = g + 4 r, so a nibble
						// per register r -- the edge pixels are the last of their group of the list, a few consecutive u
						const unsigned em = (unsigned)__builtin_amdgcn_readfirstlane((int)semask[s][P]);
#pragma unroll
						for (int r = 0; r < 4; ++r) {
							if (!((em >> (4 * r)) & 15u)) continue;
							const float dcf = pcr[r] - scf[s], drf = prr[r] - srf[s];
							const float d2f = dcf * dcf + drf * drf;
							bool inside = d2f < c2f;
							// psf.py:142  sqrt((j-col)^2 + (i-row)^2) < cutoff_radius: FP32 decides unless it is within 1e-4 of the
							// radius squared (its error is below 1e-5 for stamps up to 256 pixels wide); then the FP64 expression
							// does, and the reference's own square root when that too is within rounding
							const bool near = fabsf(d2f - c2f) <= 1e-4f * c2f;
							if (__any(near)) {
								if (near) {
									const double dc = (double)pcr[r] - a.pos_col[(s0 + s) * a.pos_pitch + kk];
									const double dr = (double)prr[r] - a.pos_row[(s0 + s) * a.pos_pitch + kk];
									const double dr2 = dr * dr;
									const double d2 = dc * dc + dr2;
									inside = (fabs(d2 - c2) > 1e-9 * c2) ? (d2 < c2) : (sqrt(d2) < cutoff);
								}
							}
							D[s][r] = inside ? D[s][r] : 0.0;
						}
					}
				}
				// the design matrix of the LAST cadence (pixels outside the cut-off and non-finite pixels zero) is what the
				// contamination is computed from (linpsf_photometry.py:203-211): the lane that owns that cadence writes it out
				if (last_here && (lane & 15) == a.n_cad - 1 - k0) {
#pragma unroll
					for (int s = 0; s < S; ++s)
#pragma unroll
						for (int r = 0; r < 4; ++r) alast[((int64_t)target * kMfmaStars + s) * kMfmaPixels + P * 16 + g + 4 * r] = D[s][r];
				}
#pragma unroll
				for (int r = 0; r < 4; ++r) {
					const double b = (double)bv[r];
					int m = S;
#pragma unroll
					for (int s = 0; s < S; ++s) {
						if (has[s]) acc[s] += D[s][r] * b;
#pragma unroll
						for (int t = s; t < S; ++t) { if (has[s] && has[t]) acc[m] += D[s][r] * D[t][r]; ++m; }
					}
				}
			};
			// three pixel tiles in flight, the loop unrolled by three so that no loaded register is copied (a copy waits for its load)
			if (RING == 3) {
#pragma unroll 1
				for (int P = 0; P < ntiles; P += 3) {
					if (P + 2 < ntiles) load_tile(P + 2, bv2);
					process(P, bv0);
					if (P + 1 >= ntiles) break;
					if (P + 3 < ntiles) load_tile(P + 3, bv0);
					process(P + 1, bv1);
					if (P + 2 >= ntiles) break;
					if (P + 4 < ntiles) load_tile(P + 4, bv1);
					process(P + 2, bv2);
				}
			} else {
#pragma unroll 1
				for (int P = 0; P < ntiles; P += 2) {
					if (P + 1 < ntiles) load_tile(P + 1, bv1);
					process(P, bv0);
					if (P + 1 >= ntiles) break;
					if (P + 2 < ntiles) load_tile(P + 2, bv0);
					process(P + 1, bv1);
				}
			}
			__builtin_amdgcn_s_setprio(3);
			// ---- sum over the four lane groups
#pragma unroll
			for (int m = 0; m < NACC; ++m) {
				double v = acc[m];
				v += __shfl_xor(v, 16, 64);
				v += __shfl_xor(v, 32, 64);
				kept[m] = v;
			}
		}
		// ---- every lane solves one cadence
		const int k = gi * GCAD + lane;
		if (lane < GCAD && k < a.n_cad) {
			double G[S][S], gv[S], x[S];
			int m = S;
#pragma unroll
			for (int s = 0; s < S; ++s) {
				gv[s] = kept[s];
#pragma unroll
				for (int t = s; t < S; ++t) { G[s][t] = kept[m]; G[t][s] = kept[m]; ++m; }
			}
			// Cholesky where the normal equations are well conditioned (almost always), the pseudo-inverse otherwise; the branch is
			// taken per wavefront so that the Jacobi sweeps run only where some cadence needs them
			const bool easy = (S > 1) && chol_solve<S>(G, gv, x);
			if (__any(!easy)) {
				double xp[S];
				pinv_solve<S>(G, gv, S, xp);
				if (!easy) {
#pragma unroll
					for (int s = 0; s < S; ++s) x[s] = xp[s];
				}
			}
			double tf = __builtin_nan("");
#pragma unroll
			for (int s = 0; s < S; ++s) {
				a.fluxes_all[(s0 + s) * a.out_pitch + k] = x[s];
				if (s == ti) tf = x[s];
			}
			a.flux[(int64_t)target * a.out_pitch + k] = tf;
			a.flux_err[(int64_t)target * a.out_pitch + k] = __builtin_nan("");
		}
	}
}

} // namespace

namespace tp_linpsf {

// the number of wavefronts (at most `most`) that leaves the fewest idle while the others finish their last group of 64 cadences
static int fit_waves(int n_cad, int most, int group_cadences)
{
	const int groups = (n_cad + group_cadences - 1) / group_cadences;
	int best = most;
	double best_par = 0.0;
	for (int w = most; w >= (most + 1) / 2; --w) {
		const double par = (double)groups / (double)((groups + w - 1) / w);   // wavefronts busy on average
		if (par >= best_par) { best_par = par; best = w; }
	}
	return best;
}

// launches the matrix-core fit, one launch per star count over the segments the plan kernel has listed for it
int fit_mfma_launch(tp_ctx* ctx, const FitArgs& a, int n_targets, const unsigned long long* seg_counts, const unsigned long long* class_counts, const SegPlan* d_segs,
	const int32_t* d_seg_lists, const MPlan* d_mplans, const uint16_t* d_ulist, const uint8_t* d_usig, const double* d_kstore, double* d_alast)
{
	// The launches are independent (one per star count): the first runs on the context's stream, the others on two side streams
	// that wait for what precedes on it (the coefficient store) and are waited for before what follows (the finalisation), so that
	// the tail of one launch -- its last workgroups on a mostly idle chip -- overlaps the body of another.
	// (one way out: a failure below still joins the side streams and returns the events to the pool)
	hipEvent_t before = ctx->get_event();
	hipError_t err = hipEventRecord(before, ctx->stream);
	int used = 0;
	hipStream_t streams[3] = {ctx->stream, nullptr, nullptr};
	for (int i = 0; i < 2; ++i) {
		if (!ctx->side[i] && err == hipSuccess) err = hipStreamCreateWithFlags(&ctx->side[i], hipStreamNonBlocking);
		streams[i + 1] = ctx->side[i];
	}
	bool waited[3] = {true, false, false};
	// (the workgroup is sized for the whole series also where the class has several segments per target: sizing it for the mean
	// segment -- fewer wavefronts, a fuller last round -- measured slower, 7.5 against 6.9 ms on the drift scene: the classes with
	// two and more stars hold one workgroup per CU, and its wavefronts are the CU's occupancy)
	(void)class_counts;
#define TP_FITM(CLS, SS, TT, WW, LDS) do { \
		if (seg_counts[CLS] > 0 && err == hipSuccess && streams[used % 3] != nullptr) { \
			const int si = used++ % 3; \
			if (!waited[si]) { err = hipStreamWaitEvent(streams[si], before, 0); waited[si] = (err == hipSuccess); } \
			if (err == hipSuccess) err = hipFuncSetAttribute(reinterpret_cast<const void*>(tp_linpsf_fitm_kernel<SS, TT, WW>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS); \
			if (err == hipSuccess) { TP_LAUNCH_ON(ctx, streams[si], TPK_LINPSF_FIT_MFMA, (tp_linpsf_fitm_kernel<SS, TT, WW>), dim3((unsigned)seg_counts[CLS]), dim3(64 * fit_waves(a.n_cad, TT / 64, 16)), (size_t)LDS, \
				a, d_segs, d_seg_lists + (size_t)(CLS) * n_targets * kMfmaSegs, d_mplans, d_ulist, d_usig, d_kstore, d_alast); \
			err = hipGetLastError(); } \
		} \
	} while (0)
	// registers decide the shape (measured, C3 batch): one star 124 VGPRs -- two workgroups of 8 wavefronts per CU; two stars 167 --
	// one workgroup of up to 12 (three per SIMD; at 128 registers two workgroups of 8 spill and lose: 8.46 against 8.12 ms per
	// step); three and four stars 226 / 256 -- two per SIMD, one workgroup of 8 whatever the size of the image (two workgroups
	// of 4 for the small images: 0.38 ms for what the large configuration does in 0.1).  Units of 16 cadences everywhere: with the
	// Cholesky solve the sixteen-lane solve costs less than the idle tail of larger units (three stars: 2.78 -> 2.47 ms)
	TP_FITM(0, 1, 512, 4, kMfmaLdsSmall);
	TP_FITM(1, 2, 768, 3, kMfmaLdsLarge);
	TP_FITM(2, 3, 512, 2, kMfmaLdsLarge);
	TP_FITM(3, 4, 512, 2, kMfmaLdsLarge);
#undef TP_FITM
	for (int i = 1; i < 3; ++i) {
		if (waited[i] && streams[i]) {
			hipEvent_t done = ctx->get_event();
			hipError_t e2 = hipEventRecord(done, streams[i]);
			if (e2 == hipSuccess) e2 = hipStreamWaitEvent(ctx->stream, done, 0);
			if (e2 != hipSuccess) (void)hipStreamSynchronize(streams[i]);   // the join of last resort
			if (err == hipSuccess) err = e2;
			ctx->pool.push_back(done);
		}
	}
	ctx->pool.push_back(before);
	if (err != hipSuccess) return ctx->fail(TP_ERR_HIP, "tp_linpsf_fitm_kernel", err);
	return TP_OK;
}

} // namespace tp_linpsf
