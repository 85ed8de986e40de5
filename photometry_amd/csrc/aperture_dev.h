// aperture_dev.h -- device code of A6 shared by the stand-alone extraction kernels (aperture.hip) and the fused
// per-target kernel (fused.hip).  See aperture.hip for the arithmetic contract and its reference citations.
#pragma once
#include "common.h"
#include <cmath>

namespace tp_ap {

constexpr int kMaxList = 128;      // small kernel: mask pixels held in LDS
constexpr int kChunk = 1024;       // big kernel: ordered mask pixels staged per round
constexpr int kMaxLeaves = 4096;   // big kernel: pairwise leaves (each 65..128 pixels)
constexpr int kMaxDepth = 24;

template <int VEC> struct Vec;
template <> struct Vec<4> {
	static __device__ __forceinline__ void load(const float* p, float (&v)[4]) {
		float4 t = *reinterpret_cast<const float4*>(p);
		v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
	}
};
template <> struct Vec<2> {
	static __device__ __forceinline__ void load(const float* p, float (&v)[2]) {
		float2 t = *reinterpret_cast<const float2*>(p);
		v[0] = t.x; v[1] = t.y;
	}
};
template <> struct Vec<1> {
	static __device__ __forceinline__ void load(const float* p, float (&v)[1]) { v[0] = *p; }
};

// Per-thread state for VEC cadences
template <int VEC>
struct CadState {
	float r[VEC][8];      // pairwise accumulators: flux
	float e[VEC][8];      // pairwise accumulators: err^2
	float bk[VEC][8];     // pairwise accumulators: background (NaN -> 0, np.nansum)
	float fres[VEC], eres[VEC], bres[VEC];
	double cw[VEC], ccol[VEC], crow[VEC];
	bool f_allnan[VEC], f_allzero[VEC], b_allnan[VEC];

	__device__ __forceinline__ void init() {
#pragma unroll
		for (int c = 0; c < VEC; c++) {
			fres[c] = 0.f; eres[c] = 0.f; bres[c] = 0.f;
			cw[c] = 0.0; ccol[c] = 0.0; crow[c] = 0.0;
			f_allnan[c] = true; f_allzero[c] = true; b_allnan[c] = true;
		}
	}
	// everything except the pairwise flux / err / background sums
	__device__ __forceinline__ void side(const float (&v)[VEC], double col, double row) {
#pragma unroll
		for (int c = 0; c < VEC; c++) {
			const float x = v[c];
			f_allnan[c] = f_allnan[c] && (x != x);
			f_allzero[c] = f_allzero[c] && (x == 0.f);
			// (branch-free: a value that is not positive enters as +0.0, which changes none of the three sums)
			const double w = (x > 0.f) ? (double)x : 0.0;
			cw[c] += w;
			ccol[c] += col * w;
			crow[c] += row * w;
		}
	}
	// np.nansum (photometry.py:201) = np.sum of the values with NaN replaced by 0: the term that enters the pairwise tree
	__device__ __forceinline__ void bkg_terms(const float (&b)[VEC], float (&y)[VEC]) {
#pragma unroll
		for (int c = 0; c < VEC; c++) {
			const bool fin = (b[c] == b[c]);
			b_allnan[c] = b_allnan[c] && !fin;
			y[c] = fin ? b[c] : 0.f;
		}
	}
};

__device__ __forceinline__ float combine8(const float (&r)[8]) {
	return ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
}

struct Args {
	const float* images; const float* images_err; const float* backgrounds;
	int32_t bkg_mode; int64_t bkg_series_pitch;   // 0 cube, 1 series per target, 2 no background (aperture-only)
	const float* subtract; int64_t subtract_pitch;
	const uint8_t* mask; const int32_t* stamps; const int32_t* status;
	double* flux; double* flux_err; double* flux_bkg; double* ccol; double* crow;
	int64_t out_pitch; int n_cad; int height; int width; int64_t t_pitch; int n_targets;
	// optional work list of the big-mask kernel: big_list[0] = count, big_list[1..] = targets (filled by the fused kernel)
	int32_t* big_list;
	// STACK mode (stack_cols > 0; tp_aperture_extract_stack): images / images_err / backgrounds are not per-target cubes but the
	// TIME-MAJOR stacks of a CCD region, [stack rows x stack_cols][t_pitch] (tp_frames_transpose): pixel (r, c) of a target's stamp
	// is the stack's row (stamps[4 t] - stack_row0 + r) * stack_cols + stamps[4 t + 2] - stack_col0 + c.  Nothing is cut.
	int32_t stack_cols = 0, stack_row0 = 0, stack_col0 = 0;
};

// where the time series of the pixels of `target` start (cube mode: its cube; stack mode: the stack) and which row of it holds
// stamp pixel p = pr * width + pc
__device__ __forceinline__ int64_t target_base(const Args& a, int target, int P) { return a.stack_cols ? 0 : (int64_t)target * P * a.t_pitch; }
__device__ __forceinline__ int stack_origin(const Args& a, int target) {
	return a.stack_cols ? ((a.stamps[target * 4 + 0] - a.stack_row0) * a.stack_cols + (a.stamps[target * 4 + 2] - a.stack_col0)) : 0;
}
__device__ __forceinline__ int pixel_row(const Args& a, int origin, int p, int pr, int pc) { return a.stack_cols ? (origin + pr * a.stack_cols + pc) : p; }

template <int VEC>
__device__ __forceinline__ void store_outputs(const Args& a, int target, int k0, const CadState<VEC>& st, int M) {
	const int64_t ob = (int64_t)target * a.out_pitch;
	const double nan = __builtin_nan("");
#pragma unroll
	for (int c = 0; c < VEC; c++) {
		const int k = k0 + c;
		if (k >= a.n_cad) continue;
		const bool bad = (M == 0) || st.f_allnan[c] || st.f_allzero[c];
		a.flux[ob + k] = bad ? nan : (double)st.fres[c];
		a.flux_err[ob + k] = bad ? nan : (double)sqrtf(st.eres[c]);
		const bool haspos = st.cw[c] > 0.0;
		a.ccol[ob + k] = (bad || !haspos) ? nan : st.ccol[c] / st.cw[c];
		a.crow[ob + k] = (bad || !haspos) ? nan : st.crow[c] / st.cw[c];
		if (a.flux_bkg) a.flux_bkg[ob + k] = (M == 0 || a.bkg_mode == 2 || st.b_allnan[c]) ? nan : (double)st.bres[c];
	}
}

// Ordered (raster) compaction of the next mask pixels starting at *p_next into list[0..cap), by
// ONE wavefront.  Returns the number stored; advances *p_next.  With count_rest, keeps counting
// (without storing) to the end of the mask and returns the total in *total.
__device__ __forceinline__ int compact_mask(const uint8_t* m, int P, int& p_next, int* list, int cap, int lane,
	bool count_rest, int* total)
{
	int n = 0;
	int p0 = p_next;
	for (; p0 < P; p0 += 64) {
		const int p = p0 + lane;
		const bool in = (p < P) && (m[p] != 0);
		const unsigned long long bal = __ballot(in);
		const int pos = n + __popcll(bal & ((1ull << lane) - 1ull));
		const int cnt = __popcll(bal);
		if (n + cnt > cap) {
			if (!count_rest) {
				// store only what fits, stop *inside* this group: find the pixel where the list fills
				if (in && pos < cap) list[pos] = p;
				// p_next = index of the first pixel NOT stored
				const unsigned long long notstored = __ballot(in && pos >= cap);
				p_next = p0 + (int)__ffsll((long long)notstored) - 1;
				return cap;
			}
			if (in && pos < cap) list[pos] = p;
			n += cnt;
			continue;
		}
		if (in && pos < cap) list[pos] = p;
		n += cnt;
	}
	p_next = P;
	if (total) *total = n;
	return n < cap ? n : cap;
}

// The pixel list of the two small-mask extractions carries a pixel's row beside its index (index in the low, row in the high 16
// bits; a stamp holds at most 65 535 pixels): the extraction turns every list entry into (row, column) once per cadence block,
// and a division by the run-time stamp width is ~30 instructions -- a third of the vector work of a pixel (lab clocks, round 4).
__device__ __forceinline__ void pack_rows(int* list, int M, int width, int lane, int nlanes) {
	for (int i = lane; i < M; i += nlanes) { const int p = list[i]; list[i] = p | ((p / width) << 16); }
}

// Extraction of the cadences q_first, q_first + q_stride, ... (VEC cadences each) of one target with a mask of
// M <= kMaxList pixels listed (raster order) in s_list: a single pairwise leaf.
template <int VEC>
__device__ __forceinline__ void extract_small(const Args& a, int target, const int* s_list, int M, int q_first, int q_stride)
{
	const int P = a.height * a.width;
	const int col0 = a.stamps[target * 4 + 2] + 1; // 1-based CCD column of stamp column 0
	const int row0 = a.stamps[target * 4 + 0] + 1;
	const int64_t tb = target_base(a, target, P);
	const int origin = stack_origin(a, target);
	const float* img = a.images + tb;
	const float* err = a.images_err + tb;
	const float* bkg = (a.bkg_mode == 0) ? (a.backgrounds + tb) : (a.backgrounds + (int64_t)target * a.bkg_series_pitch);
	const bool has_bkg = (a.bkg_mode != 2);
	const int nq = (a.n_cad + VEC - 1) / VEC;
	const int nblk = M - (M & 7);

	for (int q = q_first; q < nq; q += q_stride) {
		const int k0 = q * VEC;
		CadState<VEC> st;
		st.init();
		float bser[VEC], ssub[VEC];
		if (a.bkg_mode == 1) Vec<VEC>::load(bkg + k0, bser);
		if (a.subtract) Vec<VEC>::load(a.subtract + (int64_t)target * a.subtract_pitch + k0, ssub);

		auto fetch = [&](int idx, float (&v)[VEC], float (&e2)[VEC], float (&y)[VEC]) {
			const int pk = s_list[idx];
			const int p = pk & 0xffff;
			const int pr = (int)((unsigned)pk >> 16);
			const int pc = p - pr * a.width;
			const int64_t off = (int64_t)pixel_row(a, origin, p, pr, pc) * a.t_pitch + k0;
			float ee[VEC], bb[VEC];
			Vec<VEC>::load(img + off, v);
			if (a.subtract) {
#pragma unroll
				for (int c = 0; c < VEC; c++) v[c] = v[c] - ssub[c];
			}
			Vec<VEC>::load(err + off, ee);
			if (a.bkg_mode == 0) Vec<VEC>::load(bkg + off, bb);
			else {
#pragma unroll
				for (int c = 0; c < VEC; c++) bb[c] = (a.bkg_mode == 1) ? bser[c] : 0.f;
			}
#pragma unroll
			for (int c = 0; c < VEC; c++) e2[c] = ee[c] * ee[c];
			st.side(v, (double)(col0 + pc), (double)(row0 + pr));
			if (has_bkg) st.bkg_terms(bb, y);
		};

		if (M < 8) {
			for (int i = 0; i < M; i++) {
				float v[VEC], e2[VEC], y[VEC];
				fetch(i, v, e2, y);
#pragma unroll
				for (int c = 0; c < VEC; c++) { st.fres[c] += v[c]; st.eres[c] += e2[c]; if (has_bkg) st.bres[c] += y[c]; }
			}
		} else {
			for (int g = 0; g < nblk; g += 8) {
#pragma unroll
				for (int j = 0; j < 8; j++) {
					float v[VEC], e2[VEC], y[VEC];
					fetch(g + j, v, e2, y);
#pragma unroll
					for (int c = 0; c < VEC; c++) {
						if (g == 0) { st.r[c][j] = v[c]; st.e[c][j] = e2[c]; if (has_bkg) st.bk[c][j] = y[c]; }
						else { st.r[c][j] += v[c]; st.e[c][j] += e2[c]; if (has_bkg) st.bk[c][j] += y[c]; }
					}
				}
			}
#pragma unroll
			for (int c = 0; c < VEC; c++) { st.fres[c] = combine8(st.r[c]); st.eres[c] = combine8(st.e[c]); if (has_bkg) st.bres[c] = combine8(st.bk[c]); }
			for (int i = nblk; i < M; i++) {
				float v[VEC], e2[VEC], y[VEC];
				fetch(i, v, e2, y);
#pragma unroll
				for (int c = 0; c < VEC; c++) { st.fres[c] += v[c]; st.eres[c] += e2[c]; if (has_bkg) st.bres[c] += y[c]; }
			}
		}
		// np.sum = 0 + pairwise_sum (identity-initialised reduce)
#pragma unroll
		for (int c = 0; c < VEC; c++) { st.fres[c] = 0.f + st.fres[c]; st.eres[c] = 0.f + st.eres[c]; st.bres[c] = 0.f + st.bres[c]; }
		store_outputs<VEC>(a, target, k0, st, M);
	}
}

// The same extraction as a flat software pipeline (used by the fused per-target kernel, where one wavefront walks all
// cadence blocks of its target): the mask pixels are consumed in groups of 8 from two ping-pong register buffers, the loads
// of group g+1 -- also across cadence blocks -- are in flight while group g is accumulated, and every load is
// unconditional straight-line code (indices clamp instead of branching) so that only the ping-pong order decides the
// waitcnts.  Pixel order, accumulator assignment and operation order are those of extract_small: identical results.
template <int VEC, bool HAS_SUB, int BKG, bool LDS_SERIES = false>
__device__ __forceinline__ void extract_small_stream(const Args& a, int target, const int* s_list, int M, int q_lane, int q_stride,
	const float* lds_sub = nullptr, const float* lds_ser = nullptr)
{
	// lds_sub / lds_ser: the target's subtracted / background series staged in LDS by the caller (same values as in HBM): a lane's
	// cadences do not change over the pixel groups, so the HBM copy would be re-read once per group (1.85 GB per launch of 10 000
	// targets, measured); LDS reads also stay out of the vector-memory queue that paces the pixel loads
	const int P = a.height * a.width;
	const int col0 = a.stamps[target * 4 + 2] + 1;
	const int row0 = a.stamps[target * 4 + 0] + 1;
	const int64_t tb = (int64_t)target * P * a.t_pitch;
	const float* img = a.images + tb;
	const float* err = a.images_err + tb;
	constexpr bool BKG_CUBE = (BKG == 0), BKG_SERIES = (BKG == 1), HAS_BKG = (BKG != 2);
	const float* bkg = BKG_CUBE ? (a.backgrounds + tb) : (a.backgrounds + (int64_t)target * a.bkg_series_pitch);
	const float* subp = HAS_SUB ? (a.subtract + (int64_t)target * a.subtract_pitch) : nullptr;
	const int nq = (a.n_cad + VEC - 1) / VEC;
	const int nit = (nq + q_stride - 1) / q_stride;
	CadState<VEC> st;
	st.init();
	if (M == 0) {
		for (int it = 0; it < nit; it++) {
			const int q = q_lane + it * q_stride;
			if (q < nq) store_outputs<VEC>(a, target, q * VEC, st, 0);
		}
		return;
	}
	const int nfull = M >> 3, ntail = M & 7;
	const int spq = nfull + (ntail ? 1 : 0); // pixel groups per cadence block
	const int total = nit * spq;

	struct Buf { float v[8][VEC], e[8][VEC], b[BKG_CUBE ? 8 : 1][VEC], ser[VEC], sub[VEC]; };
	Buf bufA, bufB;
	auto issue = [&](Buf& B, int step) {
		step = (step < total) ? step : (total - 1);
		const int it = step / spq, g = step - it * spq;
		int q = q_lane + it * q_stride;
		q = (q < nq) ? q : (nq - 1);
		const int k0 = q * VEC;
		// (a compile-time choice: a pointer that may be LDS or HBM would turn these into flat loads, which wait for everything)
		if (BKG_SERIES) { if (LDS_SERIES) Vec<VEC>::load(lds_ser + k0, B.ser); else Vec<VEC>::load(bkg + k0, B.ser); }
		if (HAS_SUB) { if (LDS_SERIES) Vec<VEC>::load(lds_sub + k0, B.sub); else Vec<VEC>::load(subp + k0, B.sub); }
#pragma unroll
		for (int j = 0; j < 8; j++) {
			int idx = g * 8 + j;
			idx = (idx < M) ? idx : (M - 1);
			// (the list entry is the same in every lane: taken to a scalar register, the row's base address is scalar arithmetic and
			// the load gets a scalar base + the lane's cadence offset)
			const int64_t rowoff = (int64_t)(__builtin_amdgcn_readfirstlane(s_list[idx]) & 0xffff) * a.t_pitch;
			Vec<VEC>::load(img + rowoff + k0, B.v[j]);
			Vec<VEC>::load(err + rowoff + k0, B.e[j]);
			if (BKG_CUBE) Vec<VEC>::load(bkg + rowoff + k0, B.b[j]);
		}
	};
	auto consume = [&](const Buf& B, int step) {
		if (step >= total) return;
		const int it = step / spq, g = step - it * spq;
		const int q = q_lane + it * q_stride;
		const bool full = g < nfull;
		const int cnt = full ? 8 : ntail;
#pragma unroll
		for (int j = 0; j < 8; j++) {
			if (j < cnt) {
				const int pk = __builtin_amdgcn_readfirstlane(s_list[g * 8 + j]);
				const int p = pk & 0xffff;
				const int pr = (int)((unsigned)pk >> 16);
				const int pc = p - pr * a.width;
				float v[VEC], e2[VEC], bb[VEC], y[VEC];
#pragma unroll
				for (int c = 0; c < VEC; c++) {
					v[c] = HAS_SUB ? (B.v[j][c] - B.sub[c]) : B.v[j][c];
					e2[c] = B.e[j][c] * B.e[j][c];
					bb[c] = BKG_CUBE ? B.b[BKG_CUBE ? j : 0][c] : (BKG_SERIES ? B.ser[c] : 0.f);
				}
				st.side(v, (double)(col0 + pc), (double)(row0 + pr));
				if (HAS_BKG) st.bkg_terms(bb, y);
#pragma unroll
				for (int c = 0; c < VEC; c++) {
					if (full) {
						if (g == 0) { st.r[c][j] = v[c]; st.e[c][j] = e2[c]; if (HAS_BKG) st.bk[c][j] = y[c]; }
						else { st.r[c][j] += v[c]; st.e[c][j] += e2[c]; if (HAS_BKG) st.bk[c][j] += y[c]; }
					} else { st.fres[c] += v[c]; st.eres[c] += e2[c]; if (HAS_BKG) st.bres[c] += y[c]; }
				}
			}
		}
		if (full && g == nfull - 1) {
#pragma unroll
			for (int c = 0; c < VEC; c++) { st.fres[c] = combine8(st.r[c]); st.eres[c] = combine8(st.e[c]); if (HAS_BKG) st.bres[c] = combine8(st.bk[c]); }
		}
		if (g == spq - 1) {
			// np.sum = 0 + pairwise_sum (identity-initialised reduce)
#pragma unroll
			for (int c = 0; c < VEC; c++) { st.fres[c] = 0.f + st.fres[c]; st.eres[c] = 0.f + st.eres[c]; st.bres[c] = 0.f + st.bres[c]; }
			if (q < nq) store_outputs<VEC>(a, target, q * VEC, st, M);
			st.init();
		}
	};
	issue(bufA, 0);
	issue(bufB, 1);
	for (int step = 0; step < total; step += 2) {
		consume(bufA, step);
		issue(bufA, step + 2);
		consume(bufB, step + 1);
		issue(bufB, step + 3);
	}
}

} // namespace tp_ap

// Launches tp_aperture_big_kernel (masks above kMaxList pixels; it skips all other targets) on ctx's stream.
int tp_aperture_extract_big(tp_ctx* ctx, const tp_ap::Args& a, bool vec4);

// Grow-only device scratch of the context (at least `bytes`), nullptr on failure.
void* tp_ctx_scratch(tp_ctx* ctx, size_t bytes);
