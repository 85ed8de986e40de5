// aperture.hip -- A6: per-cadence in-mask flux / error / centroid / background sums.
//
// Replaces the extraction loop of AperturePhotometry.do_photometry
// (photometry/AperturePhotometry/photometry.py:172-201).
//
// Mapping (gfx950): workgroups of 256 threads, one THREAD per group of 2 consecutive cadences of a target
// (grid.x = cadence blocks of the target, so consecutive workgroups touch the same DRAM pages).
// The cube is time-fastest (BasePhotometry.py:732), so for a given mask pixel the 64 lanes of a
// wavefront read 64 x 16 B = 1 KiB of consecutive cadences: fully coalesced 128-bit loads, and
// only the rows of pixels that are IN the mask are ever touched.  Every thread owns its cadences'
// accumulators, so there is no cross-lane reduction at all; the only LDS use is the ordered list of
// mask pixels (raster order), built once per workgroup by wavefront 0 with ballot compaction.
//
// Arithmetic is the reference's, bit for bit where the reference is float32:
//  * flux     = numpy float32 pairwise np.sum over the mask pixels in raster order (:188):
//               n < 8 sequential from 0; n <= 128: 8 strided accumulators, combined
//               ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), tail added sequentially; n > 128: recursive
//               halves with n2 = n/2 - (n/2)%8  (numpy/core/src/umath/loops_utils.h.src).
//  * flux_err = sqrtf(pairwise float32 sum of err*err)  (:189)
//  * centroid = float64 sums of w*col, w*row, w over pixels with flux > 0 (np.average, :192-196),
//               1-based CCD coordinates (BasePhotometry.get_pixel_grid, BasePhotometry.py:696-706)
//  * background = np.nansum (:201): NaN -> 0, then the same float32 pairwise np.sum; NaN if all NaN (:198-201);
//               no background cube (aperture-only, d_backgrounds NULL) -> flux_background NaN
//  * all-NaN or all-zero flux in the mask -> flux, flux_err, centroid = NaN (:182-185)
#include "aperture_dev.h"

namespace {

using namespace tp_ap;

//--------------------------------------------------------------------------------------------------
// Small kernel: 0 <= M <= 128 mask pixels (a single pairwise leaf)
//--------------------------------------------------------------------------------------------------
template <int VEC>
__global__ __launch_bounds__(256) void tp_aperture_kernel(Args a)
{
	__shared__ int s_list[kMaxList];
	__shared__ int s_M;
	const int target = blockIdx.y + blockIdx.z * 65535;
	if (target >= a.n_targets) return;
	if (a.status && a.status[target] == TP_STATUS_ERROR) return;
	const int P = a.height * a.width;
	const int tid = threadIdx.x;
	const uint8_t* m = a.mask + (int64_t)target * P;
	if (tid < 64) {
		int pn = 0, total = 0;
		compact_mask(m, P, pn, s_list, kMaxList, tid, true, &total);
		if (tid == 0) s_M = total;
	}
	__syncthreads();
	const int M = s_M;
	if (M > kMaxList) return; // handled by tp_aperture_big_kernel
	pack_rows(s_list, M, a.width, tid, (int)blockDim.x);
	__syncthreads();
	extract_small<VEC>(a, target, s_list, M, blockIdx.x * blockDim.x + tid, gridDim.x * blockDim.x);
}

//--------------------------------------------------------------------------------------------------
// Big kernel: M > 128 (numpy's recursive pairwise tree).  Every leaf except the last is a
// multiple of 8 long, so the ordered pixel stream is consumed in groups of 8.
//--------------------------------------------------------------------------------------------------
__device__ void build_leaves(int n, int* leaf_len, unsigned char* leaf_merges, int* n_leaves)
{
	// iterative post-order of:  rec(n): n <= 128 ? leaf : (rec(n2), rec(n - n2), merge)
	int stack_n[kMaxDepth * 2];
	unsigned char stack_state[kMaxDepth * 2];
	int sp = 0, nl = 0;
	stack_n[0] = n; stack_state[0] = 0; sp = 1;
	while (sp > 0) {
		const int cur = stack_n[sp - 1];
		const int state = stack_state[sp - 1];
		if (cur <= 128) {
			if (nl < kMaxLeaves) { leaf_len[nl] = cur; leaf_merges[nl] = 0; }
			nl++;
			sp--;
			continue;
		}
		int n2 = cur / 2;
		n2 -= n2 % 8;
		if (state == 0) { stack_state[sp - 1] = 1; stack_n[sp] = n2; stack_state[sp] = 0; sp++; }
		else if (state == 1) { stack_state[sp - 1] = 2; stack_n[sp] = cur - n2; stack_state[sp] = 0; sp++; }
		else { if (nl - 1 < kMaxLeaves) leaf_merges[nl - 1]++; sp--; }
	}
	*n_leaves = nl;
}

template <int VEC>
__global__ __launch_bounds__(512) void tp_aperture_big_kernel(Args a)
{
	__shared__ int s_list[kChunk];
	__shared__ int s_leaf_len[kMaxLeaves];
	__shared__ unsigned char s_leaf_merges[kMaxLeaves];
	__shared__ int s_M, s_nleaves, s_n, s_pnext;
	int target = blockIdx.x;
	if (a.big_list) { // work list from the fused kernel: most launches have nothing to do
		if ((int)blockIdx.x >= a.big_list[0]) return;
		target = a.big_list[1 + blockIdx.x];
	}
	if (a.status && a.status[target] == TP_STATUS_ERROR) return;
	const int P = a.height * a.width;
	const int tid = threadIdx.x;
	const uint8_t* m = a.mask + (int64_t)target * P;
	if (tid < 64) {
		int pn = 0, total = 0;
		compact_mask(m, P, pn, s_list, 0, tid, true, &total);
		if (tid == 0) s_M = total;
	}
	__syncthreads();
	const int M = s_M;
	if (M <= kMaxList) return; // handled by tp_aperture_kernel
	if (tid == 0) { build_leaves(M, s_leaf_len, s_leaf_merges, &s_nleaves); s_pnext = 0; }
	__syncthreads();
	if (s_nleaves > kMaxLeaves) return; // host validates height*width so this cannot trigger

	const int col0 = a.stamps[target * 4 + 2] + 1;
	const int row0 = a.stamps[target * 4 + 0] + 1;
	const int64_t tb = target_base(a, target, P);
	const int origin = stack_origin(a, target);
	const float* img = a.images + tb;
	const float* err = a.images_err + tb;
	const float* bkg = (a.bkg_mode == 0) ? (a.backgrounds + tb) : (a.backgrounds + (int64_t)target * a.bkg_series_pitch);
	const int nq = (a.n_cad + VEC - 1) / VEC;
	const int nrounds_q = (nq + (int)blockDim.x - 1) / (int)blockDim.x;

	for (int qr = 0; qr < nrounds_q; qr++) {
		const int q = qr * blockDim.x + tid;
		const bool active = q < nq;
		const int k0 = q * VEC;
		CadState<VEC> st;
		st.init();
		float bser[VEC], ssub[VEC];
		if (active && a.bkg_mode == 1) Vec<VEC>::load(bkg + k0, bser);
		if (active && a.subtract) Vec<VEC>::load(a.subtract + (int64_t)target * a.subtract_pitch + k0, ssub);
		float stk_f[VEC][kMaxDepth], stk_e[VEC][kMaxDepth], stk_b[VEC][kMaxDepth];
		int sp = 0;
		int leaf = 0, pos_in_leaf = 0;
		int done = 0; // mask pixels consumed so far

		__syncthreads();
		if (tid == 0) s_pnext = 0;
		__syncthreads();

		while (done < M) {
			// stage the next <= kChunk ordered mask pixels (a multiple of 8 unless it is the end)
			if (tid < 64) {
				int pn = s_pnext;
				const int n = compact_mask(m, P, pn, s_list, kChunk, tid, false, nullptr);
				if (tid == 0) { s_n = n; s_pnext = pn; }
			}
			__syncthreads();
			const int n = s_n;
			if (active) {
				int i = 0;
				while (i < n) {
					const int L = s_leaf_len[leaf];
					const int Lblk = L - (L & 7);
					if (pos_in_leaf < Lblk) {
						// a full group of 8 (groups never straddle a chunk: kChunk % 8 == 0 and only the last leaf has a tail)
#pragma unroll
						for (int j = 0; j < 8; j++) {
							const int p = s_list[i + j];
							const int pr = p / a.width;
							const int pc = p - pr * a.width;
							const int64_t off = (int64_t)pixel_row(a, origin, p, pr, pc) * a.t_pitch + k0;
							float v[VEC], ee[VEC], bb[VEC];
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
							st.side(v, (double)(col0 + pc), (double)(row0 + pr));
							float y[VEC];
							st.bkg_terms(bb, y);
#pragma unroll
							for (int c = 0; c < VEC; c++) {
								const float e2 = ee[c] * ee[c];
								if (pos_in_leaf == 0) { st.r[c][j] = v[c]; st.e[c][j] = e2; st.bk[c][j] = y[c]; }
								else { st.r[c][j] += v[c]; st.e[c][j] += e2; st.bk[c][j] += y[c]; }
							}
						}
						i += 8;
						pos_in_leaf += 8;
						if (pos_in_leaf == Lblk) {
#pragma unroll
							for (int c = 0; c < VEC; c++) { st.fres[c] = combine8(st.r[c]); st.eres[c] = combine8(st.e[c]); st.bres[c] = combine8(st.bk[c]); }
						}
					} else {
						// tail element of the (last) leaf
						const int p = s_list[i];
						const int pr = p / a.width;
						const int pc = p - pr * a.width;
						const int64_t off = (int64_t)pixel_row(a, origin, p, pr, pc) * a.t_pitch + k0;
						float v[VEC], ee[VEC], bb[VEC];
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
						st.side(v, (double)(col0 + pc), (double)(row0 + pr));
						float y[VEC];
						st.bkg_terms(bb, y);
#pragma unroll
						for (int c = 0; c < VEC; c++) { st.fres[c] += v[c]; st.eres[c] += ee[c] * ee[c]; st.bres[c] += y[c]; }
						i += 1;
						pos_in_leaf += 1;
					}
					if (pos_in_leaf == L) {
						// leaf complete: push, then perform the merges that follow it in post-order
#pragma unroll
						for (int c = 0; c < VEC; c++) { stk_f[c][sp] = st.fres[c]; stk_e[c][sp] = st.eres[c]; stk_b[c][sp] = st.bres[c]; }
						sp++;
						for (int mm = 0; mm < (int)s_leaf_merges[leaf]; mm++) {
#pragma unroll
							for (int c = 0; c < VEC; c++) {
								stk_f[c][sp - 2] = stk_f[c][sp - 2] + stk_f[c][sp - 1];
								stk_e[c][sp - 2] = stk_e[c][sp - 2] + stk_e[c][sp - 1];
								stk_b[c][sp - 2] = stk_b[c][sp - 2] + stk_b[c][sp - 1];
							}
							sp--;
						}
						leaf++;
						pos_in_leaf = 0;
					}
				}
			}
			done += n;
			__syncthreads();
		}
		if (active) {
#pragma unroll
			for (int c = 0; c < VEC; c++) { st.fres[c] = 0.f + stk_f[c][0]; st.eres[c] = 0.f + stk_e[c][0]; st.bres[c] = 0.f + stk_b[c][0]; }
			store_outputs<VEC>(a, target, k0, st, M);
		}
	}
}

} // namespace

void* tp_ctx_scratch(tp_ctx* ctx, size_t bytes)
{
	if (ctx->scratch_bytes < bytes) {
		if (ctx->scratch) (void)hipFree(ctx->scratch);
		ctx->scratch = nullptr; ctx->scratch_bytes = 0;
		if (tp_device_alloc(ctx, &ctx->scratch, bytes) != hipSuccess) { ctx->scratch = nullptr; return nullptr; }
		ctx->scratch_bytes = bytes;
	}
	return ctx->scratch;
}

int tp_aperture_extract_big(tp_ctx* ctx, const tp_ap::Args& a, bool vec4)
{
	const int vec = vec4 ? 4 : 1;
	const int nq = (a.n_cad + vec - 1) / vec;
	int threads = ((nq + 63) / 64) * 64;
	if (threads > 512) threads = 512;
	dim3 grid((unsigned)a.n_targets), block((unsigned)threads);
	if (vec4) TP_LAUNCH(ctx, TPK_APERTURE_BIG, tp_aperture_big_kernel<4>, grid, block, 0, a);
	else TP_LAUNCH(ctx, TPK_APERTURE_BIG, tp_aperture_big_kernel<1>, grid, block, 0, a);
	TP_LAUNCH_CHECK(ctx, "tp_aperture_big_kernel");
	return TP_OK;
}

// the two extraction launches (masks up to 128 pixels; the others) for a filled argument block
static int launch_extract(tp_ctx* ctx, const Args& a, bool vec4)
{
	// Small masks (the common case): 2 cadences per thread (64-bit loads, 512 B per wavefront instruction) keep the
	// kernel near 100 VGPRs so that several 256-thread workgroups share a CU and overlap their mask-list prologue,
	// loads and stores; 4 cadences per thread needed 256 VGPRs -> one workgroup per CU.
	{
		const int vec = vec4 ? 2 : 1;
		const int nq = (a.n_cad + vec - 1) / vec;
		const int threads = 256;
		const unsigned gy = (unsigned)((a.n_targets < 65535) ? a.n_targets : 65535);
		const unsigned gz = (unsigned)((a.n_targets + 65534) / 65535);
		dim3 grid((unsigned)((nq + threads - 1) / threads), gy, gz), block((unsigned)threads);
		if (vec4) TP_LAUNCH(ctx, TPK_APERTURE, tp_aperture_kernel<2>, grid, block, 0, a);
		else TP_LAUNCH(ctx, TPK_APERTURE, tp_aperture_kernel<1>, grid, block, 0, a);
		TP_LAUNCH_CHECK(ctx, "tp_aperture_kernel");
	}
	// Masks above 128 pixels (rare): the recursive pairwise tree, one workgroup per target
	return tp_aperture_extract_big(ctx, a, vec4);
}

extern "C" int tp_aperture_extract(tp_ctx* ctx, const tp_cube_desc* desc,
	const float* d_images, const float* d_images_err, const float* d_backgrounds,
	int32_t bkg_mode, int64_t bkg_series_pitch, const float* d_subtract, int64_t subtract_pitch,
	const uint8_t* d_mask, const int32_t* d_stamps, const int32_t* d_status,
	double* d_flux, double* d_flux_err, double* d_flux_background,
	double* d_centroid_col, double* d_centroid_row, int64_t out_pitch)
{
	TP_CHECK_CTX(ctx);
	TP_API_BEGIN
	TP_REQUIRE(ctx, tp_desc_ok(desc), "tp_aperture_extract: bad cube descriptor");
	TP_REQUIRE(ctx, d_images && d_images_err && d_mask && d_stamps, "tp_aperture_extract: null input pointer");
	TP_REQUIRE(ctx, d_flux && d_flux_err && d_centroid_col && d_centroid_row, "tp_aperture_extract: null output pointer");
	TP_REQUIRE(ctx, d_flux_background || !d_backgrounds, "tp_aperture_extract: backgrounds given but no flux_background output");
	TP_REQUIRE(ctx, out_pitch >= desc->n_cad, "tp_aperture_extract: out_pitch < n_cad");
	TP_REQUIRE(ctx, bkg_mode == 0 || bkg_mode == 1, "tp_aperture_extract: bkg_mode must be 0 (cube) or 1 (series)");
	if (!d_backgrounds) bkg_mode = 2; // aperture-only: no background input, flux_background (if given) = NaN
	TP_REQUIRE(ctx, bkg_mode != 1 || bkg_series_pitch >= desc->n_cad, "tp_aperture_extract: bad bkg_series_pitch");
	TP_REQUIRE(ctx, (int64_t)desc->height * desc->width <= (int64_t)kMaxLeaves * 64, "tp_aperture_extract: stamp too large");
	if (desc->n_targets == 0 || desc->n_cad == 0) return TP_OK;

	Args a;
	a.images = d_images; a.images_err = d_images_err; a.backgrounds = d_backgrounds;
	a.bkg_mode = bkg_mode; a.bkg_series_pitch = bkg_series_pitch;
	a.subtract = d_subtract; a.subtract_pitch = subtract_pitch;
	a.mask = d_mask; a.stamps = d_stamps; a.status = d_status;
	a.flux = d_flux; a.flux_err = d_flux_err; a.flux_bkg = d_flux_background;
	a.ccol = d_centroid_col; a.crow = d_centroid_row;
	a.out_pitch = out_pitch; a.n_cad = desc->n_cad; a.height = desc->height; a.width = desc->width;
	a.t_pitch = desc->t_pitch; a.n_targets = desc->n_targets; a.big_list = nullptr;

	bool vec4 = tp_vec4_ok(d_images, desc->t_pitch) && tp_vec4_ok(d_images_err, desc->t_pitch);
	if (bkg_mode == 0) vec4 = vec4 && tp_vec4_ok(d_backgrounds, desc->t_pitch);
	else if (bkg_mode == 1) vec4 = vec4 && tp_vec4_ok(d_backgrounds, bkg_series_pitch);
	TP_REQUIRE(ctx, d_subtract == nullptr || subtract_pitch >= desc->n_cad, "tp_aperture_extract: bad subtract pitch");
	if (d_subtract) vec4 = vec4 && tp_vec4_ok(d_subtract, subtract_pitch);
	return launch_extract(ctx, a, vec4);
	TP_API_END(ctx)
}

// The same extraction with the pixels' time series taken from the TIME-MAJOR stacks of a CCD region (tp_frames_transpose) instead of
// per-target cubes: what BasePhotometry._load_cube + do_photometry's loop read (BasePhotometry.py:720-751, photometry.py:172-201),
// without a cut -- the series of a mask pixel is one contiguous row of the stack.  Same kernels, same order of operations: the
// results equal tp_aperture_extract's on the cut cubes bit for bit (tests/test_gpu_resize.py).
extern "C" int tp_aperture_extract_stack(tp_ctx* ctx, int32_t n_targets, int32_t n_cad, int32_t height, int32_t width,
	const float* d_images_t, const float* d_images_err_t, const float* d_backgrounds_t, int64_t t_pitch,
	int32_t stack_rows, int32_t stack_cols, int32_t stack_row0, int32_t stack_col0,
	const uint8_t* d_mask, const int32_t* d_stamps, const int32_t* d_status,
	double* d_flux, double* d_flux_err, double* d_flux_background,
	double* d_centroid_col, double* d_centroid_row, int64_t out_pitch)
{
	TP_CHECK_CTX(ctx);
	TP_API_BEGIN
	TP_REQUIRE(ctx, n_targets >= 0 && n_cad >= 0 && height > 0 && width > 0 && t_pitch >= n_cad, "tp_aperture_extract_stack: bad sizes");
	TP_REQUIRE(ctx, stack_rows > 0 && stack_cols > 0 && (int64_t)stack_rows * stack_cols < ((int64_t)1 << 31), "tp_aperture_extract_stack: bad stack shape");
	TP_REQUIRE(ctx, d_images_t && d_images_err_t && d_mask && d_stamps, "tp_aperture_extract_stack: null input pointer");
	TP_REQUIRE(ctx, d_flux && d_flux_err && d_centroid_col && d_centroid_row, "tp_aperture_extract_stack: null output pointer");
	TP_REQUIRE(ctx, d_flux_background || !d_backgrounds_t, "tp_aperture_extract_stack: backgrounds given but no flux_background output");
	TP_REQUIRE(ctx, out_pitch >= n_cad, "tp_aperture_extract_stack: out_pitch < n_cad");
	TP_REQUIRE(ctx, (int64_t)height * width <= (int64_t)kMaxLeaves * 64, "tp_aperture_extract_stack: stamp too large");
	if (n_targets == 0 || n_cad == 0) return TP_OK;
	Args a;
	a.images = d_images_t; a.images_err = d_images_err_t; a.backgrounds = d_backgrounds_t;
	a.bkg_mode = d_backgrounds_t ? 0 : 2; a.bkg_series_pitch = 0;
	a.subtract = nullptr; a.subtract_pitch = 0;
	a.mask = d_mask; a.stamps = d_stamps; a.status = d_status;
	a.flux = d_flux; a.flux_err = d_flux_err; a.flux_bkg = d_flux_background;
	a.ccol = d_centroid_col; a.crow = d_centroid_row;
	a.out_pitch = out_pitch; a.n_cad = n_cad; a.height = height; a.width = width;
	a.t_pitch = t_pitch; a.n_targets = n_targets; a.big_list = nullptr;
	a.stack_cols = stack_cols; a.stack_row0 = stack_row0; a.stack_col0 = stack_col0;
	bool vec4 = tp_vec4_ok(d_images_t, t_pitch) && tp_vec4_ok(d_images_err_t, t_pitch);
	if (d_backgrounds_t) vec4 = vec4 && tp_vec4_ok(d_backgrounds_t, t_pitch);
	return launch_extract(ctx, a, vec4);
	TP_API_END(ctx)
}
