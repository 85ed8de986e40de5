// fused.hip -- the whole aperture hot path of one target in ONE wavefront:
//   A1 sum image -> A2..A5b K2P2 mask (+A7 contamination) -> A6 extraction,
// i.e. AperturePhotometry.do_photometry (photometry/AperturePhotometry/photometry.py:44-257, the first
// pass over a fixed stamp) without the sum image or the mask ever leaving the compute unit.
//
// Why fuse (gfx950): A1 and A6 are HBM streaming, K2P2 is ~250 us of latency-bound LDS work per target that
// leaves the memory system idle; as three kernels each phase runs alone on the chip.  Here a wavefront owns a
// target from the first load to the last store; the 12 wavefronts resident per CU (LDS: 13 KB each at 15x15) are at
// different phases at any time, so the streaming of some targets hides the mask building of the others, and the
// whole batch is one launch (no launch gaps, no sum image / mask round trip through HBM).
//
// Arithmetic is that of the three stand-alone kernels (same device functions: sumimage_dev.h, k2p2_core.h,
// aperture_dev.h), so the outputs are bit-identical to tp_sumimage + tp_k2p2_masks + tp_aperture_extract.
// Masks above 128 pixels (rare) are left to tp_aperture_big_kernel, launched right after on the same stream.
//
// LDS plan per wavefront: the K2P2 work arrays (k2p2::shared_bytes); during A1 the good-cadence flags alias the
// phase-shared region of the K2P2 arrays (k.srt), during A6 the mask pixel list aliases k.lab.
#include "sumimage_dev.h"
#include "aperture_dev.h"
#include "k2p2_args.h"
#include <algorithm>
#include <vector>

namespace {

constexpr int kRows = 8; // pixel rows per A1 step: 8 x 1 KiB loads in flight per lane

// BKG: 0 = background cube, 1 = one background series per target, 2 = no background (aperture-only)
// A1: the sum image is formed here (streams the whole images cube); false: it is an input (tp_background_sumimage formed it
// while it streamed the raw cube for the background) and the kernel reads in-mask pixel rows only
template <int VEC, bool VEC4, bool HAS_SUB, int BKG, bool A1>
__global__ __launch_bounds__(64, 2) void tp_aperture_fused_kernel(tp_ap::Args a, k2p2::BatchArgs ka, k2p2::Params prm,
	const double* __restrict__ twid, const int32_t* __restrict__ quality, int64_t quality_stride, uint32_t bitmask,
	double* __restrict__ sumimage_out, const int32_t* __restrict__ order)
{
	extern __shared__ __align__(16) unsigned char smem[];
	// workgroups start in the order of their index: `order` (a permutation of the targets, brightest first) makes the launch's last,
	// partly filled round the one of the cheapest targets
	const int target = order ? order[blockIdx.x] : (int)blockIdx.x;
	const int lane = threadIdx.x;
	k2p2::Shared k;
	k2p2::shared_carve(k, smem, ka.H, ka.W, lane, twid);
	const int P = ka.H * ka.W;

	// ---------------- A1: sum image into LDS (k.S) and HBM ----------------
	if constexpr (!A1) {
		const double* in = sumimage_out + (int64_t)target * P;
		for (int p = lane; p < P; p += 64) k.S[p] = in[p];
	} else {
	unsigned char* good = reinterpret_cast<unsigned char*>(k.srt);
	tp_sum::stage_good(good, quality + (int64_t)target * quality_stride, bitmask, a.n_cad, lane, 64);
	__syncthreads();
	{
		const float* base = a.images + (int64_t)target * P * a.t_pitch;
		const float* sub = a.subtract ? (a.subtract + (int64_t)target * a.subtract_pitch) : nullptr;
		double* o = sumimage_out + (int64_t)target * P;
		int p = 0;
		if (VEC4) {
			// Flat software pipeline over (row group, quad step): the loads of step i+1 are issued before step i is
			// accumulated, also across row groups, so every lane keeps kRows..2*kRows 16-byte loads in flight all the
			// time (the wavefront count per CU is fixed by the K2P2 LDS footprint, bytes in flight must come from here).
			// Per lane the cadences are still added in increasing order and reduced by the same tree: same result.
			const int nq = ((a.n_cad + 3) & ~3) >> 2;
			const int spg = (nq + 63) >> 6;               // quad steps per row group
			const int ngroups = (P + kRows - 1) / kRows;  // the last group may be partial: rows clamp to P-1, results dropped
			const uint32_t* good4 = reinterpret_cast<const uint32_t*>(good);
			const float4* sub4 = reinterpret_cast<const float4*>(sub);
			const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
			const int total = ngroups * spg;
			// ping-pong register buffers: no copy between "next" and "current", so the waitcnt before a consume only
			// covers the older buffer and the newer one stays in flight
			float4 bufA[kRows], bufB[kRows];
			float4 subA = zero4, subB = zero4; // the step's quad of the subtracted series travels with the buffer
			auto issue = [&](float4 (&buf)[kRows], float4& sbuf, int st) {
				// unconditional straight-line loads (steps past the end re-read the last one, lanes past the row end re-read
				// its last quad; both are ignored by consume) so that nothing but the ping-pong order decides the waitcnts
				st = (st < total) ? st : (total - 1);
				const int g = st / spg, i = st - g * spg;
				int qd = lane + (i << 6);
				qd = (qd < nq) ? qd : (nq - 1);
				if (HAS_SUB) sbuf = sub4[qd];
#pragma unroll
				for (int j = 0; j < kRows; j++) {
					int r = g * kRows + j;
					r = (r < P) ? r : (P - 1);
					buf[j] = reinterpret_cast<const float4*>(base + (int64_t)r * a.t_pitch)[qd];
				}
			};
			double sacc[kRows];
			int nacc[kRows];
#pragma unroll
			for (int j = 0; j < kRows; j++) { sacc[j] = 0.0; nacc[j] = 0; }
			auto consume = [&](const float4 (&buf)[kRows], const float4& sa, int st) {
				const bool valid = st < total;
				const int g = st / spg, i = st - g * spg;
				const int qd = lane + (i << 6);
				if (valid && qd < nq) {
					const uint32_t gd = good4[qd];
#pragma unroll
					for (int j = 0; j < kRows; j++) {
						float4 v = buf[j];
						if (HAS_SUB) { v.x -= sa.x; v.y -= sa.y; v.z -= sa.z; v.w -= sa.w; }
						tp_sum::acc1(v.x, gd & 0xffu, sacc[j], nacc[j]); tp_sum::acc1(v.y, gd & 0xff00u, sacc[j], nacc[j]);
						tp_sum::acc1(v.z, gd & 0xff0000u, sacc[j], nacc[j]); tp_sum::acc1(v.w, gd & 0xff000000u, sacc[j], nacc[j]);
					}
				}
				if (valid && i == spg - 1) {
					// the kRows shuffle trees interleaved (independent chains), same tree per row as tp_sum::wave_sum_*
#pragma unroll
					for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
						for (int j = 0; j < kRows; j++) { sacc[j] += __shfl_down(sacc[j], off, 64); nacc[j] += __shfl_down(nacc[j], off, 64); }
					}
					if (lane == 0) {
#pragma unroll
						for (int j = 0; j < kRows; j++) {
							const int r = g * kRows + j;
							if (r < P) { const double m = (nacc[j] > 0) ? sacc[j] / (double)nacc[j] : __builtin_nan(""); k.S[r] = m; o[r] = m; }
						}
					}
#pragma unroll
					for (int j = 0; j < kRows; j++) { sacc[j] = 0.0; nacc[j] = 0; }
				}
			};
			issue(bufA, subA, 0);
			issue(bufB, subB, 1);
			for (int st = 0; st < total; st += 2) {
				consume(bufA, subA, st);
				issue(bufA, subA, st + 2);
				consume(bufB, subB, st + 1);
				issue(bufB, subB, st + 3);
			}
		} else {
			for (; p < P; p++) {
				const double m = tp_sum::row_mean_scalar(base + (int64_t)p * a.t_pitch, sub, good, a.n_cad, lane);
				if (lane == 0) { k.S[p] = m; o[p] = m; }
			}
		}
	}
	}
	__syncthreads();

	// ---------------- A2..A5b, A7: the mask, from the LDS-resident sum image ----------------
	k2p2::Target t;
	k2p2::make_target(ka, target, t);
	t.S = k.S;
	const int status = k2p2::run_target(k, prm, t);
	if (status == TP_STATUS_ERROR) return; // photometry.py: the plugin stops, nothing is extracted

	// ---------------- A6: extraction over the mask pixels (k.res), all cadences ----------------
	int* s_list = reinterpret_cast<int*>(k.lab);
	int pn = 0, M = 0;
	tp_ap::compact_mask(k.res, P, pn, s_list, tp_ap::kMaxList, lane, true, &M);
	__syncthreads();
	if (M > tp_ap::kMaxList) { // tp_aperture_big_kernel takes it from the mask in HBM
		if (lane == 0 && a.big_list) a.big_list[1 + atomicAdd(&a.big_list[0], 1)] = target;
		return;
	}
	tp_ap::pack_rows(s_list, M, a.width, lane, 64);
	__syncthreads();
	// the series a lane needs in every pixel group (subtracted series, background series: the same array in the background-in-
	// the-step configuration) are staged once in the part of the LDS the mask builder no longer needs: [smem, s_list)
	const float* lds_sub = nullptr;
	const float* lds_ser = nullptr;
	{
		constexpr bool SER = (BKG == 1);
		const float* gsub = HAS_SUB ? (a.subtract + (int64_t)target * a.subtract_pitch) : nullptr;
		const float* gser = SER ? (a.backgrounds + (int64_t)target * a.bkg_series_pitch) : nullptr;
		const int nflt = (a.n_cad + VEC - 1) / VEC * VEC;                       // what the VEC-wide reads touch
		const int room = (int)((reinterpret_cast<unsigned char*>(s_list) - smem) / sizeof(float));
		float* stage = reinterpret_cast<float*>(smem);
		const bool same = HAS_SUB && SER && (gsub == gser);
		const int need = ((HAS_SUB ? 1 : 0) + ((SER && !same) ? 1 : 0)) * nflt;
		if ((HAS_SUB || SER) && need > 0 && need <= room) {
			if (HAS_SUB) { for (int i = lane; i < nflt; i += 64) stage[i] = gsub[(i < a.n_cad) ? i : (a.n_cad - 1)]; lds_sub = stage; }
			if (SER) {
				if (same) lds_ser = stage;
				else { float* st2 = stage + (HAS_SUB ? nflt : 0); for (int i = lane; i < nflt; i += 64) st2[i] = gser[(i < a.n_cad) ? i : (a.n_cad - 1)]; lds_ser = st2; }
			}
			__syncthreads();
		}
	}
	if ((HAS_SUB && lds_sub) || (BKG == 1 && lds_ser)) tp_ap::extract_small_stream<VEC, HAS_SUB, BKG, true>(a, target, s_list, M, lane, 64, lds_sub, lds_ser);
	else tp_ap::extract_small_stream<VEC, HAS_SUB, BKG, false>(a, target, s_list, M, lane, 64);
}

} // namespace

// The order in which the targets of a large batch are launched: brightest first.  A target's work grows with its brightness (more
// pixels above the threshold for the mask builder, a larger mask for the extraction) and a launch of 10 000 one-wavefront targets is
// 3.26 rounds of the chip: with the cheap targets last the partly filled tail is short (1.40 -> 1.35 ms, measured).  ANY permutation
// gives the same results, so the order is cached per (magnitude array, size) without regard to the array's contents: a stale order
// costs time, never correctness.  Built on the host the first time (one small download).
static const int32_t* fused_launch_order(tp_ctx* ctx, const double* d_tmag, int n)
{
	if (n < 4096 || d_tmag == nullptr) return nullptr;      // less than a round and a half: nothing to gain
	if (ctx->order && ctx->order_key == (const void*)d_tmag && ctx->order_n == n) return ctx->order;
	std::vector<double> tm((size_t)n);
	if (hipMemcpyAsync(tm.data(), d_tmag, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) return nullptr;
	if (hipStreamSynchronize(ctx->stream) != hipSuccess) return nullptr;
	std::vector<int32_t> idx((size_t)n);
	for (int i = 0; i < n; ++i) idx[(size_t)i] = i;
	std::stable_sort(idx.begin(), idx.end(), [&](int32_t x, int32_t y) { return tm[(size_t)x] < tm[(size_t)y]; });   // (NaN magnitudes: wherever they fall)
	if (ctx->order_n < n) {
		if (ctx->order) (void)hipFree(ctx->order);
		ctx->order = nullptr; ctx->order_n = 0;
		void* p = nullptr;
		if (tp_device_alloc(ctx, &p, (size_t)n * sizeof(int32_t)) != hipSuccess) return nullptr;
		ctx->order = static_cast<int32_t*>(p);
	}
	if (hipMemcpy(ctx->order, idx.data(), (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess) { ctx->order_key = nullptr; return nullptr; }
	ctx->order_key = d_tmag; ctx->order_n = n;
	return ctx->order;
}

static int aperture_photometry_impl(tp_ctx* ctx, const tp_cube_desc* desc, bool given_sumimage,
	const float* d_images, const float* d_images_err, const float* d_backgrounds, int32_t bkg_mode, int64_t bkg_series_pitch,
	const float* d_subtract, int64_t subtract_pitch,
	const int32_t* d_quality, int64_t quality_target_stride, uint32_t bitmask,
	const int64_t* d_cat_offsets, const float* d_cat_column_stamp, const float* d_cat_row_stamp, const float* d_cat_tmag,
	const float* d_cat_column, const float* d_cat_row, const int64_t* d_cat_starid,
	const double* d_target_pos_row, const double* d_target_pos_column, const double* d_target_tmag, const int64_t* d_target_starid,
	const int32_t* d_stamps, const int32_t* d_aperture, const tp_k2p2_params* params,
	double* d_sumimage, uint8_t* d_mask, int32_t* d_status, int32_t* d_flags, double* d_contamination, double* d_diag,
	uint8_t* d_cat_in_mask,
	double* d_flux, double* d_flux_err, double* d_flux_background, double* d_centroid_col, double* d_centroid_row, int64_t out_pitch)
{
	TP_CHECK_CTX(ctx);
	TP_API_BEGIN
	TP_REQUIRE(ctx, tp_desc_ok(desc), "tp_aperture_photometry: bad cube descriptor");
	TP_REQUIRE(ctx, d_images && d_images_err && (d_quality || given_sumimage) && d_stamps && d_aperture && d_cat_offsets
		&& d_target_pos_row && d_target_pos_column && d_target_tmag && d_target_starid, "tp_aperture_photometry: null input pointer");
	TP_REQUIRE(ctx, d_sumimage && d_mask && d_status && d_flags && d_contamination, "tp_aperture_photometry: null output pointer");
	TP_REQUIRE(ctx, d_flux && d_flux_err && d_centroid_col && d_centroid_row, "tp_aperture_photometry: null output pointer");
	TP_REQUIRE(ctx, d_flux_background || !d_backgrounds, "tp_aperture_photometry: backgrounds given but no flux_background output");
	TP_REQUIRE(ctx, out_pitch >= desc->n_cad, "tp_aperture_photometry: out_pitch < n_cad");
	TP_REQUIRE(ctx, bkg_mode == 0 || bkg_mode == 1, "tp_aperture_photometry: bkg_mode must be 0 (cube) or 1 (series)");
	if (!d_backgrounds) bkg_mode = 2; // aperture-only (BASELINE configs[1]): no background input, flux_background (if given) = NaN
	TP_REQUIRE(ctx, bkg_mode != 1 || bkg_series_pitch >= desc->n_cad, "tp_aperture_photometry: bad bkg_series_pitch");
	TP_REQUIRE(ctx, quality_target_stride == 0 || quality_target_stride >= desc->n_cad, "tp_aperture_photometry: bad quality stride");
	TP_REQUIRE(ctx, d_subtract == nullptr || subtract_pitch >= desc->n_cad, "tp_aperture_photometry: bad subtract pitch");
	if (desc->n_targets == 0) return TP_OK;
	const int P = desc->height * desc->width;
	// LDS: the K2P2 arrays; the good-cadence flags must fit behind k.S while A1 runs
	const k2p2::SharedLayout lay = k2p2::shared_layout(P);
	size_t shmem = lay.total;
	const size_t a1_bytes = lay.off_region + (size_t)((desc->n_cad + 3) & ~3) + 16; // flags live in the shared region
	if (!given_sumimage && a1_bytes > shmem) shmem = a1_bytes;
	if (shmem > 160 * 1024) {
		// a stamp (or light curve) beyond the LDS-resident per-target state: the three stages one after the other, same results
		// (tp_k2p2_masks keeps its work arrays in HBM for such stamps)
		int rc = given_sumimage ? TP_OK : tp_sumimage(ctx, desc, d_images, d_quality, quality_target_stride, bitmask, d_subtract, subtract_pitch, d_sumimage);
		if (rc != TP_OK) return rc;
		rc = tp_k2p2_masks(ctx, desc->n_targets, desc->height, desc->width, d_sumimage, d_cat_offsets, d_cat_column_stamp, d_cat_row_stamp,
			d_cat_tmag, d_cat_column, d_cat_row, d_cat_starid, d_target_pos_row, d_target_pos_column, d_target_tmag, d_target_starid,
			d_stamps, d_aperture, nullptr, params, d_mask, d_status, d_flags, d_contamination, d_diag, d_cat_in_mask);
		if (rc != TP_OK) return rc;
		return tp_aperture_extract(ctx, desc, d_images, d_images_err, d_backgrounds, (bkg_mode == 2) ? 0 : bkg_mode, bkg_series_pitch,
			d_subtract, subtract_pitch, d_mask, d_stamps, d_status, d_flux, d_flux_err, d_flux_background, d_centroid_col, d_centroid_row, out_pitch);
	}

	if (!ctx->twiddle) {
		double h[2 * k2p2::kGrid];
		for (int j = 0; j < k2p2::kGrid; ++j) {
			h[j] = std::cos(2.0 * k2p2::kPi * (double)j / (double)k2p2::kGrid);
			h[k2p2::kGrid + j] = std::sin(2.0 * k2p2::kPi * (double)j / (double)k2p2::kGrid);
		}
		TP_HIP(ctx, hipMalloc(&ctx->twiddle, sizeof(h)));
		TP_HIP(ctx, hipMemcpy(ctx->twiddle, h, sizeof(h), hipMemcpyHostToDevice));
	}

	k2p2::Params prm = k2p2::default_params();
	if (params) {
		prm.thresh = params->thresh;
		prm.min_no_pixels_in_mask = params->min_no_pixels_in_mask;
		prm.min_for_cluster = params->min_for_cluster;
		prm.extend_overflow = params->extend_overflow;
		prm.ws_thres = params->ws_thres;
		prm.saturation_limit = params->saturation_limit;
	}
	k2p2::BatchArgs ka;
	ka.n_targets = desc->n_targets; ka.H = desc->height; ka.W = desc->width; ka.sumimage = d_sumimage; ka.cat_offsets = d_cat_offsets;
	ka.cat_column_stamp = d_cat_column_stamp; ka.cat_row_stamp = d_cat_row_stamp; ka.cat_tmag = d_cat_tmag;
	ka.cat_column = d_cat_column; ka.cat_row = d_cat_row; ka.cat_starid = d_cat_starid;
	ka.target_pos_row = d_target_pos_row; ka.target_pos_column = d_target_pos_column; ka.target_tmag = d_target_tmag;
	ka.target_starid = d_target_starid; ka.stamps = d_stamps; ka.aperture = d_aperture; ka.cut_override = nullptr;
	ka.mask = d_mask; ka.status = d_status; ka.flags = d_flags; ka.contamination = d_contamination; ka.diag = d_diag;
	ka.cat_in_mask = d_cat_in_mask;

	tp_ap::Args a;
	a.images = d_images; a.images_err = d_images_err; a.backgrounds = d_backgrounds;
	a.bkg_mode = bkg_mode; a.bkg_series_pitch = bkg_series_pitch;
	a.subtract = d_subtract; a.subtract_pitch = subtract_pitch;
	a.mask = d_mask; a.stamps = d_stamps; a.status = d_status;
	a.flux = d_flux; a.flux_err = d_flux_err; a.flux_bkg = d_flux_background;
	a.ccol = d_centroid_col; a.crow = d_centroid_row;
	a.out_pitch = out_pitch; a.n_cad = desc->n_cad; a.height = desc->height; a.width = desc->width;
	a.t_pitch = desc->t_pitch; a.n_targets = desc->n_targets;
	// work list of the big-mask kernel (count + targets), zeroed on the stream before the launch
	a.big_list = static_cast<int32_t*>(tp_ctx_scratch(ctx, ((size_t)desc->n_targets + 1) * sizeof(int32_t)));
	TP_REQUIRE(ctx, a.big_list != nullptr, "tp_aperture_photometry: out of device memory for the work list");
	TP_HIP(ctx, hipMemsetAsync(a.big_list, 0, sizeof(int32_t), ctx->stream));

	bool vec4 = tp_vec4_ok(d_images, desc->t_pitch) && tp_vec4_ok(d_images_err, desc->t_pitch);
	if (bkg_mode == 0) vec4 = vec4 && tp_vec4_ok(d_backgrounds, desc->t_pitch);
	else if (bkg_mode == 1) vec4 = vec4 && tp_vec4_ok(d_backgrounds, bkg_series_pitch);
	if (d_subtract) vec4 = vec4 && tp_vec4_ok(d_subtract, subtract_pitch);

	const int32_t* d_order = fused_launch_order(ctx, d_target_tmag, desc->n_targets);
	const dim3 grid((unsigned)desc->n_targets), block(64);
#define TP_FUSED_LAUNCH_A(V, V4, HS, BK, A1) do { \
		auto kern = tp_aperture_fused_kernel<V, V4, HS, BK, A1>; \
		if (shmem > 64 * 1024) TP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem)); \
		TP_LAUNCH(ctx, TPK_FUSED, kern, grid, block, shmem, a, ka, prm, (const double*)ctx->twiddle, d_quality, quality_target_stride, bitmask, d_sumimage, d_order); \
	} while (0)
#define TP_FUSED_LAUNCH(V, V4, HS, BK) do { if (given_sumimage) TP_FUSED_LAUNCH_A(V, V4, HS, BK, false); else TP_FUSED_LAUNCH_A(V, V4, HS, BK, true); } while (0)
#define TP_FUSED_BKG(V, V4, HS) do { \
		if (bkg_mode == 0) TP_FUSED_LAUNCH(V, V4, HS, 0); else if (bkg_mode == 1) TP_FUSED_LAUNCH(V, V4, HS, 1); else TP_FUSED_LAUNCH(V, V4, HS, 2); \
	} while (0)
	// the configurations of the pipeline: resident (images, errors, background) cubes; raw cubes with the stamp-constant
	// background series subtracted on the fly (B3) and reported as the background (bkg_mode 1); aperture-only (no background)
	if (vec4) { if (d_subtract) TP_FUSED_BKG(2, true, true); else TP_FUSED_BKG(2, true, false); }
	else { if (d_subtract) TP_FUSED_BKG(1, false, true); else TP_FUSED_BKG(1, false, false); }
#undef TP_FUSED_BKG
#undef TP_FUSED_LAUNCH
#undef TP_FUSED_LAUNCH_A
	TP_LAUNCH_CHECK(ctx, "tp_aperture_fused_kernel");
	// masks above 128 pixels: the recursive pairwise tree kernel picks them from the mask / status in HBM
	return tp_aperture_extract_big(ctx, a, vec4);
	TP_API_END(ctx)
}

extern "C" int tp_aperture_photometry(tp_ctx* ctx, const tp_cube_desc* desc,
	const float* d_images, const float* d_images_err, const float* d_backgrounds, int32_t bkg_mode, int64_t bkg_series_pitch,
	const float* d_subtract, int64_t subtract_pitch,
	const int32_t* d_quality, int64_t quality_target_stride, uint32_t bitmask,
	const int64_t* d_cat_offsets, const float* d_cat_column_stamp, const float* d_cat_row_stamp, const float* d_cat_tmag,
	const float* d_cat_column, const float* d_cat_row, const int64_t* d_cat_starid,
	const double* d_target_pos_row, const double* d_target_pos_column, const double* d_target_tmag, const int64_t* d_target_starid,
	const int32_t* d_stamps, const int32_t* d_aperture, const tp_k2p2_params* params,
	double* d_sumimage, uint8_t* d_mask, int32_t* d_status, int32_t* d_flags, double* d_contamination, double* d_diag,
	uint8_t* d_cat_in_mask,
	double* d_flux, double* d_flux_err, double* d_flux_background, double* d_centroid_col, double* d_centroid_row, int64_t out_pitch)
{
	return aperture_photometry_impl(ctx, desc, false, d_images, d_images_err, d_backgrounds, bkg_mode, bkg_series_pitch, d_subtract, subtract_pitch,
		d_quality, quality_target_stride, bitmask, d_cat_offsets, d_cat_column_stamp, d_cat_row_stamp, d_cat_tmag, d_cat_column, d_cat_row, d_cat_starid,
		d_target_pos_row, d_target_pos_column, d_target_tmag, d_target_starid, d_stamps, d_aperture, params, d_sumimage, d_mask, d_status, d_flags,
		d_contamination, d_diag, d_cat_in_mask, d_flux, d_flux_err, d_flux_background, d_centroid_col, d_centroid_row, out_pitch);
}

extern "C" int tp_aperture_photometry_from_sumimage(tp_ctx* ctx, const tp_cube_desc* desc,
	const float* d_images, const float* d_images_err, const float* d_backgrounds, int32_t bkg_mode, int64_t bkg_series_pitch,
	const float* d_subtract, int64_t subtract_pitch,
	const int32_t* d_quality, int64_t quality_target_stride, uint32_t bitmask,
	const int64_t* d_cat_offsets, const float* d_cat_column_stamp, const float* d_cat_row_stamp, const float* d_cat_tmag,
	const float* d_cat_column, const float* d_cat_row, const int64_t* d_cat_starid,
	const double* d_target_pos_row, const double* d_target_pos_column, const double* d_target_tmag, const int64_t* d_target_starid,
	const int32_t* d_stamps, const int32_t* d_aperture, const tp_k2p2_params* params,
	const double* d_sumimage, uint8_t* d_mask, int32_t* d_status, int32_t* d_flags, double* d_contamination, double* d_diag,
	uint8_t* d_cat_in_mask,
	double* d_flux, double* d_flux_err, double* d_flux_background, double* d_centroid_col, double* d_centroid_row, int64_t out_pitch)
{
	// (the kernel only reads the sum image on this path)
	return aperture_photometry_impl(ctx, desc, true, d_images, d_images_err, d_backgrounds, bkg_mode, bkg_series_pitch, d_subtract, subtract_pitch,
		d_quality, quality_target_stride, bitmask, d_cat_offsets, d_cat_column_stamp, d_cat_row_stamp, d_cat_tmag, d_cat_column, d_cat_row, d_cat_starid,
		d_target_pos_row, d_target_pos_column, d_target_tmag, d_target_starid, d_stamps, d_aperture, params, const_cast<double*>(d_sumimage), d_mask, d_status, d_flags,
		d_contamination, d_diag, d_cat_in_mask, d_flux, d_flux_err, d_flux_background, d_centroid_col, d_centroid_row, out_pitch);
}

