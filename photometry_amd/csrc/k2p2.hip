// k2p2.hip -- A2..A5b + A7 on the device: one wavefront (64 threads) per target, all work arrays
// in LDS (see k2p2_core.h for the algorithm and its reference citations).
//
// This stage is O(1) in the number of cadences (it sees only the 15x15 sum image), latency-bound
// integer / float64 work; it is NOT on the HBM roofline -- 10 000 targets x 13 KB of LDS each (15x15 stamps).
#include "common.h"
#ifdef TP_LAB_K2P2_CLOCK
// scratch build only (tools/k2p2_timing.py): cycles per phase of the mask builder, summed over the targets of a launch
__device__ unsigned long long tp_lab_k2clk[24];
#define TP_K2P2_CLOCK(k, i) do { const unsigned long long now_ = __builtin_readcyclecounter(); (k).clk[i] += now_ - (k).clk0; (k).clk0 = now_; } while (0)
#endif
#include "k2p2_args.h"
#include <cmath>
#include <cstdlib>

void* tp_ctx_scratch(tp_ctx* ctx, size_t bytes); // aperture.hip

namespace {

__global__ __launch_bounds__(64, 2) void tp_k2p2_kernel(k2p2::BatchArgs a, k2p2::Params prm, const double* __restrict__ twid)
{
	extern __shared__ __align__(16) unsigned char smem[];
	const int target = blockIdx.x;
	k2p2::Shared k;
	k2p2::shared_carve(k, smem, a.H, a.W, (int)threadIdx.x, twid);
	k2p2::Target t;
	k2p2::make_target(a, target, t);
#ifdef TP_LAB_K2P2_CLOCK
	for (int i = 0; i < 16; ++i) k.clk[i] = 0;
	k.clk0 = __builtin_readcyclecounter();
#endif
	k2p2::run_target(k, prm, t);
#ifdef TP_LAB_K2P2_CLOCK
	if (threadIdx.x == 0) for (int i = 0; i < 16; ++i) atomicAdd(&tp_lab_k2clk[i], k.clk[i]);
#endif
}

// Stamps whose work arrays do not fit the LDS (beyond about 54 x 54 pixels: the default stamps of stars brighter than
// Tmag ~ 3.5, BasePhotometry.py:541-564, and what the resize loop grows around Tmag 4-5 stars): the same code on work arrays in
// a context-owned HBM scratch, one wavefront per target.  A workgroup barrier orders its global accesses like its LDS accesses
// (one wavefront, s_waitcnt vmcnt(0) before the barrier), so nothing else changes; it is slow (every array access is an L2 round
// trip), which is acceptable for the few such stars of a CCD -- upstream they are the slowest targets too.
__global__ __launch_bounds__(64) void tp_k2p2_global_kernel(k2p2::BatchArgs a, k2p2::Params prm, const double* __restrict__ twid,
	unsigned char* __restrict__ scratch, size_t bytes_per_target)
{
	const int target = blockIdx.x;
	k2p2::Shared k;
	k2p2::shared_carve(k, scratch + (size_t)target * bytes_per_target, a.H, a.W, (int)threadIdx.x, twid);
	k2p2::Target t;
	k2p2::make_target(a, target, t);
	k2p2::run_target(k, prm, t);
}

} // namespace

#ifdef TP_LAB_K2P2_CLOCK
extern "C" int tp_lab_k2p2_clocks(unsigned long long* out, int reset) {
	if (reset) { unsigned long long z[24] = {}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(tp_lab_k2clk), z, sizeof(z)); }
	(void)hipDeviceSynchronize();
	return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(tp_lab_k2clk), 24 * sizeof(unsigned long long));
}
#endif

extern "C" int tp_k2p2_masks(tp_ctx* ctx, int32_t n_targets, int32_t height, int32_t width,
	const double* d_sumimage,
	const int64_t* d_cat_offsets, const float* d_cat_column_stamp, const float* d_cat_row_stamp, const float* d_cat_tmag,
	const float* d_cat_column, const float* d_cat_row, const int64_t* d_cat_starid,
	const double* d_target_pos_row, const double* d_target_pos_column, const double* d_target_tmag, const int64_t* d_target_starid,
	const int32_t* d_stamps, const int32_t* d_aperture, const double* d_cut_override, const tp_k2p2_params* params,
	uint8_t* d_mask, int32_t* d_status, int32_t* d_flags, double* d_contamination, double* d_diag, uint8_t* d_cat_in_mask)
{
	TP_CHECK_CTX(ctx);
	TP_API_BEGIN
	TP_REQUIRE(ctx, n_targets >= 0 && height > 0 && width > 0, "tp_k2p2_masks: bad geometry");
	TP_REQUIRE(ctx, d_sumimage && d_cat_offsets && d_target_pos_row && d_target_pos_column && d_target_tmag && d_target_starid
		&& d_stamps && d_aperture, "tp_k2p2_masks: null input pointer");
	TP_REQUIRE(ctx, d_mask && d_status && d_flags && d_contamination, "tp_k2p2_masks: null output pointer");
	if (n_targets == 0) return TP_OK;
	const size_t shmem = k2p2::shared_bytes(height * width);
	const bool in_lds = shmem <= 160 * 1024;
	TP_REQUIRE(ctx, (int64_t)height * width <= 32767, "tp_k2p2_masks: more than 32767 pixels per stamp (signed 16-bit labels and pixel indices)");

	k2p2::Params prm = k2p2::default_params();
	if (params) {
		prm.thresh = params->thresh;
		prm.min_no_pixels_in_mask = params->min_no_pixels_in_mask;
		prm.min_for_cluster = params->min_for_cluster;
		prm.extend_overflow = params->extend_overflow;
		prm.ws_thres = params->ws_thres;
		prm.saturation_limit = params->saturation_limit;
	}
	// twiddle table of the 128-point DFT (cos, sin), once per context
	if (!ctx->twiddle) {
		double h[2 * k2p2::kGrid];
		for (int j = 0; j < k2p2::kGrid; ++j) {
			h[j] = std::cos(2.0 * k2p2::kPi * (double)j / (double)k2p2::kGrid);
			h[k2p2::kGrid + j] = std::sin(2.0 * k2p2::kPi * (double)j / (double)k2p2::kGrid);
		}
		TP_HIP(ctx, hipMalloc(&ctx->twiddle, sizeof(h)));
		TP_HIP(ctx, hipMemcpy(ctx->twiddle, h, sizeof(h), hipMemcpyHostToDevice));
	}
	if (in_lds && shmem > 64 * 1024) {
		TP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(tp_k2p2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
	}

	k2p2::BatchArgs a;
	a.n_targets = n_targets; a.H = height; a.W = width; a.sumimage = d_sumimage; a.cat_offsets = d_cat_offsets;
	a.cat_column_stamp = d_cat_column_stamp; a.cat_row_stamp = d_cat_row_stamp; a.cat_tmag = d_cat_tmag;
	a.cat_column = d_cat_column; a.cat_row = d_cat_row; a.cat_starid = d_cat_starid;
	a.target_pos_row = d_target_pos_row; a.target_pos_column = d_target_pos_column; a.target_tmag = d_target_tmag;
	a.target_starid = d_target_starid; a.stamps = d_stamps; a.aperture = d_aperture; a.cut_override = d_cut_override;
	a.mask = d_mask; a.status = d_status; a.flags = d_flags; a.contamination = d_contamination; a.diag = d_diag;
	a.cat_in_mask = d_cat_in_mask;
	if (in_lds) TP_LAUNCH(ctx, TPK_K2P2, tp_k2p2_kernel, dim3((unsigned)n_targets), dim3(64), shmem, a, prm, (const double*)ctx->twiddle);
	else {
		const size_t per = (shmem + 255) & ~(size_t)255;
		unsigned char* scratch = static_cast<unsigned char*>(tp_ctx_scratch(ctx, per * (size_t)n_targets));
		TP_REQUIRE(ctx, scratch != nullptr, "tp_k2p2_masks: out of device memory for the work arrays of a large stamp");
		TP_LAUNCH(ctx, TPK_K2P2, tp_k2p2_global_kernel, dim3((unsigned)n_targets), dim3(64), 0, a, prm, (const double*)ctx->twiddle, scratch, per);
	}
	TP_LAUNCH_CHECK(ctx, "tp_k2p2_kernel");
	return TP_OK;
	TP_API_END(ctx)
}
