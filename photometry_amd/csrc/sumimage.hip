// sumimage.hip -- A1: the sum image (mean over good-quality cadences, NaN excluded).
//
// Replaces BasePhotometry.sumimage (photometry/BasePhotometry.py:1008-1019) and the FFI
// accumulation of photometry/prepare.py:450-453,459.
//
// Mapping (gfx950): one 256-thread workgroup (4 wavefronts) per target.  A pixel's time series
// is contiguous in HBM (time-fastest cube), so each wavefront owns a pixel row at a time and
// streams it with 128-bit loads, 64 lanes x 16 B = 1 KiB per instruction, fully coalesced.
// The good-quality flags are staged once per workgroup in LDS (one byte per cadence).  Each lane
// keeps a float64 partial sum and an int count; one 64-lane DPP/shuffle reduction per pixel.
// HBM-bound: algorithmic bytes per target = P*T*4 (images) + T*4 (quality) + P*8 (output).
#include "sumimage_dev.h"

namespace {

constexpr int kBlock = 256;
constexpr int kWaves = kBlock / 64;

template <bool VEC4>
__global__ __launch_bounds__(kBlock) void tp_sumimage_kernel(
	const float* __restrict__ images, const int32_t* __restrict__ quality, int64_t quality_stride,
	uint32_t bitmask, double* __restrict__ out, int n_cad, int n_pix, int64_t t_pitch,
	const float* __restrict__ subtract, int64_t subtract_pitch)
{
	extern __shared__ __align__(16) unsigned char good[]; // [round_up(n_cad, 4)]
	const int target = blockIdx.x;
	const int tid = threadIdx.x;
	const int lane = tid & 63;
	const int wave = tid >> 6;
	tp_sum::stage_good(good, quality + (int64_t)target * quality_stride, bitmask, n_cad, tid, kBlock);
	__syncthreads();

	const float* base = images + (int64_t)target * n_pix * t_pitch;
	double* o = out + (int64_t)target * n_pix;
	// optional on-the-fly background subtraction (prepare.py:419-420: float32 image - float32 background)
	const float* sub = subtract ? (subtract + (int64_t)target * subtract_pitch) : nullptr;

	if (VEC4) {
		// two rows (2 x 1 KiB loads in flight per lane and step) per wavefront and iteration
		// (blockIdx.y: a small batch spreads the pixel rows of a target over several workgroups -- every row mean is one
		// wavefront's own work, so the split changes nothing but the latency of a launch with a handful of targets)
		int p = ((int)blockIdx.y * kWaves + wave) * 2;
		for (; p + 1 < n_pix; p += kWaves * 2 * (int)gridDim.y) {
			double m[2];
			tp_sum::rows_mean_vec4<2>(base, t_pitch, p, sub, good, n_cad, lane, m);
			if (lane == 0) { o[p] = m[0]; o[p + 1] = m[1]; }
		}
		if (p < n_pix) {
			double m[1];
			tp_sum::rows_mean_vec4<1>(base, t_pitch, p, sub, good, n_cad, lane, m);
			if (lane == 0) o[p] = m[0];
		}
	} else {
		for (int p = (int)blockIdx.y * kWaves + wave; p < n_pix; p += kWaves * (int)gridDim.y) {
			const double m = tp_sum::row_mean_scalar(base + (int64_t)p * t_pitch, sub, good, n_cad, lane);
			if (lane == 0) o[p] = m;
		}
	}
}

} // namespace

extern "C" int tp_sumimage(tp_ctx* ctx, const tp_cube_desc* desc, const float* d_images,
	const int32_t* d_quality, int64_t quality_target_stride, uint32_t bitmask,
	const float* d_subtract, int64_t subtract_pitch, double* d_sumimage)
{
	TP_CHECK_CTX(ctx);
	TP_API_BEGIN
	TP_REQUIRE(ctx, tp_desc_ok(desc), "tp_sumimage: bad cube descriptor");
	TP_REQUIRE(ctx, d_images && d_quality && d_sumimage, "tp_sumimage: null pointer");
	TP_REQUIRE(ctx, quality_target_stride == 0 || quality_target_stride >= desc->n_cad, "tp_sumimage: bad quality stride");
	if (desc->n_targets == 0) return TP_OK;
	const int n_pix = desc->height * desc->width;
	const size_t shmem = (size_t)((desc->n_cad + 3) & ~3) + 16;
	TP_REQUIRE(ctx, shmem <= 160 * 1024, "tp_sumimage: n_cad too large for the LDS quality table");
	// VEC4 reads the row padding of the last quad: needs pitch % 4 == 0 (so the quad is inside the pitch)
	TP_REQUIRE(ctx, d_subtract == nullptr || subtract_pitch >= desc->n_cad, "tp_sumimage: bad subtract pitch");
	const bool vec4 = tp_vec4_ok(d_images, desc->t_pitch) && (d_subtract == nullptr || (tp_vec4_ok(d_subtract, subtract_pitch)));
	// few targets: up to ~1024 workgroups in all, each with at least one pass of its wavefronts over two rows
	int split = 1;
	if (desc->n_targets < 512) {
		split = 1024 / desc->n_targets;
		const int most = (n_pix + 2 * kWaves - 1) / (2 * kWaves);
		if (split > most) split = most;
		if (split < 1) split = 1;
	}
	dim3 grid((unsigned)desc->n_targets, (unsigned)split), block(kBlock);
	if (vec4) {
		TP_LAUNCH(ctx, TPK_SUMIMAGE, tp_sumimage_kernel<true>, grid, block, shmem,
			d_images, d_quality, quality_target_stride, bitmask, d_sumimage, desc->n_cad, n_pix, desc->t_pitch, d_subtract, subtract_pitch);
	} else {
		TP_LAUNCH(ctx, TPK_SUMIMAGE, tp_sumimage_kernel<false>, grid, block, shmem,
			d_images, d_quality, quality_target_stride, bitmask, d_sumimage, desc->n_cad, n_pix, desc->t_pitch, d_subtract, subtract_pitch);
	}
	TP_LAUNCH_CHECK(ctx, "tp_sumimage_kernel");
	return TP_OK;
	TP_API_END(ctx)
}

// ---- the compact output block of a multi-GPU gather -------------------------------------------------------------------------
// The light curve's flux, flux_err and flux_background are float32 sums (AperturePhotometry/photometry.py:172-201: nansum of
// float32 stamps) widened to float64 on store, so a gathered block may carry them as float32 and lose nothing; the centroids
// and the LinPSF light curve are genuine float64.  tp_block_compact copies the fields of a full block into their places in the
// compact one: kind 0 = bytes as they are, kind 1 = float64 -> float32 (exact for the planes named above).
__global__ __launch_bounds__(256) void tp_f64_to_f32_kernel(const double* __restrict__ in, float* __restrict__ out, uint64_t n)
{
	for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) out[i] = (float)in[i];
}

// 16 bytes per thread and step; the tail by bytes (both pointers 16-byte aligned, or the whole copy goes by bytes)
__global__ __launch_bounds__(256) void tp_blit_kernel(uint4* __restrict__ dst, const uint4* __restrict__ src, uint64_t n16, uint64_t nbytes)
{
	for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256) dst[i] = src[i];
	const uint64_t done = n16 * 16;
	unsigned char* d8 = reinterpret_cast<unsigned char*>(dst);
	const unsigned char* s8 = reinterpret_cast<const unsigned char*>(src);
	if (blockIdx.x == 0) for (uint64_t i = done + threadIdx.x; i < nbytes; i += 256) d8[i] = s8[i];
}

int tp_blit(tp_ctx* ctx, void* dst, const void* src, uint64_t nbytes)
{
	TP_CHECK_CTX(ctx);
	if (nbytes == 0) return TP_OK;
	TP_REQUIRE(ctx, dst && src, "tp_blit: null pointer");
	const bool aligned = ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15u) == 0;
	const uint64_t n16 = aligned ? nbytes / 16 : 0;
	const uint64_t blocks = (n16 + 255) / 256;
	const dim3 grid((unsigned)(blocks < 1 ? 1 : (blocks < 512 ? blocks : 512))), block(256);
	TP_LAUNCH(ctx, TPK_BLIT, tp_blit_kernel, grid, block, 0, static_cast<uint4*>(dst), static_cast<const uint4*>(src), n16, nbytes);
	TP_LAUNCH_CHECK(ctx, "tp_blit_kernel");
	return TP_OK;
}

extern "C" int tp_block_compact(tp_ctx* ctx, const void* d_block, void* d_compact, const tp_block_field* fields, int32_t n_fields)
{
	TP_CHECK_CTX(ctx);
	TP_API_BEGIN
	TP_REQUIRE(ctx, n_fields >= 0 && (n_fields == 0 || (d_block && d_compact && fields)), "tp_block_compact: null pointer");
	for (int i = 0; i < n_fields; ++i) {
		const tp_block_field& f = fields[i];
		if (f.count == 0) continue;
		const char* src = static_cast<const char*>(d_block) + f.src_offset;
		char* dst = static_cast<char*>(d_compact) + f.dst_offset;
		if (f.kind == 0) {
			TP_HIP(ctx, hipMemcpyAsync(dst, src, (size_t)f.count, hipMemcpyDeviceToDevice, ctx->stream));
		} else {
			TP_REQUIRE(ctx, f.kind == 1 && f.src_offset % 8 == 0 && f.dst_offset % 4 == 0, "tp_block_compact: bad field kind or alignment");
			const uint64_t blocks = (f.count + 255) / 256;
			const dim3 grid((unsigned)(blocks < 65536 ? blocks : 65536)), block(256);
			TP_LAUNCH(ctx, TPK_BLOCK_COMPACT, tp_f64_to_f32_kernel, grid, block, 0, reinterpret_cast<const double*>(src), reinterpret_cast<float*>(dst), f.count);
			TP_LAUNCH_CHECK(ctx, "tp_f64_to_f32_kernel");
		}
	}
	return TP_OK;
	TP_API_END(ctx)
}
