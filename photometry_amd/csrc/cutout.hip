// cutout.hip -- stamp cutter: per-target stamp cubes from a full-frame image stack in HBM.
//
// Replaces BasePhotometry._load_cube, FFI branch (photometry/BasePhotometry.py:720-742): for every target the
// reference reads `hdf[group/%04d][ir1:ir2, ic1:ic2]` for k = 0..T-1 (3 x T chunked HDF5 reads per target) into a
// (rows, cols, times) float32 cube.  Here the frame stack [T][R][C] of a CCD stays resident in HBM (2048 x 2048 x
// 1300 float32 = 21.8 GB per cube, three cubes fit several times into 288 GB) and ALL stamps of a batch are cut in
// one launch: a transposing gather from image-major frames to the time-fastest cube layout.
//
// Mapping (gfx950): one 256-thread workgroup per (target, block of 64 cadences, band of stamp rows).  Load phase: 16 lanes
// per stamp row segment (W <= 16 contiguous floats of a frame row), tile[k][pixel of the band] in LDS; store phase: 64
// consecutive cadences of one pixel per wavefront instruction = 256 B contiguous in the cube.  LDS row stride = band pixels | 1
// floats, so the transposed reads (stride between lanes) fall on distinct banks.  The band (about 64 pixels: 4 rows of a 15-wide
// stamp, 16 KB of LDS) is what lets 8+ workgroups share a CU: the reads are scattered 60-byte pieces whose latency only
// occupancy hides (a whole 15x15 stamp per workgroup, 58 KB of LDS and 8 waves per CU, ran at 8.0 ms per 10 k-stamp cube),
// and it removes any limit on the stamp size (the stamp-resize retries of the aperture plugin cut 39x19, 62x17 ... stamps).
// HBM-bound: algorithmic bytes per target = 2 * P * T * 4 (read + write); the reads are 60-byte segments of 8 KiB
// frame rows, so the real fetch traffic is about twice the algorithmic one (whole 128-byte lines).
// Pixels outside the frame (the reference never produces such stamps: it clips them, BasePhotometry.py:643-679) are NaN.
#include "common.h"

namespace {

constexpr int kCadBlock = 64;   // cadences per workgroup: 256-byte store segments

struct CutArgs {
	const float* frames; int n_frames; int frame_rows, frame_cols; int64_t row_pitch, frame_stride;
	int row_offset, col_offset;          // pixel_offset_row / pixel_offset_col (BasePhotometry.py:724-727)
	const int32_t* stamps; int height, width; int64_t t_pitch; float* cube;
};

__global__ __launch_bounds__(256) void tp_cut_stamps_kernel(CutArgs a, int band_rows)
{
	extern __shared__ float tile[]; // [kCadBlock][ldp]
	const int target = blockIdx.x;
	const int k0 = blockIdx.y * kCadBlock;
	const int tid = threadIdx.x;
	const int W = a.width;
	const int row_first = blockIdx.z * band_rows;                                   // first stamp row of this band
	const int H = (a.height - row_first < band_rows) ? (a.height - row_first) : band_rows;   // rows in this band
	const int P = H * W;
	const int ldp = (band_rows * W) | 1;
	const int r0 = a.stamps[target * 4 + 0] - a.row_offset + row_first;
	const int c0 = a.stamps[target * 4 + 2] - a.col_offset;
	const float nan = __builtin_nanf("");
	// ---- load: (cadence, stamp row) pairs, 16 lanes across the columns of a row segment
	const int wgrp = (W + 15) / 16;                 // 16-lane groups per stamp row (1 for W <= 16)
	const int seg = tid >> 4, lane16 = tid & 15;    // 16 segments in flight per pass
	const int nseg = kCadBlock * H * wgrp;
	constexpr int U = 8; // independent loads in flight per thread before the LDS writes
	// (row, cadence, column group) of this thread's segment, advanced by 16 segments per step without divisions
	// (the kernel is VALU-bound on index arithmetic otherwise: 78 % VALU busy with div / mod per load)
	int g = seg % wgrp, rowk = seg / wgrp;
	int i = rowk % H, kk = rowk / H;
	const int dg = 16 % wgrp, drow = 16 / wgrp;     // advance of (g, rowk) per 16 segments
	const int di = drow % H, dk = drow / H;
	for (int base = seg; base < nseg; base += 16 * U) {
		float v[U];
		int dst[U];
#pragma unroll
		for (int u = 0; u < U; ++u) {
			const int sidx = base + 16 * u;
			const int j = g * 16 + lane16;
			const int k = k0 + kk;
			const int r = r0 + i, c = c0 + j;
			const bool want = (sidx < nseg) && (j < W);
			const bool inside = want && k < a.n_frames && r >= 0 && r < a.frame_rows && c >= 0 && c < a.frame_cols;
			// clamped address, unconditional load: nothing under a branch between the loads
			const int64_t off = inside ? ((int64_t)k * a.frame_stride + (int64_t)r * a.row_pitch + c) : 0;
			const float x = a.frames[off];
			v[u] = inside ? x : nan;
			dst[u] = want ? (kk * ldp + i * W + j) : -1;
			// next segment of this thread: sidx + 16
			g += dg; int carry = 0;
			if (g >= wgrp) { g -= wgrp; carry = 1; }
			i += di + carry; kk += dk;
			if (i >= H) { i -= H; kk++; }
			if (i >= H) { i -= H; kk++; } // di + carry < 2H
		}
#pragma unroll
		for (int u = 0; u < U; ++u) if (dst[u] >= 0) tile[dst[u]] = v[u];
	}
	__syncthreads();
	// ---- store: 64 consecutive cadences of one pixel per wavefront
	const int lane = tid & 63, wave = tid >> 6;
	float* out = a.cube + ((int64_t)target * a.height * W + (int64_t)row_first * W) * a.t_pitch;
	if (k0 + lane < a.n_frames) {
		for (int p = wave; p < P; p += 4) out[(int64_t)p * a.t_pitch + k0 + lane] = tile[lane * ldp + p];
	}
}

} // namespace

extern "C" int tp_cut_stamps(tp_ctx* ctx, const float* d_frames, int32_t n_frames, int32_t frame_rows, int32_t frame_cols,
	int64_t row_pitch, int64_t frame_stride, int32_t row_offset, int32_t col_offset,
	const int32_t* d_stamps, const tp_cube_desc* desc, float* d_cube)
{
	TP_CHECK_CTX(ctx);
	TP_API_BEGIN
	TP_REQUIRE(ctx, tp_desc_ok(desc), "tp_cut_stamps: bad cube descriptor");
	TP_REQUIRE(ctx, d_frames && d_stamps && d_cube, "tp_cut_stamps: null pointer");
	TP_REQUIRE(ctx, n_frames == desc->n_cad, "tp_cut_stamps: the cube must have one cadence per frame");
	TP_REQUIRE(ctx, frame_rows > 0 && frame_cols > 0 && row_pitch >= frame_cols && frame_stride >= (int64_t)frame_rows * row_pitch, "tp_cut_stamps: bad frame geometry");
	if (desc->n_targets == 0 || desc->n_cad == 0) return TP_OK;
	// rows per band: about 64 pixels (16 KB of LDS); at least one row whatever the width
	int band_rows = 64 / desc->width;
	if (band_rows < 1) band_rows = 1;
	if (band_rows > desc->height) band_rows = desc->height;
	const size_t shmem = (size_t)kCadBlock * ((band_rows * desc->width) | 1) * sizeof(float);
	TP_REQUIRE(ctx, shmem <= 160 * 1024, "tp_cut_stamps: stamp rows wider than 639 pixels are not supported");
	const int n_bands = (desc->height + band_rows - 1) / band_rows;
	TP_REQUIRE(ctx, n_bands <= 65535 && (desc->n_cad + kCadBlock - 1) / kCadBlock <= 65535, "tp_cut_stamps: too many bands / cadence blocks");
	CutArgs a;
	a.frames = d_frames; a.n_frames = n_frames; a.frame_rows = frame_rows; a.frame_cols = frame_cols;
	a.row_pitch = row_pitch; a.frame_stride = frame_stride; a.row_offset = row_offset; a.col_offset = col_offset;
	a.stamps = d_stamps; a.height = desc->height; a.width = desc->width; a.t_pitch = desc->t_pitch; a.cube = d_cube;
	if (shmem > 64 * 1024)
		TP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(tp_cut_stamps_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
	dim3 grid((unsigned)desc->n_targets, (unsigned)((desc->n_cad + kCadBlock - 1) / kCadBlock), (unsigned)n_bands);
	TP_LAUNCH(ctx, TPK_CUTOUT, tp_cut_stamps_kernel, grid, dim3(256), shmem, a, band_rows);
	TP_LAUNCH_CHECK(ctx, "tp_cut_stamps_kernel");
	return TP_OK;
	TP_API_END(ctx)
}
