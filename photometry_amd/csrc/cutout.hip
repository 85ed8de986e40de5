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
//
// Dense batches (round 2): the FRAME-TILE-MAJOR path.  With thousands of stamps per CCD the per-stamp gather above fetches
// every 60-byte row piece as one or two whole 128-byte lines -- 36 GB for the 11.7 GB of a 10 k-stamp cube on a 2048 x 2048
// stack, more than the stack itself (21.8 GB) -- and on crowded fields the same pixels many times over.  The tile-major
// path turns the transposition around: the frame is cut into tiles of 2 rows x 64 columns; a binning pre-pass (three small
// kernels: count, scan, fill; no host round trip) lists the stamps that touch each tile; one 256-thread workgroup per (tile,
// block of 32 cadences) loads the tile of those 32 frames with fully coalesced 256-byte row segments into LDS
// ([cadence][pixel], stride 129 floats, 16.5 KB) and then serves every stamp of its list from LDS: half a wavefront stores the
// 32 cadences of one stamp pixel (one 128-byte line of the cube).  Measured tile shapes (rows x columns x frames, 10 k stamps on
// a 512^2 / 1024^2 / 2048^2 stack, ms): 2x64x32 2.8 / 3.2 / 5.8 (chosen), 2x64x64 3.0 / 3.6 / 6.7, 4x64x32 3.0 / 3.5 / 6.0,
// 2x128x32 3.4 / 3.4 / 5.7, 1x64x32 3.3 / 3.6 / 6.3, 2x32x32 3.4 / 4.0 / 6.0, 2x64x16 5.8 / 5.0 / 8.4 (64-byte stores).  Every frame pixel is read from HBM once, whatever the
// number of stamps that contain it, and never as a partial line.  Tiles without stamps exit at once.
#include "common.h"

void* tp_ctx_scratch(tp_ctx* ctx, size_t bytes); // aperture.hip

namespace {

constexpr int kCadBlock = 64;   // cadences per workgroup: 256-byte store segments

struct CutArgs {
	const float* frames; int n_frames; int frame_rows, frame_cols; int64_t row_pitch, frame_stride;
	int row_offset, col_offset;          // pixel_offset_row / pixel_offset_col (BasePhotometry.py:724-727)
	const int32_t* stamps; int height, width; int64_t t_pitch; float* cube;
	const uint8_t* mask;                 // optional [n_targets][height * width]: only the pixels with a non-zero entry are written
};

// up to kMaxStacks frame stacks cut in ONE launch (the three image groups of a CCD share stamps and geometry)
constexpr int kMaxStacks = 4;
struct StackPtrs { const float* frames[kMaxStacks]; float* cubes[kMaxStacks]; };

// blockIdx.y = cadence block + n_cad_blocks * stack: a small group (the resized stamps of a few targets) is a chain of latency-bound
// launches, and three stacks in three launches were three links of it
__global__ __launch_bounds__(256) void tp_cut_stamps_kernel(CutArgs a, StackPtrs sp, int n_cad_blocks, int band_rows)
{
	extern __shared__ float tile[]; // [kCadBlock][ldp]
	const int target = blockIdx.x;
	const int stack = (int)blockIdx.y / n_cad_blocks;
	a.frames = sp.frames[stack];
	a.cube = sp.cubes[stack];
	const int k0 = ((int)blockIdx.y - stack * n_cad_blocks) * kCadBlock;
	const int tid = threadIdx.x;
	const int W = a.width;
	const int row_first = blockIdx.z * band_rows;                                   // first stamp row of this band
	const int H = (a.height - row_first < band_rows) ? (a.height - row_first) : band_rows;   // rows in this band
	const int P = H * W;
	const int ldp = (band_rows * W) | 1;
	const int r0 = a.stamps[target * 4 + 0] - a.row_offset + row_first;
	const int c0 = a.stamps[target * 4 + 2] - a.col_offset;
	const float nan = __builtin_nanf("");
	const uint8_t* mk0 = a.mask ? (a.mask + (int64_t)target * a.height * W + (int64_t)row_first * W) : nullptr;
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
			// (with a mask only its pixels are fetched: a resized stamp of 35 x 35 holds a mask of a few dozen pixels, and the groups of
			// resized stamps read 2.0 GB per batch of 2 500 targets to write 0.1 -- profiles/r5_frames_traffic.txt, before this test)
			const bool wanted = want && (mk0 == nullptr || mk0[i * W + j] != 0);
			const bool inside = wanted && k < a.n_frames && r >= 0 && r < a.frame_rows && c >= 0 && c < a.frame_cols;
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
	// the padding of the time axis (cadences n_frames .. t_pitch) is written too, as zeros: the caller need not clear the cube
	if (k0 + lane < a.t_pitch) {
		const bool real = k0 + lane < a.n_frames;
		const uint8_t* mk = a.mask ? (a.mask + (int64_t)target * a.height * W + (int64_t)row_first * W) : nullptr;
		for (int p = wave; p < P; p += 4) {
			if (mk && !mk[p]) continue;   // (wave-uniform: p depends on the wavefront only)
			out[(int64_t)p * a.t_pitch + k0 + lane] = real ? tile[lane * ldp + p] : 0.f;
		}
	}
}


//--------------------------------------------------------------------------------------------------
// frame-tile-major path
//--------------------------------------------------------------------------------------------------
// tile shape: 2 rows x 64 columns x 32 frames (measured against 2x64x64, 4x64x32, 2x128x32, 1x64x32, 2x64x16: DESIGN.md section 3)
constexpr int kTileRows = 2, kTileCols = 64, kTilePix = kTileRows * kTileCols;
constexpr int kTileCad = 32;   // frames per tile = lanes per stored pixel run
constexpr int kTileLd = kTilePix + 1;   // LDS stride of one cadence: odd, so that the transposed reads are conflict-free

struct TileGeom { int tiles_x, tiles_y; };

// range of tiles a stamp touches (clipped to the frame); empty if the stamp lies outside
__device__ __forceinline__ bool stamp_tiles(const CutArgs& a, int target, int& ty0, int& ty1, int& tx0, int& tx1)
{
	const int r0 = a.stamps[target * 4 + 0] - a.row_offset, c0 = a.stamps[target * 4 + 2] - a.col_offset;
	const int ra = r0 > 0 ? r0 : 0, rb = (r0 + a.height < a.frame_rows) ? (r0 + a.height) : a.frame_rows;
	const int ca = c0 > 0 ? c0 : 0, cb = (c0 + a.width < a.frame_cols) ? (c0 + a.width) : a.frame_cols;
	if (ra >= rb || ca >= cb) return false;
	ty0 = ra / kTileRows; ty1 = (rb - 1) / kTileRows; tx0 = ca / kTileCols; tx1 = (cb - 1) / kTileCols;
	return true;
}

// pass 1 (FILL = false): count the stamps per tile; pass 3 (FILL = true): write the stamp indices at the scanned offsets.
// Stamps that reach outside the frame get their cube filled with NaN first (count pass; the tiles only write what exists).
template <bool FILL>
__global__ __launch_bounds__(256) void tp_cut_bin_kernel(CutArgs a, TileGeom tg, int n_targets, int* __restrict__ count_or_cursor,
	const int* __restrict__ offsets, int* __restrict__ items, int* __restrict__ outside)
{
	const int t = blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= n_targets) return;
	int ty0, ty1, tx0, tx1;
	const bool any = stamp_tiles(a, t, ty0, ty1, tx0, tx1);
	if (!FILL) {
		const int r0 = a.stamps[t * 4 + 0] - a.row_offset, c0 = a.stamps[t * 4 + 2] - a.col_offset;
		outside[t] = (r0 < 0 || c0 < 0 || r0 + a.height > a.frame_rows || c0 + a.width > a.frame_cols) ? 1 : 0;
	}
	if (!any) return;
	for (int ty = ty0; ty <= ty1; ++ty)
		for (int tx = tx0; tx <= tx1; ++tx) {
			const int tile = ty * tg.tiles_x + tx;
			const int pos = atomicAdd(&count_or_cursor[tile], 1);
			if (FILL) items[offsets[tile] + pos] = t;
		}
}

// pass 2: exclusive scan of the per-tile counts (one workgroup; n + 1 outputs), cursors reset for the fill pass
__global__ __launch_bounds__(1024) void tp_cut_scan_kernel(int* __restrict__ count, int* __restrict__ offsets, int n)
{
	__shared__ int part[1024];
	const int tid = threadIdx.x;
	const int per = (n + 1023) / 1024;
	const int lo = tid * per, hi = (lo + per < n) ? (lo + per) : n;
	int s = 0;
	for (int i = lo; i < hi; ++i) s += count[i];
	part[tid] = s;
	__syncthreads();
	for (int d = 1; d < 1024; d <<= 1) {
		const int v = (tid >= d) ? part[tid - d] : 0;
		__syncthreads();
		part[tid] += v;
		__syncthreads();
	}
	int run = part[tid] - s;
	for (int i = lo; i < hi; ++i) { const int c = count[i]; offsets[i] = run; run += c; count[i] = 0; }
	if (tid == 1023) offsets[n] = part[1023];
}

// up to kMaxStacks frame stacks cut with ONE binning of the stamps (the three image groups of a CCD share stamps and geometry):
// blockIdx.z picks the stack

__global__ __launch_bounds__(256) void tp_cut_nanfill_kernel(CutArgs a, StackPtrs sp, const int* __restrict__ outside)
{
	const int target = blockIdx.x;
	if (!outside[target]) return;
	float* out = sp.cubes[blockIdx.z] + (int64_t)target * a.height * a.width * a.t_pitch;
	const int64_t n = (int64_t)a.height * a.width * a.t_pitch;
	const float nan = __builtin_nanf("");
	for (int64_t i = threadIdx.x; i < n; i += blockDim.x)
		out[i] = ((int)(i % a.t_pitch) < a.n_frames) ? nan : 0.f;
}

__global__ __launch_bounds__(256) void tp_cut_tiles_kernel(CutArgs a, StackPtrs sp, TileGeom tg, const int* __restrict__ offsets, const int* __restrict__ items)
{
	__shared__ float tile[kTileCad * kTileLd];
	a.frames = sp.frames[blockIdx.z];
	a.cube = sp.cubes[blockIdx.z];
	const int tile_id = blockIdx.x;
	const int first = offsets[tile_id], last = offsets[tile_id + 1];
	if (first == last) return;                       // no stamp touches this tile
	const int k0 = blockIdx.y * kTileCad;
	const int tid = threadIdx.x;
	const int ty = tile_id / tg.tiles_x, tx = tile_id - ty * tg.tiles_x;
	const int tr0 = ty * kTileRows, tc0 = tx * kTileCols;
	// ---- load: segment = (frame, tile row); a group of kTileCols lanes takes one row segment (256 bytes for 64 columns), all
	// loads of a thread in flight
	constexpr int GRP = 256 / kTileCols;             // row segments per pass of the workgroup
	int grp = tid / kTileCols;
	if (kTileCols % 64 == 0) grp = __builtin_amdgcn_readfirstlane(grp);   // wave-uniform: scalar address arithmetic
	const int lcol = tid % kTileCols;
	const int col = tc0 + lcol;
	const bool col_ok = col < a.frame_cols;
	constexpr int SEG = kTileCad * kTileRows;
	constexpr int U = SEG / GRP;
	{
		float v[U];
#pragma unroll
		for (int u = 0; u < U; ++u) {
			const int sidx = grp + GRP * u;
			const int kk = sidx / kTileRows, rr = sidx % kTileRows;
			const bool ok = col_ok && (k0 + kk < a.n_frames) && (tr0 + rr < a.frame_rows);
			const int64_t off = ok ? ((int64_t)(k0 + kk) * a.frame_stride + (int64_t)(tr0 + rr) * a.row_pitch + col) : 0;
			v[u] = a.frames[off];
		}
#pragma unroll
		for (int u = 0; u < U; ++u) {
			const int sidx = grp + GRP * u;
			const int kk = sidx / kTileRows, rr = sidx % kTileRows;
			tile[kk * kTileLd + rr * kTileCols + lcol] = v[u];
		}
	}
	__syncthreads();
	// ---- serve the stamps of this tile: kTileCad lanes store the frames of one stamp pixel (256 contiguous bytes for 64)
	constexpr int RUNS = 256 / kTileCad;             // pixel runs stored per pass of the workgroup
	const int lane = tid % kTileCad;
	int run = tid / kTileCad;
	if (kTileCad % 64 == 0) run = __builtin_amdgcn_readfirstlane(run);
	const bool k_ok = k0 + lane < a.n_frames;
	const bool k_store = k0 + lane < a.t_pitch;      // the padding of the time axis is written too (zeros): no memset of the cube
	const int P = a.height * a.width;
	for (int it = first; it < last; ++it) {
		const int target = __builtin_amdgcn_readfirstlane(items[it]);
		const int r0 = a.stamps[target * 4 + 0] - a.row_offset, c0 = a.stamps[target * 4 + 2] - a.col_offset;
		// overlap of the stamp with this tile (and the frame), in tile coordinates
		int ra = r0 - tr0, rb = r0 + a.height - tr0, ca = c0 - tc0, cb = c0 + a.width - tc0;
		ra = ra > 0 ? ra : 0; ca = ca > 0 ? ca : 0;
		const int rmax = (a.frame_rows - tr0 < kTileRows) ? (a.frame_rows - tr0) : kTileRows;
		const int cmax = (a.frame_cols - tc0 < kTileCols) ? (a.frame_cols - tc0) : kTileCols;
		rb = rb < rmax ? rb : rmax; cb = cb < cmax ? cb : cmax;
		float* out = a.cube + (int64_t)target * P * a.t_pitch + k0 + lane;
		for (int rr = ra; rr < rb; ++rr) {
			const int prow = (tr0 + rr - r0) * a.width + (tc0 - c0);
			const float* src = tile + lane * kTileLd + rr * kTileCols;
#pragma unroll 4
			for (int cc = ca + run; cc < cb; cc += RUNS) {
				const float x = src[cc];
				if (k_store) out[(int64_t)(prow + cc) * a.t_pitch] = k_ok ? x : 0.f;
			}
		}
	}
}

// ---- masked cut: only the in-mask pixels of every stamp are written.  The tile lists then hold PIXELS, not stamps (an item =
// target, pixel of its stamp): a mask is a sixth of a 15 x 15 stamp, and a serve loop that walks every stamp pixel to skip five
// of six is as slow as the unmasked cut (measured: 1.83 ms for two stacks of 2 500 stamps, against 1.88 without a mask).
// pass 1 / 3: one thread per (target, stamp pixel)
template <bool FILL>
__global__ __launch_bounds__(256) void tp_cut_bin_pixels_kernel(CutArgs a, StackPtrs sp, int n_stacks, TileGeom tg, int n_targets, int* __restrict__ count_or_cursor,
	const int* __restrict__ offsets, uint32_t* __restrict__ items)
{
	const int P = a.height * a.width;
	const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (e >= (int64_t)n_targets * P) return;
	if (!a.mask[e]) return;
	const int t = (int)(e / P), p = (int)(e - (int64_t)t * P);
	const int pi = p / a.width, pj = p - pi * a.width;
	const int r = a.stamps[t * 4 + 0] - a.row_offset + pi, c = a.stamps[t * 4 + 2] - a.col_offset + pj;
	if (r < 0 || r >= a.frame_rows || c < 0 || c >= a.frame_cols) {
		// an in-mask pixel beyond the frame (the reference never cuts such stamps: BasePhotometry.py:643-679): NaN, written here
		if (!FILL) {
			const float nan = __builtin_nanf("");
			for (int k = 0; k < n_stacks; ++k) {
				float* out = sp.cubes[k] + e * a.t_pitch;
				for (int q = 0; q < (int)a.t_pitch; ++q) out[q] = (q < a.n_frames) ? nan : 0.f;
			}
		}
		return;
	}
	const int tile = (r / kTileRows) * tg.tiles_x + (c / kTileCols);
	const int pos = atomicAdd(&count_or_cursor[tile], 1);
	if (FILL) { const int64_t o = 2 * ((int64_t)offsets[tile] + pos); items[o] = (uint32_t)t; items[o + 1] = (uint32_t)p; }
}

__global__ __launch_bounds__(256) void tp_cut_tiles_pixels_kernel(CutArgs a, StackPtrs sp, TileGeom tg, const int* __restrict__ offsets, const uint32_t* __restrict__ items)
{
	__shared__ float tile[kTileCad * kTileLd];
	a.frames = sp.frames[blockIdx.z];
	a.cube = sp.cubes[blockIdx.z];
	const int tile_id = blockIdx.x;
	const int first = offsets[tile_id], last = offsets[tile_id + 1];
	if (first == last) return;                       // no in-mask pixel in this tile
	const int k0 = blockIdx.y * kTileCad;
	const int tid = threadIdx.x;
	const int ty = tile_id / tg.tiles_x, tx = tile_id - ty * tg.tiles_x;
	const int tr0 = ty * kTileRows, tc0 = tx * kTileCols;
	constexpr int GRP = 256 / kTileCols;
	const int grp = __builtin_amdgcn_readfirstlane(tid / kTileCols);
	const int lcol = tid % kTileCols;
	const int col = tc0 + lcol;
	const bool col_ok = col < a.frame_cols;
	constexpr int SEG = kTileCad * kTileRows;
	constexpr int U = SEG / GRP;
	{
		float v[U];
#pragma unroll
		for (int u = 0; u < U; ++u) {
			const int sidx = grp + GRP * u;
			const int kk = sidx / kTileRows, rr = sidx % kTileRows;
			const bool ok = col_ok && (k0 + kk < a.n_frames) && (tr0 + rr < a.frame_rows);
			const int64_t off = ok ? ((int64_t)(k0 + kk) * a.frame_stride + (int64_t)(tr0 + rr) * a.row_pitch + col) : 0;
			v[u] = a.frames[off];
		}
#pragma unroll
		for (int u = 0; u < U; ++u) {
			const int sidx = grp + GRP * u;
			const int kk = sidx / kTileRows, rr = sidx % kTileRows;
			tile[kk * kTileLd + rr * kTileCols + lcol] = v[u];
		}
	}
	__syncthreads();
	// ---- one in-mask pixel per group of kTileCad lanes and step: its frames of this block are one 128-byte line of the cube
	constexpr int RUNS = 256 / kTileCad;
	const int lane = tid % kTileCad, run = tid / kTileCad;
	const bool k_ok = k0 + lane < a.n_frames, k_store = k0 + lane < a.t_pitch;
	const int P = a.height * a.width;
	for (int it = first + run; it < last; it += RUNS) {
		const int target = (int)items[2 * (int64_t)it], p = (int)items[2 * (int64_t)it + 1];
		const int pi = p / a.width, pj = p - pi * a.width;
		const int rr = a.stamps[target * 4 + 0] - a.row_offset + pi - tr0, cc = a.stamps[target * 4 + 2] - a.col_offset + pj - tc0;
		const float x = tile[lane * kTileLd + rr * kTileCols + cc];
		if (k_store) a.cube[((int64_t)target * P + p) * a.t_pitch + k0 + lane] = k_ok ? x : 0.f;
	}
}

// BasePhotometry.py:1001-1006: the sum image of a stamp is a crop of the frame's
__global__ __launch_bounds__(256) void tp_crop_sumimage_kernel(const double* __restrict__ full, int frame_rows, int frame_cols, int64_t row_pitch,
	int row_offset, int col_offset, const int32_t* __restrict__ stamps, int64_t n_items, int height, int width, double* __restrict__ out)
{
	const int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (it >= n_items) return;
	const int P = height * width;
	const int target = (int)(it / P), p = (int)(it - (int64_t)target * P);
	const int i = p / width, j = p - i * width;
	const int r = stamps[target * 4 + 0] - row_offset + i, c = stamps[target * 4 + 2] - col_offset + j;
	const bool inside = r >= 0 && r < frame_rows && c >= 0 && c < frame_cols;
	out[it] = inside ? full[(int64_t)r * row_pitch + c] : __builtin_nan("");
}

} // namespace

static int cut_stamps_launch(tp_ctx* ctx, int n_stacks, const float* const* d_frames, int32_t n_frames, int32_t frame_rows, int32_t frame_cols,
	int64_t row_pitch, int64_t frame_stride, int32_t row_offset, int32_t col_offset,
	const int32_t* d_stamps, const tp_cube_desc* desc, float* const* d_cubes, const uint8_t* d_mask = nullptr)
{
	TP_REQUIRE(ctx, tp_desc_ok(desc), "tp_cut_stamps: bad cube descriptor");
	TP_REQUIRE(ctx, n_stacks >= 1 && n_stacks <= kMaxStacks && d_frames && d_stamps && d_cubes, "tp_cut_stamps: null pointer / 1 .. 4 stacks");
	for (int k = 0; k < n_stacks; ++k) TP_REQUIRE(ctx, d_frames[k] && d_cubes[k], "tp_cut_stamps: null pointer");
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
	TP_REQUIRE(ctx, n_bands <= 65535 && (desc->t_pitch + kCadBlock - 1) / kCadBlock <= 65535, "tp_cut_stamps: too many bands / cadence blocks");
	CutArgs a;
	a.frames = d_frames[0]; a.n_frames = n_frames; a.frame_rows = frame_rows; a.frame_cols = frame_cols;
	a.row_pitch = row_pitch; a.frame_stride = frame_stride; a.row_offset = row_offset; a.col_offset = col_offset;
	a.stamps = d_stamps; a.height = desc->height; a.width = desc->width; a.t_pitch = desc->t_pitch; a.cube = d_cubes[0];
	a.mask = d_mask;
	StackPtrs sp{};
	for (int k = 0; k < n_stacks; ++k) { sp.frames[k] = d_frames[k]; sp.cubes[k] = d_cubes[k]; }
	// dense batch (the stamps cover at least an eighth of the frame): frame-tile-major, every frame pixel fetched once
	const int64_t stamp_pixels = (int64_t)desc->n_targets * desc->height * desc->width;
	TileGeom tg;
	tg.tiles_x = (frame_cols + kTileCols - 1) / kTileCols;
	tg.tiles_y = (frame_rows + kTileRows - 1) / kTileRows;
	const int64_t n_tiles = (int64_t)tg.tiles_x * tg.tiles_y;
	const int64_t max_per_stamp = (int64_t)((desc->height - 1) / kTileRows + 2) * ((desc->width - 1) / kTileCols + 2);
	if (d_mask && stamp_pixels * 8 >= (int64_t)frame_rows * frame_cols && n_tiles <= 16 * 1024 * 1024 && stamp_pixels < 1073741823ll) {
		// masked: the tiles serve lists of in-mask pixels (a tile without any exits at once: a sparse batch reads what it needs)
		const size_t n_int = (size_t)(2 * n_tiles + 2 + 2 * stamp_pixels);
		int* base = static_cast<int*>(tp_ctx_scratch(ctx, n_int * sizeof(int)));
		TP_REQUIRE(ctx, base != nullptr, "tp_cut_stamps_masked: out of device memory (pixel lists)");
		int* count = base;
		int* offsets = count + n_tiles;
		uint32_t* items = reinterpret_cast<uint32_t*>(offsets + n_tiles + 2);
		TP_HIP(ctx, hipMemsetAsync(count, 0, (size_t)n_tiles * sizeof(int), ctx->stream));
		const dim3 bgrid((unsigned)((stamp_pixels + 255) / 256));
		{
			tp_prof_scope _ps(ctx, TPK_CUTOUT);
			hipLaunchKernelGGL(tp_cut_bin_pixels_kernel<false>, bgrid, dim3(256), 0, ctx->stream, a, sp, (int)n_stacks, tg, (int)desc->n_targets, count, (const int*)offsets, items);
			hipLaunchKernelGGL(tp_cut_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, count, offsets, (int)n_tiles);
			hipLaunchKernelGGL(tp_cut_bin_pixels_kernel<true>, bgrid, dim3(256), 0, ctx->stream, a, sp, (int)n_stacks, tg, (int)desc->n_targets, count, (const int*)offsets, items);
			dim3 grid((unsigned)n_tiles, (unsigned)((desc->t_pitch + kTileCad - 1) / kTileCad), (unsigned)n_stacks);
			hipLaunchKernelGGL(tp_cut_tiles_pixels_kernel, grid, dim3(256), 0, ctx->stream, a, sp, tg, (const int*)offsets, (const uint32_t*)items);
		}
		TP_LAUNCH_CHECK(ctx, "tp_cut_tiles_pixels_kernel");
		return TP_OK;
	}
	if (stamp_pixels * 8 >= (int64_t)frame_rows * frame_cols && n_tiles <= 16 * 1024 * 1024 && desc->n_targets * max_per_stamp < 2147483647ll) {
		const size_t n_int = (size_t)(2 * n_tiles + 2 + desc->n_targets * max_per_stamp + desc->n_targets);
		int* base = static_cast<int*>(tp_ctx_scratch(ctx, n_int * sizeof(int)));
		TP_REQUIRE(ctx, base != nullptr, "tp_cut_stamps: out of device memory (tile lists)");
		int* count = base;                       // [n_tiles]      counts, then cursors
		int* offsets = count + n_tiles;          // [n_tiles + 1]
		int* outside = offsets + n_tiles + 1;    // [n_targets]
		int* items = outside + desc->n_targets;  // [sum of counts]
		TP_HIP(ctx, hipMemsetAsync(count, 0, (size_t)n_tiles * sizeof(int), ctx->stream));
		const dim3 bgrid((unsigned)((desc->n_targets + 255) / 256));
		{
			tp_prof_scope _ps(ctx, TPK_CUTOUT); // the whole path is one profile entry
			hipLaunchKernelGGL(tp_cut_bin_kernel<false>, bgrid, dim3(256), 0, ctx->stream, a, tg, (int)desc->n_targets, count, (const int*)offsets, items, outside);
			hipLaunchKernelGGL(tp_cut_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, count, offsets, (int)n_tiles);
			hipLaunchKernelGGL(tp_cut_bin_kernel<true>, bgrid, dim3(256), 0, ctx->stream, a, tg, (int)desc->n_targets, count, (const int*)offsets, items, outside);
			hipLaunchKernelGGL(tp_cut_nanfill_kernel, dim3((unsigned)desc->n_targets, 1, (unsigned)n_stacks), dim3(256), 0, ctx->stream, a, sp, (const int*)outside);
			dim3 grid((unsigned)n_tiles, (unsigned)((desc->t_pitch + kTileCad - 1) / kTileCad), (unsigned)n_stacks);   // (blocks cover the padding of the time axis too)
			hipLaunchKernelGGL(tp_cut_tiles_kernel, grid, dim3(256), 0, ctx->stream, a, sp, tg, (const int*)offsets, (const int*)items);
		}
		TP_LAUNCH_CHECK(ctx, "tp_cut_tiles_kernel");
		return TP_OK;
	}
	if (shmem > 64 * 1024)
		TP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(tp_cut_stamps_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
	const int n_cad_blocks = (int)((desc->t_pitch + kCadBlock - 1) / kCadBlock);
	TP_REQUIRE(ctx, (int64_t)n_cad_blocks * n_stacks <= 65535, "tp_cut_stamps: too many cadence blocks");
	dim3 grid((unsigned)desc->n_targets, (unsigned)(n_cad_blocks * n_stacks), (unsigned)n_bands);
	TP_LAUNCH(ctx, TPK_CUTOUT, tp_cut_stamps_kernel, grid, dim3(256), shmem, a, sp, n_cad_blocks, band_rows);
	TP_LAUNCH_CHECK(ctx, "tp_cut_stamps_kernel");
	return TP_OK;
}

extern "C" int tp_cut_stamps(tp_ctx* ctx, const float* d_frames, int32_t n_frames, int32_t frame_rows, int32_t frame_cols,
	int64_t row_pitch, int64_t frame_stride, int32_t row_offset, int32_t col_offset,
	const int32_t* d_stamps, const tp_cube_desc* desc, float* d_cube)
{
	TP_CHECK_CTX(ctx);
	TP_API_BEGIN
	return cut_stamps_launch(ctx, 1, &d_frames, n_frames, frame_rows, frame_cols, row_pitch, frame_stride, row_offset, col_offset, d_stamps, desc, &d_cube);
	TP_API_END(ctx)
}

extern "C" int tp_cut_stamps_multi(tp_ctx* ctx, int32_t n_stacks, const float* const* d_frames, int32_t n_frames, int32_t frame_rows, int32_t frame_cols,
	int64_t row_pitch, int64_t frame_stride, int32_t row_offset, int32_t col_offset,
	const int32_t* d_stamps, const tp_cube_desc* desc, float* const* d_cubes)
{
	TP_CHECK_CTX(ctx);
	TP_API_BEGIN
	return cut_stamps_launch(ctx, n_stacks, d_frames, n_frames, frame_rows, frame_cols, row_pitch, frame_stride, row_offset, col_offset, d_stamps, desc, d_cubes);
	TP_API_END(ctx)
}

// ---- the time-major copy of a frame stack ----------------------------------------------------------------------------------
// [n_frames][n_pixels] -> [n_pixels][t_pitch] (the cadences past n_frames zero): tiles of 64 pixels x 64 frames through LDS, 256-byte
// runs on both sides.  One pass per stack and region; every batch of targets on the region then reads a mask pixel's time series
// as one contiguous row (tp_aperture_extract_stack) where it used to be cut out of 1 300 frames per call.
__global__ __launch_bounds__(256) void tp_frames_transpose_kernel(const float* __restrict__ src, int n_frames, int64_t n_pixels, int64_t frame_stride,
	float* __restrict__ dst, int64_t t_pitch)
{
	__shared__ float tile[64][65];
	const int64_t p0 = (int64_t)blockIdx.x * 64;
	const int f0 = blockIdx.y * 64;
	const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
	for (int j = w; j < 64; j += 4) {
		const int f = f0 + j;
		const int64_t p = p0 + lane;
		tile[j][lane] = (f < n_frames && p < n_pixels) ? src[(int64_t)f * frame_stride + p] : 0.f;
	}
	__syncthreads();
	for (int j = w; j < 64; j += 4) {
		const int64_t p = p0 + j;
		const int f = f0 + lane;
		if (p < n_pixels && f < t_pitch) dst[p * t_pitch + f] = tile[lane][j];
	}
}

extern "C" int tp_frames_transpose(tp_ctx* ctx, const float* d_frames, int32_t n_frames, int64_t n_pixels, int64_t frame_stride, float* d_out, int64_t t_pitch)
{
	TP_CHECK_CTX(ctx);
	TP_API_BEGIN
	TP_REQUIRE(ctx, d_frames && d_out, "tp_frames_transpose: null pointer");
	TP_REQUIRE(ctx, n_frames >= 0 && n_pixels >= 0 && frame_stride >= n_pixels && t_pitch >= n_frames, "tp_frames_transpose: bad sizes");
	if (n_pixels == 0 || t_pitch == 0) return TP_OK;
	const int64_t gx = (n_pixels + 63) / 64, gy = (t_pitch + 63) / 64;
	TP_REQUIRE(ctx, gx < ((int64_t)1 << 31) && gy <= 65535, "tp_frames_transpose: stack too large for one launch");
	TP_LAUNCH(ctx, TPK_CUTOUT, tp_frames_transpose_kernel, dim3((unsigned)gx, (unsigned)gy), dim3(256), 0, d_frames, (int)n_frames, n_pixels, frame_stride, d_out, t_pitch);
	TP_LAUNCH_CHECK(ctx, "tp_frames_transpose_kernel");
	return TP_OK;
	TP_API_END(ctx)
}

extern "C" int tp_crop_sumimage(tp_ctx* ctx, const double* d_full, int32_t frame_rows, int32_t frame_cols, int64_t row_pitch,
	int32_t row_offset, int32_t col_offset, const int32_t* d_stamps, int32_t n_targets, int32_t height, int32_t width, double* d_out)
{
	TP_CHECK_CTX(ctx);
	TP_API_BEGIN
	TP_REQUIRE(ctx, d_full && d_stamps && d_out, "tp_crop_sumimage: null pointer");
	TP_REQUIRE(ctx, frame_rows > 0 && frame_cols > 0 && row_pitch >= frame_cols && n_targets >= 0 && height > 0 && width > 0, "tp_crop_sumimage: bad geometry");
	const int64_t n_items = (int64_t)n_targets * height * width;
	if (n_items == 0) return TP_OK;
	TP_LAUNCH(ctx, TPK_SUMIMAGE, tp_crop_sumimage_kernel, dim3((unsigned)((n_items + 255) / 256)), dim3(256), 0, d_full, (int)frame_rows, (int)frame_cols, row_pitch,
		(int)row_offset, (int)col_offset, d_stamps, n_items, (int)height, (int)width, d_out);
	TP_LAUNCH_CHECK(ctx, "tp_crop_sumimage_kernel");
	return TP_OK;
	TP_API_END(ctx)
}

extern "C" int tp_cut_stamps_masked(tp_ctx* ctx, int32_t n_stacks, const float* const* d_frames, int32_t n_frames, int32_t frame_rows, int32_t frame_cols,
	int64_t row_pitch, int64_t frame_stride, int32_t row_offset, int32_t col_offset,
	const int32_t* d_stamps, const tp_cube_desc* desc, const uint8_t* d_mask, float* const* d_cubes)
{
	TP_CHECK_CTX(ctx);
	TP_API_BEGIN
	TP_REQUIRE(ctx, d_mask != nullptr, "tp_cut_stamps_masked: null mask");
	return cut_stamps_launch(ctx, n_stacks, d_frames, n_frames, frame_rows, frame_cols, row_pitch, frame_stride, row_offset, col_offset, d_stamps, desc, d_cubes, d_mask);
	TP_API_END(ctx)
}
