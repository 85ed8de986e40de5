/*
 * tessphot_hip.h -- C ABI of libtessphot_hip.so: the MI355X (gfx950) native per-target
 * photometry engine for the tasoc/photometry hot path.
 *
 * The reference is pure Python and has no FFI of its own; each entry point below names the
 * reference code (file:line, relative to the reference repository root) whose work it
 * replaces.  INTEGRATION.md shows the ctypes binding a maintainer adds on the reference side.
 *
 * Conventions
 * -----------
 *  - every function returns int: 0 = TP_OK, < 0 = error code; tp_last_error(ctx) returns a
 *    ctx-owned, NUL-terminated description of the last failure on that ctx.
 *  - bulk data pointers named d_* are DEVICE pointers (HBM) obtained from tp_malloc (or any
 *    hipMalloc in the same process); pointers named h_* are host pointers.
 *  - the caller allocates and owns every buffer; the library never frees caller memory and
 *    keeps no pointer past the call.
 *  - all work is enqueued on the ctx's own HIP stream, in call order.  Calls return when
 *    the work is enqueued; tp_sync() (or a d2h copy) waits for it.
 *  - per-target failure is reported through int32 status arrays holding the reference's
 *    STATUS integers (photometry/BasePhotometry.py:48-59): 0 UNKNOWN, 1 OK, 2 ERROR,
 *    3 WARNING, 4 ABORT, 5 SKIPPED, 6 STARTED.  No C++ exception crosses the ABI.
 *  - one ctx per host thread; calls on one ctx are not thread-safe.
 *
 * Cube layout
 * -----------
 *  A stamp cube is float32 [n_targets][height][width][t_pitch]: per target exactly the
 *  reference's (rows, cols, times) C-order cube with TIME as the fastest axis
 *  (photometry/BasePhotometry.py:732), t_pitch >= n_cad elements between the time series
 *  of consecutive pixels.  t_pitch % 4 == 0 and a 16-byte aligned base select the
 *  128-bit load path; anything else still works on the scalar path.
 */
#ifndef TESSPHOT_HIP_H
#define TESSPHOT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TP_OK               0
#define TP_ERR_INVALID     -1   /* bad argument */
#define TP_ERR_HIP         -2   /* HIP runtime error (text in tp_last_error) */
#define TP_ERR_NOMEM       -3
#define TP_ERR_COMM        -4   /* RCCL error */
#define TP_ERR_UNSUPPORTED -5

/* STATUS integers, photometry/BasePhotometry.py:48-59 */
#define TP_STATUS_UNKNOWN 0
#define TP_STATUS_OK      1
#define TP_STATUS_ERROR   2
#define TP_STATUS_WARNING 3
#define TP_STATUS_ABORT   4
#define TP_STATUS_SKIPPED 5
#define TP_STATUS_STARTED 6

typedef struct tp_ctx tp_ctx;

typedef struct tp_cube_desc {
	int32_t n_targets;
	int32_t n_cad;      /* T: number of cadences */
	int32_t height;     /* H: stamp rows */
	int32_t width;      /* W: stamp columns */
	int64_t t_pitch;    /* elements between consecutive pixels' time series (>= n_cad) */
} tp_cube_desc;

/* ---- context, memory, timing ------------------------------------------------------------ */
int tp_version(void);
int tp_device_count(int* n);
int tp_ctx_create(int device, tp_ctx** out);
/* An additional context (= one more HIP stream) on the same device; high_priority != 0 asks for the
 * device's greatest stream priority (used for the latency-bound K2P2 stage of the chunked pipeline). */
int tp_ctx_create_stream(int device, int high_priority, tp_ctx** out);
int tp_ctx_destroy(tp_ctx* ctx);
const char* tp_last_error(tp_ctx* ctx);      /* ctx may be NULL: last tp_ctx_create failure */
int tp_device_info(tp_ctx* ctx, char* name, int name_len, int32_t* n_cu, uint64_t* hbm_bytes);
/* NUMA node of the device's PCIe slot (-1: unknown), for binding the host threads that feed it (device.bind_host_to_device). */
int tp_device_numa_node(int device, int* node);

/* Device memory.  tp_free keeps a block for the next tp_malloc of its size class (blocks up to 32 GiB, 160 GiB per context;
 * hipMalloc / hipFree of multi-GB blocks cost milliseconds and synchronise the device); a recycled block is handed out only
 * after everything that was queued on the context's stream when it was freed has run, so it is idle for whichever stream writes
 * it next.  tp_cache_trim gives the cached blocks back to the driver; when an allocation fails the library does so itself, for
 * this context first and then for every other context of the device. */
int tp_malloc(tp_ctx* ctx, uint64_t nbytes, void** d_ptr);
int tp_free(tp_ctx* ctx, void* d_ptr);
int tp_cache_trim(tp_ctx* ctx);
int tp_memset(tp_ctx* ctx, void* d_ptr, int value, uint64_t nbytes);
int tp_memcpy_h2d(tp_ctx* ctx, void* d_dst, const void* h_src, uint64_t nbytes);
int tp_memcpy_d2h(tp_ctx* ctx, void* h_dst, const void* d_src, uint64_t nbytes);   /* waits */
int tp_memcpy_d2d(tp_ctx* ctx, void* d_dst, const void* d_src, uint64_t nbytes);
/* Copy n_rows time series of n_cad float32 between pitched layouts (host -> device): the host
 * cube of BasePhotometry._load_cube (BasePhotometry.py:720-751) has pitch n_cad. */
int tp_upload_cube(tp_ctx* ctx, float* d_dst, int64_t dst_pitch, const float* h_src, int64_t src_pitch,
	int64_t n_rows, int64_t n_cad);
/* Pinned (page-locked) host memory and copies that return as soon as they are enqueued on the ctx stream: with a
 * second context (= stream) for the transfers and tp_event_* ordering, the upload of the next chunk of stamp cubes
 * overlaps the photometry of the current one -- the regime of a plugin that receives host cubes from
 * BasePhotometry._load_cube (BasePhotometry.py:720-751).  h_src / h_dst must come from tp_host_alloc. */
int tp_host_alloc(tp_ctx* ctx, uint64_t nbytes, void** h_ptr);
int tp_host_free(tp_ctx* ctx, void* h_ptr);
int tp_upload_cube_async(tp_ctx* ctx, float* d_dst, int64_t dst_pitch, const float* h_src, int64_t src_pitch,
	int64_t n_rows, int64_t n_cad);
int tp_memcpy_d2h_async(tp_ctx* ctx, void* h_dst, const void* d_src, uint64_t nbytes);
int tp_sync(tp_ctx* ctx);

/* Cross-stream ordering: several contexts on one device each own a HIP stream; an event recorded on one
 * context's stream can be waited for by another's (hipStreamWaitEvent), so independent stages of the
 * pipeline (memory-bound A1 / A6, latency-bound K2P2) of different target chunks overlap on the GPU. */
int tp_event_create(tp_ctx* ctx, void** event);
int tp_event_destroy(tp_ctx* ctx, void* event);
int tp_event_record(tp_ctx* ctx, void* event);           /* on ctx's stream */
int tp_stream_wait_event(tp_ctx* ctx, void* event);      /* ctx's stream waits for the event */
int tp_event_sync(tp_ctx* ctx, void* event);             /* the HOST waits for the event (reuse of a pinned staging buffer) */

/* HIP-event stopwatch on the ctx stream; slot in [0, 16). */
int tp_timer_start(tp_ctx* ctx, int slot);
int tp_timer_stop(tp_ctx* ctx, int slot);
int tp_timer_elapsed_ms(tp_ctx* ctx, int slot, float* ms);   /* waits for the stop event */

/* Per-kernel profile: when enabled every kernel launch is bracketed by HIP events on the
 * ctx stream.  kernel ids are dense in [0, tp_kernel_count()). */
int tp_profile_enable(tp_ctx* ctx, int on);
int tp_profile_reset(tp_ctx* ctx);
int tp_kernel_count(void);
const char* tp_kernel_name(int kernel_id);
int tp_profile_get(tp_ctx* ctx, int kernel_id, int64_t* n_launches, double* total_ms);   /* waits */

/* ---- A1: sum image ------------------------------------------------------------------------
 * replaces BasePhotometry.sumimage (photometry/BasePhotometry.py:1008-1019) and the FFI
 * accumulation of prepare.py:450-453,459: per-pixel mean over the cadences with
 * (quality & bitmask) == 0 (TESSQualityFlags.DEFAULT_BITMASK = 4335, quality.py:123-124),
 * non-finite pixels excluded from sum and count, zero count -> NaN.  float64 accumulation.
 *   d_quality: int32 [n_cad] shared by all targets (quality_target_stride = 0) or
 *              [n_targets][quality_target_stride].
 *   d_subtract: optional float32 [n_targets][subtract_pitch]: a stamp-constant background series
 *              subtracted on the fly (images := raw - background, prepare.py:419-420), or NULL.
 *   d_sumimage: float64 [n_targets][height*width].                                          */
int tp_sumimage(tp_ctx* ctx, const tp_cube_desc* desc, const float* d_images,
	const int32_t* d_quality, int64_t quality_target_stride, uint32_t bitmask,
	const float* d_subtract, int64_t subtract_pitch, double* d_sumimage);

/* ---- A2..A5b + A7: K2P2 aperture masks -------------------------------------------------------
 * replaces k2p2.k2p2FixFromSum (photometry/AperturePhotometry/k2p2v2.py:344-623: KDE-mode/MAD
 * threshold, DBSCAN clustering, watershed segmentation, size filter, hole filling, overflow
 * columns) and the mask selection, minimum aperture, edge test and contamination of
 * AperturePhotometry.do_photometry (photometry/AperturePhotometry/photometry.py:31-41, 93-131,
 * 220-254), for every target of the batch.  Stamps are fixed-size cubes: resize_stamp() cannot
 * grow them (BasePhotometry.py:605-612 -> False), so edge-touching masks are used as they are
 * and reported through the edge bits of d_flags for a host-side retry on a bigger cut-out.
 * Stamps of any size up to 32 767 pixels (labels and pixel indices are signed 16-bit): up to about 54 x 54 the work arrays are in LDS, beyond that in a context-owned
 * HBM scratch (same code, same results, slow: the few bright stars of a CCD).
 *   catalog (ragged, CSR): d_cat_offsets int64 [n_targets+1]; per star float32 column_stamp,
 *     row_stamp, tmag, column, row (BasePhotometry.catalog, BasePhotometry.py:1153-1178), int64 starid.
 *   d_target_pos_row/column: float64 CCD position (target_pos_row/column); d_target_tmag float64.
 *   d_aperture: int32 [n_targets][H*W] pixel flags (BasePhotometry.aperture; bit 1 = collected).
 *   d_cut_override: optional float64 [n_targets] replacing the KDE/Powell threshold CUT.
 *   params: NULL = the plugin's settings (photometry.py:54-64).
 * outputs: d_mask uint8 [n_targets][H*W] (final_phot_mask); d_status int32 STATUS;
 *   d_flags int32: bit0 minimum aperture used, bits1-4 mask touches stamp edge (row 0, last row,
 *   column 0, last column), bit5 K2P2NoStars, bit6 no mask passed the size filter,
 *   bits 8+ error kind (1 no flux, 2 zero KDE bandwidth, 3 no watershed peak, 4 target outside
 *   stamp, 5 too many masks, 6 no catalog star in mask);
 *   d_contamination float64 (AP_CONT, NaN if undefined); d_diag optional float64 [n_targets][8] =
 *   (CUT, MODE, MAD1, KDE bandwidth, FFT-grid mode guess, n positive pixels, min|S-CUT|, n masks);
 *   d_cat_in_mask optional uint8 [n_catalog]: star falls inside the final mask (skip_targets).   */
typedef struct tp_k2p2_params {
	double thresh;                  /* 0.8 */
	int32_t min_no_pixels_in_mask;  /* 4 */
	int32_t min_for_cluster;        /* 4 */
	int32_t extend_overflow;        /* 1 */
	int32_t reserved;
	double ws_thres;                /* 0 */
	double saturation_limit;        /* 7.0 */
} tp_k2p2_params;
int tp_k2p2_masks(tp_ctx* ctx, int32_t n_targets, int32_t height, int32_t width,
	const double* d_sumimage,
	const int64_t* d_cat_offsets, const float* d_cat_column_stamp, const float* d_cat_row_stamp, const float* d_cat_tmag,
	const float* d_cat_column, const float* d_cat_row, const int64_t* d_cat_starid,
	const double* d_target_pos_row, const double* d_target_pos_column, const double* d_target_tmag, const int64_t* d_target_starid,
	const int32_t* d_stamps, const int32_t* d_aperture, const double* d_cut_override, const tp_k2p2_params* params,
	uint8_t* d_mask, int32_t* d_status, int32_t* d_flags, double* d_contamination, double* d_diag, uint8_t* d_cat_in_mask);

/* ---- A6: aperture extraction ---------------------------------------------------------------
 * replaces the per-cadence loop of AperturePhotometry.do_photometry
 * (photometry/AperturePhotometry/photometry.py:172-201): in-mask flux (float32 np.sum order,
 * bit-exact), flux error sqrt(sum err^2), flux-weighted centroid over pixels with flux > 0 in
 * 1-based CCD (column, row) coordinates, np.nansum of the background (NaN -> 0, float32 np.sum order,
 * bit-exact); NaN rules of :182-201.
 *   d_backgrounds: bkg_mode 0: cube with the layout of desc;
 *                  bkg_mode 1: one series per target, float32 [n_targets][bkg_series_pitch]
 *                  (a stamp-constant background: every pixel of a cadence has the same value);
 *                  NULL: aperture-only (no background input): d_flux_background may be NULL too, and is
 *                  filled with NaN otherwise.
 *   d_subtract: optional float32 [n_targets][subtract_pitch] subtracted from d_images on the fly
 *                  (d_images then holds the raw, not background-subtracted, flux), or NULL.
 *   d_mask:   uint8 [n_targets][height*width], non-zero = in aperture (final_phot_mask).
 *   d_stamps: int32 [n_targets][4] = (row_min, row_max, col_min, col_max), BasePhotometry.stamp.
 *   d_status: optional int32 [n_targets]; targets whose status is TP_STATUS_ERROR are skipped
 *             (outputs untouched), like the early returns of photometry.py:116,163,170.
 *   outputs:  float64 [n_targets][out_pitch] each (lightcurve columns, BasePhotometry.py:425-428);
 *             d_centroid_col / d_centroid_row are pos_centroid[:,0] / [:,1].                   */
int tp_aperture_extract(tp_ctx* ctx, const tp_cube_desc* desc,
	const float* d_images, const float* d_images_err, const float* d_backgrounds,
	int32_t bkg_mode, int64_t bkg_series_pitch, const float* d_subtract, int64_t subtract_pitch,
	const uint8_t* d_mask, const int32_t* d_stamps, const int32_t* d_status,
	double* d_flux, double* d_flux_err, double* d_flux_background,
	double* d_centroid_col, double* d_centroid_row, int64_t out_pitch);

/* ---- A1 + A2..A5b + A7 + A6 fused: AperturePhotometry.do_photometry for a batch -------------------
 * replaces one pass of photometry/AperturePhotometry/photometry.py:75-257 over a fixed stamp for every
 * target: sum image, K2P2 mask (+ minimum aperture, contamination), extraction.  One wavefront owns a
 * target from the first load to the last store, the sum image and the mask stay in LDS in between; the
 * outputs are bit-identical to tp_sumimage + tp_k2p2_masks + tp_aperture_extract called in turn
 * (argument meaning as documented there; d_sumimage is an OUTPUT here).                              */
int tp_aperture_photometry(tp_ctx* ctx, const tp_cube_desc* desc,
	const float* d_images, const float* d_images_err, const float* d_backgrounds, int32_t bkg_mode, int64_t bkg_series_pitch,
	const float* d_subtract, int64_t subtract_pitch,
	const int32_t* d_quality, int64_t quality_target_stride, uint32_t bitmask,
	const int64_t* d_cat_offsets, const float* d_cat_column_stamp, const float* d_cat_row_stamp, const float* d_cat_tmag,
	const float* d_cat_column, const float* d_cat_row, const int64_t* d_cat_starid,
	const double* d_target_pos_row, const double* d_target_pos_column, const double* d_target_tmag, const int64_t* d_target_starid,
	const int32_t* d_stamps, const int32_t* d_aperture, const tp_k2p2_params* params,
	double* d_sumimage, uint8_t* d_mask, int32_t* d_status, int32_t* d_flags, double* d_contamination, double* d_diag,
	uint8_t* d_cat_in_mask,
	double* d_flux, double* d_flux_err, double* d_flux_background, double* d_centroid_col, double* d_centroid_row, int64_t out_pitch);
/* The same from a sum image the caller already holds (d_sumimage is an INPUT: float64 [n_targets][height*width], e.g. from
 * tp_background_sumimage, which forms it while it streams the raw cube for the background): the kernel starts at the K2P2 mask
 * (k2p2v2.py:388-623) and reads only the in-mask pixel rows of the cubes.  d_quality / bitmask are not used (the sum image
 * carries them) and may be NULL / 0.  Everything else as above; outputs bit-identical to tp_k2p2_masks + tp_aperture_extract. */
int tp_aperture_photometry_from_sumimage(tp_ctx* ctx, const tp_cube_desc* desc,
	const float* d_images, const float* d_images_err, const float* d_backgrounds, int32_t bkg_mode, int64_t bkg_series_pitch,
	const float* d_subtract, int64_t subtract_pitch,
	const int32_t* d_quality, int64_t quality_target_stride, uint32_t bitmask,
	const int64_t* d_cat_offsets, const float* d_cat_column_stamp, const float* d_cat_row_stamp, const float* d_cat_tmag,
	const float* d_cat_column, const float* d_cat_row, const int64_t* d_cat_starid,
	const double* d_target_pos_row, const double* d_target_pos_column, const double* d_target_tmag, const int64_t* d_target_starid,
	const int32_t* d_stamps, const int32_t* d_aperture, const tp_k2p2_params* params,
	const double* d_sumimage, uint8_t* d_mask, int32_t* d_status, int32_t* d_flags, double* d_contamination, double* d_diag,
	uint8_t* d_cat_in_mask,
	double* d_flux, double* d_flux_err, double* d_flux_background, double* d_centroid_col, double* d_centroid_row, int64_t out_pitch);

/* ---- B*, B2, B3: background on stamps -----------------------------------------------------------
 * tp_background_stamp (B*): build-defined stamp analogue of backgrounds.fit_background
 *   (photometry/backgrounds.py:52-211, which needs 64x64 tiles of a full frame): per (target, cadence)
 *   mask non-finite / > flux_cutoff / < 0 pixels (backgrounds.py:89-94), SigmaClip(3 sigma, 5 iterations,
 *   median / std), SExtractor mode estimate; more than exclude_percentile % masked pixels -> NaN.
 *   d_raw: cube of raw (not background-subtracted) flux; d_bkg: float32 [n_targets][bkg_pitch].
 * tp_smooth_time (B2): prepare.py:317-335, out[k] = nanmean(in[max(k-w,0) : min(k+w+1,T)]),
 *   w = time_smooth/2 (time_smooth = 3 at 1800 s cadence, 9 at 600 s; prepare.py:258), float32.
 * tp_subtract_background (B3): prepare.py:419-425, images = raw - bkg[k] (float32); pixels whose
 *   flag & flag_mask != 0 (PixelQualityFlags.ManualExclude = 2) become NaN in image and error.
 *   d_pixel_flags: optional uint8 [n_targets][H*W][n_cad]; d_raw_err / d_images_err optional;
 *   in-place (d_images == d_raw) is allowed; bkg_pitch >= t_pitch.                               */
int tp_background_stamp(tp_ctx* ctx, const tp_cube_desc* desc, const float* d_raw,
	double flux_cutoff, double exclude_percentile, float* d_bkg, int64_t bkg_pitch);
int tp_smooth_time(tp_ctx* ctx, int32_t n_targets, int32_t n_cad, int64_t pitch, int32_t time_smooth,
	const float* d_in, float* d_out);
int tp_subtract_background(tp_ctx* ctx, const tp_cube_desc* desc, const float* d_raw, const float* d_raw_err,
	const float* d_bkg, int64_t bkg_pitch, const uint8_t* d_pixel_flags, uint32_t flag_mask,
	float* d_images, float* d_images_err);
/* B* + B2 + A1 in ONE pass over the raw cube: d_bkg_raw = tp_background_stamp(d_raw), d_bkg = tp_smooth_time(d_bkg_raw)
 * (prepare.py:258, 317-335), d_sumimage = tp_sumimage(d_raw, d_subtract = d_bkg), i.e. the good-cadence mean of
 * float32(raw - smoothed background) per pixel (prepare.py:419-421, 450-459; BasePhotometry.py:1008-1019).  Both series
 * float32 [n_targets][bkg_pitch] and bit-identical to the two stand-alone entries; the sum image float64
 * [n_targets][height*width], equal to tp_sumimage's to rounding (another order of the float64 additions).  Stamps up to
 * 256 pixels and time_smooth <= 17 read the cube once; anything else runs the three entries in turn.                    */
int tp_background_sumimage(tp_ctx* ctx, const tp_cube_desc* desc, const float* d_raw,
	double flux_cutoff, double exclude_percentile, int32_t time_smooth,
	const int32_t* d_quality, int64_t quality_target_stride, uint32_t bitmask,
	float* d_bkg_raw, float* d_bkg, int64_t bkg_pitch, double* d_sumimage);

/* ---- B1 + full-frame B2 / B3 / A1: the prepare stage on a frame stack ----------------------------------------
 * Frames: float32 [n_frames][frame_rows][row_pitch] resident in HBM (frame k at + k * frame_stride), as for tp_cut_stamps.
 * tp_background_mesh (B1, first half): the low-resolution mesh of backgrounds.fit_background for a plain image
 *   (photometry/backgrounds.py:89-97 pixel mask; :200-206 photutils Background2D on box_size x box_size cells with
 *   SigmaClip(3, maxiters = 5) and the SExtractor estimator): d_mesh float64 [n_frames][ny][nx] (NaN for a cell without an
 *   unmasked pixel), d_nmasked int32 [n_frames][ny][nx] (masked, padded or sigma-clipped pixels of the cell: photutils' mesh_nmasked, what its
 *   second mesh selection compares with exclude_percentile), ny = ceil(rows / box_size).
 *   d_exclude: optional uint8 manual-exclude image(s) [frame_rows][frame_cols] (exclude_frame_stride 0 = one for all frames).
 *   d_subtract: optional float32 image(s) [frame_rows][frame_cols] taken off the pixel values after the masking (the radial
 *   component of a TESS image, :200: Background2D(img0 - img_bkg_radial, mask = mask)).
 * tp_background_zoom (B1, second half): the full-resolution background from the cubic-spline coefficients of the finished
 *   mesh (after the exclusion of mostly-masked cells, their IDW fill, the 3 x 3 median filter and the spline prefilter --
 *   host work on ny x nx values per frame, photometry_amd/prepare.py), i.e. scipy.ndimage.zoom(order 3, mode 'reflect',
 *   grid_mode) as photutils' BkgZoomInterpolator calls it, clipped to [d_vmin[k], d_vmax[k]]; float32 output (what the
 *   reference's smoothing block stores, prepare.py:327).
 * tp_frames_smooth_time (B2, prepare.py:317-335), tp_frames_subtract (B3, prepare.py:419-425; flags uint8 per value) and
 *   tp_frames_sumimage (A1, prepare.py:450-453, 459) are the image-layout versions of tp_smooth_time,
 *   tp_subtract_background and tp_sumimage.                                                                          */
int tp_background_mesh(tp_ctx* ctx, const float* d_frames, int32_t n_frames, int32_t frame_rows, int32_t frame_cols,
	int64_t row_pitch, int64_t frame_stride, const uint8_t* d_exclude, int64_t exclude_frame_stride,
	const float* d_subtract, int64_t subtract_frame_stride,
	double flux_cutoff, int32_t box_size, double* d_mesh, int32_t* d_nmasked);
/* tp_background_mesh_finish: the low-resolution part of photutils Background2D after the cell statistics, per frame on the device:
 *   cells with more than exclude_percentile % masked pixels (of box_size^2) or a non-finite statistic are replaced by the
 *   inverse-distance weighted mean of the 10 nearest kept cells, the filter_size x filter_size NaN-ignoring median filter, then
 *   d_vmin / d_vmax (range of the filtered mesh) and d_coef float64 [n_frames][rows][cols] = the cubic B-spline coefficients
 *   scipy.ndimage.zoom(order 3, mode 'reflect') interpolates from (spline_filter1d along both axes) -- the inputs of
 *   tp_background_zoom.  d_filtered optional: the filtered mesh itself.  A frame without a kept cell is NaN.  <= 8192 cells.  */
int tp_background_mesh_finish(tp_ctx* ctx, const double* d_mesh, const int32_t* d_nmasked, int32_t n_frames, int32_t mesh_rows,
	int32_t mesh_cols, int32_t box_size, double exclude_percentile, int32_t filter_size, double* d_coef, double* d_vmin, double* d_vmax,
	double* d_filtered);
int tp_background_zoom(tp_ctx* ctx, const double* d_coef, const double* d_vmin, const double* d_vmax, int32_t n_frames,
	int32_t mesh_rows, int32_t mesh_cols, int32_t box_size, int32_t frame_rows, int32_t frame_cols, int64_t row_pitch, int64_t frame_stride,
	float* d_background);
int tp_frames_smooth_time(tp_ctx* ctx, int32_t n_frames, int64_t n_pixels, int64_t frame_stride, int32_t time_smooth,
	const float* d_in, float* d_out);
int tp_frames_subtract(tp_ctx* ctx, int64_t n_values, const float* d_raw, const float* d_raw_err, const float* d_bkg,
	const uint8_t* d_pixel_flags, uint32_t flag_mask, float* d_images, float* d_images_err);
int tp_frames_sumimage(tp_ctx* ctx, int32_t n_frames, int64_t n_pixels, int64_t frame_stride, const float* d_images,
	const int32_t* d_quality, uint32_t bitmask, double* d_sumimage);

/* ---- pixel flags: "background shenanigans"  -------------------------------------------------
 * tp_frames_median_filter replaces pixel_flags.pixel_background_shenanigans (photometry/pixel_flags.py:61-79):
 *   scipy.ndimage.median_filter(img - SumImage, size) with the default 'reflect' boundary, for every frame of a stack;
 *   d_reference: float64 [frame_rows][frame_cols] (the sum image) or NULL; size odd, <= 15 (the reference uses 15);
 *   float32 output (what prepare.py:533-537 stores); non-finite input values sort to the end of a window.
 * tp_frames_block_median_accumulate: one block of the "mean shenanigans" (prepare.py:558-575): acc += nanmedian over the
 *   n_block <= 32 frames listed in d_frame_index (per pixel, NaN result -> 0); the caller divides by the number of blocks.
 * tp_frames_threshold_flags: prepare.py:594-607: clears flag_bit in every pixel flag and sets it where
 *   |indicator - mean| > threshold (PixelQualityFlags.BackgroundShenanigans, bkgshe_threshold = 40).              */
/* tp_frames_pixel_flags: the per-frame pixel flags of the prepare stage (prepare.py:296-297, 406-408): bit_background
 *   (PixelQualityFlags.NotUsedForBackground) where fit_background masks the pixel (backgrounds.py:89-97: not finite, above
 *   flux_cutoff, negative, manually excluded), bit_manual (ManualExclude) where pixel_flags.pixel_manual_exclude (:13-58) does:
 *   columns >= d_first_excluded_column[k] (the host evaluates the header rules; NULL = no rule fires), and the whole frame when
 *   every pixel of it is zero and zero_is_excluded (TESS data).  d_all_zero int32 [n_frames] (out): the zero test per frame.
 *   d_pixel_flags uint8 [n_frames][rows][cols]; passed as d_exclude to tp_background_mesh / tp_radial_* it is exactly the mask.
 * tp_frames_used_in_background: backgrounds_pixels_used (prepare.py:435, 464-466): d_used uint8 [n_pixels] = the pixel's
 *   bit_background is clear in more than threshold of the frames.                                                    */
int tp_frames_pixel_flags(tp_ctx* ctx, const float* d_frames, int32_t n_frames, int32_t frame_rows, int32_t frame_cols,
	int64_t row_pitch, int64_t frame_stride, const int32_t* d_first_excluded_column, int32_t zero_is_excluded, double flux_cutoff,
	uint32_t bit_background, uint32_t bit_manual, int32_t* d_all_zero, uint8_t* d_pixel_flags);
int tp_frames_used_in_background(tp_ctx* ctx, const uint8_t* d_pixel_flags, int32_t n_frames, int64_t n_pixels, uint32_t bit_background,
	double threshold, uint8_t* d_used);
int tp_frames_median_filter(tp_ctx* ctx, const float* d_frames, int32_t n_frames, int32_t frame_rows, int32_t frame_cols,
	int64_t row_pitch, int64_t frame_stride, const double* d_reference, int32_t size, float* d_out);
int tp_frames_block_median_accumulate(tp_ctx* ctx, const float* d_frames, int64_t n_pixels, int64_t frame_stride,
	const int32_t* d_frame_index, int32_t n_block, double* d_accumulator);
int tp_frames_threshold_flags(tp_ctx* ctx, const float* d_indicator, const double* d_mean, double threshold, uint32_t flag_bit,
	int64_t n_pixels, int32_t n_frames, uint8_t* d_pixel_flags);

/* ---- B1, TESS branch: the radial component of fit_background (photometry/backgrounds.py:104-197) ---------------
 * Frames: float32 [n_frames][n_pixels] contiguous images (row_pitch == frame_cols), frame k at + k * frame_stride.  The
 * pixel mask (:89-97) is evaluated on d_frames; values are d_frames - d_square (float64) once a square (mesh) component of
 * a previous iteration exists, d_frames alone (float32 arithmetic, as numpy does it) when d_square is NULL.
 * tp_radial_zeropoint: d_zeropoint[k] = -min(values over the unmasked pixels) + 1.0 (:166-168); NaN when everything is
 *   masked.  d_partial: float64 scratch [n_frames][n_partial].
 * tp_radial_ring_modes: _reduce_mode (:20-32) of log10(values + zeropoint) over the pixels of every ring (the callable
 *   statistic of scipy.stats.binned_statistic, :171-176): d_ring_pixels int32 [n_ring_pixels] lists the pixels of ring 0,
 *   ring 1, ... (row-major inside a ring), ring j = d_ring_pixels[d_ring_offsets[j] .. d_ring_offsets[j + 1]).
 *   statsmodels KDEUnivariate(x).fit(gridsize = 2000): Gaussian kernel, FFT on 2048 grid points, cut = 3, bandwidth
 *   bandwidth_constant * min(std, IQR / 1.349) * n^(-1/5) (bw = 'normal_reference'); zero bandwidth -> median; the mode is
 *   support[argmax(density)].  d_modes float64 [n_frames][n_rings] (NaN: fewer than 2 pixels), d_counts optional int32
 *   [n_frames][n_rings], d_scratch float64 [n_frames][n_ring_pixels].
 * tp_radial_evaluate: d_out = float32(10**s(r) - zeropoint + d_add) with r = hypot(col + col_offset - xcen, row - ycen)
 *   and s the cubic spline of frame k in FITPACK form (knots d_knots[k][0..n), coefficients d_coefs[k][..], n =
 *   d_n_knots[k] <= max_knots; evaluated with ext = 3, :186-188); n == 0: no radial component (d_out = d_add or 0).
 *   d_add optional float32 images (the square component: the total background of :209).                            */
/* Images that are evaluated where they are read instead of stored (round 5: the alternation read and wrote each of them
 * several times per iteration -- 304 MB of traffic per 2048 x 2048 frame against 117 MB of algorithmic bytes):
 *   tp_zoom_image   the square component, i.e. what tp_background_zoom would write, from the outputs of
 *                   tp_background_mesh_finish (d_coef / d_vmin / d_vmax); frame_cols = columns of the frame;
 *   tp_radial_image the radial component, i.e. what tp_radial_evaluate would write without d_add, from the ring profile
 *                   (tp_radial_profiles) and the zero point.
 * The *_zoom / *_radial entries below take them in place of d_square / d_subtract / d_add and give bit-identical results. */
typedef struct tp_zoom_image {
	const double* d_coef; const double* d_vmin; const double* d_vmax;
	int32_t mesh_rows, mesh_cols, box_size, frame_cols;
} tp_zoom_image;
typedef struct tp_radial_image {
	double col_offset, xcen, ycen;
	const double* d_knots; const double* d_coefs; const int32_t* d_n_knots;
	const double* d_zeropoint;
	int32_t max_knots, reserved;
} tp_radial_image;
int tp_radial_zeropoint(tp_ctx* ctx, const float* d_frames, int32_t n_frames, int64_t n_pixels, int64_t frame_stride,
	const float* d_square, int64_t square_frame_stride, const uint8_t* d_exclude, int64_t exclude_frame_stride, double flux_cutoff,
	double* d_partial, int32_t n_partial, double* d_zeropoint);
int tp_radial_ring_modes(tp_ctx* ctx, const float* d_frames, int32_t n_frames, int64_t n_pixels, int64_t frame_stride,
	const float* d_square, int64_t square_frame_stride, const uint8_t* d_exclude, int64_t exclude_frame_stride, double flux_cutoff,
	const double* d_zeropoint, const int32_t* d_ring_pixels, const int32_t* d_ring_offsets, int32_t n_rings, int32_t n_ring_pixels,
	double bandwidth_constant, double* d_scratch, double* d_modes, int32_t* d_counts);
/* tp_radial_profiles: the ring profile of every frame on the device (backgrounds.py:178-187): utilities.move_median_central of the
 *   ring modes (width radial_smooth, 0 = none; NaN-ignoring, ends redone over the first / last k + 2 points) and the interpolating
 *   cubic spline through the rings that have a mode, in FITPACK form (interior knots on the data points x[2 .. m-3], the m x m
 *   collocation system solved in the band) -- what InterpolatedUnivariateSpline(k = 3) returns, to rounding.  d_modes float64
 *   [n_frames][n_rings] (tp_radial_ring_modes), d_bin_center float64 [n_rings]; outputs as tp_radial_evaluate takes them
 *   (d_knots / d_coefs float64 [n_frames][max_knots], d_n_knots int32; 0 = no radial component: fewer than 4 usable rings --
 *   the reference logs for fewer than 3 and FITPACK refuses exactly 3).  n_rings <= 64.                           */
int tp_radial_profiles(tp_ctx* ctx, int32_t n_frames, int32_t n_rings, const double* d_modes, const double* d_bin_center,
	int32_t radial_smooth, int32_t max_knots, double* d_knots, double* d_coefs, int32_t* d_n_knots);
int tp_radial_evaluate(tp_ctx* ctx, int32_t n_frames, int32_t frame_rows, int32_t frame_cols, int64_t frame_stride,
	double col_offset, double xcen, double ycen, const double* d_knots, const double* d_coefs, const int32_t* d_n_knots, int32_t max_knots,
	const double* d_zeropoint, const float* d_add, int64_t add_frame_stride, float* d_out);
int tp_radial_zeropoint_zoom(tp_ctx* ctx, const float* d_frames, int32_t n_frames, int64_t n_pixels, int64_t frame_stride,
	const tp_zoom_image* square, const uint8_t* d_exclude, int64_t exclude_frame_stride, double flux_cutoff,
	double* d_partial, int32_t n_partial, double* d_zeropoint);
int tp_radial_ring_modes_zoom(tp_ctx* ctx, const float* d_frames, int32_t n_frames, int64_t n_pixels, int64_t frame_stride,
	const tp_zoom_image* square, const uint8_t* d_exclude, int64_t exclude_frame_stride, double flux_cutoff,
	const double* d_zeropoint, const int32_t* d_ring_pixels, const int32_t* d_ring_offsets, int32_t n_rings, int32_t n_ring_pixels,
	double bandwidth_constant, double* d_scratch, double* d_modes, int32_t* d_counts);
/* d_out = float32(radial + square): the total background of backgrounds.py:209 from the two implicit images */
int tp_radial_evaluate_zoom(tp_ctx* ctx, int32_t n_frames, int32_t frame_rows, int32_t frame_cols, int64_t frame_stride,
	const tp_radial_image* radial, const tp_zoom_image* add, float* d_out);
/* tp_background_mesh with the radial component to subtract (backgrounds.py:200) evaluated from its ring profile; max_knots <= 72 */
int tp_background_mesh_radial(tp_ctx* ctx, const float* d_frames, int32_t n_frames, int32_t frame_rows, int32_t frame_cols,
	int64_t row_pitch, int64_t frame_stride, const uint8_t* d_exclude, int64_t exclude_frame_stride,
	const tp_radial_image* radial, double flux_cutoff, int32_t box_size, double* d_mesh, int32_t* d_nmasked);

/* ---- P1..P4: linear PSF photometry ----------------------------------------------------------------
 * tp_linpsf_prf (P1) replaces the per-target PRF construction of PSF.__init__ (photometry/psf.py:
 *   101-119).  The interpolating-spline fit is linear in the data, so the coefficient table of the
 *   inverse-distance blend is the same blend of the per-sample tables (fitted once per camera/CCD on
 *   the host with the reference's own scipy call, see photometry_amd/psf.py):
 *     d_coef[t][c] = sum_s d_weights[t][s] * d_base_coef[s][c]
 *   d_base_coef float64 [n_samples][n_coef], d_weights float64 [n_targets][n_samples] (already divided
 *   by the normalisation of psf.py:116), d_coef float64 [n_targets][n_coef].  n_samples <= 32.
 * tp_linpsf_fit (P2-P4) replaces PSF.integrate_to_image (psf.py:122-148), lsfit and the loop / status
 *   logic of LinPSFPhotometry.do_photometry (photometry/linpsf_photometry.py:22-34, 79-219).
 *   d_coef: [n_targets][n*n] spline coefficients, first axis = column direction (psf.py:119,146);
 *   d_knots_x / d_knots_y: [n+4] FITPACK knots of ANY strictly increasing sample grid, n = 4 .. 2048 (both axes alike; different
 *   lengths: tp_linpsf_fit_xy);
 *   cutoff_radius: any positive radius, +infinity = none (psf.py:142 `cutoff_radius is None`).  The SPOC layout -- evenly
 *   spaced, 9 samples per pixel, n = 32 .. 140 -- with a radius whose pixel edges stay inside the evenly spaced knots
 *   (cutoff_radius <= 5.25; LinPSFPhotometry uses 5, linpsf_photometry.py:63) runs on the fast kernels below; the library reads
 *   that off the knots on the device, and everything else is fitted by general kernels that evaluate the FITPACK box integral
 *   of psf.py:146 (dblint / fpintb: limits cut to the grid, so pixels beyond the PRF's support contribute zero) per star,
 *   pixel and cadence, with run-time sized normal equations -- same results, ~100 x slower (tp_linpsf_last_counts [13]).
 *   fitted stars (ragged, CSR): d_star_offsets int64 [n_targets+1]; d_target_index int32 [n_targets]
 *   = index of the main target among its fitted stars (the selection of linpsf_photometry.py:93-104 is
 *   host catalogue work); d_pos_row / d_pos_col float64 [n_fit_stars][pos_pitch] = row_stamp /
 *   column_stamp of every fitted star at every cadence, i.e. what catalog_attime() returns
 *   (BasePhotometry.py:1224-1258: WCS / jitter interpolation on the host).  max_stars = the largest number of
 *   fitted stars of a target, <= 64 (up to 8 stars run out of registers; more -- rare, crowded fields -- out of an HBM
 *   scratch with the same arithmetic: the reference's star selection has no limit, linpsf_photometry.py:93-104).
 *   d_subtract: optional background series subtracted from d_images on the fly (as in tp_sumimage).
 *   outputs: d_flux / d_flux_err float64 [n_targets][out_pitch] (flux_err is NaN, :169);
 *   d_fluxes_all float64 [n_fit_stars][out_pitch] every fitted flux; d_contamination float64 (PSF_CONT,
 *   :203-211); d_status int32 (OK, WARNING if contamination > 0.1, ERROR if all fluxes are NaN);
 *   d_fluxes_mean optional float64 [n_fit_stars].
 * tp_linpsf_set_path chooses between the two mappings of the fit that exist for targets with up to 4 fitted stars, up to 256
 *   pixels within reach of their cut-off circles and stars that visit at most 3 x 3 knot intervals of the PRF grid: 1 (default)
 *   the matrix-core kernel (v_mfma_f64_16x16x4_f64 on ONE tensor-product quartic spline per star and pixel, cadences in
 *   their natural order, the target's coefficients resident in LDS); 0 the vector-ALU kernels (one cadence per lane, cadences
 *   sorted by table origin, scalar-loaded polynomial coefficients) for every target.  Targets that do not qualify take the
 *   vector-ALU kernels either way.  Same results to rounding (1e-13).   */
/* tp_linpsf_fit / tp_psf_fit for a PRF spline whose two axes have different numbers of samples (psf.py:119 accepts any
 * RectBivariateSpline): d_coef [n_targets][n_coef_axis_x * n_coef_axis_y] (first axis = column direction), d_knots_x
 * [n_coef_axis_x + 4], d_knots_y [n_coef_axis_y + 4]; such a table is never the SPOC layout and is fitted by the any-grid kernels
 * (the FITPACK box integral).  With equal axis lengths they ARE tp_linpsf_fit / tp_psf_fit. */
int tp_linpsf_fit_xy(tp_ctx* ctx, const tp_cube_desc* desc, const float* d_images,
	const float* d_subtract, int64_t subtract_pitch,
	const double* d_coef, const double* d_knots_x, const double* d_knots_y, int32_t n_coef_axis_x, int32_t n_coef_axis_y, int32_t max_stars,
	const int64_t* d_star_offsets, const int32_t* d_target_index,
	const double* d_pos_row, const double* d_pos_col, int64_t pos_pitch, double cutoff_radius,
	double* d_flux, double* d_flux_err, double* d_fluxes_all, int64_t out_pitch,
	double* d_contamination, int32_t* d_status, double* d_fluxes_mean);
int tp_psf_fit_xy(tp_ctx* ctx, const tp_cube_desc* desc, const float* d_images, const float* d_backgrounds,
	const double* d_coef, const double* d_knots_x, const double* d_knots_y, int32_t n_coef_axis_x, int32_t n_coef_axis_y,
	const int64_t* d_star_offsets, const double* d_params0, const uint8_t* d_mini_aperture,
	double variance_floor, double cutoff_radius, int32_t maxiter_first, int32_t maxiter,
	double* d_flux, double* d_flux_err, double* d_centroid_row, double* d_centroid_col, int64_t out_pitch,
	double* d_params_out, int32_t* d_nit, int32_t* d_status);
int tp_linpsf_prf(tp_ctx* ctx, int32_t n_targets, int32_t n_samples, int32_t n_coef,
	const double* d_base_coef, const double* d_weights, double* d_coef);
int tp_linpsf_set_path(tp_ctx* ctx, int32_t path);
/* Which kernels fitted the targets of the LAST tp_linpsf_fit call of this context (the plan kernel decides per target):
 * counts[0] targets on the matrix cores, [1] the segments their series were cut into (a star that drifts over more than three
 * knot intervals gets one spline per stretch of the series; without drift one segment per target), [2] targets on the
 * vector-ALU polynomial kernels, [3] on the general kernel (wide excursions inside 16 cadences), [4] with more than 8 fitted
 * stars, [5..8] matrix-core targets with 1..4 fitted stars, [9..12] their segments, [13] targets fitted by the any-grid / any-radius
 * kernels (then all of them, and [0..12] are zero).  n <= 16 counters are copied. */
int tp_linpsf_last_counts(tp_ctx* ctx, int64_t* counts, int32_t n);
/* The positions a fit takes, for a field that moves as a whole: d_pos[s][k] = (double)(d_base[s] + d_shift[k]) -- the float32 sum the
 * plugin's catalogue holds after catalog_attime (BasePhotometry.py:1224-1258) for a translation, widened as
 * linpsf_photometry.py:116-121 does -- for n_stars stars and n_cad cadences, pos_pitch >= n_cad doubles per star.  (Built on the
 * host these arrays were most of the batched LinPSF entry's time: 75 MB for 2 000 targets.) */
int tp_star_positions(tp_ctx* ctx, int64_t n_stars, int32_t n_cad, const float* d_base, const float* d_shift, double* d_pos, int64_t pos_pitch);
int tp_linpsf_fit(tp_ctx* ctx, const tp_cube_desc* desc, const float* d_images,
	const float* d_subtract, int64_t subtract_pitch,
	const double* d_coef, const double* d_knots_x, const double* d_knots_y, int32_t n_coef_axis, int32_t max_stars,
	const int64_t* d_star_offsets, const int32_t* d_target_index,
	const double* d_pos_row, const double* d_pos_col, int64_t pos_pitch, double cutoff_radius,
	double* d_flux, double* d_flux_err, double* d_fluxes_all, int64_t out_pitch,
	double* d_contamination, int32_t* d_status, double* d_fluxes_mean);

/* ---- non-linear PSF photometry ------------------------------------------------
 * replaces PSFPhotometry.do_photometry (photometry/psf_photometry.py:111-196) with its likelihood (:52-90, statistic
 * 'Gaussian_d' with background) for a batch: per target and cadence a Nelder-Mead fit (scipy `_minimize_neldermead`, step for
 * step) of (row_stamp, column_stamp, flux) of the selected stars, warm-started from the previous cadence (the first one from
 * d_params0), then the aperture correction: flux = fitted flux + nansum(residuals in the mini aperture) (:164-171).
 *   d_images, d_backgrounds: cubes with the layout of desc (backgrounds may be NULL = 0);
 *   d_coef / d_knots_* / n_coef_axis: the PRF spline of every target, as for tp_linpsf_fit;
 *   fitted stars (ragged, CSR, at most 5 per target, the main target first: the selection of :117-136 is host catalogue work):
 *     d_star_offsets int64 [n_targets+1], d_params0 float64 [n_fit_stars][3] = (row_stamp, column_stamp, mag2flux(tmag));
 *   d_mini_aperture uint8 [n_targets][H*W]: _minimum_aperture (:29-41);
 *   variance_floor = n_readout * readnoise^2 / gain^2 (:84); maxiter_first / maxiter = 1500 / 500 upstream (:146-150);
 *   outputs float64 [n_targets][out_pitch]: d_flux, d_flux_err (NaN, :175), d_centroid_row / _col = pos_centroid[:, 0] / [:, 1]
 *     = the fitted (row_stamp, column_stamp) of the main target, as upstream (:176); a cadence whose fit did not finish within
 *     its iteration limit is NaN and does not update the starting point (:190-194);
 *   d_params_out optional float64 [n_fit_stars * 3][out_pitch] (every fitted parameter), d_nit optional int32
 *     [n_targets][out_pitch] (iterations used), d_status int32 (OK, :196).
 *   PRF grid and cutoff_radius as for tp_linpsf_fit: any grid (n = 4 .. 2048), any positive radius or +infinity; the SPOC
 *   layout with cutoff_radius <= 5.25 (PSFPhotometry uses 5, psf_photometry.py:26, :73)
 *   runs on the cached-biquartic kernel, anything else on its general instantiation (FITPACK box integral per evaluation). */
int tp_psf_fit(tp_ctx* ctx, const tp_cube_desc* desc, const float* d_images, const float* d_backgrounds,
	const double* d_coef, const double* d_knots_x, const double* d_knots_y, int32_t n_coef_axis,
	const int64_t* d_star_offsets, const double* d_params0, const uint8_t* d_mini_aperture,
	double variance_floor, double cutoff_radius, int32_t maxiter_first, int32_t maxiter,
	double* d_flux, double* d_flux_err, double* d_centroid_row, double* d_centroid_col, int64_t out_pitch,
	double* d_params_out, int32_t* d_nit, int32_t* d_status);

/* ---- light-curve diagnostics ---------------------------------------------
 * replaces the diagnostics block of BasePhotometry.photometry (photometry/BasePhotometry.py:1343-1407)
 * and utilities.rms_timescale (photometry/utilities.py:227-264) for a batch of light curves: the
 * reductions whose results the scheduler stores in its `diagnostics` table (taskmanager.py:543-563).
 *   d_flux, d_flux_err, d_centroid_col, d_centroid_row: float64 [n_targets][lc_pitch] (the outputs of
 *            tp_aperture_extract / tp_aperture_photometry);  d_time: float64 [n_cad] (days);
 *   d_quality / stride / bitmask as in tp_sumimage ("good" cadences, TESSQualityFlags.filter);
 *   d_status: optional int32 [n_targets]; only TP_STATUS_OK / TP_STATUS_WARNING targets are processed
 *            (all outputs NaN otherwise), like the `if self._status in (OK, WARNING)` of :1343;
 *   d_sumimage / d_mask (both or neither): for mask_size and edge_flux (:1394-1403);
 *   timescale_days: bin width of rms_hour (the reference uses 3600/86400);
 *   d_diag: float64 [n_targets][10] = mean_flux, variance, rms_hour, ptp, pos_centroid column, row,
 *            variability, mask_size, edge_flux, flags.  flags (as a float64 integer): 1 all fluxes NaN, 2 all
 *            errors NaN (both ValueError upstream, :1346-1349), 4 invalid time vector (ValueError in
 *            rms_timescale), 8 no detrending ("Could not detrend ..." warning: detrend = 0), 16 more
 *            time bins than cadences (rms_hour = NaN).  Light curves up to ~3 900 cadences are reduced out of
 *            LDS, longer ones (2-minute data) out of a context-owned HBM scratch with the same code.    */
int tp_lightcurve_diagnostics(tp_ctx* ctx, int32_t n_targets, int32_t n_cad,
	const double* d_flux, const double* d_flux_err, const double* d_centroid_col, const double* d_centroid_row, int64_t lc_pitch,
	const double* d_time, const int32_t* d_quality, int64_t quality_target_stride, uint32_t bitmask,
	const int32_t* d_status, const double* d_sumimage, const uint8_t* d_mask, int32_t height, int32_t width,
	double timescale_days, double* d_diag);

/* ---- stamp cutter -----------------------------------------------------------
 * replaces BasePhotometry._load_cube, FFI branch (photometry/BasePhotometry.py:720-742), for a batch: cuts
 * every target's stamp out of a full-frame image stack resident in HBM and writes the time-fastest cube.
 *   d_frames: float32 [n_frames][frame_rows][row_pitch] (frame k at d_frames + k*frame_stride), the HDF5 group
 *             `images/%04d` (or `images_err`, `backgrounds`) of the reference loaded once per CCD;
 *   row_offset / col_offset: PIXEL_OFFSET_ROW / PIXEL_OFFSET_COLUMN (BasePhotometry.py:724-727; 0 and 44);
 *   d_stamps: int32 [n_targets][4] = (row_min, row_max, col_min, col_max) in CCD coordinates; every stamp must be
 *             desc->height x desc->width; pixels outside the frame become NaN;
 *   d_cube:   float32 cube with the layout of desc (n_cad == n_frames); the padding of the time axis (cadences n_cad ..
 *             t_pitch of every pixel) is written as zeros: the caller need not clear the cube.             */
int tp_cut_stamps(tp_ctx* ctx, const float* d_frames, int32_t n_frames, int32_t frame_rows, int32_t frame_cols,
	int64_t row_pitch, int64_t frame_stride, int32_t row_offset, int32_t col_offset,
	const int32_t* d_stamps, const tp_cube_desc* desc, float* d_cube);
/* The same for n_stacks (1 .. 4) frame stacks of one geometry at once -- the images / images_err / backgrounds groups of a CCD
 * share their stamps: one binning of the stamps, one launch.  d_frames / d_cubes: HOST arrays of n_stacks device pointers. */
int tp_cut_stamps_multi(tp_ctx* ctx, int32_t n_stacks, const float* const* d_frames, int32_t n_frames, int32_t frame_rows, int32_t frame_cols,
	int64_t row_pitch, int64_t frame_stride, int32_t row_offset, int32_t col_offset,
	const int32_t* d_stamps, const tp_cube_desc* desc, float* const* d_cubes);
/* The same, writing only the pixel rows of every stamp's aperture mask (d_mask uint8 [n_targets][height * width], non-zero = write;
 * e.g. tp_k2p2_masks' output): tp_aperture_extract reads nothing else of the error and background cubes, and a mask is a sixth of
 * a 15 x 15 stamp on average.  The other rows of d_cubes are left as they were. */
/* The sum images of a group of stamps from the sum image of the frame (BasePhotometry.py:1001-1006: self._sumimage_full[ir1:ir2,
 * ic1:ic2]): d_out float64 [n_targets][height * width]; d_full float64 [frame_rows][row_pitch] covering the CCD rows / columns from
 * row_offset / col_offset; stamp pixels outside the frame are NaN.  d_stamps int32 [n_targets][4] as for tp_cut_stamps. */
int tp_crop_sumimage(tp_ctx* ctx, const double* d_full, int32_t frame_rows, int32_t frame_cols, int64_t row_pitch,
	int32_t row_offset, int32_t col_offset, const int32_t* d_stamps, int32_t n_targets, int32_t height, int32_t width, double* d_out);
int tp_cut_stamps_masked(tp_ctx* ctx, int32_t n_stacks, const float* const* d_frames, int32_t n_frames, int32_t frame_rows, int32_t frame_cols,
	int64_t row_pitch, int64_t frame_stride, int32_t row_offset, int32_t col_offset,
	const int32_t* d_stamps, const tp_cube_desc* desc, const uint8_t* d_mask, float* const* d_cubes);

/* ---- the batched drop-in entry as a native job engine ----------------------------------------------------
 * replaces, for every target of a CCD region at once, what run_tessphot(_mpi).py does target by target through
 * tessphot('aperture', ...): the constructor's stamp cut (BasePhotometry._load_cube, BasePhotometry.py:720-751), the
 * catalogue of the stamp (BasePhotometry.catalog, :1094-1181), AperturePhotometry.do_photometry WITH its stamp-resize loop
 * (photometry/AperturePhotometry/photometry.py:75-170; resize_stamp / _set_stamp, BasePhotometry.py:567-693) and the
 * diagnostics of BasePhotometry.photometry (:1343-1407).  The frame stacks of the region stay in HBM; the host submits a batch
 * of targets and collects it, a worker thread of the library drives the rounds in between on streams of the engine's pool (three per slot, shared by the
 * jobs in flight, and a copy stream per slot)
 * (group by stamp size, catalogue selection, tp_cut_stamps, tp_aperture_photometry / the three stand-alone kernels for small
 * groups, tp_lightcurve_diagnostics, download, the plugin's decisions), so several jobs -- one per engine slot -- overlap.
 *
 *   tp_frames_stack: the three image groups of the region, float32 [n_frames][n_rows][n_cols] in HBM, covering CCD rows
 *     [row0, row0 + n_rows) and columns [col0, col0 + n_cols) (PIXEL_OFFSET_ROW / _COLUMN); must stay valid until the job
 *     has been waited for.
 *   tp_frames_catalog: every star of the region (host arrays, copied), binned once into cells for the per-stamp selection
 *     (stars in the stamp + its 5-pixel buffer, float32 stamp coordinates as BasePhotometry.catalog); must outlive its jobs.
 *   tp_frames_submit: host arrays (copied before it returns): targets (starid, tmag, CCD row / column), their default stamps
 *     int64 [n][4] = (row_min, row_max, col_min, col_max) and h_valid (0: "Invalid stamp selected", BasePhotometry.py:671),
 *     h_attempts (retry limit, photometry.py:70-73), h_quick_break_budget (flux_limit x expected flux for a target the
 *     haloswitch quick break applies to, NaN otherwise: photometry.py:146-158), time float64 [n_frames], quality int32
 *     [n_frames]; budget_bytes: device memory the cubes of one round may take (0: a quarter of the HBM).  Fails when every
 *     slot of the engine holds a job that has not been waited for.
 *   tp_frames_wait: joins the job (its slot is free afterwards).  Then: tp_frames_counts / _targets (per target: STATUS,
 *     final stamp, number of resizes, has_result, and where its arrays are: group, position) / _group (per device pass: its
 *     size and the page-locked host block with the layout of the packed output block -- light curves [5][n][T] float64,
 *     contamination, status, flags, mask, catalogue flags, sum image, diagnostics [n][10], each field on a 256-byte boundary)
 *     / _group_lists (catalogue CSR offsets and star ids, target ids) / _events (what the reference would have logged, as
 *     codes: 1 "No flux above threshold.", 2 / 3 minimum aperture, 4 "Too many masks.", 5 an uncaught exception of the mask
 *     stage (a = kind), 6 "Could not resize stamp any further.", 7 haloswitch quick break (value = edge flux), 8 "Too many
 *     stamp resizes.", 9 "No targets in mask.", 10 / 11 a device failure (text), 12 "Invalid stamp selected").
 *   tp_frames_release: the job's host blocks go back to the engine's pool; the job is gone.                              */
typedef struct tp_frames_engine tp_frames_engine;
typedef struct tp_frames_catalog tp_frames_catalog;
typedef struct tp_frames_job tp_frames_job;
typedef struct tp_frames_stack {
	const float* d_images;
	const float* d_images_err;
	const float* d_backgrounds;
	int32_t n_frames, n_rows, n_cols, row0, col0;
	/* optional: the sum image of the region, float64 [n_rows][n_cols] -- what the reference keeps as the HDF5 dataset
	 * 'sumimage' (prepare.py:450-453, 459; tp_frames_sumimage computes it) and BasePhotometry.sumimage crops for an FFI
	 * target (BasePhotometry.py:1001-1006).  Given, every pass takes its sum images from it (tp_crop_sumimage), builds the
	 * masks first and cuts only the in-mask pixel rows of the three stacks; NULL: the sum image of every stamp is formed from
	 * its own image cube (tp_sumimage: the reference's postage-stamp branch, BasePhotometry.py:1007-1019). */
	const double* d_sumimage;
	/* (round 6: the struct grew by the four fields below -- callers built against the older header must be recompiled; zero them
	 * for the cutting path)
	 * optional (all three or none; needs d_sumimage): the TIME-MAJOR copies of the three stacks, float32 [n_rows * n_cols][t_pitch]
	 * (tp_frames_transpose; t_pitch >= n_frames, a multiple of 4; the cadences past n_frames zero).  Given, no pass cuts anything:
	 * the extraction reads a mask pixel's time series as one row of the stack (tp_aperture_extract_stack) -- the per-target cube of
	 * BasePhotometry._load_cube (BasePhotometry.py:720-751) is never materialised.  NULL: in-mask rows are cut per pass. */
	const float* d_images_t;
	const float* d_images_err_t;
	const float* d_backgrounds_t;
	int64_t t_pitch;
} tp_frames_stack;
/* [n_frames][n_pixels] (frame_stride elements from frame to frame) -> [n_pixels][t_pitch], cadences past n_frames zero */
int tp_frames_transpose(tp_ctx* ctx, const float* d_frames, int32_t n_frames, int64_t n_pixels, int64_t frame_stride, float* d_out, int64_t t_pitch);
/* tp_aperture_extract (A6, photometry.py:172-201) with the pixels' series taken from the time-major stacks of a region: stamp
 * pixel (r, c) of target t is row (d_stamps[4 t] - stack_row0 + r) * stack_cols + d_stamps[4 t + 2] - stack_col0 + c of the
 * stacks; d_backgrounds_t NULL = aperture-only (flux_background NaN).  Same kernels and operation order as on cut cubes. */
int tp_aperture_extract_stack(tp_ctx* ctx, int32_t n_targets, int32_t n_cad, int32_t height, int32_t width,
	const float* d_images_t, const float* d_images_err_t, const float* d_backgrounds_t, int64_t t_pitch,
	int32_t stack_rows, int32_t stack_cols, int32_t stack_row0, int32_t stack_col0,
	const uint8_t* d_mask, const int32_t* d_stamps, const int32_t* d_status,
	double* d_flux, double* d_flux_err, double* d_flux_background,
	double* d_centroid_col, double* d_centroid_row, int64_t out_pitch);
int tp_frames_engine_create(int device, int32_t n_slots, tp_frames_engine** out);
int tp_frames_engine_destroy(tp_frames_engine* eng);        /* every job waited for and released first */
int tp_frames_engine_info(tp_frames_engine* eng, int32_t* n_slots, int32_t* n_free, uint64_t* hbm_bytes);
int tp_frames_catalog_create(int64_t n_stars, const int64_t* h_starid, const float* h_tmag, const double* h_row, const double* h_column,
	tp_frames_catalog** out);
int tp_frames_catalog_destroy(tp_frames_catalog* cat);
int tp_frames_submit(tp_frames_engine* eng, const tp_frames_stack* stack, const tp_frames_catalog* cat,
	int32_t n_targets, const int64_t* h_starid, const double* h_tmag, const double* h_row, const double* h_column,
	const int64_t* h_stamps, const uint8_t* h_valid, const int32_t* h_attempts, const double* h_quick_break_budget,
	const double* h_time, const int32_t* h_quality, double budget_bytes, tp_frames_job** out);
int tp_frames_wait(tp_frames_job* job);
/* has the job's worker finished (tp_frames_wait would return at once)?  A caller with several jobs in flight collects whichever
 * is done and hands its slot to the next batch. */
int tp_frames_poll(tp_frames_job* job, int32_t* done);
int tp_frames_counts(tp_frames_job* job, int32_t* n_groups, int64_t* n_events);
int tp_frames_targets(tp_frames_job* job, int32_t* status, int64_t* stamps, int32_t* stamp_resizes, uint8_t* has_result, int32_t* group, int32_t* pos);
int tp_frames_group(tp_frames_job* job, int32_t g, int32_t* n_targets, int32_t* height, int32_t* width, int64_t* cat_capacity, int64_t* n_cat,
	void** h_block, uint64_t* block_nbytes);
int tp_frames_group_lists(tp_frames_job* job, int32_t g, int64_t* cat_offsets, int64_t* cat_starid, int64_t* target_starid);
int tp_frames_events(tp_frames_job* job, int32_t* target, int32_t* code, int32_t* a, int32_t* b, double* value, int32_t* text);
const char* tp_frames_text(tp_frames_job* job, int32_t k);
int tp_frames_release(tp_frames_job* job);

/* ---- multi-GPU: the final light-curve gather (RCCL over xGMI) --------------------------------
 * replaces the pickled result messages of run_tessphot_mpi.py:114-132,163-191: targets are
 * statically sharded over the ranks (one process per GPU) and the only data-path exchange is one
 * gather of the output block.  Rank 0 obtains the 128-byte id and distributes it out of band
 * (torch.distributed store, mpi4py bcast, a file ...), then every rank calls tp_comm_init.
 * With a single rank (no tp_comm_init) gather/allgather degenerate to a device copy.           */
int tp_comm_unique_id(char* id_out, int id_len);                       /* id_len >= 128 */
int tp_comm_init(tp_ctx* ctx, const char* id, int id_len, int rank, int n_ranks);
int tp_comm_destroy(tp_ctx* ctx);
int tp_comm_info(tp_ctx* ctx, int* rank, int* n_ranks);
/* d_recv (root only) holds n_ranks * nbytes_per_rank bytes, rank r's block at r * nbytes_per_rank */
int tp_comm_gather(tp_ctx* ctx, const void* d_send, void* d_recv, uint64_t nbytes_per_rank, int root);
/* The COMPACT form of a rank's output block for the gather: the light curve's flux, flux_err and flux_background planes are
 * float32 sums widened on store (AperturePhotometry/photometry.py:172-201), so the block a rank SENDS may carry them as float32
 * and lose nothing (3 of the 5 planes halve); everything else travels as it is.  One field: `count` bytes (kind 0, copied) or
 * `count` float64 values converted to float32 (kind 1) from byte `src_offset` of the full block to byte `dst_offset` of the
 * compact one.  The layouts are the host's (photometry_amd/comm.py: packed_block_layout / compact_block_layout); rank 0 widens
 * the gathered planes again (comm.expand_blocks).  Stream-ordered on the context's stream.                                      */
typedef struct { uint64_t src_offset, dst_offset, count; int32_t kind; int32_t reserved; } tp_block_field;
int tp_block_compact(tp_ctx* ctx, const void* d_block, void* d_compact, const tp_block_field* fields, int32_t n_fields);
int tp_comm_allgather(tp_ctx* ctx, const void* d_send, void* d_recv, uint64_t nbytes_per_rank);

/* ---- synthetic data (bench / test utility, not part of the reference path) -----------------
 * Fill images / images_err / backgrounds cubes on the device from scene parameters, following
 * the data model of simulation/simulateFITS.py:338-405 (see photometry_amd/simulate.py).
 *   d_star_params: float64 [n_targets][n_slots][3] = (row_stamp, col_stamp, flux); flux 0 = unused
 *   d_sigma_psf, d_bkg_level, d_bkg_phase: float64 [n_targets]
 *   d_jitter: float64 [n_cad][2] (column, row) per-cadence shift
 *   any of d_images / d_images_err / d_backgrounds / d_raw may be NULL.                        */
int tp_synth_fill(tp_ctx* ctx, const tp_cube_desc* desc, int32_t n_slots,
	const double* d_star_params, const double* d_sigma_psf, const double* d_bkg_level,
	const double* d_bkg_phase, const double* d_jitter, double readnoise, double nan_fraction,
	uint64_t seed, float* d_images, float* d_images_err, float* d_backgrounds, float* d_raw);

#ifdef __cplusplus
}
#endif
#endif /* TESSPHOT_HIP_H */
