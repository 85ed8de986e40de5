// k2p2_args.h -- batch argument block shared by the HIP kernel wrapper and the host-sim harness.
#pragma once
#include "k2p2_core.h"

namespace k2p2 {

struct BatchArgs {
	int n_targets, H, W;
	const double* sumimage;            // [Nt][P]
	const int64_t* cat_offsets;        // [Nt+1]
	const float* cat_column_stamp; const float* cat_row_stamp; const float* cat_tmag;
	const float* cat_column; const float* cat_row; const int64_t* cat_starid;
	const double* target_pos_row; const double* target_pos_column; const double* target_tmag;
	const int64_t* target_starid;
	const int32_t* stamps;             // [Nt][4]
	const int32_t* aperture;           // [Nt][P]
	const double* cut_override;        // [Nt] or null
	uint8_t* mask; int32_t* status; int32_t* flags; double* contamination; double* diag; uint8_t* cat_in_mask;
};

inline TP_DEV void make_target(const BatchArgs& a, int i, Target& t) {
	const int P = a.H * a.W;
	const int64_t c0 = a.cat_offsets[i], c1 = a.cat_offsets[i + 1];
	t.S = a.sumimage + (int64_t)i * P;
	t.H = a.H; t.W = a.W;
	t.ncat = (int)(c1 - c0);
	t.cat_col = a.cat_column_stamp + c0;
	t.cat_row = a.cat_row_stamp + c0;
	t.cat_tmag = a.cat_tmag + c0;
	t.cat_ccd_col = a.cat_column + c0;
	t.cat_ccd_row = a.cat_row + c0;
	t.cat_starid = a.cat_starid + c0;
	t.tpos_row = a.target_pos_row[i];
	t.tpos_col = a.target_pos_column[i];
	t.stamp_row0 = a.stamps[4 * i + 0];
	t.stamp_col0 = a.stamps[4 * i + 2];
	t.target_tmag = a.target_tmag[i];
	t.target_starid = a.target_starid[i];
	t.aperture = a.aperture + (int64_t)i * P;
	t.cut_override = a.cut_override ? (a.cut_override + i) : nullptr;
	t.mask = a.mask + (int64_t)i * P;
	t.status = a.status + i;
	t.flags = a.flags + i;
	t.contamination = a.contamination + i;
	t.diag = a.diag ? (a.diag + (int64_t)i * 8) : nullptr;
	t.cat_in_mask = a.cat_in_mask ? (a.cat_in_mask + c0) : nullptr;
}

// photometry.py:54-64 + the normalised taps of scipy.ndimage.gaussian_filter(sigma=0.5, truncate=4)
// (exp(-2 x^2) / sum, x = -2..2, as computed by scipy's _gaussian_kernel1d)
inline TP_HD Params default_params() {
	Params p;
	p.thresh = 0.8;
	p.min_no_pixels_in_mask = 4;
	p.min_for_cluster = 4;
	p.extend_overflow = 1;
	p.reserved = 0;
	p.ws_thres = 0.0;
	p.saturation_limit = 7.0;
	p.gauss_w0 = 0x1.92b965ef5aaeep-1;
	p.gauss_w1 = 0x1.b405b9842b206p-4;
	p.gauss_w2 = 0x1.14aebe6a24088p-12;
	return p;
}

} // namespace k2p2
