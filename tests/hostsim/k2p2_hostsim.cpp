// k2p2_hostsim.cpp -- TEST-ONLY host build of photometry_amd/csrc/k2p2_core.h (lane layer k2p2_lanes_host.h: the 64
// lanes of the wavefront become loops).  It exists to debug the K2P2 kernel logic on machines
// without a GPU; nothing in the product loads it.  Built by tests/test_k2p2_hostsim.py with g++.
#define K2P2_LANES_HEADER "k2p2_lanes_host.h"
#include "../../photometry_amd/csrc/k2p2_args.h"
#include <vector>
#include <cmath>
#include <cstring>

extern "C" int hostsim_k2p2(int n_targets, int H, int W, const double* sumimage,
	const int64_t* cat_offsets, const float* cat_column_stamp, const float* cat_row_stamp, const float* cat_tmag,
	const float* cat_column, const float* cat_row, const int64_t* cat_starid,
	const double* target_pos_row, const double* target_pos_column, const double* target_tmag, const int64_t* target_starid,
	const int32_t* stamps, const int32_t* aperture, const double* cut_override, double thresh,
	uint8_t* mask, int32_t* status, int32_t* flags, double* contamination, double* diag, uint8_t* cat_in_mask)
{
	k2p2::BatchArgs a;
	a.n_targets = n_targets; a.H = H; a.W = W; a.sumimage = sumimage; a.cat_offsets = cat_offsets;
	a.cat_column_stamp = cat_column_stamp; a.cat_row_stamp = cat_row_stamp; a.cat_tmag = cat_tmag;
	a.cat_column = cat_column; a.cat_row = cat_row; a.cat_starid = cat_starid;
	a.target_pos_row = target_pos_row; a.target_pos_column = target_pos_column; a.target_tmag = target_tmag;
	a.target_starid = target_starid; a.stamps = stamps; a.aperture = aperture; a.cut_override = cut_override;
	a.mask = mask; a.status = status; a.flags = flags; a.contamination = contamination; a.diag = diag; a.cat_in_mask = cat_in_mask;
	k2p2::Params prm = k2p2::default_params();
	prm.thresh = thresh;
	std::vector<double> twid(2 * k2p2::kGrid);
	for (int j = 0; j < k2p2::kGrid; ++j) {
		twid[j] = std::cos(2.0 * k2p2::kPi * (double)j / (double)k2p2::kGrid);
		twid[k2p2::kGrid + j] = std::sin(2.0 * k2p2::kPi * (double)j / (double)k2p2::kGrid);
	}
	std::vector<unsigned char> shm(k2p2::shared_bytes(H * W) + 64);
	for (int i = 0; i < n_targets; ++i) {
		std::memset(shm.data(), 0xCD, shm.size()); // poison: catches reads of uninitialised scratch
		k2p2::Shared k;
		void* base = (void*)(((uintptr_t)shm.data() + 15) & ~(uintptr_t)15);
		k2p2::shared_carve(k, base, H, W, 0, twid.data());
		k2p2::Target t;
		k2p2::make_target(a, i, t);
		k2p2::run_target(k, prm, t);
	}
	return 0;
}
