// k2p2_lanes_host.h -- TEST-ONLY lane layer for photometry_amd/csrc/k2p2_core.h: the 64 lanes of the wavefront become
// loops, so that the K2P2 phases can be run (and sanitised) on a machine without a GPU.  Same tree association of the
// reductions as the device layer (photometry_amd/csrc/k2p2_lanes.h): bit-identical results.
#if K2P2_LANES_SECTION == 1
#define TP_DEV
#define TP_HD
#define TP_LANE_LOOP(l) for (int l = 0; l < 64; ++l)
#define TP_PAR_FOR(i, n) for (int i = 0; i < (n); ++i)
#define TP_SYNC() do {} while (0)
#define TP_SERIAL if (true)
#define TP_ATOMIC_INC(ptr) ((*(ptr))++)
#define TP_ATOMIC_OR(ptr, v) (*(ptr) |= (v))
#define TP_ATOMIC_MIN(ptr, v) do { if ((v) < *(ptr)) *(ptr) = (v); } while (0)
#define TP_ATOMIC_MAX(ptr, v) do { if ((v) > *(ptr)) *(ptr) = (v); } while (0)
#elif K2P2_LANES_SECTION == 2
#define TP_TREE(T, arr, OP) T a_[64]; for (int l = 0; l < 64; ++l) a_[l] = (arr)[l]; \
	for (int off = 32; off > 0; off >>= 1) for (int l = 0; l < off; ++l) { const T x_ = a_[l], y_ = a_[l + off]; a_[l] = OP; } return a_[0];
inline double sum_red(const Shared& k) { TP_TREE(double, k.red, x_ + y_) }
inline int sum_ired(const Shared& k) { TP_TREE(int, k.ired, x_ + y_) }
inline int or_ired(const Shared& k) { TP_TREE(int, k.ired, x_ | y_) }
inline int and_ired(const Shared& k) { TP_TREE(int, k.ired, x_ & y_) }
inline int max_ired(const Shared& k) { TP_TREE(int, k.ired, (y_ > x_) ? y_ : x_) }
inline double min_arr(const Shared&, const double* arr) { TP_TREE(double, arr, (y_ < x_) ? y_ : x_) }
inline double max_arr(const Shared&, const double* arr) { TP_TREE(double, arr, (y_ > x_) ? y_ : x_) }
#undef TP_TREE
template <class F>
inline double wave_sum_f(const Shared& k, const F& f) {
	double a_[64];
	for (int l = 0; l < 64; ++l) a_[l] = f(l);
	for (int off = 32; off > 0; off >>= 1) for (int l = 0; l < off; ++l) a_[l] = a_[l] + a_[l + off];
	return a_[0];
}
#endif
