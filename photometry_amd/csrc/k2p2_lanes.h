// k2p2_lanes.h -- the lane layer of k2p2_core.h on the device: a phase is a loop every lane of the wavefront takes part in,
// reductions are fixed binary trees over the 64 lanes (DPP row shifts / ds_bpermute), the result leaves lane 0 through
// v_readfirstlane.  Included twice by k2p2_core.h: section 1 (macros) before its declarations, section 2 (reductions over
// k2p2::Shared) inside namespace k2p2.
#if K2P2_LANES_SECTION == 1
#define TP_DEV __device__
#define TP_HD __host__ __device__
#define TP_LANE_LOOP(l) for (int l = k.lane, _once = 0; _once < 1; ++_once)
#define TP_PAR_FOR(i, n) for (int i = k.lane; i < (n); i += 64)
#define TP_SYNC() __syncthreads()
#define TP_SERIAL if (k.lane == 0)
#define TP_ATOMIC_INC(ptr) atomicAdd((ptr), 1)
#define TP_ATOMIC_OR(ptr, v) atomicOr((ptr), (v))
#define TP_ATOMIC_MIN(ptr, v) atomicMin((ptr), (v))
#define TP_ATOMIC_MAX(ptr, v) atomicMax((ptr), (v))
#define TP_NO_UNROLL _Pragma("unroll 1")
#define TP_ALWAYS_INLINE __forceinline__
#elif K2P2_LANES_SECTION == 2
// Reductions over the 64 per-lane partials in k.red / k.ired (written by a TP_LANE_LOOP, followed by TP_SYNC).  Fixed
// binary-tree association: a[l] (op)= a[l+32], then +16, ...
// lane l <- lane l + OFF: the two top steps go through the LDS crossbar (ds_bpermute), the four steps inside a row of 16
// lanes are DPP row shifts (a few cycles instead of ~60 each); the result leaves lane 0 through v_readfirstlane.
// Same pairs, same order as __shfl_down: the tree association is unchanged.
template <int OFF> inline TP_DEV int tp_down(int x) {
	if (OFF >= 16) return __shfl_down(x, OFF, 64);
	return __builtin_amdgcn_update_dpp(0, x, 0x100 + OFF, 0xF, 0xF, true); // row_shl:OFF
}
template <int OFF> inline TP_DEV double tp_down(double x) {
	const long long b = __double_as_longlong(x);
	const int lo = tp_down<OFF>((int)(b & 0xffffffffll)), hi = tp_down<OFF>((int)(b >> 32));
	return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
inline TP_DEV int tp_first(int x) { return __builtin_amdgcn_readfirstlane(x); }
inline TP_DEV double tp_first(double x) {
	const long long b = __double_as_longlong(x);
	const int lo = __builtin_amdgcn_readfirstlane((int)(b & 0xffffffffll)), hi = __builtin_amdgcn_readfirstlane((int)(b >> 32));
	return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
#define TP_TREE_STEP(T, OFF, OP) { const T y_ = tp_down<OFF>(x_); x_ = OP; }
#define TP_TREE(T, arr, OP) T x_ = (arr)[k.lane]; \
	TP_TREE_STEP(T, 32, OP) TP_TREE_STEP(T, 16, OP) TP_TREE_STEP(T, 8, OP) TP_TREE_STEP(T, 4, OP) TP_TREE_STEP(T, 2, OP) TP_TREE_STEP(T, 1, OP) \
	return tp_first(x_);
inline TP_DEV double sum_red(const Shared& k) { TP_TREE(double, k.red, x_ + y_) }
inline TP_DEV int sum_ired(const Shared& k) { TP_TREE(int, k.ired, x_ + y_) }
inline TP_DEV int or_ired(const Shared& k) { TP_TREE(int, k.ired, x_ | y_) }
inline TP_DEV int and_ired(const Shared& k) { TP_TREE(int, k.ired, x_ & y_) }
inline TP_DEV int max_ired(const Shared& k) { TP_TREE(int, k.ired, (y_ > x_) ? y_ : x_) }
inline TP_DEV double min_arr(const Shared& k, const double* arr) { TP_TREE(double, arr, (y_ < x_) ? y_ : x_) }
inline TP_DEV double max_arr(const Shared& k, const double* arr) { TP_TREE(double, arr, (y_ > x_) ? y_ : x_) }
#undef TP_TREE
// Ascending sort of k.srt[0 .. Pp), Pp <= 64 K, in REGISTERS: a lane holds K consecutive keys (K = 4: stamps up to 16 x 16; K = 16:
// up to 32 x 32, the resized stamps of the batched entry), the bitonic network's strides below K are exchanges between its own
// registers, the strides from K up exchanges with the lane l ^ (stride / K).  The LDS version (k2p2_core.h: two dependent LDS round
// trips and a fence per stage) took 28 000 cycles of a 15 x 15 target's ~360 000 (36 stages) and 15 % of a 25 x 25 target's time
// (55 stages of 1 024 keys, one wavefront: round 6, in-kernel clocks); a sorted array is a sorted array: same result.
#define TP_HAVE_WAVE_SORT 1
template <int K>
inline TP_DEV void wave_sort_regs(Shared& k) {
	const int l = k.lane, n = k.Pp;
	const double inf = __builtin_inf();
	double v[K];
#pragma unroll
	for (int j = 0; j < K; ++j) { const int i = K * l + j; v[j] = (i < n) ? k.srt[i] : inf; }
	auto cx = [](double& a, double& b, bool up) { const double mn = fmin(a, b), mx = fmax(a, b); a = up ? mn : mx; b = up ? mx : mn; };
#pragma unroll
	for (int size = 2; size <= 64 * K; size <<= 1) {
		// the direction of a key's block of `size`: from the key's index i = K l + j -- a bit of j below K, a bit of l from K up
		const bool up_lane = ((K * l) & size) == 0;
#pragma unroll
		for (int stride = size >> 1; stride >= 1; stride >>= 1) {
			if (stride >= K) {
				const int lx = stride / K;
				const bool keep_min = (((l & lx) == 0) == up_lane);
#pragma unroll
				for (int j = 0; j < K; ++j) { const double o = __shfl_xor(v[j], lx, 64); v[j] = keep_min ? fmin(v[j], o) : fmax(v[j], o); }
			} else {
#pragma unroll
				for (int j = 0; j < K; ++j)
					if ((j & stride) == 0) cx(v[j], v[j | stride], (size < K) ? ((j & size) == 0) : up_lane);
			}
		}
	}
#pragma unroll
	for (int j = 0; j < K; ++j) { const int i = K * l + j; if (i < n) k.srt[i] = v[j]; }
	__syncthreads();
}
// Tree sum (same association as sum_red) of per-lane partials produced by f(lane), without touching LDS: the partial stays
// in a register and goes straight into the shuffle tree.
template <class F>
inline TP_DEV double wave_sum_f(const Shared& k, const F& f) {
	double x_ = f(k.lane);
	x_ = x_ + tp_down<32>(x_); x_ = x_ + tp_down<16>(x_); x_ = x_ + tp_down<8>(x_);
	x_ = x_ + tp_down<4>(x_); x_ = x_ + tp_down<2>(x_); x_ = x_ + tp_down<1>(x_);
	return tp_first(x_);
}
#endif
