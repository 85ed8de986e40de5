// background.hip -- B* (stamp-level background per cadence), B2 (time smoothing), B3 (subtraction).
//
// B2 / B3 replace photometry/prepare.py:317-335 (nanmean over a +-w cadence window, float32) and
// :419-425 (image -= background; manual-exclude pixels -> NaN).
//
// B* is the build-defined stamp analogue of backgrounds.fit_background (photometry/backgrounds.py:
// 52-211; Background2D on 64x64 tiles cannot be applied to a 15x15 stamp): pixel mask of
// backgrounds.py:89-94, one mesh cell = the stamp, SigmaClip(3 sigma, 5 iterations, median/std),
// SExtractor mode estimator, "more than 50 % masked -> no estimate".  See oracle/backgrounds.py.
//
// Mapping (gfx950).  A frame (one cadence of one target, <= 256 pixels) is owned by EIGHT lanes, 32 pixel values per
// lane in VGPRs; a wavefront holds 8 frames, a 256-thread workgroup 32 consecutive cadences (= one 128-byte line of
// every pixel's time series).  With ~70 VGPRs and 1 KiB of LDS per frame 4-5 wavefronts per SIMD are resident, so the
// loads of some wavefronts hide under the sorting of others (the round-1 kernel -- a quad per frame, 64 values per
// lane, 185 VGPRs, 2 wavefronts per SIMD -- spent 40 % of its life waiting for its loads and ran at 10.9 ms).
//   1. loads: buffer_load_dword with a per-lane voffset and a scalar soffset per pixel row (no address arithmetic on
//      the vector ALU); pixel mask of backgrounds.py:89-94 and the float64 sums of the unclipped frame on the fly;
//   2. each lane sorts its 32 values with an unrolled bitonic network (v_min_f32 / v_max_f32 on compile-time register
//      indices), then three cross-lane merge levels (2, 4, 8 lanes): a MIRROR stage (register j against register 31-j
//      of the partner lane: DPP quad_perm / row_half_mirror) turns two ascending runs into two bitonic halves, the
//      remaining stages are ascending merges; a cross-lane compare-exchange is one DPP move + one v_med3_f32 against
//      a per-lane -inf / +inf constant (min for the lower lane, max for the upper);
//   3. the sorted values are staged in LDS where they can be indexed by rank.  The kept set of the sigma clipping is
//      always a contiguous rank range: the median is two LDS reads, and the clipped ranks are found by walking in from
//      both ends of the range, eight ranks per step (one per lane of the frame), summing what leaves the range as it
//      goes.  The 3-sigma test is evaluated without division or square root: x is clipped iff
//      ((x - med) m)^2 > 9 (m s2 - s1^2), with s1, s2 the float64 sums of the m kept values.
// A wavefront stages and reads only its own 8 frames and its LDS operations execute in order: no workgroup barrier.
// Measured on the C3 cube (10 000 x 1 300 x 15x15, 13 M frames, 11.75 GB): 5.05 ms = 2.3 TB/s (round 3: the walk runs for the upper end
// alone when the lowest kept rank is not clipped -- almost always: 5.7 -> 5.2 ms; the registers past the last pixel are left out of
// the loads, the sums and the per-lane network: 5.05; rounds 1-2: 10.9, 5.4-5.7).  The kernel is bound by the
// vector ALUs, not by HBM: ~2 500 vector instructions per wavefront of 8 frames, 1 340 of them the sorting network
// (counters: 1.64 M waves, 4.3 G VALU instructions, VALU busy 57 % of the two-waves-per-SIMD issue peak, 15 of 20
// possible waves per CU resident).  Phases timed by cutting the kernel short: loads + mask + sums 2.1 ms (HBM rate
// 5.6 TB/s), + sort 2.3 ms, + clipping 1.8 ms.  Tried and dropped: staging the loads through a coalescing LDS tile
// (16-byte loads, two barriers: 6.2 ms against 5.9 -- the load phase alone got faster, 2.1 against 2.9 ms, the kernel
// did not, it is not waiting for memory); prefetching the next 32-cadence block into 32 more registers while sorting
// (5.8 -> 6.2 ms at 4 or 5 waves per SIMD: nothing left to hide); a bitonic instead of Batcher's network for the
// per-lane sort (5.9 ms); cross-lane exchanges in groups of 4 / 8 independent DPP moves (no change); four lanes per frame
// instead of eight (half the redundant clipping work, but 109 VGPRs and twice the LDS per wavefront: 6.4 ms).
// Also dropped (round 2): a wave-specialised workgroup -- eight sorter wavefronts stage 64 frames, a ninth clips them with
// one lane per frame (float thresholds, 4-ary rank searches) while the sorters work on the next block: 6.7 ms with the
// clipping switched off and 11.0 ms with it (two 66 KB workgroups per CU leave the sorters 4.5 waves per SIMD between two
// barriers per block, and the lane-per-frame clipper is a serial chain of LDS round trips that no amount of sorting hides).
// Shared staging areas (512-thread workgroups whose 8 wavefronts take one of 4-6 areas with an LDS atomic when their sort is
// done: 8 wavefronts per SIMD instead of 5): 5.62-5.66 ms against 5.53 -- occupancy is not what limits it.  The instruction
// rates do (tools/lab/valu_rate.hip): v_min / v_max / v_med3, DPP moves, v_cmp, v_cndmask and every FP64 instruction issue
// at one wave64 instruction per 4 cycles on this chip (only plain FP32 / integer add, mul, fma, logic and moves at 2), so the
// ~2 500 vector instructions of a wavefront are ~9 500 cycles and the batch ~6 ms of vector-ALU time: the kernel runs at it.
#include "common.h"
#include <cmath>
#include <utility>

namespace {

// v_min_f32 / v_max_f32 without the NaN-quieting canonicalisation fminf / fmaxf carry (no value is ever NaN here: masked
// pixels are +inf sentinels); plain asm so that the scheduler is still free to reorder
__device__ __forceinline__ float tp_min(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float tp_max(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

// Compile-time bitonic network on a register array.  One template instantiation per (size, stride)
// stage keeps every unrolled body small enough for the optimiser's full-unroll budget.
template <int N, int SIZE, int STRIDE>
struct BitonicStage {
	static __device__ __forceinline__ void run(float (&v)[N]) {
#pragma unroll
		for (int t = 0; t < N / 2; ++t) {
			const int lo = (t / STRIDE) * (STRIDE * 2) + (t % STRIDE);
			const int hi = lo + STRIDE;
			const bool up = ((lo & SIZE) == 0);
			const float a = v[lo], b = v[hi];
			const float mn = tp_min(a, b), mx = tp_max(a, b);
			v[lo] = up ? mn : mx;
			v[hi] = up ? mx : mn;
		}
		BitonicStage<N, SIZE, STRIDE / 2>::run(v);
	}
};
template <int N, int SIZE>
struct BitonicStage<N, SIZE, 0> {
	static __device__ __forceinline__ void run(float (&)[N]) {}
};

// Batcher's odd-even merge sort of N values as a compile-time list of compare-exchanges (Knuth 5.2.2 algorithm M)
template <int N> struct OemNet { unsigned char a[N * 12], b[N * 12]; int n; };
template <int N>
constexpr OemNet<N> make_oem() {
	OemNet<N> r{};
	int c = 0;
	for (int p = 1; p < N; p *= 2)
		for (int k = p; k >= 1; k /= 2)
			for (int j = k % p; j <= N - 1 - k; j += 2 * k)
				for (int i = 0; i <= ((k - 1 < N - j - k - 1) ? (k - 1) : (N - j - k - 1)); ++i)
					if ((i + j) / (2 * p) == (i + j + k) / (2 * p)) { r.a[c] = (unsigned char)(i + j); r.b[c] = (unsigned char)(i + j + k); ++c; }
	r.n = c;
	return r;
}
template <int N> struct Oem { static constexpr OemNet<N> net = make_oem<N>(); };
static_assert(Oem<32>::net.n == 191 && Oem<64>::net.n == 543, "Batcher networks of 32 / 64 keys");
// NREAL: registers NREAL..N-1 are known to hold the +inf sentinel in every lane (slots past the last pixel): an exchange whose
// upper register is one of them changes nothing (the upper register of an exchange receives the maximum) and is left out --
// 20 of the 191 exchanges for the 29 registers a 15 x 15 stamp fills
template <int N, int NREAL, size_t I>
__device__ __forceinline__ void oem_exchange(float (&v)[N]) {
	if constexpr (Oem<N>::net.b[I] < NREAL) {
		const float x = v[Oem<N>::net.a[I]], y = v[Oem<N>::net.b[I]];
		v[Oem<N>::net.a[I]] = tp_min(x, y);
		v[Oem<N>::net.b[I]] = tp_max(x, y);
	}
}
template <int N, int NREAL = N, size_t... I>
__device__ __forceinline__ void oem_sort(float (&v)[N], std::index_sequence<I...>) { (oem_exchange<N, NREAL, I>(v), ...); }

struct BkgArgs {
	const float* raw; float* out; int n_cad; int n_pix; int64_t t_pitch; int64_t out_pitch;
	float flux_cutoff; float exclude_fraction;
};

// SExtractorBackground (photutils 1.3.0: sd == 0 -> mean; |mean - med| / sd < 0.3 -> 2.5 med - 1.5 mean; else med) on the sums of
// the m kept values, as B*'s definition states it (oracle/backgrounds.py, step 4): no square root and one division.  With
// q = m s2 - s1^2 = m^2 var and e = s1 - m med = m (mean - med):  sd == 0  <=>  q <= 0;  |mean - med| / sd < 0.3  <=>  e^2 < 0.09 q.
__device__ __forceinline__ float sextractor_mode_sums(double med, double mm, double s1, double s2) {
	const double q = mm * s2 - s1 * s1;
	const double e = s1 - mm * med;
	const double mean = s1 / mm;
	double bkg = med;
	if (e * e < 0.09 * q) bkg = 2.5 * med - 1.5 * mean;
	if (!(q > 0.0)) bkg = mean;
	return (float)bkg;
}

constexpr int kFramesPerBlock = 32;   // cadences per workgroup = one 128-byte line of every pixel's time series

// DPP lane permutations inside the lanes of a frame
constexpr int kDppXor1 = 0xB1;        // quad_perm [1,0,3,2]
constexpr int kDppXor2 = 0x4E;        // quad_perm [2,3,0,1]
constexpr int kDppQuadRev = 0x1B;     // quad_perm [3,2,1,0]        (lane ^ 3)
constexpr int kDppHalfMirror = 0x141; // row_half_mirror            (lane ^ 7)
template <int CTRL> __device__ __forceinline__ int dpp_i(int x) { return __builtin_amdgcn_mov_dpp(x, CTRL, 0xF, 0xF, true); }
template <int CTRL> __device__ __forceinline__ float dpp_f(float x) { return __int_as_float(dpp_i<CTRL>(__float_as_int(x))); }
template <int CTRL> __device__ __forceinline__ double dpp_d(double x) {
	const int lo = dpp_i<CTRL>(__double2loint(x)), hi = dpp_i<CTRL>(__double2hiint(x));
	return __hiloint2double(hi, lo);
}
template <int G> __device__ __forceinline__ int frame_sum(int x) {
	x += dpp_i<kDppXor1>(x); x += dpp_i<kDppXor2>(x);
	if (G == 8) x += dpp_i<kDppHalfMirror>(x);
	return x;
}
template <int G> __device__ __forceinline__ double frame_sum(double x) {
	x += dpp_d<kDppXor1>(x); x += dpp_d<kDppXor2>(x);
	if (G == 8) x += dpp_d<kDppHalfMirror>(x);
	return x;
}

// Cross-lane compare-exchange stage: register j against register j (MIRROR: R-1 - j) of the partner lane; sel = -inf
// keeps the minimum (v_med3_f32(a, b, -inf) = min), sel = +inf the maximum.
template <int CTRL> __device__ __forceinline__ float xlane(float x) { return dpp_f<CTRL>(x); }
template <int CTRL, bool MIRROR, int R>
__device__ __forceinline__ void cross_stage(float (&v)[R], float sel) {
	if (MIRROR) {
#pragma unroll
		for (int j = 0; j < R / 2; ++j) {
			const float a = v[j], b = v[R - 1 - j];
			const float pa = xlane<CTRL>(b), pb = xlane<CTRL>(a);
			v[j] = __builtin_amdgcn_fmed3f(a, pa, sel);
			v[R - 1 - j] = __builtin_amdgcn_fmed3f(b, pb, sel);
		}
	} else {
#pragma unroll
		for (int j = 0; j < R; ++j) v[j] = __builtin_amdgcn_fmed3f(v[j], xlane<CTRL>(v[j]), sel);
	}
}
// ascending bitonic MERGE of a lane's R values (strides R/2..1)
template <int R> __device__ __forceinline__ void local_merge(float (&v)[R]) { BitonicStage<R, R, R / 2>::run(v); }

// LDS index of rank r of a staged frame: every run of 32 ranks is followed by 4 pad words, so that the 16-byte stores of the
// lanes of a frame fall into different banks
__device__ __forceinline__ int rank_idx(int r) { return r + ((r >> 5) << 2); }
// the staged value of rank r (0 <= r < 256); unsigned index arithmetic: the signed form cost two more address instructions
__device__ __forceinline__ float rank_read(const float* fr, int r) {
	const unsigned u = (unsigned)r;
	return fr[u + ((u >> 5) << 2)];
}
// wavefront votes on a bool (the header's __ballot / __any take an int: a select and a compare more per call)
__device__ __forceinline__ uint64_t tp_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ bool tp_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }
// a where the lane's bit of the wavefront mask is set, else b (v_cndmask_b32 with the mask in a scalar register pair)
__device__ __forceinline__ int tp_sel(uint64_t mask, int a, int b) { int r; asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(mask)); return r; }

// One frame (cadence) per G lanes, from the loaded values to the background estimate: mask + float64 sums, distributed sort,
// rank-staged sigma clipping, SExtractor mode.  `v` holds the lane's loaded pixel values (pixel i = G j + g in register j; the
// registers NREAL.. hold +inf) and is consumed; `fr` is the frame's staging area in LDS.  Every lane of the frame returns the
// same value (NaN: no estimate).
// G: lanes per frame (8: 32 values per lane, 8 frames per wavefront; 4: 64 values per lane, 16 frames per wavefront).
// JFULL: number of complete G-pixel rows (n_pix / G) when known at compile time (the pixel-count test is then only made for the
// last rows), -1 = test every slot.
// (the two halves of bkg_frame: the kernel that forms the sum image in the same pass does other work between them)
struct FrameSums { int n; double s1, s2; };

template <int G, int JFULL>
__device__ __forceinline__ FrameSums bkg_frame_sort(const BkgArgs& a, float (&v)[256 / G], int g)
{
	constexpr int R = 256 / G;             // values per lane
	constexpr int NREAL = (JFULL >= 0 && JFULL + 1 < R) ? (JFULL + 1) : R;   // registers 0..NREAL-1 can hold a pixel (JFULL: the first lanes only)
	const float inf = __builtin_inff();
	int n = 0;
	double s1 = 0.0, s2 = 0.0;
#pragma unroll
	for (int j = 0; j < NREAL; ++j) {
		const float x = v[j];
		// backgrounds.py:91-94: mask = ~isfinite | > flux_cutoff | < 0; slots past the last pixel are masked too
		bool ok = (x >= 0.f) && (x <= a.flux_cutoff);
		if (JFULL < 0 || j >= JFULL) ok = ok && (j * G + g < a.n_pix);
		n += ok ? 1 : 0;
		v[j] = ok ? x : inf;
		float z = ok ? x : 0.f;
		asm("" : "+v"(z));   // keep the conversion in the loop (else R doubles stay live)
		const double zd = (double)z;
		s1 += zd;
		s2 = __builtin_fma(zd, zd, s2);
	}
	n = frame_sum<G>(n);
	s1 = frame_sum<G>(s1);
	s2 = frame_sum<G>(s2);

	// --- distributed sort of the 256 values of the frame: rank = R g + j
	const float sel1 = (g & 1) ? inf : -inf, sel2 = (g & 2) ? inf : -inf, sel4 = (g & 4) ? inf : -inf;
	oem_sort<R, NREAL>(v, std::make_index_sequence<Oem<R>::net.n>());   // every lane ascending (Batcher: 191 / 543 exchanges for 32 / 64 keys)
	cross_stage<kDppXor1, true, R>(v, sel1);                     // runs of 2 lanes: mirror against lane^1,
	local_merge<R>(v);                                           //                  then ascending merges
	cross_stage<kDppQuadRev, true, R>(v, sel2);                  // runs of 4 lanes: mirror against lane^3,
	cross_stage<kDppXor1, false, R>(v, sel1);                    //                  stride R against lane^1,
	local_merge<R>(v);                                           //                  strides R/2..1
	if (G == 8) {
		cross_stage<kDppHalfMirror, true, R>(v, sel4);           // runs of 8 lanes: mirror against lane^7,
		cross_stage<kDppXor2, false, R>(v, sel2);                //                  stride 2R against lane^2,
		cross_stage<kDppXor1, false, R>(v, sel1);                //                  stride R against lane^1,
		local_merge<R>(v);                                       //                  strides R/2..1
	}
	return FrameSums{n, s1, s2};
}

template <int G, int JFULL>
__device__ __forceinline__ float bkg_frame_clip(const BkgArgs& a, float (&v)[256 / G], FrameSums fs, float* fr, int g, int lane, bool active)
{
	constexpr int R = 256 / G;
	int n = fs.n;
	double s1 = fs.s1, s2 = fs.s2;
	// --- stage the sorted frame (only the ranks that can hold a pixel; ranks >= n are +inf sentinels nobody reads)
#pragma unroll
	for (int j = 0; j < R; j += 4)
		if (g * R + j < a.n_pix) *reinterpret_cast<float4*>(fr + rank_idx(g * R + j)) = make_float4(v[j], v[j + 1], v[j + 2], v[j + 3]);
	__builtin_amdgcn_wave_barrier();

	// --- sigma clipping on the staged ranks; the lanes of a frame carry the same scalars
	const int nmasked = a.n_pix - n;
	const bool usable = active && (n > 0) && !((float)nmasked > a.exclude_fraction * (float)a.n_pix);
	int lo_i = 0, hi_i = usable ? n : 1;
	const uint64_t um = tp_ballot(usable);
	double med = 0.0;
	const int bshift = lane & (64 - G);
	constexpr unsigned BMASK = (1u << G) - 1u;
#pragma unroll 1
	for (int it = 0; ; ++it) {
		const int m = hi_i - lo_i;
		const int m1 = lo_i + (m >> 1);          // upper middle rank
		const int m0 = (m & 1) ? m1 : (m1 - 1);  // lower middle rank
		med = ((double)rank_read(fr, m0) + (double)rank_read(fr, m1)) * 0.5;
		if (it == 5) break;                        // maxiters = 5 clipping passes, then the final statistics
		const double mm = (double)m;
		double q9 = 9.0 * (mm * s2 - s1 * s1);    // 9 m^2 var
		if (!(q9 > 0.0)) q9 = 0.0;
		int top = 0, bot = 0;
		double r1 = 0.0, r2 = 0.0;
		// The lanes' conditions are kept as wavefront masks in scalar registers (a comparison writes one directly; `and` between them is
		// scalar work): as per-lane bools every vote cost a select and a compare to turn the bool back into a mask.
		uint64_t mt = um, mb = um;                  // lanes whose frame is still walking in from the top / from the bottom
		bool general;
		// The lowest kept rank decides whether anything is clipped below (the ranks are sorted): on sky-dominated stamps that is
		// almost never the case (stars clip at the top), and the walk then runs for the upper end alone -- about half the
		// instructions of a step.  Same sums, same order.
		{
			const double db0 = ((double)rank_read(fr, lo_i) - med) * mm;
			general = (um & tp_ballot(db0 < 0.0) & tp_ballot(db0 * db0 > q9)) != 0ull;
			if (!general) {
#pragma unroll 1
				for (int base = 0; ; base += G) {
					const int rt = hi_i - 1 - g - base;
					const uint64_t vt = mt & tp_ballot(rt >= lo_i);
					const float xt = rank_read(fr, tp_sel(vt, rt, lo_i));
					const double dt = ((double)xt - med) * mm;
					const uint64_t ot = vt & tp_ballot(dt > 0.0) & tp_ballot(dt * dt > q9);
					const unsigned bt = (unsigned)(ot >> bshift) & BMASK;
					const int ct = __builtin_ctz(~bt);
					{ const double x = (double)((g < ct) ? xt : 0.f); r1 += x; r2 = __builtin_fma(x, x, r2); }   // + 0.0 changes nothing
					top += ct;
					mt = tp_ballot(ct == G);
					if (mt == 0ull) break;
				}
			}
		}
		if (general)
#pragma unroll 1
		for (int base = 0; ; base += G) {
			const int rt = hi_i - 1 - g - base, rb = lo_i + g + base;
			const uint64_t vt = mt & tp_ballot(rt >= lo_i), vb = mb & tp_ballot(rb < hi_i);
			const float xt = rank_read(fr, tp_sel(vt, rt, lo_i)), xb = rank_read(fr, tp_sel(vb, rb, lo_i));
			const double dt = ((double)xt - med) * mm, db = ((double)xb - med) * mm;
			const uint64_t ot = vt & tp_ballot(dt > 0.0) & tp_ballot(dt * dt > q9), ob = vb & tp_ballot(db < 0.0) & tp_ballot(db * db > q9);
			// number of consecutive clipped ranks from the end of the range, among this step's G
			const unsigned bt = (unsigned)(ot >> bshift) & BMASK, bb = (unsigned)(ob >> bshift) & BMASK;
			const int ct = __builtin_ctz(~bt), cb = __builtin_ctz(~bb);
			{ const double x = (double)((g < ct) ? xt : 0.f); r1 += x; r2 = __builtin_fma(x, x, r2); }
			{ const double x = (double)((g < cb) ? xb : 0.f); r1 += x; r2 = __builtin_fma(x, x, r2); }
			top += ct; bot += cb;
			mt = tp_ballot(ct == G); mb = tp_ballot(cb == G);
			if ((mt | mb) == 0ull) break;
		}
		if (tp_ballot((top | bot) != 0) == 0ull) break;       // nchanged == 0 in every frame of the wavefront: the statistics are final
		s1 -= frame_sum<G>(r1);
		s2 -= frame_sum<G>(r2);
		lo_i += bot;
		hi_i -= top;
	}
	float result = __builtin_nanf("");
	{
		const double mm = usable ? (double)(hi_i - lo_i) : 1.0;
		result = usable ? sextractor_mode_sums(med, mm, s1, s2) : result;
	}
	return result;
}

template <int G, int JFULL>
__device__ __forceinline__ float bkg_frame(const BkgArgs& a, float (&v)[256 / G], float* fr, int g, int lane, bool active)
{
	const FrameSums fs = bkg_frame_sort<G, JFULL>(a, v, g);
	return bkg_frame_clip<G, JFULL>(a, v, fs, fr, g, lane, active);
}

template <int G, int JFULL>
__global__ __launch_bounds__(kFramesPerBlock * G) void tp_bkg_stamp_kernel(BkgArgs a, int frame_stride)
{
	constexpr int R = 256 / G;             // values per lane
	constexpr int FPW = 64 / G;            // frames per wavefront
	constexpr int SHIFT = (G == 8) ? 3 : 2;
	extern __shared__ __align__(16) float s_sorted[]; // [kFramesPerBlock][frame_stride]
	const int target = blockIdx.x;
	const int tid = threadIdx.x;
	const int wave = tid >> 6, lane = tid & 63;
	const int f = lane >> SHIFT, g = lane & (G - 1);   // frame within the wavefront, lane within the frame
	const int k = blockIdx.y * kFramesPerBlock + wave * FPW + f;
	const bool active = k < a.n_cad;
	const float inf = __builtin_inff();
	float* fr = s_sorted + (size_t)(wave * FPW + f) * frame_stride;

	// --- loads: pixel i = G j + g of cadence k; the descriptor covers this target's cube, the row offset is scalar
	const int pitch_b = (int)a.t_pitch * 4;
	const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(
		const_cast<float*>(a.raw + (int64_t)target * a.n_pix * a.t_pitch), 0, a.n_pix * pitch_b, 0x00020000);
	const int voff = g * pitch_b + (active ? k : (a.n_cad - 1)) * 4;
	constexpr int NREAL = (JFULL >= 0 && JFULL + 1 < R) ? (JFULL + 1) : R;
	float v[R];
#pragma unroll
	for (int j = 0; j < R; ++j) // all loads in flight before the first use; registers past the last pixel hold the sentinel
		v[j] = (j < NREAL) ? __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, j * G * pitch_b, 0)) : inf;
	const float result = bkg_frame<G, JFULL>(a, v, fr, g, lane, active);
	if (g == 0 && active) a.out[(int64_t)target * a.out_pitch + k] = result;
}

// B* + B2 + A1 in ONE pass over the raw cube (round 4).  The step of the raw-cube pipeline used to stream the cube twice: B*
// above, then the sum image of `raw - smoothed background` inside the fused aperture kernel (prepare.py:317-335 smoothing,
// :419-425 subtraction, :450-459 / BasePhotometry.py:1008-1019 sum image).  B* holds every pixel of every frame in registers, so
// the sum image is formed here and the cube is read once per step.
//   One 256-thread workgroup per TARGET walks its cadence blocks (32 cadences, frames as in the kernel above) in order.  A
//   lane keeps its loaded values (`u`, +29 VGPRs) beside the copy the sort consumes.  After the clipping the frame's staging
//   area is free and takes the RAW values [frame][pixel]; the frame's estimate goes to a ring of the last 128 cadences in LDS.
//   A split workgroup barrier per block (signal after the clipping, wait one block later: see the kernel) makes the ring entries
//   of the block visible.  Then every wavefront smooths with ONE LANE PER
//   CADENCE (lane L: cadence 32 b - 32 + L, i.e. the previous block and this one): the 2 w + 1 taps of the two windows the
//   reference uses (w = 1, 4; prepare.py:258) come from wave-shift DPP moves of one register and are added in series order
//   (nanmean in float32: tp_bkg_smooth_kernel's expression); other windows loop over the ring.  The summing phase gives a lane 4
//   pixels (one 16-byte LDS read per cadence) and walks the wavefront's eight cadences: the smoothed value of a cadence is
//   read out of its lane into a scalar register, a cadence that does not count (bad quality, no background, window not
//   complete yet) is skipped by a scalar branch, fl32(raw - smooth) is widened and added to the lane's float64 sums in
//   cadence order -- fixed order, no atomics.  A frame that passed B*'s pixel mask whole holds finite values only: no test
//   and a per-wavefront count; other frames test every difference (BasePhotometry.py:1011-1015).  The last w cadences of a
//   block need the first w estimates of the next one: their raw values wait in a small LDS buffer (owned by the last
//   wavefront) and are added one block later.  The loads of block b + 1 are issued before the barrier of block b.  At the
//   end the four wavefront sums of a pixel are added in wavefront order and divided by the count (0 -> NaN).
//   The phase after the barrier is ~200 vector instructions per wavefront and block beside the ~2 100 of B* (measured: every
//   instruction counts, the kernel runs on the vector-ALU floor like B*; the first version -- smoothing by the frames' lanes
//   through LDS, per-element finiteness tests everywhere -- had ~400 and took 6.4 ms against B*'s 5.1-5.2).
struct BkgSumArgs {
	BkgArgs b;
	float* smooth;                    // [n_targets][out_pitch]: the smoothed series (what the aperture kernel subtracts)
	const int32_t* quality; int64_t quality_stride; uint32_t bitmask;
	double* sumimage;                 // [n_targets][n_pix]
	int w;                            // time_smooth / 2, at most 8
};
constexpr int kRing = 128;
constexpr int kDppWaveShl1 = 0x130, kDppWaveShr1 = 0x138;   // lane L <- lane L + 1 / lane L - 1; lanes without a source keep `old`

// W: half width of the smoothing window when known at compile time (1, 4), -1: run-time (s.w)
template <int JFULL, int W>
__global__ __launch_bounds__(256, 4) void tp_bkg_stamp_sum_kernel(BkgSumArgs s, int frame_stride)
{
	constexpr int G = 8, R = 32, FPW = 8;
	constexpr int NREAL = (JFULL >= 0 && JFULL + 1 < R) ? (JFULL + 1) : R;
	const BkgArgs& a = s.b;
	extern __shared__ __align__(16) float s_lds[];
	float* s_sorted = s_lds;                                              // [32][frame_stride]: sorted ranks, then raw pixels
	float* s_ring = s_lds + (size_t)kFramesPerBlock * frame_stride;       // [kRing] unsmoothed estimates by cadence & 127
	float* s_pend = s_ring + kRing;                                       // [w][n_pix4]: raw values of a block's last w cadences
	const int n_pix4 = (a.n_pix + 3) & ~3;
	const int w = (W >= 0) ? W : s.w;
	unsigned char* s_good = reinterpret_cast<unsigned char*>(s_pend + (size_t)w * n_pix4);   // [n_cad]: good-quality flags (a load
	                                                                      // behind the next block's pixel loads would wait for all of them)
	const int target = blockIdx.x;
	const int tid = threadIdx.x;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
	const int f = lane >> 3, g = lane & 7;
	const float inf = __builtin_inff(), qnan = __builtin_nanf("");
	float* fr = s_sorted + (size_t)(wave * FPW + f) * frame_stride;
	float* my_frames = s_sorted + (size_t)(wave * FPW) * frame_stride;
	const int n_cad = a.n_cad;
	const int nblocks = (n_cad + kFramesPerBlock - 1) / kFramesPerBlock;
	{
		const int32_t* q = s.quality + (int64_t)target * s.quality_stride;
		for (int i = tid; i < n_cad; i += 256) s_good[i] = (((uint32_t)q[i] & s.bitmask) == 0u) ? 1 : 0;   // BasePhotometry.py:1010
	}
	float* out_raw = a.out + (int64_t)target * a.out_pitch;
	float* out_smooth = s.smooth + (int64_t)target * a.out_pitch;

	const int pitch_b = (int)a.t_pitch * 4;
	const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(
		const_cast<float*>(a.raw + (int64_t)target * a.n_pix * a.t_pitch), 0, a.n_pix * pitch_b, 0x00020000);
	float u[NREAL];
	{
		const int k0 = wave * FPW + f;
		const int voff = g * pitch_b + ((k0 < n_cad) ? k0 : (n_cad - 1)) * 4;
#pragma unroll
		for (int j = 0; j < NREAL; ++j) u[j] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, j * G * pitch_b, 0));
	}
	// a lane's pixels in the summing phase: 4 lane .. 4 lane + 3
	double acc[4] = {0.0, 0.0, 0.0, 0.0};
	int cnt[4] = {0, 0, 0, 0};
	int n_clean = 0;                  // cadences added without a test: they count for every pixel (wave-uniform)
	unsigned held_clean = 0u;         // the held-back cadences' frames passed the pixel mask whole (bit per cadence, wave-uniform)
	const int p0 = 4 * lane;
	// One cadence's four pixels; `bs` finite.  `clean`: every value is finite and every difference counts.
	auto add4 = [&](const float4 x, const float bs, const bool clean) {
		float d[4] = {x.x - bs, x.y - bs, x.z - bs, x.w - bs};   // prepare.py:421 in float32
		if (clean) n_clean++;
		else {
#pragma unroll
			for (int c = 0; c < 4; ++c) {
				const bool ok = fabsf(d[c]) <= 3.402823466e+38f;      // BasePhotometry.py:1011-1015
				d[c] = ok ? d[c] : 0.f;
				cnt[c] += ok ? 1 : 0;
			}
		}
#pragma unroll
		for (int c = 0; c < 4; ++c) acc[c] += (double)d[c];
	};
	// nanmean of the ring over [k - w, k + w] clipped to the series (tp_bkg_smooth_kernel's expression)
	auto smooth_at = [&](int k) {
		const int i1 = (k - w > 0) ? (k - w) : 0;
		const int i2 = (k + w + 1 < n_cad) ? (k + w + 1) : n_cad;
		float asum = 0.f;
		int c = 0;
		for (int i = i1; i < i2; ++i) {
			const float x = s_ring[i & (kRing - 1)];
			if (x == x) { asum += x; c++; }
		}
		return (c > 0) ? (asum / (float)c) : qnan;
	};

	// The phase that smooths and sums a block needs every wavefront's estimates of that block.  A workgroup barrier right after the
	// clipping made the four wavefronts wait for the slowest at every block (their clipping loops run for different numbers of
	// passes, their SIMDs are shared with other workgroups): 0.2 ms of the launch.  Instead the barrier is SPLIT: a wavefront
	// signals (an LDS counter) once its estimates of block b are in the ring, and waits for the four signals of block b only after
	// it has sorted block b + 1 -- by then they are almost always there.  The summing of block b therefore runs one block late,
	// between the sort and the staging of block b + 1 (the sort works in registers: block b's raw values are still in the
	// staging area), and once more after the loop for the last block.
	unsigned* s_sig = reinterpret_cast<unsigned*>(s_good + ((n_cad + 15) & ~15));
	if (tid == 0) *s_sig = 0u;
	__syncthreads();
	uint64_t clean_prev = 0ull;       // bit 8 ff: frame ff of this wavefront's previous block passed the pixel mask whole
	auto post_phase = [&](const int pb, const uint64_t clean_mask) {
		// wait for the estimates of block pb of all four wavefronts
		while (__hip_atomic_load(s_sig, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < 4u * (unsigned)(pb + 1)) __builtin_amdgcn_s_sleep(1);
		const int kb = pb * kFramesPerBlock;
		const int kmax = (kb + kFramesPerBlock - 1 < n_cad - 1) ? (kb + kFramesPerBlock - 1) : (n_cad - 1);
		const bool last = (pb + 1 == nblocks);
		const int n_held = (wave == 3 && pb > 0) ? w : 0;   // cadences of the block before that this wavefront held back
		// --- B2, one lane per cadence: lane L <-> cadence kb - 32 + L
		const int c = kb - kFramesPerBlock + lane;
		const bool mine = (lane >= 32 + wave * FPW) && (lane < 40 + wave * FPW) && (c < n_cad) && (last || c + w <= kmax);
		const bool held = (lane >= 32 - n_held) && (lane < 32);
		float bs;
		if constexpr (W > 0) {
			const float rv = (c >= 0 && c < n_cad) ? s_ring[c & (kRing - 1)] : qnan;   // outside the series: not in any window
			float dn[W], up[W];
			int t = __float_as_int(rv);
#pragma unroll
			for (int i = 0; i < W; ++i) { t = __builtin_amdgcn_update_dpp(__float_as_int(qnan), t, kDppWaveShr1, 0xF, 0xF, false); dn[i] = __int_as_float(t); }
			t = __float_as_int(rv);
#pragma unroll
			for (int i = 0; i < W; ++i) { t = __builtin_amdgcn_update_dpp(__float_as_int(qnan), t, kDppWaveShl1, 0xF, 0xF, false); up[i] = __int_as_float(t); }
			float asum = 0.f;
			int cn = 0;
#pragma unroll
			for (int j = -W; j <= W; ++j) {
				const float x = (j < 0) ? dn[-j - 1] : ((j == 0) ? rv : up[j - 1]);
				const bool ok = (x == x);
				asum += ok ? x : 0.f;
				cn += ok ? 1 : 0;
			}
			bs = (cn > 0) ? (asum / (float)cn) : qnan;
		} else {
			bs = (mine || held) ? smooth_at(c) : qnan;
		}
		float eff = qnan;
		if (mine || held) {
			out_smooth[c] = bs;
			eff = s_good[c] ? bs : qnan;        // good-quality cadences only (BasePhotometry.py:1010)
		}
		const uint64_t valid_mask = __ballot(eff == eff);
		// --- A1: the held-back cadences first (they are earlier in the series), then the block's
		if (p0 < a.n_pix) {
			for (int sl = 0; sl < n_held; ++sl) {
				const int L = 32 - n_held + sl;
				if ((valid_mask >> L) & 1ull)
					add4(*reinterpret_cast<const float4*>(s_pend + sl * n_pix4 + p0), __int_as_float(__builtin_amdgcn_readlane(__float_as_int(eff), L)), ((held_clean >> sl) & 1u) != 0u);
			}
			// four cadences' pixels are read at a time (the sorted values of the next block and its raw values are live in registers here)
#pragma unroll
			for (int h4 = 0; h4 < FPW; h4 += 4) {
				float4 x[4];
#pragma unroll
				for (int ff = 0; ff < 4; ++ff) x[ff] = *reinterpret_cast<const float4*>(my_frames + (size_t)(h4 + ff) * frame_stride + p0);
#pragma unroll
				for (int ff = 0; ff < 4; ++ff) asm volatile("" : "+v"(x[ff].x), "+v"(x[ff].y), "+v"(x[ff].z), "+v"(x[ff].w));
#pragma unroll
				for (int ff = 0; ff < 4; ++ff) {
					const int L = 32 + wave * FPW + h4 + ff;
					if ((valid_mask >> L) & 1ull)
						add4(x[ff], __int_as_float(__builtin_amdgcn_readlane(__float_as_int(eff), L)), ((clean_mask >> (8 * (h4 + ff))) & 1ull) != 0ull);
				}
			}
		}
		__builtin_amdgcn_wave_barrier();
		if (wave == 3 && !last) {
			// the block's last w cadences wait for the next block's estimates
			for (int sl = 0; sl < w; ++sl)
				if (p0 < a.n_pix) *reinterpret_cast<float4*>(s_pend + sl * n_pix4 + p0) = *reinterpret_cast<const float4*>(my_frames + (size_t)(FPW - w + sl) * frame_stride + p0);
			held_clean = 0u;
			for (int sl = 0; sl < w; ++sl) held_clean |= (unsigned)((clean_mask >> (8 * (FPW - w + sl))) & 1ull) << sl;
		}
		__builtin_amdgcn_wave_barrier();   // ... before the sorted ranks of the next block overwrite the staging area
	};

#pragma unroll 1
	for (int b = 0; b < nblocks; ++b) {
		const int kb = b * kFramesPerBlock;
		const int k = kb + wave * FPW + f;
		const bool active = k < n_cad;
		float v[R];
#pragma unroll
		for (int j = 0; j < R; ++j) v[j] = (j < NREAL) ? u[j] : inf;
		const FrameSums fs = bkg_frame_sort<G, JFULL>(a, v, g);
		if (b > 0) post_phase(b - 1, clean_prev);      // block b - 1: its raw values are still in the staging area
		const float result = bkg_frame_clip<G, JFULL>(a, v, fs, fr, g, lane, active);
		__builtin_amdgcn_wave_barrier();   // the clipping's reads of the staged ranks come before the raw values take their place
		if (g == 0 && active) { out_raw[k] = result; s_ring[k & (kRing - 1)] = result; }
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
		if (lane == 0) atomicAdd(s_sig, 1u);   // this wavefront's estimates of block b are in the ring (its LDS operations execute in order)
#pragma unroll
		for (int j = 0; j < NREAL; ++j)
			if (JFULL >= 0 ? (j < JFULL || j * G + g < a.n_pix) : (j * G + g < a.n_pix)) fr[j * G + g] = u[j];
		clean_prev = __ballot(fs.n == a.n_pix);   // no pixel masked: in particular every value is finite
		// the loads of the next block fly during its predecessor's summing phase
		if (b + 1 < nblocks) {
			const int kn = k + kFramesPerBlock;
			const int voff = g * pitch_b + ((kn < n_cad) ? kn : (n_cad - 1)) * 4;
#pragma unroll
			for (int j = 0; j < NREAL; ++j) u[j] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, j * G * pitch_b, 0));
		}
	}
	post_phase(nblocks - 1, clean_prev);
	// --- the four wavefront sums of a pixel, in wavefront order
	__syncthreads();
	double* s_acc = reinterpret_cast<double*>(s_lds);             // [4][n_pix4]
	int* s_cnt = reinterpret_cast<int*>(s_acc + 4 * n_pix4);      // [4][n_pix4]
	if (p0 < a.n_pix) {
#pragma unroll
		for (int c = 0; c < 4; ++c) { s_acc[wave * n_pix4 + p0 + c] = acc[c]; s_cnt[wave * n_pix4 + p0 + c] = cnt[c] + n_clean; }
	}
	__syncthreads();
	for (int p = tid; p < a.n_pix; p += 256) {
		const double t = ((s_acc[p] + s_acc[n_pix4 + p]) + s_acc[2 * n_pix4 + p]) + s_acc[3 * n_pix4 + p];
		const int c = s_cnt[p] + s_cnt[n_pix4 + p] + s_cnt[2 * n_pix4 + p] + s_cnt[3 * n_pix4 + p];
		s.sumimage[(int64_t)target * a.n_pix + p] = (c > 0) ? t / (double)c : __builtin_nan("");
	}
}

// Median filter of full-frame images (pixel_flags.pixel_background_shenanigans, photometry/pixel_flags.py:61-79:
// scipy.ndimage.median_filter(img - SumImage, size = 15), default boundary 'reflect'), with the sorting machinery of the
// stamp kernel above: a size x size window (<= 256 values) is one "frame" of eight lanes, a wavefront filters 8 consecutive
// pixels of an image row, a 256-thread workgroup 32.  The window values are float32((double) img - reference): rounding to
// float32 is monotone, so the median of the rounded values IS the rounded median of the float64 differences the reference
// sorts (the count is odd).  Non-finite values sort to the end (scipy's result for NaN input is unspecified); a window whose
// median is non-finite gives NaN.
struct MedianArgs {
	const float* frames; const double* reference; float* out;
	int n_rows, n_cols; int64_t row_pitch, frame_stride; int size;
};

__device__ __forceinline__ int reflect_index(int i, int n) { // (d c b a | a b c d | d c b a)
	if (n == 1) return 0;
	const int p = 2 * n;
	i %= p;
	if (i < 0) i += p;
	return (i < n) ? i : (p - 1 - i);
}

// NREAL: registers that can hold a window value (32; the 15 x 15 window the reference uses has tp_median15_quad_kernel below).  FAST: both image
// dimensions are at least the window size, so an index leaves the image by less than its length and one reflection is a
// comparison and a subtraction; the window offset of a lane's next value (8 further in raster order) follows from the last
// without a division.  (The generic index arithmetic -- two divisions and two modulo reflections per value -- cost as much as
// the sorting network: 3.4 -> 2.1 ms of kernel time per 2048 x 2048 frame; 1.7 with the pruned last merge below.)
template <int NREAL, bool FAST>
__global__ __launch_bounds__(256) void tp_median_filter_kernel(MedianArgs a)
{
	constexpr int G = 8, R = 32;
	__shared__ __align__(16) float s_win[32 * 288];
	const int tid = threadIdx.x;
	const int wave = tid >> 6, lane = tid & 63;
	const int f = lane >> 3, g = lane & 7;
	const int row = blockIdx.y, frame = blockIdx.z;
	int col = blockIdx.x * 32 + wave * 8 + f;
	const bool active = col < a.n_cols;
	if (!active) col = a.n_cols - 1;
	const int size = a.size, half = size / 2, npix = size * size;
	const float inf = __builtin_inff();
	const float* img = a.frames + (int64_t)frame * a.frame_stride;
	float v[R];
	int dy = -half, dx = g - half;   // window offset of value i = g (g < 8 <= size whenever FAST)
#pragma unroll
	for (int j = 0; j < R; ++j) {
		const int i = j * G + g;
		float x = inf;
		if (j < NREAL && i < npix) {
			int rr, cc;
			if (FAST) {
				rr = row + dy; cc = col + dx;
				rr = (rr < 0) ? (-rr - 1) : rr; rr = (rr >= a.n_rows) ? (2 * a.n_rows - 1 - rr) : rr;
				cc = (cc < 0) ? (-cc - 1) : cc; cc = (cc >= a.n_cols) ? (2 * a.n_cols - 1 - cc) : cc;
				dx += G;
				if (dx > half) { dx -= size; dy += 1; }
			} else {
				rr = reflect_index(row + i / size - half, a.n_rows); cc = reflect_index(col + i % size - half, a.n_cols);
			}
			const float raw = img[(int64_t)rr * a.row_pitch + cc];
			x = a.reference ? (float)((double)raw - a.reference[(int64_t)rr * a.n_cols + cc]) : raw;
			if (!(fabsf(x) <= 3.402823466e+38f)) x = inf;
		}
		v[j] = x;
	}
	const float sel1 = (g & 1) ? inf : -inf, sel2 = (g & 2) ? inf : -inf, sel4 = (g & 4) ? inf : -inf;
	oem_sort<R, NREAL>(v, std::make_index_sequence<Oem<R>::net.n>());
	cross_stage<kDppXor1, true, R>(v, sel1);
	local_merge<R>(v);
	cross_stage<kDppQuadRev, true, R>(v, sel2);
	cross_stage<kDppXor1, false, R>(v, sel1);
	local_merge<R>(v);
	cross_stage<kDppHalfMirror, true, R>(v, sel4);
	cross_stage<kDppXor2, false, R>(v, sel2);
	cross_stage<kDppXor1, false, R>(v, sel1);
	local_merge<R>(v);
	float* fr = s_win + (wave * 8 + f) * 288;
#pragma unroll
	for (int j = 0; j < R; j += 4) *reinterpret_cast<float4*>(fr + rank_idx(g * R + j)) = make_float4(v[j], v[j + 1], v[j + 2], v[j + 3]);
	__builtin_amdgcn_wave_barrier();
	const float med = fr[rank_idx((npix - 1) >> 1)];
	if (g == 0 && active)
		a.out[(int64_t)frame * a.frame_stride + (int64_t)row * a.row_pitch + col] = (med <= 3.402823466e+38f) ? med : __builtin_nanf("");
}

// The 15 x 15 window the reference uses, four output pixels at a time.  The windows of four horizontally adjacent pixels
// c0 .. c0+3 share twelve of their fifteen columns (c0-4 .. c0+7: 180 values); each has three columns of its own (45 values).
// The eight lanes of a group sort the shared 180 ONCE (23 registers per lane), then the 45 own values of each of the four
// pixels (6 registers per lane: a network a fifth the size), and the median of a window is the rank-112 element of the union
// of two sorted lists -- no merge is needed for that:
//     rank k of A u B  =  min over i of max(A[k - i], B[i - 1]),   i = how many of the k + 1 smallest come from B (0 .. 45),
// 46 candidates, six per lane, one cross-lane minimum.  Of the shared list only ranks 67 .. 112 can be asked for: lanes 2 and 3 of
// the group stage theirs (ranks 64 .. 127) in LDS, the others nothing.  The window values come from an LDS tile of the workgroup
// (15 rows x 128 + 14 columns of float32(img - reference), reflected at the image edges, non-finite -> +inf): the index
// arithmetic of the reflection is paid once per tile element, not once per window element.
// Per pixel ~75 vector instructions against ~260 of the kernel above (one full 256-key sort per pixel).
__global__ __launch_bounds__(256) void tp_median15_quad_kernel(MedianArgs a)
{
	constexpr int TW = 144;                       // tile row pitch: 128 output columns + 14 + 2 pad
	__shared__ float s_tile[15 * TW];
	__shared__ __align__(16) float s_a[32 * 72];  // per group: ranks 64 .. 127 of the shared list (two runs of 32 + 4 pad words each)
	__shared__ __align__(16) float s_b[32 * 72];  // per group: the sorted own values of the pixel in work (64 slots)
	const int tid = threadIdx.x;
	const int wave = tid >> 6, lane = tid & 63;
	const int f = lane >> 3, g = lane & 7;
	const int q = wave * 8 + f;                   // group of the workgroup: output columns col0 + 4 q .. + 3
	const int row = blockIdx.y, frame = blockIdx.z;
	const int col0 = blockIdx.x * 128;
	const float inf = __builtin_inff();
	const float* img = a.frames + (int64_t)frame * a.frame_stride;

	// ---- the tile: rows row-7 .. row+7, columns col0-7 .. col0+134 (both dimensions are at least 15: one reflection suffices)
	for (int e = tid; e < 15 * 142; e += 256) {
		const int dy = e / 142, t = e - dy * 142;
		int rr = row - 7 + dy, cc = col0 - 7 + t;
		float x = inf;
		if (cc < a.n_cols + 7) {                  // (columns further out belong to outputs beyond the image only)
			rr = (rr < 0) ? (-rr - 1) : rr; rr = (rr >= a.n_rows) ? (2 * a.n_rows - 1 - rr) : rr;
			cc = (cc < 0) ? (-cc - 1) : cc; cc = (cc >= a.n_cols) ? (2 * a.n_cols - 1 - cc) : cc;
			const float raw = img[(int64_t)rr * a.row_pitch + cc];
			x = a.reference ? (float)((double)raw - a.reference[(int64_t)rr * a.n_cols + cc]) : raw;
			if (!(fabsf(x) <= 3.402823466e+38f)) x = inf;
		}
		s_tile[dy * TW + t] = x;
	}
	__syncthreads();

	const float sel1 = (g & 1) ? inf : -inf, sel2 = (g & 2) ? inf : -inf, sel4 = (g & 4) ? inf : -inf;
	// ---- the shared columns: value i = 8 j + g of the 15 x 12 block in raster order, tile column 4 q + 3 + dx
	float* sa = s_a + q * 72;
	{
		constexpr int R = 32;
		float v[R];
		int dy = 0, dx = g;
		const float* base = s_tile + 4 * q + 3;
#pragma unroll
		for (int j = 0; j < R; ++j) {
			float x = inf;
			if (j < 23) {
				if (j < 22 || g < 4) x = base[dy * TW + dx];
				dx += 8;
				if (dx >= 12) { dx -= 12; dy += 1; }
			}
			v[j] = x;
		}
		oem_sort<R, 23>(v, std::make_index_sequence<Oem<R>::net.n>());
		cross_stage<kDppXor1, true, R>(v, sel1);
		local_merge<R>(v);
		cross_stage<kDppQuadRev, true, R>(v, sel2);
		cross_stage<kDppXor1, false, R>(v, sel1);
		local_merge<R>(v);
		cross_stage<kDppHalfMirror, true, R>(v, sel4);
		cross_stage<kDppXor2, false, R>(v, sel2);
		cross_stage<kDppXor1, false, R>(v, sel1);
		local_merge<R>(v);
		if (g == 2 || g == 3) {
			float* dst = sa + (g - 2) * 36;
#pragma unroll
			for (int j = 0; j < R; j += 4) *reinterpret_cast<float4*>(dst + j) = make_float4(v[j], v[j + 1], v[j + 2], v[j + 3]);
		}
	}
	// ---- the four pixels: their own three columns (tile columns 4 q + k .. 4 q + 2, then 4 q + 15 .. 4 q + 14 + k), 15 x 3 in
	// raster order, value i = 8 j + g
	float* sb = s_b + q * 72;
	float med4 = 0.f;
#pragma unroll
	for (int k = 0; k < 4; ++k) {
		constexpr int R = 8;
		float v[R];
#pragma unroll
		for (int j = 0; j < R; ++j) {
			const int i = 8 * j + g;
			float x = inf;
			if (j < 6 && i < 45) {
				const int dy = i / 3, e = i - 3 * dy;
				const int off = (k + e <= 2) ? (k + e) : (12 + k + e);
				x = s_tile[dy * TW + 4 * q + off];
			}
			v[j] = x;
		}
		oem_sort<R, 6>(v, std::make_index_sequence<Oem<R>::net.n>());
		cross_stage<kDppXor1, true, R>(v, sel1);
		local_merge<R>(v);
		cross_stage<kDppQuadRev, true, R>(v, sel2);
		cross_stage<kDppXor1, false, R>(v, sel1);
		local_merge<R>(v);
		cross_stage<kDppHalfMirror, true, R>(v, sel4);
		cross_stage<kDppXor2, false, R>(v, sel2);
		cross_stage<kDppXor1, false, R>(v, sel1);
		local_merge<R>(v);
		__builtin_amdgcn_wave_barrier();          // (the reads of the pixel before are done: LDS operations of a wavefront run in order)
		*reinterpret_cast<float4*>(sb + 4 + g * 8) = make_float4(v[0], v[1], v[2], v[3]);      // B[r] at sb[4 + r], B[-1] = -inf at sb[3]
		*reinterpret_cast<float4*>(sb + 8 + g * 8) = make_float4(v[4], v[5], v[6], v[7]);
		if (g == 0) sb[3] = -inf;
		__builtin_amdgcn_wave_barrier();
		// rank 112 of the union: candidates i = 6 g + t
		float best = inf;
#pragma unroll
		for (int t = 0; t < 6; ++t) {
			const int i = 6 * g + t;
			if (i <= 45) {
				const int ra = 112 - i - 64;          // 3 .. 48: rank of the shared list, less the 64 that are not staged
				const float xa = sa[ra + ((ra >> 5) << 2)];
				const float xb = sb[3 + i];           // B[i - 1]
				best = tp_min(best, tp_max(xa, xb));
			}
		}
		best = tp_min(best, dpp_f<kDppXor1>(best));
		best = tp_min(best, dpp_f<kDppXor2>(best));
		best = tp_min(best, dpp_f<kDppHalfMirror>(best));
		med4 = (g == k) ? best : med4;
	}
	const int col = col0 + 4 * q + g;
	if (g < 4 && col < a.n_cols)
		a.out[(int64_t)frame * a.frame_stride + (int64_t)row * a.row_pitch + col] = (med4 <= 3.402823466e+38f) ? med4 : __builtin_nanf("");
}

// nanmedian over <= 32 frames per pixel (the "mean shenanigans" blocks of prepare.py:563-575): out[p] += NaN -> 0 of the median.
// The values of a pixel sit in registers (NaN -> +inf, which sorts last), Batcher's network orders them, and the one or two middle
// ranks of the n finite ones are picked by a chain of selects: no array is indexed at run time (an insertion sort into a
// run-time indexed array lives in scratch memory: 1.6 ms per 16 frames of 2048 x 2048 against 0.1).
__global__ __launch_bounds__(256) void tp_block_median_accumulate_kernel(const float* __restrict__ frames, const int32_t* __restrict__ index, int n_block,
	int64_t n_pix, int64_t frame_stride, double* __restrict__ acc)
{
	const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (p >= n_pix) return;
	const float inf = __builtin_inff();
	float v[32];
	int n = 0;
#pragma unroll
	for (int j = 0; j < 32; ++j) {
		float x = inf;
		if (j < n_block) {   // uniform
			const float y = frames[(int64_t)index[j] * frame_stride + p];
			if (y == y) { x = y; ++n; }
		}
		v[j] = x;
	}
	oem_sort<32>(v, std::make_index_sequence<Oem<32>::net.n>());
	if (n > 0) {
		const int hi = n >> 1, lo = (n & 1) ? hi : (hi - 1);
		float a = 0.f, b = 0.f;
#pragma unroll
		for (int j = 0; j < 32; ++j) { a = (j == lo) ? v[j] : a; b = (j == hi) ? v[j] : b; }
		acc[p] += (n & 1) ? (double)b : ((double)a + (double)b) / 2.0;   // float64 block like the reference
	}
}

// flags |= bit where |indicator - mean| > threshold, bit cleared elsewhere (prepare.py:594-607)
__global__ __launch_bounds__(256) void tp_threshold_flags_kernel(const float* __restrict__ ind, const double* __restrict__ mean, double threshold,
	uint32_t bit, int64_t n_pix, int64_t n_values, uint8_t* __restrict__ flags)
{
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n_values) return;
	const double d = fabs((double)ind[i] - mean[i % n_pix]);
	const uint8_t fl = flags[i] & (uint8_t)~bit;
	flags[i] = (d > threshold) ? (uint8_t)(fl | bit) : fl;
}

// Generic fallback for stamps with more than 256 pixels: one wavefront per (target, cadence), values
// sorted in LDS.  Correct for any size that fits LDS; not tuned (large stamps are the bright-star tail).
// The arithmetic is B*'s definition (oracle/backgrounds.py, module header) operation for operation, like bkg_frame_sort /
// bkg_frame_clip above: the sums over eight interleaved accumulators (lanes 0..7 here), the division-free 3-sigma test, the
// clipped values taken off the sums eight at a time from the top and from the bottom.
__global__ __launch_bounds__(64) void tp_bkg_stamp_generic_kernel(BkgArgs a, int np2)
{
	extern __shared__ float sv[];
	const int target = blockIdx.x;
	const int k = blockIdx.y * 1 + blockIdx.z * 65535;
	const int lane = threadIdx.x;
	if (k >= a.n_cad) return;
	const float* base = a.raw + (int64_t)target * a.n_pix * a.t_pitch + k;
	const float inf = __builtin_inff();
	__shared__ int s_n;
	if (lane == 0) s_n = 0;
	__syncthreads();
	int cnt = 0;
	for (int i = lane; i < np2; i += 64) {
		float x = inf;
		if (i < a.n_pix) {
			x = base[(int64_t)i * a.t_pitch];
			const bool ok = (x >= 0.f) && (x <= a.flux_cutoff);
			cnt += ok ? 1 : 0;
			x = ok ? x : inf;
		}
		sv[i] = x;
	}
	atomicAdd(&s_n, cnt);
	__syncthreads();
	// the sums of the kept values in pixel order: accumulator g = lane g takes the pixels g, g + 8, ...
	const int g = lane & 7;
	double s1 = 0.0, s2 = 0.0;
	if (lane < 8)
		for (int i = g; i < a.n_pix; i += 8) {
			const float x = sv[i];
			const double zd = (double)((x < inf) ? x : 0.f);
			s1 += zd;
			s2 = __builtin_fma(zd, zd, s2);
		}
	s1 = frame_sum<8>(s1);
	s2 = frame_sum<8>(s2);
	__syncthreads();
	for (int size = 2; size <= np2; size <<= 1) {
		for (int stride = size >> 1; stride > 0; stride >>= 1) {
			for (int t = lane; t < np2 / 2; t += 64) {
				const int lo = ((t & ~(stride - 1)) << 1) | (t & (stride - 1)); // stride is a power of two
				const int hi = lo + stride;
				const bool up = ((lo & size) == 0);
				const float x = sv[lo], y = sv[hi];
				if ((x > y) == up) { sv[lo] = y; sv[hi] = x; }
			}
			__syncthreads();
		}
	}
	if (lane >= 8) return;
	const int n = s_n;
	float result = __builtin_nanf("");
	const int nmasked = a.n_pix - n;
	if (n > 0 && !((float)nmasked > a.exclude_fraction * (float)a.n_pix)) {
		int lo_i = 0, hi_i = n;
		double med = 0.0;
		for (int it = 0; ; ++it) {
			const int m = hi_i - lo_i;
			const int m1 = lo_i + (m >> 1);
			const int m0 = (m & 1) ? m1 : (m1 - 1);
			med = ((double)sv[m0] + (double)sv[m1]) * 0.5;
			if (it == 5) break;
			const double mm = (double)m;
			double q9 = 9.0 * (mm * s2 - s1 * s1);
			if (!(q9 > 0.0)) q9 = 0.0;
			// the clipped ranks: na from the top, nb from the bottom (every lane counts them all: the test is monotone in the rank)
			int na = 0, nb = 0;
			while (na < m) { const double d = ((double)sv[hi_i - 1 - na] - med) * mm; if ((d > 0.0) && (d * d > q9)) ++na; else break; }
			while (nb < m) { const double d = ((double)sv[lo_i + nb] - med) * mm; if ((d < 0.0) && (d * d > q9)) ++nb; else break; }
			if (na == 0 && nb == 0) break;
			double r1 = 0.0, r2 = 0.0;
			for (int t = g; t < na || t < nb; t += 8) {
				if (t < na) { const double x = (double)sv[hi_i - 1 - t]; r1 += x; r2 = __builtin_fma(x, x, r2); }
				if (t < nb) { const double x = (double)sv[lo_i + t]; r1 += x; r2 = __builtin_fma(x, x, r2); }
			}
			s1 -= frame_sum<8>(r1);
			s2 -= frame_sum<8>(r2);
			lo_i += nb;
			hi_i -= na;
		}
		result = sextractor_mode_sums(med, (double)(hi_i - lo_i), s1, s2);
	}
	if (lane == 0) a.out[(int64_t)target * a.out_pitch + k] = result;
}

// B2: bottleneck.nanmean over the window [k-w, k+w] clipped to the series (float32 accumulate)
__global__ __launch_bounds__(256) void tp_bkg_smooth_kernel(const float* __restrict__ in, float* __restrict__ out,
	int n_cad, int64_t pitch, int w)
{
	const int target = blockIdx.x;
	const int k = blockIdx.y * blockDim.x + threadIdx.x;
	if (k >= n_cad) return;
	const float* x = in + (int64_t)target * pitch;
	const int i1 = (k - w > 0) ? (k - w) : 0;
	const int i2 = (k + w + 1 < n_cad) ? (k + w + 1) : n_cad;
	float asum = 0.f;
	int cnt = 0;
	for (int n = i1; n < i2; ++n) {
		const float v = x[n];
		if (v == v) { asum += v; cnt++; }
	}
	out[(int64_t)target * pitch + k] = (cnt > 0) ? (asum / (float)cnt) : __builtin_nanf("");
}

// B3: images = raw - bkg[k]; optional manual-exclude flags (PixelQualityFlags.ManualExclude = 2) -> NaN
template <bool VEC4>
__global__ __launch_bounds__(256) void tp_bkg_subtract_kernel(const float* __restrict__ raw, const float* __restrict__ raw_err,
	const float* __restrict__ bkg, int64_t bkg_pitch, const uint8_t* __restrict__ flags, uint32_t flag_mask,
	float* __restrict__ img, float* __restrict__ err, int n_cad, int n_pix, int64_t t_pitch)
{
	const int target = blockIdx.x;
	const int64_t nq = VEC4 ? (t_pitch >> 2) : t_pitch;
	const int64_t idx = (int64_t)blockIdx.y * blockDim.x + threadIdx.x;
	if (idx >= (int64_t)n_pix * nq) return;
	const int p = (int)(idx / nq);
	const int64_t q = idx - (int64_t)p * nq;
	const int64_t off = ((int64_t)target * n_pix + p) * t_pitch + (VEC4 ? q * 4 : q);
	const float* b = bkg + (int64_t)target * bkg_pitch + (VEC4 ? q * 4 : q);
	constexpr int V = VEC4 ? 4 : 1;
	float x[V], e[V], bb[V];
	if (VEC4) {
		const float4 t = *reinterpret_cast<const float4*>(raw + off); x[0] = t.x; x[1 % V] = t.y; x[2 % V] = t.z; x[3 % V] = t.w;
		const float4 u = *reinterpret_cast<const float4*>(b); bb[0] = u.x; bb[1 % V] = u.y; bb[2 % V] = u.z; bb[3 % V] = u.w;
		if (raw_err) { const float4 s = *reinterpret_cast<const float4*>(raw_err + off); e[0] = s.x; e[1 % V] = s.y; e[2 % V] = s.z; e[3 % V] = s.w; }
	} else {
		x[0] = raw[off]; bb[0] = b[0];
		if (raw_err) e[0] = raw_err[off];
	}
#pragma unroll
	for (int j = 0; j < V; ++j) {
		const int64_t kk = (VEC4 ? q * 4 : q) + j;
		float r = x[j] - bb[j];
		bool excl = false;
		if (flags && kk < n_cad) excl = (flags[((int64_t)target * n_pix + p) * n_cad + kk] & flag_mask) != 0;
		x[j] = excl ? __builtin_nanf("") : r;
		if (raw_err) e[j] = excl ? __builtin_nanf("") : e[j];
	}
	if (VEC4) {
		*reinterpret_cast<float4*>(img + off) = make_float4(x[0], x[1 % V], x[2 % V], x[3 % V]);
		if (raw_err && err) *reinterpret_cast<float4*>(err + off) = make_float4(e[0], e[1 % V], e[2 % V], e[3 % V]);
	} else {
		img[off] = x[0];
		if (raw_err && err) err[off] = e[0];
	}
}

} // namespace

extern "C" int tp_background_stamp(tp_ctx* ctx, const tp_cube_desc* desc, const float* d_raw,
	double flux_cutoff, double exclude_percentile, float* d_bkg, int64_t bkg_pitch)
{
	TP_CHECK_CTX(ctx);
	TP_API_BEGIN
	TP_REQUIRE(ctx, tp_desc_ok(desc), "tp_background_stamp: bad cube descriptor");
	TP_REQUIRE(ctx, d_raw && d_bkg, "tp_background_stamp: null pointer");
	TP_REQUIRE(ctx, bkg_pitch >= desc->n_cad, "tp_background_stamp: bkg_pitch < n_cad");
	if (desc->n_targets == 0 || desc->n_cad == 0) return TP_OK;
	BkgArgs a;
	a.raw = d_raw; a.out = d_bkg; a.n_cad = desc->n_cad; a.n_pix = desc->height * desc->width;
	a.t_pitch = desc->t_pitch; a.out_pitch = bkg_pitch;
	a.flux_cutoff = (float)flux_cutoff; a.exclude_fraction = (float)(exclude_percentile / 100.0);
	if (a.n_pix <= 256) {
		// staged frame: rank_idx of the last 16-byte store + 1
		const int last = (a.n_pix - 1) | 3;
		const int frame_stride = ((last + ((last >> 5) << 2) + 1) + 3) & ~3;
		const size_t shmem = (size_t)kFramesPerBlock * frame_stride * sizeof(float);
		TP_REQUIRE(ctx, (int64_t)a.n_pix * a.t_pitch * 4 < 2147483647ll, "tp_background_stamp: stamp cube too large");
		dim3 grid((unsigned)desc->n_targets, (unsigned)((desc->n_cad + kFramesPerBlock - 1) / kFramesPerBlock));
		constexpr int G = 8;   // lanes per frame; 4 (16 frames per wavefront, 64 values per lane, 109 VGPRs, 16 KB of LDS per wavefront) measured 6.4 ms against 5.6
		dim3 block(kFramesPerBlock * G);
		const int jfull = a.n_pix / G;
		if (jfull == 225 / G) TP_LAUNCH(ctx, TPK_BKG_STAMP, (tp_bkg_stamp_kernel<G, 225 / G>), grid, block, shmem, a, frame_stride);        // 15 x 15
		else if (jfull == 121 / G) TP_LAUNCH(ctx, TPK_BKG_STAMP, (tp_bkg_stamp_kernel<G, 121 / G>), grid, block, shmem, a, frame_stride);   // 11 x 11
		else TP_LAUNCH(ctx, TPK_BKG_STAMP, (tp_bkg_stamp_kernel<G, -1>), grid, block, shmem, a, frame_stride);
	} else {
		int np2 = 1;
		while (np2 < a.n_pix) np2 <<= 1;
		const size_t shmem = (size_t)np2 * sizeof(float);
		TP_REQUIRE(ctx, shmem <= 150 * 1024, "tp_background_stamp: stamp too large");
		TP_REQUIRE(ctx, desc->n_cad <= 2147483647, "tp_background_stamp: too many cadences");
		if (shmem > 64 * 1024)
			TP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(tp_bkg_stamp_generic_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
		const unsigned gy = (unsigned)((desc->n_cad < 65535) ? desc->n_cad : 65535);
		const unsigned gz = (unsigned)((desc->n_cad + 65534) / 65535);
		dim3 block(64), grid((unsigned)desc->n_targets, gy, gz);
		TP_LAUNCH(ctx, TPK_BKG_STAMP, tp_bkg_stamp_generic_kernel, grid, block, shmem, a, np2);
	}
	TP_LAUNCH_CHECK(ctx, "tp_bkg_stamp_kernel");
	return TP_OK;
	TP_API_END(ctx)
}

extern "C" int tp_background_sumimage(tp_ctx* ctx, const tp_cube_desc* desc, const float* d_raw,
	double flux_cutoff, double exclude_percentile, int32_t time_smooth,
	const int32_t* d_quality, int64_t quality_target_stride, uint32_t bitmask,
	float* d_bkg_raw, float* d_bkg, int64_t bkg_pitch, double* d_sumimage)
{
	TP_CHECK_CTX(ctx);
	TP_API_BEGIN
	TP_REQUIRE(ctx, tp_desc_ok(desc), "tp_background_sumimage: bad cube descriptor");
	TP_REQUIRE(ctx, d_raw && d_quality && d_bkg_raw && d_bkg && d_sumimage && d_bkg_raw != d_bkg, "tp_background_sumimage: null or aliased pointers");
	TP_REQUIRE(ctx, bkg_pitch >= desc->n_cad && time_smooth >= 1, "tp_background_sumimage: bkg_pitch < n_cad or time_smooth < 1");
	TP_REQUIRE(ctx, quality_target_stride == 0 || quality_target_stride >= desc->n_cad, "tp_background_sumimage: bad quality stride");
	if (desc->n_targets == 0) return TP_OK;
	const int n_pix = desc->height * desc->width;
	const int w = time_smooth / 2;
	// a frame's area holds its sorted ranks (rank_idx of the last 16-byte store + 1), then its raw pixels; an area stride that is
	// a multiple of 32 words would put the eight frames of a wavefront on the same banks when the raw values are written
	const int last = (n_pix - 1) | 3;
	int frame_stride = ((last + ((last >> 5) << 2) + 1) + 3) & ~3;
	if (frame_stride % 32 == 0) frame_stride += 8;
	const int n_pix4 = (n_pix + 3) & ~3;
	const size_t shmem = ((size_t)kFramesPerBlock * frame_stride + kRing + (size_t)(w <= 8 ? w : 0) * n_pix4) * sizeof(float) + (((size_t)desc->n_cad + 15) & ~(size_t)15) + 16;
	if (n_pix > 256 || w > 8 || desc->n_cad == 0 || shmem > 160 * 1024 || (int64_t)n_pix * desc->t_pitch * 4 >= 2147483647ll) {
		// stamps above 256 pixels (the bright-star tail), windows beyond the reference's two (prepare.py:258), series whose quality
		// flags do not fit the LDS: the three stages one after the other -- same series, the sum image to rounding (another
		// order of the float64 additions)
		int rc = tp_background_stamp(ctx, desc, d_raw, flux_cutoff, exclude_percentile, d_bkg_raw, bkg_pitch);
		if (rc != TP_OK) return rc;
		rc = tp_smooth_time(ctx, desc->n_targets, desc->n_cad, bkg_pitch, time_smooth, d_bkg_raw, d_bkg);
		if (rc != TP_OK) return rc;
		return tp_sumimage(ctx, desc, d_raw, d_quality, quality_target_stride, bitmask, d_bkg, bkg_pitch, d_sumimage);
	}
	BkgSumArgs s;
	s.b.raw = d_raw; s.b.out = d_bkg_raw; s.b.n_cad = desc->n_cad; s.b.n_pix = n_pix;
	s.b.t_pitch = desc->t_pitch; s.b.out_pitch = bkg_pitch;
	s.b.flux_cutoff = (float)flux_cutoff; s.b.exclude_fraction = (float)(exclude_percentile / 100.0);
	s.smooth = d_bkg; s.quality = d_quality; s.quality_stride = quality_target_stride; s.bitmask = bitmask; s.sumimage = d_sumimage; s.w = w;
	const dim3 grid((unsigned)desc->n_targets), block(256);
	const int jfull = n_pix / 8;
#define TP_BKG_SUM_LAUNCH(JF, WW) do { \
		auto kern = tp_bkg_stamp_sum_kernel<JF, WW>; \
		if (shmem > 64 * 1024) TP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem)); \
		TP_LAUNCH(ctx, TPK_BKG_STAMP_SUM, kern, grid, block, shmem, s, frame_stride); \
	} while (0)
#define TP_BKG_SUM_W(JF) do { if (w == 1) TP_BKG_SUM_LAUNCH(JF, 1); else if (w == 4) TP_BKG_SUM_LAUNCH(JF, 4); else TP_BKG_SUM_LAUNCH(JF, -1); } while (0)
	if (jfull == 225 / 8) TP_BKG_SUM_W(225 / 8);        // 15 x 15
	else if (jfull == 121 / 8) TP_BKG_SUM_W(121 / 8);   // 11 x 11
	else TP_BKG_SUM_W(-1);
#undef TP_BKG_SUM_W
#undef TP_BKG_SUM_LAUNCH
	TP_LAUNCH_CHECK(ctx, "tp_bkg_stamp_sum_kernel");
	return TP_OK;
	TP_API_END(ctx)
}

extern "C" int tp_smooth_time(tp_ctx* ctx, int32_t n_targets, int32_t n_cad, int64_t pitch, int32_t time_smooth,
	const float* d_in, float* d_out)
{
	TP_CHECK_CTX(ctx);
	TP_API_BEGIN
	TP_REQUIRE(ctx, n_targets >= 0 && n_cad >= 0 && pitch >= n_cad && time_smooth >= 1, "tp_smooth_time: bad geometry");
	TP_REQUIRE(ctx, d_in && d_out && d_in != d_out, "tp_smooth_time: null or aliased pointers");
	if (n_targets == 0 || n_cad == 0) return TP_OK;
	dim3 block(256), grid((unsigned)n_targets, (unsigned)((n_cad + 255) / 256));
	TP_LAUNCH(ctx, TPK_BKG_SMOOTH, tp_bkg_smooth_kernel, grid, block, 0, d_in, d_out, (int)n_cad, pitch, (int)(time_smooth / 2));
	TP_LAUNCH_CHECK(ctx, "tp_bkg_smooth_kernel");
	return TP_OK;
	TP_API_END(ctx)
}

extern "C" int tp_subtract_background(tp_ctx* ctx, const tp_cube_desc* desc, const float* d_raw, const float* d_raw_err,
	const float* d_bkg, int64_t bkg_pitch, const uint8_t* d_pixel_flags, uint32_t flag_mask,
	float* d_images, float* d_images_err)
{
	TP_CHECK_CTX(ctx);
	TP_API_BEGIN
	TP_REQUIRE(ctx, tp_desc_ok(desc), "tp_subtract_background: bad cube descriptor");
	TP_REQUIRE(ctx, d_raw && d_bkg && d_images, "tp_subtract_background: null pointer");
	TP_REQUIRE(ctx, bkg_pitch >= desc->t_pitch, "tp_subtract_background: bkg_pitch must be >= the cube's t_pitch");
	if (desc->n_targets == 0 || desc->n_cad == 0) return TP_OK;
	const int n_pix = desc->height * desc->width;
	bool vec4 = tp_vec4_ok(d_raw, desc->t_pitch) && tp_vec4_ok(d_images, desc->t_pitch) && tp_vec4_ok(d_bkg, bkg_pitch)
;
	if (d_raw_err) vec4 = vec4 && tp_vec4_ok(d_raw_err, desc->t_pitch) && (!d_images_err || tp_vec4_ok(d_images_err, desc->t_pitch));
	const int64_t nq = vec4 ? (desc->t_pitch / 4) : desc->t_pitch;
	const int64_t per_target = (int64_t)n_pix * nq;
	TP_REQUIRE(ctx, (per_target + 255) / 256 <= 65535, "tp_subtract_background: stamp cube too large");
	dim3 block(256), grid((unsigned)desc->n_targets, (unsigned)((per_target + 255) / 256));
	if (vec4) {
		TP_LAUNCH(ctx, TPK_BKG_SUBTRACT, tp_bkg_subtract_kernel<true>, grid, block, 0, d_raw, d_raw_err, d_bkg, bkg_pitch,
			d_pixel_flags, flag_mask, d_images, d_images_err, desc->n_cad, n_pix, desc->t_pitch);
	} else {
		TP_LAUNCH(ctx, TPK_BKG_SUBTRACT, tp_bkg_subtract_kernel<false>, grid, block, 0, d_raw, d_raw_err, d_bkg, bkg_pitch,
			d_pixel_flags, flag_mask, d_images, d_images_err, desc->n_cad, n_pix, desc->t_pitch);
	}
	TP_LAUNCH_CHECK(ctx, "tp_bkg_subtract_kernel");
	return TP_OK;
	TP_API_END(ctx)
}

extern "C" int tp_frames_median_filter(tp_ctx* ctx, const float* d_frames, int32_t n_frames, int32_t frame_rows, int32_t frame_cols,
	int64_t row_pitch, int64_t frame_stride, const double* d_reference, int32_t size, float* d_out)
{
	TP_CHECK_CTX(ctx);
	TP_API_BEGIN
	TP_REQUIRE(ctx, d_frames && d_out && d_frames != d_out, "tp_frames_median_filter: null or aliased pointers");
	TP_REQUIRE(ctx, n_frames >= 0 && n_frames <= 65535 && frame_rows > 0 && frame_rows <= 65535 && frame_cols > 0 && row_pitch >= frame_cols
		&& frame_stride >= (int64_t)frame_rows * row_pitch, "tp_frames_median_filter: bad frame geometry");
	TP_REQUIRE(ctx, size >= 1 && (size & 1) == 1 && size * size <= 256, "tp_frames_median_filter: size must be odd, at most 15");
	if (n_frames == 0) return TP_OK;
	MedianArgs a;
	a.frames = d_frames; a.reference = d_reference; a.out = d_out; a.n_rows = frame_rows; a.n_cols = frame_cols;
	a.row_pitch = row_pitch; a.frame_stride = frame_stride; a.size = size;
	dim3 grid((unsigned)((frame_cols + 31) / 32), (unsigned)frame_rows, (unsigned)n_frames);
	const bool fast = frame_rows >= size && frame_cols >= size && size >= 8;
	if (size == 15 && fast) {
		dim3 grid4((unsigned)((frame_cols + 127) / 128), (unsigned)frame_rows, (unsigned)n_frames);
		TP_LAUNCH(ctx, TPK_MEDIAN_FILTER, tp_median15_quad_kernel, grid4, dim3(256), 0, a);
	}
	else if (fast) TP_LAUNCH(ctx, TPK_MEDIAN_FILTER, (tp_median_filter_kernel<32, true>), grid, dim3(256), 0, a);
	else TP_LAUNCH(ctx, TPK_MEDIAN_FILTER, (tp_median_filter_kernel<32, false>), grid, dim3(256), 0, a);
	TP_LAUNCH_CHECK(ctx, "tp_median_filter_kernel");
	return TP_OK;
	TP_API_END(ctx)
}

extern "C" int tp_frames_block_median_accumulate(tp_ctx* ctx, const float* d_frames, int64_t n_pixels, int64_t frame_stride,
	const int32_t* d_frame_index, int32_t n_block, double* d_accumulator)
{
	TP_CHECK_CTX(ctx);
	TP_API_BEGIN
	TP_REQUIRE(ctx, d_frames && d_frame_index && d_accumulator, "tp_frames_block_median_accumulate: null pointer");
	TP_REQUIRE(ctx, n_block >= 1 && n_block <= 32 && n_pixels >= 0 && frame_stride >= n_pixels, "tp_frames_block_median_accumulate: at most 32 frames per block");
	if (n_pixels == 0) return TP_OK;
	TP_LAUNCH(ctx, TPK_MEDIAN_FILTER, tp_block_median_accumulate_kernel, dim3((unsigned)((n_pixels + 255) / 256)), dim3(256), 0, d_frames, d_frame_index,
		(int)n_block, n_pixels, frame_stride, d_accumulator);
	TP_LAUNCH_CHECK(ctx, "tp_block_median_accumulate_kernel");
	return TP_OK;
	TP_API_END(ctx)
}

extern "C" int tp_frames_threshold_flags(tp_ctx* ctx, const float* d_indicator, const double* d_mean, double threshold, uint32_t flag_bit,
	int64_t n_pixels, int32_t n_frames, uint8_t* d_pixel_flags)
{
	TP_CHECK_CTX(ctx);
	TP_API_BEGIN
	TP_REQUIRE(ctx, d_indicator && d_mean && d_pixel_flags, "tp_frames_threshold_flags: null pointer");
	TP_REQUIRE(ctx, n_pixels > 0 && n_frames >= 0 && flag_bit != 0 && flag_bit < 256, "tp_frames_threshold_flags: bad arguments");
	const int64_t nv = n_pixels * n_frames;
	if (nv == 0) return TP_OK;
	TP_REQUIRE(ctx, (nv + 255) / 256 <= 2147483647ll, "tp_frames_threshold_flags: too many values for one launch");
	TP_LAUNCH(ctx, TPK_MEDIAN_FILTER, tp_threshold_flags_kernel, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, d_indicator, d_mean, threshold, flag_bit,
		n_pixels, nv, d_pixel_flags);
	TP_LAUNCH_CHECK(ctx, "tp_threshold_flags_kernel");
	return TP_OK;
	TP_API_END(ctx)
}
