// common.h -- context, error handling, launch/profile helpers shared by every translation unit
// of libtessphot_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>
#include <utility>
#include <map>
#include <mutex>
#include "../../include/tessphot_hip.h"

// Dense kernel ids for the per-kernel profile (tp_kernel_name / tp_profile_get).
enum tp_kernel_id {
	TPK_SUMIMAGE = 0,
	TPK_APERTURE,
	TPK_APERTURE_BIG,
	TPK_K2P2,
	TPK_FUSED,
	TPK_BKG_STAMP,
	TPK_BKG_SMOOTH,
	TPK_BKG_SUBTRACT,
	TPK_LINPSF_PRF,
	TPK_LINPSF_FIT,
	TPK_LINPSF_FIT_DIRECT,
	TPK_LINPSF_FIN,
	TPK_DIAGNOSTICS,
	TPK_CUTOUT,
	TPK_PSF_FIT,
	TPK_BKG_MESH,
	TPK_BKG_ZOOM,
	TPK_MEDIAN_FILTER,
	TPK_BKG_RADIAL,
	TPK_LINPSF_PLAN,
	TPK_LINPSF_COEF,
	TPK_SYNTH,
	TPK_LINPSF_FIT_MFMA,
	TPK_BKG_STAMP_SUM,
	TPK_STAR_POSITIONS,
	TPK_BLOCK_COMPACT,
	TPK_BLIT,
	TPK_COUNT
};

struct tp_ctx {
	int device = 0;
	hipStream_t stream = nullptr;
	hipStream_t side[2] = {nullptr, nullptr};   // created on demand: independent launches of one entry whose tails may overlap (tp_linpsf_fit)
	std::string err;
	hipEvent_t tstart[16] = {};
	hipEvent_t tstop[16] = {};
	bool profile = false;
	// pending (start, stop) event pairs per kernel id + accumulated totals
	std::vector<std::pair<hipEvent_t, hipEvent_t>> pending[TPK_COUNT];
	std::vector<hipEvent_t> pool;
	int64_t prof_n[TPK_COUNT] = {};
	double prof_ms[TPK_COUNT] = {};
	void* twiddle = nullptr;    // device table of the K2P2 128-point DFT (k2p2.hip)
	// launch order of the fused aperture kernel (fused.hip): the targets of a batch brightest first, cached per (tmag array, size)
	const void* order_key = nullptr; int order_n = 0; int32_t* order = nullptr;
	void* scratch = nullptr;    // grow-only device scratch owned by the context (linpsf.hip)
	size_t scratch_bytes = 0;
	void* store = nullptr;      // second grow-only buffer: the polynomial coefficient store of tp_linpsf_fit
	size_t store_bytes = 0;
	// tp_malloc / tp_free: blocks a caller frees go to a size-keyed cache instead of back to the driver (hipFree synchronises the
	// whole device, hipMalloc costs tens of microseconds -- and ~12 ms for a multi-GB block: the stamp cubes of the batched frames
	// entry, measured -- a batch of 10 000 stamps of 15 x 15 is three blocks of 11.8 GB, and outside the cache every call paid 1.2 s
	// for them); reuse is ordered by the context's stream.  Blocks up to cache_block (32 GiB), cache_limit (160 GiB) per context (a batch of
	// 20 000 stamps is three blocks of 23.6 GB: with 64 GiB the third one went back to the driver every call, 1.2 s instead of 0.08).
	// When an allocation fails the context's own cache goes back to the driver first, then the caches of every other context of
	// the device (api.cpp keeps a registry; cache_mutex serialises a context's cache against such a visit from another thread).
	// A cached block carries an event recorded on the context's stream when it was freed; tp_malloc hands it out again only once
	// that event has completed (it prefers a block whose event already has, and waits otherwise), so a recycled block is idle
	// whichever stream or context writes to it next.  tp_device_alloc (scratch, stores, work lists) and tp_malloc give the cache
	// back to the driver when an allocation fails; tp_cache_trim does so on request.
	struct cached_block { void* ptr; hipEvent_t freed; };
	std::multimap<size_t, cached_block> cache;
	std::map<void*, size_t> live;   // blocks handed out by tp_malloc -> capacity
	size_t cache_bytes = 0, cache_limit = (size_t)160 << 30, cache_block = (size_t)32 << 30;
	std::recursive_mutex cache_mutex;
	// A context whose blocks are only ever used on its OWN stream (the contexts of the frames engine: one per stream of a job) may take
	// a cached block back while the work queued on it when it was freed is still running: the stream orders the new use behind it.
	// Without this a tp_malloc with no idle block of the size either waits on the host for the freeing event or goes to the driver --
	// with four jobs in flight both stall the worker thread for milliseconds (measured: queueing a job's groups 0.5 ms alone, 2 - 17 ms
	// with four jobs in flight).
	bool reuse_in_stream_order = false;
	// pinned staging area of the synchronous copy entries (tp_memcpy_h2d / _d2h): pageable transfers go through it in pieces
	void* stage = nullptr;
	size_t stage_bytes = 0;
	// small uploads (<= 256 KiB) go through a pinned ring and return without waiting for their DMA (the ring is reused only
	// after the stream has been synchronised): a batched entry makes a hundred of them per call
	void* ring = nullptr;
	size_t ring_cursor = 0;
	int linpsf_path = 1;        // tp_linpsf_set_path: 1 = matrix-core fit where a target qualifies, 0 = vector-ALU fit kernels only
	int64_t linpsf_counts[16] = {};   // tp_linpsf_last_counts: which kernels fitted the targets of the last tp_linpsf_fit call
	void* comm = nullptr;       // ncclComm_t (comm.cpp)
	int comm_rank = 0, comm_size = 1;

	int fail(int code, const char* what, hipError_t e = hipSuccess) {
		err = what;
		if (e != hipSuccess) {
			err += ": ";
			err += hipGetErrorString(e);
		}
		return code;
	}
	hipEvent_t get_event() {
		if (!pool.empty()) {
			hipEvent_t e = pool.back();
			pool.pop_back();
			return e;
		}
		hipEvent_t e = nullptr;
		(void)hipEventCreate(&e);
		return e;
	}
};

extern thread_local std::string tp_global_err;

// hipMalloc for the library's own buffers (context scratch, coefficient store, work lists): when the driver is out of memory
// the blocks idling in the tp_malloc cache are given back first (api.cpp)
hipError_t tp_device_alloc(tp_ctx* ctx, void** ptr, size_t bytes);

#define TP_HIP(ctx, call) do { hipError_t _e = (call); if (_e != hipSuccess) return (ctx)->fail(TP_ERR_HIP, #call, _e); } while (0)
#define TP_REQUIRE(ctx, cond, msg) do { if (!(cond)) return (ctx)->fail(TP_ERR_INVALID, msg); } while (0)
#define TP_CHECK_CTX(ctx) do { if ((ctx) == nullptr) { tp_global_err = "null ctx"; return TP_ERR_INVALID; } (void)hipSetDevice((ctx)->device); } while (0)

// Brackets one kernel launch with events when profiling is on.
struct tp_prof_scope {
	tp_ctx* ctx;
	int kid;
	hipStream_t st;
	hipEvent_t e0 = nullptr, e1 = nullptr;
	tp_prof_scope(tp_ctx* c, int k, hipStream_t s = nullptr) : ctx(c), kid(k), st(s ? s : c->stream) {
		if (ctx->profile) {
			e0 = ctx->get_event();
			e1 = ctx->get_event();
			(void)hipEventRecord(e0, st);
		}
	}
	~tp_prof_scope() {
		if (ctx->profile) {
			(void)hipEventRecord(e1, st);
			ctx->pending[kid].emplace_back(e0, e1);
		}
	}
};

#define TP_LAUNCH(ctx, kid, kernel, grid, block, shmem, ...) do { \
	tp_prof_scope _ps((ctx), (kid)); \
	hipLaunchKernelGGL(kernel, grid, block, shmem, (ctx)->stream, __VA_ARGS__); \
} while (0)
// the same on another stream of the context (the caller orders it against ctx->stream with events)
#define TP_LAUNCH_ON(ctx, st, kid, kernel, grid, block, shmem, ...) do { \
	tp_prof_scope _ps((ctx), (kid), (st)); \
	hipLaunchKernelGGL(kernel, grid, block, shmem, (st), __VA_ARGS__); \
} while (0)

#define TP_LAUNCH_CHECK(ctx, name) do { hipError_t _e = hipGetLastError(); if (_e != hipSuccess) return (ctx)->fail(TP_ERR_HIP, name, _e); } while (0)

static inline bool tp_desc_ok(const tp_cube_desc* d) {
	return d && d->n_targets >= 0 && d->n_cad >= 0 && d->height > 0 && d->width > 0 && d->t_pitch >= d->n_cad;
}

// 16-byte aligned base and pitch % 4 == 0 -> 128-bit load path
static inline bool tp_vec4_ok(const void* p, int64_t pitch) {
	return ((reinterpret_cast<uintptr_t>(p) & 15u) == 0) && (pitch % 4 == 0);
}

// internal (sumimage.hip): nbytes from src to dst by a kernel on the context's stream -- either side may be page-locked host memory.
// The frames engine moves a group's metadata and the data of its decisions this way: an asynchronous copy goes through a DMA engine
// that streams share, and a small copy queued behind a large one that is still waiting for its kernels waits with it.
int tp_blit(tp_ctx* ctx, void* dst, const void* src, uint64_t nbytes);

#define TP_API_BEGIN try {
#define TP_API_END(ctx) } catch (const std::exception& ex) { if (ctx) { (ctx)->err = ex.what(); } else { tp_global_err = ex.what(); } return TP_ERR_INVALID; } catch (...) { if (ctx) { (ctx)->err = "unknown C++ exception"; } return TP_ERR_INVALID; }
