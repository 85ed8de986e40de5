// api.cpp -- context, memory, timers and the per-kernel profile of libtessphot_hip.so.
#include "common.h"
#include <algorithm>
#include <atomic>
#include <mutex>
#include <thread>
#include <cstring>
#include <cstdlib>
#include <exception>

// every live context, so that an allocation that fails can ask the others of its device for their cached blocks
static std::mutex tp_registry_mutex;
static std::vector<tp_ctx*> tp_registry;

thread_local std::string tp_global_err;

static const char* const kKernelNames[TPK_COUNT] = {
	"tp_sumimage_kernel",
	"tp_aperture_kernel",
	"tp_aperture_big_kernel",
	"tp_k2p2_kernel",
	"tp_aperture_fused_kernel",
	"tp_bkg_stamp_kernel",
	"tp_bkg_smooth_kernel",
	"tp_bkg_subtract_kernel",
	"tp_linpsf_prf_kernel",
	"tp_linpsf_fit_kernel",
	"tp_linpsf_fit_direct_kernel",
	"tp_linpsf_finalize_kernel",
	"tp_diagnostics_kernel",
	"tp_cut_stamps_kernel",
	"tp_psf_fit_kernel",
	"tp_bkg_mesh_kernel",
	"tp_bkg_zoom_kernel",
	"tp_median_filter_kernel",
	"tp_radial_kernels",
	"tp_linpsf_plan_kernel",
	"tp_linpsf_coef_kernel",
	"tp_synth_kernel",
	"tp_linpsf_fitm_kernel",
	"tp_bkg_stamp_sum_kernel",
	"tp_star_positions_kernel",
	"tp_f64_to_f32_kernel",
	"tp_blit_kernel",
};

extern "C" {

int tp_version(void) { return 100; } // 0.1.0

int tp_device_count(int* n) {
	if (!n) return TP_ERR_INVALID;
	int c = 0;
	hipError_t e = hipGetDeviceCount(&c);
	if (e != hipSuccess) {
		tp_global_err = std::string("hipGetDeviceCount: ") + hipGetErrorString(e);
		*n = 0;
		return TP_ERR_HIP;
	}
	*n = c;
	return TP_OK;
}

static int tp_ctx_create_impl(int device, int high_priority, tp_ctx** out) {
	if (!out) return TP_ERR_INVALID;
	*out = nullptr;
	TP_API_BEGIN
	int n = 0;
	hipError_t e = hipGetDeviceCount(&n);
	if (e != hipSuccess || n <= 0) {
		tp_global_err = std::string("no HIP device available: ") + hipGetErrorString(e);
		return TP_ERR_HIP;
	}
	if (device < 0 || device >= n) {
		tp_global_err = "device index out of range";
		return TP_ERR_INVALID;
	}
	e = hipSetDevice(device);
	if (e != hipSuccess) {
		tp_global_err = std::string("hipSetDevice: ") + hipGetErrorString(e);
		return TP_ERR_HIP;
	}
	hipDeviceProp_t prop;
	e = hipGetDeviceProperties(&prop, device);
	if (e != hipSuccess) {
		tp_global_err = std::string("hipGetDeviceProperties: ") + hipGetErrorString(e);
		return TP_ERR_HIP;
	}
	if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
		tp_global_err = std::string("libtessphot_hip is built for gfx950 only, device is ") + prop.gcnArchName;
		return TP_ERR_UNSUPPORTED;
	}
	tp_ctx* ctx = new tp_ctx();
	ctx->device = device;
	if (high_priority) {
		int least = 0, greatest = 0;
		(void)hipDeviceGetStreamPriorityRange(&least, &greatest);
		e = hipStreamCreateWithPriority(&ctx->stream, hipStreamNonBlocking, greatest);
	} else {
		e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
	}
	if (e != hipSuccess) {
		tp_global_err = std::string("hipStreamCreate: ") + hipGetErrorString(e);
		delete ctx;
		return TP_ERR_HIP;
	}
	for (int i = 0; i < 16; i++) {
		(void)hipEventCreate(&ctx->tstart[i]);
		(void)hipEventCreate(&ctx->tstop[i]);
	}
	{
		std::lock_guard<std::mutex> lk(tp_registry_mutex);
		tp_registry.push_back(ctx);
	}
	*out = ctx;
	return TP_OK;
	TP_API_END((tp_ctx*)nullptr)
}

int tp_comm_destroy(tp_ctx* ctx);
static void tp_cache_release(tp_ctx* ctx, bool own = true);


int tp_ctx_create(int device, tp_ctx** out) {
	return tp_ctx_create_impl(device, 0, out);
}

int tp_ctx_create_stream(int device, int high_priority, tp_ctx** out) {
	return tp_ctx_create_impl(device, high_priority, out);
}

int tp_ctx_destroy(tp_ctx* ctx) {
	if (!ctx) return TP_OK;
	{
		std::lock_guard<std::mutex> lk(tp_registry_mutex);
		tp_registry.erase(std::remove(tp_registry.begin(), tp_registry.end(), ctx), tp_registry.end());
	}
	(void)hipSetDevice(ctx->device);
	(void)hipStreamSynchronize(ctx->stream);
	(void)tp_comm_destroy(ctx);
	if (ctx->twiddle) (void)hipFree(ctx->twiddle);
	if (ctx->order) (void)hipFree(ctx->order);
	if (ctx->scratch) (void)hipFree(ctx->scratch);
	if (ctx->store) (void)hipFree(ctx->store);
	tp_cache_release(ctx);
	if (ctx->stage) (void)hipHostFree(ctx->stage);
	if (ctx->ring) (void)hipHostFree(ctx->ring);
	for (int k = 0; k < TPK_COUNT; k++)
		for (auto& p : ctx->pending[k]) {
			(void)hipEventDestroy(p.first);
			(void)hipEventDestroy(p.second);
		}
	for (auto e : ctx->pool) (void)hipEventDestroy(e);
	for (int i = 0; i < 16; i++) {
		(void)hipEventDestroy(ctx->tstart[i]);
		(void)hipEventDestroy(ctx->tstop[i]);
	}
	for (auto& st : ctx->side) if (st) (void)hipStreamDestroy(st);
	(void)hipStreamDestroy(ctx->stream);
	delete ctx;
	return TP_OK;
}

const char* tp_last_error(tp_ctx* ctx) {
	return ctx ? ctx->err.c_str() : tp_global_err.c_str();
}

int tp_device_info(tp_ctx* ctx, char* name, int name_len, int32_t* n_cu, uint64_t* hbm_bytes) {
	TP_CHECK_CTX(ctx);
	hipDeviceProp_t prop;
	TP_HIP(ctx, hipGetDeviceProperties(&prop, ctx->device));
	if (name && name_len > 0) {
		std::snprintf(name, (size_t)name_len, "%s (%s)", prop.name, prop.gcnArchName);
	}
	if (n_cu) *n_cu = prop.multiProcessorCount;
	if (hbm_bytes) *hbm_bytes = (uint64_t)prop.totalGlobalMem;
	return TP_OK;
}

// NUMA node of the device's PCIe slot (sysfs), -1 when unknown: the host threads that feed a GPU and the page-locked buffers it
// copies into are best kept on that node's cores (a process whose threads float over both sockets of a two-socket host ran the
// batched frames entry in 20 ms instead of 13, a third of its starts)
int tp_device_numa_node(int device, int* node) {
	if (!node) return TP_ERR_INVALID;
	*node = -1;
	char bus[64] = {0};
	hipError_t e = hipDeviceGetPCIBusId(bus, (int)sizeof(bus), device);
	if (e != hipSuccess) {
		tp_global_err = std::string("hipDeviceGetPCIBusId: ") + hipGetErrorString(e);
		return TP_ERR_HIP;
	}
	for (char* c = bus; *c; ++c) if (*c >= 'A' && *c <= 'F') *c = (char)(*c - 'A' + 'a');
	const std::string path = std::string("/sys/bus/pci/devices/") + bus + "/numa_node";
	if (FILE* f = std::fopen(path.c_str(), "r")) {
		int n = -1;
		if (std::fscanf(f, "%d", &n) == 1) *node = n;
		std::fclose(f);
	}
	return TP_OK;
}

// capacity classes of the allocation cache: powers of two from 4 KiB up to 64 KiB (the metadata blocks of a batched entry differ
// by a few bytes from group to group: with finer classes every one of them went to the driver, and a hipMalloc that has to map a
// new chunk takes milliseconds), then steps of 1/8 of the power of two below
static size_t tp_alloc_class(size_t n) {
	// up to 16 MiB: powers of two (the metadata block of a group of the frames engine is 0.1 - 2 MB and changes with the number of
	// catalogue stars in the batch: with steps of an eighth successive batches kept falling into classes the cache had not seen,
	// and a hipMalloc issued while four jobs have work queued returns after 8 - 9 ms -- measured, TESSPHOT_FRAMES_TIMING)
	if (n <= ((size_t)16 << 20)) { size_t p = 4096; while (p < n) p *= 2; return p; }
	size_t p = 65536;
	while (p * 2 <= n) p *= 2;
	const size_t step = p / 8;
	return (n + step - 1) / step * step;
}

// The caches of ALL contexts of a device together stay below a share of its memory (TESSPHOT_CACHE_FRACTION, default 0.6 of the
// device's total): a context's own limit (cache_limit) bounds one context, but a frames engine alone brings twenty contexts, and
// what they hold is invisible to allocations that do not go through tp_device_alloc (RCCL's buffers, another library, a second
// process on the device) -- those cannot ask for it back.
static std::atomic<size_t> tp_cache_total[64];
static size_t tp_cache_device_limit(int device) {
	static std::mutex m;
	static size_t limit[64] = {};
	if (device < 0 || device >= 64) return (size_t)-1;
	std::lock_guard<std::mutex> lk(m);
	if (limit[device] == 0) {
		double frac = 0.6;
		if (const char* e = std::getenv("TESSPHOT_CACHE_FRACTION")) { const double v = std::atof(e); if (v >= 0.0 && v <= 1.0) frac = v; }
		hipDeviceProp_t prop;
		size_t total = (size_t)64 << 30;
		if (hipGetDeviceProperties(&prop, device) == hipSuccess) total = prop.totalGlobalMem;
		(void)hipGetLastError();
		limit[device] = (size_t)std::max(1.0, frac * (double)total);
	}
	return limit[device];
}
static inline std::atomic<size_t>& tp_cache_total_of(const tp_ctx* ctx) { return tp_cache_total[(ctx->device >= 0 && ctx->device < 64) ? ctx->device : 0]; }

// every cached block back to the driver (their freeing events have to have completed: the stream is synchronised first)
static void tp_cache_release(tp_ctx* ctx, bool own) {
	std::lock_guard<std::recursive_mutex> lk(ctx->cache_mutex);
	if (ctx->cache.empty()) return;
	(void)hipStreamSynchronize(ctx->stream);
	for (auto& kv : ctx->cache) {
		(void)hipFree(kv.second.ptr);
		if (kv.second.freed) {
			// (a visitor does not touch the owner's event pool: that one is not behind the mutex)
			if (own) ctx->pool.push_back(kv.second.freed);
			else (void)hipEventDestroy(kv.second.freed);
		}
	}
	ctx->cache.clear();
	tp_cache_total_of(ctx) -= ctx->cache_bytes;
	ctx->cache_bytes = 0;
}

// the cached blocks of the OTHER contexts of ctx's device back to the driver; a context whose cache is in use right now
// (another thread inside tp_malloc / tp_free on it) is passed over rather than waited for
static bool tp_cache_release_others(tp_ctx* ctx) {
	bool any = false;
	std::lock_guard<std::mutex> lk(tp_registry_mutex);
	for (tp_ctx* other : tp_registry) {
		if (other == ctx || other->device != ctx->device) continue;
		std::unique_lock<std::recursive_mutex> ol(other->cache_mutex, std::try_to_lock);
		if (!ol.owns_lock() || other->cache.empty()) continue;
		tp_cache_release(other, false);
		any = true;
	}
	return any;
}

} // extern "C"
hipError_t tp_device_alloc(tp_ctx* ctx, void** ptr, size_t bytes) {
	hipError_t e = hipMalloc(ptr, bytes);
	if (e == hipErrorOutOfMemory && !ctx->cache.empty()) {
		tp_cache_release(ctx);
		(void)hipGetLastError();
		e = hipMalloc(ptr, bytes);
	}
	if (e == hipErrorOutOfMemory && tp_cache_release_others(ctx)) {
		(void)hipGetLastError();
		e = hipMalloc(ptr, bytes);
	}
	return e;
}
extern "C" {

int tp_cache_trim(tp_ctx* ctx) {
	TP_CHECK_CTX(ctx);
	tp_cache_release(ctx);
	return TP_OK;
}

int tp_malloc(tp_ctx* ctx, uint64_t nbytes, void** d_ptr) {
	TP_CHECK_CTX(ctx);
	TP_REQUIRE(ctx, d_ptr != nullptr, "tp_malloc: null output pointer");
	*d_ptr = nullptr;
	if (nbytes == 0) nbytes = 16;
	const size_t cap = tp_alloc_class((size_t)nbytes);
	std::lock_guard<std::recursive_mutex> lk(ctx->cache_mutex);
	auto range = ctx->cache.equal_range(cap);
	if (range.first != range.second) {
		// a block whose freeing event has completed is idle for every stream.  If none is: a small block is cheaper to get from
		// the driver than to wait for (tens of microseconds against the kernels still queued on the freed one); a large one waits
		// for the oldest (what was queued on the context's stream when it was freed: the wait the old hipFree paid at once)
		auto pick = range.first;
		bool idle = false;
		for (auto it = range.first; it != range.second; ++it)
			if (!it->second.freed || hipEventQuery(it->second.freed) == hipSuccess) { pick = it; idle = true; break; }
		(void)hipGetLastError();   // hipErrorNotReady of the queries
		constexpr size_t kWaitAbove = (size_t)16 << 20;
		if (idle || cap >= kWaitAbove || ctx->reuse_in_stream_order) {
			if (!idle && !ctx->reuse_in_stream_order) TP_HIP(ctx, hipEventSynchronize(pick->second.freed));
			*d_ptr = pick->second.ptr;
			if (pick->second.freed) ctx->pool.push_back(pick->second.freed);
			ctx->cache.erase(pick);
			ctx->cache_bytes -= cap;
			tp_cache_total_of(ctx) -= cap;
			ctx->live[*d_ptr] = cap;
			return TP_OK;
		}
	}
	const hipError_t e = tp_device_alloc(ctx, d_ptr, cap);
	if (e == hipErrorOutOfMemory) return ctx->fail(TP_ERR_NOMEM, "tp_malloc: out of device memory");
	if (e != hipSuccess) return ctx->fail(TP_ERR_HIP, "hipMalloc", e);
	ctx->live[*d_ptr] = cap;
	return TP_OK;
}

int tp_free(tp_ctx* ctx, void* d_ptr) {
	TP_CHECK_CTX(ctx);
	if (!d_ptr) return TP_OK;
	std::lock_guard<std::recursive_mutex> lk(ctx->cache_mutex);
	auto it = ctx->live.find(d_ptr);
	if (it != ctx->live.end()) {
		const size_t cap = it->second;
		ctx->live.erase(it);
		if (cap <= ctx->cache_block && ctx->cache_bytes + cap <= ctx->cache_limit && tp_cache_total_of(ctx).load() + cap <= tp_cache_device_limit(ctx->device)) {
			// kept for the next tp_malloc of this capacity; the event marks the end of what is queued on the context's stream for
			// the block (tp_malloc hands it out again once that has run)
			hipEvent_t ev = ctx->get_event();
			if (ev) (void)hipEventRecord(ev, ctx->stream);
			ctx->cache.emplace(cap, tp_ctx::cached_block{d_ptr, ev});
			ctx->cache_bytes += cap;
			tp_cache_total_of(ctx) += cap;
			return TP_OK;
		}
	}
	TP_HIP(ctx, hipStreamSynchronize(ctx->stream));
	TP_HIP(ctx, hipFree(d_ptr));
	return TP_OK;
}

int tp_memset(tp_ctx* ctx, void* d_ptr, int value, uint64_t nbytes) {
	TP_CHECK_CTX(ctx);
	if (nbytes == 0) return TP_OK;
	TP_REQUIRE(ctx, d_ptr != nullptr, "tp_memset: null pointer");
	TP_HIP(ctx, hipMemsetAsync(d_ptr, value, (size_t)nbytes, ctx->stream));
	return TP_OK;
}

// memcpy between the pinned staging area and pageable memory: one thread moves ~10 GB/s, a fifth of what the link delivers, so
// pieces of 4 MiB and more are split over four threads (the box gives a GPU 16 cores)
static void tp_host_copy(void* dst, const void* src, size_t n) {
	constexpr size_t kMin = (size_t)4 << 20;
	constexpr int kThreads = 4;
	if (n < kMin) { memcpy(dst, src, n); return; }
	const size_t part = ((n / kThreads) + 4095) & ~(size_t)4095;
	std::thread th[kThreads - 1];
	int started = 0;
	for (int i = 1; i < kThreads; ++i) {
		const size_t off = (size_t)i * part;
		if (off >= n) break;
		const size_t len = (off + part < n && i + 1 < kThreads) ? part : (n - off);
		th[started++] = std::thread([=] { memcpy(static_cast<char*>(dst) + off, static_cast<const char*>(src) + off, len); });
	}
	memcpy(dst, src, part < n ? part : n);
	for (int i = 0; i < started; ++i) th[i].join();
}

// the pinned staging area of the synchronous copies (two halves of 16 MiB: one is being filled / drained by the host while the
// other is in flight)
static int tp_stage(tp_ctx* ctx) {
	constexpr size_t kStage = (size_t)32 << 20;
	if (!ctx->stage) {
		TP_HIP(ctx, hipHostMalloc(&ctx->stage, kStage, hipHostMallocDefault));
		ctx->stage_bytes = kStage;
	}
	return TP_OK;
}

int tp_memcpy_h2d(tp_ctx* ctx, void* d_dst, const void* h_src, uint64_t nbytes) {
	TP_CHECK_CTX(ctx);
	if (nbytes == 0) return TP_OK;
	TP_REQUIRE(ctx, d_dst && h_src, "tp_memcpy_h2d: null pointer");
	constexpr size_t kRing = (size_t)4 << 20, kSmall = (size_t)256 << 10;
	if (nbytes <= kSmall) {
		if (!ctx->ring) { TP_HIP(ctx, hipHostMalloc(&ctx->ring, kRing, hipHostMallocDefault)); ctx->ring_cursor = 0; }
		const size_t n = ((size_t)nbytes + 63) & ~(size_t)63;
		if (ctx->ring_cursor + n > kRing) { TP_HIP(ctx, hipStreamSynchronize(ctx->stream)); ctx->ring_cursor = 0; }
		char* buf = static_cast<char*>(ctx->ring) + ctx->ring_cursor;
		memcpy(buf, h_src, (size_t)nbytes);
		TP_HIP(ctx, hipMemcpyAsync(d_dst, buf, (size_t)nbytes, hipMemcpyHostToDevice, ctx->stream));
		ctx->ring_cursor += n;
		return TP_OK;
	}
	// pageable host memory goes through the pinned staging area in pieces: the host copy of piece i + 1 overlaps the DMA of
	// piece i, and the call returns once h_src is consumed (the last DMA may still be in flight: it is ordered on the stream)
	int rc = tp_stage(ctx);
	if (rc != TP_OK) return rc;
	const size_t half = ctx->stage_bytes / 2;
	hipEvent_t ev[2] = {ctx->get_event(), ctx->get_event()};
	bool used[2] = {false, false};
	size_t off = 0;
	int b = 0;
	while (off < nbytes) {
		const size_t n = ((size_t)nbytes - off < half) ? ((size_t)nbytes - off) : half;
		char* buf = static_cast<char*>(ctx->stage) + b * half;
		if (used[b]) TP_HIP(ctx, hipEventSynchronize(ev[b]));
		tp_host_copy(buf, static_cast<const char*>(h_src) + off, n);
		TP_HIP(ctx, hipMemcpyAsync(static_cast<char*>(d_dst) + off, buf, n, hipMemcpyHostToDevice, ctx->stream));
		TP_HIP(ctx, hipEventRecord(ev[b], ctx->stream));
		used[b] = true;
		off += n;
		b ^= 1;
	}
	// the staging halves are reused by the next call: wait for the DMAs that read them (not for the rest of the stream's
	// work ahead of them -- events, not a stream synchronisation, would be enough, but the copies ARE the tail of the stream)
	for (int i = 0; i < 2; i++) if (used[i]) TP_HIP(ctx, hipEventSynchronize(ev[i]));
	ctx->pool.push_back(ev[0]);
	ctx->pool.push_back(ev[1]);
	return TP_OK;
}

int tp_memcpy_d2h(tp_ctx* ctx, void* h_dst, const void* d_src, uint64_t nbytes) {
	TP_CHECK_CTX(ctx);
	if (nbytes == 0) return TP_OK;
	TP_REQUIRE(ctx, h_dst && d_src, "tp_memcpy_d2h: null pointer");
	int rc = tp_stage(ctx);
	if (rc != TP_OK) return rc;
	const size_t half = ctx->stage_bytes / 2;
	hipEvent_t ev[2] = {ctx->get_event(), ctx->get_event()};
	// piece i + 1 is on its way into one half of the pinned area while the host copies piece i out of the other
	size_t issued = 0, drained = 0;
	size_t len[2] = {0, 0};
	int bi = 0, bd = 0;
	while (drained < nbytes) {
		while (issued < nbytes && len[bi] == 0) {
			const size_t n = ((size_t)nbytes - issued < half) ? ((size_t)nbytes - issued) : half;
			TP_HIP(ctx, hipMemcpyAsync(static_cast<char*>(ctx->stage) + bi * half, static_cast<const char*>(d_src) + issued, n, hipMemcpyDeviceToHost, ctx->stream));
			TP_HIP(ctx, hipEventRecord(ev[bi], ctx->stream));
			len[bi] = n;
			issued += n;
			bi ^= 1;
		}
		TP_HIP(ctx, hipEventSynchronize(ev[bd]));
		tp_host_copy(static_cast<char*>(h_dst) + drained, static_cast<char*>(ctx->stage) + bd * half, len[bd]);
		drained += len[bd];
		len[bd] = 0;
		bd ^= 1;
	}
	ctx->pool.push_back(ev[0]);
	ctx->pool.push_back(ev[1]);
	return TP_OK;
}

int tp_memcpy_d2d(tp_ctx* ctx, void* d_dst, const void* d_src, uint64_t nbytes) {
	TP_CHECK_CTX(ctx);
	if (nbytes == 0) return TP_OK;
	TP_REQUIRE(ctx, d_dst && d_src, "tp_memcpy_d2d: null pointer");
	TP_HIP(ctx, hipMemcpyAsync(d_dst, d_src, (size_t)nbytes, hipMemcpyDeviceToDevice, ctx->stream));
	return TP_OK;
}

int tp_upload_cube(tp_ctx* ctx, float* d_dst, int64_t dst_pitch, const float* h_src, int64_t src_pitch,
	int64_t n_rows, int64_t n_cad) {
	TP_CHECK_CTX(ctx);
	if (n_rows == 0 || n_cad == 0) return TP_OK;
	TP_REQUIRE(ctx, d_dst && h_src, "tp_upload_cube: null pointer");
	TP_REQUIRE(ctx, dst_pitch >= n_cad && src_pitch >= n_cad && n_rows > 0 && n_cad > 0, "tp_upload_cube: bad geometry");
	if (dst_pitch == n_cad && src_pitch == n_cad) {
		TP_HIP(ctx, hipMemcpyAsync(d_dst, h_src, (size_t)(n_rows * n_cad) * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
	} else {
		TP_HIP(ctx, hipMemcpy2DAsync(d_dst, (size_t)dst_pitch * sizeof(float), h_src, (size_t)src_pitch * sizeof(float),
			(size_t)n_cad * sizeof(float), (size_t)n_rows, hipMemcpyHostToDevice, ctx->stream));
	}
	TP_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return TP_OK;
}

int tp_host_alloc(tp_ctx* ctx, uint64_t nbytes, void** h_ptr) {
	TP_CHECK_CTX(ctx);
	TP_REQUIRE(ctx, h_ptr != nullptr, "tp_host_alloc: null output pointer");
	*h_ptr = nullptr;
	if (nbytes == 0) nbytes = 16;
	hipError_t e = hipHostMalloc(h_ptr, (size_t)nbytes, hipHostMallocDefault);
	if (e == hipErrorOutOfMemory) return ctx->fail(TP_ERR_NOMEM, "tp_host_alloc: out of pinned host memory");
	if (e != hipSuccess) return ctx->fail(TP_ERR_HIP, "hipHostMalloc", e);
	return TP_OK;
}

int tp_host_free(tp_ctx* ctx, void* h_ptr) {
	TP_CHECK_CTX(ctx);
	if (!h_ptr) return TP_OK;
	TP_HIP(ctx, hipHostFree(h_ptr));
	return TP_OK;
}

int tp_upload_cube_async(tp_ctx* ctx, float* d_dst, int64_t dst_pitch, const float* h_src, int64_t src_pitch,
	int64_t n_rows, int64_t n_cad) {
	TP_CHECK_CTX(ctx);
	if (n_rows == 0 || n_cad == 0) return TP_OK;
	TP_REQUIRE(ctx, d_dst && h_src, "tp_upload_cube_async: null pointer");
	TP_REQUIRE(ctx, dst_pitch >= n_cad && src_pitch >= n_cad && n_rows > 0 && n_cad > 0, "tp_upload_cube_async: bad geometry");
	if (dst_pitch == n_cad && src_pitch == n_cad) {
		TP_HIP(ctx, hipMemcpyAsync(d_dst, h_src, (size_t)(n_rows * n_cad) * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
	} else {
		TP_HIP(ctx, hipMemcpy2DAsync(d_dst, (size_t)dst_pitch * sizeof(float), h_src, (size_t)src_pitch * sizeof(float),
			(size_t)n_cad * sizeof(float), (size_t)n_rows, hipMemcpyHostToDevice, ctx->stream));
	}
	return TP_OK;
}

int tp_memcpy_d2h_async(tp_ctx* ctx, void* h_dst, const void* d_src, uint64_t nbytes) {
	TP_CHECK_CTX(ctx);
	if (nbytes == 0) return TP_OK;
	TP_REQUIRE(ctx, h_dst && d_src, "tp_memcpy_d2h_async: null pointer");
	TP_HIP(ctx, hipMemcpyAsync(h_dst, d_src, (size_t)nbytes, hipMemcpyDeviceToHost, ctx->stream));
	return TP_OK;
}

int tp_sync(tp_ctx* ctx) {
	TP_CHECK_CTX(ctx);
	TP_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return TP_OK;
}

int tp_event_create(tp_ctx* ctx, void** event) {
	TP_CHECK_CTX(ctx);
	TP_REQUIRE(ctx, event != nullptr, "tp_event_create: null output pointer");
	hipEvent_t e = nullptr;
	TP_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
	*event = (void*)e;
	return TP_OK;
}

int tp_event_destroy(tp_ctx* ctx, void* event) {
	TP_CHECK_CTX(ctx);
	if (event) TP_HIP(ctx, hipEventDestroy((hipEvent_t)event));
	return TP_OK;
}

int tp_event_record(tp_ctx* ctx, void* event) {
	TP_CHECK_CTX(ctx);
	TP_REQUIRE(ctx, event != nullptr, "tp_event_record: null event");
	TP_HIP(ctx, hipEventRecord((hipEvent_t)event, ctx->stream));
	return TP_OK;
}

int tp_event_sync(tp_ctx* ctx, void* event) {
	TP_CHECK_CTX(ctx);
	TP_REQUIRE(ctx, event != nullptr, "tp_event_sync: null event");
	TP_HIP(ctx, hipEventSynchronize((hipEvent_t)event));
	return TP_OK;
}

int tp_stream_wait_event(tp_ctx* ctx, void* event) {
	TP_CHECK_CTX(ctx);
	TP_REQUIRE(ctx, event != nullptr, "tp_stream_wait_event: null event");
	TP_HIP(ctx, hipStreamWaitEvent(ctx->stream, (hipEvent_t)event, 0));
	return TP_OK;
}

int tp_timer_start(tp_ctx* ctx, int slot) {
	TP_CHECK_CTX(ctx);
	TP_REQUIRE(ctx, slot >= 0 && slot < 16, "timer slot out of range");
	TP_HIP(ctx, hipEventRecord(ctx->tstart[slot], ctx->stream));
	return TP_OK;
}

int tp_timer_stop(tp_ctx* ctx, int slot) {
	TP_CHECK_CTX(ctx);
	TP_REQUIRE(ctx, slot >= 0 && slot < 16, "timer slot out of range");
	TP_HIP(ctx, hipEventRecord(ctx->tstop[slot], ctx->stream));
	return TP_OK;
}

int tp_timer_elapsed_ms(tp_ctx* ctx, int slot, float* ms) {
	TP_CHECK_CTX(ctx);
	TP_REQUIRE(ctx, slot >= 0 && slot < 16 && ms, "timer slot out of range");
	TP_HIP(ctx, hipEventSynchronize(ctx->tstop[slot]));
	TP_HIP(ctx, hipEventElapsedTime(ms, ctx->tstart[slot], ctx->tstop[slot]));
	return TP_OK;
}

int tp_profile_enable(tp_ctx* ctx, int on) {
	TP_CHECK_CTX(ctx);
	ctx->profile = (on != 0);
	return TP_OK;
}

static int tp_profile_drain(tp_ctx* ctx) {
	TP_HIP(ctx, hipStreamSynchronize(ctx->stream));
	for (int k = 0; k < TPK_COUNT; k++) {
		for (auto& p : ctx->pending[k]) {
			float ms = 0.f;
			if (hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess) {
				ctx->prof_ms[k] += (double)ms;
				ctx->prof_n[k] += 1;
			}
			ctx->pool.push_back(p.first);
			ctx->pool.push_back(p.second);
		}
		ctx->pending[k].clear();
	}
	return TP_OK;
}

int tp_profile_reset(tp_ctx* ctx) {
	TP_CHECK_CTX(ctx);
	int rc = tp_profile_drain(ctx);
	if (rc != TP_OK) return rc;
	for (int k = 0; k < TPK_COUNT; k++) {
		ctx->prof_ms[k] = 0.0;
		ctx->prof_n[k] = 0;
	}
	return TP_OK;
}

int tp_kernel_count(void) { return TPK_COUNT; }

const char* tp_kernel_name(int kernel_id) {
	if (kernel_id < 0 || kernel_id >= TPK_COUNT) return "";
	return kKernelNames[kernel_id];
}

int tp_profile_get(tp_ctx* ctx, int kernel_id, int64_t* n_launches, double* total_ms) {
	TP_CHECK_CTX(ctx);
	TP_REQUIRE(ctx, kernel_id >= 0 && kernel_id < TPK_COUNT, "kernel id out of range");
	int rc = tp_profile_drain(ctx);
	if (rc != TP_OK) return rc;
	if (n_launches) *n_launches = ctx->prof_n[kernel_id];
	if (total_ms) *total_ms = ctx->prof_ms[kernel_id];
	return TP_OK;
}

} // extern "C"
