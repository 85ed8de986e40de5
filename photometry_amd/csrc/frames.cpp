// frames.cpp -- the batched drop-in entry as a native host engine: AperturePhotometry.do_photometry INCLUDING its stamp-resize loop
// (photometry/AperturePhotometry/photometry.py:75-170, BasePhotometry.resize_stamp / _set_stamp BasePhotometry.py:567-693) for every
// target of a CCD region whose frame stacks are resident in HBM, from the target list to columnar results.
//
// What the reference does one target at a time in Python -- cut the stamp out of the HDF5 groups, run the plugin, look at the mask,
// grow the stamp, try again -- is here a JOB: the host submits a batch (tp_frames_submit) and collects it (tp_frames_wait); in
// between a worker thread of the library drives the rounds: group the targets still in play by stamp size, select the catalogue
// stars of every stamp from a cell-binned index, and queue each group's pass on a stream of the engine's pool -- first the halves
// that produce the masks of ALL groups of the round (sum images cropped from the region's, tp_k2p2_masks, the download of what the
// decisions read, an event), then their second halves (extraction, light-curve diagnostics, download of the packed output block
// into page-locked memory).  With the region's TIME-MAJOR stacks at hand (tp_frames_stack.d_images_t ...) nothing is cut: the
// extraction reads a mask pixel's series as one row of the stack (tp_aperture_extract_stack); without them the in-mask rows of
// the stamps are cut per pass (tp_cut_stamps_masked).  The worker then decides with the plugin's rules who is finished, who gets a
// bigger stamp and who gives up.  Small transfers go by a kernel (tp_blit), only the large group's light curves through a DMA
// engine, chunk by chunk on the job's copy stream while the later chunks are still being extracted.  No Python runs between
// submit and collect, so several jobs in flight (one per engine slot) keep the device busy: the first round of one batch runs
// under the latency-bound resize rounds of another.  The rules are those of photometry_amd/stamps.py and plugins.mask_outcome
// (which stay the per-target plugin's implementation and the reference of tests/test_gpu_resize.py); messages travel as codes
// that the Python layer turns into the reference's log strings.
#include "common.h"
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <stdexcept>
#include <thread>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <vector>

namespace {

constexpr uint32_t kBitmask = 1 | 2 | 4 | 8 | 32 | 64 | 128 | 4096;   // TESSQualityFlags.DEFAULT_BITMASK (quality.py:123-124)
// Streams per job: stream 0 for the large groups, three for the small, latency-bound ones.  Four active streams is what a single
// job runs fastest with (measured, 2 500 targets: 15 / 13 / 20 ms with 2 / 3 / 4 streams for the small groups: beyond four active
// queues of a process the hardware time-slices them).
constexpr int kStreams = 4;               // per slot: three streams of the engine's pool and a copy stream
static int g_small_streams = 3;          // (experiment: TESSPHOT_FRAMES_STREAMS = 1 .. 3 pool streams per slot)
constexpr int kResizeStep = 10;          // photometry.py:124-131
constexpr int kFusedFrom = 1024;         // from this many targets on a group is "large": stream 0, the error / background stacks cut after the mask
constexpr int kEdgeBits = 2 | 4 | 8 | 16;

inline int64_t round_up(int64_t n, int64_t m) { return (n + m - 1) / m * m; }

struct Fail : std::runtime_error { using std::runtime_error::runtime_error; };
inline void ck(tp_ctx* g, int rc) { if (rc != TP_OK) throw Fail(g->err.empty() ? std::string("error ") + std::to_string(rc) : g->err); }
inline void ckh(hipError_t e, const char* what) { if (e != hipSuccess) throw Fail(std::string(what) + ": " + hipGetErrorString(e)); }

// ---- page-locked host memory, pooled by size class (hipHostMalloc takes milliseconds) ---------------------------------------
struct PinnedPool {
	std::mutex m;
	std::multimap<size_t, void*> free_blocks;
	static size_t size_class(size_t n) {
		size_t p = 65536;
		while (p < n && p < ((size_t)1 << 20)) p *= 2;
		if (n <= p) return p;
		p = (size_t)1 << 20;
		while (p * 2 <= n) p *= 2;
		const size_t step = p / 8;
		return (n + step - 1) / step * step;
	}
	void* get(size_t n, size_t* cap) {
		const size_t c = size_class(n ? n : 16);
		{
			std::lock_guard<std::mutex> lk(m);
			auto it = free_blocks.find(c);
			if (it != free_blocks.end()) { void* p = it->second; free_blocks.erase(it); *cap = c; return p; }
		}
		void* p = nullptr;
		ckh(hipHostMalloc(&p, c, hipHostMallocDefault), "hipHostMalloc");
		*cap = c;
		return p;
	}
	void put(void* p, size_t cap) {
		if (!p) return;
		std::lock_guard<std::mutex> lk(m);
		free_blocks.emplace(cap, p);
	}
	~PinnedPool() { for (auto& kv : free_blocks) (void)hipHostFree(kv.second); }
};

// numpy's pairwise summation of a contiguous float64 vector (np.add.reduce): what np.nansum does after replacing the NaNs
double np_pairwise_sum(const double* a, int64_t n) {
	if (n < 8) { double r = 0.0; for (int64_t i = 0; i < n; ++i) r += a[i]; return r; }
	if (n <= 128) {
		double r[8];
		for (int j = 0; j < 8; ++j) r[j] = a[j];
		int64_t i = 8;
		for (; i < n - (n % 8); i += 8) for (int j = 0; j < 8; ++j) r[j] += a[i + j];
		double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
		for (; i < n; ++i) res += a[i];
		return res;
	}
	int64_t n2 = n / 2;
	n2 -= n2 % 8;
	return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
}

} // namespace

// A few helper threads of the engine, started once: the catalogue selection of a large group is cut into runs of stamps and the
// runs are selected side by side (threads started per group cost more than they saved: 2.3 ms against 0.75 for 2 500 stamps).
struct HelperPool {
	std::mutex m;
	std::condition_variable cv;
	std::deque<std::function<void()>> tasks;
	std::vector<std::thread> threads;
	bool stop = false;
	void start(int n) {
		for (int i = 0; i < n; ++i) {
			try {
				threads.emplace_back([this] {
					for (;;) {
						std::function<void()> f;
						{
							std::unique_lock<std::mutex> lk(m);
							cv.wait(lk, [this] { return stop || !tasks.empty(); });
							if (tasks.empty()) return;      // (stop, and nothing left to do)
							f = std::move(tasks.front());
							tasks.pop_front();
						}
						f();
					}
				});
			} catch (...) { break; }                   // fewer helpers, or none: the callers run what nobody takes
		}
	}
	void post(std::function<void()> f) { { std::lock_guard<std::mutex> lk(m); tasks.push_back(std::move(f)); } cv.notify_one(); }
	// a posted task that no helper has taken yet, for the poster to run itself rather than wait
	bool take(std::function<void()>& f) {
		std::lock_guard<std::mutex> lk(m);
		if (tasks.empty()) return false;
		f = std::move(tasks.front());
		tasks.pop_front();
		return true;
	}
	~HelperPool() {
		{ std::lock_guard<std::mutex> lk(m); stop = true; }
		cv.notify_all();
		for (auto& t : threads) if (t.joinable()) t.join();
	}
};

// ---- the catalogue of a region, binned into cells of 16 x 16 pixels (stars sorted by cell) ------------------------------------
struct tp_frames_catalog {
	int64_t n = 0;
	std::vector<int64_t> starid;
	std::vector<float> tmag;
	std::vector<double> row, col;
	int64_t cell = 16, r0 = 0, c0 = 0, n_cr = 1, n_cc = 1;
	std::vector<int64_t> order, cell_start;
};

struct tp_frames_engine {
	int device = 0;
	int n_slots = 0;
	std::vector<tp_ctx*> ctxs;          // kStreams - 1 per slot: the engine's pool of streams, shared by the jobs in flight (SmallStream
	                                    // below); the kStreams-th stream of a slot is its copy stream.  The process should stay below
	                                    // ~24 streams in all: beyond that the hardware queues are time-sliced, and with six idle streams
	                                    // more in the process four jobs in flight fell from 8.1 to 4.8 x 10^5 targets/s (round 6)
	// A stream of the pool.  A job's worker CLAIMS one per group of a round while it queues the round (nobody else queues
	// on a claimed stream: a context's host-side state has one user at a time), marks it with an event when it lets go, and the next
	// claimant -- of any job -- prefers a stream whose event has completed (idle), in index order (so that the same few contexts are
	// used and their allocation caches stay warm), else the one with the least work queued since it was last seen idle.
	struct SmallStream { tp_ctx* c = nullptr; hipEvent_t busy = nullptr; double load = 0.0; bool claimed = false; };
	std::vector<SmallStream> small;
	std::mutex sm;
	std::vector<hipStream_t> copy_streams;   // one per slot (nullptr if it could not be created: the light curves then leave on the job's stream)
	std::vector<char> busy;
	std::mutex m;
	PinnedPool pinned;
	HelperPool helpers;
	uint64_t hbm_bytes = 0;
	std::atomic<int> running{0};         // worker threads inside run(): tp_frames_engine_destroy waits for them
};

namespace {

struct Event { int32_t target, code, a, b; double value; int32_t text; };

struct Group {
	int32_t n = 0, H = 0, W = 0;
	int64_t n_cat = 0, cat_capacity = 1;
	std::vector<int64_t> cat_offsets, cat_starid, target_starid;
	void* h_block = nullptr; size_t h_cap = 0; uint64_t nbytes = 0;
	// offsets of the fields of the packed block (comm.packed_block_layout(n, T, H, W, n_cat = cat_capacity, extras = True))
	uint64_t off_lc = 0, off_cont = 0, off_status = 0, off_flags = 0, off_mask = 0, off_cim = 0, off_sum = 0, off_diag = 0;
};

// a group of one round while its pass is in flight
struct Launched {
	tp_ctx* g = nullptr;
	int small = -1;                       // index of the claimed stream of the engine's pool
	bool chunked = false;                 // its light curves leave on the job's copy stream
	std::vector<int32_t> idx;
	Group grp;
	hipEvent_t ev = nullptr;              // the decisions' data have arrived
	bool failed = false;
	std::string error;
	// what the first half of the pass (metadata, masks, the decisions' download) leaves for the second (cut, extraction, diagnostics)
	std::vector<void*> dev;               // device blocks of this group: freed (stream-ordered) once everything is queued
	tp_cube_desc desc{};
	float* cubes[3] = {nullptr, nullptr, nullptr};
	char* blk = nullptr;
	const int32_t* d_stamps = nullptr; const int32_t* d_quality = nullptr; const double* d_time = nullptr;
	bool crop = false, large = false, time_major = false;
};

} // namespace

struct tp_frames_job {
	tp_frames_engine* eng = nullptr;
	int slot = -1;
	hipStream_t copy_stream = nullptr;        // the slot's copy stream: the light curves of that group, chunk by chunk
	std::vector<hipEvent_t> tails;            // one per group: everything the group queued (its light curves last) has run
	std::vector<std::pair<tp_ctx*, void*>> late_frees;   // output blocks still read by the copy stream when their group was queued
	tp_frames_stack stack{};
	const tp_frames_catalog* cat = nullptr;
	int32_t n = 0, T = 0;
	std::vector<int64_t> starid;
	std::vector<double> tmag, row, col, budget_flux, time;
	std::vector<int64_t> cur;                 // [n][4]
	std::vector<uint8_t> valid;
	std::vector<int32_t> attempts, quality;
	double budget = 0.0;
	// results
	std::vector<int32_t> status, stamp_resizes, group, pos;
	std::vector<uint8_t> has_result;
	std::vector<int64_t> stamp;               // [n][4]
	std::vector<Group> groups;
	std::vector<Event> events;
	std::vector<std::string> texts;
	std::map<int32_t, std::vector<Event>> pending;   // logged, not yet flushed into `events` (plugins._Messages of a target)
	std::vector<std::pair<void*, size_t>> host_scratch;   // pinned metadata blocks: back to the pool when the job is done
	std::thread worker;
	// lab (TESSPHOT_FRAMES_TIMING=1): where the worker thread's time goes, microseconds -- [0] catalogue selection + metadata block,
	// [1] queueing a group's device work, [2] waiting for the decisions of a round, [3] deciding, [4] waiting for the last light curves
	double lab_us[5] = {0, 0, 0, 0, 0};
	int lab_groups = 0, lab_rounds = 0;
	std::map<std::string, double> lab_steps;   // the queueing of a group, step by step
	std::string lab_timeline;                  // when the rounds were queued and decided, microseconds from the start of run()
	std::chrono::steady_clock::time_point lab_run0;
	void lab_mark(const char* what, int a, int b) {
		char buf[96];
		std::snprintf(buf, sizeof buf, " %s%d/%d@%.0f", what, a, b, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - lab_run0).count());
		lab_timeline += buf;
	}
	int rc = TP_OK;
	std::string err;
	bool joined = false, released = false;
	std::atomic<int> done{0};               // set by the worker when run() has returned (tp_frames_poll)

	void log(int32_t i, int32_t code, int32_t a = 0, int32_t b = 0, double v = 0.0, int32_t text = -1) { pending[i].push_back(Event{i, code, a, b, v, text}); }
	void direct(int32_t i, int32_t code, int32_t a = 0, int32_t b = 0, double v = 0.0, int32_t text = -1) { events.push_back(Event{i, code, a, b, v, text}); }
	void flush(int32_t i) {
		auto it = pending.find(i);
		if (it == pending.end()) return;
		for (auto& e : it->second) events.push_back(e);
		pending.erase(it);
	}
	void finish(int32_t i, int32_t st) {
		status[i] = st;
		for (int k = 0; k < 4; ++k) stamp[(size_t)i * 4 + k] = cur[(size_t)i * 4 + k];
		flush(i);
	}
	int32_t add_text(const std::string& s) { texts.push_back(s); return (int32_t)texts.size() - 1; }
	void run();
	void select_catalog(const std::vector<int32_t>& idx, Group& g, std::vector<float>& c_tmag, std::vector<float>& c_row, std::vector<float>& c_col,
		std::vector<float>& c_row_stamp, std::vector<float>& c_col_stamp) const;
	void launch_masks(Launched& L, std::vector<hipEvent_t>& event_pool);
	void launch_tail(Launched& L, std::vector<hipEvent_t>& event_pool);
	void fail_group(Launched& L, const char* what, std::vector<hipEvent_t>& event_pool);
	void claim(Launched& L, double work, const std::vector<Launched>& round);
	void drain_all();
	void decide(Launched& L, std::vector<int32_t>& still);
};

// the stars inside every stamp plus its 5-pixel buffer, in catalogue order, with the float32 stamp coordinates of
// BasePhotometry.catalog (BasePhotometry.py:1094-1181) -- pipeline._catalogs_of_stamps, stamp by stamp
void tp_frames_job::select_catalog(const std::vector<int32_t>& idx, Group& g, std::vector<float>& c_tmag, std::vector<float>& c_row,
	std::vector<float>& c_col, std::vector<float>& c_row_stamp, std::vector<float>& c_col_stamp) const
{
	const tp_frames_catalog& c = *cat;
	const double buffer = 5.0;
	const int64_t B = c.cell;
	g.cat_offsets.assign(1, 0);
	g.cat_starid.clear();
	std::vector<int64_t> found;
	auto clipi = [](int64_t v, int64_t lo, int64_t hi) { return v < lo ? lo : (v > hi ? hi : v); };
	for (int32_t i : idx) {
		const int64_t* st = &cur[(size_t)i * 4];
		const double rlo = (double)st[0] - 0.5 - buffer, rhi = (double)st[1] - 0.5 + buffer;
		const double clo = (double)st[2] - 0.5 - buffer, chi = (double)st[3] - 0.5 + buffer;
		found.clear();
		if (c.n > 0) {
			const int64_t cr0 = clipi((int64_t)std::floor((rlo - (double)c.r0) / (double)B), 0, c.n_cr - 1);
			const int64_t cr1 = clipi((int64_t)std::floor((rhi - (double)c.r0) / (double)B), -1, c.n_cr - 1);
			const int64_t cc0 = clipi((int64_t)std::floor((clo - (double)c.c0) / (double)B), 0, c.n_cc - 1);
			const int64_t cc1 = clipi((int64_t)std::floor((chi - (double)c.c0) / (double)B), -1, c.n_cc - 1);
			if (cc1 >= cc0)
				for (int64_t cr = cr0; cr <= cr1; ++cr) {
					const int64_t a = c.cell_start[cr * c.n_cc + cc0], b = c.cell_start[cr * c.n_cc + cc1 + 1];
					for (int64_t p = a; p < b; ++p) {
						const int64_t s = c.order[p];
						if (c.row[s] >= rlo && c.row[s] < rhi && c.col[s] >= clo && c.col[s] < chi) found.push_back(s);
					}
				}
			std::sort(found.begin(), found.end());
		}
		for (int64_t s : found) {
			g.cat_starid.push_back(c.starid[s]);
			c_tmag.push_back(c.tmag[s]);
			c_col.push_back((float)c.col[s]);
			c_row.push_back((float)c.row[s]);
			c_col_stamp.push_back((float)(c.col[s] - (double)st[2]));
			c_row_stamp.push_back((float)(c.row[s] - (double)st[0]));
		}
		g.cat_offsets.push_back((int64_t)g.cat_starid.size());
	}
	g.n_cat = (int64_t)g.cat_starid.size();
	g.cat_capacity = g.n_cat > 0 ? g.n_cat : 1;
}

namespace {

struct MetaField { const void* src; size_t nbytes; size_t off; };

} // namespace

// The pass of a group of same-sized stamps on stream L.g, queued in two halves.  launch_masks: metadata upload, the sum images, the
// masks, the download of what the round's decisions read, an event behind it.  launch_tail: the cut of the in-mask rows, the
// extraction, the diagnostics and the download of the light curves, an event behind everything.  The worker queues the first
// halves of ALL groups of a round before any second half: a round is decided from the masks alone, and a mask kernel queued behind
// another group's cut and extraction waited for them (round 6, timeline of a 2 500-target batch: the fourth and fifth group of the
// second round delivered their decisions 2 and 3 ms after the first three).
void tp_frames_job::fail_group(Launched& L, const char* what, std::vector<hipEvent_t>& event_pool)
{
	L.failed = true;
	L.error = what;
	if (L.chunked && copy_stream) (void)hipStreamSynchronize(copy_stream);
	(void)hipStreamSynchronize(L.g->stream);
	(void)hipGetLastError();
	for (void* p : L.dev) (void)tp_free(L.g, p);
	L.dev.clear();
	if (L.grp.h_block) { eng->pinned.put(L.grp.h_block, L.grp.h_cap); L.grp.h_block = nullptr; }
	if (L.ev) { event_pool.push_back(L.ev); L.ev = nullptr; }
}

void tp_frames_job::launch_masks(Launched& L, std::vector<hipEvent_t>& event_pool)
{
	tp_ctx* g = L.g;
	Group& G = L.grp;
	const int32_t m = (int32_t)L.idx.size(), H = G.H, W = G.W;
	void* h_meta = nullptr; size_t h_meta_cap = 0;
	auto dalloc = [&](size_t nbytes) { void* p = nullptr; ck(g, tp_malloc(g, nbytes, &p)); L.dev.push_back(p); return p; };
	const auto lab_t0 = std::chrono::steady_clock::now();
	auto lab_t1 = lab_t0;
	try {
		if ((int64_t)H * W > 32767) throw Fail("a " + std::to_string(H) + "x" + std::to_string(W) + " stamp is beyond the 32 767 pixels of the mask builder");
		G.n = m;
		G.target_starid.resize(m);
		std::vector<float> c_tmag, c_row, c_col, c_row_stamp, c_col_stamp;
		if (m < 1024 || eng->helpers.threads.empty()) select_catalog(L.idx, G, c_tmag, c_row, c_col, c_row_stamp, c_col_stamp);
		else {
			// a large group: the stamps in four runs, three of them offered to the engine's helper threads (what no helper has taken when
			// this thread is through with its own run it does itself), joined in order.  The selection of 2 500 stamps is 0.5 of the
			// 0.75 ms a worker needs before it can queue anything, 2 of 2.7 ms for 10 000.
			constexpr int K = 4;
			struct Run { Group part; std::vector<float> tmag, row, col, rs, cs; std::vector<int32_t> idx; bool done = false; };
			Run run[K];
			std::mutex dm;
			std::condition_variable dcv;
			int pending = K - 1;
			for (int k = 0; k < K; ++k) run[k].idx.assign(L.idx.begin() + (size_t)m * k / K, L.idx.begin() + (size_t)m * (k + 1) / K);
			for (int k = 1; k < K; ++k)
				eng->helpers.post([this, &run, &dm, &dcv, &pending, k] {
					try { select_catalog(run[k].idx, run[k].part, run[k].tmag, run[k].row, run[k].col, run[k].rs, run[k].cs); run[k].done = true; }
					catch (...) {}                      // (out of memory on a helper: reported below by the thread that waits)
					{ std::lock_guard<std::mutex> lk(dm); pending -= 1; }
					dcv.notify_one();
				});
			select_catalog(run[0].idx, run[0].part, run[0].tmag, run[0].row, run[0].col, run[0].rs, run[0].cs);
			{
				std::function<void()> f;           // (tasks of other jobs may be among them: any posted run is as good to do)
				while (eng->helpers.take(f)) f();
				std::unique_lock<std::mutex> lk(dm);
				dcv.wait(lk, [&] { return pending == 0; });
			}
			for (int k = 1; k < K; ++k) if (!run[k].done) throw Fail("the catalogue selection of the group failed on a helper thread");
			G.cat_offsets.assign(1, 0);
			G.cat_starid.clear();
			for (int k = 0; k < K; ++k) {
				const int64_t base = (int64_t)G.cat_starid.size();
				for (size_t j = 1; j < run[k].part.cat_offsets.size(); ++j) G.cat_offsets.push_back(base + run[k].part.cat_offsets[j]);
				G.cat_starid.insert(G.cat_starid.end(), run[k].part.cat_starid.begin(), run[k].part.cat_starid.end());
				c_tmag.insert(c_tmag.end(), run[k].tmag.begin(), run[k].tmag.end());
				c_row.insert(c_row.end(), run[k].row.begin(), run[k].row.end());
				c_col.insert(c_col.end(), run[k].col.begin(), run[k].col.end());
				c_row_stamp.insert(c_row_stamp.end(), run[k].rs.begin(), run[k].rs.end());
				c_col_stamp.insert(c_col_stamp.end(), run[k].cs.begin(), run[k].cs.end());
			}
			G.n_cat = (int64_t)G.cat_starid.size();
			G.cat_capacity = G.n_cat > 0 ? G.n_cat : 1;
		}
		// ---- the metadata of the group as ONE block: one upload
		std::vector<int32_t> stamps32((size_t)m * 4);
		std::vector<double> t_row(m), t_col(m), t_tmag(m);
		for (int32_t j = 0; j < m; ++j) {
			const int32_t i = L.idx[j];
			for (int k = 0; k < 4; ++k) stamps32[(size_t)j * 4 + k] = (int32_t)cur[(size_t)i * 4 + k];
			t_row[j] = row[i]; t_col[j] = col[i]; t_tmag[j] = tmag[i];
			G.target_starid[j] = starid[i];
		}
		const size_t nc = (size_t)G.n_cat;
		MetaField f[14] = {
			{quality.data(), (size_t)T * 4, 0}, {time.data(), (size_t)T * 8, 0}, {stamps32.data(), (size_t)m * 16, 0},
			{G.cat_offsets.data(), (size_t)(m + 1) * 8, 0}, {G.cat_starid.data(), nc * 8, 0}, {c_tmag.data(), nc * 4, 0},
			{c_row.data(), nc * 4, 0}, {c_col.data(), nc * 4, 0}, {c_row_stamp.data(), nc * 4, 0}, {c_col_stamp.data(), nc * 4, 0},
			{t_row.data(), (size_t)m * 8, 0}, {t_col.data(), (size_t)m * 8, 0}, {t_tmag.data(), (size_t)m * 8, 0}, {G.target_starid.data(), (size_t)m * 8, 0}};
		size_t total = 0;
		for (auto& x : f) { x.off = total; total = (size_t)round_up((int64_t)(total + std::max(x.nbytes, (size_t)16)), 256); }
		h_meta = eng->pinned.get(total, &h_meta_cap);
		host_scratch.emplace_back(h_meta, h_meta_cap);
		std::memset(h_meta, 0, total);
		for (auto& x : f) if (x.nbytes) std::memcpy(static_cast<char*>(h_meta) + x.off, x.src, x.nbytes);
		lab_t1 = std::chrono::steady_clock::now();
		auto lab_prev = lab_t1;
		auto lap = [&](const char* what) { const auto now = std::chrono::steady_clock::now(); lab_steps[what] += std::chrono::duration<double, std::micro>(now - lab_prev).count(); lab_prev = now; };
		char* d_meta = static_cast<char*>(dalloc(total));
		lap("alloc meta");
		// (measured, TESSPHOT_FRAMES_TIMING: in the first runs of a process this call can return after 8 - 20 ms while other jobs have
		// work queued; in the steady state it takes 30 us.  A kernel that reads the page-locked block through its device mapping never
		// waits, but its system-scope accesses slowed every concurrent kernel: 3.0 x 10^5 targets/s pipelined instead of 5 x 10^5)
		// by a kernel, not by a DMA engine: the streams of a process share the engines, and this upload -- the head of the chain that
		// decides the round -- sat behind the first round's 130 MB of light curves on some streams until THEY had been extracted and
		// copied (round 6, copy trace: the metadata of two of five groups arrived 4 ms late)
		ck(g, tp_blit(g, d_meta, h_meta, total));
		lap("alloc+h2d");
		const int32_t* d_quality = reinterpret_cast<const int32_t*>(d_meta + f[0].off);
		const double* d_time = reinterpret_cast<const double*>(d_meta + f[1].off);
		const int32_t* d_stamps = reinterpret_cast<const int32_t*>(d_meta + f[2].off);
		const int64_t* d_cat_offsets = reinterpret_cast<const int64_t*>(d_meta + f[3].off);
		const int64_t* d_cat_starid = reinterpret_cast<const int64_t*>(d_meta + f[4].off);
		const float* d_cat_tmag = reinterpret_cast<const float*>(d_meta + f[5].off);
		const float* d_cat_row = reinterpret_cast<const float*>(d_meta + f[6].off);
		const float* d_cat_col = reinterpret_cast<const float*>(d_meta + f[7].off);
		const float* d_cat_row_stamp = reinterpret_cast<const float*>(d_meta + f[8].off);
		const float* d_cat_col_stamp = reinterpret_cast<const float*>(d_meta + f[9].off);
		const double* d_t_row = reinterpret_cast<const double*>(d_meta + f[10].off);
		const double* d_t_col = reinterpret_cast<const double*>(d_meta + f[11].off);
		const double* d_t_tmag = reinterpret_cast<const double*>(d_meta + f[12].off);
		const int64_t* d_t_starid = reinterpret_cast<const int64_t*>(d_meta + f[13].off);
		L.d_stamps = d_stamps; L.d_quality = d_quality; L.d_time = d_time;
		// ---- the three stamp cubes (BasePhotometry._load_cube for the whole group; the cutter writes the padding of the time axis)
		tp_cube_desc& desc = L.desc;
		desc.n_targets = m; desc.n_cad = T; desc.height = H; desc.width = W; desc.t_pitch = round_up(T, 32);
		const size_t cube_bytes = (size_t)m * H * W * (size_t)desc.t_pitch * 4;
		const float* frames[3] = {stack.d_images, stack.d_images_err, stack.d_backgrounds};
		float** cubes = L.cubes;
		// the region's sum image is at hand (the FFI branch of BasePhotometry.sumimage): no cube is needed before the masks are known
		const bool crop = L.crop = stack.d_sumimage != nullptr;
		// ... and with the time-major stacks no cube is needed at all (launch_tail)
		L.time_major = crop && stack.d_images_t != nullptr;
		if (!L.time_major) for (int k = 0; k < 3; ++k) cubes[k] = static_cast<float*>(dalloc(cube_bytes));
		lap("alloc cubes");
		const bool large = L.large = m >= kFusedFrom;
		// a small group: one binning of the stamps and one launch for the three stacks.  A large group: the images now, the error and
		// background stacks once the masks are known -- only their in-mask pixel rows are ever read (below)
		if (!crop)
			ck(g, tp_cut_stamps_multi(g, large ? 1 : 3, frames, stack.n_frames, stack.n_rows, stack.n_cols, stack.n_cols, (int64_t)stack.n_rows * stack.n_cols,
				stack.row0, stack.col0, d_stamps, &desc, cubes));
		// ---- the packed output block (comm.packed_block_layout with the catalogue flags, the sum image and the diagnostics)
		const size_t P = (size_t)H * W;
		uint64_t off = 0;
		auto field = [&](uint64_t nbytes) { const uint64_t o = off; off = (uint64_t)round_up((int64_t)(off + nbytes), 256); return o; };
		G.off_lc = field((uint64_t)5 * m * T * 8);
		G.off_cont = field((uint64_t)m * 8);
		G.off_status = field((uint64_t)m * 4);
		G.off_flags = field((uint64_t)m * 4);
		G.off_mask = field((uint64_t)m * P);
		G.off_cim = field((uint64_t)G.cat_capacity);
		G.off_sum = field((uint64_t)m * P * 8);
		G.off_diag = field((uint64_t)m * 10 * 8);
		G.nbytes = off;
		char* blk = L.blk = static_cast<char*>(dalloc((size_t)G.nbytes));
		ckh(hipMemsetAsync(blk, 0, (size_t)G.nbytes, g->stream), "hipMemsetAsync(block)");
		double* d_cont = reinterpret_cast<double*>(blk + G.off_cont);
		int32_t* d_status = reinterpret_cast<int32_t*>(blk + G.off_status);
		int32_t* d_flags = reinterpret_cast<int32_t*>(blk + G.off_flags);
		uint8_t* d_mask = reinterpret_cast<uint8_t*>(blk + G.off_mask);
		uint8_t* d_cim = reinterpret_cast<uint8_t*>(blk + G.off_cim);
		double* d_sum = reinterpret_cast<double*>(blk + G.off_sum);
		// scratch: the mask builder's diagnostics and the aperture image (bit 1 = collected: every pixel, BasePhotometry.py:1043)
		double* d_diag8 = static_cast<double*>(dalloc((size_t)m * 8 * 8));
		ckh(hipMemsetAsync(d_diag8, 0, (size_t)m * 64, g->stream), "hipMemsetAsync(diag)");
		int32_t* d_aperture = static_cast<int32_t*>(dalloc((size_t)m * P * 4));
		ckh(hipMemsetAsync(d_aperture, 1, (size_t)m * P * 4, g->stream), "hipMemsetAsync(aperture)");
		lap("alloc+memsets");
		// ---- the masks.  The three stand-alone kernels (bit-identical to the fused launch; a small group is latency-bound and spreads
		// better over the chip this way).  With the region's sum image: crop, masks, and then (second half) ONE cut of the in-mask rows of
		// all three stacks (a sixth of a 15 x 15 stamp: 5.5 GB of traffic per 2 500 stamps instead of 10.8).
		if (crop) ck(g, tp_crop_sumimage(g, stack.d_sumimage, stack.n_rows, stack.n_cols, stack.n_cols, stack.row0, stack.col0, d_stamps, m, H, W, d_sum));
		else ck(g, tp_sumimage(g, &desc, cubes[0], d_quality, 0, kBitmask, nullptr, 0, d_sum));
		lap("crop");
		ck(g, tp_k2p2_masks(g, m, H, W, d_sum, d_cat_offsets, d_cat_col_stamp, d_cat_row_stamp, d_cat_tmag, d_cat_col, d_cat_row, d_cat_starid,
			d_t_row, d_t_col, d_t_tmag, d_t_starid, d_stamps, d_aperture, nullptr, nullptr, d_mask, d_status, d_flags, d_cont, d_diag8, d_cim));
		lap("k2p2");
		// ---- downloads: what the decisions read (status, flags, mask, catalogue flags, sum image) is complete once the masks are --
		// it leaves now, with an event, and the worker decides the job's next round while this group's extraction and diagnostics run
		// (nothing of a round is decided from the light curves; a target that is cut again has its extraction redone anyway)
		G.h_block = eng->pinned.get((size_t)G.nbytes, &G.h_cap);
		lap("pinned");
		const uint64_t lc_bytes = G.off_cont;
		ck(g, tp_blit(g, static_cast<char*>(G.h_block) + lc_bytes, blk + lc_bytes, G.off_diag - lc_bytes));   // (by a kernel: see the metadata)
		if (event_pool.empty()) { hipEvent_t e = nullptr; ckh(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate"); event_pool.push_back(e); }
		L.ev = event_pool.back(); event_pool.pop_back();
		ckh(hipEventRecord(L.ev, g->stream), "hipEventRecord");
		lap("d2h+event");
		const auto lab_t2 = std::chrono::steady_clock::now();
		lab_us[0] += std::chrono::duration<double, std::micro>(lab_t1 - lab_t0).count();
		lab_us[1] += std::chrono::duration<double, std::micro>(lab_t2 - lab_t1).count();
		lab_groups += 1;
	} catch (const std::exception& e) {
		const std::string what = e.what();
		fail_group(L, what.c_str(), event_pool);
	}
}

void tp_frames_job::launch_tail(Launched& L, std::vector<hipEvent_t>& event_pool)
{
	if (L.failed) return;
	tp_ctx* g = L.g;
	Group& G = L.grp;
	const int32_t m = G.n, H = G.H, W = G.W;
	const auto lab_t1 = std::chrono::steady_clock::now();
	try {
		auto lab_prev = lab_t1;
		auto lap = [&](const char* what) { const auto now = std::chrono::steady_clock::now(); lab_steps[what] += std::chrono::duration<double, std::micro>(now - lab_prev).count(); lab_prev = now; };
		const tp_cube_desc& desc = L.desc;
		float** cubes = L.cubes;
		char* blk = L.blk;
		const float* frames[3] = {stack.d_images, stack.d_images_err, stack.d_backgrounds};
		double* lc[5];
		for (int k = 0; k < 5; ++k) lc[k] = reinterpret_cast<double*>(blk + G.off_lc) + (size_t)k * m * T;
		int32_t* d_status = reinterpret_cast<int32_t*>(blk + G.off_status);
		uint8_t* d_mask = reinterpret_cast<uint8_t*>(blk + G.off_mask);
		double* d_sum = reinterpret_cast<double*>(blk + G.off_sum);
		double* d_diagn = reinterpret_cast<double*>(blk + G.off_diag);
		const uint64_t lc_bytes = G.off_cont;
		// For a large group the cut of the error and background stacks comes BETWEEN mask and extraction and writes in-mask rows only:
		// of 8.9 GB of cubes per 2 500 stamps of 15 x 15 the passes read 4.4 (the images for the sum image, a sixth of the rows of all
		// three for the extraction), so two thirds of the old cut's writes were never read
		if (L.time_major) {}     // nothing to cut: the extraction reads the rows of the time-major stacks
		else if (L.crop)
			ck(g, tp_cut_stamps_masked(g, 3, frames, stack.n_frames, stack.n_rows, stack.n_cols, stack.n_cols, (int64_t)stack.n_rows * stack.n_cols,
				stack.row0, stack.col0, L.d_stamps, &desc, d_mask, cubes));
		else if (L.large)
			ck(g, tp_cut_stamps_masked(g, 2, frames + 1, stack.n_frames, stack.n_rows, stack.n_cols, stack.n_cols, (int64_t)stack.n_rows * stack.n_cols,
				stack.row0, stack.col0, L.d_stamps, &desc, d_mask, cubes + 1));
		lap("masked cut");
		auto new_event = [&]() {
			if (event_pool.empty()) { hipEvent_t e = nullptr; ckh(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate"); event_pool.push_back(e); }
			hipEvent_t e = event_pool.back(); event_pool.pop_back();
			tails.push_back(e);               // (destroyed with the tail events when the job ends)
			return e;
		};
		// The light curves of a large group are most of what the call downloads (130 MB per 2 500 targets: 2.3 ms of the link), and they
		// used to leave when extraction AND diagnostics of the whole group were done.  Now the group is extracted in chunks of targets,
		// and a chunk's five planes leave on the job's copy stream as soon as the chunk is extracted: the link starts 0.15 ms after the
		// cut instead of 1.2 ms, and the diagnostics run under the copies.
		const int32_t n_chunks = (copy_stream && m >= 2048) ? std::min<int32_t>(8, m / 512) : 1;
		L.chunked = n_chunks > 1;
		for (int32_t c = 0; c < n_chunks; ++c) {
			const int32_t j0 = (int32_t)((int64_t)m * c / n_chunks), j1 = (int32_t)((int64_t)m * (c + 1) / n_chunks);
			tp_cube_desc part = desc;
			part.n_targets = j1 - j0;
			const size_t cube_off = (size_t)j0 * H * W * (size_t)desc.t_pitch;
			if (L.time_major)
				ck(g, tp_aperture_extract_stack(g, j1 - j0, T, H, W, stack.d_images_t, stack.d_images_err_t, stack.d_backgrounds_t, stack.t_pitch,
					stack.n_rows, stack.n_cols, stack.row0, stack.col0,
					d_mask + (size_t)j0 * H * W, L.d_stamps + (size_t)j0 * 4, d_status + j0,
					lc[0] + (size_t)j0 * T, lc[1] + (size_t)j0 * T, lc[2] + (size_t)j0 * T, lc[3] + (size_t)j0 * T, lc[4] + (size_t)j0 * T, T));
			else
			ck(g, tp_aperture_extract(g, &part, cubes[0] + cube_off, cubes[1] + cube_off, cubes[2] + cube_off, 0, 0, nullptr, 0,
				d_mask + (size_t)j0 * H * W, L.d_stamps + (size_t)j0 * 4, d_status + j0,
				lc[0] + (size_t)j0 * T, lc[1] + (size_t)j0 * T, lc[2] + (size_t)j0 * T, lc[3] + (size_t)j0 * T, lc[4] + (size_t)j0 * T, T));
			if (n_chunks > 1) {
				hipEvent_t e = new_event();
				ckh(hipEventRecord(e, g->stream), "hipEventRecord");
				ckh(hipStreamWaitEvent(copy_stream, e, 0), "hipStreamWaitEvent");
				// the five planes of the chunk as ONE rectangular copy (five rows, a plane apart): 48 DMA commands per 2 500 targets instead
				// of 240, each followed by ~20 us of idle link (copy trace of four jobs in flight: the link was busy 87 % of the time;
				// 7.55 -> 7.98 x 10^5 targets/s).  TESSPHOT_FRAMES_RECT=0: plane by plane
				static const bool rect = [] { const char* e = std::getenv("TESSPHOT_FRAMES_RECT"); return !(e && e[0] == '0'); }();
				if (rect) {
					const size_t o = (size_t)G.off_lc + (size_t)j0 * T * 8;
					ckh(hipMemcpy2DAsync(static_cast<char*>(G.h_block) + o, (size_t)m * T * 8, blk + o, (size_t)m * T * 8, (size_t)(j1 - j0) * T * 8, 5,
						hipMemcpyDeviceToHost, copy_stream), "hipMemcpy2DAsync(light curves)");
				} else
				for (int k = 0; k < 5; ++k) {
					const size_t o = (size_t)G.off_lc + ((size_t)k * m + (size_t)j0) * T * 8;
					ckh(hipMemcpyAsync(static_cast<char*>(G.h_block) + o, blk + o, (size_t)(j1 - j0) * T * 8, hipMemcpyDeviceToHost, copy_stream), "hipMemcpyAsync(light curves)");
				}
			}
		}
		lap("extract");
		ck(g, tp_lightcurve_diagnostics(g, m, T, lc[0], lc[1], lc[3], lc[4], T, L.d_time, L.d_quality, 0, kBitmask, d_status, d_sum, d_mask, H, W,
			3600.0 / 86400.0, d_diagn));
		lap("diagnostics");
		// (by a kernel, like everything but the large group's light curves: a copy that waits in a DMA engine's queue behind another
		// job's 130 MB holds this STREAM, and the next round's masks queued on it, for as long)
		ck(g, tp_blit(g, static_cast<char*>(G.h_block) + G.off_diag, blk + G.off_diag, G.nbytes - G.off_diag));
		if (n_chunks > 1) {
			// the copies' end is one of the job's tail events.  The group's stream does NOT wait for them (it would be held, and whatever
			// another job queues on it next, for the 2 - 10 ms the link takes): the output block they read is the one block of the group
			// that is not given back in stream order below -- it goes back when the job has seen its tail events
			hipEvent_t e = new_event();
			ckh(hipEventRecord(e, copy_stream), "hipEventRecord");
			for (size_t i = 0; i < L.dev.size(); ++i)
				if (L.dev[i] == static_cast<void*>(blk)) { L.dev.erase(L.dev.begin() + (long)i); late_frees.emplace_back(g, static_cast<void*>(blk)); break; }
		} else if (lc_bytes <= ((uint64_t)32 << 20)) {
			ck(g, tp_blit(g, G.h_block, blk, lc_bytes));
		} else {
			ckh(hipMemcpyAsync(G.h_block, blk, (size_t)lc_bytes, hipMemcpyDeviceToHost, g->stream), "hipMemcpyAsync(light curves)");
		}
		lap("d2h light curves");
		// the group's last word: the job ends when the tail events of all its groups have completed (its streams are shared)
		if (event_pool.empty()) { hipEvent_t e = nullptr; ckh(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate"); event_pool.push_back(e); }
		hipEvent_t te = event_pool.back(); event_pool.pop_back();
		tails.push_back(te);
		ckh(hipEventRecord(te, g->stream), "hipEventRecord");
		for (void* p : L.dev) (void)tp_free(g, p);      // stream-ordered: handed out again only after what is queued above has run
		L.dev.clear();
		lab_us[1] += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - lab_t1).count();
	} catch (const std::exception& e) {
		// (the round's decisions are taken after both halves of all its groups have been queued: the group counts as failed, as if
		// its first half had)
		const std::string what = e.what();
		fail_group(L, what.c_str(), event_pool);
	}
}

// the plugin's rules on the results of one group (photometry.py:93-170; plugins.mask_outcome, stamps.py)
void tp_frames_job::decide(Launched& L, std::vector<int32_t>& still)
{
	Group& G = L.grp;
	const int32_t m = G.n, H = G.H, W = G.W;
	const int32_t gid = (int32_t)groups.size();
	const char* blk = static_cast<const char*>(G.h_block);
	const int32_t* r_status = reinterpret_cast<const int32_t*>(blk + G.off_status);
	const int32_t* r_flags = reinterpret_cast<const int32_t*>(blk + G.off_flags);
	const uint8_t* r_mask = reinterpret_cast<const uint8_t*>(blk + G.off_mask);
	const double* r_sum = reinterpret_cast<const double*>(blk + G.off_sum);
	const int64_t limits[4] = {stack.row0, (int64_t)stack.row0 + stack.n_rows, stack.col0, (int64_t)stack.col0 + stack.n_cols};
	static const int side_bit[4] = {2, 4, 8, 16};      // down, up, left, right (stamps.SIDES)
	static const int side_sign[4] = {-1, +1, -1, +1};
	std::vector<double> vals;
	for (int32_t j = 0; j < m; ++j) {
		const int32_t i = L.idx[j];
		attempts[i] -= 1;
		const int32_t fl = r_flags[j], kind = fl >> 8;
		auto stands = [&]() {
			has_result[i] = 1; group[i] = gid; pos[i] = j;
			finish(i, r_status[j]);
		};
		if ((fl & (1 | 32 | kEdgeBits)) == 0 && kind == 0) { stands(); continue; }   // the common case: nothing to log, no edge touched
		// plugins.mask_outcome
		if (fl & 32) log(i, 1);
		if (fl & 1) log(i, (fl & (32 | 64)) ? 2 : 3);
		if (kind == 5) { log(i, 4); finish(i, TP_STATUS_ERROR); continue; }
		if (kind >= 1 && kind <= 4) { direct(i, 5, kind); finish(i, TP_STATUS_ERROR); continue; }   // an uncaught exception upstream
		if (fl & kEdgeBits) {
			int64_t before[4], after[4];
			for (int k = 0; k < 4; ++k) before[k] = after[k] = cur[(size_t)i * 4 + k];
			for (int s = 0; s < 4; ++s) if (fl & side_bit[s]) after[s] += side_sign[s] * kResizeStep;
			// stamps.clip_stamp (growing a valid stamp cannot empty it)
			after[0] = std::max(after[0], limits[0]); after[2] = std::max(after[2], limits[2]);
			after[1] = std::min(after[1], limits[1]); after[3] = std::min(after[3], limits[3]);
			if (std::equal(before, before + 4, after)) {
				log(i, 6);                             // "Could not resize stamp any further.": the attempt just made stands
			} else {
				stamp_resizes[i] += 1;
				for (int k = 0; k < 4; ++k) cur[(size_t)i * 4 + k] = after[k];
				bool quick = false;
				double stuck = 0.0;
				if (budget_flux[i] == budget_flux[i]) {     // bright target (not NaN): stamps.quick_break_flux
					bool side_stuck[4], any = false;
					for (int s = 0; s < 4; ++s) { side_stuck[s] = (fl & side_bit[s]) && before[s] == after[s]; any = any || side_stuck[s]; }
					if (any) {
						vals.clear();
						const uint8_t* mk = r_mask + (size_t)j * H * W;
						const double* sm = r_sum + (size_t)j * H * W;
						for (int r = 0; r < H; ++r)
							for (int c = 0; c < W; ++c) {
								const bool edge = (side_stuck[0] && r == 0) || (side_stuck[1] && r == H - 1) || (side_stuck[2] && c == 0) || (side_stuck[3] && c == W - 1);
								if (edge && mk[r * W + c]) { const double v = sm[r * W + c]; vals.push_back(v == v ? v : 0.0); }
							}
						stuck = np_pairwise_sum(vals.data(), (int64_t)vals.size());
						quick = stuck > budget_flux[i];
					}
				}
				if (quick) { log(i, 7, 0, 0, stuck); finish(i, TP_STATUS_ERROR); }
				else if (attempts[i] == 0) { log(i, 8); finish(i, TP_STATUS_ERROR); }
				else still.push_back(i);
				continue;
			}
		}
		if (kind == 6) log(i, 9);                          // "No targets in mask."
		stands();
	}
	groups.push_back(std::move(G));
}

// a stream of the engine's pool for a small group: an idle one (its last claimant's event has completed) in index order, else the one
// with the least work queued since it was last seen idle; claimed until the round is queued
void tp_frames_job::claim(Launched& L, double work, const std::vector<Launched>& round)
{
	for (;;) {
		{
			std::lock_guard<std::mutex> lk(eng->sm);
			int pick = -1;
			for (size_t k = 0; k < eng->small.size(); ++k) {
				auto& S = eng->small[k];
				if (S.claimed) continue;
				if (S.busy && S.load > 0.0 && hipEventQuery(S.busy) == hipSuccess) S.load = 0.0;   // drained
				if (S.load == 0.0) { pick = (int)k; break; }
				if (pick < 0 || S.load < eng->small[(size_t)pick].load) pick = (int)k;
			}
			(void)hipGetLastError();   // hipErrorNotReady of the queries
			if (pick < 0)                // none unclaimed.  A round with more groups than the pool has streams: one this job holds already
				for (const Launched& o : round)
					if (o.small >= 0 && (pick < 0 || eng->small[(size_t)o.small].load < eng->small[(size_t)pick].load)) pick = o.small;
			if (pick >= 0) {
				auto& S = eng->small[(size_t)pick];
				S.claimed = true;
				S.load += work;
				L.small = pick;
				L.g = S.c;
				return;
			}
		}
		std::this_thread::yield();     // every stream of the pool is being queued on by the other jobs' workers: a matter of microseconds
	}
}

// error paths: everything this job may have queued anywhere has run (its copy stream and the whole pool)
void tp_frames_job::drain_all()
{
	if (copy_stream) (void)hipStreamSynchronize(copy_stream);
	for (auto& S : eng->small) (void)hipStreamSynchronize(S.c->stream);
	(void)hipGetLastError();
}

void tp_frames_job::run()
{
	(void)hipSetDevice(eng->device);
	lab_run0 = std::chrono::steady_clock::now();
	std::vector<hipEvent_t> event_pool;
	try {
		status.assign(n, 0); stamp_resizes.assign(n, 0); group.assign(n, -1); pos.assign(n, 0); has_result.assign(n, 0);
		stamp.assign((size_t)n * 4, -1);
		std::vector<int32_t> active;
		for (int32_t i = 0; i < n; ++i) {
			if (valid[i]) { active.push_back(i); continue; }
			status[i] = TP_STATUS_ERROR;                    // BasePhotometry.py:671-672: the constructor raises
			direct(i, 12);
			stamp[(size_t)i * 4] = -1; stamp[(size_t)i * 4 + 1] = -2; stamp[(size_t)i * 4 + 2] = -1; stamp[(size_t)i * 4 + 3] = -2;
		}
		const int64_t pitch = round_up(T, 32);
		while (!active.empty()) {
			// ---- the groups of this round (targets that share a stamp size), cut into parts that fit the memory budget
			std::map<int64_t, std::vector<int32_t>> by_size;
			for (int32_t i : active) {
				const int64_t h = cur[(size_t)i * 4 + 1] - cur[(size_t)i * 4], w = cur[(size_t)i * 4 + 3] - cur[(size_t)i * 4 + 2];
				by_size[h * 100000 + w].push_back(i);
			}
			struct Piece { std::vector<int32_t> idx; int32_t H, W; double nbytes; };
			std::vector<std::vector<Piece>> parts(1);
			double acc = 0.0;
			for (auto& kv : by_size) {
				const int32_t H = (int32_t)(kv.first / 100000), W = (int32_t)(kv.first % 100000);
				const bool cubes_needed = !(stack.d_sumimage && stack.d_images_t);
				const double per_target = (cubes_needed ? 3.0 * H * W * (double)pitch * 4 : 0.0) + 5.0 * T * 8 + (double)H * W * 13 + 256;
				const int64_t nmax = std::max<int64_t>(1, (int64_t)std::floor(budget / per_target));
				for (size_t a0 = 0; a0 < kv.second.size(); a0 += (size_t)nmax) {
					Piece p;
					p.idx.assign(kv.second.begin() + a0, kv.second.begin() + std::min(kv.second.size(), a0 + (size_t)nmax));
					p.H = H; p.W = W; p.nbytes = per_target * (double)p.idx.size();
					if (!parts.back().empty() && acc + p.nbytes > budget) { parts.emplace_back(); acc = 0.0; }
					acc += p.nbytes;
					parts.back().push_back(std::move(p));
				}
			}
			std::vector<int32_t> still;
			for (auto& part : parts) {
				std::vector<Launched> launched(part.size());
				// the claims on the pool's streams end when the round is queued -- or when anything on the way throws
				struct Claims {
					tp_frames_engine* e; std::vector<Launched>& ls; bool held = true;
					void release() {
						if (!held) return;
						held = false;
						std::lock_guard<std::mutex> lk(e->sm);
						for (auto& L : ls) {
							if (L.small < 0) continue;
							auto& S = e->small[(size_t)L.small];
							if (!S.busy) (void)hipEventCreateWithFlags(&S.busy, hipEventDisableTiming);
							if (S.busy) (void)hipEventRecord(S.busy, S.c->stream);
							S.claimed = false;
						}
					}
					~Claims() { release(); }
				} claims{eng, launched};
				for (size_t gi = 0; gi < part.size(); ++gi) {
					Launched& L = launched[gi];
					L.idx = std::move(part[gi].idx);
					L.grp.H = part[gi].H; L.grp.W = part[gi].W;
					// Every group takes a stream of the engine's pool, an IDLE one if there is one: the large group of the first round as
					// well as the resized stamps of a few targets -- chains of latency-bound launches that decide when the job's next
					// round can start.  Until round 6 a job had four streams of its own, three of them for the small groups, taken in
					// turn: the fourth and fifth group of a round queued behind the first two's cut, extraction and diagnostics, and a
					// round of five groups was decided 3 ms after its first three masks were done.  (More streams per JOB do not help:
					// with 7 per job, 35 in the engine, a call alone took 11 - 14 ms instead of 7.6 -- see the note at the pool.)
					const double work = (double)L.idx.size() * (double)L.grp.H * (double)L.grp.W;   // what its tail costs, roughly
					claim(L, work, launched);
				}
				for (auto& L : launched) launch_masks(L, event_pool);
				for (auto& L : launched) launch_tail(L, event_pool);
				claims.release();
				lab_mark("queued", lab_rounds + 1, (int)launched.size());
				std::string lost;                        // a device error that surfaces at an event costs every group of the part
				bool drained = false;                    // ... and the job's streams are drained once before any of its blocks is given back
				lab_rounds += 1;
				for (auto& L : launched) {
					const auto lab_a = std::chrono::steady_clock::now();
					if (!L.failed && lost.empty()) {
						const hipError_t e = hipEventSynchronize(L.ev);
						if (e != hipSuccess) { lost = std::string("hipEventSynchronize: ") + hipGetErrorString(e); (void)hipGetLastError(); }
					}
					const auto lab_b = std::chrono::steady_clock::now();
					lab_mark("ev", lab_rounds, (int)L.idx.size());
					lab_us[2] += std::chrono::duration<double, std::micro>(lab_b - lab_a).count();
					struct LabDecide { double& acc; std::chrono::steady_clock::time_point t; ~LabDecide() { acc += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t).count(); } } lab_d{lab_us[3], lab_b};
					if (L.ev) { event_pool.push_back(L.ev); L.ev = nullptr; }
					if (L.failed || !lost.empty()) {
						// the copies into this part's page-locked blocks may still be queued (on this group's stream, or -- once an error
						// has surfaced and the remaining groups are no longer waited for one by one -- on any of the job's streams): a block
						// goes back to the engine-wide pool, where another job's thread may take it, only after they have drained
						if (!drained) { drain_all(); drained = true; }
						if (L.grp.h_block) { eng->pinned.put(L.grp.h_block, L.grp.h_cap); L.grp.h_block = nullptr; }
						const int32_t t = add_text(L.failed ? L.error : lost);
						for (int32_t i : L.idx) { log(i, 10, L.grp.H, L.grp.W, 0.0, t); finish(i, TP_STATUS_ERROR); }
						continue;
					}
					decide(L, still);
				}
			}
			std::sort(still.begin(), still.end());
			active.swap(still);
		}
		// ---- the light curves of every round have arrived
		std::string copy_error;
		const auto lab_w = std::chrono::steady_clock::now();
		for (hipEvent_t te : tails) {
			const hipError_t e = hipEventSynchronize(te);
			if (e != hipSuccess && copy_error.empty()) { copy_error = std::string("hipEventSynchronize: ") + hipGetErrorString(e); (void)hipGetLastError(); }
		}
		for (auto& lf : late_frees) (void)tp_free(lf.first, lf.second);   // (the allocator of a context is serialised by its own mutex)
		late_frees.clear();
		lab_us[4] += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - lab_w).count();
		if (const char* le = std::getenv("TESSPHOT_FRAMES_TIMING"))
			if (le[0] == '1')
				std::fprintf(stderr, "[frames job] %d targets, %d rounds, %d groups: select+metadata %.0f us, queueing %.0f, waiting for decisions %.0f, deciding %.0f, last light curves %.0f\n",
					(int)n, lab_rounds, lab_groups, lab_us[0], lab_us[1], lab_us[2], lab_us[3], lab_us[4]);
		if (const char* le = std::getenv("TESSPHOT_FRAMES_TIMING"))
			if (le[0] == '1' && le[1] == '1') {
				std::string line = "[frames job]   queueing:";
				for (auto& kv : lab_steps) { char buf[96]; std::snprintf(buf, sizeof buf, " %s %.0f,", kv.first.c_str(), kv.second); line += buf; }
				std::fprintf(stderr, "%s\n[frames job]   timeline:%s end@%.0f\n", line.c_str(), lab_timeline.c_str(),
					std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - lab_run0).count());
			}
		if (!copy_error.empty()) {       // nothing that was extracted can be trusted
			const int32_t t = add_text(copy_error);
			for (int32_t i = 0; i < n; ++i) if (has_result[i]) { has_result[i] = 0; direct(i, 11, 0, 0, 0.0, t); status[i] = TP_STATUS_ERROR; }
		}
	} catch (const std::exception& e) {
		rc = TP_ERR_HIP;
		err = e.what();
		drain_all();
		for (auto& lf : late_frees) (void)tp_free(lf.first, lf.second);
		late_frees.clear();
	}
	for (auto e : tails) (void)hipEventDestroy(e);
	tails.clear();
	for (auto e : event_pool) (void)hipEventDestroy(e);
	for (auto& hs : host_scratch) eng->pinned.put(hs.first, hs.second);
	host_scratch.clear();
}

extern "C" {

int tp_frames_engine_create(int device, int32_t n_slots, tp_frames_engine** out)
{
	if (!out) { tp_global_err = "tp_frames_engine_create: null output pointer"; return TP_ERR_INVALID; }
	*out = nullptr;
	if (n_slots < 1 || n_slots > 64) { tp_global_err = "tp_frames_engine_create: 1 .. 64 slots"; return TP_ERR_INVALID; }
	TP_API_BEGIN
	tp_frames_engine* eng = new tp_frames_engine();
	eng->device = device;
	eng->n_slots = n_slots;
	if (const char* se = std::getenv("TESSPHOT_FRAMES_STREAMS")) { const int v = std::atoi(se); if (v >= 1 && v <= kStreams - 1) g_small_streams = v; }
	for (int i = 0; i < n_slots * g_small_streams; ++i) {
		tp_ctx* c = nullptr;
		const char* pe = std::getenv("TESSPHOT_FRAMES_PRIO");
		const int rc = tp_ctx_create_stream(device, (pe && pe[0] == '1') ? 1 : 0, &c);
		if (rc != TP_OK) {
			for (tp_ctx* x : eng->ctxs) (void)tp_ctx_destroy(x);
			delete eng;
			return rc;
		}
		c->reuse_in_stream_order = true;      // (every block of a group is allocated from, used on and freed to the context of its stream)
		eng->ctxs.push_back(c);
		tp_frames_engine::SmallStream S; S.c = c; eng->small.push_back(S);
	}
	eng->busy.assign(n_slots, 0);
	eng->helpers.start(3);
	eng->copy_streams.assign(n_slots, nullptr);
	(void)hipSetDevice(device);
	for (int i = 0; i < n_slots; ++i)
		if (hipStreamCreateWithFlags(&eng->copy_streams[(size_t)i], hipStreamNonBlocking) != hipSuccess) { eng->copy_streams[(size_t)i] = nullptr; (void)hipGetLastError(); }
	hipDeviceProp_t prop;
	if (hipGetDeviceProperties(&prop, device) == hipSuccess) eng->hbm_bytes = (uint64_t)prop.totalGlobalMem;
	*out = eng;
	return TP_OK;
	TP_API_END((tp_ctx*)nullptr)
}

int tp_frames_engine_destroy(tp_frames_engine* eng)
{
	if (!eng) return TP_OK;
	while (eng->running.load() > 0) std::this_thread::yield();   // (a job still running: its streams go only once it is through)
	for (auto& S : eng->small) if (S.busy) (void)hipEventDestroy(S.busy);
	for (hipStream_t cs : eng->copy_streams) if (cs) (void)hipStreamDestroy(cs);
	for (tp_ctx* c : eng->ctxs) (void)tp_ctx_destroy(c);
	delete eng;
	return TP_OK;
}

int tp_frames_engine_info(tp_frames_engine* eng, int32_t* n_slots, int32_t* n_free, uint64_t* hbm_bytes)
{
	if (!eng) { tp_global_err = "null engine"; return TP_ERR_INVALID; }
	std::lock_guard<std::mutex> lk(eng->m);
	if (n_slots) *n_slots = eng->n_slots;
	if (n_free) { int f = 0; for (char b : eng->busy) f += b ? 0 : 1; *n_free = f; }
	if (hbm_bytes) *hbm_bytes = eng->hbm_bytes;
	return TP_OK;
}

int tp_frames_catalog_create(int64_t n_stars, const int64_t* h_starid, const float* h_tmag, const double* h_row, const double* h_column,
	tp_frames_catalog** out)
{
	if (!out || n_stars < 0 || (n_stars > 0 && !(h_starid && h_tmag && h_row && h_column))) { tp_global_err = "tp_frames_catalog_create: bad arguments"; return TP_ERR_INVALID; }
	*out = nullptr;
	TP_API_BEGIN
	tp_frames_catalog* c = new tp_frames_catalog();
	c->n = n_stars;
	c->starid.assign(h_starid, h_starid + n_stars);
	c->tmag.assign(h_tmag, h_tmag + n_stars);
	c->row.assign(h_row, h_row + n_stars);
	c->col.assign(h_column, h_column + n_stars);
	// pipeline._CatalogIndex: cells of 16 x 16 pixels from the floor of the smallest row / column; stars without a position go
	// to a cell no stamp asks for
	double rmin = INFINITY, cmin = INFINITY;
	for (int64_t i = 0; i < n_stars; ++i) {
		if (std::isfinite(c->row[i])) rmin = std::min(rmin, c->row[i]);
		if (std::isfinite(c->col[i])) cmin = std::min(cmin, c->col[i]);
	}
	c->r0 = std::isfinite(rmin) ? (int64_t)std::floor(rmin) : 0;
	c->c0 = std::isfinite(cmin) ? (int64_t)std::floor(cmin) : 0;
	std::vector<int64_t> cr(n_stars, 0), cc(n_stars, 0);
	int64_t crmax = 0, ccmax = 0;
	for (int64_t i = 0; i < n_stars; ++i) {
		if (std::isfinite(c->row[i]) && std::isfinite(c->col[i])) {
			cr[i] = (int64_t)std::floor((c->row[i] - (double)c->r0) / (double)c->cell);
			cc[i] = (int64_t)std::floor((c->col[i] - (double)c->c0) / (double)c->cell);
			crmax = std::max(crmax, cr[i]); ccmax = std::max(ccmax, cc[i]);
		}
	}
	c->n_cr = crmax + 1; c->n_cc = ccmax + 1;
	const int64_t n_cells = c->n_cr * c->n_cc;
	std::vector<int64_t> cid(n_stars);
	for (int64_t i = 0; i < n_stars; ++i)
		cid[i] = (std::isfinite(c->row[i]) && std::isfinite(c->col[i]) && cr[i] >= 0 && cc[i] >= 0) ? cr[i] * c->n_cc + cc[i] : n_cells;
	c->order.resize(n_stars);
	for (int64_t i = 0; i < n_stars; ++i) c->order[i] = i;
	std::stable_sort(c->order.begin(), c->order.end(), [&](int64_t a, int64_t b) { return cid[a] < cid[b]; });
	c->cell_start.assign(n_cells + 2, 0);
	for (int64_t i = 0; i < n_stars; ++i) c->cell_start[cid[i] + 1] += 1;
	for (int64_t k = 0; k <= n_cells; ++k) c->cell_start[k + 1] += c->cell_start[k];
	*out = c;
	return TP_OK;
	TP_API_END((tp_ctx*)nullptr)
}

int tp_frames_catalog_destroy(tp_frames_catalog* cat)
{
	delete cat;
	return TP_OK;
}

int tp_frames_submit(tp_frames_engine* eng, const tp_frames_stack* stack, const tp_frames_catalog* cat,
	int32_t n_targets, const int64_t* h_starid, const double* h_tmag, const double* h_row, const double* h_column,
	const int64_t* h_stamps, const uint8_t* h_valid, const int32_t* h_attempts, const double* h_quick_break_budget,
	const double* h_time, const int32_t* h_quality, double budget_bytes, tp_frames_job** out)
{
	if (!eng || !out) { tp_global_err = "tp_frames_submit: null engine / output pointer"; return TP_ERR_INVALID; }
	*out = nullptr;
	if (!stack || !cat || n_targets < 0 || !stack->d_images || !stack->d_images_err || !stack->d_backgrounds || stack->n_frames <= 0 ||
		stack->n_rows <= 0 || stack->n_cols <= 0 || !h_time || !h_quality ||
		(n_targets > 0 && !(h_starid && h_tmag && h_row && h_column && h_stamps && h_valid && h_attempts && h_quick_break_budget))) {
		tp_global_err = "tp_frames_submit: bad arguments";
		return TP_ERR_INVALID;
	}
	if (stack->d_images_t || stack->d_images_err_t || stack->d_backgrounds_t) {
		if (!(stack->d_images_t && stack->d_images_err_t && stack->d_backgrounds_t && stack->d_sumimage) || stack->t_pitch < stack->n_frames || stack->t_pitch % 4 != 0 ||
			(int64_t)stack->n_rows * stack->n_cols >= ((int64_t)1 << 31)) {
			tp_global_err = "tp_frames_submit: the time-major stacks come all three, with the region's sum image, and t_pitch >= n_frames, a multiple of 4";
			return TP_ERR_INVALID;
		}
	}
	TP_API_BEGIN
	int slot = -1;
	{
		std::lock_guard<std::mutex> lk(eng->m);
		for (int s = 0; s < eng->n_slots; ++s) if (!eng->busy[s]) { slot = s; eng->busy[s] = 1; break; }
	}
	if (slot < 0) { tp_global_err = "tp_frames_submit: every slot of the engine holds a job (wait for and release one first)"; return TP_ERR_INVALID; }
	tp_frames_job* job = new tp_frames_job();
	job->eng = eng; job->slot = slot;
	job->copy_stream = eng->copy_streams[(size_t)slot];
	job->stack = *stack; job->cat = cat;
	job->n = n_targets; job->T = stack->n_frames;
	job->starid.assign(h_starid, h_starid + n_targets);
	job->tmag.assign(h_tmag, h_tmag + n_targets);
	job->row.assign(h_row, h_row + n_targets);
	job->col.assign(h_column, h_column + n_targets);
	job->cur.assign(h_stamps, h_stamps + (size_t)n_targets * 4);
	job->valid.assign(h_valid, h_valid + n_targets);
	job->attempts.assign(h_attempts, h_attempts + n_targets);
	job->budget_flux.assign(h_quick_break_budget, h_quick_break_budget + n_targets);
	job->time.assign(h_time, h_time + job->T);
	job->quality.assign(h_quality, h_quality + job->T);
	job->budget = budget_bytes > 0 ? budget_bytes : (double)eng->hbm_bytes / 4.0;
	eng->running.fetch_add(1);
	try {
		job->worker = std::thread([job] { job->run(); job->done.store(1, std::memory_order_release); job->eng->running.fetch_sub(1); });
	} catch (...) {          // no thread to be had: the slot is free again, the job never existed
		eng->running.fetch_sub(1);
		{ std::lock_guard<std::mutex> lk(eng->m); eng->busy[slot] = 0; }
		delete job;
		tp_global_err = "tp_frames_submit: could not start the job's worker thread";
		return TP_ERR_INVALID;
	}
	*out = job;
	return TP_OK;
	TP_API_END((tp_ctx*)nullptr)
}

int tp_frames_wait(tp_frames_job* job)
{
	if (!job) { tp_global_err = "null job"; return TP_ERR_INVALID; }
	if (!job->joined) {
		job->worker.join();
		job->joined = true;
		// the job's streams are idle: its slot can take the next job while the caller still holds this one's results
		std::lock_guard<std::mutex> lk(job->eng->m);
		job->eng->busy[job->slot] = 0;
	}
	if (job->rc != TP_OK) tp_global_err = job->err;
	return job->rc;
}

int tp_frames_poll(tp_frames_job* job, int32_t* done)
{
	if (!job || !done) { tp_global_err = "tp_frames_poll: null pointer"; return TP_ERR_INVALID; }
	*done = (job->joined || job->done.load(std::memory_order_acquire)) ? 1 : 0;
	return TP_OK;
}

int tp_frames_counts(tp_frames_job* job, int32_t* n_groups, int64_t* n_events)
{
	if (!job || !job->joined) { tp_global_err = "tp_frames_counts: wait for the job first"; return TP_ERR_INVALID; }
	if (n_groups) *n_groups = (int32_t)job->groups.size();
	if (n_events) *n_events = (int64_t)job->events.size();
	return TP_OK;
}

int tp_frames_targets(tp_frames_job* job, int32_t* status, int64_t* stamps, int32_t* stamp_resizes, uint8_t* has_result, int32_t* group, int32_t* pos)
{
	if (!job || !job->joined) { tp_global_err = "tp_frames_targets: wait for the job first"; return TP_ERR_INVALID; }
	const size_t n = (size_t)job->n;
	if (status) std::memcpy(status, job->status.data(), n * 4);
	if (stamps) std::memcpy(stamps, job->stamp.data(), n * 32);
	if (stamp_resizes) std::memcpy(stamp_resizes, job->stamp_resizes.data(), n * 4);
	if (has_result) std::memcpy(has_result, job->has_result.data(), n);
	if (group) std::memcpy(group, job->group.data(), n * 4);
	if (pos) std::memcpy(pos, job->pos.data(), n * 4);
	return TP_OK;
}

int tp_frames_group(tp_frames_job* job, int32_t g, int32_t* n_targets, int32_t* height, int32_t* width, int64_t* cat_capacity, int64_t* n_cat,
	void** h_block, uint64_t* block_nbytes)
{
	if (!job || !job->joined || g < 0 || g >= (int32_t)job->groups.size()) { tp_global_err = "tp_frames_group: no such group"; return TP_ERR_INVALID; }
	const Group& G = job->groups[g];
	if (n_targets) *n_targets = G.n;
	if (height) *height = G.H;
	if (width) *width = G.W;
	if (cat_capacity) *cat_capacity = G.cat_capacity;
	if (n_cat) *n_cat = G.n_cat;
	if (h_block) *h_block = G.h_block;
	if (block_nbytes) *block_nbytes = G.nbytes;
	return TP_OK;
}

int tp_frames_group_lists(tp_frames_job* job, int32_t g, int64_t* cat_offsets, int64_t* cat_starid, int64_t* target_starid)
{
	if (!job || !job->joined || g < 0 || g >= (int32_t)job->groups.size()) { tp_global_err = "tp_frames_group_lists: no such group"; return TP_ERR_INVALID; }
	const Group& G = job->groups[g];
	if (cat_offsets) std::memcpy(cat_offsets, G.cat_offsets.data(), G.cat_offsets.size() * 8);
	if (cat_starid && G.n_cat) std::memcpy(cat_starid, G.cat_starid.data(), (size_t)G.n_cat * 8);
	if (target_starid) std::memcpy(target_starid, G.target_starid.data(), (size_t)G.n * 8);
	return TP_OK;
}

int tp_frames_events(tp_frames_job* job, int32_t* target, int32_t* code, int32_t* a, int32_t* b, double* value, int32_t* text)
{
	if (!job || !job->joined) { tp_global_err = "tp_frames_events: wait for the job first"; return TP_ERR_INVALID; }
	for (size_t k = 0; k < job->events.size(); ++k) {
		const Event& e = job->events[k];
		if (target) target[k] = e.target;
		if (code) code[k] = e.code;
		if (a) a[k] = e.a;
		if (b) b[k] = e.b;
		if (value) value[k] = e.value;
		if (text) text[k] = e.text;
	}
	return TP_OK;
}

const char* tp_frames_text(tp_frames_job* job, int32_t k)
{
	if (!job || k < 0 || k >= (int32_t)job->texts.size()) return "";
	return job->texts[k].c_str();
}

int tp_frames_release(tp_frames_job* job)
{
	if (!job) return TP_OK;
	(void)tp_frames_wait(job);
	for (auto& G : job->groups) if (G.h_block) { job->eng->pinned.put(G.h_block, G.h_cap); G.h_block = nullptr; }
	delete job;
	return TP_OK;
}

} // extern "C"
