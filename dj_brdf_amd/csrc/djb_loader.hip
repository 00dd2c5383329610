// djb_loader.hip -- end-to-end batch fit of MERL files: what examples/merl_params.cpp:53-67 does per
// file (djb::merl(path) -> djb::tabular(merl, 90) -> two fits), for a list of files, as a pipeline:
//
//   reader threads  --pread-->  pinned host ring  --hipMemcpyAsync-->  HBM raw ring
//                                                   k_merl_convert (stream)  -->  texel tables
//   ... all tables resident ...  one k_fit launch (one workgroup per material)  -->  alphas
//
// The reference spends 0.135 s per file in fstream::read + 0.117 s in the fit, serially
// (SURVEY.md section 6); here file reads, PCIe uploads and the conversion kernel overlap, and the
// fit of the whole batch is one 2 ms launch.  Files are independent: with several GPUs each
// context gets a share of the list (dj_brdf_amd/merl_params.py), no collective.
#include "../../include/djb_hip.h"
#include "djb_internal.hpp"
#include "djb_cpu.hpp"
#include "djb_merl_file.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

extern "C" {
// implemented in djb_host.hip
djb_status djb_fit_brdf_batch(djb_ctx *, int, const djb_brdf *const *, int, int, float *, float *, float *,
                              float *, float *, float *, float *);
djb_status djb_brdf_destroy(djb_brdf *);
}

namespace djbk {
// defined in djb_host.hip: wrap an already converted texel table into a djb_brdf (takes ownership)
djb_status wrap_merl_table(djb_ctx *ctx, djbdev::MerlTexel *table, djb_brdf **out, bool own);
// the same for a per-slot (sparse) texel array of the file-fit pipeline (djbdev::Brdf::merl_sparse)
djb_status wrap_merl_slots(djb_ctx *ctx, djbdev::MerlTexel *slots, djb_brdf **out);
int ctx_option_fit_files_dense(djb_ctx *ctx);
// the slot of the context that holds this file's LoaderState (freed by djb_ctx_destroy through free_fn)
void **ctx_loader_state(djb_ctx *ctx, void (*free_fn)(void *));
hipStream_t ctx_stream(djb_ctx *ctx);
int ctx_device(djb_ctx *ctx);
// every entry point that enqueues on the ctx stream holds the context's call mutex (see djb_ctx)
void ctx_lock(djb_ctx *ctx);
void ctx_unlock(djb_ctx *ctx);
} // namespace djbk

namespace {

constexpr long long MERL_N = 90LL * 90 * 180;
constexpr size_t PAYLOAD = sizeof(double) * 3 * MERL_N;   // 34 992 000 bytes after the 12-byte header
constexpr size_t CHUNK = 4u << 20;                        // pinned staging granularity
constexpr int CHUNKS_PER_FILE = (int)((PAYLOAD + CHUNK - 1) / CHUNK);

// one pinned staging chunk
struct Chunk {
	char *host = nullptr;         // pinned, CHUNK bytes
	hipEvent_t done = nullptr;    // its upload has finished -> reusable
	int file = -1, part = -1;
	size_t bytes = 0;
	int state = 0;                // 0 free, 1 being filled, 2 filled, 3 upload in flight
	djb_status st = DJB_OK;
	std::string err;
};

double now_s()
{
	return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// one chunk of a file's payload; part 0 also validates the header.  Same checks and messages as djb::merl::merl
// (dj_brdf.h:963-983); the header is untrusted: 64-bit product of positive dims, MERL shape only.
djb_status read_part(const char *path, int part, char *dst, size_t *bytes, std::string *err)
{
	char buf[256];
	int fd = open(path, O_RDONLY);
	if (fd < 0) { snprintf(buf, sizeof buf, "djb_error: Failed to open %s\n", path); *err = buf; return DJB_ERR_OPEN_FAILED; }
	if (part == 0) {
		int32_t dims[3] = { 0, 0, 0 };
		ssize_t got = pread(fd, dims, 12, 0);
		const bool positive = got == 12 && dims[0] > 0 && dims[1] > 0 && dims[2] > 0;
		long long n = positive ? (long long)dims[0] * (long long)dims[1] * (long long)dims[2] : 0;
		if (n <= 0) { close(fd); *err = "djb_error: Failed to read MERL header\n"; return DJB_ERR_BAD_HEADER; }
		if (n != MERL_N) {
			close(fd);
			snprintf(buf, sizeof buf, "djb_error: MERL table has %lld samples per channel, expected %lld\n", n, MERL_N);
			*err = buf; return DJB_ERR_BAD_HEADER;
		}
	}
	const size_t begin = (size_t)part * CHUNK, want = PAYLOAD - begin < CHUNK ? PAYLOAD - begin : CHUNK;
	size_t off = 0;
	while (off < want) {
		ssize_t r = pread(fd, dst + off, want - off, 12 + (off_t)(begin + off));
		if (r <= 0) break;
		off += (size_t)r;
	}
	close(fd);
	if (off != want) { snprintf(buf, sizeof buf, "djb_error: Reading %s failed\n", path); *err = buf; return DJB_ERR_READ_FAILED; }
	*bytes = want;
	return DJB_OK;
}


// ---- releasing the file mappings of a gather, off the caller's critical path: a short-lived thread per call unmaps them
// (~0.08 ms of kernel work per 35 MB mapping); the library's static destructor waits for the stragglers, so that neither
// exit() nor dlclose() pulls the code out from under one.
struct Reaper {
	std::atomic<int> live{ 0 };
	~Reaper() { while (live.load() > 0) std::this_thread::sleep_for(std::chrono::milliseconds(1)); }
};
Reaper g_reaper;
void reap_mappings(std::vector<std::vector<std::pair<void *, size_t>>> &&kept)
{
	size_t n = 0;
	for (auto &v : kept) n += v.size();
	if (!n) return;
	auto maps = std::make_shared<std::vector<std::vector<std::pair<void *, size_t>>>>(std::move(kept));
	g_reaper.live.fetch_add(1);
	try {
		std::thread([maps]() {
			// paced: every munmap holds the process's mmap_lock for writing for ~80 us (8.5 k page-table entries to tear down);
			// back to back they starve the page faults of every other thread -- the caller's next 15 ms, measured -- so the
			// reaper yields between two mappings
			for (auto &v : *maps) for (auto &m : v) { munmap(m.first, m.second); std::this_thread::sleep_for(std::chrono::microseconds(200)); }
			g_reaper.live.fetch_sub(1);
		}).detach();
	} catch (...) {                     // no thread to be had: unmap here
		g_reaper.live.fetch_sub(1);
		for (auto &v : *maps) for (auto &m : v) munmap(m.first, m.second);
	}
}

// ---- what the sparse form keeps between calls, per context (all of it used under the context's call mutex) -----
// A call on 100 files spends ~2.2 ms gathering; creating 31 threads (0.6-1.2 ms before the last one starts), pinning a
// 6.6 MB buffer (0.4-0.9 ms), computing the slot plan on the GPU (0.27 ms) and releasing all of it again used to cost as
// much once more: the threads now park between calls, the buffers grow and stay, the plan is cached per resolution.
class WorkerPool {
	std::vector<std::thread> th;
	std::mutex mu;
	std::condition_variable cv, done_cv;
	const std::function<void()> *job = nullptr;
	unsigned long long gen = 0;
	int want = 0, pending = 0;
	bool stop = false;
	void loop(int id)
	{
		unsigned long long seen = 0;
		for (;;) {
			const std::function<void()> *fn = nullptr;
			{
				std::unique_lock<std::mutex> lk(mu);
				cv.wait(lk, [&] { return stop || gen != seen; });
				if (stop) return;
				seen = gen;
				if (id < want) fn = job;
			}
			if (fn) {
				(*fn)();                                  // never throws (the job catches)
				std::lock_guard<std::mutex> lk(mu);
				if (--pending == 0) done_cv.notify_all();
			}
		}
	}
public:
	// runs fn on `n` threads (the caller is one of them) and returns when all have returned
	void run(int n, const std::function<void()> &fn)
	{
		const int helpers = n - 1;
		if (helpers > 0) {
			{ std::lock_guard<std::mutex> lk(mu); job = &fn; want = helpers; pending = helpers; ++gen; }
			cv.notify_all();
		}
		// threads the pool does not have yet join the job as they come up (creating 31 of them takes ~1 ms: the first call's
		// gather is under way meanwhile)
		while ((int)th.size() < helpers) {
			try { th.emplace_back(&WorkerPool::loop, this, (int)th.size()); }
			catch (...) { std::lock_guard<std::mutex> lk(mu); pending -= helpers - (int)th.size(); break; }   // fewer threads, same result
		}
		fn();
		if (helpers > 0) {
			std::unique_lock<std::mutex> lk(mu);
			done_cv.wait(lk, [&] { return pending == 0; });
			job = nullptr;
		}
	}
	~WorkerPool()
	{
		{ std::lock_guard<std::mutex> lk(mu); stop = true; }
		cv.notify_all();
		for (std::thread &t : th) t.join();
	}
};
struct LoaderState {
	std::map<int, djbfile::SlotPlan> plans;              // by fit resolution
	djbdev::MerlTexel *host = nullptr, *dev = nullptr;   // [file][slot] texels: pinned staging and its HBM copy, grow-only
	size_t bytes = 0;
	int device = 0;
	WorkerPool pool;
	~LoaderState()
	{
		(void)hipSetDevice(device);
		if (host) (void)hipHostFree(host);
		if (dev) (void)hipFree(dev);
	}
};
void free_loader_state(void *p) { delete (LoaderState *)p; }
LoaderState *loader_state(djb_ctx *ctx)
{
	void **slot = djbk::ctx_loader_state(ctx, free_loader_state);
	if (!*slot) { LoaderState *ls = new LoaderState(); ls->device = djbk::ctx_device(ctx); *slot = ls; }
	return (LoaderState *)*slot;
}

// ---- the sparse form: fetch only what the fit reads ----------------------------------------------------------
// djb::tabular(merl, res) evaluates its source at a fixed set of directions (djb_device.hpp: fit_merl_slot_count):
// cnt back-scattering configurations + the (theta_d, theta_h) Fresnel pairs, 5 545 of a MERL file's 4 374 000
// doubles at res 90.  The reference loads all 35 MB to read them (0.135 s per file) and round 1
// moved all 35 MB over PCIe.  Here the table indices of the slots are computed once on the GPU (the same device
// code the fit kernel would run: k_fit_merl_slots), worker threads map each file and gather just those entries
// into per-slot texels -- float(double(sample) * channel scale), below-horizon bins zeroed, exactly what
// k_merl_convert writes for them (dj_brdf.h:1010-1023) -- and 97 KB per material goes to the GPU instead of 35 MB.
// Same alphas, bit for bit (tests/test_gpu_golden.py::test_native_file_pipeline compares the two forms).
djb_status fit_merl_files_sparse(djb_ctx *ctx, int n_files, const char *const *paths, int res, int shadow, int threads,
                                 float *alpha_beckmann, float *alpha_ggx, double *timing)
{
	hipStream_t stream = djbk::ctx_stream(ctx);
	const double t_begin = now_s();
	const int n_slots = djbk::fit_merl_slots(res);
	LoaderState *ls = loader_state(ctx);
	// ---- which table entries does a fit at this resolution read?  (device code, once per context and resolution: 8 k indices)
	if (!ls->plans.count(res)) {
		int32_t *d_idx = nullptr;
		std::vector<int32_t> idx(n_slots);
		hipError_t e = hipMalloc((void **)&d_idx, sizeof(int32_t) * n_slots);
		if (e == hipSuccess) e = djbk::launch_fit_merl_slots(stream, res, d_idx);
		if (e == hipSuccess) e = hipMemcpyAsync(idx.data(), d_idx, sizeof(int32_t) * n_slots, hipMemcpyDeviceToHost, stream);
		hipError_t se = hipStreamSynchronize(stream);
		if (e == hipSuccess) e = se;
		if (d_idx) (void)hipFree(d_idx);
		if (e != hipSuccess) { (void)hipGetLastError(); return djbk::set_error(DJB_ERR_HIP, "djb_error: fit slot indices: %s", hipGetErrorString(e)); }
		ls->plans[res] = djbfile::make_plan(idx);
	}
	const djbfile::SlotPlan &plan = ls->plans[res];
	const double t_plan = now_s();
	// ---- gather: files are independent -> worker threads; results land in one pinned block [file][slot]
	const size_t bytes = sizeof(djbdev::MerlTexel) * (size_t)n_slots * n_files;
	if (ls->bytes < bytes) {
		(void)hipStreamSynchronize(stream);
		if (ls->host) (void)hipHostFree(ls->host);
		if (ls->dev) (void)hipFree(ls->dev);
		ls->host = ls->dev = nullptr; ls->bytes = 0;
		if (hipHostMalloc((void **)&ls->host, bytes, hipHostMallocDefault) != hipSuccess || hipMalloc((void **)&ls->dev, bytes) != hipSuccess) {
			(void)hipGetLastError();
			if (ls->host) (void)hipHostFree(ls->host);
			if (ls->dev) (void)hipFree(ls->dev);
			ls->host = ls->dev = nullptr;
			return djbk::set_error(DJB_ERR_HIP, "djb_error: cannot allocate the per-slot tables of %d files", n_files);
		}
		ls->bytes = bytes;
		memset(ls->host, 0, bytes);          // slots the fit never evaluates are never written either: they stay zero
	}
	djbdev::MerlTexel *host = ls->host, *dev = ls->dev;
	const double t_alloc = now_s();
	if (threads < 1) {
		// measured on the GPU box, 100 files (profiles/r02/fit_files_rates.txt): 1 thread 30 ms, 4: 11.4, 8: 9.8, 16: 10.4
		// round 3 (mappings released outside the loop; 100 files, profiles/r03/fit_files_rates.txt): 8 threads 6.3 ms, 16: 5.9, 32: 4.1, 64: 4.7
		unsigned hc = std::thread::hardware_concurrency();
		threads = hc >= 128 ? 32 : hc > 8 ? (int)(hc / 4) : hc > 1 ? (int)hc - 1 : 1;
		if (const char *ev = getenv("DJB_READER_THREADS")) { int v = atoi(ev); if (v >= 1 && v <= 256) threads = v; }
	}
	if (threads > n_files) threads = n_files;
	std::atomic<int> next(0);
	std::mutex mu;
	djb_status status = DJB_OK; std::string status_msg; int status_file = n_files;
	// Mappings are NOT unmapped inside the gather loop: munmap takes the process's mmap_lock for writing and broadcasts a
	// TLB shootdown to every core that runs one of its threads, which serialised the readers (profiles/r03/fit_files_rates.txt:
	// 8 -> 32 threads 9.4 -> 10.2 ms with munmap in the loop, 6.3 -> 4.1 ms without; MAP_POPULATE is worse still: 38 ms,
	// it holds the lock while it fills the page tables).  They are released after the alphas are on their way, by a
	// reaper thread (below).
	std::vector<std::vector<std::pair<void *, size_t>>> kept((size_t)threads);
	std::atomic<int> worker_id(0);
	// no exception may leave a worker thread (std::terminate): allocation failures inside one become the call's status
	auto worker = [&]() {
		auto &keep = kept[(size_t)worker_id.fetch_add(1)];
		for (;;) {
			const int f = next.fetch_add(1);
			if (f >= n_files) return;
			djb_status st = DJB_OK;
			std::string err;
			try { st = djbfile::gather_file(paths[f], plan, (float *)(host + (size_t)f * n_slots), &err, &keep); }
			catch (const std::bad_alloc &) { st = DJB_ERR_OUT_OF_MEMORY; }
			catch (...) { st = DJB_ERR_INTERNAL; }
			if (st != DJB_OK) {
				std::lock_guard<std::mutex> lk(mu);
				try { if (f < status_file) { status_file = f; status = st; status_msg = err.empty() ? "djb_error: out of host memory" : err; } }   // the reference stops at the first bad file
				catch (...) { status = st; }
			}
		}
	};
	ls->pool.run(threads, worker);
	const double t_loaded0 = now_s();
	hipError_t e = hipSuccess;
	if (status == DJB_OK) {
		e = hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, stream);
		hipError_t se = hipStreamSynchronize(stream);
		if (e == hipSuccess) e = se;
		if (e != hipSuccess) { (void)hipGetLastError(); status = DJB_ERR_HIP; status_msg = std::string("djb_error: upload failed: ") + hipGetErrorString(e); }
	}
	const double t_loaded = now_s();
	std::vector<djb_brdf *> mats(n_files, nullptr);
	for (int k = 0; k < n_files && status == DJB_OK; ++k) status = djbk::wrap_merl_slots(ctx, dev + (size_t)k * n_slots, &mats[k]);
	if (status == DJB_OK && status_msg.empty())
		status = djb_fit_brdf_batch(ctx, n_files, mats.data(), res, shadow, alpha_beckmann, alpha_ggx, nullptr, nullptr, nullptr, nullptr, nullptr);
	const double t_end = now_s();
	if (getenv("DJB_LOADER_TRACE")) {
		// how scattered is the plan?  distinct 4 KB pages / 64 KB fault-around windows one file's gather touches (3 channel planes)
		size_t pages = 0, windows = 0; long long lp = -1, lw = -1;
		for (int ch = 0; ch < 3; ++ch) for (int32_t i : plan.idx) {
			const long long byte = 12 + 8 * ((long long)i + (long long)ch * MERL_N), pg = byte >> 12, w = byte >> 16;
			if (pg != lp) { ++pages; lp = pg; }
			if (w != lw) { ++windows; lw = w; }
		}
		fprintf(stderr, "djb_loader: %zu entries per channel on %zu pages / %zu 64-KB windows per file\n", plan.idx.size(), pages, windows);
	}
	if (getenv("DJB_LOADER_TRACE"))
		fprintf(stderr, "djb_loader: plan %.3f ms, alloc %.3f, gather (%d threads) %.3f, upload %.3f, fit %.3f\n", 1e3 * (t_plan - t_begin),
		        1e3 * (t_alloc - t_plan), threads, 1e3 * (t_loaded0 - t_alloc), 1e3 * (t_loaded - t_loaded0), 1e3 * (t_end - t_loaded));
	reap_mappings(std::move(kept));
	for (djb_brdf *b : mats) if (b) djb_brdf_destroy(b);
	djbfile::t_failed_file = status != DJB_OK && status_file < n_files ? status_file : -1;
	if (status != DJB_OK) return status_msg.empty() ? status : djbk::set_error(status, "%s", status_msg.c_str());
	if (timing) {
		timing[0] = t_end - t_begin; timing[1] = t_loaded - t_begin; timing[2] = t_end - t_loaded;
		timing[3] = (double)n_files * (double)plan.idx.size() * 24.0;       // bytes actually read from the files
	}
	return DJB_OK;
}

} // namespace

static djb_status fit_merl_files(djb_ctx *ctx, int n_files, const char *const *paths, int res, int shadow,
                                 int reader_threads, float *alpha_beckmann, float *alpha_ggx, double *timing);

extern "C" djb_status djb_set_file_map_observer(void (*fn)(const char *path, void *user), void *user)
{
	djbfile::g_map_observer_user.store(user, std::memory_order_release);
	djbfile::g_map_observer.store(fn, std::memory_order_release);
	return DJB_OK;
}

extern "C" djb_status djb_fit_merl_files(djb_ctx *ctx, int n_files, const char *const *paths, int res, int shadow,
                                         int reader_threads, float *alpha_beckmann, float *alpha_ggx,
                                         double *timing /* optional [4]: total, read+upload, fit, bytes */)
try {
	return fit_merl_files(ctx, n_files, paths, res, shadow, reader_threads, alpha_beckmann, alpha_ggx, timing);
}
catch (const std::bad_alloc &) { return djbk::set_error(DJB_ERR_OUT_OF_MEMORY, "djb_error: out of host memory"); }
catch (...) { return djbk::set_error(DJB_ERR_INTERNAL, "djb_error: internal error in djb_fit_merl_files"); }

// ---- the same job over SEVERAL contexts (examples/merl_params.cpp:53-69 is the loop it stands for): file k belongs to
// context k mod n_ctx, every context's share is one fit_merl_files call on a host thread of its own (the calling thread takes context 0),
// rows come back in input order.  No exchange between contexts -- the fits are independent.
static djb_status fit_merl_files_multi(djb_ctx *const *ctxs, int n_ctx, int n_files, const char *const *paths, int res, int shadow,
                                       int reader_threads, float *alpha_beckmann, float *alpha_ggx, double *timing)
{
	if (!ctxs || n_ctx < 1 || !paths || n_files < 0 || !alpha_beckmann || !alpha_ggx)
		return djbk::set_error(DJB_ERR_INVALID_ARGUMENT, "djb_error: invalid argument");
	for (int g = 0; g < n_ctx; ++g) {
		if (!ctxs[g]) return djbk::set_error(DJB_ERR_INVALID_ARGUMENT, "djb_error: null context in the context list");
		for (int h = 0; h < g; ++h)
			if (ctxs[h] == ctxs[g]) return djbk::set_error(DJB_ERR_INVALID_ARGUMENT, "djb_error: a context is listed twice (two shares would serialise on its lock)");
	}
	if (timing) for (int k = 0; k < 4 * n_ctx; ++k) timing[k] = 0.0;
	if (n_files == 0) return DJB_OK;
	const int used = std::min(n_ctx, n_files);
	struct Share { std::vector<const char *> paths; std::vector<float> ab, ag; djb_status st = DJB_OK; std::string msg; int bad = -1; };
	std::vector<Share> shares(used);
	for (int k = 0; k < n_files; ++k) shares[k % used].paths.push_back(paths[k]);
	auto run = [&](int g) {
		Share &s = shares[g];
		s.ab.resize(s.paths.size()); s.ag.resize(s.paths.size());
		djbfile::t_failed_file = -1;
		try { s.st = fit_merl_files(ctxs[g], (int)s.paths.size(), s.paths.data(), res, shadow, reader_threads, s.ab.data(), s.ag.data(), timing ? timing + 4 * g : nullptr); }
		catch (const std::bad_alloc &) { s.st = djbk::set_error(DJB_ERR_OUT_OF_MEMORY, "djb_error: out of host memory"); }
		catch (...) { s.st = djbk::set_error(DJB_ERR_INTERNAL, "djb_error: internal error in djb_fit_merl_files_multi"); }
		if (s.st != DJB_OK) { s.msg = djb_last_error(); s.bad = djbfile::t_failed_file; }     // the message is the worker thread's: carried over below
	};
	std::vector<std::thread> workers;
	for (int g = 1; g < used; ++g) workers.emplace_back(run, g);
	run(0);
	for (std::thread &t : workers) t.join();
	// the reference's loop stops at the first bad file: the error of the lowest-indexed bad file over all shares wins (a failure that is
	// not a file's -- a device error -- ranks before every file of its context)
	int best = -1; long long best_index = 0;
	for (int g = 0; g < used; ++g) {
		if (shares[g].st == DJB_OK) continue;
		const long long index = shares[g].bad >= 0 ? (long long)shares[g].bad * used + g : (long long)g - used;
		if (best < 0 || index < best_index) { best = g; best_index = index; }
	}
	if (best >= 0) return djbk::set_error(shares[best].st, "%s", shares[best].msg.c_str());
	for (int g = 0; g < used; ++g)
		for (size_t j = 0; j < shares[g].paths.size(); ++j) { alpha_beckmann[j * used + g] = shares[g].ab[j]; alpha_ggx[j * used + g] = shares[g].ag[j]; }
	return DJB_OK;
}

extern "C" djb_status djb_fit_merl_files_multi(djb_ctx *const *ctxs, int n_ctx, int n_files, const char *const *paths, int res, int shadow,
                                               int reader_threads, float *alpha_beckmann, float *alpha_ggx, double *timing)
try {
	return fit_merl_files_multi(ctxs, n_ctx, n_files, paths, res, shadow, reader_threads, alpha_beckmann, alpha_ggx, timing);
}
catch (const std::bad_alloc &) { return djbk::set_error(DJB_ERR_OUT_OF_MEMORY, "djb_error: out of host memory"); }
catch (...) { return djbk::set_error(DJB_ERR_INTERNAL, "djb_error: internal error in djb_fit_merl_files_multi"); }

static djb_status fit_merl_files(djb_ctx *ctx, int n_files, const char *const *paths, int res, int shadow,
                                 int reader_threads, float *alpha_beckmann, float *alpha_ggx, double *timing)
{
	if (!ctx || !paths || n_files < 0 || !alpha_beckmann || !alpha_ggx)
		return djbk::set_error(DJB_ERR_INVALID_ARGUMENT, "djb_error: invalid argument");
	if (n_files == 0) return DJB_OK;
	if (djbcpu::is_cpu(ctx)) return djbcpu::fit_merl_files(ctx, n_files, paths, res, shadow, reader_threads, alpha_beckmann, alpha_ggx, timing);
	hipError_t e = hipSetDevice(djbk::ctx_device(ctx));
	if (e != hipSuccess) return djbk::set_error(DJB_ERR_HIP, "djb_error: hipSetDevice: %s", hipGetErrorString(e));
	hipStream_t stream = djbk::ctx_stream(ctx);
	struct CallLock { djb_ctx *c; explicit CallLock(djb_ctx *c_) : c(c_) { djbk::ctx_lock(c); } ~CallLock() { djbk::ctx_unlock(c); } } call_lock(ctx);
	// default: fetch only the entries the fit reads (fit_merl_files_sparse); DJB_OPT_FIT_FILES_DENSE / DJB_FIT_FILES_DENSE=1
	// uploads and converts every table in full, as round 1 did -- same alphas, and what a caller that goes on to
	// evaluate the tables would want
	const char *dense_env = getenv("DJB_FIT_FILES_DENSE");
	if (!(djbk::ctx_option_fit_files_dense(ctx) || (dense_env && atoi(dense_env) != 0)))
		return fit_merl_files_sparse(ctx, n_files, paths, res, shadow, reader_threads, alpha_beckmann, alpha_ggx, timing);
	const double t_begin = now_s();
	// Readers copy file chunks from the page cache into pinned memory; the consumer below feeds them to the DMA engine.
	// Measured on the GPU box (2 x EPYC 9575F, 100 files, profiles/r02/fit_files_rates.txt): 2 readers 88 ms = 40 GB/s
	// (PCIe Gen5 takes ~56 GB/s one way, tools/pcie_probe.hip), 4 readers 108 ms, 8 readers 123 ms -- more readers
	// only add cross-socket memory traffic.  Round 1 staged whole files (4 x 35 MB pinned slots): 125-143 ms, of which
	// ~35 ms went into pinning and unpinning the slots.  DJB_READER_THREADS overrides.
	if (reader_threads < 1) {
		reader_threads = 2;
		if (const char *ev = getenv("DJB_READER_THREADS")) { int v = atoi(ev); if (v >= 1 && v <= 64) reader_threads = v; }
	}
	const long long n_parts = (long long)n_files * CHUNKS_PER_FILE;
	if ((long long)reader_threads > n_parts) reader_threads = (int)n_parts;
	// Staging is a ring of 4 MiB pinned chunks, not of whole files: pinning memory is what a first call pays for
	// (hipHostMalloc of ten 35 MB slots cost ~35 ms of a 110 ms job), and 4 MiB copies already run at link speed.
	const int n_chunks = (int)std::min<long long>(n_parts, std::min(3LL * reader_threads + 2, 16LL));
	const int n_raw = n_files < 6 ? n_files : 6;                  // raw payloads (doubles) in HBM awaiting conversion

	std::vector<Chunk> chunks(n_chunks);
	char *pinned = nullptr;
	double *raw = nullptr;                                        // n_raw x PAYLOAD
	std::vector<int> parts_up(n_files, 0);
	djbdev::MerlTexel *all_tables = nullptr;
	djb_status status = DJB_OK;
	std::string status_msg;
	int status_file = n_files;                                    // the reference stops at the first bad file: report the lowest index
	auto cleanup = [&]() {
		for (Chunk &c : chunks) if (c.done) (void)hipEventDestroy(c.done);
		if (pinned) (void)hipHostFree(pinned);
		if (raw) (void)hipFree(raw);
		if (all_tables) (void)hipFree(all_tables);
	};
	bool ok = hipMalloc((void **)&all_tables, sizeof(djbdev::MerlTexel) * (size_t)MERL_N * n_files) == hipSuccess &&
	          hipMalloc((void **)&raw, PAYLOAD * (size_t)n_raw) == hipSuccess &&
	          hipHostMalloc((void **)&pinned, CHUNK * (size_t)n_chunks, hipHostMallocDefault) == hipSuccess;
	for (int k = 0; ok && k < n_chunks; ++k) {
		chunks[k].host = pinned + CHUNK * (size_t)k;
		ok = hipEventCreateWithFlags(&chunks[k].done, hipEventDisableTiming) == hipSuccess;
	}
	if (!ok) {
		(void)hipGetLastError();
		cleanup();
		return djbk::set_error(DJB_ERR_HIP, "djb_error: cannot allocate the upload ring / %d MERL tables in HBM", n_files);
	}

	// ---- producer side: reader threads claim (file, part) items in order and a free pinned chunk each
	std::mutex mu;
	std::condition_variable cv;
	long long next_part = 0;
	bool abort_flag = false;
	auto reader = [&]() {
		for (;;) {
			int slot = -1; long long item;
			{
				std::unique_lock<std::mutex> lk(mu);
				cv.wait(lk, [&] {
					if (abort_flag || next_part >= n_parts) return true;
					for (int s = 0; s < n_chunks; ++s) if (chunks[s].state == 0) return true;
					return false;
				});
				if (abort_flag || next_part >= n_parts) return;
				for (int s = 0; s < n_chunks; ++s) if (chunks[s].state == 0) { slot = s; break; }
				item = next_part++;
				chunks[slot].state = 1; chunks[slot].file = (int)(item / CHUNKS_PER_FILE); chunks[slot].part = (int)(item % CHUNKS_PER_FILE);
			}
			std::string err; size_t bytes = 0;
			djb_status st;
			try { st = read_part(paths[chunks[slot].file], chunks[slot].part, chunks[slot].host, &bytes, &err); }
			catch (...) { st = DJB_ERR_OUT_OF_MEMORY; }              // nothing may leave a reader thread (std::terminate)
			{
				std::lock_guard<std::mutex> lk(mu);
				chunks[slot].st = st; chunks[slot].bytes = bytes; chunks[slot].state = 2;
				try { chunks[slot].err = err; } catch (...) { chunks[slot].st = DJB_ERR_OUT_OF_MEMORY; }
			}
			cv.notify_all();
		}
	};
	std::vector<std::thread> readers;
	for (int t = 0; t < reader_threads; ++t) { try { readers.emplace_back(reader); } catch (...) { if (readers.empty()) throw; break; } }   // fewer readers, same result

	// ---- consumer side (this thread): upload filled chunks, convert a file once its last chunk is on its way
	long long uploaded = 0;
	while (uploaded < n_parts && status == DJB_OK) {
		int slot = -1;
		{
			std::unique_lock<std::mutex> lk(mu);
			cv.wait_for(lk, std::chrono::microseconds(200), [&] {
				for (int s = 0; s < n_chunks; ++s) if (chunks[s].state == 2) return true;
				return false;
			});
			// oldest filled chunk first, so that files complete (and free their raw slot) in order
			// (a chunk of file F may only go to raw slot F % n_raw once file F - n_raw has been handed to the
			// conversion kernel: a slow reader can hold a part of an old file while the others run far ahead)
			long long best = -1;
			for (int s = 0; s < n_chunks; ++s)
				if (chunks[s].state == 2) {
					const int F = chunks[s].file;
					if (chunks[s].st == DJB_OK && F >= n_raw && parts_up[F - n_raw] < CHUNKS_PER_FILE) continue;
					long long id = (long long)F * CHUNKS_PER_FILE + chunks[s].part;
					if (best < 0 || id < best) { best = id; slot = s; }
				}
		}
		// recycle chunks whose upload has completed
		bool freed = false;
		for (int s = 0; s < n_chunks; ++s) {
			bool inflight;
			{ std::lock_guard<std::mutex> lk(mu); inflight = chunks[s].state == 3; }
			if (inflight && hipEventQuery(chunks[s].done) == hipSuccess) {
				{ std::lock_guard<std::mutex> lk(mu); chunks[s].state = 0; }
				freed = true;
			}
		}
		if (freed) cv.notify_all();
		if (slot < 0) continue;
		Chunk &c = chunks[slot];
		if (c.st != DJB_OK) {
			if (c.file < status_file) { status_file = c.file; status = c.st; status_msg = c.err; }
			break;
		}
		// raw slot r was last used by file c.file - n_raw, whose conversion kernel is already enqueued on this stream
		// (checked above): the copy below is ordered after it
		const int r = c.file % n_raw;
		char *dst = (char *)raw + PAYLOAD * (size_t)r + CHUNK * (size_t)c.part;
		e = hipMemcpyAsync(dst, c.host, c.bytes, hipMemcpyHostToDevice, stream);
		if (e == hipSuccess) e = hipEventRecord(c.done, stream);
		if (e == hipSuccess && ++parts_up[c.file] == CHUNKS_PER_FILE) {
			// one block for the batch (a hipMalloc per file costs ~0.2 ms)
			e = djbk::launch_merl_convert(stream, (const double *)((char *)raw + PAYLOAD * (size_t)r), MERL_N, all_tables + (size_t)c.file * MERL_N);
		}
		if (e != hipSuccess) {
			status = DJB_ERR_HIP; status_msg = std::string("djb_error: upload failed: ") + hipGetErrorString(e);
			break;
		}
		{ std::lock_guard<std::mutex> lk(mu); c.state = 3; }
		++uploaded;
	}
	{ std::lock_guard<std::mutex> lk(mu); abort_flag = true; }
	cv.notify_all();
	for (std::thread &t : readers) t.join();
	if (status != DJB_OK || uploaded < n_parts) {
		// a lower-indexed file may have failed in a chunk that was filled but not yet consumed
		for (Chunk &c : chunks) if (c.state == 2 && c.st != DJB_OK && c.file < status_file) { status_file = c.file; status = c.st; status_msg = c.err; }
	}
	// uploads / conversions may still be in flight, also on the failure path: the ring must not be freed under them
	{
		hipError_t se = hipStreamSynchronize(stream);
		if (se != hipSuccess) (void)hipGetLastError();
		if (status == DJB_OK && se != hipSuccess) { status = DJB_ERR_HIP; status_msg = "djb_error: stream sync failed"; }
	}
	const double t_loaded = now_s();
	djbfile::t_failed_file = status != DJB_OK && status_file < n_files ? status_file : -1;
	if (status != DJB_OK) { cleanup(); return djbk::set_error(status, "%s", status_msg.c_str()); }

	// ---- one fit launch for the whole batch
	std::vector<djb_brdf *> mats(n_files, nullptr);
	for (int k = 0; k < n_files && status == DJB_OK; ++k)
		status = djbk::wrap_merl_table(ctx, all_tables + (size_t)k * MERL_N, &mats[k], false);   // views into all_tables
	if (status == DJB_OK)
		status = djb_fit_brdf_batch(ctx, n_files, mats.data(), res, shadow, alpha_beckmann, alpha_ggx,
		                            nullptr, nullptr, nullptr, nullptr, nullptr);
	const double t_end = now_s();
	for (djb_brdf *b : mats) if (b) djb_brdf_destroy(b);
	cleanup();
	if (timing) {
		timing[0] = t_end - t_begin; timing[1] = t_loaded - t_begin; timing[2] = t_end - t_loaded;
		timing[3] = (double)n_files * (double)(PAYLOAD + 12);
	}
	return status;
}
