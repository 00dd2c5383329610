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

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include <fcntl.h>
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
hipStream_t ctx_stream(djb_ctx *ctx);
int ctx_device(djb_ctx *ctx);
// every entry point that enqueues on the ctx stream holds the context's call mutex (see djb_ctx)
void ctx_lock(djb_ctx *ctx);
void ctx_unlock(djb_ctx *ctx);
} // namespace djbk

namespace {

constexpr long long MERL_N = 90LL * 90 * 180;
constexpr size_t PAYLOAD = sizeof(double) * 3 * MERL_N;   // 34 992 000 bytes after the 12-byte header

struct Slot {
	double *host = nullptr;       // pinned
	double *dev = nullptr;        // raw payload in HBM
	hipEvent_t done = nullptr;    // conversion finished -> slot reusable
	int file = -1;
	int state = 0;                // 0 free, 1 being filled, 2 filled, 3 in flight on the GPU
	djb_status st = DJB_OK;
	std::string err;
};

double now_s()
{
	return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// same checks and messages as djb::merl::merl (dj_brdf.h:963-983)
djb_status read_payload(const char *path, double *dst, std::string *err)
{
	char buf[256];
	int fd = open(path, O_RDONLY);
	if (fd < 0) { snprintf(buf, sizeof buf, "djb_error: Failed to open %s\n", path); *err = buf; return DJB_ERR_OPEN_FAILED; }
	int32_t dims[3] = { 0, 0, 0 };
	ssize_t got = pread(fd, dims, 12, 0);
	// the header is untrusted: 64-bit product of positive dims only (an int32 product overflows: UB)
	const bool positive = got == 12 && dims[0] > 0 && dims[1] > 0 && dims[2] > 0;
	long long n = positive ? (long long)dims[0] * (long long)dims[1] * (long long)dims[2] : 0;
	if (n <= 0) { close(fd); *err = "djb_error: Failed to read MERL header\n"; return DJB_ERR_BAD_HEADER; }
	if (n != MERL_N) {
		close(fd);
		snprintf(buf, sizeof buf, "djb_error: MERL table has %lld samples per channel, expected %lld\n", n, MERL_N);
		*err = buf; return DJB_ERR_BAD_HEADER;
	}
	size_t off = 0;
	while (off < PAYLOAD) {
		ssize_t r = pread(fd, (char *)dst + off, PAYLOAD - off, 12 + (off_t)off);
		if (r <= 0) break;
		off += (size_t)r;
	}
	close(fd);
	if (off != PAYLOAD) { snprintf(buf, sizeof buf, "djb_error: Reading %s failed\n", path); *err = buf; return DJB_ERR_READ_FAILED; }
	return DJB_OK;
}

} // namespace

static djb_status fit_merl_files(djb_ctx *ctx, int n_files, const char *const *paths, int res, int shadow,
                                 int reader_threads, float *alpha_beckmann, float *alpha_ggx, double *timing);

extern "C" djb_status djb_fit_merl_files(djb_ctx *ctx, int n_files, const char *const *paths, int res, int shadow,
                                         int reader_threads, float *alpha_beckmann, float *alpha_ggx,
                                         double *timing /* optional [4]: total, read+upload, fit, bytes */)
try {
	return fit_merl_files(ctx, n_files, paths, res, shadow, reader_threads, alpha_beckmann, alpha_ggx, timing);
}
catch (const std::bad_alloc &) { return djbk::set_error(DJB_ERR_OUT_OF_MEMORY, "djb_error: out of host memory"); }
catch (...) { return djbk::set_error(DJB_ERR_INTERNAL, "djb_error: internal error in djb_fit_merl_files"); }

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
	const double t_begin = now_s();
	const int n_slots = n_files < 4 ? n_files : 4;
	if (reader_threads < 1) reader_threads = 4;
	if (reader_threads > n_slots) reader_threads = n_slots;

	std::vector<Slot> slots(n_slots);
	std::vector<djbdev::MerlTexel *> tables(n_files, nullptr);   // views into all_tables
	djbdev::MerlTexel *all_tables = nullptr;
	djb_status status = DJB_OK;
	std::string status_msg;
	auto cleanup = [&]() {
		for (Slot &s : slots) {
			if (s.host) (void)hipHostFree(s.host);
			if (s.dev) (void)hipFree(s.dev);
			if (s.done) (void)hipEventDestroy(s.done);
		}
		if (all_tables) (void)hipFree(all_tables);
	};
	if (hipMalloc((void **)&all_tables, sizeof(djbdev::MerlTexel) * (size_t)MERL_N * n_files) != hipSuccess) {
		(void)hipGetLastError();
		return djbk::set_error(DJB_ERR_HIP, "djb_error: cannot allocate %d MERL tables in HBM", n_files);
	}
	for (Slot &s : slots) {
		if (hipHostMalloc((void **)&s.host, PAYLOAD, hipHostMallocDefault) != hipSuccess ||
		    hipMalloc((void **)&s.dev, PAYLOAD) != hipSuccess ||
		    hipEventCreateWithFlags(&s.done, hipEventDisableTiming) != hipSuccess) {
			cleanup();
			return djbk::set_error(DJB_ERR_HIP, "djb_error: cannot allocate the upload ring");
		}
	}

	// ---- producer side: reader threads claim (file, free slot) pairs in file order
	std::mutex mu;
	std::condition_variable cv;
	int next_file = 0;
	bool abort_flag = false;
	auto reader = [&]() {
		for (;;) {
			int file, slot = -1;
			{
				std::unique_lock<std::mutex> lk(mu);
				cv.wait(lk, [&] {
					if (abort_flag || next_file >= n_files) return true;
					for (int s = 0; s < n_slots; ++s) if (slots[s].state == 0) return true;
					return false;
				});
				if (abort_flag || next_file >= n_files) return;
				for (int s = 0; s < n_slots; ++s) if (slots[s].state == 0) { slot = s; break; }
				file = next_file++;
				slots[slot].state = 1; slots[slot].file = file;
			}
			std::string err;
			djb_status st = read_payload(paths[file], slots[slot].host, &err);
			{
				std::lock_guard<std::mutex> lk(mu);
				slots[slot].st = st; slots[slot].err = err; slots[slot].state = 2;
			}
			cv.notify_all();
		}
	};
	std::vector<std::thread> readers;
	for (int t = 0; t < reader_threads; ++t) readers.emplace_back(reader);

	// ---- consumer side (this thread): upload + convert filled slots, recycle finished ones
	int uploaded = 0;
	while (uploaded < n_files && status == DJB_OK) {
		int slot = -1;
		{
			std::unique_lock<std::mutex> lk(mu);
			cv.wait_for(lk, std::chrono::milliseconds(1), [&] {
				for (int s = 0; s < n_slots; ++s) if (slots[s].state == 2) return true;
				return false;
			});
			for (int s = 0; s < n_slots; ++s) if (slots[s].state == 2) { slot = s; break; }
		}
		// recycle slots whose conversion has completed
		for (int s = 0; s < n_slots; ++s) {
			bool inflight;
			{ std::lock_guard<std::mutex> lk(mu); inflight = slots[s].state == 3; }
			if (inflight && hipEventQuery(slots[s].done) == hipSuccess) {
				{ std::lock_guard<std::mutex> lk(mu); slots[s].state = 0; }
				cv.notify_all();
			}
		}
		if (slot < 0) continue;
		Slot &s = slots[slot];
		if (s.st != DJB_OK) { status = s.st; status_msg = s.err; break; }
		djbdev::MerlTexel *tab = all_tables + (size_t)s.file * MERL_N;   // one block for the batch (a hipMalloc per file costs ~0.2 ms)
		e = hipMemcpyAsync(s.dev, s.host, PAYLOAD, hipMemcpyHostToDevice, stream);
		if (e == hipSuccess) e = djbk::launch_merl_convert(stream, s.dev, MERL_N, tab);
		if (e == hipSuccess) e = hipEventRecord(s.done, stream);
		if (e != hipSuccess) {
			status = DJB_ERR_HIP; status_msg = std::string("djb_error: upload failed: ") + hipGetErrorString(e);
			break;
		}
		tables[s.file] = tab;
		{ std::lock_guard<std::mutex> lk(mu); s.state = 3; }
		++uploaded;
	}
	{ std::lock_guard<std::mutex> lk(mu); abort_flag = true; }
	cv.notify_all();
	for (std::thread &t : readers) t.join();
	// uploads / conversions of earlier slots may still be in flight, also on the failure path: the ring
	// (pinned host + HBM) must not be freed under them
	{
		hipError_t se = hipStreamSynchronize(stream);
		if (se != hipSuccess) (void)hipGetLastError();
		if (status == DJB_OK && se != hipSuccess) { status = DJB_ERR_HIP; status_msg = "djb_error: stream sync failed"; }
	}
	const double t_loaded = now_s();
	if (status != DJB_OK) { cleanup(); return djbk::set_error(status, "%s", status_msg.c_str()); }

	// ---- one fit launch for the whole batch
	std::vector<djb_brdf *> mats(n_files, nullptr);
	for (int k = 0; k < n_files && status == DJB_OK; ++k) {
		status = djbk::wrap_merl_table(ctx, tables[k], &mats[k], false);   // views into all_tables
	}
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
