// djb_merl_file.hpp -- reading what a tabular(merl, res) fit needs out of a MERL file, shared by the GPU file
// pipeline (djb_loader.hip) and the host path (djb_cpu.cpp).  Plain C++ (no HIP).
//
// djb::tabular(merl, res) evaluates its source at a fixed set of directions (djb_device.hpp: fit_merl_slot_count):
// cnt back-scattering configurations + the (theta_d, theta_h) Fresnel pairs -- 5 545 of a file's 4 374 000 doubles
// at res 90.  The reference loads all 35 MB to read them (0.135 s per file).  Here the caller
// computes the table index of every query slot once (the same code the fit runs), and each file is mapped and only
// those entries are turned into per-slot texels: float(double(sample) * channel scale) with below-horizon bins
// zeroed, exactly what k_merl_convert / merl_convert_one write for them (dj_brdf.h:1010-1023).
#pragma once

#include <algorithm>
#include <atomic>
#include <csetjmp>
#include <csignal>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "../../include/djb_hip.h"

namespace djbfile {

constexpr long long MERL_N = 90LL * 90 * 180;
constexpr size_t PAYLOAD = sizeof(double) * 3 * MERL_N;   // 34 992 000 bytes after the 12-byte header

struct SlotPlan { std::vector<int32_t> slot, idx; };     // used slots sorted by table index (monotone walk over the mapping)

// idx[s]: table index query slot s reads, or < 0 for a slot the fit never evaluates
inline SlotPlan make_plan(const std::vector<int32_t> &idx)
{
	std::vector<int32_t> order;
	for (int s = 0; s < (int)idx.size(); ++s) if (idx[s] >= 0 && idx[s] < MERL_N) order.push_back(s);
	std::sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return idx[a] < idx[b] || (idx[a] == idx[b] && a < b); });
	SlotPlan plan;
	for (int32_t s : order) { plan.slot.push_back(s); plan.idx.push_back(idx[s]); }
	return plan;
}

// ---- a file that shrinks under its mapping ---------------------------------------------------------------------------
// The size is checked before the file is mapped, but another process can truncate it afterwards; touching a page that lies
// beyond the new end of file raises SIGBUS, where the reference -- which fread()s -- reports "Reading %s failed"
// (dj_brdf.h:979-982).  pread per entry would give that verdict for free but costs a system call per double (measured:
// 24 files 6 ms mapped vs 190 ms with pread on one thread; process_vm_readv on the own mapping: 8 ms per FILE), so the
// gather keeps the mapping and runs under a guard: a process-wide SIGBUS handler, installed on first use, that jumps back
// into gather_file when -- and only when -- the faulting address lies inside the mapping the current thread is gathering
// from; any other SIGBUS goes to the handler that was installed before (or to the default action).
// Ownership rules of the handler (a host such as Mitsuba or Python's faulthandler may install its own later):
//   * every gather_file entry queries the current disposition (one sigaction call per file) and re-installs the guard if it is
//     no longer ours, chaining to whatever it found -- so a handler installed after ours does not disable the guard, it becomes
//     the one foreign faults are forwarded to;
//   * a fault that is not ours and has no previous handler resets the disposition to the default and returns, so that the
//     faulting access re-executes under it; the next gather_file re-arms the guard;
//   * the per-thread guard pointer lives in initial-exec TLS: reading it from the handler never enters __tls_get_addr / malloc,
//     also for a foreign SIGBUS on a thread that has never run gather_file.
struct BusGuard { sigjmp_buf env; const char *lo, *hi; };
inline thread_local BusGuard *t_bus_guard __attribute__((tls_model("initial-exec"))) = nullptr;
// the handler that was installed before ours: an immutable copy behind an atomic pointer (the handler may read it on any thread while
// install_sigbus_guard publishes a newer one; the few superseded copies are never freed)
inline std::atomic<const struct sigaction *> g_prev_sigbus{nullptr};
inline std::mutex g_sigbus_mu;
// a foreign fault forwarded once that comes back at the same address on the same thread was not resolved by the chain -- e.g. a
// handler installed after ours that disables itself and falls back to ours (Python's faulthandler): the second time the default
// action gets it, instead of bouncing between the two handlers forever
inline thread_local const void *t_bus_last_addr __attribute__((tls_model("initial-exec"))) = nullptr;
inline thread_local int t_bus_repeats __attribute__((tls_model("initial-exec"))) = 0;
inline void on_sigbus(int sig, siginfo_t *si, void *uc)
{
	BusGuard *g = t_bus_guard;
	const char *addr = (const char *)si->si_addr;
	if (g && addr >= g->lo && addr < g->hi) siglongjmp(g->env, 1);
	const bool repeat = addr == t_bus_last_addr && ++t_bus_repeats >= 1;
	if (addr != t_bus_last_addr) { t_bus_last_addr = addr; t_bus_repeats = 0; }
	const struct sigaction *prev = g_prev_sigbus.load(std::memory_order_acquire);
	if (!repeat && prev) {
		if ((prev->sa_flags & SA_SIGINFO) && prev->sa_sigaction && prev->sa_sigaction != on_sigbus) { prev->sa_sigaction(sig, si, uc); return; }
		if (!(prev->sa_flags & SA_SIGINFO) && prev->sa_handler == SIG_IGN) return;
		if (!(prev->sa_flags & SA_SIGINFO) && prev->sa_handler != SIG_DFL) { prev->sa_handler(sig); return; }
	}
	struct sigaction dfl; memset(&dfl, 0, sizeof dfl); dfl.sa_handler = SIG_DFL; sigemptyset(&dfl.sa_mask);
	sigaction(SIGBUS, &dfl, nullptr);          // not ours and nobody else's (or unresolved by the chain): the faulting access re-executes under the default action
}
inline void install_sigbus_guard()
{
	struct sigaction cur;
	if (sigaction(SIGBUS, nullptr, &cur) == 0 && (cur.sa_flags & SA_SIGINFO) && cur.sa_sigaction == on_sigbus) return;
	std::lock_guard<std::mutex> lk(g_sigbus_mu);
	if (sigaction(SIGBUS, nullptr, &cur) == 0 && (cur.sa_flags & SA_SIGINFO) && cur.sa_sigaction == on_sigbus) return;
	struct sigaction sa; memset(&sa, 0, sizeof sa);
	sa.sa_sigaction = on_sigbus; sa.sa_flags = SA_SIGINFO; sigemptyset(&sa.sa_mask);
	g_prev_sigbus.store(new struct sigaction(cur), std::memory_order_release);      // published before the handler that reads it is installed
	sigaction(SIGBUS, &sa, nullptr);
}

// index (within the call's path list) of the file whose error a failed fit_merl_files call reports -- the lowest-indexed bad file --
// for djb_fit_merl_files_multi, which must pick the lowest index over several contexts' shares; -1: the failure was not a file's
inline thread_local int t_failed_file = -1;

// Observer of the file pipeline (djb_set_file_map_observer, include/djb_hip.h): called with the path after a file has passed its
// size check and has been mapped, before the gather.  The library itself never writes to an input file; the tests of the guard
// above register a callback that truncates the file at exactly this point.
typedef void (*map_observer_fn)(const char *path, void *user);
inline std::atomic<map_observer_fn> g_map_observer{nullptr};
inline std::atomic<void *> g_map_observer_user{nullptr};

// out: 3 floats (r, g, b) per slot, slot-major.  Same checks and messages as djb::merl::merl (dj_brdf.h:963-983); the
// header is untrusted (64-bit product of positive dims, MERL shape only); the reference reads the whole payload and
// fails if the file is short (dj_brdf.h:979-982): same verdict here, from the file size.
// `keep` (optional): the mapping is handed back instead of unmapped -- see the caller (djb_loader.hip: unmapping inside
// the gather loop is what stopped the gather from scaling with reader threads).
inline djb_status gather_file(const char *path, const SlotPlan &plan, float *out, std::string *err,
                              std::vector<std::pair<void *, size_t>> *keep = nullptr)
{
	char buf[256];
	int fd = open(path, O_RDONLY);
	if (fd < 0) { snprintf(buf, sizeof buf, "djb_error: Failed to open %s\n", path); *err = buf; return DJB_ERR_OPEN_FAILED; }
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
	struct stat sb;
	if (fstat(fd, &sb) != 0 || (size_t)sb.st_size < 12 + PAYLOAD) {
		close(fd);
		snprintf(buf, sizeof buf, "djb_error: Reading %s failed\n", path); *err = buf; return DJB_ERR_READ_FAILED;
	}
	void *map = mmap(nullptr, 12 + PAYLOAD, PROT_READ, MAP_PRIVATE, fd, 0);
	close(fd);
	if (map == MAP_FAILED) { snprintf(buf, sizeof buf, "djb_error: Reading %s failed\n", path); *err = buf; return DJB_ERR_READ_FAILED; }
	if (map_observer_fn obs = g_map_observer.load(std::memory_order_acquire)) obs(path, g_map_observer_user.load(std::memory_order_acquire));
	const char *base = (const char *)map + 12;                   // the payload is 4 bytes off 8-byte alignment: memcpy each double
	const size_t m = plan.slot.size();
	install_sigbus_guard();
	BusGuard guard;
	guard.lo = (const char *)map; guard.hi = guard.lo + 12 + PAYLOAD;
	if (sigsetjmp(guard.env, 1) != 0) {                         // a page of the mapping is gone: the file shrank under us
		t_bus_guard = nullptr;
		munmap(map, 12 + PAYLOAD);
		snprintf(buf, sizeof buf, "djb_error: Reading %s failed\n", path); *err = buf; return DJB_ERR_READ_FAILED;
	}
	t_bus_guard = &guard;
	for (size_t k = 0; k < m; ++k) {
		const long long i = plan.idx[k];
		double s[3];
		memcpy(&s[0], base + 8 * (size_t)i, 8);
		memcpy(&s[1], base + 8 * (size_t)(i + MERL_N), 8);
		memcpy(&s[2], base + 8 * (size_t)(i + 2 * MERL_N), 8);
		// merl_convert_one on this entry (same expressions; host IEEE arithmetic == the device's)
		float r = (float)(s[0] * (1.00 / 1500.0)), g = (float)(s[1] * (1.15 / 1500.0)), b = (float)(s[2] * (1.66 / 1500.0));
		if ((double)r < 0.0 || (double)g < 0.0 || (double)b < 0.0) r = g = b = 0.0f;
		float *o = out + 3 * (size_t)plan.slot[k];
		o[0] = r; o[1] = g; o[2] = b;
	}
	t_bus_guard = nullptr;
	if (keep) keep->emplace_back(map, 12 + PAYLOAD);
	else munmap(map, 12 + PAYLOAD);
	return DJB_OK;
}

} // namespace djbfile
