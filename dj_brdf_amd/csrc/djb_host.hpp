// djb_host.hpp -- internals shared by the host-side translation units of libdjb_hip.so:
//   djb_host.hip      errors, microfacet::params resolution, handle lifetime (contexts, objects, the host twin), options
//   djb_host_ops.hip  the operator surface: staging pipelines, the two-tier worklist, eval / sample / query / ... entry points,
//                     diagnostics and self-tests
//   djb_host_fit.hip  the fit drivers (tabular, tabular_anisotropic, batch fits)
//   djb_loader.hip    the file pipeline (djb_fit_merl_files)
// Not installed; the public surface is include/djb_hip.h.
#pragma once
#include "../../include/djb_hip.h"
#include "djb_internal.hpp"
#include "djb_cpu.hpp"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdint>
#include <cstdlib>
#include <condition_variable>
#include <thread>
#include <unistd.h>
#include <map>
#include <atomic>
#include <mutex>
#include <string>
#include <vector>

using djbdev::Brdf;
using djbdev::Params;
using djbdev::View;
using djbcpu::is_cpu;

struct djb_ctx {
	int device;
	hipStream_t stream;
	bool owns_stream;
	hipEvent_t ev0, ev1;
	void *scratch;            // worklist of the two-tier kernels (utia, contract mode; grown on demand)
	size_t scratch_bytes;
	int merl_exact_only;      // DJB_OPT_MERL_EXACT_ONLY
	int aniso_qf2_aligned = 0; // DJB_OPT_ANISO_QF2_ALIGNED
	int fit_files_dense = 0;   // DJB_OPT_FIT_FILES_DENSE
	// what djb_fit_merl_files keeps between calls (djb_loader.hip: slot plans, a pinned / device buffer pair, its worker threads);
	// created on first use under call_mu, released by djb_ctx_destroy through loader_state_free
	std::map<int, int> fit_seen;               // by resolution: single-material fits so far (the tables below are built from the second on)
	std::map<int, float *> fit_fresnel_dirs;   // by resolution: djbk::FitSplit::fres_dirs (device memory, freed by djb_ctx_destroy)
	void *loader_state = nullptr;
	void (*loader_state_free)(void *) = nullptr;
	int utia_exact_only = 0;   // DJB_OPT_UTIA_EXACT_ONLY: utia eval batches run k_eval<UTIA> (one kernel, exact fall-backs inline) instead of the two tiers
	// tier-2 worklist of the two-tier kernels: capacity as a share of the batch.  2 % covers the bench distribution 8x over;
	// after a call whose list overflowed (hostile distributions: 6 % of uniformly drawn BINS sit in the reference's snap
	// region) the share grows, so that only the first such call pays the full rescan (wl_note / wl_adapt)
	double wl_frac = 1.0 / 48;
	hipEvent_t wl_ev = nullptr;
	unsigned int *wl_host = nullptr;      // pinned: the count of the last large call
	size_t wl_last_cap = 0; long long wl_last_n = 0; bool wl_pending = false; int wl_words = 1;
	double wl_last_share = 0.0;            // tier-2 pairs / pairs of the last large two-tier call
	unsigned long long wl_note_key = 0, ct_key = 0; double ct_key_share = 0.0;   // contract mode: (lobe, params) of that call and its share
	unsigned int ct_hopeless_calls = 0;        // calls answered by the exact kernel because of ct_key_share: every 16th re-probes
	long long test_worklist_cap = -1;   // DJB_OPT_TEST_WORKLIST_CAP (tests): >= 0 overrides the tier-2 worklist capacity
	int contract_1e5 = 0;      // DJB_OPT_CONTRACT_1E5: dense GGX eval batches run the two-tier value-contract kernels
	std::atomic<long long> host_batch_max{DJB_SCALAR_HOST_MAX};   // DJB_OPT_HOST_BATCH_MAX: host-array calls up to this size are answered by the host twin
	int scalar_on_device = 0;  // DJB_OPT_SCALAR_ON_DEVICE: scalar-size host calls go through the GPU too (A/B testing)
	// HBM staging blocks of the DJB_MEM_HOST path, recycled across calls (hipMalloc costs more than
	// a small batch); bounded by POOL_MAX_BYTES
	std::mutex pool_mu;
	std::vector<std::pair<void *, size_t>> pool;
	// Every entry point that enqueues work holds this for the duration of the call: the reference's
	// operators are const and safe to call concurrently on one object (Mitsuba's render threads do),
	// so concurrent callers of one context are serialised here (its stream serialises them anyway)
	// and multi-launch sequences that share per-context scratch (the two-tier MERL lookup) stay atomic.
	std::recursive_mutex call_mu;
	// small DJB_MEM_HOST calls (scalar facade calls, <= SMALL_N units): inputs are memcpy'd into this pinned,
	// device-visible arena and the kernels read / write it directly over PCIe -- no hipMemcpy, one sync
	char *pin = nullptr;
	size_t pin_bytes = 0;
	int n_cus = 0;            // compute units of the device (how many fit workgroups run at once)
	// large DJB_MEM_HOST batches (eval_host_pipelined): results of chunk c leave on this second stream while
	// chunk c+1 comes in on `stream`, so both PCIe directions carry data; created on first use
	hipStream_t d2h_stream = nullptr;
	hipEvent_t pipe_ev[2] = { nullptr, nullptr };
	hipStream_t owned_stream = nullptr;   // the stream djb_ctx_create made, after djb_ctx_set_stream moved the ctx off it
};

struct djb_brdf {
	int device;                      // MUST stay the first member (djbcpu::is_cpu): device of the creating context, kept here
	                                 // because the handle may be destroyed after its context
	djb_ctx *ctx;
	Brdf dev;                        // device view (pointers into HBM)
	std::vector<void *> allocs;      // HBM blocks owned by this object
	// tabular: host copies for the accessors
	std::vector<float> p22, sigma, cdf, qf, fresnel;
	float alpha_beckmann, alpha_ggx;
	// tabular_anisotropic: host copies of the 8 tables (+ fresnel above) and the two 5-parameter fits
	std::vector<float> aniso[8];
	float aniso_fit[10];
	int elev = 0, azim = 0;
	int aniso_qf2_entries = 0;       // size of the reference's m_qf2 (== elev * azim unless rows came up short)
	// merl / utia created from a file or from memory: the file's double payload stays in HBM (one of
	// `allocs`) for get_samples(); 35 MB per MERL material, 2 MB per UTIA material
	const double *raw_samples = nullptr;
	long long raw_count = 0;
	std::vector<double> model_host;   // sgd / abc: the table row (host copy)
	// host twin (djb_cpu.cpp object with the same tables in host memory) that answers scalar-size DJB_MEM_HOST
	// calls on the caller's thread; built on first use, kept in step by set_shadow / set_fresnel
	mutable std::once_flag twin_once;
	mutable djb_brdf *twin = nullptr;
	// 1 once build_twin has run: std::call_once costs two thread-local stores and a pthread_once call even when the flag is set
	// (6 ns of a 57 ns one-pair call), a load does not
	mutable std::atomic<int> twin_built{0};
};

namespace djbh {

djb_status fail(djb_status st, const char *fmt, ...);      // sets the thread's djb_last_error() message (djb_host.hip)

#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) \
	return djbh::fail(DJB_ERR_HIP, "djb_error: HIP %s at %s:%d: %s", #expr, __FILE__, __LINE__, \
	            hipGetErrorString(e_)); } while (0)

constexpr long long MERL_N = 90LL * 90 * 180;
constexpr long long UTIA_N = 3LL * 288 * 288;
constexpr long long SCALAR_HOST_MAX = DJB_SCALAR_HOST_MAX;   // scalar-size DJB_MEM_HOST calls: answered by the host twin

// microfacet::params on the host (djb_host.hip)
djb_status resolve_params(const djb_params *in, djb_params_resolved *p);
djb_status device_params(const djb_params *in, Params *out, int brdf_kind = -1, bool want_reciprocals = true);

// ------------------------------------------------------------------ host <-> HBM staging
// DJB_MEM_HOST callers: every array is copied to / from HBM **in the caller's own layout** with
// plain hipMemcpy straight from / into the caller's memory -- an array of djb::vec3 (stride 3)
// becomes one 12n-byte copy and the kernels read it with stride 3; SoA (stride 1) arrays are
// copied per component (one copy when the three are contiguous).  No host-side packing: a
// single-threaded AoS<->SoA loop runs at ~3 GB/s, the copy itself at ~56 GB/s (tools/pcie_probe.hip).
// Only exotic strides fall back to a packed SoA block.  Device-resident callers bypass all of this.
constexpr size_t POOL_MAX_BYTES = 8ull << 30;
constexpr long long SMALL_N = 4096;            // units per call that go through the pinned arena
constexpr size_t PIN_BYTES = 1u << 20;         // >= SMALL_N * (largest per-unit footprint of any entry point)

struct Staged {
	djb_ctx *ctx; long long n; int mem;
	std::vector<std::pair<void *, size_t>> blocks;
	struct Out { View dev; djb_vec3_view host; int layout; };   // layout: 0 interleaved, 1 SoA stride 1, 2 packed fallback
	std::vector<Out> outs;
	std::vector<std::pair<void *, std::pair<void *, size_t>>> out_raw;   // dev -> (host, bytes)

	bool small = false, synced = false;
	size_t pin_off = 0;

	Staged(djb_ctx *c, long long n_, int mem_) : ctx(c), n(n_), mem(mem_)
	{
		// the arena is per context and the caller holds ctx->call_mu for the whole entry point
		small = mem == DJB_MEM_HOST && n <= SMALL_N && ctx && ctx->pin;
	}
	~Staged()
	{
		if (blocks.empty()) return;
		std::lock_guard<std::mutex> g(ctx->pool_mu);
		size_t total = 0;
		for (auto &p : ctx->pool) total += p.second;
		for (auto &b : blocks) {
			if (total + b.second <= POOL_MAX_BYTES && ctx->pool.size() < 32) { ctx->pool.push_back(b); total += b.second; }
			else (void)hipFree(b.first);
		}
	}

	// One pageable copy at a time: the runtime pins the caller's pages for the duration of an
	// asynchronous copy, and two in-flight copies whose host ranges share a page (x/y/z of one SoA
	// allocation, or two small heap arrays) fail with hipErrorInvalidValue.
	djb_status copy(void *dst, const void *src, size_t bytes, hipMemcpyKind kind)
	{
		if (small) {   // both ends are host-addressable: inputs before the launch, outputs after one sync
			if (kind == hipMemcpyDeviceToHost && !synced) { HIP_TRY(hipStreamSynchronize(ctx->stream)); synced = true; }
			memcpy(dst, src, bytes);
			return DJB_OK;
		}
		hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, ctx->stream);
		if (e != hipSuccess) {
			hipPointerAttribute_t ad, as;
			hipError_t e1 = hipPointerGetAttributes(&ad, dst), e2 = hipPointerGetAttributes(&as, src);
			(void)hipGetLastError();
			return fail(DJB_ERR_HIP, "djb_error: staging copy failed (%s): dst %p [attr %d type %d dev %d] src %p [attr %d type %d dev %d] "
			            "bytes %zu kind %d n %lld", hipGetErrorString(e), dst, (int)e1, e1 == hipSuccess ? (int)ad.type : -1,
			            e1 == hipSuccess ? ad.device : -1, src, (int)e2, e2 == hipSuccess ? (int)as.type : -1,
			            e2 == hipSuccess ? as.device : -1, bytes, (int)kind, n);
		}
		HIP_TRY(hipStreamSynchronize(ctx->stream));
		return DJB_OK;
	}
	static bool valid(const djb_vec3_view *v) { return v && v->x && v->y && v->z; }
	static int layout_of(const djb_vec3_view *v)
	{
		if (v->stride == 3 && v->y == v->x + 1 && v->z == v->x + 2) return 0;
		if (v->stride == 1) return 1;
		return 2;
	}

	djb_status alloc(size_t bytes, void **out)
	{
		if (bytes == 0) bytes = 4;
		if (small) {
			size_t off = (pin_off + 255) & ~(size_t)255;
			if (off + bytes <= ctx->pin_bytes) { *out = ctx->pin + off; pin_off = off + bytes; return DJB_OK; }
			if (pin_off == 0) small = false;      // nothing handed out yet: fall back to the HBM path
			else return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: pinned staging arena exhausted");
		}
		{
			std::lock_guard<std::mutex> g(ctx->pool_mu);
			int best = -1;
			for (int k = 0; k < (int)ctx->pool.size(); ++k)
				if (ctx->pool[k].second >= bytes && (best < 0 || ctx->pool[k].second < ctx->pool[best].second)) best = k;
			if (best >= 0 && ctx->pool[best].second <= 2 * bytes + (1u << 20)) {
				blocks.push_back(ctx->pool[best]);
				*out = ctx->pool[best].first;
				ctx->pool.erase(ctx->pool.begin() + best);
				return DJB_OK;
			}
		}
		void *d = nullptr;
		hipError_t e = hipMalloc(&d, bytes);
		if (e != hipSuccess) {   // give the recycled blocks back and retry once
			(void)hipGetLastError();
			std::lock_guard<std::mutex> g(ctx->pool_mu);
			for (auto &p : ctx->pool) (void)hipFree(p.first);
			ctx->pool.clear();
			e = hipMalloc(&d, bytes);
		}
		HIP_TRY(e);
		blocks.push_back({ d, bytes });
		*out = d;
		return DJB_OK;
	}

	djb_status in_vec(const djb_vec3_view *v, View *out)
	{
		if (!valid(v)) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null vec3 view");
		if (mem == DJB_MEM_DEVICE) { *out = View{ v->x, v->y, v->z, (long long)v->stride }; return DJB_OK; }
		float *d = nullptr;
		djb_status st = alloc(sizeof(float) * 3 * (size_t)n, (void **)&d);
		if (st != DJB_OK) return st;
		const size_t nb = sizeof(float) * (size_t)n;
		switch (layout_of(v)) {
		case 0:
			if (n) { djb_status cs_ = copy(d, v->x, 3 * nb, hipMemcpyHostToDevice); if (cs_ != DJB_OK) return cs_; }
			*out = View{ d, d + 1, d + 2, 3 };
			break;
		case 1:
			if (n && v->y == v->x + n && v->z == v->x + 2 * n) { djb_status cs_ = copy(d, v->x, 3 * nb, hipMemcpyHostToDevice); if (cs_ != DJB_OK) return cs_; }
			else if (n) {
				{ djb_status cs_ = copy(d, v->x, nb, hipMemcpyHostToDevice); if (cs_ != DJB_OK) return cs_; }
				{ djb_status cs_ = copy(d + n, v->y, nb, hipMemcpyHostToDevice); if (cs_ != DJB_OK) return cs_; }
				{ djb_status cs_ = copy(d + 2 * n, v->z, nb, hipMemcpyHostToDevice); if (cs_ != DJB_OK) return cs_; }
			}
			*out = View{ d, d + n, d + 2 * n, 1 };
			break;
		default: {
			std::vector<float> pack(3 * (size_t)n);
			for (long long k = 0; k < n; ++k) {
				pack[k] = v->x[k * v->stride];
				pack[n + k] = v->y[k * v->stride];
				pack[2 * n + k] = v->z[k * v->stride];
			}
			if (n) { djb_status cs_ = copy(d, pack.data(), 3 * nb, hipMemcpyHostToDevice); if (cs_ != DJB_OK) return cs_; }
			HIP_TRY(hipStreamSynchronize(ctx->stream));   // pack goes out of scope
			*out = View{ d, d + n, d + 2 * n, 1 };
		}
		}
		return DJB_OK;
	}
	djb_status in_f(const float *h, const float **out)
	{
		if (!h) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null input array");
		if (mem == DJB_MEM_DEVICE) { *out = h; return DJB_OK; }
		float *d = nullptr;
		djb_status st = alloc(sizeof(float) * (size_t)n, (void **)&d);
		if (st != DJB_OK) return st;
		if (n) { djb_status cs_ = copy(d, h, sizeof(float) * (size_t)n, hipMemcpyHostToDevice); if (cs_ != DJB_OK) return cs_; }
		*out = d;
		return DJB_OK;
	}
	djb_status out_vec(const djb_vec3_view *v, View *out)
	{
		if (!valid(v)) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null output vec3 view");
		if (mem == DJB_MEM_DEVICE) { *out = View{ v->x, v->y, v->z, (long long)v->stride }; return DJB_OK; }
		float *d = nullptr;
		djb_status st = alloc(sizeof(float) * 3 * (size_t)n, (void **)&d);
		if (st != DJB_OK) return st;
		int lay = layout_of(v);
		*out = lay == 0 ? View{ d, d + 1, d + 2, 3 } : View{ d, d + n, d + 2 * n, 1 };
		outs.push_back(Out{ *out, *v, lay });
		return DJB_OK;
	}
	template <typename T> djb_status out_arr(T *h, T **out)
	{
		if (!h) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null output array");
		if (mem == DJB_MEM_DEVICE) { *out = h; return DJB_OK; }
		T *d = nullptr;
		djb_status st = alloc(sizeof(T) * (size_t)n, (void **)&d);
		if (st != DJB_OK) return st;
		out_raw.push_back({ d, { h, sizeof(T) * (size_t)n } });
		*out = d;
		return DJB_OK;
	}
	djb_status finish()
	{
		if (mem == DJB_MEM_DEVICE) return DJB_OK;
		const size_t nb = sizeof(float) * (size_t)n;
		for (auto &o : outs) {
			if (!n) continue;
			if (o.layout == 0) { djb_status cs_ = copy(o.host.x, o.dev.x, 3 * nb, hipMemcpyDeviceToHost); if (cs_ != DJB_OK) return cs_; }
			else if (o.layout == 1) {
				if (o.host.y == o.host.x + n && o.host.z == o.host.x + 2 * n)
					{ djb_status cs_ = copy(o.host.x, o.dev.x, 3 * nb, hipMemcpyDeviceToHost); if (cs_ != DJB_OK) return cs_; }
				else {
					{ djb_status cs_ = copy(o.host.x, o.dev.x, nb, hipMemcpyDeviceToHost); if (cs_ != DJB_OK) return cs_; }
					{ djb_status cs_ = copy(o.host.y, o.dev.y, nb, hipMemcpyDeviceToHost); if (cs_ != DJB_OK) return cs_; }
					{ djb_status cs_ = copy(o.host.z, o.dev.z, nb, hipMemcpyDeviceToHost); if (cs_ != DJB_OK) return cs_; }
				}
			} else {
				std::vector<float> pack(3 * (size_t)n);
				{ djb_status cs_ = copy(pack.data(), o.dev.x, 3 * nb, hipMemcpyDeviceToHost); if (cs_ != DJB_OK) return cs_; }
				HIP_TRY(hipStreamSynchronize(ctx->stream));
				for (long long k = 0; k < n; ++k) {
					o.host.x[k * o.host.stride] = pack[k];
					o.host.y[k * o.host.stride] = pack[n + k];
					o.host.z[k * o.host.stride] = pack[2 * n + k];
				}
			}
		}
		for (auto &o : out_raw)
			if (o.second.second) { djb_status cs_ = copy(o.second.first, o.first, o.second.second, hipMemcpyDeviceToHost); if (cs_ != DJB_OK) return cs_; }
		if (!(small && synced)) HIP_TRY(hipStreamSynchronize(ctx->stream));
		return DJB_OK;
	}
};

djb_status check_call(djb_ctx *ctx, const djb_brdf *b, long long n, int mem);
// scalar-size DJB_MEM_HOST calls: the host twin of a GPU object, or NULL when the call belongs on the GPU (djb_host.hip)
const djb_brdf *scalar_twin(const djb_ctx *ctx, const djb_brdf *b, long long n, int mem);
djb_status cpu_pair_check(const djb_ctx *ctx, const djb_brdf *b);
// object construction helpers (djb_host.hip)
djb_status alloc_brdf(djb_ctx *ctx, int kind, djb_brdf **out);
djb_status upload_floats(djb_brdf *b, const float *host, size_t count, const float **dev_out);
djb_status set_fresnel(djb_brdf *b, const djb_fresnel_desc *f);
djb_status create_microfacet(djb_ctx *ctx, int kind, const djb_fresnel_desc *f, int shadow, djb_brdf **out);
} // namespace djbh
namespace djbk {
// per-slot rgb samples in HBM as a fit source (djbdev::Brdf::merl_sparse; djb_host.hip): never handed to the eval kernels
djb_status wrap_merl_slots(djb_ctx *ctx, djbdev::MerlTexel *slots, djb_brdf **out);
}
namespace djbh {

} // namespace djbh

// No C++ exception may cross the C ABI (a ctypes / C caller would abort): every entry point is a
// function-try-block that maps std::bad_alloc and anything else to a status + message.
#define DJB_ABI_CATCH \
	catch (const std::bad_alloc &) { return djbh::fail(DJB_ERR_OUT_OF_MEMORY, "djb_error: out of host memory"); } \
	catch (const std::exception &ex_) { return djbh::fail(DJB_ERR_INTERNAL, "djb_error: internal error: %s", ex_.what()); } \
	catch (...) { return djbh::fail(DJB_ERR_INTERNAL, "djb_error: internal error"); }

