// djb_host.hip -- the C ABI of libdjb_hip.so (include/djb_hip.h): handle lifetime, argument
// checking, host<->HBM staging, and kernel launches.  On a GPU context every batch runs on the gfx950
// kernels (or fails with DJB_ERR_NO_DEVICE / DJB_ERR_HIP: there is no silent fallback); the two uses of
// the product's host instantiation of the same per-unit code (djb_cpu.cpp) are explicit: a CPU context
// (djb_ctx_create(DJB_DEVICE_CPU)) and scalar-size DJB_MEM_HOST calls (<= DJB_SCALAR_HOST_MAX units).
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
#include <mutex>
#include <string>
#include <vector>

using djbdev::Brdf;
using djbdev::Params;
using djbdev::View;
using djbcpu::is_cpu;

namespace {

thread_local std::string g_err;

djb_status fail(djb_status st, const char *fmt, ...)
{
	char buf[256];   // same 256-byte budget as djb::exc (dj_brdf.h:578-587)
	va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
	g_err = buf;
	return st;
}

#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) \
	return fail(DJB_ERR_HIP, "djb_error: HIP %s at %s:%d: %s", #expr, __FILE__, __LINE__, \
	            hipGetErrorString(e_)); } while (0)

constexpr long long MERL_N = 90LL * 90 * 180;
constexpr long long UTIA_N = 3LL * 288 * 288;

} // namespace

struct djb_ctx {
	int device;
	hipStream_t stream;
	bool owns_stream;
	hipEvent_t ev0, ev1;
	void *scratch;            // worklist of the two-tier MERL kernel (grown on demand)
	size_t scratch_bytes;
	int merl_exact_only;      // DJB_OPT_MERL_EXACT_ONLY
	int aniso_qf2_aligned = 0; // DJB_OPT_ANISO_QF2_ALIGNED
	int fit_files_dense = 0;   // DJB_OPT_FIT_FILES_DENSE
	int utia_exact_only = 0;   // DJB_OPT_UTIA_EXACT_ONLY: utia eval batches run k_eval<UTIA> (one kernel, exact fall-backs inline) instead of the two tiers
	// tier-2 worklist of the two-tier kernels: capacity as a share of the batch.  2 % covers the bench distribution 8x over;
	// after a call whose list overflowed (hostile distributions: 6 % of uniformly drawn BINS sit in the reference's snap
	// region) the share grows, so that only the first such call pays the full rescan (wl_note / wl_adapt)
	double wl_frac = 1.0 / 48;
	hipEvent_t wl_ev = nullptr;
	unsigned int *wl_host = nullptr;      // pinned: the count of the last large call
	size_t wl_last_cap = 0; long long wl_last_n = 0; bool wl_pending = false;
	int contract_1e5 = 0;      // DJB_OPT_CONTRACT_1E5: dense GGX eval batches run the two-tier value-contract kernels
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
};

namespace {

// ------------------------------------------------------------------ microfacet::params on the host
// (scalar set-up code, dj_brdf.h:1355-1506; same float/double evaluation order as the reference)
float Ff(double x) { return (float)x; }

void resolve_location(djb_params_resolved *p, float tx, float ty)
{
	p->tx_n = tx; p->ty_n = ty;
	float x = -tx, y = -ty, z = 1.0f;
	float m = x * x + y * y + z * z;
	float r = Ff(1.0 / std::sqrt((double)m));
	p->n[0] = r * x; p->n[1] = r * y; p->n[2] = r * z;
}

void resolve_ellipse(djb_params_resolved *p, float a1, float a2, float phi_a)
{
	p->a1 = a1; p->a2 = a2; p->phi_a = phi_a;
	float c = Ff(std::cos((double)phi_a)), s = Ff(std::sin((double)phi_a));
	float c2 = Ff(2.0 * (double)c * (double)c - (double)1.0f);
	float a1s = a1 * a1, a2s = a2 * a2, t1 = a1s + a2s, t2 = a1s - a2s;
	p->ax = Ff(std::sqrt(0.5 * (double)(t1 + t2 * c2)));
	p->ay = Ff(std::sqrt(0.5 * (double)(t1 - t2 * c2)));
	p->rho = (a2s - a1s) * c * s / (p->ax * p->ay);
	p->sqrt_one_minus_rho_sqr = Ff(std::sqrt(1.0 - (double)(p->rho * p->rho)));
}

djb_status resolve_params(const djb_params *in, djb_params_resolved *p)
{
	memset(p, 0, sizeof *p);
	int kind = in ? in->kind : DJB_PARAMS_STANDARD;
	if (kind == DJB_PARAMS_STANDARD) {
		resolve_ellipse(p, 1.0f, 1.0f, 0.0f);
		resolve_location(p, 0.0f, 0.0f);
	} else if (kind == DJB_PARAMS_ELLIPTIC) {
		if (!(in->v[0] > 0.0f && in->v[1] > 0.0f))
			return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: Invalid ellipse radii");   // dj_brdf.h:1453
		resolve_ellipse(p, in->v[0], in->v[1], in->v[2]);
		resolve_location(p, 0.0f, 0.0f);
	} else if (kind == DJB_PARAMS_PDFPARAMS) {
		float ax = in->v[0], ay = in->v[1], rho = in->v[2];
		if (!(ax > 0.0f && ay > 0.0f))
			return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: Invalid scale parameters");  // :1466
		if (!(std::fabs((double)rho) < 1.0))
			return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: Invalid correlation parameter"); // :1467
		p->ax = ax; p->ay = ay; p->rho = rho;
		p->sqrt_one_minus_rho_sqr = Ff(std::sqrt(1.0 - (double)(rho * rho)));
		float axs = ax * ax, ays = ay * ay;
		float cov = Ff((double)(rho * ax * ay) * 2.0);
		float t1 = axs + ays, t2 = axs - ays;
		float t3 = Ff(std::sqrt((double)(t2 * t2 + cov * cov)));
		p->a1 = Ff(std::sqrt(0.5 * (double)(t1 + t3)));
		p->a2 = Ff(std::sqrt(0.5 * (double)(t1 - t3)));
		p->phi_a = ((double)cov != 0.0) ? Ff(std::atan((double)((axs - ays - t3) / cov))) : 0.0f;
		resolve_location(p, in->v[3], in->v[4]);
	} else {
		return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: unknown params kind %d", kind);
	}
	return DJB_OK;
}

djb_status device_params(const djb_params *in, Params *out, int brdf_kind = -1)
{
	// lambert::params(reflectance) (dj_brdf.h:114-119, 861-868): carried to the kernel in the n slot
	if (brdf_kind == DJB_KIND_LAMBERT) {
		if (in && in->kind != DJB_PARAMS_STANDARD && in->kind != DJB_PARAMS_LAMBERT)
			return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: a lambert brdf takes lambert::params");
		memset(out, 0, sizeof *out);
		const bool has = in && in->kind == DJB_PARAMS_LAMBERT;
		out->nx = has ? in->v[0] : 1.0f; out->ny = has ? in->v[1] : 1.0f; out->nz = has ? in->v[2] : 1.0f;
		return DJB_OK;
	}
	if (in && in->kind == DJB_PARAMS_LAMBERT)
		return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: lambert::params passed to a brdf that is not a lambert");
	djb_params_resolved r;
	djb_status st = resolve_params(in, &r);
	if (st != DJB_OK) return st;
	out->nx = r.n[0]; out->ny = r.n[1]; out->nz = r.n[2];
	out->ax = r.ax; out->ay = r.ay; out->rho = r.rho; out->s = r.sqrt_one_minus_rho_sqr;
	out->tx = r.tx_n; out->ty = r.ty_n;
	// reciprocals of the two launch-uniform denominators of mf_p22 (djb_device.hpp: fdiv_r): correctly rounded doubles
	// of exactly the floats the kernel divides by (this TU is built with -ffp-contract=off: no FMA in ax * ay * s)
	const float t2 = out->ax * out->ay * out->s;
	out->r_ax = 1.0 / (double)out->ax;
	out->r_t2 = 1.0 / (double)t2;
	if (!(std::fabs(out->r_ax) <= 1e300)) out->r_ax = 0.0;      // ax == 0 / NaN: leave it to the IEEE division
	if (!(std::fabs(out->r_t2) <= 1e300)) out->r_t2 = 0.0;
	return DJB_OK;
}

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

djb_status check_call(djb_ctx *ctx, const djb_brdf *b, long long n, int mem)
{
	if (!ctx) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null ctx");
	if (b && b->device != ctx->device)
		return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: brdf lives on device %d, ctx on %d", b->device, ctx->device);
	if (n < 0) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: negative batch size");
	if (mem != DJB_MEM_DEVICE && mem != DJB_MEM_HOST)
		return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: unknown memory space %d", mem);
	HIP_TRY(hipSetDevice(ctx->device));
	return DJB_OK;
}


// ------------------------------------------------------------------ scalar-size host calls: the host twin
// Calls of <= DJB_SCALAR_HOST_MAX units with DJB_MEM_HOST arrays -- the one-pair virtuals of the djb:: facade, a
// renderer's per-hit eval / sample / pdf -- are evaluated on the CALLING thread by the product's host instantiation
// of the same per-unit code (djb_cpu.cpp), from a host copy of the object's tables: no staging, no launch, no
// context mutex (the reference's operators are const and concurrent; a 15 us GPU round trip per pair behind a
// mutex is not a drop-in for them).  Everything larger runs on the GPU.  DJB_OPT_SCALAR_ON_DEVICE = 1 sends these
// calls through the GPU as well (tests compare the two bit for bit).
constexpr long long SCALAR_HOST_MAX = DJB_SCALAR_HOST_MAX;

djb_fresnel_desc current_fresnel_desc(const djb_brdf *b)
{
	djb_fresnel_desc d;
	memset(&d, 0, sizeof d);
	d.kind = b->dev.fr.kind;
	for (int c = 0; c < 3; ++c) { d.a[c] = b->dev.fr.a[c]; d.b[c] = b->dev.fr.b[c]; }
	if (d.kind == DJB_FRESNEL_SPLINE) { d.points = b->fresnel.data(); d.npoints = b->dev.fr.npts; }
	return d;
}

void build_twin(const djb_brdf *b)
{
	djb_ctx *tc = djbcpu::twin_ctx();
	djb_brdf *t = nullptr;
	djb_status st = DJB_ERR_NOT_IMPLEMENTED;
	const djb_fresnel_desc fd = current_fresnel_desc(b);
	auto download = [&](std::vector<char> &host, const void *dev, size_t bytes) -> bool {
		host.resize(bytes);
		std::lock_guard<std::recursive_mutex> call_lock(b->ctx->call_mu);
		if (hipSetDevice(b->ctx->device) != hipSuccess) return false;
		if (hipMemcpyAsync(host.data(), dev, bytes, hipMemcpyDeviceToHost, b->ctx->stream) != hipSuccess) { (void)hipGetLastError(); return false; }
		return hipStreamSynchronize(b->ctx->stream) == hipSuccess;
	};
	switch (b->dev.kind) {
	case DJB_KIND_BECKMANN: case DJB_KIND_GGX:
		st = djbcpu::create_microfacet(tc, b->dev.kind, &fd, b->dev.shadow, &t); break;
	case DJB_KIND_LAMBERT: st = djbcpu::create_lambert(tc, &t); break;
	case DJB_KIND_SGD: case DJB_KIND_ABC:
		if (!b->model_host.empty()) st = djbcpu::create_model(tc, b->dev.kind, b->model_host.data(), (int)b->model_host.size(), &t);
		break;
	case DJB_KIND_TABULAR: {
		const int res = b->dev.n_p22;
		std::vector<float> fz(3 * (size_t)res, 1.0f);
		const float *fp = b->fresnel.size() == 3 * (size_t)res ? b->fresnel.data() : fz.data();
		st = djbcpu::create_tabular_from_tables(tc, b->dev.shadow, res, b->p22.data(), b->sigma.data(), b->cdf.data(), b->qf.data(),
		                                        (int)b->qf.size(), fp, b->alpha_beckmann, b->alpha_ggx, &t);
		if (st == DJB_OK) st = djbcpu::set_fresnel(t, &fd);
		break;
	}
	case DJB_KIND_TABULAR_ANISO: {
		const float *tabs[8]; int counts[8];
		for (int k = 0; k < 8; ++k) { tabs[k] = b->aniso[k].data(); counts[k] = (int)b->aniso[k].size(); }
		std::vector<float> fz(3 * (size_t)b->elev, 1.0f);
		const float *fp = b->fresnel.size() == 3 * (size_t)b->elev ? b->fresnel.data() : fz.data();
		st = djbcpu::create_aniso_from_tables(tc, b->dev.shadow, b->elev, b->azim, tabs, counts, fp, b->aniso_fit, b->aniso_qf2_entries, &t);
		if (st == DJB_OK) st = djbcpu::set_fresnel(t, &fd);
		break;
	}
	case DJB_KIND_MERL: {
		if (b->dev.merl_sparse) break;            // per-slot texels of the file pipeline: internal, never evaluated
		std::vector<char> host;
		if (download(host, b->dev.merl, sizeof(djbdev::MerlTexel) * (size_t)MERL_N))
			st = djbcpu::create_merl_from_texels(tc, (const float *)host.data(), &t);
		break;
	}
	case DJB_KIND_UTIA: {
		std::vector<char> host;
		if (download(host, b->dev.utia, sizeof(float4) * 8 * (size_t)(UTIA_N / 3)))
			st = djbcpu::create_utia_from_records(tc, (const float *)host.data(), &t);
		break;
	}
	}
	b->twin = st == DJB_OK ? t : nullptr;
}

// the host twin of a GPU object for a scalar-size host call, or NULL (then the call takes the GPU path)
const djb_brdf *scalar_twin(const djb_ctx *ctx, const djb_brdf *b, long long n, int mem)
{
	if (mem != DJB_MEM_HOST || n > SCALAR_HOST_MAX || n < 0 || !b || ctx->scalar_on_device) return nullptr;
	std::call_once(b->twin_once, build_twin, b);
	return b->twin;
}

// CPU context: both operands must belong to it
djb_status cpu_pair_check(const djb_ctx *ctx, const djb_brdf *b)
{
	if (b && is_cpu(ctx) != is_cpu(b))
		return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: brdf and ctx belong to different back ends (CPU / GPU)");
	return DJB_OK;
}

djb_status eval_common(djb_ctx *ctx, const djb_brdf *b, int64_t n, const djb_vec3_view *i,
                       const djb_vec3_view *o, const djb_params *params, const djb_vec3_view *out_fr,
                       float *out_pdf, int mem, int want);

// ------------------------------------------------------------------ large host batches: both PCIe directions in flight
// A DJB_MEM_HOST batch of >= 2 chunks is cut into chunks of DJB_HOST_PIPE_CHUNK units (default: n/8 clamped
// to [2^19, 2^23]; 0 disables).  The calling thread copies chunk c+1 in and enqueues its kernels on the context's stream
// while a helper thread copies the results of chunk c out on a second stream (two HBM slots).  Each
// thread keeps the one-pageable-copy-at-a-time rule of Staged::copy, and the path is only taken when
// no input array shares a host page with an output array (see Staged::copy on why).  Results are those
// of the unchunked call: every unit is independent and the chunk kernels are the same kernels.
long long host_pipe_chunk(long long n)
{
	if (const char *e = getenv("DJB_HOST_PIPE_CHUNK")) { long long c = atoll(e); return c < 0 ? 0 : c; }
	// default: eight chunks for mid-sized batches (the first copy in and the last copy out are not overlapped:
	// time ~ input time x (1 + 1/(2 chunks)); >= 2^19 units each keeps a chunk's copies well above the per-copy
	// overhead), 2^23 units for large ones (tools/host_path_rate.py: 2^22..2^24 are within 3 %)
	long long c = ((n + 7) / 8 + 4095) & ~4095LL;
	if (c < (1LL << 19)) c = 1LL << 19;
	if (c > (1LL << 23)) c = 1LL << 23;
	return c;
}

struct HostSpan { uintptr_t lo, hi; };
HostSpan span_of(const djb_vec3_view *v, long long n)
{
	const float *a = v->x < v->y ? v->x : v->y; a = a < v->z ? a : v->z;
	const float *z = v->x > v->y ? v->x : v->y; z = z > v->z ? z : v->z;
	return HostSpan{ (uintptr_t)a, (uintptr_t)(z + (n - 1) * v->stride + 1) };
}
bool share_page(HostSpan a, HostSpan b)
{
	// base pages: separately allocated large arrays are usually adjacent mappings, so anything coarser than
	// the real page size would see every pair of arrays as sharing one
	static const uintptr_t PG = (uintptr_t)sysconf(_SC_PAGESIZE);
	return (a.lo & ~(PG - 1)) < ((b.hi + PG - 1) & ~(PG - 1)) && (b.lo & ~(PG - 1)) < ((a.hi + PG - 1) & ~(PG - 1));
}

// one per-unit array of a chunked host batch: a vec3 view (in the caller's layout) or a float array
struct PipeArr {
	const djb_vec3_view *v = nullptr; int layout = 0;   // vec3
	float *f = nullptr; int width = 1;                    // `width` contiguous floats per unit (1 = scalar, 5 = params record)
	float *dev[2] = { nullptr, nullptr };                 // the two HBM slots
	long long C = 0;
	static PipeArr vec(const djb_vec3_view *v) { PipeArr a; a.v = v; a.layout = Staged::layout_of(v); return a; }
	static PipeArr arr(const float *f, int width = 1) { PipeArr a; a.f = const_cast<float *>(f); a.width = width; return a; }
	HostSpan span(long long n) const { return v ? span_of(v, n) : HostSpan{ (uintptr_t)f, (uintptr_t)(f + (size_t)width * n) }; }
	size_t floats_per_unit() const { return v ? 3 : (size_t)width; }
	djb_vec3_view view(int s) const   // device view of slot s (vec3 arrays)
	{
		float *d = dev[s];
		return layout == 0 ? djb_vec3_view{ d, d + 1, d + 2, 3 } : djb_vec3_view{ d, d + C, d + 2 * C, 1 };
	}
	// units [lo, lo + m) between the caller's memory and slot s; one pageable copy at a time (Staged::copy)
	hipError_t move(int s, long long lo, long long m, bool to_dev, hipStream_t st) const
	{
		const hipMemcpyKind k = to_dev ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost;
		float *hp[3], *dp[3]; size_t cnt; int parts;
		if (!v) { hp[0] = f + (size_t)width * lo; dp[0] = dev[s]; cnt = (size_t)width * m; parts = 1; }
		else if (layout == 0) { hp[0] = v->x + 3 * lo; dp[0] = dev[s]; cnt = 3 * (size_t)m; parts = 1; }
		else { hp[0] = v->x + lo; hp[1] = v->y + lo; hp[2] = v->z + lo; dp[0] = dev[s]; dp[1] = dev[s] + C; dp[2] = dev[s] + 2 * C; cnt = (size_t)m; parts = 3; }
		for (int c = 0; c < parts; ++c) {
			hipError_t e = to_dev ? hipMemcpyAsync(dp[c], hp[c], sizeof(float) * cnt, k, st) : hipMemcpyAsync(hp[c], dp[c], sizeof(float) * cnt, k, st);
			if (e != hipSuccess) return e;
			if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;
		}
		return hipSuccess;
	}
};

// Runs launch(m, slot) -- which enqueues the kernels of one chunk on ctx->stream, reading ins[*].dev[slot] and
// writing outs[*].dev[slot] -- over all chunks.  Returns DJB_OK with *taken = false when the batch does not
// qualify (the caller then uses the plain copy-in / run / copy-out path).
template <class Launch>
djb_status host_pipeline(djb_ctx *ctx, long long n, std::vector<PipeArr> &ins, std::vector<PipeArr> &outs, Launch launch, bool *taken)
{
	*taken = false;
	const long long C = host_pipe_chunk(n);
	if (C <= 0 || n < 2 * C || n <= SMALL_N) return DJB_OK;
	// tests set DJB_HOST_PIPE_REQUIRE to turn "fell back to the plain path" into an error
	auto skip = [](const char *why) -> djb_status {
		if (getenv("DJB_HOST_PIPE_REQUIRE")) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: chunked host path not taken: %s", why);
		return DJB_OK;
	};
	for (auto *set : { &ins, &outs })
		for (const PipeArr &a : *set) {
			if (a.v ? !Staged::valid(a.v) : !a.f) return DJB_OK;          // the plain path reports the error
			if (a.v && a.layout == 2) return skip("exotic stride (packed on the host)");
		}
	for (const PipeArr &a : ins)
		for (const PipeArr &o : outs)
			if (share_page(a.span(n), o.span(n))) return skip("an input shares a host page with an output");
	if (!ctx->d2h_stream) {
		HIP_TRY(hipStreamCreateWithFlags(&ctx->d2h_stream, hipStreamNonBlocking));
		for (hipEvent_t &e : ctx->pipe_ev) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
	}
	*taken = true;

	// two slots per array in HBM, recycled through the context's pool
	Staged pool(ctx, C, DJB_MEM_HOST);
	pool.small = false;
	for (auto *set : { &ins, &outs })
		for (PipeArr &a : *set) {
			a.C = C;
			for (int s = 0; s < 2; ++s) {
				djb_status st = pool.alloc(sizeof(float) * a.floats_per_unit() * (size_t)C, (void **)&a.dev[s]);
				if (st != DJB_OK) return st;
			}
		}

	const long long nch = (n + C - 1) / C;
	std::mutex mu;
	std::condition_variable cv;
	long long issued = 0, drained = 0;
	bool abort_ = false;
	hipError_t werr = hipSuccess;
	std::thread drain([&]() {
		hipError_t e = hipSetDevice(ctx->device);
		for (long long c = 0; c < nch; ++c) {
			{
				std::unique_lock<std::mutex> lk(mu);
				cv.wait(lk, [&] { return issued > c || abort_; });
				if (abort_) return;
			}
			const int s = (int)(c & 1);
			const long long lo = c * C, m = n - lo < C ? n - lo : C;
			if (e == hipSuccess) e = hipStreamWaitEvent(ctx->d2h_stream, ctx->pipe_ev[s], 0);
			for (const PipeArr &a : outs)
				if (e == hipSuccess) e = a.move(s, lo, m, false, ctx->d2h_stream);
			{
				std::lock_guard<std::mutex> lk(mu);
				if (e != hipSuccess) werr = e;
				drained = c + 1;
			}
			cv.notify_all();
		}
	});
	auto stop = [&](djb_status st) {
		{ std::lock_guard<std::mutex> lk(mu); abort_ = true; }
		cv.notify_all();
		drain.join();
		(void)hipStreamSynchronize(ctx->stream);
		(void)hipStreamSynchronize(ctx->d2h_stream);
		return st;
	};
	for (long long c = 0; c < nch; ++c) {
		const int s = (int)(c & 1);
		const long long lo = c * C, m = n - lo < C ? n - lo : C;
		hipError_t late;
		{   // slot s is free once chunk c-2 has left
			std::unique_lock<std::mutex> lk(mu);
			cv.wait(lk, [&] { return drained >= c - 1; });
			late = werr;
		}
		if (late != hipSuccess)   // the helper keeps waiting for the remaining chunks: release it before joining
			return stop(fail(DJB_ERR_HIP, "djb_error: staging copy failed (%s) while returning a host batch", hipGetErrorString(late)));
		hipError_t e = hipSuccess;
		for (const PipeArr &a : ins)
			if (e == hipSuccess) e = a.move(s, lo, m, true, ctx->stream);
		if (e != hipSuccess) return stop(fail(DJB_ERR_HIP, "djb_error: staging copy failed (%s) in chunk %lld of a host batch", hipGetErrorString(e), c));
		djb_status st = launch(m, s);
		if (st != DJB_OK) return stop(st);
		if ((e = hipEventRecord(ctx->pipe_ev[s], ctx->stream)) != hipSuccess)
			return stop(fail(DJB_ERR_HIP, "djb_error: hipEventRecord: %s", hipGetErrorString(e)));
		{ std::lock_guard<std::mutex> lk(mu); issued = c + 1; }
		cv.notify_all();
	}
	drain.join();
	if (werr != hipSuccess) {
		(void)hipStreamSynchronize(ctx->stream);
		return fail(DJB_ERR_HIP, "djb_error: staging copy failed (%s) while returning a host batch", hipGetErrorString(werr));
	}
	HIP_TRY(hipStreamSynchronize(ctx->stream));
	return DJB_OK;
}

djb_status eval_host_pipelined(djb_ctx *ctx, const djb_brdf *b, long long n, const djb_vec3_view *i,
                               const djb_vec3_view *o, const djb_params *params, const djb_vec3_view *out_fr,
                               float *out_pdf, int want, bool *taken)
{
	*taken = false;
	const bool wfr = (want & 3) != 0, wpdf = (want & 4) != 0;
	if (!i || !o || (wfr && !out_fr)) return DJB_OK;                      // the plain path reports the error
	std::vector<PipeArr> ins{ PipeArr::vec(i), PipeArr::vec(o) }, outs;
	if (wfr) outs.push_back(PipeArr::vec(out_fr));
	if (wpdf) outs.push_back(PipeArr::arr(out_pdf));
	return host_pipeline(ctx, n, ins, outs, [&](long long m, int s) {
		djb_vec3_view vi = ins[0].view(s), vo = ins[1].view(s), vf = wfr ? outs[0].view(s) : djb_vec3_view{ nullptr, nullptr, nullptr, 0 };
		return eval_common(ctx, b, m, &vi, &vo, params, wfr ? &vf : nullptr, wpdf ? outs.back().dev[s] : nullptr, DJB_MEM_DEVICE, want);
	}, taken);
}

// worklist capacity bookkeeping (ctx->call_mu held).  wl_adapt: if the previous large call has finished and its list
// overflowed, grow the share (never blocks: an unfinished call is looked at next time).  wl_note: remember this call.
void wl_adapt(djb_ctx *ctx)
{
	if (!ctx->wl_pending || hipEventQuery(ctx->wl_ev) != hipSuccess) return;
	ctx->wl_pending = false;
	const unsigned int count = *ctx->wl_host;
	if ((size_t)count > ctx->wl_last_cap && ctx->wl_last_n > 0) {
		const double need = 1.25 * (double)count / (double)ctx->wl_last_n;
		ctx->wl_frac = std::min(0.25, std::max(need, 2.0 * ctx->wl_frac));
	}
}
void wl_note(djb_ctx *ctx, const unsigned int *count, size_t cap, long long n)
{
	if (n < (1LL << 20) || ctx->wl_pending) return;
	if (!ctx->wl_ev && hipEventCreateWithFlags(&ctx->wl_ev, hipEventDisableTiming) != hipSuccess) { ctx->wl_ev = nullptr; return; }
	if (!ctx->wl_host && hipHostMalloc((void **)&ctx->wl_host, 16) != hipSuccess) { ctx->wl_host = nullptr; return; }
	if (hipMemcpyAsync(ctx->wl_host, count, sizeof(unsigned int), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) return;
	if (hipEventRecord(ctx->wl_ev, ctx->stream) != hipSuccess) return;
	ctx->wl_last_cap = cap; ctx->wl_last_n = n; ctx->wl_pending = true;
}

djb_status eval_common(djb_ctx *ctx, const djb_brdf *b, int64_t n, const djb_vec3_view *i,
                       const djb_vec3_view *o, const djb_params *params, const djb_vec3_view *out_fr,
                       float *out_pdf, int mem, int want)
{
	if (!b) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null brdf");
	if (!ctx) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null ctx");
	djb_status st = cpu_pair_check(ctx, b);
	if (st != DJB_OK) return st;
	if (is_cpu(ctx)) return djbcpu::eval(ctx, b, n, i, o, params, out_fr, out_pdf, want);
	if (const djb_brdf *tw = scalar_twin(ctx, b, n, mem)) return djbcpu::eval(djbcpu::twin_ctx(), tw, n, i, o, params, out_fr, out_pdf, want);
	st = check_call(ctx, b, n, mem);
	if (st != DJB_OK) return st;
	std::lock_guard<std::recursive_mutex> call_lock(ctx->call_mu);
	Params p;
	if ((st = device_params(params, &p, b->dev.kind)) != DJB_OK) return st;
	if (mem == DJB_MEM_HOST && n > SMALL_N) {
		bool taken = false;
		st = eval_host_pipelined(ctx, b, n, i, o, params, out_fr, out_pdf, want, &taken);
		if (taken || st != DJB_OK) return st;
	}
	Staged sg(ctx, n, mem);
	View vi, vo, vout{ nullptr, nullptr, nullptr, 0 };
	float *dpdf = nullptr;
	if ((st = sg.in_vec(i, &vi)) != DJB_OK) return st;
	if ((st = sg.in_vec(o, &vo)) != DJB_OK) return st;
	if ((want & 3) && (st = sg.out_vec(out_fr, &vout)) != DJB_OK) return st;
	if ((want & 4) && (st = sg.out_arr(out_pdf, &dpdf)) != DJB_OK) return st;
	if (b->dev.kind == DJB_KIND_MERL && (want & 3) && !ctx->merl_exact_only) {
		// two-tier exact lookup; pair indices travel as uint32, so very large batches are chunked
		const long long CH = 1LL << 31;
		for (long long lo = 0; lo < n; lo += CH) {
			long long m = n - lo < CH ? n - lo : CH;
			// worklist: 16-byte header (count) + cap records of 32 bytes {k, i, o}; ~1 % of uniformly
			// distributed pairs are ambiguous, 2 % capacity; overflow falls back to a rescan
			const size_t REC = 32;
			wl_adapt(ctx);
			size_t cap = (size_t)((double)m * ctx->wl_frac) + 4096;
			size_t need = 16 + REC * cap;
			if (ctx->scratch_bytes < need) {
				HIP_TRY(hipStreamSynchronize(ctx->stream));
				if (ctx->scratch) (void)hipFree(ctx->scratch);
				ctx->scratch = nullptr; ctx->scratch_bytes = 0;
				HIP_TRY(hipMalloc(&ctx->scratch, need));
				ctx->scratch_bytes = need;
			}
			if (cap > 0xfffffff0ull) cap = 0xfffffff0ull;
			unsigned int *count = (unsigned int *)ctx->scratch, *list = count + 4;
			auto off = [&](const View &v) { return View{ v.x + lo * v.stride, v.y + lo * v.stride, v.z + lo * v.stride, v.stride }; };
			View oi = off(vi), oo = off(vo), ou = (want & 3) ? off(vout) : vout;
			HIP_TRY(djbk::launch_merl_twotier(ctx->stream, b->dev, m, oi, oo, ou, dpdf ? dpdf + lo : nullptr, want,
			                                  list, (unsigned int)cap, count));
			wl_note(ctx, count, cap, m);
		}
		return sg.finish();
	}
	if (b->dev.kind == DJB_KIND_UTIA && (want & 3) && !ctx->utia_exact_only) {
		// two-tier (djb_kernels_eval.hip): pair indices travel as uint32, so very large batches are chunked; the
		// worklist (16-byte header + 4 bytes per entry; ~2e-5 of the pairs need it) shares the context's scratch
		const long long CH = 1LL << 31;
		for (long long lo = 0; lo < n; lo += CH) {
			long long m = n - lo < CH ? n - lo : CH;
			size_t cap = (size_t)(m / 256 + 4096);
			if (const char *e = getenv("DJB_UTIA_WORKLIST_CAP")) cap = (size_t)strtoull(e, nullptr, 10);   // test hook: force the overflow path
			size_t need = 16 + 4 * (cap ? cap : 1);
			if (ctx->scratch_bytes < need) {
				HIP_TRY(hipStreamSynchronize(ctx->stream));
				if (ctx->scratch) (void)hipFree(ctx->scratch);
				ctx->scratch = nullptr; ctx->scratch_bytes = 0;
				HIP_TRY(hipMalloc(&ctx->scratch, need));
				ctx->scratch_bytes = need;
			}
			unsigned int *count = (unsigned int *)ctx->scratch, *list = count + 4;
			auto off = [&](const View &v) { return View{ v.x + lo * v.stride, v.y + lo * v.stride, v.z + lo * v.stride, v.stride }; };
			HIP_TRY(djbk::launch_utia_twotier(ctx->stream, b->dev, m, off(vi), off(vo), off(vout), dpdf ? dpdf + lo : nullptr, want,
			                                  list, (unsigned int)cap, count));
		}
		return sg.finish();
	}
	if (ctx->contract_1e5 && b->dev.kind == DJB_KIND_GGX && djbk::contract_supported(b->dev, p)) {
		auto al16 = [](const void *q) { return ((uintptr_t)q & 15) == 0; };
		const bool dense16 = vi.stride == 1 && vo.stride == 1 && al16(vi.x) && al16(vi.y) && al16(vi.z) && al16(vo.x) && al16(vo.y) && al16(vo.z) &&
		                     (!(want & 3) || (vout.stride == 1 && al16(vout.x) && al16(vout.y) && al16(vout.z))) && (!(want & 4) || al16(dpdf));
		if (dense16) {
			// as for MERL: pair indices travel as uint32; worklist = 16-byte header + 32-byte records {k, i, o}
			const long long CH = 1LL << 31;
			for (long long lo = 0; lo < n; lo += CH) {
				long long m = n - lo < CH ? n - lo : CH;
				const size_t REC = 32;
				wl_adapt(ctx);
				size_t cap = (size_t)((double)m * ctx->wl_frac) + 4096;
				size_t need = 16 + REC * cap;
				if (ctx->scratch_bytes < need) {
					HIP_TRY(hipStreamSynchronize(ctx->stream));
					if (ctx->scratch) (void)hipFree(ctx->scratch);
					ctx->scratch = nullptr; ctx->scratch_bytes = 0;
					HIP_TRY(hipMalloc(&ctx->scratch, need));
					ctx->scratch_bytes = need;
				}
				if (cap > 0xfffffff0ull) cap = 0xfffffff0ull;
				unsigned int *count = (unsigned int *)ctx->scratch, *list = count + 4;
				auto off = [&](const View &v) { return View{ v.x ? v.x + lo : nullptr, v.y ? v.y + lo : nullptr, v.z ? v.z + lo : nullptr, v.stride }; };
				HIP_TRY(djbk::launch_eval_contract(ctx->stream, b->dev, p, m, off(vi), off(vo), off(vout), dpdf ? dpdf + lo : nullptr, want,
				                                   list, (unsigned int)cap, count));
				wl_note(ctx, count, cap, m);
			}
			return sg.finish();
		}
	}
	HIP_TRY(djbk::launch_eval(ctx->stream, b->dev, p, n, vi, vo, vout, dpdf, want));
	return sg.finish();
}

djb_status alloc_brdf(djb_ctx *ctx, int kind, djb_brdf **out)
{
	djb_brdf *b = new djb_brdf();
	b->ctx = ctx;
	b->device = ctx->device;
	memset(&b->dev, 0, sizeof b->dev);
	b->dev.kind = kind;
	b->dev.shadow = 1;
	b->dev.fr.kind = djbdev::FR_IDEAL;
	b->alpha_beckmann = b->alpha_ggx = 0.0f;
	*out = b;
	return DJB_OK;
}

djb_status upload_floats(djb_brdf *b, const float *host, size_t count, const float **dev_out)
{
	float *d = nullptr;
	HIP_TRY(hipMalloc((void **)&d, sizeof(float) * (count ? count : 1)));
	b->allocs.push_back(d);
	HIP_TRY(hipMemcpy(d, host, sizeof(float) * count, hipMemcpyHostToDevice));
	*dev_out = d;
	return DJB_OK;
}

djb_status set_fresnel(djb_brdf *b, const djb_fresnel_desc *f)
{
	djbdev::Fresnel &fr = b->dev.fr;
	fr.kind = f ? f->kind : DJB_FRESNEL_IDEAL;
	fr.pts = nullptr; fr.npts = 0;
	if (!f) return DJB_OK;
	if (f->kind < DJB_FRESNEL_IDEAL || f->kind > DJB_FRESNEL_SPLINE)
		return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: unknown fresnel kind %d", f->kind);
	for (int c = 0; c < 3; ++c) { fr.a[c] = f->a[c]; fr.b[c] = f->b[c]; }
	if (f->kind == DJB_FRESNEL_SPLINE) {
		if (!f->points || f->npoints < 1)
			return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: fresnel::spline needs >= 1 point");
		b->fresnel.assign(f->points, f->points + 3 * (size_t)f->npoints);
		fr.npts = f->npoints;
		return upload_floats(b, b->fresnel.data(), b->fresnel.size(), &fr.pts);
	}
	return DJB_OK;
}

djb_status create_microfacet(djb_ctx *ctx, int kind, const djb_fresnel_desc *f, int shadow, djb_brdf **out)
{
	if (!ctx || !out) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null argument");
	HIP_TRY(hipSetDevice(ctx->device));
	djb_brdf *b;
	alloc_brdf(ctx, kind, &b);
	b->dev.shadow = shadow != 0;
	djb_status st = set_fresnel(b, f);
	if (st != DJB_OK) { djb_brdf_destroy(b); return st; }
	*out = b;
	return DJB_OK;
}

djb_status read_file(const char *path, size_t header_bytes, std::vector<char> *header,
                     size_t payload_bytes, std::vector<double> *payload, bool header_is_merl)
{
	FILE *f = fopen(path, "rb");
	if (!f) return fail(DJB_ERR_OPEN_FAILED, "djb_error: Failed to open %s\n", path);
	if (header_is_merl) {
		// dj_brdf.h:973-976 accepts any positive dims product and then indexes as 90x90x180; the product
		// is taken in 64 bits here (the header is untrusted) and anything but the MERL shape is refused
		// BEFORE the payload buffer is sized (a corrupt header must not be able to request gigabytes)
		int32_t dims[3] = { 0, 0, 0 };
		size_t got = fread(dims, 4, 3, f);
		const bool positive = got == 3 && dims[0] > 0 && dims[1] > 0 && dims[2] > 0;
		const long long n = positive ? (long long)dims[0] * (long long)dims[1] * (long long)dims[2] : 0;
		if (n <= 0) { fclose(f); return fail(DJB_ERR_BAD_HEADER, "djb_error: Failed to read MERL header\n"); }
		if (n != MERL_N) {
			fclose(f);
			return fail(DJB_ERR_BAD_HEADER, "djb_error: MERL table has %lld samples per channel, expected %lld\n", n, MERL_N);
		}
		payload_bytes = sizeof(double) * 3 * (size_t)n;
		header->assign((char *)dims, (char *)dims + 12);
	}
	(void)header_bytes;
	payload->resize(payload_bytes / sizeof(double));
	size_t got = fread(payload->data(), 1, payload_bytes, f);
	fclose(f);
	if (got != payload_bytes) return fail(DJB_ERR_READ_FAILED, "djb_error: Reading %s failed\n", path);
	return DJB_OK;
}

} // namespace

// No C++ exception may cross the C ABI (a ctypes / C caller would abort): every entry point is a
// function-try-block that maps std::bad_alloc and anything else to a status + message.
#define DJB_ABI_CATCH \
	catch (const std::bad_alloc &) { return fail(DJB_ERR_OUT_OF_MEMORY, "djb_error: out of host memory"); } \
	catch (const std::exception &ex_) { return fail(DJB_ERR_INTERNAL, "djb_error: internal error: %s", ex_.what()); } \
	catch (...) { return fail(DJB_ERR_INTERNAL, "djb_error: internal error"); }

// ============================================================================ C ABI
extern "C" {

const char *djb_last_error(void) { return g_err.c_str(); }
int djb_version(void) { return DJB_HIP_VERSION; }

djb_status djb_device_count(int *count)
try {
	if (!count) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null argument");
	*count = 0;
	int n = 0;
	hipError_t e = hipGetDeviceCount(&n);
	if (e != hipSuccess || n <= 0)
		return fail(DJB_ERR_NO_DEVICE, "djb_error: no HIP device (%s); libdjb_hip has no CPU path",
		            e != hipSuccess ? hipGetErrorString(e) : "0 devices");
	*count = n;
	return DJB_OK;
}
DJB_ABI_CATCH

static djb_status ctx_create(int device, void *hip_stream, bool own, djb_ctx **out)
{
	if (!out) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null argument");
	int n = 0;
	djb_status st = djb_device_count(&n);
	if (st != DJB_OK) return st;
	if (device < 0 || device >= n)
		return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: device %d out of range [0,%d)", device, n);
	HIP_TRY(hipSetDevice(device));
	djb_ctx *c = new djb_ctx();
	c->device = device;
	c->owns_stream = own;
	c->stream = (hipStream_t)hip_stream;
	c->scratch = nullptr; c->scratch_bytes = 0; c->merl_exact_only = 0;
	if (hipDeviceGetAttribute(&c->n_cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) { c->n_cus = 0; (void)hipGetLastError(); }
	if (own) {
		hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
		if (e != hipSuccess) { delete c; return fail(DJB_ERR_HIP, "djb_error: hipStreamCreate: %s", hipGetErrorString(e)); }
	}
	if (hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) {
		delete c; return fail(DJB_ERR_HIP, "djb_error: hipEventCreate failed");
	}
	// pinned, device-visible arena for small host-memory calls; without it they use the HBM staging path
	if (hipHostMalloc((void **)&c->pin, PIN_BYTES, hipHostMallocDefault) == hipSuccess) c->pin_bytes = PIN_BYTES;
	else { c->pin = nullptr; (void)hipGetLastError(); }
	*out = c;
	return DJB_OK;
}

djb_status djb_ctx_create(int device, djb_ctx **out)
try {
	djbhostlibm::init();      // once per process: is the host's libm the glibc the kernels restate?  (djb_cpu_libm.cpp)
	if (device == DJB_DEVICE_CPU) return djbcpu::ctx_create(out);
	return ctx_create(device, nullptr, true, out); }
DJB_ABI_CATCH
djb_status djb_ctx_create_on_stream(int device, void *hip_stream, djb_ctx **out)
try {
	djbhostlibm::init();
	if (device == DJB_DEVICE_CPU) return djbcpu::ctx_create(out);
	return ctx_create(device, hip_stream, false, out);
}
DJB_ABI_CATCH

// 1: the host's libm returned glibc 2.35's bits on the probe set (the host path calls it); 0: it did not, and the host
// path runs the kernels' restatements instead (unless DJB_HOST_LIBM=host); -1: not checked (CPU without FMA)
int djb_ctx_libm_matches_host(const djb_ctx *) { return djbhostlibm::init(); }
// 0: host-side libm calls go to the host's libm; 1: to the kernels' restatements of glibc 2.35's functions
int djb_host_libm_mode(void) { djbhostlibm::init(); return djbhostlibm::use_restated; }
// 1: the host's atan / log (not restated) gave glibc 2.35's values on the known-answer set; 0: they did not; -1: not checked
int djb_host_atan_log_kat(void) { return djbhostlibm::atan_log_kat(); }

djb_status djb_ctx_destroy(djb_ctx *ctx)
try {
	if (is_cpu(ctx)) return djbcpu::ctx_destroy(ctx);
	if (!ctx) return DJB_OK;
	(void)hipSetDevice(ctx->device);
	(void)hipStreamSynchronize(ctx->stream);
	(void)hipEventDestroy(ctx->ev0); (void)hipEventDestroy(ctx->ev1);
	if (ctx->scratch) (void)hipFree(ctx->scratch);
	if (ctx->wl_ev) (void)hipEventDestroy(ctx->wl_ev);
	if (ctx->wl_host) (void)hipHostFree(ctx->wl_host);
	for (auto &p : ctx->pool) (void)hipFree(p.first);
	if (ctx->pin) (void)hipHostFree(ctx->pin);
	if (ctx->d2h_stream) (void)hipStreamDestroy(ctx->d2h_stream);
	for (hipEvent_t e : ctx->pipe_ev) if (e) (void)hipEventDestroy(e);
	if (ctx->owns_stream) (void)hipStreamDestroy(ctx->stream);
	if (ctx->owned_stream) (void)hipStreamDestroy(ctx->owned_stream);
	delete ctx;
	return DJB_OK;
}
DJB_ABI_CATCH

djb_status djb_ctx_synchronize(djb_ctx *ctx)
try {
	if (is_cpu(ctx)) return DJB_OK;
	if (!ctx) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null ctx");
	HIP_TRY(hipStreamSynchronize(ctx->stream));
	return DJB_OK;
}
DJB_ABI_CATCH

void *djb_ctx_stream(djb_ctx *ctx) { return ctx && !is_cpu(ctx) ? (void *)ctx->stream : nullptr; }

djb_status djb_ctx_set_stream(djb_ctx *ctx, void *hip_stream)
try {
	if (!ctx) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null ctx");
	if (is_cpu(ctx)) return DJB_OK;
	std::lock_guard<std::recursive_mutex> call_lock(ctx->call_mu);
	hipStream_t ns = (hipStream_t)hip_stream;
	if (ns == ctx->stream) return DJB_OK;
	HIP_TRY(hipSetDevice(ctx->device));
	// work already enqueued by this context (and its per-context scratch) stays ordered before anything the
	// new stream will run: the new stream waits for an event recorded on the old one
	HIP_TRY(hipEventRecord(ctx->ev1, ctx->stream));
	HIP_TRY(hipStreamWaitEvent(ns, ctx->ev1, 0));
	if (ctx->owns_stream) { ctx->owned_stream = ctx->stream; ctx->owns_stream = false; }
	ctx->stream = ns;
	return DJB_OK;
}
DJB_ABI_CATCH

djb_status djb_timer_start(djb_ctx *ctx)
try {
	if (is_cpu(ctx)) return djbcpu::timer_start(ctx);
	if (!ctx) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null ctx");
	HIP_TRY(hipEventRecord(ctx->ev0, ctx->stream));
	return DJB_OK;
}
DJB_ABI_CATCH

djb_status djb_timer_stop_ms(djb_ctx *ctx, float *ms)
try {
	if (is_cpu(ctx) && ms) return djbcpu::timer_stop_ms(ctx, ms);
	if (!ctx || !ms) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null argument");
	HIP_TRY(hipEventRecord(ctx->ev1, ctx->stream));
	HIP_TRY(hipEventSynchronize(ctx->ev1));
	HIP_TRY(hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
	return DJB_OK;
}
DJB_ABI_CATCH

// ---------------------------------------------------------------- constructors
djb_status djb_brdf_create_beckmann(djb_ctx *ctx, const djb_fresnel_desc *f, int shadow, djb_brdf **out)
try {
	if (is_cpu(ctx) && out) return djbcpu::create_microfacet(ctx, DJB_KIND_BECKMANN, f, shadow, out);
	return create_microfacet(ctx, DJB_KIND_BECKMANN, f, shadow, out);
}
DJB_ABI_CATCH
djb_status djb_brdf_create_ggx(djb_ctx *ctx, const djb_fresnel_desc *f, int shadow, djb_brdf **out)
try {
	if (is_cpu(ctx) && out) return djbcpu::create_microfacet(ctx, DJB_KIND_GGX, f, shadow, out);
	return create_microfacet(ctx, DJB_KIND_GGX, f, shadow, out);
}
DJB_ABI_CATCH

djb_status djb_brdf_create_merl_from_memory(djb_ctx *ctx, const double *samples, int64_t n, djb_brdf **out)
try {
	if (is_cpu(ctx) && samples && out) return djbcpu::create_merl_from_memory(ctx, samples, n, out);
	if (!ctx || !samples || !out) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null argument");
	if (n <= 0) return fail(DJB_ERR_BAD_HEADER, "djb_error: Failed to read MERL header\n");
	// The reference accepts any positive dims product but indexes as 90x90x180 (dj_brdf.h:997-1008);
	// a smaller table would be read out of bounds there.  Refuse it here.
	if (n != MERL_N)
		return fail(DJB_ERR_BAD_HEADER, "djb_error: MERL table has %lld samples per channel, expected %lld\n",
		            (long long)n, MERL_N);
	HIP_TRY(hipSetDevice(ctx->device));
	djb_brdf *b;
	alloc_brdf(ctx, DJB_KIND_MERL, &b);
	double *raw = nullptr; djbdev::MerlTexel *tab = nullptr;
	hipError_t e = hipMalloc((void **)&raw, sizeof(double) * 3 * (size_t)n);
	if (e == hipSuccess) e = hipMalloc((void **)&tab, sizeof(djbdev::MerlTexel) * (size_t)n);
	if (e == hipSuccess) e = hipMemcpyAsync(raw, samples, sizeof(double) * 3 * (size_t)n, hipMemcpyHostToDevice, ctx->stream);
	if (e == hipSuccess) e = djbk::launch_merl_convert(ctx->stream, raw, n, tab);
	if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
	if (e != hipSuccess) {
		if (raw) (void)hipFree(raw);
		if (tab) (void)hipFree(tab);
		delete b;
		return fail(DJB_ERR_HIP, "djb_error: MERL upload failed: %s", hipGetErrorString(e));
	}
	b->allocs.push_back(tab);
	b->allocs.push_back(raw);
	b->raw_samples = raw; b->raw_count = 3 * (long long)n;
	b->dev.merl = tab;
	*out = b;
	return DJB_OK;
}
DJB_ABI_CATCH

djb_status djb_brdf_create_merl_from_file(djb_ctx *ctx, const char *path, djb_brdf **out)
try {
	if (is_cpu(ctx) && path && out) return djbcpu::create_merl_from_file(ctx, path, out);
	if (!ctx || !path || !out) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null argument");
	std::vector<char> hdr; std::vector<double> payload;
	djb_status st = read_file(path, 12, &hdr, 0, &payload, true);
	if (st != DJB_OK) return st;
	return djb_brdf_create_merl_from_memory(ctx, payload.data(), (int64_t)(payload.size() / 3), out);
}
DJB_ABI_CATCH

djb_status djb_brdf_create_utia_from_memory(djb_ctx *ctx, const double *samples, djb_brdf **out)
try {
	if (is_cpu(ctx) && samples && out) return djbcpu::create_utia_from_memory(ctx, samples, out);
	if (!ctx || !samples || !out) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null argument");
	HIP_TRY(hipSetDevice(ctx->device));
	djb_brdf *b;
	alloc_brdf(ctx, DJB_KIND_UTIA, &b);
	double *raw = nullptr; float4 *tab = nullptr;    // 288*288 records of eight float4 (k_utia_convert)
	hipError_t e = hipMalloc((void **)&raw, sizeof(double) * (size_t)UTIA_N);
	if (e == hipSuccess) e = hipMalloc((void **)&tab, sizeof(float4) * 8 * (size_t)(UTIA_N / 3));
	if (e == hipSuccess) e = hipMemcpyAsync(raw, samples, sizeof(double) * (size_t)UTIA_N, hipMemcpyHostToDevice, ctx->stream);
	if (e == hipSuccess) e = djbk::launch_utia_convert(ctx->stream, raw, UTIA_N, tab);
	if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
	if (e != hipSuccess) {
		if (raw) (void)hipFree(raw);
		if (tab) (void)hipFree(tab);
		delete b;
		return fail(DJB_ERR_HIP, "djb_error: UTIA upload failed: %s", hipGetErrorString(e));
	}
	b->allocs.push_back(tab);
	b->allocs.push_back(raw);
	b->raw_samples = raw; b->raw_count = UTIA_N;
	b->dev.utia = tab;
	*out = b;
	return DJB_OK;
}
DJB_ABI_CATCH

djb_status djb_brdf_create_utia_from_file(djb_ctx *ctx, const char *path, djb_brdf **out)
try {
	if (is_cpu(ctx) && path && out) return djbcpu::create_utia_from_file(ctx, path, out);
	if (!ctx || !path || !out) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null argument");
	std::vector<char> hdr; std::vector<double> payload;
	djb_status st = read_file(path, 0, &hdr, sizeof(double) * (size_t)UTIA_N, &payload, false);
	if (st != DJB_OK) return st;
	return djb_brdf_create_utia_from_memory(ctx, payload.data(), out);
}
DJB_ABI_CATCH

djb_status djb_brdf_create_lambert(djb_ctx *ctx, djb_brdf **out)
try {
	if (is_cpu(ctx) && out) return djbcpu::create_lambert(ctx, out);
	if (!ctx || !out) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null argument");
	return alloc_brdf(ctx, DJB_KIND_LAMBERT, out);
}
DJB_ABI_CATCH

// ---- sgd / abc: one row of the published parameter tables + the model's own Fresnel
struct SgdRow { const char *name, *other_name; double v[33]; };
struct AbcRow { const char *name; double v[9]; };
#include "build/djb_param_tables.inc"

static djb_status create_model(djb_ctx *ctx, int kind, const double *row, int count, djb_brdf **out)
{
	if (is_cpu(ctx)) return djbcpu::create_model(ctx, kind, row, count, out);
	HIP_TRY(hipSetDevice(ctx->device));
	djb_brdf *b;
	alloc_brdf(ctx, kind, &b);
	b->model_host.assign(row, row + count);
	double *d = nullptr;
	hipError_t e = hipMalloc((void **)&d, sizeof(double) * count);
	if (e == hipSuccess) e = hipMemcpy(d, row, sizeof(double) * count, hipMemcpyHostToDevice);
	if (e != hipSuccess) { if (d) (void)hipFree(d); delete b; return fail(DJB_ERR_HIP, "djb_error: %s", hipGetErrorString(e)); }
	b->allocs.push_back(d);
	b->dev.model = d;
	djbdev::Fresnel &fr = b->dev.fr;
	if (kind == DJB_KIND_SGD) {          // fresnel::sgd(vec3::from_raw(f0), vec3::from_raw(f1)), dj_brdf.h:3443
		fr.kind = djbdev::FR_SGD;
		for (int c = 0; c < 3; ++c) { fr.a[c] = (float)row[12 + c]; fr.b[c] = (float)row[15 + c]; }
	} else {                             // fresnel::unpolarized(vec3(ior)), dj_brdf.h:3623
		fr.kind = djbdev::FR_UNPOLARIZED;
		for (int c = 0; c < 3; ++c) fr.a[c] = (float)row[8];
	}
	*out = b;
	return DJB_OK;
}

djb_status djb_brdf_create_sgd_from_params(djb_ctx *ctx, const double *params33, djb_brdf **out)
try {
	if (!ctx || !params33 || !out) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null argument");
	return create_model(ctx, DJB_KIND_SGD, params33, 33, out);
}
DJB_ABI_CATCH
djb_status djb_brdf_create_abc_from_params(djb_ctx *ctx, const double *params9, djb_brdf **out)
try {
	if (!ctx || !params9 || !out) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null argument");
	return create_model(ctx, DJB_KIND_ABC, params9, 9, out);
}
DJB_ABI_CATCH
djb_status djb_brdf_create_sgd(djb_ctx *ctx, const char *name, djb_brdf **out)
try {
	if (!ctx || !name || !out) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null argument");
	for (const SgdRow &r : k_sgd_rows)
		if (!strcmp(r.name, name) || !strcmp(r.other_name, name))
			return create_model(ctx, DJB_KIND_SGD, r.v, 33, out);
	return fail(DJB_ERR_UNKNOWN_MATERIAL, "djb_error: No SGD parameters for %s\n", name);     // dj_brdf.h:3449
}
DJB_ABI_CATCH
djb_status djb_brdf_create_abc(djb_ctx *ctx, const char *name, djb_brdf **out)
try {
	if (!ctx || !name || !out) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null argument");
	for (const AbcRow &r : k_abc_rows)
		if (!strcmp(r.name, name))
			return create_model(ctx, DJB_KIND_ABC, r.v, 9, out);
	return fail(DJB_ERR_UNKNOWN_MATERIAL, "djb_error: No ABC parameters for %s\n", name);     // dj_brdf.h:3628
}
DJB_ABI_CATCH

djb_status djb_brdf_destroy(djb_brdf *b)
try {
	if (is_cpu(b)) return djbcpu::destroy(b);
	if (b && b->twin) djbcpu::destroy(b->twin);
	if (!b) return DJB_OK;
	(void)hipSetDevice(b->device);
	for (void *p : b->allocs) (void)hipFree(p);
	delete b;
	return DJB_OK;
}
DJB_ABI_CATCH

int djb_brdf_kind(const djb_brdf *b) { return !b ? -1 : is_cpu(b) ? djbcpu::kind(b) : b->dev.kind; }

djb_status djb_brdf_get_samples(const djb_brdf *b, double *out, int64_t capacity, int64_t *count)
try {
	if (is_cpu(b) && count) return djbcpu::get_samples(b, out, capacity, count);
	if (!b || !count) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null argument");
	if (b->dev.kind != DJB_KIND_MERL && b->dev.kind != DJB_KIND_UTIA)
		return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: get_samples needs a merl or utia BRDF");
	if (!b->raw_samples)
		return fail(DJB_ERR_NOT_IMPLEMENTED, "djb_error: this table was built by the file pipeline, which does not keep the payload");
	*count = b->raw_count;
	if (!out) return DJB_OK;
	if (capacity < b->raw_count) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: get_samples needs room for %lld doubles", b->raw_count);
	HIP_TRY(hipSetDevice(b->ctx->device));
	std::lock_guard<std::recursive_mutex> call_lock(b->ctx->call_mu);
	HIP_TRY(hipMemcpyAsync(out, b->raw_samples, sizeof(double) * (size_t)b->raw_count, hipMemcpyDeviceToHost, b->ctx->stream));
	HIP_TRY(hipStreamSynchronize(b->ctx->stream));
	if (b->dev.kind == DJB_KIND_UTIA) {      // utia::normalize, dj_brdf.h:1162-1177: clamp to zero, then *= (float_t)(1.f / 140.f)
		const float k = 1.f / 140.f;
		for (long long j = 0; j < b->raw_count; ++j) { double v = out[j] > 0.0 ? out[j] : 0.0; out[j] = v * k; }
	}
	return DJB_OK;
}
DJB_ABI_CATCH
int djb_brdf_get_shadow(const djb_brdf *b) { return !b ? -1 : is_cpu(b) ? djbcpu::get_shadow(b) : b->dev.shadow; }

static bool is_microfacet_kind(int k)
{
	return k == DJB_KIND_BECKMANN || k == DJB_KIND_GGX || k == DJB_KIND_TABULAR || k == DJB_KIND_TABULAR_ANISO;
}

djb_status djb_brdf_set_shadow(djb_brdf *b, int shadow)
try {
	if (is_cpu(b)) return djbcpu::set_shadow(b, shadow);
	if (!b || !is_microfacet_kind(b->dev.kind))
		return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: set_shadow needs a microfacet BRDF");
	b->dev.shadow = shadow != 0;
	if (b->twin) djbcpu::set_shadow(b->twin, shadow);
	return DJB_OK;
}
DJB_ABI_CATCH

djb_status djb_brdf_set_fresnel(djb_brdf *b, const djb_fresnel_desc *f)
try {
	if (is_cpu(b)) return djbcpu::set_fresnel(b, f);
	if (!b || !is_microfacet_kind(b->dev.kind))
		return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: set_fresnel needs a microfacet BRDF");
	HIP_TRY(hipSetDevice(b->ctx->device));
	// kernels receive the descriptor by value at launch; a replaced spline table stays allocated
	// until the handle is destroyed, so launches in flight are unaffected
	djbdev::Fresnel saved = b->dev.fr;
	std::vector<float> saved_pts = b->fresnel;
	djb_status st = set_fresnel(b, f);
	if (st != DJB_OK) { b->dev.fr = saved; b->fresnel = saved_pts; }
	else if (b->twin) st = djbcpu::set_fresnel(b->twin, f);
	return st;
}
DJB_ABI_CATCH

// ---------------------------------------------------------------- the fitter
static djb_status run_fit(djb_ctx *ctx, const std::vector<Brdf> &srcs, int src_kind, int res, int shadow,
                          float *alpha_b, float *alpha_g, float *p22, float *sigma, float *cdf,
                          float *qf, float *fresnel, int *n_qf_host)
{
	const int n_mat = (int)srcs.size(), cnt = res - 1;
	if (res <= 2) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: Invalid Resolution");   // dj_brdf.h:2218
	if (djbk::fit_lds_bytes(res) > 160 * 1024)
		return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: resolution %d exceeds the LDS budget of the fit kernel", res);
	Params std_p;
	djb_status st = device_params(nullptr, &std_p);
	if (st != DJB_OK) return st;
	// one HBM block from the context's recycled staging pool, carved into the kernel's work arrays and
	// outputs (eleven hipMalloc / hipFree pairs per call cost more than the fit of 100 materials)
	Staged pool(ctx, SMALL_N + 1, DJB_MEM_HOST);
	size_t total = 0;
	auto reserve = [&](size_t bytes) { size_t o = total; total += (bytes + 255) & ~(size_t)255; return o; };
	const size_t o_srcs = reserve(sizeof(Brdf) * n_mat);
	djbk::FitSplit split;
	split.parts = djbk::fit_parts(n_mat, ctx->n_cus);
	const size_t o_km = reserve(sizeof(double) * (size_t)n_mat * split.parts * cnt * cnt);
	const size_t o_sigx = reserve(sizeof(float) * (size_t)n_mat * res), o_done = reserve(sizeof(unsigned int) * 2 * n_mat);
	const size_t o_ratio = reserve(sizeof(float) * 3 * (size_t)n_mat * cnt * (cnt + 1));
	const size_t o_p22 = reserve(sizeof(float) * (size_t)n_mat * res), o_sigma = reserve(sizeof(float) * (size_t)n_mat * res);
	const size_t o_cdf = reserve(sizeof(float) * (size_t)n_mat * res), o_qf = reserve(sizeof(float) * (size_t)n_mat * res);
	const size_t o_fres = reserve(sizeof(float) * 3 * (size_t)n_mat * res);
	const size_t o_ab = reserve(sizeof(float) * n_mat), o_ag = reserve(sizeof(float) * n_mat), o_nqf = reserve(sizeof(int) * n_mat);
	char *base = nullptr;
	if ((st = pool.alloc(total, (void **)&base)) != DJB_OK) return st;
	Brdf *d_srcs = (Brdf *)(base + o_srcs);
	double *km = (double *)(base + o_km);
	float *ratio = (float *)(base + o_ratio);
	djbk::FitOut o;
	o.p22 = (float *)(base + o_p22); o.sigma = (float *)(base + o_sigma); o.cdf = (float *)(base + o_cdf);
	o.qf = (float *)(base + o_qf); o.fresnel = (float *)(base + o_fres);
	o.alpha_beckmann = (float *)(base + o_ab); o.alpha_ggx = (float *)(base + o_ag); o.n_qf = (int *)(base + o_nqf);
	{
		hipError_t ce = hipMemcpyAsync(d_srcs, srcs.data(), sizeof(Brdf) * n_mat, hipMemcpyHostToDevice, ctx->stream);
		if (ce != hipSuccess) { (void)hipStreamSynchronize(ctx->stream); (void)hipGetLastError(); return fail(DJB_ERR_HIP, "djb_error: fit upload failed: %s", hipGetErrorString(ce)); }
	}
	split.sig_x = (float *)(base + o_sigx); split.sig_done = (unsigned int *)(base + o_done);
	// from here on the kernel may be running on `base`: every exit synchronises the stream before `pool`
	// hands the block back to the context (and before `staging` goes out of scope)
	hipError_t e = djbk::launch_fit(ctx->stream, d_srcs, src_kind, std_p, n_mat, res, shadow != 0, km, ratio, o, split);
	// the outputs are one contiguous range of the block [o_p22, total): ONE pageable device-to-host copy
	// (the one-copy-at-a-time rule of Staged::copy), unpacked on the host after the sync
	std::vector<char> staging(total - o_p22);
	if (e == hipSuccess) e = hipMemcpyAsync(staging.data(), base + o_p22, staging.size(), hipMemcpyDeviceToHost, ctx->stream);
	hipError_t se = hipStreamSynchronize(ctx->stream);
	if (e == hipSuccess) e = se;
	if (e != hipSuccess) {
		(void)hipGetLastError();
		return fail(DJB_ERR_HIP, "djb_error: fit failed: %s", hipGetErrorString(e));
	}
	auto back = [&](void *h, size_t off, size_t bytes) { if (h) memcpy(h, staging.data() + (off - o_p22), bytes); };
	back(alpha_b, o_ab, sizeof(float) * n_mat);
	back(alpha_g, o_ag, sizeof(float) * n_mat);
	back(p22, o_p22, sizeof(float) * (size_t)n_mat * res);
	back(sigma, o_sigma, sizeof(float) * (size_t)n_mat * res);
	back(cdf, o_cdf, sizeof(float) * (size_t)n_mat * res);
	back(qf, o_qf, sizeof(float) * (size_t)n_mat * res);
	back(fresnel, o_fres, sizeof(float) * 3 * (size_t)n_mat * res);
	back(n_qf_host, o_nqf, sizeof(int) * n_mat);
	return DJB_OK;
}

djb_status djb_brdf_create_tabular(djb_ctx *ctx, const djb_brdf *src, int res, int shadow, djb_brdf **out)
try {
	if (is_cpu(ctx) && src && out) {
		if (!is_cpu(src)) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: a CPU context fits BRDFs of a CPU context");
		return djbcpu::create_tabular(ctx, src, res, shadow, out);
	}
	if (src && is_cpu(src)) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: brdf belongs to a CPU context");
	if (!ctx || !src || !out) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null argument");
	djb_status st = check_call(ctx, src, 0, DJB_MEM_DEVICE);
	if (st != DJB_OK) return st;
	std::lock_guard<std::recursive_mutex> call_lock(ctx->call_mu);
	if (res <= 2) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: Invalid Resolution");
	djb_brdf *t;
	alloc_brdf(ctx, DJB_KIND_TABULAR, &t);
	t->dev.shadow = shadow != 0;
	t->p22.resize(res); t->sigma.resize(res); t->cdf.resize(res); t->qf.resize(res); t->fresnel.resize(3 * (size_t)res);
	int n_qf = 0;
	std::vector<Brdf> srcs(1, src->dev);
	st = run_fit(ctx, srcs, src->dev.kind, res, shadow, &t->alpha_beckmann, &t->alpha_ggx, t->p22.data(),
	             t->sigma.data(), t->cdf.data(), t->qf.data(), t->fresnel.data(), &n_qf);
	if (st != DJB_OK) { djb_brdf_destroy(t); return st; }
	t->qf.resize(n_qf);
	t->dev.n_p22 = res; t->dev.n_sigma = res; t->dev.n_cdf = res; t->dev.n_qf = n_qf;
	t->dev.fr.kind = djbdev::FR_SPLINE; t->dev.fr.npts = res;
	if ((st = upload_floats(t, t->p22.data(), res, &t->dev.p22)) != DJB_OK ||
	    (st = upload_floats(t, t->sigma.data(), res, &t->dev.sigma)) != DJB_OK ||
	    (st = upload_floats(t, t->cdf.data(), res, &t->dev.cdf)) != DJB_OK ||
	    (st = upload_floats(t, t->qf.data(), n_qf, &t->dev.qf)) != DJB_OK ||
	    (st = upload_floats(t, t->fresnel.data(), 3 * (size_t)res, &t->dev.fr.pts)) != DJB_OK) {
		djb_brdf_destroy(t); return st;
	}
	*out = t;
	return DJB_OK;
}
DJB_ABI_CATCH

// djb::tabular_anisotropic(brdf, elevation_res, azimuthal_res, shadow), dj_brdf.h:2238-2273
djb_status djb_brdf_create_tabular_anisotropic(djb_ctx *ctx, const djb_brdf *src, int elev, int azim,
                                               int shadow, djb_brdf **out)
try {
	if (is_cpu(ctx) && src && out) {
		if (!is_cpu(src)) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: a CPU context fits BRDFs of a CPU context");
		return djbcpu::create_tabular_anisotropic(ctx, src, elev, azim, shadow, out);
	}
	if (src && is_cpu(src)) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: brdf belongs to a CPU context");
	if (!ctx || !src || !out) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null argument");
	djb_status st = check_call(ctx, src, 0, DJB_MEM_DEVICE);
	if (st != DJB_OK) return st;
	std::lock_guard<std::recursive_mutex> call_lock(ctx->call_mu);
	if (elev <= 1 || azim <= 1 || elev > 1024 || azim > 1024)
		return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: Invalid Resolution");           // dj_brdf.h:2244
	Params std_p;
	if ((st = device_params(nullptr, &std_p)) != DJB_OK) return st;
	const size_t E = elev, A = azim, w = E - 1, N = w * A, G = E * A;
	// one HBM block: outputs first (they stay alive with the object), work arrays after
	struct Carve { size_t off = 0; size_t take(size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; } } cv;
	size_t o_p22 = cv.take(4 * G), o_sig = cv.take(4 * G), o_pdf1 = cv.take(4 * A), o_cdf1 = cv.take(4 * A),
	       o_qf1 = cv.take(4 * A), o_pdf2 = cv.take(4 * G), o_cdf2 = cv.take(4 * G), o_qf2 = cv.take(4 * G),
	       o_fres = cv.take(12 * E), o_fit = cv.take(4 * 10), o_cnt = cv.take(4 * 4);
	size_t o_f8[8]; for (int k = 0; k < 8; ++k) o_f8[k] = cv.take(4 * N);
	size_t o_v0 = cv.take(8 * N), o_v1 = cv.take(8 * N), o_terms = cv.take(4 * djbk::aniso_terms_count()),
	       o_ndf = cv.take(4 * djbk::aniso_ndf_count()), o_cosd = cv.take(8 * djbk::aniso_cosd_count(azim)),
	       o_st = cv.take(4 * djbk::aniso_sig_nodes()), o_ss = cv.take(4 * djbk::aniso_sig_nodes()),
	       o_sc = cv.take(8 * djbk::aniso_sig_nodes()), o_ratio = cv.take(12 * w * E),
	       o_probes = cv.take(4 * A * 8 * w), o_rowk = cv.take(4 * A), o_qrows = cv.take(4 * G), o_qlen = cv.take(4 * A);
	unsigned char *blk = nullptr;
	HIP_TRY(hipMalloc((void **)&blk, cv.off));
	hipError_t e = hipMemsetAsync(blk, 0, cv.off, ctx->stream);
	djbk::AnisoScratch S;
	S.elev = elev; S.azim = azim;
	auto F4 = [&](size_t o) { return (float *)(blk + o); };
	S.p22 = F4(o_p22); S.sigma = F4(o_sig); S.pdf1 = F4(o_pdf1); S.cdf1 = F4(o_cdf1); S.qf1 = F4(o_qf1);
	S.pdf2 = F4(o_pdf2); S.cdf2 = F4(o_cdf2); S.qf2 = F4(o_qf2); S.fres = F4(o_fres); S.fit = F4(o_fit);
	S.counts = (int *)(blk + o_cnt);
	S.k1 = F4(o_f8[0]); S.xo = F4(o_f8[1]); S.yo = F4(o_f8[2]); S.zo = F4(o_f8[3]);
	S.s1 = F4(o_f8[4]); S.s2 = F4(o_f8[5]); S.tn = F4(o_f8[6]); S.dn = F4(o_f8[7]);
	S.v0 = (double *)(blk + o_v0); S.v1 = (double *)(blk + o_v1);
	S.terms = F4(o_terms); S.ndf_tab = F4(o_ndf); S.cosd = (double *)(blk + o_cosd);
	S.sig_theta = F4(o_st); S.sig_sin = F4(o_ss); S.sig_cosd = (double *)(blk + o_sc);
	S.ratio = F4(o_ratio); S.probes = F4(o_probes); S.rowk = F4(o_rowk);
	S.qf2_rows = F4(o_qrows); S.qf2_len = (int *)(blk + o_qlen); S.qf2_aligned = ctx->aniso_qf2_aligned;
	if (e == hipSuccess) e = djbk::launch_fit_aniso(ctx->stream, src->dev, std_p, S, shadow != 0);
	djb_brdf *t;
	alloc_brdf(ctx, DJB_KIND_TABULAR_ANISO, &t);
	t->allocs.push_back(blk);
	t->elev = elev; t->azim = azim;
	const size_t sizes[8] = { G, G, A, A, A, G, G, G };
	float *const srcs8[8] = { S.p22, S.sigma, S.pdf1, S.cdf1, S.qf1, S.pdf2, S.cdf2, S.qf2 };
	for (int k = 0; k < 8 && e == hipSuccess; ++k) {
		t->aniso[k].resize(sizes[k]);
		e = hipMemcpyAsync(t->aniso[k].data(), srcs8[k], 4 * sizes[k], hipMemcpyDeviceToHost, ctx->stream);
	}
	t->fresnel.resize(3 * E);
	int counts[4] = { 0, 0, 0, 0 };
	if (e == hipSuccess) e = hipMemcpyAsync(t->fresnel.data(), S.fres, 12 * E, hipMemcpyDeviceToHost, ctx->stream);
	if (e == hipSuccess) e = hipMemcpyAsync(t->aniso_fit, S.fit, 40, hipMemcpyDeviceToHost, ctx->stream);
	if (e == hipSuccess) e = hipMemcpyAsync(counts, S.counts, 16, hipMemcpyDeviceToHost, ctx->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
	if (e != hipSuccess) { djb_brdf_destroy(t); return fail(DJB_ERR_HIP, "djb_error: anisotropic fit failed: %s", hipGetErrorString(e)); }
	// counts[1] = azimuth rows whose conditional CDF could not be inverted for every quantile (the w-node CDF
	// can stay below (w-1)/w at the last probe for grazing-heavy data).  The reference's m_qf2 then comes up
	// short and every later row is misaligned (dj_brdf.h:3005-3034); ka_qf2_layout reproduces exactly that
	// vector (counts[2] entries; what the reference reads past its end is 1.0 here) unless
	// DJB_OPT_ANISO_QF2_ALIGNED is set on the context.  eval / pdf never touch this table.
	t->aniso_qf2_entries = counts[2];
	t->aniso[4].resize(counts[0]);                      // m_qf1 may be shorter than azim (scan quirk)
	Brdf &d = t->dev;
	d.shadow = shadow != 0;
	d.p22 = S.p22; d.sigma = S.sigma; d.n_p22 = d.n_sigma = (int)G;
	d.a_pdf1 = S.pdf1; d.a_cdf1 = S.cdf1; d.a_qf1 = S.qf1; d.a_pdf2 = S.pdf2; d.a_cdf2 = S.cdf2; d.a_qf2 = S.qf2;
	d.elev = elev; d.azim = azim; d.n_a_cdf1 = azim; d.n_a_qf1 = counts[0];
	d.fr.kind = djbdev::FR_SPLINE; d.fr.pts = S.fres; d.fr.npts = elev;
	*out = t;
	return DJB_OK;
}
DJB_ABI_CATCH

djb_status djb_tabular_anisotropic_get(const djb_brdf *tab, int which, float *outp, int *count, int *elev, int *azim)
try {
	if (is_cpu(tab)) return djbcpu::aniso_get(tab, which, outp, count, elev, azim);
	if (!tab || tab->dev.kind != DJB_KIND_TABULAR_ANISO)
		return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: not a tabular_anisotropic brdf");
	if (elev) *elev = tab->elev;
	if (azim) *azim = tab->azim;
	const std::vector<float> *v;
	if (which == DJB_ATAB_QF2_ENTRIES) { if (count) *count = tab->aniso_qf2_entries; return DJB_OK; }
	if (which >= 0 && which < 8) v = &tab->aniso[which];
	else if (which == DJB_ATAB_FRESNEL) v = &tab->fresnel;
	else return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: unknown table %d", which);
	if (count) *count = (int)(which == DJB_ATAB_FRESNEL ? v->size() / 3 : v->size());
	if (outp) memcpy(outp, v->data(), sizeof(float) * v->size());
	return DJB_OK;
}
DJB_ABI_CATCH

djb_status djb_tabular_anisotropic_fit(const djb_brdf *tab, djb_params *beckmann, djb_params *ggx)
try {
	if (is_cpu(tab)) return djbcpu::aniso_fit(tab, beckmann, ggx);
	if (!tab || tab->dev.kind != DJB_KIND_TABULAR_ANISO)
		return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: not a tabular_anisotropic brdf");
	for (int k = 0; k < 2; ++k) {
		djb_params *p = k == 0 ? beckmann : ggx;
		if (!p) continue;
		p->kind = DJB_PARAMS_PDFPARAMS;
		for (int c = 0; c < 5; ++c) p->v[c] = tab->aniso_fit[5 * k + c];
	}
	return DJB_OK;
}
DJB_ABI_CATCH

djb_status djb_tabular_get(const djb_brdf *tab, int which, float *outp, int *count)
try {
	if (is_cpu(tab)) return djbcpu::tabular_get(tab, which, outp, count);
	if (!tab || tab->dev.kind != DJB_KIND_TABULAR)
		return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: not a tabular brdf");
	const std::vector<float> *v;
	switch (which) {
	case DJB_TAB_P22: v = &tab->p22; break;
	case DJB_TAB_SIGMA: v = &tab->sigma; break;
	case DJB_TAB_CDF: v = &tab->cdf; break;
	case DJB_TAB_QF: v = &tab->qf; break;
	case DJB_TAB_FRESNEL: v = &tab->fresnel; break;
	default: return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: unknown table %d", which);
	}
	if (count) *count = (int)(which == DJB_TAB_FRESNEL ? v->size() / 3 : v->size());
	if (outp) memcpy(outp, v->data(), sizeof(float) * v->size());
	return DJB_OK;
}
DJB_ABI_CATCH

djb_status djb_tabular_fit(const djb_brdf *tab, float *alpha_beckmann, float *alpha_ggx)
try {
	if (is_cpu(tab)) return djbcpu::tabular_fit(tab, alpha_beckmann, alpha_ggx);
	if (!tab || tab->dev.kind != DJB_KIND_TABULAR)
		return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: not a tabular brdf");
	if (alpha_beckmann) *alpha_beckmann = tab->alpha_beckmann;
	if (alpha_ggx) *alpha_ggx = tab->alpha_ggx;
	return DJB_OK;
}
DJB_ABI_CATCH

djb_status djb_fit_merl_batch(djb_ctx *ctx, int n_mat, const double *const *tables, int res, int shadow,
                              float *alpha_beckmann, float *alpha_ggx, float *p22, float *sigma,
                              float *cdf, float *qf, float *fresnel)
try {
	if (is_cpu(ctx) && tables && n_mat >= 0) return n_mat == 0 ? DJB_OK : djbcpu::fit_merl_batch(ctx, n_mat, tables, res, shadow, alpha_beckmann, alpha_ggx, p22, sigma, cdf, qf, fresnel);
	if (!ctx || !tables || n_mat < 0) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: invalid argument");
	if (n_mat == 0) return DJB_OK;
	djb_status st = check_call(ctx, nullptr, 0, DJB_MEM_DEVICE);
	if (st != DJB_OK) return st;
	std::lock_guard<std::recursive_mutex> call_lock(ctx->call_mu);
	// upload + convert every table (raw doubles -> texel table), double-buffered on the stream
	std::vector<djb_brdf *> mats(n_mat, nullptr);
	std::vector<Brdf> srcs(n_mat);
	for (int m = 0; m < n_mat; ++m) {
		st = djb_brdf_create_merl_from_memory(ctx, tables[m], MERL_N, &mats[m]);
		if (st != DJB_OK) break;
		srcs[m] = mats[m]->dev;
	}
	if (st == DJB_OK)
		st = run_fit(ctx, srcs, DJB_KIND_MERL, res, shadow, alpha_beckmann, alpha_ggx, p22, sigma, cdf, qf, fresnel, nullptr);
	for (djb_brdf *b : mats) djb_brdf_destroy(b);
	return st;
}
DJB_ABI_CATCH

djb_status djb_fit_brdf_batch(djb_ctx *ctx, int n_mat, const djb_brdf *const *srcs_in, int res, int shadow,
                              float *alpha_beckmann, float *alpha_ggx, float *p22, float *sigma,
                              float *cdf, float *qf, float *fresnel)
try {
	if (is_cpu(ctx) && srcs_in && n_mat >= 0) return n_mat == 0 ? DJB_OK : djbcpu::fit_brdf_batch(ctx, n_mat, srcs_in, res, shadow, alpha_beckmann, alpha_ggx, p22, sigma, cdf, qf, fresnel);
	if (!ctx || !srcs_in || n_mat < 0) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: invalid argument");
	if (n_mat == 0) return DJB_OK;
	djb_status st = check_call(ctx, srcs_in[0], 0, DJB_MEM_DEVICE);
	if (st != DJB_OK) return st;
	std::lock_guard<std::recursive_mutex> call_lock(ctx->call_mu);
	std::vector<Brdf> srcs(n_mat);
	for (int m = 0; m < n_mat; ++m) {
		if (!srcs_in[m] || srcs_in[m]->dev.kind != srcs_in[0]->dev.kind || srcs_in[m]->device != ctx->device)
			return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: batch fit needs BRDFs of one kind on the ctx device");
		srcs[m] = srcs_in[m]->dev;
	}
	return run_fit(ctx, srcs, srcs[0].kind, res, shadow, alpha_beckmann, alpha_ggx, p22, sigma, cdf, qf, fresnel, nullptr);
}
DJB_ABI_CATCH

// ---------------------------------------------------------------- the operator surface
djb_status djb_eval_batch(djb_ctx *ctx, const djb_brdf *b, int64_t n, const djb_vec3_view *i,
                          const djb_vec3_view *o, const djb_params *params, const djb_vec3_view *out, int mem)
try {
	return eval_common(ctx, b, n, i, o, params, out, nullptr, mem, 1);
}
DJB_ABI_CATCH
djb_status djb_evalp_batch(djb_ctx *ctx, const djb_brdf *b, int64_t n, const djb_vec3_view *i,
                           const djb_vec3_view *o, const djb_params *params, const djb_vec3_view *out, int mem)
try {
	return eval_common(ctx, b, n, i, o, params, out, nullptr, mem, 2);
}
DJB_ABI_CATCH
djb_status djb_pdf_batch(djb_ctx *ctx, const djb_brdf *b, int64_t n, const djb_vec3_view *i,
                         const djb_vec3_view *o, const djb_params *params, float *out_pdf, int mem)
try {
	return eval_common(ctx, b, n, i, o, params, nullptr, out_pdf, mem, 4);
}
DJB_ABI_CATCH
djb_status djb_eval_pdf_batch(djb_ctx *ctx, const djb_brdf *b, int64_t n, const djb_vec3_view *i,
                              const djb_vec3_view *o, const djb_params *params, int want_cos,
                              const djb_vec3_view *out_fr, float *out_pdf, int mem)
try {
	return eval_common(ctx, b, n, i, o, params, out_fr, out_pdf, mem, want_cos ? 6 : 5);
}
DJB_ABI_CATCH

static djb_status sample_common(djb_ctx *ctx, const djb_brdf *b, int64_t n, const float *u1, const float *u2,
                                const djb_vec3_view *o, const djb_params *params, const djb_vec3_view *out_w,
                                const djb_vec3_view *out_i, float *out_pdf, int mem, bool is)
{
	if (!b) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null brdf");
	if (!ctx) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null ctx");
	djb_status st = cpu_pair_check(ctx, b);
	if (st != DJB_OK) return st;
	if (is_cpu(ctx) || scalar_twin(ctx, b, n, mem)) {
		const bool on_cpu = is_cpu(ctx);
		if (is && (!out_w || !out_pdf)) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null argument");
		if (!u1 || !u2) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null input array");
		return djbcpu::sample(on_cpu ? ctx : djbcpu::twin_ctx(), on_cpu ? b : b->twin, n, u1, u2, 0, 0, 0, o, params, is ? out_w : nullptr, out_i, out_pdf);
	}
	st = check_call(ctx, b, n, mem);
	if (st != DJB_OK) return st;
	std::lock_guard<std::recursive_mutex> call_lock(ctx->call_mu);
	Params p;
	if ((st = device_params(params, &p, b->dev.kind)) != DJB_OK) return st;
	if (mem == DJB_MEM_HOST && n > SMALL_N && o && out_i && (!is || out_w)) {   // large host batch: chunked, both PCIe directions busy
		bool taken = false;
		std::vector<PipeArr> ins{ PipeArr::arr(u1), PipeArr::arr(u2), PipeArr::vec(o) }, outs{ PipeArr::vec(out_i) };
		if (is) { outs.push_back(PipeArr::vec(out_w)); outs.push_back(PipeArr::arr(out_pdf)); }
		st = host_pipeline(ctx, n, ins, outs, [&](long long m, int s) {
			djb_vec3_view dvo = ins[2].view(s), dvi = outs[0].view(s), dvw = is ? outs[1].view(s) : djb_vec3_view{ nullptr, nullptr, nullptr, 0 };
			return sample_common(ctx, b, m, ins[0].dev[s], ins[1].dev[s], &dvo, params, is ? &dvw : nullptr, &dvi,
			                     is ? outs[2].dev[s] : nullptr, DJB_MEM_DEVICE, is);
		}, &taken);
		if (taken || st != DJB_OK) return st;
	}
	Staged sg(ctx, n, mem);
	View vo, vi, vw; const float *d1, *d2; float *dpdf = nullptr;
	if ((st = sg.in_f(u1, &d1)) != DJB_OK) return st;
	if ((st = sg.in_f(u2, &d2)) != DJB_OK) return st;
	if ((st = sg.in_vec(o, &vo)) != DJB_OK) return st;
	if ((st = sg.out_vec(out_i, &vi)) != DJB_OK) return st;
	if (is) {
		if ((st = sg.out_vec(out_w, &vw)) != DJB_OK) return st;
		if ((st = sg.out_arr(out_pdf, &dpdf)) != DJB_OK) return st;
	}
	HIP_TRY(djbk::launch_sample(ctx->stream, b->dev, p, n, d1, d2, 0, 0, 0, vo, vi, is ? &vw : nullptr, dpdf));
	return sg.finish();
}

djb_status djb_sample_batch(djb_ctx *ctx, const djb_brdf *b, int64_t n, const float *u1, const float *u2,
                            const djb_vec3_view *o, const djb_params *params, const djb_vec3_view *out_i, int mem)
try {
	return sample_common(ctx, b, n, u1, u2, o, params, nullptr, out_i, nullptr, mem, false);
}
DJB_ABI_CATCH

djb_status djb_evalp_is_batch(djb_ctx *ctx, const djb_brdf *b, int64_t n, const float *u1, const float *u2,
                              const djb_vec3_view *o, const djb_params *params, const djb_vec3_view *out_w,
                              const djb_vec3_view *out_i, float *out_pdf, int mem)
try {
	return sample_common(ctx, b, n, u1, u2, o, params, out_w, out_i, out_pdf, mem, true);
}
DJB_ABI_CATCH

djb_status djb_sample_rng_batch(djb_ctx *ctx, const djb_brdf *b, int64_t n, uint32_t seed_u1, uint32_t seed_u2,
                                uint64_t start, const djb_vec3_view *o, const djb_params *params,
                                const djb_vec3_view *out_i)
try {
	if (is_cpu(ctx)) {
		if (b && !is_cpu(b)) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: brdf belongs to a GPU context");
		return djbcpu::sample(ctx, b, n, nullptr, nullptr, seed_u1, seed_u2, start, o, params, nullptr, out_i, nullptr);
	}
	if (b && is_cpu(b)) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: brdf belongs to a CPU context");
	if (!b) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null brdf");
	djb_status st = check_call(ctx, b, n, DJB_MEM_DEVICE);
	if (st != DJB_OK) return st;
	std::lock_guard<std::recursive_mutex> call_lock(ctx->call_mu);
	Params p;
	if ((st = device_params(params, &p, b->dev.kind)) != DJB_OK) return st;
	if (!Staged::valid(o) || !Staged::valid(out_i)) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null vec3 view");
	View vo{ o->x, o->y, o->z, (long long)o->stride }, vi{ out_i->x, out_i->y, out_i->z, (long long)out_i->stride };
	HIP_TRY(djbk::launch_sample(ctx->stream, b->dev, p, n, nullptr, nullptr, seed_u1, seed_u2, start, vo, vi, nullptr, nullptr));
	return DJB_OK;
}
DJB_ABI_CATCH

static djb_status hd_common(djb_ctx *ctx, int64_t n, const djb_vec3_view *a, const djb_vec3_view *b,
                            const djb_vec3_view *c, const djb_vec3_view *d, int mem, bool inverse)
{
	if (!ctx) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null ctx");
	if (is_cpu(ctx)) return djbcpu::io_hd(ctx, n, a, b, c, d, inverse);
	if (mem == DJB_MEM_HOST && n >= 0 && n <= SCALAR_HOST_MAX && !ctx->scalar_on_device) return djbcpu::io_hd(djbcpu::twin_ctx(), n, a, b, c, d, inverse);
	djb_status st = check_call(ctx, nullptr, n, mem);
	if (st != DJB_OK) return st;
	std::lock_guard<std::recursive_mutex> call_lock(ctx->call_mu);
	Staged sg(ctx, n, mem);
	View va, vb, vc, vd;
	if ((st = sg.in_vec(a, &va)) != DJB_OK) return st;
	if ((st = sg.in_vec(b, &vb)) != DJB_OK) return st;
	if ((st = sg.out_vec(c, &vc)) != DJB_OK) return st;
	if ((st = sg.out_vec(d, &vd)) != DJB_OK) return st;
	HIP_TRY(djbk::launch_io_to_hd(ctx->stream, n, va, vb, vc, vd, inverse));
	return sg.finish();
}
djb_status djb_io_to_hd_batch(djb_ctx *ctx, int64_t n, const djb_vec3_view *i, const djb_vec3_view *o,
                              const djb_vec3_view *h, const djb_vec3_view *d, int mem)
try {
	return hd_common(ctx, n, i, o, h, d, mem, false);
}
DJB_ABI_CATCH
djb_status djb_hd_to_io_batch(djb_ctx *ctx, int64_t n, const djb_vec3_view *h, const djb_vec3_view *d,
                              const djb_vec3_view *i, const djb_vec3_view *o, int mem)
try {
	return hd_common(ctx, n, h, d, i, o, mem, true);
}
DJB_ABI_CATCH

djb_status djb_query_batch(djb_ctx *ctx, const djb_brdf *b, int which, int64_t n, const djb_vec3_view *a,
                           const djb_vec3_view *bb, const djb_vec3_view *c, const djb_params *params,
                           const djb_vec3_view *out, int mem)
try {
	if (!b) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null brdf");
	if (!ctx) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null ctx");
	const int bkind = djb_brdf_kind(b);
	const bool aniso = bkind == DJB_KIND_TABULAR_ANISO;
	const bool model = bkind == DJB_KIND_SGD || bkind == DJB_KIND_ABC;
	const bool model_q = which >= DJB_Q_MODEL_NDF && which <= DJB_Q_MODEL_G1;
	if (model) {
		if (!(model_q || which == DJB_Q_FRESNEL) || (which == DJB_Q_MODEL_G1 && bkind != DJB_KIND_SGD))
			return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: sgd / abc answer ndf, gaf, fresnel (and g1 for sgd) only");
	} else if (model_q)
		return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: DJB_Q_MODEL_* need an sgd or abc brdf");
	if (bkind > DJB_KIND_TABULAR && !aniso && !model)
		return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: queries need a microfacet brdf");
	if ((which >= DJB_Q_QF2_RADIAL && which <= DJB_Q_QF1) && (bkind == DJB_KIND_TABULAR || aniso))
		return fail(DJB_ERR_NOT_IMPLEMENTED, "djb_error: Not Implemented");          // dj_brdf.h:1854, 1859
	if ((which >= DJB_Q_P22_RADIAL && which <= DJB_Q_QF1) && aniso)
		return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: tabular_anisotropic is not a radial microfacet");
	if ((which >= DJB_Q_ANISO_PDF1 && which <= DJB_Q_ANISO_QF2) && !aniso)
		return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: pdf1/cdf1/qf1/pdf2/cdf2/qf2 need a tabular_anisotropic");
	djb_status st = cpu_pair_check(ctx, b);
	if (st != DJB_OK) return st;
	if (!Staged::valid(a) || !Staged::valid(out)) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null vec3 view");
	if (is_cpu(ctx)) return n <= 0 ? DJB_OK : djbcpu::query(ctx, b, which, n, a, bb ? bb : a, c ? c : a, params, out);
	if (const djb_brdf *tw = scalar_twin(ctx, b, n, mem)) return n <= 0 ? DJB_OK : djbcpu::query(djbcpu::twin_ctx(), tw, which, n, a, bb ? bb : a, c ? c : a, params, out);
	st = check_call(ctx, b, n, mem);
	if (st != DJB_OK) return st;
	std::lock_guard<std::recursive_mutex> call_lock(ctx->call_mu);
	Params p;
	if ((st = device_params(params, &p, b->dev.kind)) != DJB_OK) return st;
	Staged sg(ctx, n, mem);
	View va, vb, vc, vo;
	if ((st = sg.in_vec(a, &va)) != DJB_OK) return st;
	vb = va; vc = va;
	if (bb && (st = sg.in_vec(bb, &vb)) != DJB_OK) return st;
	if (c && (st = sg.in_vec(c, &vc)) != DJB_OK) return st;
	if ((st = sg.out_vec(out, &vo)) != DJB_OK) return st;
	HIP_TRY(djbk::launch_query(ctx->stream, b->dev, p, which, n, va, vb, vc, vo));
	return sg.finish();
}
DJB_ABI_CATCH

djb_status djb_merl_index_batch(djb_ctx *ctx, int64_t n, const djb_vec3_view *i, const djb_vec3_view *o,
                                int32_t *out_index, int mem)
try {
	if (!ctx) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null ctx");
	if (is_cpu(ctx)) return djbcpu::merl_index(ctx, n, i, o, out_index);
	if (mem == DJB_MEM_HOST && n >= 0 && n <= SCALAR_HOST_MAX && !ctx->scalar_on_device) return djbcpu::merl_index(djbcpu::twin_ctx(), n, i, o, out_index);
	djb_status st = check_call(ctx, nullptr, n, mem);
	if (st != DJB_OK) return st;
	std::lock_guard<std::recursive_mutex> call_lock(ctx->call_mu);
	Staged sg(ctx, n, mem);
	View vi, vo; int32_t *didx;
	if ((st = sg.in_vec(i, &vi)) != DJB_OK) return st;
	if ((st = sg.in_vec(o, &vo)) != DJB_OK) return st;
	if ((st = sg.out_arr(out_index, &didx)) != DJB_OK) return st;
	HIP_TRY(djbk::launch_merl_index(ctx->stream, n, vi, vo, didx));
	return sg.finish();
}
DJB_ABI_CATCH

// ---------------------------------------------------------------- beckmann::lrep (host scalars)
// dj_brdf.h:1959-2051, float arithmetic in the reference's order (this TU is built with
// -ffp-contract=off).  lrep = {E1, E2, E3, E4, E5}.
djb_status djb_lrep_op(int op, const float *a, const float *b, float x, float y, float *out)
try {
	if (!a || !out) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null argument");
	float E1 = a[0], E2 = a[1], E3 = a[2], E4 = a[3], E5 = a[4];
	const float dflt[5] = { 0, 0, 1, 1, 0 };
	const float *r = b ? b : dflt;
	switch (op) {
	case DJB_LREP_ADD:                                                  // operator+, :1992-1999
		out[0] = E1 + r[0]; out[1] = E2 + r[1];
		out[2] = E3 + r[2] + 2.0f * E1 * r[0];
		out[3] = E4 + r[3] + 2.0f * E2 * r[1];
		out[4] = E5 + r[4] + E1 * r[1] + E2 * r[0];
		return DJB_OK;
	case DJB_LREP_MUL: case DJB_LREP_IMUL: {                            // operator*, *=, :2001-2033
		if (!(x >= 0.0f)) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: Invalid scale");
		float s2 = x * x;
		out[0] = E1 * x; out[1] = E2 * x; out[2] = E3 * s2; out[3] = E4 * s2; out[4] = E5 * s2;
		return DJB_OK;
	}
	case DJB_LREP_IADD:                                                 // operator+=, :2011-2020 (uses the
		E1 += r[0]; E2 += r[1];                                         //  already-updated E1/E2: kept)
		E3 += r[2] + 2.0f * E1 * r[0];
		E4 += r[3] + 2.0f * E2 * r[1];
		E5 += r[4] + E1 * r[1] + E2 * r[0];
		break;
	case DJB_LREP_SHEAR:                                                // :2035-2042
		E1 += x; E2 += y; E3 += x * x; E4 += y * y; E5 += x * y;
		break;
	case DJB_LREP_SCALE:                                                // :2044-2051
		E1 *= x; E2 *= y; E3 *= x * x; E4 *= y * y; E5 *= x * y;
		break;
	default:
		return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: unknown lrep op %d", op);
	}
	out[0] = E1; out[1] = E2; out[2] = E3; out[3] = E4; out[4] = E5;
	return DJB_OK;
}
DJB_ABI_CATCH

djb_status djb_params_to_lrep(const djb_params *params, float *out)                      // :1965-1974
try {
	if (!out) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null argument");
	djb_params_resolved r;
	djb_status st = resolve_params(params, &r);
	if (st != DJB_OK) return st;
	out[0] = r.tx_n; out[1] = r.ty_n;
	out[2] = 0.5f * r.ax * r.ax + r.tx_n * r.tx_n;
	out[3] = 0.5f * r.ay * r.ay + r.ty_n * r.ty_n;
	out[4] = 0.5f * r.rho * r.ax * r.ay + r.tx_n * r.ty_n;
	return DJB_OK;
}
DJB_ABI_CATCH

djb_status djb_lrep_to_params(const float *l, djb_params *out)                           // :1976-1990
try {
	if (!l || !out) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null argument");
	float t1 = l[2] - l[0] * l[0], t2 = l[3] - l[1] * l[1];
	t1 = t1 > 0.0f ? t1 : 0.0f; t2 = t2 > 0.0f ? t2 : 0.0f;
	double sx = std::sqrt(2.0 * (double)t1), sy = std::sqrt(2.0 * (double)t2);
	float ax = (float)(sx > 1e-5 ? sx : 1e-5), ay = (float)(sy > 1e-5 ? sy : 1e-5);
	float rho = 2.0f * (l[4] - l[0] * l[1]) / (ax * ay);
	rho = rho > -0.99f ? rho : -0.99f; rho = rho < 0.99f ? rho : 0.99f;
	out->kind = DJB_PARAMS_PDFPARAMS;
	out->v[0] = ax; out->v[1] = ay; out->v[2] = rho; out->v[3] = l[0]; out->v[4] = l[1];
	return DJB_OK;
}
DJB_ABI_CATCH

static djb_status eval_pp_common(djb_ctx *ctx, const djb_brdf *b, int64_t n, const djb_vec3_view *i,
                                 const djb_vec3_view *o, const float *rec, int mode, const float *base5,
                                 int want, const djb_vec3_view *out_fr, float *out_pdf, float *out_pp, int mem)
{
	if (!b || !rec || !ctx) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null argument");
	const int bkind = djb_brdf_kind(b);
	if (bkind > DJB_KIND_TABULAR && bkind != DJB_KIND_TABULAR_ANISO)
		return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: per-pair params need a microfacet brdf");
	if (want != 1 && want != 2 && want != 4 && want != 5 && want != 6)
		return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: want must be eval(1)|evalp(2) and/or pdf(4)");
	djb_status st = cpu_pair_check(ctx, b);
	if (st != DJB_OK) return st;
	if (is_cpu(ctx)) return n <= 0 ? DJB_OK : djbcpu::eval_pp(ctx, b, n, i, o, rec, mode, base5, want, out_fr, out_pdf, out_pp);
	if (const djb_brdf *tw = scalar_twin(ctx, b, n, mem)) return n <= 0 ? DJB_OK : djbcpu::eval_pp(djbcpu::twin_ctx(), tw, n, i, o, rec, mode, base5, want, out_fr, out_pdf, out_pp);
	st = check_call(ctx, b, n, mem);
	if (st != DJB_OK) return st;
	std::lock_guard<std::recursive_mutex> call_lock(ctx->call_mu);
	if (mem == DJB_MEM_HOST && n > SMALL_N && i && o && (!(want & 3) || out_fr)) {   // large host batch: chunked, both PCIe directions busy
		bool taken = false;
		const bool wfr = (want & 3) != 0, wpdf = (want & 4) != 0;
		std::vector<PipeArr> ins{ PipeArr::vec(i), PipeArr::vec(o), PipeArr::arr(rec, 5) }, outs;
		int kf = -1, kp = -1, kq = -1;
		if (wfr) { kf = (int)outs.size(); outs.push_back(PipeArr::vec(out_fr)); }
		if (wpdf) { kp = (int)outs.size(); outs.push_back(PipeArr::arr(out_pdf)); }
		if (out_pp) { kq = (int)outs.size(); outs.push_back(PipeArr::arr(out_pp, 5)); }
		st = host_pipeline(ctx, n, ins, outs, [&](long long m, int s) {
			djb_vec3_view dvi = ins[0].view(s), dvo = ins[1].view(s), dvf = wfr ? outs[kf].view(s) : djb_vec3_view{ nullptr, nullptr, nullptr, 0 };
			return eval_pp_common(ctx, b, m, &dvi, &dvo, ins[2].dev[s], mode, base5, want, wfr ? &dvf : nullptr,
			                      wpdf ? outs[kp].dev[s] : nullptr, out_pp ? outs[kq].dev[s] : nullptr, DJB_MEM_DEVICE);
		}, &taken);
		if (taken || st != DJB_OK) return st;
	}
	Staged sg(ctx, n, mem);
	View vi, vo, vout{ nullptr, nullptr, nullptr, 0 };
	float *dpdf = nullptr, *dpp = nullptr; const float *drec = rec;
	if ((st = sg.in_vec(i, &vi)) != DJB_OK) return st;
	if ((st = sg.in_vec(o, &vo)) != DJB_OK) return st;
	if (mem == DJB_MEM_HOST) {
		float *d = nullptr;
		if ((st = sg.alloc(sizeof(float) * 5 * (size_t)n, (void **)&d)) != DJB_OK) return st;
		if (n && (st = sg.copy(d, rec, sizeof(float) * 5 * (size_t)n, hipMemcpyHostToDevice)) != DJB_OK) return st;
		drec = d;
	}
	if ((want & 3) && (st = sg.out_vec(out_fr, &vout)) != DJB_OK) return st;
	if ((want & 4) && (st = sg.out_arr(out_pdf, &dpdf)) != DJB_OK) return st;
	if (out_pp) {
		if (mem == DJB_MEM_DEVICE) dpp = out_pp;
		else {
			if ((st = sg.alloc(sizeof(float) * 5 * (size_t)n, (void **)&dpp)) != DJB_OK) return st;
			sg.out_raw.push_back({ dpp, { out_pp, sizeof(float) * 5 * (size_t)n } });
		}
	}
	HIP_TRY(djbk::launch_eval_pp(ctx->stream, b->dev, n, vi, vo, drec, mode, base5, vout, dpdf, dpp, want));
	return sg.finish();
}

djb_status djb_eval_pp_batch(djb_ctx *ctx, const djb_brdf *b, int64_t n, const djb_vec3_view *i,
                             const djb_vec3_view *o, const float *pdfparams, int want,
                             const djb_vec3_view *out_fr, float *out_pdf, int mem)
try {
	return eval_pp_common(ctx, b, n, i, o, pdfparams, 0, nullptr, want, out_fr, out_pdf, nullptr, mem);
}
DJB_ABI_CATCH

djb_status djb_eval_lean_batch(djb_ctx *ctx, const djb_brdf *b, int64_t n, const djb_vec3_view *i,
                               const djb_vec3_view *o, const djb_params *base, float scale, const float *lean,
                               int want, const djb_vec3_view *out_fr, float *out_pdf, float *out_pdfparams, int mem)
try {
	float l1[5], base5[5];
	djb_status st = djb_params_to_lrep(base, l1);
	if (st != DJB_OK) return st;
	if ((st = djb_lrep_op(DJB_LREP_IMUL, l1, nullptr, scale, 0.0f, base5)) != DJB_OK) return st;
	return eval_pp_common(ctx, b, n, i, o, lean, 1, base5, want, out_fr, out_pdf, out_pdfparams, mem);
}
DJB_ABI_CATCH

djb_status djb_ctx_set_option(djb_ctx *ctx, int option, int value)
try {
	if (is_cpu(ctx)) return DJB_OK;           // the options select GPU code paths
	if (ctx && option == DJB_OPT_SCALAR_ON_DEVICE) { ctx->scalar_on_device = value != 0; return DJB_OK; }
	if (ctx && option == DJB_OPT_FIT_FILES_DENSE) { ctx->fit_files_dense = value != 0; return DJB_OK; }
	if (!ctx) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null ctx");
	if (option == DJB_OPT_MERL_EXACT_ONLY) { ctx->merl_exact_only = value != 0; return DJB_OK; }
	if (option == DJB_OPT_ANISO_QF2_ALIGNED) { ctx->aniso_qf2_aligned = value != 0; return DJB_OK; }
	if (option == DJB_OPT_UTIA_EXACT_ONLY) { ctx->utia_exact_only = value != 0; return DJB_OK; }
	if (option == DJB_OPT_CONTRACT_1E5) { ctx->contract_1e5 = value != 0; return DJB_OK; }
	return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: unknown option %d", option);
}
DJB_ABI_CATCH

djb_status djb_merl_guard_stats(djb_ctx *ctx, int64_t n, const djb_vec3_view *i, const djb_vec3_view *o,
                                const float *guard6, float *max_ratio3, unsigned long long *counters4)
try {
	if (is_cpu(ctx)) return fail(DJB_ERR_NOT_IMPLEMENTED, "djb_error: this diagnostic needs a GPU context");
	djb_status st = check_call(ctx, nullptr, n, DJB_MEM_DEVICE);
	if (st != DJB_OK) return st;
	std::lock_guard<std::recursive_mutex> call_lock(ctx->call_mu);
	if (!Staged::valid(i) || !Staged::valid(o) || !max_ratio3 || !counters4)
		return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null argument");
	unsigned char *d = nullptr;
	HIP_TRY(hipMalloc((void **)&d, 64));
	hipError_t e = hipMemsetAsync(d, 0, 64, ctx->stream);
	if (e == hipSuccess)
		e = djbk::launch_merl_guard_stats(ctx->stream, n, View{ i->x, i->y, i->z, (long long)i->stride },
		                                  View{ o->x, o->y, o->z, (long long)o->stride }, guard6,
		                                  (unsigned int *)d, (unsigned long long *)(d + 16));
	unsigned char h[64];
	if (e == hipSuccess) e = hipMemcpyAsync(h, d, 64, hipMemcpyDeviceToHost, ctx->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
	(void)hipFree(d);
	if (e != hipSuccess) return fail(DJB_ERR_HIP, "djb_error: guard stats: %s", hipGetErrorString(e));
	memcpy(max_ratio3, h, 12);
	memcpy(counters4, h + 16, 32);
	return DJB_OK;
}
DJB_ABI_CATCH

djb_status djb_merl_guard_attack(djb_ctx *ctx, int64_t n, const djb_vec3_view *i, const djb_vec3_view *o, const float *guard6,
                                 int iters, uint32_t seed, float *best_ratio, unsigned long long *counters3)
try {
	if (is_cpu(ctx)) return fail(DJB_ERR_NOT_IMPLEMENTED, "djb_error: this diagnostic needs a GPU context");
	djb_status st = check_call(ctx, nullptr, n, DJB_MEM_DEVICE);
	if (st != DJB_OK) return st;
	std::lock_guard<std::recursive_mutex> call_lock(ctx->call_mu);
	if (!Staged::valid(i) || !Staged::valid(o) || !best_ratio || !counters3 || iters < 0)
		return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: invalid argument");
	unsigned long long *d = nullptr;
	HIP_TRY(hipMalloc((void **)&d, 32));
	hipError_t e = hipMemsetAsync(d, 0, 32, ctx->stream);
	if (e == hipSuccess)
		e = djbk::launch_merl_guard_attack(ctx->stream, n, View{ i->x, i->y, i->z, (long long)i->stride },
		                                   View{ o->x, o->y, o->z, (long long)o->stride }, guard6, iters, seed, best_ratio, d);
	if (e == hipSuccess) e = hipMemcpyAsync(counters3, d, 24, hipMemcpyDeviceToHost, ctx->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
	(void)hipFree(d);
	if (e != hipSuccess) return fail(DJB_ERR_HIP, "djb_error: guard attack: %s", hipGetErrorString(e));
	return DJB_OK;
}
DJB_ABI_CATCH

djb_status djb_params_resolve(const djb_params *params, djb_params_resolved *out)
try {
	if (!out) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null argument");
	return resolve_params(params, out);
}
DJB_ABI_CATCH

// ---------------------------------------------------------------- synthetic workloads
djb_status djb_gen_directions(djb_ctx *ctx, int64_t n, uint32_t seed, uint64_t start, const djb_vec3_view *out)
try {
	if (is_cpu(ctx) && out) return n <= 0 ? DJB_OK : djbcpu::gen_directions(ctx, n, seed, start, out);
	djb_status st = check_call(ctx, nullptr, n, DJB_MEM_DEVICE);
	if (st != DJB_OK) return st;
	std::lock_guard<std::recursive_mutex> call_lock(ctx->call_mu);
	if (!Staged::valid(out)) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null vec3 view");
	HIP_TRY(djbk::launch_gen_directions(ctx->stream, n, seed, start, View{ out->x, out->y, out->z, (long long)out->stride }));
	return DJB_OK;
}
DJB_ABI_CATCH
djb_status djb_gen_uniforms(djb_ctx *ctx, int64_t n, uint32_t seed, uint64_t start, float *out)
try {
	if (is_cpu(ctx)) return n <= 0 ? DJB_OK : djbcpu::gen_uniforms(ctx, n, seed, start, out);
	djb_status st = check_call(ctx, nullptr, n, DJB_MEM_DEVICE);
	if (st != DJB_OK) return st;
	std::lock_guard<std::recursive_mutex> call_lock(ctx->call_mu);
	if (!out) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null output array");
	HIP_TRY(djbk::launch_gen_uniforms(ctx->stream, n, seed, start, out));
	return DJB_OK;
}
DJB_ABI_CATCH
djb_status djb_selftest_guarded_math(djb_ctx *ctx, int64_t n, uint32_t seed, unsigned long long *counters10)
try {
	if (is_cpu(ctx)) return fail(DJB_ERR_NOT_IMPLEMENTED, "djb_error: this diagnostic needs a GPU context");
	djb_status st = check_call(ctx, nullptr, n, DJB_MEM_DEVICE);
	if (st != DJB_OK) return st;
	std::lock_guard<std::recursive_mutex> call_lock(ctx->call_mu);
	if (!counters10) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null argument");
	unsigned long long *d = nullptr;
	HIP_TRY(hipMalloc((void **)&d, 80));
	hipError_t e = hipMemsetAsync(d, 0, 80, ctx->stream);
	if (e == hipSuccess) e = djbk::launch_guard_selftest(ctx->stream, n, seed, d);
	if (e == hipSuccess) e = hipMemcpyAsync(counters10, d, 80, hipMemcpyDeviceToHost, ctx->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
	(void)hipFree(d);
	if (e != hipSuccess) return fail(DJB_ERR_HIP, "djb_error: selftest: %s", hipGetErrorString(e));
	return DJB_OK;
}
DJB_ABI_CATCH

djb_status djb_selftest_contract(djb_ctx *ctx, const djb_brdf *b, const djb_params *params, int64_t n, uint32_t seed, int family,
                                 float *max_rel2, unsigned long long *counters4)
try {
	if (is_cpu(ctx)) return fail(DJB_ERR_NOT_IMPLEMENTED, "djb_error: this diagnostic needs a GPU context");
	if (!b) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null brdf");
	djb_status st = check_call(ctx, b, n, DJB_MEM_DEVICE);
	if (st != DJB_OK) return st;
	std::lock_guard<std::recursive_mutex> call_lock(ctx->call_mu);
	if (!max_rel2 || !counters4) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null argument");
	Params p;
	if ((st = device_params(params, &p, b->dev.kind)) != DJB_OK) return st;
	if (b->dev.kind != DJB_KIND_GGX || !djbk::contract_supported(b->dev, p))
		return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: brdf / params outside the domain of the contract-mode fast path");
	unsigned char *d = nullptr;
	HIP_TRY(hipMalloc((void **)&d, 64));
	hipError_t e = hipMemsetAsync(d, 0, 64, ctx->stream);
	if (e == hipSuccess)
		e = djbk::launch_contract_selftest(ctx->stream, b->dev, p, n, seed, seed ^ 0x9e3779b9u, 0ull, family,
		                                   (unsigned int *)d, (unsigned long long *)(d + 16));
	unsigned char h[64];
	if (e == hipSuccess) e = hipMemcpyAsync(h, d, 64, hipMemcpyDeviceToHost, ctx->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
	(void)hipFree(d);
	if (e != hipSuccess) return fail(DJB_ERR_HIP, "djb_error: contract selftest: %s", hipGetErrorString(e));
	memcpy(max_rel2, h, 8);
	memcpy(counters4, h + 16, 32);
	return DJB_OK;
}
DJB_ABI_CATCH

djb_status djb_selftest_libm(djb_ctx *ctx, int fn, int64_t n, const double *x, const double *y, double *out)
try {
	if (is_cpu(ctx)) return fail(DJB_ERR_NOT_IMPLEMENTED, "djb_error: this diagnostic needs a GPU context");
	djb_status st = check_call(ctx, nullptr, n, DJB_MEM_HOST);
	if (st != DJB_OK) return st;
	std::lock_guard<std::recursive_mutex> call_lock(ctx->call_mu);
	if (fn < 0 || fn > 11 || !x || !y || !out) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: invalid selftest arguments");
	if (n == 0) return DJB_OK;
	const size_t nb = sizeof(double) * (size_t)n;
	double *d = nullptr;
	HIP_TRY(hipMalloc((void **)&d, 3 * nb));
	hipError_t e = hipMemcpy(d, x, nb, hipMemcpyHostToDevice);
	if (e == hipSuccess) e = hipMemcpy(d + n, y, nb, hipMemcpyHostToDevice);
	if (e == hipSuccess) e = djbk::launch_libm_probe(ctx->stream, fn, n, d, d + n, d + 2 * n);
	if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
	if (e == hipSuccess) e = hipMemcpy(out, d + 2 * n, nb, hipMemcpyDeviceToHost);
	(void)hipFree(d);
	if (e != hipSuccess) return fail(DJB_ERR_HIP, "djb_error: selftest: %s", hipGetErrorString(e));
	return DJB_OK;
}
DJB_ABI_CATCH

static_assert(djbdev::TRIG_SITES == DJB_TRIG_SITES && djbdev::TRIG_DOUBLE == DJB_TRIG_DOUBLE && djbdev::TRIG_DOUBLE_SITES == DJB_TRIG_DOUBLE_SITES, "djb_hip.h and djb_device.hpp number the trig sites differently");
static bool trig_site_valid(int fn)
{
	return (fn >= 0 && fn < DJB_TRIG_SITES) || (fn >= DJB_TRIG_DOUBLE && fn < DJB_TRIG_DOUBLE + DJB_TRIG_DOUBLE_SITES);
}
djb_status djb_selftest_trig_sweep(djb_ctx *ctx, int fn, int host_fn, uint32_t first_bits, int64_t count, int threads,
                                   unsigned long long *n_bad, uint32_t *bad3, int cap)
try {
	if (is_cpu(ctx)) return fail(DJB_ERR_NOT_IMPLEMENTED, "djb_error: this diagnostic needs a GPU context");
	djb_status st = check_call(ctx, nullptr, count, DJB_MEM_HOST);
	if (st != DJB_OK) return st;
	if (!trig_site_valid(fn) || !trig_site_valid(host_fn) || (fn >= DJB_TRIG_DOUBLE) != (host_fn >= DJB_TRIG_DOUBLE) || !n_bad ||
	    cap < 0 || (cap > 0 && !bad3) || count > ((int64_t)1 << 32) - (int64_t)first_bits)
		return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: invalid selftest arguments");
	*n_bad = 0;
	if (count == 0) return DJB_OK;
	const size_t nb = (fn >= DJB_TRIG_DOUBLE ? sizeof(double) : sizeof(float)) * (size_t)count;
	std::vector<char> host(nb);
	{
		std::lock_guard<std::recursive_mutex> call_lock(ctx->call_mu);
		void *d = nullptr;
		HIP_TRY(hipMalloc(&d, nb));
		hipError_t e = djbk::launch_trig_sweep(ctx->stream, fn, first_bits, count, d);
		if (e == hipSuccess) e = hipMemcpyAsync(host.data(), d, nb, hipMemcpyDeviceToHost, ctx->stream);
		if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
		(void)hipFree(d);
		if (e != hipSuccess) return fail(DJB_ERR_HIP, "djb_error: selftest: %s", hipGetErrorString(e));
	}
	*n_bad = djbcpu::trig_sweep_compare(host_fn, first_bits, count, host.data(), threads, bad3, cap);
	return DJB_OK;
}
DJB_ABI_CATCH

djb_status djb_histogram_xy(djb_ctx *ctx, int64_t n, const djb_vec3_view *v, int bins, unsigned long long *counts)
try {
	if (is_cpu(ctx)) return djbcpu::histogram_xy(ctx, n, v, bins, counts);
	djb_status st = check_call(ctx, nullptr, n, DJB_MEM_DEVICE);
	if (st != DJB_OK) return st;
	std::lock_guard<std::recursive_mutex> call_lock(ctx->call_mu);
	if (!Staged::valid(v) || !counts || bins < 1 || bins > 128)
		return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: invalid histogram arguments");
	HIP_TRY(djbk::launch_histogram_xy(ctx->stream, n, View{ v->x, v->y, v->z, (long long)v->stride }, bins, counts));
	return DJB_OK;
}
DJB_ABI_CATCH

} // extern "C"

// ---------------------------------------------------------------- hooks for djb_loader.hip
namespace djbk {

djb_status resolve_device_params(const djb_params *in, float out9[9], int brdf_kind)
{
	Params p;
	djb_status st = device_params(in, &p, brdf_kind);
	if (st != DJB_OK) return st;
	out9[0] = p.nx; out9[1] = p.ny; out9[2] = p.nz; out9[3] = p.ax; out9[4] = p.ay; out9[5] = p.rho; out9[6] = p.s; out9[7] = p.tx; out9[8] = p.ty;
	// (the host path divides: r_ax / r_t2 stay 0 there)
	return DJB_OK;
}

hipStream_t ctx_stream(djb_ctx *ctx) { return ctx->stream; }
int ctx_device(djb_ctx *ctx) { return ctx->device; }
void ctx_lock(djb_ctx *ctx) { ctx->call_mu.lock(); }
void ctx_unlock(djb_ctx *ctx) { ctx->call_mu.unlock(); }

djb_status set_error(djb_status st, const char *fmt, ...)
{
	char buf[256];
	va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
	g_err = buf;
	return st;
}

int ctx_option_fit_files_dense(djb_ctx *ctx) { return ctx->fit_files_dense; }

// per-slot texels of the file-fit pipeline (see djb_loader.hip): a source for djb_fit_brdf_batch only
djb_status wrap_merl_slots(djb_ctx *ctx, djbdev::MerlTexel *slots, djb_brdf **out)
{
	djb_brdf *b;
	alloc_brdf(ctx, DJB_KIND_MERL, &b);
	b->dev.merl = slots;
	b->dev.merl_sparse = 1;
	*out = b;
	return DJB_OK;
}

// a texel table already converted in HBM becomes a djb::merl object (which owns it if `own`)
djb_status wrap_merl_table(djb_ctx *ctx, djbdev::MerlTexel *table, djb_brdf **out, bool own)
{
	djb_brdf *b;
	alloc_brdf(ctx, DJB_KIND_MERL, &b);
	if (own) b->allocs.push_back(table);
	b->dev.merl = table;
	*out = b;
	return DJB_OK;
}

} // namespace djbk
