// djb_cpu.cpp -- the product's host execution path (see djb_cpu.hpp): the per-unit code of djb_device.hpp compiled
// for the CPU (DJB_HOST_MATH: every libm call is the host's glibc, i.e. what the reference itself calls; no FMA
// contraction: -ffp-contract=off, as for the kernels), driven by plain loops over the caller's arrays, chunked over
// std::threads for large batches.  The fitters restate djb_kernels_fit.hip / djb_kernels_fit_aniso.hip phase by
// phase with the sums in the reference's order (dj_brdf.h:2215-2762, 3133-3184).
#define DJB_HOST_MATH 1
#include "djb_device.hpp"
#include "djb_cpu.hpp"
#include "djb_merl_file.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <exception>
#include <mutex>
#include <thread>
#include <vector>

using namespace djbdev;

namespace {

constexpr long long MERL_N = 90LL * 90 * 180;
constexpr long long UTIA_N = 3LL * 288 * 288;

struct CpuCtx {
	int device = -1;          // MUST stay the first member (djbcpu::is_cpu)
	int threads = 1;
	std::chrono::steady_clock::time_point t0;
};

struct CpuBrdf {
	int device = -1;          // MUST stay the first member (djbcpu::is_cpu)
	CpuCtx *ctx;
	Brdf dev;                 // the same view the kernels take; every pointer is host memory owned below
	std::vector<float> p22, sigma, cdf, qf, fresnel;
	float alpha_beckmann = 0.0f, alpha_ggx = 0.0f;
	std::vector<float> aniso[8];
	float aniso_fit[10] = { 0 };
	int elev = 0, azim = 0, aniso_qf2_entries = 0;
	std::vector<MerlTexel> merl;
	std::vector<float4> utia;
	std::vector<double> model, raw;
	UserNdf ndf = {};         // KIND_USER: the caller's callbacks (dev.user_ndf points here)
};

CpuCtx *C(djb_ctx *c) { return (CpuCtx *)c; }
const CpuBrdf *B(const djb_brdf *b) { return (const CpuBrdf *)b; }
CpuBrdf *B(djb_brdf *b) { return (CpuBrdf *)b; }

View view_of(const djb_vec3_view *v) { return View{ v->x, v->y, v->z, (long long)v->stride }; }
bool valid(const djb_vec3_view *v) { return v && v->x && v->y && v->z; }

// contiguous chunks of [0, n) over up to ctx->threads threads; small batches run on the calling thread
template <class F> void parallel_for(const CpuCtx *ctx, long long n, long long grain, F f)
{
	int t = ctx ? ctx->threads : 1;
	if ((long long)t > n / grain) t = (int)(n / grain);
	if (t <= 1) { f(0LL, n); return; }
	// No exception may leave a worker thread (std::terminate) and none may leave this function before every thread has been
	// joined: a worker's exception is carried to the caller's thread and rethrown there (the C ABI maps it to a status);
	// if a thread cannot be created, its slice runs on the calling thread.
	std::vector<std::thread> th;
	th.reserve(t - 1);
	std::exception_ptr first;
	std::mutex mu;
	auto guarded = [&](long long k0, long long k1) {
		try { f(k0, k1); }
		catch (...) { std::lock_guard<std::mutex> lk(mu); if (!first) first = std::current_exception(); }
	};
	std::vector<int> inline_slices;
	for (int k = 1; k < t; ++k) {
		try { th.emplace_back([&guarded, n, k, t] { guarded(n * k / t, n * (k + 1) / t); }); }
		catch (...) { inline_slices.push_back(k); }
	}
	guarded(0LL, n / t);
	for (int k : inline_slices) guarded(n * k / t, n * (k + 1) / t);
	for (auto &x : th) x.join();
	if (first) std::rethrow_exception(first);
}

CpuBrdf *alloc_brdf(djb_ctx *ctx, int kind)
{
	CpuBrdf *b = new CpuBrdf();
	b->ctx = C(ctx);
	memset(&b->dev, 0, sizeof b->dev);
	b->dev.kind = kind;
	b->dev.shadow = 1;
	b->dev.fr.kind = FR_IDEAL;
	return b;
}

djb_status apply_fresnel(CpuBrdf *b, const djb_fresnel_desc *f)
{
	Fresnel &fr = b->dev.fr;
	fr.kind = f ? f->kind : DJB_FRESNEL_IDEAL;
	fr.pts = nullptr; fr.npts = 0;
	if (!f) return DJB_OK;
	if (f->kind < DJB_FRESNEL_IDEAL || f->kind > DJB_FRESNEL_SPLINE)
		return djbk::set_error(DJB_ERR_INVALID_ARGUMENT, "djb_error: unknown fresnel kind %d", f->kind);
	for (int c = 0; c < 3; ++c) { fr.a[c] = f->a[c]; fr.b[c] = f->b[c]; }
	if (f->kind == DJB_FRESNEL_SPLINE) {
		if (!f->points || f->npoints < 1)
			return djbk::set_error(DJB_ERR_INVALID_ARGUMENT, "djb_error: fresnel::spline needs >= 1 point");
		b->fresnel.assign(f->points, f->points + 3 * (size_t)f->npoints);
		fr.npts = f->npoints; fr.pts = b->fresnel.data();
	}
	return DJB_OK;
}

djb_status params_for(const djb_params *in, int brdf_kind, Params *p)
{
	// a scalar call costs ~100 ns: do not re-derive params::standard() (cos / sin / sqrt chain, what the reference does
	// on every call with user_param == NULL, dj_brdf.h:1532-1534) each time, and not at all for the kinds that ignore it
	const bool uses_params = DJB_IS_MICROFACET(brdf_kind) || brdf_kind == KIND_LAMBERT;
	if (!uses_params && (!in || DJB_PARAMS_KIND(in->kind) != DJB_PARAMS_LAMBERT)) { memset(p, 0, sizeof *p); return DJB_OK; }
	if (!in && brdf_kind != KIND_LAMBERT) {
		static const Params std_p = [] {
			float v[9]; Params q; memset(&q, 0, sizeof q);
			if (djbk::resolve_device_params(nullptr, v, -1) == DJB_OK) { q.nx = v[0]; q.ny = v[1]; q.nz = v[2]; q.ax = v[3]; q.ay = v[4]; q.rho = v[5]; q.s = v[6]; q.tx = v[7]; q.ty = v[8]; }
			return q;
		}();
		*p = std_p;
		return DJB_OK;
	}
	if (in && brdf_kind != KIND_LAMBERT && (in->kind & DJB_PARAMS_RESOLVED_FOLLOWS) && DJB_PARAMS_KIND(in->kind) != DJB_PARAMS_LAMBERT) {   // (in == NULL reaches here for a lambert)
		// a parameter set that carries its resolved form (include/djb_hip.h: djb_params_cached; the facade's params objects): read it
		const djb_params_resolved &r = reinterpret_cast<const djb_params_cached *>(in)->r;
		if (!(r.ax > 0.0f && r.ay > 0.0f && r.rho > -1.0f && r.rho < 1.0f))     // a stray flag on a plain djb_params: refuse, do not read garbage as parameters
			return djbk::set_error(DJB_ERR_INVALID_ARGUMENT, "djb_error: DJB_PARAMS_RESOLVED_FOLLOWS is set but no resolved parameter set follows the djb_params");
		p->nx = r.n[0]; p->ny = r.n[1]; p->nz = r.n[2]; p->ax = r.ax; p->ay = r.ay; p->rho = r.rho; p->s = r.sqrt_one_minus_rho_sqr;
		p->tx = r.tx_n; p->ty = r.ty_n; p->r_ax = 0.0; p->r_t2 = 0.0;
		return DJB_OK;
	}
	float v[9];
	djb_status st = djbk::resolve_device_params(in, v, brdf_kind);
	if (st != DJB_OK) return st;
	p->nx = v[0]; p->ny = v[1]; p->nz = v[2]; p->ax = v[3]; p->ay = v[4]; p->rho = v[5]; p->s = v[6]; p->tx = v[7]; p->ty = v[8];
	p->r_ax = 0.0; p->r_t2 = 0.0;
	return DJB_OK;
}

// ------------------------------------------------------------------ loops
template <int KIND, int WANT>
void eval_loop(const Brdf &b, const Params &p, long long k0, long long k1, const View &vi, const View &vo, const View &vout, float *out_pdf)
{
	for (long long k = k0; k < k1; ++k) {
		v3 fr = mk(0, 0, 0); float pdf = 0.0f;
		eval_one<KIND, WANT>(b, p, load3(vi, k), load3(vo, k), fr, pdf);
		if (WANT & 3) store3(vout, k, fr);
		if (WANT & 4) out_pdf[k] = pdf;
	}
}
template <int KIND>
void eval_kind(const Brdf &b, const Params &p, long long k0, long long k1, const View &vi, const View &vo, const View &vout, float *out_pdf, int want)
{
	switch (want) {
	case 1: eval_loop<KIND, 1>(b, p, k0, k1, vi, vo, vout, out_pdf); break;
	case 2: eval_loop<KIND, 2>(b, p, k0, k1, vi, vo, vout, out_pdf); break;
	case 4: eval_loop<KIND, 4>(b, p, k0, k1, vi, vo, vout, out_pdf); break;
	case 5: eval_loop<KIND, 5>(b, p, k0, k1, vi, vo, vout, out_pdf); break;
	case 6: eval_loop<KIND, 6>(b, p, k0, k1, vi, vo, vout, out_pdf); break;
	}
}
#define DJB_KIND_SWITCH(kind_, CALL) \
	switch (kind_) { \
	case KIND_BECKMANN: { constexpr int K = KIND_BECKMANN; CALL; } break; \
	case KIND_GGX: { constexpr int K = KIND_GGX; CALL; } break; \
	case KIND_TABULAR: { constexpr int K = KIND_TABULAR; CALL; } break; \
	case KIND_TABULAR_ANISO: { constexpr int K = KIND_TABULAR_ANISO; CALL; } break; \
	case KIND_MERL: { constexpr int K = KIND_MERL; CALL; } break; \
	case KIND_UTIA: { constexpr int K = KIND_UTIA; CALL; } break; \
	case KIND_LAMBERT: { constexpr int K = KIND_LAMBERT; CALL; } break; \
	case KIND_SGD: { constexpr int K = KIND_SGD; CALL; } break; \
	case KIND_ABC: { constexpr int K = KIND_ABC; CALL; } break; \
	case KIND_USER: { constexpr int K = KIND_USER; CALL; } break; \
	}

template <int KIND, bool IS>
void sample_loop(const Brdf &b, const Params &p, long long k0, long long k1, const float *u1a, const float *u2a, uint32_t s1, uint32_t s2,
                 unsigned long long start, const View &vo, const View &vi_out, const View &vw_out, float *out_pdf)
{
	const GlibcTabs gt = glibc_tabs_global();
	for (long long k = k0; k < k1; ++k) {
		float u1 = u1a ? u1a[k] : gen_uniform(s1, start + (unsigned long long)k);
		float u2 = u1a ? u2a[k] : gen_uniform(s2, start + (unsigned long long)k);
		v3 i_out, w; float pdf;
		sample_one<KIND, IS>(b, p, u1, u2, load3(vo, k), gt, i_out, w, pdf);
		store3(vi_out, k, i_out);
		if (IS) { store3(vw_out, k, w); out_pdf[k] = pdf; }
	}
}

template <int KIND, int WANT, int MODE>
void pp_loop(const Brdf &b, long long k0, long long k1, const View &vi, const View &vo, const float *rec, const LeanCfg &base,
             const View &vout, float *out_pdf, float *out_pp)
{
	for (long long k = k0; k < k1; ++k) {
		v3 fr = mk(0, 0, 0); float pdf = 0.0f;
		pp_one<KIND, WANT, MODE>(b, load3(vi, k), load3(vo, k), rec + 5 * k, base, out_pp ? out_pp + 5 * k : nullptr, fr, pdf);
		if (WANT & 3) store3(vout, k, fr);
		if (WANT & 4) out_pdf[k] = pdf;
	}
}
template <int KIND, int MODE>
void pp_kind(const Brdf &b, long long k0, long long k1, const View &vi, const View &vo, const float *rec, const LeanCfg &base,
             const View &vout, float *out_pdf, float *out_pp, int want)
{
	switch (want) {
	case 1: pp_loop<KIND, 1, MODE>(b, k0, k1, vi, vo, rec, base, vout, out_pdf, out_pp); break;
	case 2: pp_loop<KIND, 2, MODE>(b, k0, k1, vi, vo, rec, base, vout, out_pdf, out_pp); break;
	case 4: pp_loop<KIND, 4, MODE>(b, k0, k1, vi, vo, rec, base, vout, out_pdf, out_pp); break;
	case 5: pp_loop<KIND, 5, MODE>(b, k0, k1, vi, vo, rec, base, vout, out_pdf, out_pp); break;
	case 6: pp_loop<KIND, 6, MODE>(b, k0, k1, vi, vo, rec, base, vout, out_pdf, out_pp); break;
	}
}

// brdf.eval of any kind (the fitters' source look-ups); slot: the query slot (a sparse MERL source holds one texel per slot)
v3 src_eval(const Brdf &src, const Params &std_p, v3 i, v3 o, int slot = -1)
{
	if (src.kind == KIND_MERL && src.merl_sparse) { MerlTexel t = src.merl[slot]; return mk(t.x, t.y, t.z); }
	// the fitters call brdf.eval(i, o) with user_param == NULL: a Lambertian source has the default reflectance (1, 1, 1)
	// (dj_brdf.h:863-865) -- std_p are MICROFACET params, whose n = (0, 0, 1) eval_one would read as a reflectance (found in
	// round 5 by examples/c_abi_demo.c: tabular(lambert) on a CPU context fitted a blue-only source; the GPU kernels were right)
	if (src.kind == KIND_LAMBERT) return divs(mk(1, 1, 1), F(DJB_PI));
	v3 fr = mk(0, 0, 0); float pdf = 0.0f;
	DJB_KIND_SWITCH(src.kind, (eval_one<K, 1>(src, std_p, i, o, fr, pdf)))
	return fr;
}

// ------------------------------------------------------------------ djb::tabular(brdf, res, shadow) + the two fits
// Restates k_fit (djb_kernels_fit.hip) on one thread: same per-term expressions, every sum in the reference's order.
struct FitResult { std::vector<float> p22, sigma, cdf, qf, fresnel; int n_qf; float alpha_beckmann, alpha_ggx; };

void fit_tabular(const Brdf &src, const Params &std_p, int res, int shadow, FitResult &R)
{
	const int cnt = res - 1;
	const int NTHETA_SIGMA = 90, NPHI_SIGMA = 180, NNODE = NTHETA_SIGMA * NPHI_SIGMA, NTHETA_FIT = 128, MAX_PHI_STEPS = 512;
	R.p22.assign(res, 0.0f); R.sigma.assign(res, 0.0f); R.cdf.assign(res, 0.0f); R.qf.assign(res, 0.0f); R.fresnel.assign(3 * (size_t)res, 0.0f);
	float *p22 = R.p22.data(), *sigma = R.sigma.data(), *cdf = R.cdf.data(), *qf = R.qf.data(), *fres = R.fresnel.data();
	Brdf self;
	memset(&self, 0, sizeof self);
	self.kind = KIND_TABULAR; self.shadow = shadow; self.fr.kind = FR_IDEAL;
	self.p22 = p22; self.sigma = sigma; self.cdf = cdf; self.qf = qf;
	self.n_p22 = res; self.n_sigma = res; self.n_cdf = res; self.n_qf = res;

	// ---- compute_p22_smith (dj_brdf.h:2482-2522)
	const float dtheta_k = F(sqrt(DJB_PI * 0.5) / D((float)cnt));
	const float dphi_h = F(DJB_PI / 180.0);
	std::vector<float> cphi; cphi.reserve(MAX_PHI_STEPS);
	for (float phi = 0.0f; D(phi) < 2.0 * DJB_PI && (int)cphi.size() < MAX_PHI_STEPS; phi += dphi_h) cphi.push_back(cos_f(phi));   // 361 float-stepped values
	const int nphi = (int)cphi.size();
	std::vector<float> theta(cnt), cosv(cnt), tanv(cnt), kji(cnt);
	std::vector<double> v0(cnt, 1.0), v1(cnt, 0.0), kmT((size_t)cnt * cnt);
	for (int k = 0; k < cnt; ++k) {
		float th = fit_backscatter_theta(k, cnt);
		float th2 = th * th;
		float c = cos_f(th2), t = tan_f(th2);
		theta[k] = th; cosv[k] = c; tanv[k] = t;
		v3 w = from_angles(th2, 0.0f);
		float fr_i = intensity(src_eval(src, std_p, w, w, k));
		kji[k] = F((D(dtheta_k) * glibc_pow(D(c), D(6.0f))) * (8.0 * D(fr_i)));
	}
	for (int io = 0; io < cnt; ++io)
		for (int jh = 0; jh < cnt; ++jh) {
			float tan_product = tanv[jh] * tanv[io];
			float nint = 0.0f;
			for (int q = 0; q < nphi; ++q) nint += fmax_(1.0f, tan_product * cphi[q]);
			nint *= dphi_h;
			float ch = cosv[jh];
			kmT[(size_t)jh * cnt + io] = D(theta[jh] * kji[io] * nint * tanv[jh] / (ch * ch));
		}
	for (int it = 0; it < 4; ++it) {               // matrix::eigenvector(4): un-normalised, sums in index order
		std::vector<double> &vin = (it & 1) ? v1 : v0, &vout = (it & 1) ? v0 : v1;
		for (int j = 0; j < cnt; ++j) {
			double acc = 0.0;
			for (int i = 0; i < cnt; ++i) acc += kmT[(size_t)i * cnt + j] * vin[i];
			vout[j] = acc;
		}
	}
	for (int k = 0; k < res; ++k) p22[k] = k < cnt ? F(1e-2 * v0[k]) : 0.0f;

	// ---- normalize_p22 (dj_brdf.h:2277-2304)
	{
		float nint = 0.0f;
		for (int k = 0; k < NTHETA_FIT; ++k) {
			float u = (float)k / (float)NTHETA_FIT;
			float th = F(D(u * u) * DJB_PI * 0.5);
			float r = tan_f(th), c = cos_f(th);
			float pr = p22_radial<KIND_TABULAR>(self, r * r);
			nint += (u * pr * r) / (c * c);
		}
		nint *= F(DJB_PI / D((float)NTHETA_FIT)) * F(2.0 * DJB_PI);
		const float scale = F(1.0 / D(nint));
		for (int k = 0; k < res; ++k) p22[k] *= scale;
	}

	// ---- compute_sigma (dj_brdf.h:2348-2386): the ndf of a node does not depend on theta_k
	{
		std::vector<double> cphid(NPHI_SIGMA), cthd(NTHETA_SIGMA);
		std::vector<float> sh(NTHETA_SIGMA), ui(NTHETA_SIGMA), ndf_tab(NNODE);
		for (int k = 0; k < NPHI_SIGMA; ++k) cphid[k] = glibc_cos(D(F(D((float)k / (float)NPHI_SIGMA) * 2.0 * DJB_PI)));
		for (int k = 0; k < NTHETA_SIGMA; ++k) {
			float u = (float)k / (float)NTHETA_SIGMA;
			float th = F(D(u * u) * DJB_PI * 0.5);
			ui[k] = u; sh[k] = sin_f(th); cthd[k] = glibc_cos(D(th));
		}
		for (int e = 0; e < NNODE; ++e) {
			int j2 = e / NTHETA_SIGMA, j1 = e - j2 * NTHETA_SIGMA;
			float phi_h = F(D((float)j2 / (float)NPHI_SIGMA) * 2.0 * DJB_PI);
			float u = (float)j1 / (float)NTHETA_SIGMA;
			float th = F(D(u * u) * DJB_PI * 0.5);
			ndf_tab[e] = mf_ndf<KIND_TABULAR>(self, from_angles(th, phi_h), std_p);
		}
		const float dth = F(DJB_PI / D((float)NTHETA_SIGMA));
		const float dph = F(2.0 * DJB_PI / D((float)NPHI_SIGMA));
		for (int k = 0; k < cnt; ++k) {
			float tmp = (float)k / (float)cnt;
			float theta_k = F(D(tmp) * 0.5 * DJB_PI);
			const float ck = cos_f(theta_k), sk = sin_f(theta_k);
			float nint = 0.0f;
			for (int e = 0; e < NNODE; ++e) {
				const int j2 = e / NTHETA_SIGMA, j1 = e - j2 * NTHETA_SIGMA;
				const float s1 = sh[j1];
				float kh = F(D(sk * s1) * cphid[j2] + D(ck) * cthd[j1]);
				nint += fmax_(0.0f, kh) * ndf_tab[e] * ui[j1] * s1;
			}
			nint *= dth * dph;
			sigma[k] = fmax_(ck, nint);
		}
		sigma[cnt] = sigma[cnt - 1];
	}

	// ---- compute_fresnel (dj_brdf.h:2583-2641)
	for (int i = 0; i < cnt; ++i) {
		float fx = 0, fy = 0, fz = 0; int cx = 0, cy = 0, cz = 0;
		for (int j = 0; j <= cnt; ++j) {
			v3 dir_i, dir_o;
			if (!fit_fresnel_dirs(i, j, cnt, dir_i, dir_o)) continue;
			v3 fr1 = src_eval(src, std_p, dir_i, dir_o, cnt + i * (cnt + 1) + j);
			v3 fr2; float pdf;
			mf_eval_pdf<KIND_TABULAR, 1>(self, std_p, dir_i, dir_o, fr2, pdf);
			if (D(fr2.x) > 1e-4) { fx += fr1.x / fr2.x; ++cx; }
			if (D(fr2.y) > 1e-4) { fy += fr1.y / fr2.y; ++cy; }
			if (D(fr2.z) > 1e-4) { fz += fr1.z / fr2.z; ++cz; }
		}
		fres[3 * i] = cx == 0 ? 1.0f : fmin_(1.0f, fx / (float)cx);
		fres[3 * i + 1] = cy == 0 ? 1.0f : fmin_(1.0f, fy / (float)cy);
		fres[3 * i + 2] = cz == 0 ? 1.0f : fmin_(1.0f, fz / (float)cz);
	}
	for (int c = 0; c < 3; ++c) fres[3 * cnt + c] = fres[3 * (cnt - 1) + c];

	// ---- compute_cdf (dj_brdf.h:2705-2727)
	{
		const float dth = F(DJB_PI / D((float)cnt));
		float nint = 0.0f;
		for (int k = 0; k < cnt; ++k) {
			float u = (float)k / (float)cnt;
			float th = F(D(u * u) * DJB_PI * 0.5);
			float c = cos_f(th), r = tan_f(th);
			float pr = p22_radial<KIND_TABULAR>(self, r * r);
			nint += (u * r * pr) / (c * c);
			cdf[k] = F(D(nint * dth) * (2.0 * DJB_PI));
		}
		cdf[cnt] = 1.0f;
	}

	// ---- compute_qf (dj_brdf.h:2731-2762): one forward scan, j persists across i
	{
		const int qres = cnt * 8;
		int nq = 0, j = 0;
		qf[nq++] = 0.0f;
		for (int i = 1; i < cnt; ++i) {
			float c = (float)i / (float)cnt;
			for (; j < qres; ++j) {
				float u = (float)j / (float)qres;
				float th = F(D(u) * DJB_PI * 0.5);
				if (tab_cdf_radial(self, tan_f(th)) >= c) { qf[nq++] = u; break; }
			}
		}
		qf[nq++] = 1.0f;
		R.n_qf = nq;
	}

	// ---- fit_beckmann_parameters / fit_ggx_parameters (dj_brdf.h:3133-3184)
	{
		const float dth = F(DJB_PI / D((float)NTHETA_FIT));
		float nb = 0.0f, ng = 0.0f;
		for (int k = 0; k < NTHETA_FIT; ++k) {
			float u = (float)k / (float)NTHETA_FIT;
			float th = F(D(u * u) * DJB_PI * 0.5);
			float c = cos_f(th), r = tan_f(th);
			float r2 = r * r;
			float pr = p22_radial<KIND_TABULAR>(self, r2);
			nb += (u * r2 * r * pr) / (c * c);
			ng += (u * r2 * pr) / (c * c);
		}
		nb = F(D(nb) * (D(dth) * DJB_PI));
		ng = F(D(ng) * (D(dth) * 4.0));
		R.alpha_beckmann = F(sqrt(2.0 * D(nb)));
		R.alpha_ggx = ng;
	}
}

djb_status read_doubles(const char *path, bool merl_header, std::vector<double> *payload)
{
	FILE *f = fopen(path, "rb");
	if (!f) return djbk::set_error(DJB_ERR_OPEN_FAILED, "djb_error: Failed to open %s\n", path);
	size_t count = (size_t)UTIA_N;
	if (merl_header) {     // dj_brdf.h:973-976; untrusted: 64-bit product, MERL shape only (see djb_host.hip read_file)
		int32_t dims[3] = { 0, 0, 0 };
		size_t got = fread(dims, 4, 3, f);
		const bool positive = got == 3 && dims[0] > 0 && dims[1] > 0 && dims[2] > 0;
		const long long n = positive ? (long long)dims[0] * (long long)dims[1] * (long long)dims[2] : 0;
		if (n <= 0) { fclose(f); return djbk::set_error(DJB_ERR_BAD_HEADER, "djb_error: Failed to read MERL header\n"); }
		if (n != MERL_N) { fclose(f); return djbk::set_error(DJB_ERR_BAD_HEADER, "djb_error: MERL table has %lld samples per channel, expected %lld\n", n, MERL_N); }
		count = 3 * (size_t)n;
	}
	payload->resize(count);
	size_t got = fread(payload->data(), sizeof(double), count, f);
	fclose(f);
	if (got != count) return djbk::set_error(DJB_ERR_READ_FAILED, "djb_error: Reading %s failed\n", path);
	return DJB_OK;
}

void finish_tabular(CpuBrdf *t, int res, int shadow)
{
	Brdf &d = t->dev;
	d.shadow = shadow != 0;
	d.p22 = t->p22.data(); d.sigma = t->sigma.data(); d.cdf = t->cdf.data(); d.qf = t->qf.data();
	d.n_p22 = res; d.n_sigma = res; d.n_cdf = res; d.n_qf = (int)t->qf.size();
	d.fr.kind = FR_SPLINE; d.fr.npts = res; d.fr.pts = t->fresnel.data();
}

} // namespace

#include "djb_cpu_aniso.inc"

namespace djbcpu {

// ------------------------------------------------------------------ context
djb_status ctx_create(djb_ctx **out)
{
	if (!out) return djbk::set_error(DJB_ERR_INVALID_ARGUMENT, "djb_error: null argument");
	CpuCtx *c = new CpuCtx();
	unsigned hc = std::thread::hardware_concurrency();
	c->threads = hc ? (int)std::min(hc, 64u) : 1;
	if (const char *e = getenv("DJB_CPU_THREADS")) { int v = atoi(e); if (v >= 1) c->threads = v; }
	c->t0 = std::chrono::steady_clock::now();
	*out = (djb_ctx *)c;
	return DJB_OK;
}
djb_status ctx_destroy(djb_ctx *ctx) { delete C(ctx); return DJB_OK; }
djb_status timer_start(djb_ctx *ctx) { C(ctx)->t0 = std::chrono::steady_clock::now(); return DJB_OK; }
djb_status timer_stop_ms(djb_ctx *ctx, float *ms)
{
	*ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - C(ctx)->t0).count();
	return DJB_OK;
}
djb_ctx *twin_ctx()
{
	static CpuCtx *c = [] { CpuCtx *x = new CpuCtx(); x->threads = 1; return x; }();   // scalar calls: the caller's thread only
	return (djb_ctx *)c;
}

// ------------------------------------------------------------------ constructors
djb_status create_microfacet(djb_ctx *ctx, int kind, const djb_fresnel_desc *f, int shadow, djb_brdf **out)
{
	CpuBrdf *b = alloc_brdf(ctx, kind);
	b->dev.shadow = shadow != 0;
	djb_status st = apply_fresnel(b, f);
	if (st != DJB_OK) { delete b; return st; }
	*out = (djb_brdf *)b;
	return DJB_OK;
}

djb_status create_merl_from_memory(djb_ctx *ctx, const double *samples, int64_t n, djb_brdf **out)
{
	if (n <= 0) return djbk::set_error(DJB_ERR_BAD_HEADER, "djb_error: Failed to read MERL header\n");
	if (n != MERL_N)
		return djbk::set_error(DJB_ERR_BAD_HEADER, "djb_error: MERL table has %lld samples per channel, expected %lld\n", (long long)n, MERL_N);
	CpuBrdf *b = alloc_brdf(ctx, KIND_MERL);
	b->raw.assign(samples, samples + 3 * (size_t)n);
	b->merl.resize((size_t)n);
	MerlTexel *tab = b->merl.data();
	const double *raw = b->raw.data();
	parallel_for(C(ctx), n, 1 << 16, [=](long long k0, long long k1) { for (long long k = k0; k < k1; ++k) tab[k] = merl_convert_one(raw, n, k); });
	b->dev.merl = tab;
	*out = (djb_brdf *)b;
	return DJB_OK;
}
djb_status create_merl_from_file(djb_ctx *ctx, const char *path, djb_brdf **out)
{
	std::vector<double> payload;
	djb_status st = read_doubles(path, true, &payload);
	if (st != DJB_OK) return st;
	return create_merl_from_memory(ctx, payload.data(), (int64_t)(payload.size() / 3), out);
}
djb_status create_merl_from_texels(djb_ctx *ctx, const float *texels3, djb_brdf **out)
{
	CpuBrdf *b = alloc_brdf(ctx, KIND_MERL);
	b->merl.resize((size_t)MERL_N);
	memcpy(b->merl.data(), texels3, sizeof(MerlTexel) * (size_t)MERL_N);
	b->dev.merl = b->merl.data();
	*out = (djb_brdf *)b;
	return DJB_OK;
}

djb_status create_utia_from_memory(djb_ctx *ctx, const double *samples, djb_brdf **out)
{
	CpuBrdf *b = alloc_brdf(ctx, KIND_UTIA);
	b->raw.assign(samples, samples + (size_t)UTIA_N);
	b->utia.resize(8 * (size_t)(UTIA_N / 3));
	float4 *tab = b->utia.data();
	const double *raw = b->raw.data();
	parallel_for(C(ctx), UTIA_N / 3, 1 << 13, [=](long long e0, long long e1) { for (long long e = e0; e < e1; ++e) utia_convert_one(raw, UTIA_N, e, tab); });
	b->dev.utia = tab;
	*out = (djb_brdf *)b;
	return DJB_OK;
}
djb_status create_utia_from_records(djb_ctx *ctx, const float *records, djb_brdf **out)
{
	CpuBrdf *b = alloc_brdf(ctx, KIND_UTIA);
	b->utia.resize(8 * (size_t)(UTIA_N / 3));
	memcpy(b->utia.data(), records, sizeof(float4) * b->utia.size());
	b->dev.utia = b->utia.data();
	*out = (djb_brdf *)b;
	return DJB_OK;
}
djb_status create_utia_from_file(djb_ctx *ctx, const char *path, djb_brdf **out)
{
	std::vector<double> payload;
	djb_status st = read_doubles(path, false, &payload);
	if (st != DJB_OK) return st;
	return create_utia_from_memory(ctx, payload.data(), out);
}

djb_status create_lambert(djb_ctx *ctx, djb_brdf **out) { *out = (djb_brdf *)alloc_brdf(ctx, KIND_LAMBERT); return DJB_OK; }

djb_status create_model(djb_ctx *ctx, int kind, const double *row, int count, djb_brdf **out)
{
	CpuBrdf *b = alloc_brdf(ctx, kind);
	b->model.assign(row, row + count);
	b->dev.model = b->model.data();
	Fresnel &fr = b->dev.fr;
	if (kind == KIND_SGD) {          // fresnel::sgd(vec3::from_raw(f0), vec3::from_raw(f1)), dj_brdf.h:3443
		fr.kind = FR_SGD;
		for (int c = 0; c < 3; ++c) { fr.a[c] = (float)row[12 + c]; fr.b[c] = (float)row[15 + c]; }
	} else {                         // fresnel::unpolarized(vec3(ior)), dj_brdf.h:3623
		fr.kind = FR_UNPOLARIZED;
		for (int c = 0; c < 3; ++c) fr.a[c] = (float)row[8];
	}
	*out = (djb_brdf *)b;
	return DJB_OK;
}

djb_status create_tabular(djb_ctx *ctx, const djb_brdf *src, int res, int shadow, djb_brdf **out)
{
	if (res <= 2) return djbk::set_error(DJB_ERR_INVALID_ARGUMENT, "djb_error: Invalid Resolution");   // dj_brdf.h:2218
	Params std_p;
	djb_status st = params_for(nullptr, -1, &std_p);
	if (st != DJB_OK) return st;
	FitResult R;
	fit_tabular(B(src)->dev, std_p, res, shadow != 0, R);
	CpuBrdf *t = alloc_brdf(ctx, KIND_TABULAR);
	t->p22.swap(R.p22); t->sigma.swap(R.sigma); t->cdf.swap(R.cdf); t->qf.swap(R.qf); t->fresnel.swap(R.fresnel);
	t->qf.resize(R.n_qf);
	t->alpha_beckmann = R.alpha_beckmann; t->alpha_ggx = R.alpha_ggx;
	finish_tabular(t, res, shadow);
	*out = (djb_brdf *)t;
	return DJB_OK;
}

djb_status create_tabular_from_tables(djb_ctx *ctx, int shadow, int res, const float *p22, const float *sigma, const float *cdf,
                                      const float *qf, int n_qf, const float *fresnel3, float alpha_b, float alpha_g, djb_brdf **out)
{
	CpuBrdf *t = alloc_brdf(ctx, KIND_TABULAR);
	t->p22.assign(p22, p22 + res); t->sigma.assign(sigma, sigma + res); t->cdf.assign(cdf, cdf + res);
	t->qf.assign(qf, qf + n_qf); t->fresnel.assign(fresnel3, fresnel3 + 3 * (size_t)res);
	t->alpha_beckmann = alpha_b; t->alpha_ggx = alpha_g;
	finish_tabular(t, res, shadow);
	*out = (djb_brdf *)t;
	return DJB_OK;
}

// ---- user-defined sources (dj_brdf.h:74-109: any class deriving from djb::brdf can be fitted).  The (i, o) pairs at which
// tabular::tabular(const brdf &, res, shadow) calls brdf.eval, in the reference's call order: the res - 1 back-scatter
// directions of compute_p22_smith (dj_brdf.h:2488-2494), then the pairs of compute_fresnel (dj_brdf.h:2589-2611; dir_i is
// forced to (0, 0, 1) there).  A pair the reference's loop never reaches reads NaN.  Slot numbering = fit_merl_slot_count.
int fit_query_count(int res) { return res > 2 ? fit_merl_slot_count(res) : 0; }
int fit_aniso_query_count(int elev, int azim) { return elev > 1 && azim > 1 ? aniso_query_count(elev, azim) : 0; }
djb_status fit_query_dirs(int res, float *i3, float *o3)
{
	if (res <= 2) return djbk::set_error(DJB_ERR_INVALID_ARGUMENT, "djb_error: Invalid Resolution");   // dj_brdf.h:2218
	const int cnt = res - 1;
	auto put = [](float *p, int s, v3 v) { p[3 * (size_t)s] = v.x; p[3 * (size_t)s + 1] = v.y; p[3 * (size_t)s + 2] = v.z; };
	for (int k = 0; k < cnt; ++k) {
		float th = fit_backscatter_theta(k, cnt);
		v3 w = from_angles(th * th, 0.0f);
		put(i3, k, w); put(o3, k, w);
	}
	const float qnan = __builtin_nanf("");
	for (int i = 0; i < cnt; ++i)
		for (int j = 0; j <= cnt; ++j) {
			const int s = cnt + i * (cnt + 1) + j;
			v3 dir_i, dir_o;
			if (!fit_fresnel_dirs(i, j, cnt, dir_i, dir_o)) dir_i = dir_o = mk(qnan, qnan, qnan);
			put(i3, s, dir_i); put(o3, s, dir_o);
		}
	return DJB_OK;
}
djb_status fit_aniso_query_dirs(int elev, int azim, float *i3, float *o3)
{
	if (elev <= 1 || azim <= 1 || elev > 1024 || azim > 1024)
		return djbk::set_error(DJB_ERR_INVALID_ARGUMENT, "djb_error: Invalid Resolution");           // dj_brdf.h:2244
	aniso_query_dirs(elev, azim, i3, o3);
	return DJB_OK;
}

// the source of a fit from per-slot samples: what brdf.eval returned at each query slot (rgb, 3 floats per slot)
static Brdf sampled_source(const float *rgb)
{
	Brdf src;
	memset(&src, 0, sizeof src);
	src.kind = KIND_MERL; src.shadow = 1; src.fr.kind = FR_IDEAL;
	src.merl = (const MerlTexel *)rgb; src.merl_sparse = 1;
	return src;
}
djb_status create_tabular_from_samples(djb_ctx *ctx, int res, int shadow, const float *rgb, djb_brdf **out)
{
	if (res <= 2) return djbk::set_error(DJB_ERR_INVALID_ARGUMENT, "djb_error: Invalid Resolution");   // dj_brdf.h:2218
	Params std_p;
	djb_status st = params_for(nullptr, -1, &std_p);
	if (st != DJB_OK) return st;
	FitResult R;
	fit_tabular(sampled_source(rgb), std_p, res, shadow != 0, R);
	CpuBrdf *t = alloc_brdf(ctx, KIND_TABULAR);
	t->p22.swap(R.p22); t->sigma.swap(R.sigma); t->cdf.swap(R.cdf); t->qf.swap(R.qf); t->fresnel.swap(R.fresnel);
	t->qf.resize(R.n_qf);
	t->alpha_beckmann = R.alpha_beckmann; t->alpha_ggx = R.alpha_ggx;
	finish_tabular(t, res, shadow);
	*out = (djb_brdf *)t;
	return DJB_OK;
}

static void finish_aniso(CpuBrdf *t, int shadow)
{
	Brdf &d = t->dev;
	d.shadow = shadow != 0;
	d.p22 = t->aniso[0].data(); d.sigma = t->aniso[1].data(); d.n_p22 = d.n_sigma = t->elev * t->azim;
	d.a_pdf1 = t->aniso[2].data(); d.a_cdf1 = t->aniso[3].data(); d.a_qf1 = t->aniso[4].data();
	d.a_pdf2 = t->aniso[5].data(); d.a_cdf2 = t->aniso[6].data(); d.a_qf2 = t->aniso[7].data();
	d.elev = t->elev; d.azim = t->azim; d.n_a_cdf1 = t->azim; d.n_a_qf1 = (int)t->aniso[4].size();
	d.fr.kind = FR_SPLINE; d.fr.pts = t->fresnel.data(); d.fr.npts = t->elev;
}

djb_status create_aniso_from_tables(djb_ctx *ctx, int shadow, int elev, int azim, const float *const tabs[8], const int counts[8],
                                    const float *fresnel3, const float fit10[10], int qf2_entries, djb_brdf **out)
{
	CpuBrdf *t = alloc_brdf(ctx, KIND_TABULAR_ANISO);
	t->elev = elev; t->azim = azim; t->aniso_qf2_entries = qf2_entries;
	for (int k = 0; k < 8; ++k) t->aniso[k].assign(tabs[k], tabs[k] + counts[k]);
	t->fresnel.assign(fresnel3, fresnel3 + 3 * (size_t)elev);
	memcpy(t->aniso_fit, fit10, sizeof t->aniso_fit);
	finish_aniso(t, shadow);
	*out = (djb_brdf *)t;
	return DJB_OK;
}

djb_status create_tabular_anisotropic(djb_ctx *ctx, const djb_brdf *src, int elev, int azim, int shadow, djb_brdf **out)
{
	if (elev <= 1 || azim <= 1 || elev > 1024 || azim > 1024)
		return djbk::set_error(DJB_ERR_INVALID_ARGUMENT, "djb_error: Invalid Resolution");           // dj_brdf.h:2244
	Params std_p;
	djb_status st = params_for(nullptr, -1, &std_p);
	if (st != DJB_OK) return st;
	AnisoFit A;
	fit_aniso(C(ctx), B(src)->dev, std_p, elev, azim, shadow != 0, A);
	CpuBrdf *t = alloc_brdf(ctx, KIND_TABULAR_ANISO);
	t->elev = elev; t->azim = azim; t->aniso_qf2_entries = A.qf2_entries;
	for (int k = 0; k < 8; ++k) t->aniso[k].swap(A.tab[k]);
	t->fresnel.swap(A.fresnel);
	memcpy(t->aniso_fit, A.fit, sizeof t->aniso_fit);
	finish_aniso(t, shadow);
	*out = (djb_brdf *)t;
	return DJB_OK;
}

djb_status create_tabular_anisotropic_from_samples(djb_ctx *ctx, int elev, int azim, int shadow, const float *rgb, djb_brdf **out)
{
	if (elev <= 1 || azim <= 1 || elev > 1024 || azim > 1024)
		return djbk::set_error(DJB_ERR_INVALID_ARGUMENT, "djb_error: Invalid Resolution");           // dj_brdf.h:2244
	Params std_p;
	djb_status st = params_for(nullptr, -1, &std_p);
	if (st != DJB_OK) return st;
	AnisoFit A;
	fit_aniso(C(ctx), sampled_source(rgb), std_p, elev, azim, shadow != 0, A);
	CpuBrdf *t = alloc_brdf(ctx, KIND_TABULAR_ANISO);
	t->elev = elev; t->azim = azim; t->aniso_qf2_entries = A.qf2_entries;
	for (int k = 0; k < 8; ++k) t->aniso[k].swap(A.tab[k]);
	t->fresnel.swap(A.fresnel);
	memcpy(t->aniso_fit, A.fit, sizeof t->aniso_fit);
	finish_aniso(t, shadow);
	*out = (djb_brdf *)t;
	return DJB_OK;
}

// a microfacet BRDF around a user-defined NDF (host code): dj_brdf.h:283-295 / 307-314
djb_status create_user_microfacet(djb_ctx *ctx, const djb_user_ndf *ndf, const djb_fresnel_desc *f, int shadow, djb_brdf **out)
{
	static_assert(sizeof(UserNdf) == sizeof(djb_user_ndf), "UserNdf mirrors djb_user_ndf");
	if (!ndf || !ndf->supports_smith_vndf_sampling) return djbk::set_error(DJB_ERR_INVALID_ARGUMENT, "djb_error: user NDF without supports_smith_vndf_sampling");
	const bool radial = ndf->p22_radial != nullptr;
	if (radial ? !(ndf->sigma_std_radial && ndf->qf_radial) : !(ndf->p22_std && ndf->sigma_std && ndf->sample_vp22_std))
		return djbk::set_error(DJB_ERR_INVALID_ARGUMENT, "djb_error: user NDF lacks a required callback (radial: p22_radial, sigma_std_radial, qf_radial; "
		                                                 "microfacet: p22_std, sigma_std, sample_vp22_std)");
	CpuBrdf *b = alloc_brdf(ctx, KIND_USER);
	memcpy(&b->ndf, ndf, sizeof b->ndf);
	b->dev.user_ndf = &b->ndf;
	b->dev.shadow = shadow != 0;
	djb_status st = apply_fresnel(b, f);
	if (st != DJB_OK) { delete b; return st; }
	*out = (djb_brdf *)b;
	return DJB_OK;
}

djb_status destroy(djb_brdf *b) { delete B(b); return DJB_OK; }
int kind(const djb_brdf *b) { return B(b)->dev.kind; }
int get_shadow(const djb_brdf *b) { return B(b)->dev.shadow; }
static bool is_microfacet_kind(int k) { return DJB_IS_MICROFACET(k); }
djb_status set_shadow(djb_brdf *b, int shadow)
{
	if (!is_microfacet_kind(B(b)->dev.kind)) return djbk::set_error(DJB_ERR_INVALID_ARGUMENT, "djb_error: set_shadow needs a microfacet BRDF");
	B(b)->dev.shadow = shadow != 0;
	return DJB_OK;
}
djb_status set_fresnel(djb_brdf *b_, const djb_fresnel_desc *f)
{
	CpuBrdf *b = B(b_);
	if (!is_microfacet_kind(b->dev.kind)) return djbk::set_error(DJB_ERR_INVALID_ARGUMENT, "djb_error: set_fresnel needs a microfacet BRDF");
	Fresnel saved = b->dev.fr;
	std::vector<float> saved_pts = b->fresnel;
	djb_status st = apply_fresnel(b, f);
	if (st != DJB_OK) { b->fresnel = saved_pts; b->dev.fr = saved; if (saved.kind == FR_SPLINE) b->dev.fr.pts = b->fresnel.data(); }
	return st;
}

djb_status get_fresnel(const djb_brdf *b_, djb_fresnel_desc *out)
{
	const CpuBrdf *b = B(b_);
	const int k = b->dev.kind;
	if (!is_microfacet_kind(k) && k != KIND_SGD && k != KIND_ABC) return djbk::set_error(DJB_ERR_INVALID_ARGUMENT, "djb_error: this brdf has no Fresnel term");
	const Fresnel &fr = b->dev.fr;
	memset(out, 0, sizeof *out);
	out->kind = fr.kind;
	for (int c = 0; c < 3; ++c) { out->a[c] = fr.a[c]; out->b[c] = fr.b[c]; }
	if (fr.kind == FR_SPLINE) { out->points = b->fresnel.data(); out->npoints = fr.npts; }
	return DJB_OK;
}

djb_status get_samples(const djb_brdf *b_, double *out, int64_t capacity, int64_t *count)
{
	const CpuBrdf *b = B(b_);
	if (b->dev.kind != KIND_MERL && b->dev.kind != KIND_UTIA)
		return djbk::set_error(DJB_ERR_INVALID_ARGUMENT, "djb_error: get_samples needs a merl or utia BRDF");
	if (b->raw.empty())
		return djbk::set_error(DJB_ERR_NOT_IMPLEMENTED, "djb_error: this table was built from converted texels; the payload was not kept");
	*count = (int64_t)b->raw.size();
	if (!out) return DJB_OK;
	if (capacity < *count) return djbk::set_error(DJB_ERR_INVALID_ARGUMENT, "djb_error: get_samples needs room for %lld doubles", (long long)*count);
	memcpy(out, b->raw.data(), sizeof(double) * b->raw.size());
	if (b->dev.kind == KIND_UTIA) {      // utia::normalize, dj_brdf.h:1162-1177
		const float k = 1.f / 140.f;
		for (size_t j = 0; j < b->raw.size(); ++j) { double v = out[j] > 0.0 ? out[j] : 0.0; out[j] = v * k; }
	}
	return DJB_OK;
}

djb_status tabular_get(const djb_brdf *b, int which, float *outp, int *count)
{
	const CpuBrdf *tab = B(b);
	if (tab->dev.kind != KIND_TABULAR) return djbk::set_error(DJB_ERR_INVALID_ARGUMENT, "djb_error: not a tabular brdf");
	const std::vector<float> *v;
	switch (which) {
	case DJB_TAB_P22: v = &tab->p22; break;
	case DJB_TAB_SIGMA: v = &tab->sigma; break;
	case DJB_TAB_CDF: v = &tab->cdf; break;
	case DJB_TAB_QF: v = &tab->qf; break;
	case DJB_TAB_FRESNEL: v = &tab->fresnel; break;
	default: return djbk::set_error(DJB_ERR_INVALID_ARGUMENT, "djb_error: unknown table %d", which);
	}
	if (count) *count = (int)(which == DJB_TAB_FRESNEL ? v->size() / 3 : v->size());
	if (outp) memcpy(outp, v->data(), sizeof(float) * v->size());
	return DJB_OK;
}
djb_status tabular_fit(const djb_brdf *b, float *ab, float *ag)
{
	if (B(b)->dev.kind != KIND_TABULAR) return djbk::set_error(DJB_ERR_INVALID_ARGUMENT, "djb_error: not a tabular brdf");
	if (ab) *ab = B(b)->alpha_beckmann;
	if (ag) *ag = B(b)->alpha_ggx;
	return DJB_OK;
}
djb_status aniso_get(const djb_brdf *b, int which, float *outp, int *count, int *elev, int *azim)
{
	const CpuBrdf *tab = B(b);
	if (tab->dev.kind != KIND_TABULAR_ANISO) return djbk::set_error(DJB_ERR_INVALID_ARGUMENT, "djb_error: not a tabular_anisotropic brdf");
	if (elev) *elev = tab->elev;
	if (azim) *azim = tab->azim;
	if (which == DJB_ATAB_QF2_ENTRIES) { if (count) *count = tab->aniso_qf2_entries; return DJB_OK; }
	const std::vector<float> *v;
	if (which >= 0 && which < 8) v = &tab->aniso[which];
	else if (which == DJB_ATAB_FRESNEL) v = &tab->fresnel;
	else return djbk::set_error(DJB_ERR_INVALID_ARGUMENT, "djb_error: unknown table %d", which);
	if (count) *count = (int)(which == DJB_ATAB_FRESNEL ? v->size() / 3 : v->size());
	if (outp) memcpy(outp, v->data(), sizeof(float) * v->size());
	return DJB_OK;
}
djb_status aniso_fit(const djb_brdf *b, djb_params *beckmann, djb_params *ggx)
{
	const CpuBrdf *tab = B(b);
	if (tab->dev.kind != KIND_TABULAR_ANISO) return djbk::set_error(DJB_ERR_INVALID_ARGUMENT, "djb_error: not a tabular_anisotropic brdf");
	for (int k = 0; k < 2; ++k) {
		djb_params *p = k == 0 ? beckmann : ggx;
		if (!p) continue;
		p->kind = DJB_PARAMS_PDFPARAMS;
		for (int c = 0; c < 5; ++c) p->v[c] = tab->aniso_fit[5 * k + c];
	}
	return DJB_OK;
}

// ------------------------------------------------------------------ batch operators
djb_status eval(djb_ctx *ctx, const djb_brdf *b_, int64_t n, const djb_vec3_view *i, const djb_vec3_view *o, const djb_params *params,
                const djb_vec3_view *out_fr, float *out_pdf, int want)
{
	if (!b_) return djbk::set_error(DJB_ERR_INVALID_ARGUMENT, "djb_error: null brdf");
	if (n < 0) return djbk::set_error(DJB_ERR_INVALID_ARGUMENT, "djb_error: negative batch size");
	const Brdf &b = B(b_)->dev;
	Params p;
	djb_status st = params_for(params, b.kind, &p);
	if (st != DJB_OK) return st;
	if (!valid(i) || !valid(o)) return djbk::set_error(DJB_ERR_INVALID_ARGUMENT, "djb_error: null vec3 view");
	if ((want & 3) && !valid(out_fr)) return djbk::set_error(DJB_ERR_INVALID_ARGUMENT, "djb_error: null output vec3 view");
	if ((want & 4) && !out_pdf) return djbk::set_error(DJB_ERR_INVALID_ARGUMENT, "djb_error: null output array");
	const View vi = view_of(i), vo = view_of(o), vout = (want & 3) ? view_of(out_fr) : View{ nullptr, nullptr, nullptr, 0 };
	parallel_for(C(ctx), n, 4096, [&](long long k0, long long k1) {
		DJB_KIND_SWITCH(b.kind, (eval_kind<K>(b, p, k0, k1, vi, vo, vout, out_pdf, want)))
	});
	return DJB_OK;
}

djb_status sample(djb_ctx *ctx, const djb_brdf *b_, int64_t n, const float *u1, const float *u2, uint32_t s1, uint32_t s2, uint64_t start,
                  const djb_vec3_view *o, const djb_params *params, const djb_vec3_view *out_w, const djb_vec3_view *out_i, float *out_pdf)
{
	if (!b_) return djbk::set_error(DJB_ERR_INVALID_ARGUMENT, "djb_error: null brdf");
	if (n < 0) return djbk::set_error(DJB_ERR_INVALID_ARGUMENT, "djb_error: negative batch size");
	const Brdf &b = B(b_)->dev;
	Params p;
	djb_status st = params_for(params, b.kind, &p);
	if (st != DJB_OK) return st;
	const bool is = out_w != nullptr;
	if (!valid(o) || !valid(out_i) || (is && (!valid(out_w) || !out_pdf)) || ((u1 == nullptr) != (u2 == nullptr)))
		return djbk::set_error(DJB_ERR_INVALID_ARGUMENT, "djb_error: null argument");
	if (b.kind == KIND_USER) {      // a radial user NDF that samples the Smith way needs qf2_radial / qf3_radial (include/djb_hip.h); the base class of
		const UserNdf &u = B(b_)->ndf;   // the reference throws "Not Implemented" there (dj_brdf.h:1785-1790), a raw C caller must not jump through NULL
		if (u.p22_radial && (!u.qf2_radial || !u.qf3_radial) && u.supports_smith_vndf_sampling(u.user))
			return djbk::set_error(DJB_ERR_NOT_IMPLEMENTED, "djb_error: Not Implemented");
	}
	const View vo = view_of(o), vi = view_of(out_i), vw = is ? view_of(out_w) : View{ nullptr, nullptr, nullptr, 0 };
	parallel_for(C(ctx), n, 2048, [&](long long k0, long long k1) {
		if (is) { DJB_KIND_SWITCH(b.kind, (sample_loop<K, true>(b, p, k0, k1, u1, u2, s1, s2, start, vo, vi, vw, out_pdf))) }
		else { DJB_KIND_SWITCH(b.kind, (sample_loop<K, false>(b, p, k0, k1, u1, u2, s1, s2, start, vo, vi, vw, out_pdf))) }
	});
	return DJB_OK;
}

djb_status eval_pp(djb_ctx *ctx, const djb_brdf *b_, int64_t n, const djb_vec3_view *i, const djb_vec3_view *o, const float *rec,
                   int mode, const float *base5, float scale, int lean_flags, int want, const djb_vec3_view *out_fr, float *out_pdf,
                   float *out_pp)
{
	if (!b_ || !rec) return djbk::set_error(DJB_ERR_INVALID_ARGUMENT, "djb_error: null argument");
	const Brdf &b = B(b_)->dev;
	if (!is_microfacet_kind(b.kind) || b.kind == KIND_USER) return djbk::set_error(DJB_ERR_INVALID_ARGUMENT, "djb_error: per-pair params need one of the library's microfacet BRDFs");
	if (!valid(i) || !valid(o) || ((want & 3) && !valid(out_fr)) || ((want & 4) && !out_pdf))
		return djbk::set_error(DJB_ERR_INVALID_ARGUMENT, "djb_error: null argument");
	const LeanCfg base = lean_cfg(base5, scale, lean_flags);
	const View vi = view_of(i), vo = view_of(o), vout = (want & 3) ? view_of(out_fr) : View{ nullptr, nullptr, nullptr, 0 };
	parallel_for(C(ctx), n, 4096, [&](long long k0, long long k1) {
#define DJB_PP_(K_) (mode == 0 ? pp_kind<K_, 0>(b, k0, k1, vi, vo, rec, base, vout, out_pdf, out_pp, want) \
                               : pp_kind<K_, 1>(b, k0, k1, vi, vo, rec, base, vout, out_pdf, out_pp, want))
		switch (b.kind) {
		case KIND_BECKMANN: DJB_PP_(KIND_BECKMANN); break;
		case KIND_GGX: DJB_PP_(KIND_GGX); break;
		case KIND_TABULAR: DJB_PP_(KIND_TABULAR); break;
		case KIND_TABULAR_ANISO: DJB_PP_(KIND_TABULAR_ANISO); break;
		}
#undef DJB_PP_
	});
	return DJB_OK;
}

template <int KIND, int MODE>
static void pp_sample_loop(const Brdf &b, long long k0, long long k1, const float *u1, const float *u2, const View &vo, const float *rec,
                           const LeanCfg &base, bool is, const View &vi_out, const View &vw_out, float *out_pdf, float *out_pp)
{
	const GlibcTabs gt = glibc_tabs_global();
	for (long long k = k0; k < k1; ++k) {
		v3 i_out, w; float pdf;
		if (is) pp_sample_one<KIND, true, MODE>(b, u1[k], u2[k], load3(vo, k), rec + 5 * k, base, out_pp ? out_pp + 5 * k : nullptr, gt, i_out, w, pdf);
		else pp_sample_one<KIND, false, MODE>(b, u1[k], u2[k], load3(vo, k), rec + 5 * k, base, out_pp ? out_pp + 5 * k : nullptr, gt, i_out, w, pdf);
		store3(vi_out, k, i_out);
		if (is) { store3(vw_out, k, w); out_pdf[k] = pdf; }
	}
}

djb_status sample_pp(djb_ctx *ctx, const djb_brdf *b_, int64_t n, const float *u1, const float *u2, const djb_vec3_view *o,
                     const float *rec, int mode, const float *base5, float scale, int lean_flags, const djb_vec3_view *out_w,
                     const djb_vec3_view *out_i, float *out_pdf, float *out_pp)
{
	if (!b_ || !rec || !u1 || !u2) return djbk::set_error(DJB_ERR_INVALID_ARGUMENT, "djb_error: null argument");
	const Brdf &b = B(b_)->dev;
	if (!is_microfacet_kind(b.kind) || b.kind == KIND_USER) return djbk::set_error(DJB_ERR_INVALID_ARGUMENT, "djb_error: per-pair params need one of the library's microfacet BRDFs");
	const bool is = out_w != nullptr;
	if (!valid(o) || !valid(out_i) || (is && (!valid(out_w) || !out_pdf)))
		return djbk::set_error(DJB_ERR_INVALID_ARGUMENT, "djb_error: null argument");
	const LeanCfg base = lean_cfg(base5, scale, lean_flags);
	const View vo = view_of(o), vi = view_of(out_i), vw = is ? view_of(out_w) : View{ nullptr, nullptr, nullptr, 0 };
	parallel_for(C(ctx), n, 2048, [&](long long k0, long long k1) {
#define DJB_SPP_(K_) (mode == 0 ? pp_sample_loop<K_, 0>(b, k0, k1, u1, u2, vo, rec, base, is, vi, vw, out_pdf, out_pp) \
                                : pp_sample_loop<K_, 1>(b, k0, k1, u1, u2, vo, rec, base, is, vi, vw, out_pdf, out_pp))
		switch (b.kind) {
		case KIND_BECKMANN: DJB_SPP_(KIND_BECKMANN); break;
		case KIND_GGX: DJB_SPP_(KIND_GGX); break;
		case KIND_TABULAR: DJB_SPP_(KIND_TABULAR); break;
		case KIND_TABULAR_ANISO: DJB_SPP_(KIND_TABULAR_ANISO); break;
		}
#undef DJB_SPP_
	});
	return DJB_OK;
}

djb_status query(djb_ctx *ctx, const djb_brdf *b_, int which, int64_t n, const djb_vec3_view *a, const djb_vec3_view *bb,
                 const djb_vec3_view *c, const djb_params *params, const djb_vec3_view *out)
{
	const Brdf &b = B(b_)->dev;
	if (b.kind == KIND_USER && which >= Q_P22_RADIAL && which <= Q_QF1) {   // the radial queries of a user NDF: only what the user's class has
		const UserNdf &u = B(b_)->ndf;
		const bool have = which == Q_P22_RADIAL ? u.p22_radial != nullptr : which == Q_SIGMA_STD_RADIAL ? u.sigma_std_radial != nullptr
		                : which == Q_CDF_RADIAL ? u.cdf_radial != nullptr : which == Q_QF_RADIAL ? u.qf_radial != nullptr
		                : which == Q_QF2_RADIAL ? u.qf2_radial != nullptr : which == Q_QF3_RADIAL ? u.qf3_radial != nullptr : false;
		if (!have) return djbk::set_error(DJB_ERR_NOT_IMPLEMENTED, "djb_error: Not Implemented");
	}
	Params p;
	djb_status st = params_for(params, b.kind, &p);
	if (st != DJB_OK) return st;
	const View null_view{ nullptr, nullptr, nullptr, 0 };
	const View va = view_of(a), vb = bb ? view_of(bb) : null_view, vc = c ? view_of(c) : null_view, vout = view_of(out);
	parallel_for(C(ctx), n, 4096, [&](long long k0, long long k1) {
		for (long long k = k0; k < k1; ++k) {
			v3 r = mk(0, 0, 0);
			switch (b.kind) {
			case KIND_BECKMANN: r = query_one<KIND_BECKMANN>(b, p, which, k, va, vb, vc); break;
			case KIND_GGX: r = query_one<KIND_GGX>(b, p, which, k, va, vb, vc); break;
			case KIND_TABULAR: r = query_one<KIND_TABULAR>(b, p, which, k, va, vb, vc); break;
			case KIND_TABULAR_ANISO: r = query_one<KIND_TABULAR_ANISO>(b, p, which, k, va, vb, vc); break;
			case KIND_USER: r = query_one<KIND_USER>(b, p, which, k, va, vb, vc); break;
			case KIND_SGD: r = model_query_one<KIND_SGD>(b, which, k, va, vb, vc); break;
			case KIND_ABC: r = model_query_one<KIND_ABC>(b, which, k, va, vb, vc); break;
			}
			store3(vout, k, r);
		}
	});
	return DJB_OK;
}

djb_status io_hd(djb_ctx *ctx, int64_t n, const djb_vec3_view *a, const djb_vec3_view *b, const djb_vec3_view *c, const djb_vec3_view *d, bool inverse)
{
	if (!valid(a) || !valid(b) || !valid(c) || !valid(d)) return djbk::set_error(DJB_ERR_INVALID_ARGUMENT, "djb_error: null vec3 view");
	const View va = view_of(a), vb = view_of(b), vc = view_of(c), vd = view_of(d);
	parallel_for(C(ctx), n, 4096, [&](long long k0, long long k1) {
		for (long long k = k0; k < k1; ++k) {
			v3 r1, r2;
			if (!inverse) io_to_hd(load3(va, k), load3(vb, k), r1, r2);
			else hd_to_io(load3(va, k), load3(vb, k), r1, r2);
			store3(vc, k, r1); store3(vd, k, r2);
		}
	});
	return DJB_OK;
}

djb_status merl_index(djb_ctx *ctx, int64_t n, const djb_vec3_view *i, const djb_vec3_view *o, int32_t *out)
{
	if (!valid(i) || !valid(o) || !out) return djbk::set_error(DJB_ERR_INVALID_ARGUMENT, "djb_error: null argument");
	const View vi = view_of(i), vo = view_of(o);
	parallel_for(C(ctx), n, 4096, [&](long long k0, long long k1) {
		for (long long k = k0; k < k1; ++k) out[k] = djbdev::merl_index(load3(vi, k), load3(vo, k));
	});
	return DJB_OK;
}

// ------------------------------------------------------------------ batch fits: materials are independent -> one thread each
static djb_status fit_many(djb_ctx *ctx, const std::vector<const Brdf *> &srcs, int res, int shadow, float *ab, float *ag,
                           float *p22, float *sigma, float *cdf, float *qf, float *fresnel)
{
	if (res <= 2) return djbk::set_error(DJB_ERR_INVALID_ARGUMENT, "djb_error: Invalid Resolution");
	Params std_p;
	djb_status st = params_for(nullptr, -1, &std_p);
	if (st != DJB_OK) return st;
	const long long n = (long long)srcs.size();
	parallel_for(C(ctx), n, 1, [&](long long m0, long long m1) {
		for (long long m = m0; m < m1; ++m) {
			FitResult R;
			fit_tabular(*srcs[m], std_p, res, shadow != 0, R);
			if (ab) ab[m] = R.alpha_beckmann;
			if (ag) ag[m] = R.alpha_ggx;
			auto put = [&](float *dst, const std::vector<float> &v, int w) { if (dst) memcpy(dst + (size_t)m * res * w, v.data(), sizeof(float) * (size_t)res * w); };
			for (int k = R.n_qf; k < res; ++k) R.qf[k] = 0.0f;
			put(p22, R.p22, 1); put(sigma, R.sigma, 1); put(cdf, R.cdf, 1); put(qf, R.qf, 1); put(fresnel, R.fresnel, 3);
		}
	});
	return DJB_OK;
}

djb_status fit_brdf_batch(djb_ctx *ctx, int n_mat, const djb_brdf *const *srcs_in, int res, int shadow, float *ab, float *ag,
                          float *p22, float *sigma, float *cdf, float *qf, float *fresnel)
{
	std::vector<const Brdf *> srcs(n_mat);
	for (int m = 0; m < n_mat; ++m) {
		if (!srcs_in[m] || !is_cpu(srcs_in[m])) return djbk::set_error(DJB_ERR_INVALID_ARGUMENT, "djb_error: batch fit on a CPU context needs BRDFs of that context");
		srcs[m] = &B(srcs_in[m])->dev;
	}
	return fit_many(ctx, srcs, res, shadow, ab, ag, p22, sigma, cdf, qf, fresnel);
}

djb_status fit_merl_batch(djb_ctx *ctx, int n_mat, const double *const *tables, int res, int shadow, float *ab, float *ag,
                          float *p22, float *sigma, float *cdf, float *qf, float *fresnel)
{
	std::vector<djb_brdf *> mats(n_mat, nullptr);
	djb_status st = DJB_OK;
	for (int m = 0; m < n_mat && st == DJB_OK; ++m) st = create_merl_from_memory(ctx, tables[m], MERL_N, &mats[m]);
	if (st == DJB_OK) st = fit_brdf_batch(ctx, n_mat, mats.data(), res, shadow, ab, ag, p22, sigma, cdf, qf, fresnel);
	for (djb_brdf *b : mats) if (b) destroy(b);
	return st;
}

// what examples/merl_params.cpp:53-67 does per file; files are independent -> threads.  Only the entries a
// tabular(merl, res) fit reads are fetched from each (mapped) file: djb_merl_file.hpp
djb_status fit_merl_files(djb_ctx *ctx, int n_files, const char *const *paths, int res, int shadow, int threads, float *ab,
                          float *ag, double *timing)
{
	if (res <= 2) return djbk::set_error(DJB_ERR_INVALID_ARGUMENT, "djb_error: Invalid Resolution");
	Params std_p;
	djb_status st = params_for(nullptr, -1, &std_p);
	if (st != DJB_OK) return st;
	const auto t_begin = std::chrono::steady_clock::now();
	const int n_slots = fit_merl_slot_count(res);
	std::vector<int32_t> idx(n_slots);
	CpuCtx pool = *C(ctx);
	if (threads >= 1) pool.threads = threads;
	parallel_for(&pool, n_slots, 256, [&](long long s0, long long s1) { for (long long s = s0; s < s1; ++s) idx[s] = fit_merl_slot_index((int)s, res); });
	const djbfile::SlotPlan plan = djbfile::make_plan(idx);
	std::mutex mu;
	djb_status first = DJB_OK;
	std::string first_msg;
	int first_file = n_files;
	double load_s = 0.0;
	parallel_for(&pool, n_files, 1, [&](long long f0, long long f1) {
		std::vector<MerlTexel> slots(n_slots);
		for (long long f = f0; f < f1; ++f) {
			const auto t0 = std::chrono::steady_clock::now();
			std::string err;
			memset(slots.data(), 0, sizeof(MerlTexel) * slots.size());
			djb_status s = djbfile::gather_file(paths[f], plan, (float *)slots.data(), &err);
			const auto t1 = std::chrono::steady_clock::now();
			if (s != DJB_OK) {
				std::lock_guard<std::mutex> g(mu);
				if ((int)f < first_file) { first_file = (int)f; first = s; first_msg = err; }
				continue;
			}
			Brdf src;
			memset(&src, 0, sizeof src);
			src.kind = KIND_MERL; src.shadow = 1; src.fr.kind = FR_IDEAL; src.merl = slots.data(); src.merl_sparse = 1;
			FitResult R;
			fit_tabular(src, std_p, res, shadow != 0, R);
			ab[f] = R.alpha_beckmann; ag[f] = R.alpha_ggx;
			std::lock_guard<std::mutex> g(mu);
			load_s += std::chrono::duration<double>(t1 - t0).count();
		}
	});
	djbfile::t_failed_file = first != DJB_OK ? first_file : -1;
	if (first != DJB_OK) return djbk::set_error(first, "%s", first_msg.c_str());
	if (timing) {
		timing[0] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
		timing[1] = load_s; timing[2] = timing[0] - load_s < 0 ? 0 : timing[0] - load_s;   // thread-summed load vs wall: indicative only
		timing[3] = (double)n_files * (double)plan.idx.size() * 24.0;
	}
	return DJB_OK;
}

djb_status gen_directions(djb_ctx *ctx, int64_t n, uint32_t seed, uint64_t start, const djb_vec3_view *out)
{
	if (!valid(out)) return djbk::set_error(DJB_ERR_INVALID_ARGUMENT, "djb_error: null vec3 view");
	const View v = view_of(out);
	parallel_for(C(ctx), n, 1 << 14, [&](long long k0, long long k1) { for (long long k = k0; k < k1; ++k) store3(v, k, gen_direction(seed, start + (unsigned long long)k)); });
	return DJB_OK;
}
// the helpers the reference defines as file-static functions of its implementation section (dj_brdf.h:650-765), for callers that
// compiled against them: evaluated here by the per-unit code every operator uses
djb_status helper(int which, const float *in, float *out)
{
	const GlibcTabs gt{};
	switch (which) {
	case DJB_HELPER_ERF: out[0] = erf_(in[0]); return DJB_OK;
	case DJB_HELPER_ERFINV: out[0] = erfinv_(in[0], gt); return DJB_OK;
	case DJB_HELPER_XYZ_TO_THETA_PHI: xyz_to_theta_phi(mk(in[0], in[1], in[2]), out[0], out[1]); return DJB_OK;
	case DJB_HELPER_UNIFORM_TO_CONCENTRIC: uniform_to_concentric(in[0], in[1], out[0], out[1]); return DJB_OK;
	case DJB_HELPER_ROTATE_VECTOR: { const v3 r = rotate_axis(mk(in[0], in[1], in[2]), mk(in[3], in[4], in[5]), in[6]); out[0] = r.x; out[1] = r.y; out[2] = r.z; return DJB_OK; }
	default: return djbk::set_error(DJB_ERR_INVALID_ARGUMENT, "djb_error: unknown helper");
	}
}

djb_status gen_uniforms(djb_ctx *ctx, int64_t n, uint32_t seed, uint64_t start, float *out)
{
	if (!out) return djbk::set_error(DJB_ERR_INVALID_ARGUMENT, "djb_error: null argument");
	parallel_for(C(ctx), n, 1 << 14, [&](long long k0, long long k1) { for (long long k = k0; k < k1; ++k) out[k] = gen_uniform(seed, start + (unsigned long long)k); });
	return DJB_OK;
}
djb_status histogram_xy(djb_ctx *, int64_t n, const djb_vec3_view *v, int bins, unsigned long long *counts)
{
	if (!valid(v) || !counts) return djbk::set_error(DJB_ERR_INVALID_ARGUMENT, "djb_error: null argument");
	for (long long k = 0; k < n; ++k) {
		float x = v->x[k * v->stride], y = v->y[k * v->stride];
		int bx = (int)((x + 1.0f) * 0.5f * (float)bins), by = (int)((y + 1.0f) * 0.5f * (float)bins);
		bx = bx < 0 ? 0 : (bx >= bins ? bins - 1 : bx);
		by = by < 0 ? 0 : (by >= bins ? bins - 1 : by);
		++counts[by * bins + bx];
	}
	return DJB_OK;
}

unsigned long long trig_sweep_compare(int host_fn, uint32_t first_bits, int64_t count, const void *dev, int threads,
                                      uint32_t *bad3, int cap)
{
	CpuCtx pool;
	pool.threads = threads >= 1 ? threads : (int)std::max(1u, std::min(std::thread::hardware_concurrency(), 64u));
	std::mutex mu;
	unsigned long long n_bad = 0;
	const bool dbl = host_fn >= TRIG_DOUBLE;
	parallel_for(&pool, count, 1 << 16, [&](long long k0, long long k1) {
		for (long long k = k0; k < k1; ++k) {
			uint32_t xb = first_bits + (uint32_t)k, r1, r2 = 0;
			float x;
			memcpy(&x, &xb, 4);
			if (dbl) {
				double h = trig_site_d(host_fn, x), d = ((const double *)dev)[k];
				long long hb, db;
				memcpy(&hb, &h, 8); memcpy(&db, &d, 8);
				if (hb == db || (h != h && d != d)) continue;
				// same-sign finite doubles order as their bit patterns
				long long diff = ((hb ^ db) < 0) ? 0x7fffffffLL : (hb > db ? hb - db : db - hb);
				r1 = (uint32_t)std::min(diff, 0x7fffffffLL);
			} else {
				float h = trig_site(host_fn, x), d = ((const float *)dev)[k];
				memcpy(&r2, &h, 4); memcpy(&r1, &d, 4);
				if (r1 == r2 || (h != h && d != d)) continue;
			}
			std::lock_guard<std::mutex> g(mu);
			if (n_bad < (unsigned long long)cap && bad3) { bad3[3 * n_bad] = xb; bad3[3 * n_bad + 1] = r1; bad3[3 * n_bad + 2] = r2; }
			++n_bad;
		}
	});
	return n_bad;
}

} // namespace djbcpu
