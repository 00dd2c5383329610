// djb_host_fit.hip -- the C ABI of libdjb_hip.so, part 3: the fit drivers -- djb::tabular and djb::tabular_anisotropic
// construction (one k_fit launch / the 18 launches of launch_fit_aniso), their accessors, and the batch fits
// (djb_fit_merl_batch, djb_fit_brdf_batch).  Shared internals: djb_host.hpp.
#include "djb_host.hpp"

using namespace djbh;

extern "C" {

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
	// the directions of the Fresnel-ratio pass depend on the resolution only: once per context (the launch below is stream-ordered before the fit)
	split.fres_dirs = nullptr;
	{
		auto it = ctx->fit_fresnel_dirs.find(res);
		// built from the second material on: a context that fits one material once would only pay for them
		if (it == ctx->fit_fresnel_dirs.end() && (n_mat >= 2 || ++ctx->fit_seen[res] >= 2)) {
			float *d = nullptr;
			if (hipMalloc((void **)&d, sizeof(float) * djbk::fit_fresnel_dirs_floats(res)) == hipSuccess) {
				if (djbk::launch_fit_fresnel_dirs(ctx->stream, res, std_p, d) == hipSuccess) it = ctx->fit_fresnel_dirs.emplace(res, d).first;
				else { (void)hipGetLastError(); (void)hipFree(d); }
			} else (void)hipGetLastError();
		}
		if (it != ctx->fit_fresnel_dirs.end()) split.fres_dirs = it->second;
	}
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

// the tabular object of a fit of `src` (a device view: a resident BRDF, or the per-slot samples of a user-defined one)
static djb_status gpu_tabular(djb_ctx *ctx, const Brdf &src, int res, int shadow, djb_brdf **out)
{
	if (res <= 2) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: Invalid Resolution");
	djb_brdf *t;
	alloc_brdf(ctx, DJB_KIND_TABULAR, &t);
	t->dev.shadow = shadow != 0;
	t->p22.resize(res); t->sigma.resize(res); t->cdf.resize(res); t->qf.resize(res); t->fresnel.resize(3 * (size_t)res);
	int n_qf = 0;
	std::vector<Brdf> srcs(1, src);
	djb_status st = run_fit(ctx, srcs, src.kind, res, shadow, &t->alpha_beckmann, &t->alpha_ggx, t->p22.data(),
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
	return gpu_tabular(ctx, src->dev, res, shadow, out);
}
DJB_ABI_CATCH

// ---- fits of user-defined sources (dj_brdf.h:74-109: `eval` is the one pure virtual; tabular's constructor only ever calls
// brdf.eval, at directions that depend on the resolution alone).  The caller evaluates its BRDF on the host at the query slots
// (djb_fit_query_dirs) and hands the rgb samples over; the fit itself is the same k_fit launch, reading one rgb per slot.
static djb_status store_dirs(const std::vector<float> &v3s, int64_t n, const djb_vec3_view *out)
{
	if (!out) return DJB_OK;
	if (!out->x || !out->y || !out->z) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null view");
	for (int64_t k = 0; k < n; ++k) {
		out->x[k * out->stride] = v3s[3 * (size_t)k]; out->y[k * out->stride] = v3s[3 * (size_t)k + 1]; out->z[k * out->stride] = v3s[3 * (size_t)k + 2];
	}
	return DJB_OK;
}
djb_status djb_fit_query_dirs(int res, int64_t capacity, const djb_vec3_view *out_i, const djb_vec3_view *out_o, int64_t *count)
try {
	if (res <= 2) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: Invalid Resolution");            // dj_brdf.h:2218
	const int64_t n = djbcpu::fit_query_count(res);
	if (count) *count = n;
	if (!out_i && !out_o) return DJB_OK;
	if (capacity < n) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: a fit at resolution %d has %lld query slots, capacity %lld", res, (long long)n, (long long)capacity);
	djbhostlibm::init();
	std::vector<float> vi(3 * (size_t)n), vo(3 * (size_t)n);
	djb_status st = djbcpu::fit_query_dirs(res, vi.data(), vo.data());
	if (st == DJB_OK) st = store_dirs(vi, n, out_i);
	if (st == DJB_OK) st = store_dirs(vo, n, out_o);
	return st;
}
DJB_ABI_CATCH

djb_status djb_fit_aniso_query_dirs(int elev, int azim, int64_t capacity, const djb_vec3_view *out_i, const djb_vec3_view *out_o, int64_t *count)
try {
	if (elev <= 1 || azim <= 1 || elev > 1024 || azim > 1024)
		return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: Invalid Resolution");                       // dj_brdf.h:2244
	const int64_t n = djbcpu::fit_aniso_query_count(elev, azim);
	if (count) *count = n;
	if (!out_i && !out_o) return DJB_OK;
	if (capacity < n) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: a %d x %d fit has %lld query slots, capacity %lld", elev, azim, (long long)n, (long long)capacity);
	djbhostlibm::init();
	std::vector<float> vi(3 * (size_t)n), vo(3 * (size_t)n);
	djb_status st = djbcpu::fit_aniso_query_dirs(elev, azim, vi.data(), vo.data());
	if (st == DJB_OK) st = store_dirs(vi, n, out_i);
	if (st == DJB_OK) st = store_dirs(vo, n, out_o);
	return st;
}
DJB_ABI_CATCH

// the samples in HBM as a per-slot source (djbdev::Brdf::merl_sparse); `holder` owns the block
static djb_status upload_samples(djb_ctx *ctx, const float *rgb, int64_t n, djb_brdf **holder)
{
	djbdev::MerlTexel *d = nullptr;
	HIP_TRY(hipSetDevice(ctx->device));
	HIP_TRY(hipMalloc((void **)&d, sizeof(djbdev::MerlTexel) * (size_t)n));
	hipError_t e = hipMemcpyAsync(d, rgb, sizeof(float) * 3 * (size_t)n, hipMemcpyHostToDevice, ctx->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);          // pageable source: the caller may free `rgb` on return
	if (e != hipSuccess) { (void)hipFree(d); (void)hipGetLastError(); return fail(DJB_ERR_HIP, "djb_error: sample upload failed: %s", hipGetErrorString(e)); }
	djb_status st = djbk::wrap_merl_slots(ctx, d, holder);
	if (st != DJB_OK) { (void)hipFree(d); return st; }
	(*holder)->allocs.push_back(d);
	return DJB_OK;
}

djb_status djb_brdf_create_tabular_from_samples(djb_ctx *ctx, int res, int shadow, const float *rgb, int64_t count, djb_brdf **out)
try {
	if (!ctx || !rgb || !out) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null argument");
	if (res <= 2) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: Invalid Resolution");            // dj_brdf.h:2218
	if (count != djbcpu::fit_query_count(res))
		return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: a fit at resolution %d takes %d samples, got %lld", res, djbcpu::fit_query_count(res), (long long)count);
	if (is_cpu(ctx)) return djbcpu::create_tabular_from_samples(ctx, res, shadow, rgb, out);
	djb_status st = check_call(ctx, nullptr, 0, DJB_MEM_DEVICE);
	if (st != DJB_OK) return st;
	std::lock_guard<std::recursive_mutex> call_lock(ctx->call_mu);
	djb_brdf *holder = nullptr;
	if ((st = upload_samples(ctx, rgb, count, &holder)) != DJB_OK) return st;
	st = gpu_tabular(ctx, holder->dev, res, shadow, out);
	djb_brdf_destroy(holder);
	return st;
}
DJB_ABI_CATCH

// djb::tabular_anisotropic(brdf, elevation_res, azimuthal_res, shadow), dj_brdf.h:2238-2273
static djb_status gpu_tabular_aniso(djb_ctx *ctx, const Brdf &src_dev, int elev, int azim, int shadow, djb_brdf **out)
{
	djb_status st;
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
	if (e == hipSuccess) e = djbk::launch_fit_aniso(ctx->stream, src_dev, std_p, S, shadow != 0);
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
	return gpu_tabular_aniso(ctx, src->dev, elev, azim, shadow, out);
}
DJB_ABI_CATCH

// tabular_anisotropic of a user-defined source, dj_brdf.h:2238-2273 with brdf.eval sampled by the caller at djb_fit_aniso_query_dirs
djb_status djb_brdf_create_tabular_anisotropic_from_samples(djb_ctx *ctx, int elev, int azim, int shadow, const float *rgb,
                                                            int64_t count, djb_brdf **out)
try {
	if (!ctx || !rgb || !out) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null argument");
	if (elev <= 1 || azim <= 1 || elev > 1024 || azim > 1024)
		return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: Invalid Resolution");           // dj_brdf.h:2244
	if (count != djbcpu::fit_aniso_query_count(elev, azim))
		return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: a %d x %d fit takes %d samples, got %lld", elev, azim, djbcpu::fit_aniso_query_count(elev, azim), (long long)count);
	if (is_cpu(ctx)) return djbcpu::create_tabular_anisotropic_from_samples(ctx, elev, azim, shadow, rgb, out);
	djb_status st = check_call(ctx, nullptr, 0, DJB_MEM_DEVICE);
	if (st != DJB_OK) return st;
	std::lock_guard<std::recursive_mutex> call_lock(ctx->call_mu);
	djb_brdf *holder = nullptr;
	if ((st = upload_samples(ctx, rgb, count, &holder)) != DJB_OK) return st;
	st = gpu_tabular_aniso(ctx, holder->dev, elev, azim, shadow, out);
	djb_brdf_destroy(holder);
	return st;
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

} // extern "C"
