// djb_host.hip -- the C ABI of libdjb_hip.so (include/djb_hip.h), part 1: error reporting, microfacet::params
// resolution, handle lifetime -- contexts, BRDF objects and their host twins -- and the context options.  On a GPU
// context every batch runs on the gfx950 kernels (or fails with DJB_ERR_NO_DEVICE / DJB_ERR_HIP: there is no silent
// fallback); the two uses of the product's host instantiation of the same per-unit code (djb_cpu.cpp) are explicit: a
// CPU context (djb_ctx_create(DJB_DEVICE_CPU)) and scalar-size DJB_MEM_HOST calls (<= DJB_SCALAR_HOST_MAX units).
// The operator surface lives in djb_host_ops.hip, the fit drivers in djb_host_fit.hip (shared internals: djb_host.hpp).
#include "djb_host.hpp"

namespace djbh {

thread_local std::string g_err;

djb_status fail(djb_status st, const char *fmt, ...)
{
	char buf[256];   // same 256-byte budget as djb::exc (dj_brdf.h:578-587)
	va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
	g_err = buf;
	return st;
}

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
	if (in && (in->kind & DJB_PARAMS_RESOLVED_FOLLOWS) && DJB_PARAMS_KIND(in->kind) != DJB_PARAMS_LAMBERT) {
		*p = reinterpret_cast<const djb_params_cached *>(in)->r;       // resolved once by djb_params_resolve (include/djb_hip.h)
		// the flag is a promise about the bytes behind `in`; a stray 0x100 in a plain djb_params must not turn into silently wrong
		// parameters: what djb_params_resolve writes always satisfies these (three compares, no arithmetic)
		if (!(p->ax > 0.0f && p->ay > 0.0f && p->rho > -1.0f && p->rho < 1.0f))
			return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: DJB_PARAMS_RESOLVED_FOLLOWS is set but no resolved parameter set follows the djb_params");
		return DJB_OK;
	}
	memset(p, 0, sizeof *p);
	int kind = in ? DJB_PARAMS_KIND(in->kind) : DJB_PARAMS_STANDARD;
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

djb_status device_params(const djb_params *in, Params *out, int brdf_kind, bool want_reciprocals)
{
	const int in_kind = in ? DJB_PARAMS_KIND(in->kind) : DJB_PARAMS_STANDARD;
	// lambert::params(reflectance) (dj_brdf.h:114-119, 861-868): carried to the kernel in the n slot
	if (brdf_kind == DJB_KIND_LAMBERT) {
		if (in && in_kind != DJB_PARAMS_STANDARD && in_kind != DJB_PARAMS_LAMBERT)
			return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: a lambert brdf takes lambert::params");
		memset(out, 0, sizeof *out);
		const bool has = in && in_kind == DJB_PARAMS_LAMBERT;
		out->nx = has ? in->v[0] : 1.0f; out->ny = has ? in->v[1] : 1.0f; out->nz = has ? in->v[2] : 1.0f;
		return DJB_OK;
	}
	if (in && in_kind == DJB_PARAMS_LAMBERT)
		return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: lambert::params passed to a brdf that is not a lambert");
	djb_params_resolved r;
	djb_status st = resolve_params(in, &r);
	if (st != DJB_OK) return st;
	out->nx = r.n[0]; out->ny = r.n[1]; out->nz = r.n[2];
	out->ax = r.ax; out->ay = r.ay; out->rho = r.rho; out->s = r.sqrt_one_minus_rho_sqr;
	out->tx = r.tx_n; out->ty = r.ty_n;
	// reciprocals of the two launch-uniform denominators of mf_p22 (djb_device.hpp: fdiv_r): correctly rounded doubles
	// of exactly the floats the kernel divides by (this TU is built with -ffp-contract=off: no FMA in ax * ay * s)
	out->r_ax = out->r_t2 = 0.0;
	if (!want_reciprocals) return DJB_OK;                        // the host path divides
	const float t2 = out->ax * out->ay * out->s;
	out->r_ax = 1.0 / (double)out->ax;
	out->r_t2 = 1.0 / (double)t2;
	if (!(std::fabs(out->r_ax) <= 1e300)) out->r_ax = 0.0;      // ax == 0 / NaN: leave it to the IEEE division
	if (!(std::fabs(out->r_t2) <= 1e300)) out->r_t2 = 0.0;
	return DJB_OK;
}

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

djb_fresnel_desc current_fresnel_desc(const djb_brdf *b)
{
	djb_fresnel_desc d;
	memset(&d, 0, sizeof d);
	d.kind = b->dev.fr.kind;
	for (int c = 0; c < 3; ++c) { d.a[c] = b->dev.fr.a[c]; d.b[c] = b->dev.fr.b[c]; }
	if (d.kind == DJB_FRESNEL_SPLINE) { d.points = b->fresnel.data(); d.npoints = b->dev.fr.npts; }
	return d;
}

// `via`: the context of the CALL that needs the twin (validated by the caller: same device as the object).  The object's
// creating context may be gone by now -- handles outlive their contexts -- so nothing here touches b->ctx.
void build_twin(const djb_brdf *b, djb_ctx *via)
{
	djb_ctx *tc = djbcpu::twin_ctx();
	djb_brdf *t = nullptr;
	djb_status st = DJB_ERR_NOT_IMPLEMENTED;
	const djb_fresnel_desc fd = current_fresnel_desc(b);
	auto download = [&](std::vector<char> &host, const void *dev, size_t bytes) -> bool {
		host.resize(bytes);
		std::lock_guard<std::recursive_mutex> call_lock(via->call_mu);
		if (hipSetDevice(via->device) != hipSuccess) return false;
		if (hipMemcpyAsync(host.data(), dev, bytes, hipMemcpyDeviceToHost, via->stream) != hipSuccess) { (void)hipGetLastError(); return false; }
		return hipStreamSynchronize(via->stream) == hipSuccess;
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
// (a brdf / ctx device mismatch is NOT answered here: the caller's check_call reports it, as for every other call.)
// Concurrency: any number of threads may call the operators of one object at once -- they are const in the reference and
// lock-free here -- but djb_brdf_set_fresnel / set_shadow replace tables of the twin that such readers may be walking: as
// with the reference's microfacet::set_fresnel (delete + copy, dj_brdf.h:1521-1525) a setter must not run concurrently
// with calls on the same object.
const djb_brdf *scalar_twin(const djb_ctx *ctx, const djb_brdf *b, long long n, int mem)
{
	if (mem != DJB_MEM_HOST || n > ctx->host_batch_max || n < 0 || !b || ctx->scalar_on_device) return nullptr;
	if (b->device != ctx->device) return nullptr;
	if (!b->twin_built.load(std::memory_order_acquire)) {
		std::call_once(b->twin_once, build_twin, b, const_cast<djb_ctx *>(ctx));       // returns once the twin exists, whoever built it
		b->twin_built.store(1, std::memory_order_release);
	}
	return b->twin;
}

// CPU context: both operands must belong to it
djb_status cpu_pair_check(const djb_ctx *ctx, const djb_brdf *b)
{
	if (b && is_cpu(ctx) != is_cpu(b))
		return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: brdf and ctx belong to different back ends (CPU / GPU)");
	return DJB_OK;
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

} // namespace djbh

using namespace djbh;

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
	// DJB_SCALAR_ON_DEVICE=1: the initial value of DJB_OPT_SCALAR_ON_DEVICE for every GPU context of the process, so that
	// programs written against the C++ facade (which never call djb_ctx_set_option) can be A/B-tested on the kernel path
	if (const char *e = getenv("DJB_SCALAR_ON_DEVICE")) c->scalar_on_device = atoi(e) != 0;
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
	if (ctx->loader_state && ctx->loader_state_free) ctx->loader_state_free(ctx->loader_state);
	for (auto &kv : ctx->fit_fresnel_dirs) (void)hipFree(kv.second);
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

// a microfacet BRDF around a user-defined NDF: host code of the caller, so a CPU-context object (include/djb_hip.h)
djb_status djb_brdf_create_user_microfacet(djb_ctx *ctx, const djb_user_ndf *ndf, const djb_fresnel_desc *f, int shadow, djb_brdf **out)
try {
	if (!ctx || !ndf || !out) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null argument");
	if (!is_cpu(ctx)) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: a user-defined NDF is host code: create the object on a CPU context (DJB_DEVICE_CPU)");
	return djbcpu::create_user_microfacet(ctx, ndf, f, shadow, out);
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
	// sgd: the device copy carries, behind the row, the constants of the decided fast tier (djb_fast_models.inc: logarithms, a reciprocal and
	// the bound's coefficients per channel; [33] = 0 when the row is outside that tier's domain -- the kernels then run the exact chains only)
	double ext[djbdev::SGD_FAST_ROW];
	const double *src = row;
	int n_dev = count;
	if (kind == DJB_KIND_SGD) {
		(void)djbdev::sgd_fast_row(row, ext); src = ext; n_dev = djbdev::SGD_FAST_ROW;
		const char *ev = getenv("DJB_SGD_FAST");                 // "0": objects created from here on run the exact chains only (tests, A/B timing)
		if (ev && ev[0] == '0') ext[djbdev::SGD_FAST_FLAG] = 0.0;
	}
	double *d = nullptr;
	hipError_t e = hipMalloc((void **)&d, sizeof(double) * n_dev);
	if (e == hipSuccess) e = hipMemcpy(d, src, sizeof(double) * n_dev, hipMemcpyHostToDevice);
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
	// The payload was complete (stream synchronised) when the constructor returned and is immutable since: a blocking copy
	// on the object's device needs neither a stream nor a lock -- and must not touch b->ctx, which may have been destroyed
	// (handles outlive their creating context, :141)
	HIP_TRY(hipSetDevice(b->device));
	HIP_TRY(hipMemcpy(out, b->raw_samples, sizeof(double) * (size_t)b->raw_count, hipMemcpyDeviceToHost));
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
	return k == DJB_KIND_BECKMANN || k == DJB_KIND_GGX || k == DJB_KIND_TABULAR || k == DJB_KIND_TABULAR_ANISO || k == DJB_KIND_USER;
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
	HIP_TRY(hipSetDevice(b->device));          // not b->ctx->device: the creating context may be gone (:141)
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

djb_status djb_brdf_get_fresnel(const djb_brdf *b, djb_fresnel_desc *out)
try {
	if (!b || !out) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null argument");
	if (is_cpu(b)) return djbcpu::get_fresnel(b, out);
	const int k = b->dev.kind;
	if (!is_microfacet_kind(k) && k != DJB_KIND_SGD && k != DJB_KIND_ABC) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: this brdf has no Fresnel term");
	const djbdev::Fresnel &fr = b->dev.fr;
	memset(out, 0, sizeof *out);
	out->kind = fr.kind;
	for (int c = 0; c < 3; ++c) { out->a[c] = fr.a[c]; out->b[c] = fr.b[c]; }
	if (fr.kind == djbdev::FR_SPLINE) { out->points = b->fresnel.data(); out->npoints = fr.npts; }      // the host copy of the table
	return DJB_OK;
}
DJB_ABI_CATCH

djb_status djb_ctx_set_option(djb_ctx *ctx, int option, int value)
try {
	if (is_cpu(ctx)) return DJB_OK;           // the options select GPU code paths
	if (ctx && option == DJB_OPT_SCALAR_ON_DEVICE) { ctx->scalar_on_device = value != 0; return DJB_OK; }
	if (ctx && option == DJB_OPT_FIT_FILES_DENSE) { ctx->fit_files_dense = value != 0; return DJB_OK; }
	if (ctx && option == DJB_OPT_HOST_BATCH_MAX) { ctx->host_batch_max = value < 0 ? 0 : value > 65536 ? 65536 : value; return DJB_OK; }
	if (!ctx) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null ctx");
	if (option == DJB_OPT_MERL_EXACT_ONLY) { ctx->merl_exact_only = value != 0; return DJB_OK; }
	if (option == DJB_OPT_ANISO_QF2_ALIGNED) { ctx->aniso_qf2_aligned = value != 0; return DJB_OK; }
	if (option == DJB_OPT_UTIA_EXACT_ONLY) { ctx->utia_exact_only = value != 0; return DJB_OK; }
	if (option == DJB_OPT_CONTRACT_1E5) {
		ctx->contract_1e5 = value != 0;
		// the "hopeless lobe" verdict does not outlive a toggle -- nor does the key of a note still in flight, which would re-arm it
		ctx->ct_key = 0; ctx->ct_key_share = 0.0; ctx->ct_hopeless_calls = 0; ctx->wl_note_key = 0;
		return DJB_OK;
	}
	if (option == DJB_OPT_TEST_WORKLIST_CAP) { ctx->test_worklist_cap = value; return DJB_OK; }
	return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: unknown option %d", option);
}
DJB_ABI_CATCH

djb_status djb_params_resolve(const djb_params *params, djb_params_resolved *out)
try {
	if (!out) return fail(DJB_ERR_INVALID_ARGUMENT, "djb_error: null argument");
	if (!params) return resolve_params(nullptr, out);
	djb_params plain = *params;                                   // always computed here: this is where a cached form comes from
	plain.kind = DJB_PARAMS_KIND(plain.kind);
	return resolve_params(&plain, out);
}
DJB_ABI_CATCH

} // extern "C"

// ---------------------------------------------------------------- hooks for djb_loader.hip
namespace djbk {

djb_status resolve_device_params(const djb_params *in, float out9[9], int brdf_kind)
{
	Params p;
	djb_status st = device_params(in, &p, brdf_kind, false);
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
void **ctx_loader_state(djb_ctx *ctx, void (*free_fn)(void *)) { ctx->loader_state_free = free_fn; return &ctx->loader_state; }

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
