// include/djb_hip.hpp -- header-only C++ facade: `namespace djb` over the C ABI of libdjb_hip.so.
//
// Same class names, method names, argument meaning and error behaviour as the reference's public
// surface (jdupuy/dj_brdf, dj_brdf.h:41-537) for the hot path, so code written against
// `djb::brdf::eval/evalp/pdf/sample/evalp_is`, `djb::merl`, `djb::utia`, `djb::beckmann`,
// `djb::ggx`, `djb::tabular`, `djb::microfacet::params`, `djb::fresnel::*` recompiles against this
// header (link with -ldjb_hip).  Two differences, both additive:
//   * every operator also has a BATCH overload (n pairs per call) -- the form that makes sense
//     on a GPU.  The scalar virtuals are batches of one, which the library answers on the calling
//     thread from a host twin of the object (no launch, no lock: 60-200 ns per call);
//     a renderer that can gather a wavefront of intersections should still call the batch form.
//   * objects live on a djb::hip::context (one GPU + one stream); a process-wide default exists.
//   * the reference's EXTENSION POINTS are kept: `class brdf` has a public default constructor and `eval` as its one pure
//     virtual (dj_brdf.h:74-109), `fresnel::impl` has `eval` and `copy` (dj_brdf.h:157-162).  A class a user derives from
//     either is host code; the library never sees it.  Such an object works wherever the reference accepts it: its other
//     operators are the base-class defaults of dj_brdf.h:795-845, tabular / tabular_anisotropic fit it by sampling its
//     eval() on the host at the fit's query directions (djb_fit_query_dirs) and running the fit kernels on the samples,
//     and a microfacet BRDF holding a user-defined Fresnel term evaluates D G on the GPU and multiplies by the user's
//     F(cos theta_d) on the host ("user-defined classes").
// Errors: constructors and calls throw djb::exc carrying the library's djb_error message.
// All arithmetic of the library's own classes runs in the HIP kernels; this header only holds the reference's vec3
// helpers and the one-line base-class compositions a user-derived object needs.
#ifndef DJB_HIP_HPP
#define DJB_HIP_HPP

#include <cmath>
#include <cstdarg>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <string>
#include <vector>
#if __cplusplus >= 201103L
#	include <atomic>
#	include <mutex>          /* the slot a user callback's exception waits in (class brdf) */
#endif

#include "djb_hip.h"

/* Configuration macros of the reference header (dj_brdf.h:7-12, 43-48, 552-560), honoured here:
 *   DJB_USE_DOUBLE_PRECISION  the kernels implement the reference's DEFAULT configuration -- float storage with the
 *                             reference's double sub-expressions kept -- so a double float_t facade over them would
 *                             silently return float-precision values: refused at compile time instead.
 *   DJB_ASSERT(x)             user-overridable.  The reference asserts on invalid ARGUMENTS (radii, correlation, indices
 *                             of refraction, resolutions, variates: dj_brdf.h:1257, 1453, 1466-1467, 2003, 2173, 2218, 2244,
 *                             2780); a user macro is invoked at the same sites with the same conditions before the call
 *                             reaches the library.  Without one the library's own validation throws djb::exc (the
 *                             reference would abort in assert.h or, under NDEBUG, compute garbage).
 *   DJB_EPSILON               see below: only the reference's default value is supported.
 *   DJB_LOG(fmt, ...)         user-overridable, default stdout like the reference.  Unless NVERBOSE is defined the
 *                             constructors of tabular / tabular_anisotropic print the reference's progress lines
 *                             (dj_brdf.h:2383-2384, 2429-2430, 2638-2639, 2698-2699, 2724-2725, 2759-2760).  The two lines that carry a value
 *                             internal to the fit (the slope-pdf normalisation constants, :2301, :2335) and the
 *                             per-sample chatter of merl::eval / utia::normalize (:1017, :1166) are not reproduced.   */
#if defined(DJB_USE_DOUBLE_PRECISION) && DJB_USE_DOUBLE_PRECISION
#error "dj_brdf_amd: DJB_USE_DOUBLE_PRECISION=1 is not supported -- the MI355X kernels implement the reference's default single-precision configuration (float_t = float); see include/djb_hip.hpp"
#endif
#ifdef DJB_ASSERT
#	define DJB_USER_ASSERT(x) DJB_ASSERT(x)
#else
#	define DJB_USER_ASSERT(x) ((void)0)      /* the library validates the same conditions and the facade throws djb::exc */
#	include <cassert>
#	define DJB_ASSERT(x) assert(x)          /* dj_brdf.h:552-554: user code that uses DJB_ASSERT itself keeps its assertions */
#endif
#ifndef DJB_LOG
#	define DJB_LOG(format, ...) fprintf(stdout, format, ##__VA_ARGS__)
#endif
/* DJB_EPSILON (dj_brdf.h:49-51, used once: microfacet::ndf is zero unless h.z > DJB_EPSILON, :1561).  The kernels are built for the
 * reference's default, 1e-4; a program that defines another value before including the header is refused when it creates its first
 * context (a float cannot be compared by the preprocessor), instead of silently getting the default's results.  M_PI: dj_brdf.h:562. */
#ifndef DJB_EPSILON
#	define DJB_EPSILON (float_t)1e-4
#endif
#ifndef M_PI
#	define M_PI 3.1415926535897932384626433832795
#endif

namespace djb {

typedef float float_t;                      // dj_brdf.h:44-48 (single precision build)

/* Exception API, dj_brdf.h:54-59 */
struct exc : public std::exception {
	// the reference's constructor: a printf-style message, 255 characters kept (dj_brdf.h:55, 578-587)
	exc(const char *fmt, ...) __attribute__((format(printf, 2, 3))) : m_status(0)
	{
		char buf[256];
		va_list args;
		va_start(args, fmt);
		vsnprintf(buf, sizeof buf, fmt, args);
		va_end(args);
		m_str = buf;
	}
	// what the facade throws when the library reports an error: its message as it is, and the djb_status (an extension)
	exc(djb_status status, const std::string &msg) : m_str(msg), m_status((int)status) {}
	virtual ~exc() throw() {}
	const char *what() const throw() { return m_str.c_str(); }
	std::string m_str;
	int m_status;
};

/* Standalone vec3 utility, dj_brdf.h:62-71 (storage only; algebra happens on the device) */
struct vec3 {
	static vec3 from_raw(const double *v) { return vec3((float_t)v[0], (float_t)v[1], (float_t)v[2]); }
	static vec3 from_raw(const float *v) { return vec3(v[0], v[1], v[2]); }
	static const float_t *to_raw(const vec3 &v) { return &v.x; }
	explicit vec3(float_t s = 0) : x(s), y(s), z(s) {}
	vec3(float_t x_, float_t y_, float_t z_) : x(x_), y(y_), z(z_) {}
	explicit vec3(float_t theta, float_t phi)   // dj_brdf.h:589-595 (double libm, rounded where the reference rounds)
	{
		float_t s = (float_t)std::sin((double)theta);
		x = (float_t)((double)s * std::cos((double)phi));
		y = (float_t)((double)s * std::sin((double)phi));
		z = (float_t)std::cos((double)theta);
	}
	float_t intensity() const { return (float_t)0.2126 * x + (float_t)0.7152 * y + (float_t)0.0722 * z; }
	float_t x, y, z;
};

/* vec3 algebra of dj_brdf.h:597-637: host-side scalar helpers with the reference's float/double order
 * (user code such as tests/nrm_utia.cpp accumulates evalp() results with them) */
inline vec3 operator*(float_t a, const vec3 &b) { return vec3(a * b.x, a * b.y, a * b.z); }
inline vec3 operator*(const vec3 &a, float_t b) { return vec3(b * a.x, b * a.y, b * a.z); }
inline vec3 operator/(const vec3 &a, float_t b) { return (float_t)(1.0 / (double)b) * a; }
inline vec3 operator*(const vec3 &a, const vec3 &b) { return vec3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline vec3 operator/(const vec3 &a, const vec3 &b) { return vec3(a.x / b.x, a.y / b.y, a.z / b.z); }
inline vec3 operator+(const vec3 &a, const vec3 &b) { return vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline vec3 operator-(const vec3 &a, const vec3 &b) { return vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline vec3 &operator+=(vec3 &a, const vec3 &b) { a.x += b.x; a.y += b.y; a.z += b.z; return a; }
inline vec3 &operator*=(vec3 &a, const vec3 &b) { a.x *= b.x; a.y *= b.y; a.z *= b.z; return a; }
inline vec3 &operator*=(vec3 &a, float_t b) { a.x *= b; a.y *= b; a.z *= b; return a; }
inline float_t dot(const vec3 &a, const vec3 &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline vec3 cross(const vec3 &a, const vec3 &b) { return vec3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
inline vec3 normalize(const vec3 &v) { return (float_t)(1.0 / std::sqrt((double)dot(v, v))) * v; }
/* utility API, dj_brdf.h:574-576, 612-616, 639-645 (user-derived classes written against the reference use them) */
template <typename T> static T min(const T &a, const T &b) { return a < b ? a : b; }
template <typename T> static T max(const T &a, const T &b) { return a > b ? a : b; }
template <typename T> static T sat(const T &x) { return min(T(1), max(T(0), x)); }
template <typename T> static T max3(const T &x, const T &y, const T &z) { T m = x; if (m < y) m = y; if (m < z) m = z; return m; }
inline float_t inversesqrt(float_t x) { return (float_t)(1.0 / std::sqrt((double)x)); }
/* The other file-static helpers of the reference's implementation section (dj_brdf.h:650-765), visible to every program that defines
 * DJ_BRDF_IMPLEMENTATION: a user-defined lobe's own sample() or NDF may call them.  Answered by the library (djb_helper: the
 * arithmetic its operators use); the ABI status is checked further down (hip::check is not declared yet here). */
namespace hip { inline void helper(int which, const float *in, float *out); }
inline float_t erf(float_t x) { float_t r; hip::helper(DJB_HELPER_ERF, &x, &r); return r; }
inline float_t erfinv(float_t u) { float_t r; hip::helper(DJB_HELPER_ERFINV, &u, &r); return r; }
inline void xyz_to_theta_phi(const vec3 &p, float_t *theta, float_t *phi)
{ const float_t in[3] = { p.x, p.y, p.z }; float_t out[2]; hip::helper(DJB_HELPER_XYZ_TO_THETA_PHI, in, out); *theta = out[0]; *phi = out[1]; }
inline void uniform_to_concentric(float_t u1, float_t u2, float_t *x, float_t *y)
{ const float_t in[2] = { u1, u2 }; float_t out[2]; hip::helper(DJB_HELPER_UNIFORM_TO_CONCENTRIC, in, out); *x = out[0]; *y = out[1]; }
inline vec3 rotate_vector(const vec3 &x, const vec3 &axis, float_t angle)
{ const float_t in[7] = { x.x, x.y, x.z, axis.x, axis.y, axis.z, angle }; float_t out[3]; hip::helper(DJB_HELPER_ROTATE_VECTOR, in, out); return vec3(out[0], out[1], out[2]); }
/* the private linear-table helpers, dj_brdf.h:1181-1249: plain host templates (they work on the caller's own std::vector) */
namespace spline {
typedef int (*uwrap_callback)(int, int);
inline int uwrap_repeat(int i, int edge) { i %= edge; return i < 0 ? i + edge : i; }
inline int uwrap_edge(int i, int edge) { return i >= edge ? edge - 1 : (i < 0 ? 0 : i); }
template <typename T> T lerp(const T &x1, const T &x2, float_t u) { return x1 + u * (x2 - x1); }
// node and fraction of u * n - u, both formed in float as the reference does
inline float_t locate(float_t u, int n, int *whole) { double w; const float_t f = (float_t)modf((double)(u * n - u), &w); *whole = (int)w; return f; }
template <typename T> static T eval(const std::vector<T> &points, uwrap_callback wrap, float_t u)
{
	const int n = (int)points.size();
	int k; const float_t f = locate(u, n, &k);
	return lerp(points[wrap(k, n)], points[wrap(k + 1, n)], f);
}
template <typename T> static T eval2d(const std::vector<T> &points, int w, int h, uwrap_callback wrap1, float_t u1, uwrap_callback wrap2, float_t u2)
{
	int k1, k2;
	const float_t f1 = locate(u1, w, &k1), f2 = locate(u2, h, &k2);
	const int i1 = wrap1(k1, w), i2 = wrap1(k1 + 1, w), j1 = wrap2(k2, h), j2 = wrap2(k2 + 1, h);
	const float_t lo = lerp(points[i1 + w * j1], points[i2 + w * j1], f1), hi = lerp(points[i1 + w * j2], points[i2 + w * j2], f1);
	return lerp(lo, hi, f2);
}
} // namespace spline

namespace hip {

inline void check(djb_status st)
{
	if (st != DJB_OK) throw exc(st, djb_last_error());
}
inline void helper(int which, const float *in, float *out) { check(djb_helper(which, in, out)); }

/* one GPU + one HIP stream -- or, with device == DJB_DEVICE_CPU, the library's host execution path */
class context {
public:
	explicit context(int device = 0) : m_ctx(NULL) { abi_check(); check(djb_ctx_create(device, &m_ctx)); }
	context(int device, void *hip_stream) : m_ctx(NULL) { abi_check(); check(djb_ctx_create_on_stream(device, hip_stream, &m_ctx)); }
	// the loaded libdjb_hip.so must have the ABI major this header was written against (djb_hip.h: DJB_HIP_VERSION)
	static void abi_check()
	{
		if (!((float_t)(DJB_EPSILON) == (float_t)1e-4))
			throw exc(DJB_ERR_INVALID_ARGUMENT, "djb_error: DJB_EPSILON was redefined; libdjb_hip.so is built for the reference's default (1e-4)");
		if (DJB_HIP_VERSION_MAJOR(djb_version()) != DJB_HIP_VERSION_MAJOR(DJB_HIP_VERSION)) {
			char msg[160];
			snprintf(msg, sizeof msg, "djb_error: libdjb_hip.so has ABI version %d, this program was compiled against %d", djb_version(), (int)DJB_HIP_VERSION);
			throw exc(DJB_ERR_INVALID_ARGUMENT, msg);
		}
	}
	~context() { djb_ctx_destroy(m_ctx); }
	djb_ctx *get() const { return m_ctx; }
	void synchronize() const { check(djb_ctx_synchronize(m_ctx)); }
	// The numerical contract of this context's BATCH calls.  Default (false): every value is the reference's, bit
	// for bit.  true = DJB_OPT_CONTRACT_1E5: values within 1e-5 relative of the reference's (sampled directions: every component
	// within 1e-5), zeros and NaNs exactly where the reference has them -- the north star's tolerance, spent on speed (GGX eval+pdf
	// 0.53 -> 0.76 of the HBM roofline).  What changes bits under it: eval / evalp / pdf / eval_pdf batches of ggx and beckmann
	// (ideal, Schlick f0 >= 0.01, unpolarized ior >= 1.05; no mean-normal offset, |rho| <= 0.9), sgd and abc eval, sample and the
	// weights / pdfs of evalp_is of ggx and beckmann (evalp_is directions stay bit-identical), utia eval / evalp (the sRGB power of
	// the decode only: cells, weights and the 16-tap sums stay the reference's bits).  What never changes: MERL look-ups, every
	// MERL / UTIA bin index, tabular and tabular_anisotropic (fits, eval, sampling), lambert, the queries, LEAN / per-pair
	// parameter calls, every scalar (one-pair) call, and everything on a CPU context.
	void set_contract_1e5(bool on) { set_option(DJB_OPT_CONTRACT_1E5, on ? 1 : 0); }
	// host-array batches of up to `units` units are answered on the calling thread by the host twin (default DJB_SCALAR_HOST_MAX = 96):
	// a renderer submitting a few hundred pairs per call from several threads gains from ~512 (include/djb_hip.h, DJB_OPT_HOST_BATCH_MAX)
	void set_host_batch_max(int units) { set_option(DJB_OPT_HOST_BATCH_MAX, units); }
	void set_option(int option, int value) { check(djb_ctx_set_option(m_ctx, option, value)); }
	// The process-wide default context, used by every djb:: object that is not given one.  DJB_DEVICE=<n> selects GPU n,
	// DJB_DEVICE=cpu the host path.  Without the variable it is GPU 0 -- and, ONLY on a machine that has no HIP device at
	// all, the host path (announced once on stderr): the reference is a CPU library and its programs, compiled against
	// this header, have to run on such a machine too (BASELINE configs[0]).  A GPU that is present but failing is an
	// error, never a reason to fall back.
	static int standard_device()
	{
		const char *e = getenv("DJB_DEVICE");
		if (e && (!strcmp(e, "cpu") || !strcmp(e, "CPU"))) return DJB_DEVICE_CPU;
		if (e && *e) return atoi(e);
		if (device_count() > 0) return 0;
		if (!getenv("DJB_QUIET")) fprintf(stderr, "djb: no HIP device on this machine -- running on the host path (DJB_DEVICE=cpu)\n");
		return DJB_DEVICE_CPU;
	}
	static context &standard() { static context c(standard_device()); return c; }
	// the process-wide HOST-path context: where objects whose NDF is user code live (a GPU cannot call host code)
	static context &host() { static context c(DJB_DEVICE_CPU); return c; }
	static int device_count() { int n = 0; return djb_device_count(&n) == DJB_OK ? n : 0; }
private:
	context(const context &);
	context &operator=(const context &);
	djb_ctx *m_ctx;
};

/* merl / utia get_samples(): fetched from HBM on first use, cached in the object */
inline const std::vector<double> &fetch_samples(const djb_brdf *h, std::vector<double> &cache)
{
	if (cache.empty()) {
		int64_t n = 0;
		check(djb_brdf_get_samples(h, NULL, 0, &n));
		cache.resize((size_t)n);
		check(djb_brdf_get_samples(h, &cache[0], n, &n));
	}
	return cache;
}

/* view of an array of djb::vec3 (AoS, stride 3 floats) */
inline djb_vec3_view view(const vec3 *p)
{
	float *f = const_cast<float *>(&p->x);
	djb_vec3_view v = { f, f + 1, f + 2, 3 };
	return v;
}

} // namespace hip

/* BRDF interface, dj_brdf.h:74-109.
 * Two kinds of object implement it:
 *   * the library's classes below: RESIDENT in HBM behind a djb_brdf handle, every operator a call into libdjb_hip.so;
 *   * classes a user derives from djb::brdf (public default constructor, `eval` overridden): host code with no handle.
 *     Their other operators are the reference's base-class defaults (dj_brdf.h:795-845), their batch overloads loop over
 *     the scalar virtuals, and tabular / tabular_anisotropic fit them from host-side samples of eval().               */
class brdf {
public:
	// ---- the reference's virtuals.  evaluate f_r: the one pure virtual (dj_brdf.h:77-78)
	virtual vec3 eval(const vec3 &i, const vec3 &o, const void *user_param = NULL) const = 0;
	virtual vec3 eval_hd(const vec3 &h, const vec3 &d, const void *user_param = NULL) const            // dj_brdf.h:795-801
	{ vec3 i, o; hd_to_io(h, d, &i, &o); return eval(i, o, user_param); }
	virtual vec3 evalp(const vec3 &i, const vec3 &o, const void *user_param = NULL) const               // dj_brdf.h:803-806
	{
		if (!resident()) return eval(i, o, user_param) * i.z;
		vec3 r; evalp(1, &i, &o, &r, user_param); return r;
	}
	virtual vec3 evalp_hd(const vec3 &h, const vec3 &d, const void *user_param = NULL) const           // dj_brdf.h:808-814: eval * cos,
	{ vec3 i, o; hd_to_io(h, d, &i, &o); return eval(i, o, user_param) * i.z; }                        // also where evalp is overridden
	virtual vec3 evalp_is(float_t u1, float_t u2, const vec3 &o, vec3 *i, float_t *pdf,
	                      const void *user_param = NULL) const
	{
		if (!resident()) {                                                                               // dj_brdf.h:816-828
			const vec3 i_ = sample(u1, u2, o, user_param);
			float_t pdf_ = this->pdf(i_, o);
			if (i) *i = i_;
			if (pdf) *pdf = pdf_;
			return evalp(i_, o, user_param) / pdf_;
		}
		vec3 w, i_; float_t pdf_ = 0;
		evalp_is(1, &u1, &u2, &o, &w, &i_, &pdf_, user_param);
		if (i) *i = i_;
		if (pdf) *pdf = pdf_;
		return w;
	}
	virtual vec3 sample(float_t u1, float_t u2, const vec3 &o, const void *user_param = NULL) const    // dj_brdf.h:830-840
	{
		vec3 r; djb_vec3_view vo = hip::view(&o), vi = hip::view(&r);
		checked(djb_sample_batch(op_ctx(), op_handle(), 1, &u1, &u2, &vo, m_h ? params_of(user_param) : NULL, &vi, DJB_MEM_HOST));
		return r;
	}
	virtual float_t pdf(const vec3 &i, const vec3 &o, const void *user_param = NULL) const              // dj_brdf.h:842-845
	{
		float_t r = 0; djb_vec3_view vi = hip::view(&i), vo = hip::view(&o);
		checked(djb_pdf_batch(op_ctx(), op_handle(), 1, &vi, &vo, m_h ? params_of(user_param) : NULL, &r, DJB_MEM_HOST));
		return r;
	}
	static void io_to_hd(const vec3 &i, const vec3 &o, vec3 *h, vec3 *d)
	{
		djb_vec3_view vi = hip::view(&i), vo = hip::view(&o), vh = hip::view(h), vd = hip::view(d);
		hip::check(djb_io_to_hd_batch(hip::context::standard().get(), 1, &vi, &vo, &vh, &vd, DJB_MEM_HOST));
	}
	static void hd_to_io(const vec3 &h, const vec3 &d, vec3 *i, vec3 *o)
	{
		djb_vec3_view vh = hip::view(&h), vd = hip::view(&d), vi = hip::view(i), vo = hip::view(o);
		hip::check(djb_hd_to_io_batch(hip::context::standard().get(), 1, &vh, &vd, &vi, &vo, DJB_MEM_HOST));
	}

	// ---- batch overloads: n pairs, host arrays of djb::vec3.  A resident object runs them as one launch; a user-derived
	// object (host code) is called once per pair.
	void eval(size_t n, const vec3 *i, const vec3 *o, vec3 *out, const void *user_param = NULL) const
	{
		if (!resident()) { host_eval_batch(false, n, i, o, out, user_param); return; }
		djb_vec3_view vi = hip::view(i), vo = hip::view(o), vr = hip::view(out);
		checked(djb_eval_batch(ctx(), m_h, (int64_t)n, &vi, &vo, params_of(user_param), &vr, DJB_MEM_HOST));
	}
	void evalp(size_t n, const vec3 *i, const vec3 *o, vec3 *out, const void *user_param = NULL) const
	{
		if (!resident()) { host_eval_batch(true, n, i, o, out, user_param); return; }
		djb_vec3_view vi = hip::view(i), vo = hip::view(o), vr = hip::view(out);
		checked(djb_evalp_batch(ctx(), m_h, (int64_t)n, &vi, &vo, params_of(user_param), &vr, DJB_MEM_HOST));
	}
	void pdf(size_t n, const vec3 *i, const vec3 *o, float_t *out, const void *user_param = NULL) const
	{
		if (!overrides_resident_ops()) { for (size_t k = 0; k < n; ++k) out[k] = pdf(i[k], o[k], user_param); return; }
		djb_vec3_view vi = hip::view(i), vo = hip::view(o);
		checked(djb_pdf_batch(ctx(), m_h, (int64_t)n, &vi, &vo, params_of(user_param), out, DJB_MEM_HOST));
	}
	void sample(size_t n, const float_t *u1, const float_t *u2, const vec3 *o, vec3 *out_i,
	            const void *user_param = NULL) const
	{
		if (!overrides_resident_ops()) { for (size_t k = 0; k < n; ++k) out_i[k] = sample(u1[k], u2[k], o[k], user_param); return; }
		djb_vec3_view vo = hip::view(o), vi = hip::view(out_i);
		checked(djb_sample_batch(ctx(), m_h, (int64_t)n, u1, u2, &vo, params_of(user_param), &vi, DJB_MEM_HOST));
	}
	void evalp_is(size_t n, const float_t *u1, const float_t *u2, const vec3 *o, vec3 *out_weight,
	              vec3 *out_i, float_t *out_pdf, const void *user_param = NULL) const
	{
		if (!resident()) { host_evalp_is_batch(n, u1, u2, o, out_weight, out_i, out_pdf, user_param); return; }
		djb_vec3_view vo = hip::view(o), vw = hip::view(out_weight), vi = hip::view(out_i);
		checked(djb_evalp_is_batch(ctx(), m_h, (int64_t)n, u1, u2, &vo, params_of(user_param), &vw, &vi,
		                              out_pdf, DJB_MEM_HOST));
	}
	// ---- batch, device-resident (SoA or strided views in HBM; asynchronous on the context stream): resident objects only
	void eval_device(int64_t n, const djb_vec3_view &i, const djb_vec3_view &o, const djb_vec3_view &out,
	                 const void *user_param = NULL) const
	{ need_resident("eval_device"); checked(djb_eval_batch(ctx(), m_h, n, &i, &o, params_of(user_param), &out, DJB_MEM_DEVICE)); }
	void eval_pdf_device(int64_t n, const djb_vec3_view &i, const djb_vec3_view &o, const djb_vec3_view &out,
	                     float_t *out_pdf, bool cos = false, const void *user_param = NULL) const
	{ need_resident("eval_pdf_device"); checked(djb_eval_pdf_batch(ctx(), m_h, n, &i, &o, params_of(user_param), cos, &out, out_pdf, DJB_MEM_DEVICE)); }

	// NULL for a user-derived object; for a microfacet BRDF with a user-defined Fresnel term the handle of its D G part
	const djb_brdf *handle() const { return m_h; }
	// true: every operator of this object is answered by the library from the handle (kernels / host twin);
	// false: host code is involved (a user-derived class, a user-defined fresnel::impl) and fits sample eval() on the host
	bool resident() const { return m_h != NULL && !m_host_eval && !m_host_only; }
	// the context a fit of this object should run on: its own, unless it lives on the host path because its NDF is user code --
	// then its eval() is sampled there and the fit itself runs on the default context (the GPU, where there is one)
	hip::context &fit_context() const { return m_host_only ? hip::context::standard() : get_context(); }
	hip::context &get_context() const { return m_ctx ? *m_ctx : hip::context::standard(); }
	brdf() : m_h(NULL), m_ctx(NULL), m_host_eval(true), m_host_only(false) { init_exc(); }                            // dj_brdf.h:102
	virtual ~brdf() { djb_brdf_destroy(m_h); }
protected:
	explicit brdf(hip::context *c) : m_h(NULL), m_ctx(c ? c : &hip::context::standard()), m_host_eval(false), m_host_only(false) { init_exc(); }
	djb_ctx *ctx() const { return get_context().get(); }
	virtual const djb_params *params_of(const void *) const { return NULL; }   // ignored by merl/utia/...
	// eval of a resident object: what the library's classes override `eval` with
	vec3 eval_resident(const vec3 &i, const vec3 &o, const void *user_param) const
	{
		vec3 r; djb_vec3_view vi = hip::view(&i), vo = hip::view(&o), vr = hip::view(&r);
		checked(djb_eval_batch(ctx(), m_h, 1, &vi, &vo, params_of(user_param), &vr, DJB_MEM_HOST));
		return r;
	}
	// one-pair calls on the handle whatever resident() says (the D G part of a microfacet BRDF with a user-defined Fresnel term)
	vec3 handle_evalp(const vec3 &i, const vec3 &o, const void *user_param) const
	{
		vec3 r; djb_vec3_view vi = hip::view(&i), vo = hip::view(&o), vr = hip::view(&r);
		checked(djb_evalp_batch(ctx(), m_h, 1, &vi, &vo, params_of(user_param), &vr, DJB_MEM_HOST));
		return r;
	}
	vec3 handle_evalp_is(float_t u1, float_t u2, const vec3 &o, vec3 *i, float_t *pdf, const void *user_param) const
	{
		vec3 w; djb_vec3_view vo = hip::view(&o), vw = hip::view(&w), vi = hip::view(i);
		checked(djb_evalp_is_batch(ctx(), m_h, 1, &u1, &u2, &vo, params_of(user_param), &vw, &vi, pdf, DJB_MEM_HOST));
		return w;
	}
	// sample / pdf never involve the Fresnel term: a handle answers them even when eval is composed on the host
	bool overrides_resident_ops() const { return m_h != NULL; }
	// batches of an object evaluated by host code: one scalar virtual call per pair (microfacet overrides these: D G of the whole
	// batch in one library call, the user's Fresnel term per pair)
	virtual void host_eval_batch(bool cosine, size_t n, const vec3 *i, const vec3 *o, vec3 *out, const void *user_param) const
	{ for (size_t k = 0; k < n; ++k) out[k] = cosine ? evalp(i[k], o[k], user_param) : eval(i[k], o[k], user_param); }
	virtual void host_evalp_is_batch(size_t n, const float_t *u1, const float_t *u2, const vec3 *o, vec3 *out_weight, vec3 *out_i,
	                                 float_t *out_pdf, const void *user_param) const
	{ for (size_t k = 0; k < n; ++k) out_weight[k] = evalp_is(u1[k], u2[k], o[k], &out_i[k], &out_pdf[k], user_param); }
	void need_resident(const char *what) const
	{ if (!resident()) throw exc(DJB_ERR_INVALID_ARGUMENT, std::string("djb_error: ") + what + " needs a BRDF resident on the GPU (this object is evaluated by host code)"); }
	// An exception thrown by the CALLER'S code while the library was calling it back (a user-defined NDF: the cb_* trampolines of
	// microfacet / radial) must reach the caller as it is -- type and message -- as it does in the reference, where the call is direct.
	// It cannot travel through the C ABI: the trampoline keeps it here (and hands the library a NaN), and every library call made on
	// behalf of this object re-throws it when it returns (`checked`).  Callbacks may run on the library's worker threads: one slot per
	// object, first exception wins.  Before C++11 there is no std::exception_ptr: the exception then crosses the library and arrives as
	// a djb::exc carrying its message.
#if __cplusplus >= 201103L
	void stash_user_exception() const
	{ std::lock_guard<std::mutex> lock(m_exc_mu); if (!m_exc) { m_exc = std::current_exception(); m_exc_set.store(1, std::memory_order_release); } }
	void rethrow_user_exception() const
	{
		if (!m_exc_set.load(std::memory_order_acquire)) return;          // the common case: one relaxed-cost load, no lock (render threads share a BSDF)
		std::exception_ptr e;
		{ std::lock_guard<std::mutex> lock(m_exc_mu); e = m_exc; m_exc = std::exception_ptr(); m_exc_set.store(0, std::memory_order_release); }
		if (e) std::rethrow_exception(e);
	}
	mutable std::exception_ptr m_exc;
	mutable std::mutex m_exc_mu;
	mutable std::atomic<int> m_exc_set;
	void init_exc() { m_exc_set.store(0, std::memory_order_relaxed); }
#else
	void stash_user_exception() const {}
	void rethrow_user_exception() const {}
	void init_exc() {}
#endif
	void checked(djb_status st) const { rethrow_user_exception(); hip::check(st); }
	djb_brdf *m_h;
	hip::context *m_ctx;
	bool m_host_eval;        // eval involves host code beyond the handle (a user-derived brdf; a user-defined Fresnel term)
	bool m_host_only;        // the handle itself lives on the host path (a user-defined NDF: callbacks)
private:
	// the base-class sample / pdf (dj_brdf.h:830-845) are what djb::lambert inherits unchanged: a user-derived object
	// borrows them from a Lambertian on the default context
	static const djb_brdf *base_ops()
	{
		struct holder { djb_brdf *h; holder() : h(NULL) { hip::check(djb_brdf_create_lambert(hip::context::standard().get(), &h)); } ~holder() { djb_brdf_destroy(h); } };
		static holder b;
		return b.h;
	}
	const djb_brdf *op_handle() const { return m_h ? m_h : base_ops(); }
	djb_ctx *op_ctx() const { return m_h ? ctx() : hip::context::standard().get(); }
	// noncopyable, dj_brdf.h:104-108
	brdf(const brdf &);
	brdf &operator=(const brdf &);
};
/* what each of the library's classes declares: eval answered from the handle; the batch overloads stay visible */
#define DJB_HIP_RESIDENT_EVAL \
	using brdf::eval; \
	vec3 eval(const vec3 &i, const vec3 &o, const void *user_param = NULL) const { return eval_resident(i, o, user_param); }

namespace hip {
/* brdf.eval(i, o) of a host-evaluated source at the query slots of a fit, in the reference's call order
 * (dj_brdf.h:2494, 2610; 2545, 2671).  A slot its loops never reach (NaN direction) is not evaluated. */
inline std::vector<float_t> sample_source(const brdf &src, const std::vector<vec3> &qi, const std::vector<vec3> &qo)
{
	std::vector<float_t> rgb(3 * qi.size(), (float_t)0);
	for (size_t s = 0; s < qi.size(); ++s) {
		if (qo[s].x != qo[s].x) continue;
		const vec3 fr = src.eval(qi[s], qo[s]);
		rgb[3 * s] = fr.x; rgb[3 * s + 1] = fr.y; rgb[3 * s + 2] = fr.z;
	}
	return rgb;
}
} // namespace hip

/* Lambertian BRDF, dj_brdf.h:112-123 */
class lambert : public brdf {
public:
	/* Lambertian Parameters, dj_brdf.h:114-119: passed as `const void *user_param` */
	class params {
	public:
		params(const vec3 &reflectance = vec3(1)) : m_reflectance(reflectance) {}
		vec3 m_reflectance;
	};
	explicit lambert(hip::context *c = NULL) : brdf(c) { checked(djb_brdf_create_lambert(ctx(), &m_h)); }
	DJB_HIP_RESIDENT_EVAL
protected:
	const djb_params *params_of(const void *user_param) const   // dj_brdf.h:863-865
	{
		if (!user_param) return NULL;
		static __thread djb_params d;                              // lives across the ABI call of this thread
		const vec3 &r = reinterpret_cast<const params *>(user_param)->m_reflectance;
		d.kind = DJB_PARAMS_LAMBERT; d.v[0] = r.x; d.v[1] = r.y; d.v[2] = r.z; d.v[3] = d.v[4] = 0;
		return &d;
	}
};

/* MERL BRDF, dj_brdf.h:126-133 */
class merl : public brdf {
public:
	merl(const char *filename, hip::context *c = NULL) : brdf(c)
	{ checked(djb_brdf_create_merl_from_file(ctx(), filename, &m_h)); }
	merl(const double *samples, int64_t n_per_channel, hip::context *c = NULL) : brdf(c)
	{ checked(djb_brdf_create_merl_from_memory(ctx(), samples, n_per_channel, &m_h)); }
	const std::vector<double> &get_samples() const { return hip::fetch_samples(m_h, m_samples); }   // dj_brdf.h:132
	DJB_HIP_RESIDENT_EVAL
private:
	mutable std::vector<double> m_samples;
};

/* UTIA BRDF, dj_brdf.h:136-146 */
class utia : public brdf {
public:
	utia(const char *filename, hip::context *c = NULL) : brdf(c)
	{ checked(djb_brdf_create_utia_from_file(ctx(), filename, &m_h)); }
	const std::vector<double> &get_samples() const { return hip::fetch_samples(m_h, m_samples); }   // dj_brdf.h:143
	DJB_HIP_RESIDENT_EVAL
private:
	mutable std::vector<double> m_samples;
};

/* Fresnel API, dj_brdf.h:149-207 */
namespace fresnel {
	/* Utilities, dj_brdf.h:151-154, 1255-1290 (host-side scalar helpers, same float/double order) */
	inline void ior_to_f0(float_t ior, float_t *f0)
	{ float_t tmp = (float_t)(((double)ior - 1.0) / ((double)ior + 1.0)); *f0 = tmp * tmp; }
	inline void f0_to_ior(float_t f0, float_t *ior)
	{
		if ((double)f0 == 1.0) { *ior = (float_t)1.0; return; }
		float_t sqrt_f0 = (float_t)std::sqrt((double)f0);
		*ior = (float_t)((1.0 + (double)sqrt_f0) / (1.0 - (double)sqrt_f0));
	}
	inline void ior_to_f0(const vec3 &ior, vec3 *f0) { ior_to_f0(ior.x, &f0->x); ior_to_f0(ior.y, &f0->y); ior_to_f0(ior.z, &f0->z); }
	inline void f0_to_ior(const vec3 &f0, vec3 *ior) { f0_to_ior(f0.x, &ior->x); f0_to_ior(f0.y, &ior->y); f0_to_ior(f0.z, &ior->z); }
	/* dj_brdf.h:157-162.  The five terms below are fused into the eval kernels (desc() names them to the library).  A class
	 * a user derives from impl -- eval() and copy() overridden, as in the reference -- is host code: a microfacet BRDF that
	 * holds one keeps D and G on the GPU and calls the user's eval() on the host per pair (microfacet::evalp below). */
	class impl {
	public:
		virtual ~impl() {}
		virtual vec3 eval(float_t cos_theta_d) const = 0;
		virtual impl *copy() const = 0;
		// how the library runs this term; the default marks a term only the caller can evaluate
		virtual djb_fresnel_desc desc() const { djb_fresnel_desc d = djb_fresnel_desc(); d.kind = DJB_FRESNEL_HOST; return d; }
	protected:
		/* stand-alone evaluation of a term the library knows: through a temporary microfacet object on the default
		 * context (answered by its host twin) */
		vec3 eval_desc(float_t cos_theta_d) const
		{
			djb_fresnel_desc d = desc();
			djb_ctx *c = hip::context::standard().get();
			djb_brdf *h = NULL;
			hip::check(djb_brdf_create_ggx(c, &d, 1, &h));
			vec3 a(cos_theta_d, 0, 0), r;
			djb_vec3_view va = hip::view(&a), vr = hip::view(&r);
			djb_status st = djb_query_batch(c, h, DJB_Q_FRESNEL, 1, &va, NULL, NULL, NULL, &vr, DJB_MEM_HOST);
			djb_brdf_destroy(h);
			hip::check(st);
			return r;
		}
	};
	class ideal : public impl {
	public:
		vec3 eval(float_t cos_theta_d) const { return eval_desc(cos_theta_d); }
		impl *copy() const { return new ideal(); }
		djb_fresnel_desc desc() const { djb_fresnel_desc d = djb_fresnel_desc(); d.kind = DJB_FRESNEL_IDEAL; return d; }
	};
	class unpolarized : public impl {
		vec3 ior;
	public:
		unpolarized(const vec3 &ior_) : ior(ior_) { DJB_USER_ASSERT(ior.x > 0.0 && ior.y > 0.0 && ior.z > 0.0 && "Invalid ior"); }   // dj_brdf.h:1257
		vec3 eval(float_t cos_theta_d) const { return eval_desc(cos_theta_d); }
		impl *copy() const { return new unpolarized(*this); }
		djb_fresnel_desc desc() const
		{ djb_fresnel_desc d = djb_fresnel_desc(); d.kind = DJB_FRESNEL_UNPOLARIZED; d.a[0] = ior.x; d.a[1] = ior.y; d.a[2] = ior.z; return d; }
	};
	class schlick : public impl {
		vec3 f0;
	public:
		schlick(const vec3 &f0_) : f0(f0_) {}
		vec3 eval(float_t cos_theta_d) const { return eval_desc(cos_theta_d); }
		impl *copy() const { return new schlick(*this); }
		djb_fresnel_desc desc() const
		{ djb_fresnel_desc d = djb_fresnel_desc(); d.kind = DJB_FRESNEL_SCHLICK; d.a[0] = f0.x; d.a[1] = f0.y; d.a[2] = f0.z; return d; }
	};
	class sgd : public impl {
		vec3 f0, f1;
	public:
		sgd(const vec3 &f0_, const vec3 &f1_) : f0(f0_), f1(f1_) {}
		vec3 eval(float_t cos_theta_d) const { return eval_desc(cos_theta_d); }
		impl *copy() const { return new sgd(*this); }
		djb_fresnel_desc desc() const
		{
			djb_fresnel_desc d = djb_fresnel_desc(); d.kind = DJB_FRESNEL_SGD;
			d.a[0] = f0.x; d.a[1] = f0.y; d.a[2] = f0.z; d.b[0] = f1.x; d.b[1] = f1.y; d.b[2] = f1.z; return d;
		}
	};
	class spline : public impl {
		std::vector<vec3> m_points;
	public:
		explicit spline(const std::vector<vec3> &points) : m_points(points) {}
		const std::vector<vec3> &get_points() const { return m_points; }
		vec3 eval(float_t cos_theta_d) const { return eval_desc(cos_theta_d); }
		impl *copy() const { return new spline(*this); }
		djb_fresnel_desc desc() const
		{
			djb_fresnel_desc d = djb_fresnel_desc(); d.kind = DJB_FRESNEL_SPLINE;
			d.points = m_points.empty() ? NULL : &m_points[0].x; d.npoints = (int)m_points.size(); return d;
		}
	};
	/* the library's term behind a handle as one of the classes above (sgd::get_fresnel, abc::get_fresnel) */
	inline impl *from_handle(const djb_brdf *h)
	{
		djb_fresnel_desc d;
		hip::check(djb_brdf_get_fresnel(h, &d));
		switch (d.kind) {
		case DJB_FRESNEL_UNPOLARIZED: return new unpolarized(vec3(d.a[0], d.a[1], d.a[2]));
		case DJB_FRESNEL_SCHLICK: return new schlick(vec3(d.a[0], d.a[1], d.a[2]));
		case DJB_FRESNEL_SGD: return new sgd(vec3(d.a[0], d.a[1], d.a[2]), vec3(d.b[0], d.b[1], d.b[2]));
		case DJB_FRESNEL_SPLINE: {
			std::vector<vec3> pts;
			for (int k = 0; k < d.npoints; ++k) pts.push_back(vec3(d.points[3 * k], d.points[3 * k + 1], d.points[3 * k + 2]));
			return new spline(pts);
		}
		default: return new ideal();
		}
	}
} // namespace fresnel

/* Shifted Gamma Distribution BRDF, dj_brdf.h:481-511 (published per-material parameters) */
class sgd : public brdf {
public:
	sgd(const char *name, hip::context *c = NULL) : brdf(c), m_fresnel(NULL)
	{ checked(djb_brdf_create_sgd(ctx(), name, &m_h)); m_fresnel = fresnel::from_handle(m_h); }     // fresnel::sgd(f0, f1), dj_brdf.h:3443
	~sgd() { delete m_fresnel; }
	const fresnel::impl &get_fresnel() const { return *m_fresnel; }                                   // dj_brdf.h:510
	DJB_HIP_RESIDENT_EVAL
	vec3 ndf(const vec3 &h) const { return mq(DJB_Q_MODEL_NDF, h, NULL, NULL); }
	vec3 gaf(const vec3 &h, const vec3 &i, const vec3 &o) const { return mq(DJB_Q_MODEL_GAF, h, &i, &o); }
	vec3 g1(const vec3 &k) const { return mq(DJB_Q_MODEL_G1, k, NULL, NULL); }
	vec3 fresnel(float_t cos_theta_d) const { return mq(DJB_Q_FRESNEL, vec3(cos_theta_d, 0, 0), NULL, NULL); }
protected:
	vec3 mq(int which, const vec3 &a, const vec3 *b, const vec3 *c) const
	{
		vec3 r; djb_vec3_view va = hip::view(&a), vb = hip::view(b ? b : &a), vc = hip::view(c ? c : &a), vr = hip::view(&r);
		checked(djb_query_batch(ctx(), m_h, which, 1, &va, &vb, &vc, NULL, &vr, DJB_MEM_HOST));
		return r;
	}
private:
	const fresnel::impl *m_fresnel;
};

/* ABC Distribution BRDF, dj_brdf.h:514-535 */
class abc : public brdf {
public:
	abc(const char *name, hip::context *c = NULL) : brdf(c), m_fresnel(NULL)
	{ checked(djb_brdf_create_abc(ctx(), name, &m_h)); m_fresnel = fresnel::from_handle(m_h); }     // fresnel::unpolarized(vec3(ior)), dj_brdf.h:3623
	~abc() { delete m_fresnel; }
	const fresnel::impl &get_fresnel() const { return *m_fresnel; }                                   // dj_brdf.h:534
	DJB_HIP_RESIDENT_EVAL
	vec3 ndf(const vec3 &h) const { return mq(DJB_Q_MODEL_NDF, h, NULL, NULL); }
	float_t gaf(const vec3 &h, const vec3 &i, const vec3 &o) const { return mq(DJB_Q_MODEL_GAF, h, &i, &o).x; }
	vec3 fresnel(float_t cos_theta_d) const { return mq(DJB_Q_FRESNEL, vec3(cos_theta_d, 0, 0), NULL, NULL); }
protected:
	vec3 mq(int which, const vec3 &a, const vec3 *b, const vec3 *c) const
	{
		vec3 r; djb_vec3_view va = hip::view(&a), vb = hip::view(b ? b : &a), vc = hip::view(c ? c : &a), vr = hip::view(&r);
		checked(djb_query_batch(ctx(), m_h, which, 1, &va, &vb, &vc, NULL, &vr, DJB_MEM_HOST));
		return r;
	}
private:
	const fresnel::impl *m_fresnel;
};

/* Microfacet API, dj_brdf.h:210-298 */
class microfacet : public brdf {
public:
	/* microfacet parameters, dj_brdf.h:213-243.  Passed as `const void *user_param`. */
	class params {
	public:
		static params standard() { return params(); }
		static params isotropic(float_t a) { return elliptic(a, a, 0); }
		static params elliptic(float_t a1, float_t a2, float_t phi_a = 0.0)
		{ DJB_USER_ASSERT(a1 > 0.0 && a2 > 0.0 && "Invalid ellipse radii");                                   // dj_brdf.h:1453
		  params p; p.m_c.p.kind = DJB_PARAMS_ELLIPTIC; p.m_c.p.v[0] = a1; p.m_c.p.v[1] = a2; p.m_c.p.v[2] = phi_a; p.resolve(); return p; }
		static params pdfparams(float_t ax, float_t ay, float_t rho = 0.0, float_t tx_n = 0.0, float_t ty_n = 0.0)
		{
			DJB_USER_ASSERT(ax > 0.0 && ay > 0.0 && "Invalid scale parameters");                                  // dj_brdf.h:1466
			DJB_USER_ASSERT(std::fabs(rho) < 1.0 && "Invalid correlation parameter");                            // dj_brdf.h:1467
			params p; p.m_c.p.kind = DJB_PARAMS_PDFPARAMS;
			p.m_c.p.v[0] = ax; p.m_c.p.v[1] = ay; p.m_c.p.v[2] = rho; p.m_c.p.v[3] = tx_n; p.m_c.p.v[4] = ty_n;
			p.resolve(); return p;
		}
		void set_ellipse(float_t a1, float_t a2, float_t phi_a = 0.0)
		{ float_t tx = m_c.r.tx_n, ty = m_c.r.ty_n; *this = elliptic(a1, a2, phi_a); if (tx != 0 || ty != 0) set_location(tx, ty); }
		void set_pdfparams(float_t ax, float_t ay, float_t rho = 0.0, float_t tx_n = 0.0, float_t ty_n = 0.0)
		{ *this = pdfparams(ax, ay, rho, tx_n, ty_n); }
		void set_location(float_t tx_n, float_t ty_n) { *this = pdfparams(m_c.r.ax, m_c.r.ay, m_c.r.rho, tx_n, ty_n); }
		/* dj_brdf.h:1444-1449: tx = -n.x / n.z, ty = -n.y / n.z.  The kernels use the unit normal of
		 * (tx, ty); it agrees with `n` in direction whenever n.z > 0. */
		void set_location(const vec3 &n) { set_location(-n.x / n.z, -n.y / n.z); }
		void get_ellipse(float_t *a1, float_t *a2, float_t *phi_a = NULL) const
		{ if (a1) *a1 = m_c.r.a1; if (a2) *a2 = m_c.r.a2; if (phi_a) *phi_a = m_c.r.phi_a; }
		void get_pdfparams(float_t *ax, float_t *ay, float_t *rho = NULL, float_t *tx_n = NULL, float_t *ty_n = NULL) const
		{ if (ax) *ax = m_c.r.ax; if (ay) *ay = m_c.r.ay; if (rho) *rho = m_c.r.rho; if (tx_n) *tx_n = m_c.r.tx_n; if (ty_n) *ty_n = m_c.r.ty_n; }
		void get_location(float_t *tx_n, float_t *ty_n) const { if (tx_n) *tx_n = m_c.r.tx_n; if (ty_n) *ty_n = m_c.r.ty_n; }
		void get_location(vec3 *n) const { if (n) *n = vec3(m_c.r.n[0], m_c.r.n[1], m_c.r.n[2]); }
		params(float_t ax, float_t ay, float_t rho, float_t tx_n, float_t ty_n) { *this = pdfparams(ax, ay, rho, tx_n, ty_n); }   // dj_brdf.h:236
		params(float_t a1 = 1.0, float_t a2 = 1.0, float_t phi_a = 0.0)
		{ m_c.p.kind = DJB_PARAMS_ELLIPTIC; m_c.p.v[0] = a1; m_c.p.v[1] = a2; m_c.p.v[2] = phi_a; m_c.p.v[3] = m_c.p.v[4] = 0; resolve(); }
		const djb_params *desc() const { return &m_c.p; }
	private:
		// the set-up arithmetic runs once, here, as in the reference's factories; the calls read the result (djb_params_cached)
		void resolve()
		{
			m_c.p.kind = DJB_PARAMS_KIND(m_c.p.kind);
			hip::check(djb_params_resolve(&m_c.p, &m_c.r));              // DJB_ASSERT sites -> djb::exc
			m_c.p.kind |= DJB_PARAMS_RESOLVED_FOLLOWS;
		}
		djb_params_cached m_c;
	};

	// false for the two tabulated classes, which sample with the "nmap" scheme (dj_brdf.h:412, 439); a user-defined NDF class
	// overrides it (pure virtual in the reference, dj_brdf.h:274)
	virtual bool supports_smith_vndf_sampling() const
	{
		const int k = djb_brdf_kind(m_h);
		return k != DJB_KIND_TABULAR && k != DJB_KIND_TABULAR_ANISO;
	}
	int get_shadow() const { return djb_brdf_get_shadow(m_h); }
	void set_shadow(bool shadow) { checked(djb_brdf_set_shadow(m_h, shadow ? 1 : 0)); }          // dj_brdf.h:278
	void set_fresnel(const fresnel::impl &f)                                                         // dj_brdf.h:1521-1525
	{
		const fresnel::impl *copy = f.copy();
		bool host = false;
		djb_fresnel_desc d = resident_desc(*copy, &host);
		djb_status st = djb_brdf_set_fresnel(m_h, &d);
		if (st != DJB_OK) { delete copy; hip::check(st); }
		delete m_fresnel;
		m_fresnel = copy;
		m_host_eval = host;
	}
	const fresnel::impl &get_fresnel() const { return *m_fresnel; }
	virtual ~microfacet() { delete m_fresnel; }

	// eval / sampling queries (dj_brdf.h:258-276), scalar form = batch of one
	vec3 fresnel(float_t cos_theta_d) const                                                           // dj_brdf.h:258
	{ return m_host_eval ? m_fresnel->eval(cos_theta_d) : q3(DJB_Q_FRESNEL, vec3(cos_theta_d, 0, 0)); }
	float_t ndf(const vec3 &h, const params &p = params::standard()) const { return q(DJB_Q_NDF, &h, NULL, NULL, p); }
	float_t gaf(const vec3 &h, const vec3 &i, const vec3 &o, const params &p = params::standard()) const { return q(DJB_Q_GAF, &h, &i, &o, p); }
	float_t g1(const vec3 &h, const vec3 &k, const params &p = params::standard()) const { return q(DJB_Q_G1, &h, &k, NULL, p); }
	float_t sigma(const vec3 &k, const params &p = params::standard()) const { return q(DJB_Q_SIGMA, &k, NULL, NULL, p); }
	float_t p22(float_t x, float_t y, const params &p = params::standard()) const { vec3 a(x, y, 0); return q(DJB_Q_P22, &a, NULL, NULL, p); }
	float_t vp22(float_t x, float_t y, const vec3 &k, const params &p = params::standard()) const { vec3 a(x, y, 0); return q(DJB_Q_VP22, &a, &k, NULL, p); }
	float_t vndf(const vec3 &h, const vec3 &k, const params &p = params::standard()) const { return q(DJB_Q_VNDF, &h, &k, NULL, p); }
	// the base-class quantile functions are stubs in the reference too (dj_brdf.h:1783-1791)
	virtual float_t qf2(float_t, const vec3 &) const { throw exc(DJB_ERR_NOT_IMPLEMENTED, "djb_error: Not Implemented"); }
	virtual float_t qf3(float_t, const vec3 &, float_t) const { throw exc(DJB_ERR_NOT_IMPLEMENTED, "djb_error: Not Implemented"); }
	// ---- the operators that involve the Fresnel term.  With one of the library's terms everything runs behind the handle.
	// With a USER-DEFINED fresnel::impl (m_host_eval) the handle holds the same lobe with fresnel::ideal -- F = (1, 1, 1)
	// exactly -- and the reference's expressions are finished here with the user's eval(), operation for operation.
	using brdf::eval; using brdf::evalp; using brdf::evalp_is;
	vec3 evalp(const vec3 &i, const vec3 &o, const void *user_param = NULL) const                    // dj_brdf.h:1524-1546
	{
		if (!m_host_eval) return handle_evalp(i, o, user_param);
		const params p = user_param ? *reinterpret_cast<const params *>(user_param) : params::standard();
		vec3 h = normalize(i + o);
		if (gaf(h, i, o, p) > (float_t)0.0) {
			float_t cos_theta_d = sat(dot(o, h));
			vec3 F = m_fresnel->eval(cos_theta_d);
			return F * handle_evalp(i, o, user_param).x;           // F * ((D * G) / (4.0 * o.z))
		}
		return vec3(0);
	}
	vec3 eval(const vec3 &i, const vec3 &o, const void *user_param = NULL) const                     // dj_brdf.h:1550-1554
	{ return m_host_eval ? evalp(i, o, user_param) / i.z : eval_resident(i, o, user_param); }
	vec3 evalp_is(float_t u1, float_t u2, const vec3 &o, vec3 *i, float_t *pdf, const void *user_param = NULL) const   // dj_brdf.h:1731-1765
	{
		if (!m_host_eval) {
			vec3 i0; float_t pdf0 = 0;
			const vec3 w0 = handle_evalp_is(u1, u2, o, &i0, &pdf0, user_param);
			if (i) *i = i0;
			if (pdf) *pdf = pdf0;
			return w0;
		}
		const params p = user_param ? *reinterpret_cast<const params *>(user_param) : params::standard();
		vec3 i_; float_t pdf_ = 0;
		vec3 w = handle_evalp_is(u1, u2, o, &i_, &pdf_, user_param);    // G / G1 (Smith VNDF kinds), the direction, its pdf
		vec3 h = normalize(i_ + o);
		if (pdf) *pdf = (float_t)0;
		if (gaf(h, i_, o, p) > (float_t)0.0) {
			if (i) *i = i_;
			if (pdf) *pdf = pdf_;
			if (!supports_smith_vndf_sampling()) return evalp(i_, o, user_param) / pdf_;
			return m_fresnel->eval(sat(dot(o, h))) * w.x;
		}
		return vec3(0);
	}
protected:
	// Batches under a user-defined Fresnel term: the D G part of EVERY pair in one call on the handle (a kernel launch on a GPU
	// context), then the user's F(cos theta_d) per pair on the host.  A non-zero, non-NaN D G value can only come from the reference's
	// G > 0 branch, where the result is F * that value (dj_brdf.h:1545, 1762); every other pair (zeros, NaNs: a handful) takes the
	// one-pair path, which asks for G itself.  Same bits as n scalar calls.
	void host_eval_batch(bool cosine, size_t n, const vec3 *i, const vec3 *o, vec3 *out, const void *user_param) const
	{
		if (!n) return;
		if (!m_host_eval) {      // a user-defined NDF with one of the library's Fresnel terms: the handle answers the whole batch
			djb_vec3_view vi0 = hip::view(i), vo0 = hip::view(o), vr0 = hip::view(out);
			checked((cosine ? djb_evalp_batch : djb_eval_batch)(ctx(), m_h, (int64_t)n, &vi0, &vo0, params_of(user_param), &vr0, DJB_MEM_HOST));
			return;
		}
		std::vector<vec3> dg(n);
		djb_vec3_view vi = hip::view(i), vo = hip::view(o), vr = hip::view(&dg[0]);
		checked(djb_evalp_batch(ctx(), m_h, (int64_t)n, &vi, &vo, params_of(user_param), &vr, DJB_MEM_HOST));
		for (size_t k = 0; k < n; ++k) {
			const float_t s = dg[k].x;
			if (s == s && s != (float_t)0) {
				const vec3 h = normalize(i[k] + o[k]);
				const vec3 fr_cos = m_fresnel->eval(sat(dot(o[k], h))) * s;
				out[k] = cosine ? fr_cos : fr_cos / i[k].z;
			} else out[k] = cosine ? evalp(i[k], o[k], user_param) : eval(i[k], o[k], user_param);
		}
	}
	void host_evalp_is_batch(size_t n, const float_t *u1, const float_t *u2, const vec3 *o, vec3 *out_weight, vec3 *out_i,
	                         float_t *out_pdf, const void *user_param) const
	{
		if (!n) return;
		if (m_host_eval && !supports_smith_vndf_sampling()) { brdf::host_evalp_is_batch(n, u1, u2, o, out_weight, out_i, out_pdf, user_param); return; }
		djb_vec3_view vo = hip::view(o), vw = hip::view(out_weight), vi = hip::view(out_i);
		checked(djb_evalp_is_batch(ctx(), m_h, (int64_t)n, u1, u2, &vo, params_of(user_param), &vw, &vi, out_pdf, DJB_MEM_HOST));
		if (!m_host_eval) return;
		for (size_t k = 0; k < n; ++k) {
			const float_t s = out_weight[k].x;                     // G / G1 under fresnel::ideal
			if (s == s && s != (float_t)0) {
				const vec3 h = normalize(out_i[k] + o[k]);
				out_weight[k] = m_fresnel->eval(sat(dot(o[k], h))) * s;
			} else out_weight[k] = evalp_is(u1[k], u2[k], o[k], &out_i[k], &out_pdf[k], user_param);
		}
	}
	microfacet(hip::context *c, const fresnel::impl &f) : brdf(c), m_fresnel(f.copy()) {}
	/* ---- a USER-DEFINED NDF (the reference's protected constructor and pure virtuals, dj_brdf.h:283-295): a class derived from
	 * microfacet implements sigma_std, p22_std, sample_vp22_std_nmap and supports_smith_vndf_sampling (and may override
	 * sample_vp22_std_smith, qf2, qf3).  The object is a handle on the library's HOST path (hip::context::host()) whose NDF calls
	 * back into these virtuals; params, sigma's stretch, G1 / G2, eval / evalp / pdf / sample / evalp_is and the queries are the
	 * library's own per-unit code, the same the kernels run. */
	microfacet(const fresnel::impl &f = fresnel::ideal(), bool shadow = true) : brdf(&hip::context::host()), m_fresnel(f.copy())
	{ m_host_only = true; create_user_handle(false, shadow); }
	virtual float_t sigma_std(const vec3 &) const { throw exc(DJB_ERR_NOT_IMPLEMENTED, "djb_error: Not Implemented"); }
	virtual float_t p22_std(float_t, float_t) const { throw exc(DJB_ERR_NOT_IMPLEMENTED, "djb_error: Not Implemented"); }
	virtual void sample_vp22_std_smith(float_t u1, float_t u2, const vec3 &k, float_t *xslope, float_t *yslope) const   // dj_brdf.h:1769-1781
	{
		if (supports_smith_vndf_sampling()) { *xslope = qf2(u1, k); *yslope = qf3(u2, k, *xslope); }
		else sample_vp22_std_nmap(u1, u2, k, xslope, yslope);
	}
	virtual void sample_vp22_std_nmap(float_t, float_t, const vec3 &, float_t *, float_t *) const
	{ throw exc(DJB_ERR_NOT_IMPLEMENTED, "djb_error: Not Implemented"); }
	struct user_tag {};
	microfacet(user_tag, const fresnel::impl &f) : brdf(&hip::context::host()), m_fresnel(f.copy()) { m_host_only = true; }   // radial's user constructor
	// the callbacks only forward to the virtuals above.  They are called from the library's host path -- for batches from several of
	// its worker threads at once (the operators are const, as in the reference).  What the user's code throws is kept in the object
	// and re-thrown to the caller when the library call returns (brdf::checked); the library is handed a NaN meanwhile
#if __cplusplus >= 201103L
#	define DJB_HIP_CB(self, expr, fallback) try { return (expr); } catch (...) { (self)->stash_user_exception(); return fallback; }
#else
#	define DJB_HIP_CB(self, expr, fallback) return (expr);
#endif
	static float cb_nan() { unsigned int w = 0x7fc00000u; float f; memcpy(&f, &w, 4); return f; }
	static int cb_smith(void *u) { const microfacet *s = static_cast<const microfacet *>(u); DJB_HIP_CB(s, s->supports_smith_vndf_sampling() ? 1 : 0, 0) }
	static float cb_p22_std(void *u, float x, float y) { const microfacet *s = static_cast<const microfacet *>(u); DJB_HIP_CB(s, s->p22_std(x, y), cb_nan()) }
	static float cb_sigma_std(void *u, const float *k) { const microfacet *s = static_cast<const microfacet *>(u); DJB_HIP_CB(s, s->sigma_std(vec3(k[0], k[1], k[2])), cb_nan()) }
	static void cb_sample_std(void *u, float u1, float u2, const float *k, float *x, float *y)
	{
		const microfacet *s = static_cast<const microfacet *>(u);
		*x = *y = cb_nan();
		DJB_HIP_CB(s, (s->sample_vp22_std_smith(u1, u2, vec3(k[0], k[1], k[2]), x, y), void()), void())
	}
	void create_user_handle(bool, bool shadow)
	{
		djb_user_ndf n = djb_user_ndf();
		n.user = const_cast<microfacet *>(this);
		n.supports_smith_vndf_sampling = cb_smith; n.p22_std = cb_p22_std; n.sigma_std = cb_sigma_std; n.sample_vp22_std = cb_sample_std;
		djb_fresnel_desc d = resident_desc(*m_fresnel, &m_host_eval);
		checked(djb_brdf_create_user_microfacet(ctx(), &n, &d, shadow ? 1 : 0, &m_h));
	}
	// the term the handle is created with: the library's own, or ideal under a user-defined one (*host = true)
	static djb_fresnel_desc resident_desc(const fresnel::impl &f, bool *host)
	{
		djb_fresnel_desc d = f.desc();
		*host = d.kind == DJB_FRESNEL_HOST;
		if (*host) { d = djb_fresnel_desc(); d.kind = DJB_FRESNEL_IDEAL; }
		return d;
	}
	const djb_params *params_of(const void *user_param) const
	{ return user_param ? reinterpret_cast<const params *>(user_param)->desc() : NULL; }   // dj_brdf.h:1532-1534
	float_t q(int which, const vec3 *a, const vec3 *b, const vec3 *c, const params &p) const
	{
		vec3 out;
		djb_vec3_view va = hip::view(a), vb = hip::view(b ? b : a), vc = hip::view(c ? c : a), vo = hip::view(&out);
		checked(djb_query_batch(ctx(), m_h, which, 1, &va, b ? &vb : NULL, c ? &vc : NULL, p.desc(), &vo, DJB_MEM_HOST));
		return out.x;
	}
	vec3 q3(int which, const vec3 &a) const
	{
		vec3 out;
		djb_vec3_view va = hip::view(&a), vo = hip::view(&out);
		checked(djb_query_batch(ctx(), m_h, which, 1, &va, NULL, NULL, NULL, &vo, DJB_MEM_HOST));
		return out;
	}
	const fresnel::impl *m_fresnel;
};

/* Radial microfacets, dj_brdf.h:301-324.  The six queries are the reference's public virtuals: the library's lobes answer them
 * from the handle; a class a USER derives from radial (public constructor, as in the reference) overrides p22_radial,
 * sigma_std_radial, cdf_radial, qf_radial (+ qf2_radial / qf3_radial for Smith VNDF sampling) and supports_smith_vndf_sampling,
 * and is evaluated on the library's host path with those as callbacks (see microfacet). */
class radial : public microfacet {
public:
	radial(const fresnel::impl &f = fresnel::ideal(), bool shadow = true) : microfacet(user_tag(), f)
	{
		djb_user_ndf n = djb_user_ndf();
		n.user = this;
		n.supports_smith_vndf_sampling = cb_smith_r; n.p22_radial = cb_p22_radial; n.sigma_std_radial = cb_sigma_std_radial;
		n.cdf_radial = cb_cdf_radial; n.qf_radial = cb_qf_radial; n.qf2_radial = cb_qf2_radial; n.qf3_radial = cb_qf3_radial;
		djb_fresnel_desc d = resident_desc(*m_fresnel, &m_host_eval);
		checked(djb_brdf_create_user_microfacet(ctx(), &n, &d, shadow ? 1 : 0, &m_h));
	}
	virtual float_t p22_radial(float_t r_sqr) const = 0;
	virtual float_t sigma_std_radial(float_t cos_theta_k) const = 0;
	virtual float_t cdf_radial(float_t r) const = 0;
	virtual float_t qf_radial(float_t u) const = 0;
	virtual float_t qf2_radial(float_t, float_t, float_t) const { throw exc(DJB_ERR_NOT_IMPLEMENTED, "djb_error: Not Implemented"); }   // dj_brdf.h:1848-1855
	virtual float_t qf3_radial(float_t, float_t) const { throw exc(DJB_ERR_NOT_IMPLEMENTED, "djb_error: Not Implemented"); }             // dj_brdf.h:1857-1860
protected:
	radial(hip::context *c, const fresnel::impl &f) : microfacet(c, f) {}
	float_t rq(int which, float_t a, float_t b = 0, float_t c = 0) const
	{ vec3 v(a, b, c); return q(which, &v, NULL, NULL, params::standard()); }
private:
	static int cb_smith_r(void *u) { const radial *s = static_cast<const radial *>(u); DJB_HIP_CB(s, s->supports_smith_vndf_sampling() ? 1 : 0, 0) }
	static float cb_p22_radial(void *u, float r) { const radial *s = static_cast<const radial *>(u); DJB_HIP_CB(s, s->p22_radial(r), cb_nan()) }
	static float cb_sigma_std_radial(void *u, float c) { const radial *s = static_cast<const radial *>(u); DJB_HIP_CB(s, s->sigma_std_radial(c), cb_nan()) }
	static float cb_cdf_radial(void *u, float r) { const radial *s = static_cast<const radial *>(u); DJB_HIP_CB(s, s->cdf_radial(r), cb_nan()) }
	static float cb_qf_radial(void *u, float x) { const radial *s = static_cast<const radial *>(u); DJB_HIP_CB(s, s->qf_radial(x), cb_nan()) }
	static float cb_qf2_radial(void *u, float x, float c, float sn) { const radial *s = static_cast<const radial *>(u); DJB_HIP_CB(s, s->qf2_radial(x, c, sn), cb_nan()) }
	static float cb_qf3_radial(void *u, float x, float q2) { const radial *s = static_cast<const radial *>(u); DJB_HIP_CB(s, s->qf3_radial(x, q2), cb_nan()) }
};
/* what the library's radial lobes declare: the radial queries answered from the handle */
/* The NDF virtuals of the library's own lobes are FINAL (C++11 and later): their operators are answered from the object in HBM, so an
 * override in a class derived from djb::ggx / beckmann / tabular would be ignored by every operator -- in the reference it would
 * change them (eval -> p22_std -> p22_radial is virtual all the way).  A compile error instead of a silent difference: a lobe with its
 * own NDF derives from djb::radial or djb::microfacet (INTEGRATION.md "User-defined classes"). */
#if __cplusplus >= 201103L
#	define DJB_HIP_FINAL final
#else
#	define DJB_HIP_FINAL
#endif
#define DJB_HIP_RESIDENT_RADIAL \
	float_t p22_radial(float_t r_sqr) const DJB_HIP_FINAL { return rq(DJB_Q_P22_RADIAL, r_sqr); } \
	float_t sigma_std_radial(float_t cos_theta_k) const DJB_HIP_FINAL { return rq(DJB_Q_SIGMA_STD_RADIAL, cos_theta_k); } \
	float_t cdf_radial(float_t r) const DJB_HIP_FINAL { return rq(DJB_Q_CDF_RADIAL, r); } \
	float_t qf_radial(float_t u) const DJB_HIP_FINAL { return rq(DJB_Q_QF_RADIAL, u); }
#define DJB_HIP_RESIDENT_RADIAL_SMITH \
	float_t qf2_radial(float_t u, float_t cos_theta_k, float_t sin_theta_k) const DJB_HIP_FINAL { return rq(DJB_Q_QF2_RADIAL, u, cos_theta_k, sin_theta_k); } \
	float_t qf3_radial(float_t u, float_t qf2) const DJB_HIP_FINAL { return rq(DJB_Q_QF3_RADIAL, u, qf2); }

/* Beckmann Microfacet NDF, dj_brdf.h:327-371 */
class beckmann : public radial {
public:
	/* Linear Representation (LEAN / LEADR slope moments), dj_brdf.h:330-353 */
	class lrep {
		friend class beckmann;
	public:
		lrep(float_t E1 = 0, float_t E2 = 0, float_t E3 = 1, float_t E4 = 1, float_t E5 = 0)
		{ m_E[0] = E1; m_E[1] = E2; m_E[2] = E3; m_E[3] = E4; m_E[4] = E5; }
		lrep operator+(const lrep &r) const { return op(DJB_LREP_ADD, &r, 0, 0); }
		lrep operator*(float_t sc) const { DJB_USER_ASSERT(sc >= (float_t)0.0 && "Invalid scale"); return op(DJB_LREP_MUL, NULL, sc, 0); }   // dj_brdf.h:2003
		lrep &operator+=(const lrep &r) { *this = op(DJB_LREP_IADD, &r, 0, 0); return *this; }
		lrep &operator*=(float_t sc) { DJB_USER_ASSERT(sc >= (float_t)0.0 && "Invalid scale"); *this = op(DJB_LREP_IMUL, NULL, sc, 0); return *this; }   // dj_brdf.h:2024
		void scale(float_t x, float_t y) { *this = op(DJB_LREP_SCALE, NULL, x, y); }
		void shear(float_t x, float_t y) { *this = op(DJB_LREP_SHEAR, NULL, x, y); }
		const float_t *moments() const { return m_E; }
	private:
		lrep op(int which, const lrep *r, float_t x, float_t y) const
		{ lrep o; hip::check(djb_lrep_op(which, m_E, r ? r->m_E : NULL, x, y, o.m_E)); return o; }
		float_t m_E[5];
	};
	static void params_to_lrep(const microfacet::params &params, lrep *l)
	{ hip::check(djb_params_to_lrep(params.desc(), l->m_E)); }
	static void lrep_to_params(const lrep &l, microfacet::params *params)
	{
		djb_params d;
		hip::check(djb_lrep_to_params(l.m_E, &d));
		*params = microfacet::params::pdfparams(d.v[0], d.v[1], d.v[2], d.v[3], d.v[4]);
	}
	/* batch form of dj_beckmannconductor's per-hit evaluation (mitsuba/dj_beckmannconductor.cpp:296-319):
	 * lean[n][5] texel moments; params_k = lrep_to_params(lrep(lean_k) * dmapscale + params_to_lrep(base));
	 * lean_flags = DJB_LEAN_NAIVE_MIP (leanFiltering = false) | DJB_LEAN_BIASED (raw texels, bias 25 / 625) */
	void evalp_lean(size_t n, const vec3 *i, const vec3 *o, const microfacet::params &base, float_t dmapscale,
	                const float_t *lean, vec3 *out_fr_cos, float_t *out_pdf = NULL, int lean_flags = 0) const
	{
		djb_vec3_view vi = hip::view(i), vo = hip::view(o), vr = hip::view(out_fr_cos);
		hip::check(djb_eval_lean_batch(ctx(), m_h, (int64_t)n, &vi, &vo, base.desc(), dmapscale, lean_flags, lean,
		                               out_pdf ? 6 : 2, &vr, out_pdf, NULL, DJB_MEM_HOST));
	}
	/* batch form of dj_beckmann_conductor::sample (mitsuba/dj_beckmannconductor.cpp:373-413): per-hit params as above,
	 * then evalp_is; returns through out_fr_cos (weights), out_i, out_pdf */
	void evalp_is_lean(size_t n, const float_t *u1, const float_t *u2, const vec3 *o, const microfacet::params &base,
	                   float_t dmapscale, const float_t *lean, vec3 *out_fr_cos, vec3 *out_i, float_t *out_pdf,
	                   int lean_flags = 0) const
	{
		djb_vec3_view vo = hip::view(o), vr = hip::view(out_fr_cos), vi = hip::view(out_i);
		hip::check(djb_sample_lean_batch(ctx(), m_h, (int64_t)n, u1, u2, &vo, base.desc(), dmapscale, lean_flags, lean,
		                                 &vr, &vi, out_pdf, NULL, DJB_MEM_HOST));
	}
	beckmann(const fresnel::impl &f = fresnel::ideal(), bool shadow = true, hip::context *c = NULL) : radial(c, f)
	{ djb_fresnel_desc d = resident_desc(f, &m_host_eval); hip::check(djb_brdf_create_beckmann(ctx(), &d, shadow, &m_h)); }
	DJB_HIP_RESIDENT_RADIAL
	DJB_HIP_RESIDENT_RADIAL_SMITH
	float_t qf1(float_t u) const { return rq(DJB_Q_QF1, u); }
};

/* GGX Microfacet NDF, dj_brdf.h:374-391 */
class ggx : public radial {
public:
	ggx(const fresnel::impl &f = fresnel::ideal(), bool shadow = true, hip::context *c = NULL) : radial(c, f)
	{ djb_fresnel_desc d = resident_desc(f, &m_host_eval); hip::check(djb_brdf_create_ggx(ctx(), &d, shadow, &m_h)); }
	DJB_HIP_RESIDENT_RADIAL
	DJB_HIP_RESIDENT_RADIAL_SMITH
	float_t qf1(float_t u) const { return rq(DJB_Q_QF1, u); }
};

/* Tabulated Microfacet NDF -- the power-iteration fit, dj_brdf.h:394-425 */
class tabular : public radial {
public:
	tabular(const brdf &src, int resolution, bool shadow = true) : radial(&src.fit_context(), fresnel::ideal())
	{
		DJB_USER_ASSERT(resolution > 2 && "Invalid Resolution");                                                // dj_brdf.h:2218
		if (src.resident()) hip::check(djb_brdf_create_tabular(ctx(), src.handle(), resolution, shadow, &m_h));
		else {   // host code (a user-derived brdf, a user-defined Fresnel term): its eval() sampled where the reference calls it
			int64_t n = 0;
			hip::check(djb_fit_query_dirs(resolution, 0, NULL, NULL, &n));
			std::vector<vec3> qi((size_t)n), qo((size_t)n);
			djb_vec3_view vi = hip::view(&qi[0]), vo = hip::view(&qo[0]);
			hip::check(djb_fit_query_dirs(resolution, n, &vi, &vo, NULL));
			std::vector<float_t> rgb = hip::sample_source(src, qi, qo);
			hip::check(djb_brdf_create_tabular_from_samples(ctx(), resolution, shadow, &rgb[0], n, &m_h));
		}
#ifndef NVERBOSE   // the reference's progress lines, in its order (the whole construction is one launch here)
		DJB_LOG("djb_verbose: Projected area term ready\n");
		DJB_LOG("djb_verbose: Fresnel function ready\n");
		DJB_LOG("djb_verbose: Slope CDF ready\n");
		DJB_LOG("djb_verbose: Slope QF ready\n");
#endif
		m_p22 = fetch(DJB_TAB_P22, 1); m_sigma = fetch(DJB_TAB_SIGMA, 1);
		m_cdf = fetch(DJB_TAB_CDF, 1); m_qf = fetch(DJB_TAB_QF, 1);
		std::vector<float_t> f = fetch(DJB_TAB_FRESNEL, 3);
		std::vector<vec3> pts;
		for (size_t k = 0; k + 2 < f.size(); k += 3) pts.push_back(vec3(f[k], f[k + 1], f[k + 2]));
		delete m_fresnel;
		m_fresnel = new fresnel::spline(pts);
	}
	static microfacet::params fit_beckmann_parameters(const tabular &t)
	{ float_t a = 0; hip::check(djb_tabular_fit(t.m_h, &a, NULL)); return microfacet::params::isotropic(a); }
	static microfacet::params fit_ggx_parameters(const tabular &t)
	{ float_t a = 0; hip::check(djb_tabular_fit(t.m_h, NULL, &a)); return microfacet::params::isotropic(a); }
	DJB_HIP_RESIDENT_RADIAL
	const std::vector<float_t> &get_p22v() const { return m_p22; }
	const std::vector<float_t> &get_sigmav() const { return m_sigma; }
	const std::vector<float_t> &get_cdfv() const { return m_cdf; }
	const std::vector<float_t> &get_qfv() const { return m_qf; }
private:
	std::vector<float_t> fetch(int which, int width) const
	{
		int n = 0;
		hip::check(djb_tabular_get(m_h, which, NULL, &n));
		std::vector<float_t> v((size_t)n * width);
		if (n) hip::check(djb_tabular_get(m_h, which, &v[0], NULL));
		return v;
	}
	std::vector<float_t> m_p22, m_sigma, m_cdf, m_qf;
};

/* Tabulated Anisotropic Microfacet NDF, dj_brdf.h:428-478 */
class tabular_anisotropic : public microfacet {
public:
	tabular_anisotropic(const brdf &src, int elevation_res, int azimuthal_res, bool shadow = true)
		: microfacet(&src.fit_context(), fresnel::ideal()), m_elev(elevation_res), m_azim(azimuthal_res)
	{
		DJB_USER_ASSERT(elevation_res > 1 && azimuthal_res > 1 && "Invalid Resolution");                       // dj_brdf.h:2244
		if (src.resident()) hip::check(djb_brdf_create_tabular_anisotropic(ctx(), src.handle(), elevation_res, azimuthal_res, shadow, &m_h));
		else {   // as in tabular: eval() of the host-evaluated source at the (elev - 1) * azim + (elev - 1) * elev query slots
			int64_t n = 0;
			hip::check(djb_fit_aniso_query_dirs(elevation_res, azimuthal_res, 0, NULL, NULL, &n));
			std::vector<vec3> qi((size_t)n), qo((size_t)n);
			djb_vec3_view vi = hip::view(&qi[0]), vo = hip::view(&qo[0]);
			hip::check(djb_fit_aniso_query_dirs(elevation_res, azimuthal_res, n, &vi, &vo, NULL));
			std::vector<float_t> rgb = hip::sample_source(src, qi, qo);
			hip::check(djb_brdf_create_tabular_anisotropic_from_samples(ctx(), elevation_res, azimuthal_res, shadow, &rgb[0], n, &m_h));
		}
#ifndef NVERBOSE
		DJB_LOG("djb_verbose: Anisotropic projected area term ready\n");
		DJB_LOG("djb_verbose: Fresnel function ready\n");
#endif
		m_p22 = fetch(DJB_ATAB_P22, 1); m_sigma = fetch(DJB_ATAB_SIGMA, 1);
		std::vector<float_t> f = fetch(DJB_ATAB_FRESNEL, 3);
		std::vector<vec3> pts;
		for (size_t k = 0; k + 2 < f.size(); k += 3) pts.push_back(vec3(f[k], f[k + 1], f[k + 2]));
		delete m_fresnel;
		m_fresnel = new fresnel::spline(pts);
	}
	static microfacet::params fit_beckmann_parameters(const tabular_anisotropic &t) { return t.fit(0); }
	static microfacet::params fit_ggx_parameters(const tabular_anisotropic &t) { return t.fit(1); }
	const std::vector<float_t> &get_p22v(int *elev_cnt, int *azim_cnt) const
	{ if (elev_cnt) *elev_cnt = m_elev; if (azim_cnt) *azim_cnt = m_azim; return m_p22; }
	const std::vector<float_t> &get_sigmav(int *elev_cnt, int *azim_cnt) const
	{ if (elev_cnt) *elev_cnt = m_elev; if (azim_cnt) *azim_cnt = m_azim; return m_sigma; }
	// queries (scalar = batch of one)
	float_t pdf1(float_t phi) const { return aq(DJB_Q_ANISO_PDF1, phi, 0); }
	float_t pdf2(float_t theta, float_t phi) const { return aq(DJB_Q_ANISO_PDF2, theta, phi); }
	float_t cdf1(float_t phi) const { return aq(DJB_Q_ANISO_CDF1, phi, 0); }
	float_t cdf2(float_t theta, float_t phi) const { return aq(DJB_Q_ANISO_CDF2, theta, phi); }
	float_t qf1(float_t u1) const { return aq(DJB_Q_ANISO_QF1, u1, 0); }
	float_t qf2(float_t u, float_t phi) const { return aq(DJB_Q_ANISO_QF2, u, phi); }
private:
	float_t aq(int which, float_t a, float_t b) const { vec3 v(a, b, 0); return q(which, &v, NULL, NULL, params::standard()); }
	microfacet::params fit(int which) const
	{
		djb_params b, g;
		hip::check(djb_tabular_anisotropic_fit(m_h, &b, &g));
		const djb_params &d = which == 0 ? b : g;
		return microfacet::params::pdfparams(d.v[0], d.v[1], d.v[2], d.v[3], d.v[4]);
	}
	std::vector<float_t> fetch(int which, int width) const
	{
		int n = 0;
		hip::check(djb_tabular_anisotropic_get(m_h, which, NULL, &n, NULL, NULL));
		std::vector<float_t> v((size_t)n * width);
		if (n) hip::check(djb_tabular_anisotropic_get(m_h, which, &v[0], NULL, NULL, NULL));
		return v;
	}
	std::vector<float_t> m_p22, m_sigma;
	int m_elev, m_azim;
};

} // namespace djb

#endif // DJB_HIP_HPP
