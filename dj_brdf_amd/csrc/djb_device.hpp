// djb_device.hpp -- gfx950 device-side math for the dj_brdf hot path.
//
// Numerical contract: values are stored as float; wherever the reference
// (jdupuy/dj_brdf, dj_brdf.h) evaluates a sub-expression in double (M_PI, 1.0-style literals,
// unqualified libm calls under <cmath>) this code does the same and rounds once, at the same
// place.  Shortcuts are taken only where IEEE-754 guarantees the identical float:
//   * float(sqrt(double(x)))      == sqrtf(x)   (correctly rounded; double rounding is innocuous
//   * float(double(a)/double(b))  == a / b       for sqrt and for one division: 53 >= 2*24+2)
// This header MUST be compiled with -ffp-contract=off: an FMA-contracted a*b+c*d chain moves
// MERL bin indices ("hard parts").
//
// Two instantiations of the SAME source:
//   * the gfx950 device code of the kernels (default; compiled by hipcc), and
//   * DJB_HOST_MATH: plain C++ for the host (djb_cpu.cpp) -- the product's own CPU path for scalar calls and for
//     machines without a GPU.  There every libm call IS the host's glibc call the reference makes, and the
//     guarded / restated device shortcuts below collapse to the exact expressions they stand in for.
#pragma once

#include <stdint.h>
#if defined(DJB_HOST_MATH)
#include <math.h>
#include <string.h>
#define DJB_DEV static inline
struct float4 { float x, y, z, w; };
#if defined(DJB_HOST_RESTATED)
// third instantiation (djb_cpu_libm.cpp only): the host code with the kernels' restatements of glibc 2.35's libm
// functions compiled FOR THE HOST, so that the host path can run the very algorithms the kernels run on a machine
// whose own libm is not glibc 2.35 / x86-64-FMA (see djbhostlibm below).  The device intrinsics the restatements use:
#define __device__
static inline int __double2loint(double x) { uint64_t u; memcpy(&u, &x, 8); return (int)(uint32_t)u; }
static inline int __double2hiint(double x) { uint64_t u; memcpy(&u, &x, 8); return (int)(uint32_t)(u >> 32); }
static inline double __hiloint2double(int hi, int lo) { uint64_t u = ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo; double x; memcpy(&x, &u, 8); return x; }
static inline long long __double_as_longlong(double x) { long long u; memcpy(&u, &x, 8); return u; }
static inline double __longlong_as_double(long long u) { double x; memcpy(&x, &u, 8); return x; }
static inline unsigned int __float_as_uint(float x) { unsigned int u; memcpy(&u, &x, 4); return u; }
static inline float __uint_as_float(unsigned int u) { float x; memcpy(&x, &u, 4); return x; }
#else
// The host path's libm.  The kernels reproduce glibc 2.35's x86-64 (FMA ifunc) exp / pow / atan2 / sin / cos / tan /
// acos / logf / expf / powf; the host path calls the HOST's libm, which is the same thing only on such a host.
// djbhostlibm::init() (first context creation) compares the two on a few thousand arguments per function; when they
// differ, `use_restated` sends every one of these calls of the host path to the kernels' restatements compiled for
// the host (r_*), so scalar-size host calls and GPU batches keep returning the same bits (djb_cpu_libm.cpp).
namespace djbhostlibm {
extern int use_restated;
double r_exp(double), r_pow(double, double), r_atan2(double, double), r_sin(double), r_cos(double), r_tan(double), r_acos(double);
float r_logf(float), r_expf(float), r_powf(float, float);
// 1: the host libm agrees with the restatements on the probe set; 0: it does not (restatements in use unless
// DJB_HOST_LIBM=host); -1: not checked (CPU without FMA: the restatements cannot run on this host)
int init();
int atan_log_kat();    // 1: the host's atan / log (not restated: float -> float sites only) give glibc 2.35's values on the known-answer set
}
#endif
#else
#include <hip/hip_runtime.h>
#define DJB_DEV __device__ __forceinline__
#endif
#define DJB_PI 3.14159265358979323846

namespace djbdev {

struct v3 { float x, y, z; };

// resolved microfacet::params (dj_brdf.h:237-242), filled by the host (djb_host.cpp)
// r_ax, r_t2: doubles 1 / ax and 1 / (ax * ay * s) rounded once (host: 1.0 / double(float)), or 0 = not available
// (per-pair parameters, host path).  With them the three divisions of mf_p22 by launch-uniform denominators become a
// conversion, one fp64 multiply and a conversion -- exactly, see fdiv_r.
struct Params { float nx, ny, nz, ax, ay, rho, s, tx, ty; double r_ax, r_t2; };

struct Fresnel {
	int kind;
	float a[3], b[3];
	const float *pts;   // spline points, 3 floats each (device)
	int npts;
};

// one MERL bin in HBM: packed RGB floats (12 B, one dwordx3 gather).  Packing 10.67 bins per 128 B L2
// line instead of 8 (float4) raises the per-XCD L2 hit rate of the random gather (MERL).
struct MerlTexel { float x, y, z; };

// device view of a djb_brdf
struct Brdf {
	int kind;
	int shadow;
	Fresnel fr;
	const float *p22, *sigma, *cdf, *qf;   // tabular tables (device)
	int n_p22, n_sigma, n_cdf, n_qf;
	const MerlTexel *merl;                  // [1458000] pre-scaled float RGB; below-horizon -> 0
	int merl_sparse;                        // != 0 (file-fit pipeline only): `merl` holds just the texels the fitter reads, one per
	                                        // query slot (fit_merl_slot_count), not the table: never handed to the eval kernels
	const float4 *utia;                     // [288*288][8]: 128-byte records, RGB of the 2x2x2 (theta_v, phi_i, phi_v) taps (k_utia_convert)
	const double *model;                    // sgd: 33 doubles, abc: 9 doubles (one published table row)
	// tabular_anisotropic: p22 / sigma above are elev x azim grids (element (i, j) at [i + elev*j]);
	// two-level sampling tables below (dj_brdf.h:429-438)
	const float *a_pdf1, *a_cdf1, *a_qf1, *a_pdf2, *a_cdf2, *a_qf2;
	int elev, azim, n_a_cdf1, n_a_qf1;
	// where glibc_exp() / glibc_pow() / glibc_acos() read their tables (LdsTab: 0 = the global copy, else 1 + the LDS byte offset
	// of a copy staged by the kernel, which sets these on its own copy of the struct; the host leaves them 0)
	unsigned int exp_lds, pow_lds, acos_lds;
};

struct View { float *x, *y, *z; long long stride; };

enum { KIND_BECKMANN = 0, KIND_GGX = 1, KIND_TABULAR = 2, KIND_MERL = 3, KIND_UTIA = 4, KIND_LAMBERT = 5,
       KIND_SGD = 6, KIND_ABC = 7, KIND_TABULAR_ANISO = 8 };
// the two tabulated microfacet classes sample with the non-VNDF "nmap" scheme (supports_smith_vndf_sampling() == false)
#define DJB_NMAP(K) ((K) == KIND_TABULAR || (K) == KIND_TABULAR_ANISO)
enum { FR_IDEAL = 0, FR_UNPOLARIZED = 1, FR_SCHLICK = 2, FR_SGD = 3, FR_SPLINE = 4 };

// ------------------------------------------------------------------ L0 helpers (dj_brdf.h:574-765)
DJB_DEV float F(double x) { return (float)x; }
DJB_DEV double D(float x) { return (double)x; }
DJB_DEV float fmin_(float a, float b) { return a < b ? a : b; }   // djb::min, dj_brdf.h:574
DJB_DEV float fmax_(float a, float b) { return a > b ? a : b; }   // djb::max, dj_brdf.h:575
DJB_DEV float sat_(float x) { return fmin_(1.0f, fmax_(0.0f, x)); }

// ------------------------------------------------------------------ float -> float sites of the fp64 trig family
// Every place where the path rounds a double libm trig result of ONE float argument straight to float goes through
// one of these, so that each is a float -> float map with 2^32 inputs: tools/exhaustive_trig.py sweeps all of them on
// the device against the host's glibc (djb_selftest_trig_sweep).  The two-argument atan2 sites and the sites that keep
// the double (cos(phi) * sin(theta) products, the fitters' integrands, sgd's g1) are not of this shape: they run glibc's
// own algorithms (glibc_atan2 / sin / cos / tan / acos below).
enum { TRIG_COS = 0, TRIG_SIN, TRIG_TAN, TRIG_ACOS, TRIG_ACOS_U, TRIG_ACOS_U32, TRIG_ATAN_SQU, TRIG_ATAN_U,
       TRIG_ATAN_SQRT, TRIG_BECK_QF, TRIG_ACOS_DEG, TRIG_UTIA_BIN15, TRIG_UTIA_BIN7P5, TRIG_SITES };
// hl_*: the libm call of a site.  Device: ROCm's libm (each site is swept against glibc over all 2^32 inputs); host: the
// host's libm, or the kernels' restatement of glibc's function when the host's libm is not glibc 2.35 (djbhostlibm).
// atan and log have no restatement: their sites follow the host's libm (djbhostlibm::atan_log_kat reports on them).
#if defined(DJB_HOST_MATH) && !defined(DJB_HOST_RESTATED)
DJB_DEV double hl_cos(double x) { return djbhostlibm::use_restated ? djbhostlibm::r_cos(x) : cos(x); }
DJB_DEV double hl_sin(double x) { return djbhostlibm::use_restated ? djbhostlibm::r_sin(x) : sin(x); }
DJB_DEV double hl_tan(double x) { return djbhostlibm::use_restated ? djbhostlibm::r_tan(x) : tan(x); }
DJB_DEV double hl_acos(double x) { return djbhostlibm::use_restated ? djbhostlibm::r_acos(x) : acos(x); }
#else
DJB_DEV double hl_cos(double x) { return cos(x); }
DJB_DEV double hl_sin(double x) { return sin(x); }
DJB_DEV double hl_tan(double x) { return tan(x); }
DJB_DEV double hl_acos(double x) { return acos(x); }
#endif
DJB_DEV float cos_f(float x) { return F(hl_cos(D(x))); }
DJB_DEV float sin_f(float x) { return F(hl_sin(D(x))); }
DJB_DEV float tan_f(float x) { return F(hl_tan(D(x))); }
DJB_DEV float acos_f(float x) { return F(hl_acos(D(x))); }
DJB_DEV float acos_u_f(float c) { return F(2.0 * hl_acos(D(c)) / DJB_PI); }                    // dj_brdf.h:1341 (spline fresnel)
DJB_DEV float acos_u32_f(float c) { return F(D(2.0f) * hl_acos(D(c)) / D(F(DJB_PI))); }       // dj_brdf.h:2158 (tabular sigma)
DJB_DEV float atan_squ_f(float r) { return F(sqrt(D(2.0f) * atan(D(r)) / D(F(DJB_PI)))); }  // dj_brdf.h:2152 (tabular p22)
DJB_DEV float atan_u_f(float r) { return F(atan(D(r)) * D(2.0f) / D(F(DJB_PI))); }          // dj_brdf.h:2165 (tabular cdf)
DJB_DEV float atan_sqrt_f(float x) { return F(atan(sqrt(D(x)))); }                          // dj_brdf.h:2285 (aniso p22)
DJB_DEV float beck_qf_f(float u) { return F(sqrt(-log(1.0 - D(u)))); }                      // dj_brdf.h:1887 (beckmann qf)
DJB_DEV float acos_deg_f(float z) { return F(D(F(180.0 / DJB_PI)) * hl_acos(D(z))); }          // dj_brdf.h:1633 (utia)
// utia's grid cells (dj_brdf.h:1639-1646): (int)floor(double(theta) / 15.0) (clamped to 4) and (int)floor(double(phi) / 7.5)
// are functions of one float too.  The device evaluates them without the fp64 division: RN(theta / 15) >= k  <=>
// theta >= 15 k for a float theta, because a float below 15 k is at least 2^-20 below it while the quotient is rounded at
// 2^-50, and 15 k (7.5 k) is a float; the sweep compares them with the host's division for every float in range.
#if defined(DJB_HOST_MATH)
DJB_DEV int utia_bin15(float theta) { int k = (int)floor(D(theta) / 15.0); return k > 4 ? 4 : k; }
DJB_DEV int utia_bin7p5(float phi) { return (int)floor(D(phi) / 7.5); }
#else
DJB_DEV int utia_bin15(float theta) { return (theta >= 15.0f) + (theta >= 30.0f) + (theta >= 45.0f) + (theta >= 60.0f); }   // theta in [0, 90)
DJB_DEV int utia_bin7p5(float phi)                                                                                          // phi in [0, 360)
{
	int k = (int)(phi * 0.13333334f);              // off by one at most
	if (7.5f * (float)k > phi) --k;
	if (7.5f * (float)(k + 1) <= phi) ++k;
	return k;
}
#endif
// the sites that keep the double (products such as float(double(s) * cos(double(phi))), the sgd / abc models, the
// sigma integrand): site TRIG_DOUBLE + {0 cos, 1 sin, 2 tan, 3 acos} of a float argument, as a double
enum { TRIG_DOUBLE = 16, TRIG_DOUBLE_SITES = 4 };
DJB_DEV double trig_site_d(int fn, float x)
{
	switch (fn - TRIG_DOUBLE) {
	case 0: return cos(D(x));
	case 1: return sin(D(x));
	case 2: return tan(D(x));
	default: return acos(D(x));
	}
}
DJB_DEV float trig_site(int fn, float x)
{
	switch (fn) {
	case TRIG_COS: return cos_f(x);
	case TRIG_SIN: return sin_f(x);
	case TRIG_TAN: return tan_f(x);
	case TRIG_ACOS: return acos_f(x);
	case TRIG_ACOS_U: return acos_u_f(x);
	case TRIG_ACOS_U32: return acos_u32_f(x);
	case TRIG_ATAN_SQU: return atan_squ_f(x);
	case TRIG_ATAN_U: return atan_u_f(x);
	case TRIG_ATAN_SQRT: return atan_sqrt_f(x);
	case TRIG_BECK_QF: return beck_qf_f(x);
	case TRIG_ACOS_DEG: return acos_deg_f(x);
	case TRIG_UTIA_BIN15: return x >= 0.0f && x < 90.0f ? (float)utia_bin15(x) : 0.0f;      // 0 outside the range utia feeds
	default: return x >= 0.0f && x < 360.0f ? (float)utia_bin7p5(x) : 0.0f;
	}
}

DJB_DEV v3 mk(float x, float y, float z) { v3 v; v.x = x; v.y = y; v.z = z; return v; }
DJB_DEV v3 add(v3 a, v3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
DJB_DEV v3 sub(v3 a, v3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
DJB_DEV v3 scale(float s, v3 a) { return mk(s * a.x, s * a.y, s * a.z); }
DJB_DEV float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }          // dj_brdf.h:618
DJB_DEV v3 cross(v3 a, v3 b)                                                           // dj_brdf.h:623
{
	return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// vec3 / float_t == (1.0 / b) * a with the reciprocal rounded to float (dj_brdf.h:601);
// float(1.0 / double(b)) == 1.0f / b (one correctly-rounded division)
DJB_DEV v3 divs(v3 a, float b) { return scale(1.0f / b, a); }
// float(double(a) / (4.0 * double(b))) == a / (4.0f * b): 4*b is exact in float and one division
// of two floats rounds identically through double (dj_brdf.h:1544, 1724-1726, 1754-1760)
DJB_DEV float fdiv4(float a, float b) { return a / (4.0f * b); }
#if defined(DJB_HOST_MATH)
// host: the exact expressions themselves (dj_brdf.h:612; float(1.0 / q))
DJB_DEV float inversesqrt_(float x) { return F(1.0 / sqrt(D(x))); }
DJB_DEV float recip_to_f32(double q) { return F(1.0 / q); }
DJB_DEV float sqrt_to_f32(double a) { return F(sqrt(a)); }
#else
// ---- guarded fast paths for float(<double expression>) -----------------------------------------
// The reference rounds a correctly-rounded double result e to float.  A cheaper double y with
// |y - e| <= 2^-44 |e| rounds to the SAME float unless y lies within 2^-44 (relative) of a float
// rounding boundary, i.e. of a double whose low 29 mantissa bits are 0x10000000.  near_f32_midpoint
// tests that (256 ulp64 either side; probability 2^-20), and the caller then takes the exact path.
// y comes from v_rsq_f64 / v_rcp_f64 refined by two Newton steps (error <= a few 2^-53 for any
// seed accuracy >= 2^-14); e itself is within 2^-52 of the true value.
DJB_DEV bool near_f32_midpoint(double y, int width = 256)
{
	// the 29 mantissa bits a float does not keep sit in the low word: 32-bit arithmetic (4 VALU instead of ~10)
	const int d = (int)((unsigned int)__double2loint(y) & 0x1FFFFFFFu) - 0x10000000;
	return (d < 0 ? -d : d) <= width;
}
// inversesqrt = float(1.0 / sqrt(double(x))): two double roundings (dj_brdf.h:612).
// Seed: v_rsq_f32 (1 ulp, 3 issue slots; v_rsq_f64 costs 5.9 -- profiles/r03/valu_issue_cost.txt), then ONE step of the
// cubically convergent iteration y (1 + e/2 + 3 e^2/8), e = 1 - x y^2, in fp64: x y0 is exact in a double (24 x 24 bits),
// so e carries one rounding; |e| <= 2^-22 leaves (5/16) e^3 < 2^-67 plus three roundings -- within 2 ulp64 of the true value.
DJB_DEV double inversesqrt_fast(float x)
{
	const double xd = D(x), yd = D(__builtin_amdgcn_rsqf(x));
	const double e = __builtin_fma(-(xd * yd), yd, 1.0);
	return __builtin_fma(yd * e, __builtin_fma(e, 0.375, 0.5), yd);
}
DJB_DEV float inversesqrt_(float x)
{
	const double y = inversesqrt_fast(x);
	if (__builtin_expect(near_f32_midpoint(y) || !(x > 1e-30f && x < 1e30f), 0))
		return F(1.0 / sqrt(D(x)));                    // exact path (also zero / inf / NaN / tiny)
	return F(y);
}
// float(1.0 / q) for a double q: v_rcp_f32 seed of float(q) (error < 2^-22 together), one step r (1 + e + e^2), e = 1 - q r
DJB_DEV double recip_fast(double q)
{
	const double r = D(__builtin_amdgcn_rcpf(F(q)));
	const double e = __builtin_fma(-q, r, 1.0);
	return __builtin_fma(r, __builtin_fma(e, e, e), r);
}
DJB_DEV float recip_to_f32(double q)
{
	const double r = recip_fast(q);
	double aq = q < 0 ? -q : q;
	if (__builtin_expect(near_f32_midpoint(r) || !(aq > 1e-30 && aq < 1e30), 0))
		return F(1.0 / q);
	return F(r);
}
// float(sqrt(a)) for a double a that is not a float (1 - c^2 and the like): v_rsq_f32 seed, two coupled Newton steps
// (g -> sqrt(a), h -> 1 / (2 sqrt(a))): 7 fp64 operations instead of the ~17 + v_rsq_f64 of the IEEE expansion
DJB_DEV double sqrt_fast(double a)
{
	const double y = D(__builtin_amdgcn_rsqf(F(a)));
	double g = a * y, h = 0.5 * y;
	const double r = __builtin_fma(-g, h, 0.5);
	g = __builtin_fma(g, r, g); h = __builtin_fma(h, r, h);
	return __builtin_fma(__builtin_fma(-g, g, a), h, g);
}
DJB_DEV float sqrt_to_f32(double a)
{
	const double g = sqrt_fast(a);
	if (__builtin_expect(near_f32_midpoint(g) || !(a > 1e-30 && a < 1e30), 0))
		return F(sqrt(a));                             // exact path (also zero / negative / inf / NaN / tiny)
	return F(g);
}
#endif
// a / b for floats, given R = double(1 / b) to within 2^-52 (or 0: plain division).  float(double(a) * R) is the
// correctly rounded quotient whenever that quotient is a normal float: a / b is never within 2^-49 (relative) of the
// midpoint m of two adjacent floats -- a - m b is a non-zero multiple of ulp(m) ulp(b), so |a/b - m| >= |a/b| / (M B)
// for the integer significands M < 2^25, B < 2^24 -- and never on one (M is odd with 25 bits: M B has more than 24
// significant bits, it cannot equal a), while the product carries at most 2^-52 + 2^-53 of error.  Sub-normal and
// NaN quotients (coarser grid: ties exist) take the IEEE sequence.  Verified against a / b on the device for 1e11
// operand pairs over all exponents (tools/div_probe.hip, profiles/r02/div_probe.txt: 0 mismatches); 2.2 VALU issue
// slots instead of 8.8.
DJB_DEV float fdiv_r(float a, float b, double R)
{
#if !defined(DJB_HOST_MATH)
	if (R != 0.0) {
		float q = F(D(a) * R);
		if (__builtin_expect(!(fabsf(q) >= 1.17549435e-38f) && a != 0.0f, 0)) q = a / b;
		return q;
	}
#endif
	return a / b;
}
DJB_DEV v3 normalize(v3 v) { return scale(inversesqrt_(dot(v, v)), v); }              // dj_brdf.h:630
DJB_DEV float intensity(v3 v) { return 0.2126f * v.x + 0.7152f * v.y + 0.0722f * v.z; } // dj_brdf.h:69

typedef unsigned int LdsTab;   // where a device kernel staged a libm table (0 = the global copy); unused on the host
DJB_DEV double glibc_sin(double x);   // the host libm's sin / cos, for the places that keep the double (defined with the
DJB_DEV double glibc_cos(double x);   // other glibc restatements below); the float -> float sites sin_f / cos_f are swept exhaustively instead
DJB_DEV double glibc_tan(double x);
DJB_DEV double glibc_acos(double x, LdsTab AT);
// vec3(theta, phi), dj_brdf.h:589-595
DJB_DEV v3 from_angles(float theta, float phi)
{
	float s = sin_f(theta);
	return mk(F(D(s) * glibc_cos(D(phi))), F(D(s) * glibc_sin(D(phi))), cos_f(theta));
}

DJB_DEV double glibc_atan2(double y, double x);   // the host libm's atan2 (defined with the other glibc restatements below)
// float(scale * atan2(double y, double x)) with the HOST libm's atan2 (dj_brdf.h:659, 1634).  On the device glibc's
// routine (two IEEE fp64 divisions, a 13.5 KB table) costs twice the device libm's, so the device libm goes first:
// it is within a few ulp64 of glibc's value (both are accurate to <= 2 ulp), and two doubles that close round to the
// same float unless they sit next to a float rounding boundary.  near_f32_midpoint tests 1024 ulp64 either side
// (probability 2^-18); there -- and for zeros, NaNs and results in the float subnormal range -- glibc's own
// algorithm decides (glibc_atan2 below, bit-identical to the host libm: tests/test_gpu_parity.py).  So the value is
// the reference's by construction, at the device libm's speed.
DJB_DEV float atan2_to_f32(float y, float x, double scale);
// dj_brdf.h:650-661
DJB_DEV void xyz_to_theta_phi(v3 p, float &theta, float &phi)
{
	if (D(p.z) > 0.99999) { theta = 0.0f; phi = 0.0f; }
	else if (D(p.z) < -0.99999) { theta = F(DJB_PI); phi = 0.0f; }
	else { theta = acos_f(p.z); phi = atan2_to_f32(p.y, p.x, 1.0); }
}

#if defined(DJB_HOST_MATH) && !defined(DJB_HOST_RESTATED)
// host: the reference's unqualified exp() / pow() ... are the host's glibc functions -- or, on a host whose
// libm is not the glibc the kernels restate, the restatements themselves (djbhostlibm, top of this file)
DJB_DEV double glibc_exp(double x, LdsTab = 0u) { return djbhostlibm::use_restated ? djbhostlibm::r_exp(x) : exp(x); }
DJB_DEV double glibc_pow(double x, double y, LdsTab = 0u, LdsTab = 0u) { return djbhostlibm::use_restated ? djbhostlibm::r_pow(x, y) : pow(x, y); }
DJB_DEV double glibc_atan2(double y, double x) { return djbhostlibm::use_restated ? djbhostlibm::r_atan2(y, x) : atan2(y, x); }
DJB_DEV double glibc_sin(double x) { return hl_sin(x); }
DJB_DEV double glibc_cos(double x) { return hl_cos(x); }
DJB_DEV double glibc_tan(double x) { return hl_tan(x); }
DJB_DEV double glibc_acos(double x, LdsTab) { return hl_acos(x); }
DJB_DEV float atan2_to_f32(float y, float x, double scale) { return F(scale * glibc_atan2(D(y), D(x))); }
#else
// ---- glibc 2.35's double exp / pow, restated --------------------------------------------------------
// The reference's unqualified exp() / pow() are glibc's double functions: ~0.51 ulp, not
// correctly rounded, so ROCm's device libm -- equally close to the true value -- rounds the other way
// once in ~2^28 calls after the cast to float (seen by tools/fuzz_parity.py in an `abc` value).  These are
// glibc's own table-driven algorithms (sysdeps/ieee754/dbl-64/e_exp.c, e_pow.c), with the fusion of the
// x86-64 FMA ifunc variants as read off their disassembly (every a*b+c of the source is one fma); tables
// in djb_glibc_dbl64_tables.hpp (tools/extract_glibc_dbl64_tables.py).  Pinned against the host libm in
// tests/test_oracle_golden.py::test_glibc_double_libm_restatement (CPU) and, as compiled here, in
// tests/test_gpu_parity.py::test_device_libm_restatements.  exp is complete; pow hands zero / negative /
// subnormal / Inf / NaN bases and exponents outside [2^-65, 2^63) -- exact special values -- to the device libm.
#include "djb_glibc_dbl64_tables.hpp"
// LDS copies of the tables are addressed through address-space-3 pointers rebuilt from a 32-bit offset, so that
// the look-ups compile to ds_read (a generic pointer that may be global or LDS compiles to flat_load); the
// offset form also keeps `Brdf` the same size for the host and the device compilation.
#if defined(DJB_HOST_MATH)
typedef const unsigned long long *lds_u64p;   // host instantiation: the tables are only ever read from their global copies (T == 0)
typedef const double *lds_f64p;
#else
typedef const __attribute__((address_space(3))) unsigned long long *lds_u64p;
typedef const __attribute__((address_space(3))) double *lds_f64p;
#endif
DJB_DEV LdsTab glibc_exp_tab_to_lds(unsigned long long *lds, int tid, int nthreads)   // caller: __syncthreads() afterwards
{
	for (int k = tid; k < 256; k += nthreads) lds[k] = DJB_GLIBC_EXP_TAB[k];
	return 1u + (unsigned int)(uintptr_t)(lds_u64p)lds;
}
DJB_DEV LdsTab glibc_pow_tab_to_lds(double *lds, int tid, int nthreads)               // caller: __syncthreads() afterwards
{
	for (int k = tid; k < 384; k += nthreads) lds[k] = DJB_GLIBC_POW_LOG_TAB[k];
	return 1u + (unsigned int)(uintptr_t)(lds_f64p)lds;
}
// tmp and the scale bits of exp()/exp_inline(): r = x - k ln2/N, tmp = tail + r + r^2 p(r), scale = 2^(k/N)
DJB_DEV double glibc_exp_tmp(double x, double xtail, unsigned int &klo, unsigned int &sh, int &sl, LdsTab T)
{
	constexpr double InvLn2N = DJB_GLIBC_EXP_C[0], Shift = DJB_GLIBC_EXP_C[1], NegLn2hiN = DJB_GLIBC_EXP_C[2],
	                 NegLn2loN = DJB_GLIBC_EXP_C[3], C2 = DJB_GLIBC_EXP_C[4], C3 = DJB_GLIBC_EXP_C[5],
	                 C4 = DJB_GLIBC_EXP_C[6], C5 = DJB_GLIBC_EXP_C[7];
	double kd = __builtin_fma(x, InvLn2N, Shift);
	klo = (unsigned int)__double2loint(kd);                              // ki: only bits 0..31 are used
	kd -= Shift;
	double r = __builtin_fma(kd, NegLn2hiN, x);
	r = __builtin_fma(kd, NegLn2loN, r);
	r += xtail;
	const unsigned int idx = 2u * (klo & 127u);
	unsigned long long tb, sb;                                            // {tail bits, scale bits}: one 16-byte entry
	if (T) { lds_u64p L = (lds_u64p)(uintptr_t)(T - 1u); tb = L[idx]; sb = L[idx + 1]; }
	else { tb = DJB_GLIBC_EXP_TAB[idx]; sb = DJB_GLIBC_EXP_TAB[idx + 1]; }
	const double tail = __longlong_as_double((long long)tb);
	sh = (unsigned int)(sb >> 32) + (klo << 13);                         // sbits = tab + (ki << 45): only the high word changes
	sl = (int)(unsigned int)sb;
	double r2 = r * r;
	double p = __builtin_fma(r, C3, C2), q = __builtin_fma(r, C5, C4);
	double t = __builtin_fma(p, r2, tail + r);
	return __builtin_fma(r2 * r2, q, t);
}
// everything outside 2^-54 <= |x| < 512 (cold): tiny, huge, Inf, NaN, and specialcase() of e_exp.c
static __device__ __attribute__((noinline)) double glibc_exp_cold(double x, double xtail, unsigned int abstop, bool is_pow)
{
	const unsigned int hx = (unsigned int)__double2hiint(x);
	if (abstop < 0x3c9u) return 1.0 + x;                                    // |x| < 2^-54
	if (abstop >= 0x409u) {                                                 // |x| >= 1024, Inf, NaN
		if (!is_pow) {
			if (hx == 0xfff00000u && __double2loint(x) == 0) return 0.0;
			if (abstop >= 0x7ffu) return 1.0 + x;
		}
		return (hx >> 31) ? 0.0 : __longlong_as_double(0x7ff0000000000000ll);
	}
	unsigned int klo, sh; int sl;
	const double tmp = glibc_exp_tmp(x, xtail, klo, sh, sl, 0u);
	if ((klo & 0x80000000u) == 0) {
		const double scale = __hiloint2double((int)(sh - (1009u << 20)), sl);
		return 0x1p1009 * __builtin_fma(scale, tmp, scale);
	}
	const double scale = __hiloint2double((int)(sh + (1022u << 20)), sl);
	const double m = tmp * scale;             // not fused in __exp_fma: the product is used twice
	double y = scale + m;
	if (y < 1.0) {
		double lo = scale - y + m;
		double hi = 1.0 + y;
		lo = 1.0 - hi + y + lo;
		y = (hi + lo) - 1.0;
		if (y == 0.0) y = 0.0;
	}
	return 0x1p-1022 * y;
}
// the main path runs unconditionally (garbage outside its domain) and one rarely taken branch replaces it
DJB_DEV double glibc_exp_inline(double x, double xtail, bool is_pow, LdsTab T)
{
	const unsigned int abstop = ((unsigned int)__double2hiint(x) >> 20) & 0x7ffu;
	unsigned int klo, sh; int sl;
	const double tmp = glibc_exp_tmp(x, xtail, klo, sh, sl, T);
	const double scale = __hiloint2double((int)sh, sl);
	double y = __builtin_fma(scale, tmp, scale);
	if (__builtin_expect(abstop - 0x3c9u >= 0x3fu, 0)) y = glibc_exp_cold(x, xtail, abstop, is_pow);
	return y;
}
// T: where the table is read from (0 = global copy, or the handle of an LDS copy)
DJB_DEV double glibc_exp(double x, LdsTab T = 0u) { return glibc_exp_inline(x, 0.0, false, T); }
// PT / ET: the log and exp tables (0 = the global copies; see Brdf::pow_lds / exp_lds)
DJB_DEV double glibc_pow(double x, double y, LdsTab PT = 0u, LdsTab ET = 0u)
{
	constexpr double Ln2hi = DJB_GLIBC_POW_C[0], Ln2lo = DJB_GLIBC_POW_C[1], A0 = DJB_GLIBC_POW_C[2], A1 = DJB_GLIBC_POW_C[3],
	                 A2 = DJB_GLIBC_POW_C[4], A3 = DJB_GLIBC_POW_C[5], A4 = DJB_GLIBC_POW_C[6], A5 = DJB_GLIBC_POW_C[7],
	                 A6 = DJB_GLIBC_POW_C[8];
	const unsigned int hx = (unsigned int)__double2hiint(x), hy = (unsigned int)__double2hiint(y);
	const unsigned int topx = hx >> 20, topy = hy >> 20;
	const bool other = topx - 0x001u >= 0x7ffu - 0x001u || (topy & 0x7ffu) - 0x3beu >= 0x43eu - 0x3beu;
	// tmp = ix - 0x3fe6955500000000: the low word of OFF is zero, so only the high word takes part
	const unsigned int tmph = hx - 0x3fe69555u;
	const int i = (int)((tmph >> 13) & 127u);
	const int k = (int)tmph >> 20;
	const double z = __hiloint2double((int)(hx - (tmph & 0xfff00000u)), __double2loint(x));
	const double kd = (double)k;
	double invc, logc, logctail;
	if (PT) { lds_f64p L = (lds_f64p)(uintptr_t)(PT - 1u) + 3 * i; invc = L[0]; logc = L[1]; logctail = L[2]; }
	else { const double *T = DJB_GLIBC_POW_LOG_TAB + 3 * i; invc = T[0]; logc = T[1]; logctail = T[2]; }
	double r = __builtin_fma(z, invc, -1.0);
	double t1 = __builtin_fma(kd, Ln2hi, logc);
	double t2 = t1 + r;
	double lo1 = __builtin_fma(kd, Ln2lo, logctail);
	double lo2 = t1 - t2 + r;
	double ar = A0 * r, ar2 = r * ar, ar3 = r * ar2;
	double hi = t2 + ar2;
	double lo3 = __builtin_fma(ar, r, -ar2);
	double lo4 = t2 - hi + ar2;
	double p1 = __builtin_fma(r, A2, A1), p2 = __builtin_fma(r, A4, A3), p3 = __builtin_fma(r, A6, A5);
	double q = __builtin_fma(p3, ar2, p2);
	double rr = __builtin_fma(ar2, q, p1);
	double lo = __builtin_fma(ar3, rr, lo1 + lo2 + lo3 + lo4);
	double lhi = hi + lo;
	double llo = hi - lhi + lo;
	double ehi = y * lhi;
	double elo = __builtin_fma(y, llo, __builtin_fma(lhi, y, -ehi));
	double res = glibc_exp_inline(ehi, elo, true, ET);
	if (__builtin_expect(other, 0)) {
		// +0 base, finite non-zero exponent in range (sgd's max(0, theta - theta0)^k): e_pow.c returns x*x or 1/(x*x)
		if ((hx | (unsigned int)__double2loint(x)) == 0u && (topy & 0x7ffu) - 0x3beu < 0x43eu - 0x3beu)
			res = (hy >> 31) ? __longlong_as_double(0x7ff0000000000000ll) : 0.0;
		// negative / subnormal / Inf / NaN bases, |y| outside [2^-65, 2^63): exact special values, device libm
		else res = pow(x, y);
	}
	return res;
}

// ---- glibc 2.35's double atan2, restated ----------------------------------------------------------
// __ieee754_atan2 of sysdeps/ieee754/dbl-64/e_atan2.c (IBM Accurate Mathematical Library; since glibc 2.34 without its
// multi-precision fall-back) as the x86-64 FMA ifunc variant computes it: branch structure, operation order and the
// placement of every fused multiply-add read off the disassembly of __ieee754_atan2_fma; the 241 x 7 table cij out of
// libm.so.6 (tools/extract_glibc_dbl64_tables.py).  u = min / max of the magnitudes by an IEEE division and du its
// residual; u < 1/16: odd polynomial d3 .. d13; else the Taylor expansion about the table point next to u; then the
// quadrant identity with the two-term pi/2 or pi.  Complete: zeros, infinities, NaNs, the exponent-difference
// shortcuts, the 2^+-500 rescaling.  Pinned against the host libm in oracle/ (0 mismatches over 5e7 argument pairs of
// every class) and, as compiled here, in tests/test_gpu_parity.py::test_device_libm_restatements.  It is what
// atan2_to_f32 (above) falls back on next to a float rounding boundary: with it the phi of xyz_to_theta_phi -- the last
// libm call the MERL bin indices went through that was only observed to agree -- and utia's azimuths are the
// reference's by construction.
DJB_DEV double glibc_atan2(double y, double x)
{
	constexpr double d3 = -0x1.5555555555555p-2, d5 = 0x1.99999999997fdp-3, d7 = -0x1.24924923f7603p-3,
	                 d9 = 0x1.c71c6e5129a3bp-4, d11 = -0x1.7458022b13c25p-4, d13 = 0x1.375f08b31cbcep-4,
	                 hpi = 0x1.921fb54442d18p+0, hpi1 = 0x1.1a62633145c07p-54, opi = 0x1.921fb54442d18p+1,
	                 opi1 = 0x1.1a62633145c07p-53, qpi = 0x1.921fb54442d18p-1, tqpi = 0x1.2d97c7f3321d2p+1,
	                 twom500 = 0x1p-500, two500 = 0x1p+500, inv16 = 0x1p-4, TWO52 = 0x1p+52, TWO8 = 0x1p+8;
	const int ux = __double2hiint(x), uy = __double2hiint(y);
	const unsigned int dx = (unsigned int)__double2loint(x), dy = (unsigned int)__double2loint(y);
	// x = NaN or y = NaN
	if ((ux & 0x7ff00000) == 0x7ff00000 && (((ux & 0xfffff) | dx) != 0)) return x + y;
	if ((uy & 0x7ff00000) == 0x7ff00000 && (((uy & 0xfffff) | dy) != 0)) return y + y;
	// y = +-0
	if (uy == 0 && dy == 0) return ux < 0 ? opi : 0.0;
	if ((unsigned int)uy == 0x80000000u && dy == 0) return ux < 0 ? -opi : -0.0;
	// x = +-0
	if (x == 0.0) return uy < 0 ? -hpi : hpi;
	// x = +-Inf
	if (ux == 0x7ff00000 && dx == 0) {
		if (uy == 0x7ff00000 && dy == 0) return qpi;
		if ((unsigned int)uy == 0xfff00000u && dy == 0) return -qpi;
		return uy < 0 ? -0.0 : 0.0;
	}
	if ((unsigned int)ux == 0xfff00000u && dx == 0) {
		if (uy == 0x7ff00000 && dy == 0) return tqpi;
		if ((unsigned int)uy == 0xfff00000u && dy == 0) return -tqpi;
		return uy < 0 ? -opi : opi;
	}
	// y = +-Inf
	if (uy == 0x7ff00000 && dy == 0) return hpi;
	if ((unsigned int)uy == 0xfff00000u && dy == 0) return -hpi;
	double ax = x < 0.0 ? -x : x, ay = y < 0.0 ? -y : y;
	const int de = (uy & 0x7ff00000) - (ux & 0x7ff00000);
	// either x/y or y/x is very close to zero
	if (de >= 0x3900000) return y > 0.0 ? hpi : -hpi;
	if (de <= -0x3900000) {
		if (x > 0.0) return __builtin_copysign(ay / ax, y);
		return y > 0.0 ? opi : -opi;
	}
	if (ax < twom500 || ay < twom500) { ax *= two500; ay *= two500; }
	if (ax > two500 || ay > two500) { ax *= twom500; ay *= twom500; }
	const bool y_lt_x = ay < ax;
	const double mx = y_lt_x ? ax : ay, mn = y_lt_x ? ay : ax;
	const double u = mn / mx;
	double v = mx * u;
	const double vv = __builtin_fma(mx, u, -v);
	const double du = ((mn - v) - vv) / mx;
	// which of (i) x > 0, |y| < |x|: atan(u); (ii) x > 0, |x| <= |y|: pi/2 - atan(u); (iii) x < 0, |x| < |y|: pi/2 + atan(u);
	// (iv) x < 0, |y| <= |x|: pi - atan(u)
	const bool pos = x > 0.0, c3 = !pos && ay > ax;
	double z;
	if (u < inv16) {
		v = u * u;
		double p = __builtin_fma(d13, v, d11);
		p = __builtin_fma(p, v, d9); p = __builtin_fma(p, v, d7); p = __builtin_fma(p, v, d5); p = __builtin_fma(p, v, d3);
		if (pos && y_lt_x) z = u + __builtin_fma(u * v, p, du);
		else {
			const double zz = (u * v) * p, au = u < 0.0 ? -u : u;
			if (pos) {
				const double t2 = hpi - u, cor = hpi > au ? (hpi - t2) - u : hpi - (u + t2);
				z = (((cor + hpi1) - du) - zz) + t2;
			} else if (c3) {
				const double t2 = u + hpi, cor = hpi > au ? (hpi - t2) + u : (u - t2) + hpi;
				z = (((cor + hpi1) + du) + zz) + t2;
			} else {
				const double t2 = opi - u, cor = opi > au ? (opi - t2) - u : opi - (t2 + u);
				z = (((cor + opi1) - du) - zz) + t2;
			}
		}
		return __builtin_copysign(z, y);
	}
	const int i = (int)(__builtin_fma(u, TWO8, TWO52) - TWO52) - 16;
	const double *c = DJB_GLIBC_ATAN_CIJ + 7 * i;
	const double c0 = c[0], c1 = c[1], c2 = c[2], c3_ = c[3], c4 = c[4], c5 = c[5], c6 = c[6];
	const double t3 = u - c0;
	if (pos && y_lt_x) {
		const double w = du + t3, at3 = t3 < 0.0 ? -t3 : t3, adu = du < 0.0 ? -du : du;
		const double dv = at3 > adu ? (t3 - w) + du : (du - w) + t3;
		double p = __builtin_fma(c6, w, c5);
		p = __builtin_fma(p, w, c4); p = __builtin_fma(p, w, c3_);
		p = (w * w) * p;
		p = __builtin_fma(dv, c2, p);
		z = __builtin_fma(w, c2, p) + c1;
		return __builtin_copysign(z, y);
	}
	const double w = t3 + du;
	double p = __builtin_fma(c6, w, c5);
	p = __builtin_fma(p, w, c4); p = __builtin_fma(p, w, c3_); p = __builtin_fma(p, w, c2);
	if (pos) z = (hpi - c1) + __builtin_fma(-w, p, hpi1);
	else if (c3) z = (hpi + c1) + __builtin_fma(w, p, hpi1);
	else z = (opi - c1) + __builtin_fma(-w, p, opi1);
	return __builtin_copysign(z, y);
}
// out of line: inlined into the rarely taken branch of atan2_to_f32 its divisions, constants and table reads cost
// the hot loops more registers and scratch than the branch ever saves (utia eval: 2.8 -> 5.6 ms per 1e8)
#if defined(DJB_HOST_MATH)
DJB_DEV float atan2_to_f32(float y, float x, double scale) { return F(scale * glibc_atan2(D(y), D(x))); }
#else
__device__ __attribute__((noinline)) double glibc_atan2_cold(double y, double x) { return glibc_atan2(y, x); }
// the device-libm value and whether it is decided (tier 1 of a two-tier kernel: undecided units go to a second kernel)
DJB_DEV float atan2_to_f32_t1(float y, float x, double scale, bool &ok)
{
	const double d = scale * atan2(D(y), D(x));
	const double ad = d < 0.0 ? -d : d;
	ok = !near_f32_midpoint(d, 1024) && ad >= 1e-37;
	return F(d);
}
DJB_DEV float atan2_to_f32(float y, float x, double scale)
{
	bool ok;
	const float r = atan2_to_f32_t1(y, x, scale, ok);
	if (__builtin_expect(!ok, 0)) return F(scale * glibc_atan2_cold(D(y), D(x)));
	return r;
}
#endif

// ---- glibc 2.35's double sin / cos, restated -------------------------------------------------------
// __sin / __cos of sysdeps/ieee754/dbl-64/s_sin.c (IBM Accurate Mathematical Library as cleaned up in glibc 2.28: no
// slow paths) as the x86-64 FMA ifunc variants compute them -- operation order and fusion read off __sin_fma /
// __cos_fma.  |x| < 0.126: odd Taylor polynomial; else x = x_k + r with x_k = k / 128 out of the 440-entry
// __sincostab (sin and cos of x_k as double-doubles) and short polynomials in r; 0.855 < |x| < 2.43 through
// pi/2 - |x|; up to 105414350 the three-constant reduction by pi/2.  Beyond that (__branred) the device libm answers:
// the BRDF code's angles never get there.  Pinned against the host libm in oracle/ (0 mismatches over 7.5e7 arguments
// of every class) and, as compiled here, in tests/test_gpu_parity.py::test_device_libm_restatements.  Used where the
// reference keeps the double (float(double(s) * cos(double(phi))) in vec3(theta, phi) and the samplers, the sigma
// integrands of the fitters): those values are the reference's by construction.
DJB_DEV double glibc_do_sin(double x, double dx)                                      // sin(x + dx), |x| < 0.855
{
	constexpr double sn3 = -0x1.5555555555515p-3, sn5 = 0x1.11110e829872fp-7, cs2 = 0.5, cs4 = -0x1.5555555555535p-5,
	                 cs6 = 0x1.6c16bedd9e239p-10, s1 = -0x1.5555555555555p-3, s2 = 0x1.1111111110ecep-7,
	                 s3 = -0x1.a01a019db08b8p-13, s4 = 0x1.71de27b9a7ed9p-19, s5 = -0x1.addffc2fcdf59p-26, big = 0x1.8p+45;
	const double ax = x < 0.0 ? -x : x;
	if (ax < 0.126) {
		const double xx = x * x;
		double p = __builtin_fma(s5, xx, s4);
		p = __builtin_fma(p, xx, s3); p = __builtin_fma(p, xx, s2); p = __builtin_fma(p, xx, s1);
		return x + __builtin_fma(__builtin_fma(p, x, -(0.5 * dx)), xx, dx);
	}
	if (x <= 0.0) dx = -dx;
	const double u = big + ax;
	const double *T = DJB_GLIBC_SINCOS_TAB + 4 * __double2loint(u);
	const double r = ax - (u - big), xx = r * r;
	const double s = r + __builtin_fma(r * xx, __builtin_fma(sn5, xx, sn3), dx);
	const double c = __builtin_fma(r, dx, xx * __builtin_fma(__builtin_fma(cs6, xx, cs4), xx, cs2));
	const double sn = T[0], ssn = T[1], cs = T[2], ccs = T[3];
	const double cor = __builtin_fma(s, cs, __builtin_fma(-c, sn, __builtin_fma(s, ccs, ssn)));
	return __builtin_copysign(sn + cor, x);
}
DJB_DEV double glibc_do_cos(double x, double dx)                                      // cos(x + dx), |x| < 0.855
{
	constexpr double sn3 = -0x1.5555555555515p-3, sn5 = 0x1.11110e829872fp-7, cs2 = 0.5, cs4 = -0x1.5555555555535p-5,
	                 cs6 = 0x1.6c16bedd9e239p-10, big = 0x1.8p+45;
	if (x < 0.0) dx = -dx;
	const double ax = x < 0.0 ? -x : x, u = big + ax;
	const double *T = DJB_GLIBC_SINCOS_TAB + 4 * __double2loint(u);
	const double r = (ax - (u - big)) + dx, xx = r * r;
	const double s = __builtin_fma(r * xx, __builtin_fma(sn5, xx, sn3), r);
	const double c = xx * __builtin_fma(__builtin_fma(cs6, xx, cs4), xx, cs2);
	const double sn = T[0], ssn = T[1], cs = T[2], ccs = T[3];
	const double cor = __builtin_fma(-s, sn, __builtin_fma(-c, cs, __builtin_fma(-s, ssn, ccs)));
	return cs + cor;
}
// reduce_sincos: x = n pi/2 + a + da, |a| <= pi/4, for 2.43 < |x| < 105414350
DJB_DEV int glibc_reduce_sincos(double x, double &a, double &da)
{
	constexpr double toint = 0x1.8p+52, hpinv = 0x1.45f306dc9c883p-1, mp1 = 0x1.921fb58000000p+0, mp2 = -0x1.dde973c000000p-27,
	                 pp3 = -0x1.cb3b398000000p-55, pp4 = -0x1.d747f23e32ed7p-83;
	const double t = __builtin_fma(x, hpinv, toint), xn = t - toint;
	const double y = __builtin_fma(-xn, mp2, __builtin_fma(-xn, mp1, x));
	const double t2 = __builtin_fma(-xn, pp3, y);
	double db = __builtin_fma(-pp3, xn, y - t2);
	const double b = __builtin_fma(-xn, pp4, t2);
	db = db + __builtin_fma(-xn, pp4, t2 - b);
	a = b; da = db;
	return __double2loint(t) & 3;
}
DJB_DEV double glibc_do_sincos(double a, double da, int n)
{
	const double r = (n & 1) ? glibc_do_cos(a, da) : glibc_do_sin(a, da);
	return (n & 2) ? -r : r;
}
DJB_DEV double glibc_sin(double x)
{
	constexpr double hp0 = 0x1.921fb54442d18p+0, hp1 = 0x1.1a62633145c07p-54;
	const int k = __double2hiint(x) & 0x7fffffff;
	if (k < 0x3e500000) return x;                                                       // |x| < 2^-26
	if (k < 0x3feb6000) return glibc_do_sin(x, 0.0);                                    // |x| < 0.855469
	if (k < 0x400368fd) return __builtin_copysign(glibc_do_cos(hp0 - (x < 0.0 ? -x : x), hp1), x);   // |x| < 2.426265
	if (k < 0x419921fb) { double a, da; const int n = glibc_reduce_sincos(x, a, da); return glibc_do_sincos(a, da, n); }
	return sin(x);                                                                      // __branred / Inf / NaN: device libm
}
DJB_DEV double glibc_cos(double x)
{
	constexpr double hp0 = 0x1.921fb54442d18p+0, hp1 = 0x1.1a62633145c07p-54;
	const int k = __double2hiint(x) & 0x7fffffff;
	if (k < 0x3e400000) return 1.0;                                                     // |x| < 2^-27
	if (k < 0x3feb6000) return glibc_do_cos(x, 0.0);
	if (k < 0x400368fd) {
		const double y = hp0 - (x < 0.0 ? -x : x), a = y + hp1, da = (y - a) + hp1;
		return glibc_do_sin(a, da);
	}
	if (k < 0x419921fb) { double a, da; const int n = glibc_reduce_sincos(x, a, da); return glibc_do_sincos(a, da, n + 1); }
	return cos(x);
}

// ---- glibc 2.35's double tan, restated (|x| <= 25) ----------------------------------------------------
// __tan of sysdeps/ieee754/dbl-64/s_tan.c (no slow paths) as __tan_fma computes it.  |x| <= 0.0608: odd polynomial
// d3 .. d11; <= 0.787: x = x_i + z with x_i out of the 186 x 4 table xfg (tan and cot of x_i):
// tan = fi + pz (fi + gi) / (gi - pz); <= 25: x = n pi/2 + a + da (mp1, mp2, mp3), then the same two forms for a, or
// -cot through a double-double division (polynomial) / gi - pz (fi + gi) / (fi + pz) (table) when n is odd.  Larger
// arguments (a longer reduction, __branred) go to the device libm: the anisotropic fitter's angles stay below pi/2.
// Pinned like the others (oracle: 4.6e7 arguments; test_device_libm_restatements on the GPU).
DJB_DEV double glibc_tan(double x)
{
	constexpr double g1 = 0x1.b096c00000000p-27, g2 = 0x1.f212d00000000p-5, g3 = 0x1.92f1a00000000p-1, g4 = 25.0,
	                 d3 = 0x1.5555555555555p-2, d5 = 0x1.11111111107c6p-3, d7 = 0x1.ba1ba1cdb8745p-5, d9 = 0x1.664ed49cfc666p-6,
	                 d11 = 0x1.2385a3cf2e4eap-7, e0 = 0x1.5555555554dbdp-2, e1 = 0x1.11112e0a6b45fp-3, mfftnhf = -15.5, TWO8 = 256.0,
	                 toint = 0x1.8p+52, hpinv = 0x1.45f306dc9c883p-1, mp1 = 0x1.921fb58000000p+0, mp2 = -0x1.dde973c000000p-27,
	                 mp3 = -0x1.cb3b399d747f2p-55;
	if ((__double2hiint(x) & 0x7ff00000) == 0x7ff00000) return x - x;
	const double w = x < 0.0 ? -x : x;
	if (w <= g1) return x;
	if (w <= g2) {
		const double x2 = x * x;
		double t = __builtin_fma(d11, x2, d9);
		t = __builtin_fma(t, x2, d7); t = __builtin_fma(t, x2, d5); t = __builtin_fma(t, x2, d3);
		return __builtin_fma(x * x2, t, x);
	}
	if (w <= g3) {
		const int i = (int)__builtin_fma(TWO8, w, mfftnhf);
		const double *r = DJB_GLIBC_TAN_XFG + 4 * i;
		const double z = w - r[0], z2 = z * z;
		const double pz = __builtin_fma(z * z2, __builtin_fma(z2, e1, e0), z), fi = r[1], gi = r[2];
		return (((fi + gi) * pz) / (gi - pz) + fi) * (x < 0.0 ? -1.0 : 1.0);
	}
	if (!(w <= g4)) return tan(x);
	const double t = __builtin_fma(x, hpinv, toint), xn = t - toint;
	const int n = __double2loint(t) & 1;
	const double t1 = __builtin_fma(-xn, mp2, __builtin_fma(-xn, mp1, x));
	const double a = __builtin_fma(-xn, mp3, t1), da = __builtin_fma(-xn, mp3, t1 - a);
	const bool neg = a < 0.0;
	const double ya = neg ? -a : a, yya = neg ? -da : da, sy = neg ? -1.0 : 1.0;
	if (ya <= g2) {
		const double a2 = a * a;
		double p = __builtin_fma(d11, a2, d9);
		p = __builtin_fma(p, a2, d7); p = __builtin_fma(p, a2, d5); p = __builtin_fma(p, a2, d3);
		const double t2 = __builtin_fma(a * a2, p, da), y = a + t2;
		if (n == 0) return y;
		// -cot(a + da): b + db = a + t2 exactly, then 1 / (b + db) as a double-double
		const double at2 = t2 < 0.0 ? -t2 : t2;
		const double db = ya > at2 ? (a - y) + t2 : (t2 - y) + a;
		const double c = 1.0 / y, ch = c * y, cl = __builtin_fma(c, y, -ch);
		const double cc = __builtin_fma(-db, c, ((1.0 - ch) - cl) + 0.0) / y;
		const double z = c + cc, zz = (c - z) + cc;
		return -(zz + z);
	}
	const int i = (int)__builtin_fma(TWO8, ya, mfftnhf);
	const double *r = DJB_GLIBC_TAN_XFG + 4 * i;
	const double z = (ya - r[0]) + yya, z2 = z * z;
	const double pz = __builtin_fma(z * z2, __builtin_fma(z2, e1, e0), z), fi = r[1], gi = r[2];
	const double num = (fi + gi) * pz;
	if (n) return (gi - num / (pz + fi)) * -sy;
	return (num / (gi - pz) + fi) * sy;
}

// ---- glibc 2.35's double acos, restated -------------------------------------------------------------
// __ieee754_acos of sysdeps/ieee754/dbl-64/e_asin.c (no slow paths) as __ieee754_acos_fma computes it.  |x| < 1/8:
// pi/2 - x - x^3 p(x^2) with a two-term pi/2; seven intervals up to 0.96875 with a Taylor expansion about the nearest
// point of asincos.tbl (rows of 11 .. 15 entries: x_i, the coefficients, acos(x_i)); from 0.96875 to 1:
// 2 asin(sqrt((1 - |x|) / 2)) with the square root seeded from root.tbl and refined as a double-double.  Complete.
// Used by sgd's g1, the one place that keeps the double of an acos (dj_brdf.h:3431).  Pinned like the others
// (oracle: 5.4e7 arguments; test_device_libm_restatements on the GPU).
// AT: 0 = the global tables, else the handle of the LDS copy (asncs followed by inroot) of glibc_acos_tab_to_lds
DJB_DEV LdsTab glibc_acos_tab_to_lds(double *lds, int tid, int nthreads)              // caller: __syncthreads() afterwards
{
	for (int k = tid; k < 2568; k += nthreads) lds[k] = DJB_GLIBC_ASNCS[k];
	for (int k = tid; k < 128; k += nthreads) lds[2568 + k] = DJB_GLIBC_INROOT[k];
	return 1u + (unsigned int)(uintptr_t)(lds_f64p)lds;
}
DJB_DEV double glibc_acos(double x, LdsTab AT)
{
	constexpr double hp0 = 0x1.921fb54442d18p+0, hp1 = 0x1.1a62633145c07p-54, f1 = 0x1.55555555554f9p-3, f2 = 0x1.333333336127dp-4,
	                 f3 = 0x1.6db6dae42c0e4p-5, f4 = 0x1.f1c7e04f4ad99p-6, f5 = 0x1.6e442c822d419p-6, f6 = 0x1.292d80f453c72p-6,
	                 rt0 = 0x1.fffffffecc1ddp-1, rt1 = 0x1.fffffff757304p-2, rt2 = 0x1.800496769c91ap-2, rt3 = 0x1.4006318d1dab9p-2,
	                 t27 = 0x1p+27;
	const int m = __double2hiint(x), k = m & 0x7fffffff;
	if (k < 0x3c880000) return hp0;
	if (k < 0x3fc00000) {
		const double x2 = x * x;
		double p = __builtin_fma(f6, x2, f5);
		p = __builtin_fma(p, x2, f4); p = __builtin_fma(p, x2, f3); p = __builtin_fma(p, x2, f2); p = __builtin_fma(p, x2, f1);
		const double r = hp0 - x;
		return r + __builtin_fma(-p, x * x2, ((hp0 - r) - x) + hp1);
	}
	if (k < 0x3fef0000) {
		int S, n;
		if (k < 0x3fd00000) { S = 11; n = 11 * ((k >> 15) & 0x1f); }
		else if (k < 0x3fe00000) { S = 11; n = 352 + 11 * ((k >> 14) & 0x3f); }
		else if (k < 0x3fe80000) { S = 12; n = 1056 + 12 * ((k >> 13) & 0x7f); }
		else if (k < 0x3fed8000) { S = 13; n = 992 + 13 * ((k >> 13) & 0x7f); }
		else if (k < 0x3fee8000) { S = 14; n = 884 + 14 * ((k >> 13) & 0x7f); }
		else { S = 15; n = 768 + 15 * ((k >> 13) & 0x7f); }
		const lds_f64p L = (lds_f64p)(uintptr_t)(AT - 1u) + n;
		const double *G = DJB_GLIBC_ASNCS + n;
		auto T = [&](int j) { return AT ? L[j] : G[j]; };
		const double xx = (m > 0 ? x : -x) - T(0);
		double p = T(S - 5);
		for (int j = S - 6; j >= 2; --j) p = __builtin_fma(p, xx, T(j));
		p = __builtin_fma(p, xx * xx, T(S - 4));
		const double t = __builtin_fma(xx, T(1), p), y = T(S - 3);
		return m > 0 ? (hp1 - t) + (hp0 - y) : (t + hp1) + (y + hp0);
	}
	if (k < 0x3ff00000) {
		const double z = (m > 0 ? 1.0 - x : x + 1.0) * 0.5;
		const int hz = __double2hiint(z);
		const int ir = (hz >> 14) & 0x7f;
		double t = (AT ? ((lds_f64p)(uintptr_t)(AT - 1u))[2568 + ir] : DJB_GLIBC_INROOT[ir]) * __hiloint2double((1023 + 511 - (hz >> 21)) << 20, 0);   // inroot * powtwo
		const double r = __builtin_fma(-(t * t), z, 1.0);
		double q = __builtin_fma(rt3, r, rt2);
		q = __builtin_fma(q, r, rt1); q = __builtin_fma(q, r, rt0);
		t = q * t;
		const double c = z * t;
		const double h = __builtin_fma(-c, t * 0.5, 1.5);
		const double y = __builtin_fma(-t27, c, __builtin_fma(c, t27, c));
		const double cc = __builtin_fma(-y, y, z) / __builtin_fma(h, c, y);
		double p = __builtin_fma(f6, z, f5);
		p = __builtin_fma(p, z, f4); p = __builtin_fma(p, z, f3); p = __builtin_fma(p, z, f2); p = __builtin_fma(p, z, f1);
		p = (p * z) * (y + cc);
		if (m < 0) return 2.0 * (((hp1 - cc) - p) + (hp0 - y));
		return 2.0 * ((cc + p) + y);
	}
	const unsigned int lo = (unsigned int)__double2loint(x);
	if (k == 0x3ff00000 && lo == 0) return m > 0 ? 0.0 : 2.0 * hp0;
	if (k > 0x7ff00000 || (k == 0x7ff00000 && lo != 0)) return x + x;
	return (x - x) / (x - x);
}

#endif

// A&S 7.1.26 as the reference writes it, dj_brdf.h:667-688
// e must be exp(double(-x*x)) (the same for +x and -x): callers that need that exponential
// themselves (beckmann_qf2_radial) evaluate the fp64 exp once
DJB_DEV float erf_given_exp(float x, double e)
{
	const float a1 = 0.254829592f, a2 = -0.284496736f, a3 = 1.421413741f,
	            a4 = -1.453152027f, a5 = 1.061405429f, p = 0.3275911f;
	float sign = x < 0 ? -1.0f : 1.0f;
	x = fabsf(x);
	float t = recip_to_f32(1.0 + D(p * x));
	float poly = ((((a5 * t + a4) * t) + a3) * t + a2) * t + a1;
	float y = F(1.0 - D(poly * t) * e);
	return sign * y;
}
DJB_DEV float erf_(float x, LdsTab T = 0u) { return erf_given_exp(x, glibc_exp(D(-x * x), T)); }

#if defined(DJB_HOST_MATH) && !defined(DJB_HOST_RESTATED)
// host: logf / std::exp(float) / std::pow(float, float) of the reference (dj_brdf.h:695, 1917, 1935) ARE glibc's
// (or the restatements, djbhostlibm)
struct GlibcTabs { LdsTab exp64; };
DJB_DEV GlibcTabs glibc_tabs_global() { GlibcTabs t = { 0u }; return t; }
DJB_DEV float glibc_logf(float x, const GlibcTabs &) { return djbhostlibm::use_restated ? djbhostlibm::r_logf(x) : logf(x); }
DJB_DEV float glibc_expf(float x, const GlibcTabs &) { return djbhostlibm::use_restated ? djbhostlibm::r_expf(x) : expf(x); }
DJB_DEV float glibc_powf(float x, float y, const GlibcTabs &) { return djbhostlibm::use_restated ? djbhostlibm::r_powf(x, y) : powf(x, y); }
#else
// ---- glibc 2.35's float logf / expf / powf, restated -------------------------------------------
// The reference calls the float libm in erfinv (logf) and in Beckmann's Newton inversion (powf, expf),
// dj_brdf.h:691-721, 1897-1952, so its values are those of the host's glibc -- not of a correctly
// rounded function, and not of ROCm's device libm (which differs in ~13 % of the Beckmann samples).
// These are the table-driven double-arithmetic algorithms glibc uses (sysdeps/ieee754/flt-32/e_logf.c,
// e_expf.c, e_powf.c = ARM optimized-routines), with the multiply-add contractions of the x86-64 FMA
// ifunc variant; tables in djb_glibc_flt32_tables.hpp (tools/extract_glibc_flt32_tables.py).  The
// restatement is pinned against the host libm in oracle/ (0 mismatches over 1.2e8-2e8 arguments per
// function) and, as compiled here, in tests/test_gpu_parity.py::test_device_libm_restatements.  logf and expf
// are complete; powf hands zero / Inf / NaN arguments and negative bases to the device libm.
#include "djb_glibc_flt32_tables.hpp"
// where the indexed tables are read from: the global copies by default; hot kernels stage them in LDS
// (glibc_tabs_to_lds: 768 B) because the Newton loop looks them up twice per iteration.  The scalar
// coefficients are compile-time constants (SGPRs / literals, not per-lane registers).
struct GlibcTabs { const double *logf, *powlog; const unsigned long long *exp2; LdsTab exp64; };
DJB_DEV GlibcTabs glibc_tabs_global() { GlibcTabs t = { DJB_GLIBC_LOGF, DJB_GLIBC_POWF_LOG2, DJB_GLIBC_EXP2F_TAB, 0u }; return t; }
constexpr int GLIBC_LDS_WORDS = 32 + 32 + 32;    // 8-byte words
DJB_DEV GlibcTabs glibc_tabs_to_lds(double *lds, int tid, int nthreads)   // caller: __syncthreads() afterwards
{
	for (int k = tid; k < GLIBC_LDS_WORDS; k += nthreads) {
		double v;
		if (k < 32) v = DJB_GLIBC_LOGF[k];
		else if (k < 64) v = DJB_GLIBC_POWF_LOG2[k - 32];
		else v = __longlong_as_double((long long)DJB_GLIBC_EXP2F_TAB[k - 64]);
		lds[k] = v;
	}
	GlibcTabs t = { lds, lds + 32, (const unsigned long long *)(lds + 64), 0u };
	return t;
}
DJB_DEV float glibc_logf(float x, const GlibcTabs &gt)
{
	const double *T = gt.logf;
	constexpr double Ln2 = DJB_GLIBC_LOGF_C[0], A0 = DJB_GLIBC_LOGF_C[1], A1 = DJB_GLIBC_LOGF_C[2], A2 = DJB_GLIBC_LOGF_C[3];
	unsigned int ix = __float_as_uint(x);
	if (ix == 0x3f800000u) return 0.0f;
	if (__builtin_expect(ix - 0x00800000u >= 0x7f800000u - 0x00800000u, 0)) {
		if (ix * 2u == 0u) return -__builtin_inff();
		if (ix == 0x7f800000u) return x;
		if ((ix & 0x80000000u) || ix * 2u >= 0xff000000u) return __builtin_nanf("");
		ix = __float_as_uint(x * 0x1p23f) - (23u << 23);                 // subnormal: normalise
	}
	unsigned int tmp = ix - 0x3f330000u;
	int i = (int)((tmp >> 19) % 16u), k = (int)tmp >> 23;
	unsigned int iz = ix - (tmp & (0x1ffu << 23));
	double invc = T[2 * i], logc = T[2 * i + 1], z = D(__uint_as_float(iz));
	double r = __builtin_fma(z, invc, -1.0);
	double y0 = __builtin_fma((double)k, Ln2, logc);
	double r2 = r * r;
	double y = __builtin_fma(A1, r, A2);
	y = __builtin_fma(A0, r2, y);
	y = __builtin_fma(y, r2, y0 + r);
	return F(y);
}
// C0..C2: poly (powf) or poly_scaled (expf)
DJB_DEV float glibc_exp2_tail(unsigned long long ki, double r, double C0, double C1, double C2, const GlibcTabs &gt)
{
	// t = tab[ki % 32] + (ki << 47): only the high word changes, and only bits 0..16 of ki reach it
	const unsigned int klo = (unsigned int)ki;
	const unsigned long long tb = gt.exp2[klo & 31u];
	double s = __hiloint2double((int)((unsigned int)(tb >> 32) + (klo << 15)), (int)(unsigned int)tb);
	double zz = __builtin_fma(C0, r, C1);
	double r2 = r * r;
	double y = __builtin_fma(C2, r, 1.0);
	y = __builtin_fma(zz, r2, y);
	return F(y * s);
}
DJB_DEV float glibc_expf(float x, const GlibcTabs &gt)
{
	constexpr double Shift = DJB_GLIBC_EXP2F_C[4], InvLn2N = DJB_GLIBC_EXP2F_C[5];
	unsigned int abstop = (__float_as_uint(x) >> 20) & 0x7ffu;
	if (__builtin_expect(abstop >= (0x42b00000u >> 20), 0)) {          // |x| >= 88 or nan
		if (__float_as_uint(x) == 0xff800000u) return 0.0f;
		if (abstop >= (0x7f800000u >> 20)) return x + x;
		if (x > 0x1.62e42ep6f) return __builtin_inff();                 // x > log(0x1p128)
		if (x < -0x1.9fe368p6f) return 0.0f;                            // x < log(0x1p-150)
		if (x < -0x1.9d1d9ep6f) return __uint_as_float(1u);             // x < log(0x1p-149): __math_may_uflowf = 2^-149
	}
	double xd = D(x), z = InvLn2N * xd;
	double kd = z + Shift;
	unsigned long long ki = (unsigned long long)__double_as_longlong(kd);
	kd -= Shift;
	double r = __builtin_fma(InvLn2N, xd, -kd);
	return glibc_exp2_tail(ki, r, DJB_GLIBC_EXP2F_C[6], DJB_GLIBC_EXP2F_C[7], DJB_GLIBC_EXP2F_C[8], gt);
}
DJB_DEV float glibc_powf(float x, float y, const GlibcTabs &gt)
{
	const double *T = gt.powlog;
	constexpr double A0 = DJB_GLIBC_POWF_C[0], A1 = DJB_GLIBC_POWF_C[1], A2 = DJB_GLIBC_POWF_C[2], A3 = DJB_GLIBC_POWF_C[3],
	                 A4 = DJB_GLIBC_POWF_C[4], ShiftScaled = DJB_GLIBC_EXP2F_C[0];
	unsigned int ix = __float_as_uint(x), iy = __float_as_uint(y);
	if (__builtin_expect(ix - 0x00800000u >= 0x7f800000u - 0x00800000u || 2u * iy - 1u >= 2u * 0x7f800000u - 1u, 0)) {
		// zero / Inf / NaN arguments and negative bases: exact special values (or glibc's sign_bias path): device libm
		if (2u * iy - 1u >= 2u * 0x7f800000u - 1u || 2u * ix - 1u >= 2u * 0x7f800000u - 1u || (ix & 0x80000000u)) return powf(x, y);
		ix = (__float_as_uint(x * 0x1p23f) & 0x7fffffffu) - (23u << 23);   // positive subnormal: normalise
	}
	unsigned int tmp = ix - 0x3f330000u;
	int i = (int)((tmp >> 19) % 16u);
	unsigned int top = tmp & 0xff800000u, iz = ix - top;
	int k = (int)top >> 23;
	double invc = T[2 * i], logc = T[2 * i + 1], z = D(__uint_as_float(iz));
	double r = __builtin_fma(z, invc, -1.0), y0 = logc + (double)k;
	double r2 = r * r;
	double p0 = __builtin_fma(A0, r, A1), p = __builtin_fma(A2, r, A3), r4 = r2 * r2;
	double q = __builtin_fma(A4, r, y0);
	q = __builtin_fma(p, r2, q);
	double logx = __builtin_fma(p0, r4, q);
	double ylogx = D(y) * logx;
	if (__builtin_expect((((unsigned int)__double2hiint(ylogx) >> 15) & 0xffffu) >= (0x405f8000u >> 15), 0)) {   // |y log2 x| >= 126
		if (ylogx > 0x1.fffffffd1d571p+6) return __builtin_inff();
		if (ylogx <= -150.0) return 0.0f;
		if (ylogx < -149.0) return __uint_as_float(1u);                   // __math_may_uflowf
	}
	double kd = ylogx + ShiftScaled;
	unsigned long long ki = (unsigned long long)__double_as_longlong(kd);
	kd -= ShiftScaled;
	double rr = __builtin_fma(D(y), logx, -kd);
	return glibc_exp2_tail(ki, rr, DJB_GLIBC_EXP2F_C[1], DJB_GLIBC_EXP2F_C[2], DJB_GLIBC_EXP2F_C[3], gt);
}

#endif

// Giles' single-precision erfinv, dj_brdf.h:691-721
DJB_DEV float erfinv_(float u, const GlibcTabs &gt)
{
	float w = -glibc_logf((1.0f - u) * (1.0f + u), gt), p;
	if (w < 5.0f) {
		w = w - 2.5f;
		p = 2.81022636e-08f;
		p = 3.43273939e-07f + p * w;
		p = -3.5233877e-06f + p * w;
		p = -4.39150654e-06f + p * w;
		p = 0.00021858087f + p * w;
		p = -0.00125372503f + p * w;
		p = -0.00417768164f + p * w;
		p = 0.246640727f + p * w;
		p = 1.50140941f + p * w;
	} else {
		w = F(sqrt(D(w)) - 3.0);
		p = -0.000200214257f;
		p = 0.000100950558f + p * w;
		p = 0.00134934322f + p * w;
		p = -0.00367342844f + p * w;
		p = 0.00573950773f + p * w;
		p = -0.0076224613f + p * w;
		p = 0.00943887047f + p * w;
		p = 1.00167406f + p * w;
		p = 2.83297682f + p * w;
	}
	return p * u;
}

// Cline's concentric map, dj_brdf.h:726-747
DJB_DEV void uniform_to_concentric(float u1, float u2, float &x, float &y)
{
	float r1 = F(2.0 * D(u1) - 1.0), r2 = F(2.0 * D(u2) - 1.0), phi, r;
	if (r1 == 0 && r2 == 0) { r = phi = 0; }
	else if (r1 * r1 > r2 * r2) { r = r1; phi = F((DJB_PI / 4.0) * D(r2 / r1)); }
	else { r = r2; phi = F((DJB_PI / 2.0) - D(r1 / r2) * (DJB_PI / 4.0)); }
	x = F(D(r) * glibc_cos(D(phi)));
	y = F(D(r) * glibc_sin(D(phi)));
}

// Rodrigues rotation about +z / +y with the reference's exact operation order (dj_brdf.h:754-765).
// For axis = z: dot(axis, x) = x.z, cross(axis, x) = (-x.y, x.x, 0); for axis = y:
// dot = x.y, cross = (x.z, 0, -x.x); the zero products vanish exactly in IEEE arithmetic
// (finite inputs), so only the surviving terms are evaluated.
DJB_DEV v3 rotate_z(v3 x, float angle)
{
	float c = cos_f(angle), s = sin_f(angle);
	float t2 = F(D(x.z) * (1.0 - D(c)));
	// out = c*x; out += axis*t2 (adds 0 to x,y; t2 to z); out += s*cross
	return mk((c * x.x + 0.0f * t2) + s * (0.0f * x.z - x.y),
	          (c * x.y + 0.0f * t2) + s * (x.x - 0.0f * x.z),
	          (c * x.z + t2) + s * (0.0f * x.y - 0.0f * x.x));
}
DJB_DEV v3 rotate_y(v3 x, float angle)
{
	float c = cos_f(angle), s = sin_f(angle);
	float t2 = F(D(x.y) * (1.0 - D(c)));
	return mk((c * x.x + 0.0f * t2) + s * (x.z - 0.0f * x.y),
	          (c * x.y + t2) + s * (0.0f * x.x - 0.0f * x.z),
	          (c * x.z + 0.0f * t2) + s * (0.0f * x.y - x.x));
}

// dj_brdf.h:771-781
DJB_DEV void io_to_hd(v3 i, v3 o, v3 &h, v3 &d)
{
	float th, ph;
	h = normalize(add(i, o));
	xyz_to_theta_phi(h, th, ph);
	v3 tmp = rotate_z(i, -ph);
	d = normalize(rotate_y(tmp, -th));
}

// dj_brdf.h:783-793
DJB_DEV void hd_to_io(v3 h, v3 d, v3 &i, v3 &o)
{
	float th, ph;
	xyz_to_theta_phi(h, th, ph);
	v3 tmp = rotate_y(d, th);
	i = normalize(rotate_z(tmp, ph));
	o = normalize(sub(scale(F(2.0 * D(dot(i, h))), h), i));
}

// ------------------------------------------------------------------ private spline (dj_brdf.h:1181-1249)
DJB_DEV void spline_locate(int edge, float u, int &i1, int &i2, float &frac)
{
	// modf(double(u*edge - u)): integer part by truncation, fractional part exact in float
	float t = u * (float)edge - u;
	float ip = truncf(t);
	frac = t - ip;                       // exact (Sterbenz / same-binade subtraction)
	int k = (int)ip;
	i1 = k >= edge ? edge - 1 : (k < 0 ? 0 : k);          // uwrap_edge
	int k2 = k + 1;
	i2 = k2 >= edge ? edge - 1 : (k2 < 0 ? 0 : k2);
}
DJB_DEV float spline_f(const float *pts, int n, float u)
{
	int i1, i2; float fr;
	spline_locate(n, u, i1, i2, fr);
	float p1 = pts[i1], p2 = pts[i2];
	return p1 + fr * (p2 - p1);
}
DJB_DEV v3 spline_v3(const float *pts, int n, float u)
{
	int i1, i2; float fr;
	spline_locate(n, u, i1, i2, fr);
	v3 p1 = mk(pts[3 * i1], pts[3 * i1 + 1], pts[3 * i1 + 2]);
	v3 p2 = mk(pts[3 * i2], pts[3 * i2 + 1], pts[3 * i2 + 2]);
	return add(p1, scale(fr, sub(p2, p1)));
}

// ------------------------------------------------------------------ Fresnel (dj_brdf.h:1253-1346)
DJB_DEV float unpolarized1(float c, float n)   // dj_brdf.h:1292-1303
{
	float g = F(sqrt(D(n * n + c * c) - 1.0));
	float t1 = F(D(c * (g + c)) - 1.0);
	float t2 = F(D(c * (g - c)) + 1.0);
	float t3 = (t1 * t1) / (t2 * t2);
	float t4 = ((g - c) * (g - c)) / ((g + c) * (g + c));
	return F((0.5 * D(t4)) * (1.0 + D(t3)));
}

DJB_DEV v3 fresnel_eval(const Fresnel &f, float c)
{
	switch (f.kind) {
	case FR_UNPOLARIZED:
		return mk(unpolarized1(c, f.a[0]), unpolarized1(c, f.a[1]), unpolarized1(c, f.a[2]));
	case FR_SCHLICK: {   // dj_brdf.h:1320-1328
		float c1 = 1.0f - c /* == float(1.0 - double(c)) */, c2 = c1 * c1, c5 = c2 * c2 * c1;
		v3 f0 = mk(f.a[0], f.a[1], f.a[2]);
		return add(f0, scale(c5, sub(mk(1, 1, 1), f0)));
	}
	case FR_SGD: {       // dj_brdf.h:1330-1336
		float pw = F(glibc_pow(1.0 - D(c), 5.0));
		v3 f0 = mk(f.a[0], f.a[1], f.a[2]), f1 = mk(f.b[0], f.b[1], f.b[2]);
		return add(sub(f0, scale(c, f1)), scale(pw, sub(mk(1, 1, 1), f0)));
	}
	case FR_SPLINE: {    // dj_brdf.h:1338-1344
		float u = acos_u_f(c);
		return spline_v3(f.pts, f.npts, u);
	}
	default: return mk(1, 1, 1);
	}
}

// ------------------------------------------------------------------ radial NDFs (dj_brdf.h:1866-2176)
template <int KIND> DJB_DEV float p22_radial(const Brdf &b, float r_sqr)
{
	if (KIND == KIND_BECKMANN) return F(glibc_exp(D(-r_sqr), b.exp_lds) / DJB_PI);                      // :1866
	if (KIND == KIND_GGX) { float t = 1.0f + r_sqr; /* == float(1.0 + double(r_sqr)) */ return recip_to_f32(DJB_PI * D(t) * D(t)); } // :2056
	float r = sqrtf(r_sqr);                                                             // :2151
	float u = atan_squ_f(r);
	return spline_f(b.p22, b.n_p22, u);
}

template <int KIND> DJB_DEV float sigma_std_radial(const Brdf &b, float c)
{
	if (KIND == KIND_BECKMANN) {                                                        // :1871
		if (D(c) == 1.0) return 1.0f;
		float s = sqrt_to_f32(1.0 - D(c * c));
		float nu = c / s;
		const double e = glibc_exp(D(-nu * nu), b.exp_lds);              // also the exponential inside erf(nu)
		float tmp = F(e * D(inversesqrt_(F(DJB_PI))));
		return F((D(c) * (1.0 + D(erf_given_exp(nu, e))) + D(s * tmp)) / 2.0);
	}
	if (KIND == KIND_GGX) return (1.0f + c) * 0.5f; /* == float((1.0 + double(c)) / 2.0) */                                // :2062
	float u = acos_u32_f(c);                                   // :2158
	return spline_f(b.sigma, b.n_sigma, u);
}

DJB_DEV float tab_cdf_radial(const Brdf &b, float r)                                   // :2164
{
	float u = atan_u_f(r);
	if (u < 0.0f) u = 0.0f;
	return spline_f(b.cdf, b.n_cdf, sqrtf(u));
}
DJB_DEV float tab_qf_radial(const Brdf &b, float u)                                    // :2171
{
	float qf = spline_f(b.qf, b.n_qf, u);
	return tan_f(qf * F(DJB_PI) / 2.0f);
}

// ------------------------------------------------------------------ tabular_anisotropic fetches
// spline::uwrap_repeat (dj_brdf.h:1183) subtracts / adds `edge` in a loop; the remainder form gives the same
// value for every int and cannot spin for millions of iterations on a wild (saturated) coordinate
DJB_DEV int uwrap_repeat(int i, int edge) { int r = i % edge; return r < 0 ? r + edge : r; }
DJB_DEV int uwrap_edge(int i, int edge) { return i >= edge ? edge - 1 : (i < 0 ? 0 : i); }                      // :1191
DJB_DEV float spline_rep(const float *pts, int n, float u)                             // spline::eval, uwrap_repeat
{
	float t = u * (float)n - u, ip = truncf(t), fr = t - ip;
	int k = (int)ip;
	float p1 = pts[uwrap_repeat(k, n)], p2 = pts[uwrap_repeat(k + 1, n)];
	return p1 + fr * (p2 - p1);
}
DJB_DEV float spline_2d(const float *pts, int w, int h, float u1, float u2)            // spline::eval2d, :1220
{
	float t1 = u1 * (float)w - u1, ip1 = truncf(t1), f1 = t1 - ip1;
	float t2 = u2 * (float)h - u2, ip2 = truncf(t2), f2 = t2 - ip2;
	int i1 = uwrap_edge((int)ip1, w), i2 = uwrap_edge((int)ip1 + 1, w);
	int j1 = uwrap_repeat((int)ip2, h), j2 = uwrap_repeat((int)ip2 + 1, h);
	float p1 = pts[i1 + w * j1], p2 = pts[i2 + w * j1], p3 = pts[i1 + w * j2], p4 = pts[i2 + w * j2];
	float a = p1 + f1 * (p2 - p1), c = p3 + f1 * (p4 - p3);
	return a + f2 * (c - a);
}
DJB_DEV float aniso_grid(const Brdf &b, const float *tab, float theta, float phi)      // :2185-2211
{
	if (D(phi) < 0.0) phi = F(D(phi) + 2.0 * DJB_PI);
	return spline_2d(tab, b.elev, b.azim, F(D(theta) * 2.0 / DJB_PI), F(D(phi) * 0.5 / DJB_PI));
}
DJB_DEV float aniso_p22_theta_phi(const Brdf &b, float theta, float phi) { return aniso_grid(b, b.p22, theta, phi); }
DJB_DEV float aniso_p22_std(const Brdf &b, float x, float y)                           // :2178
{
	return aniso_p22_theta_phi(b, atan_sqrt_f(x * x + y * y), atan2_to_f32(-y, -x, 1.0));
}
DJB_DEV float aniso_sigma_std(const Brdf &b, v3 k)                                     // :2198
{
	return aniso_grid(b, b.sigma, acos_f(k.z), atan2_to_f32(k.y, k.x, 1.0));
}
DJB_DEV float aniso_pdf1(const Brdf &b, float phi) { return spline_rep(b.a_pdf1, b.azim, F(D(phi) * 0.5 / DJB_PI)); }       // :2768
DJB_DEV float aniso_cdf1(const Brdf &b, float phi) { return spline_rep(b.a_cdf1, b.n_a_cdf1, F(D(phi) * 0.5 / DJB_PI)); }
DJB_DEV float aniso_qf1(const Brdf &b, float u1) { return F(D(spline_f(b.a_qf1, b.n_a_qf1, u1)) * 2.0 * DJB_PI); }          // :2780
DJB_DEV float aniso_pdf2(const Brdf &b, float theta, float phi)                        // :2786
{
	if (D(theta) >= 0.5 * DJB_PI) return 0.0f;
	return spline_2d(b.a_pdf2, b.elev, b.azim, F(D(theta) * 2.0 / DJB_PI), F(D(phi) * 0.5 / DJB_PI));
}
DJB_DEV float aniso_cdf2(const Brdf &b, float theta, float phi)                        // :2800
{
	if (D(theta) >= 0.5 * DJB_PI) return 1.0f;
	return spline_2d(b.a_cdf2, b.elev, b.azim, F(D(theta) * 2.0 / DJB_PI), F(D(phi) * 0.5 / DJB_PI));
}
DJB_DEV float aniso_qf2(const Brdf &b, float u, float phi)                             // :2814
{
	return F(D(spline_2d(b.a_qf2, b.elev, b.azim, u, F(D(phi) / (2.0 * DJB_PI)))) * 0.5 * DJB_PI);
}

// analytic cdf / quantile of the radial slope distribution (dj_brdf.h:1881-1889, 2067-2076)
template <int KIND> DJB_DEV float cdf_radial(const Brdf &b, float r)
{
	if (KIND == KIND_BECKMANN) return F(1.0 - glibc_exp(D(-r * r), b.exp_lds));
	if (KIND == KIND_GGX) { float t = r * r; return F(D(t) / (1.0 + D(t))); }
	return tab_cdf_radial(b, r);
}
template <int KIND> DJB_DEV float qf_radial(const Brdf &b, float u)
{
	if (KIND == KIND_BECKMANN) return beck_qf_f(u);
	if (KIND == KIND_GGX) return F(sqrt(D(u) / (1.0 - D(u))));
	return tab_qf_radial(b, u);
}
DJB_DEV float ggx_qf1(float u)                                                         // :2078
{
	if (D(u) < 0.5) { u = F((0.5 - D(u)) * 2.0); return -u * inversesqrt_(F(1.0 - D(u * u))); }
	u = F((D(u) - 0.5) * 2.0);
	return u * inversesqrt_(F(1.0 - D(u * u)));
}

DJB_DEV float beckmann_qf1(float u, const GlibcTabs &gt) { return erfinv_(F(2.0 * D(u) - 1.0), gt); }   // :1891

// Newton + bisection in the erf domain, dj_brdf.h:1897-1952
DJB_DEV float beckmann_qf2_radial(float u, float cos_k, float sin_k, const GlibcTabs &gt)
{
	const float sqrt_pi_inv = F(1. / sqrt(DJB_PI));
	float cot_k = cos_k / sin_k, tan_k = sin_k / cos_k;
	const double e_cot = glibc_exp(D(-cot_k * cot_k), gt.exp64);          // shared by erf(cot_k) and the normalization
	float a = -1, c = erf_given_exp(cot_k, e_cot);
	u = fmax_(u, 1e-6f);
	float fit = 1 + cos_k * (-0.876f + cos_k * (0.4265f - 0.0594f * cos_k));
	float b = c - (1 + c) * glibc_powf(1 - u, fit, gt);
	float normalization = recip_to_f32(D(1 + c) + D(sqrt_pi_inv * tan_k) * e_cot);
	int it = 0;
	float inv_erf = 0.0f;
	bool converged = false;
	// not unrolled: nine copies of the body (each with the out-of-line branches of erfinv / logf / expf) made the
	// sampling kernels 27 KB of code for no gain -- the trip count is data dependent (3.2 on average)
	int it_end = 10;
#if !defined(DJB_HOST_MATH)
	asm volatile("" : "+s"(it_end));       // opaque bound: unroll pragmas alone do not survive the inlining of this loop
#endif
	while (++it < it_end) {
		if (!(b >= a && b <= c)) b = 0.5f * (a + c);
		inv_erf = erfinv_(b, gt);
		float value = normalization * (1 + b + sqrt_pi_inv * tan_k * glibc_expf(-inv_erf * inv_erf, gt)) - u;
		float derivative = normalization * (1 - inv_erf * tan_k);
#if defined(DJB_EXP_NEWTON_FIXED)      // timing experiment only (tools/exp): every lane runs exactly this many iterations
		if (it == DJB_EXP_NEWTON_FIXED) { converged = true; break; }
#else
		if (fabsf(value) < 1e-5f) { converged = true; break; }
#endif
		if (value > 0) c = b; else a = b;
		b -= value / derivative;
	}
	// the reference returns erfinv(max(-0.9999, b)); after a converged exit b is the argument inv_erf was
	// just evaluated at, so the value is reused (same function, same argument) unless the clamp moves it
	if (converged && b >= -0.9999f) return inv_erf;
	return erfinv_(fmax_(-0.9999f, b), gt);
}

DJB_DEV float ggx_qf2_radial(float u, float cos_k, float sin_k)                        // :2089
{
	float sin_t = F(D(u) * (1.0 + D(cos_k)) - 1.0);
	float cos_t = sqrt_to_f32(1.0 - D(sin_t * sin_t));
	if (D(cos_t) > 0.707107) {
		float tan_t = sin_t / cos_t;
		if (D(sin_k) < 0.707107) {
			float tan_k = sin_k / cos_k;
			return F(D(-(tan_t + tan_k)) / (1.0 - D(tan_t * tan_k)));
		} else {
			float cot_k = cos_k / sin_k;
			return F((1.0 + D(tan_t * cot_k)) / D(tan_t - cot_k));
		}
	} else {
		float cot_t = cos_t / sin_t;
		if (D(sin_k) < 0.707107) {
			float tan_k = sin_k / cos_k;
			return F((1.0 + D(tan_k * cot_t)) / D(tan_k - cot_t));
		} else {
			float cot_k = cos_k / sin_k;
			return F(D(cot_t + cot_k) / (1.0 - D(cot_t * cot_k)));
		}
	}
}

DJB_DEV float ggx_qf3_radial(float u, float qf2)                                       // :2121
{
	float alpha = sqrt_to_f32(1.0 + D(qf2 * qf2));
	float S;
	if (D(u) < 0.5) { u = F(2.0 * (0.5 - D(u))); S = -1.0f; }
	else { u = F(2.0 * (D(u) - 0.5)); S = 1.0f; }
	double x = D(u);
	float p = F(x * (x * (x * (-0.365728915865723) + 0.790235037209296) - 0.424965825137544)
	            + 0.000152998850436920);
	float q = F(x * (x * (x * (x * 0.169507819808272 - 0.397203533833404) - 0.232500544458471) + 1)
	            - 0.539825872510702);
	return S * alpha * (p / q);
}

// ------------------------------------------------------------------ microfacet (dj_brdf.h:1529-1765)
template <int KIND> DJB_DEV float mf_p22(const Brdf &b, float x, float y, const Params &p)  // :1574
{
	x -= p.tx; y -= p.ty;
	float nrm = p.ax * p.ay * p.s;
	float x_ = fdiv_r(x, p.ax, p.r_ax);
	float t1 = p.ax * y - p.rho * p.ay * x;
	float t2 = p.ax * p.ay * p.s;                    // == nrm: the same expression
	float y_ = fdiv_r(t1, t2, p.r_t2);
	if (KIND == KIND_TABULAR_ANISO) return fdiv_r(aniso_p22_std(b, x_, y_), nrm, p.r_t2);
	return fdiv_r(p22_radial<KIND>(b, x_ * x_ + y_ * y_), nrm, p.r_t2);
}

template <int KIND> DJB_DEV float mf_ndf(const Brdf &b, v3 h, const Params &p)              // :1559
{
	if (h.z > 1e-4f) {
		float c2 = h.z * h.z, c4 = c2 * c2;
		float xs = -h.x / h.z, ys = -h.y / h.z;
		return mf_p22<KIND>(b, xs, ys, p) / c4;
	}
	return 0.0f;
}

template <int KIND> DJB_DEV float mf_sigma(const Brdf &b, v3 k, const Params &p)            // :1619
{
	float a = k.x * p.ax + k.y * p.ay * p.rho;
	float bb = k.y * p.ay * p.s;
	float c = k.z - k.x * p.tx - k.y * p.ty;
	float nrm = sqrtf(a * a + bb * bb + c * c);
	float rn = 1.0f / nrm;
	if (KIND == KIND_TABULAR_ANISO) return nrm * aniso_sigma_std(b, mk(rn * a, rn * bb, rn * c));
	float kz = rn * c;
	return nrm * sigma_std_radial<KIND>(b, kz);
}

// g1 given a precomputed sigma(k) (sigma is a pure function of k), dj_brdf.h:1633-1642
DJB_DEV float mf_g1_from_sigma(v3 k, float sigma_k, const Params &p)
{
	if (D(dot(k, mk(p.nx, p.ny, p.nz))) > 0.0) return k.z / sigma_k;
	return 0.0f;
}

DJB_DEV float mf_gaf_from_g1(int shadow, float g1i, float g1o)                              // :1644
{
	if (shadow) {
		float t = g1i * g1o;
		if (D(t) > 0.0) return t / (g1i + g1o - t);
		return 0.0f;
	}
	return g1o;
}

// eval / evalp / pdf of one pair, sharing h, sigma(o), sigma(i), D (all pure).
// WANT bits: 1 eval, 2 evalp, 4 pdf.
// FRK >= 0 fixes the Fresnel kind at compile time (the eval kernels are specialised for the two
// cheap, common cases -- ideal and schlick -- so the fp64 branches of the others cost no registers).
template <int FRK> DJB_DEV v3 fresnel_eval_k(const Fresnel &f, float c)
{
	if (FRK == FR_IDEAL) return mk(1, 1, 1);
	if (FRK == FR_SCHLICK) {
		float c1 = 1.0f - c, c2 = c1 * c1, c5 = c2 * c2 * c1;
		v3 f0 = mk(f.a[0], f.a[1], f.a[2]);
		return add(f0, scale(c5, sub(mk(1, 1, 1), f0)));
	}
	if (FRK == FR_UNPOLARIZED) return mk(unpolarized1(c, f.a[0]), unpolarized1(c, f.a[1]), unpolarized1(c, f.a[2]));
	if (FRK == FR_SPLINE) {                              // dj_brdf.h:1338-1344: what every fitted (tabular) lobe carries
		float u = acos_u_f(c);
		return spline_v3(f.pts, f.npts, u);
	}
	return fresnel_eval(f, c);
}

template <int KIND, int WANT, int FRK = -1>
DJB_DEV void mf_eval_pdf(const Brdf &b, const Params &p, v3 i, v3 o, v3 &fr, float &pdf)
{
	v3 h = normalize(add(i, o));
	float sig_o = mf_sigma<KIND>(b, o, p);
	float g1o = mf_g1_from_sigma(o, sig_o, p);
	float g1i = 0.0f;
	if (b.shadow) g1i = mf_g1_from_sigma(i, mf_sigma<KIND>(b, i, p), p);
	float G = mf_gaf_from_g1(b.shadow, g1i, g1o);
	fr = mk(0, 0, 0);
	pdf = 0.0f;
	if (D(G) > 0.0) {
		float Dn = mf_ndf<KIND>(b, h, p);
		float oh = dot(o, h);
		if (WANT & 3) {                                                                      // :1529-1555
			float cd = sat_(oh);
			v3 Fr = fresnel_eval_k<FRK>(b.fr, cd);
			v3 e = scale(fdiv4(Dn * G, o.z), Fr);
			fr = (WANT & 1) ? divs(e, i.z) : e;
		}
		if (WANT & 4) {                                                                      // :1713-1730
			float ih4 = dot(i, h);
			if (DJB_NMAP(KIND)) pdf = fdiv4(h.z * Dn, ih4);
			else {
				float vndf = D(oh) > 0.0 ? oh * Dn / sig_o : 0.0f;                           // :1602-1615
				pdf = fdiv4(vndf, ih4);
			}
		}
	}
}

// radial::sample_vp22_std_smith / _nmap, dj_brdf.h:1806-1846
template <int KIND>
DJB_DEV void mf_sample_vp22_std(const Brdf &b, float u1, float u2, v3 k, float &xs, float &ys, const GlibcTabs &gt)
{
	if (!DJB_NMAP(KIND)) {
		float cos_k = k.z;
		float sin_k = D(k.z) < 1.0 ? sqrt_to_f32(1.0 - D(k.z * k.z)) : 0.0f;
		float tx, ty;
		if (KIND == KIND_BECKMANN) { tx = beckmann_qf2_radial(u1, cos_k, sin_k, gt); ty = beckmann_qf1(u2, gt); }
		else { tx = ggx_qf2_radial(u1, cos_k, sin_k); ty = ggx_qf3_radial(u2, tx); }
		if (D(sin_k) == 0.0) { xs = tx; ys = ty; }
		else {
			float nrm = inversesqrt_(k.x * k.x + k.y * k.y);
			float cp = k.x * nrm, sp = k.y * nrm;
			xs = cp * tx - sp * ty;
			ys = sp * tx + cp * ty;
		}
	} else if (KIND == KIND_TABULAR_ANISO) {                                                    // :2828
		float phi = aniso_qf1(b, u1);
		float theta = aniso_qf2(b, u2, phi);
		float tan_theta = tan_f(theta);
		xs = F(D(-tan_theta) * glibc_cos(D(phi)));
		ys = F(D(-tan_theta) * glibc_sin(D(phi)));
	} else {
		float phi_h = F(D(u1) * DJB_PI * 2.0);
		float r_h = tab_qf_radial(b, u2);
		xs = F(D(r_h) * glibc_cos(D(phi_h)));
		ys = F(D(r_h) * glibc_sin(D(phi_h)));
	}
}

template <int KIND>
DJB_DEV v3 mf_sample(const Brdf &b, const Params &p, float u1, float u2, v3 o, const GlibcTabs &gt)   // :1669
{
	u1 = sat_(u1) * 0.99998f + 0.00001f;
	u2 = sat_(u2) * 0.99998f + 0.00001f;
	float a = o.x * p.ax + o.y * p.ay * p.rho;
	float bb = o.y * p.ay * p.s;
	float c = o.z - o.x * p.tx - o.y * p.ty;
	v3 o_std = normalize(mk(a, bb, c));
	if (D(o_std.z) > 0.0) {
		float txm, tym;
		mf_sample_vp22_std<KIND>(b, u1, u2, o_std, txm, tym, gt);
		float txh = p.ax * txm + p.tx;
		float chol = p.rho * txm + p.s * tym;
		float tyh = p.ay * chol + p.ty;
		v3 h = normalize(mk(-txh, -tyh, 1));
		return sub(scale(F(2.0 * D(dot(o, h))), h), o);
	}
	return mk(0, 0, 1);
}

template <int KIND, int FRK = -1>
DJB_DEV v3 mf_evalp_is(const Brdf &b, const Params &p, float u1, float u2, v3 o, v3 &i_out,
                       float &pdf_out, const GlibcTabs &gt)                                  // :1734
{
	v3 i_ = mf_sample<KIND>(b, p, u1, u2, o, gt);
	v3 h = normalize(add(i_, o));
	float sig_o = mf_sigma<KIND>(b, o, p);
	float g1o = mf_g1_from_sigma(o, sig_o, p);
	float g1i = 0.0f;
	if (b.shadow) g1i = mf_g1_from_sigma(i_, mf_sigma<KIND>(b, i_, p), p);
	float G = mf_gaf_from_g1(b.shadow, g1i, g1o);
	pdf_out = 0.f;
	if (D(G) > 0.0) {
		float oh = dot(o, h);
		float cd = sat_(oh);
		i_out = i_;
		float Dn = mf_ndf<KIND>(b, h, p);
		v3 Fr = fresnel_eval_k<FRK>(b.fr, cd);
		if (DJB_NMAP(KIND)) {
			float pdf_ = fdiv4(h.z * Dn, cd);
			pdf_out = pdf_;
			v3 e = scale(fdiv4(Dn * G, o.z), Fr);   // evalp(i_, o)
			return divs(e, pdf_);
		} else {
			float vndf = D(oh) > 0.0 ? oh * Dn / sig_o : 0.0f;
			pdf_out = fdiv4(vndf, cd);
			return scale(G / g1o, Fr);
		}
	}
	return mk(0, 0, 0);
}

// ------------------------------------------------------------------ MERL (dj_brdf.h:893-1024)
DJB_DEV int theta_half_index(float th)                                                       // :906
{
	if (D(th) <= 0.0) return 0;
	float deg = F((D(th) / (DJB_PI / 2.0)) * 90);
	float t = deg * 90.0f;
	t = sqrtf(t);
	int r = (int)t;
	return r < 0 ? 0 : (r >= 90 ? 89 : r);
}
DJB_DEV int theta_diff_index(float td)                                                       // :926
{
	int t = (int)(D(td) / (DJB_PI * 0.5) * 90);
	return t < 0 ? 0 : (t < 89 ? t : 89);
}
DJB_DEV int phi_diff_index(float pd)                                                         // :940
{
	if (D(pd) < 0.0) pd = F(D(pd) + DJB_PI);
	int t = (int)(D(pd) / DJB_PI * 360 / 2);
	return t < 0 ? 0 : (t < 179 ? t : 179);
}
DJB_DEV int merl_index(v3 i, v3 o)                                                           // :987-1002
{
	v3 h, d; float th, ph, td, pd;
	h = normalize(add(i, o));
	xyz_to_theta_phi(h, th, ph);
	v3 tmp = rotate_z(i, -ph);
	d = normalize(rotate_y(tmp, -th));
	xyz_to_theta_phi(d, td, pd);
	return phi_diff_index(pd) + theta_diff_index(td) * 180 + theta_half_index(th) * 16200;
}
DJB_DEV v3 merl_eval(const Brdf &b, v3 i, v3 o)
{
	// table entries are float(double sample * channel scale) with below-horizon bins zeroed at
	// load time (djb_host.cpp): exactly what dj_brdf.h:1010-1023 returns per lookup.
	MerlTexel t = b.merl[merl_index(i, o)];
	return mk(t.x, t.y, t.z);
}

// the reference's three float angles (theta_h, theta_d, phi_d) -- calibration / diagnostics
DJB_DEV void merl_angles_exact(v3 i, v3 o, float &th, float &td, float &pd)
{
	v3 h, d; float ph;
	h = normalize(add(i, o));
	xyz_to_theta_phi(h, th, ph);
	v3 tmp = rotate_z(i, -ph);
	d = normalize(rotate_y(tmp, -th));
	xyz_to_theta_phi(d, td, pd);
}

// ------------------------------------------------------------------ the source look-ups of djb::tabular's constructor
// tabular(brdf, res) evaluates its source BRDF at a FIXED set of directions that depend on `res` only: cnt = res - 1
// back-scattering configurations for the slope pdf (dj_brdf.h:2488-2499) and, for the Fresnel ratio, the pairs
// (theta_d(i), theta_h(j)), i < cnt, j <= cnt, with dir_i overwritten by (0, 0, 1) (dj_brdf.h:2595-2612).  They are
// numbered as query slots: slot k < cnt = back-scattering direction k, slot cnt + i * (cnt + 1) + j = Fresnel pair (i, j).
// For a MERL source that is all a fit ever reads of the 4.37 M table entries (5.5 k of them at res 90), which lets the
// file pipeline fetch just those (djb_loader.hip).  The same functions give k_fit its directions.
#if defined(DJB_HOST_MATH)
static inline int fit_merl_slot_count(int res) { const int cnt = res - 1; return cnt + cnt * (cnt + 1); }
#else
__host__ __device__ inline int fit_merl_slot_count(int res) { const int cnt = res - 1; return cnt + cnt * (cnt + 1); }
#endif
DJB_DEV float fit_backscatter_theta(int k, int cnt)          // theta of eval(w, w), w = vec3(theta^2, 0)
{
	float tmp = (float)k / (float)cnt;
	return F(D(tmp) * sqrt(DJB_PI * 0.5));
}
// false: the reference's loop skips this pair
DJB_DEV bool fit_fresnel_dirs(int i, int j, int cnt, v3 &dir_i, v3 &dir_o)
{
	float theta_d = F(D((float)i / (float)cnt) * DJB_PI * 0.5);
	float prev = 0.0f;
	if (j > 0) { float t1 = (float)(j - 1) / (float)cnt; prev = F(D(t1 * t1) * DJB_PI * 0.5); }
	float t1 = (float)j / (float)cnt;
	float theta_h = F(D(t1 * t1) * DJB_PI * 0.5);
	if (!(D(prev) < DJB_PI * 0.5 - D(theta_d) && !(D(theta_h) > DJB_PI * 0.5))) return false;
	v3 dir_h = from_angles(theta_h, 0.0f), dir_d = from_angles(theta_d, F(DJB_PI * 0.5));
	hd_to_io(dir_h, dir_d, dir_i, dir_o);
	dir_i = mk(0, 0, 1);                        // dj_brdf.h:2609
	return true;
}
// the MERL table index slot `s` reads, or -1 for a skipped pair
DJB_DEV int fit_merl_slot_index(int s, int res)
{
	const int cnt = res - 1;
	if (s < cnt) {
		float th = fit_backscatter_theta(s, cnt);
		v3 w = from_angles(th * th, 0.0f);
		return merl_index(w, w);
	}
	const int e = s - cnt, i = e / (cnt + 1), j = e - i * (cnt + 1);
	v3 dir_i, dir_o;
	if (!fit_fresnel_dirs(i, j, cnt, dir_i, dir_o)) return -1;
	return merl_index(dir_i, dir_o);
}

#if !defined(DJB_HOST_MATH)   // tier 1 is a device optimisation; the host runs merl_index as written
// ------------------------------------------------------------------ two-tier exact MERL binning
// Tier 1 (this function): the three half/diff angles from closed-form geometry in fp32 --
//     theta_h = angle(h, z),  theta_d = angle(i, h),  phi_d = azimuth of i around h
// with h = (i+o)/|i+o| -- no rotations, no fp64, three polynomial atan2.  Each estimate carries a
// guard band that bounds |estimate - the reference's own float value|: the reference's chain of
// float roundings (amplified by cot(theta_h) and 1/sin(theta_d)) plus this path's own error.
// If an estimate lies inside the guard band of a bin boundary -- or in the regions where the
// reference snaps angles (|z| > 0.99999, dj_brdf.h:652-656) -- the pair is AMBIGUOUS and is
// handed to tier 2, the operation-by-operation fp64 path (merl_index).  Outside the bands both
// paths land in the same bin, so the composite is bit-exact while ~99.5 % of the pairs never touch fp64.
// The band constants are DERIVED (a first-order worst-case bound of every rounding on both paths, u = 2^-24):
//     |theta_h est - ref| <= u (5 cot(theta_h) + 13.1)                                   -> a_h = 12 (>= 5), b_h = 16
//     |theta_d est - ref| <= u (25.5 + 5 cot(theta_h)) / sin(theta_d) + 32.4 u           -> a_d = 26, b_d = 12, c_d = 36
//     |phi_d   est - ref| <= u (36.9 + 5 cot(theta_h)) / sin(theta_d) + 25.6 u           -> a_p = 40, b_d = 12, c_d = 36
// and ATTACKED: tools/merl_guard_attack.py hill-climbs input bit patterns to maximise |estimate - reference| / band
// (profiles/r03/merl_guard_attack.txt); k_merl_guard_stats samples the same ratio, and tests/test_gpu_verification.py
// asserts it stays below 0.5 with 0 index mismatches on the bench distribution and on twelve adversarial families.
struct MerlGuard { float a_h, b_h, a_d, b_d, c_d, a_p; };   // multiples of 2^-24
#define MERL_GUARD_DEFAULT { 12.0f, 16.0f, 26.0f, 12.0f, 36.0f, 40.0f }

DJB_DEV float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
DJB_DEV float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }

// atan2 with |error| < ~2.5e-7 rad: octant reduction + Cephes atanf core (4 coefficients)
DJB_DEV float fast_atan2(float y, float x)
{
	float ax = fabsf(x), ay = fabsf(y);
	float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
	float z = mn * fast_rcp(mx);                                   // [0, 1]
	bool hi = z > 0.41421356f;
	float w = hi ? (z - 1.0f) * fast_rcp(z + 1.0f) : z;            // |w| <= tan(pi/8)
	float w2 = w * w;
	float p = ((8.05374449538e-2f * w2 - 1.38776856032e-1f) * w2 + 1.99777106478e-1f) * w2 - 3.33329491539e-1f;
	float a = p * w2 * w + w;
	a = hi ? a + 0.78539816339f : a;
	a = ay > ax ? 1.57079632679f - a : a;
	a = x < 0.0f ? 3.14159265359f - a : a;
	return y < 0.0f ? -a : a;
}

struct MerlFast {
	float t_h, x_d, x_p;     // continuous bin coordinates: floor() gives the bin
	float m_h, m_d, m_p;     // guard-band half widths in the same coordinates
	bool special;            // snap regions / degenerate input: always tier 2
};

DJB_DEV MerlFast merl_fast_coords(v3 i, v3 o, const MerlGuard g)
{
	const float U = 5.9604644775390625e-08f;   // 2^-24
	MerlFast f;
	float sx = i.x + o.x, sy = i.y + o.y, sz = i.z + o.z;
	float r = __builtin_amdgcn_rsqf(sx * sx + sy * sy + sz * sz);
	float hx = sx * r, hy = sy * r, hz = sz * r;
	float sh2 = hx * hx + hy * hy;
	float sh = fast_sqrt(sh2);                                      // sin(theta_h)
	float th = fast_atan2(sh, hz);
	// theta_d = angle(i, h): |i x h| and i . h share the factor |i|
	float cx = i.y * hz - i.z * hy, cy = i.z * hx - i.x * hz, cz = i.x * hy - i.y * hx;
	float sdn = fast_sqrt(cx * cx + cy * cy + cz * cz);
	float cdn = i.x * hx + i.y * hy + i.z * hz;
	float td = fast_atan2(sdn, cdn);
	// phi_d: components of i along e_theta, e_phi of the h frame, both scaled by |i| sin(theta_h)
	float ny = hx * i.y - hy * i.x;
	float nx = hz * (hx * i.x + hy * i.y) - sh2 * i.z;
	float pd = fast_atan2(ny, nx);
	float ilen = fast_sqrt(i.x * i.x + i.y * i.y + i.z * i.z);
	float sd = sdn * fast_rcp(ilen);                                // sin(theta_d)
	float cd = cdn * fast_rcp(ilen);
	// guard bands, radians
	float rsh = fast_rcp(sh), rsd = fast_rcp(sd);
	float e_h = U * (g.a_h * rsh + g.b_h);
	float e_d = U * (g.a_d + g.b_d * rsh) * rsd + U * g.c_d;
	float e_p = U * (g.a_p + g.b_d * rsh) * rsd + U * g.c_d;
	// bin coordinates (dj_brdf.h:906-957)
	const float R2D = 57.29577951308232f;
	float deg_h = th * R2D;
	f.t_h = fast_sqrt(deg_h * 90.0f);
	f.m_h = e_h * (2578.3100780887044f * fast_rcp(fmaxf(f.t_h, 1e-3f))) + f.t_h * (8.0f * U);
	f.x_d = td * R2D;
	f.m_d = e_d * R2D;
	float pw = pd < 0.0f ? pd + 3.14159265359f : pd;
	f.x_p = pw * R2D;
	f.m_p = e_p * R2D;
	// The reference snaps (theta, phi) to (0, 0) / (pi, 0) when |z| > 0.99999, i.e. when the angle is
	// within acos(0.99999) = 4.4721e-3 rad of a pole (dj_brdf.h:652-656).  A pair is "special"
	// (always tier 2) unless both estimates, shrunk by their guard bands, clear that zone.
	const float SNAP = 4.6e-3f, PI_F = 3.14159265359f;
	f.special = !(th > SNAP + e_h) || !(th < PI_F - SNAP - e_h) || !(td > SNAP + e_d) || !(td < PI_F - SNAP - e_d);
	(void)cd;
	return f;
}

// true iff the bin index is certain; idx is then identical to merl_index(i, o)
DJB_DEV bool merl_index_fast(v3 i, v3 o, const MerlGuard g, int &idx)
{
	MerlFast f = merl_fast_coords(i, o, g);
	// distance to the nearest integer boundary in each coordinate
	float rh = rintf(f.t_h), rd = rintf(f.x_d), rp = rintf(f.x_p);
	bool amb_h = fabsf(f.t_h - rh) < f.m_h && rh >= 1.0f && rh <= 89.0f;    // boundaries 1..89
	bool amb_d = fabsf(f.x_d - rd) < f.m_d && rd >= 1.0f && rd <= 89.0f;    // boundaries 1..89
	bool amb_p = fabsf(f.x_p - rp) < f.m_p;                                  // 0..180 (0 == 180 wrap)
	// comparisons are false on NaN, so a NaN coordinate must force tier 2 explicitly
	bool finite = (f.t_h == f.t_h) && (f.x_d == f.x_d) && (f.x_p == f.x_p) &&
	              (f.m_h < 0.45f) && (f.m_d < 0.45f) && (f.m_p < 0.45f);
	int kh = (int)f.t_h, kd = (int)f.x_d, kp = (int)f.x_p;
	kh = kh > 89 ? 89 : kh; kd = kd > 89 ? 89 : kd; kp = kp > 179 ? 179 : kp;
	idx = kp + kd * 180 + kh * 16200;
	return finite && !f.special && !amb_h && !amb_d && !amb_p;
}

#endif

// ------------------------------------------------------------------ UTIA (dj_brdf.h:1063-1157)
// sRGB decode of dj_brdf.h:1147-1150: float(pow(double(float(double(v) + 0.055)) / 1.055, double(2.4f))).
// Exact form: an IEEE fp64 division + the fp64 libm pow (~150 fp64 instructions).  Guarded form
// (same idea as inversesqrt_): with t = num / 1.055 and p = double(2.4f) = 2.4 + d,
//     t^p = (t z)^3 * exp(d ln t),   z = t^(-1/5),
// z from a division-free Newton iteration z <- z (6 - t z^5) / 5 seeded by v_log_f32 / v_exp_f32
// (error 3 e^2 per step: 2e-7 -> 1e-13 -> rounding level), exp(d ln t) = 1 + x + x^2/2 with
// x = d ln t <= 3e-7 and ln t from the same v_log_f32 (absolute error <= 4e-7 -> 4e-14 relative in
// the result).  Total error < 2^-44; the result is used only if it is not within 2^-42 (1024 ulp64)
// of an fp32 rounding boundary, otherwise -- or outside t in [1/16, 16] -- the exact form runs.
DJB_DEV float srgb_decode_exact(float v)
{
	return F(glibc_pow(D(F(D(v) + 0.055)) / 1.055, D(2.4f)));
}
#if defined(DJB_HOST_MATH)
DJB_DEV float srgb_decode(float v) { return srgb_decode_exact(v); }
#else
DJB_DEV double srgb_decode_fast(float v, bool &ok)
{
	const double num = D(F(D(v) + 0.055));
	const double t = num * (1.0 / 1.055);                       // <= 1 ulp64 from the IEEE quotient
	const float tf = F(t);
	const float L = __builtin_amdgcn_logf(tf);                  // log2(t), ~1 ulp
	double z = D(__builtin_amdgcn_exp2f(-0.2f * L));            // t^(-1/5), ~3e-7
#pragma unroll
	for (int it = 0; it < 2; ++it) {
		double z2 = z * z, z4 = z2 * z2, z5 = z4 * z;
		double e = __builtin_fma(-t, z5, 1.0);
		z = __builtin_fma(z * 0.2, e, z);
	}
	double u = t * z, r = u * u * u;                            // t^2.4
	const double delta = D(2.4f) - 2.4;                         // 9.5367431640625e-08
	double x = delta * (D(L) * 0.6931471805599453);
	r = r * __builtin_fma(x, __builtin_fma(x, 0.5, 1.0), 1.0);
	ok = !near_f32_midpoint(r, 1024) && (tf > 0.0625f && tf < 16.0f);
	return r;
}
DJB_DEV float srgb_decode(float v)
{
	bool ok;
	double r = srgb_decode_fast(v, ok);
	if (__builtin_expect(!ok, 0)) return srgb_decode_exact(v);
	return F(r);
}

#endif

// T1 (device only): tier 1 of the batch kernel -- the device-libm azimuths without glibc's atan2 behind them; ok = false
// when one of the two was not decided away from a float rounding boundary (k_eval_utia_t1 then lists the pair for
// k_eval_utia_fix, which runs utia_eval).  Everything else is the same code.
template <bool T1> DJB_DEV v3 utia_eval_t(const Brdf &b, v3 i, v3 o, bool &ok)
{
	float r2d = F(180.0 / DJB_PI);
	float theta_i = acos_deg_f(i.z), theta_o = acos_deg_f(o.z);
	float phi_i, phi_o;
	ok = true;
#if !defined(DJB_HOST_MATH)
	if (T1) {
		bool ok_i, ok_o;
		phi_i = atan2_to_f32_t1(i.y, i.x, D(r2d), ok_i); phi_o = atan2_to_f32_t1(o.y, o.x, D(r2d), ok_o);
		ok = ok_i && ok_o;
	} else
#endif
	{ phi_i = atan2_to_f32(i.y, i.x, D(r2d)); phi_o = atan2_to_f32(o.y, o.x, D(r2d)); }
	if (D(theta_i) >= 90.0 || D(theta_o) >= 90.0) return mk(0, 0, 0);
	if (!(phi_i == phi_i) || !(phi_o == phi_o)) return mk(0, 0, 0);   // NaN guard: reference would spin
	while (D(phi_i) < 0.0) phi_i = F(D(phi_i) + 360.0);
	while (D(phi_o) < 0.0) phi_o = F(D(phi_o) + 360.0);
	while (phi_i >= 360.0f) phi_i = F(D(phi_i) - 360.0);
	while (phi_o >= 360.0f) phi_o = F(D(phi_o) - 360.0);
	int iti0 = utia_bin15(theta_i), iti1 = iti0 + 1;          // (int)floor(theta / 15.0), > 4 -> 4 (then iti1 = 5)
	int itv0 = utia_bin15(theta_o), itv1 = itv0 + 1;
	int ipi0 = utia_bin7p5(phi_i), ipi1 = ipi0 + 1;            // (int)floor(phi / 7.5)
	int ipv0 = utia_bin7p5(phi_o), ipv1 = ipv0 + 1;
	float sum, wti[2], wtv[2], wpi[2], wpv[2];
	wti[1] = theta_i - F(15.0 * iti0); wti[0] = F(15.0 * iti1) - theta_i;
	sum = wti[0] + wti[1]; wti[0] /= sum; wti[1] /= sum;
	wtv[1] = theta_o - F(15.0 * itv0); wtv[0] = F(15.0 * itv1) - theta_o;
	sum = wtv[0] + wtv[1]; wtv[0] /= sum; wtv[1] /= sum;
	wpi[1] = phi_i - F(7.5 * ipi0); wpi[0] = F(7.5 * ipi1) - phi_i;
	sum = wpi[0] + wpi[1]; wpi[0] /= sum; wpi[1] /= sum;
	wpv[1] = phi_o - F(7.5 * ipv0); wpv[0] = F(7.5 * ipv1) - phi_o;
	sum = wpv[0] + wpv[1]; wpv[0] /= sum; wpv[1] /= sum;
	if (ipi1 == 48) ipi1 = 0;
	if (ipv1 == 48) ipv1 = 0;
	int iti[2] = { iti0, iti1 };
	(void)ipv1; (void)ipi1; (void)itv1;
	// The reference walks the three colour planes one after the other, 16 taps each (48 scattered
	// 4-byte reads).  The HBM table stores, per (theta_i, phi_i, theta_v, phi_v) node, the RGB of the
	// eight taps (theta_v + {0,1}) x (phi_i + {0,1}) x (phi_v + {0,1}) (azimuths wrapped) in one
	// 128-byte record = one L2 line, so the 16 taps are 2 aligned line gathers; each plane still
	// accumulates its 16 terms in the reference's order (a, c, k, l nested, l innermost), so the
	// sums are bit-identical.
	float acc[3] = { 0.0f, 0.0f, 0.0f };
#pragma unroll
	for (int a = 0; a < 2; ++a) {
		int e = 288 * (48 * iti[a] + ipi0) + 48 * itv0 + ipv0;
		const float4 *rec = b.utia + 8 * (size_t)e;
		float4 q[6];
#pragma unroll
		for (int j = 0; j < 6; ++j) q[j] = rec[j];
		const float t[24] = { q[0].x, q[0].y, q[0].z, q[0].w, q[1].x, q[1].y, q[1].z, q[1].w, q[2].x, q[2].y, q[2].z, q[2].w,
		                      q[3].x, q[3].y, q[3].z, q[3].w, q[4].x, q[4].y, q[4].z, q[4].w, q[5].x, q[5].y, q[5].z, q[5].w };
#pragma unroll
		for (int c = 0; c < 2; ++c)
#pragma unroll
		for (int k = 0; k < 2; ++k)
#pragma unroll
		for (int l = 0; l < 2; ++l) {
			float w = wti[a] * wtv[c] * wpi[k] * wpv[l];
			const int tap = 3 * (4 * c + 2 * k + l);
			acc[0] += w * t[tap]; acc[1] += w * t[tap + 1]; acc[2] += w * t[tap + 2];
		}
	}
	float RGB[3];
#pragma unroll
	for (int isp = 0; isp < 3; ++isp) {
		float v = acc[isp];
		if (D(v) > 0.0375) v = srgb_decode(v);
		else v /= 12.92f;
		RGB[isp] = v * 100.0f;
	}
	return mk(fmax_(0.f, RGB[0]), fmax_(0.f, RGB[1]), fmax_(0.f, RGB[2]));
}
DJB_DEV v3 utia_eval(const Brdf &b, v3 i, v3 o) { bool ok; return utia_eval_t<false>(b, i, o, ok); }

// ------------------------------------------------------------------ per-pair params / beckmann::lrep
// params::pdfparams(ax, ay, rho, tx, ty) -> the members eval needs (dj_brdf.h:1437-1474)
DJB_DEV Params params_from_pdfparams(float ax, float ay, float rho, float tx, float ty)
{
	Params p;
	p.r_ax = 0.0; p.r_t2 = 0.0;                      // per-pair denominators: a reciprocal per lane costs more than the division
	p.ax = ax; p.ay = ay; p.rho = rho; p.tx = tx; p.ty = ty;
	p.s = F(sqrt(1.0 - D(rho * rho)));
	v3 n = normalize(mk(-tx, -ty, 1.0f));
	p.nx = n.x; p.ny = n.y; p.nz = n.z;
	return p;
}
struct Lrep { float E1, E2, E3, E4, E5; };                                                    // :350-352
DJB_DEV Lrep lrep_add(Lrep a, Lrep r)                                                         // :1992
{
	Lrep o;
	o.E1 = a.E1 + r.E1; o.E2 = a.E2 + r.E2;
	o.E3 = a.E3 + r.E3 + 2.0f * a.E1 * r.E1;
	o.E4 = a.E4 + r.E4 + 2.0f * a.E2 * r.E2;
	o.E5 = a.E5 + r.E5 + a.E1 * r.E2 + a.E2 * r.E1;
	return o;
}
DJB_DEV void lrep_to_pdfparams(Lrep l, float &ax, float &ay, float &rho, float &tx, float &ty)  // :1976
{
	float t1 = fmax_(0.0f, l.E3 - l.E1 * l.E1), t2 = fmax_(0.0f, l.E4 - l.E2 * l.E2);
	ax = F(fmax(1e-5, sqrt(2.0 * D(t1))));
	ay = F(fmax(1e-5, sqrt(2.0 * D(t2))));
	rho = 2.0f * (l.E5 - l.E1 * l.E2) / (ax * ay);
	rho = fmin_(0.99f, fmax_(-0.99f, rho));
	tx = l.E1; ty = l.E2;
}

// ------------------------------------------------------------------ SGD (dj_brdf.h:3415-3500)
// theta_k = acos(double(k.z)): the same for the three channels, evaluated once by the callers
DJB_DEV double sgd_g1(const Brdf &b, double theta_k, double theta0, double c, double k_, double lambda)   // :3415
{
	double t1 = fmax(0.0, theta_k - theta0);
	double t2 = 1.0 - glibc_exp(c * glibc_pow(t1, k_, b.pow_lds, b.exp_lds), b.exp_lds);
	double t3 = 1.0 + lambda * t2;
	return fmin(1.0, fmax(0.0, t3));
}
DJB_DEV double sgd_ndf(const Brdf &b, double ch, double alpha, double p, double kap)                     // :3424
{
	const double inv_pi = 1.0 / DJB_PI;
	double c2 = ch * ch;
	double t2 = (1.0 - c2) / c2;
	double ax = alpha + t2 / alpha;
	return (kap * glibc_exp(-ax, b.exp_lds) * inv_pi) / (glibc_pow(ax, p, b.pow_lds, b.exp_lds) * c2 * c2);
}
// model row: rhoD rhoS alpha p f0 f1 kap lambda c k theta0 (3 doubles each)
DJB_DEV v3 sgd_g1_rgb(const Brdf &b, v3 k)                                                // sgd::g1, :3477
{
	const double *m = b.model;
	const double theta_k = glibc_acos(D(k.z), b.acos_lds);
	return mk(F(sgd_g1(b, theta_k, m[30], m[24], m[27], m[21])), F(sgd_g1(b, theta_k, m[31], m[25], m[28], m[22])),
	          F(sgd_g1(b, theta_k, m[32], m[26], m[29], m[23])));
}
DJB_DEV v3 sgd_ndf_rgb(const Brdf &b, v3 h)                                               // sgd::ndf, :3490
{
	const double *m = b.model;
	return mk(F(sgd_ndf(b, D(h.z), m[6], m[9], m[18])), F(sgd_ndf(b, D(h.z), m[7], m[10], m[19])),
	          F(sgd_ndf(b, D(h.z), m[8], m[11], m[20])));
}
DJB_DEV v3 sgd_gaf_rgb(const Brdf &b, v3 i, v3 o)                                         // sgd::gaf = g1(i) * g1(o), :3472
{
	v3 gi = sgd_g1_rgb(b, i), go = sgd_g1_rgb(b, o);
	return mk(gi.x * go.x, gi.y * go.y, gi.z * go.z);
}
DJB_DEV v3 sgd_eval(const Brdf &b, v3 i, v3 o)                                            // :3454
{
	const double *m = b.model;
	if (D(i.z) > 0.0 && D(o.z) > 0.0) {
		v3 h = normalize(add(i, o));
		v3 Kd = mk(F(m[0]), F(m[1]), F(m[2])), Ks = mk(F(m[3]), F(m[4]), F(m[5]));
		v3 Fr = fresnel_eval(b.fr, sat_(dot(i, h)));
		v3 G = sgd_gaf_rgb(b, i, o), Dn = sgd_ndf_rgb(b, h);
		v3 FDG = mk((Fr.x * Dn.x) * G.x, (Fr.y * Dn.y) * G.y, (Fr.z * Dn.z) * G.z);
		v3 spec = divs(mk(Ks.x * FDG.x, Ks.y * FDG.y, Ks.z * FDG.z), i.z * o.z);
		return divs(add(Kd, spec), F(DJB_PI));
	}
	return mk(0, 0, 0);
}

// ------------------------------------------------------------------ ABC (dj_brdf.h:3608-3668)
// model row: kD[3] A[3] B C ior
DJB_DEV float abc_gaf(v3 h, v3 i, v3 o)                                                   // abc::gaf, :3647
{
	float g1_i = fmin_(1.0f, 2.0f * (h.z * i.z / dot(h, i)));
	float g1_o = fmin_(1.0f, 2.0f * (h.z * o.z / dot(h, o)));
	return fmin_(g1_i, g1_o);
}
DJB_DEV v3 abc_ndf_rgb(const Brdf &b, v3 h)                                               // abc::ndf, :3657 + abc__ndf :3608
{
	const double *m = b.model;
	double den = glibc_pow(1.0 + m[6] * (1.0 - D(h.z)), m[7], b.pow_lds, b.exp_lds);
	return mk(F(m[3] / den), F(m[4] / den), F(m[5] / den));
}
DJB_DEV v3 abc_eval(const Brdf &b, v3 i, v3 o)                                            // :3633
{
	const double *m = b.model;
	if (D(i.z) > 0.0 && D(o.z) > 0.0) {
		v3 h = normalize(add(i, o));
		v3 Kd = mk(F(m[0]), F(m[1]), F(m[2]));
		v3 Fr = fresnel_eval(b.fr, sat_(dot(i, h)));
		float G = abc_gaf(h, i, o);
		v3 Dn = abc_ndf_rgb(b, h);
		v3 spec = divs(scale(G, mk(Fr.x * Dn.x, Fr.y * Dn.y, Fr.z * Dn.z)), F(DJB_PI * D(i.z) * D(o.z)));
		return add(divs(Kd, F(DJB_PI)), spec);
	}
	return mk(0, 0, 0);
}

// ------------------------------------------------------------------ array access
DJB_DEV v3 load3(const View &v, long long k)
{
	long long off = k * v.stride;
	return mk(v.x[off], v.y[off], v.z[off]);
}
DJB_DEV void store3(const View &v, long long k, v3 a)
{
	long long off = k * v.stride;
	v.x[off] = a.x; v.y[off] = a.y; v.z[off] = a.z;
}

// ------------------------------------------------------------------ counter-based RNG (synth.py)
DJB_DEV uint32_t pcg(uint32_t x)
{
	uint32_t state = x * 747796405u + 2891336453u;
	uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
	return (word >> 22u) ^ word;
}
DJB_DEV uint32_t hash_u32(uint32_t seed, uint64_t k, uint32_t c)
{
	uint32_t h = pcg(seed + c * 0x9E3779B9u);
	h = pcg(h ^ (uint32_t)(k & 0xFFFFFFFFull));
	h = pcg(h + (uint32_t)(k >> 32));
	return h;
}
DJB_DEV float gen_uniform(uint32_t seed, uint64_t k)
{
	return (float)(hash_u32(seed, k, 0) >> 8) * 5.9604644775390625e-08f;   // 2^-24
}
DJB_DEV v3 gen_direction(uint32_t seed, uint64_t k)
{
	float x = (float)(hash_u32(seed, k, 0) >> 8) * 1.1920928955078125e-07f - 1.0f;   // 2^-23
	float y = (float)(hash_u32(seed, k, 1) >> 8) * 1.1920928955078125e-07f - 1.0f;
	float r2 = x * x + y * y;
	if (r2 >= 0.998f) { x *= 0.5f; y *= 0.5f; }
	float z = sqrtf((1.0f - x * x) - y * y);
	return mk(x, y, z);
}

// ================================================================== one unit of each batch operator
// The bodies the kernels (djb_kernels_eval.hip) and the host loops (djb_cpu.cpp) both run, one (i, o) pair /
// sample / query per call.
template <int KIND, int WANT, int FRK = -1>
DJB_DEV void eval_one(const Brdf &b, const Params &p, v3 i, v3 o, v3 &fr, float &pdf)
{
	if (KIND <= KIND_TABULAR || KIND == KIND_TABULAR_ANISO) {
		mf_eval_pdf<KIND, WANT, FRK>(b, p, i, o, fr, pdf);
	} else {
		if (WANT & 3) {
			v3 e;
			if (KIND == KIND_MERL) e = merl_eval(b, i, o);
			else if (KIND == KIND_UTIA) e = utia_eval(b, i, o);
			else if (KIND == KIND_SGD) e = sgd_eval(b, i, o);
			else if (KIND == KIND_ABC) e = abc_eval(b, i, o);
			else e = divs(mk(p.nx, p.ny, p.nz), F(DJB_PI));        // lambert: reflectance / M_PI, dj_brdf.h:861-868
			fr = (WANT & 2) ? scale(i.z, e) : e;                   // brdf::evalp, dj_brdf.h:803-806
		}
		if (WANT & 4) pdf = F(D(i.z) / DJB_PI);                    // brdf::pdf, dj_brdf.h:842-845
	}
}


// sample (IS == false) / evalp_is (IS == true) of one unit; FRK as in mf_eval_pdf
template <int KIND, bool IS, int FRK = -1>
DJB_DEV void sample_one(const Brdf &b, const Params &p, float u1, float u2, v3 o, const GlibcTabs &gt,
                        v3 &i_out, v3 &w_out, float &pdf_out)
{
	w_out = mk(0, 0, 0); pdf_out = 0.0f;
	if (KIND <= KIND_TABULAR || KIND == KIND_TABULAR_ANISO) {
		if (!IS) i_out = mf_sample<KIND>(b, p, u1, u2, o, gt);
		else {
			i_out = mk(0, 0, 0);
			w_out = mf_evalp_is<KIND, FRK>(b, p, u1, u2, o, i_out, pdf_out, gt);
		}
	} else {
		// brdf::sample / brdf::evalp_is defaults (cosine hemisphere), dj_brdf.h:816-845
		float x, y;
		uniform_to_concentric(u1, u2, x, y);
		i_out = mk(x, y, F(sqrt(1.0 - D(x * x) - D(y * y))));
		if (IS) {
			v3 fr; float pdf;
			eval_one<KIND, 6>(b, p, i_out, o, fr, pdf);
			w_out = divs(fr, pdf);
			pdf_out = pdf;
		}
	}
}

// per-pair microfacet::params.  MODE 0: pdfparams record (ax, ay, rho, tx, ty); MODE 1: LEAN moments (E1..E5)
// combined with the scaled base lobe, params = lrep_to_params(base + lean) (mitsuba/dj_beckmannconductor.cpp:291-319),
// optionally written back to out_pp5
template <int KIND, int WANT, int MODE, int FRK = -1>
DJB_DEV void pp_one(const Brdf &b, v3 i, v3 o, const float *r, const Lrep &base, float *out_pp5, v3 &fr, float &pdf)
{
	float ax, ay, rho, tx, ty;
	if (MODE == 0) { ax = r[0]; ay = r[1]; rho = r[2]; tx = r[3]; ty = r[4]; }
	else {
		Lrep l; l.E1 = r[0]; l.E2 = r[1]; l.E3 = r[2]; l.E4 = r[3]; l.E5 = r[4];
		lrep_to_pdfparams(lrep_add(base, l), ax, ay, rho, tx, ty);
		if (out_pp5) { out_pp5[0] = ax; out_pp5[1] = ay; out_pp5[2] = rho; out_pp5[3] = tx; out_pp5[4] = ty; }
	}
	Params p = params_from_pdfparams(ax, ay, rho, tx, ty);
	mf_eval_pdf<KIND, WANT, FRK>(b, p, i, o, fr, pdf);
}

// ------------------------------------------------------------------ microfacet / radial queries
// (dj_brdf.h:258-276, 307-314).  `which` is wave-uniform.
enum { Q_NDF = 0, Q_GAF, Q_G1, Q_SIGMA, Q_P22, Q_VP22, Q_VNDF, Q_FRESNEL,
       Q_P22_RADIAL = 16, Q_SIGMA_STD_RADIAL, Q_CDF_RADIAL, Q_QF_RADIAL, Q_QF2_RADIAL, Q_QF3_RADIAL, Q_QF1,
       Q_A_PDF1 = 32, Q_A_CDF1, Q_A_QF1, Q_A_PDF2, Q_A_CDF2, Q_A_QF2,
       Q_MODEL_NDF = 48, Q_MODEL_GAF, Q_MODEL_G1 };


template <int KIND>
DJB_DEV v3 query_one(const Brdf &b, const Params &p, int which, long long k, const View &va, const View &vb, const View &vc)
{
	v3 a = load3(va, k);
	v3 r = mk(0, 0, 0);
	switch (which) {
	case Q_NDF: r.x = mf_ndf<KIND>(b, a, p); break;
	case Q_GAF: {   // gaf(h, i, o): a = h (unused by Smith), vb = i, vc = o
		v3 i = load3(vb, k), o = load3(vc, k);
		float g1o = mf_g1_from_sigma(o, mf_sigma<KIND>(b, o, p), p);
		float g1i = b.shadow ? mf_g1_from_sigma(i, mf_sigma<KIND>(b, i, p), p) : 0.0f;
		r.x = mf_gaf_from_g1(b.shadow, g1i, g1o); break;
	}
	case Q_G1: { v3 kk = load3(vb, k); r.x = mf_g1_from_sigma(kk, mf_sigma<KIND>(b, kk, p), p); break; }
	case Q_SIGMA: r.x = mf_sigma<KIND>(b, a, p); break;
	case Q_P22: r.x = mf_p22<KIND>(b, a.x, a.y, p); break;
	case Q_VP22: case Q_VNDF: {
		v3 kk = load3(vb, k);
		v3 h = which == Q_VNDF ? a : normalize(mk(-a.x, -a.y, 1));
		float kh = dot(kk, h);
		float vn = D(kh) > 0.0 ? kh * mf_ndf<KIND>(b, h, p) / mf_sigma<KIND>(b, kk, p) : 0.0f;
		r.x = which == Q_VNDF ? vn : (h.z * h.z * h.z) * vn; break;
	}
	case Q_FRESNEL: r = fresnel_eval(b.fr, a.x); break;
	case Q_P22_RADIAL: r.x = p22_radial<KIND>(b, a.x); break;
	case Q_SIGMA_STD_RADIAL: r.x = sigma_std_radial<KIND>(b, a.x); break;
	case Q_CDF_RADIAL: r.x = cdf_radial<KIND>(b, a.x); break;
	case Q_QF_RADIAL: r.x = qf_radial<KIND>(b, a.x); break;
	case Q_QF2_RADIAL: r.x = KIND == KIND_BECKMANN ? beckmann_qf2_radial(a.x, a.y, a.z, glibc_tabs_global())
	                       : KIND == KIND_GGX ? ggx_qf2_radial(a.x, a.y, a.z) : 0.0f; break;
	case Q_QF3_RADIAL: r.x = KIND == KIND_BECKMANN ? beckmann_qf1(a.x, glibc_tabs_global())
	                       : KIND == KIND_GGX ? ggx_qf3_radial(a.x, a.y) : 0.0f; break;
	case Q_QF1: r.x = KIND == KIND_BECKMANN ? beckmann_qf1(a.x, glibc_tabs_global()) : KIND == KIND_GGX ? ggx_qf1(a.x) : 0.0f; break;
	// tabular_anisotropic::{pdf1, cdf1, qf1, pdf2, cdf2, qf2} (dj_brdf.h:450-455)
	case Q_A_PDF1: r.x = KIND == KIND_TABULAR_ANISO ? aniso_pdf1(b, a.x) : 0.0f; break;
	case Q_A_CDF1: r.x = KIND == KIND_TABULAR_ANISO ? aniso_cdf1(b, a.x) : 0.0f; break;
	case Q_A_QF1:  r.x = KIND == KIND_TABULAR_ANISO ? aniso_qf1(b, a.x) : 0.0f; break;
	case Q_A_PDF2: r.x = KIND == KIND_TABULAR_ANISO ? aniso_pdf2(b, a.x, a.y) : 0.0f; break;
	case Q_A_CDF2: r.x = KIND == KIND_TABULAR_ANISO ? aniso_cdf2(b, a.x, a.y) : 0.0f; break;
	case Q_A_QF2:  r.x = KIND == KIND_TABULAR_ANISO ? aniso_qf2(b, a.x, a.y) : 0.0f; break;
	}
	return r;
}

// sgd::{ndf, gaf, g1, fresnel} and abc::{ndf, gaf, fresnel} (dj_brdf.h:505-509, 530-533)
template <int KIND>
DJB_DEV v3 model_query_one(const Brdf &b, int which, long long k, const View &va, const View &vb, const View &vc)
{
	v3 a = load3(va, k), r = mk(0, 0, 0);
	switch (which) {
	case Q_FRESNEL: r = fresnel_eval(b.fr, a.x); break;
	case Q_MODEL_NDF: r = KIND == KIND_SGD ? sgd_ndf_rgb(b, a) : abc_ndf_rgb(b, a); break;
	case Q_MODEL_GAF: {
		v3 i = load3(vb, k), o = load3(vc, k);
		if (KIND == KIND_SGD) r = sgd_gaf_rgb(b, i, o);
		else r.x = abc_gaf(a, i, o);
		break;
	}
	case Q_MODEL_G1: if (KIND == KIND_SGD) r = sgd_g1_rgb(b, a); break;
	}
	return r;
}

// dj_brdf.h:1010-1023 applied once per table entry instead of once per lookup: n = 1458000 texels from 3*n doubles
DJB_DEV MerlTexel merl_convert_one(const double *s, long long n, long long k)
{
	float r = F(s[k] * (1.00 / 1500.0));
	float g = F(s[k + n] * (1.15 / 1500.0));
	float b = F(s[k + 2 * n] * (1.66 / 1500.0));
	if (D(r) < 0.0 || D(g) < 0.0 || D(b) < 0.0) r = g = b = 0.0f;
	MerlTexel t; t.x = r; t.y = g; t.z = b;
	return t;
}

// utia::normalize (dj_brdf.h:1162-1177) then the (float_t) cast of dj_brdf.h:1144: record e of the 288*288
// records of eight float4 (see utia_eval); n = 3*288*288 samples (three planes)
DJB_DEV void utia_convert_one(const double *s, long long n, long long e, float4 *table)
{
	const float kf = 1.f / 140.f;
	const long long plane = n / 3;
	// e = 288 * (48 * iti + ipi) + 48 * itv + ipv
	long long ipv = e % 48, itv = (e / 48) % 6, row = e / 288, ipi = row % 48;
	float t[32];
	for (int c = 0; c < 2; ++c)
		for (int k = 0; k < 2; ++k)
			for (int l = 0; l < 2; ++l) {
				long long tv = itv + c > 5 ? 5 : itv + c;
				long long src = 288 * (row - ipi + (ipi + k) % 48) + 48 * tv + (ipv + l) % 48;
				for (int ch = 0; ch < 3; ++ch) {
					double v = s[ch * plane + src] > 0.0 ? s[ch * plane + src] : 0.0;
					t[3 * (4 * c + 2 * k + l) + ch] = F(v * D(kf));
				}
			}
	for (int j = 24; j < 32; ++j) t[j] = 0.0f;
	for (int j = 0; j < 8; ++j) { float4 q; q.x = t[4 * j]; q.y = t[4 * j + 1]; q.z = t[4 * j + 2]; q.w = t[4 * j + 3]; table[8 * e + j] = q; }
}

} // namespace djbdev
