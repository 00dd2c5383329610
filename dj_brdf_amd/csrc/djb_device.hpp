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
	unsigned int atan_lds;                  // the arctangent core's table (atan_tab_to_lds), 0 = not staged: the trig sites keep their previous forms
	// KIND_USER (host path only): the callbacks of a user-defined NDF (UserNdf below); never set on an object a kernel sees
	const void *user_ndf;
};

struct View { float *x, *y, *z; long long stride; };

// KIND_USER: a microfacet BRDF whose NDF is HOST CODE of the caller -- a class derived from djb::radial (its public virtuals
// p22_radial / sigma_std_radial / cdf_radial / qf_radial / qf2_radial / qf3_radial, dj_brdf.h:307-314) or from djb::microfacet
// (the protected virtuals p22_std / sigma_std / sample_vp22_std_*, dj_brdf.h:283-295).  Host instantiation only: everything
// around the NDF (params, sigma's stretch, G1 / G2, eval / pdf / sample / evalp_is, the queries) is the same per-unit code.
enum { KIND_BECKMANN = 0, KIND_GGX = 1, KIND_TABULAR = 2, KIND_MERL = 3, KIND_UTIA = 4, KIND_LAMBERT = 5,
       KIND_SGD = 6, KIND_ABC = 7, KIND_TABULAR_ANISO = 8, KIND_USER = 9 };
// same layout as djb_user_ndf (include/djb_hip.h)
struct UserNdf {
	void *user;
	int (*supports_smith_vndf_sampling)(void *);
	float (*p22_radial)(void *, float);                 // != NULL: a radial NDF
	float (*sigma_std_radial)(void *, float);
	float (*cdf_radial)(void *, float);
	float (*qf_radial)(void *, float);
	float (*qf2_radial)(void *, float, float, float);
	float (*qf3_radial)(void *, float, float);
	float (*p22_std)(void *, float, float);             // used when p22_radial == NULL
	float (*sigma_std)(void *, const float *);
	void (*sample_vp22_std)(void *, float, float, const float *, float *, float *);
};
#define DJB_IS_MICROFACET(K) ((K) <= KIND_TABULAR || (K) == KIND_TABULAR_ANISO || (K) == KIND_USER)
// the two tabulated microfacet classes sample with the non-VNDF "nmap" scheme (supports_smith_vndf_sampling() == false)
#define DJB_NMAP(K) ((K) == KIND_TABULAR || (K) == KIND_TABULAR_ANISO)
enum { FR_IDEAL = 0, FR_UNPOLARIZED = 1, FR_SCHLICK = 2, FR_SGD = 3, FR_SPLINE = 4 };

// ------------------------------------------------------------------ L0 helpers (dj_brdf.h:574-765)
DJB_DEV float F(double x) { return (float)x; }
DJB_DEV double D(float x) { return (double)x; }
DJB_DEV float fmin_(float a, float b) { return a < b ? a : b; }   // djb::min, dj_brdf.h:574
DJB_DEV float fmax_(float a, float b) { return a > b ? a : b; }   // djb::max, dj_brdf.h:575
DJB_DEV float sat_(float x) { return fmin_(1.0f, fmax_(0.0f, x)); }
// the same templates on doubles: NOT IEEE fmin / fmax -- djb::max(a, NaN) is NaN (the comparison is false and b is returned), which is
// how a NaN acos (an un-normalised direction with z > 1 in sgd::g1) or a NaN LEAN record reaches the reference's result
DJB_DEV double dmin_(double a, double b) { return a < b ? a : b; }
DJB_DEV double dmax_(double a, double b) { return a > b ? a : b; }

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
#if defined(DJB_HOST_MATH)
DJB_DEV float acos_u_f(float c) { return F(2.0 * hl_acos(D(c)) / DJB_PI); }                    // dj_brdf.h:1341 (spline fresnel)
DJB_DEV float acos_u32_f(float c) { return F(D(2.0f) * hl_acos(D(c)) / D(F(DJB_PI))); }       // dj_brdf.h:2158 (tabular sigma)
DJB_DEV float atan_squ_f(float r) { return F(sqrt(D(2.0f) * atan(D(r)) / D(F(DJB_PI)))); }  // dj_brdf.h:2152 (tabular p22)
DJB_DEV float atan_u_f(float r) { return F(atan(D(r)) * D(2.0f) / D(F(DJB_PI))); }          // dj_brdf.h:2165 (tabular cdf)
#else
// Device: the same four maps without their fp64 divisions by pi / float(pi).  num * RN(1 / c) is within 1.5 * 2^-52 of the
// IEEE quotient num / c, so it (or, for atan_squ, its square root: another 2^-52) rounds to the same float unless it sits
// next to a float rounding boundary (near_f32_midpoint, 256 ulp64 either side), where the reference's own expression
// decides.  Like every site they are swept over all 2^32 inputs against the host's values (tools/exhaustive_trig.py).
DJB_DEV bool near_f32_midpoint(double y, int width = 256);
DJB_DEV double sqrt_fast(double a);
DJB_DEV float div_c_to_f32(double num, double c, double rc)
{
	const double q = num * rc, aq = q < 0 ? -q : q;
	if (__builtin_expect(near_f32_midpoint(q) || !(aq > 1e-30 && aq < 1e30), 0)) return F(num / c);
	return F(q);
}
DJB_DEV float acos_u_f(float c) { return div_c_to_f32(2.0 * hl_acos(D(c)), DJB_PI, 0x1.45f306dc9c883p-2); }                    // dj_brdf.h:1341
DJB_DEV float acos_u32_f(float c) { return div_c_to_f32(D(2.0f) * hl_acos(D(c)), D(F(DJB_PI)), 0x1.45f306446f9b4p-2); }        // dj_brdf.h:2158
DJB_DEV float atan_u_f(float r) { return div_c_to_f32(atan(D(r)) * D(2.0f), D(F(DJB_PI)), 0x1.45f306446f9b4p-2); }             // dj_brdf.h:2165
DJB_DEV float atan_squ_f(float r)                                                                                                 // dj_brdf.h:2152
{
	const double num = D(2.0f) * atan(D(r));
	const double v = num * 0x1.45f306446f9b4p-2;
	const double g = sqrt_fast(v);
	if (__builtin_expect(near_f32_midpoint(g) || !(v > 1e-30 && v < 1e30), 0)) return F(sqrt(num / D(F(DJB_PI))));
	return F(g);
}
#endif
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
DJB_DEV float div_to_f32(double num, double den) { return F(num / den); }
DJB_DEV float div_pi_to_f32(double num) { return F(num / DJB_PI); }
DJB_DEV float div_2pi_to_f32(double num) { return F(num / (2.0 * DJB_PI)); }
DJB_DEV float inv_sqrt_pi_f() { return inversesqrt_(F(DJB_PI)); }
#else
// ---- guarded fast paths for float(<double expression>) -----------------------------------------
// The reference rounds a correctly-rounded double result e to float.  A cheaper double y with
// |y - e| <= 2^-44 |e| rounds to the SAME float unless y lies within 2^-44 (relative) of a float
// rounding boundary, i.e. of a double whose low 29 mantissa bits are 0x10000000.  near_f32_midpoint
// tests that (256 ulp64 either side; probability 2^-20), and the caller then takes the exact path.
// y comes from v_rsq_f64 / v_rcp_f64 refined by two Newton steps (error <= a few 2^-53 for any
// seed accuracy >= 2^-14); e itself is within 2^-52 of the true value.
DJB_DEV bool near_f32_midpoint(double y, int width)
{
	// the 29 mantissa bits a float does not keep sit in the low word.  With x = lo & 0x1FFFFFFF the test is
	// |x - 2^28| <= width, i.e. x - (2^28 - width) in [0, 2 width]; lo << 3 = 8 x (mod 2^32) drops the three bits the
	// float keeps, so one shift-add and one unsigned compare do it: 8 (x - (2^28 - width)) <= 16 width, negatives wrap to
	// >= 2^31 (checked over all 2^32 low words, tools/near_midpoint_check.c).  2 VALU (v_lshl_add_u32, v_cmp) instead of 5.
	const unsigned int d = ((unsigned int)__double2loint(y) << 3) + (0u - (((unsigned int)0x10000000 - (unsigned int)width) << 3));
	return d <= ((unsigned int)width << 4);
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
// float(num / den) for doubles: num * recip_fast(den) is within 2^-50 (relative) of the correctly rounded double quotient
// (reciprocal <= 1.5 * 2^-52, one more rounding, the quotient's own 2^-53), so it rounds to the same float unless it sits
// within near_f32_midpoint's 256 ulp64 of a float rounding boundary; results outside the normal float range and
// divisors outside recip_fast's take the IEEE division.  ~18 issue slots instead of ~32 (v_rcp_f64 alone costs 5.9).
DJB_DEV float div_to_f32(double num, double den)
{
	const double q = num * recip_fast(den);
	const double aq = q < 0 ? -q : q, ad = den < 0 ? -den : den;
	if (__builtin_expect(near_f32_midpoint(q) || !(aq > 1e-30 && aq < 1e30) || !(ad > 1e-30 && ad < 1e30), 0))
		return F(num / den);
	return F(q);
}
// float(num / pi): num * RN(1 / pi) is within 1.5 * 2^-52 of the IEEE quotient -- same guard, no division at all
DJB_DEV float div_pi_to_f32(double num)
{
	const double q = num * 0x1.45f306dc9c883p-2;
	const double aq = q < 0 ? -q : q;
	if (__builtin_expect(near_f32_midpoint(q) || !(aq > 1e-30 && aq < 1e30), 0)) return F(num / DJB_PI);
	return F(q);
}
// float(num / (2.0 * pi)) (the azimuth coordinate of tabular_anisotropic's quantile grid, dj_brdf.h:2814): 2 pi and RN(1 / pi) / 2 are exact scalings
DJB_DEV float div_2pi_to_f32(double num)
{
	const double q = num * 0x1.45f306dc9c883p-3;
	const double aq = q < 0 ? -q : q;
	if (__builtin_expect(near_f32_midpoint(q) || !(aq > 1e-30 && aq < 1e30), 0)) return F(num / (2.0 * DJB_PI));
	return F(q);
}
// inversesqrt(float(pi)) = float(1.0 / sqrt(double(3.14159274f))): a constant (the compiler does not fold v_rsq_f32 of a literal)
DJB_DEV float inv_sqrt_pi_f() { return 0x1.20dd74p-1f; }
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
#include "djb_glibc_restated_f64.inc"
#endif

// A&S 7.1.26 as the reference writes it, dj_brdf.h:667-688
// e must be exp(double(-x*x)) (the same for +x and -x): callers that need that exponential
// themselves (beckmann_qf2_radial) evaluate the fp64 exp once
// the part before the exponential: poly(t) t as the reference rounds it, and the sign
DJB_DEV float erf_poly_t(float x, float &sign)
{
	const float a1 = 0.254829592f, a2 = -0.284496736f, a3 = 1.421413741f,
	            a4 = -1.453152027f, a5 = 1.061405429f, p = 0.3275911f;
	sign = x < 0 ? -1.0f : 1.0f;
	x = fabsf(x);
	float t = recip_to_f32(1.0 + D(p * x));
	float poly = ((((a5 * t + a4) * t) + a3) * t + a2) * t + a1;
	return poly * t;
}
DJB_DEV float erf_given_exp(float x, double e)
{
	float sign;
	const float pt = erf_poly_t(x, sign);
	float y = F(1.0 - D(pt) * e);
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
#include "djb_glibc_restated_f32.inc"
#endif

// (float(double(r) * cos(double(phi))), float(double(r) * sin(double(phi)))): the polar -> slope step of the normal-map sampling scheme
// (radial::sample_vp22_std_nmap, dj_brdf.h:1806-1816; tabular_anisotropic, :2828; Cline's concentric map, :745-746).
#if defined(DJB_HOST_MATH)
DJB_DEV void polar_to_f32(float r, float phi, float &x, float &y) { x = F(D(r) * glibc_cos(D(phi))); y = F(D(r) * glibc_sin(D(phi))); }
#else
// Device (round 6): glibc's sin and cos -- table-driven, ~135 fp64 instructions each as restated -- were 270 of the 590 VALU instructions
// of a tabular `sample`.  Guarded shortcut, the pattern of div_to_f32: sine and cosine from a Cody-Waite reduction by pi/2 (two fma
// steps; |phi| <= 8, so k <= 5 and the reduced argument keeps full relative accuracy for a float-valued phi) and fdlibm's kernel
// polynomials (|error| <= 2^-58 on |t| <= pi/4): within 2 ulp64 of the host libm's values for EVERY float phi in [-8, 8]
// (tools/sincos_fast_check.c, all 2.2e9 arguments, profiles/r06/sincos_fast_check.txt).  The products then differ from the reference's
// doubles by < 4 ulp64 and round to the same floats unless one sits within 256 ulp64 of a float rounding boundary (near_f32_midpoint,
// probability 2^-19 per sample); there, for zero / subnormal-range / huge products and for phi outside [-8, 8] or NaN, glibc's own
// algorithms decide -- the previous code, verbatim.
// fma(a, b, k) with a literal addend k held in an SGPR pair.  The compiler selects v_fmac_f64 for a Horner step -- a tied accumulator, so
// the coefficient is first moved into a VGPR pair (two v_mov_b32: 3 issue slots per step instead of 1.6); v_fma_f64 takes it as an SGPR
// operand.  The same operation, the same bits.  Only for literals (an "s" operand that is not uniform would be read from one lane).
DJB_DEV double fma_sk(double a, double b, double k)
{
	double r;
	asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(k));
	return r;
}
DJB_DEV void sincos_fast(double x, double &s, double &c)
{
	const double kf = __builtin_rint(x * 0x1.45f306dc9c883p-1);                    // 2 / pi
	double t = __builtin_fma(-kf, 0x1.921fb54442d18p+0, x);                        // pi / 2: its high 53 bits ...
	t = __builtin_fma(-kf, 0x1.1a62633145c07p-54, t);                              // ... and the next 53
	const double z = t * t;
	double ps = fma_sk(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);   // the coefficients as SGPR operands (fma_sk): same operations
	ps = fma_sk(z, ps, 2.75573137070700676789e-06);
	ps = fma_sk(z, ps, -1.98412698298579493134e-04);
	ps = fma_sk(z, ps, 8.33333333332248946124e-03);
	ps = fma_sk(z, ps, -1.66666666666666324348e-01);
	const double st = __builtin_fma(t * z, ps, t);
	double pc = fma_sk(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
	pc = fma_sk(z, pc, -2.75573143513906633035e-07);
	pc = fma_sk(z, pc, 2.48015872894767294178e-05);
	pc = fma_sk(z, pc, -1.38888888888741095749e-03);
	pc = fma_sk(z, pc, 4.16666666666666019037e-02);
	const double ct = __builtin_fma(z * z, pc, __builtin_fma(z, -0.5, 1.0));
	const int q = (int)kf & 3;
	const double a = (q & 1) ? ct : st, b = (q & 1) ? st : ct;                     // the quadrant: sin takes a, cos takes b
	s = (q & 2) ? -a : a;
	c = ((q + 1) & 2) ? -b : b;
}
DJB_DEV void polar_to_f32(float r, float phi, float &x, float &y)
{
	const double ph = D(phi);
	double s, c;
	sincos_fast(ph, s, c);
	const double px = D(r) * c, py = D(r) * s;
	const double ax = px < 0 ? -px : px, ay = py < 0 ? -py : py;
	const bool fast = (ph >= -8.0) & (ph <= 8.0) & (ax > 1e-30) & (ax < 1e30) & (ay > 1e-30) & (ay < 1e30)
	                & !near_f32_midpoint(px) & !near_f32_midpoint(py);
	if (__builtin_expect(!fast, 0)) { x = F(D(r) * glibc_cos(ph)); y = F(D(r) * glibc_sin(ph)); return; }
	x = F(px); y = F(py);
}
#endif

// ---- one branch-free fp64 arctangent for every float -> float trig site of the table-driven kinds (round 6).  The device libm's fp64
// acos / atan / atan2 were 436 of the utia kernel's ~850 VALU instructions per pair and a third of the tabulated lobes' (both arms of
// acos under divergent exec masks; an IEEE division inside atan2; Horner steps as v_fmac_f64, whose tied accumulator makes the compiler
// move every coefficient into a VGPR pair first: 3 issue slots per step).  A site only needs a double within a guard of the libm's --
// anything further from a float rounding boundary than the core's own error rounds to the same float, the rest takes the site's
// previous form (djb_device.hpp above: the forms the exhaustive sweeps were run on) -- so all of them share
//     atan(mn / mx), 0 <= mn <= mx:  centre c = j / 8 from an fp32 estimate of the quotient (j = 0..8), then
//     atan(mn / mx) = atan(c) + atan(r),  r = (mn - c mx) / (mx + c mn),  |r| <= 1/16 (+ the estimate's error)
// with atan(c) from a 9-entry table in LDS, ONE division (v_rcp_f32 seed, two Newton steps) and r (1 - s/3 + s^2/5 - s^3/7 + s^4/9 -
// s^5/11), s = r^2 (truncation 2^-51); the coefficients ride in SGPRs (fma_sk).  acos(z) = atan2(sqrt(1 - z^2), z).  Error of the
// double: a few 2^-52 relative (division 2^-51, polynomial 2^-50, sqrt 2^-50, ~20 roundings) against a guard of 4096 ulp64 = 2^-40.
// Measured by djb_selftest_fast_trig: the one-argument sites against their previous forms over ALL 2^32 floats -- identical, decided
// or not -- and 2^33 (y, x) pairs for atan2; largest distance of a decided double from the device libm's: 3 ulp64
// (profiles/r06/fast_trig_selftest.txt; tests/test_gpu_verification.py).
#if !defined(DJB_HOST_MATH)
constexpr double DJB_ATAN_EIGHTHS[9] = { 0.0, 0x1.fd5ba9aac2f6ep-4, 0x1.f5b75f92c80ddp-3, 0x1.6f61941e4def1p-2, 0x1.dac670561bb4fp-2,
                                         0x1.1e00babdefeb4p-1, 0x1.4978fa3269ee1p-1, 0x1.700a7c5784634p-1, 0x1.921fb54442d18p-1 };
constexpr int ATAN_GUARD = 4096;
// the table staged in LDS by a kernel: 16 doubles reserved, caller: __syncthreads() afterwards.  Brdf::atan_lds carries the handle (LdsTab)
DJB_DEV LdsTab atan_tab_to_lds(double *lds, int tid)
{
	if (tid < 9) lds[tid] = DJB_ATAN_EIGHTHS[tid];
	return 1u + (unsigned int)(uintptr_t)(lds_f64p)lds;
}
DJB_DEV lds_f64p atan_tab(LdsTab AT) { return (lds_f64p)(uintptr_t)(AT - 1u); }
DJB_DEV double atan_core(double mn, double mx, lds_f64p T)
{
	const float jf = __builtin_rintf(8.0f * (F(mn) * __builtin_amdgcn_rcpf(F(mx))));
	const int j = (int)fminf(fmaxf(jf, 0.0f), 8.0f);
	const double c = 0.125 * D((float)j);
	const double num = __builtin_fma(-c, mx, mn), den = __builtin_fma(c, mn, mx);
	double rc = D(__builtin_amdgcn_rcpf(F(den)));
	double e = __builtin_fma(-den, rc, 1.0);
	rc = __builtin_fma(rc, e, rc);
	e = __builtin_fma(-den, rc, 1.0);
	rc = __builtin_fma(rc, e, rc);
	const double r = num * rc, s = r * r;
	double p = s * (-1.0 / 11.0) + (1.0 / 9.0);
	p = fma_sk(p, s, -1.0 / 7.0);
	p = fma_sk(p, s, 1.0 / 5.0);
	p = fma_sk(p, s, -1.0 / 3.0);
	return T[j] + __builtin_fma(r * s, p, r);
}
// acos(x) for -1 <= x <= 1 and atan(r) for 0 <= r < 1e18, as doubles; ok = false outside (and where 1 - x^2 is too small for the core)
DJB_DEV double acos_fast(float x, lds_f64p T, bool &ok)
{
	const double zd = D(fabsf(x));
	const double w = __builtin_fma(-zd, zd, 1.0);            // 1 - x^2: one rounding (exact where it is small)
	const double sn = sqrt_fast(w);
	const bool swap = sn > zd;
	double a = atan_core(swap ? zd : sn, swap ? sn : zd, T);
	a = swap ? 0x1.921fb54442d18p+0 - a : a;
	ok = w > 1e-18;                                          // false for |x| > 1 and NaN as well
	return x < 0.0f ? 0x1.921fb54442d18p+1 - a : a;
}
DJB_DEV double atan_fast(double r, lds_f64p T, bool &ok)
{
	const bool swap = r > 1.0;
	double a = atan_core(swap ? 1.0 : r, swap ? r : 1.0, T);
	ok = (r >= 0.0) & (r < 1e18);
	return swap ? 0x1.921fb54442d18p+0 - a : a;
}
DJB_DEV bool fast_decided(double d) { const double ad = d < 0.0 ? -d : d; return !near_f32_midpoint(d, ATAN_GUARD) & (ad > 1e-30) & (ad < 1e30); }
// the one-argument sites: the site's float from the core (ok = decided), its previous form, and the two together
// (AT = Brdf::atan_lds; 0: the kernel staged no table -- the previous form)
enum { FT_ACOS = 0, FT_ACOS_U, FT_ACOS_U32, FT_ATAN_U, FT_ATAN_SQU, FT_ATAN_SQRT, FT_SITES };
template <int S> DJB_DEV float trig_fast(float x, lds_f64p T, bool &ok)
{
	bool in;
	double d;
	if (S == FT_ACOS) d = acos_fast(x, T, in);                                                           // float(acos(double x))
	else if (S == FT_ACOS_U) d = (2.0 * acos_fast(x, T, in)) * 0x1.45f306dc9c883p-2;                       // float(2 acos(x) / pi)
	else if (S == FT_ACOS_U32) d = (2.0 * acos_fast(x, T, in)) * 0x1.45f306446f9b4p-2;                     // float(2 acos(x) / double(float(pi)))
	else if (S == FT_ATAN_U) d = (atan_fast(D(x), T, in) * 2.0) * 0x1.45f306446f9b4p-2;                    // float(atan(x) 2 / double(float(pi)))
	else if (S == FT_ATAN_SQU) {                                                                          // float(sqrt(2 atan(x) / double(float(pi))))
		const double v = (2.0 * atan_fast(D(x), T, in)) * 0x1.45f306446f9b4p-2;
		d = sqrt_fast(v);
		in &= v > 1e-30;
	} else {                                                                                              // float(atan(sqrt(double x)))
		const double xd = D(x);
		d = atan_fast(sqrt_fast(xd), T, in);
		in &= (xd > 1e-30) & (xd < 1e30);
	}
	ok = in & fast_decided(d);
	return F(d);
}
template <int S> DJB_DEV float trig_prev(float x)
{
	return S == FT_ACOS ? acos_f(x) : S == FT_ACOS_U ? acos_u_f(x) : S == FT_ACOS_U32 ? acos_u32_f(x) : S == FT_ATAN_U ? atan_u_f(x)
	     : S == FT_ATAN_SQU ? atan_squ_f(x) : atan_sqrt_f(x);
}
template <int S> DJB_DEV float trig_at(float x, LdsTab AT)
{
	if (AT) { bool ok; const float r = trig_fast<S>(x, atan_tab(AT), ok); if (__builtin_expect(ok, 1)) return r; }
	return trig_prev<S>(x);
}
DJB_DEV float acos_f(float x, LdsTab AT) { return trig_at<FT_ACOS>(x, AT); }                 // dj_brdf.h:650-661 (aniso sigma)
DJB_DEV float acos_u_f(float c, LdsTab AT) { return trig_at<FT_ACOS_U>(c, AT); }            // dj_brdf.h:1341
DJB_DEV float acos_u32_f(float c, LdsTab AT) { return trig_at<FT_ACOS_U32>(c, AT); }        // dj_brdf.h:2158
DJB_DEV float atan_u_f(float r, LdsTab AT) { return trig_at<FT_ATAN_U>(r, AT); }            // dj_brdf.h:2165
DJB_DEV float atan_squ_f(float r, LdsTab AT) { return trig_at<FT_ATAN_SQU>(r, AT); }        // dj_brdf.h:2152
DJB_DEV float atan_sqrt_f(float x, LdsTab AT) { return trig_at<FT_ATAN_SQRT>(x, AT); }      // dj_brdf.h:2285
// float(scale * atan2(double y, double x)) decided by the core, or ok = false
DJB_DEV float atan2_fast_f32(float y, float x, double scale, lds_f64p T, bool &ok, double *as_double = nullptr)
{
	const float ayf = fabsf(y), axf = fabsf(x);
	const bool swap = ayf > axf;
	const double mx = D(swap ? ayf : axf), mn = D(swap ? axf : ayf);
	double a = atan_core(mn, mx, T);
	a = swap ? 0x1.921fb54442d18p+0 - a : a;
	a = x < 0.0f ? 0x1.921fb54442d18p+1 - a : a;
	a = (__float_as_uint(y) >> 31) ? -a : a;                  // the sign BIT: atan2(-0, x < 0) is -pi
	const double d = scale * a;
	ok = (mx > 1e-18) & (mx < 1e18) & fast_decided(d);
	if (as_double) *as_double = d;
	return F(d);
}
DJB_DEV float atan2_to_f32(float y, float x, double scale, LdsTab AT)                                      // dj_brdf.h:659, 1634
{
	if (AT) { bool ok; const float r = atan2_fast_f32(y, x, scale, atan_tab(AT), ok); if (__builtin_expect(ok, 1)) return r; }
	return atan2_to_f32(y, x, scale);
}
// float(tan(double x)) (the quantile -> slope step of the tabulated lobes' sampling, dj_brdf.h:2171-2176, 2828): sine over cosine from
// sincos_fast (2 ulp64 each for |x| <= 8) through one reciprocal, decided like the sites above; the device libm's tan otherwise.
// djb_selftest_fast_trig mode 9: identical to tan_f over all 2^32 floats.
DJB_DEV float tan_fast_f32(float x, bool &ok)
{
	const double xd = D(x);
	double sn, cs;
	sincos_fast(xd, sn, cs);
	const double q = sn * recip_fast(cs), ac = cs < 0.0 ? -cs : cs;
	ok = (xd >= -8.0) & (xd <= 8.0) & (ac > 1e-30) & fast_decided(q);
	return F(q);
}
DJB_DEV float tan_fast_f(float x)
{
	bool ok;
	const float r = tan_fast_f32(x, ok);
	if (__builtin_expect(ok, 1)) return r;
	return tan_f(x);
}
#else
// host: the expressions themselves
DJB_DEV float tan_fast_f(float x) { return tan_f(x); }
DJB_DEV float acos_f(float x, LdsTab) { return acos_f(x); }
DJB_DEV float acos_u_f(float c, LdsTab) { return acos_u_f(c); }
DJB_DEV float acos_u32_f(float c, LdsTab) { return acos_u32_f(c); }
DJB_DEV float atan_u_f(float r, LdsTab) { return atan_u_f(r); }
DJB_DEV float atan_squ_f(float r, LdsTab) { return atan_squ_f(r); }
DJB_DEV float atan_sqrt_f(float x, LdsTab) { return atan_sqrt_f(x); }
DJB_DEV float atan2_to_f32(float y, float x, double scale, LdsTab) { return atan2_to_f32(y, x, scale); }
#endif

#if !defined(DJB_HOST_MATH) || defined(DJB_HOST_RESTATED)
#include "djb_fast_models.inc"      // flog / fexp and the decided fast tier of the sgd / abc chains (device; the host only compiles it for tools/sgd_fast_check.cpp)
#endif
#include "djb_device_microfacet.inc"
#include "djb_device_tables.inc"
#include "djb_device_units.inc"

} // namespace djbdev
