// djb_cpu_libm.cpp -- the kernels' restatements of glibc 2.35's libm functions, compiled FOR THE HOST, and the check that
// decides whether the host path needs them.
//
// The reference (jdupuy/dj_brdf, dj_brdf.h) calls the host's libm: exp / pow / atan2 / sin / cos / tan / acos in double
// (hdr:659, 685, 1634, 1868, 3419, 3431, 3612 ...) and logf / expf / powf in float (hdr:695, 1917, 1935).  The gfx950
// kernels reproduce what glibc 2.35's x86-64 FMA-ifunc variants return for those calls (djb_device.hpp, "restated"
// sections; sources restated: see NOTICE).  The product's host path (djb_cpu.cpp: CPU contexts and the scalar-size host
// calls of a GPU context) calls the host's own libm -- the same bits only on a host with that glibc and an FMA CPU.
// On any other host the two execution paths of one object would disagree in the last place, so:
//
//   * init() -- run once, by the first context created -- evaluates a fixed probe set (a few thousand arguments per
//     function, drawn from the ranges the BRDF code feeds them plus the special values) with the host's libm and with
//     the restatements compiled here, and compares every bit;
//   * if any differs, use_restated = 1: every one of these libm calls of the host path (the glibc_* / hl_* wrappers of
//     djb_device.hpp's host instantiation) runs the restatement instead.  Scalar path == batch path then holds by
//     construction on every host, and the values are what the reference returns on a glibc 2.35 / FMA host;
//   * one line on stderr says so, and djb_ctx_libm_matches_host() (include/djb_hip.h) reports it.
//   DJB_HOST_LIBM=restated | host overrides the choice (tests run the whole CPU suite under `restated`).
//
// atan (three float -> float sites of the tabulated lobes) and log (one site) have no restatement; atan_log_kat() checks
// the host's functions against known answers of glibc 2.35 so that a deviating host is at least reported.
//
// Built with -mfma (the restatements spell every fused multiply-add of the FMA ifunc variants as __builtin_fma; without
// hardware FMA each would be a libm call) and -ffp-contract=off; nothing here runs unless the CPU has FMA.
#define DJB_HOST_MATH 1
#define DJB_HOST_RESTATED 1
#include <gnu/libc-version.h>
#include "djb_device.hpp"

#include <stdio.h>
#include <stdlib.h>
#include <mutex>

namespace djbhostlibm {

int use_restated = 0;

double r_exp(double x) { return djbdev::glibc_exp(x, 0u); }
double r_pow(double x, double y) { return djbdev::glibc_pow(x, y, 0u, 0u); }
double r_atan2(double y, double x) { return djbdev::glibc_atan2(y, x); }
double r_sin(double x) { return djbdev::glibc_sin(x); }
double r_cos(double x) { return djbdev::glibc_cos(x); }
double r_tan(double x) { return djbdev::glibc_tan(x); }
double r_acos(double x) { return djbdev::glibc_acos(x, 0u); }
float r_logf(float x) { return djbdev::glibc_logf(x, djbdev::glibc_tabs_global()); }
float r_expf(float x) { return djbdev::glibc_expf(x, djbdev::glibc_tabs_global()); }
float r_powf(float x, float y) { return djbdev::glibc_powf(x, y, djbdev::glibc_tabs_global()); }

namespace {

// counter hash -> uniform in [0, 1)
inline double u01(uint32_t k, uint32_t stream)
{
	uint32_t x = k * 747796405u + stream * 2891336453u + 12345u;
	x = ((x >> ((x >> 28u) + 4u)) ^ x) * 277803737u;
	x = (x >> 22u) ^ x;
	return (double)x * (1.0 / 4294967296.0);
}
inline bool same(double a, double b) { uint64_t x, y; memcpy(&x, &a, 8); memcpy(&y, &b, 8); return x == y || (a != a && b != b); }
inline bool samef(float a, float b) { uint32_t x, y; memcpy(&x, &a, 4); memcpy(&y, &b, 4); return x == y || (a != a && b != b); }
// volatile function pointers: the host's functions are looked up at run time (an interposed libm is what is measured),
// and the compiler cannot fold a call on a constant
double (*volatile h_exp)(double) = exp; double (*volatile h_pow)(double, double) = pow;
double (*volatile h_atan2)(double, double) = atan2; double (*volatile h_sin)(double) = sin;
double (*volatile h_cos)(double) = cos; double (*volatile h_tan)(double) = tan;
double (*volatile h_acos)(double) = acos; double (*volatile h_atan)(double) = atan; double (*volatile h_log)(double) = log;
float (*volatile h_logf)(float) = logf; float (*volatile h_expf)(float) = expf; float (*volatile h_powf)(float, float) = powf;

constexpr int NPROBE = 4096;

// number of probe arguments on which the host libm and the restatement differ, per function (order: exp pow atan2
// sin cos tan acos logf expf powf)
void compare(int bad[10])
{
	for (int f = 0; f < 10; ++f) bad[f] = 0;
	for (int k = 0; k < NPROBE; ++k) {
		const double a = u01(k, 1), b = u01(k, 2), c = u01(k, 3);
		// exp: Beckmann arguments -nu^2, -r^2 in [-90, 0]; the sgd model's in [-40, 5]; a few near zero
		{ double x = (k & 3) == 0 ? -90.0 * a * a : (k & 3) == 1 ? -45.0 * a + 5.0 * b : (k & 3) == 2 ? -a * 1e-3 : (a - 0.5) * 1400.0;
		  bad[0] += !same(h_exp(x), r_exp(x)); }
		// pow: bases in (0, 1] and [1, 100], exponents 2.4 (sRGB), 5, and the sgd / abc rows' ranges
		{ double x = (k & 1) ? a + 1e-9 : 1.0 + 99.0 * a * a, y = (k & 3) == 0 ? (double)2.4f : (k & 3) == 1 ? 5.0 : (k & 3) == 2 ? -3.0 * b : 40.0 * b - 20.0;
		  bad[1] += !same(h_pow(x, y), r_pow(x, y)); }
		// atan2: components of unit vectors, all four quadrants, small denominators
		{ double y = 2.0 * a - 1.0, x = (k & 7) == 0 ? (2.0 * b - 1.0) * 1e-6 : 2.0 * b - 1.0;
		  bad[2] += !same(h_atan2(y, x), r_atan2(y, x)); }
		// sin / cos: angles in [-2 pi, 2 pi] (and a few up to 1e4); tan: [0, pi/2) and [-25, 25]
		{ double x = (k & 7) == 0 ? (a - 0.5) * 2e4 : (a - 0.5) * 4.0 * DJB_PI;
		  bad[3] += !same(h_sin(x), r_sin(x)); bad[4] += !same(h_cos(x), r_cos(x)); }
		{ double x = (k & 3) == 0 ? (a - 0.5) * 50.0 : a * (DJB_PI / 2.0) * 0.999999;
		  bad[5] += !same(h_tan(x), r_tan(x)); }
		// acos: [-1, 1], dense near the ends
		{ double x = (k & 3) == 0 ? 1.0 - a * a * 1e-3 : (k & 3) == 1 ? -1.0 + a * a * 1e-3 : 2.0 * a - 1.0;
		  bad[6] += !same(h_acos(x), r_acos(x)); }
		// logf: erfinv's (1 - u)(1 + u) in (0, 1]; expf: -erfinv^2 in [-30, 0]; powf: (1 - u)^fit, fit in [0.49, 1]
		{ float x = (float)((k & 1) ? a : a * a * a * 1e-3 + 1e-30);
		  bad[7] += !samef(h_logf(x), r_logf(x)); }
		{ float x = (float)((k & 1) ? -30.0 * a * a : (a - 0.5) * 170.0);
		  bad[8] += !samef(h_expf(x), r_expf(x)); }
		{ float x = (float)((k & 1) ? 1.0 - a * 0.999999 : a * 50.0 + 1e-6), y = (float)((k & 1) ? 0.49 + 0.51 * c : 6.0 * c - 3.0);
		  bad[9] += !samef(h_powf(x, y), r_powf(x, y)); }
	}
}

std::once_flag g_once;
int g_status = -1, g_kat = -1;

// glibc 2.35 x86-64 values of atan / log at fixed arguments (tools/gen_atan_log_kat.py); bit patterns
#include "djb_atan_log_kat.inc"

void do_init()
{
	const bool fma = __builtin_cpu_supports("fma");
	const char *env = getenv("DJB_HOST_LIBM");
	if (!fma) {
		g_status = -1;
		fprintf(stderr, "dj_brdf_amd: this CPU has no FMA: the host libm cannot be checked against the kernels' glibc 2.35 restatements; "
		                "scalar-size host calls follow the host's libm\n");
		return;
	}
	int bad[10];
	compare(bad);
	int total = 0;
	for (int f = 0; f < 10; ++f) total += bad[f];
	g_status = total == 0 ? 1 : 0;
	int kat_bad = 0;
	for (int k = 0; k < DJB_KAT_N; ++k) {
		double x, wa, wl; memcpy(&x, &DJB_KAT_ATAN_X[k], 8); memcpy(&wa, &DJB_KAT_ATAN_Y[k], 8);
		kat_bad += !same(h_atan(x), wa);
		memcpy(&x, &DJB_KAT_LOG_X[k], 8); memcpy(&wl, &DJB_KAT_LOG_Y[k], 8);
		kat_bad += !same(h_log(x), wl);
	}
	g_kat = kat_bad == 0 ? 1 : 0;
	use_restated = g_status == 0;
	// The probe set is a SAMPLE (4096 arguments per function): a libm that differs from glibc 2.35 on one argument in 1e4 ...
	// 1e6 -- e.g. a later glibc with correctly rounded float functions -- would pass it.  So the host libm is trusted only if
	// it also SAYS it is glibc 2.35; any other version runs the restatements (validated against 2.35 on every bit), which
	// keeps "scalar path == batch path" true by construction there as well.
	const char *ver = gnu_get_libc_version();
	const bool known_good = ver && !strcmp(ver, "2.35");
	if (g_status == 1 && !known_good) {
		use_restated = 1;
		fprintf(stderr, "dj_brdf_amd: host libc is glibc %s, not 2.35: the probe set found no difference, but it is a sample -- host-side calls "
		                "run the kernels' restatements of glibc 2.35's functions (DJB_HOST_LIBM=host keeps the host's)\n", ver ? ver : "?");
	}
	if (env && !strcmp(env, "restated")) use_restated = 1;
	if (env && !strcmp(env, "host")) use_restated = 0;
	if (g_status == 0) {
		static const char *names[10] = { "exp", "pow", "atan2", "sin", "cos", "tan", "acos", "logf", "expf", "powf" };
		char list[160]; size_t n = 0; list[0] = 0;
		for (int f = 0; f < 10; ++f)
			if (bad[f] && n + 24 < sizeof list) n += (size_t)snprintf(list + n, sizeof list - n, " %s(%d/%d)", names[f], bad[f], NPROBE);
		fprintf(stderr, "dj_brdf_amd: the host libm differs from glibc 2.35 (x86-64, FMA) in:%s -- host-side calls %s\n", list,
		        use_restated ? "run the kernels' restatements of those functions instead (same bits as the GPU batches)"
		                     : "keep the host libm (DJB_HOST_LIBM=host): scalar-size host calls may differ from GPU batches in the last place");
	}
	if (g_kat == 0)
		fprintf(stderr, "dj_brdf_amd: the host's atan / log differ from glibc 2.35's on the known-answer set (%d values): the float -> float sites "
		                "of the tabulated lobes that call them follow the host's libm\n", kat_bad);
}

} // namespace

int init() { std::call_once(g_once, do_init); return g_status; }
int atan_log_kat() { init(); return g_kat; }

} // namespace djbhostlibm
