// djb_kernels_sample.hip -- Beckmann VNDF sampling (sample / evalp_is, BASELINE configs[3]) as a two-path kernel.
//
// The reference's sampler (dj_brdf.h:1669-1700, 1806-1846, 1897-1952) is one long dependent chain per sample with a
// data-dependent Newton loop (2.7 % of the samples take 2 trips, 72 % take 3, 25 % take 4, 0.5 % up to 9) and a dozen
// rarely taken arms: Giles' tail polynomial of erfinv (an fp64 sqrt), the special cases of glibc's logf / expf / powf /
// exp, the exact fall-backs of the guarded fp64 shortcuts.  Each arm is taken by well under 1 % of the lanes but, at
// 64 lanes a wave, by 20-30 % of the waves, and one slow lane keeps its wave in the loop: the one-kernel form
// (k_sample in djb_kernels_eval.hip) pays 4.85 trip-equivalents for a mean of 3.2.
//
// Here every lane runs the COMMON path only: the same operations in the same order as the reference on that path --
// every result it produces is the reference's, bit for bit -- with exactly four Newton trips, no arm at all, and a
// `rare` flag raised wherever the reference would have left the common path (any arm above, a fifth trip, a
// degenerate direction).  Flagged samples are not stored; their inputs go to a per-wave queue in LDS, and whenever a
// wave has collected 64 of them it runs the full per-sample code (sample_one, the one the one-kernel form runs) on
// them as one dense wave.  1.3 % of the samples take that route on the bench distribution (by cause: profiles/r03/
// beckmann_sample_two_path.txt, section 3; the DJB_EXP_RARE_COUNT build of this file counts them).  A flag is only raised where
// the operands at that call site can actually meet the special case (the range arguments stand next to each site).  Nothing is approximated
// anywhere: both paths are the reference's arithmetic, the split is by control flow only.
#include "djb_internal.hpp"
#include <stdio.h>
#include <stdlib.h>

using namespace djbdev;

namespace {

constexpr int BLOCK = 256;
constexpr int WAVES = BLOCK / 64;
constexpr unsigned int QCAP = 128;            // queue slots per wave: < 64 waiting + <= 64 new per iteration
constexpr int TRIPS = 4;                      // Newton trips of the common path; a sample that needs more is deferred

// why a sample leaves the common path (the site numbers only matter to the measurement build, DJB_EXP_RARE_COUNT)
enum { R_LOGF = 0, R_EXPF, R_POWF, R_EXP64, R_GUARD, R_TAIL_LOOP, R_TAIL_QF1, R_TRIPS, R_CLAMP, R_DEGENERATE, R_SITES };
struct Rare {
	bool any = false;
#ifdef DJB_EXP_RARE_COUNT
	unsigned int bits = 0;
	DJB_DEV void flag(int site, bool c) { any |= c; bits |= c ? 1u << site : 0u; }
	DJB_DEV void merge(const Rare &r, bool act) { any |= act & r.any; bits |= act ? r.bits : 0u; }
#else
	DJB_DEV void flag(int, bool c) { any |= c; }
	DJB_DEV void merge(const Rare &r, bool act) { any |= act & r.any; }
#endif
};
#ifdef DJB_EXP_RARE_COUNT          // measurement builds only (tools/exp): samples on the common / deferred path, and per site
__device__ unsigned long long g_rare[2 + R_SITES];
#endif
#ifdef DJB_EXP_TRIP4_CHECK         // measurement builds only: {samples that need trip 4, shown to converge there, shown but NOT converged (must be 0),
__device__ unsigned int g_trip4_printed;
__device__ unsigned long long g_trip4[4];   //                not shown although they converge (the price: they take the exact path)}
#endif

// Measured on 2.5e8 and 1e9 samples (profiles/r03/beckmann_sample_two_path.txt, section 4): one workgroup per resident slot (256 CUs x 5)
// is 19 % slower than ~48 tiles per workgroup -- the slots do not finish together -- and one tile per workgroup 60 % slower
// (table staging and a nearly empty queue drain per tile).
constexpr long long TILES_PER_WG = 48;
inline int grid_persistent(long long n)
{
	const long long tiles = (n + BLOCK - 1) / BLOCK;
	long long blocks = (tiles + TILES_PER_WG - 1) / TILES_PER_WG;
	if (blocks < 5120) blocks = tiles < 5120 ? tiles : 5120;      // small batches: fill the chip first
	if (blocks > 0x7fffffffLL) blocks = 0x7fffffffLL;
	if (blocks < 1) blocks = 1;
	return (int)blocks;
}

// ---- the common paths of the float libm restatements (djb_glibc_restated_f32.inc), special cases flagged instead of taken
// RANGE_FLAG = false: the caller knows x is a float in [0, 2) -- never negative, Inf or NaN -- and flags the sample itself
// whenever this returns <= -5.  That covers glibc's other special case, x zero or sub-normal: for ix < 0x00800000 the code
// below has k = -127 or -126, z in [0.699, 1.399) and |r| <= 1/32, i.e. it returns k ln2 + logc + log1p(r) <= -86.9.
template <bool RANGE_FLAG>
DJB_DEV float logf_main(float x, const GlibcTabs &gt, Rare &rare)
{
	const double *T = gt.logf;
	constexpr double Ln2 = DJB_GLIBC_LOGF_C[0], A0 = DJB_GLIBC_LOGF_C[1], A1 = DJB_GLIBC_LOGF_C[2], A2 = DJB_GLIBC_LOGF_C[3];
	const unsigned int ix = __float_as_uint(x);
	// x == 1 needs no arm: table entry 9 is {1, 0}, so the polynomial below returns +0 like glibc's shortcut
	if (RANGE_FLAG) rare.flag(R_LOGF, ix - 0x00800000u >= 0x7f800000u - 0x00800000u);
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
// |x| < 88 or NaN only (glibc's overflow / underflow arms start at 88): the one caller passes -ie^2 with |ie| <= 2.2, see there
DJB_DEV float expf_main(float x, const GlibcTabs &gt)
{
	constexpr double Shift = DJB_GLIBC_EXP2F_C[4], InvLn2N = DJB_GLIBC_EXP2F_C[5];
	double xd = D(x), z = InvLn2N * xd;
	double kd = z + Shift;
	unsigned long long ki = (unsigned long long)__double_as_longlong(kd);
	kd -= Shift;
	double r = __builtin_fma(InvLn2N, xd, -kd);
	return glibc_exp2_tail(ki, r, DJB_GLIBC_EXP2F_C[6], DJB_GLIBC_EXP2F_C[7], DJB_GLIBC_EXP2F_C[8], gt);
}
// x a normal positive float, y finite and non-zero, |y log2 x| < 126 (glibc's special cases are exactly the complement): the one
// caller passes x = 1 - u in [1e-5, 1 - 1e-6] and y = fit in [0.49, 1], see there
DJB_DEV float powf_main(float x, float y, const GlibcTabs &gt)
{
	const double *T = gt.powlog;
	constexpr double A0 = DJB_GLIBC_POWF_C[0], A1 = DJB_GLIBC_POWF_C[1], A2 = DJB_GLIBC_POWF_C[2], A3 = DJB_GLIBC_POWF_C[3],
	                 A4 = DJB_GLIBC_POWF_C[4], ShiftScaled = DJB_GLIBC_EXP2F_C[0];
	const unsigned int ix = __float_as_uint(x);
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
	double kd = ylogx + ShiftScaled;
	unsigned long long ki = (unsigned long long)__double_as_longlong(kd);
	kd -= ShiftScaled;
	double rr = __builtin_fma(D(y), logx, -kd);
	return glibc_exp2_tail(ki, rr, DJB_GLIBC_EXP2F_C[1], DJB_GLIBC_EXP2F_C[2], DJB_GLIBC_EXP2F_C[3], gt);
}
// glibc's fp64 exp of a non-positive argument, main path (glibc_exp_inline without its cold branch).  x <= -512 returns 0
// instead of glibc's tiny value: the two places the sampler uses e = exp(-cot^2) -- 1 - poly t e in erf and 2 + k e in the
// normalisation, both in double -- give the same result for every e below 2^-60, let alone below exp(-512)
// The one caller passes x = -cot_k^2 with the sign bit set and |x| <= 8.5e6 (see bk_qf2_common): of glibc's special cases only
// the tiny argument (|x| < 2^-54, where it returns 1 + x) can occur.
DJB_DEV double exp_main_neg(double x, LdsTab T, Rare &rare)
{
	const unsigned int hx = (unsigned int)__double2hiint(x), abstop = (hx >> 20) & 0x7ffu;
	const bool big = abstop >= 0x408u;                                   // |x| >= 512
	rare.flag(R_EXP64, abstop < 0x3c9u);
	unsigned int klo, sh; int sl;
	const double tmp = glibc_exp_tmp(x, 0.0, klo, sh, sl, T);
	const double scale = __hiloint2double((int)sh, sl);
	return big ? 0.0 : __builtin_fma(scale, tmp, scale);
}
// ---- the guarded fp64 shortcuts (djb_device.hpp) with the exact fall-back flagged instead of taken
// RANGE: what the caller does NOT know about the operand -- R_BOTH: nothing (the shortcut's domain is (1e-30, 1e30), anything else
// is flagged); R_UPPER: it is >= 1 or Inf / NaN; R_NONE: it is inside the domain for every sample no other flag has caught
enum { R_NONE = 0, R_UPPER, R_BOTH };
template <int RANGE>
DJB_DEV float inversesqrt_g(float x, Rare &rare)
{
	const double y = inversesqrt_fast(x);
	rare.flag(R_GUARD, near_f32_midpoint(y) | (RANGE == R_BOTH ? !((x > 1e-30f) & (x < 1e30f)) : RANGE == R_UPPER ? !(x < 1e30f) : false));
	return F(y);
}
template <int RANGE>
DJB_DEV float recip_g(double q, Rare &rare)
{
	const double r = recip_fast(q);
	const double aq = q < 0 ? -q : q;
	rare.flag(R_GUARD, near_f32_midpoint(r) | (RANGE == R_BOTH ? !((aq > 1e-30) & (aq < 1e30)) : false));
	return F(r);
}
template <int RANGE>
DJB_DEV float sqrt_g(double a, Rare &rare)
{
	const double g = sqrt_fast(a);
	rare.flag(R_GUARD, near_f32_midpoint(g) | (RANGE == R_BOTH ? !((a > 1e-30) & (a < 1e30)) : false));
	return F(g);
}
template <int RANGE>
DJB_DEV v3 normalize_g(v3 v, Rare &rare) { return scale(inversesqrt_g<RANGE>(dot(v, v), rare), v); }

DJB_DEV float erf_given_exp_g(float x, double e, Rare &rare)               // erf_given_exp, djb_device.hpp
{
	const float a1 = 0.254829592f, a2 = -0.284496736f, a3 = 1.421413741f,
	            a4 = -1.453152027f, a5 = 1.061405429f, p = 0.3275911f;
	float sign = x < 0 ? -1.0f : 1.0f;
	x = fabsf(x);
	float t = recip_g<R_NONE>(1.0 + D(p * x), rare);                        // x = cot_k in (0, 2900]: the operand is in [1, 952]
	float poly = ((((a5 * t + a4) * t) + a3) * t + a2) * t + a1;
	float y = F(1.0 - D(poly * t) * e);
	return sign * y;
}
// Giles' erfinv, central arm (w < 5); the tail arm is flagged (dj_brdf.h:691-721).  IN_UNIT: the caller guarantees u in [-1, 1]
// and not NaN; then (1 - u)(1 + u) is a float in [0, 1 + 2^-23] and logf's own range flag is covered by `w < 5` (logf_main).
// Either way a sample that is not flagged here has w in [-2e-7, 5), so |p| <= sum |c_k| 2.5^k < 2.18 and |result| <= 2.18 |u|.
template <bool IN_UNIT>
DJB_DEV float erfinv_central(float u, const GlibcTabs &gt, Rare &rare, int site)
{
	float w = -logf_main<!IN_UNIT>((1.0f - u) * (1.0f + u), gt, rare), p;
	rare.flag(site, !(w < 5.0f));
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
	return p * u;
}

// beckmann_qf2_radial (djb_device_microfacet.inc, dj_brdf.h:1897-1952) for a sample that converges within TRIPS trips
// UNROLL: four copies of the trip (sample: 15.7 vs 16.3 ms per 1e9) or a loop (evalp_is, whose tail needs the registers:
// 107 VGPRs = 4 waves per SIMD instead of 160 = 3; 6.30 vs 6.80 ms per 2e8)
template <bool UNROLL>
DJB_DEV float bk_qf2_common(float u, float cos_k, float sin_k, const GlibcTabs &gt, Rare &rare)
{
	const float sqrt_pi_inv = F(1. / sqrt(DJB_PI));
	float cot_k = cos_k / sin_k, tan_k = sin_k / cos_k;
	const double e_cot = exp_main_neg(D(-cot_k * cot_k), gt.exp64, rare);
	float a = -1, c = erf_given_exp_g(cot_k, e_cot, rare);
	u = fmax_(u, 1e-6f);
	float fit = 1 + cos_k * (-0.876f + cos_k * (0.4265f - 0.0594f * cos_k));
	// u = max(u, 1e-6) is never NaN (djb::max returns its second argument then) and at most 0.99999; fit = 1 + c (-0.876 + c (0.4265
	// - 0.0594 c)) lies in [0.49, 1] for the cos_k in (0, 1) that are not flagged: powf's special cases cannot occur
	float b = c - (1 + c) * powf_main(1 - u, fit, gt);
	// c >= -2.4e-7 (erf's poly t is at most 1.0000002 and e_cot at most 1) and the second term lies in [0, 7.6e7]: an argument
	// tiny enough for tan_k > 2^27 (1 + 2^-22) has cot_k^2 < 2^-54, which exp_main_neg flags
	float normalization = recip_g<R_NONE>(D(1 + c) + D(sqrt_pi_inv * tan_k) * e_cot, rare);
	float inv_erf = 0.0f, b_at = 0.0f;
	bool done = false;
#ifdef DJB_BK_TRIP4_FULL           // A/B and measurement builds: the last trip evaluates its value like the others (the round-4 form)
	constexpr bool SHORT_LAST = false;
#else
	constexpr bool SHORT_LAST = true;
#endif
	// The LAST trip (round 5).  What a sample needs from trip 4 is erfinv(b3) -- and the knowledge that the reference leaves its loop
	// there, |value(b3)| < 1e-5.  That value is not computed any more (glibc's expf, the CDF, its derivative: a quarter of a trip):
	// b3 = b2 - q, q = value(b2) / derivative(b2), is a Newton step on a smooth function, so
	//     |value(b3)| <= 1/2 max|f''| q^2 + rounding,   f'' = -N tan_k sqrt(pi)/2 exp(erfinv(b)^2)   (f' = N (1 - erfinv(b) tan_k)),
	// and the sample is KNOWN to converge there when 0.4431 N tan_k q^2 / E(b2) < 4e-6 with |q| < 2e-3: exp(erfinv^2) = 1 / E moves by
	// at most e^(125 |q|) = 1.28 over the step inside the central arm (E >= 0.027, |erfinv| <= 1.9: both ends are flagged otherwise),
	// the float evaluation of value adds < 1e-6 at either end (terms of magnitude <= 1, a few ulps each), the error of the float
	// derivative enters as q * |d derivative| < 2e-7: 1.28 * 4e-6 + 2.2e-6 = 7.3e-6 < 1e-5.  A sample that needs trip 4 and cannot show
	// this, or whose b3 leaves [a, c] (the reference bisects), is flagged: the exact per-sample code decides.  Nothing is approximated --
	// a sample that is kept has taken the reference's decisions and carries the reference's erfinv(b3).  Measured (DJB_EXP_RARE_COUNT
	// build with DJB_EXP_TRIP4_CHECK): profiles/r05/beckmann_trip4.txt.
	float q_step = 0.0f, e_last = 1.0f;
	auto trip = [&](bool last) {
		const bool inside = (b >= a) & (b <= c);
		const float bt = !inside ? 0.5f * (a + c) : b;
		// bt lies in [a, c], a sub-interval of [-1, erf(cot_k)] with finite ends (c = erf_given_exp_g of a finite positive cot_k is in
		// [0, 1]; a NaN b takes the midpoint): IN_UNIT holds, and |ie| <= 2.18 keeps -ie^2 inside expf's main path
		const float ie = erfinv_central<true>(bt, gt, rare, R_TAIL_LOOP);
		inv_erf = ie; b_at = bt;
		if (last && SHORT_LAST) {
			const float nt = 0.4431f * (normalization * tan_k);
			const bool sure = inside & (tan_k > 0.0f) & (fabsf(q_step) < 2e-3f) & (nt * (q_step * q_step) < 4e-6f * e_last);   // tan_k > 0: the bound's derivation assumes it (a k.z <= 0 sample is R_DEGENERATE anyway -- this test does not rely on that)
#ifdef DJB_EXP_TRIP4_CHECK           // measurement: the decision above against the value it stands in for
			{
				const float value = normalization * (1 + bt + sqrt_pi_inv * tan_k * expf_main(-ie * ie, gt)) - u;
				const bool truly = fabsf(value) < 1e-5f;
				// (samples that are flagged anyway -- k.z <= 0 runs this code on a negative tan_k, the tail arm of erfinv -- are not counted)
				const bool counted = !done & !rare.any;
				const unsigned int n4 = (unsigned int)__popcll(__ballot(counted)), ns = (unsigned int)__popcll(__ballot(counted & sure)),
				                   nw = (unsigned int)__popcll(__ballot(counted & sure & !truly)), nl = (unsigned int)__popcll(__ballot(counted & !sure & truly));
				if (counted & sure & !truly) {
					if (atomicAdd(&g_trip4_printed, 1u) < 12u)
						printf("djb_exp trip4 miss: u %.9g cos_k %.9g sin_k %.9g tan_k %.9g N %.9g a %.9g c %.9g b3 %.9g q %.9g E2 %.9g ie3 %.9g value3 %.9g nt*q*q %.9g\n",
						       u, cos_k, sin_k, tan_k, normalization, a, c, bt, q_step, e_last, ie, value, nt * (q_step * q_step));
				}
				if ((threadIdx.x & 63u) == 0) {
					atomicAdd(&g_trip4[0], (unsigned long long)n4); atomicAdd(&g_trip4[1], (unsigned long long)ns);
					atomicAdd(&g_trip4[2], (unsigned long long)nw); atomicAdd(&g_trip4[3], (unsigned long long)nl);
				}
			}
#endif
			done |= sure;                                    // a lane frozen since an earlier trip stays done (and repeats its erfinv)
			return;
		}
		const float e_ie = expf_main(-ie * ie, gt);
		const float value = normalization * (1 + bt + sqrt_pi_inv * tan_k * e_ie) - u;
		const float derivative = normalization * (1 - ie * tan_k);
		done = fabsf(value) < 1e-5f;
		const bool pos = value > 0;
		c = pos ? bt : c; a = pos ? a : bt;
		// a converged lane is frozen instead of masked: with b = bt -- now an end of [a, c] -- the next trip takes the same bt and so
		// repeats this one exactly (same ie, value, flags; the updates of a, c and done are idempotent): the last trip's ie and
		// bt are the converged ones, and no per-trip select of the results is needed
		if (!last) {
			float qv = value / derivative;
			asm volatile("" : "+v"(qv));            // computed by all lanes: no branch around the division for the converged ones
			b = done ? bt : bt - qv;
			q_step = qv; e_last = e_ie;
		}
	};
	if (UNROLL) {
#pragma unroll
		for (int t = 0; t < TRIPS; ++t) trip(t == TRIPS - 1);
	} else {
		int trips = SHORT_LAST ? TRIPS - 1 : TRIPS;
		asm volatile("" : "+s"(trips));               // opaque bound: a loop, not copies of the body
		for (int t = 0; t < trips; ++t) trip(false);
		if (SHORT_LAST) trip(true);
	}
	// not converged: more trips; b < -0.9999: the reference re-evaluates erfinv at the clamped argument
	rare.flag(R_TRIPS, !done);
	rare.flag(R_CLAMP, !(b_at >= -0.9999f));
	return inv_erf;
}

// microfacet::sample for Beckmann (mf_sample, mf_sample_vp22_std; dj_brdf.h:1669-1700, 1806-1832), common path
template <bool UNROLL>
DJB_DEV v3 bk_sample_common(const Params &p, float u1, float u2, v3 o, const GlibcTabs &gt, Rare &rare)
{
	u1 = sat_(u1) * 0.99998f + 0.00001f;
	u2 = sat_(u2) * 0.99998f + 0.00001f;
	float a = o.x * p.ax + o.y * p.ay * p.rho;
	float bb = o.y * p.ay * p.s;
	float c = o.z - o.x * p.tx - o.y * p.ty;
	v3 k = normalize_g<R_BOTH>(mk(a, bb, c), rare);
	// k.z <= 0 returns (0, 0, 1) in the reference, k.z >= 1 skips the rotation: both to the full code
	rare.flag(R_DEGENERATE, !(D(k.z) > 0.0) | !(D(k.z) < 1.0));
	float cos_k = k.z;
	// a float k.z in (0, 1) is at most 1 - 2^-24, its float square at most 1 - 2^-23: the operand lies in [2^-23, 1], sin_k in
	// [3.4e-4, 1] (the reference's `sin_k > 0` test cannot fail), cot_k in (0, 2900] and tan_k in [3.4e-4, Inf)
	float sin_k = sqrt_g<R_NONE>(1.0 - D(k.z * k.z), rare);
	float tx = bk_qf2_common<UNROLL>(u1, cos_k, sin_k, gt, rare);
	float ty = erfinv_central<false>(F(2.0 * D(u2) - 1.0), gt, rare, R_TAIL_QF1);                     // beckmann_qf1
	float nrm = inversesqrt_g<R_BOTH>(k.x * k.x + k.y * k.y, rare);
	float cp = k.x * nrm, sp = k.y * nrm;
	float txm = cp * tx - sp * ty;
	float tym = sp * tx + cp * ty;
	float txh = p.ax * txm + p.tx;
	float chol = p.rho * txm + p.s * tym;
	float tyh = p.ay * chol + p.ty;
	v3 h = normalize_g<R_UPPER>(mk(-txh, -tyh, 1), rare);                              // (txh^2 + tyh^2) + 1 >= 1, or Inf / NaN
	return sub(scale(F(2.0 * D(dot(o, h))), h), o);
}

// ---- DJB_OPT_CONTRACT_1E5: the sampled DIRECTION inside a value contract -- every component of the returned unit vector within
// 1e-5 of the reference's -- instead of bit-identically.  Same structure as above: a straight-line common path for every lane,
// the full per-sample code (sample_one: the reference's arithmetic) for the flagged ones; what changes is the arithmetic of
// the common path: v_rcp / v_rsq / v_log / v_exp in fp32 where the exact path runs glibc's logf / expf / powf / exp restated
// in fp64 and correctly rounded divisions (~300 instead of ~690 VALU per sample).  What keeps a sample inside the contract:
//  * the stretched view direction k and sin(theta_k) are the REFERENCE's floats (guarded exact normalize / sqrt, as in
//    the common path above): near normal incidence 1 - k.z^2 amplifies an ulp of k.z by 1 / sin^2;
//  * every decision the reference takes on a computed value is either taken on bit-identical operands (k.z > 0, k.z < 1) or
//    guarded by a band: Newton's exit |value| < 1e-5 (2 epsv(u) around the threshold, below), the bisection safeguard
//    b in [a, c] (max(CTS_B_BAND, 2 epsv / |f'|) at both ends), erfinv's
//    arm w < 5 and the final clamp b >= -0.9999.  Inside a band the sample goes to the exact path.  With the same
//    decisions the two Newton sequences stay within epsv / |derivative| of each other (the iteration contracts);
//  * the error that reaches the direction is bounded per sample -- d slope / d b = sqrt(pi)/2 exp(slope^2), the lobe's
//    stretch, |d h / d slope| <= h.z, |d i / d h| <= 4 -- and a sample whose bound exceeds CTS_DIR_MAX goes to the exact
//    path as well.  djb_selftest_contract_sample measures the actual maximum (tests/test_gpu_contract.py).
// epsv(u) = CTS_EPSV_U u + CTS_EPSV_0 models |value_contract - value_reference| at equal b: value = N S - u with N S ~ u near the
// exit, N carrying ~3e-7 relative (one v_rcp, erf's reciprocal and polynomial, the split-argument exponential) and S ~2e-7 (v_exp;
// erfinv's 1.5e-7 enters through 2 ie^2, which is large only where exp(-ie^2) has made the term small).  It is a first-order
// estimate for the typical case, NOT a proven worst case (a grazing view with tan_k ~ 1 can reach ~1e-6 u in N alone); what it is
// held to is measurement: djb_selftest_contract_sample reports the largest (observed difference - 1.5e-6) / per-sample bound --
// 0.19 over 1e10 samples of seven lobes x five input families (profiles/r04/contract_sample.txt), i.e. the bound has a factor
// of five in hand, and the largest observed difference, 3.0e-6, a factor of three to the contract itself.
// The two sequences sit epsv / |f'| apart (the iteration contracts), which the value sees as another epsv: the exit test is
// decided only outside 2 epsv(u) of its threshold, the safeguard's only outside max(CTS_B_BAND, 2 epsv / |f'|) of an end
// (CTS_B_BAND covers the first trip: b0's own error, 4e-7).  CTS_DIR_MAX leaves 2e-6 of the 1e-5 to everything that is not the
// Newton sequence (the rotation, the two rsq normalisations, the reflection: ~5e-7 measured).
#ifndef DJB_EXP_CTS_SCALE            // sensitivity builds only (tools/exp/r05/beckmann_share_sensitivity.sh): the error model's constants scaled
#define DJB_EXP_CTS_SCALE 1.0f
#endif
constexpr float CTS_B_BAND = 4.0e-6f, CTS_EPSV_U = 5.0e-7f * DJB_EXP_CTS_SCALE, CTS_EPSV_0 = 1.5e-7f * DJB_EXP_CTS_SCALE, CTS_DIR_MAX = 8.0e-6f;

DJB_DEV float cts_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
// exp(y), y <= 0, ~2 ulp: y log2(e) split into hi + lo (a plain exp2(y * log2e) loses |y| 2^-24)
DJB_DEV float cts_exp_neg(float y)
{
	const float L = 1.44269502f, L_LO = 1.92596299e-8f;
	const float hi = y * L;
	const float lo = __builtin_fmaf(y, L, -hi) + y * L_LO;
	const float e = __builtin_amdgcn_exp2f(hi);
	return __builtin_fmaf(e, lo * 0.693147182f, e);
}
// x^y for x in [1e-6, 1], y in [0.4, 1.1]: x = m 2^e, y log2(x) split into an integer and a fraction whose absolute error
// stays ~1.5e-7 (the product y e is carried with its fma residual) -> relative error < 3e-7
DJB_DEV float cts_pow(float x, float y)
{
#pragma clang fp contract(fast)
	const float m = __builtin_amdgcn_frexp_mantf(x), fe = (float)__builtin_amdgcn_frexp_expf(x);
	const float lm = __builtin_amdgcn_logf(m);                 // log2 m in [-1, 0]
	const float t1 = y * fe, r1 = __builtin_fmaf(y, fe, -t1);
	const float n = rintf(t1);
	const float f = ((t1 - n) + y * lm) + r1;
	return __builtin_amdgcn_ldexpf(__builtin_amdgcn_exp2f(f), (int)n);
}
// Giles' erfinv, central arm, with v_log_f32; tail = the reference is in (or within 1e-2 of) the other arm
DJB_DEV float cts_erfinv(float u, bool &tail)
{
#pragma clang fp contract(fast)
	float w = -0.693147182f * __builtin_amdgcn_logf((1.0f - u) * (1.0f + u)), p;
	tail = !(w < 4.99f);
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
	return p * u;
}

// is the lobe inside the domain the bounds above were made for?  (anything else keeps the exact kernel)
inline bool cts_params_ok(const Params &p)
{
	return p.ax >= 1e-3f && p.ax <= 1e2f && p.ay >= 1e-3f && p.ay <= 1e2f && fabsf(p.rho) <= 0.99f &&
	       fabsf(p.tx) <= 10.0f && fabsf(p.ty) <= 10.0f;
}

DJB_DEV v3 bk_sample_contract(const Params &p, float u1, float u2, v3 o, Rare &rare, float *bound_out = nullptr)
{
	u1 = sat_(u1) * 0.99998f + 0.00001f;
	u2 = sat_(u2) * 0.99998f + 0.00001f;
	const float sa = o.x * p.ax + o.y * p.ay * p.rho;
	const float sb = o.y * p.ay * p.s;
	const float sc = o.z - o.x * p.tx - o.y * p.ty;
	const v3 k = normalize_g<R_BOTH>(mk(sa, sb, sc), rare);                    // the reference's k
	// k.z <= 0 / k.z >= 1: the reference's special cases; below 1e-3 (tan_k > 1e3) the bounds of the header were not made
	rare.flag(R_DEGENERATE, !(k.z > 1e-3f) | !(D(k.z) < 1.0));
	const float cos_k = k.z;
	const float sin_k = sqrt_g<R_NONE>(1.0 - D(k.z * k.z), rare);              // the reference's sin_k, in [3.4e-4, 1]
	const float fit = 1 + cos_k * (-0.876f + cos_k * (0.4265f - 0.0594f * cos_k));
	const float u = fmax_(u1, 1e-6f);
	const float epsv = CTS_EPSV_U * u + CTS_EPSV_0;
	float ie = 0.0f, b_at = 0.0f, E = 1.0f, rder = 0.0f, q_step = 0.0f, tx, ty, hz, oh, ol2;
	bool doubt = false;                     // a decision of the Newton loop fell inside its band (or met a NaN: the tests are "clearly outside")
	bool done = false, tails = false;
	v3 h;
	{
#pragma clang fp contract(fast)          // approximate arithmetic from here on: fused multiply-adds only remove roundings
		const float cot_k = cos_k * cts_rcp(sin_k), tan_k = sin_k * cts_rcp(cos_k);
		const float e_cot = cts_exp_neg(-fminf(cot_k * cot_k, 100.0f));
		// erf(cot_k), Abramowitz & Stegun 7.1.26 as the reference evaluates it (dj_brdf.h:667-688)
		const float t = cts_rcp(1.0f + 0.3275911f * cot_k);
		const float poly = ((((1.061405429f * t - 1.453152027f) * t) + 1.421413741f) * t - 0.284496736f) * t + 0.254829592f;
		float c = 1.0f - (poly * t) * e_cot, a = -1.0f;
		float b = c - (1 + c) * cts_pow(1 - u, fit);
		const float K = 0.564189584f * tan_k;                                  // tan_k / sqrt(pi)
		const float N = cts_rcp((1 + c) + K * e_cot);                         // the normalisation of the CDF
		const float vband = 2.0f * epsv;
#pragma unroll
		for (int trip = 0; trip < TRIPS; ++trip) {
			// the safeguard's decision (b in [a, c]) is certain only away from both ends; the exit's (|value| < 1e-5) only away
			// from the threshold.  A converged lane sits ON an end and repeats its last trip: it takes no decisions any more
			const bool clear_ends = (fabsf(b - a) > fmaxf(CTS_B_BAND, vband * fabsf(rder))) & (fabsf(b - c) > fmaxf(CTS_B_BAND, vband * fabsf(rder)));
			const bool inside = (b >= a) & (b <= c);
			const float bt = inside ? b : 0.5f * (a + c);
			bool tail;
			ie = cts_erfinv(bt, tail);
			tails |= tail;
#ifndef DJB_BK_TRIP4_FULL
			if (trip == TRIPS - 1) {
				// the last trip (round 5, as in bk_qf2_common): only erfinv(b3) is needed; that the reference leaves its loop there is shown
				// from the Newton step that led here instead of being computed -- 1/2 |f''| q^2 with room for both sequences' rounding --
				// and a sample that cannot show it is in doubt (exact path).  E and rder stay those of b2: the error estimate below reads
				// them within the factor e^(125 |q|) <= 1.28 the step can move them by (it has a factor of five in hand)
				const bool sure = inside & clear_ends & (tan_k > 0.0f) & (fabsf(q_step) < 2e-3f) & ((0.4431f * (N * tan_k)) * (q_step * q_step) < 3e-6f * E);
				doubt |= !done & !sure;
				done |= sure;
				b_at = bt;
				break;
			}
#endif
			E = cts_exp_neg(-ie * ie);
			const float value = N * ((1 + bt) + K * E) - u;
			const float derivative = N * (1 - ie * tan_k);
			rder = cts_rcp(derivative);
			const float av = fabsf(value);
			doubt |= !done & !(clear_ends & (fabsf(av - 1e-5f) > vband));      // `done` is still the previous trip's: a frozen lane decides nothing
			done = av < 1e-5f;                                                 // a frozen lane: the same value again
			b_at = bt;
			const bool pos = value > 0;
			c = pos ? bt : c; a = pos ? a : bt;
			q_step = value * rder;
			b = done ? bt : bt - q_step;
		}
		tx = ie;
		bool tail2;
		ty = cts_erfinv(2.0f * u2 - 1.0f, tail2);                             // beckmann_qf1: float(2.0 * u2 - 1.0) is one rounding (fused here)
		tails |= tail2;
		const float nrm = __builtin_amdgcn_rsqf(k.x * k.x + k.y * k.y);
		const float cp = k.x * nrm, sp = k.y * nrm;
		const float txm = cp * tx - sp * ty, tym = sp * tx + cp * ty;
		const float txh = p.ax * txm + p.tx;
		const float chol = p.rho * txm + p.s * tym;
		const float tyh = p.ay * chol + p.ty;
		hz = __builtin_amdgcn_rsqf((txh * txh + tyh * tyh) + 1.0f);
		h = mk(-txh * hz, -tyh * hz, hz);
		oh = dot(o, h);
		ol2 = dot(o, o);
	}
	rare.flag(R_TRIPS, !done | doubt);
	rare.flag(R_TAIL_LOOP, tails);
	rare.flag(R_CLAMP, !(b_at > -0.99989f));
	// the error that can reach the direction (header): the Newton sequence's distance to the reference's, through erfinv's slope,
	// the lobe's stretch (the norm of [[ax, 0], [ay rho, ay s]] is at most sqrt(ax^2 + ay^2)), |d h / d slope| <= h.z, and the
	// reflection i = 2 (o.h) h - o: |d i| <= (2 |o| + 2 |o.h|) |d h|.  The contract is relative to |o| (1 for a direction).
	const float err_tx = (epsv * 0.886226925f) * fabsf(rder) * cts_rcp(E) + 3e-7f * (fabsf(tx) + fabsf(ty));
	const float stretch = sqrtf(p.ax * p.ax + p.ay * p.ay);                    // launch-uniform
	const float ol = ol2 * __builtin_amdgcn_rsqf(fmaxf(ol2, 1e-30f));
	const float bound = ((2.0f * ol + 2.0f * fabsf(oh)) * hz) * (stretch * err_tx);
	rare.flag(R_GUARD, !(bound < CTS_DIR_MAX * fmaxf(ol, 1.0f)));
	if (bound_out) *bound_out = bound;
	return sub(scale(2.0f * oh, h), o);
}


// ---- the same for GGX (round 4).  ggx::qf2_radial / qf3_radial are closed forms (dj_brdf.h:2089-2146): there is no Newton sequence to
// follow, and every decision the reference takes on a computed value is harmless -- the four tangent / cotangent addition forms are
// one function, so picking another form than the reference near 0.707107 moves the result by rounding only; u2 < 0.5 and k.z in
// (0, 1) are decided on operands bit-identical to the reference's.  What is kept exact, cheaply: the stretched view direction k (guarded exact normalize as above: 1 - k.z^2 amplifies an ulp of k.z by 1 / sin^2 near normal incidence),
// sin_t = float(u (1 + cos_k) - 1) (three fp64 operations: its square is cancelled against 1) and the two cubic / quartic
// polynomials of qf3 (fp64 Horner as in the reference: their quotient is a difference of nearly equal terms towards u2 -> 0 and 1).
// What is approximated: five divisions and three square roots (v_rcp / v_rsq / v_sqrt_f32, ~1 ulp each) and the final products.
// Slopes come out with a relative error of a few ulp (an absolute one where a + b cancels, |a|, |b| <= 1); through
// |d h| <= |d slope| h.z and the reflection that is ~1e-6 of the 1e-5 contract; the per-sample bound below is the same first-order
// estimate as Beckmann's (the lobe's stretch, h.z, 2 |o| + 2 |o.h|), checked by the same self-test and directed search.
// Error model (first order, this path's roundings plus the reference's own, which are of the same kind and half the size): a, b carry
// 1.5 ulp each (v_rcp + a product), their product 4; whichever of the four forms is taken, numerator and denominator are then off by at
// most CTG_OPS (|a| + |b| + |a b| + 1) in absolute terms, and tx = num / den by that over |den| for the numerator plus |tx| times that
// over |den| for the denominator: tx = (1 + rho) tx + e_abs with rho = CTG_REL + CTG_OPS asum, e_abs = CTG_OPS asum, asum = (|a| + |b| + |a b|
// + 1) / |den|.  rho is the term that matters: where a b -> 1 (the sampled normal at the horizon of the stretched frame) it grows like
// |tx|, the lobe's stretch turns the slope back into a moderate one, and a sharp lobe (alpha = 0.02) seen at grazing incidence leaves
// the contract by eps / alpha ~ 1.5 if it is ignored (measured: 1.27e-5 before the term was in the bound).
#ifndef DJB_CTG_OPS
#define DJB_CTG_OPS 6e-8f
#define DJB_CTG_REL 1e-7f
#endif
constexpr float CTG_OPS = DJB_CTG_OPS, CTG_REL = DJB_CTG_REL;
DJB_DEV v3 ggx_sample_contract(const Params &p, float u1, float u2, v3 o, Rare &rare, float *bound_out = nullptr)
{
	u1 = sat_(u1) * 0.99998f + 0.00001f;
	u2 = sat_(u2) * 0.99998f + 0.00001f;
	const float sa = o.x * p.ax + o.y * p.ay * p.rho;
	const float sb = o.y * p.ay * p.s;
	const float sc = o.z - o.x * p.tx - o.y * p.ty;
	const v3 k = normalize_g<R_BOTH>(mk(sa, sb, sc), rare);                    // the reference's k
	rare.flag(R_DEGENERATE, !(k.z > 1e-3f));                                   // k.z <= 0: the reference returns (0, 0, 1); NaN; a view too grazing for the bounds
	// k.z == 1 (a sharp lobe seen near the normal: the stretched view direction rounds to the pole) is the reference's other special case
	// -- sin_k = 0 and no rotation of the sampled slope (dj_brdf.h:1806-1832) -- and common enough to be kept here
	const bool pole = !(D(k.z) < 1.0);
	const float cos_k = k.z;
	// sin_k = float(sqrt(1.0 - double(k.z * k.z))) in the reference: the float product is formed as there, 1 - it loses at most half
	// an ulp of the difference in float (nothing above 0.5), v_sqrt_f32 another ulp: 1.5 ulp of sin_k, not amplified (what IS amplified,
	// 1 / sin^2 times an ulp of k.z, sits in k.z itself, which is the reference's)
	const float sin_k = pole ? 0.0f : __builtin_amdgcn_sqrtf(1.0f - k.z * k.z);
	const float sin_t = F(D(u1) * (1.0 + D(cos_k)) - 1.0);                      // the reference's (ggx_qf2_radial)
	const float s2 = sin_t * sin_t;                                            // as the reference forms it: 1 - s2 below loses nothing more
	// qf3's remap and polynomials, the reference's own values (ggx_qf3_radial)
	const bool lower = D(u2) < 0.5;
	const float ur = lower ? F(2.0 * (0.5 - D(u2))) : F(2.0 * (D(u2) - 0.5));
	const double x = D(ur);
	const float pn = F(x * (x * (x * (-0.365728915865723) + 0.790235037209296) - 0.424965825137544) + 0.000152998850436920);
	const float qd = F(x * (x * (x * (x * 0.169507819808272 - 0.397203533833404) - 0.232500544458471) + 1) - 0.539825872510702);
	float tx, ty, hz, oh, ol2, asum;
	v3 h;
	{
#pragma clang fp contract(fast)
		const float cos_t = __builtin_amdgcn_sqrtf(fmaxf(1.0f - s2, 0.0f));
		const bool tan_t = cos_t > 0.707107f, tan_k = sin_k < 0.707107f;
		const float a = (tan_t ? sin_t : cos_t) * cts_rcp(tan_t ? cos_t : sin_t);       // tan_t or cot_t
		const float b = (tan_k ? sin_k : cos_k) * cts_rcp(tan_k ? cos_k : sin_k);       // tan_k or cot_k
		const float prod = a * b;
		float num, den;
		if (tan_t == tan_k) { const float s = a + b; num = tan_t ? -s : s; den = 1.0f - prod; }
		else { num = 1.0f + prod; den = tan_t ? a - b : b - a; }
		const float rden = cts_rcp(den);
		tx = num * rden;
		asum = (fabsf(a) + fabsf(b) + fabsf(prod) + 1.0f) * fabsf(rden);                // an absolute error of num / den, over |den|
		const float alpha = __builtin_amdgcn_sqrtf(1.0f + tx * tx);
		ty = (lower ? -alpha : alpha) * (pn * cts_rcp(qd));
		const float nrm = __builtin_amdgcn_rsqf(k.x * k.x + k.y * k.y);
		const float cp = pole ? 1.0f : k.x * nrm, sp = pole ? 0.0f : k.y * nrm;
		const float txm = cp * tx - sp * ty, tym = sp * tx + cp * ty;
		const float txh = p.ax * txm + p.tx;
		const float chol = p.rho * txm + p.s * tym;
		const float tyh = p.ay * chol + p.ty;
		hz = __builtin_amdgcn_rsqf((txh * txh + tyh * tyh) + 1.0f);
		h = mk(-txh * hz, -tyh * hz, hz);
		oh = dot(o, h);
		ol2 = dot(o, o);
	}
	// den = 0 (the sampled normal at the horizon of the stretched frame) or a vanishing quotient polynomial: not a number to bound
	rare.flag(R_CLAMP, !(fabsf(tx) < 1e18f) | !(fabsf(ty) < 1e18f));
	// tx is off by rho tx + e_abs (header), and ty = +-sqrt(1 + tx^2) p / q follows tx's relative error as far as sqrt(1 + tx^2) follows
	// tx: where the error is large (a b -> 1, |tx| -> 1e5) the slope vector is scaled as a whole.  A scaling survives the lobe's linear
	// stretch as a scaling (about the mean slope (p.tx, p.ty)) and moves h by |slope| h.z^2 = sin(theta_h) h.z per unit, not by
	// |slope| h.z; what is not a scaling -- e_abs, the part of ty that does not follow, the roundings of the rest -- goes through the stretch
	// and h.z as before.
	const float rho = CTG_REL + CTG_OPS * asum, e_abs = CTG_OPS * asum;
	const float rest = e_abs + rho * fabsf(ty) * cts_rcp(1.0f + tx * tx) + CTG_REL * (fabsf(tx) + fabsf(ty));
	const float stretch = sqrtf(p.ax * p.ax + p.ay * p.ay);                    // launch-uniform
	const float sin_h = __builtin_amdgcn_sqrtf(fmaxf(1.0f - hz * hz, 0.0f));
	const float dh = hz * (rho * (sin_h + fabsf(p.tx) + fabsf(p.ty)) + stretch * rest);
	const float ol = ol2 * __builtin_amdgcn_rsqf(fmaxf(ol2, 1e-30f));
	const float bound = (2.0f * ol + 2.0f * fabsf(oh)) * dh;
	rare.flag(R_TRIPS, !(bound < CTS_DIR_MAX * fmaxf(ol, 1.0f)));             // (the Newton-trips site: GGX has no other use for it)
	if (bound_out) *bound_out = bound;
	return sub(scale(2.0f * oh, h), o);
}
// one name for both lobes' contract paths
template <int KIND>
DJB_DEV v3 sample_contract(const Params &p, float u1, float u2, v3 o, Rare &rare, float *bound_out = nullptr)
{
	if (KIND == KIND_GGX) return ggx_sample_contract(p, u1, u2, o, rare, bound_out);
	return bk_sample_contract(p, u1, u2, o, rare, bound_out);
}

#include "djb_contract_device.inc"   // ct_is_tail: the evalp_is tail under DJB_OPT_CONTRACT_1E5

// CT (DJB_OPT_CONTRACT_1E5): sample -> bk_sample_contract (directions within 1e-5); evalp_is -> the EXACT direction of the common
// path with ct_is_tail for weight and pdf.  Either way a declined sample takes the exact per-sample code through the queue.
template <bool IS, bool RNG, int FRK, bool DENSE, bool CT = false, int KIND = KIND_BECKMANN>
__global__ __launch_bounds__(BLOCK) void k_sample_bk(Brdf b, Params p, long long n, const float *u1a,
                                                     const float *u2a, uint32_t seed1, uint32_t seed2,
                                                     unsigned long long start, View vo, View vi_out,
                                                     View vw_out, float *out_pdf, djbk::CtParams ct)
{
	static_assert(KIND == KIND_BECKMANN || (KIND == KIND_GGX && CT && !IS), "GGX: the contract-mode sampler only (its exact sampler is k_sample)");
	__shared__ double s_glibc[KIND == KIND_BECKMANN ? GLIBC_LDS_WORDS : 1];
	__shared__ unsigned long long s_exp[KIND == KIND_BECKMANN ? 256 : 1];
	__shared__ unsigned int s_q[WAVES][7][QCAP];       // deferred samples: {k lo, k hi, u1, u2, o.xyz}
	GlibcTabs gt = glibc_tabs_global();                // GGX's sampler calls no libm function
	if (KIND == KIND_BECKMANN) {
		gt = glibc_tabs_to_lds(s_glibc, threadIdx.x, BLOCK);
		gt.exp64 = b.exp_lds = glibc_exp_tab_to_lds(s_exp, threadIdx.x, BLOCK);
		__syncthreads();
	}
	const unsigned int t = threadIdx.x, wave = t >> 6, lane = t & 63u;
	unsigned int (&q)[7][QCAP] = s_q[wave];
	unsigned int qn = 0;                               // wave-uniform
	// the full per-sample code on `cnt` queued samples starting at slot `first` (one per lane)
	auto drain = [&](unsigned int first, unsigned int cnt) {
		if (lane < cnt) {
			const unsigned int j = first + lane;
			const long long k = (long long)(((unsigned long long)q[1][j] << 32) | q[0][j]);
			const float u1 = __uint_as_float(q[2][j]), u2 = __uint_as_float(q[3][j]);
			const v3 o = mk(__uint_as_float(q[4][j]), __uint_as_float(q[5][j]), __uint_as_float(q[6][j]));
			v3 i_out, w; float pdf;
			sample_one<KIND, IS, FRK>(b, p, u1, u2, o, gt, i_out, w, pdf);
			store3(vi_out, k, i_out);
			if (IS) { store3(vw_out, k, w); out_pdf[k] = pdf; }
		}
	};
	const long long stride = (long long)gridDim.x * BLOCK;
	for (long long k0 = (long long)blockIdx.x * BLOCK; k0 < n; k0 += stride) {     // k0: workgroup-uniform
		const long long k = k0 + t;
		// the tile's live lanes from a scalar bound; dense accesses as SGPR base + 32-bit lane offset (djb_device_units.inc: lane_byte_offset)
		const unsigned int rem = n - k0 >= (long long)BLOCK ? (unsigned int)BLOCK : (unsigned int)(n - k0);
		const bool live = t < rem;
		float u1 = 0.5f, u2 = 0.5f; v3 o = mk(0, 0, 1);
		if (live) {
			const unsigned int toff = lane_byte_offset(t);
			u1 = RNG ? gen_uniform(seed1, start + (unsigned long long)k) : (DENSE ? (*dense_off(u1a + k0, toff)) : u1a[k]);
			u2 = RNG ? gen_uniform(seed2, start + (unsigned long long)k) : (DENSE ? (*dense_off(u2a + k0, toff)) : u2a[k]);
			o = DENSE ? load3_dense_off(vo, k0, toff) : load3(vo, k);
		}
		Rare why;
		v3 i_;
		if constexpr (CT && !IS) i_ = sample_contract<KIND>(p, u1, u2, o, why);
		else i_ = bk_sample_common<!IS>(p, u1, u2, o, gt, why);
		v3 i_out = i_, w = mk(0, 0, 0); float pdf = 0.0f;
		if (IS && CT) {
			bool alive;
			why.flag(R_GUARD, !ct_is_tail<KIND_BECKMANN, FRK == -1 ? FR_IDEAL : FRK>(ct, i_, o, w, pdf, alive));
			i_out = alive ? i_ : mk(0, 0, 0);
		} else if (IS) { i_out = mk(0, 0, 0); w = mf_evalp_is_tail<KIND_BECKMANN, FRK>(b, p, i_, o, i_out, pdf); }
#if defined(DJB_EXP_NO_DEFER) && DJB_EXP_NO_DEFER == 2   // timing-only builds (tools/exp/r05/beckmann_defer_cost.sh): the flags are computed, nobody is deferred
		const bool rare = why.any & live & (n < 0);          //   (n < 0 never holds, the compiler cannot know)
#elif defined(DJB_EXP_NO_DEFER)      // ... nobody is deferred AND the flag computations are dead code: the bare arithmetic of the common path
		const bool rare = false;
#else
		const bool rare = why.any & live;
#endif
		if (live && !rare) {
			const unsigned int soff = lane_byte_offset(t);
			if (DENSE) store3_dense_off(vi_out, k0, soff, i_out); else store3(vi_out, k, i_out);
			if (IS) {
				if (DENSE) { store3_dense_off(vw_out, k0, soff, w); (*dense_off(out_pdf + k0, soff)) = pdf; }
				else { store3(vw_out, k, w); out_pdf[k] = pdf; }
			}
		}
#ifdef DJB_EXP_RARE_COUNT
		{
			const unsigned int c0 = (unsigned int)__popcll(__ballot(live & !rare)), c1 = (unsigned int)__popcll(__ballot(rare));
			if (lane == 0) { atomicAdd(&g_rare[0], (unsigned long long)c0); if (c1) atomicAdd(&g_rare[1], (unsigned long long)c1); }
			for (int site = 0; site < R_SITES; ++site) {
				const unsigned int c = (unsigned int)__popcll(__ballot(rare && ((why.bits >> site) & 1u)));
				if (c && lane == 0) atomicAdd(&g_rare[2 + site], (unsigned long long)c);
			}
		}
#endif
		const unsigned long long mask = __ballot(rare);
		if (mask) {
			if (rare) {
				const unsigned int j = qn + (unsigned int)__popcll(mask & ((1ull << lane) - 1ull));
				q[0][j] = (unsigned int)(unsigned long long)k; q[1][j] = (unsigned int)((unsigned long long)k >> 32);
				q[2][j] = __float_as_uint(u1); q[3][j] = __float_as_uint(u2);
				q[4][j] = __float_as_uint(o.x); q[5][j] = __float_as_uint(o.y); q[6][j] = __float_as_uint(o.z);
			}
			qn += (unsigned int)__popcll(mask);
			if (qn >= 64u) {
				__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
				__builtin_amdgcn_wave_barrier();
				qn -= 64u;
				drain(qn, 64u);
				__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
				__builtin_amdgcn_wave_barrier();
			}
		}
	}
	if (qn) {
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
		__builtin_amdgcn_wave_barrier();
		drain(0u, qn);
	}
}

// ---- measurement: the contract path against the full per-sample code on generated (u1, u2, o).  max_bits[0]: largest
// |component difference| among the samples the contract path kept (float bits, atomicMax), max_bits[1]: largest
// difference / per-sample bound among them (how much of the bound is used; must stay below 1); counters: [0] samples, [1] samples
// handed to the exact path, [2] kept samples with a component outside 1e-5, [3] kept samples where the reference returns its
// degenerate (0, 0, 1)
template <int KIND>
__global__ __launch_bounds__(BLOCK) void k_sample_ct_selftest(Brdf b, Params p, long long n, uint32_t seed, unsigned long long start,
                                                              int family, unsigned int *max_bits, unsigned long long *counters)
{
	__shared__ double s_glibc[GLIBC_LDS_WORDS];
	__shared__ unsigned long long s_exp[256];
	GlibcTabs gt = glibc_tabs_to_lds(s_glibc, threadIdx.x, BLOCK);
	gt.exp64 = b.exp_lds = glibc_exp_tab_to_lds(s_exp, threadIdx.x, BLOCK);
	__syncthreads();
	const long long stride = (long long)gridDim.x * BLOCK;
	float worst = 0.0f, used = 0.0f;
	unsigned long long n_all = 0, n_def = 0, n_out = 0, n_deg = 0;
	for (long long k = (long long)blockIdx.x * BLOCK + threadIdx.x; k < n; k += stride) {
		const unsigned long long kk = start + (unsigned long long)k;
		float u1 = gen_uniform(seed ^ 0x1111u, kk), u2 = gen_uniform(seed ^ 0x2222u, kk);
		v3 o = gen_direction(seed, kk);
		if (family == 1) { o.z = 0.02f + 0.05f * o.z; o = normalize(o); }                 // grazing view
		else if (family == 2) o = normalize(mk(0.01f * o.x, 0.01f * o.y, 1.0f));          // near-normal view
		else if (family == 3) { u1 = u1 < 0.5f ? 1e-4f * u1 : 1.0f - 1e-4f * (1.0f - u1); u2 = u2 < 0.5f ? 1e-3f * u2 : 1.0f - 1e-3f * (1.0f - u2); }   // the tails of both uniforms
		else if (family == 4) o = scale(0.25f + 3.0f * gen_uniform(seed ^ 0x77u, kk), o); // un-normalised view
		++n_all;
		Rare why;
		float bound;
		const v3 ia = sample_contract<KIND>(p, u1, u2, o, why, &bound);
		if (why.any) { ++n_def; continue; }
		v3 ie, w; float pdf;
		sample_one<KIND, false, -1>(b, p, u1, u2, o, gt, ie, w, pdf);
		const float scale_o = fmaxf(1.0f, sqrtf(dot(o, o)));                  // the contract is relative to |o| (1 for a direction)
		const float d = fmaxf(fabsf(ia.x - ie.x), fmaxf(fabsf(ia.y - ie.y), fabsf(ia.z - ie.z))) / scale_o;
		if (!(d <= 1e-5f)) ++n_out;
		if (ie.x == 0.0f && ie.y == 0.0f && ie.z == 1.0f) ++n_deg;
		worst = fmaxf(worst, d == d ? d : 3.0e38f);
		used = fmaxf(used, (d - 1.5e-6f) * scale_o / bound);     // 1.5e-6: the allowance for everything that is not the Newton sequence (CTS_DIR_MAX)
	}
	atomicMax(&max_bits[0], __float_as_uint(worst));
	atomicMax(&max_bits[1], __float_as_uint(used));
	atomicAdd(&counters[0], n_all); atomicAdd(&counters[1], n_def); atomicAdd(&counters[2], n_out); atomicAdd(&counters[3], n_deg);
}


// ---- directed search instead of sampling: every lane hill-climbs ONE sample over the bit patterns of its five inputs
// (u1, u2, o.xyz), maximising the component difference between the contract path and the full per-sample code (in units of
// the contract, 1e-5 max(1, |o|)); a sample the contract path hands to the exact path scores 0.  A move adds +-2^e units in the
// last place (e = 0..20, hash-drawn) to one input and is kept if the score grows.  counters: [0] evaluations, [1] evaluated
// samples the fast path KEPT with a component outside the contract (must stay 0), [2] accepted moves.  best[k]: the score reached.
template <int KIND>
DJB_DEV float cts_attack_score(const Brdf &b, const Params &p, float u1, float u2, v3 o, const GlibcTabs &gt, unsigned long long &n_out)
{
	Rare why;
	const v3 ia = sample_contract<KIND>(p, u1, u2, o, why);
	if (why.any) return 0.0f;
	v3 ie, w; float pdf;
	sample_one<KIND, false, -1>(b, p, u1, u2, o, gt, ie, w, pdf);
	const float scale_o = fmaxf(1.0f, sqrtf(dot(o, o)));
	const float d = fmaxf(fabsf(ia.x - ie.x), fmaxf(fabsf(ia.y - ie.y), fabsf(ia.z - ie.z))) / (1e-5f * scale_o);
	if (!(d <= 1.0f)) ++n_out;                                       // NaN on one side only counts too
	return d == d ? d : 3.0e38f;
}
template <int KIND>
__global__ __launch_bounds__(BLOCK) void k_sample_ct_attack(Brdf b, Params p, long long n, float *u1a, float *u2a, View vo, int iters, uint32_t seed,
                                                            float *best, unsigned long long *counters)
{
	__shared__ double s_glibc[GLIBC_LDS_WORDS];
	__shared__ unsigned long long s_exp[256];
	GlibcTabs gt = glibc_tabs_to_lds(s_glibc, threadIdx.x, BLOCK);
	gt.exp64 = b.exp_lds = glibc_exp_tab_to_lds(s_exp, threadIdx.x, BLOCK);
	__syncthreads();
	const long long stride = (long long)gridDim.x * BLOCK;
	unsigned long long n_eval = 0, n_out = 0, n_acc = 0;
	for (long long k = (long long)blockIdx.x * BLOCK + threadIdx.x; k < n; k += stride) {
		float c[5];
		{ const v3 o = load3(vo, k); c[0] = u1a[k]; c[1] = u2a[k]; c[2] = o.x; c[3] = o.y; c[4] = o.z; }
		float cur = cts_attack_score<KIND>(b, p, c[0], c[1], mk(c[2], c[3], c[4]), gt, n_out);
		++n_eval;
		for (int it = 0; it < iters; ++it) {
			const uint32_t h = hash_u32(seed, (uint64_t)k * 4096ull + (uint64_t)it, 11u);
			const int w = (int)(h % 5u), e = (int)((h >> 3) % 21u);
			const int delta = (h & 0x80000000u) ? (1 << e) : -(1 << e);
			const float old = c[w];
			const float cand = __uint_as_float(__float_as_uint(old) + (uint32_t)delta);
			if (!(fabsf(cand) < 16.0f)) continue;                        // stay finite and near the family
			c[w] = cand;
			const float r = cts_attack_score<KIND>(b, p, c[0], c[1], mk(c[2], c[3], c[4]), gt, n_out);
			++n_eval;
			if (r > cur) { cur = r; ++n_acc; } else c[w] = old;
		}
		u1a[k] = c[0]; u2a[k] = c[1]; store3(vo, k, mk(c[2], c[3], c[4]));
		best[k] = cur;
	}
	atomicAdd(&counters[0], n_eval); atomicAdd(&counters[1], n_out); atomicAdd(&counters[2], n_acc);
}

} // namespace

namespace djbk {

bool sample_contract_supported(const Brdf &b, const Params &p) { return (b.kind == KIND_BECKMANN || b.kind == KIND_GGX) && cts_params_ok(p); }

hipError_t launch_sample_contract_attack(hipStream_t s, const Brdf &b, const Params &p, long long n, float *u1, float *u2, const View &o, int iters,
                                         uint32_t seed, float *best, unsigned long long *counters)
{
	if (!sample_contract_supported(b, p)) return hipErrorInvalidValue;
	if (b.kind == KIND_GGX) hipLaunchKernelGGL(k_sample_ct_attack<KIND_GGX>, dim3(grid_persistent(n)), dim3(BLOCK), 0, s, b, p, n, u1, u2, o, iters, seed, best, counters);
	else hipLaunchKernelGGL(k_sample_ct_attack<KIND_BECKMANN>, dim3(grid_persistent(n)), dim3(BLOCK), 0, s, b, p, n, u1, u2, o, iters, seed, best, counters);
	return hipGetLastError();
}

hipError_t launch_sample_contract_selftest(hipStream_t s, const Brdf &b, const Params &p, long long n, uint32_t seed, unsigned long long start,
                                           int family, unsigned int *max_bits, unsigned long long *counters)
{
	if (!sample_contract_supported(b, p)) return hipErrorInvalidValue;
	if (b.kind == KIND_GGX) hipLaunchKernelGGL(k_sample_ct_selftest<KIND_GGX>, dim3(grid_persistent(n)), dim3(BLOCK), 0, s, b, p, n, seed, start, family, max_bits, counters);
	else hipLaunchKernelGGL(k_sample_ct_selftest<KIND_BECKMANN>, dim3(grid_persistent(n)), dim3(BLOCK), 0, s, b, p, n, seed, start, family, max_bits, counters);
	return hipGetLastError();
}

// ggx `sample` under DJB_OPT_CONTRACT_1E5 (ggx_sample_contract + the exact per-sample code for declined samples, one launch)
hipError_t launch_sample_ggx_contract(hipStream_t s, const Brdf &b, const Params &p, long long n, const float *u1, const float *u2,
                                      uint32_t s1, uint32_t s2, unsigned long long start, const View &o, const View &out_i)
{
	if (b.kind != KIND_GGX || !cts_params_ok(p)) return hipErrorInvalidValue;
	const dim3 g(grid_persistent(n)), t(BLOCK);
	const View w{ nullptr, nullptr, nullptr, 0 };
	const djbk::CtParams ct{};
	const bool rng = u1 == nullptr, dn = o.stride == 1 && out_i.stride == 1;
#ifdef DJB_EXP_RARE_COUNT
	struct Report { hipStream_t s; ~Report() { unsigned long long h[2 + R_SITES]; (void)hipStreamSynchronize(s); (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_rare), sizeof h);
		fprintf(stderr, "djb_exp: ggx contract sample: common %llu deferred %llu | guard %llu bound %llu clamp %llu degenerate %llu (cumulative)\n",
		        h[0], h[1], h[2 + R_GUARD], h[2 + R_TRIPS], h[2 + R_CLAMP], h[2 + R_DEGENERATE]); } } report{ s };
#endif
	if (rng) { if (dn) hipLaunchKernelGGL((k_sample_bk<false, true, -1, true, true, KIND_GGX>), g, t, 0, s, b, p, n, u1, u2, s1, s2, start, o, out_i, w, (float *)nullptr, ct);
	           else hipLaunchKernelGGL((k_sample_bk<false, true, -1, false, true, KIND_GGX>), g, t, 0, s, b, p, n, u1, u2, s1, s2, start, o, out_i, w, (float *)nullptr, ct); }
	else { if (dn) hipLaunchKernelGGL((k_sample_bk<false, false, -1, true, true, KIND_GGX>), g, t, 0, s, b, p, n, u1, u2, s1, s2, start, o, out_i, w, (float *)nullptr, ct);
	       else hipLaunchKernelGGL((k_sample_bk<false, false, -1, false, true, KIND_GGX>), g, t, 0, s, b, p, n, u1, u2, s1, s2, start, o, out_i, w, (float *)nullptr, ct); }
	return hipGetLastError();
}

// sample / evalp_is (ideal, Schlick or unpolarized Fresnel) of a Beckmann lobe; same contract as launch_sample (djb_kernels_eval.hip), which forwards here
hipError_t launch_sample_beckmann(hipStream_t s, const Brdf &b, const Params &p, long long n, const float *u1, const float *u2,
                                  uint32_t s1, uint32_t s2, unsigned long long start, const View &o, const View &out_i,
                                  const View *out_w, float *out_pdf, bool contract)
{
	dim3 g(grid_persistent(n)), t(BLOCK);
#ifdef DJB_EXPERIMENT
	if (const char *e = getenv("DJB_SAMPLE_GRID_ENV")) g.x = (unsigned int)atoll(e);
#endif
	View w = out_w ? *out_w : View{ nullptr, nullptr, nullptr, 0 };
	const bool is = out_w != nullptr, rng = u1 == nullptr;
	djbk::CtParams ct{};
	const bool ct_is = is && contract && contract_params(b, p, nullptr, &ct);     // evalp_is under the contract: exact direction, contract tail
	auto dense1 = [](const View &v) { return !v.x || v.stride == 1; };
	const bool dn = dense1(o) && dense1(out_i) && dense1(w);
#ifdef DJB_EXP_RARE_COUNT
	struct Report { hipStream_t s; ~Report() { unsigned long long h[2 + R_SITES]; (void)hipStreamSynchronize(s); (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_rare), sizeof h);
		fprintf(stderr, "djb_exp: sample_bk common %llu deferred %llu | logf %llu expf %llu powf %llu exp64 %llu guard %llu tail_loop %llu tail_qf1 %llu trips %llu clamp %llu degenerate %llu (cumulative)\n",
		        h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8], h[9], h[10], h[11]);
#ifdef DJB_EXP_TRIP4_CHECK
		unsigned long long t4[4]; (void)hipMemcpyFromSymbol(t4, HIP_SYMBOL(g_trip4), sizeof t4);
		fprintf(stderr, "djb_exp: trip 4: needed by %llu, shown to converge %llu, SHOWN BUT NOT CONVERGED %llu (must be 0), converge but not shown %llu (cumulative)\n", t4[0], t4[1], t4[2], t4[3]);
#endif
		} } report{ s };
#endif
#define DJB_LAUNCH_S(IS_, RNG_, FRK_, DN_) hipLaunchKernelGGL((k_sample_bk<IS_, RNG_, FRK_, DN_>), g, t, 0, s, b, p, n, u1, u2, s1, s2, start, o, out_i, w, out_pdf, ct)
#define DJB_LAUNCH_S2(IS_, FRK_) do { if (rng) { if (dn) DJB_LAUNCH_S(IS_, true, FRK_, true); else DJB_LAUNCH_S(IS_, true, FRK_, false); } \
                                      else { if (dn) DJB_LAUNCH_S(IS_, false, FRK_, true); else DJB_LAUNCH_S(IS_, false, FRK_, false); } \
                                      return hipGetLastError(); } while (0)
	if (!is && contract && cts_params_ok(p)) {       // DJB_OPT_CONTRACT_1E5: directions within 1e-5 (bk_sample_contract)
		if (rng) { if (dn) hipLaunchKernelGGL((k_sample_bk<false, true, -1, true, true>), g, t, 0, s, b, p, n, u1, u2, s1, s2, start, o, out_i, w, out_pdf, ct);
		           else hipLaunchKernelGGL((k_sample_bk<false, true, -1, false, true>), g, t, 0, s, b, p, n, u1, u2, s1, s2, start, o, out_i, w, out_pdf, ct); }
		else { if (dn) hipLaunchKernelGGL((k_sample_bk<false, false, -1, true, true>), g, t, 0, s, b, p, n, u1, u2, s1, s2, start, o, out_i, w, out_pdf, ct);
		       else hipLaunchKernelGGL((k_sample_bk<false, false, -1, false, true>), g, t, 0, s, b, p, n, u1, u2, s1, s2, start, o, out_i, w, out_pdf, ct); }
		return hipGetLastError();
	}
	if (!is) DJB_LAUNCH_S2(false, -1);
	if (ct_is) {
#define DJB_LAUNCH_CT(FRK_) do { \
		if (rng) { if (dn) hipLaunchKernelGGL((k_sample_bk<true, true, FRK_, true, true>), g, t, 0, s, b, p, n, u1, u2, s1, s2, start, o, out_i, w, out_pdf, ct); \
		           else hipLaunchKernelGGL((k_sample_bk<true, true, FRK_, false, true>), g, t, 0, s, b, p, n, u1, u2, s1, s2, start, o, out_i, w, out_pdf, ct); } \
		else { if (dn) hipLaunchKernelGGL((k_sample_bk<true, false, FRK_, true, true>), g, t, 0, s, b, p, n, u1, u2, s1, s2, start, o, out_i, w, out_pdf, ct); \
		       else hipLaunchKernelGGL((k_sample_bk<true, false, FRK_, false, true>), g, t, 0, s, b, p, n, u1, u2, s1, s2, start, o, out_i, w, out_pdf, ct); } \
		return hipGetLastError(); } while (0)
		if (b.fr.kind == FR_IDEAL) DJB_LAUNCH_CT(FR_IDEAL);
		if (b.fr.kind == FR_SCHLICK) DJB_LAUNCH_CT(FR_SCHLICK);
		if (b.fr.kind == FR_UNPOLARIZED) DJB_LAUNCH_CT(FR_UNPOLARIZED);
#undef DJB_LAUNCH_CT
	}
	if (b.fr.kind == FR_IDEAL) DJB_LAUNCH_S2(true, FR_IDEAL);
	if (b.fr.kind == FR_SCHLICK) DJB_LAUNCH_S2(true, FR_SCHLICK);
	if (b.fr.kind == FR_UNPOLARIZED) DJB_LAUNCH_S2(true, FR_UNPOLARIZED);
	return hipErrorInvalidValue;           // evalp_is with a run-time Fresnel kind: launch_sample keeps those on k_sample
#undef DJB_LAUNCH_S2
#undef DJB_LAUNCH_S
}

} // namespace djbk
