// djb_kernels_contract.hip -- DJB_OPT_CONTRACT_1E5: eval / evalp / pdf of the analytic lobes inside the
// north star's VALUE contract (|result - reference| <= 1e-5 |reference|) instead of bit-identically.
//
// The default kernels (djb_kernels_eval.hip) spend ~350 VALU issue slots per GGX pair on reproducing the reference's
// roundings: 15 correctly rounded divisions, 2 square roots, ~35 fp64 operations (microfacet::eval / pdf,
// dj_brdf.h:1529-1555, 1713-1730 with ndf 1559, p22 1574, sigma 1619, g1 1633, gaf 1644).  This file evaluates the
// same formulas with v_rsq_f32 / v_rcp_f32 (1 ulp each), algebraically merged denominators (6 transcendental-rate
// instructions per eval+pdf pair instead of 15 divisions) and float4 non-temporal streams, which makes the GGX kernel
// HBM-bound.  What keeps it inside the contract for EVERY input, not only for typical ones:
//
//  * every branch the reference takes on a computed value (g1 > 0, G > 0, h.z > 1e-4, dot(o, h) > 0) is decided from
//    quantities that are bit-identical to the reference's (sums and products of the inputs in the reference's order),
//    or the pair is handed to tier 2 when the approximate operand lies near the threshold;
//  * the places where the reference's own float chain is ill-conditioned -- dot(o, h) and dot(i, h) of the pdf near
//    theta_d = 90 deg, where one ulp of h moves the quotient by more than 1e-5 -- are detected (CT_DOT_MIN) and handed
//    to tier 2 as well: an approximate evaluation cannot track the reference's rounding noise there;
//  * tier 2 (k_ct_fixup) runs eval_one -- the bit-exact per-pair code of the default kernels -- on the listed pairs.
//    Like the MERL tiers it streams {k, i, o} records, and on worklist overflow it redoes the whole batch exactly.
//
// Error budget of the fast path (u = 2^-24, all relative, tx = ty = 0, |rho| <= CT_RHO_MAX): D <= ~36 u (68 u at the
// rho limit), shadowing-masking and the merged 1 / (4 o.z i.z) <= 15 u, the pdf's two dot products <= 18 u each with
// CT_DOT_MIN = 0.25: worst case sum <= ~1e-5 * 0.5; measured maxima are reported by djb_selftest_contract
// (tests/test_gpu_contract.py, bench.py secondary.ggx_eval_pdf_contract).
// MERL / UTIA indices and cell decisions are never approximated: they do not go through this file.
#include "djb_internal.hpp"
#include "djb_worklist.hpp"

using namespace djbdev;

namespace {

constexpr int BLOCK = 256;

inline int grid_for(long long n, long long cap = 256LL * 64)
{
	long long blocks = (n + BLOCK - 1) / BLOCK;
	if (blocks > cap) blocks = cap;
	if (blocks < 1) blocks = 1;
	return (int)blocks;
}

#include "djb_contract_device.inc"   // CtParams, the shortcuts, ct_sigma, the Fresnel terms, the evalp_is tail (shared with the samplers)

// Beckmann.  exp(-r^2) makes D as sensitive as r^2 is large: an ulp of the half vector's slope moves D by r^2 2^-23, and
// the reference's own float chain carries several -- so the slope of h is computed with the REFERENCE's operations
// (h = normalize(i + o) through the guarded inverse square root, two IEEE divisions, the exact divisions by the
// launch-uniform denominators: r^2 is the reference's, bit for bit), which also makes h.z > 1e-4, dot(o, h) and dot(i, h)
// the reference's own values: no guard bands are needed for them.  Everything else -- both sigmas with their exp / erf,
// the NDF's exp, shadowing-masking, the final quotients -- runs on the reciprocal / rsq / exp2 instructions.
template <int WANT, int FRK>
DJB_DEV bool ct_eval_beckmann(const CtParams &c, v3 i, v3 o, v3 &fr, float &pdf)
{
	const bool live = (o.z > 0.0f) & (!c.shadow | (i.z > 0.0f));
	bool ok = (o.z > CT_LO) & (i.z > CT_LO);
	const v3 h = normalize(add(i, o));                        // exact (dj_brdf.h:1536, 630-637)
	const bool facing = h.z > 1e-4f;                          // microfacet::ndf's cut (dj_brdf.h:1561): the reference's decision
	const float xs = -h.x / h.z, ys = -h.y / h.z;             // dj_brdf.h:1564
	const float x_ = fdiv_r(xs, c.ax, c.R_ax);                 // microfacet::p22, dj_brdf.h:1574-1587
	const float y_ = fdiv_r(c.ax * ys - c.rho_ay * xs, c.t2, c.R_t2);
	const float r2 = x_ * x_ + y_ * y_;
	// r^2 >= 104: the reference's float(exp(-r^2) / pi) IS zero (exp(-103.5) / pi = 3.6e-46 is below half of the smallest float
	// denormal, 7.0e-46) and with it p22, D, eval and pdf -- decided on the reference's own r^2, so the zeros are its zeros.  A sharp lobe
	// (alpha = 0.05: r^2 = 400 tan^2 theta_h) puts most random pairs there.  Below 80 the results are normal floats and ct_exp_neg keeps its
	// 2 ulp (its argument split holds down to 2^-126 = exp(-87.3)); between 80 and 104 D heads for the denormals: tier 2.
	const bool zero_d = facing & (r2 >= 104.0f);
	ok &= (r2 < 80.0f) | !facing | zero_d;
	const float c2 = h.z * h.z, c4 = c2 * c2;
	const float en = ct_exp_neg(-fminf(r2, 100.0f));          // D = exp(-r^2) / (pi t2 c4)
	const float sig_o = ct_sigma<KIND_BECKMANN>(c, o, ok);
	float den4;
	if (c.shadow) {
		const float sig_i = ct_sigma<KIND_BECKMANN>(c, i, ok);
		den4 = 4.0f * ((i.z * sig_o + o.z * sig_i) - i.z * o.z);
	} else den4 = 4.0f * (sig_o * i.z);
	const float oh = dot(o, h);
	const bool on = live & facing;
	fr = mk(0, 0, 0); pdf = 0.0f;
	if (WANT & 3) {
		float e = (c.k_d * en) * rcp_(c4 * den4);
		if (WANT & 2) e *= i.z;
		ok &= ((e < 1e30f) & (e > 1e-33f)) | !on | zero_d;    // the exponential tail towards the denormals: tier 2 (the relative
		                                                       // contract has no meaning there; zeros must match exactly)
		e = (on & !zero_d) ? e : 0.0f;
		fr = ct_fresnel_times<FRK>(c, oh, e);
		// eval = evalp / i.z even where evalp is vec3(0) (dj_brdf.h:1551-1555): NaN for a dead pair with i.z = 0 (or NaN)
		if ((WANT & 1) && !live && ((i.z == 0.0f) | (i.z != i.z))) fr = mk(__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""));
	}
	if (WANT & 4) {
		const float ih = dot(i, h);
		float q = (oh * (c.k_d * en)) * rcp_(c4 * (4.0f * ih * sig_o));
		const bool pos = oh > 0.0f;                            // vndf is 0 unless dot(o, h) > 0 (dj_brdf.h:1605): the reference's decision
		ok &= ((((q < 1e30f) & (q > 1e-33f)) | zero_d) & (ih > CT_LO)) | !(on & pos);   // zero_d: +0 / (4 dot(i, h)) with dot(i, h) > 0
		pdf = (on & pos & !zero_d) ? q : 0.0f;
	}
	return ok | !live;
}

// ABC (abc::eval, dj_brdf.h:3633-3645: kD / pi + G F D / (pi i.z o.z), D = A / (1 + B (1 - h.z))^C, G = min of the two
// min(1, 2 h.z k.z / dot(h, k)), F = unpolarized(ior)).  B reaches 8e6 in the published fits, so D moves by B C 2^-24 /
// (1 + B (1 - h.z)) per ulp of h.z -- percent: like Beckmann's slope, h is computed with the REFERENCE's operations
// (normalize through the guarded inverse square root), which also makes dot(i, h), dot(h, k) and the Fresnel term's
// g = float(sqrt(n^2 + c^2 - 1)) (guarded fp64 square root) the reference's own floats: its ill-conditioned
// differences g - c (ior -> 1) and 1 - h.z start from identical operands.  What is approximated: the power (frexp, v_log_f32
// of the mantissa, the product with C in fp64, v_exp_f32 of the fraction: ~3e-7 relative for C <= 16), three divisions
// by v_rcp_f32, the final products.  Sum of two non-negative terms: the relative error of the result is at most that of
// the specular term, ~20 ulp.  pdf = i.z / pi (brdf::pdf, dj_brdf.h:842): one multiplication.
template <int WANT>
DJB_DEV bool ct_eval_abc(const CtParams &c, v3 i, v3 o, v3 &fr, float &pdf)
{
	const bool live = (i.z > 0.0f) & (o.z > 0.0f);            // dj_brdf.h:3636; false for NaN like the reference's test
	fr = mk(0, 0, 0); pdf = 0.0f;
	if (WANT & 4) pdf = i.z * 0.318309886f;                   // computed for every pair, live or not, as in eval_one
	bool ok = true;
	if (WANT & 3) {
		const v3 h = normalize(add(i, o));                    // exact
		const float cd = sat_(dot(i, h));
		// fresnel::unpolarized, dj_brdf.h:1292-1303 (one ior for the three channels: abc's constructor, :3623)
		const float g = sqrt_to_f32(D(c.ior * c.ior + cd * cd) - 1.0);
		const float t1 = cd * (g + cd) - 1.0f, t2 = cd * (g - cd) + 1.0f;   // == float(double(c (g +- c)) -+ 1.0): one rounding of an exact difference
		const float q3 = t1 * rcp_(t2), gm = g - cd, gp = g + cd, q4 = gm * rcp_(gp);
		const float Fr = (0.5f * (q4 * q4)) * (1.0f + q3 * q3);
		ok &= (gm > 0.0f) & in_range(gp) & in_range(t2);
		// abc::gaf, dj_brdf.h:3647-3655
		const float di = dot(h, i), dO = dot(h, o);
		ok &= in_range(di) & in_range(dO);
		const float g1i = fminf(1.0f, 2.0f * ((h.z * i.z) * rcp_(di))), g1o = fminf(1.0f, 2.0f * ((h.z * o.z) * rcp_(dO)));
		const float G = fminf(g1i, g1o);
		// (1 + B (1 - h.z))^C
		const double td = __builtin_fma(c.B, 1.0 - D(h.z), 1.0);
		const int e2 = __builtin_amdgcn_frexp_exp(td);
		const float lm = __builtin_amdgcn_logf(F(__builtin_amdgcn_frexp_mant(td)));   // log2 of the mantissa in [0.5, 1)
		const double y = c.C * (D(lm) + (double)e2);
		const double yn = __builtin_rint(y);
		const float den = __builtin_amdgcn_ldexpf(__builtin_amdgcn_exp2f(F(y - yn)), (int)yn);
		ok &= (td >= 1.0) & (td < 1e30) & (y < 100.0);
		const float k = ((G * Fr) * rcp_(den)) * rcp_((3.14159274f * i.z) * o.z);
		const float amin = fminf(c.A[0], fminf(c.A[1], c.A[2]));
		ok &= ((k * amin > 1e-30f) | (k == 0.0f)) & (k < 1e30f) & (i.z > CT_LO) & (o.z > CT_LO);
		v3 e = live ? mk(c.kd_pi[0] + c.A[0] * k, c.kd_pi[1] + c.A[1] * k, c.kd_pi[2] + c.A[2] * k) : mk(0, 0, 0);
		if (WANT & 2) e = scale(i.z, e);                      // brdf::evalp = eval * i.z, dead pairs included (NaN i.z -> NaN, as in eval_one)
		fr = e;
	}
	return ok | !live;
}

// ---- SGD (sgd::eval, dj_brdf.h:3454-3468: (Kd + Ks F D G / (i.z o.z)) / pi per channel; 9 pow + 9 exp + 2 acos in fp64 per pair
// in the bit-exact kernel, every one of them glibc's algorithm).  Tier 1:
//   h            the reference's own float (exact normalize, as for ABC): c = sat(dot(i, h)) and h.z are its operands.
//   F            f0 - c f1 + (1 - c)^5 (1 - f0) in float (the power by three products, 4 u); the sum can cancel when f1 is
//                large: |F| < 0.05 (|f0| + |c f1| + |...|) -> tier 2.
//   D            kap e^-ax / (pi ax^p hz^4), ax = alpha + (1 - hz^2) / (hz^2 alpha): alpha goes down to 1.6e-5, so ax reaches the
//                hundreds before e^-ax underflows and an ulp32 of ax would move D by ax 2^-24.  ax is therefore formed in fp64
//                (one division), and e^-ax / ax^p is ONE v_exp_f32 of the fp64 exponent -ax log2(e) - p log2(ax) split into
//                integer + fraction (log2 of the mantissa by v_log_f32: |p| <= 4 keeps its 6e-8 below 2.4e-7).  Where e^-ax has
//                underflowed the specular term vanishes against Kd > 0 and the pair stays in tier 1.
//   G = g1 g1    g1 = clamp(1 - w), w = lambda expm1(x), x = c t1^k, t1 = max(0, acos(k.z) - theta0): a WALL -- k reaches 856, c
//                1e38, lambda 1.5e7 in the published rows.  Tier 1 evaluates it with a float acos (A&S 4.4.46, |err| <= SGD_DT
//                with the float roundings) and v_log / v_exp; the error this makes in g1 is
//                    dg1 = lambda e^x x (dx / x),     dx / x <= R = k (SGD_DT / t1 + 1.2e-7 |ln t1|) + 4e-7
//                with R bounded per channel over the t1 that matter (ct_params_sgd), so "dg1 <= 1.5e-6 g1" is a threshold on x
//                alone: x <= x_max (the plateau and the foot of the wall) -> tier 1; x >= x_zero (beyond the wall, g1 = 0 with
//                certainty) -> tier 1; in between -- ON the wall -- the fp64 acos / pow / exp of tier 2 answer.  Which share of
//                the pairs that is depends on the material (profiles/r04/contract_sgd_materials.txt: 22 of the 100 published
//                rows below 5 %, 40 below 10 % on the bench distribution); the host routes a material whose last large
//                batch listed more than 15 % straight to the bit-exact kernel.
constexpr double SGD_DT = 4e-7;        // |float acos - acos| bound, radians (A&S eps 2e-8 + Horner / sqrt roundings)

DJB_DEV float ct_acos01(float x)      // Abramowitz & Stegun 4.4.46 on [0, 1]: sqrt(1 - x) P7(x)
{
	float p = -0.0012624911f;
	p = p * x + 0.0066700901f; p = p * x - 0.0170881256f; p = p * x + 0.0308918810f;
	p = p * x - 0.0501743046f; p = p * x + 0.0889789874f; p = p * x - 0.2145988016f; p = p * x + 1.5707963050f;
	const float s = 1.0f - x;
	return (s * rsq_(fmaxf(s, 1e-30f))) * p;
}
// g1 of one direction / channel.  x <= x_max <= 0.5: expm1 by its series (six terms: 3e-7 relative at 0.5); x >= x_zero: 0.
DJB_DEV float ct_sgd_g1(float theta_k, float th0_hi, float th0_lo, float lam, float l2c, float kk, float x_max, float x_zero, bool &ok)
{
	const float d = (theta_k - th0_hi) - th0_lo;
	const float t1 = fmaxf(d, 1e-30f);                          // d <= 0: the plateau (x underflows to 0, g1 = 1)
	const float lx = l2c + kk * __builtin_amdgcn_logf(t1);
	const float x = __builtin_amdgcn_exp2f(fminf(lx, 7.0f));
	const float xs = fminf(x, 0.5f);
	const float em1 = xs * (1.0f + xs * (0.5f + xs * (0.166666672f + xs * (0.0416666679f + xs * (0.00833333377f + xs * 0.00138888892f)))));
	const float g1 = fmaxf(1.0f - lam * em1, 0.0f);
	const bool beyond = x >= x_zero;
	ok &= (x <= x_max) | beyond;
	return beyond ? 0.0f : g1;
}
template <int WANT>
DJB_DEV bool ct_eval_sgd(const CtParams &c, v3 i, v3 o, v3 &fr, float &pdf)
{
	const bool live = (i.z > 0.0f) & (o.z > 0.0f);            // dj_brdf.h:3456
	fr = mk(0, 0, 0); pdf = 0.0f;
	if (WANT & 4) pdf = i.z * 0.318309886f;                   // brdf::pdf = i.z / pi
	bool ok = true;
	if (WANT & 3) {
		const v3 h = normalize(add(i, o));                    // exact
		const float cd = sat_(dot(i, h));
		ok &= (h.z > 1e-3f) & (i.z > 1e-6f) & (o.z > 1e-6f) & (i.z <= 1.0f) & (o.z <= 1.0f);
		// Fresnel (fresnel::sgd, dj_brdf.h:1330-1336)
		const float c1 = 1.0f - cd, c2_ = c1 * c1, c5 = c2_ * c2_ * c1;
		// NDF: shared fp64 part
		const double c2d = D(h.z) * D(h.z);
		const double t2 = (1.0 - c2d) / c2d;
		const float hz2 = h.z * h.z, rc4 = rcp_(hz2 * hz2);
		const float th_i = ct_acos01(fminf(i.z, 1.0f)), th_o = ct_acos01(fminf(o.z, 1.0f));
		const float riz = rcp_(i.z * o.z);
		float e[3];
#pragma unroll
		for (int ch = 0; ch < 3; ++ch) {
			const float t_f1 = cd * c.sf1[ch], t_p = c5 * c.s1mf0[ch];
			const float Fr = (c.sf0[ch] - t_f1) + t_p;
			ok &= fabsf(Fr) >= 0.05f * (fabsf(c.sf0[ch]) + fabsf(t_f1) + fabsf(t_p));
			const double ax = __builtin_fma(t2, c.inv_alpha[ch], c.alpha[ch]);
			const int ea = __builtin_amdgcn_frexp_exp(ax);
			const float lm = __builtin_amdgcn_logf(F(__builtin_amdgcn_frexp_mant(ax)));
			double y = __builtin_fma(-1.4426950408889634, ax, D(c.lkap[ch])) - D(c.p_[ch]) * (D(lm) + (double)ea);
			ok &= (ax > 0.0) & (y < 100.0);                      // overflow side: tier 2; a NaN h.z fails the comparison too
			y = __builtin_fmax(y, -200.0);                       // e^-ax underflowed: D = 0 against Kd >= SGD_KD_MIN (ct_params_sgd)
			const double yn = __builtin_rint(y);
			const float Dn = __builtin_amdgcn_ldexpf(__builtin_amdgcn_exp2f(F(y - yn)), (int)yn) * rc4;
			const float G = ct_sgd_g1(th_i, c.th0_hi[ch], c.th0_lo[ch], c.lam[ch], c.l2c[ch], c.kk[ch], c.x_max[ch], c.x_zero[ch], ok) *
			                ct_sgd_g1(th_o, c.th0_hi[ch], c.th0_lo[ch], c.lam[ch], c.l2c[ch], c.kk[ch], c.x_max[ch], c.x_zero[ch], ok);
			const float spec = (c.ks[ch] * ((Fr * Dn) * G)) * riz;
			ok &= (spec >= 0.0f) & (spec < 1e30f);
			e[ch] = (c.kd[ch] + spec) * 0.318309886f;
		}
		v3 v = live ? mk(e[0], e[1], e[2]) : mk(0, 0, 0);
		if (WANT & 2) v = scale(i.z, v);
		fr = v;
	}
	return ok | !live;
}

// one pair; false = tier 2.  fr / pdf follow eval_one's WANT convention (1 eval, 2 evalp, 4 pdf)
template <int KIND, int WANT, int FRK>
DJB_DEV bool ct_eval_one(const CtParams &c, v3 i, v3 o, v3 &fr, float &pdf)
{
	static_assert(KIND == KIND_GGX || KIND == KIND_BECKMANN || KIND == KIND_ABC || KIND == KIND_SGD, "contract mode: GGX, Beckmann, ABC and SGD");
	if (KIND == KIND_ABC) return ct_eval_abc<WANT>(c, i, o, fr, pdf);
	if (KIND == KIND_SGD) return ct_eval_sgd<WANT>(c, i, o, fr, pdf);
	if (KIND == KIND_BECKMANN) return ct_eval_beckmann<WANT, FRK>(c, i, o, fr, pdf);
	// g1(k) > 0 <=> dot(k, m_n) = k.z > 0 (dj_brdf.h:1633-1642); gaf > 0 <=> both (shadow) / g1(o) (dj_brdf.h:1644-1665)
	// -- decided on the inputs themselves, NaN included: the reference's comparisons are false for NaN and return zeros.
	// Branch-free: dead pairs run the arithmetic on whatever they hold and select the zeros at the end.
	const bool live = (o.z > 0.0f) & (!c.shadow | (i.z > 0.0f));
	bool ok = (o.z > CT_LO) & (i.z > CT_LO);
	v3 s = add(i, o);
	float m = dot(s, s);
	ok &= (in_range(m));
	float r = rsq_(m), hz = r * s.z;
	ok &= (hz > CT_HZ_MIN);
	float sig_o = ct_sigma<KIND>(c, o, ok);
	// 4 * o.z * i.z / G:  G = t / (g1i + g1o - t) with g1 = k.z / sigma  ->  4 (i.z sig_o + o.z sig_i - i.z o.z)
	float den4;
	if (c.shadow) {
		float sig_i = ct_sigma<KIND>(c, i, ok);
		den4 = 4.0f * ((i.z * sig_o + o.z * sig_i) - i.z * o.z);
	} else den4 = 4.0f * (sig_o * i.z);
	// slope of h and the stretched slope (microfacet::ndf / p22, dj_brdf.h:1559-1587)
	float rz = rcp_(s.z), xs = -s.x * rz, ys = -s.y * rz;
	float x_ = xs * c.r_ax;
	float y_ = (c.ax * ys - c.rho_ay * xs) * c.r_t2;
	float r2 = x_ * x_ + y_ * y_;
	ok &= (r2 < 1e14f);
	float t = 1.0f + r2, c2 = hz * hz;
	float T = (t * t) * (c2 * c2);                            // D = k_d / T   (ggx::p22_radial, dj_brdf.h:2056)
	float oh = r * dot(o, s);
	fr = mk(0, 0, 0); pdf = 0.0f;
	if (WANT & 3) {
		float e = c.k_d * rcp_(T * den4);                      // D G / (4 o.z i.z)
		if (WANT & 2) e *= i.z;                                // evalp
		ok &= (e < 1e30f);
		e = live ? e : 0.0f;
		fr = ct_fresnel_times<FRK>(c, oh, e);
		// eval = evalp / i.z even where evalp is vec3(0) (dj_brdf.h:1551-1555): NaN for a dead pair with i.z = 0 (or NaN)
		if ((WANT & 1) && !live && ((i.z == 0.0f) | (i.z != i.z))) fr = mk(__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""));
	}
	if (WANT & 4) {
		float ih = r * dot(i, s);
		ok &= (oh > CT_DOT_MIN) & (ih > CT_DOT_MIN);
		float q = (oh * c.k_d) * rcp_(T * (4.0f * ih * sig_o));  // (o.h) D / (sigma(o) 4 (i.h)), dj_brdf.h:1602-1615, 1724
		ok &= (q < 1e30f);
		pdf = live ? q : 0.0f;
	}
	return ok | !live;
}

// The tier-2 worklist is SHARDED: CT_SHARDS counters and list segments, workgroup b appends to shard b mod CT_SHARDS.
// With one workgroup per 1024 pairs (the grid that streams fastest, launch_ct) nearly every wave has something to append
// when ~1 % of the pairs are tier-2 (Beckmann's exp(-r^2) tail), i.e. one returning atomic per wave: 3.5e5 of them per 1e8
// pairs on ONE address ran at the ~88 per microsecond a single word sustains -- 4.0 ms for a kernel whose streams take
// 0.7 ms (profiles/r03/contract_beckmann_worklist.txt).  64 words take the same traffic in 0.06 ms.
constexpr unsigned int CT_SHARDS = djbk::CONTRACT_SHARDS;
constexpr unsigned int CT_STRIDE = djbk::CONTRACT_COUNTER_STRIDE;    // in words: atomics on words of one cache line still serialise

template <int KIND, int WANT, int FRK>
__global__ __launch_bounds__(BLOCK) void k_ct_fast_v4(CtParams c, long long n4, View vi, View vo, View vout, float *out_pdf,
                                                      uint4 *list_all, unsigned int cap, unsigned int *counts)
{
	__shared__ WaveBuf wbuf[BLOCK / 64];
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	unsigned int wcount = 0;
	const unsigned int shard = blockIdx.x % CT_SHARDS;
	uint4 *list = list_all + 2 * (size_t)shard * cap;            // cap = records per shard
	unsigned int *count = counts + (size_t)shard * CT_STRIDE;    // one counter per 128-byte line
	const float4 *ix4 = (const float4 *)vi.x, *iy4 = (const float4 *)vi.y, *iz4 = (const float4 *)vi.z;
	const float4 *ox4 = (const float4 *)vo.x, *oy4 = (const float4 *)vo.y, *oz4 = (const float4 *)vo.z;
	const long long stride = (long long)gridDim.x * BLOCK;
	for (long long q0 = (long long)blockIdx.x * BLOCK; q0 < n4; q0 += stride) {
		const long long q = q0 + threadIdx.x;
		bool amb[4] = { false, false, false, false };
		float ixs[4], iys[4], izs[4], oxs[4], oys[4], ozs[4];
		if (q < n4) {
			float4 ax = nt_load4(ix4 + q), ay = nt_load4(iy4 + q), az = nt_load4(iz4 + q),
			       bx = nt_load4(ox4 + q), by = nt_load4(oy4 + q), bz = nt_load4(oz4 + q);
			nt_load_wait6(ax, ay, az, bx, by, bz);
			ixs[0] = ax.x; ixs[1] = ax.y; ixs[2] = ax.z; ixs[3] = ax.w;
			iys[0] = ay.x; iys[1] = ay.y; iys[2] = ay.z; iys[3] = ay.w;
			izs[0] = az.x; izs[1] = az.y; izs[2] = az.z; izs[3] = az.w;
			oxs[0] = bx.x; oxs[1] = bx.y; oxs[2] = bx.z; oxs[3] = bx.w;
			oys[0] = by.x; oys[1] = by.y; oys[2] = by.z; oys[3] = by.w;
			ozs[0] = bz.x; ozs[1] = bz.y; ozs[2] = bz.z; ozs[3] = bz.w;
			float rr[4], gg[4], bb[4], pp[4];
#pragma unroll
			for (int j = 0; j < 4; ++j) {
				v3 fr; float pdf;
				amb[j] = !ct_eval_one<KIND, WANT, FRK>(c, mk(ixs[j], iys[j], izs[j]), mk(oxs[j], oys[j], ozs[j]), fr, pdf);
				rr[j] = fr.x; gg[j] = fr.y; bb[j] = fr.z; pp[j] = pdf;
			}
			// tier-2 pairs get a placeholder here; k_ct_fixup runs after this kernel on the same stream and overwrites it
			if (WANT & 3) {
				nt_store4(rr[0], rr[1], rr[2], rr[3], (float4 *)vout.x + q);
				nt_store4(gg[0], gg[1], gg[2], gg[3], (float4 *)vout.y + q);
				nt_store4(bb[0], bb[1], bb[2], bb[3], (float4 *)vout.z + q);
			}
			if (WANT & 4) nt_store4(pp[0], pp[1], pp[2], pp[3], (float4 *)out_pdf + q);
		}
		if (__ballot(amb[0] | amb[1] | amb[2] | amb[3])) {
#pragma unroll
			for (int j = 0; j < 4; ++j)
				wl_push(wbuf[wave], wcount, lane, list, cap, count, amb[j], (unsigned int)(4 * q + j),
				        mk(ixs[j], iys[j], izs[j]), mk(oxs[j], oys[j], ozs[j]));
		}
	}
	if (wcount) wl_flush(wbuf[wave], wcount, lane, list, cap, count);
}

// tier 2: the bit-exact per-pair code on the listed pairs (or on the whole batch when a shard of the list overflowed)
template <int KIND, int WANT, int FRK>
__global__ __launch_bounds__(BLOCK) void k_ct_fixup(Brdf b, Params p, long long n, View vi, View vo, View vout, float *out_pdf,
                                                    const uint4 *list, unsigned int cap, const unsigned int *counts)
{
	__shared__ unsigned int s_counts[CT_SHARDS];
	__shared__ int s_over;
	// the fp64 exp / pow / acos of glibc that eval_one<SGD | ABC | BECKMANN> calls: tables to LDS, as in k_eval (sgd's tier 2 can
	// be a large share of the batch: the wall of its shadowing term)
	constexpr bool EXPT = KIND == KIND_BECKMANN || KIND == KIND_SGD || KIND == KIND_ABC, POWT = KIND == KIND_SGD || KIND == KIND_ABC,
	               ACOST = KIND == KIND_SGD;
	__shared__ unsigned long long s_exp[EXPT ? 256 : 1];
	__shared__ double s_pow[POWT ? 384 : 1];
	__shared__ double s_acos[ACOST ? 2568 + 128 : 1];
	if (EXPT) b.exp_lds = glibc_exp_tab_to_lds(s_exp, threadIdx.x, BLOCK);
	if (POWT) b.pow_lds = glibc_pow_tab_to_lds(s_pow, threadIdx.x, BLOCK);
	if (ACOST) b.acos_lds = glibc_acos_tab_to_lds(s_acos, threadIdx.x, BLOCK);
	if (threadIdx.x == 0) s_over = 0;
	__syncthreads();
	if (threadIdx.x < CT_SHARDS) { const unsigned int cnt = counts[(size_t)threadIdx.x * CT_STRIDE]; s_counts[threadIdx.x] = cnt; if (cnt > cap) s_over = 1; }
	__syncthreads();
	if (!s_over) {
		// records numbered through, shard after shard, and dealt to the threads by that number (as in k_merl_fixup)
		__shared__ unsigned int s_first[CT_SHARDS + 1];
		if (threadIdx.x == 0) {
			unsigned int acc = 0;
			for (unsigned int sh = 0; sh < CT_SHARDS; ++sh) { s_first[sh] = acc; acc += s_counts[sh]; }
			s_first[CT_SHARDS] = acc;
		}
		__syncthreads();
		const unsigned int total = s_first[CT_SHARDS], stride = gridDim.x * BLOCK;
		for (unsigned long long g = (unsigned long long)blockIdx.x * BLOCK + threadIdx.x; g < total; g += stride) {
			unsigned int sh = 0;
#pragma unroll
			for (unsigned int step = CT_SHARDS / 2; step; step >>= 1) if (s_first[sh + step] <= (unsigned int)g) sh += step;
			const size_t j = (size_t)sh * cap + ((unsigned int)g - s_first[sh]);
			uint4 ra = list[2 * j], rb = list[2 * j + 1];
			const long long k = (long long)ra.x;
			v3 i = mk(__uint_as_float(ra.y), __uint_as_float(ra.z), __uint_as_float(ra.w));
			v3 o = mk(__uint_as_float(rb.x), __uint_as_float(rb.y), __uint_as_float(rb.z));
			v3 fr; float pdf;
			eval_one<KIND, WANT, FRK>(b, p, i, o, fr, pdf);
			if (WANT & 3) store3(vout, k, fr);
			if (WANT & 4) out_pdf[k] = pdf;
		}
	} else {
		const long long stride = (long long)gridDim.x * BLOCK;
		for (long long k = (long long)blockIdx.x * BLOCK + threadIdx.x; k < n; k += stride) {
			v3 fr; float pdf;
			eval_one<KIND, WANT, FRK>(b, p, load3(vi, k), load3(vo, k), fr, pdf);
			if (WANT & 3) store3(vout, k, fr);
			if (WANT & 4) out_pdf[k] = pdf;
		}
	}
}

// ---- measurement: fast path vs the bit-exact path on generated directions.  stats[0..1]: max relative difference of
// eval rgb / pdf among fast-path pairs (float bits, atomicMax); counters: [0] pairs, [1] tier-2 pairs, [2] fast-path
// values where exactly one of the two results is zero, [3] fast-path values outside 1e-5
template <int KIND, int FRK>
__global__ __launch_bounds__(BLOCK) void k_ct_selftest(Brdf b, Params p, CtParams c, long long n, uint32_t seed_i, uint32_t seed_o,
                                                       unsigned long long start, int family, unsigned int *max_bits,
                                                       unsigned long long *counters)
{
	const long long stride = (long long)gridDim.x * BLOCK;
	float me = 0, mp = 0;
	unsigned long long n_t2 = 0, n_zero = 0, n_out = 0, n_all = 0;
	for (long long k = (long long)blockIdx.x * BLOCK + threadIdx.x; k < n; k += stride) {
		v3 i = gen_direction(seed_i, start + (unsigned long long)k), o = gen_direction(seed_o, start + (unsigned long long)k);
		if (family == 1) {            // grazing, opposite azimuths: theta_d near 90 deg
			float a = 0.02f + 0.3f * gen_uniform(seed_i ^ 0x51u, start + (unsigned long long)k);
			o = mk(-i.x, -i.y, i.z); i.z = a * i.z; o.z = a * o.z;
			i = normalize(i); o = normalize(o);
		} else if (family == 2) {     // near-normal incidence
			i = normalize(mk(0.01f * i.x, 0.01f * i.y, 1.0f)); o = normalize(mk(0.02f * o.x, 0.02f * o.y, 1.0f));
		} else if (family == 3) {     // one direction at the horizon
			o.z = 1e-3f * o.z; o = normalize(o);
		} else if (family == 4) {     // un-normalised inputs
			i = scale(0.5f + gen_uniform(seed_i ^ 0x77u, start + (unsigned long long)k), i); o = scale(3.0f, o);
		}
		v3 fa, fe; float pa, pe;
		++n_all;
		if (!ct_eval_one<KIND, 5, FRK>(c, i, o, fa, pa)) { ++n_t2; continue; }
		eval_one<KIND, 5, FRK>(b, p, i, o, fe, pe);
		const float va[4] = { fa.x, fa.y, fa.z, pa }, ve[4] = { fe.x, fe.y, fe.z, pe };
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			if (va[j] == ve[j]) continue;
			if (va[j] == 0.0f || ve[j] == 0.0f || !(fabsf(ve[j]) >= 1.2e-38f)) { ++n_zero; continue; }   // a zero on one side only, or a sub-normal reference value: must be equal
			float rel = fabsf(va[j] - ve[j]) / fabsf(ve[j]);
			if (!(rel <= 1e-5f)) ++n_out;
			if (j < 3) me = fmaxf(me, rel); else mp = fmaxf(mp, rel);
		}
	}
	atomicMax(&max_bits[0], __float_as_uint(me));
	atomicMax(&max_bits[1], __float_as_uint(mp));
	atomicAdd(&counters[0], n_all); atomicAdd(&counters[1], n_t2);
	atomicAdd(&counters[2], n_zero); atomicAdd(&counters[3], n_out);
}

// abc: the published rows and anything like them (one ior > 1 for the three channels, the constructor's Fresnel term)
bool ct_params_abc(const Brdf &b, const double *m, CtParams *c)
{
	if (!m || b.fr.kind != FR_UNPOLARIZED) return false;
	const float ior = (float)m[8];
	if (!(b.fr.a[0] == ior && b.fr.a[1] == ior && b.fr.a[2] == ior) || !(ior > 1.0f && ior < 1e4f)) return false;
	if (!(m[6] >= 0.0 && m[6] < 1e12) || !(m[7] >= 1e-3 && m[7] <= 16.0)) return false;
	*c = CtParams{};
	const float inv_pi = 1.0f / (float)DJB_PI;               // divs(kD, float(pi)) = (1.0f / float(pi)) * kD, dj_brdf.h:601, 3644
	for (int k = 0; k < 3; ++k) {
		const float kd = (float)m[k], a = (float)m[3 + k];
		if (!(kd >= 0.0f && kd < 1e6f) || !(a > 1e-12f && a < 1e12f)) return false;
		c->kd_pi[k] = inv_pi * kd; c->A[k] = a;
	}
	c->ior = ior; c->B = m[6]; c->C = m[7];
	return true;
}

// sgd: the published rows and anything like them (constructor's fresnel::sgd(f0, f1), positive alpha / kap / lambda / c, k >= 1,
// a diffuse term that is not negligible: where e^-ax underflows tier 1 returns Kd / pi, which needs Kd to dominate).
// Per channel the error bound of g1 (header of ct_eval_sgd) is turned into two thresholds on x = c t1^k:
//   t_neg  the t1 below which w = lambda expm1(x) < 1e-8 whatever the relative error of x (as long as it is below 1);
//   R      the bound of dx / x for t1 in [t_neg, pi/2 - theta0];
//   x_max  the largest x <= 0.5 with lambda e^x x R <= 1.5e-6 (1 - w);   x_zero: the smallest x with w - 1 >= 4 lambda e^x x R + 1e-6.
constexpr double SGD_KD_MIN = 1e-4;
bool ct_params_sgd(const Brdf &b, const double *m, CtParams *c)
{
	if (!m || b.fr.kind != FR_SGD) return false;
	*c = CtParams{};
	for (int k = 0; k < 3; ++k) {
		const double rd = m[k], rs = m[3 + k], al = m[6 + k], pp = m[9 + k], f0 = m[12 + k], f1 = m[15 + k], kap = m[18 + k],
		             lam = m[21 + k], cc = m[24 + k], kk = m[27 + k], th0 = m[30 + k];
		if ((float)f0 != b.fr.a[k] || (float)f1 != b.fr.b[k]) return false;          // a replaced Fresnel term: exact kernels
		if (!(rd >= SGD_KD_MIN && rd < 1e6) || !(rs >= 0.0 && rs < 1e6)) return false;
		if (!(al >= 1e-6 && al <= 10.0) || !(pp >= 0.0 && pp <= 4.0) || !(kap > 1e-6 && kap < 1e8)) return false;
		if (!(lam > 1e-12 && lam < 1e9) || !(cc > 1e-30 && cc < 1e39) || !(kk >= 2.0 && kk <= 2000.0) || !(th0 > -1.5 && th0 < 1.5)) return false;
		if (!(std::fabs(f0) < 1e3) || !(std::fabs(f1) < 1e3)) return false;
		c->kd[k] = (float)rd; c->ks[k] = (float)rs;
		c->sf0[k] = (float)f0; c->sf1[k] = (float)f1; c->s1mf0[k] = 1.0f - (float)f0;
		c->alpha[k] = al; c->inv_alpha[k] = 1.0 / al;
		c->p_[k] = (float)pp; c->lkap[k] = (float)std::log2(kap / DJB_PI);
		c->lam[k] = (float)lam; c->l2c[k] = (float)std::log2(cc); c->kk[k] = (float)kk;
		c->th0_hi[k] = (float)th0; c->th0_lo[k] = (float)(th0 - (double)c->th0_hi[k]);
		// the corner t1 = 0 sits inside the acos error: up to t1 = 2 SGD_DT the reference may be on either side; nothing may happen there
		const double l2lam = std::log2(lam), l2c = std::log2(cc);
		if (!(l2lam + l2c + kk * std::log2(2.0 * SGD_DT) + std::log2(1.0 + kk) < std::log2(3e-8))) return false;
		const double t_hi = DJB_PI * 0.5 - th0 + 1e-3;
		c->x_max[k] = 0.5f; c->x_zero[k] = 3.0e38f;
		if (t_hi <= 4.0 * SGD_DT) continue;                                           // theta_k never exceeds theta0: all plateau
		double t_neg = std::exp2((std::log2(1e-8) - l2lam - l2c) / kk);               // lambda c t^k = 1e-8
		t_neg = std::min(std::max(t_neg, 4.0 * SGD_DT), t_hi);
		const double Lmax = std::max(std::fabs(std::log(t_neg)), std::fabs(std::log(t_hi)));
		const double R = kk * (SGD_DT / t_neg + 1.2e-7 * Lmax) + 4e-7;
		if (!(R < 0.25)) return false;
		auto w_of = [&](double x) { return lam * std::expm1(x); };
		auto good = [&](double x) { const double w = w_of(x); return w < 1.0 && (w + lam) * x * R <= 1.5e-6 * (1.0 - w); };
		double lo = 0.0, hi = 0.5;
		if (good(hi)) lo = hi;
		else for (int it = 0; it < 60; ++it) { const double mid = 0.5 * (lo + hi); if (good(mid)) lo = mid; else hi = mid; }
		c->x_max[k] = (float)(lo * (1.0 - 1e-6));
		auto dead = [&](double x) { const double w = w_of(x); return w - 1.0 >= 4.0 * (w + lam) * x * R + 1e-6; };
		lo = 0.0; hi = 100.0;
		if (dead(hi)) { for (int it = 0; it < 80; ++it) { const double mid = 0.5 * (lo + hi); if (dead(mid)) hi = mid; else lo = mid; } c->x_zero[k] = (float)(hi * (1.0 + 1e-6)); }
	}
	return true;
}

bool ct_params_any(const Brdf &b, const Params &p, const double *model_host, CtParams *c);

bool ct_params(const Brdf &b, const Params &p, CtParams *c)
{
	if (p.tx != 0.0f || p.ty != 0.0f || !(p.nx == 0.0f && p.ny == 0.0f && p.nz == 1.0f)) return false;
	if (!(fabsf(p.rho) <= djbk::CT_RHO_MAX) || !(p.ax >= 1e-4f && p.ax <= 1e4f) || !(p.ay >= 1e-4f && p.ay <= 1e4f)) return false;
	const double t2 = (double)p.ax * (double)p.ay * (double)p.s;
	if (!(t2 > 1e-9 && t2 < 1e9)) return false;
	c->ax = p.ax; c->ay = p.ay; c->rho = p.rho; c->s = p.s; c->rho_ay = p.rho * p.ay;
	c->r_ax = (float)(1.0 / (double)p.ax);
	c->r_t2 = (float)(1.0 / (double)(p.ax * p.ay * p.s));
	c->k_d = (float)((1.0 / (double)(p.ax * p.ay * p.s)) / DJB_PI);
	c->t2 = p.ax * p.ay * p.s;
	c->R_ax = p.r_ax; c->R_t2 = p.r_t2;
	c->shadow = b.shadow;
	for (int k = 0; k < 3; ++k) { c->f0[k] = 1.0f; c->f1[k] = 0.0f; c->n2m1[k] = 1.25f; }
	if (b.fr.kind == FR_UNPOLARIZED) {
		for (int k = 0; k < 3; ++k) {
			if (!(b.fr.a[k] >= CT_IOR_MIN && b.fr.a[k] <= CT_IOR_MAX)) return false;      // ct_unpolarized: the reference's own noise below 1.05
			c->n2m1[k] = (float)((double)b.fr.a[k] * (double)b.fr.a[k] - 1.0);
		}
	} else if (b.fr.kind == FR_SCHLICK) {
		for (int k = 0; k < 3; ++k) {
			// below f0 = 0.01 the term f0 + (1 - f0)(1 - c)^5 is ill-conditioned in c near 1 (header): exact kernels
			if (!(b.fr.a[k] >= 0.01f && b.fr.a[k] <= 1.0f)) return false;
			c->f0[k] = b.fr.a[k]; c->f1[k] = 1.0f - b.fr.a[k];
		}
	} else if (b.fr.kind != FR_IDEAL) return false;
	return true;
}

bool ct_params_any(const Brdf &b, const Params &p, const double *model_host, CtParams *c)
{
	if (b.kind == KIND_ABC) return ct_params_abc(b, model_host, c);
	if (b.kind == KIND_SGD) return ct_params_sgd(b, model_host, c);
	return (b.kind == KIND_GGX || b.kind == KIND_BECKMANN) && ct_params(b, p, c);
}

template <int KIND, int FRK>
hipError_t launch_ct(hipStream_t s, const Brdf &b, const Params &p, const CtParams &c, long long n, const View &i, const View &o,
                     const View &out, float *out_pdf, int want, uint4 *list, unsigned int cap /* records in total */, unsigned int *count /* CT_SHARDS words */)
{
	hipError_t e = hipMemsetAsync(count, 0, sizeof(unsigned int) * CT_SHARDS * CT_STRIDE, s);
	if (e != hipSuccess) return e;
	cap /= CT_SHARDS;                                          // records per shard
	if (cap == 0) cap = 1;
	const long long n4 = n / 4;
	// one workgroup per 1024 pairs, no grid-stride cap: measured (profiles/r03/contract_grid.txt, 1e8 pairs) 0.696 ms with
	// the full grid against 0.77-0.82 ms with 2048 ... 32768 persistent workgroups -- the hardware dispatcher keeps more
	// loads in flight across workgroup boundaries than a wave's in-order loop does
	const dim3 g(grid_for(n4, 0x7fffffffLL)), t(BLOCK), gf(grid_for((long long)CT_SHARDS * cap, 2048));
#define DJB_CT(W_) do { \
		if (n4 > 0) hipLaunchKernelGGL((k_ct_fast_v4<KIND, W_, FRK>), g, t, 0, s, c, n4, i, o, out, out_pdf, list, cap, count); \
		hipLaunchKernelGGL((k_ct_fixup<KIND, W_, FRK>), gf, t, 0, s, b, p, n, i, o, out, out_pdf, list, cap, count); } while (0)
	switch (want) {
	case 1: DJB_CT(1); break;
	case 2: DJB_CT(2); break;
	case 4: DJB_CT(4); break;
	case 5: DJB_CT(5); break;
	case 6: DJB_CT(6); break;
	default: return hipErrorInvalidValue;
	}
#undef DJB_CT
	return hipGetLastError();
}

} // namespace

namespace djbk {

bool contract_supported(const Brdf &b, const Params &p, const double *model_host)
{
	CtParams c;
	return ct_params_any(b, p, model_host, &c);
}

bool contract_params(const Brdf &b, const Params &p, const double *model_host, CtParams *c) { return ct_params_any(b, p, model_host, c); }

hipError_t launch_eval_contract(hipStream_t s, const Brdf &b, const Params &p, const double *model_host, long long n, const View &i, const View &o,
                                const View &out, float *out_pdf, int want, unsigned int *list, unsigned int cap, unsigned int *count)
{
	CtParams c;
	if (!ct_params_any(b, p, model_host, &c)) return hipErrorInvalidValue;
	if (n <= 0) return hipSuccess;
	// the < 4-pair tail of the batch: the exact kernel (launched first: the fix-up kernel's overflow rescan covers it too)
	const long long n4 = n / 4;
	if (4 * n4 < n) {
		auto off = [&](const View &v) { return View{ v.x ? v.x + 4 * n4 : nullptr, v.y ? v.y + 4 * n4 : nullptr, v.z ? v.z + 4 * n4 : nullptr, v.stride }; };
		hipError_t e = launch_eval(s, b, p, n - 4 * n4, off(i), off(o), off(out), out_pdf ? out_pdf + 4 * n4 : nullptr, want);
		if (e != hipSuccess) return e;
	}
	if (b.kind == KIND_ABC) return launch_ct<KIND_ABC, -1>(s, b, p, c, n, i, o, out, out_pdf, want, (uint4 *)list, cap, count);
	if (b.kind == KIND_SGD) return launch_ct<KIND_SGD, -1>(s, b, p, c, n, i, o, out, out_pdf, want, (uint4 *)list, cap, count);
	// the pdf has no Fresnel term: one instantiation serves every kind
	const int frk = want == 4 ? FR_IDEAL : b.fr.kind;
	if (b.kind == KIND_BECKMANN) {
		if (frk == FR_SCHLICK) return launch_ct<KIND_BECKMANN, FR_SCHLICK>(s, b, p, c, n, i, o, out, out_pdf, want, (uint4 *)list, cap, count);
		if (frk == FR_UNPOLARIZED) return launch_ct<KIND_BECKMANN, FR_UNPOLARIZED>(s, b, p, c, n, i, o, out, out_pdf, want, (uint4 *)list, cap, count);
		return launch_ct<KIND_BECKMANN, FR_IDEAL>(s, b, p, c, n, i, o, out, out_pdf, want, (uint4 *)list, cap, count);
	}
	if (frk == FR_SCHLICK) return launch_ct<KIND_GGX, FR_SCHLICK>(s, b, p, c, n, i, o, out, out_pdf, want, (uint4 *)list, cap, count);
	if (frk == FR_UNPOLARIZED) return launch_ct<KIND_GGX, FR_UNPOLARIZED>(s, b, p, c, n, i, o, out, out_pdf, want, (uint4 *)list, cap, count);
	return launch_ct<KIND_GGX, FR_IDEAL>(s, b, p, c, n, i, o, out, out_pdf, want, (uint4 *)list, cap, count);
}

hipError_t launch_contract_selftest(hipStream_t s, const Brdf &b, const Params &p, const double *model_host, long long n, uint32_t seed_i, uint32_t seed_o,
                                    unsigned long long start, int family, unsigned int *max_bits, unsigned long long *counters)
{
	CtParams c;
	if (!ct_params_any(b, p, model_host, &c)) return hipErrorInvalidValue;
	const dim3 g(grid_for(n, 256LL * 16)), t(BLOCK);
	if (b.kind == KIND_ABC) hipLaunchKernelGGL((k_ct_selftest<KIND_ABC, -1>), g, t, 0, s, b, p, c, n, seed_i, seed_o, start, family, max_bits, counters);
	else if (b.kind == KIND_SGD) hipLaunchKernelGGL((k_ct_selftest<KIND_SGD, -1>), g, t, 0, s, b, p, c, n, seed_i, seed_o, start, family, max_bits, counters);
	else if (b.kind == KIND_BECKMANN) {
		if (b.fr.kind == FR_SCHLICK) hipLaunchKernelGGL((k_ct_selftest<KIND_BECKMANN, FR_SCHLICK>), g, t, 0, s, b, p, c, n, seed_i, seed_o, start, family, max_bits, counters);
		else if (b.fr.kind == FR_UNPOLARIZED) hipLaunchKernelGGL((k_ct_selftest<KIND_BECKMANN, FR_UNPOLARIZED>), g, t, 0, s, b, p, c, n, seed_i, seed_o, start, family, max_bits, counters);
		else hipLaunchKernelGGL((k_ct_selftest<KIND_BECKMANN, FR_IDEAL>), g, t, 0, s, b, p, c, n, seed_i, seed_o, start, family, max_bits, counters);
	} else if (b.fr.kind == FR_UNPOLARIZED) hipLaunchKernelGGL((k_ct_selftest<KIND_GGX, FR_UNPOLARIZED>), g, t, 0, s, b, p, c, n, seed_i, seed_o, start, family, max_bits, counters);
	else if (b.fr.kind == FR_SCHLICK) hipLaunchKernelGGL((k_ct_selftest<KIND_GGX, FR_SCHLICK>), g, t, 0, s, b, p, c, n, seed_i, seed_o, start, family, max_bits, counters);
	else hipLaunchKernelGGL((k_ct_selftest<KIND_GGX, FR_IDEAL>), g, t, 0, s, b, p, c, n, seed_i, seed_o, start, family, max_bits, counters);
	return hipGetLastError();
}

} // namespace djbk
