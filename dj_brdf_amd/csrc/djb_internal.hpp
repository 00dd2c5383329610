// djb_internal.hpp -- launch interface between the C-ABI host code (djb_host.cpp) and the
// gfx950 kernels (djb_kernels_*.hip).  Not installed; the public surface is include/djb_hip.h.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "djb_device.hpp"

namespace djbk {

using djbdev::Brdf;
using djbdev::Params;
using djbdev::View;

// WANT bits: 1 eval, 2 evalp, 4 pdf (1 and 2 are exclusive)
hipError_t launch_eval(hipStream_t s, const Brdf &b, const Params &p, long long n,
                       const View &i, const View &o, const View &out_fr, float *out_pdf, int want);

// sample / evalp_is.  If u1 == nullptr the uniforms come from the on-chip counter RNG
// (seed_u1, seed_u2, start); out_w / out_pdf may be null-views (sample only).
// contract (DJB_OPT_CONTRACT_1E5): Beckmann and GGX `sample` may return directions within 1e-5 per component instead of the reference's bits
hipError_t launch_sample(hipStream_t s, const Brdf &b, const Params &p, long long n,
                         const float *u1, const float *u2, uint32_t seed_u1, uint32_t seed_u2,
                         unsigned long long start, const View &o, const View &out_i,
                         const View *out_w, float *out_pdf, bool contract = false);

// the Beckmann lobe's sample / evalp_is (djb_kernels_sample.hip: common path + deferred full path); launch_sample forwards here
hipError_t launch_sample_beckmann(hipStream_t s, const Brdf &b, const Params &p, long long n,
                                  const float *u1, const float *u2, uint32_t seed_u1, uint32_t seed_u2,
                                  unsigned long long start, const View &o, const View &out_i,
                                  const View *out_w, float *out_pdf, bool contract = false);
// ggx `sample` under DJB_OPT_CONTRACT_1E5: directions within 1e-5 per component (djb_kernels_sample.hip, ggx_sample_contract)
hipError_t launch_sample_ggx_contract(hipStream_t s, const Brdf &b, const Params &p, long long n, const float *u1, const float *u2,
                                      uint32_t seed_u1, uint32_t seed_u2, unsigned long long start, const View &o, const View &out_i);
bool sample_contract_supported(const Brdf &b, const Params &p);
// the contract-mode sampler against the full per-sample code on n generated samples (k_sample_ct_selftest)
hipError_t launch_sample_contract_selftest(hipStream_t s, const Brdf &b, const Params &p, long long n, uint32_t seed, unsigned long long start,
                                           int family, unsigned int *max_bits, unsigned long long *counters);
// directed search over the bit patterns of (u1, u2, o) for the largest contract-vs-exact difference (k_sample_ct_attack); u1, u2, o are updated in place
hipError_t launch_sample_contract_attack(hipStream_t s, const Brdf &b, const Params &p, long long n, float *u1, float *u2, const View &o, int iters,
                                         uint32_t seed, float *best, unsigned long long *counters);

// per-pair params: rec = n x 5 floats; mode 0 = pdfparams records, mode 1 = LEAN texel moments composed with
// base5 = params_to_lrep(base) (unscaled), scale = dmapscale, lean_flags = DJB_LEAN_* as dj_beckmannconductor does;
// out_pp (optional, mode 1) receives the resolved pdfparams
hipError_t launch_eval_pp(hipStream_t s, const Brdf &b, long long n, const View &i, const View &o,
                          const float *rec, int mode, const float *base5, float scale, int lean_flags, const View &out,
                          float *out_pdf, float *out_pp, int want);

// sample (out_w == NULL) / evalp_is with the same per-pair records
hipError_t launch_sample_pp(hipStream_t s, const Brdf &b, long long n, const float *u1, const float *u2, const View &o,
                            const float *rec, int mode, const float *base5, float scale, int lean_flags, const View &out_i,
                            const View *out_w, float *out_pdf, float *out_pp);

// microfacet / radial queries; out.x holds scalar results (out.xyz for the Fresnel query)
hipError_t launch_query(hipStream_t s, const Brdf &b, const Params &p, int which, long long n,
                        const View &a, const View &bb, const View &c, const View &out);

hipError_t launch_io_to_hd(hipStream_t s, long long n, const View &i, const View &o,
                           const View &h, const View &d, bool inverse);
hipError_t launch_merl_index(hipStream_t s, long long n, const View &i, const View &o, int32_t *idx);
hipError_t launch_merl_keys(hipStream_t s, long long n, const View &i, const View &o, uint32_t *keys);     // tier-1 bin keys (djb_kernels_merl.hip)

// two-tier exact MERL lookup (djb_kernels_merl.hip): one kernel, the ambiguous pairs of tier 1 wait in per-wave LDS queues and are
// drained by the exact path as dense waves (no worklist in HBM, no second launch)
hipError_t launch_merl_twotier(hipStream_t s, const Brdf &b, long long n, const View &i, const View &o,
                               const View &out, float *out_pdf, int want);
hipError_t launch_merl_guard_stats(hipStream_t s, long long n, const View &i, const View &o,
                                   const float *guard6, unsigned int *max_bits, unsigned long long *counters);

// directed search for the worst |estimate - reference| / guard band: i / o are candidates in device memory, updated in place;
// best[n] receives each candidate's final ratio; counters = {evaluations, index mismatches among certain pairs, accepted moves}
hipError_t launch_merl_guard_attack(hipStream_t s, long long n, const View &i, const View &o, const float *guard6, int iters, uint32_t seed,
                                    float *best, unsigned long long *counters);

// MERL payload (3*n doubles) -> packed RGB texel table (pre-scaled, below-horizon zeroed); n = 1458000
hipError_t launch_merl_convert(hipStream_t s, const double *samples, long long n, djbdev::MerlTexel *table);
// UTIA payload (n = 3*288*288 doubles) -> float(max(0, s) * double(1.f/140.f)) in 288*288 records of 8 float4
hipError_t launch_utia_convert(hipStream_t s, const double *samples, long long n, float4 *table);

hipError_t launch_gen_directions(hipStream_t s, long long n, uint32_t seed, unsigned long long start,
                                 const View &out);
hipError_t launch_gen_uniforms(hipStream_t s, long long n, uint32_t seed, unsigned long long start,
                               float *out);
// utia eval / evalp (want 1, 2, 5, 6), two-tier: k_utia_v2 (no exact fall-backs; lists the undecided pairs) +
// k_eval_utia_fix for the pairs of the worklist (count: 16 bytes, list: cap entries).  contract: DJB_OPT_CONTRACT_1E5 -- the
// sRGB power of the decode on the fast transcendentals, everything else (cells, weights, the 16-tap sums) the reference's bits
hipError_t launch_utia_twotier(hipStream_t s, const Brdf &b, long long n, const View &i, const View &o, const View &out,
                               float *out_pdf, int want, unsigned int *list, unsigned int cap, unsigned int *count, bool contract);
hipError_t launch_fast_trig_selftest(hipStream_t s, long long n, int mode, uint32_t first, uint32_t seed, unsigned long long *counters4);
// DJB_OPT_CONTRACT_1E5 (djb_kernels_contract.hip): GGX eval / evalp / pdf inside the 1e-5 value contract, two-tier like
// the MERL lookup, with a sharded worklist (list: cap records of 32 bytes in total, count: CONTRACT_SHARDS uint32, CONTRACT_COUNTER_STRIDE words apart: the record
// list is cut into that many equal segments).  Views must be dense (stride 1) and 16-byte aligned.
constexpr unsigned int WL_SHARDS = 64;
constexpr unsigned int WL_COUNTER_STRIDE = 32;         // words between two counters (one 128-byte line each)
constexpr unsigned int CONTRACT_SHARDS = WL_SHARDS, CONTRACT_COUNTER_STRIDE = WL_COUNTER_STRIDE;
constexpr float CT_RHO_MAX = 0.9f;
// launch-uniform constants of the contract-mode fast paths (filled by contract_params on the host, a kernel argument)
struct CtParams {
	float ax, ay, rho, s, rho_ay;      // microfacet::params (tx = ty = 0, mean normal = +z)
	float r_ax, r_t2;                  // float(1 / ax), float(1 / (ax ay s))
	float k_d;                         // float(r_t2 / pi): the constant factor of D
	float t2;                          // ax * ay * s as the reference rounds it (the divisor of mf_p22)
	double R_ax, R_t2;                 // 1 / ax and 1 / t2 to within 2^-52: fdiv_r's exact divisions (Beckmann)
	float f0[3], f1[3];                // schlick: f0 and 1 - f0
	float n2m1[3];                     // unpolarized: ior^2 - 1 per channel (ior >= CT_IOR_MIN)
	int shadow;
	// abc (model row kD[3] A[3] B C ior, dj_brdf.h:3608-3668)
	float kd_pi[3];                    // kD / pi as the reference rounds it: float(kD) * (1.0f / float(pi))
	float A[3], ior;
	double B, C;
	// sgd (model row rhoD rhoS alpha p f0 f1 kap lambda c k theta0, 3 doubles each; dj_brdf.h:3415-3500)
	float kd[3], ks[3], sf0[3], sf1[3], s1mf0[3];      // rhoD, rhoS, Fresnel f0, f1, 1 - f0 as floats (the reference's vec3::from_raw)
	double alpha[3], inv_alpha[3];
	float p_[3], lkap[3];                              // NDF exponent; log2(kap / pi)
	float lam[3], l2c[3], kk[3], th0_hi[3], th0_lo[3]; // g1: lambda, log2(c), k, theta0 = hi + lo
	float x_max[3], x_zero[3];                         // g1: tier-1 range of x = c t1^k (ct_params_sgd)
};
// false: brdf / params outside the fast path's domain
bool contract_params(const Brdf &b, const Params &p, const double *model_host, CtParams *c);
// model_host: the host copy of b.model (sgd / abc rows), or NULL
bool contract_supported(const Brdf &b, const Params &p, const double *model_host = nullptr);
hipError_t launch_eval_contract(hipStream_t s, const Brdf &b, const Params &p, const double *model_host, long long n, const View &i, const View &o,
                                const View &out, float *out_pdf, int want, unsigned int *list, unsigned int cap, unsigned int *count);
hipError_t launch_contract_selftest(hipStream_t s, const Brdf &b, const Params &p, const double *model_host, long long n, uint32_t seed_i, uint32_t seed_o,
                                    unsigned long long start, int family, unsigned int *max_bits, unsigned long long *counters);
hipError_t launch_guard_selftest(hipStream_t s, long long n, uint32_t seed, unsigned long long *counters);
hipError_t launch_model_fast_selftest(hipStream_t s, const Brdf &b, long long n, uint32_t seed, uint32_t first, unsigned long long *counters6);
hipError_t launch_libm_probe(hipStream_t s, int fn, long long n, const double *x, const double *y, double *out);
hipError_t launch_trig_sweep(hipStream_t s, int fn, uint32_t first, long long n, void *out);
hipError_t launch_histogram_xy(hipStream_t s, long long n, const View &v, int bins,
                               unsigned long long *counts);

// ---- the power-iteration fitter (djb::tabular ctor + fits), one workgroup per material
struct FitOut {           // device pointers, [n_mat][res] (fresnel [n_mat][res][3]); alphas [n_mat]
	float *p22, *sigma, *cdf, *qf, *fresnel;
	float *alpha_beckmann, *alpha_ggx;
	int *n_qf;            // number of valid qf entries per material (reference quirk, dj_brdf.h:2731)
};
// The two heavy passes of one material's fit -- the sigma quadrature (by rows) and the Fresnel-ratio pass (by
// (theta_d, theta_h) pairs) -- can be sliced over `parts` workgroups (fit_parts(n_mat)): parts - 1 helper
// workgroups per material redo the cheap phases before them and exchange their slices through sig_x [n_mat][res],
// the ratio scratch and the arrival counters sig_done [n_mat][2] (zeroed by launch_fit).
// fres_dirs (optional): one record of 5 floats per (theta_d, theta_h) pair of the Fresnel-ratio pass -- the outgoing direction (x = NaN
// for the pairs the reference skips) and the fitted lobe's two table coordinates that the pair's geometry fixes -- plus one trailing float
// (launch_fit_fresnel_dirs; fit_fresnel_dirs_floats(res) floats).  They depend on the resolution only, so the host computes them once per
// context and resolution instead of every workgroup of every fit recomputing 5 456 double rotations and 11 k fp64 libm calls; NULL:
// computed in place.
struct FitSplit { int parts; float *sig_x; unsigned int *sig_done; const float *fres_dirs; };
size_t fit_fresnel_dirs_floats(int res);
hipError_t launch_fit_fresnel_dirs(hipStream_t s, int res, const Params &std_p, float *recs);
int fit_parts(int n_mat, int n_cus);
// srcs: device array of n_mat Brdf views (all of kind `src_kind`); std_p: params::standard().
// km_scratch: n_mat*parts*(res-1)^2 doubles; ratio_scratch: n_mat*(res-1)*res*3 floats.
hipError_t launch_fit(hipStream_t s, const Brdf *srcs, int src_kind, const Params &std_p, int n_mat,
                      int res, int shadow, double *km_scratch, float *ratio_scratch,
                      const FitOut &out, const FitSplit &split);
size_t fit_lds_bytes(int res);
// query slots of a tabular(merl, res) fit and the MERL table index each one reads (-1: unused slot); idx: device, fit_merl_slots(res) ints
int fit_merl_slots(int res);
hipError_t launch_fit_merl_slots(hipStream_t s, int res, int32_t *idx);

// ---- tabular_anisotropic (djb_kernels_fit_aniso.hip): all pointers are device memory
struct AnisoScratch {
	int elev, azim;
	// outputs: grids are elev x azim, element (i_elev, j_azim) at [i + elev*j]
	float *p22, *sigma, *pdf1, *cdf1, *qf1, *pdf2, *cdf2, *qf2, *fres /* 3*elev */, *fit /* 10 */;
	int *counts;                               // [0] entries in qf1, [1] short qf2 rows, [2] entries in the reference's m_qf2
	// work arrays: N = (elev-1)*azim
	float *k1, *xo, *yo, *zo, *s1, *s2, *tn, *dn;   // N each
	double *v0, *v1;                                // N each
	float *terms;                                   // aniso_terms_count()
	float *ndf_tab;                                 // aniso_ndf_count()
	double *cosd;                                   // aniso_cosd_count(azim)
	float *sig_theta, *sig_sin; double *sig_cosd;   // aniso_sig_nodes() each
	float *ratio;                                   // 3*(elev-1)*elev
	float *probes;                                  // azim*8*(elev-1)
	float *rowk;                                    // azim
	float *qf2_rows; int *qf2_len;                  // elev*azim / azim: the rows of compute_qf2 before they are laid out
	int qf2_aligned;                                // 0: the reference's push_back layout; 1: every row at elev*k (DJB_OPT_ANISO_QF2_ALIGNED)
};
size_t aniso_terms_count();
size_t aniso_ndf_count();
size_t aniso_cosd_count(int azim);
size_t aniso_sig_nodes();
hipError_t launch_fit_aniso(hipStream_t s, const Brdf &src, const Params &std_p, const AnisoScratch &S, int shadow);

} // namespace djbk
