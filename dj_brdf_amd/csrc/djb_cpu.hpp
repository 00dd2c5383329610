// djb_cpu.hpp -- interface between the C ABI (djb_host.hip) and the product's HOST execution path
// (djb_cpu.cpp): the same per-unit code as the gfx950 kernels (djb_device.hpp), instantiated for the CPU with
// the host's own libm -- which is exactly what the reference (jdupuy/dj_brdf, a CPU library, dj_brdf.h:74-109)
// calls.  Two uses:
//   * a CPU context (djb_ctx_create(DJB_DEVICE_CPU)): every entry point of include/djb_hip.h on a machine
//     without a GPU -- BASELINE.json configs[0], "examples/merl_params.cpp ... CPU only, runs without a GPU";
//   * scalar-size DJB_MEM_HOST calls on a GPU context (the facade's one-pair virtuals, Mitsuba's per-hit calls):
//     n <= DJB_SCALAR_HOST_MAX units are evaluated here, on the caller's thread, with no staging, no launch and no
//     context mutex, from a host twin of the object's tables.
// Not a fallback for batches: a GPU context runs every batch above the scalar threshold on the GPU or fails.
// Nothing under oracle/ is used: that is test infrastructure.
#pragma once

#include <stdint.h>

#include "../../include/djb_hip.h"

// the host path's libm check (djb_cpu_libm.cpp): init() runs once, by the first context of either kind
namespace djbhostlibm {
extern int use_restated;
int init();
int atan_log_kat();
}

namespace djbcpu {

// handles: a CPU context / object starts with the same leading int as its GPU counterpart (djb_ctx: the device,
// djb_brdf: the device of the context that created it), which is how the C ABI tells them apart (device < 0).  The
// object carries its own copy: a handle may outlive its context (garbage collectors destroy them in any order).
inline bool is_cpu(const djb_ctx *c) { return c && *(const int *)c < 0; }
inline bool is_cpu(const djb_brdf *b) { return b && *(const int *)b < 0; }

// ---- context
djb_status ctx_create(djb_ctx **out);
djb_status ctx_destroy(djb_ctx *ctx);
djb_status timer_start(djb_ctx *ctx);
djb_status timer_stop_ms(djb_ctx *ctx, float *ms);
djb_ctx *twin_ctx();                 // process-wide CPU context that owns the host twins of GPU objects

// ---- constructors (same contracts as the djb_brdf_create_* entry points)
djb_status create_microfacet(djb_ctx *, int kind, const djb_fresnel_desc *, int shadow, djb_brdf **);
djb_status create_merl_from_memory(djb_ctx *, const double *samples, int64_t n, djb_brdf **);
djb_status create_merl_from_file(djb_ctx *, const char *path, djb_brdf **);
// a MERL object from an already converted texel table (n = 1458000 x 3 floats; copied)
djb_status create_merl_from_texels(djb_ctx *, const float *texels3, djb_brdf **);
djb_status create_utia_from_memory(djb_ctx *, const double *samples, djb_brdf **);
// a UTIA object from the converted table (288*288 records of 8 float4 = 32 floats; copied)
djb_status create_utia_from_records(djb_ctx *, const float *records, djb_brdf **);
djb_status create_utia_from_file(djb_ctx *, const char *path, djb_brdf **);
djb_status create_lambert(djb_ctx *, djb_brdf **);
djb_status create_model(djb_ctx *, int kind, const double *row, int count, djb_brdf **);
djb_status create_user_microfacet(djb_ctx *, const djb_user_ndf *, const djb_fresnel_desc *, int shadow, djb_brdf **);
djb_status create_tabular(djb_ctx *, const djb_brdf *src, int res, int shadow, djb_brdf **);
// a tabular / tabular_anisotropic object from tables fitted elsewhere (the host twin of a GPU-fitted object)
djb_status create_tabular_from_tables(djb_ctx *, int shadow, int res, const float *p22, const float *sigma, const float *cdf,
                                      const float *qf, int n_qf, const float *fresnel3, float alpha_b, float alpha_g, djb_brdf **);
djb_status create_aniso_from_tables(djb_ctx *, int shadow, int elev, int azim, const float *const tabs[8], const int counts[8],
                                    const float *fresnel3, const float fit10[10], int qf2_entries, djb_brdf **);
djb_status create_tabular_anisotropic(djb_ctx *, const djb_brdf *src, int elev, int azim, int shadow, djb_brdf **);
// fits of user-defined sources: the query slots of a fit (host code, any context) and the fit from per-slot samples
int fit_query_count(int res);
int fit_aniso_query_count(int elev, int azim);
djb_status fit_query_dirs(int res, float *i3, float *o3);
djb_status fit_aniso_query_dirs(int elev, int azim, float *i3, float *o3);
djb_status create_tabular_from_samples(djb_ctx *, int res, int shadow, const float *rgb, djb_brdf **);
djb_status create_tabular_anisotropic_from_samples(djb_ctx *, int elev, int azim, int shadow, const float *rgb, djb_brdf **);
djb_status destroy(djb_brdf *);
int kind(const djb_brdf *);
int get_shadow(const djb_brdf *);
djb_status set_shadow(djb_brdf *, int shadow);
djb_status set_fresnel(djb_brdf *, const djb_fresnel_desc *);
djb_status get_fresnel(const djb_brdf *, djb_fresnel_desc *out);
djb_status get_samples(const djb_brdf *, double *out, int64_t capacity, int64_t *count);
djb_status tabular_get(const djb_brdf *, int which, float *out, int *count);
djb_status tabular_fit(const djb_brdf *, float *alpha_beckmann, float *alpha_ggx);
djb_status aniso_get(const djb_brdf *, int which, float *out, int *count, int *elev, int *azim);
djb_status aniso_fit(const djb_brdf *, djb_params *beckmann, djb_params *ggx);

// ---- batch operators on host arrays (want bits: 1 eval, 2 evalp, 4 pdf)
djb_status eval(djb_ctx *, const djb_brdf *, int64_t n, const djb_vec3_view *i, const djb_vec3_view *o, const djb_params *,
                const djb_vec3_view *out_fr, float *out_pdf, int want);
// u1 == NULL: uniforms from the counter RNG (seed_u1, seed_u2, start); out_w == NULL: sample(), else evalp_is()
djb_status sample(djb_ctx *, const djb_brdf *, int64_t n, const float *u1, const float *u2, uint32_t seed_u1, uint32_t seed_u2,
                  uint64_t start, const djb_vec3_view *o, const djb_params *, const djb_vec3_view *out_w,
                  const djb_vec3_view *out_i, float *out_pdf);
djb_status eval_pp(djb_ctx *, const djb_brdf *, int64_t n, const djb_vec3_view *i, const djb_vec3_view *o, const float *rec,
                   int mode, const float *base5, float scale, int lean_flags, int want, const djb_vec3_view *out_fr, float *out_pdf,
                   float *out_pp);
djb_status sample_pp(djb_ctx *, const djb_brdf *, int64_t n, const float *u1, const float *u2, const djb_vec3_view *o, const float *rec,
                     int mode, const float *base5, float scale, int lean_flags, const djb_vec3_view *out_w, const djb_vec3_view *out_i,
                     float *out_pdf, float *out_pp);
djb_status query(djb_ctx *, const djb_brdf *, int which, int64_t n, const djb_vec3_view *a, const djb_vec3_view *b,
                 const djb_vec3_view *c, const djb_params *, const djb_vec3_view *out);
djb_status io_hd(djb_ctx *, int64_t n, const djb_vec3_view *a, const djb_vec3_view *b, const djb_vec3_view *c,
                 const djb_vec3_view *d, bool inverse);
djb_status merl_index(djb_ctx *, int64_t n, const djb_vec3_view *i, const djb_vec3_view *o, int32_t *out);
djb_status fit_merl_batch(djb_ctx *, int n_mat, const double *const *tables, int res, int shadow, float *ab, float *ag,
                          float *p22, float *sigma, float *cdf, float *qf, float *fresnel);
djb_status fit_brdf_batch(djb_ctx *, int n_mat, const djb_brdf *const *srcs, int res, int shadow, float *ab, float *ag,
                          float *p22, float *sigma, float *cdf, float *qf, float *fresnel);
djb_status fit_merl_files(djb_ctx *, int n_files, const char *const *paths, int res, int shadow, int threads, float *ab,
                          float *ag, double *timing);
djb_status gen_directions(djb_ctx *, int64_t n, uint32_t seed, uint64_t start, const djb_vec3_view *out);
djb_status gen_uniforms(djb_ctx *, int64_t n, uint32_t seed, uint64_t start, float *out);
djb_status helper(int which, const float *in, float *out);    // djb_helper: the reference's file-static helpers, host arithmetic
djb_status histogram_xy(djb_ctx *, int64_t n, const djb_vec3_view *v, int bins, unsigned long long *counts);
// host values of trig site `host_fn` (djb_device.hpp TRIG_*; float sites < TRIG_DOUBLE, double sites from TRIG_DOUBLE)
// for the inputs with bit patterns first_bits .. first_bits + count - 1, compared with `dev` (a site evaluated on the
// GPU: floats or doubles); NaN == NaN.  Returns the number of differing inputs and stores the first `cap` of them
// (input bits, device bits, host bits; for a double site: input bits, |difference| in units of the last place, 0).
unsigned long long trig_sweep_compare(int host_fn, uint32_t first_bits, int64_t count, const void *dev, int threads,
                                      uint32_t *bad3, int cap);

} // namespace djbcpu

// helpers of djb_host.hip that the host path shares (scalar set-up code: microfacet::params resolution, the
// published sgd / abc rows, error reporting)
namespace djbk {
djb_status set_error(djb_status st, const char *fmt, ...);
djb_status resolve_device_params(const djb_params *in, float out9[9], int brdf_kind);   // Params {nx,ny,nz,ax,ay,rho,s,tx,ty}
} // namespace djbk
