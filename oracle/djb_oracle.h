/* oracle/djb_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C99) of the dj_brdf hot path: microfacet eval / pdf /
 * VNDF sample (Beckmann, GGX, tabular), MERL and UTIA table lookup, and the
 * power-iteration fitter (djb::tabular ctor + fit_{beckmann,ggx}_parameters).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / reported CPU baseline.  The
 * shipped product (dj_brdf_amd + libdjb_hip.so) never links, imports or calls
 * anything in oracle/.
 *
 * Parity status: PINNED.  Every function here is checked (bit-for-bit unless
 * stated) against the real reference compiled in place from
 * /root/reference/dj_brdf.h (oracle/ref_shim.cpp -> oracle/_ref/libdjb_ref.so)
 * by tests/test_oracle_vs_ref.py in the build container, and against the golden
 * vectors committed under tests/golden/ (generated from the real reference by
 * tests/golden/make_golden.py) everywhere else.
 *
 * Arithmetic contract: storage is float; every expression that
 * the reference evaluates in double (M_PI, 1.0-style literals, unqualified
 * libm calls) is evaluated in double here and rounded once where the reference
 * rounds.  Build with: gcc -O2 -ffp-contract=off (no -ffast-math, no -march=native).
 */
#ifndef DJB_ORACLE_H
#define DJB_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float x, y, z; } o_vec3;

/* microfacet::params (hdr:213-243) */
typedef struct {
	o_vec3 n;                 /* mean normal */
	float a1, a2, phi_a;      /* ellipse */
	float ax, ay;             /* scales */
	float rho, sqrt_1mrho2;   /* correlation */
	float tx, ty;             /* location */
} o_params;

/* same wire format as ref_shim.cpp's shim_params */
typedef struct {
	int   kind;               /* 0 NULL/standard, 1 elliptic(a1,a2,phi), 2 pdfparams(ax,ay,rho,tx,ty), 3 lambert::params(rgb) */
	float v[5];
} o_param_desc;

/* O_FRESNEL_CUSTOM / O_BRDF_CUSTOM: the test-suite's USER-DEFINED classes -- the lobes and the Fresnel term that
 * oracle/ref_shim.cpp derives from djb::brdf (hdr:74-109) and djb::fresnel::impl (hdr:157-162), restated here so that the
 * reference's handling of user classes (base-class operators hdr:795-845, fits of arbitrary sources hdr:2482-2701) has an
 * oracle too.  They are fixtures of this repository, not reference code. */
enum { O_FRESNEL_IDEAL = 0, O_FRESNEL_UNPOLARIZED = 1, O_FRESNEL_SCHLICK = 2,
       O_FRESNEL_SGD = 3, O_FRESNEL_SPLINE = 4, O_FRESNEL_CUSTOM = 5 };

enum { O_BRDF_BECKMANN = 0, O_BRDF_GGX = 1, O_BRDF_TABULAR = 2, O_BRDF_MERL = 3,
       O_BRDF_UTIA = 4, O_BRDF_LAMBERT = 5, O_BRDF_SGD = 6, O_BRDF_ABC = 7, O_BRDF_TABULAR_ANISO = 8,
       O_BRDF_CUSTOM = 9 };
/* which = 0: Phong lobe {kd[3], ks[3], exponent}; 1: Ward lobe {kd[3], ks[3], ax, ay} (ref_shim.cpp: user_phong, user_ward) */
struct o_brdf *o_create_custom(int which, const float *params, int n);

/* djb::tabular_anisotropic(brdf, elevation_res, azimuthal_res, shadow) (hdr:428-478, 2238-2273) */
struct o_brdf *o_create_tabular_anisotropic(const struct o_brdf *src, int elev, int azim, int shadow);
int  o_aniso_get(const struct o_brdf *t, int which, float *out);            /* 0 p22v 1 sigmav 4 fresnel */
int  o_aniso_get_table(const struct o_brdf *t, int which, float *out);      /* 0 pdf1 1 cdf1 2 qf1 3 pdf2 4 cdf2 5 qf2; 6: count of the reference's m_qf2 */
void o_aniso_query(const struct o_brdf *t, int which, int64_t n, const float *a, const float *b, float *out);
void o_aniso_fit(const struct o_brdf *t, float *beckmann5, float *ggx5);

/* sgd: rhoD[3] rhoS[3] alpha[3] p[3] f0[3] f1[3] kap[3] lambda[3] c[3] k[3] theta0[3] (33 doubles);
 * abc: kD[3] A[3] B C ior (9 doubles) -- one row of the published tables (hdr:3312-3413, 3505-3606) */
struct o_brdf *o_create_sgd(const double *params33);
struct o_brdf *o_create_abc(const double *params9);

typedef struct o_brdf o_brdf;

/* construction */
o_brdf *o_create_microfacet(int ndf, int fkind, const float *fdata, int nf, int shadow);
o_brdf *o_create_merl_from_memory(const double *samples, int64_t n_per_channel); /* copies */
o_brdf *o_create_merl(const char *path);
o_brdf *o_create_utia_from_memory(const double *samples); /* 3*288*288 raw file doubles; copies+normalizes */
o_brdf *o_create_utia(const char *path);
o_brdf *o_create_lambert(void);
o_brdf *o_create_tabular(const o_brdf *src, int res, int shadow);
void    o_destroy(o_brdf *b);
const char *o_last_error(void);

/* operator surface; vectors are AoS float[n][3] */
void o_eval(const o_brdf *b, int op, int64_t n, const float *i, const float *o,
            const o_param_desc *pd, float *out);   /* op 0 eval,1 evalp (n x 3); 2 pdf (n) */
void o_sample(const o_brdf *b, int64_t n, const float *u1, const float *u2, const float *o,
              const o_param_desc *pd, float *out_i);
void o_evalp_is(const o_brdf *b, int64_t n, const float *u1, const float *u2, const float *o,
                const o_param_desc *pd, float *out_w, float *out_i, float *out_pdf);
void o_io_to_hd(int64_t n, const float *i, const float *o, float *h, float *d);
void o_hd_to_io(int64_t n, const float *h, const float *d, float *i, float *o);
void o_merl_index(int64_t n, const float *i, const float *o, int *idx);

void o_params_get(const o_param_desc *pd, float *out12);
void o_microfacet_query(const o_brdf *b, int which, int64_t n, const float *a, const float *bb,
                        const float *c, const o_param_desc *pd, float *out);
void o_radial_query(const o_brdf *b, int which, int64_t n, const float *a, const float *bb,
                    const float *c, float *out);
void o_fresnel_eval(const o_brdf *b, int64_t n, const float *c, float *out);
/* host libm float functions (fn 0 logf, 1 expf, 2 powf) and the restatement of glibc 2.35's algorithms
 * for them that the HIP kernels implement (use_fma: contraction of the x86-64 FMA ifunc variants) */
void o_libm_f32(int fn, int64_t n, const float *x, const float *y, float *out);
void o_glibc_f32(int fn, int use_fma, int64_t n, const float *x, const float *y, float *out);
/* host libm double exp (fn 0) / pow (fn 1) and the restatement of glibc 2.35's algorithms the HIP kernels implement */
void o_libm_f64(int fn, int64_t n, const double *x, const double *y, double *out);
void o_glibc_f64(int fn, int64_t n, const double *x, const double *y, double *out);
/* vec3::vec3(theta, phi), dj_brdf.h:589-595 */
void o_vec3_angles(int64_t n, const float *theta, const float *phi, float *out);
/* sgd / abc member queries, dj_brdf.h:505-509, 530-533: which 0 ndf(h), 1 gaf(h, i, o), 2 g1(k) [sgd], 3 fresnel(a.x) */
void o_model_query(const o_brdf *b, int which, int64_t n, const float *a, const float *i, const float *o, float *out);
/* fresnel::ior_to_f0 (dir 0) / f0_to_ior (dir 1), dj_brdf.h:1255-1290 */
void o_ior_f0(int dir, int64_t n, const float *x, float *y);
void o_erf(int64_t n, const float *x, float *y);
void o_erfinv(int64_t n, const float *x, float *y);

int  o_tabular_get(const o_brdf *t, int which, float *out);
void o_tabular_fit(const o_brdf *t, float *alpha_beckmann, float *alpha_ggx);

/* multi-threaded batch drivers used by bench.py's cpu_baseline leg (pthread static chunks) */
void o_eval_mt(const o_brdf *b, int op, int64_t n, const float *i, const float *o,
               const o_param_desc *pd, float *out, int threads);

#ifdef __cplusplus
}
#endif
#endif
