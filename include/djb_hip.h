/* include/djb_hip.h -- C ABI of libdjb_hip.so, the MI355X (gfx950) batch engine behind the
 * dj_brdf operator surface.
 *
 * The reference (jdupuy/dj_brdf, dj_brdf.h) has no FFI: its boundary is the abstract C++ class
 * djb::brdf (dj_brdf.h:74-109) whose virtuals take ONE (i, o) pair per call.  This header is the
 * batch equivalent of that surface: one entry point per virtual, taking n pairs, plus the
 * constructors of the concrete classes on the hot path.  Each declaration cites the reference
 * interface it replaces.  No C++ / torch / STL types cross this boundary: plain pointers, sizes,
 * POD structs, status codes.  INTEGRATION.md shows the djb:: facade and the Mitsuba-plugin side.
 *
 * Conventions
 *  - Directions follow the reference (dj_brdf.h:23-26): i = towards the light, o = towards the
 *    viewer, both in the local frame with z the surface normal.
 *  - Arrays are described by djb_vec3_view {x, y, z, stride}: element k is (x[k*stride],
 *    y[k*stride], z[k*stride]).  SoA = three arrays with stride 1 (fast path);
 *    an array of djb::vec3 (AoS) = {p, p+1, p+2, 3}.
 *  - output arrays may coincide with input arrays only as index-aligned in-place views (out[k] shares storage with
 *    in[k]); any other overlap of an output with an input is undefined, as for the reference's own loops.
 *  - `mem` says where the array pointers live: DJB_MEM_DEVICE (HBM of the ctx's GPU; the call is
 *    asynchronous on the ctx's stream) or DJB_MEM_HOST (the call stages through HBM and returns
 *    when the outputs are back in host memory).  Host batches of >= 2^20 units of the operator
 *    calls (eval / evalp / pdf / sample / evalp_is and the per-pair-parameter forms) are cut into
 *    chunks so that inputs travel to HBM while earlier results travel back (a helper thread and a
 *    second stream per call; results do not depend on the chunking).
 *    A DJB_MEM_DEVICE operator call of fewer than 2^20 units is a sequence of kernel launches (and at most one memset) on the
 *    ctx's stream and nothing else -- once the kind has run once on the ctx, which may allocate scratch -- so it can be recorded
 *    by hipStreamBeginCapture on that stream and replayed from a hipGraph (tests/test_gpu_graph_capture.py).
 *  - Every function returns a djb_status; djb_last_error() returns the thread-local message
 *    (the text djb::exc would have carried, dj_brdf.h:54-59 / 578-587).
 *  - Handles are immutable after creation; batch calls on one ctx are serialised on its stream.
 */
#ifndef DJB_HIP_H
#define DJB_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI version = 100 * major + 10 * minor (+ patch).  A change of an existing signature or of the meaning of an argument bumps the
 * MAJOR digit, additions bump the minor one; djb_version() returns the value the loaded library was built with, and the C++
 * facade / the Python mirror refuse a library whose major differs from the header they were written against.
 *   100  rounds 1-3
 *   200  round 4: djb_eval_lean_batch / djb_sample_lean_batch gained `int lean_flags` (after `dmapscale`) and compose the per-hit
 *        lobe as the plugin does, lrep(lean) * dmapscale + params_to_lrep(base) (mitsuba/dj_beckmannconductor.cpp:296-314) --
 *        before: lrep(base) * scale + lean.  Callers built against 100 must be recompiled.
 *   210  round 5: + djb_fit_query_dirs, djb_fit_aniso_query_dirs, djb_brdf_create_tabular_from_samples,
 *        djb_brdf_create_tabular_anisotropic_from_samples (fits of user-defined sources), djb_set_file_map_observer,
 *        DJB_FRESNEL_HOST.  No existing entry changed.
 *   220  round 5: + djb_brdf_create_user_microfacet / djb_user_ndf (user-defined NDFs on the host path), DJB_KIND_USER,
 *        djb_brdf_get_fresnel.
 *   230  round 5: + djb_helper (the reference's file-static erf / erfinv / xyz_to_theta_phi / uniform_to_concentric / rotate_vector).
 *   231  round 5: + DJB_PARAMS_RESOLVED_FOLLOWS / djb_params_cached (a parameter set that carries its resolved form: one-pair calls
 *        skip the set-up arithmetic).  A plain djb_params means what it always meant.
 *   232  round 5: + DJB_OPT_HOST_BATCH_MAX (the size up to which host-array calls are answered by the host twin; default unchanged).
 *   233  round 6: + djb_fit_merl_files_multi (the file pipeline over several contexts), djb_merl_bin_keys_batch.  The on-chip uniforms of
 *        djb_sample_rng_batch / djb_gen_uniforms are a cheaper counter hash (dj_brdf_amd/synth.py: rng_uniforms); same interface.
 *   234  round 6: + djb_selftest_fast_trig; DJB_OPT_CONTRACT_1E5 also covers utia eval / evalp (the sRGB power of the decode only).
 *   235  round 6: + djb_selftest_model_fast (the decided fast tier of the sgd / abc models). */
#define DJB_HIP_VERSION 235
#define DJB_HIP_VERSION_MAJOR(v) ((v) / 100)

typedef enum {
	DJB_OK = 0,
	DJB_ERR_INVALID_ARGUMENT = 1,
	DJB_ERR_OPEN_FAILED = 2,      /* "djb_error: Failed to open %s"          dj_brdf.h:970, 1044 */
	DJB_ERR_BAD_HEADER = 3,       /* "djb_error: Failed to read MERL header" dj_brdf.h:976       */
	DJB_ERR_READ_FAILED = 4,      /* "djb_error: Reading %s failed"          dj_brdf.h:982, 1058 */
	DJB_ERR_NOT_IMPLEMENTED = 5,  /* "djb_error: Not Implemented"            dj_brdf.h:1785-1790 */
	DJB_ERR_HIP = 6,              /* a HIP runtime call failed                                  */
	DJB_ERR_NO_DEVICE = 7,        /* no gfx950 device / HIP runtime unusable                     */
	DJB_ERR_UNKNOWN_MATERIAL = 8, /* "djb_error: No SGD/ABC parameters for %s" dj_brdf.h:3449, 3628 */
	DJB_ERR_OUT_OF_MEMORY = 9,    /* host allocation failed (std::bad_alloc never crosses this ABI) */
	DJB_ERR_INTERNAL = 10         /* any other C++ exception caught at the ABI boundary            */
} djb_status;

enum { DJB_MEM_DEVICE = 0, DJB_MEM_HOST = 1 };

/* The device argument of djb_ctx_create that selects the product's HOST execution path: the same per-unit code as
 * the kernels, compiled for the CPU with the host's own libm (what the reference -- a CPU library, dj_brdf.h:74-109
 * -- calls).  Every entry point works on such a context (memory space flags are ignored: all pointers are host
 * memory), objects belong to the context that created them, batches are chunked over DJB_CPU_THREADS threads.
 * It exists for machines without a GPU (BASELINE configs[0]: examples/merl_params.cpp "runs without a GPU"); it is
 * never entered implicitly: a GPU context fails with DJB_ERR_NO_DEVICE / DJB_ERR_HIP rather than fall back.   */
#define DJB_DEVICE_CPU (-1)
/* DJB_MEM_HOST calls of at most this many units on a GPU context -- the scalar virtuals of the djb:: facade, a
 * renderer's per-hit calls -- are answered by that same host code on the calling thread, from a host copy of the
 * object's tables (no staging, no launch, no context lock): bit-identical to the batch path, ~100 ns instead of
 * ~15 us per pair.  DJB_OPT_SCALAR_ON_DEVICE = 1 routes them through the GPU as well.  The threshold is the smallest
 * measured break-even of the operators (profiles/r03/scalar_latency.txt, one host thread vs one GPU round trip of ~22 us:
 * beckmann sample ~100 units, merl eval ~190, ggx eval ~320).                                                  */
#define DJB_SCALAR_HOST_MAX 96

typedef struct djb_ctx djb_ctx;     /* one GPU + one HIP stream */
typedef struct djb_brdf djb_brdf;   /* an immutable BRDF object resident in HBM (djb::brdf subclass) */

typedef struct { float *x, *y, *z; int64_t stride; } djb_vec3_view;

/* djb::microfacet::params factories (dj_brdf.h:217-221).  The batch shares one parameter set. */
enum { DJB_PARAMS_STANDARD = 0,   /* user_param == NULL -> params::standard()            */
       DJB_PARAMS_ELLIPTIC = 1,   /* params::elliptic(v[0]=a1, v[1]=a2, v[2]=phi_a)      */
       DJB_PARAMS_PDFPARAMS = 2,  /* params::pdfparams(ax, ay, rho, tx_n, ty_n)          */
       DJB_PARAMS_LAMBERT = 3     /* lambert::params(reflectance = v[0..2]); lambert handles only   dj_brdf.h:114-119 */ };
typedef struct { int kind; float v[5]; } djb_params;

/* resolved view of microfacet::params, all private members (dj_brdf.h:237-242) */
typedef struct {
	float n[3];
	float a1, a2, phi_a;
	float ax, ay, rho, sqrt_one_minus_rho_sqr;
	float tx_n, ty_n;
} djb_params_resolved;

/* A parameter set together with its resolved form, as the reference's params object holds it (dj_brdf.h:237-242: the factories do the
 * cos / sin / sqrt / atan set-up once, eval / pdf / sample copy the result).  `p.kind` carries DJB_PARAMS_RESOLVED_FOLLOWS and `r` is what
 * djb_params_resolve(&p, &r) returned for the same p (with the flag cleared): every entry point that takes a `const djb_params *` then
 * reads `r` instead of redoing the set-up -- 40-60 ns of a ~100 ns one-pair call.  The djb:: facade's params objects are of this form.
 * The flag is a promise about the bytes BEHIND the djb_params: a copy of `p` alone must have it cleared (DJB_PARAMS_KIND). */
#define DJB_PARAMS_RESOLVED_FOLLOWS 0x100
#define DJB_PARAMS_KIND(k) ((k) & 0xff)
typedef struct { djb_params p; djb_params_resolved r; } djb_params_cached;

/* djb::fresnel::{ideal,unpolarized,schlick,sgd,spline} (dj_brdf.h:149-207).
 * DJB_FRESNEL_HOST marks a term only the CALLER can evaluate -- a class a user derived from fresnel::impl
 * (dj_brdf.h:157-162).  It never reaches a kernel: the constructors and djb_brdf_set_fresnel refuse it
 * (DJB_ERR_INVALID_ARGUMENT).  The caller creates the object with DJB_FRESNEL_IDEAL instead -- that term is exactly
 * (1, 1, 1), so evalp returns the bare D G / (4 o.z) and evalp_is G / G1 -- and multiplies by its own F(cos theta_d) on
 * the host, which is the reference's expression `F * scalar` (dj_brdf.h:1545, 1762) operation for operation;
 * include/djb_hip.hpp's microfacet::evalp / evalp_is do exactly that.                                               */
enum { DJB_FRESNEL_IDEAL = 0, DJB_FRESNEL_UNPOLARIZED = 1, DJB_FRESNEL_SCHLICK = 2,
       DJB_FRESNEL_SGD = 3, DJB_FRESNEL_SPLINE = 4, DJB_FRESNEL_HOST = 5 };
typedef struct {
	int kind;
	float a[3];           /* unpolarized: ior; schlick: f0; sgd: f0 */
	float b[3];           /* sgd: f1 */
	const float *points;  /* spline: npoints x 3 floats (host memory; copied) */
	int npoints;
} djb_fresnel_desc;

enum { DJB_KIND_BECKMANN = 0, DJB_KIND_GGX = 1, DJB_KIND_TABULAR = 2, DJB_KIND_MERL = 3,
       DJB_KIND_UTIA = 4, DJB_KIND_LAMBERT = 5, DJB_KIND_SGD = 6, DJB_KIND_ABC = 7,
       DJB_KIND_TABULAR_ANISO = 8, DJB_KIND_USER = 9 };

/* ---------------------------------------------------------------- library / context */
const char *djb_last_error(void);
int         djb_version(void);
/* number of usable gfx950 devices (0 when there is no GPU; never falls back to a CPU path) */
djb_status  djb_device_count(int *count);
/* a context with its own (non-blocking) HIP stream; device = DJB_DEVICE_CPU: the host execution path */
djb_status  djb_ctx_create(int device, djb_ctx **out);
/* a context that runs on the caller's hipStream_t (e.g. torch's current stream); NULL means the
 * device's default (null) stream, so work is ordered with everything else issued there */
djb_status  djb_ctx_create_on_stream(int device, void *hip_stream, djb_ctx **out);
djb_status  djb_ctx_destroy(djb_ctx *ctx);
djb_status  djb_ctx_synchronize(djb_ctx *ctx);
void       *djb_ctx_stream(djb_ctx *ctx);
/* move the context onto another hipStream_t of its device (NULL = the default stream).  The new stream
 * first waits for everything the context enqueued so far, so per-context scratch stays ordered.  Callers
 * that follow a framework's "current stream" (torch) call this before each batch call; cheap when unchanged. */
djb_status  djb_ctx_set_stream(djb_ctx *ctx, void *hip_stream);
/* The host's libm and the two execution paths.  The kernels reproduce what glibc 2.35 (x86-64, FMA ifunc variants)
 * returns for the libm calls the reference makes (exp / pow / atan2 / sin / cos / tan / acos, logf / expf / powf); the host
 * path (CPU contexts, scalar-size DJB_MEM_HOST calls) calls the host's libm.  The first context creation compares the two on
 * a fixed probe set.  djb_ctx_libm_matches_host: 1 = identical (the host path keeps calling the host's libm), 0 = the host's
 * libm differs -- the host path then runs the kernels' restatements compiled for the host, so scalar calls and GPU batches
 * still agree bit for bit, and one line on stderr says so -- , -1 = not checked (CPU without FMA).  DJB_HOST_LIBM=restated|host
 * (environment) forces the choice; djb_host_libm_mode returns it (1 = restatements).  atan and log, which occur only in
 * float -> float sites, have no restatement: djb_host_atan_log_kat reports whether the host's match glibc 2.35's known answers. */
int djb_ctx_libm_matches_host(const djb_ctx *);
int djb_host_libm_mode(void);
int djb_host_atan_log_kat(void);

/* options.  DJB_OPT_MERL_EXACT_ONLY = 1 makes merl eval/evalp run the operation-by-operation fp64
 * kernel for every pair instead of the two-tier kernel (fp32 fast path + guard bands + fp64 path
 * for ambiguous pairs); both give the same bits, the option exists to verify that.           */
enum { DJB_OPT_MERL_EXACT_ONLY = 1,
/* DJB_OPT_ANISO_QF2_ALIGNED = 1: tabular_anisotropic objects created afterwards keep every row of the
 * conditional quantile table at its own offset (padded with 1.0).  Default 0 = the reference's vector,
 * in which a row whose conditional CDF cannot be inverted for every quantile comes up short and shifts
 * all later rows (dj_brdf.h:3005-3034) -- the two differ only for such (grazing-heavy) data.          */
       DJB_OPT_ANISO_QF2_ALIGNED = 2,
/* DJB_OPT_SCALAR_ON_DEVICE = 1: scalar-size DJB_MEM_HOST calls (<= DJB_SCALAR_HOST_MAX units) run on the GPU too
 * (initial value: the environment variable DJB_SCALAR_ON_DEVICE, default 0) */
       DJB_OPT_SCALAR_ON_DEVICE = 3,
/* DJB_OPT_FIT_FILES_DENSE = 1: djb_fit_merl_files uploads and converts every 35 MB table in full before the fit (the
 * round-1 pipeline) instead of fetching only the ~5.5 k entries per file that tabular(merl, res) reads; same alphas */
       DJB_OPT_FIT_FILES_DENSE = 4,
/* DJB_OPT_UTIA_EXACT_ONLY = 1: utia eval / evalp batches run one kernel that carries the exact fall-backs of the azimuths
 * (glibc's atan2) and of the sRGB power inline, instead of the two-tier form (tier 1 without them + a worklist of the
 * pairs that sit next to a float rounding boundary, re-evaluated by a second kernel); both give the same bits, the
 * option exists to verify that. */
       DJB_OPT_UTIA_EXACT_ONLY = 5,
/* DJB_OPT_CONTRACT_1E5 = 1 (off by default): dense device-resident eval / evalp / pdf batches of GGX and Beckmann (ideal or
 * schlick Fresnel, f0 >= 0.01; params without mean-normal offset, |rho| <= 0.9) and of ABC (one ior > 1, 0 <= B < 1e12,
 * 1e-3 <= C <= 16: every published row) are evaluated inside the VALUE contract of the north star --
 * every result within 1e-5 relative of the reference's, zeros exactly where the reference returns zeros -- instead of
 * bit-identically: reciprocal / rsqrt instructions and merged denominators in place of the reference's 15 correctly
 * rounded divisions, pairs whose reference value is ill-conditioned re-done by the bit-exact code (two tiers, as for
 * MERL).  Since round 4 also: unpolarized Fresnel (ior >= 1.05) for GGX / Beckmann, sgd::eval, and `sample` of a Beckmann or GGX lobe
 * (djb_sample_batch / djb_sample_rng_batch; 1e-3 <= ax, ay <= 100, |rho| <= 0.99): every component of the returned unit vector
 * within 1e-5 of the reference's, samples whose decisions or conditioning are in doubt re-done by the bit-exact code in the same
 * launch; and evalp_is of GGX / Beckmann (same Fresnel / params domain as eval): the sampled direction stays the reference's
 * bit for bit -- its pdf moves by 1e-3 for a 1e-5 change of direction -- weight and pdf within 1e-5 relative.  Since ABI 234 also
 * utia eval / evalp: the grid cells, the weights and the 16-tap sums remain the reference's bits (so do the sRGB knee and the clamp
 * decisions), only pow(t, 2.4f) of the decode runs on the fast transcendentals (<= 1.5e-6 relative).  Everything else -- MERL look-ups,
 * every MERL / UTIA bin decision, the fitters, other lobes and layouts -- is unaffected and stays bit-identical.  djb_selftest_contract and
 * djb_selftest_contract_sample measure the actual maximum difference. */
       DJB_OPT_CONTRACT_1E5 = 6,
/* DJB_OPT_TEST_WORKLIST_CAP = <entries> (tests only; -1 = automatic, the default): overrides the capacity of the tier-2
 * worklist of the two-tier kernels (utia, contract mode; the MERL look-up drains its tier 2 in-kernel and has none), to exercise
 * the overflow path in which the second kernel redoes the whole batch.  Results never depend on the capacity. */
       DJB_OPT_TEST_WORKLIST_CAP = 7,
/* DJB_OPT_HOST_BATCH_MAX = <units> (default DJB_SCALAR_HOST_MAX = 96; 0 .. 65536): DJB_MEM_HOST calls of up to that many units are
 * answered on the CALLING thread by the object's host twin (no staging, no launch, no context lock) instead of a ~22 us GPU round
 * trip.  96 is the break-even of the most expensive operator (Beckmann sample, ~160 ns per unit on one core); a GGX / tabulated eval
 * costs ~31 ns per pair, so a caller that submits host batches of a few hundred pairs from several render threads -- where batch
 * calls on one context also serialise on its lock -- gains from 512 or so.  The results are the same bits either way. */
       DJB_OPT_HOST_BATCH_MAX = 8 };
djb_status  djb_ctx_set_option(djb_ctx *ctx, int option, int value);
/* Observer of the MERL file pipeline (djb_fit_merl_files, both context kinds): `fn(path, user)` is called from the reader thread
 * after a file has passed its size check and has been mapped, before its entries are gathered; NULL removes it.  Diagnostics /
 * tests (a callback that truncates the file there exercises the "file shrinks under the mapping" guard: the verdict is the
 * reference's "Reading %s failed", dj_brdf.h:979-982, not SIGBUS).  The library itself never writes to an input file.  */
djb_status  djb_set_file_map_observer(void (*fn)(const char *path, void *user), void *user);
/* HIP-event timing on the ctx stream (what bench.py's roofline leg uses) */
djb_status  djb_timer_start(djb_ctx *ctx);
djb_status  djb_timer_stop_ms(djb_ctx *ctx, float *ms);

/* ---------------------------------------------------------------- constructors */
/* djb::beckmann(fresnel, shadow) / djb::ggx(fresnel, shadow)          dj_brdf.h:358-360, 376-378 */
djb_status djb_brdf_create_beckmann(djb_ctx *, const djb_fresnel_desc *, int shadow, djb_brdf **);
djb_status djb_brdf_create_ggx(djb_ctx *, const djb_fresnel_desc *, int shadow, djb_brdf **);
/* djb::merl(const char *filename)                                     dj_brdf.h:963-983 */
djb_status djb_brdf_create_merl_from_file(djb_ctx *, const char *path, djb_brdf **);
/* same object from the file payload already in host memory: 3*n doubles, planes R,G,B */
djb_status djb_brdf_create_merl_from_memory(djb_ctx *, const double *samples, int64_t n_per_channel,
                                            djb_brdf **);
/* djb::utia(const char *filename)                                     dj_brdf.h:1039-1059 */
djb_status djb_brdf_create_utia_from_file(djb_ctx *, const char *path, djb_brdf **);
djb_status djb_brdf_create_utia_from_memory(djb_ctx *, const double *samples, djb_brdf **);
/* djb::lambert                                                        dj_brdf.h:112-123 */
djb_status djb_brdf_create_lambert(djb_ctx *, djb_brdf **);
/* djb::sgd(const char *name) / djb::abc(const char *name): Shifted-Gamma and ABC models with the
 * published per-material parameters (compiled in from the CSVs in dj_brdf_amd/data; SGD rows also
 * answer to their alias).  Unknown names -> DJB_ERR_UNKNOWN_MATERIAL.  dj_brdf.h:502, 527, 3436, 3617
 * The _from_params forms take one table row: sgd = rhoD rhoS alpha p f0 f1 kap lambda c k theta0
 * (11 x RGB = 33 doubles), abc = kD[3] A[3] B C ior (9 doubles).                               */
djb_status djb_brdf_create_sgd(djb_ctx *, const char *name, djb_brdf **);
djb_status djb_brdf_create_abc(djb_ctx *, const char *name, djb_brdf **);
djb_status djb_brdf_create_sgd_from_params(djb_ctx *, const double *params33, djb_brdf **);
djb_status djb_brdf_create_abc_from_params(djb_ctx *, const double *params9, djb_brdf **);
/* djb::tabular(const brdf&, int res, bool shadow): the power-iteration fit, run on the GPU
 *                                                                     dj_brdf.h:2215-2236 */
djb_status djb_brdf_create_tabular(djb_ctx *, const djb_brdf *src, int res, int shadow, djb_brdf **);
/* djb::tabular_anisotropic(const brdf&, int elevation_res, int azimuthal_res, bool shadow): the
 * anisotropic power-iteration fit on a (theta, phi) grid; the (w*h)^2 Smith kernel matrix (513 MB at
 * 90 x 90 in the reference) is recomputed on the fly, never stored.           dj_brdf.h:441-444 */
djb_status djb_brdf_create_tabular_anisotropic(djb_ctx *, const djb_brdf *src, int elevation_res,
                                               int azimuthal_res, int shadow, djb_brdf **);
/* ---- USER-DEFINED microfacet NDFs.  The reference's third extension point: a class derived from djb::radial overrides its public
 * virtuals p22_radial / sigma_std_radial / cdf_radial / qf_radial (+ qf2_radial / qf3_radial for Smith VNDF sampling,
 * dj_brdf.h:307-314), one derived from djb::microfacet the protected p22_std / sigma_std / sample_vp22_std_* (dj_brdf.h:283-295);
 * everything else -- params, the stretch of sigma, G1 / G2, eval / evalp / pdf / sample / evalp_is, the queries -- is base-class code.
 * Here the base-class code is the library's per-unit code on its HOST path and the user's functions are callbacks: the object lives
 * on a CPU context (a GPU cannot call host code; batches are spread over the context's threads), fits of it run wherever the
 * caller likes through the *_from_samples constructors above.  p22_radial != NULL makes it a radial NDF (required then:
 * sigma_std_radial, qf_radial; qf2_radial / qf3_radial when supports_smith_vndf_sampling returns non-zero); otherwise p22_std,
 * sigma_std and sample_vp22_std (= the virtual sample_vp22_std_smith, whose default forwards to _nmap) are required.       */
typedef struct {
	void  *user;
	int   (*supports_smith_vndf_sampling)(void *user);
	float (*p22_radial)(void *user, float r_sqr);
	float (*sigma_std_radial)(void *user, float cos_theta_k);
	float (*cdf_radial)(void *user, float r);
	float (*qf_radial)(void *user, float u);
	float (*qf2_radial)(void *user, float u, float cos_theta_k, float sin_theta_k);
	float (*qf3_radial)(void *user, float u, float qf2);
	float (*p22_std)(void *user, float x, float y);
	float (*sigma_std)(void *user, const float k[3]);
	void  (*sample_vp22_std)(void *user, float u1, float u2, const float k[3], float *xslope, float *yslope);
} djb_user_ndf;
djb_status djb_brdf_create_user_microfacet(djb_ctx *cpu_ctx, const djb_user_ndf *ndf, const djb_fresnel_desc *, int shadow,
                                           djb_brdf **);
/* ---- fits of USER-DEFINED sources.  The reference's extension point is `class brdf` with `eval` as its one pure virtual
 * (dj_brdf.h:74-109); tabular's and tabular_anisotropic's constructors only ever call brdf.eval, at directions fixed by
 * the resolution: the res-1 back-scatter pairs eval(w, w) of compute_p22_smith (dj_brdf.h:2482-2522; (elev-1)*azim of
 * them for the anisotropic grid, dj_brdf.h:2525-2579) and the pairs of compute_fresnel with dir_i = (0,0,1)
 * (dj_brdf.h:2583-2641, 2643-2701).  djb_fit_query_dirs / djb_fit_aniso_query_dirs return those (i, o) pairs (host arrays,
 * any view layout) in the reference's CALL ORDER; a pair its loop never reaches has NaN components and must be skipped.
 * The caller evaluates its BRDF there (host code: a C++ virtual, a Python callable, a cgo callback ...), writes rgb[3*s]
 * for slot s (skipped slots: any value) and the *_from_samples constructors run the same fit kernels on them.  Pure host
 * functions; out_i / out_o may be NULL to query *count only.                                                        */
djb_status djb_fit_query_dirs(int res, int64_t capacity, const djb_vec3_view *out_i, const djb_vec3_view *out_o,
                              int64_t *count);
djb_status djb_fit_aniso_query_dirs(int elevation_res, int azimuthal_res, int64_t capacity, const djb_vec3_view *out_i,
                                    const djb_vec3_view *out_o, int64_t *count);
/* djb::tabular(const brdf &user_defined, res, shadow)                 dj_brdf.h:2215-2236 */
djb_status djb_brdf_create_tabular_from_samples(djb_ctx *, int res, int shadow, const float *rgb, int64_t count,
                                                djb_brdf **);
/* djb::tabular_anisotropic(const brdf &user_defined, elev, azim, shadow)   dj_brdf.h:2238-2273 */
djb_status djb_brdf_create_tabular_anisotropic_from_samples(djb_ctx *, int elevation_res, int azimuthal_res, int shadow,
                                                            const float *rgb, int64_t count, djb_brdf **);
djb_status djb_brdf_destroy(djb_brdf *);
int        djb_brdf_kind(const djb_brdf *);
/* merl::get_samples() / utia::get_samples(): the table as the reference holds it -- the file's
 * doubles for MERL (3 x 90 x 90 x 180), the clamped and scaled doubles for UTIA (3 x 288 x 288,
 * after utia::normalize).  out == NULL only reports *count.                dj_brdf.h:132, 143 */
djb_status djb_brdf_get_samples(const djb_brdf *, double *out, int64_t capacity, int64_t *count);
/* microfacet::set_shadow / get_shadow                                 dj_brdf.h:278-281 */
int        djb_brdf_get_shadow(const djb_brdf *);
/* The two microfacet mutators (beckmann, ggx, tabular, tabular_anisotropic handles only).  Like the
 * reference's non-const members they must not race with batch calls on the same handle from other
 * threads; launches already enqueued keep the state they were launched with.
 * microfacet::set_shadow(bool)                                        dj_brdf.h:278      */
djb_status djb_brdf_set_shadow(djb_brdf *, int shadow);
/* microfacet::set_fresnel(const fresnel::impl &) -- e.g. tab->set_fresnel(fresnel::ideal()) after a
 * fit (mitsuba/dj_brdf.cpp:214).  NULL = fresnel::ideal.               dj_brdf.h:279, 1521-1525 */
djb_status djb_brdf_set_fresnel(djb_brdf *, const djb_fresnel_desc *fresnel);
/* microfacet::get_fresnel / sgd::get_fresnel / abc::get_fresnel: the term the object evaluates (sgd: fresnel::sgd(f0, f1) of its
 * table row, abc: fresnel::unpolarized(vec3(ior))); for a spline `points` stays valid as long as the object and its Fresnel term
 * do.                                                                              dj_brdf.h:282, 510, 534, 3443, 3623 */
djb_status djb_brdf_get_fresnel(const djb_brdf *, djb_fresnel_desc *out);

/* ---------------------------------------------------------------- the operator surface */
/* brdf::eval(i, o, user_param) -> vec3                                dj_brdf.h:77-78   */
djb_status djb_eval_batch(djb_ctx *, const djb_brdf *, int64_t n, const djb_vec3_view *i,
                          const djb_vec3_view *o, const djb_params *params,
                          const djb_vec3_view *out_fr, int mem);
/* brdf::evalp(i, o, user_param) = f_r * cos(theta_i)                  dj_brdf.h:82-83   */
djb_status djb_evalp_batch(djb_ctx *, const djb_brdf *, int64_t n, const djb_vec3_view *i,
                           const djb_vec3_view *o, const djb_params *params,
                           const djb_vec3_view *out_fr_cos, int mem);
/* brdf::pdf(i, o, user_param)                                         dj_brdf.h:96-97   */
djb_status djb_pdf_batch(djb_ctx *, const djb_brdf *, int64_t n, const djb_vec3_view *i,
                         const djb_vec3_view *o, const djb_params *params,
                         float *out_pdf, int mem);
/* eval (or evalp when want_cos != 0) and pdf of the same pairs in one pass over HBM */
djb_status djb_eval_pdf_batch(djb_ctx *, const djb_brdf *, int64_t n, const djb_vec3_view *i,
                              const djb_vec3_view *o, const djb_params *params, int want_cos,
                              const djb_vec3_view *out_fr, float *out_pdf, int mem);
/* brdf::sample(u1, u2, o, user_param) -> i                            dj_brdf.h:92-94   */
djb_status djb_sample_batch(djb_ctx *, const djb_brdf *, int64_t n, const float *u1,
                            const float *u2, const djb_vec3_view *o, const djb_params *params,
                            const djb_vec3_view *out_i, int mem);
/* brdf::sample with the two uniforms drawn on chip from the counter RNG of djb_gen_uniforms
 * (u_k = uniform(seed, start + k)); device memory only.  Same results as djb_gen_uniforms
 * followed by djb_sample_batch, without the 8 B/sample of HBM traffic.                  */
djb_status djb_sample_rng_batch(djb_ctx *, const djb_brdf *, int64_t n, uint32_t seed_u1,
                                uint32_t seed_u2, uint64_t start, const djb_vec3_view *o,
                                const djb_params *params, const djb_vec3_view *out_i);
/* brdf::evalp_is(u1, u2, o, &i, &pdf, user_param) -> weight           dj_brdf.h:87-90   */
djb_status djb_evalp_is_batch(djb_ctx *, const djb_brdf *, int64_t n, const float *u1,
                              const float *u2, const djb_vec3_view *o, const djb_params *params,
                              const djb_vec3_view *out_weight, const djb_vec3_view *out_i,
                              float *out_pdf, int mem);
/* microfacet and radial queries, batched (dj_brdf.h:258-276, 307-314, 366, 384).  a/b/c are the
 * call's arguments in declaration order, each as a vec3 view (scalars in .x, (x,y) slopes in
 * .x/.y, qf2_radial's (u, cos, sin) in .x/.y/.z); the result is written to out.x (Fresnel: xyz).
 *   NDF(h) GAF(h,i,o) G1(h,k) SIGMA(k) P22(x,y) VP22(x,y | k) VNDF(h,k) FRESNEL(cos_theta_d)
 *   P22_RADIAL(r_sqr) SIGMA_STD_RADIAL(cos) CDF_RADIAL(r) QF_RADIAL(u) QF2_RADIAL(u,cos,sin)
 *   QF3_RADIAL(u,qf2) QF1(u)                                                            */
enum { DJB_Q_NDF = 0, DJB_Q_GAF = 1, DJB_Q_G1 = 2, DJB_Q_SIGMA = 3, DJB_Q_P22 = 4, DJB_Q_VP22 = 5,
       DJB_Q_VNDF = 6, DJB_Q_FRESNEL = 7, DJB_Q_P22_RADIAL = 16, DJB_Q_SIGMA_STD_RADIAL = 17,
       DJB_Q_CDF_RADIAL = 18, DJB_Q_QF_RADIAL = 19, DJB_Q_QF2_RADIAL = 20, DJB_Q_QF3_RADIAL = 21,
       DJB_Q_QF1 = 22,
       /* tabular_anisotropic only (dj_brdf.h:450-455): PDF1(phi) CDF1(phi) QF1(u) PDF2(theta,phi)
        * CDF2(theta,phi) QF2(u,phi), arguments in a.x / a.y */
       DJB_Q_ANISO_PDF1 = 32, DJB_Q_ANISO_CDF1 = 33, DJB_Q_ANISO_QF1 = 34, DJB_Q_ANISO_PDF2 = 35,
       DJB_Q_ANISO_CDF2 = 36, DJB_Q_ANISO_QF2 = 37,
       /* sgd / abc handles (dj_brdf.h:505-509, 530-533): ndf(h) -> rgb; gaf(h, i, o) -> rgb (sgd) or out.x (abc);
        * g1(k) -> rgb (sgd only); DJB_Q_FRESNEL works for them too */
       DJB_Q_MODEL_NDF = 48, DJB_Q_MODEL_GAF = 49, DJB_Q_MODEL_G1 = 50 };
djb_status djb_query_batch(djb_ctx *, const djb_brdf *, int which, int64_t n, const djb_vec3_view *a,
                           const djb_vec3_view *b, const djb_vec3_view *c, const djb_params *params,
                           const djb_vec3_view *out, int mem);
/* ---- per-pair microfacet parameters (what dj_beckmannconductor builds per hit,
 * mitsuba/dj_beckmannconductor.cpp:291-319).  want = 1 eval | 2 evalp, optionally | 4 pdf.
 * djb_eval_pp_batch:   pdfparams = n records (ax, ay, rho, tx_n, ty_n), i.e. evalp(i, o, &params_k).
 * djb_eval_lean_batch: lean = n records of LEAN/LEADR texel moments (E1..E5); per pair, in the plugin's order
 *   (mitsuba/dj_beckmannconductor.cpp:296-314, repeated at 344-362 and 384-402):
 *     lrep1 = lrep(E1, E2, E3, E4, E5)               [DJB_LEAN_NAIVE_MIP: lrep(E1, E2, E1*E1, E2*E2, E1*E2), leanFiltering=false]
 *     lrep1 *= scale                                  [m_dmapScale]
 *     params_k = lrep_to_params(lrep1 + params_to_lrep(base))          (dj_brdf.h:1965-2033; operator+ is not commutative in float)
 *   DJB_LEAN_BIASED: the records are raw texels, E1 -= 25, E2 -= 25, E5 -= 625 first (l.300-303).
 *   If out_pdfparams != NULL, the resolved (ax, ay, rho, tx, ty) are written back.
 * Records live in the same memory space as the directions.                                 */
enum { DJB_LEAN_NAIVE_MIP = 1, DJB_LEAN_BIASED = 2 };
djb_status djb_eval_pp_batch(djb_ctx *, const djb_brdf *, int64_t n, const djb_vec3_view *i,
                             const djb_vec3_view *o, const float *pdfparams, int want,
                             const djb_vec3_view *out_fr, float *out_pdf, int mem);
djb_status djb_eval_lean_batch(djb_ctx *, const djb_brdf *, int64_t n, const djb_vec3_view *i,
                               const djb_vec3_view *o, const djb_params *base, float scale, int lean_flags,
                               const float *lean, int want, const djb_vec3_view *out_fr,
                               float *out_pdf, float *out_pdfparams, int mem);
/* sample() (out_w == out_pdf == NULL) / evalp_is() with the same per-pair records: the batch form of
 * dj_beckmann_conductor::sample (mitsuba/dj_beckmannconductor.cpp:373-413): params_k as above, then
 * evalp_is(u1_k, u2_k, o_k, &i_k, &pdf_k, &params_k) (dj_brdf.h:1734-1765). */
djb_status djb_sample_pp_batch(djb_ctx *, const djb_brdf *, int64_t n, const float *u1, const float *u2,
                               const djb_vec3_view *o, const float *pdfparams, const djb_vec3_view *out_w,
                               const djb_vec3_view *out_i, float *out_pdf, int mem);
djb_status djb_sample_lean_batch(djb_ctx *, const djb_brdf *, int64_t n, const float *u1, const float *u2,
                                 const djb_vec3_view *o, const djb_params *base, float scale, int lean_flags,
                                 const float *lean, const djb_vec3_view *out_w, const djb_vec3_view *out_i,
                                 float *out_pdf, float *out_pdfparams, int mem);
/* beckmann::lrep algebra on {E1..E5} (host scalars; dj_brdf.h:330-356, 1959-2051).  b may be NULL
 * (= the default lrep(0,0,1,1,0)); x (and y) are the scalar arguments of mul / shear / scale.
 * IADD keeps the reference's operator+= ordering (dj_brdf.h:2013-2017), which differs from ADD.   */
enum { DJB_LREP_ADD = 0, DJB_LREP_MUL = 1, DJB_LREP_IADD = 2, DJB_LREP_IMUL = 3, DJB_LREP_SHEAR = 4,
       DJB_LREP_SCALE = 5 };
djb_status djb_lrep_op(int op, const float *a, const float *b, float x, float y, float *out);
djb_status djb_params_to_lrep(const djb_params *params, float *out_lrep);      /* beckmann::params_to_lrep */
djb_status djb_lrep_to_params(const float *lrep, djb_params *out_pdfparams);   /* beckmann::lrep_to_params */

/* The file-static helpers of the reference's implementation section (dj_brdf.h:650-765) -- erf (A&S 7.1.26), erfinv (Giles),
 * xyz_to_theta_phi, uniform_to_concentric (Cline), rotate_vector (Rodrigues) -- for callers that were compiled against them (a user-defined
 * lobe's own sample() or NDF): one call, host scalars, the arithmetic of the library's operators.  in / out:
 *   ERF, ERFINV: in[0] -> out[0];  XYZ_TO_THETA_PHI: in[0..2] -> out[0] theta, out[1] phi;  UNIFORM_TO_CONCENTRIC: in[0..1] -> out[0..1];
 *   ROTATE_VECTOR: in[0..2] x, in[3..5] axis, in[6] angle -> out[0..2].                                                          (ABI 230) */
enum { DJB_HELPER_ERF = 0, DJB_HELPER_ERFINV = 1, DJB_HELPER_XYZ_TO_THETA_PHI = 2, DJB_HELPER_UNIFORM_TO_CONCENTRIC = 3, DJB_HELPER_ROTATE_VECTOR = 4 };
djb_status djb_helper(int which, const float *in, float *out);

/* brdf::io_to_hd / brdf::hd_to_io (static)                            dj_brdf.h:99-100  */
djb_status djb_io_to_hd_batch(djb_ctx *, int64_t n, const djb_vec3_view *i, const djb_vec3_view *o,
                              const djb_vec3_view *out_h, const djb_vec3_view *out_d, int mem);
djb_status djb_hd_to_io_batch(djb_ctx *, int64_t n, const djb_vec3_view *h, const djb_vec3_view *d,
                              const djb_vec3_view *out_i, const djb_vec3_view *out_o, int mem);
/* the table index merl::eval composes (diagnostic; dj_brdf.h:997-1002)                  */
djb_status djb_merl_index_batch(djb_ctx *, int64_t n, const djb_vec3_view *i,
                                const djb_vec3_view *o, int32_t *out_index, int mem);
/* The 21-bit MERL bin key of every pair from the look-up's TIER-1 arithmetic only (fp32 closed forms; no fp64 path, no table access):
 * key == the index djb_merl_index_batch returns for every pair tier 1 is certain of (99.6 % of random directions; all of them on a
 * CPU context), the neighbouring bin its estimate falls into otherwise.  For ORDERING a batch before djb_eval_batch on a merl object --
 * the look-up's rate depends on how many distinct 128-byte table lines a launch touches per unit time (bench.py
 * secondary.merl_eval_order_lever: as generated / bucketed per 4096-pair tile / sorted by key) -- e.g. a wavefront renderer that
 * already sorts its hits by material can append these bits to its sort key.  Not a substitute for the exact index.   (ABI 233) */
djb_status djb_merl_bin_keys_batch(djb_ctx *, int64_t n, const djb_vec3_view *i, const djb_vec3_view *o,
                                   uint32_t *out_keys, int mem);

/* calibration of the two-tier MERL kernel on n device-resident pairs: max over the batch of
 * |fp32 estimate - reference value| / guard band for (theta_h, theta_d, phi_d), and the counters
 * {special-region pairs, ambiguous pairs, index mismatches among "certain" pairs (must be 0),
 * certain pairs}.  guard6 = {a_h, b_h, a_d, b_d, c_d, a_p} in units of 2^-24; NULL = the shipped constants.                           */
djb_status djb_merl_guard_stats(djb_ctx *, int64_t n, const djb_vec3_view *i, const djb_vec3_view *o,
                                const float *guard6, float *max_ratio3, unsigned long long *counters4);

/* directed search against the same guard bands: every one of the n device-resident candidate pairs (i, o: updated in
 * place) hill-climbs over the bit patterns of its six floats for `iters` moves (+-2^e units in the last place of one
 * coordinate, e = 0..20), keeping a move when |estimate - reference| / band grows.  best_ratio: device float[n], the
 * final ratio of each candidate; counters3 (host) = {evaluations, index mismatches among pairs tier 1 called certain
 * (must be 0), accepted moves}.  tools/merl_guard_attack.py restarts it from every adversarial family.            */
djb_status djb_merl_guard_attack(djb_ctx *, int64_t n, const djb_vec3_view *i, const djb_vec3_view *o, const float *guard6,
                                 int iters, uint32_t seed, float *best_ratio, unsigned long long *counters3);

/* self-test of the kernels' guarded fp64 shortcuts (float(1/sqrt(double x)), float(1/q), the sRGB
 * decode float(pow(t, 2.4f))) against the exact double sequences on n hash-generated inputs:
 * counters[12] = {rsqrt mismatches, reciprocal mismatches, rsqrt exact-path fallbacks, reciprocal
 * fallbacks, sRGB-decode mismatches, sRGB-decode fallbacks, mismatches of the exact division through a double
 * reciprocal (float(double(a) * R) vs a / b), its IEEE fallbacks, mismatches of float(sqrt(a)) for a double a, its fallbacks,
 * mismatches of float(num / den) for two doubles (the quotient of GGX's quantile function), its fallbacks};
 * every mismatch count must be 0. */
djb_status djb_selftest_guarded_math(djb_ctx *, int64_t n, uint32_t seed, unsigned long long *counters12);
/* The table-driven kinds take their float -> float trig sites (the polar / azimuth angles of utia::eval, the table coordinates of
 * tabular and tabular_anisotropic) from one branch-free fp64 arctangent core and keep a value only where it is decided -- further from
 * a float rounding boundary than the core's error -- otherwise the site's previous form answers.  This runs the sites
 * against those forms.  mode 0: the n floats whose bit patterns follow `first` as polar cosines (float(r2d * acos(z)), dj_brdf.h:1066),
 * decided ones; mode 1 / 8: n hash-generated float pairs as (y, x) of float(r2d * atan2(y, x)) (:1068) / float(atan2(y, x)) (:659),
 * decided ones; mode 2..7: the n floats after `first` through the site with the core against the site without it, every one of them
 * (acos, 2 acos / pi, 2 acos / float(pi), 2 atan / float(pi), sqrt(2 atan / float(pi)), atan(sqrt)); mode 9: float(tan(double x)) of the
 * tabulated lobes' samplers the same way.  counters4 = {decided, different
 * (must be 0), left to the previous form, the largest distance of a decided double from the device libm's in units of 2^-52 of the
 * value (modes 0, 1, 8; the guard is 4096)}. */
djb_status djb_selftest_fast_trig(djb_ctx *, int64_t n, int mode, uint32_t first, uint32_t seed, unsigned long long *counters4);
/* The sgd and abc models round the ends of fp64 chains (glibc's pow and exp) to float: nine values per sgd pair, three per abc pair.  The
 * kernels evaluate each chain once in plain double together with a bound on its distance from the reference's own double, and keep the
 * float only when both ends of that interval round to it; the reference's chain answers for the rest (csrc/djb_fast_models.inc,
 * DESIGN.md 4.2).  This runs n generated polar cosines (uniform, hugging the wall of sgd's shadowing term, grazing, next to the normal)
 * through the product's g1 / ndf and through the exact chains alone: counters6 = {g1 values, g1 values left to the exact chain, g1 values
 * that differ (must be 0), ndf values, left, different (must be 0)} (abc: the ndf half only).  seed == 0: the n floats whose bit patterns follow
 * `first` instead (bits 1 .. 0x3f800000 are every float polar cosine of (0, 1]: an exhaustive run per row takes ~0.1 s). */
djb_status djb_selftest_model_fast(djb_ctx *, const djb_brdf *, int64_t n, uint32_t seed, uint32_t first, unsigned long long *counters6);
/* the DJB_OPT_CONTRACT_1E5 fast path against the bit-exact per-pair code on n generated pairs (family 0: the bench
 * distribution; 1: grazing with opposite azimuths; 2: near-normal incidence; 3: o at the horizon; 4: un-normalised):
 * max_rel2 = {max relative difference of the eval rgb, of the pdf} over the fast-path pairs, counters4 = {pairs, pairs
 * handed to the exact tier, values where exactly one side is zero (must be 0), values outside 1e-5 (must be 0)}.
 * DJB_ERR_INVALID_ARGUMENT when brdf / params are outside the fast path's domain. */
djb_status djb_selftest_contract(djb_ctx *, const djb_brdf *, const djb_params *params, int64_t n, uint32_t seed, int family,
                                 float *max_rel2, unsigned long long *counters4);
/* the DJB_OPT_CONTRACT_1E5 sampler (`sample` of a Beckmann or GGX lobe) against the bit-exact per-sample code on n generated (u1, u2, o)
 * (family 0: the bench distribution; 1: grazing view; 2: near-normal view; 3: both uniforms in their tails; 4: un-normalised
 * view): max_abs2 = {largest |component difference| among the samples the fast path kept, largest difference / per-sample
 * error bound among them (the share of the bound that is ever used; < 1)}, counters4 = {samples, samples
 * handed to the exact path, kept samples with a component outside 1e-5 (must be 0), kept samples for which the
 * reference returns its degenerate (0, 0, 1)}.  DJB_ERR_INVALID_ARGUMENT when brdf / params are outside the sampler's domain. */
djb_status djb_selftest_contract_sample(djb_ctx *, const djb_brdf *, const djb_params *params, int64_t n, uint32_t seed, int family,
                                        float *max_abs2, unsigned long long *counters4);
/* directed search against the same sampler: n candidate samples (u1[n], u2[n], o: device arrays, updated IN PLACE) hill-climb
 * over the bit patterns of their five inputs (iters moves each, +-2^e units in the last place of one input) to maximise the
 * component difference between the contract path and the bit-exact code, in units of the contract (1e-5 max(1, |o|)); a sample
 * the fast path hands to the exact path scores 0.  best[n] (device) = the score each candidate reached (< 1 = inside the
 * contract); counters3 = {evaluations, evaluated samples the fast path kept OUTSIDE the contract (must be 0), accepted moves}.
 * tools/contract_sample_attack.py restarts it from several input families. */
djb_status djb_contract_sample_attack(djb_ctx *, const djb_brdf *, const djb_params *params, int64_t n, float *u1, float *u2,
                                      const djb_vec3_view *o, int iters, uint32_t seed, float *best, unsigned long long *counters3);
/* the kernels' restatements of the host libm functions the reference calls (glibc 2.35: double exp / pow / atan2 / sin / cos / tan / acos,
 * float logf / expf / powf -- dj_brdf.h:659, 685, 695, 1634, 1868, 1917, 1935, 3419, 3431, 3612), evaluated on the GPU for
 * host arrays: fn 0 exp(x), 1 pow(x, y), 2 logf(x), 3 expf(x), 4 powf(x, y) (float functions on the values cast
 * to float), 5 atan2(x, y) (x = the first, "y" argument of atan2), 6 / 7 the kernels' float(atan2(x, y)) and
 * float(float(180 / pi) * atan2(x, y)) of float arguments (device libm, glibc's algorithm next to a float rounding
 * boundary), 8 sin(x), 9 cos(x), 10 tan(x), 11 acos(x) (double).  The test-suite compares `out` with the libm of the host, bit for bit. */
djb_status djb_selftest_libm(djb_ctx *, int fn, int64_t n, const double *x, const double *y, double *out);

/* exhaustive check of the sites of the fp64 trig family.  Float sites (0 .. DJB_TRIG_SITES - 1, the TRIG_* enum of
 * csrc/djb_device.hpp): every place where the path rounds the double libm result of ONE float argument to float
 * (float(cos(double x)), float(2 acos(x) / pi), ..., utia's grid cells floor(x / 15.0)), i.e. a float -> float map
 * with 2^32 inputs.  Double sites
 * (DJB_TRIG_DOUBLE + 0 cos, 1 sin, 2 tan, 3 acos): the double result itself, for the places that keep it.
 * Evaluates site fn on the GPU for the `count` floats whose bit patterns start at first_bits
 * (count <= 2^32 - first_bits), evaluates site host_fn (= fn, except for the test-suite's negative control) with the
 * library's host instantiation (the host's libm: glibc, what the reference links) on `threads` threads (0 = all
 * cores), and returns the number of inputs whose results differ (NaN == NaN); the first `cap` of them go to bad3 as
 * {input bits, device bits, host bits} (double sites: {input bits, |difference| in units of the last place, 0}).
 * tools/exhaustive_trig.py sweeps all 2^32 inputs of every site. */
#define DJB_TRIG_SITES 13
#define DJB_TRIG_DOUBLE 16
#define DJB_TRIG_DOUBLE_SITES 4
djb_status djb_selftest_trig_sweep(djb_ctx *, int fn, int host_fn, uint32_t first_bits, int64_t count, int threads,
                                   unsigned long long *n_bad, uint32_t *bad3, int cap);

/* microfacet::params -> private members (host side, no GPU work)       dj_brdf.h:1355-1506 */
djb_status djb_params_resolve(const djb_params *params, djb_params_resolved *out);

/* ---------------------------------------------------------------- tabular accessors / fits */
enum { DJB_TAB_P22 = 0, DJB_TAB_SIGMA = 1, DJB_TAB_CDF = 2, DJB_TAB_QF = 3, DJB_TAB_FRESNEL = 4 };
/* tabular::get_p22v / get_sigmav / get_cdfv / get_qfv, and fresnel::spline::get_points of
 * tabular::get_fresnel() (3 floats per point).  out may be NULL to query *count only.
 *                                                                     dj_brdf.h:404-407, 203 */
djb_status djb_tabular_get(const djb_brdf *tab, int which, float *out, int *count);
/* tabular::fit_beckmann_parameters / fit_ggx_parameters -> isotropic alpha
 *                                                                     dj_brdf.h:3133-3184 */
djb_status djb_tabular_fit(const djb_brdf *tab, float *alpha_beckmann, float *alpha_ggx);

/* tabular_anisotropic accessors: get_p22v / get_sigmav with their (elev, azim) counts
 * (dj_brdf.h:447-448), the sampling tables behind pdf1/cdf1/qf1/pdf2/cdf2/qf2, the Fresnel spline
 * points; grids are elev x azim floats, element (i_elev, j_azim) at [i + elev*j].           */
enum { DJB_ATAB_P22 = 0, DJB_ATAB_SIGMA = 1, DJB_ATAB_PDF1 = 2, DJB_ATAB_CDF1 = 3, DJB_ATAB_QF1 = 4,
       DJB_ATAB_PDF2 = 5, DJB_ATAB_CDF2 = 6, DJB_ATAB_QF2 = 7, DJB_ATAB_FRESNEL = 8,
       DJB_ATAB_QF2_ENTRIES = 9 /* count only: entries of the reference's m_qf2 (< elev*azim if rows came up short) */ };
djb_status djb_tabular_anisotropic_get(const djb_brdf *tab, int which, float *out, int *count,
                                       int *elev_cnt, int *azim_cnt);
/* tabular_anisotropic::fit_beckmann_parameters / fit_ggx_parameters -> params::pdfparams(alphax,
 * alphay, rho, mux, muy) (the GGX rho is the reference's "TODO": 0).          dj_brdf.h:3186-3307 */
djb_status djb_tabular_anisotropic_fit(const djb_brdf *tab, djb_params *beckmann, djb_params *ggx);

/* ---------------------------------------------------------------- batch fitter
 * What examples/merl_params.cpp:53-67 does per file, for n_materials tables at once:
 * tabular(merl, res, shadow) + both fits.  tables[m] points to 3*n doubles (MERL file
 * payload, host memory).  Outputs (host): alpha arrays [n_materials]; optional per-material
 * tables p22/sigma/cdf/qf [n_materials][res] and fresnel [n_materials][res][3] (may be NULL). */
djb_status djb_fit_merl_batch(djb_ctx *, int n_materials, const double *const *tables,
                              int res, int shadow, float *alpha_beckmann, float *alpha_ggx,
                              float *p22, float *sigma, float *cdf, float *qf, float *fresnel);

/* Same fit for n_materials BRDF objects already resident in HBM (all of one kind, e.g. MERL
 * tables created with djb_brdf_create_merl_*): no host->device traffic inside the call.    */
djb_status djb_fit_brdf_batch(djb_ctx *, int n_materials, const djb_brdf *const *srcs,
                              int res, int shadow, float *alpha_beckmann, float *alpha_ggx,
                              float *p22, float *sigma, float *cdf, float *qf, float *fresnel);

/* End to end: what examples/merl_params.cpp:53-67 does per file, for a list of MERL files: the table indices a
 * tabular(merl, res) fit reads are computed on the GPU, worker threads gather just those entries from the (mapped)
 * files, ~97 KB per material travel to HBM, then ONE fit launch for the whole batch.  With DJB_OPT_FIT_FILES_DENSE
 * the whole tables travel instead (reader threads -> pinned 4 MiB chunk ring -> async upload + conversion kernel).
 * Errors carry djb::merl's messages (dj_brdf.h:970-982), the lowest-indexed bad file wins.  reader_threads <= 0
 * picks a default (32 on a large host).  timing (optional, 4 doubles): total seconds, seconds until every table was
 * resident in HBM, seconds of the fit, bytes read from the files.  The mapped files are released by a helper thread
 * after the call has its alphas (unmapping inside the gather loop serialised the readers).  The context keeps what a call
 * sets up -- its reader threads (parked between calls), a pinned staging buffer with its HBM twin (6.6 MB per 100 files) and
 * the slot plan of the resolution -- until djb_ctx_destroy: the first call of a shape costs about twice the later ones.
 * The default form reads
 * through mmap: a file that is TRUNCATED by another process while the call runs raises SIGBUS like any mapped read
 * (size and header are checked before mapping); callers that cannot rule that out -- network file systems with
 * concurrent writers -- should set DJB_OPT_FIT_FILES_DENSE, whose pread() path returns "Reading <file> failed".  */
djb_status djb_fit_merl_files(djb_ctx *, int n_files, const char *const *paths, int res, int shadow,
                              int reader_threads, float *alpha_beckmann, float *alpha_ggx,
                              double *timing);
/* The same job over SEVERAL contexts -- one per GPU of a node (BASELINE configs[4]; the loop of examples/merl_params.cpp:53-69 is what it
 * stands for).  File k belongs to context k mod n_ctx; every context's share is one djb_fit_merl_files on a host thread inside the
 * library (the calling thread takes context 0); alpha_*[k] are in INPUT order.  The fits are independent: no exchange between
 * contexts, no collective.  Contexts must be distinct; CPU and GPU contexts may be mixed.  On failure nothing is written and the
 * error is that of the lowest-indexed bad file over all shares (the reference's loop stops at its first bad file).  timing
 * (optional): 4 doubles PER CONTEXT, each as in djb_fit_merl_files.                                                             */
djb_status djb_fit_merl_files_multi(djb_ctx *const *ctxs, int n_ctx, int n_files, const char *const *paths,
                                    int res, int shadow, int reader_threads, float *alpha_beckmann,
                                    float *alpha_ggx, double *timing);

/* ---------------------------------------------------------------- synthetic workloads
 * (not reference behaviour: the reference has no RNG).  Bit-identical to
 * dj_brdf_amd/synth.py.  Device pointers only. */
djb_status djb_gen_directions(djb_ctx *, int64_t n, uint32_t seed, uint64_t start,
                              const djb_vec3_view *out);
djb_status djb_gen_uniforms(djb_ctx *, int64_t n, uint32_t seed, uint64_t start, float *out);
/* bins x bins histogram of (x, y) in [-1,1]^2 (LDS atomics, one global flush per workgroup) */
djb_status djb_histogram_xy(djb_ctx *, int64_t n, const djb_vec3_view *v, int bins,
                            unsigned long long *counts /* device, bins*bins */);

#ifdef __cplusplus
}
#endif
#endif /* DJB_HIP_H */
