/* A plain C99 caller of the C ABI (include/djb_hip.h) -- the shape a cgo / JNI / N-API binding takes: opaque handles, views of
 * float arrays, status codes, djb_last_error().  No C++, no HIP headers.
 *     c_abi_demo [cpu|gpu]        (default: the GPU if there is one, else the host path)
 * Evaluates the pair of SURVEY.md section 8-N's known answers -- i = (0.3, 0.2, .), o = (-0.4, 0.1, .), z = sqrtf(1 - x^2 - y^2) --
 * on ggx isotropic alpha = 0.3 (reference: eval 0.621380985, pdf 0.581518769; sample(0.25, 0.75) = (0.657071352, 0.080957301,
 * 0.749468625)) and prints the results with %.9g, as a caller of djb::ggx::eval / pdf / sample would (dj_brdf.h:77-97). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "djb_hip.h"

/* a BRDF the CALLER defines (the reference's extension point: a class derived from djb::brdf, dj_brdf.h:74-109 -- here a C
 * function, what a cgo / JNI callback would be).  It restates djb::lambert::eval (dj_brdf.h:861-868: reflectance / M_PI with
 * vec3 / float_t = (1.0 / b) * a, b = float(M_PI)) so that the fit of the callback can be held against the fit of the library's
 * own Lambertian, table by table */
static void my_brdf(const float *i, const float *o, float *rgb)
{
	const float pi_f = (float)3.14159265358979323846;
	const float inv_pi = (float)(1.0 / (double)pi_f);
	(void)i; (void)o;
	rgb[0] = inv_pi * 1.0f; rgb[1] = inv_pi * 1.0f; rgb[2] = inv_pi * 1.0f;
}

/* an NDF the CALLER defines (the reference's third extension point: a class derived from djb::radial, dj_brdf.h:301-324 -- here C
 * callbacks): GGX's own radial functions restated (dj_brdf.h:2056-2076), sampled with the "nmap" scheme.  eval of such an object must
 * equal the library's ggx bit for bit: everything around the NDF is the same per-unit code */
static int   cb_no_smith(void *u) { (void)u; return 0; }
static float cb_p22_radial(void *u, float r_sqr) { const float t = 1.0f + r_sqr; (void)u; return (float)(1.0 / (3.14159265358979323846 * (double)t * (double)t)); }
static float cb_sigma_std_radial(void *u, float c) { (void)u; return (float)((1.0 + (double)c) / 2.0); }
static float cb_cdf_radial(void *u, float r) { const float t = r * r; (void)u; return (float)((double)t / (1.0 + (double)t)); }
static float cb_qf_radial(void *u, float x) { (void)u; return (float)sqrt((double)x / (1.0 - (double)x)); }

#define CHECK(call) do { djb_status s_ = (call); if (s_ != DJB_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, (int)s_, djb_last_error()); return 1; } } while (0)

int main(int argc, char **argv)
{
	int device = DJB_DEVICE_CPU, count = 0;
	if (argc > 1 && !strcmp(argv[1], "gpu")) device = 0;
	else if (argc <= 1 && djb_device_count(&count) == DJB_OK && count > 0) device = 0;
	djb_ctx *ctx = NULL;
	djb_brdf *ggx = NULL;
	CHECK(djb_ctx_create(device, &ctx));
	djb_fresnel_desc ideal;
	memset(&ideal, 0, sizeof ideal);
	ideal.kind = DJB_FRESNEL_IDEAL;
	CHECK(djb_brdf_create_ggx(ctx, &ideal, /*shadow*/ 1, &ggx));

	/* an array of djb::vec3 (stride 3), host memory: what a drop-in caller already holds */
	float i[3] = { 0.3f, 0.2f, 0.0f }, o[3] = { -0.4f, 0.1f, 0.0f }, fr[3], in[3], pdf = 0.0f;
	i[2] = sqrtf(1.0f - i[0] * i[0] - i[1] * i[1]);
	o[2] = sqrtf(1.0f - o[0] * o[0] - o[1] * o[1]);
	djb_vec3_view vi = { i, i + 1, i + 2, 3 }, vo = { o, o + 1, o + 2, 3 }, vfr = { fr, fr + 1, fr + 2, 3 }, vin = { in, in + 1, in + 2, 3 };
	djb_params p;
	memset(&p, 0, sizeof p);
	p.kind = DJB_PARAMS_ELLIPTIC; p.v[0] = 0.3f; p.v[1] = 0.3f; p.v[2] = 0.0f;      /* params::isotropic(0.3) */
	CHECK(djb_eval_pdf_batch(ctx, ggx, 1, &vi, &vo, &p, /*want_cos*/ 0, &vfr, &pdf, DJB_MEM_HOST));
	const float u1 = 0.25f, u2 = 0.75f;
	CHECK(djb_sample_batch(ctx, ggx, 1, &u1, &u2, &vo, &p, &vin, DJB_MEM_HOST));
	printf("device %s\n", device == DJB_DEVICE_CPU ? "cpu" : "gpu");
	printf("eval %.9g %.9g %.9g\npdf %.9g\nsample %.9g %.9g %.9g\n", fr[0], fr[1], fr[2], pdf, in[0], in[1], in[2]);
	/* fit a caller-defined BRDF: the library says where tabular's constructor evaluates its source (dj_brdf.h:2494, 2610), the caller
	 * evaluates there, the fit runs on the samples; held against tabular(lambert) on the same context: every table identical */
	{
		const int res = 32;
		int64_t nq = 0, k, evaluated = 0;
		CHECK(djb_fit_query_dirs(res, 0, NULL, NULL, &nq));
		float *qi = (float *)malloc(sizeof(float) * 3 * (size_t)nq), *qo = (float *)malloc(sizeof(float) * 3 * (size_t)nq);
		float *rgb = (float *)calloc(3 * (size_t)nq, sizeof(float));
		djb_vec3_view vqi = { qi, qi + 1, qi + 2, 3 }, vqo = { qo, qo + 1, qo + 2, 3 };
		CHECK(djb_fit_query_dirs(res, nq, &vqi, &vqo, NULL));
		for (k = 0; k < nq; ++k)
			if (qo[3 * k] == qo[3 * k]) { my_brdf(qi + 3 * k, qo + 3 * k, rgb + 3 * k); ++evaluated; }     /* NaN: a pair the reference skips */
		djb_brdf *tab = NULL, *lam = NULL, *tab_lam = NULL;
		CHECK(djb_brdf_create_tabular_from_samples(ctx, res, 1, rgb, nq, &tab));
		CHECK(djb_brdf_create_lambert(ctx, &lam));
		CHECK(djb_brdf_create_tabular(ctx, lam, res, 1, &tab_lam));
		float a_user = 0, a_lam = 0, t_user[3 * 32], t_lam[3 * 32];
		int which, same = 1;
		CHECK(djb_tabular_fit(tab, NULL, &a_user));
		CHECK(djb_tabular_fit(tab_lam, NULL, &a_lam));
		for (which = DJB_TAB_P22; which <= DJB_TAB_FRESNEL; ++which) {
			int n_user = 0, n_lam = 0;
			memset(t_user, 0, sizeof t_user); memset(t_lam, 0, sizeof t_lam);
			CHECK(djb_tabular_get(tab, which, NULL, &n_user)); CHECK(djb_tabular_get(tab_lam, which, NULL, &n_lam));
			CHECK(djb_tabular_get(tab, which, t_user, NULL)); CHECK(djb_tabular_get(tab_lam, which, t_lam, NULL));
			same = same && n_user == n_lam && memcmp(t_user, t_lam, sizeof t_user) == 0;
		}
		printf("user-defined fit: %lld of %lld query slots evaluated, alpha_ggx %.3f (tabular(lambert): %.3f), tables %s\n", (long long)evaluated,
		       (long long)nq, a_user, a_lam, same ? "identical" : "DIFFERENT");
		CHECK(djb_brdf_destroy(tab)); CHECK(djb_brdf_destroy(tab_lam)); CHECK(djb_brdf_destroy(lam));
		free(qi); free(qo); free(rgb);
	}
	/* a caller-defined NDF: host code, so the object lives on a CPU context whatever `ctx` is */
	{
		djb_ctx *host = NULL;
		djb_brdf *mine = NULL, *ggx_host = NULL;
		djb_user_ndf ndf;
		float fr_mine[3], fr_ggx[3];
		djb_vec3_view vm = { fr_mine, fr_mine + 1, fr_mine + 2, 3 }, vg = { fr_ggx, fr_ggx + 1, fr_ggx + 2, 3 };
		memset(&ndf, 0, sizeof ndf);
		ndf.supports_smith_vndf_sampling = cb_no_smith; ndf.p22_radial = cb_p22_radial; ndf.sigma_std_radial = cb_sigma_std_radial;
		ndf.cdf_radial = cb_cdf_radial; ndf.qf_radial = cb_qf_radial;
		CHECK(djb_ctx_create(DJB_DEVICE_CPU, &host));
		CHECK(djb_brdf_create_user_microfacet(host, &ndf, &ideal, 1, &mine));
		CHECK(djb_brdf_create_ggx(host, &ideal, 1, &ggx_host));
		CHECK(djb_eval_batch(host, mine, 1, &vi, &vo, &p, &vm, DJB_MEM_HOST));
		CHECK(djb_eval_batch(host, ggx_host, 1, &vi, &vo, &p, &vg, DJB_MEM_HOST));
		printf("user-defined NDF (GGX restated as callbacks): eval %.9g, %s the library's ggx\n", fr_mine[0],
		       memcmp(fr_mine, fr_ggx, sizeof fr_mine) == 0 ? "identical to" : "DIFFERENT from");
		CHECK(djb_brdf_destroy(mine)); CHECK(djb_brdf_destroy(ggx_host)); CHECK(djb_ctx_destroy(host));
	}
	CHECK(djb_brdf_destroy(ggx));
	CHECK(djb_ctx_destroy(ctx));
	return 0;
}
