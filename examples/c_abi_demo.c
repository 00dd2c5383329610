/* A plain C99 caller of the C ABI (include/djb_hip.h) -- the shape a cgo / JNI / N-API binding takes: opaque handles, views of
 * float arrays, status codes, djb_last_error().  No C++, no HIP headers.
 *     c_abi_demo [cpu|gpu]        (default: the GPU if there is one, else the host path)
 * Evaluates the pair of SURVEY.md section 8-N's known answers -- i = (0.3, 0.2, .), o = (-0.4, 0.1, .), z = sqrtf(1 - x^2 - y^2) --
 * on ggx isotropic alpha = 0.3 (reference: eval 0.621380985, pdf 0.581518769; sample(0.25, 0.75) = (0.657071352, 0.080957301,
 * 0.749468625)) and prints the results with %.9g, as a caller of djb::ggx::eval / pdf / sample would (dj_brdf.h:77-97). */
#include <math.h>
#include <stdio.h>
#include <string.h>
#include "djb_hip.h"

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
	CHECK(djb_brdf_destroy(ggx));
	CHECK(djb_ctx_destroy(ctx));
	return 0;
}
