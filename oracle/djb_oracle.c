/* oracle/djb_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see djb_oracle.h).
 *
 * Plain-C restatement of the dj_brdf hot path.  "hdr:N" = /root/reference/dj_brdf.h line N.
 * Conventions: F(x) rounds a double expression to float exactly where the
 * reference assigns/returns a float_t; everything inside D(...) is double.
 */
#include "djb_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define F(x) ((float)(x))
#define D(x) ((double)(x))
#define O_PI 3.14159265358979323846 /* M_PI */
#define O_EPSILON ((float)1e-4)     /* DJB_EPSILON, hdr:49-51 */

static __thread char g_err[512];
const char *o_last_error(void) { return g_err; }
static void set_err(const char *fmt, ...)
{
	va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
}

/* ------------------------------------------------------------------ L0 math (hdr:574-765) */
static float fminf_(float a, float b) { return a < b ? a : b; }   /* djb::min hdr:574 */
static float fmaxf_(float a, float b) { return a > b ? a : b; }   /* djb::max hdr:575 */
static float satf_(float x) { return fminf_(1.0f, fmaxf_(0.0f, x)); } /* hdr:576 */

static o_vec3 v3(float x, float y, float z) { o_vec3 v = { x, y, z }; return v; }
static o_vec3 v3_add(o_vec3 a, o_vec3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
static o_vec3 v3_sub(o_vec3 a, o_vec3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
static o_vec3 v3_mul(o_vec3 a, o_vec3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
static o_vec3 v3_scale(float s, o_vec3 a) { return v3(s * a.x, s * a.y, s * a.z); }
/* vec3 / float_t = (1.0 / b) * a : double reciprocal rounded to float, hdr:601 */
static o_vec3 v3_div(o_vec3 a, float b) { return v3_scale(F(1.0 / D(b)), a); }
static float v3_dot(o_vec3 a, o_vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; } /* hdr:618 */
static o_vec3 v3_cross(o_vec3 a, o_vec3 b) /* hdr:623 */
{
	return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static float inversesqrt_(float x) { return F(1.0 / sqrt(D(x))); } /* hdr:612 */
static o_vec3 v3_normalize(o_vec3 v) { return v3_scale(inversesqrt_(v3_dot(v, v)), v); } /* hdr:630 */
static float v3_intensity(o_vec3 v) { return 0.2126f * v.x + 0.7152f * v.y + 0.0722f * v.z; } /* hdr:69 */

/* vec3(theta, phi) ctor, hdr:589-595 */
static o_vec3 v3_from_angles(float theta, float phi)
{
	float s = F(sin(D(theta)));
	return v3(F(D(s) * cos(D(phi))), F(D(s) * sin(D(phi))), F(cos(D(theta))));
}

/* hdr:650-661 */
static void xyz_to_theta_phi(o_vec3 p, float *theta, float *phi)
{
	if (D(p.z) > 0.99999) { *theta = 0.0f; *phi = 0.0f; }
	else if (D(p.z) < -0.99999) { *theta = F(O_PI); *phi = 0.0f; }
	else { *theta = F(acos(D(p.z))); *phi = F(atan2(D(p.y), D(p.x))); }
}

/* A&S 7.1.26, hdr:667-688 */
static float erf_(float x)
{
	const float a1 = 0.254829592f, a2 = -0.284496736f, a3 = 1.421413741f,
	            a4 = -1.453152027f, a5 = 1.061405429f, p = 0.3275911f;
	int sign = x < 0 ? -1 : 1;
	x = F(fabs(D(x)));
	float t = F(1.0 / (1.0 + D(p * x)));
	float poly = ((((a5 * t + a4) * t) + a3) * t + a2) * t + a1;
	float y = F(1.0 - D(poly * t) * exp(D(-x * x)));
	return (float)sign * y;
}

/* Giles single-precision erfinv, hdr:691-721 */
static float erfinv_(float u)
{
	float w = -logf((1.0f - u) * (1.0f + u)), p;
	if (w < 5.0f) {
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
	} else {
		w = F(sqrt(D(w)) - D(3.0f));
		p = -0.000200214257f;
		p = 0.000100950558f + p * w;
		p = 0.00134934322f + p * w;
		p = -0.00367342844f + p * w;
		p = 0.00573950773f + p * w;
		p = -0.0076224613f + p * w;
		p = 0.00943887047f + p * w;
		p = 1.00167406f + p * w;
		p = 2.83297682f + p * w;
	}
	return p * u;
}

/* Cline's concentric map, hdr:726-747 */
static void uniform_to_concentric(float u1, float u2, float *x, float *y)
{
	float r1 = F(2.0 * D(u1) - 1.0), r2 = F(2.0 * D(u2) - 1.0), phi, r;
	if (r1 == 0 && r2 == 0) { r = phi = 0; }
	else if (r1 * r1 > r2 * r2) { r = r1; phi = F((O_PI / 4.0) * D(r2 / r1)); }
	else { r = r2; phi = F((O_PI / 2.0) - D(r1 / r2) * (O_PI / 4.0)); }
	*x = F(D(r) * cos(D(phi)));
	*y = F(D(r) * sin(D(phi)));
}

/* Rodrigues, hdr:754-765 */
static o_vec3 rotate_vector(o_vec3 x, o_vec3 axis, float angle)
{
	float c = F(cos(D(angle))), s = F(sin(D(angle)));
	o_vec3 out = v3_scale(c, x);
	float t1 = v3_dot(axis, x);
	float t2 = F(D(t1) * (1.0 - D(c)));
	out = v3_add(out, v3_scale(t2, axis));
	out = v3_add(out, v3_scale(s, v3_cross(axis, x)));
	return out;
}

/* hdr:771-781 */
static void io_to_hd(o_vec3 i, o_vec3 o, o_vec3 *h, o_vec3 *d)
{
	float th, ph;
	*h = v3_normalize(v3_add(i, o));
	xyz_to_theta_phi(*h, &th, &ph);
	o_vec3 tmp = rotate_vector(i, v3(0, 0, 1), -ph);
	*d = v3_normalize(rotate_vector(tmp, v3(0, 1, 0), -th));
}

/* hdr:783-793 */
static void hd_to_io(o_vec3 h, o_vec3 d, o_vec3 *i, o_vec3 *o)
{
	float th, ph;
	xyz_to_theta_phi(h, &th, &ph);
	o_vec3 tmp = rotate_vector(d, v3(0, 1, 0), th);
	*i = v3_normalize(rotate_vector(tmp, v3(0, 0, 1), ph));
	*o = v3_normalize(v3_sub(v3_scale(F(2.0 * D(v3_dot(*i, h))), h), *i));
}

/* ------------------------------------------------------------------ spline (hdr:1181-1249) */
static int uwrap_edge(int i, int edge) { return i >= edge ? edge - 1 : (i < 0 ? 0 : i); }

static void spline_locate(int edge, float u, int *i1, int *i2, float *frac)
{
	double ip;
	*frac = F(modf(D(u * (float)edge - u), &ip));
	*i1 = uwrap_edge((int)ip, edge);
	*i2 = uwrap_edge((int)ip + 1, edge);
}
static float spline_eval_f(const float *pts, int n, float u)
{
	int i1, i2; float fr;
	spline_locate(n, u, &i1, &i2, &fr);
	return pts[i1] + fr * (pts[i2] - pts[i1]);
}
static o_vec3 spline_eval_v3(const o_vec3 *pts, int n, float u)
{
	int i1, i2; float fr;
	spline_locate(n, u, &i1, &i2, &fr);
	return v3_add(pts[i1], v3_scale(fr, v3_sub(pts[i2], pts[i1])));
}

/* ------------------------------------------------------------------ BRDF object */
typedef struct {
	int kind;
	o_vec3 a, b;          /* ior | f0 | (f0, f1) */
	o_vec3 *pts; int npts;
} o_fresnel;

struct o_brdf {
	int kind;
	int shadow;
	o_fresnel fresnel;
	/* tabular */
	float *p22, *sigma, *cdf, *qf;
	int n_p22, n_sigma, n_cdf, n_qf;
	/* merl / utia */
	double model[36];      /* sgd / abc parameter row */
	/* tabular_anisotropic: grids are elev x azim, element (i_elev, j_azim) at [i + elev*j] */
	int elev, azim;
	float *a_p22, *a_sigma, *a_pdf1, *a_cdf1, *a_qf1, *a_pdf2, *a_cdf2, *a_qf2;
	int n_a_pdf1, n_a_cdf1, n_a_qf1, n_a_pdf2, n_a_cdf2, n_a_qf2;
	int n_a_qf2_ref;   /* entries the reference's m_qf2 holds (== elev*azim unless rows came up short) */
	double *samples;
	int64_t n_samples;
};

/* ------------------------------------------------------------------ Fresnel (hdr:1253-1346) */
static float unpolarized_eval1(float c, float n) /* hdr:1292-1303 */
{
	float g = F(sqrt(D(n * n + c * c) - 1.0));
	float t1 = F(D(c * (g + c)) - 1.0);
	float t2 = F(D(c * (g - c)) + 1.0);
	float t3 = (t1 * t1) / (t2 * t2);
	float t4 = ((g - c) * (g - c)) / ((g + c) * (g + c));
	return F((0.5 * D(t4)) * (1.0 + D(t3)));
}

static o_vec3 fresnel_eval(const o_fresnel *f, float c)
{
	switch (f->kind) {
	case O_FRESNEL_UNPOLARIZED:
		return v3(unpolarized_eval1(c, f->a.x), unpolarized_eval1(c, f->a.y),
		          unpolarized_eval1(c, f->a.z));
	case O_FRESNEL_SCHLICK: { /* hdr:1320-1328 */
		float c1 = F(1.0 - D(c)), c2 = c1 * c1, c5 = c2 * c2 * c1;
		return v3_add(f->a, v3_scale(c5, v3_sub(v3(1, 1, 1), f->a)));
	}
	case O_FRESNEL_SGD: { /* hdr:1330-1336 */
		float pw = F(pow(1.0 - D(c), 5.0));
		return v3_add(v3_sub(f->a, v3_scale(c, f->b)),
		              v3_scale(pw, v3_sub(v3(1, 1, 1), f->a)));
	}
	case O_FRESNEL_SPLINE: { /* hdr:1338-1344 */
		float u = F(2.0 * acos(D(c)) / O_PI);
		return spline_eval_v3(f->pts, f->npts, u);
	}
	case O_FRESNEL_CUSTOM: { /* ref_shim.cpp user_lazanyi::eval: a = f0, b.x = the correction's weight */
		double m = 1.0 - D(c);
		float p5 = F(m * m * m * m * m);
		float p7 = F(D(p5) * m * m);
		float t = f->b.x * c * p7;
		return v3_sub(v3_add(f->a, v3_scale(p5, v3_sub(v3(1, 1, 1), f->a))), v3(t, t, t));
	}
	default: return v3(1, 1, 1);
	}
}

/* ------------------------------------------------------------------ params (hdr:1355-1506) */
static void params_set_location(o_params *p, float tx, float ty) /* hdr:1437 */
{
	p->tx = tx; p->ty = ty;
	p->n = v3_normalize(v3(-tx, -ty, 1.0f));
}

static void params_set_ellipse(o_params *p, float a1, float a2, float phi_a) /* hdr:1451, 1356-1371 */
{
	p->a1 = a1; p->a2 = a2; p->phi_a = phi_a;
	float c = F(cos(D(phi_a))), s = F(sin(D(phi_a)));
	float c2 = F(2.0 * D(c) * D(c) - D(1.0f));
	float a1s = a1 * a1, a2s = a2 * a2, t1 = a1s + a2s, t2 = a1s - a2s;
	p->ax = F(sqrt(0.5 * D(t1 + t2 * c2)));
	p->ay = F(sqrt(0.5 * D(t1 - t2 * c2)));
	p->rho = (a2s - a1s) * c * s / (p->ax * p->ay);
	p->sqrt_1mrho2 = F(sqrt(1.0 - D(p->rho * p->rho)));
}

static void params_set_pdfparams(o_params *p, float ax, float ay, float rho, float tx, float ty)
{ /* hdr:1461-1474, 1378-1393 */
	p->ax = ax; p->ay = ay; p->rho = rho;
	p->sqrt_1mrho2 = F(sqrt(1.0 - D(rho * rho)));
	float axs = ax * ax, ays = ay * ay;
	float cov = F(D(rho * ax * ay) * 2.0);
	float t1 = axs + ays, t2 = axs - ays;
	float t3 = F(sqrt(D(t2 * t2 + cov * cov)));
	p->a1 = F(sqrt(0.5 * D(t1 + t3)));
	p->a2 = F(sqrt(0.5 * D(t1 - t3)));
	p->phi_a = (D(cov) != 0.0) ? F(atan(D((axs - ays - t3) / cov))) : 0.0f;
	params_set_location(p, tx, ty);
}

static o_params params_elliptic(float a1, float a2, float phi_a)
{
	o_params p;
	params_set_ellipse(&p, a1, a2, phi_a);
	params_set_location(&p, 0, 0);
	return p;
}
static o_params params_standard(void) { return params_elliptic(1, 1, 0); }

static o_params params_from_desc(const o_param_desc *pd)
{
	if (pd && pd->kind == 3) {   /* lambert::params(reflectance), hdr:114-119: carried in the n slot, a1 = -1 marks it */
		o_params p = params_elliptic(1, 1, 0);
		p.n = v3(pd->v[0], pd->v[1], pd->v[2]);
		p.a1 = -1.0f;
		return p;
	}
	if (pd && pd->kind == 1) return params_elliptic(pd->v[0], pd->v[1], pd->v[2]);
	if (pd && pd->kind == 2) {
		o_params p;
		params_set_pdfparams(&p, pd->v[0], pd->v[1], pd->v[2], pd->v[3], pd->v[4]);
		return p;
	}
	return params_standard();
}

/* ------------------------------------------------------------------ radial queries */
static float tab_p22_radial(const o_brdf *b, float r_sqr) /* hdr:2151-2156 */
{
	float r = F(sqrt(D(r_sqr)));
	float u = F(sqrt(D(2.0f) * atan(D(r)) / D(F(O_PI))));
	return spline_eval_f(b->p22, b->n_p22, u);
}
static float tab_sigma_std_radial(const o_brdf *b, float c) /* hdr:2158-2162 */
{
	float u = F(D(2.0f) * acos(D(c)) / D(F(O_PI)));
	return spline_eval_f(b->sigma, b->n_sigma, u);
}
static float tab_cdf_radial(const o_brdf *b, float r) /* hdr:2164-2169 */
{
	float u = F(atan(D(r)) * D(2.0f) / D(F(O_PI)));
	if (u < 0.0f) u = 0.0f;
	return spline_eval_f(b->cdf, b->n_cdf, F(sqrt(D(u))));
}
static float tab_qf_radial(const o_brdf *b, float u) /* hdr:2171-2176 */
{
	float qf = spline_eval_f(b->qf, b->n_qf, u);
	return F(tan(D(qf * F(O_PI) / 2.0f)));
}

static float p22_radial(const o_brdf *b, float r_sqr)
{
	switch (b->kind) {
	case O_BRDF_BECKMANN: return F(exp(D(-r_sqr)) / O_PI);                    /* hdr:1866 */
	case O_BRDF_GGX: { float t = F(1.0 + D(r_sqr)); return F(1.0 / (O_PI * D(t) * D(t))); } /* hdr:2056 */
	default: return tab_p22_radial(b, r_sqr);
	}
}

static float sigma_std_radial(const o_brdf *b, float c)
{
	switch (b->kind) {
	case O_BRDF_BECKMANN: { /* hdr:1871-1879 */
		if (D(c) == 1.0) return 1.0f;
		float s = F(sqrt(1.0 - D(c * c)));
		float nu = c / s;
		float tmp = F(exp(D(-nu * nu)) * D(inversesqrt_(F(O_PI))));
		return F((D(c) * (1.0 + D(erf_(nu))) + D(s * tmp)) / 2.0);
	}
	case O_BRDF_GGX: return F((1.0 + D(c)) / 2.0);                            /* hdr:2062 */
	default: return tab_sigma_std_radial(b, c);
	}
}

static float cdf_radial(const o_brdf *b, float r)
{
	switch (b->kind) {
	case O_BRDF_BECKMANN: return F(1.0 - exp(D(-r * r)));                     /* hdr:1881 */
	case O_BRDF_GGX: { float t = r * r; return F(D(t) / (1.0 + D(t))); }      /* hdr:2067 */
	default: return tab_cdf_radial(b, r);
	}
}

static float qf_radial(const o_brdf *b, float u)
{
	switch (b->kind) {
	case O_BRDF_BECKMANN: return F(sqrt(-log(1.0 - D(u))));                   /* hdr:1886 */
	case O_BRDF_GGX: return F(sqrt(D(u) / (1.0 - D(u))));                     /* hdr:2073 */
	default: return tab_qf_radial(b, u);
	}
}

static float beckmann_qf1(float u) { return erfinv_(F(2.0 * D(u) - 1.0)); } /* hdr:1891 */

/* hdr:1897-1952 */
static float beckmann_qf2_radial(float u, float cos_k, float sin_k)
{
	const float sqrt_pi_inv = F(1. / sqrt(O_PI));
	float cot_k = cos_k / sin_k, tan_k = sin_k / cos_k;
	float a = -1, c = erf_(cot_k);
	u = fmaxf_(u, 1e-6f);
	float fit = 1 + cos_k * (-0.876f + cos_k * (0.4265f - 0.0594f * cos_k));
	float b = c - (1 + c) * powf(1 - u, fit);
	float normalization = F(1 / (D(1 + c) + D(sqrt_pi_inv * tan_k) * exp(D(-cot_k * cot_k))));
	int it = 0;
	while (++it < 10) {
		if (!(b >= a && b <= c)) b = 0.5f * (a + c);
		float inv_erf = erfinv_(b);
		float value = normalization * (1 + b + sqrt_pi_inv * tan_k * expf(-inv_erf * inv_erf)) - u;
		float derivative = normalization * (1 - inv_erf * tan_k);
		if (fabs(D(value)) < D(1e-5f)) break;
		if (value > 0) c = b; else a = b;
		b -= value / derivative;
	}
	return erfinv_(fmaxf_(-0.9999f, b));
}

/* hdr:2078-2087 */
static float ggx_qf1(float u)
{
	if (D(u) < 0.5) { u = F((0.5 - D(u)) * 2.0); return -u * inversesqrt_(F(1.0 - D(u * u))); }
	u = F((D(u) - 0.5) * 2.0);
	return u * inversesqrt_(F(1.0 - D(u * u)));
}

/* hdr:2089-2119 */
static float ggx_qf2_radial(float u, float cos_k, float sin_k)
{
	float sin_t = F(D(u) * (1.0 + D(cos_k)) - 1.0);
	float cos_t = F(sqrt(1.0 - D(sin_t * sin_t)));
	if (D(cos_t) > 0.707107) {
		float tan_t = sin_t / cos_t;
		if (D(sin_k) < 0.707107) {
			float tan_k = sin_k / cos_k;
			return F(D(-(tan_t + tan_k)) / (1.0 - D(tan_t * tan_k)));
		} else {
			float cot_k = cos_k / sin_k;
			return F((1.0 + D(tan_t * cot_k)) / D(tan_t - cot_k));
		}
	} else {
		float cot_t = cos_t / sin_t;
		if (D(sin_k) < 0.707107) {
			float tan_k = sin_k / cos_k;
			return F((1.0 + D(tan_k * cot_t)) / D(tan_k - cot_t));
		} else {
			float cot_k = cos_k / sin_k;
			return F(D(cot_t + cot_k) / (1.0 - D(cot_t * cot_k)));
		}
	}
}

/* hdr:2121-2146 */
static float ggx_qf3_radial(float u, float qf2)
{
	float alpha = F(sqrt(1.0 + D(qf2 * qf2)));
	float S;
	if (D(u) < 0.5) { u = F(2.0 * (0.5 - D(u))); S = -1.0f; }
	else { u = F(2.0 * (D(u) - 0.5)); S = 1.0f; }
	double x = D(u);
	float p = F(x * (x * (x * (-0.365728915865723) + 0.790235037209296) - 0.424965825137544)
	            + 0.000152998850436920);
	float q = F(x * (x * (x * (x * 0.169507819808272 - 0.397203533833404) - 0.232500544458471) + 1)
	            - 0.539825872510702);
	return S * alpha * (p / q);
}

static int supports_smith_vndf(const o_brdf *b) { return b->kind != O_BRDF_TABULAR && b->kind != O_BRDF_TABULAR_ANISO; }

/* ------------------------------------------------------------------ tabular_anisotropic fetches */
static int uwrap_repeat(int i, int edge) { while (i >= edge) i -= edge; while (i < 0) i += edge; return i; } /* hdr:1183 */

/* spline::eval with uwrap_repeat (hdr:1207-1218) */
static float spline_eval_rep(const float *pts, int n, float u)
{
	double ip; float fr = F(modf(D(u * (float)n - u), &ip));
	int i1 = uwrap_repeat((int)ip, n), i2 = uwrap_repeat((int)ip + 1, n);
	return pts[i1] + fr * (pts[i2] - pts[i1]);
}
/* spline::eval2d(points, w, h, uwrap_edge, u1, uwrap_repeat, u2) (hdr:1220-1247) */
static float spline_eval2d(const float *pts, int w, int h, float u1, float u2)
{
	double ip1, ip2;
	float f1 = F(modf(D(u1 * (float)w - u1), &ip1));
	int i1 = uwrap_edge((int)ip1, w), i2 = uwrap_edge((int)ip1 + 1, w);
	float f2 = F(modf(D(u2 * (float)h - u2), &ip2));
	int j1 = uwrap_repeat((int)ip2, h), j2 = uwrap_repeat((int)ip2 + 1, h);
	float p1 = pts[i1 + w * j1], p2 = pts[i2 + w * j1], p3 = pts[i1 + w * j2], p4 = pts[i2 + w * j2];
	float t1 = p1 + f1 * (p2 - p1), t2 = p3 + f1 * (p4 - p3);
	return t1 + f2 * (t2 - t1);
}
static float aniso_grid(const o_brdf *b, const float *tab, float theta, float phi) /* hdr:2185-2196 */
{
	if (D(phi) < 0.0) phi = F(D(phi) + 2.0 * O_PI);
	float u1 = F(D(theta) * 2.0 / O_PI), u2 = F(D(phi) * 0.5 / O_PI);
	return spline_eval2d(tab, b->elev, b->azim, u1, u2);
}
static float aniso_p22_theta_phi(const o_brdf *b, float theta, float phi) { return aniso_grid(b, b->a_p22, theta, phi); }
static float aniso_p22_std(const o_brdf *b, float x, float y) /* hdr:2178-2183 */
{
	float theta = F(atan(sqrt(D(x * x + y * y))));
	float phi = F(atan2(D(-y), D(-x)));
	return aniso_p22_theta_phi(b, theta, phi);
}
static float aniso_sigma_std(const o_brdf *b, o_vec3 k) /* hdr:2198-2211 */
{
	return aniso_grid(b, b->a_sigma, F(acos(D(k.z))), F(atan2(D(k.y), D(k.x))));
}
static float aniso_pdf1(const o_brdf *b, float phi) { return spline_eval_rep(b->a_pdf1, b->n_a_pdf1, F(D(phi) * 0.5 / O_PI)); } /* hdr:2768 */
static float aniso_cdf1(const o_brdf *b, float phi) { return spline_eval_rep(b->a_cdf1, b->n_a_cdf1, F(D(phi) * 0.5 / O_PI)); }
static float aniso_qf1(const o_brdf *b, float u1) { return F(D(spline_eval_f(b->a_qf1, b->n_a_qf1, u1)) * 2.0 * O_PI); } /* hdr:2780 */
static float aniso_pdf2(const o_brdf *b, float theta, float phi) /* hdr:2786-2798 */
{
	if (D(theta) >= 0.5 * O_PI) return 0.0f;
	return spline_eval2d(b->a_pdf2, b->elev, b->azim, F(D(theta) * 2.0 / O_PI), F(D(phi) * 0.5 / O_PI));
}
static float aniso_cdf2(const o_brdf *b, float theta, float phi) /* hdr:2800-2812 */
{
	if (D(theta) >= 0.5 * O_PI) return 1.0f;
	return spline_eval2d(b->a_cdf2, b->elev, b->azim, F(D(theta) * 2.0 / O_PI), F(D(phi) * 0.5 / O_PI));
}
static float aniso_qf2(const o_brdf *b, float u, float phi) /* hdr:2814-2824 */
{
	float u1 = F(D(phi) / (2.0 * O_PI));
	return F(D(spline_eval2d(b->a_qf2, b->elev, b->azim, u, u1)) * 0.5 * O_PI);
}

/* ------------------------------------------------------------------ microfacet (hdr:1529-1765) */
static float mf_p22_std(const o_brdf *b, float x, float y)
{
	if (b->kind == O_BRDF_TABULAR_ANISO) return aniso_p22_std(b, x, y);
	return p22_radial(b, x * x + y * y);
}

static float mf_p22(const o_brdf *b, float x, float y, const o_params *p) /* hdr:1574-1587 */
{
	x -= p->tx; y -= p->ty;
	float nrm = p->ax * p->ay * p->sqrt_1mrho2;
	float x_ = x / p->ax;
	float t1 = p->ax * y - p->rho * p->ay * x;
	float t2 = p->ax * p->ay * p->sqrt_1mrho2;
	float y_ = t1 / t2;
	return mf_p22_std(b, x_, y_) / nrm;
}

static float mf_ndf(const o_brdf *b, o_vec3 h, const o_params *p) /* hdr:1559-1570 */
{
	if (h.z > O_EPSILON) {
		float c2 = h.z * h.z, c4 = c2 * c2;
		float xs = -h.x / h.z, ys = -h.y / h.z;
		return mf_p22(b, xs, ys, p) / c4;
	}
	return 0.0f;
}

static float mf_sigma(const o_brdf *b, o_vec3 k, const o_params *p) /* hdr:1619-1631 */
{
	float a = k.x * p->ax + k.y * p->ay * p->rho;
	float bb = k.y * p->ay * p->sqrt_1mrho2;
	float c = k.z - k.x * p->tx - k.y * p->ty;
	float nrm = F(sqrt(D(a * a + bb * bb + c * c)));
	o_vec3 kn = v3_div(v3(a, bb, c), nrm);
	if (b->kind == O_BRDF_TABULAR_ANISO) return nrm * aniso_sigma_std(b, kn);
	return nrm * sigma_std_radial(b, kn.z);
}

static float mf_g1(const o_brdf *b, o_vec3 k, const o_params *p) /* hdr:1633-1642 */
{
	if (D(v3_dot(k, p->n)) > 0.0) return k.z / mf_sigma(b, k, p);
	return 0.0f;
}

static float mf_gaf(const o_brdf *b, o_vec3 i, o_vec3 o, const o_params *p) /* hdr:1644-1665 */
{
	float g1o = mf_g1(b, o, p);
	if (b->shadow) {
		float g1i = mf_g1(b, i, p);
		float t = g1i * g1o;
		if (D(t) > 0.0) return t / (g1i + g1o - t);
		return 0.0f;
	}
	return g1o;
}

static float mf_vndf(const o_brdf *b, o_vec3 h, o_vec3 k, const o_params *p) /* hdr:1602-1615 */
{
	float kh = v3_dot(k, h);
	if (D(kh) > 0.0) {
		float Dn = mf_ndf(b, h, p);
		return kh * Dn / mf_sigma(b, k, p);
	}
	return 0.0f;
}

static float mf_vp22(const o_brdf *b, float x, float y, o_vec3 k, const o_params *p) /* hdr:1591-1598 */
{
	o_vec3 h = v3_normalize(v3(-x, -y, 1));
	float jac = h.z * h.z * h.z;
	return jac * mf_vndf(b, h, k, p);
}

static o_vec3 mf_evalp(const o_brdf *b, o_vec3 i, o_vec3 o, const o_params *p) /* hdr:1529-1547 */
{
	o_vec3 h = v3_normalize(v3_add(i, o));
	float G = mf_gaf(b, i, o, p);
	if (D(G) > 0.0) {
		float cd = satf_(v3_dot(o, h));
		o_vec3 Fr = fresnel_eval(&b->fresnel, cd);
		float Dn = mf_ndf(b, h, p);
		return v3_scale(F(D(Dn * G) / (4.0 * D(o.z))), Fr);
	}
	return v3(0, 0, 0);
}

static o_vec3 mf_eval(const o_brdf *b, o_vec3 i, o_vec3 o, const o_params *p) /* hdr:1551-1555 */
{
	return v3_div(mf_evalp(b, i, o, p), i.z);
}

static float mf_pdf(const o_brdf *b, o_vec3 i, o_vec3 o, const o_params *p) /* hdr:1713-1730 */
{
	o_vec3 h = v3_normalize(v3_add(i, o));
	float G = mf_gaf(b, i, o, p);
	if (D(G) > 0.0) {
		if (!supports_smith_vndf(b))
			return F(D(h.z * mf_ndf(b, h, p)) / (4.0 * D(v3_dot(i, h))));
		return F(D(mf_vndf(b, h, o, p)) / (4.0 * D(v3_dot(i, h))));
	}
	return 0.0f;
}

/* radial::sample_vp22_std_smith / _nmap, hdr:1806-1846 */
static void mf_sample_vp22_std(const o_brdf *b, float u1, float u2, o_vec3 k, float *xs, float *ys)
{
	if (supports_smith_vndf(b)) {
		float cos_k = k.z;
		float sin_k = D(k.z) < 1.0 ? F(sqrt(1.0 - D(k.z * k.z))) : 0.0f;
		float tx, ty;
		if (b->kind == O_BRDF_BECKMANN) {
			tx = beckmann_qf2_radial(u1, cos_k, sin_k);
			ty = beckmann_qf1(u2);                          /* qf3_radial, hdr:1954 */
		} else {
			tx = ggx_qf2_radial(u1, cos_k, sin_k);
			ty = ggx_qf3_radial(u2, tx);
		}
		if (D(sin_k) == 0.0) { *xs = tx; *ys = ty; }
		else {
			float nrm = inversesqrt_(k.x * k.x + k.y * k.y);
			float cp = k.x * nrm, sp = k.y * nrm;
			*xs = cp * tx - sp * ty;
			*ys = sp * tx + cp * ty;
		}
	} else if (b->kind == O_BRDF_TABULAR_ANISO) { /* hdr:2828-2839 */
		float phi = aniso_qf1(b, u1);
		float theta = aniso_qf2(b, u2, phi);
		float tan_theta = F(tan(D(theta)));
		*xs = F(D(-tan_theta) * cos(D(phi)));
		*ys = F(D(-tan_theta) * sin(D(phi)));
	} else {
		float phi_h = F(D(u1) * O_PI * 2.0);
		float r_h = qf_radial(b, u2);
		*xs = F(D(r_h) * cos(D(phi_h)));
		*ys = F(D(r_h) * sin(D(phi_h)));
	}
}

static o_vec3 mf_sample(const o_brdf *b, float u1, float u2, o_vec3 o, const o_params *p) /* hdr:1669-1709 */
{
	u1 = satf_(u1) * 0.99998f + 0.00001f;
	u2 = satf_(u2) * 0.99998f + 0.00001f;
	float a = o.x * p->ax + o.y * p->ay * p->rho;
	float bb = o.y * p->ay * p->sqrt_1mrho2;
	float c = o.z - o.x * p->tx - o.y * p->ty;
	o_vec3 o_std = v3_normalize(v3(a, bb, c));
	if (D(o_std.z) > 0.0) {
		float txm, tym;
		mf_sample_vp22_std(b, u1, u2, o_std, &txm, &tym);
		float txh = p->ax * txm + p->tx;
		float chol = p->rho * txm + p->sqrt_1mrho2 * tym;
		float tyh = p->ay * chol + p->ty;
		o_vec3 h = v3_normalize(v3(-txh, -tyh, 1));
		return v3_sub(v3_scale(F(2.0 * D(v3_dot(o, h))), h), o);
	}
	return v3(0, 0, 1);
}

/* hdr:1734-1765 */
static o_vec3 mf_evalp_is(const o_brdf *b, float u1, float u2, o_vec3 o, const o_params *p,
                          o_vec3 *i_out, float *pdf_out)
{
	o_vec3 i_ = mf_sample(b, u1, u2, o, p);
	o_vec3 h = v3_normalize(v3_add(i_, o));
	float G = mf_gaf(b, i_, o, p);
	*pdf_out = 0.f;
	if (D(G) > 0.0) {
		float cd = satf_(v3_dot(o, h));
		*i_out = i_;
		if (!supports_smith_vndf(b)) {
			float pdf_ = F(D(h.z * mf_ndf(b, h, p)) / (4.0 * D(cd)));
			*pdf_out = pdf_;
			return v3_div(mf_evalp(b, i_, o, p), pdf_);
		} else {
			o_vec3 Fr = fresnel_eval(&b->fresnel, cd);
			float G1 = mf_g1(b, o, p);
			*pdf_out = F(D(mf_vndf(b, h, o, p)) / (4.0 * D(cd)));
			return v3_scale(G / G1, Fr);
		}
	}
	return v3(0, 0, 0);
}

/* ------------------------------------------------------------------ MERL (hdr:893-1024) */
static int theta_half_index(float th) /* hdr:906-920 */
{
	if (D(th) <= 0.0) return 0;
	float deg = F((D(th) / (O_PI / 2.0)) * 90);
	float t = deg * 90;
	t = F(sqrt(D(t)));
	int r = (int)t;
	return r < 0 ? 0 : (r >= 90 ? 89 : r);
}
static int theta_diff_index(float td) /* hdr:926-936 */
{
	int t = (int)(D(td) / (O_PI * 0.5) * 90);
	return t < 0 ? 0 : (t < 89 ? t : 89);
}
static int phi_diff_index(float pd) /* hdr:940-957 */
{
	if (D(pd) < 0.0) pd = F(D(pd) + O_PI);
	int t = (int)(D(pd) / O_PI * 360 / 2);
	return t < 0 ? 0 : (t < 179 ? t : 179);
}
static int merl_index(o_vec3 i, o_vec3 o)
{
	o_vec3 h, d; float th, ph, td, pd;
	io_to_hd(i, o, &h, &d);
	xyz_to_theta_phi(h, &th, &ph);
	xyz_to_theta_phi(d, &td, &pd);
	return phi_diff_index(pd) + theta_diff_index(td) * 180 + theta_half_index(th) * 16200;
}
static o_vec3 merl_eval(const o_brdf *b, o_vec3 i, o_vec3 o) /* hdr:987-1024 */
{
	int64_t n = b->n_samples / 3; /* 1 458 000 for real files; index constants are fixed */
	(void)n;
	int ir = merl_index(i, o), ig = ir + 1458000, ib = ir + 2916000;
	o_vec3 rgb = v3(F(b->samples[ir] * (1.00 / 1500.0)),
	                F(b->samples[ig] * (1.15 / 1500.0)),
	                F(b->samples[ib] * (1.66 / 1500.0)));
	if (D(rgb.x) < 0.0 || D(rgb.y) < 0.0 || D(rgb.z) < 0.0) return v3(0, 0, 0);
	return rgb;
}

/* ------------------------------------------------------------------ UTIA (hdr:1029-1177) */
static o_vec3 utia_eval(const o_brdf *b, o_vec3 i, o_vec3 o) /* hdr:1063-1157 */
{
	float r2d = F(180.0 / O_PI);
	float theta_i = F(D(r2d) * acos(D(i.z))), theta_o = F(D(r2d) * acos(D(o.z)));
	float phi_i = F(D(r2d) * atan2(D(i.y), D(i.x))), phi_o = F(D(r2d) * atan2(D(o.y), D(o.x)));
	if (D(theta_i) >= 90.0 || D(theta_o) >= 90.0) return v3(0, 0, 0);
	while (D(phi_i) < 0.0) phi_i = F(D(phi_i) + 360.0);
	while (D(phi_o) < 0.0) phi_o = F(D(phi_o) + 360.0);
	while (phi_i >= 360) phi_i = F(D(phi_i) - 360.0);
	while (phi_o >= 360) phi_o = F(D(phi_o) - 360.0);
	int iti[2], itv[2], ipi[2], ipv[2];
	iti[0] = (int)floor(D(theta_i) / 15.0); iti[1] = iti[0] + 1;
	if (iti[0] > 4) { iti[0] = 4; iti[1] = 5; }
	itv[0] = (int)floor(D(theta_o) / 15.0); itv[1] = itv[0] + 1;
	if (itv[0] > 4) { itv[0] = 4; itv[1] = 5; }
	ipi[0] = (int)floor(D(phi_i) / 7.5); ipi[1] = ipi[0] + 1;
	ipv[0] = (int)floor(D(phi_o) / 7.5); ipv[1] = ipv[0] + 1;
	float sum, wti[2], wtv[2], wpi[2], wpv[2];
	wti[1] = theta_i - F(15.0 * iti[0]); wti[0] = F(15.0 * iti[1]) - theta_i;
	sum = wti[0] + wti[1]; wti[0] /= sum; wti[1] /= sum;
	wtv[1] = theta_o - F(15.0 * itv[0]); wtv[0] = F(15.0 * itv[1]) - theta_o;
	sum = wtv[0] + wtv[1]; wtv[0] /= sum; wtv[1] /= sum;
	wpi[1] = phi_i - F(7.5 * ipi[0]); wpi[0] = F(7.5 * ipi[1]) - phi_i;
	sum = wpi[0] + wpi[1]; wpi[0] /= sum; wpi[1] /= sum;
	wpv[1] = phi_o - F(7.5 * ipv[0]); wpv[0] = F(7.5 * ipv[1]) - phi_o;
	sum = wpv[0] + wpv[1]; wpv[0] /= sum; wpv[1] /= sum;
	if (ipi[1] == 48) ipi[1] = 0;
	if (ipv[1] == 48) ipv[1] = 0;
	const int nc = 48 * 6, nr = 48 * 6;
	float RGB[3];
	for (int isp = 0; isp < 3; ++isp) {
		RGB[isp] = 0.0f;
		for (int a = 0; a < 2; ++a) for (int c = 0; c < 2; ++c)
		for (int k = 0; k < 2; ++k) for (int l = 0; l < 2; ++l) {
			float w = wti[a] * wtv[c] * wpi[k] * wpv[l];
			int idx = isp * nr * nc + nc * (48 * iti[a] + ipi[k]) + 48 * itv[c] + ipv[l];
			RGB[isp] += w * F(b->samples[idx]);
		}
		if (D(RGB[isp]) > 0.0375) RGB[isp] = F(pow(D(F(D(RGB[isp]) + 0.055)) / 1.055, D(2.4f)));
		else RGB[isp] /= 12.92f;
		RGB[isp] *= 100.0f;
	}
	return v3(fmaxf_(0.f, RGB[0]), fmaxf_(0.f, RGB[1]), fmaxf_(0.f, RGB[2]));
}

/* ------------------------------------------------------------------ SGD (hdr:3415-3500) */
static double dmin(double a, double b) { return a < b ? a : b; }
static double dmax(double a, double b) { return a > b ? a : b; }

static double sgd_g1(o_vec3 k, double theta0, double c, double k_, double lambda) /* hdr:3415-3422 */
{
	double t1 = dmax(0.0, acos(D(k.z)) - theta0);
	double t2 = 1.0 - exp(c * pow(t1, k_));
	double t3 = 1.0 + lambda * t2;
	return dmin(1.0, dmax(0.0, t3));
}
static double sgd_ndf(double ch, double alpha, double p, double kap) /* hdr:3424-3432 */
{
	const double inv_pi = 1.0 / O_PI;
	double c2 = ch * ch;
	double t2 = (1.0 - c2) / c2;
	double ax = alpha + t2 / alpha;
	return (kap * exp(-ax) * inv_pi) / (pow(ax, p) * c2 * c2);
}
static o_vec3 sgd_eval(const o_brdf *b, o_vec3 i, o_vec3 o) /* hdr:3454-3468 */
{
	const double *m = b->model; /* rhoD rhoS alpha p f0 f1 kap lambda c k theta0 */
	if (D(i.z) > 0.0 && D(o.z) > 0.0) {
		o_vec3 h = v3_normalize(v3_add(i, o));
		o_vec3 Kd = v3(F(m[0]), F(m[1]), F(m[2])), Ks = v3(F(m[3]), F(m[4]), F(m[5]));
		o_vec3 Fr = fresnel_eval(&b->fresnel, satf_(v3_dot(i, h)));
		float g1i[3], g1o[3], nd[3];
		for (int c = 0; c < 3; ++c) {
			g1i[c] = F(sgd_g1(i, m[30 + c], m[24 + c], m[27 + c], m[21 + c]));
			g1o[c] = F(sgd_g1(o, m[30 + c], m[24 + c], m[27 + c], m[21 + c]));
			nd[c] = F(sgd_ndf(D(h.z), m[6 + c], m[9 + c], m[18 + c]));
		}
		o_vec3 G = v3_mul(v3(g1i[0], g1i[1], g1i[2]), v3(g1o[0], g1o[1], g1o[2]));
		o_vec3 Dn = v3(nd[0], nd[1], nd[2]);
		o_vec3 spec = v3_div(v3_mul(Ks, v3_mul(v3_mul(Fr, Dn), G)), i.z * o.z);
		return v3_div(v3_add(Kd, spec), F(O_PI));
	}
	return v3(0, 0, 0);
}

/* ------------------------------------------------------------------ ABC (hdr:3608-3668) */
static o_vec3 abc_eval(const o_brdf *b, o_vec3 i, o_vec3 o) /* hdr:3633-3645 */
{
	const double *m = b->model; /* kD[3] A[3] B C ior */
	if (D(i.z) > 0.0 && D(o.z) > 0.0) {
		o_vec3 h = v3_normalize(v3_add(i, o));
		o_vec3 Kd = v3(F(m[0]), F(m[1]), F(m[2]));
		o_vec3 Fr = fresnel_eval(&b->fresnel, satf_(v3_dot(i, h)));
		float g1_i = fminf_(1.0f, 2.0f * (h.z * i.z / v3_dot(h, i)));   /* hdr:3649-3655 */
		float g1_o = fminf_(1.0f, 2.0f * (h.z * o.z / v3_dot(h, o)));
		float G = fminf_(g1_i, g1_o);
		double ch = D(h.z), tmp = 1.0 - ch;                              /* hdr:3608-3613 */
		o_vec3 Dn = v3(F(m[3] / pow(1.0 + m[6] * tmp, m[7])), F(m[4] / pow(1.0 + m[6] * tmp, m[7])),
		               F(m[5] / pow(1.0 + m[6] * tmp, m[7])));
		o_vec3 spec = v3_div(v3_scale(G, v3_mul(Fr, Dn)), F(O_PI * D(i.z) * D(o.z)));
		return v3_add(v3_div(Kd, F(O_PI)), spec);
	}
	return v3(0, 0, 0);
}

static o_vec3 ld3(const float *p, int64_t k);
static void st3(float *p, int64_t k, o_vec3 v);
/* vec3::vec3(theta, phi), hdr:589-595 */
void o_vec3_angles(int64_t n, const float *theta, const float *phi, float *out)
{
	for (int64_t k = 0; k < n; ++k) st3(out, k, v3_from_angles(theta[k], phi[k]));
}
/* sgd::{ndf,gaf,g1} (hdr:3472-3500) and abc::{ndf,gaf} (hdr:3647-3668) as their own entry point.
 * which: 0 ndf(h) -> rgb, 1 gaf(h, i, o) -> rgb (sgd) / out[0] (abc), 2 g1(k) -> rgb (sgd), 3 fresnel(a[0]) */
void o_model_query(const o_brdf *b, int which, int64_t n, const float *a, const float *bi, const float *co, float *out)
{
	const double *m = b->model;
	for (int64_t k = 0; k < n; ++k) {
		o_vec3 A = ld3(a, k), r = v3(0, 0, 0);
		if (which == 3) r = fresnel_eval(&b->fresnel, A.x);
		else if (b->kind == O_BRDF_SGD) {
			if (which == 0)
				r = v3(F(sgd_ndf(D(A.z), m[6], m[9], m[18])), F(sgd_ndf(D(A.z), m[7], m[10], m[19])), F(sgd_ndf(D(A.z), m[8], m[11], m[20])));
			else {
				o_vec3 g[2];
				for (int w = 0; w < 2; ++w) {
					o_vec3 kk = which == 2 ? A : ld3(w == 0 ? bi : co, k);
					g[w] = v3(F(sgd_g1(kk, m[30], m[24], m[27], m[21])), F(sgd_g1(kk, m[31], m[25], m[28], m[22])),
					          F(sgd_g1(kk, m[32], m[26], m[29], m[23])));
				}
				r = which == 2 ? g[0] : v3_mul(g[0], g[1]);
			}
		} else {
			if (which == 0) {
				double tmp = 1.0 - D(A.z);
				r = v3(F(m[3] / pow(1.0 + m[6] * tmp, m[7])), F(m[4] / pow(1.0 + m[6] * tmp, m[7])), F(m[5] / pow(1.0 + m[6] * tmp, m[7])));
			} else if (which == 1) {
				o_vec3 i = ld3(bi, k), o = ld3(co, k);
				float g1_i = fminf_(1.0f, 2.0f * (A.z * i.z / v3_dot(A, i)));
				float g1_o = fminf_(1.0f, 2.0f * (A.z * o.z / v3_dot(A, o)));
				r.x = fminf_(g1_i, g1_o);
			}
		}
		st3(out, k, r);
	}
}

/* ------------------------------------------------------------------ generic dispatch */
static int is_microfacet(const o_brdf *b) { return b->kind <= O_BRDF_TABULAR || b->kind == O_BRDF_TABULAR_ANISO; }

/* the user-defined lobes of ref_shim.cpp (user_phong::eval, user_ward::eval); model = {which, kd[3], ks[3], p0, p1} */
static o_vec3 custom_eval(const o_brdf *b, o_vec3 i, o_vec3 o)
{
	const double *m = b->model;
	o_vec3 kd = v3(F(m[1]), F(m[2]), F(m[3])), ks = v3(F(m[4]), F(m[5]), F(m[6]));
	if ((int)m[0] == 0) {
		float n = F(m[7]);
		o_vec3 r = v3(-o.x, -o.y, o.z);
		float c = v3_dot(r, i);
		if (!(c > 0.0f)) c = 0.0f;
		float s = F((D(n) + 2.0) / (2.0 * O_PI) * pow(D(c), D(n)));
		return v3_add(v3_div(kd, F(O_PI)), v3_scale(s, ks));
	}
	float ax = F(m[7]), ay = F(m[8]);
	if (!(i.z > 0.0f && o.z > 0.0f)) return v3(0, 0, 0);
	o_vec3 h = v3_normalize(v3_add(i, o));
	float tx = h.x / ax, ty = h.y / ay;
	float q = (tx * tx + ty * ty) / (h.z * h.z);
	float e = F(exp(-D(q)));
	float den = F(4.0 * O_PI * D(ax * ay) * sqrt(D(i.z * o.z)));
	return v3_add(v3_div(kd, F(O_PI)), v3_scale(e / den, ks));
}

static o_vec3 brdf_eval(const o_brdf *b, o_vec3 i, o_vec3 o, const o_params *p)
{
	switch (b->kind) {
	case O_BRDF_CUSTOM: return custom_eval(b, i, o);
	case O_BRDF_MERL: return merl_eval(b, i, o);
	case O_BRDF_UTIA: return utia_eval(b, i, o);
	case O_BRDF_LAMBERT: return v3_div(p && p->a1 == -1.0f ? p->n : v3(1, 1, 1), F(O_PI)); /* hdr:861-868: reflectance / M_PI */
	case O_BRDF_SGD: return sgd_eval(b, i, o);
	case O_BRDF_ABC: return abc_eval(b, i, o);
	default: return mf_eval(b, i, o, p);
	}
}
static o_vec3 brdf_evalp(const o_brdf *b, o_vec3 i, o_vec3 o, const o_params *p)
{
	if (is_microfacet(b)) return mf_evalp(b, i, o, p);
	return v3_scale(i.z, brdf_eval(b, i, o, p)); /* hdr:803-806 */
}
static float brdf_pdf(const o_brdf *b, o_vec3 i, o_vec3 o, const o_params *p)
{
	if (is_microfacet(b)) return mf_pdf(b, i, o, p);
	return F(D(i.z) / O_PI); /* hdr:842-845 */
}
static o_vec3 brdf_sample(const o_brdf *b, float u1, float u2, o_vec3 o, const o_params *p)
{
	if (is_microfacet(b)) return mf_sample(b, u1, u2, o, p);
	float x, y; /* hdr:830-840 */
	uniform_to_concentric(u1, u2, &x, &y);
	return v3(x, y, F(sqrt(1.0 - D(x * x) - D(y * y))));
}

/* ------------------------------------------------------------------ the fitter (hdr:2215-2762) */
static void tab_compute_p22_smith(o_brdf *t, const o_brdf *src, int res) /* hdr:2482-2522 */
{
	int cnt = res - 1;
	float dtheta = F(sqrt(O_PI * 0.5) / D((float)cnt));
	double *km = (double *)calloc((size_t)cnt * cnt, sizeof(double));
	o_params std_p = params_standard();
	for (int i = 0; i < cnt; ++i) {
		float tmp = (float)i / (float)cnt;
		float theta = F(D(tmp) * sqrt(O_PI * 0.5));
		float theta_o = theta * theta;
		float cos_o = F(cos(D(theta_o))), tan_o = F(tan(D(theta_o)));
		o_vec3 w = v3_from_angles(theta_o, 0.0f);
		o_vec3 fr = brdf_eval(src, w, w, &std_p);
		float fr_i = v3_intensity(fr);
		float kji_tmp = F((D(dtheta) * pow(D(cos_o), D(6.0f))) * (8.0 * D(fr_i)));
		for (int j = 0; j < cnt; ++j) {
			const float dphi_h = F(O_PI / 180.0);
			float tmpj = (float)j / (float)cnt;
			float thj = F(D(tmpj) * sqrt(O_PI * 0.5));
			float theta_h = thj * thj;
			float cos_h = F(cos(D(theta_h))), tan_h = F(tan(D(theta_h)));
			float tan_product = tan_h * tan_o;
			float nint = 0.0f;
			for (float phi_h = 0.0f; D(phi_h) < 2.0 * O_PI; phi_h += dphi_h)
				nint += fmaxf_(1.0f, tan_product * F(cos(D(phi_h))));
			nint *= dphi_h;
			/* km(j, i) -> mij[i*size + j], hdr:2447 */
			km[(size_t)i * cnt + j] = D(thj * kji_tmp * nint * tan_h / (cos_h * cos_h));
		}
	}
	/* matrix::eigenvector(4): 4 un-normalised matvecs from ones, hdr:2455-2480 */
	double *v0 = (double *)malloc(sizeof(double) * cnt), *v1 = (double *)malloc(sizeof(double) * cnt);
	for (int i = 0; i < cnt; ++i) v0[i] = 1.0;
	for (int it = 0; it < 4; ++it) {
		for (int j = 0; j < cnt; ++j) {
			double acc = 0;
			for (int i = 0; i < cnt; ++i) acc += km[(size_t)j * cnt + i] * v0[i];
			v1[j] = acc;
		}
		double *sw = v0; v0 = v1; v1 = sw;
	}
	t->n_p22 = res;
	t->p22 = (float *)malloc(sizeof(float) * res);
	for (int i = 0; i < cnt; ++i) t->p22[i] = F(1e-2 * v0[i]);
	t->p22[cnt] = 0.0f;
	free(v0); free(v1); free(km);
}

static void tab_normalize_p22(o_brdf *t) /* hdr:2277-2304 */
{
	const int ntheta = 128;
	const float dphi = F(2.0 * O_PI);
	const float dtheta = F(O_PI / D((float)ntheta));
	float nint = 0.0f;
	for (int i = 0; i < ntheta; ++i) {
		float u = (float)i / (float)ntheta;
		float theta_h = F(D(u * u) * O_PI * 0.5);
		float r_h = F(tan(D(theta_h))), cos_h = F(cos(D(theta_h)));
		float p22_r = tab_p22_radial(t, r_h * r_h);
		nint += (u * p22_r * r_h) / (cos_h * cos_h);
	}
	nint *= dtheta * dphi;
	nint = F(1.0 / D(nint));
	for (int i = 0; i < t->n_p22; ++i) t->p22[i] *= nint;
}

static void tab_compute_sigma(o_brdf *t) /* hdr:2348-2386 */
{
	const int ntheta = 90, nphi = 180;
	float dtheta = F(O_PI / D((float)ntheta));
	float dphi = F(2.0 * O_PI / D((float)nphi));
	int cnt = t->n_p22 - 1;
	o_params std_p = params_standard();
	t->sigma = (float *)malloc(sizeof(float) * (cnt + 1));
	t->n_sigma = 0;
	/* ndf(vec3(theta_h, phi_h)) does not depend on theta_k: evaluate once (same values) */
	float *ndf_tab = (float *)malloc(sizeof(float) * ntheta * nphi);
	for (int j2 = 0; j2 < nphi; ++j2) {
		float phi_h = F(D((float)j2 / (float)nphi) * 2.0 * O_PI);
		for (int j1 = 0; j1 < ntheta; ++j1) {
			float u_i = (float)j1 / (float)ntheta;
			float theta_h = F(D(u_i * u_i) * O_PI * 0.5);
			ndf_tab[j2 * ntheta + j1] = mf_ndf(t, v3_from_angles(theta_h, phi_h), &std_p);
		}
	}
	for (int i = 0; i < cnt; ++i) {
		float tmp = (float)i / (float)cnt;
		float theta_k = F(D(tmp) * 0.5 * O_PI);
		float cos_k = F(cos(D(theta_k))), sin_k = F(sin(D(theta_k)));
		float nint = 0.0f;
		for (int j2 = 0; j2 < nphi; ++j2) {
			float u_j = (float)j2 / (float)nphi;
			float phi_h = F(D(u_j) * 2.0 * O_PI);
			for (int j1 = 0; j1 < ntheta; ++j1) {
				float u_i = (float)j1 / (float)ntheta;
				float theta_h = F(D(u_i * u_i) * O_PI * 0.5);
				float sin_h = F(sin(D(theta_h)));
				float kh = F(D(sin_k * sin_h) * cos(D(phi_h)) + D(cos_k) * cos(D(theta_h)));
				nint += fmaxf_(0.0f, kh) * ndf_tab[j2 * ntheta + j1] * u_i * sin_h;
			}
		}
		nint *= dtheta * dphi;
		t->sigma[t->n_sigma++] = fmaxf_(cos_k, nint);
	}
	t->sigma[t->n_sigma] = t->sigma[t->n_sigma - 1];
	t->n_sigma++;
	free(ndf_tab);
}

static void tab_compute_fresnel(o_brdf *t, const o_brdf *src, int res) /* hdr:2583-2641 */
{
	o_vec3 *fres = (o_vec3 *)malloc(sizeof(o_vec3) * res);
	int cnt = res - 1;
	o_params std_p = params_standard();
	for (int i = 0; i < cnt; ++i) {
		const float phi_d = F(O_PI * 0.5), phi_h = 0.0f;
		float tmp = (float)i / (float)cnt;
		float theta_d = F(D(tmp) * O_PI * 0.5);
		o_vec3 f = v3(0, 0, 0);
		int count[3] = { 0, 0, 0 };
		float theta_h = 0.0f;
		for (int j = 0; D(theta_h) < O_PI * 0.5 - D(theta_d); ++j) {
			float tmp1 = (float)j / (float)cnt;
			theta_h = F(D(tmp1 * tmp1) * O_PI * 0.5);
			if (D(theta_h) > O_PI * 0.5) continue;
			o_vec3 dir_h = v3_from_angles(theta_h, phi_h), dir_d = v3_from_angles(theta_d, phi_d);
			o_vec3 dir_i, dir_o;
			hd_to_io(dir_h, dir_d, &dir_i, &dir_o);
			dir_i = v3(0, 0, 1);
			o_vec3 fr1 = brdf_eval(src, dir_i, dir_o, &std_p);
			o_vec3 fr2 = mf_eval(t, dir_i, dir_o, &std_p);
			if (D(fr2.x) > 1e-4) { f.x += fr1.x / fr2.x; ++count[0]; }
			if (D(fr2.y) > 1e-4) { f.y += fr1.y / fr2.y; ++count[1]; }
			if (D(fr2.z) > 1e-4) { f.z += fr1.z / fr2.z; ++count[2]; }
		}
		fres[i].x = count[0] == 0 ? 1.0f : fminf_(1.0f, f.x / (float)count[0]);
		fres[i].y = count[1] == 0 ? 1.0f : fminf_(1.0f, f.y / (float)count[1]);
		fres[i].z = count[2] == 0 ? 1.0f : fminf_(1.0f, f.z / (float)count[2]);
	}
	fres[res - 1] = fres[res - 2];
	t->fresnel.kind = O_FRESNEL_SPLINE;
	t->fresnel.pts = fres;
	t->fresnel.npts = res;
}

static void tab_compute_cdf(o_brdf *t) /* hdr:2705-2727 */
{
	int cnt = t->n_p22 - 1;
	float dtheta = F(O_PI / D((float)cnt));
	float nint = 0.0f;
	t->cdf = (float *)malloc(sizeof(float) * (cnt + 1));
	t->n_cdf = 0;
	for (int i = 0; i < cnt; ++i) {
		float u = (float)i / (float)cnt;
		float theta_h = F(D(u * u) * O_PI * 0.5);
		float cos_h = F(cos(D(theta_h))), r_h = F(tan(D(theta_h)));
		float p22_r = tab_p22_radial(t, r_h * r_h);
		nint += (u * r_h * p22_r) / (cos_h * cos_h);
		t->cdf[t->n_cdf++] = F(D(nint * dtheta) * (2.0 * O_PI));
	}
	t->cdf[t->n_cdf++] = 1.0f;
}

static void tab_compute_qf(o_brdf *t) /* hdr:2731-2762 */
{
	int cnt = t->n_p22 - 1, res = cnt * 8, j = 0;
	t->qf = (float *)malloc(sizeof(float) * (cnt + 1));
	t->n_qf = 0;
	t->qf[t->n_qf++] = 0.0f;
	for (int i = 1; i < cnt; ++i) {
		float cdf = (float)i / (float)cnt;
		for (; j < res; ++j) {
			float u = (float)j / (float)res;
			float theta_h = F(D(u) * O_PI * 0.5);
			float qf = tab_cdf_radial(t, F(tan(D(theta_h))));
			if (qf >= cdf) { t->qf[t->n_qf++] = u; break; }
		}
	}
	t->qf[t->n_qf++] = 1.0f;
}

o_brdf *o_create_tabular(const o_brdf *src, int res, int shadow) /* hdr:2215-2236 */
{
	if (res <= 2) { set_err("Invalid Resolution"); return NULL; }
	o_brdf *t = (o_brdf *)calloc(1, sizeof *t);
	t->kind = O_BRDF_TABULAR;
	t->shadow = shadow != 0;
	t->fresnel.kind = O_FRESNEL_IDEAL;
	tab_compute_p22_smith(t, src, res);
	tab_normalize_p22(t);
	tab_compute_sigma(t);
	tab_compute_fresnel(t, src, res);
	tab_compute_cdf(t);
	tab_compute_qf(t);
	return t;
}

/* ------------------------------------------------------------------ tabular_anisotropic fitter */
typedef struct { float *v; int n, cap; } fvec;
static void fv_push(fvec *f, float x)
{
	if (f->n == f->cap) { f->cap = f->cap ? 2 * f->cap : 64; f->v = (float *)realloc(f->v, sizeof(float) * f->cap); }
	f->v[f->n++] = x;
}

static void aniso_compute_p22_smith(o_brdf *t, const o_brdf *src) /* hdr:2525-2579 */
{
	int w = t->elev - 1, h = t->azim, N = w * h;
	float dtheta = F(sqrt(O_PI * 0.5) / D((float)w)), dphi = F(2.0 * O_PI / D((float)h));
	o_params std_p = params_standard();
	float *k1 = (float *)malloc(sizeof(float) * N), *xo = (float *)malloc(sizeof(float) * N),
	      *yo = (float *)malloc(sizeof(float) * N), *zo = (float *)malloc(sizeof(float) * N),
	      *s1 = (float *)malloc(sizeof(float) * N), *s2 = (float *)malloc(sizeof(float) * N),
	      *tn = (float *)malloc(sizeof(float) * N), *dn = (float *)malloc(sizeof(float) * N);
	for (int i2 = 0; i2 < h; ++i2) for (int i1 = 0; i1 < w; ++i1) {
		int a = i2 * w + i1;
		float theta = F(D((float)i1 / (float)w) * 0.5 * O_PI), phi = F(D((float)i2 / (float)h) * 2.0 * O_PI);
		float st = F(sin(D(theta)));
		zo[a] = F(cos(D(theta))); xo[a] = F(D(st) * cos(D(phi))); yo[a] = F(D(st) * sin(D(phi)));
		o_vec3 wv = v3_from_angles(theta, phi);
		float fr_i = v3_intensity(brdf_eval(src, wv, wv, &std_p));
		k1[a] = F(D(dtheta * dphi) * (4.0 * D(fr_i) * pow(D(zo[a]), D(5.0f))));
		float ct = F(cos(D(theta))), tt = F(tan(D(theta)));
		tn[a] = tt; dn[a] = ct * ct;
		s1[a] = F(D(-tt) * cos(D(phi))); s2[a] = F(D(-tt) * sin(D(phi)));
	}
	/* eigenvector(4): out[a] = sum_b double(float(k1[a] * k2(a, b))) * v[b], b ascending (hdr:2455-2480) */
	double *v0 = (double *)malloc(sizeof(double) * N), *v1 = (double *)malloc(sizeof(double) * N);
	for (int a = 0; a < N; ++a) v0[a] = 1.0;
	for (int it = 0; it < 4; ++it) {
		for (int a = 0; a < N; ++a) {
			double acc = 0;
			for (int b = 0; b < N; ++b) {
				float m_dot_o = zo[a] - xo[a] * s1[b] - yo[a] * s2[b];
				float k2 = tn[b] * fmaxf_(0.0f, m_dot_o) / dn[b];
				acc += D(k1[a] * k2) * v0[b];
			}
			v1[a] = acc;
		}
		double *sw = v0; v0 = v1; v1 = sw;
	}
	t->a_p22 = (float *)malloc(sizeof(float) * t->elev * t->azim);
	for (int j = 0; j < h; ++j) {
		for (int i = 0; i < w; ++i) t->a_p22[i + t->elev * j] = F(v0[j * w + i]);
		t->a_p22[w + t->elev * j] = 0.0f;
	}
	free(k1); free(xo); free(yo); free(zo); free(s1); free(s2); free(tn); free(dn); free(v0); free(v1);
}

static void aniso_normalize_p22(o_brdf *t) /* hdr:2306-2338 */
{
	const int ntheta = 128, nphi = 256;
	float dtheta = F(sqrt(0.5 * O_PI) / D((float)ntheta)), dphi = F(2.0 * O_PI / D((float)nphi));
	float k = 0.0f;
	for (int j = 0; j < nphi; ++j) {
		float phi = F(D((float)j / (float)nphi) * 2.0 * O_PI);
		for (int i = 0; i < ntheta; ++i) {
			float theta = F(D((float)i / (float)ntheta) * sqrt(O_PI * 0.5));
			float ts = theta * theta;
			float c = F(cos(D(ts)));
			float pdf = aniso_p22_theta_phi(t, ts, phi);
			float weight = F(D(theta) * tan(D(ts)) / D(c * c));
			k += weight * pdf;
		}
	}
	k = F(D(k) * (2.0 * D(dtheta) * D(dphi)));
	k = F(1.0 / D(k));
	for (int i = 0; i < t->elev * t->azim; ++i) t->a_p22[i] *= k;
}

static void aniso_compute_sigma(o_brdf *t) /* hdr:2388-2432 */
{
	const int ntheta = 45, nphi = 90;
	float dtheta = F(sqrt(O_PI * 0.5) / D((float)ntheta)), dphi = F(2.0 * O_PI / D((float)nphi));
	int w = t->elev - 1, h = t->azim;
	o_params std_p = params_standard();
	float *ndf_tab = (float *)malloc(sizeof(float) * ntheta * nphi);
	for (int j2 = 0; j2 < nphi; ++j2) for (int j1 = 0; j1 < ntheta; ++j1) {
		float phi = F(D((float)j2 / (float)nphi) * 2.0 * O_PI);
		float theta = F(D((float)j1 / (float)ntheta) * sqrt(O_PI * 0.5));
		ndf_tab[j2 * ntheta + j1] = mf_ndf(t, v3_from_angles(theta * theta, phi), &std_p);
	}
	t->a_sigma = (float *)malloc(sizeof(float) * t->elev * t->azim);
	for (int i2 = 0; i2 < h; ++i2) {
		float phi_k = F(D((float)i2 / (float)h) * 2.0 * O_PI);
		for (int i1 = 0; i1 < w; ++i1) {
			float theta_k = F(D((float)i1 / (float)w) * 0.5 * O_PI);
			float cos_k = F(cos(D(theta_k)));
			float nint = 0.0f;
			for (int j2 = 0; j2 < nphi; ++j2) {
				float phi = F(D((float)j2 / (float)nphi) * 2.0 * O_PI);
				for (int j1 = 0; j1 < ntheta; ++j1) {
					float theta = F(D((float)j1 / (float)ntheta) * sqrt(O_PI * 0.5));
					float ts = theta * theta;
					float sin_t = F(sin(D(ts)));
					float m_dot_k = F(sin(D(theta_k)) * D(sin_t) * cos(D(phi - phi_k)) + D(cos_k) * cos(D(ts)));
					float weight = theta * sin_t;
					float masking = fmaxf_(0.0f, m_dot_k) * ndf_tab[j2 * ntheta + j1];
					nint += weight * masking;
				}
			}
			nint = F(D(nint) * (2.0 * D(dtheta) * D(dphi)));
			t->a_sigma[i1 + t->elev * i2] = fmaxf_(cos_k, nint);
		}
		t->a_sigma[w + t->elev * i2] = t->a_sigma[w - 1 + t->elev * i2];
	}
	free(ndf_tab);
}

/* nint += (val * tan(theta)) / (cos_theta * cos_theta): val float, tan double, the sum in double,
 * rounded to float on assignment (hdr:2866, 2965, 3078) */
static float acc_tan_over_cos2(float nint, float val, float theta)
{
	float c = F(cos(D(theta)));
	return F(D(nint) + (D(val) * tan(D(theta))) / D(c * c));
}

static void aniso_compute_pdf1(o_brdf *t) /* hdr:2849-2875 + normalize_pdf1 3038-3058 */
{
	const int ntheta = 256; int nphi = t->azim;
	float dtheta = F(0.5 * O_PI / D((float)ntheta));
	t->a_pdf1 = (float *)malloc(sizeof(float) * nphi); t->n_a_pdf1 = nphi;
	for (int i = 0; i < nphi; ++i) {
		float phi = F(D((float)i / (float)nphi) * 2.0 * O_PI);
		float nint = 0.0f;
		for (int j = 0; j < ntheta; ++j) {
			float theta = F(D((float)j / (float)ntheta) * 0.5 * O_PI);
			nint = acc_tan_over_cos2(nint, aniso_p22_theta_phi(t, theta, phi), theta);
		}
		t->a_pdf1[i] = nint * dtheta;
	}
	const int cnt = 512;
	float dphi = F(2.0 * O_PI / D((float)cnt)), nint = 0.0f;
	for (int i = 0; i < cnt; ++i) nint += aniso_pdf1(t, F(D((float)i / (float)cnt) * 2.0 * O_PI));
	nint *= dphi;
	float k = F(1.0 / D(nint));
	for (int i = 0; i < nphi; ++i) t->a_pdf1[i] *= k;
}

static void aniso_compute_cdf1(o_brdf *t) /* hdr:2879-2901 */
{
	int cnt = t->azim - 1;
	float dphi = F(2.0 * O_PI / D((float)cnt)), nint = 0.0f;
	fvec f = { 0, 0, 0 };
	fv_push(&f, 0.0f);
	for (int i = 1; i < cnt; ++i) {
		nint += aniso_pdf1(t, F(D((float)i / (float)cnt) * 2.0 * O_PI));
		fv_push(&f, nint * dphi);
	}
	fv_push(&f, 1.0f);
	t->a_cdf1 = f.v; t->n_a_cdf1 = f.n;
}

static void aniso_compute_qf1(o_brdf *t) /* hdr:2905-2936 */
{
	int cnt = t->n_a_cdf1 - 1, res = cnt * 8, j = 0;
	fvec f = { 0, 0, 0 };
	fv_push(&f, 0.0f);
	for (int i = 1; i < cnt; ++i) {
		float cdf = (float)i / (float)cnt;
		for (; j < res; ++j) {
			float u = (float)j / (float)res;
			if (aniso_cdf1(t, F(D(u) * 2.0 * O_PI)) >= cdf) { fv_push(&f, u); break; }
		}
	}
	fv_push(&f, 1.0f);
	t->a_qf1 = f.v; t->n_a_qf1 = f.n;
}

static void aniso_compute_pdf2(o_brdf *t) /* hdr:2945-2970 + normalize_pdf2 3062-3094 */
{
	int ntheta = t->elev - 1, nphi = t->azim;
	fvec f = { 0, 0, 0 };
	for (int i = 0; i < nphi; ++i) {
		float phi = F(D((float)i / (float)nphi) * 2.0 * O_PI);
		for (int j = 0; j < ntheta; ++j) {
			float theta = F(D((float)j / (float)ntheta) * 0.5 * O_PI);
			fv_push(&f, aniso_p22_theta_phi(t, theta, phi) / aniso_pdf1(t, phi));
		}
		fv_push(&f, 0.0f);
	}
	t->a_pdf2 = f.v; t->n_a_pdf2 = f.n;
	const int nt = 256;
	float dtheta = F(0.5 * O_PI / D((float)nt));
	float *k = (float *)malloc(sizeof(float) * nphi);
	for (int j = 0; j < nphi; ++j) {
		float phi = F(D((float)j / (float)nphi) * 2.0 * O_PI), nint = 0.0f;
		for (int i = 0; i < nt; ++i) {
			float theta = F(D((float)i / (float)nt) * 0.5 * O_PI);
			nint = acc_tan_over_cos2(nint, aniso_pdf2(t, theta, phi), theta);
		}
		nint *= dtheta;
		k[j] = F(1.0 / D(nint));
	}
	for (int j = 0; j < nphi; ++j) for (int i = 0; i < t->elev; ++i) t->a_pdf2[i + t->elev * j] *= k[j];
	free(k);
}

static void aniso_compute_cdf2(o_brdf *t) /* hdr:2974-3001 */
{
	int ntheta = t->elev - 1, nphi = t->azim;
	float dtheta = F(0.5 * O_PI / D((float)ntheta));
	fvec f = { 0, 0, 0 };
	for (int i = 0; i < nphi; ++i) {
		float phi = F(D((float)i / (float)nphi) * 2.0 * O_PI), nint = 0.0f;
		for (int j = 0; j < ntheta; ++j) {
			float theta = F(D((float)j / (float)ntheta) * 0.5 * O_PI);
			nint = acc_tan_over_cos2(nint, aniso_pdf2(t, theta, phi), theta);
			fv_push(&f, nint * dtheta);
		}
		fv_push(&f, 1.0f);
	}
	t->a_cdf2 = f.v; t->n_a_cdf2 = f.n;
}

static void aniso_compute_qf2(o_brdf *t) /* hdr:3005-3034 */
{
	int ntheta = t->elev - 1, nphi = t->azim, res = ntheta * 8;
	fvec f = { 0, 0, 0 };
	for (int k = 0; k < nphi; ++k) {
		float phi = F(D((float)k / (float)nphi) * 2.0 * O_PI);
		int j = 0;
		fv_push(&f, 0.0f);
		for (int i = 1; i < ntheta; ++i) {
			float cdf = (float)i / (float)ntheta;
			for (; j < res; ++j) {
				float u = (float)j / (float)res;
				if (aniso_cdf2(t, F(D(u) * 0.5 * O_PI), phi) >= cdf) { fv_push(&f, u); break; }
			}
		}
		fv_push(&f, 1.0f);
	}
	/* eval2d indexes the full elev x azim grid whatever the vector holds (hdr:2814-2824): a short row shifts all
	 * later rows, and the reference then reads past the end of m_qf2 (undefined).  The vector is kept exactly as
	 * the reference builds it; only the undefined tail is given a value (1.0). */
	t->n_a_qf2_ref = f.n;
	while (f.n < t->elev * t->azim) fv_push(&f, 1.0f);
	t->a_qf2 = f.v; t->n_a_qf2 = f.n;
}

o_brdf *o_create_tabular_anisotropic(const o_brdf *src, int elev, int azim, int shadow) /* hdr:2238-2273 */
{
	if (elev <= 1 || azim <= 1) { set_err("Invalid Resolution"); return NULL; }
	o_brdf *t = (o_brdf *)calloc(1, sizeof *t);
	t->kind = O_BRDF_TABULAR_ANISO;
	t->shadow = shadow != 0;
	t->fresnel.kind = O_FRESNEL_IDEAL;
	t->elev = elev; t->azim = azim;
	aniso_compute_p22_smith(t, src);
	aniso_normalize_p22(t);
	aniso_compute_sigma(t);
	tab_compute_fresnel(t, src, elev);         /* same loop as the isotropic class (hdr:2643-2701) */
	aniso_compute_pdf1(t);
	aniso_compute_cdf1(t);
	aniso_compute_qf1(t);
	aniso_compute_pdf2(t);
	aniso_compute_cdf2(t);
	aniso_compute_qf2(t);
	return t;
}

int o_aniso_get(const o_brdf *t, int which, float *out)
{
	if (which == 4) {
		if (t->fresnel.kind != O_FRESNEL_SPLINE) return 0;
		if (out) memcpy(out, t->fresnel.pts, sizeof(float) * 3 * t->fresnel.npts);
		return t->fresnel.npts;
	}
	const float *src = which == 0 ? t->a_p22 : t->a_sigma;
	if (out) memcpy(out, src, sizeof(float) * t->elev * t->azim);
	return t->elev * t->azim;
}

/* the six sampling tables as stored: which 0 pdf1 1 cdf1 2 qf1 3 pdf2 4 cdf2 5 qf2; returns the count.
 * which 6: no data, returns the number of entries of the reference's m_qf2 */
int o_aniso_get_table(const o_brdf *t, int which, float *out)
{
	const float *src[6] = { t->a_pdf1, t->a_cdf1, t->a_qf1, t->a_pdf2, t->a_cdf2, t->a_qf2 };
	const int n[6] = { t->n_a_pdf1, t->n_a_cdf1, t->n_a_qf1, t->n_a_pdf2, t->n_a_cdf2, t->n_a_qf2 };
	if (which == 6) return t->n_a_qf2_ref;
	if (which < 0 || which > 5) return 0;
	if (out) memcpy(out, src[which], sizeof(float) * n[which]);
	return n[which];
}

void o_aniso_query(const o_brdf *t, int which, int64_t n, const float *a, const float *b, float *out)
{
	for (int64_t k = 0; k < n; ++k) {
		switch (which) {
		case 0: out[k] = aniso_pdf1(t, a[k]); break;
		case 1: out[k] = aniso_cdf1(t, a[k]); break;
		case 2: out[k] = aniso_qf1(t, a[k]); break;
		case 3: out[k] = aniso_pdf2(t, a[k], b[k]); break;
		case 4: out[k] = aniso_cdf2(t, a[k], b[k]); break;
		default: out[k] = aniso_qf2(t, a[k], b[k]); break;
		}
	}
}

void o_aniso_fit(const o_brdf *t, float *bk, float *gg) /* hdr:3186-3307 */
{
	const int ntheta = 128, nphi = 512;
	float dtheta = F(sqrt(O_PI * 0.5) / D((float)ntheta)), dphi = F(2.0 * O_PI / D((float)nphi));
	float nb[5] = { 0, 0, 0, 0, 0 }, ng[5] = { 0, 0, 0, 0, 0 };
	for (int j = 0; j < nphi; ++j) {
		float phi = F(D((float)j / (float)nphi) * 2.0 * O_PI);
		float cp = F(cos(D(phi))), sp = F(sin(D(phi)));
		float cp2 = cp * cp, sp2 = sp * sp;
		for (int i = 0; i < ntheta; ++i) {
			float theta = F(D((float)i / (float)ntheta) * sqrt(O_PI * 0.5));
			float ts = theta * theta;
			float p22 = aniso_p22_theta_phi(t, ts, phi);
			float tt = F(tan(D(ts))), ct = F(cos(D(ts)));
			float tt2 = tt * tt, ct2 = ct * ct;
			float tmp2 = theta * p22 * tt / ct2;
			float e1 = -tt * cp, e2 = -tt * sp;
			nb[0] += tmp2 * e1; nb[1] += tmp2 * e2;
			nb[2] += tmp2 * (tt2 * cp2); nb[3] += tmp2 * (tt2 * sp2); nb[4] += tmp2 * (tt2 * cp * sp);
			ng[0] += tmp2 * e1; ng[1] += tmp2 * e2;
			ng[2] += tmp2 * F(fabs(D(e1))); ng[3] += tmp2 * F(fabs(D(e2))); ng[4] += tmp2 * 0.0f;
		}
	}
	for (int i = 0; i < 5; ++i) {
		nb[i] = F(D(nb[i]) * (2.0 * D(dtheta) * D(dphi)));
		ng[i] = F(D(ng[i]) * (2.0 * D(dtheta) * D(dphi)));
	}
	float mux = nb[0], muy = nb[1];
	bk[0] = F(sqrt(D(2.0f * (nb[2] - mux * mux))));
	bk[1] = F(sqrt(D(2.0f * (nb[3] - muy * muy))));
	bk[2] = F(2.0 * D(nb[4] - mux * muy) / D(bk[0] * bk[1]));
	bk[3] = mux; bk[4] = muy;
	mux = ng[0]; muy = ng[1];
	gg[0] = F(sqrt(D(ng[2] * ng[2] - mux * mux)));
	gg[1] = F(sqrt(D(ng[3] * ng[3] - muy * muy)));
	gg[2] = 0.0f; gg[3] = mux; gg[4] = muy;
}

void o_tabular_fit(const o_brdf *t, float *alpha_beckmann, float *alpha_ggx) /* hdr:3133-3184 */
{
	const int ntheta = 128;
	float dtheta = F(O_PI / D((float)ntheta));
	float nb = 0.0f, ng = 0.0f;
	for (int i = 0; i < ntheta; ++i) {
		float u = (float)i / (float)ntheta;
		float theta_h = F(D(u * u) * O_PI * 0.5);
		float cos_h = F(cos(D(theta_h))), r_h = F(tan(D(theta_h)));
		float r2 = r_h * r_h;
		float p22_r = tab_p22_radial(t, r2);
		nb += (u * r2 * r_h * p22_r) / (cos_h * cos_h);
		ng += (u * r2 * p22_r) / (cos_h * cos_h);
	}
	nb = F(D(nb) * (D(dtheta) * O_PI));
	ng = F(D(ng) * (D(dtheta) * 4.0));
	*alpha_beckmann = F(sqrt(2.0 * D(nb)));
	*alpha_ggx = ng;
}

int o_tabular_get(const o_brdf *t, int which, float *out)
{
	const float *src; int n;
	switch (which) {
	case 0: src = t->p22; n = t->n_p22; break;
	case 1: src = t->sigma; n = t->n_sigma; break;
	case 2: src = t->cdf; n = t->n_cdf; break;
	case 3: src = t->qf; n = t->n_qf; break;
	default:
		if (t->fresnel.kind != O_FRESNEL_SPLINE) return 0;
		src = &t->fresnel.pts[0].x; n = t->fresnel.npts;
		if (out) memcpy(out, src, sizeof(float) * 3 * n);
		return n;
	}
	if (out) memcpy(out, src, sizeof(float) * n);
	return n;
}

/* ------------------------------------------------------------------ beckmann::lrep (hdr:1959-2051) */
typedef struct { float E1, E2, E3, E4, E5; } o_lrep;
static o_vec3 ld3(const float *p, int64_t k);
static void st3(float *p, int64_t k, o_vec3 v);

static o_lrep lrep_add(o_lrep a, o_lrep r) /* hdr:1992-1999 */
{
	o_lrep o = { a.E1 + r.E1, a.E2 + r.E2, a.E3 + r.E3 + 2.0f * a.E1 * r.E1,
	             a.E4 + r.E4 + 2.0f * a.E2 * r.E2, a.E5 + r.E5 + a.E1 * r.E2 + a.E2 * r.E1 };
	return o;
}
static o_lrep lrep_mul(o_lrep a, float sc) /* hdr:2001-2009 */
{
	float s2 = sc * sc;
	o_lrep o = { a.E1 * sc, a.E2 * sc, a.E3 * s2, a.E4 * s2, a.E5 * s2 };
	return o;
}
static o_lrep lrep_add_assign(o_lrep a, o_lrep r) /* hdr:2011-2020: uses the UPDATED E1/E2 */
{
	a.E1 += r.E1; a.E2 += r.E2;
	a.E3 += r.E3 + 2.0f * a.E1 * r.E1;
	a.E4 += r.E4 + 2.0f * a.E2 * r.E2;
	a.E5 += r.E5 + a.E1 * r.E2 + a.E2 * r.E1;
	return a;
}
static o_lrep lrep_shear(o_lrep a, float tx, float ty) /* hdr:2035-2042 */
{
	a.E1 += tx; a.E2 += ty; a.E3 += tx * tx; a.E4 += ty * ty; a.E5 += tx * ty;
	return a;
}
static o_lrep lrep_scale_xy(o_lrep a, float x, float y) /* hdr:2044-2051 */
{
	a.E1 *= x; a.E2 *= y; a.E3 *= x * x; a.E4 *= y * y; a.E5 *= x * y;
	return a;
}
static o_lrep params_to_lrep(const o_params *p) /* hdr:1965-1974 */
{
	o_lrep l = { p->tx, p->ty, 0.5f * p->ax * p->ax + p->tx * p->tx, 0.5f * p->ay * p->ay + p->ty * p->ty,
	             0.5f * p->rho * p->ax * p->ay + p->tx * p->ty };
	return l;
}
static void lrep_to_pdfparams(o_lrep l, float *out5) /* hdr:1976-1990 */
{
	float t1 = fmaxf_(0.0f, l.E3 - l.E1 * l.E1), t2 = fmaxf_(0.0f, l.E4 - l.E2 * l.E2);
	float ax = F(dmax(1e-5, sqrt(2.0 * D(t1)))), ay = F(dmax(1e-5, sqrt(2.0 * D(t2))));
	float rho = 2.0f * (l.E5 - l.E1 * l.E2) / (ax * ay);
	rho = fminf_(0.99f, fmaxf_(-0.99f, rho));
	out5[0] = ax; out5[1] = ay; out5[2] = rho; out5[3] = l.E1; out5[4] = l.E2;
}
static o_lrep lrep_from(const float *a) { o_lrep l = { a[0], a[1], a[2], a[3], a[4] }; return l; }

/* op: 0 a+b 1 a*x 2 a+=b 3 a*=x 4 shear(x,y) 5 scale(x,y); raw5 (optional) gets the moments */
void o_lrep_op_raw(int op, const float *a, const float *b, float x, float y, float *raw5, float *out_pdfparams)
{
	o_lrep A = lrep_from(a), B = { 0, 0, 1, 1, 0 }, R;
	if (b) B = lrep_from(b);
	switch (op) {
	case 0: R = lrep_add(A, B); break;
	case 1: case 3: R = lrep_mul(A, x); break;
	case 2: R = lrep_add_assign(A, B); break;
	case 4: R = lrep_shear(A, x, y); break;
	default: R = lrep_scale_xy(A, x, y); break;
	}
	if (raw5) { raw5[0] = R.E1; raw5[1] = R.E2; raw5[2] = R.E3; raw5[3] = R.E4; raw5[4] = R.E5; }
	if (out_pdfparams) lrep_to_pdfparams(R, out_pdfparams);
}
void o_lrep_op(int op, const float *a, const float *b, float x, float y, float *out_pdfparams)
{
	o_lrep_op_raw(op, a, b, x, y, NULL, out_pdfparams);
}
void o_params_lrep_roundtrip(const o_param_desc *pd, float *out_pdfparams)
{
	o_params p = params_from_desc(pd);
	lrep_to_pdfparams(params_to_lrep(&p), out_pdfparams);
}
/* one hit of the plugin (mitsuba/dj_beckmannconductor.cpp:296-314; again at 344-362, 384-402):
 *   flags & 2: the record is a raw texel, E1 -= BIAS, E2 -= BIAS, E5 -= BIAS*BIAS (BIAS = 25.f, l.300-303)
 *   lrep1 = lrep(E1..E5), or with flags & 1 (leanFiltering = false) lrep(E1, E2, E1*E1, E2*E2, E1*E2)   (l.306-309)
 *   lrep1 *= dmapscale (l.311, hdr:2022-2033); lrep2 = params_to_lrep(base) (l.312); lrep_to_params(lrep1 + lrep2) (l.314) */
static void lean_hit_pdfparams(const float *rec, const o_params *base, float scale, int flags, float *pp)
{
	float E1 = rec[0], E2 = rec[1], E3 = rec[2], E4 = rec[3], E5 = rec[4];
	const float BIAS = 25.f;
	if (flags & 2) { E1 -= BIAS; E2 -= BIAS; E5 -= BIAS * BIAS; }
	o_lrep l1 = { E1, E2, E3, E4, E5 };
	if (flags & 1) { o_lrep naive = { E1, E2, E1 * E1, E2 * E2, E1 * E2 }; l1 = naive; }
	l1 = lrep_mul(l1, scale);          /* operator*= computes the same values as operator* (hdr:2001-2009 vs 2022-2033) */
	lrep_to_pdfparams(lrep_add(l1, params_to_lrep(base)), pp);
}
void o_lean_params(int64_t n, const o_param_desc *base, float scale, int flags, const float *lean, float *out_pdfparams)
{
	o_params p0 = params_from_desc(base);
	for (int64_t k = 0; k < n; ++k) lean_hit_pdfparams(lean + 5 * k, &p0, scale, flags, out_pdfparams + 5 * k);
}
void o_eval_lean(const o_brdf *b, int op, int64_t n, const float *i, const float *o, const o_param_desc *base,
                 float scale, int flags, const float *lean, float *out, float *out_pdfparams)
{
	o_params p0 = params_from_desc(base);
	for (int64_t k = 0; k < n; ++k) {
		float pp[5];
		lean_hit_pdfparams(lean + 5 * k, &p0, scale, flags, pp);
		if (out_pdfparams) memcpy(out_pdfparams + 5 * k, pp, sizeof pp);
		o_params p;
		params_set_pdfparams(&p, pp[0], pp[1], pp[2], pp[3], pp[4]);
		o_vec3 vi = ld3(i, k), vo = ld3(o, k);
		if (op == 0) st3(out, k, brdf_eval(b, vi, vo, &p));
		else if (op == 1) st3(out, k, brdf_evalp(b, vi, vo, &p));
		else out[k] = brdf_pdf(b, vi, vo, &p);
	}
}
/* dj_beckmann_conductor::sample per hit (mitsuba/dj_beckmannconductor.cpp:373-413): params as above, then
 * evalp_is(u1, u2, o, &i, &pdf, &params) (hdr:1734-1765); is == 0: sample() only (hdr:1669-1709) */
void o_sample_lean(const o_brdf *b, int is, int64_t n, const float *u1, const float *u2, const float *o,
                   const o_param_desc *base, float scale, int flags, const float *lean, float *out_w, float *out_i,
                   float *out_pdf, float *out_pdfparams)
{
	o_params p0 = params_from_desc(base);
	for (int64_t k = 0; k < n; ++k) {
		float pp[5];
		lean_hit_pdfparams(lean + 5 * k, &p0, scale, flags, pp);
		if (out_pdfparams) memcpy(out_pdfparams + 5 * k, pp, sizeof pp);
		o_params p;
		params_set_pdfparams(&p, pp[0], pp[1], pp[2], pp[3], pp[4]);
		o_vec3 vo = ld3(o, k);
		if (!is) { st3(out_i, k, mf_sample(b, u1[k], u2[k], vo, &p)); continue; }
		o_vec3 vi = v3(0, 0, 0); float pdf = 0;
		o_vec3 w = mf_evalp_is(b, u1[k], u2[k], vo, &p, &vi, &pdf);
		st3(out_w, k, w); st3(out_i, k, vi); out_pdf[k] = pdf;
	}
}
/* per-pair pdfparams (n x 5: ax, ay, rho, tx, ty) */
void o_eval_pp(const o_brdf *b, int op, int64_t n, const float *i, const float *o, const float *pp, float *out)
{
	for (int64_t k = 0; k < n; ++k) {
		o_params p;
		params_set_pdfparams(&p, pp[5 * k], pp[5 * k + 1], pp[5 * k + 2], pp[5 * k + 3], pp[5 * k + 4]);
		o_vec3 vi = ld3(i, k), vo = ld3(o, k);
		if (op == 0) st3(out, k, brdf_eval(b, vi, vo, &p));
		else if (op == 1) st3(out, k, brdf_evalp(b, vi, vo, &p));
		else out[k] = brdf_pdf(b, vi, vo, &p);
	}
}

/* ------------------------------------------------------------------ construction */
o_brdf *o_create_microfacet(int ndf, int fkind, const float *fd, int nf, int shadow)
{
	o_brdf *b = (o_brdf *)calloc(1, sizeof *b);
	b->kind = ndf == 0 ? O_BRDF_BECKMANN : O_BRDF_GGX;
	b->shadow = shadow != 0;
	b->fresnel.kind = fkind;
	if (fkind == O_FRESNEL_UNPOLARIZED || fkind == O_FRESNEL_SCHLICK || fkind == O_FRESNEL_SGD)
		b->fresnel.a = v3(fd[0], fd[1], fd[2]);
	if (fkind == O_FRESNEL_SGD) b->fresnel.b = v3(fd[3], fd[4], fd[5]);
	if (fkind == O_FRESNEL_CUSTOM) { b->fresnel.a = v3(fd[0], fd[1], fd[2]); b->fresnel.b = v3(fd[3], 0, 0); }
	if (fkind == O_FRESNEL_SPLINE) {
		b->fresnel.npts = nf;
		b->fresnel.pts = (o_vec3 *)malloc(sizeof(o_vec3) * nf);
		memcpy(b->fresnel.pts, fd, sizeof(o_vec3) * nf);
	}
	return b;
}

o_brdf *o_create_merl_from_memory(const double *samples, int64_t n)
{
	if (n <= 0) { set_err("djb_error: Failed to read MERL header\n"); return NULL; }
	o_brdf *b = (o_brdf *)calloc(1, sizeof *b);
	b->kind = O_BRDF_MERL;
	b->n_samples = 3 * n;
	b->samples = (double *)malloc(sizeof(double) * 3 * n);
	memcpy(b->samples, samples, sizeof(double) * 3 * n);
	return b;
}

o_brdf *o_create_merl(const char *path) /* hdr:963-983 */
{
	FILE *f = fopen(path, "rb");
	if (!f) { set_err("djb_error: Failed to open %s\n", path); return NULL; }
	int32_t dims[3];
	if (fread(dims, 4, 3, f) != 3) dims[0] = dims[1] = dims[2] = 0;
	int32_t n = dims[0] * dims[1] * dims[2];
	if (n <= 0) { fclose(f); set_err("djb_error: Failed to read MERL header\n"); return NULL; }
	o_brdf *b = (o_brdf *)calloc(1, sizeof *b);
	b->kind = O_BRDF_MERL;
	b->n_samples = 3 * (int64_t)n;
	b->samples = (double *)malloc(sizeof(double) * b->n_samples);
	size_t got = fread(b->samples, sizeof(double), (size_t)b->n_samples, f);
	fclose(f);
	if ((int64_t)got != b->n_samples) {
		o_destroy(b); set_err("djb_error: Reading %s failed\n", path); return NULL;
	}
	return b;
}

#define UTIA_CNT (3 * 288 * 288)
static void utia_normalize(o_brdf *b) /* hdr:1162-1177 */
{
	float k = 1.f / 140.f;
	for (int i = 0; i < UTIA_CNT; ++i) {
		double s = b->samples[i] > 0.0 ? b->samples[i] : 0.0;
		b->samples[i] = s * D(k);
	}
}
o_brdf *o_create_utia_from_memory(const double *samples)
{
	o_brdf *b = (o_brdf *)calloc(1, sizeof *b);
	b->kind = O_BRDF_UTIA;
	b->n_samples = UTIA_CNT;
	b->samples = (double *)malloc(sizeof(double) * UTIA_CNT);
	memcpy(b->samples, samples, sizeof(double) * UTIA_CNT);
	utia_normalize(b);
	return b;
}
o_brdf *o_create_utia(const char *path) /* hdr:1039-1059 */
{
	FILE *f = fopen(path, "rb");
	if (!f) { set_err("djb_error: Failed to open %s\n", path); return NULL; }
	double *tmp = (double *)calloc(UTIA_CNT, sizeof(double));
	size_t got = fread(tmp, sizeof(double), UTIA_CNT, f);
	fclose(f);
	if (got != UTIA_CNT) { free(tmp); set_err("djb_error: Reading %s failed\n", path); return NULL; }
	o_brdf *b = o_create_utia_from_memory(tmp);
	free(tmp);
	return b;
}

o_brdf *o_create_sgd(const double *p) /* hdr:3436-3450: fresnel::sgd(f0, f1) */
{
	o_brdf *b = (o_brdf *)calloc(1, sizeof *b);
	b->kind = O_BRDF_SGD;
	memcpy(b->model, p, sizeof(double) * 33);
	b->fresnel.kind = O_FRESNEL_SGD;
	b->fresnel.a = v3(F(p[12]), F(p[13]), F(p[14]));
	b->fresnel.b = v3(F(p[15]), F(p[16]), F(p[17]));
	return b;
}

o_brdf *o_create_abc(const double *p) /* hdr:3617-3629: fresnel::unpolarized(vec3(ior)) */
{
	o_brdf *b = (o_brdf *)calloc(1, sizeof *b);
	b->kind = O_BRDF_ABC;
	memcpy(b->model, p, sizeof(double) * 9);
	b->fresnel.kind = O_FRESNEL_UNPOLARIZED;
	b->fresnel.a = v3(F(p[8]), F(p[8]), F(p[8]));
	return b;
}

o_brdf *o_create_custom(int which, const float *params, int n)
{
	if (which < 0 || which > 1 || n != (which == 0 ? 7 : 8)) { set_err("o_create_custom: bad arguments"); return NULL; }
	o_brdf *b = (o_brdf *)calloc(1, sizeof *b);
	b->kind = O_BRDF_CUSTOM;
	b->model[0] = which;
	for (int k = 0; k < n; ++k) b->model[1 + k] = params[k];
	return b;
}

o_brdf *o_create_lambert(void)
{
	o_brdf *b = (o_brdf *)calloc(1, sizeof *b);
	b->kind = O_BRDF_LAMBERT;
	return b;
}

void o_destroy(o_brdf *b)
{
	if (!b) return;
	free(b->fresnel.pts); free(b->p22); free(b->sigma); free(b->cdf); free(b->qf);
	free(b->a_p22); free(b->a_sigma); free(b->a_pdf1); free(b->a_cdf1); free(b->a_qf1);
	free(b->a_pdf2); free(b->a_cdf2); free(b->a_qf2);
	free(b->samples); free(b);
}

/* ------------------------------------------------------------------ batch entry points */
static o_vec3 ld3(const float *p, int64_t k) { return v3(p[3 * k], p[3 * k + 1], p[3 * k + 2]); }
static void st3(float *p, int64_t k, o_vec3 v) { p[3 * k] = v.x; p[3 * k + 1] = v.y; p[3 * k + 2] = v.z; }

void o_eval(const o_brdf *b, int op, int64_t n, const float *i, const float *o,
            const o_param_desc *pd, float *out)
{
	o_params p = params_from_desc(pd);
	for (int64_t k = 0; k < n; ++k) {
		o_vec3 vi = ld3(i, k), vo = ld3(o, k);
		if (op == 0) st3(out, k, brdf_eval(b, vi, vo, &p));
		else if (op == 1) st3(out, k, brdf_evalp(b, vi, vo, &p));
		else if (op == 3 || op == 4) { /* eval_hd / evalp_hd (hdr:795-801, 808-814): vi, vo hold h, d; both go through EVAL */
			o_vec3 wi, wo;
			hd_to_io(vi, vo, &wi, &wo);
			o_vec3 fr = brdf_eval(b, wi, wo, &p);
			st3(out, k, op == 3 ? fr : v3_scale(wi.z, fr));
		}
		else out[k] = brdf_pdf(b, vi, vo, &p);
	}
}

typedef struct {
	const o_brdf *b; int op; int64_t n; const float *i, *o; const o_param_desc *pd; float *out;
} mt_job;
static void *mt_worker(void *arg)
{
	mt_job *j = (mt_job *)arg;
	o_eval(j->b, j->op, j->n, j->i, j->o, j->pd, j->out);
	return NULL;
}
void o_eval_mt(const o_brdf *b, int op, int64_t n, const float *i, const float *o,
               const o_param_desc *pd, float *out, int threads)
{
	if (threads < 1) threads = 1;
	if (threads > 256) threads = 256;
	pthread_t th[256]; mt_job jobs[256];
	int64_t chunk = (n + threads - 1) / threads;
	int ostride = op == 2 ? 1 : 3;
	int used = 0;
	for (int t = 0; t < threads; ++t) {
		int64_t lo = t * chunk, hi = lo + chunk > n ? n : lo + chunk;
		if (lo >= hi) break;
		mt_job jb = { b, op, hi - lo, i + 3 * lo, o + 3 * lo, pd, out + ostride * lo };
		jobs[t] = jb;
		pthread_create(&th[t], NULL, mt_worker, &jobs[t]);
		used++;
	}
	for (int t = 0; t < used; ++t) pthread_join(th[t], NULL);
}

void o_sample(const o_brdf *b, int64_t n, const float *u1, const float *u2, const float *o,
              const o_param_desc *pd, float *out_i)
{
	o_params p = params_from_desc(pd);
	for (int64_t k = 0; k < n; ++k) st3(out_i, k, brdf_sample(b, u1[k], u2[k], ld3(o, k), &p));
}

void o_evalp_is(const o_brdf *b, int64_t n, const float *u1, const float *u2, const float *o,
                const o_param_desc *pd, float *out_w, float *out_i, float *out_pdf)
{
	o_params p = params_from_desc(pd);
	for (int64_t k = 0; k < n; ++k) {
		o_vec3 vi = v3(0, 0, 0), vo = ld3(o, k), w; float pdf = 0;
		if (is_microfacet(b)) w = mf_evalp_is(b, u1[k], u2[k], vo, &p, &vi, &pdf);
		else { /* brdf::evalp_is, hdr:816-828 */
			vi = brdf_sample(b, u1[k], u2[k], vo, &p);
			pdf = brdf_pdf(b, vi, vo, &p);
			w = v3_div(brdf_evalp(b, vi, vo, &p), pdf);
		}
		st3(out_w, k, w); st3(out_i, k, vi); out_pdf[k] = pdf;
	}
}

void o_io_to_hd(int64_t n, const float *i, const float *o, float *h, float *d)
{
	for (int64_t k = 0; k < n; ++k) { o_vec3 vh, vd; io_to_hd(ld3(i, k), ld3(o, k), &vh, &vd); st3(h, k, vh); st3(d, k, vd); }
}
void o_hd_to_io(int64_t n, const float *h, const float *d, float *i, float *o)
{
	for (int64_t k = 0; k < n; ++k) { o_vec3 vi, vo; hd_to_io(ld3(h, k), ld3(d, k), &vi, &vo); st3(i, k, vi); st3(o, k, vo); }
}
void o_merl_index(int64_t n, const float *i, const float *o, int *idx)
{
	for (int64_t k = 0; k < n; ++k) idx[k] = merl_index(ld3(i, k), ld3(o, k));
}

void o_params_get(const o_param_desc *pd, float *out)
{
	o_params p = params_from_desc(pd);
	out[0] = p.n.x; out[1] = p.n.y; out[2] = p.n.z;
	out[3] = p.a1; out[4] = p.a2; out[5] = p.phi_a;
	out[6] = p.ax; out[7] = p.ay; out[8] = p.rho; out[9] = p.tx; out[10] = p.ty; out[11] = 0;
}

void o_microfacet_query(const o_brdf *b, int which, int64_t n, const float *a, const float *bb,
                        const float *c, const o_param_desc *pd, float *out)
{
	o_params p = params_from_desc(pd);
	for (int64_t k = 0; k < n; ++k) {
		switch (which) {
		case 0: out[k] = mf_ndf(b, ld3(a, k), &p); break;
		case 1: out[k] = mf_gaf(b, ld3(bb, k), ld3(c, k), &p); break;
		case 2: out[k] = mf_g1(b, ld3(bb, k), &p); break;
		case 3: out[k] = mf_sigma(b, ld3(a, k), &p); break;
		case 4: out[k] = mf_p22(b, a[3 * k], a[3 * k + 1], &p); break;
		case 5: out[k] = mf_vp22(b, a[3 * k], a[3 * k + 1], ld3(bb, k), &p); break;
		case 6: out[k] = mf_vndf(b, ld3(a, k), ld3(bb, k), &p); break;
		}
	}
}

void o_radial_query(const o_brdf *b, int which, int64_t n, const float *a, const float *bb,
                    const float *c, float *out)
{
	for (int64_t k = 0; k < n; ++k) {
		switch (which) {
		case 0: out[k] = p22_radial(b, a[k]); break;
		case 1: out[k] = sigma_std_radial(b, a[k]); break;
		case 2: out[k] = cdf_radial(b, a[k]); break;
		case 3: out[k] = qf_radial(b, a[k]); break;
		case 4: out[k] = b->kind == O_BRDF_BECKMANN ? beckmann_qf2_radial(a[k], bb[k], c[k])
		                                            : ggx_qf2_radial(a[k], bb[k], c[k]); break;
		case 5: out[k] = b->kind == O_BRDF_BECKMANN ? beckmann_qf1(a[k])
		                                            : ggx_qf3_radial(a[k], bb[k]); break;
		case 6: out[k] = b->kind == O_BRDF_BECKMANN ? beckmann_qf1(a[k]) : ggx_qf1(a[k]); break;
		}
	}
}

void o_fresnel_eval(const o_brdf *b, int64_t n, const float *c, float *out)
{
	for (int64_t k = 0; k < n; ++k) st3(out, k, fresnel_eval(&b->fresnel, c[k]));
}
/* dj_brdf.h:1255-1261: tmp = (ior - 1.0) / (ior + 1.0) in double, stored as float; f0 = tmp * tmp (float).
 * dj_brdf.h:1272-1282: f0 == 1.0 -> 1; else sqrt_f0 = float(sqrt(double f0)); ior = float((1.0 + s) / (1.0 - s)). */
void o_ior_f0(int dir, int64_t n, const float *x, float *y)
{
	for (int64_t k = 0; k < n; ++k) {
		if (dir == 0) {
			float tmp = (float)(((double)x[k] - 1.0) / ((double)x[k] + 1.0));
			y[k] = tmp * tmp;
		} else if ((double)x[k] == 1.0) {
			y[k] = 1.0f;
		} else {
			float s = (float)sqrt((double)x[k]);
			y[k] = (float)((1.0 + (double)s) / (1.0 - (double)s));
		}
	}
}

void o_erf(int64_t n, const float *x, float *y) { for (int64_t k = 0; k < n; ++k) y[k] = erf_(x[k]); }
void o_erfinv(int64_t n, const float *x, float *y) { for (int64_t k = 0; k < n; ++k) y[k] = erfinv_(x[k]); }

/* ------------------------------------------------------------------ glibc 2.35 float libm, restated
 * The reference's erfinv / Beckmann quantile code calls logf, expf and powf (hdr:691-721, 1897-1952): its
 * results are those of the host libm.  o_libm_f32 calls the host libm itself (what the reference build
 * does); o_glibc_f32 is the restatement of glibc's algorithms (sysdeps/ieee754/flt-32/e_logf.c, e_expf.c,
 * e_powf.c = ARM optimized-routines) that the HIP kernels implement: complete for logf and expf, and for powf
 * with a positive finite base (zero / Inf / NaN arguments and negative bases go to the host libm here, to the
 * device libm in the kernels); `fma` selects the contraction of the x86-64 FMA ifunc variants.
 * fn: 0 logf(x), 1 expf(x), 2 powf(x, y). */
#include "glibc_flt32_tables.h"
static inline uint32_t asu32(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float asf32(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint64_t asu64(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }
static inline double asf64(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }
static inline double mad(int use_fma, double a, double b, double c) { return use_fma ? fma(a, b, c) : a * b + c; }

static float glibc_logf(float x, int f)
{
	const double *T = DJB_GLIBC_LOGF, Ln2 = T[32], *A = T + 33;
	uint32_t ix = asu32(x);
	if (ix == 0x3f800000) return 0.0f;
	if (ix - 0x00800000 >= 0x7f800000 - 0x00800000) {
		if (ix * 2 == 0) return -INFINITY;
		if (ix == 0x7f800000) return x;
		if ((ix & 0x80000000) || ix * 2 >= 0xff000000) return NAN;
		ix = asu32(x * 0x1p23f);                                                   /* subnormal: normalise */
		ix -= 23 << 23;
	}
	uint32_t tmp = ix - 0x3f330000;
	int i = (tmp >> 19) % 16, k = (int32_t)tmp >> 23;
	uint32_t iz = ix - (tmp & 0x1ffu << 23);
	double invc = T[2 * i], logc = T[2 * i + 1], z = (double)asf32(iz);
	double r = mad(f, z, invc, -1.0);
	double y0 = mad(f, (double)k, Ln2, logc);
	double r2 = r * r;
	double y = mad(f, A[1], r, A[2]);
	y = mad(f, A[0], r2, y);
	y = mad(f, y, r2, y0 + r);
	return (float)y;
}
static float glibc_expf(float x, int f)
{
	const double *E = DJB_GLIBC_EXP2F, InvLn2N = E[5], SHIFT = E[4], *C = E + 6;
	uint32_t abstop = (asu32(x) >> 20) & 0x7ff;
	if (abstop >= (asu32(88.0f) >> 20)) {                                         /* |x| >= 88 or nan */
		if (asu32(x) == asu32(-INFINITY)) return 0.0f;
		if (abstop >= (asu32(INFINITY) >> 20)) return x + x;
		if (x > 0x1.62e42ep6f) return INFINITY;                                   /* x > log(0x1p128) */
		if (x < -0x1.9fe368p6f) return 0.0f;                                      /* x < log(0x1p-150) */
		if (x < -0x1.9d1d9ep6f) return 0x1.4p-75f * 0x1.4p-75f;                   /* x < log(0x1p-149): __math_may_uflowf */
	}
	double xd = (double)x, z = InvLn2N * xd;
	double kd = z + SHIFT; uint64_t ki = asu64(kd); kd -= SHIFT;
	double r = f >= 2 ? fma(InvLn2N, xd, -kd) : z - kd;      /* f >= 2: the product contracted into the subtraction */
	uint64_t t = DJB_GLIBC_EXP2F_TAB[ki % 32]; t += ki << (52 - 5);
	double s = asf64(t);
	z = mad(f, C[0], r, C[1]);
	double r2 = r * r;
	double y = mad(f, C[2], r, 1.0);
	y = mad(f, z, r2, y);
	y = y * s;
	return (float)y;
}
static float glibc_powf(float x, float y, int f)
{
	const double *T = DJB_GLIBC_POWF_LOG2, *A = T + 32, *E = DJB_GLIBC_EXP2F, SHIFT = E[0], *C = E + 1;
	uint32_t ix = asu32(x), iy = asu32(y);
	if (ix - 0x00800000 >= 0x7f800000 - 0x00800000 || 2 * iy - 1 >= 2u * 0x7f800000 - 1) {
		/* zero / Inf / NaN arguments and negative bases: exact special values (or the sign_bias path), host libm */
		if (2 * iy - 1 >= 2u * 0x7f800000 - 1 || 2 * ix - 1 >= 2u * 0x7f800000 - 1 || (ix & 0x80000000)) return powf(x, y);
		ix = asu32(x * 0x1p23f);                                                   /* positive subnormal: normalise */
		ix &= 0x7fffffff;
		ix -= 23 << 23;
	}
	uint32_t tmp = ix - 0x3f330000;
	int i = (tmp >> 19) % 16;
	uint32_t top = tmp & 0xff800000, iz = ix - top;
	int k = (int32_t)top >> 23;
	double invc = T[2 * i], logc = T[2 * i + 1], z = (double)asf32(iz);
	double r = mad(f, z, invc, -1.0), y0 = logc + (double)k;
	double r2 = r * r;
	double p0 = mad(f, A[0], r, A[1]), p = mad(f, A[2], r, A[3]), r4 = r2 * r2;
	double q = mad(f, A[4], r, y0);
	q = mad(f, p, r2, q);
	double logx = mad(f, p0, r4, q);
	double ylogx = (double)y * logx;
	if ((asu64(ylogx) >> 47 & 0xffff) >= asu64(126.0) >> 47) {                     /* |y log2 x| >= 126 */
		if (ylogx > 0x1.fffffffd1d571p+6) return INFINITY;
		if (ylogx <= -150.0) return 0.0f;
		if (ylogx < -149.0) return 0x1.4p-75f * 0x1.4p-75f;                       /* __math_may_uflowf */
	}
	double kd = ylogx + SHIFT; uint64_t ki = asu64(kd); kd -= SHIFT;
	double rr = f >= 2 ? fma((double)y, logx, -kd) : ylogx - kd;
	uint64_t t = DJB_GLIBC_EXP2F_TAB[ki % 32]; t += ki << (52 - 5);
	double s = asf64(t);
	double zz = mad(f, C[0], rr, C[1]), rr2 = rr * rr, yy = mad(f, C[2], rr, 1.0);
	yy = mad(f, zz, rr2, yy);
	return (float)(yy * s);
}
void o_libm_f32(int fn, int64_t n, const float *x, const float *y, float *out)
{
	for (int64_t k = 0; k < n; ++k) out[k] = fn == 0 ? logf(x[k]) : fn == 1 ? expf(x[k]) : powf(x[k], y[k]);
}
void o_glibc_f32(int fn, int use_fma, int64_t n, const float *x, const float *y, float *out)
{
	for (int64_t k = 0; k < n; ++k)
		out[k] = fn == 0 ? glibc_logf(x[k], use_fma) : fn == 1 ? glibc_expf(x[k], use_fma) : glibc_powf(x[k], y[k], use_fma);
}

/* ------------------------------------------------------------------ glibc 2.35 double exp / pow, restated
 * The reference's unqualified exp() / pow() are the host libm's double functions.  o_libm_f64
 * calls the host libm itself; o_glibc_f64 restates glibc's algorithms (sysdeps/ieee754/dbl-64/e_exp.c,
 * e_pow.c) as the HIP kernels implement them: the main paths, with the multiply-add fusion of the x86-64 FMA
 * ifunc variants (__exp_fma, __pow_fma; read off their disassembly -- every a*b+c of the source is one fma,
 * except in specialcase()).  exp is complete; pow hands zero / negative / subnormal / Inf / NaN bases and
 * exponents outside [2^-65, 2^63) -- whose results are exact special values -- to the host libm here and to the
 * device libm in the kernels.   fn: 0 exp(x), 1 pow(x, y). */
#include "glibc_dbl64_tables.h"
/* specialcase() of e_exp.c: 512 <= |x| < 1024, the scale factor would over/underflow */
static double glibc_exp_specialcase(double tmp, uint64_t sbits, uint64_t ki)
{
	if ((ki & 0x80000000) == 0) {
		sbits -= 1009ull << 52;
		double scale = asf64(sbits);
		return 0x1p1009 * fma(scale, tmp, scale);
	}
	sbits += 1022ull << 52;
	double scale = asf64(sbits);
	double m = tmp * scale;                  /* not fused in __exp_fma: the product is used twice */
	double y = scale + m;
	if (y < 1.0) {
		double lo = scale - y + m;
		double hi = 1.0 + y;
		lo = 1.0 - hi + y + lo;
		y = (hi + lo) - 1.0;
		if (y == 0.0) y = 0.0;
	}
	return 0x1p-1022 * y;
}
/* the common part of exp() and pow()'s exp_inline(): 2^-54 <= |x| < 1024 */
static double glibc_exp_core(double x, double xtail, int special)
{
	const double *E = DJB_GLIBC_EXP_C;       /* invln2N, shift, negln2hiN, negln2loN, C2, C3, C4, C5 */
	double kd = fma(x, E[0], E[1]);
	uint64_t ki = asu64(kd);
	kd -= E[1];
	double r = fma(kd, E[2], x);
	r = fma(kd, E[3], r);
	r += xtail;                              /* pow's exp_inline; exp() itself has xtail == 0: r + 0.0 == r */
	uint64_t idx = 2 * (ki % 128);
	double tail = asf64(DJB_GLIBC_EXP_TAB[idx]);
	uint64_t sbits = DJB_GLIBC_EXP_TAB[idx + 1] + (ki << 45);
	double r2 = r * r;
	double p = fma(r, E[5], E[4]);           /* C2 + r C3 */
	double q = fma(r, E[7], E[6]);           /* C4 + r C5 */
	double t = fma(p, r2, tail + r);
	double tmp = fma(r2 * r2, q, t);
	if (special) return glibc_exp_specialcase(tmp, sbits, ki);
	double scale = asf64(sbits);
	return fma(scale, tmp, scale);
}
static double glibc_exp(double x)
{
	uint32_t abstop = (uint32_t)(asu64(x) >> 52) & 0x7ff;
	int special = 0;
	if (abstop - 0x3c9 >= 0x3f) {
		if (abstop < 0x3c9) return 1.0 + x;                  /* |x| < 2^-54 */
		if (abstop >= 0x409) {                               /* |x| >= 1024, inf, nan */
			if (asu64(x) == asu64(-INFINITY)) return 0.0;
			if (abstop >= 0x7ff) return 1.0 + x;
			return (asu64(x) >> 63) ? 0.0 : INFINITY;
		}
		special = 1;
	}
	return glibc_exp_core(x, 0.0, special);
}
static double glibc_pow(double x, double y)
{
	const double *P = DJB_GLIBC_POW_C, *A = P + 2;    /* ln2hi, ln2lo, A[0..6] */
	uint64_t ix = asu64(x), iy = asu64(y);
	uint32_t topx = (uint32_t)(ix >> 52), topy = (uint32_t)(iy >> 52);
	/* zero, negative, subnormal, Inf, NaN bases and |y| outside [2^-65, 2^63): host libm (exact special values) */
	if (topx - 0x001 >= 0x7ff - 0x001 || (topy & 0x7ff) - 0x3be >= 0x43e - 0x3be) {
		if (ix == 0 && (topy & 0x7ff) - 0x3be < 0x43e - 0x3be) return (iy >> 63) ? INFINITY : 0.0;   /* +0 base: x*x or 1/(x*x) */
		return pow(x, y);
	}
	uint64_t tmp = ix - 0x3fe6955500000000ull;
	int i = (int)((tmp >> 45) % 128);
	int64_t k = (int64_t)tmp >> 52;
	uint64_t iz = ix - (tmp & 0xfffull << 52);
	double z = asf64(iz), kd = (double)k;
	const double *T = DJB_GLIBC_POW_LOG_TAB + 3 * i;  /* invc, logc, logctail */
	double r = fma(z, T[0], -1.0);
	double t1 = fma(kd, P[0], T[1]);
	double t2 = t1 + r;
	double lo1 = fma(kd, P[1], T[2]);
	double lo2 = t1 - t2 + r;
	double ar = A[0] * r, ar2 = r * ar, ar3 = r * ar2;
	double hi = t2 + ar2;
	double lo3 = fma(ar, r, -ar2);
	double lo4 = t2 - hi + ar2;
	double p1 = fma(r, A[2], A[1]), p2 = fma(r, A[4], A[3]), p3 = fma(r, A[6], A[5]);
	double q = fma(p3, ar2, p2);
	double rr = fma(ar2, q, p1);
	double lo = fma(ar3, rr, lo1 + lo2 + lo3 + lo4);
	double lhi = hi + lo;
	double llo = hi - lhi + lo;
	double ehi = y * lhi;
	double elo = fma(y, llo, fma(lhi, y, -ehi));
	uint32_t abstop = (uint32_t)(asu64(ehi) >> 52) & 0x7ff;
	int special = 0;
	if (abstop - 0x3c9 >= 0x3f) {
		if (abstop < 0x3c9) return 1.0 + ehi;                /* |y log x| < 2^-54 */
		if (abstop >= 0x409) return (asu64(ehi) >> 63) ? 0.0 : INFINITY;
		special = 1;
	}
	return glibc_exp_core(ehi, elo, special);
}
/* ------------------------------------------------------------------ glibc 2.35 double atan2, restated
 * __ieee754_atan2 of sysdeps/ieee754/dbl-64/e_atan2.c (IBM Accurate Mathematical Library; since glibc 2.34 without the
 * multi-precision fall-back), as the x86-64 FMA ifunc variant of this image computes it: branch structure, operation
 * order and the placement of every fused multiply-add read off the disassembly of __ieee754_atan2_fma; the 241 x 7
 * table cij out of libm.so.6 (tools/extract_glibc_dbl64_tables.py).  u = min / max of the magnitudes by an IEEE division,
 * du its residual; u < 1/16: odd polynomial d3 .. d13; else Taylor expansion about the table point next to u; then
 * the quadrant identity with the two-term pi/2 or pi.  Complete (zeros, infinities, NaNs, the exponent-difference
 * shortcuts, the 2^+-500 rescaling).  The default rounding mode is assumed (the original saves and restores it).
 * o_libm_f64 / o_glibc_f64: fn 2 = atan2(x[k] = y-argument, y[k] = x-argument). */
static double glibc_atan2(double y, double x)
{
	static const double d3 = -0x1.5555555555555p-2, d5 = 0x1.99999999997fdp-3, d7 = -0x1.24924923f7603p-3,
	                    d9 = 0x1.c71c6e5129a3bp-4, d11 = -0x1.7458022b13c25p-4, d13 = 0x1.375f08b31cbcep-4,
	                    hpi = 0x1.921fb54442d18p+0, hpi1 = 0x1.1a62633145c07p-54, opi = 0x1.921fb54442d18p+1,
	                    opi1 = 0x1.1a62633145c07p-53, qpi = 0x1.921fb54442d18p-1, tqpi = 0x1.2d97c7f3321d2p+1,
	                    twom500 = 0x1p-500, two500 = 0x1p+500, inv16 = 0x1p-4, TWO52 = 0x1p+52, TWO8 = 0x1p+8;
	uint64_t bx, by;
	memcpy(&bx, &x, 8); memcpy(&by, &y, 8);
	const int32_t ux = (int32_t)(bx >> 32), uy = (int32_t)(by >> 32);
	const uint32_t dx = (uint32_t)bx, dy = (uint32_t)by;
	/* x = NaN or y = NaN */
	if ((ux & 0x7ff00000) == 0x7ff00000 && (((ux & 0xfffff) | dx) != 0)) return x + y;
	if ((uy & 0x7ff00000) == 0x7ff00000 && (((uy & 0xfffff) | dy) != 0)) return y + y;
	/* y = +-0 */
	if (uy == 0 && dy == 0) return ux < 0 ? opi : 0.0;
	if ((uint32_t)uy == 0x80000000u && dy == 0) return ux < 0 ? -opi : -0.0;
	/* x = +-0 */
	if (x == 0.0) return uy < 0 ? -hpi : hpi;
	/* x = +-Inf */
	if (ux == 0x7ff00000 && dx == 0) {
		if (uy == 0x7ff00000 && dy == 0) return qpi;
		if ((uint32_t)uy == 0xfff00000u && dy == 0) return -qpi;
		return uy < 0 ? -0.0 : 0.0;
	}
	if ((uint32_t)ux == 0xfff00000u && dx == 0) {
		if (uy == 0x7ff00000 && dy == 0) return tqpi;
		if ((uint32_t)uy == 0xfff00000u && dy == 0) return -tqpi;
		return uy < 0 ? -opi : opi;
	}
	/* y = +-Inf */
	if (uy == 0x7ff00000 && dy == 0) return hpi;
	if ((uint32_t)uy == 0xfff00000u && dy == 0) return -hpi;
	double ax = x < 0.0 ? -x : x, ay = y < 0.0 ? -y : y;
	const int32_t de = (uy & 0x7ff00000) - (ux & 0x7ff00000);
	/* either x/y or y/x is very close to zero */
	if (de >= 0x3900000) return y > 0.0 ? hpi : -hpi;
	if (de <= -0x3900000) {
		if (x > 0.0) return copysign(ay / ax, y);
		return y > 0.0 ? opi : -opi;
	}
	if (ax < twom500 || ay < twom500) { ax *= two500; ay *= two500; }
	if (ax > two500 || ay > two500) { ax *= twom500; ay *= twom500; }
	double u, du, v, vv;
	const int y_lt_x = ay < ax;
	if (y_lt_x) { u = ay / ax; v = ax * u; vv = fma(ax, u, -v); du = ((ay - v) - vv) / ax; }
	else { u = ax / ay; v = ay * u; vv = fma(ay, u, -v); du = ((ax - v) - vv) / ay; }
	double z;
	if (u < inv16) {
		v = u * u;
		double p = fma(d13, v, d11);
		p = fma(p, v, d9); p = fma(p, v, d7); p = fma(p, v, d5); p = fma(p, v, d3);
		if (x > 0.0) {
			if (y_lt_x) z = u + fma(u * v, p, du);                                     /* (i)   atan(ay/ax) */
			else {                                                                     /* (ii)  pi/2 - atan(ax/ay) */
				const double zz = (u * v) * p, t2 = hpi - u;
				const double cor = hpi > fabs(u) ? (hpi - t2) - u : hpi - (u + t2);
				z = (((cor + hpi1) - du) - zz) + t2;
			}
		} else {
			if (!y_lt_x && ay > ax) {                                                  /* (iii) pi/2 + atan(ax/ay) */
				const double zz = (v * u) * p, t2 = u + hpi;
				const double cor = hpi > fabs(u) ? (hpi - t2) + u : (u - t2) + hpi;
				z = (((cor + hpi1) + du) + zz) + t2;
			} else {                                                                   /* (iv)  pi - atan(ay/ax) */
				const double zz = (v * u) * p, t2 = opi - u;
				const double cor = opi > fabs(u) ? (opi - t2) - u : opi - (t2 + u);
				z = (((cor + opi1) - du) - zz) + t2;
			}
		}
		return copysign(z, y);
	}
	const int i = (int)(fma(u, TWO8, TWO52) - TWO52) - 16;
	const double *c = DJB_GLIBC_ATAN_CIJ + 7 * i;
	const double t3 = u - c[0];
	if (x > 0.0 && y_lt_x) {                                                           /* (i) */
		const double w = du + t3;
		const double dv = fabs(t3) > fabs(du) ? (t3 - w) + du : (du - w) + t3;
		double p = fma(c[6], w, c[5]);
		p = fma(p, w, c[4]); p = fma(p, w, c[3]);
		p = (w * w) * p;
		p = fma(dv, c[2], p);
		z = fma(w, c[2], p) + c[1];
		return copysign(z, y);
	}
	const double w = t3 + du;
	double p = fma(c[6], w, c[5]);
	p = fma(p, w, c[4]); p = fma(p, w, c[3]); p = fma(p, w, c[2]);
	if (x > 0.0) z = (hpi - c[1]) + fma(-w, p, hpi1);                                    /* (ii) */
	else if (!y_lt_x && ay > ax) z = (hpi + c[1]) + fma(w, p, hpi1);                     /* (iii) */
	else z = (opi - c[1]) + fma(-w, p, opi1);                                            /* (iv) */
	return copysign(z, y);
}

/* ------------------------------------------------------------------ glibc 2.35 double sin / cos, restated
 * __sin / __cos of sysdeps/ieee754/dbl-64/s_sin.c (IBM Accurate Mathematical Library, as cleaned up in glibc 2.28:
 * no slow paths), as the x86-64 FMA ifunc variants compute them (operation order and fusion read off __sin_fma /
 * __cos_fma): |x| < 0.126: odd Taylor polynomial; else x = x_k + r with x_k = k/128 from the 440-entry __sincostab
 * (sin and cos of x_k as double-doubles) and short polynomials in r; 0.855 < |x| < 2.43 through pi/2 - |x|;
 * up to 105414350 a three-constant reduction by pi/2 (mp1, mp2, pp3, pp4); beyond that (__branred) the host libm is
 * called -- the BRDF code never gets there.  fn 3 = sin(x[k]), 4 = cos(x[k]). */
static const double SC_sn3 = -0x1.5555555555515p-3, SC_sn5 = 0x1.11110e829872fp-7, SC_cs2 = 0.5, SC_cs4 = -0x1.5555555555535p-5,
                    SC_cs6 = 0x1.6c16bedd9e239p-10, SC_s1 = -0x1.5555555555555p-3, SC_s2 = 0x1.1111111110ecep-7,
                    SC_s3 = -0x1.a01a019db08b8p-13, SC_s4 = 0x1.71de27b9a7ed9p-19, SC_s5 = -0x1.addffc2fcdf59p-26,
                    SC_big = 0x1.8p+45, SC_hp0 = 0x1.921fb54442d18p+0, SC_hp1 = 0x1.1a62633145c07p-54;
/* do_sin (s_sin.c): sin(x + dx), |x| < 0.855 */
static double glibc_do_sin(double x, double dx)
{
	const double ax = fabs(x);
	if (ax < 0.126) {                                   /* TAYLOR_SIN */
		const double xx = x * x;
		double p = fma(SC_s5, xx, SC_s4);
		p = fma(p, xx, SC_s3); p = fma(p, xx, SC_s2); p = fma(p, xx, SC_s1);
		return x + fma(fma(p, x, -(0.5 * dx)), xx, dx);
	}
	if (x <= 0.0) dx = -dx;
	const double u = SC_big + ax;
	uint64_t ub; memcpy(&ub, &u, 8);
	const double *T = DJB_GLIBC_SINCOS_TAB + 4 * (int32_t)(uint32_t)ub;
	const double r = ax - (u - SC_big), xx = r * r;
	const double s = r + fma(r * xx, fma(SC_sn5, xx, SC_sn3), dx);
	const double c = fma(r, dx, xx * fma(fma(SC_cs6, xx, SC_cs4), xx, SC_cs2));
	const double sn = T[0], ssn = T[1], cs = T[2], ccs = T[3];
	const double cor = fma(s, cs, fma(-c, sn, fma(s, ccs, ssn)));
	return copysign(sn + cor, x);
}
/* do_cos: cos(x + dx), |x| < 0.855 */
static double glibc_do_cos(double x, double dx)
{
	if (x < 0.0) dx = -dx;
	const double ax = fabs(x), u = SC_big + ax;
	uint64_t ub; memcpy(&ub, &u, 8);
	const double *T = DJB_GLIBC_SINCOS_TAB + 4 * (int32_t)(uint32_t)ub;
	const double r = (ax - (u - SC_big)) + dx, xx = r * r;
	const double s = fma(r * xx, fma(SC_sn5, xx, SC_sn3), r);
	const double c = xx * fma(fma(SC_cs6, xx, SC_cs4), xx, SC_cs2);
	const double sn = T[0], ssn = T[1], cs = T[2], ccs = T[3];
	const double cor = fma(-s, sn, fma(-c, cs, fma(-s, ssn, ccs)));
	return cs + cor;
}
/* reduce_sincos: x = n pi/2 + a + da, |a| <= pi/4, 2.43 < |x| < 105414350 */
static int glibc_reduce_sincos(double x, double *a, double *da)
{
	static const double toint = 0x1.8p+52, hpinv = 0x1.45f306dc9c883p-1, mp1 = 0x1.921fb58000000p+0, mp2 = -0x1.dde973c000000p-27,
	                    pp3 = -0x1.cb3b398000000p-55, pp4 = -0x1.d747f23e32ed7p-83;
	const double t = fma(x, hpinv, toint), xn = t - toint;
	uint64_t tb; memcpy(&tb, &t, 8);
	const double y = fma(-xn, mp2, fma(-xn, mp1, x));
	const double t2 = fma(-xn, pp3, y);
	double db = fma(-pp3, xn, y - t2);
	const double b = fma(-xn, pp4, t2);
	db = db + fma(-xn, pp4, t2 - b);
	*a = b; *da = db;
	return (int)(tb & 3);
}
static double glibc_do_sincos(double a, double da, int n)
{
	const double r = (n & 1) ? glibc_do_cos(a, da) : glibc_do_sin(a, da);
	return (n & 2) ? -r : r;
}
static double glibc_sin(double x)
{
	uint64_t b; memcpy(&b, &x, 8);
	const int32_t k = (int32_t)(b >> 32) & 0x7fffffff;
	if (k < 0x3e500000) return x;                                                   /* |x| < 2^-26 */
	if (k < 0x3feb6000) return glibc_do_sin(x, 0.0);                                /* |x| < 0.855469 */
	if (k < 0x400368fd) return copysign(glibc_do_cos(SC_hp0 - fabs(x), SC_hp1), x); /* |x| < 2.426265 */
	if (k < 0x419921fb) { double a, da; const int n = glibc_reduce_sincos(x, &a, &da); return glibc_do_sincos(a, da, n); }
	return sin(x);                                                                  /* __branred / Inf / NaN */
}
static double glibc_cos(double x)
{
	uint64_t b; memcpy(&b, &x, 8);
	const int32_t k = (int32_t)(b >> 32) & 0x7fffffff;
	if (k < 0x3e400000) return 1.0;                                                 /* |x| < 2^-27 */
	if (k < 0x3feb6000) return glibc_do_cos(x, 0.0);
	if (k < 0x400368fd) {
		const double y = SC_hp0 - fabs(x), a = y + SC_hp1, da = (y - a) + SC_hp1;
		return glibc_do_sin(a, da);
	}
	if (k < 0x419921fb) { double a, da; const int n = glibc_reduce_sincos(x, &a, &da); return glibc_do_sincos(a, da, n + 1); }
	return cos(x);
}

/* ------------------------------------------------------------------ glibc 2.35 double tan, restated
 * __tan of sysdeps/ieee754/dbl-64/s_tan.c (IBM Accurate Mathematical Library, without the slow paths) as the x86-64 FMA
 * ifunc variant computes it (read off __tan_fma), for |x| <= 25 -- the anisotropic fitter's arguments stay below pi/2;
 * larger arguments (a longer reduction, __branred) go to the host libm here and to the device libm in the kernels.
 * |x| <= 0.0608: odd polynomial d3 .. d11; <= 0.787: x = x_i + z with x_i from the 186 x 4 table xfg (tan and cot of
 * x_i): tan = fi + pz (fi + gi) / (gi - pz); <= 25: x = n pi/2 + a + da (mp1, mp2, mp3), then the same two forms for
 * a, or -cot through a double-double division (polynomial) / gi - pz (fi + gi) / (fi + pz) (table) when n is odd.
 * fn 5 = tan(x[k]). */
static double glibc_tan(double x)
{
	static const double g1 = 0x1.b096c00000000p-27, g2 = 0x1.f212d00000000p-5, g3 = 0x1.92f1a00000000p-1, g4 = 25.0,
	                    d3 = 0x1.5555555555555p-2, d5 = 0x1.11111111107c6p-3, d7 = 0x1.ba1ba1cdb8745p-5, d9 = 0x1.664ed49cfc666p-6,
	                    d11 = 0x1.2385a3cf2e4eap-7, e0 = 0x1.5555555554dbdp-2, e1 = 0x1.11112e0a6b45fp-3, mfftnhf = -15.5, TWO8 = 256.0,
	                    toint = 0x1.8p+52, hpinv = 0x1.45f306dc9c883p-1, mp1 = 0x1.921fb58000000p+0, mp2 = -0x1.dde973c000000p-27,
	                    mp3 = -0x1.cb3b399d747f2p-55;
	uint64_t bx; memcpy(&bx, &x, 8);
	if (((bx >> 32) & 0x7ff00000u) == 0x7ff00000u) return x - x;
	const double w = x < 0.0 ? -x : x;
	if (w <= g1) return x;
	if (w <= g2) {                                                              /* (II) */
		const double x2 = x * x;
		double t = fma(d11, x2, d9);
		t = fma(t, x2, d7); t = fma(t, x2, d5); t = fma(t, x2, d3);
		return fma(x * x2, t, x);
	}
	if (w <= g3) {                                                              /* (III) */
		const int i = (int)fma(TWO8, w, mfftnhf);
		const double *r = DJB_GLIBC_TAN_XFG + 4 * i;
		const double z = w - r[0], z2 = z * z;
		const double pz = fma(z * z2, fma(z2, e1, e0), z), fi = r[1], gi = r[2];
		return (((fi + gi) * pz) / (gi - pz) + fi) * (x < 0.0 ? -1.0 : 1.0);
	}
	if (!(w <= g4)) return tan(x);
	/* (IV) 0.787 < |x| <= 25: range reduction */
	const double t = fma(x, hpinv, toint), xn = t - toint;
	uint64_t tb; memcpy(&tb, &t, 8);
	const int n = (int)(tb & 1);
	const double t1 = fma(-xn, mp2, fma(-xn, mp1, x));
	const double a = fma(-xn, mp3, t1), da = fma(-xn, mp3, t1 - a);
	double ya, yya, sy;
	if (a < 0.0) { ya = -a; yya = -da; sy = -1.0; } else { ya = a; yya = da; sy = 1.0; }
	if (ya <= g2) {
		const double a2 = a * a;
		double p = fma(d11, a2, d9);
		p = fma(p, a2, d7); p = fma(p, a2, d5); p = fma(p, a2, d3);
		const double t2 = fma(a * a2, p, da), y = a + t2;
		if (n == 0) return y;
		/* -cot(a + da): b + db = a + t2 exactly, then 1 / (b + db) as a double-double */
		const double db = fabs(a) > fabs(t2) ? (a - y) + t2 : (t2 - y) + a;
		const double c = 1.0 / y, ch = c * y, cl = fma(c, y, -ch);
		const double cc = fma(-db, c, ((1.0 - ch) - cl) + 0.0) / y;
		const double z = c + cc, zz = (c - z) + cc;
		return -(zz + z);
	}
	const int i = (int)fma(TWO8, ya, mfftnhf);
	const double *r = DJB_GLIBC_TAN_XFG + 4 * i;
	const double z = (ya - r[0]) + yya, z2 = z * z;
	const double pz = fma(z * z2, fma(z2, e1, e0), z), fi = r[1], gi = r[2];
	const double num = (fi + gi) * pz;
	if (n) return (gi - num / (pz + fi)) * -sy;
	return (num / (gi - pz) + fi) * sy;
}

/* ------------------------------------------------------------------ glibc 2.35 double acos, restated
 * __ieee754_acos of sysdeps/ieee754/dbl-64/e_asin.c (IBM Accurate Mathematical Library, without the slow paths) as
 * __ieee754_acos_fma computes it: |x| < 1/8: pi/2 - x - x^3 p(x^2) with a two-term pi/2; seven intervals up to
 * 0.96875 with a Taylor expansion about the nearest point of asincos.tbl (rows of 11 .. 15 entries: x_i, the
 * coefficients, acos(x_i) as hi + the expansion's constant term); from 0.96875 to 1: 2 asin(sqrt((1 - |x|) / 2)) with
 * the square root built from root.tbl and refined as a double-double.  Complete.  fn 6 = acos(x[k]). */
static double glibc_acos(double x)
{
	static const double hp0 = 0x1.921fb54442d18p+0, hp1 = 0x1.1a62633145c07p-54, f1 = 0x1.55555555554f9p-3, f2 = 0x1.333333336127dp-4,
	                    f3 = 0x1.6db6dae42c0e4p-5, f4 = 0x1.f1c7e04f4ad99p-6, f5 = 0x1.6e442c822d419p-6, f6 = 0x1.292d80f453c72p-6,
	                    rt0 = 0x1.fffffffecc1ddp-1, rt1 = 0x1.fffffff757304p-2, rt2 = 0x1.800496769c91ap-2, rt3 = 0x1.4006318d1dab9p-2,
	                    t27 = 0x1p+27;
	uint64_t bx; memcpy(&bx, &x, 8);
	const int32_t m = (int32_t)(bx >> 32), k = m & 0x7fffffff;
	const uint32_t lo = (uint32_t)bx;
	if (k < 0x3c880000) return hp0;                                                /* |x| < 2^-55 */
	if (k < 0x3fc00000) {                                                          /* |x| < 1/8 */
		const double x2 = x * x;
		double p = fma(f6, x2, f5);
		p = fma(p, x2, f4); p = fma(p, x2, f3); p = fma(p, x2, f2); p = fma(p, x2, f1);
		const double r = hp0 - x;
		return r + fma(-p, x * x2, ((hp0 - r) - x) + hp1);
	}
	if (k < 0x3fef0000) {                                                          /* 1/8 <= |x| < 0.96875: seven table intervals */
		int S, n;
		if (k < 0x3fd00000) { S = 11; n = 11 * ((k >> 15) & 0x1f); }
		else if (k < 0x3fe00000) { S = 11; n = 352 + 11 * ((k >> 14) & 0x3f); }
		else if (k < 0x3fe80000) { S = 12; n = 1056 + 12 * ((k >> 13) & 0x7f); }
		else if (k < 0x3fed8000) { S = 13; n = 992 + 13 * ((k >> 13) & 0x7f); }
		else if (k < 0x3fee8000) { S = 14; n = 884 + 14 * ((k >> 13) & 0x7f); }
		else { S = 15; n = 768 + 15 * ((k >> 13) & 0x7f); }
		const double *T = DJB_GLIBC_ASNCS + n;
		const double xx = (m > 0 ? x : -x) - T[0];
		double p = T[S - 5];
		for (int j = S - 6; j >= 2; --j) p = fma(p, xx, T[j]);
		p = fma(p, xx * xx, T[S - 4]);
		const double t = fma(xx, T[1], p), y = T[S - 3];
		return m > 0 ? (hp1 - t) + (hp0 - y) : (t + hp1) + (y + hp0);
	}
	if (k < 0x3ff00000) {                                                          /* 0.96875 <= |x| < 1 */
		const double z = (m > 0 ? 1.0 - x : x + 1.0) * 0.5;
		uint64_t bz; memcpy(&bz, &z, 8);
		const int32_t hz = (int32_t)(bz >> 32);
		double t = DJB_GLIBC_INROOT[(hz >> 14) & 0x7f] * ldexp(1.0, 511 - (hz >> 21));
		const double r = fma(-(t * t), z, 1.0);
		double q = fma(rt3, r, rt2);
		q = fma(q, r, rt1); q = fma(q, r, rt0);
		t = q * t;
		const double c = z * t;
		const double h = fma(-c, t * 0.5, 1.5);
		const double y = fma(-t27, c, fma(c, t27, c));
		const double den = fma(h, c, y);
		const double cc = fma(-y, y, z) / den;
		double p = fma(f6, z, f5);
		p = fma(p, z, f4); p = fma(p, z, f3); p = fma(p, z, f2); p = fma(p, z, f1);
		p = (p * z) * (y + cc);
		if (m < 0) return 2.0 * (((hp1 - cc) - p) + (hp0 - y));
		return 2.0 * ((cc + p) + y);
	}
	if (k == 0x3ff00000 && lo == 0) return m > 0 ? 0.0 : 2.0 * hp0;                /* |x| = 1 */
	if (k > 0x7ff00000 || (k == 0x7ff00000 && lo != 0)) return x + x;              /* NaN */
	return (x - x) / (x - x);                                                      /* |x| > 1 */
}

void o_libm_f64(int fn, int64_t n, const double *x, const double *y, double *out)
{
	for (int64_t k = 0; k < n; ++k)
		out[k] = fn == 0 ? exp(x[k]) : fn == 1 ? pow(x[k], y[k]) : fn == 2 ? atan2(x[k], y[k]) : fn == 3 ? sin(x[k]) : fn == 4 ? cos(x[k]) : fn == 5 ? tan(x[k]) : acos(x[k]);
}
void o_glibc_f64(int fn, int64_t n, const double *x, const double *y, double *out)
{
	for (int64_t k = 0; k < n; ++k)
		out[k] = fn == 0 ? glibc_exp(x[k]) : fn == 1 ? glibc_pow(x[k], y[k]) : fn == 2 ? glibc_atan2(x[k], y[k])
		       : fn == 3 ? glibc_sin(x[k]) : fn == 4 ? glibc_cos(x[k]) : fn == 5 ? glibc_tan(x[k]) : glibc_acos(x[k]);
}
