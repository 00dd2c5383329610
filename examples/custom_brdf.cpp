// examples/custom_brdf.cpp -- the library's extension points, used the way the reference's README advertises
// ("Extracting Microfacet-based BRDF Parameters from Arbitrary Materials"): BRDFs and a Fresnel term DEFINED BY THE USER,
// fitted and evaluated through the djb:: classes.
//
// The program is written against the REFERENCE's interface only -- `class brdf` with its public constructor and `eval` as
// the one pure virtual (dj_brdf.h:74-109), `fresnel::impl` with `eval` and `copy` (dj_brdf.h:157-162) -- and compiles
// unchanged against either header:
//   g++ -I/root/reference/..   -> the reference, everything on the CPU      (tests/golden/make_reftests.sh: expected output)
//   g++ -I include -ldjb_hip   -> this repository: the user's eval() runs on the host at the fit's query directions, the
//                                 power iteration / quadratures / moment fits in the gfx950 kernels
// and must print the same bytes (tests/test_gpu_golden.py::test_reference_programs_unchanged, tests/test_cpu_path.py).
// Floats are printed with %a: every bit counts.
#include <cmath>
#include <cstdio>
#include <vector>

#define DJ_BRDF_IMPLEMENTATION 1
#include "dj_brdf.h"

namespace {

// a normalised Phong lobe around the mirror direction plus a diffuse floor
class phong : public djb::brdf {
public:
	phong(const djb::vec3 &kd, const djb::vec3 &ks, float exponent) : m_kd(kd), m_ks(ks), m_n(exponent) {}
	djb::vec3 eval(const djb::vec3 &i, const djb::vec3 &o, const void *user_param = NULL) const
	{
		(void)user_param;
		const djb::vec3 r(-o.x, -o.y, o.z);
		float c = djb::dot(r, i);
		if (!(c > 0.0f)) c = 0.0f;
		const float s = (float)(((double)m_n + 2.0) / (2.0 * M_PI) * std::pow((double)c, (double)m_n));
		return m_kd / (float)M_PI + m_ks * s;
	}
private:
	djb::vec3 m_kd, m_ks;
	float m_n;
};

// an anisotropic Gaussian lobe of the half vector (Ward's form)
class ward : public djb::brdf {
public:
	ward(const djb::vec3 &kd, const djb::vec3 &ks, float ax, float ay) : m_kd(kd), m_ks(ks), m_ax(ax), m_ay(ay) {}
	djb::vec3 eval(const djb::vec3 &i, const djb::vec3 &o, const void *user_param = NULL) const
	{
		(void)user_param;
		if (!(i.z > 0.0f && o.z > 0.0f)) return djb::vec3(0);
		const djb::vec3 h = djb::normalize(i + o);
		const float tx = h.x / m_ax, ty = h.y / m_ay;
		const float q = (tx * tx + ty * ty) / (h.z * h.z);
		const float e = (float)std::exp(-(double)q);
		const float den = (float)(4.0 * M_PI * (double)(m_ax * m_ay) * std::sqrt((double)(i.z * o.z)));
		return m_kd / (float)M_PI + m_ks * (e / den);
	}
private:
	djb::vec3 m_kd, m_ks;
	float m_ax, m_ay;
};

// Schlick's term with Lazanyi's grazing-angle correction: not one of the five the library ships
class lazanyi : public djb::fresnel::impl {
public:
	lazanyi(const djb::vec3 &f0, float a) : m_f0(f0), m_a(a) {}
	djb::vec3 eval(float cos_theta_d) const
	{
		const double m = 1.0 - (double)cos_theta_d;
		const float p5 = (float)(m * m * m * m * m);
		const float p7 = (float)((double)p5 * m * m);
		const float t = m_a * cos_theta_d * p7;
		return m_f0 + (djb::vec3(1) - m_f0) * p5 - djb::vec3(t);
	}
	djb::fresnel::impl *copy() const { return new lazanyi(*this); }
private:
	djb::vec3 m_f0;
	float m_a;
};

// a radial NDF defined by the user (dj_brdf.h:301-324: the public virtuals of class radial): a Student-t-like slope distribution
// p22(r^2) = (g - 1) / (pi (1 + r^2)^g) with g = 3, sampled with the "nmap" scheme through its quantile function
class student : public djb::radial {
public:
	student() : djb::radial() {}
	explicit student(const djb::fresnel::impl &f, bool shadow = true) : djb::radial(f, shadow) {}
	bool supports_smith_vndf_sampling() const { return false; }
	float p22_radial(float r_sqr) const
	{ const double t = 1.0 + (double)r_sqr; return (float)(2.0 / (M_PI * t * t * t)); }
	float sigma_std_radial(float cos_theta_k) const                   // a smooth stand-in for the projected area: (1 + c (1 + c) / 2) / 2
	{ const float c = cos_theta_k; return (float)((1.0 + (double)(c * (1.0f + c)) * 0.5) * 0.5); }
	float cdf_radial(float r) const
	{ const double t = 1.0 + (double)(r * r); return (float)(1.0 - 1.0 / (t * t)); }
	float qf_radial(float u) const
	{ return (float)std::sqrt(1.0 / std::sqrt(1.0 - (double)u) - 1.0); }
};

// an NDF defined at the microfacet level (dj_brdf.h:283-295: the protected virtuals of class microfacet): not radial --
// a slope distribution that is a product of two different 1D laws
class separable : public djb::microfacet {
public:
	separable() : djb::microfacet() {}
	bool supports_smith_vndf_sampling() const { return false; }
protected:
	float sigma_std(const djb::vec3 &k) const
	{ return (float)(0.5 * ((double)k.z + std::sqrt((double)(k.z * k.z) + 0.5 * (double)(k.x * k.x) + 0.25 * (double)(k.y * k.y)))); }
	float p22_std(float x, float y) const
	{
		const float gx = (float)(std::exp(-(double)(x * x)) / std::sqrt(M_PI));         // Gaussian in x
		const double ty = 1.0 + (double)(y * y);
		const float cy = (float)(1.0 / (2.0 * ty * std::sqrt(ty)));                     // 1 / (2 (1 + y^2)^(3/2)) in y
		return gx * cy;
	}
	void sample_vp22_std_nmap(float u1, float u2, const djb::vec3 &k, float *xslope, float *yslope) const
	{
		(void)k;
		const double a = 2.0 * (double)u1 - 1.0;                       // a crude but deterministic sampler: what matters here is that it is called
		*xslope = (float)(a * std::sqrt(-std::log(1.0 - std::fabs(a) * 0.999)));
		const double b = 2.0 * (double)u2 - 1.0;
		*yslope = (float)(b / std::sqrt(1.0 - b * b * 0.999));
	}
};

void show(const char *tag, const djb::vec3 &v) { printf("%s %a %a %a\n", tag, v.x, v.y, v.z); }
void show_table(const char *tag, const std::vector<djb::float_t> &v)
{
	printf("%s n=%d", tag, (int)v.size());
	for (size_t k = 0; k < v.size(); k += v.size() / 6 + 1) printf(" [%d]=%a", (int)k, v[k]);
	printf(" [last]=%a\n", v.back());
}

} // namespace

int main()
{
	// ---- 1. isotropic fits of user-defined lobes
	const float exponents[3] = { 10.0f, 50.0f, 400.0f };
	for (int k = 0; k < 3; ++k) {
		phong p(djb::vec3(0.05f, 0.04f, 0.03f), djb::vec3(0.9f, 0.8f, 0.7f), exponents[k]);
		djb::tabular tab(p, 90);
		float ab, ag;
		djb::tabular::fit_beckmann_parameters(tab).get_ellipse(&ab, NULL);
		djb::tabular::fit_ggx_parameters(tab).get_ellipse(&ag, NULL);
		printf("phong n=%g: beckmann %.3f ggx %.3f  (%a %a)\n", exponents[k], ab, ag, ab, ag);
		if (k == 1) {
			show_table("  p22", tab.get_p22v()); show_table("  sigma", tab.get_sigmav());
			show_table("  cdf", tab.get_cdfv()); show_table("  qf", tab.get_qfv());
			show("  fresnel(0.3)", tab.fresnel(0.3f));
			show("  tab.eval", tab.eval(djb::vec3(0.3f, 0.1f), djb::vec3(0.5f, 2.0f)));
		}
	}
	{
		phong p(djb::vec3(0.1f), djb::vec3(0.5f), 30.0f);
		djb::tabular tab(p, 32, false);                       // another resolution, no shadowing term
		float ab, ag;
		djb::tabular::fit_beckmann_parameters(tab).get_ellipse(&ab, NULL);
		djb::tabular::fit_ggx_parameters(tab).get_ellipse(&ag, NULL);
		printf("phong n=30 res 32 noshadow: %a %a\n", ab, ag);
	}

	// ---- 2. the base-class operators of a user-derived object (dj_brdf.h:795-845)
	{
		phong p(djb::vec3(0.2f, 0.3f, 0.4f), djb::vec3(0.6f, 0.5f, 0.4f), 20.0f);
		const djb::brdf &b = p;
		const djb::vec3 i(0.4f, 0.7f), o(0.6f, 3.5f);
		show("eval", b.eval(i, o)); show("evalp", b.evalp(i, o));
		printf("pdf %a\n", b.pdf(i, o));
		show("sample", b.sample(0.31f, 0.77f, o));
		djb::vec3 wi; float pdf = 0;
		show("evalp_is", b.evalp_is(0.31f, 0.77f, o, &wi, &pdf)); show("  i", wi); printf("  pdf %a\n", pdf);
		djb::vec3 h, d;
		djb::brdf::io_to_hd(i, o, &h, &d);
		show("eval_hd", b.eval_hd(h, d)); show("evalp_hd", b.evalp_hd(h, d));
	}

	// ---- 3. anisotropic fit of a user-defined lobe
	{
		ward w(djb::vec3(0.02f), djb::vec3(0.8f, 0.7f, 0.6f), 0.15f, 0.4f);
		djb::tabular_anisotropic tab(w, 12, 24);
		float v[5];
		djb::tabular_anisotropic::fit_beckmann_parameters(tab).get_pdfparams(&v[0], &v[1], &v[2], &v[3], &v[4]);
		printf("ward beckmann %a %a %a %a %a\n", v[0], v[1], v[2], v[3], v[4]);
		djb::tabular_anisotropic::fit_ggx_parameters(tab).get_pdfparams(&v[0], &v[1], &v[2], &v[3], &v[4]);
		printf("ward ggx      %a %a %a %a %a\n", v[0], v[1], v[2], v[3], v[4]);
		int e = 0, a = 0;
		show_table("  p22", tab.get_p22v(&e, &a)); show_table("  sigma", tab.get_sigmav(&e, &a));
		show("  fresnel(0.5)", tab.fresnel(0.5f));
		show("  tab.evalp", tab.evalp(djb::vec3(0.3f, 0.1f), djb::vec3(0.5f, 2.0f)));
	}

	// ---- 4. a user-defined Fresnel term inside the library's microfacet BRDFs
	{
		lazanyi f(djb::vec3(0.95f, 0.64f, 0.54f), 1.5f);
		djb::ggx g(f);
		djb::beckmann bk(f, false);
		const djb::microfacet::params pr = djb::microfacet::params::elliptic(0.3f, 0.1f, 0.4f);
		const djb::vec3 i(0.5f, 0.2f), o(0.7f, 2.9f), below(2.0f, 1.0f);
		show("fresnel(0.2)", g.fresnel(0.2f)); show("get_fresnel", g.get_fresnel().eval(0.2f));
		show("ggx eval", g.eval(i, o)); show("ggx evalp", g.evalp(i, o, &pr)); printf("ggx pdf %a\n", g.pdf(i, o, &pr));
		show("ggx eval below", g.eval(below, o));
		show("bk eval", bk.eval(i, o, &pr)); show("bk evalp", bk.evalp(i, o));
		djb::vec3 wi; float pdf = 0;
		show("ggx evalp_is", g.evalp_is(0.42f, 0.13f, o, &wi, &pdf, &pr)); show("  i", wi); printf("  pdf %a\n", pdf);
		show("bk evalp_is", bk.evalp_is(0.9f, 0.6f, o, &wi, &pdf)); show("  i", wi); printf("  pdf %a\n", pdf);
		djb::vec3 h, d;
		djb::brdf::io_to_hd(i, o, &h, &d);
		show("ggx eval_hd", g.eval_hd(h, d)); show("ggx evalp_hd", g.evalp_hd(h, d, &pr));
		// swap the term on a live object, both ways
		g.set_fresnel(djb::fresnel::schlick(djb::vec3(0.9f, 0.6f, 0.5f)));
		show("ggx eval (schlick)", g.eval(i, o));
		g.set_fresnel(f);
		show("ggx eval (lazanyi again)", g.eval(i, o));
		// and fit it: the source's eval involves the user's Fresnel, the fitted object gets a spline
		djb::tabular tab(g, 64);
		float ab, ag;
		djb::tabular::fit_beckmann_parameters(tab).get_ellipse(&ab, NULL);
		djb::tabular::fit_ggx_parameters(tab).get_ellipse(&ag, NULL);
		printf("ggx(lazanyi) res 64: beckmann %.3f ggx %.3f  (%a %a)\n", ab, ag, ab, ag);
		show("  fresnel(0.1)", tab.fresnel(0.1f)); show("  fresnel(0.9)", tab.fresnel(0.9f));
	}

	// ---- 5. user-defined NDFs: a class derived from djb::radial, one derived from djb::microfacet
	{
		student st;
		lazanyi f(djb::vec3(0.9f, 0.7f, 0.5f), 0.5f);
		student st_f(f, false);
		separable sp;
		const djb::microfacet::params pr = djb::microfacet::params::elliptic(0.35f, 0.2f, 1.1f);
		const djb::microfacet::params pp = djb::microfacet::params::pdfparams(0.4f, 0.25f, 0.3f, 0.1f, -0.05f);
		const djb::vec3 i(0.5f, 0.2f), o(0.7f, 2.9f), h = djb::normalize(i + o);
		const djb::microfacet *objs[3] = { &st, &st_f, &sp };
		const char *names[3] = { "student", "student+lazanyi", "separable" };
		for (int w = 0; w < 3; ++w) {
			const djb::microfacet &m = *objs[w];
			printf("%s\n", names[w]);
			show("  eval", m.eval(i, o)); show("  eval(pr)", m.eval(i, o, &pr)); show("  evalp(pp)", m.evalp(i, o, &pp));
			printf("  pdf %a %a\n", m.pdf(i, o), m.pdf(i, o, &pr));
			show("  sample", m.sample(0.31f, 0.77f, o, &pr));
			djb::vec3 wi; float pdf = 0;
			show("  evalp_is", m.evalp_is(0.62f, 0.18f, o, &wi, &pdf, &pp)); show("    i", wi); printf("    pdf %a\n", pdf);
			printf("  ndf %a gaf %a g1 %a sigma %a p22 %a vp22 %a vndf %a\n", m.ndf(h, pr), m.gaf(h, i, o, pr), m.g1(h, o, pp), m.sigma(o, pr),
			       m.p22(0.3f, -0.2f, pp), m.vp22(0.3f, -0.2f, o, pr), m.vndf(h, o));
			show("  fresnel(0.4)", m.fresnel(0.4f));
			djb::vec3 hh, dd;
			djb::brdf::io_to_hd(i, o, &hh, &dd);
			show("  evalp_hd", m.evalp_hd(hh, dd, &pr));
		}
		printf("student radial queries %a %a %a %a\n", st.p22_radial(0.7f), st.sigma_std_radial(0.6f), st.cdf_radial(0.9f), st.qf_radial(0.35f));
		// fit them: the sources' eval runs the user's NDF on the host, the fit itself where the library likes
		djb::tabular t1(st, 48);
		float ab, ag;
		djb::tabular::fit_beckmann_parameters(t1).get_ellipse(&ab, NULL);
		djb::tabular::fit_ggx_parameters(t1).get_ellipse(&ag, NULL);
		printf("tabular(student, 48): %a %a\n", ab, ag);
		show_table("  p22", t1.get_p22v());
		djb::tabular_anisotropic t2(sp, 10, 12);
		float v[5];
		djb::tabular_anisotropic::fit_ggx_parameters(t2).get_pdfparams(&v[0], &v[1], &v[2], &v[3], &v[4]);
		printf("tabular_anisotropic(separable, 10, 12) ggx %a %a %a %a %a\n", v[0], v[1], v[2], v[3], v[4]);
	}
	return 0;
}
