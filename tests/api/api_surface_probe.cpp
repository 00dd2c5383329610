// tests/api/api_surface_probe.cpp -- every PUBLIC name of the reference's interface (jdupuy/dj_brdf, dj_brdf.h:41-537), used once
// with the reference's signatures.  Compiled with -fsyntax-only against the reference header (where it is mounted: proves the
// probe is right) and against include/dj_brdf.h (proves the facade has the name with a compatible signature).  Never run.
#include <cmath>
#include <vector>
#define DJ_BRDF_IMPLEMENTATION 1
#include "dj_brdf.h"

namespace {
struct my_brdf : public djb::brdf {                              // :73-109
	djb::vec3 eval(const djb::vec3 &, const djb::vec3 &, const void * = NULL) const { return djb::vec3(1); }
};
struct my_fresnel : public djb::fresnel::impl {                  // :157-162
	djb::vec3 eval(djb::float_t) const { return djb::vec3(1); }
	djb::fresnel::impl *copy() const { return new my_fresnel(*this); }
};
struct my_radial : public djb::radial {                          // :301-324
	my_radial() : djb::radial() {}
	my_radial(const djb::fresnel::impl &f, bool s) : djb::radial(f, s) {}
	bool supports_smith_vndf_sampling() const { return true; }
	djb::float_t p22_radial(djb::float_t) const { return 1; }
	djb::float_t sigma_std_radial(djb::float_t) const { return 1; }
	djb::float_t cdf_radial(djb::float_t) const { return 1; }
	djb::float_t qf_radial(djb::float_t) const { return 1; }
	djb::float_t qf2_radial(djb::float_t, djb::float_t, djb::float_t) const { return 1; }
	djb::float_t qf3_radial(djb::float_t, djb::float_t) const { return 1; }
};
struct my_microfacet : public djb::microfacet {                  // :210-298
	my_microfacet() : djb::microfacet() {}
	my_microfacet(const djb::fresnel::impl &f, bool s) : djb::microfacet(f, s) {}
	bool supports_smith_vndf_sampling() const { return false; }
	djb::float_t qf2(djb::float_t, const djb::vec3 &) const { return 0; }
	djb::float_t qf3(djb::float_t, const djb::vec3 &, djb::float_t) const { return 0; }
protected:
	djb::float_t sigma_std(const djb::vec3 &) const { return 1; }
	djb::float_t p22_std(djb::float_t, djb::float_t) const { return 1; }
	void sample_vp22_std_smith(djb::float_t u1, djb::float_t u2, const djb::vec3 &k, djb::float_t *x, djb::float_t *y) const
	{ djb::microfacet::sample_vp22_std_smith(u1, u2, k, x, y); }
	void sample_vp22_std_nmap(djb::float_t, djb::float_t, const djb::vec3 &, djb::float_t *x, djb::float_t *y) const { *x = *y = 0; }
};

template <class T> void use(const T &) {}

void operators(const djb::brdf &b)                               // :77-100
{
	const djb::vec3 i(0.1f, 0.2f), o(0.3f, 0.4f);
	djb::vec3 w, h, d; djb::float_t pdf;
	use(b.eval(i, o)); use(b.eval(i, o, NULL)); use(b.eval_hd(i, o)); use(b.evalp(i, o)); use(b.evalp_hd(i, o));
	use(b.evalp_is(0.1f, 0.2f, o, &w, &pdf)); use(b.evalp_is(0.1f, 0.2f, o, NULL, NULL, NULL));
	use(b.sample(0.1f, 0.2f, o)); use(b.pdf(i, o));
	djb::brdf::io_to_hd(i, o, &h, &d); djb::brdf::hd_to_io(h, d, &w, &w);
}
void microfacets(djb::microfacet &m)                             // :246-282
{
	const djb::vec3 k(0.1f, 0.2f);
	const djb::microfacet::params p = djb::microfacet::params::standard();
	use(m.fresnel(0.5f)); use(m.ndf(k)); use(m.ndf(k, p)); use(m.gaf(k, k, k)); use(m.gaf(k, k, k, p)); use(m.g1(k, k)); use(m.g1(k, k, p));
	use(m.sigma(k)); use(m.sigma(k, p)); use(m.p22(0.1f, 0.2f)); use(m.p22(0.1f, 0.2f, p)); use(m.vp22(0.1f, 0.2f, k)); use(m.vp22(0.1f, 0.2f, k, p));
	use(m.vndf(k, k)); use(m.vndf(k, k, p)); use(m.supports_smith_vndf_sampling()); use(m.qf2(0.5f, k)); use(m.qf3(0.5f, k, 0.1f));
	m.set_shadow(true); m.set_fresnel(djb::fresnel::ideal()); use(m.get_shadow()); use(m.get_fresnel().eval(0.5f));
}
void radials(const djb::radial &r)                               // :307-314
{ use(r.p22_radial(1)); use(r.sigma_std_radial(1)); use(r.cdf_radial(1)); use(r.qf_radial(0.5f)); use(r.qf2_radial(0.5f, 0.5f, 0.5f)); use(r.qf3_radial(0.5f, 0.1f)); }
} // namespace

int main()
{
	try { throw djb::exc("x"); } catch (const std::exception &e) { use(e.what()); }                       // :54-59
	// vec3, :62-71, 589-637
	const double rd[3] = { 1, 2, 3 }; const float rf[3] = { 1, 2, 3 };
	djb::vec3 a = djb::vec3::from_raw(rd), b = djb::vec3::from_raw(rf), c(1.0f), d(1.0f, 2.0f, 3.0f), e(0.3f, 0.7f);
	use(djb::vec3::to_raw(a)); use(a.intensity()); use(a.x + a.y + a.z);
	a = 2.0f * b; a = b * 2.0f; a = b / 2.0f; a = b * c; a = b / c; a = b + c; a = b - c; a += b; a *= b; a *= 2.0f;
	use(djb::dot(a, b)); use(djb::cross(a, b)); use(djb::normalize(d)); use(e);
	// lambert, :112-123
	djb::lambert lam; djb::lambert::params lp(djb::vec3(0.5f)); use(lp.m_reflectance); use(lam.eval(a, b, &lp)); operators(lam);
	// fresnel, :149-207
	djb::float_t f0, ior; djb::vec3 v0, v1;
	djb::fresnel::ior_to_f0(1.5f, &f0); djb::fresnel::f0_to_ior(0.04f, &ior); djb::fresnel::ior_to_f0(a, &v0); djb::fresnel::f0_to_ior(a, &v1);
	djb::fresnel::ideal fi; djb::fresnel::unpolarized fu(a); djb::fresnel::schlick fs(a); djb::fresnel::sgd fg(a, b);
	std::vector<djb::vec3> pts(4, djb::vec3(1)); djb::fresnel::spline fp(pts); use(fp.get_points());
	const djb::fresnel::impl *impls[6] = { &fi, &fu, &fs, &fg, &fp, NULL }; my_fresnel mf; impls[5] = &mf;
	for (int k = 0; k < 6; ++k) { use(impls[k]->eval(0.5f)); delete impls[k]->copy(); }
	// microfacet::params, :213-243
	djb::microfacet::params p0, p1(0.1f, 0.2f, 0.3f), p2(0.1f, 0.2f, 0.3f, 0.4f, 0.5f);
	p0 = djb::microfacet::params::standard(); p0 = djb::microfacet::params::isotropic(0.3f); p0 = djb::microfacet::params::elliptic(0.1f, 0.2f);
	p0 = djb::microfacet::params::elliptic(0.1f, 0.2f, 0.3f); p0 = djb::microfacet::params::pdfparams(0.1f, 0.2f); p0 = djb::microfacet::params::pdfparams(0.1f, 0.2f, 0.3f, 0.4f, 0.5f);
	p0.set_ellipse(0.1f, 0.2f); p0.set_ellipse(0.1f, 0.2f, 0.3f); p0.set_pdfparams(0.1f, 0.2f); p0.set_pdfparams(0.1f, 0.2f, 0.3f, 0.4f, 0.5f);
	p0.set_location(0.1f, 0.2f); p0.set_location(d);
	djb::float_t g[5]; p0.get_ellipse(&g[0], &g[1]); p0.get_ellipse(&g[0], &g[1], &g[2]); p0.get_pdfparams(&g[0], &g[1]); p0.get_pdfparams(&g[0], &g[1], &g[2], &g[3], &g[4]);
	p0.get_location(&g[0], &g[1]); p0.get_location(&a); use(p1); use(p2);
	// beckmann / ggx, :327-391
	djb::beckmann bk, bk2(fs), bk3(fs, false); djb::ggx gx, gx2(fs), gx3(fs, false);
	operators(bk); microfacets(bk2); radials(bk3); use(bk.qf1(0.5f)); operators(gx); microfacets(gx2); radials(gx3); use(gx.qf1(0.5f));
	use(bk.eval(a, b, &p0)); use(bk.sample(0.1f, 0.2f, a, &p0));
	djb::beckmann::lrep l0, l1(0.1f, 0.2f, 0.3f, 0.4f, 0.5f);
	l0 = l0 + l1; l0 = l0 * 2.0f; l0 += l1; l0 *= 2.0f; l0.scale(1.0f, 2.0f); l0.shear(0.1f, 0.2f);
	djb::beckmann::params_to_lrep(p0, &l0); djb::beckmann::lrep_to_params(l0, &p0);
	// merl / utia, :126-146 (constructors throw on a missing file: never reached at run time)
	if (rd[0] < 0) {
		djb::merl m("x.binary"); use(m.get_samples().size()); operators(m);
		djb::utia u("x.bin"); use(u.get_samples().size()); operators(u);
		// tabular / tabular_anisotropic, :394-478
		djb::tabular t(m, 90), t2(m, 90, false);
		use(djb::tabular::fit_beckmann_parameters(t)); use(djb::tabular::fit_ggx_parameters(t2));
		use(t.get_p22v().size()); use(t.get_sigmav().size()); use(t.get_cdfv().size()); use(t.get_qfv().size());
		operators(t); microfacets(t); use(t.p22_radial(1)); use(t.sigma_std_radial(1)); use(t.cdf_radial(1)); use(t.qf_radial(0.5f)); use(t.supports_smith_vndf_sampling());
		djb::tabular_anisotropic ta(u, 16, 32), ta2(u, 16, 32, false);
		use(djb::tabular_anisotropic::fit_beckmann_parameters(ta)); use(djb::tabular_anisotropic::fit_ggx_parameters(ta2));
		int ec, ac; use(ta.get_p22v(&ec, &ac).size()); use(ta.get_sigmav(&ec, &ac).size());
		use(ta.pdf1(0.1f)); use(ta.pdf2(0.1f, 0.2f)); use(ta.cdf1(0.1f)); use(ta.cdf2(0.1f, 0.2f)); use(ta.qf1(0.1f)); use(ta.qf2(0.1f, 0.2f));
		operators(ta); microfacets(ta);
		// sgd / abc, :481-535
		djb::sgd s("gold-metallic-paint"); use(s.ndf(a)); use(s.gaf(a, a, b)); use(s.g1(a)); use(s.fresnel(0.5f)); use(s.get_fresnel().eval(0.5f)); operators(s);
		djb::abc ab("gold-metallic-paint"); use(ab.ndf(a)); djb::float_t gg = ab.gaf(a, a, b); use(gg); use(ab.fresnel(0.5f)); use(ab.get_fresnel().eval(0.5f)); operators(ab);
		// the user's own classes wherever the reference takes a brdf / a fresnel::impl
		my_brdf mb; operators(mb); djb::tabular t3(mb, 32); djb::tabular_anisotropic t4(mb, 8, 8);
		my_radial mr, mr2(mf, false); operators(mr); microfacets(mr2); radials(mr); djb::tabular t5(mr, 32);
		my_microfacet mm, mm2(mf, true); operators(mm); microfacets(mm2); djb::ggx gu(mf); djb::beckmann bu(mf, false); bu.set_fresnel(mf);
	}
	return 0;
}
