// examples/facade_check.cpp -- exercises the scalar djb:: surface of include/djb_hip.hpp on the
// GPU and prints the known answers of SURVEY.md 8-N (measured on the reference at survey time).
// Exit code 0 iff every value matches to 1e-5 relative.
#include <cmath>
#include <cstdio>

#include <cstring>
#include <vector>

#include "djb_hip.hpp"

static int g_fail = 0;
static void expect(const char *what, double got, double want)
{
	double rel = want == 0.0 ? std::fabs(got) : std::fabs(got - want) / std::fmax(std::fabs(want), 1e-30);
	printf("%-34s %.9g (want %.9g) %s\n", what, got, want, rel <= 1e-5 ? "ok" : "MISMATCH");
	if (!(rel <= 1e-5)) g_fail = 1;
}

int main()
{
	try {
		float iz = sqrtf(1.f - 0.3f * 0.3f - 0.2f * 0.2f), oz = sqrtf(1.f - 0.4f * 0.4f - 0.1f * 0.1f);
		djb::vec3 i(0.3f, 0.2f, iz), o(-0.4f, 0.1f, oz);
		djb::ggx ggx;
		djb::microfacet::params iso = djb::microfacet::params::isotropic(0.3f);
		expect("ggx iso 0.3 eval", ggx.eval(i, o, &iso).x, 0.621380985);
		expect("ggx iso 0.3 pdf", ggx.pdf(i, o, &iso), 0.581518769);
		djb::vec3 s = ggx.sample(0.25f, 0.75f, o, &iso);
		expect("ggx sample.x", s.x, 0.657071352); expect("ggx sample.z", s.z, 0.749468625);
		djb::beckmann beck;
		djb::microfacet::params ell = djb::microfacet::params::elliptic(0.2f, 0.5f, 0.7f);
		float ax, ay, rho; ell.get_pdfparams(&ax, &ay, &rho);
		expect("elliptic ax", ax, 0.356585801); expect("elliptic ay", ay, 0.403542489); expect("elliptic rho", rho, 0.719068825);
		expect("beckmann elliptic eval", beck.eval(i, o, &ell).x, 0.562808752);
		expect("beckmann elliptic pdf", beck.pdf(i, o, &ell), 0.524953067);
		djb::vec3 wi; float pdf;
		djb::vec3 w = beck.evalp_is(0.25f, 0.75f, o, &wi, &pdf, &ell);
		expect("beckmann evalp_is weight", w.x, 0.99999994); expect("beckmann evalp_is pdf", pdf, 0.545059502);
		djb::ggx ggs(djb::fresnel::schlick(djb::vec3(1.0f, 0.71f, 0.29f)));
		djb::vec3 c = ggs.eval(i, o, &iso);
		expect("ggx+schlick eval.g", c.y, 0.441180676); expect("ggx+schlick eval.b", c.z, 0.180200979);
		// microfacet::set_fresnel / set_shadow: a mutated plain ggx must equal the schlick-constructed one
		djb::ggx ggm;
		ggm.set_fresnel(djb::fresnel::schlick(djb::vec3(1.0f, 0.71f, 0.29f)));
		c = ggm.eval(i, o, &iso);
		expect("set_fresnel(schlick) eval.g", c.y, 0.441180676); expect("set_fresnel(schlick) eval.b", c.z, 0.180200979);
		ggm.set_shadow(false);
		expect("set_shadow(false) get_shadow", (double)ggm.get_shadow(), 0);
		djb::vec3 ang(0.7f, 1.9f);                     // vec3(theta, phi), dj_brdf.h:589-595
		expect("vec3(theta,phi).x", ang.x, -0.208268836); expect("vec3(theta,phi).z", ang.z, 0.764842212);
		djb::lambert lam; djb::lambert::params lamp(djb::vec3(0.5f, 0.25f, 0.9f));
		expect("lambert eval", lam.eval(i, o).x, 0.318309873); expect("lambert(reflectance) eval.g", lam.eval(i, o, &lamp).y, 0.0795774683);
		float f0, ior; djb::fresnel::ior_to_f0(1.5f, &f0); djb::fresnel::f0_to_ior(0.04f, &ior);
		expect("fresnel::ior_to_f0(1.5)", f0, 0.04); expect("fresnel::f0_to_ior(0.04)", ior, 1.5);
		djb::microfacet::params p5(0.4f, 0.25f, 0.3f, 0.1f, -0.2f); float ptx, pty; p5.get_location(&ptx, &pty);
		expect("params(ax,ay,rho,tx,ty).tx", ptx, 0.1); expect("params(ax,ay,rho,tx,ty).ty", pty, -0.2);
		p5.set_location(djb::vec3(-0.1f, 0.2f, 1.0f)); p5.get_location(&ptx, &pty);
		expect("set_location(vec3).tx", ptx, 0.1); expect("set_location(vec3).ty", pty, -0.2);
		djb::vec3 h, d; djb::brdf::io_to_hd(i, o, &h, &d);
		expect("io_to_hd h.y", h.y, 0.160367534); expect("io_to_hd d.y", d.y, -0.347850591);
		djb::tabular tab(djb::ggx(), 90);
		float a, dummy;
		djb::tabular::fit_beckmann_parameters(tab).get_ellipse(&a, &dummy); expect("tabular(ggx,90) alpha_beckmann", a, 2.75112224);
		djb::tabular::fit_ggx_parameters(tab).get_ellipse(&a, &dummy); expect("tabular(ggx,90) alpha_ggx", a, 0.866066337);
		expect("tabular p22v.size", (double)tab.get_p22v().size(), 90);
		// supports_smith_vndf_sampling(): true for the analytic lobes, false for BOTH tabulated classes (dj_brdf.h:412, 439)
		djb::tabular_anisotropic tan_(djb::ggx(), 8, 8);
		expect("ggx supports_smith_vndf_sampling", ggx.supports_smith_vndf_sampling() ? 1.0 : 0.0, 1.0);
		expect("tabular supports_smith_vndf_sampling", tab.supports_smith_vndf_sampling() ? 1.0 : 0.0, 0.0);
		expect("tabular_anisotropic supports_smith_vndf_sampling", tan_.supports_smith_vndf_sampling() ? 1.0 : 0.0, 0.0);
		// what dj_beckmannconductor / dj_brdf do at load time: new djb::beckmann(tab->get_fresnel()) (mitsuba/dj_beckmannconductor.cpp:189)
		djb::beckmann from_tab(tab.get_fresnel());
		expect("beckmann(tab.get_fresnel()).fresnel(0.5).g", from_tab.fresnel(0.5f).y, tab.fresnel(0.5f).y);
		djb::beckmann::lrep l1, l2(0.1f, -0.05f, 0.02f, 0.03f, 0.001f);          // mitsuba/dj_beckmannconductor.cpp:304-314
		djb::beckmann::params_to_lrep(ell, &l1); l1 *= 0.7f; djb::microfacet::params lp; djb::beckmann::lrep_to_params(l1 + l2, &lp);
		float lax, lay; lp.get_pdfparams(&lax, &lay); expect("lrep path ax > 0", lax > 0 ? 1.0 : 0.0, 1.0);
		// the numerical contract is a property of the context (hip::context::set_contract_1e5): a batch of a context in 1e-5 mode stays
		// within 1e-5 relative of the bit-exact one; one-pair calls never change
		{
			djb::hip::context &c0 = djb::hip::context::standard();
			const size_t nb = 1u << 16;                       // a batch well above the scalar threshold
			std::vector<djb::vec3> bi(nb), bo(nb), exact(nb), fast(nb);
			for (size_t k = 0; k < nb; ++k) {
				const float t = 0.05f + 1.4f * (float)k / (float)nb, ph = 0.37f * (float)k;
				bi[k] = djb::vec3(t, ph); bo[k] = djb::vec3(1.45f - t, 1.3f * ph + 0.5f);
			}
			ggs.eval(nb, &bi[0], &bo[0], &exact[0], &iso);
			c0.set_contract_1e5(true);
			ggs.eval(nb, &bi[0], &bo[0], &fast[0], &iso);
			const float one_pair = ggs.eval(i, o, &iso).y;
			c0.set_contract_1e5(false);
			double worst = 0;
			for (size_t k = 0; k < nb; ++k) {
				const double e = exact[k].y, f = fast[k].y;
				if (e != f) worst = std::fmax(worst, std::fabs(f - e) / std::fmax(std::fabs(e), 1e-30));
			}
			expect("contract mode: worst rel. diff < 1e-5", worst < 1e-5 ? 1.0 : 0.0, 1.0);
			expect("contract mode: one-pair call unchanged", one_pair, 0.441180676);
		}
		// a user-defined Fresnel term (dj_brdf.h:157-162) in batches: D G of all pairs in one library call + the user's F per pair
		// on the host must give the bits of n one-pair calls (which ask for G and compose the reference's expressions one by one)
		{
			struct tint : public djb::fresnel::impl {
				djb::vec3 eval(float c) const { const float m = 1.0f - c; return djb::vec3(0.9f, 0.6f, 0.3f) + djb::vec3(0.1f, 0.4f, 0.7f) * (m * m * m); }
				djb::fresnel::impl *copy() const { return new tint(*this); }
			};
			tint user_f; djb::ggx gu(user_f); djb::beckmann bu(user_f, false);
			const size_t nb = 4096;
			std::vector<djb::vec3> bi(nb), bo(nb), a(nb), b1(nb), wi(nb), wi1(nb);
			std::vector<float> u1(nb), u2(nb), pd(nb), pd1(nb);
			for (size_t k = 0; k < nb; ++k) {
				const float t = 0.02f + 1.7f * (float)k / (float)nb, ph = 0.37f * (float)k;      // some pairs below the horizon
				bi[k] = djb::vec3(t, ph); bo[k] = djb::vec3(1.75f - t, 1.3f * ph + 0.5f);
				u1[k] = (float)((k * 2654435761u) >> 8 & 0xffffff) / 16777216.0f; u2[k] = (float)((k * 40503u + 77u) & 0xffff) / 65536.0f;
			}
			int diff = 0;
			const djb::microfacet *objs[2] = { &gu, &bu };
			for (int w = 0; w < 2; ++w) {
				const djb::microfacet &m = *objs[w];
				m.eval(nb, &bi[0], &bo[0], &a[0], &ell);
				for (size_t k = 0; k < nb; ++k) { djb::vec3 r = m.eval(bi[k], bo[k], &ell); diff += std::memcmp(&r, &a[k], sizeof r) != 0; }
				m.evalp(nb, &bi[0], &bo[0], &a[0]);
				for (size_t k = 0; k < nb; ++k) { djb::vec3 r = m.evalp(bi[k], bo[k]); diff += std::memcmp(&r, &a[k], sizeof r) != 0; }
				m.evalp_is(nb, &u1[0], &u2[0], &bo[0], &a[0], &wi[0], &pd[0], &ell);
				for (size_t k = 0; k < nb; ++k) {
					djb::vec3 ii(0); float pp = 0; djb::vec3 r = m.evalp_is(u1[k], u2[k], bo[k], &ii, &pp, &ell);
					diff += std::memcmp(&r, &a[k], sizeof r) != 0 || std::memcmp(&pp, &pd[k], sizeof pp) != 0;
				}
			}
			expect("user Fresnel: batch == one-pair calls (bits differing)", (double)diff, 0.0);
		}
		try { djb::merl bad("/nonexistent/file.binary"); g_fail = 1; }
		catch (const djb::exc &e) { printf("djb::exc as expected: %s", e.what()); }
	} catch (const djb::exc &e) {
		fprintf(stderr, "djb::exc: %s\n", e.what());
		return 2;
	}
	return g_fail;
}
