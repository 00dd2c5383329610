// examples/api_fuzz.cpp -- the library's OWN classes driven through the reference's public interface with random arguments: lobes,
// parameterisations, the five Fresnel terms, every operator and query, the LEAN representation, the two published models, a UTIA file
// written by the program itself, isotropic and anisotropic fits.  Every result is printed bit for bit.
//
// Written against the REFERENCE's interface only; compiles unchanged against either header and must print the same bytes per seed:
//   api_fuzz <first seed> <number of seeds> [scratch directory [merl] [threads=N]]
//     merl: each seed also writes and fits a 35 MB MERL file;  threads=N: the seeds run concurrently on N host threads (same output)
//   g++ -I/root/reference -> the reference (oracle/Makefile: oracle/_ref/api_fuzz; seeds 1..4 kept as tests/golden/reftests/api_fuzz.txt)
//   g++ -I include -ldjb_hip -> this repository (host path or GPU)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdint.h>
#include <string>
#include <vector>
#include <thread>
#include <sys/stat.h>
#include <unistd.h>

#define DJ_BRDF_IMPLEMENTATION 1
#include "dj_brdf.h"

namespace {

// every line goes through `out`: stdout, or (threads mode) a buffer of the seed's own, printed in seed order afterwards
thread_local FILE *out = stdout;

struct rng {                                     // splitmix64: the same stream on every machine
	uint64_t s;
	explicit rng(uint64_t seed) : s(seed * 0x9E3779B97F4A7C15ull + 777u) {}
	uint64_t bits() { uint64_t z = (s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
	float u() { return (float)(bits() >> 40) * (1.0f / 16777216.0f); }
	float in(float a, float b) { return a + (b - a) * u(); }
	float log_in(float a, float b) { return (float)std::exp((double)in((float)std::log((double)a), (float)std::log((double)b))); }
	int below(int n) { return (int)(bits() % (uint64_t)n); }
	djb::vec3 dir(float zmin = 0.02f)
	{
		const float z = in(zmin, 1.0f), phi = in(0.0f, 6.2831853f), r = (float)std::sqrt(1.0 - (double)z * z);
		return djb::vec3(r * (float)std::cos((double)phi), r * (float)std::sin((double)phi), z);
	}
	// what a renderer's stray rays look like: now and then below the horizon, on it, on the normal, un-normalised
	djb::vec3 any_dir()
	{
		djb::vec3 d = dir();
		switch (below(12)) {
		case 0: d.z = -d.z; break;
		case 1: d = djb::vec3(d.x, d.y, 0.0f); break;
		case 2: d = djb::vec3(0, 0, 1); break;
		case 3: d = d * in(0.3f, 3.0f); break;
		default: break;
		}
		return d;
	}
};

void put(float v) { if (v != v) fprintf(out, " nan"); else fprintf(out, " %a", v); }
void show(const char *tag, const djb::vec3 &v) { fprintf(out, "%s", tag); put(v.x); put(v.y); put(v.z); fprintf(out, "\n"); }
void show1(const char *tag, float v) { fprintf(out, "%s", tag); put(v); fprintf(out, "\n"); }
void show_table(const char *tag, const std::vector<djb::float_t> &v)
{
	uint64_t h = 0xcbf29ce484222325ull;
	for (size_t k = 0; k < v.size(); ++k) {
		uint32_t w; float f = v[k]; memcpy(&w, &f, 4);
		if (f != f) w = 0x7fc00000u;
		for (int b = 0; b < 4; ++b) { h ^= (w >> (8 * b)) & 0xffu; h *= 0x100000001b3ull; }
	}
	fprintf(out, "%s n=%d fnv=%016llx", tag, (int)v.size(), (unsigned long long)h);
	if (!v.empty()) { put(v[0]); put(v[v.size() / 2]); put(v.back()); }
	fprintf(out, "\n");
}

djb::microfacet::params random_params(rng &g)
{
	switch (g.below(4)) {
	case 0: return djb::microfacet::params::standard();
	case 1: return djb::microfacet::params::isotropic(g.log_in(0.02f, 2.0f));
	case 2: return djb::microfacet::params::elliptic(g.log_in(0.02f, 2.0f), g.log_in(0.02f, 2.0f), g.in(-3.2f, 3.2f));
	default: return djb::microfacet::params::pdfparams(g.log_in(0.02f, 2.0f), g.log_in(0.02f, 2.0f), g.in(-0.95f, 0.95f),
	                                                   g.below(2) ? g.in(-0.5f, 0.5f) : 0.0f, g.below(2) ? g.in(-0.5f, 0.5f) : 0.0f);
	}
}

djb::fresnel::impl *random_fresnel(rng &g)
{
	switch (g.below(5)) {
	case 0: return new djb::fresnel::ideal();
	case 1: return new djb::fresnel::unpolarized(djb::vec3(g.in(1.02f, 3.0f), g.in(1.02f, 3.0f), g.log_in(1.02f, 60.0f)));
	case 2: return new djb::fresnel::schlick(djb::vec3(g.u(), g.u(), g.u()));
	case 3: return new djb::fresnel::sgd(djb::vec3(g.u(), g.u(), g.u()), djb::vec3(g.in(-0.3f, 0.3f), g.in(-0.3f, 0.3f), g.in(-0.3f, 0.3f)));
	default: {
		std::vector<djb::vec3> pts;
		const int n = 2 + g.below(40);
		for (int k = 0; k < n; ++k) pts.push_back(djb::vec3(g.in(0.0f, 1.5f), g.in(0.0f, 1.5f), g.in(0.0f, 1.5f)));
		return new djb::fresnel::spline(pts);
	}
	}
}

void params_round_trip(rng &g)
{
	djb::microfacet::params p = random_params(g);
	float a1, a2, phi, ax, ay, rho, tx, ty;
	p.get_ellipse(&a1, &a2, &phi); p.get_pdfparams(&ax, &ay, &rho, &tx, &ty);
	fprintf(out, "params ellipse"); put(a1); put(a2); put(phi); fprintf(out, " pdf"); put(ax); put(ay); put(rho); put(tx); put(ty); fprintf(out, "\n");
	djb::vec3 n; p.get_location(&n); show("  location", n);
	p.set_ellipse(g.log_in(0.05f, 1.0f), g.log_in(0.05f, 1.0f), g.in(-3.0f, 3.0f));
	p.set_location(djb::normalize(djb::vec3(g.in(-0.3f, 0.3f), g.in(-0.3f, 0.3f), 1.0f)));
	p.get_pdfparams(&ax, &ay, &rho, &tx, &ty);
	fprintf(out, "  after set_ellipse / set_location"); put(ax); put(ay); put(rho); put(tx); put(ty); fprintf(out, "\n");
	p.set_pdfparams(g.log_in(0.05f, 1.0f), g.log_in(0.05f, 1.0f), g.in(-0.9f, 0.9f), g.in(-0.2f, 0.2f), g.in(-0.2f, 0.2f));
	p.get_ellipse(&a1, &a2, &phi); p.get_location(&tx, &ty);
	fprintf(out, "  after set_pdfparams"); put(a1); put(a2); put(phi); put(tx); put(ty); fprintf(out, "\n");
	// the constructors ("prefer factories", dj_brdf.h:236-238) and a default-constructed object
	{
		djb::microfacet::params c0, c1(g.log_in(0.05f, 1.0f), g.log_in(0.05f, 1.0f), g.in(-3.0f, 3.0f)), c2(g.log_in(0.05f, 1.0f), g.log_in(0.05f, 1.0f), g.in(-0.9f, 0.9f), g.in(-0.3f, 0.3f), g.in(-0.3f, 0.3f));
		float v[5];
		c0.get_pdfparams(&v[0], &v[1], &v[2], &v[3], &v[4]); fprintf(out, "  ctor()"); for (int k = 0; k < 5; ++k) put(v[k]);
		c1.get_pdfparams(&v[0], &v[1], &v[2]); fprintf(out, " ctor(3)"); for (int k = 0; k < 3; ++k) put(v[k]);
		c2.get_ellipse(&v[0], &v[1], &v[2]); c2.get_location(&v[3], &v[4]); fprintf(out, " ctor(5)"); for (int k = 0; k < 5; ++k) put(v[k]);
		c2.set_location(g.in(-0.2f, 0.2f), g.in(-0.2f, 0.2f)); djb::vec3 nn; c2.get_location(&nn); show(" n", nn);
	}
	// LEAN / LEADR representation
	djb::beckmann::lrep l1, l2(g.in(-0.3f, 0.3f), g.in(-0.3f, 0.3f), g.in(0.1f, 1.0f), g.in(0.1f, 1.0f), g.in(-0.05f, 0.05f));
	djb::beckmann::params_to_lrep(p, &l1);
	djb::beckmann::lrep l3 = (l1 + l2) * g.in(0.2f, 2.0f);
	l3 *= g.in(0.5f, 1.5f);
	l3.shear(g.in(-0.2f, 0.2f), g.in(-0.2f, 0.2f));
	l3.scale(g.in(0.5f, 2.0f), g.in(0.5f, 2.0f));
	djb::beckmann::lrep l4 = l2; l4 += l1;
	djb::microfacet::params q, q2;
	djb::beckmann::lrep_to_params(l3, &q); djb::beckmann::lrep_to_params(l4, &q2);
	q.get_pdfparams(&ax, &ay, &rho, &tx, &ty);
	fprintf(out, "  lrep"); put(ax); put(ay); put(rho); put(tx); put(ty);
	q2.get_pdfparams(&ax, &ay, &rho, &tx, &ty);
	fprintf(out, " +="); put(ax); put(ay); put(rho); put(tx); put(ty); fprintf(out, "\n");
}

template <class Lobe>
void lobe_calls(const char *name, rng &g, const Lobe &b)
{
	for (int k = 0; k < 3; ++k) {
		const djb::microfacet::params p = random_params(g);
		const bool hostile = k == 2;
		const djb::vec3 i = hostile ? g.any_dir() : g.dir(), o = hostile ? g.any_dir() : g.dir();
		fprintf(out, "%s call %d%s\n", name, k, hostile ? " (stray)" : "");
		show("  eval", b.eval(i, o, &p)); show("  evalp", b.evalp(i, o, &p)); show1("  pdf", b.pdf(i, o, &p));
		const float u1 = g.u(), u2 = g.u();
		show("  sample", b.sample(u1, u2, o, &p));
		djb::vec3 wi; float pdf;
		show("  evalp_is", b.evalp_is(u1, u2, o, &wi, &pdf, &p)); show("    i", wi); show1("    pdf", pdf);
		if (!hostile) {
			djb::vec3 h, d; djb::brdf::io_to_hd(i, o, &h, &d);
			show("  io_to_hd h", h); show("    d", d);
			djb::vec3 i2, o2; djb::brdf::hd_to_io(h, d, &i2, &o2);
			show("  hd_to_io i", i2); show("    o", o2);
			show("  eval_hd", b.eval_hd(h, d, &p)); show("  evalp_hd", b.evalp_hd(h, d, &p));
			show1("  ndf", b.ndf(h, p)); show1("  gaf", b.gaf(h, i, o, p)); show1("  g1", b.g1(h, o, p)); show1("  sigma", b.sigma(o, p));
			const float x = g.in(-2.0f, 2.0f), y = g.in(-2.0f, 2.0f);
			show1("  p22", b.p22(x, y, p)); show1("  vp22", b.vp22(x, y, o, p)); show1("  vndf", b.vndf(h, o, p));
		}
	}
	const float c = g.u(), r = g.log_in(0.01f, 20.0f), u = g.in(0.001f, 0.999f);
	show("  fresnel", b.fresnel(c));
	show1("  p22_radial", b.p22_radial(r * r)); show1("  sigma_std_radial", b.sigma_std_radial(c));
	show1("  cdf_radial", b.cdf_radial(r)); show1("  qf_radial", b.qf_radial(u));
}

template <class Lobe>
void smith_queries(rng &g, const Lobe &b)
{
	const float u = g.in(0.001f, 0.999f), c = g.in(0.05f, 0.999f), s = (float)std::sqrt(1.0 - (double)c * c);
	const float q2 = b.qf2_radial(u, c, s);
	show1("  qf1", b.qf1(u)); show1("  qf2_radial", q2); show1("  qf3_radial", b.qf3_radial(g.in(0.001f, 0.999f), q2));
	// microfacet::qf2 / qf3 are not overridden by the radial lobes: the base class throws (dj_brdf.h:1782-1791)
	const djb::vec3 k = g.dir(0.1f);
	try { show1("  qf2", b.qf2(u, k)); } catch (const djb::exc &e) { fprintf(out, "  qf2: %s\n", e.what()); }
	try { show1("  qf3", b.qf3(u, k, q2)); } catch (const djb::exc &e) { fprintf(out, "  qf3: %s\n", e.what()); }
}

void one_seed(unsigned seed, const std::string &scratch, bool with_merl)
{
	rng g(seed);
	fprintf(out, "== seed %u\n", seed);
	params_round_trip(g);
	// the two analytic lobes with a random Fresnel term
	{
		djb::fresnel::impl *f = random_fresnel(g);
		const bool shadow = g.below(4) != 0;
		djb::ggx gx(*f, shadow);
		djb::beckmann bk(*f, shadow);
		fprintf(out, "fresnel(0.37) of the term:"); { const djb::vec3 v = f->eval(0.37f); put(v.x); put(v.y); put(v.z); } fprintf(out, " shadow %d\n", (int)gx.get_shadow());
		lobe_calls("ggx", g, gx); smith_queries(g, gx);
		lobe_calls("beckmann", g, bk); smith_queries(g, bk);
		// mutators
		djb::fresnel::impl *f2 = random_fresnel(g);
		gx.set_fresnel(*f2); gx.set_shadow(!shadow);
		show("  after set_fresnel / set_shadow", gx.eval(g.dir(), g.dir()));
		show("  get_fresnel().eval", gx.get_fresnel().eval(g.u()));
		// a fit of the lobe as it stands
		const int res = 8 + g.below(56);
		djb::tabular tab(bk, res, g.below(2) != 0);
		float ab, ag;
		djb::tabular::fit_beckmann_parameters(tab).get_ellipse(&ab, NULL);
		djb::tabular::fit_ggx_parameters(tab).get_ellipse(&ag, NULL);
		fprintf(out, "tabular(beckmann, %d)", res); put(ab); put(ag); fprintf(out, "\n");
		show_table("  p22", tab.get_p22v()); show_table("  sigma", tab.get_sigmav()); show_table("  cdf", tab.get_cdfv()); show_table("  qf", tab.get_qfv());
		lobe_calls("tabular", g, tab);
		// fits of fits: a table as the source of another table, isotropic and anisotropic
		{
			djb::tabular tt(tab, 8 + g.below(24), g.below(2) != 0);
			djb::tabular_anisotropic ta(tab, 6 + g.below(5), 8 + g.below(8));
			djb::tabular tat(ta, 8 + g.below(16));
			int ec, ac;
			show_table("  tabular(tabular) p22", tt.get_p22v()); show_table("  aniso(tabular) sigma", ta.get_sigmav(&ec, &ac)); show_table("  tabular(aniso(tabular)) qf", tat.get_qfv());
			float a1, a2; djb::tabular::fit_ggx_parameters(tt).get_ellipse(&a1, NULL); djb::tabular::fit_beckmann_parameters(tat).get_ellipse(&a2, NULL);
			fprintf(out, "  alphas"); put(a1); put(a2); fprintf(out, "\n");
		}
		delete f; delete f2;
	}
	// the published models
	{
		static const char *const names[6] = { "gold-metallic-paint", "chrome", "blue-fabric", "alum-bronze", "white-marble", "teflon" };
		const char *name = names[g.below(6)];
		djb::sgd s(name); djb::abc a(name);
		const djb::vec3 i = g.dir(), o = g.dir(), h = djb::normalize(i + o);
		fprintf(out, "models %s\n", name);
		show("  sgd.eval", s.eval(i, o)); show("  sgd.ndf", s.ndf(h)); show("  sgd.gaf", s.gaf(h, i, o)); show("  sgd.g1", s.g1(o)); show("  sgd.fresnel", s.fresnel(g.u()));
		show("  abc.eval", a.eval(i, o)); show("  abc.ndf", a.ndf(h)); show1("  abc.gaf", a.gaf(h, i, o)); show("  abc.fresnel", a.fresnel(g.u()));
		show("  sgd.eval stray", s.eval(g.any_dir(), g.any_dir())); show("  abc.eval stray", a.eval(g.any_dir(), g.any_dir()));
		show1("  sgd.pdf", s.pdf(i, o)); show("  abc.sample", a.sample(g.u(), g.u(), o));
		djb::tabular tab(a, 12 + g.below(30));
		float ag;
		djb::tabular::fit_ggx_parameters(tab).get_ellipse(&ag, NULL);
		fprintf(out, "  tabular(abc) ggx"); put(ag); fprintf(out, "\n");
		show1("  tab.pdf", tab.pdf(i, o)); show("  tab.sample", tab.sample(g.u(), g.u(), o));
	}
	// lambert
	{
		djb::lambert l;
		djb::lambert::params lp(djb::vec3(g.u(), g.u(), g.u())), lp0;
		lp0.m_reflectance = lp.m_reflectance * 0.5f + lp0.m_reflectance * 0.25f;
		const djb::vec3 i = g.any_dir(), o = g.any_dir();
		show("lambert.eval", l.eval(i, o, &lp)); show("  eval(lp0)", l.eval(i, o, &lp0)); show("  eval()", l.eval(i, o)); show("  evalp", l.evalp(i, o)); show1("  pdf", l.pdf(i, o)); show("  sample", l.sample(g.u(), g.u(), o));
	}
	// a UTIA file written here: look-ups, and an anisotropic fit of it
	{
		const std::string path = scratch + "/api_fuzz_utia.bin";
		std::vector<double> tab(3 * 288 * 288);
		const double base = g.in(5.0f, 60.0f), gloss = g.in(10.0f, 120.0f), width = g.in(0.5f, 6.0f);
		for (int c = 0; c < 3; ++c)
		for (int ti = 0; ti < 6; ++ti) for (int pi = 0; pi < 48; ++pi) for (int tv = 0; tv < 6; ++tv) for (int pv = 0; pv < 48; ++pv) {
			const int dp = (pi - pv + 72) % 48 - 24, dt = ti - tv;                     // 0 = mirror azimuth, same elevation
			const double spec = gloss / (1.0 + width * (double)(dp * dp) / 16.0 + 2.0 * (double)(dt * dt));
			tab[(size_t)c * 288 * 288 + 288 * (48 * ti + pi) + 48 * tv + pv] = base * (1.0 + 0.2 * c) + spec + ((ti * 7 + pi * 3 + tv * 5 + pv) % 11) * 0.25;
		}
		if (g.below(3) == 0) tab[(size_t)g.below(3 * 288 * 288)] = -4.0;             // utia::normalize clamps negative samples
		FILE *f = fopen(path.c_str(), "wb");
		if (!f || fwrite(&tab[0], sizeof(double), tab.size(), f) != tab.size()) { fprintf(out, "cannot write %s\n", path.c_str()); exit(2); }
		fclose(f);
		djb::utia u(path.c_str());
		for (int k = 0; k < 4; ++k) { const djb::vec3 i = k == 3 ? g.any_dir() : g.dir(), o = k == 3 ? g.any_dir() : g.dir(); show("utia.eval", u.eval(i, o)); show("  evalp", u.evalp(i, o)); }
		const int elev = 6 + g.below(6), azim = 8 + g.below(10);
		djb::tabular_anisotropic ta(u, elev, azim);
		float v[5];
		djb::tabular_anisotropic::fit_beckmann_parameters(ta).get_pdfparams(&v[0], &v[1], &v[2], &v[3], &v[4]);
		fprintf(out, "tabular_anisotropic(utia, %d, %d) beckmann", elev, azim); for (int k = 0; k < 5; ++k) put(v[k]);
		djb::tabular_anisotropic::fit_ggx_parameters(ta).get_pdfparams(&v[0], &v[1], &v[2], &v[3], &v[4]);
		fprintf(out, " ggx"); for (int k = 0; k < 5; ++k) put(v[k]); fprintf(out, "\n");
		int ec, ac;
		show_table("  p22", ta.get_p22v(&ec, &ac)); fprintf(out, "  grid %d %d\n", ec, ac); show_table("  sigma", ta.get_sigmav(&ec, &ac));
		const float phi = g.in(0.0f, 6.28f), th = g.in(0.0f, 1.5f), w = g.in(0.01f, 0.99f);
		show1("  pdf1", ta.pdf1(phi)); show1("  cdf1", ta.cdf1(phi)); show1("  qf1", ta.qf1(w));
		show1("  pdf2", ta.pdf2(th, phi)); show1("  cdf2", ta.cdf2(th, phi)); show1("  qf2", ta.qf2(w, phi));
		const djb::vec3 i = g.dir(), o = g.dir();
		const djb::microfacet::params p = random_params(g);
		show("  eval", ta.eval(i, o, &p)); show1("  pdf", ta.pdf(i, o, &p));
		djb::vec3 wi; float pdf;
		show("  evalp_is", ta.evalp_is(g.u(), g.u(), o, &wi, &pdf, &p)); show("    i", wi); show1("    pdf", pdf);
		remove(path.c_str());
	}
	// optional (35 MB per seed): a MERL file written here -- nearest-bin look-ups, the samples, a fit at a random resolution
	if (with_merl) {
		const std::string path = scratch + "/api_fuzz_merl.binary";
		const int dims[3] = { 90, 90, 180 };
		const size_t n = 90u * 90u * 180u;
		std::vector<double> tab(3 * n);
		const double alpha = g.log_in(0.03f, 0.6f), kd = g.in(0.0f, 0.4f), a2 = alpha * alpha;
		const uint64_t salt = g.bits();
		for (int th = 0; th < 90; ++th) for (int td = 0; td < 90; ++td) for (int pd = 0; pd < 180; ++pd) {
			const size_t k = (size_t)pd + 180u * ((size_t)td + 90u * (size_t)th);
			const double theta_h = (double)(th * th) / 8100.0 * 1.5707963267948966;             // MERL's non-linear theta_h bins
			const double t = std::tan(theta_h), c = std::cos(theta_h);
			const double d = a2 / (3.141592653589793 * c * c * c * c * (a2 + t * t) * (a2 + t * t) + 1e-300);
			const double fr = 0.04 + 0.96 * std::pow(1.0 - std::cos((double)td * 0.017453292519943295), 5.0);
			uint64_t hsh = (k + 1) * 0x9E3779B97F4A7C15ull ^ salt; hsh ^= hsh >> 29; hsh *= 0xBF58476D1CE4E5B9ull; hsh ^= hsh >> 32;
			const double noise = 1.0 + 0.1 * ((double)(hsh & 0xffff) / 65536.0 - 0.5);
			const bool hole = (hsh >> 20) % 257 == 0;                                               // MERL marks invalid bins with negative samples
			for (int ch = 0; ch < 3; ++ch) {
				const double scale = ch == 0 ? 1500.0 : (ch == 1 ? 1500.0 / 1.15 : 1500.0 / 1.66);
				tab[ch * n + k] = hole ? -1.0 : (kd * (1.0 - 0.2 * ch) / 3.141592653589793 + fr * d * 0.25 * (1.0 + 0.001 * pd)) * noise * scale;
			}
		}
		FILE *f = fopen(path.c_str(), "wb");
		if (!f || fwrite(dims, sizeof(int), 3, f) != 3 || fwrite(&tab[0], sizeof(double), tab.size(), f) != tab.size()) { fprintf(out, "cannot write %s\n", path.c_str()); exit(2); }
		fclose(f);
		djb::merl m(path.c_str());
		fprintf(out, "merl alpha %a samples %d", (float)alpha, (int)m.get_samples().size());
		{ const std::vector<double> &sm = m.get_samples(); double acc = 0; for (size_t k = 0; k < sm.size(); k += 9973) acc += sm[k]; fprintf(out, " sum %a\n", acc); }
		for (int k = 0; k < 6; ++k) { const djb::vec3 i = k == 5 ? g.any_dir() : g.dir(), o = k == 5 ? g.any_dir() : g.dir(); show("  eval", m.eval(i, o)); show("  evalp", m.evalp(i, o)); }
		const int res = 16 + g.below(90);
		djb::tabular tab_m(m, res, g.below(2) != 0);
		float ab, ag;
		djb::tabular::fit_beckmann_parameters(tab_m).get_ellipse(&ab, NULL);
		djb::tabular::fit_ggx_parameters(tab_m).get_ellipse(&ag, NULL);
		fprintf(out, "  tabular(merl, %d)", res); put(ab); put(ag); fprintf(out, "\n");
		show_table("  p22", tab_m.get_p22v()); show_table("  sigma", tab_m.get_sigmav()); show_table("  cdf", tab_m.get_cdfv()); show_table("  qf", tab_m.get_qfv());
		show("  fitted fresnel", tab_m.fresnel(g.u()));
		const djb::vec3 i = g.dir(), o = g.dir();
		show("  tab.eval", tab_m.eval(i, o)); show1("  tab.pdf", tab_m.pdf(i, o)); show("  tab.sample", tab_m.sample(g.u(), g.u(), o));
		remove(path.c_str());
	}
	// the same objects behind base-class pointers, as a renderer holds them (mitsuba/dj_brdf.cpp keeps a djb::microfacet *)
	{
		djb::fresnel::impl *f = random_fresnel(g);
		std::vector<djb::brdf *> all;
		djb::radial *rad[2] = { new djb::ggx(*f), new djb::beckmann(*f, false) };
		djb::microfacet *mf = new djb::tabular(*rad[1], 10 + g.below(20));
		all.push_back(rad[0]); all.push_back(rad[1]); all.push_back(mf); all.push_back(new djb::lambert); all.push_back(new djb::sgd("chrome")); all.push_back(new djb::abc("teflon"));
		const djb::microfacet::params p = random_params(g);
		const djb::vec3 i = g.dir(), o = g.dir();
		for (size_t k = 0; k < all.size(); ++k) {
			const djb::brdf *b = all[k];
			const void *up = k < 3 ? (const void *)&p : NULL;
			fprintf(out, "brdf* %d", (int)k); { const djb::vec3 v = b->eval(i, o, up); put(v.x); put(v.y); put(v.z); } put(b->pdf(i, o, up));
			djb::vec3 wi; float pdf; const djb::vec3 w = b->evalp_is(g.u(), g.u(), o, &wi, &pdf, up); put(w.x); put(wi.z); put(pdf); fprintf(out, "\n");
		}
		for (int k = 0; k < 2; ++k) { fprintf(out, "radial* %d", k); put(rad[k]->p22_radial(g.log_in(1e-3f, 30.0f))); put(rad[k]->qf_radial(g.in(0.01f, 0.99f))); put(rad[k]->qf2_radial(g.in(0.01f, 0.99f), 0.6f, 0.8f)); fprintf(out, " vndf %d\n", (int)rad[k]->supports_smith_vndf_sampling()); }
		djb::fresnel::impl *f2 = random_fresnel(g);
		mf->set_fresnel(*f2); mf->set_shadow(false);
		fprintf(out, "microfacet* shadow %d vndf %d", (int)mf->get_shadow(), (int)mf->supports_smith_vndf_sampling()); { const djb::vec3 v = mf->fresnel(g.u()); put(v.x); put(v.y); put(v.z); }
		put(mf->ndf(djb::normalize(i + o), p)); put(mf->sigma(i, p)); fprintf(out, " tabular? %d\n", dynamic_cast<djb::tabular *>(mf) != NULL && dynamic_cast<djb::ggx *>(mf) == NULL);
		for (size_t k = 0; k < all.size(); ++k) delete all[k];            // virtual destructors, through the base
		delete f; delete f2;
	}
	// the file-static helpers of the implementation section (dj_brdf.h:650-765, 1181-1249): visible to any program that defines
	// DJ_BRDF_IMPLEMENTATION, e.g. to a user-defined lobe's own sample()
	{
		const float x = g.in(-3.0f, 3.0f), u = g.in(-0.9999f, 0.9999f);
		fprintf(out, "helpers erf"); put(djb::erf(x)); put(djb::erf(g.log_in(1e-4f, 6.0f))); fprintf(out, " erfinv"); put(djb::erfinv(u)); put(djb::erfinv(1.0f - g.log_in(1e-7f, 1e-2f))); fprintf(out, "\n");
		float th, ph; const djb::vec3 d = g.below(6) ? g.any_dir() : djb::vec3(0, 0, g.below(2) ? 1.0f : -1.0f);
		djb::xyz_to_theta_phi(d, &th, &ph); fprintf(out, "  theta phi"); put(th); put(ph);
		float cx, cy; djb::uniform_to_concentric(g.below(8) ? g.u() : 0.5f, g.below(8) ? g.u() : 0.5f, &cx, &cy); fprintf(out, " concentric"); put(cx); put(cy); fprintf(out, "\n");
		show("  rotate_vector", djb::rotate_vector(g.dir(), djb::normalize(djb::vec3(g.in(-1.0f, 1.0f), g.in(-1.0f, 1.0f), g.in(-1.0f, 1.0f))), g.in(-4.0f, 4.0f)));
		show("  cross", djb::cross(g.dir(), g.dir())); fprintf(out, "  max3 %d inversesqrt", djb::max3(g.below(9), g.below(9), g.below(9))); put(djb::inversesqrt(g.log_in(1e-3f, 1e3f))); fprintf(out, "\n");
		{	// vec3 and the Fresnel utilities
			djb::vec3 a = g.dir(), b = g.dir(), c(g.in(0.0f, 3.1f), g.in(0.0f, 6.2f));
			const double raw[3] = { (double)g.u() * 1.000000123, (double)g.u() / 3.0, 1e-40 };
			show("  vec3(theta, phi)", c); show("  from_raw", djb::vec3::from_raw(raw));
			show("  a*b a/b", a * b + a / (b + djb::vec3(0.5f))); show("  a/s s*a", a / g.in(0.1f, 3.0f) + g.u() * b);
			a += b; a *= b; a *= g.u(); show("  += *= *=", a); show("  vec3(s)", djb::vec3(g.u()) - b);
			float f0, ior; djb::vec3 v0, v1;
			djb::fresnel::ior_to_f0(g.in(1.0f, 4.0f), &f0); djb::fresnel::f0_to_ior(g.in(0.0f, 0.99f), &ior);
			djb::fresnel::ior_to_f0(djb::vec3(g.in(1.0f, 3.0f), g.in(1.0f, 3.0f), g.in(1.0f, 3.0f)), &v0); djb::fresnel::f0_to_ior(djb::vec3(g.u(), g.u(), g.u()) * 0.98f, &v1);
			fprintf(out, "  ior_to_f0"); put(f0); put(ior); show(" v", v0 + v1);
		}
		std::vector<float> tab; std::vector<djb::vec3> tab3;
		const int n = 2 + g.below(30), w = 2 + g.below(6), h = 2 + g.below(6);
		for (int k = 0; k < n; ++k) { tab.push_back(g.in(-1.0f, 2.0f)); tab3.push_back(djb::vec3(g.u(), g.u(), g.u())); }
		std::vector<float> grid; for (int k = 0; k < w * h; ++k) grid.push_back(g.in(0.0f, 5.0f));
		const float s = g.in(-0.2f, 1.2f);
		fprintf(out, "  spline"); put(djb::spline::eval(tab, djb::spline::uwrap_edge, s)); put(djb::spline::eval(tab, djb::spline::uwrap_repeat, g.u()));
		show(" vec3", djb::spline::eval(tab3, djb::spline::uwrap_edge, g.u()));
		fprintf(out, "  eval2d"); put(djb::spline::eval2d(grid, w, h, djb::spline::uwrap_edge, g.u(), djb::spline::uwrap_repeat, g.u()));
		fprintf(out, " lerp"); put(djb::spline::lerp(g.u(), g.u(), g.u())); fprintf(out, " wrap %d %d\n", djb::spline::uwrap_repeat(g.below(40) - 20, 7), djb::spline::uwrap_edge(g.below(40) - 20, 7));
	}
	// the exception type is the caller's to throw as well: a printf-style message, 255 characters kept (dj_brdf.h:54-59, 578-587)
	try { throw djb::exc("user error %d: %s / %.3f\n", g.below(100), "lobe", (double)g.u()); } catch (const std::exception &e) { fprintf(out, "caught %s", e.what()); }
	try { throw djb::exc("%s", std::string(300 + g.below(50), 'x').c_str()); } catch (const djb::exc &e) { fprintf(out, "long message kept %d\n", (int)e.m_str.size()); }
	// errors are the reference's
	try { djb::sgd nope("no-such-material"); fprintf(out, "no exception\n"); } catch (const djb::exc &e) { fprintf(out, "exc: %s", e.what()); }
	try { djb::utia nope((scratch + "/does-not-exist.bin").c_str()); fprintf(out, "no exception\n"); } catch (const djb::exc &e) { fprintf(out, "exc raised for a missing file\n"); }
}

} // namespace

// one seed into a buffer of its own
std::string seed_to_string(unsigned seed, const std::string &scratch, bool with_merl)
{
	char *buf = NULL; size_t len = 0;
	FILE *f = open_memstream(&buf, &len);
	out = f;
	char sub[32]; snprintf(sub, sizeof sub, "/s%u", seed);                 // the files the seed writes: a name of its own
	mkdir((scratch + sub).c_str(), 0777);
	one_seed(seed, scratch + sub, with_merl);
	rmdir((scratch + sub).c_str());
	fclose(f); out = stdout;
	std::string r(buf, len); free(buf);
	return r;
}

int main(int argc, char **argv)
{
	const unsigned first = argc > 1 ? (unsigned)atoi(argv[1]) : 1u, count = argc > 2 ? (unsigned)atoi(argv[2]) : 2u;
	const std::string scratch = argc > 3 ? argv[3] : "/tmp";
	bool with_merl = false; int threads = 0;
	for (int a = 4; a < argc; ++a) { if (!strcmp(argv[a], "merl")) with_merl = true; else if (!strncmp(argv[a], "threads=", 8)) threads = atoi(argv[a] + 8); }
	if (threads <= 0) {
		for (unsigned s = first; s < first + count; ++s) one_seed(s, scratch, with_merl);
		return 0;
	}
	// threads=N: the seeds are dealt to N host threads that share the process's default context and run CONCURRENTLY; the output is
	// printed in seed order and must be the sequential run's
	std::vector<std::string> text(count);
	std::vector<std::thread> pool;
	for (int t = 0; t < threads; ++t)
		pool.push_back(std::thread([&, t]() { for (unsigned k = (unsigned)t; k < count; k += (unsigned)threads) text[k] = seed_to_string(first + k, scratch, with_merl); }));
	for (size_t t = 0; t < pool.size(); ++t) pool[t].join();
	for (unsigned k = 0; k < count; ++k) fputs(text[k].c_str(), stdout);
	return 0;
}
