// examples/custom_brdf_fuzz.cpp -- randomised companion of custom_brdf.cpp: user-defined BRDFs, Fresnel terms and NDFs with RANDOM
// parameters, fitted at random resolutions and evaluated at random directions, every result printed bit for bit.
//
// Written against the REFERENCE's interface only; it compiles unchanged against either header and must print the same bytes for the
// same seeds:   custom_brdf_fuzz <first seed> <number of seeds> [threads=N]      (threads=N: the seeds run concurrently, same output)
//   g++ -I/root/reference -> the reference (tests/golden/make_reftests.sh keeps the output of seeds 1..6 as a fixture;
//                            tests/test_user_fuzz.py compares more seeds live where the reference is present)
//   g++ -I include -ldjb_hip -> this repository (host path or GPU)
// Tables are summarised by a 64-bit FNV hash over their float bits (any differing bit changes the line) plus a few entries.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdint.h>
#include <string>
#include <vector>
#include <stdexcept>
#include <thread>

#define DJ_BRDF_IMPLEMENTATION 1
#include "dj_brdf.h"

namespace {

// every line goes through `out`: stdout, or (threads mode) a buffer of the seed's own, printed in seed order afterwards
thread_local FILE *out = stdout;

struct rng {                                     // splitmix64: the same stream on every machine
	uint64_t s;
	explicit rng(uint64_t seed) : s(seed * 0x9E3779B97F4A7C15ull + 12345u) {}
	uint64_t bits() { uint64_t z = (s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
	float u() { return (float)(bits() >> 40) * (1.0f / 16777216.0f); }                 // [0, 1)
	float in(float a, float b) { return a + (b - a) * u(); }
	float log_in(float a, float b) { return (float)std::exp((double)in((float)std::log((double)a), (float)std::log((double)b))); }
	int below(int n) { return (int)(bits() % (uint64_t)n); }
	djb::vec3 dir()                                  // upper hemisphere, not normalised to the last bit on purpose: what a caller passes
	{
		const float z = in(0.05f, 1.0f), phi = in(0.0f, 6.2831853f), r = (float)std::sqrt(1.0 - (double)z * z);
		return djb::vec3(r * (float)std::cos((double)phi), r * (float)std::sin((double)phi), z);
	}
};

// ---- user-defined BRDFs: one class, five shapes, random coefficients
class lobe : public djb::brdf {
public:
	lobe(rng &g) : m_shape(g.below(5)), m_kd(g.in(0.0f, 0.3f), g.in(0.0f, 0.3f), g.in(0.0f, 0.3f)), m_ks(g.in(0.1f, 1.0f), g.in(0.1f, 1.0f), g.in(0.1f, 1.0f)),
	               m_a(g.log_in(0.05f, 0.9f)), m_b(g.log_in(0.05f, 0.9f)), m_n(g.log_in(2.0f, 300.0f)) {}
	djb::vec3 eval(const djb::vec3 &i, const djb::vec3 &o, const void *user_param = NULL) const
	{
		(void)user_param;
		if (!(i.z > 0.0f && o.z > 0.0f)) return djb::vec3(0);
		const djb::vec3 h = djb::normalize(i + o);
		float s = 0.0f;
		switch (m_shape) {
		case 0: {                                 // Phong around the mirror direction
			float c = djb::dot(djb::vec3(-o.x, -o.y, o.z), i);
			if (!(c > 0.0f)) c = 0.0f;
			s = (float)(((double)m_n + 2.0) / (2.0 * M_PI) * std::pow((double)c, (double)m_n));
		} break;
		case 1: {                                 // Ward
			const float tx = h.x / m_a, ty = h.y / m_b, q = (tx * tx + ty * ty) / (h.z * h.z);
			s = (float)(std::exp(-(double)q) / (4.0 * M_PI * (double)(m_a * m_b) * std::sqrt((double)(i.z * o.z))));
		} break;
		case 2: {                                 // Blinn with a Schlick-like rise towards grazing angles
			const double m = 1.0 - (double)djb::dot(o, h);
			s = (float)(((double)m_n + 2.0) / (8.0 * M_PI) * std::pow((double)h.z, (double)m_n) * (0.2 + 0.8 * m * m * m * m * m) / (double)(i.z + o.z));
		} break;
		case 3: {                                 // a Lorentzian of the half-vector slope (long tail)
			const double r2 = (double)(h.x * h.x + h.y * h.y) / ((double)(h.z * h.z) * (double)(m_a * m_a));
			s = (float)(1.0 / (M_PI * (double)(m_a * m_a) * (1.0 + r2) * (1.0 + r2) * 4.0 * (double)(i.z * o.z)));
		} break;
		default: {                                // a lobe with a hard edge: discontinuous in the half vector
			s = h.z > 1.0f - 0.5f * m_a ? 1.0f / (m_a * (float)M_PI) : 0.02f;
		} break;
		}
		return m_kd / (float)M_PI + m_ks * s;
	}
	int shape() const { return m_shape; }
private:
	int m_shape;
	djb::vec3 m_kd, m_ks;
	float m_a, m_b, m_n;
};

// ---- a user-defined Fresnel term with random coefficients
class my_fresnel : public djb::fresnel::impl {
public:
	my_fresnel(rng &g) : m_f0(g.in(0.02f, 0.95f), g.in(0.02f, 0.95f), g.in(0.02f, 0.95f)), m_p(g.in(2.0f, 7.0f)) {}
	djb::vec3 eval(float cos_theta_d) const
	{
		const float w = (float)std::pow(1.0 - (double)cos_theta_d, (double)m_p);
		return m_f0 + (djb::vec3(1) - m_f0) * w;
	}
	djb::fresnel::impl *copy() const { return new my_fresnel(*this); }
private:
	djb::vec3 m_f0;
	float m_p;
};

// ---- a user-defined radial NDF: p22(r^2) = (g - 1) / (pi (1 + r^2)^g) with a random g
class student : public djb::radial {
public:
	student(float g, const djb::fresnel::impl &f, bool shadow) : djb::radial(f, shadow), m_g(g) {}
	bool supports_smith_vndf_sampling() const { return false; }
	float p22_radial(float r_sqr) const { return (float)(((double)m_g - 1.0) / (M_PI * std::pow(1.0 + (double)r_sqr, (double)m_g))); }
	float sigma_std_radial(float cos_theta_k) const { const float c = cos_theta_k; return (float)((1.0 + (double)(c * (1.0f + c)) * 0.5) * 0.5); }
	float cdf_radial(float r) const { return (float)(1.0 - std::pow(1.0 + (double)(r * r), 1.0 - (double)m_g)); }
	float qf_radial(float u) const { return (float)std::sqrt(std::pow(1.0 - (double)u, 1.0 / (1.0 - (double)m_g)) - 1.0); }
private:
	float m_g;
};

// ---- a user-defined BRDF that brings its own importance sampling, built from the helpers the reference's implementation section
// offers (uniform_to_concentric, rotate_vector, erf): sample / pdf overridden, evalp_is inherited (dj_brdf.h:816-828 calls the overrides)
class sampled_lobe : public djb::brdf {
public:
	sampled_lobe(rng &g) : m_w(g.in(0.2f, 2.0f)), m_tilt(g.in(-0.4f, 0.4f)) {}
	djb::vec3 eval(const djb::vec3 &i, const djb::vec3 &o, const void *user_param = NULL) const
	{ (void)user_param; if (!(i.z > 0.0f && o.z > 0.0f)) return djb::vec3(0); return djb::vec3(0.3f + 0.5f * djb::erf(m_w * djb::dot(i, o)), 0.2f, 0.1f * i.z); }
	djb::vec3 sample(float u1, float u2, const djb::vec3 &o, const void *user_param = NULL) const
	{
		(void)user_param; (void)o;
		float x, y; djb::uniform_to_concentric(u1, u2, &x, &y);
		const djb::vec3 d(x, y, (float)std::sqrt(djb::max(0.0, 1.0 - (double)x * x - (double)y * y)));
		return djb::rotate_vector(d, djb::vec3(0, 1, 0), m_tilt);        // a tilted cosine lobe
	}
	float pdf(const djb::vec3 &i, const djb::vec3 &o, const void *user_param = NULL) const
	{ (void)user_param; (void)o; const djb::vec3 d = djb::rotate_vector(i, djb::vec3(0, 1, 0), -m_tilt); return djb::max(d.z, 0.0f) / (float)M_PI; }
private:
	float m_w, m_tilt;
};

// ---- user code that throws: its exceptions must reach the caller as they are (type and message), also when the throwing function was
// called back by the library (an NDF), and the object must stay usable
class throwing_ndf : public djb::radial {
public:
	throwing_ndf(float limit, bool smith) : m_limit(limit), m_smith(smith) {}
	bool supports_smith_vndf_sampling() const { return m_smith; }        // true without qf2_radial / qf3_radial: the base class throws "Not Implemented"
	float p22_radial(float r_sqr) const { if (r_sqr > m_limit) throw std::out_of_range("slope beyond the table"); return (float)(1.0 / (M_PI * (1.0 + (double)r_sqr) * (1.0 + (double)r_sqr))); }
	float sigma_std_radial(float cos_theta_k) const { return 0.5f * (1.0f + cos_theta_k); }
	float cdf_radial(float r) const { return r * r / (1.0f + r * r); }
	float qf_radial(float u) const { return (float)std::sqrt((double)u / (1.0 - (double)u)); }
private:
	float m_limit;
	bool m_smith;
};

// ---- an NDF defined at the microfacet level that samples the Smith way: qf2 / qf3 overridden (dj_brdf.h:273-275, 1769-1791)
class smith_user : public djb::microfacet {
public:
	smith_user(float a, const djb::fresnel::impl &f) : djb::microfacet(f), m_a(a) {}
	bool supports_smith_vndf_sampling() const { return true; }
	float qf2(float u, const djb::vec3 &k) const { return (float)(((double)u - 0.5) * 2.0 * (double)m_a / (0.2 + (double)k.z)); }
	float qf3(float u, const djb::vec3 &k, float qf2_) const { return (float)(((double)u - 0.5) * (double)m_a * (1.0 + (double)(qf2_ * qf2_)) + 0.1 * (double)k.y); }
protected:
	float sigma_std(const djb::vec3 &k) const { return (float)(0.5 * ((double)k.z + std::sqrt((double)(k.z * k.z) + (double)(m_a * m_a) * (double)(k.x * k.x + k.y * k.y)))); }
	float p22_std(float x, float y) const { const double t = 1.0 + (double)(x * x + y * y) / (double)(m_a * m_a); return (float)(1.0 / (M_PI * (double)(m_a * m_a) * t * t)); }
	void sample_vp22_std_nmap(float, float, const djb::vec3 &, float *x, float *y) const { *x = *y = 0.0f; }
private:
	float m_a;
};

void put(float v) { if (v != v) fprintf(out, " nan"); else fprintf(out, " %a", v); }          // the sign of a NaN is not part of the contract
void show(const char *tag, const djb::vec3 &v) { fprintf(out, "%s", tag); put(v.x); put(v.y); put(v.z); fprintf(out, "\n"); }
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

void one_seed(unsigned seed)
{
	rng g(seed);
	fprintf(out, "== seed %u\n", seed);
	// 1. an isotropic fit of a random lobe at a random resolution
	{
		lobe l(g);
		const int res = 8 + g.below(90);
		const bool shadow = g.below(2) != 0;
		djb::tabular tab(l, res, shadow);
		float ab, ag;
		djb::tabular::fit_beckmann_parameters(tab).get_ellipse(&ab, NULL);
		djb::tabular::fit_ggx_parameters(tab).get_ellipse(&ag, NULL);
		fprintf(out, "tabular(shape %d, %d, %d):", l.shape(), res, (int)shadow); put(ab); put(ag); fprintf(out, "\n");
		show_table("  p22", tab.get_p22v()); show_table("  sigma", tab.get_sigmav());
		show_table("  cdf", tab.get_cdfv()); show_table("  qf", tab.get_qfv());
		for (int k = 0; k < 3; ++k) {
			const djb::vec3 i = g.dir(), o = g.dir();
			djb::microfacet::params pr = djb::microfacet::params::elliptic(g.log_in(0.05f, 1.5f), g.log_in(0.05f, 1.5f), g.in(0.0f, 3.0f));
			show("  eval", tab.eval(i, o, k ? &pr : NULL)); fprintf(out, "  pdf"); put(tab.pdf(i, o, k ? &pr : NULL)); fprintf(out, "\n");
			djb::vec3 wi; float pdf;
			show("  evalp_is", tab.evalp_is(g.u(), g.u(), o, &wi, &pdf, k ? &pr : NULL)); show("    i", wi); fprintf(out, "    pdf"); put(pdf); fprintf(out, "\n");
			show("  fresnel", tab.fresnel(g.u()));
		}
		// the base-class operators of the user's own object
		const djb::vec3 i = g.dir(), o = g.dir();
		show("  lobe.evalp", l.evalp(i, o)); fprintf(out, "  lobe.pdf"); put(l.pdf(i, o)); fprintf(out, "\n");
		djb::vec3 wi; float pdf;
		show("  lobe.evalp_is", l.evalp_is(g.u(), g.u(), o, &wi, &pdf)); show("    i", wi);
	}
	// 2. an anisotropic fit of another one on a small grid
	{
		lobe l(g);
		const int elev = 6 + g.below(8), azim = 8 + g.below(12);
		djb::tabular_anisotropic tab(l, elev, azim);
		float v[5];
		djb::tabular_anisotropic::fit_beckmann_parameters(tab).get_pdfparams(&v[0], &v[1], &v[2], &v[3], &v[4]);
		fprintf(out, "tabular_anisotropic(shape %d, %d, %d) beckmann", l.shape(), elev, azim); for (int k = 0; k < 5; ++k) put(v[k]); fprintf(out, "\n");
		djb::tabular_anisotropic::fit_ggx_parameters(tab).get_pdfparams(&v[0], &v[1], &v[2], &v[3], &v[4]);
		fprintf(out, "  ggx"); for (int k = 0; k < 5; ++k) put(v[k]); fprintf(out, "\n");
		const djb::vec3 i = g.dir(), o = g.dir();
		show("  eval", tab.eval(i, o)); fprintf(out, "  pdf"); put(tab.pdf(i, o)); fprintf(out, "\n");
		djb::vec3 wi; float pdf;
		show("  evalp_is", tab.evalp_is(g.u(), g.u(), o, &wi, &pdf)); show("    i", wi);
	}
	// 3. the library's lobes with a user-defined Fresnel term, and fitted from there
	{
		my_fresnel f(g);
		djb::ggx gx(f, g.below(2) != 0);
		djb::beckmann bk(f);
		djb::microfacet::params pr = djb::microfacet::params::elliptic(g.log_in(0.05f, 1.0f), g.log_in(0.05f, 1.0f), g.in(0.0f, 3.0f));
		for (int k = 0; k < 2; ++k) {
			const djb::vec3 i = g.dir(), o = g.dir();
			show("ggx(user F).eval", gx.eval(i, o, &pr)); show("  beckmann.evalp", bk.evalp(i, o, &pr));
			djb::vec3 wi; float pdf;
			show("  ggx.evalp_is", gx.evalp_is(g.u(), g.u(), o, &wi, &pdf, &pr)); show("    i", wi); fprintf(out, "    pdf"); put(pdf); fprintf(out, "\n");
		}
		djb::tabular tab(gx, 16 + g.below(40));
		float ag;
		djb::tabular::fit_ggx_parameters(tab).get_ellipse(&ag, NULL);
		fprintf(out, "  tabular(ggx(user F)) ggx"); put(ag); fprintf(out, "\n");
		show_table("  p22", tab.get_p22v());
		show("  fitted fresnel", tab.fresnel(g.u()));
	}
	// 4. a user-defined radial NDF
	{
		my_fresnel f(g);
		student st(g.in(1.6f, 5.0f), f, g.below(2) != 0);
		djb::microfacet::params pr = djb::microfacet::params::elliptic(g.log_in(0.1f, 1.0f), g.log_in(0.1f, 1.0f), g.in(0.0f, 3.0f));
		const djb::vec3 i = g.dir(), o = g.dir();
		show("student.eval", st.eval(i, o, &pr)); fprintf(out, "  pdf"); put(st.pdf(i, o, &pr)); fprintf(out, "\n");
		show("  sample", st.sample(g.u(), g.u(), o, &pr));
		djb::vec3 wi; float pdf;
		show("  evalp_is", st.evalp_is(g.u(), g.u(), o, &wi, &pdf, &pr)); show("    i", wi); fprintf(out, "    pdf"); put(pdf); fprintf(out, "\n");
		fprintf(out, "  ndf"); put(st.ndf(djb::normalize(i + o), pr)); fprintf(out, " sigma"); put(st.sigma(o, pr)); fprintf(out, " g1"); put(st.g1(djb::normalize(i + o), o, pr)); fprintf(out, "\n");
		djb::tabular tab(st, 12 + g.below(50));
		float ab, ag;
		djb::tabular::fit_beckmann_parameters(tab).get_ellipse(&ab, NULL);
		djb::tabular::fit_ggx_parameters(tab).get_ellipse(&ag, NULL);
		fprintf(out, "  tabular(student)"); put(ab); put(ag); fprintf(out, "\n");
		show_table("  qf", tab.get_qfv());
	}
	// 5. a user-defined lobe with its own sample() / pdf(): the inherited evalp_is and the batch overloads must go through them
	{
		sampled_lobe sl(g);
		const djb::vec3 o = g.dir();
		for (int k = 0; k < 3; ++k) {
			const float u1 = g.u(), u2 = g.u();
			djb::vec3 wi; float pdf;
			show("sampled_lobe.evalp_is", sl.evalp_is(u1, u2, o, &wi, &pdf)); show("  i", wi); fprintf(out, "  pdf"); put(pdf); put(sl.pdf(wi, o)); fprintf(out, "\n");
		}
		const djb::brdf &base = sl;
		show("  via brdf&", base.sample(g.u(), g.u(), o)); show("  evalp_hd", base.evalp_hd(djb::normalize(g.dir() + o), o));
		djb::tabular tab(sl, 10 + g.below(20));
		float ag; djb::tabular::fit_ggx_parameters(tab).get_ellipse(&ag, NULL);
		fprintf(out, "  tabular(sampled_lobe)"); put(ag); fprintf(out, "\n");
	}
	// 7 (placed before 6 for no reason but history). a microfacet-level NDF with its own qf2 / qf3
	{
		my_fresnel f(g);
		smith_user su(g.log_in(0.1f, 1.0f), f);
		djb::microfacet::params pr = djb::microfacet::params::elliptic(g.log_in(0.1f, 1.0f), g.log_in(0.1f, 1.0f), g.in(0.0f, 3.0f));
		const djb::vec3 i = g.dir(), o = g.dir();
		show("smith_user.eval", su.eval(i, o, &pr)); fprintf(out, "  pdf"); put(su.pdf(i, o, &pr)); fprintf(out, " qf2"); put(su.qf2(g.u(), o)); fprintf(out, "\n");
		show("  sample", su.sample(g.u(), g.u(), o, &pr));
		djb::vec3 wi; float pdf;
		show("  evalp_is", su.evalp_is(g.u(), g.u(), o, &wi, &pdf, &pr)); show("    i", wi); fprintf(out, "    pdf"); put(pdf); fprintf(out, "\n");
		const djb::microfacet &base = su;
		fprintf(out, "  vndf"); put(base.vndf(djb::normalize(i + o), o, pr)); put(base.vp22(g.in(-1.0f, 1.0f), g.in(-1.0f, 1.0f), o, pr)); put(base.gaf(djb::normalize(i + o), i, o, pr)); fprintf(out, "\n");
	}
	// 6. exceptions out of user code
	{
		throwing_ndf t(g.log_in(0.05f, 2.0f), g.below(2) != 0);
		for (int k = 0; k < 4; ++k) {
			const djb::vec3 i = g.dir(), o = g.dir();
			try { show("throwing_ndf.eval", t.eval(i, o)); }
			catch (const std::out_of_range &e) { fprintf(out, "throwing_ndf.eval: out_of_range: %s\n", e.what()); }
			try { show("  sample", t.sample(g.u(), g.u(), o)); }
			catch (const djb::exc &e) { fprintf(out, "  sample: djb::exc: %s\n", e.what()); }
			catch (const std::exception &e) { fprintf(out, "  sample: %s\n", e.what()); }
		}
		try { djb::tabular tab(t, 8 + g.below(16)); fprintf(out, "  a fit went through\n"); }
		catch (const std::out_of_range &e) { fprintf(out, "  fit: out_of_range: %s\n", e.what()); }
	}
}

} // namespace

int main(int argc, char **argv)
{
	const unsigned first = argc > 1 ? (unsigned)atoi(argv[1]) : 1u, count = argc > 2 ? (unsigned)atoi(argv[2]) : 4u;
	const int threads = argc > 3 && !strncmp(argv[3], "threads=", 8) ? atoi(argv[3] + 8) : 0;
	if (threads <= 0) {
		for (unsigned s = first; s < first + count; ++s) one_seed(s);
		return 0;
	}
	// threads=N: the seeds are dealt to N host threads that share the process's default context and run CONCURRENTLY (user-defined
	// objects being fitted from several threads at once); the output is printed in seed order and must be the sequential run's
	std::vector<std::string> text(count);
	std::vector<std::thread> pool;
	for (int t = 0; t < threads; ++t)
		pool.push_back(std::thread([&, t]() {
			for (unsigned k = (unsigned)t; k < count; k += (unsigned)threads) {
				char *buf = NULL; size_t len = 0;
				FILE *f = open_memstream(&buf, &len);
				out = f; one_seed(first + k); fclose(f); out = stdout;
				text[k].assign(buf, len); free(buf);
			}
		}));
	for (size_t t = 0; t < pool.size(); ++t) pool[t].join();
	for (unsigned k = 0; k < count; ++k) fputs(text[k].c_str(), stdout);
	return 0;
}
