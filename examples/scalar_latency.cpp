// examples/scalar_latency.cpp -- what one scalar call of the djb:: surface costs (the reference's real callers:
// one (i, o) pair per virtual call, from many render threads).  Prints ns per call for a few operators on the
// process default context (a GPU object answers scalar-size host calls from its host twin, include/djb_hip.h
// DJB_SCALAR_HOST_MAX) and the aggregate rate of T threads sharing ONE object; exits 0 iff every single-thread
// figure is below 1 us.      usage: scalar_latency [threads]
// Written against the reference's interface only: oracle/Makefile builds the same source on the real reference
// (oracle/_ref/scalar_latency), so the two can be timed side by side on one machine (tools/exp/r05/host_path_o3.sh).
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define DJ_BRDF_IMPLEMENTATION 1
#include "dj_brdf.h"

int main(int argc, char **argv)
{
	using clk = std::chrono::steady_clock;
	const int T = argc > 1 ? atoi(argv[1]) : 8, N = 400000;
	try {
		float iz = sqrtf(1.f - 0.3f * 0.3f - 0.2f * 0.2f), oz = sqrtf(1.f - 0.4f * 0.4f - 0.1f * 0.1f);
		djb::vec3 o(-0.4f, 0.1f, oz);
		djb::ggx ggx; djb::beckmann beck;
		djb::microfacet::params iso = djb::microfacet::params::isotropic(0.3f);
		djb::tabular tab(ggx, 64);
		double worst = 0.0, sink = 0.0;
		auto run = [&](const char *what, auto &&call) {
			call(0);                                       // builds the host twin on first use
			auto t0 = clk::now();
			for (int k = 1; k <= N; ++k) sink += call(k);
			double ns = std::chrono::duration<double, std::nano>(clk::now() - t0).count() / N;
			printf("%-28s %8.0f ns per call\n", what, ns);
			if (ns > worst) worst = ns;
		};
		// both directions move with k: nothing of a pair is loop-invariant for a compiler that inlines the reference's header
		run("ggx.eval(i, o, &params)", [&](int k) { djb::vec3 i(0.3f + 1e-7f * k, 0.2f, iz), ok(-0.4f + 1e-7f * k, 0.1f, oz); return ggx.eval(i, ok, &iso).x; });
		run("ggx.pdf(i, o)", [&](int k) { djb::vec3 i(0.3f + 1e-7f * k, 0.2f, iz), ok(-0.4f + 1e-7f * k, 0.1f, oz); return ggx.pdf(i, ok); });
		run("beckmann.sample(u1, u2, o)", [&](int k) { djb::vec3 ok(-0.4f + 1e-7f * k, 0.1f, oz); return beck.sample(0.25f + 1e-7f * k, 0.75f, ok, &iso).x; });
		run("tabular.evalp(i, o)", [&](int k) { djb::vec3 i(0.3f + 1e-7f * k, 0.2f, iz), ok(-0.4f + 1e-7f * k, 0.1f, oz); return tab.evalp(i, ok).x; });
		// what BSDF::sample() of the plugins issues: direction, weight and pdf in one call
		run("beckmann.evalp_is(u1, u2, o)", [&](int k) { djb::vec3 ok(-0.4f + 1e-7f * k, 0.1f, oz), i; djb::float_t pdf; return beck.evalp_is(0.25f + 1e-7f * k, 0.75f, ok, &i, &pdf, &iso).x + pdf; });
		run("tabular.evalp_is(u1, u2, o)", [&](int k) { djb::vec3 ok(-0.4f + 1e-7f * k, 0.1f, oz), i; djb::float_t pdf; return tab.evalp_is(0.25f + 1e-7f * k, 0.75f, ok, &i, &pdf).x + pdf; });
		// the same call through a base-class pointer the compiler cannot see through (how a renderer holds its BSDFs), first with
		// independent calls (the core overlaps consecutive ones where it can), then with each call's input depending on the previous
		// result (the latency of ONE call: what a path tracer's dependent chain of hits sees)
		{
			djb::brdf *bp = &ggx; asm volatile("" : "+r"(bp));
			float prev = 0.0f;
			run("brdf*->eval, independent", [&](int k) { djb::vec3 i(0.3f + 1e-7f * k, 0.2f, iz), ok(-0.4f + 1e-7f * k, 0.1f, oz); return bp->eval(i, ok, &iso).x; });
			run("brdf*->eval, dependent", [&](int k) { djb::vec3 i(0.3f + 1e-7f * k + 1e-30f * prev, 0.2f, iz), ok(-0.4f + 1e-7f * k, 0.1f, oz); prev = bp->eval(i, ok, &iso).x; return prev; });
		}
		// render threads sharing one BSDF: no lock on the scalar path
		std::vector<std::thread> th;
		auto t0 = clk::now();
		for (int t = 0; t < T; ++t)
			th.emplace_back([&, t] { double s = 0; for (int k = 0; k < N; ++k) { djb::vec3 i(0.3f + 1e-7f * k, 0.2f + 1e-3f * t, iz); s += ggx.eval(i, o, &iso).x; } (void)s; });
		for (auto &x : th) x.join();
		double sec = std::chrono::duration<double>(clk::now() - t0).count();
		printf("%d threads on one ggx object: %.2f M calls/s in total (sink %g)\n", T, T * (double)N / sec * 1e-6, sink);
		printf("worst single-thread figure: %.0f ns %s\n", worst, worst < 1000.0 ? "(< 1 us: ok)" : "(>= 1 us: FAIL)");
		return worst < 1000.0 ? 0 : 1;
	} catch (const djb::exc &e) {
		fprintf(stderr, "djb::exc: %s\n", e.what());
		return 2;
	}
}
