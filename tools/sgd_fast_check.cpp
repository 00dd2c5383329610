/* tools/sgd_fast_check.cpp -- the evidence behind csrc/djb_fast_models.inc (the decided fast tier of sgd's g1 / ndf), on any host with
 * glibc 2.35: the SAME source as the device code (djb_device.hpp in its host-restated instantiation) against
 *   (1) __float128 values: the error of flog and fexp in units of U = 2^-53 (what the bounds assume);
 *   (2) the reference's own expressions on the host's libm (sgd__g1, sgd__ndf, dj_brdf.h:3416-3432) over the 100 published rows and
 *       random rows: |fast double - reference double| / bound (must stay below 1; the bounds carry a factor 2), decided values that
 *       differ from the reference's float (must be 0), share of undecided values.
 *     g++ -O2 -std=c++17 -mfma -ffp-contract=off -fopenmp -I dj_brdf_amd/csrc tools/sgd_fast_check.cpp -o /tmp/sgd_fast_check -lquadmath
 *     /tmp/sgd_fast_check dj_brdf_amd/data/sgd_params.csv [samples per row, default 2e6] [abc table] [flog / fexp arguments, default 4e8]
 */
#define DJB_HOST_MATH 1
#define DJB_HOST_RESTATED 1
#include "djb_device.hpp"
#include <quadmath.h>
#include <stdio.h>
#include <stdlib.h>
#include <string>
#include <vector>
#include <omp.h>

using namespace djbdev;

static inline uint64_t mix64(uint64_t x) { x += 0x9e3779b97f4a7c15ull; x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull; x = (x ^ (x >> 27)) * 0x94d049bb133111ebull; return x ^ (x >> 31); }
static inline double u01(uint64_t &s) { s = mix64(s); return (double)(s >> 11) * 0x1p-53; }

// the reference's expressions, operation for operation (dj_brdf.h:3416-3432), before the rounding to float
static double ref_g1(double theta_k, double theta0, double c, double k_, double lambda)
{
	double tmp1 = fmax(0.0, theta_k - theta0);       // djb::max on doubles without NaNs
	double tmp2 = 1.0 - exp(c * pow(tmp1, k_));
	double tmp3 = 1.0 + lambda * tmp2;
	return tmp3;
}
static double ref_ndf(double cos_theta_h, double alpha, double p, double kap)
{
	const double inv_pi = 1.0 / M_PI;
	double c2 = cos_theta_h * cos_theta_h;
	double t2 = (1.0 - c2) / c2;
	double ax = alpha + t2 / alpha;
	return ((kap * exp(-ax) * inv_pi) / (pow(ax, p) * c2 * c2));
}

int main(int argc, char **argv)
{
	const char *csv = argc > 1 ? argv[1] : "dj_brdf_amd/data/sgd_params.csv";
	const long long per_row = argc > 2 ? atoll(argv[2]) : 2000000;
	const long long n_unit = argc > 4 ? atoll(argv[4]) : 400000000ll;       // arguments of the flog / fexp part
	// ---- (1) flog / fexp against __float128
	{
		double wl_rel = 0, wl_abs = 0, we = 0;
#pragma omp parallel
		{
			double l_rel = 0, l_abs = 0, e_ = 0;
			uint64_t s = 1234567ull * (omp_get_thread_num() + 1);
#pragma omp for
			for (long long n = 0; n < n_unit; ++n) {
				const int fam = (int)(n & 3);
				double x;
				if (fam == 0) x = exp2(2000.0 * u01(s) - 1000.0);              // any magnitude
				else if (fam == 1) x = 0.5 + 1.5 * u01(s);                     // around 1
				else if (fam == 2) x = 1.0 + (u01(s) - 0.5) * 0x1p-6;          // next to 1 (cancellation of k ln2 + ln c against r)
				else x = exp2(40.0 * u01(s) - 30.0);                           // what t1 and ax are
				const double L = flog(x, 0u);
				const __float128 t = logq((__float128)x);
				const double err = (double)fabsq((__float128)L - t);
				const double aL = fabs((double)t);
				// in units of U |L| + 2^-62
				const double r1 = err / (SGD_U * aL + 0x1p-62);
				if (r1 > l_rel) l_rel = r1;
				if (err > l_abs && aL < 0.01) l_abs = err;
				const double y = (fam == 1 ? 1400.0 * u01(s) - 700.0 : fam == 2 ? 2.0 * u01(s) - 1.0 : fam == 3 ? exp2(-60.0 * u01(s)) * (u01(s) < 0.5 ? -1 : 1) : 80.0 * u01(s) - 40.0);
				const double Y = fexp(y, 0u);
				const __float128 ty = expq((__float128)y);
				const double er = (double)(fabsq((__float128)Y - ty) / ty) / SGD_U;
				if (er > e_) e_ = er;
			}
#pragma omp critical
			{ if (l_rel > wl_rel) wl_rel = l_rel; if (l_abs > wl_abs) wl_abs = l_abs; if (e_ > we) we = e_; }
		}
		printf("flog: worst |err| / (U |ln x| + 2^-62) = %.3f   (bound used: 4 U |L| + 2^-57; worst abs err for |ln x| < 0.01: %.3g = 2^%.1f)\n", wl_rel, wl_abs, log2(wl_abs));
		printf("fexp: worst relative err = %.3f U   (bound used: 4 U = 2^-51; with glibc's own exp 3 U for the pair)\n", we);
	}
	// ---- (2) rows
	std::vector<std::vector<double>> rows;
	std::vector<std::string> names;
	FILE *f = fopen(csv, "r");
	if (!f) { fprintf(stderr, "cannot open %s\n", csv); return 2; }
	char line[4096];
	fgets(line, sizeof line, f);
	while (fgets(line, sizeof line, f)) {
		std::vector<double> v; std::string nm;
		int col = 0; char *tok = line;                       // empty fields stay fields (strtok would skip them)
		while (tok) {
			char *end = strchr(tok, ',');
			if (end) *end = 0;
			if (col == 0) nm = tok; else if (col >= 2 && col < 35) v.push_back(atof(tok));
			++col; tok = end ? end + 1 : NULL;
		}
		if (v.size() == 33) { rows.push_back(v); names.push_back(nm); }
	}
	fclose(f);
	const size_t published = rows.size();
	// random rows: every parameter log-uniform over (and beyond) the published span
	uint64_t rs = 42;
	for (int r = 0; r < 200; ++r) {
		std::vector<double> v(33, 0.1);
		for (int ch = 0; ch < 3; ++ch) {
			v[6 + ch] = exp(log(1e-6) + u01(rs) * (log(2.0) - log(1e-6)));        // alpha
			v[9 + ch] = u01(rs) < 0.1 ? 1e-14 : 3.0 * u01(rs);                   // p
			v[18 + ch] = exp(log(0.5) + u01(rs) * (log(1e5) - log(0.5)));        // kap
			v[21 + ch] = exp(log(1e-8) + u01(rs) * (log(1e8) - log(1e-8)));      // lambda
			v[24 + ch] = exp(log(1e-9) + u01(rs) * (log(1e38) - log(1e-9)));     // c
			v[27 + ch] = exp(log(1.0) + u01(rs) * (log(900.0) - log(1.0)));      // k
			v[30 + ch] = -0.7 + 2.0 * u01(rs);                                   // theta0
		}
		rows.push_back(v); names.push_back("random");
	}
	long long tot_g = 0, und_g = 0, bad_g = 0, tot_n = 0, und_n = 0, bad_n = 0, rows_out = 0, pub_g = 0, pub_ug = 0, pub_n = 0, pub_un = 0;
	double worst_g = 0, worst_n = 0, worst_und_g = 0, worst_und_n = 0; std::string at_g, at_n;
	for (size_t r = 0; r < rows.size(); ++r) {
		double m[SGD_FAST_ROW];
		if (!sgd_fast_row(rows[r].data(), m)) { ++rows_out; printf("row %zu (%s): outside the fast tier's domain\n", r, names[r].c_str()); continue; }
		long long tg = 0, ug = 0, bg = 0, tn = 0, un = 0, bn = 0, ur = 0, tr = 0; double wg = 0, wn = 0;
#pragma omp parallel reduction(+ : tg, ug, bg, tn, un, bn, ur, tr) reduction(max : wg, wn)
		{
			uint64_t s = mix64(r * 1000003ull + omp_get_thread_num());
#pragma omp for
			for (long long n = 0; n < per_row; ++n) {
				// a polar cosine the way the kernels see it: a float in (0, 1]; one family hugging the wall theta_k = theta0, one grazing
				const int fam = (int)(n % 5);
				const int ch = (int)((n / 5) % 3);
				float kz;
				if (fam == 3) { const double th = m[30 + ch] + (u01(s) - 0.3) * exp2(-30.0 * u01(s)); kz = (float)cos(th); }
				else if (fam == 4) kz = (float)(u01(s) * 0.05);
				else kz = (float)u01(s);
				if (!(kz > 0.0f)) kz = 1e-3f;
				if (kz > 1.0f) kz = 1.0f;
				const double theta_k = acos((double)kz);
				const double t1 = dmax_(0.0, theta_k - m[30 + ch]);
				bool dec; double dbg[3];
				(void)t1;
				// odd samples: theta_k as the kernels have it -- anywhere within 2^-48 (relative) of glibc's -- with the caller's bound on that
				const bool approx = (n & 1) != 0;
				const double th_f = theta_k * (1.0 + (2.0 * u01(s) - 1.0) * 0x1p-48), dth = 1.01 * 0x1p-48 * th_f;
				const float got = approx ? sgd_g1_fast<true>(m, ch, th_f - m[30 + ch], dth, 0u, 0u, dec, dbg)
				                         : sgd_g1_fast<false>(m, ch, theta_k - m[30 + ch], 0.0, 0u, 0u, dec, dbg);
				const double t3r = ref_g1(theta_k, m[30 + ch], m[24 + ch], m[27 + ch], m[21 + ch]);
				const float want = (float)fmin(1.0, fmax(0.0, t3r));
				++tg; if (fam < 3) ++tr;
				if (!dec) { ++ug; if (fam < 3) ++ur; }
				else {
					if (memcmp(&got, &want, 4)) { if (bg < 5) printf("  g1 MISMATCH row %zu ch %d kz %.9g: got %.9g want %.9g t3 %.17g ref %.17g B %.3g\n", r, ch, kz, got, want, dbg[0], t3r, dbg[1]); ++bg; }
					if (t1 > 0.0 && dbg[2] == 0.0 && got != 1.0f) { const double q = fabs(dbg[0] - t3r) / dbg[1]; if (q > wg) wg = q; }
				}
				// ndf: h.z a float in (0, 1]
				float hz = fam == 4 ? (float)(1.0 - u01(s) * exp2(-20.0 * u01(s))) : (float)sqrt(u01(s));
				if (!(hz > 1e-4f)) hz = 1e-4f;
				const double chd = (double)hz, c2 = chd * chd;
				const double rc = recip_fast_m(c2), t2 = (1.0 - c2) * rc;
				const float gotn = sgd_ndf_fast(m, ch, c2, t2, rc * rc, 0u, 0u, dec, dbg);
				const double vr = ref_ndf(chd, m[6 + ch], m[9 + ch], m[18 + ch]);
				const float wantn = (float)vr;
				++tn;
				if (!dec) ++un;
				else {
					if (memcmp(&gotn, &wantn, 4)) { if (bn < 5) printf("  ndf MISMATCH row %zu ch %d hz %.9g: got %.9g want %.9g V %.17g ref %.17g b %.3g\n", r, ch, hz, gotn, wantn, dbg[0], vr, dbg[1]); ++bn; }
					if (vr > 1e-290 && vr < 1e290) { const double q = fabs(dbg[0] / vr - 1.0) / dbg[1]; if (q > wn) wn = q; }
				}
			}
		}
		tot_g += tg; und_g += ug; bad_g += bg; tot_n += tn; und_n += un; bad_n += bn;
		if (r < published) { pub_g += tr; pub_ug += ur; pub_n += tn; pub_un += un; }
		if (wg > worst_g) { worst_g = wg; at_g = names[r]; }
		if (wn > worst_n) { worst_n = wn; at_n = names[r]; }
		const double sg = (double)ur / (tr ? tr : 1), sn = (double)un / tn;      // g1: over the uniformly drawn polar cosines (the wall-hugging family is undecided by design)
		if (sg > worst_und_g) worst_und_g = sg;
		if (sn > worst_und_n) worst_und_n = sn;
		if (sg > 1e-3 || sn > 1e-3) printf("row %zu (%s): undecided g1 %.2e ndf %.2e\n", r, names[r].c_str(), sg, sn);
	}
	printf("rows: %zu published + %zu random, %lld outside the domain; %lld values per row and term\n", published, rows.size() - published, rows_out, per_row);
	printf("g1 : %lld values, %lld undecided (%.3g; worst row %.3g), %lld decided-but-different (must be 0), worst |fast - ref| / bound %.4f (%s)\n",
	       tot_g, und_g, (double)und_g / tot_g, worst_und_g, bad_g, worst_g, at_g.c_str());
	printf("ndf: %lld values, %lld undecided (%.3g; worst row %.3g), %lld decided-but-different (must be 0), worst |fast / ref - 1| / bound %.4f (%s)\n",
	       tot_n, und_n, (double)und_n / tot_n, worst_und_n, bad_n, worst_n, at_n.c_str());
	printf("published rows alone: g1 undecided %.3g, ndf undecided %.3g\n", (double)pub_ug / (pub_g ? pub_g : 1), (double)pub_un / (pub_n ? pub_n : 1));
	// ---- (3) abc__ndf (dj_brdf.h:3608-3613) over the published rows
	long long tot_a = 0, und_a = 0, bad_a = 0; double worst_a = 0;
	{
		FILE *fa = fopen(argc > 3 ? argv[3] : "dj_brdf_amd/data/abc_params.csv", "r");
		if (!fa) { fprintf(stderr, "cannot open the abc table\n"); return 2; }
		std::vector<std::vector<double>> arows;
		if (!fgets(line, sizeof line, fa)) return 2;
		while (fgets(line, sizeof line, fa)) {
			std::vector<double> v; int col = 0; char *tok = line;
			while (tok) { char *end = strchr(tok, ','); if (end) *end = 0; if (col >= 1 && col < 10) v.push_back(atof(tok)); ++col; tok = end ? end + 1 : NULL; }
			if (v.size() == 9) arows.push_back(v);
		}
		fclose(fa);
		for (int r = 0; r < 100; ++r) { std::vector<double> v(9, 0.1); for (int c = 3; c < 6; ++c) v[c] = exp(log(1e-3) + u01(rs) * log(1e7)); v[6] = exp(log(1.0) + u01(rs) * log(1e7)); v[7] = 0.05 + 4.0 * u01(rs); arows.push_back(v); }
		for (size_t r = 0; r < arows.size(); ++r) {
			const double *m = arows[r].data();
			long long ta = 0, ua = 0, ba = 0; double wa = 0;
#pragma omp parallel reduction(+ : ta, ua, ba) reduction(max : wa)
			{
				uint64_t s = mix64(r * 7919ull + omp_get_thread_num());
#pragma omp for
				for (long long n = 0; n < per_row; ++n) {
					float hz = (n & 3) == 3 ? (float)(1.0 - u01(s) * exp2(-24.0 * u01(s))) : (float)sqrt(u01(s));
					if (!(hz > 0.0f)) hz = 1e-3f;
					const double tmp = 1.0 - (double)hz, w = 1.0 + m[6] * tmp;
					float v[3]; bool dec[3]; double dbg[6];
					abc_ndf_fast(m, w, 0u, 0u, v, dec, dbg);
					const double den = pow(w, m[7]);
					for (int c = 0; c < 3; ++c) {
						const double vr = m[3 + c] / den; const float want = (float)vr;
						++ta;
						if (!dec[c]) { ++ua; continue; }
						if (memcmp(&v[c], &want, 4)) { if (ba < 5) printf("  abc MISMATCH row %zu hz %.9g: got %.9g want %.9g\n", r, hz, v[c], want); ++ba; }
						const double q = fabs(dbg[2 * c] / vr - 1.0) / dbg[2 * c + 1]; if (q > wa) wa = q;
					}
				}
			}
			tot_a += ta; und_a += ua; bad_a += ba; if (wa > worst_a) worst_a = wa;
		}
		printf("abc: %zu rows, %lld values, %lld undecided (%.3g), %lld decided-but-different (must be 0), worst |fast / ref - 1| / bound %.4f\n", arows.size(), tot_a, und_a, (double)und_a / tot_a, bad_a, worst_a);
	}
	if (bad_a || worst_a >= 1.0) return 1;
	return (bad_g || bad_n || worst_g >= 1.0 || worst_n >= 1.0) ? 1 : 0;
}
