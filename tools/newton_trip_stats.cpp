// tools/newton_trip_stats.cpp -- trip count of beckmann::qf2_radial's Newton loop (dj_brdf.h:1897-1952) on the bench distribution of configs[3]:
// its histogram, the per-wave maximum a 64-lane wave pays, how much of it a (u, cos_k) table predicts, and what sorting the 256 samples of a
// workgroup by predicted / true trip count would buy (profiles/r03/NOTES.md 4.4).  Host code: g++ -O2 -ffp-contract=off -o tools/bin/newton_trip_stats tools/newton_trip_stats.cpp
#define DJB_HOST_MATH 1
#include "../dj_brdf_amd/csrc/djb_device.hpp"
#include <cstdio>
#include <cmath>
namespace djbhostlibm { int use_restated = 0; double r_exp(double x){return exp(x);} double r_pow(double x,double y){return pow(x,y);} double r_atan2(double y,double x){return atan2(y,x);} double r_sin(double x){return sin(x);} double r_cos(double x){return cos(x);} double r_tan(double x){return tan(x);} double r_acos(double x){return acos(x);} float r_logf(float x){return logf(x);} float r_expf(float x){return expf(x);} float r_powf(float x,float y){return powf(x,y);} }
using namespace djbdev;
static int trips(float u, float cos_k, float sin_k)
{
	GlibcTabs gt = glibc_tabs_global();
	const float sqrt_pi_inv = F(1. / sqrt(DJB_PI));
	float cot_k = cos_k / sin_k, tan_k = sin_k / cos_k;
	const double e_cot = glibc_exp(D(-cot_k * cot_k), gt.exp64);
	float a = -1, c = erf_given_exp(cot_k, e_cot);
	u = fmax_(u, 1e-6f);
	float fit = 1 + cos_k * (-0.876f + cos_k * (0.4265f - 0.0594f * cos_k));
	float b = c - (1 + c) * glibc_powf(1 - u, fit, gt);
	float normalization = recip_to_f32(D(1 + c) + D(sqrt_pi_inv * tan_k) * e_cot);
	int it = 0;
	while (++it < 10) {
		if (!(b >= a && b <= c)) b = 0.5f * (a + c);
		float inv_erf = erfinv_(b, gt);
		float value = normalization * (1 + b + sqrt_pi_inv * tan_k * glibc_expf(-inv_erf * inv_erf, gt)) - u;
		float derivative = normalization * (1 - inv_erf * tan_k);
		if (fabsf(value) < 1e-5f) break;
		if (value > 0) c = b; else a = b;
		b -= value / derivative;
	}
	return it;
}
int main()
{
	const float ax_e = 0.2f, ay_e = 0.5f, phi = 0.7f;
	// params::elliptic(0.2, 0.5, 0.7) -> ax, ay, rho (dj_brdf.h:1453-1463), computed here in double: good enough for a histogram
	double c = cos(phi), s = sin(phi), a1 = ax_e * ax_e, a2 = ay_e * ay_e;
	double sxx = a1 * c * c + a2 * s * s, syy = a1 * s * s + a2 * c * c, sxy = (a1 - a2) * c * s;
	float ax = (float)sqrt(sxx), ay = (float)sqrt(syy), rho = (float)(sxy / sqrt(sxx * syy)), sq = (float)sqrt(1 - (double)rho * rho);
	const int N = 1 << 22;
	long hist[12] = {0}; double sum = 0;
	static int tr[1 << 22];
	static float us[1 << 22], cs[1 << 22];
	for (int k = 0; k < N; ++k) {
		v3 o = gen_direction(0xD1B00002u, (uint64_t)k);
		float u1 = gen_uniform(0xD1B00003u, (uint64_t)k);
		u1 = sat_(u1) * 0.99998f + 0.00001f;
		float a = o.x * ax + o.y * ay * rho, bb = o.y * ay * sq, cc = o.z;
		v3 os = normalize(mk(a, bb, cc));
		float cos_k = os.z, sin_k = D(os.z) < 1.0 ? F(sqrt(1.0 - D(os.z * os.z))) : 0.0f;
		int t = trips(u1, cos_k, sin_k);
		tr[k] = t; us[k] = u1; cs[k] = cos_k; hist[t]++; sum += t;
	}
	printf("mean trips %.3f\n", sum / N);
	for (int t = 1; t < 11; ++t) printf("  trips %2d: %.4f\n", t, (double)hist[t] / N);
	// per-wave max over consecutive 64
	double wsum = 0; for (int w = 0; w < N / 64; ++w) { int m = 0; for (int j = 0; j < 64; ++j) m = tr[w * 64 + j] > m ? tr[w * 64 + j] : m; wsum += m; }
	printf("mean per-wave max (64 consecutive samples) %.3f\n", wsum / (N / 64));
	// predictability: 2-D table over (u bucket 32, cos_k bucket 16): mean and the residual spread
	static double acc[32][16], acc2[32][16]; static long cnt[32][16];
	for (int k = 0; k < N; ++k) { int iu = (int)(us[k] * 32); if (iu > 31) iu = 31; int ic = (int)(cs[k] * 16); if (ic > 15) ic = 15; if (ic < 0) ic = 0; acc[iu][ic] += tr[k]; acc2[iu][ic] += (double)tr[k] * tr[k]; cnt[iu][ic]++; }
	double within = 0; for (int i = 0; i < 32; ++i) for (int j = 0; j < 16; ++j) if (cnt[i][j]) { double m = acc[i][j] / cnt[i][j]; within += acc2[i][j] - cnt[i][j] * m * m; }
	double tot = 0, mean = sum / N; for (int k = 0; k < N; ++k) tot += (tr[k] - mean) * (tr[k] - mean);
	printf("variance of the trip count: total %.3f, within (u, cos_k) cells %.3f -> %.1f %% explained by a 32 x 16 table\n", tot / N, within / N, 100 * (1 - within / tot));
	// if a workgroup of 256 sorted its samples by the table's prediction into 4 waves: mean per-wave max
	double ssum = 0; for (int g = 0; g < N / 256; ++g) {
		int idx[256]; double pred[256];
		for (int j = 0; j < 256; ++j) { int k = g * 256 + j; int iu = (int)(us[k] * 32); if (iu > 31) iu = 31; int ic = (int)(cs[k] * 16); if (ic > 15) ic = 15; if (ic < 0) ic = 0; pred[j] = acc[iu][ic] / cnt[iu][ic]; idx[j] = j; }
		for (int a = 1; a < 256; ++a) { int v = idx[a]; int b = a - 1; while (b >= 0 && pred[idx[b]] > pred[v]) { idx[b + 1] = idx[b]; --b; } idx[b + 1] = v; }
		for (int w = 0; w < 4; ++w) { int m = 0; for (int j = 0; j < 64; ++j) { int t = tr[g * 256 + idx[w * 64 + j]]; m = t > m ? t : m; } ssum += m; }
	}
	printf("mean per-wave max after sorting each 256-sample workgroup by the predicted trips: %.3f\n", ssum / (N / 64));
	// oracle sort (by the true trip count): the bound of any reordering within a workgroup
	double osum = 0; for (int g = 0; g < N / 256; ++g) { int c[12] = {0}; for (int j = 0; j < 256; ++j) c[tr[g * 256 + j]]++; int pos = 0, t = 1; for (int w = 0; w < 4; ++w) { int need = 64, m = 0; while (need > 0) { while (c[t] == 0) ++t; int take = c[t] < need ? c[t] : need; c[t] -= take; need -= take; m = t; } osum += m; (void)pos; } }
	printf("... by the true trips (bound): %.3f\n", osum / (N / 64));
	return 0;
}
