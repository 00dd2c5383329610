/* tools/sincos_fast_check.c -- the error bound behind the guarded sin / cos of the normal-map sampling scheme (csrc/djb_device.hpp:
 * sincos_fast, mul_sincos_to_f32): for EVERY float x in [-8, 8] the cheap fp64 sine and cosine (Cody-Waite reduction by pi/2 in two
 * fma steps + fdlibm's kernel polynomials) against the host libm's sin / cos of the same double -- largest difference in ulp64.  The
 * guard (near_f32_midpoint, 256 ulp64 either side of a float rounding boundary) needs it to stay below ~100.
 *     gcc -O2 -ffp-contract=off -fopenmp tools/sincos_fast_check.c -o /tmp/sincos_fast_check -lm && /tmp/sincos_fast_check
 * (any host; 2.2e9 arguments, a few minutes on 8 threads).  The arithmetic below is the device function's, operation for operation. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

static void sincos_fast(double x, double *s, double *c)
{
	const double kf = rint(x * 0x1.45f306dc9c883p-1);              /* 2 / pi */
	double r = fma(-kf, 0x1.921fb54442d18p+0, x);                   /* pi / 2, high 53 bits */
	r = fma(-kf, 0x1.1a62633145c07p-54, r);                         /* ... and the next 53 */
	const double z = r * r;
	double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
	ps = fma(z, ps, 2.75573137070700676789e-06);
	ps = fma(z, ps, -1.98412698298579493134e-04);
	ps = fma(z, ps, 8.33333333332248946124e-03);
	ps = fma(z, ps, -1.66666666666666324348e-01);
	const double sr = fma(r * z, ps, r);
	double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
	pc = fma(z, pc, -2.75573143513906633035e-07);
	pc = fma(z, pc, 2.48015872894767294178e-05);
	pc = fma(z, pc, -1.38888888888741095749e-03);
	pc = fma(z, pc, 4.16666666666666019037e-02);
	const double cr = fma(z * z, pc, fma(z, -0.5, 1.0));
	const int q = (int)kf & 3;
	const double a = (q & 1) ? cr : sr, b = (q & 1) ? sr : cr;      /* sin takes a, cos takes b */
	*s = (q & 2) ? -a : a;
	*c = ((q + 1) & 2) ? -b : b;
}

static double ulps(double got, double want)
{
	if (got == want) return 0.0;
	int e; frexp(want, &e);
	return fabs(got - want) / ldexp(1.0, e - 53);
}

int main(void)
{
	double worst_s = 0, worst_c = 0; float at_s = 0, at_c = 0;
	const uint32_t top = 0x41000000u;                                /* 8.0f */
#pragma omp parallel
	{
		double ws = 0, wc = 0; float as = 0, ac = 0;
#pragma omp for schedule(static, 1 << 20)
		for (int64_t bits = 0; bits <= (int64_t)top; ++bits) {
			for (int neg = 0; neg < 2; ++neg) {
				uint32_t u = (uint32_t)bits | (neg ? 0x80000000u : 0u);
				float xf; memcpy(&xf, &u, 4);
				double s, c; sincos_fast((double)xf, &s, &c);
				const double es = ulps(s, sin((double)xf)), ec = ulps(c, cos((double)xf));
				if (es > ws) { ws = es; as = xf; }
				if (ec > wc) { wc = ec; ac = xf; }
			}
		}
#pragma omp critical
		{ if (ws > worst_s) { worst_s = ws; at_s = as; } if (wc > worst_c) { worst_c = wc; at_c = ac; } }
	}
	printf("sincos_fast vs libm over every float in [-8, 8] (%.3g arguments): max |sin| difference %.2f ulp64 at x = %a, max |cos| difference %.2f ulp64 at x = %a\n",
	       2.0 * (double)top, worst_s, at_s, worst_c, at_c);
	return worst_s < 64 && worst_c < 64 ? 0 : 1;
}
