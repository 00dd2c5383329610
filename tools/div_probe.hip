// div_probe.hip -- exact fp32 division through fp64 on gfx950: cost and exhaustive-style verification.
//
// Claim (profiles/archive/r03_NOTES.md 4.1, "exact division by a shared or uniform denominator"): for floats a, b with a normal (or zero /
// inf / NaN) quotient,   float(double(a) * R) == a / b   whenever R is within 2^-52 (relative) of 1/b.  Reason: a/b can
// never lie within 2^-49 (relative) of the midpoint m of two adjacent floats -- a - m*b is a non-zero multiple of
// ulp(m)*ulp(b), i.e. |a/b - m| >= |a/b| / (M*B) > 2^-49 |a/b| for the integer significands M < 2^25, B < 2^24 -- and an
// exact tie is impossible (M is odd with 25 bits, so M*B has > 24 significant bits and cannot equal a).  The product
// double(a)*R carries <= 2^-52 + 2^-53 of error, so it rounds to the float a/b rounds to.  Subnormal quotients (coarser
// grid: ties exist) are excluded by a magnitude test and take the IEEE sequence.
//
// This program (run on the GPU box) measures the VALU cost of the variants and counts mismatches against the IEEE
// division over hash-generated operands (all exponents, edge mantissas).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/div_probe.hip -o /tmp/div_probe && /tmp/div_probe [n_check]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ inline unsigned int pcg(unsigned int v)
{
	unsigned int s = v * 747796405u + 2891336453u;
	unsigned int w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
	return (w >> 22u) ^ w;
}

// R ~ 1/b to <= 2^-52: v_rcp_f32 seed (1 ulp), two Newton steps in fp64 (2^-23 -> 2^-46 -> rounding level)
__device__ inline double recip_d_from_f32(float b)
{
	double bd = (double)b, r = (double)__builtin_amdgcn_rcpf(b);
	r = __builtin_fma(__builtin_fma(-bd, r, 1.0), r, r);
	r = __builtin_fma(__builtin_fma(-bd, r, 1.0), r, r);
	return r;
}
__device__ inline double recip_d_from_f64(float b)
{
	double bd = (double)b, r = __builtin_amdgcn_rcp(bd);
	r = __builtin_fma(__builtin_fma(-bd, r, 1.0), r, r);
	r = __builtin_fma(__builtin_fma(-bd, r, 1.0), r, r);
	return r;
}
__device__ inline float div_by_recip(float a, float b, double R)
{
	float q = (float)((double)a * R);
	// subnormal / underflowing quotient, or a seed that broke down (b subnormal, huge, inf, NaN): IEEE sequence
	if (__builtin_expect(!(__builtin_fabsf(q) >= 1.17549435e-38f) && a != 0.0f, 0)) q = a / b;
	return q;
}
__device__ inline bool seed_ok(float b) { float ab = __builtin_fabsf(b); return ab >= 1e-30f && ab <= 1e30f; }

template <int MODE>
__global__ __launch_bounds__(256) void k_cost(float *out, int iters, float b0, double R0)
{
	float a = 1.0f + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 1e-4f, acc = 0.0f, acc2 = 0.0f;
	for (int k = 0; k < iters; ++k) {
		float x = a + (float)k * 1e-6f, y = b + (float)k * 1e-7f, x2 = x * 1.5f;
		if (MODE == 0) acc += x / y;                                                    // IEEE
		if (MODE == 1) acc += div_by_recip(x, b0, R0);                                  // uniform denominator, R from the host
		if (MODE == 2) acc += div_by_recip(x, y, recip_d_from_f32(y));                  // per-lane, f32 seed
		if (MODE == 3) acc += div_by_recip(x, y, recip_d_from_f64(y));                  // per-lane, f64 seed
		if (MODE == 4) { acc += x / y; acc2 += x2 / y; }                                // two IEEE divisions, one denominator
		if (MODE == 5) { double R = recip_d_from_f32(y); acc += div_by_recip(x, y, R); acc2 += div_by_recip(x2, y, R); }
		if (MODE == 6) acc += x * y;                                                    // loop overhead
	}
	out[blockIdx.x * 256 + threadIdx.x] = acc + acc2;
}

__device__ inline float rand_float(unsigned int h0, unsigned int h1, int k)
{
	// every 4th: full random bit pattern; else random mantissa with an exponent in a chosen band; edge mantissas mixed in
	unsigned int m = h0 & 0x7fffffu;
	if ((h1 & 7u) == 1u) m = 0u; else if ((h1 & 7u) == 2u) m = 0x7fffffu; else if ((h1 & 7u) == 3u) m = 1u;
	int e;
	switch (k & 3) {
	case 0: return __uint_as_float(h0 ^ (h1 << 7));
	case 1: e = 127 + (int)(h1 >> 8) % 5 - 2; break;          // near 1
	case 2: e = 127 + (int)(h1 >> 8) % 80 - 40; break;        // +-2^40
	default: e = 1 + (int)(h1 >> 8) % 253; break;             // all normal exponents
	}
	return __uint_as_float(((h0 >> 31) << 31) | ((unsigned int)e << 23) | m);
}

__global__ __launch_bounds__(256) void k_check(long long n, unsigned int seed, unsigned long long *counters)
{
	unsigned long long bad_u = 0, bad_l = 0, bad_l64 = 0, fb = 0;
	long long stride = (long long)gridDim.x * 256;
	for (long long k = (long long)blockIdx.x * 256 + threadIdx.x; k < n; k += stride) {
		unsigned int h0 = pcg(seed ^ (unsigned int)k), h1 = pcg(h0 + (unsigned int)(k >> 32)), h2 = pcg(h1 ^ 0x9E3779B9u), h3 = pcg(h2 + 7u);
		float a = rand_float(h0, h1, (int)(k & 3)), b = rand_float(h2, h3, (int)((k >> 2) & 3));
		float want = a / b;
		// uniform-denominator form: R = correctly rounded 1/b (what the host passes)
		float q1 = div_by_recip(a, b, 1.0 / (double)b);
		float q2 = seed_ok(b) ? div_by_recip(a, b, recip_d_from_f32(b)) : a / b;
		float q3 = seed_ok(b) ? div_by_recip(a, b, recip_d_from_f64(b)) : a / b;
		bool nan_ok = (want != want);
		if (!(q1 == want || (nan_ok && q1 != q1))) ++bad_u;
		if (!(q2 == want || (nan_ok && q2 != q2))) ++bad_l;
		if (!(q3 == want || (nan_ok && q3 != q3))) ++bad_l64;
		float q = (float)((double)a * (1.0 / (double)b));
		if (!(__builtin_fabsf(q) >= 1.17549435e-38f) && a != 0.0f) ++fb;
	}
	atomicAdd(&counters[0], bad_u); atomicAdd(&counters[1], bad_l); atomicAdd(&counters[2], bad_l64); atomicAdd(&counters[3], fb);
}

int main(int argc, char **argv)
{
	long long n_check = argc > 1 ? atoll(argv[1]) : 20000000000LL;
	float *out; unsigned long long *cnt;
	hipMalloc((void **)&out, 256 * 4096 * sizeof(float)); hipMalloc((void **)&cnt, 64); hipMemset(cnt, 0, 64);
	const int iters = 4096;
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	const char *names[] = { "IEEE a/b", "uniform b, host R: cvt*mul*cvt", "per-lane b, f32 seed + 2 Newton", "per-lane b, f64 seed + 2 Newton",
	                        "2 x IEEE, same b", "2 numerators, one f32-seeded R", "(loop overhead)" };
	float base = 0.0f;
	for (int mode = 6; mode >= 0; --mode) {
		auto launch = [&]() {
			switch (mode) {
			case 0: hipLaunchKernelGGL(k_cost<0>, dim3(4096), dim3(256), 0, 0, out, iters, 3.7f, 1.0 / 3.7f); break;
			case 1: hipLaunchKernelGGL(k_cost<1>, dim3(4096), dim3(256), 0, 0, out, iters, 3.7f, 1.0 / (double)3.7f); break;
			case 2: hipLaunchKernelGGL(k_cost<2>, dim3(4096), dim3(256), 0, 0, out, iters, 3.7f, 0.0); break;
			case 3: hipLaunchKernelGGL(k_cost<3>, dim3(4096), dim3(256), 0, 0, out, iters, 3.7f, 0.0); break;
			case 4: hipLaunchKernelGGL(k_cost<4>, dim3(4096), dim3(256), 0, 0, out, iters, 3.7f, 0.0); break;
			case 5: hipLaunchKernelGGL(k_cost<5>, dim3(4096), dim3(256), 0, 0, out, iters, 3.7f, 0.0); break;
			default: hipLaunchKernelGGL(k_cost<6>, dim3(4096), dim3(256), 0, 0, out, iters, 3.7f, 0.0); break;
			}
		};
		launch(); hipDeviceSynchronize();
		hipEventRecord(e0); launch(); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
		float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 2;
		if (mode == 6) base = ms;
		// wave-instruction issue slots per iteration per wave: 4096*4 waves on 256*4 SIMDs = 16 waves per SIMD, 4 cycles per wave64 VALU op at 2.4 GHz
		double cyc = (ms - base) * 1e-3 * 2.4e9 / (16.0 * iters) / 4.0;
		printf("%-36s %8.3f ms   ~%5.1f VALU issue slots per iteration above the loop overhead\n", names[mode], ms, mode == 6 ? 0.0 : cyc);
	}
	hipLaunchKernelGGL(k_check, dim3(16384), dim3(256), 0, 0, n_check, 20260928u, cnt);
	unsigned long long h[4];
	hipMemcpy(h, cnt, 32, hipMemcpyDeviceToHost);
	printf("checked %lld operand pairs against IEEE a/b: mismatches uniform-R %llu, f32-seeded R %llu, f64-seeded R %llu; IEEE fallbacks taken (subnormal quotients) %llu\n",
	       n_check, h[0], h[1], h[2], h[3]);
	return (h[0] | h[1] | h[2]) ? 1 : 0;
}
