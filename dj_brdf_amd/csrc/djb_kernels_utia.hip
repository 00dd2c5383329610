// djb_kernels_utia.hip -- utia::eval / evalp batches (dj_brdf.h:1063-1157), two-tier and bit-identical; under DJB_OPT_CONTRACT_1E5 the sRGB
// power of the decode runs on the fast transcendentals (everything else stays the reference's bits).  Split from djb_kernels_eval.hip in
// round 6 (the one-kernel form k_eval<UTIA> of DJB_OPT_UTIA_EXACT_ONLY / in-place calls stays there).
#include "djb_internal.hpp"
#include <stdlib.h>

using namespace djbdev;

namespace {

constexpr int BLOCK = 256;
#ifndef DJB_UTIA_MIN_WAVES
#define DJB_UTIA_MIN_WAVES 4
#endif

inline int grid_for(long long n)
{
	long long blocks = (n + BLOCK - 1) / BLOCK;
	const long long cap = 256LL * 64;   // 4 workgroups per CU resident, 16 rounds of them (2.58 ms per 1e8 pairs; 4096: 2.66, 1024: 2.94, one per tile: 2.75)
	if (blocks > cap) blocks = cap;
	if (blocks < 1) blocks = 1;
	return (int)blocks;
}
inline bool dense(const View &v) { return v.stride == 1 || v.x == nullptr; }

// utia::eval, two-tier.  Tier 1 (k_utia_v2) decides every pair it can with cheap arithmetic and lists the rest (the index of every pair
// with an angle next to a float rounding boundary, a cell estimate next to a cell boundary, a guarded shortcut that wants its exact
// form: ~2e-4 of the pairs) on a worklist; tier 2 (k_eval_utia_fix) re-evaluates those with utia_eval -- the reference as written, with
// glibc's atan2 behind the azimuths -- and overwrites the result.  If the list overflows, tier 2 redoes the whole batch, so the result
// never depends on the capacity.  (Kept inside one kernel as a rarely taken branch the exact code doubles the kernel's time.)
//
// Tier 1, round 6: the record fetch starts from ESTIMATED cells and the reference's angles are computed under it
// (djb_device_tables.inc: utia_cells_estimate / utia_weights / utia_decode_*).  CT = DJB_OPT_CONTRACT_1E5 (fast sRGB power only:
// the 16-tap sums stay the reference's bits).  The two 96-byte payloads of a pair's records are fetched wave-cooperatively,
// straight into LDS (global_load_lds_dwordx4): 384 chunks of 16 bytes per record set = 6 wave-instructions, chunk g = 64 s + lane
// belongs to lane g / 6 and lands at float4 slot g of the wave's 6 KB tile, so a wave-instruction touches ~11 table lines instead of
// 64 and the owner reads its six chunks back from slots 6 L .. 6 L + 5.  Measured forms (profiles/r06/utia_v2_forms.txt, ms per 1e8
// pairs, exact / contract): angles first + lane-private fetch (rounds 2-5) 2.88 / -; this kernel 2.72 / 2.54 before the angles
// were rebuilt; lane-private fetch from estimated cells 3.63 / 3.36; two tiles in flight 2.84 / 2.60 (LDS limits it to 3 waves).
template <int WANT, bool CT, bool DENSE>
__global__ __launch_bounds__(BLOCK, DJB_UTIA_MIN_WAVES) void k_utia_v2(Brdf b, long long n, View vi, View vo, View vout, float *out_pdf,
                                                                      unsigned int *list, unsigned int cap, unsigned int *count)
{
	__shared__ float4 s_tile[BLOCK / 64][384];
	__shared__ double s_atan[16];
	const lds_f64p T = atan_tab(atan_tab_to_lds(s_atan, threadIdx.x));
	__syncthreads();
	const long long stride = (long long)gridDim.x * BLOCK;
	const unsigned int t = threadIdx.x, wave = t >> 6, lane = t & 63u;
	typedef __attribute__((address_space(3))) void lds_void;
	typedef __attribute__((address_space(1))) const void glb_void;
	float4 *tile = s_tile[wave];
	for (long long k0 = (long long)blockIdx.x * BLOCK; k0 < n; k0 += stride) {     // k0: workgroup-uniform
		const long long k = k0 + t;
		const bool live = k < n;
		v3 i = mk(0, 0, 1), o = mk(0, 0, 1);
		// the streams carry the non-temporal hint: the record gathers hit an XCD's L2 for 62 % (LRU; the statically hottest 4 MB would give 84 %) and
		// every stream line left behind costs them some of it -- same-box A/B 2.42 -> 2.38 ms per 1e8 pairs (contract 2.27 -> 2.21)
		if (live) { i = DENSE ? load3_dense_nt(vi, k0, t) : load3(vi, k); o = DENSE ? load3_dense_nt(vo, k0, t) : load3(vo, k); }
		const UtiaCells c = utia_cells_estimate(i, o);
		int e[2];
		utia_record_index(c, e);
		float4 q[6];
		auto fetch = [&](int a) {
#pragma unroll
			for (unsigned int s = 0; s < 6u; ++s) {
				const unsigned int g = s * 64u + lane, r = (g * 10923u) >> 16, chunk = g - 6u * r;     // r = g / 6 for g < 384
				const unsigned int e_src = (unsigned int)__shfl(e[a], (int)r);
				// uniform base + 32-bit byte offset (the table is 10.6 MB): the saddr form of the load, no 64-bit address arithmetic per lane
				const char *src = (const char *)b.utia + (size_t)((8u * e_src + chunk) << 4);
				__builtin_amdgcn_global_load_lds((glb_void *)src, (lds_void *)(tile + s * 64u), 16, 0, 0);
			}
		};
		auto take = [&]() {
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                        // this wave's six LDS-direct loads have landed
#pragma unroll
			for (unsigned int j = 0; j < 6u; ++j) q[j] = tile[lane * 6u + j];
			asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                      // read before the tile is overwritten
		};
		fetch(0);
		UtiaTaps u;
		bool ok = utia_weights(i, o, c, u, T);                                      // under the fetch
		float acc[3] = { 0.0f, 0.0f, 0.0f };
		take();
		fetch(1);                                                                   // in flight while record 0 is accumulated
		utia_accumulate(u, 0, q, acc);
		take();
		utia_accumulate(u, 1, q, acc);
		const v3 ev = CT ? utia_decode_ct(u, acc, ok) : utia_decode_t1(u, acc, ok);
		if (live) {
			v3 fr = (WANT & 2) ? scale(i.z, ev) : ev;                                   // brdf::evalp, dj_brdf.h:803-806
			if (DENSE) store3_dense_nt(vout, k0, t, fr); else store3(vout, k, fr);
			if (WANT & 4) { float pdf = F(D(i.z) / DJB_PI); if (DENSE) (*dense_at(out_pdf + k0, t)) = pdf; else out_pdf[k] = pdf; }   // dj_brdf.h:842-845
			if (__builtin_expect(!ok, 0)) {
				const unsigned int slot = atomicAdd(count, 1u);
				if (slot < cap) list[slot] = (unsigned int)k;
			}
		}
	}
}
// djb_selftest_fast_trig: the arctangent core's sites against their previous forms.
//   mode 0: the floats bits(first + k), k < n, as polar cosines through utia_acos_deg_t1 against acos_deg_f (identical to the host's by
//           exhaustion); counted over the decided ones;
//   mode 1 / 8: hash-generated float pairs (y, x) -- unit-circle points, axis-hugging, tiny / huge magnitudes, signed zeros, grid lines --
//           through atan2_fast_f32 with scale r2d / 1 against atan2_to_f32 (glibc's atan2 behind a guard); decided ones;
//   mode 2..7: site FT_ACOS .. FT_ATAN_SQRT: the floats bits(first + k) through the site WITH the table against the site without it --
//           every float, decided or not (an undecided one takes the previous form: it can only differ if the plumbing is wrong);
//   mode 9: float(tan(double x)) from sincos_fast against the device libm's, every float the same way.
// counters = {decided, different floats (must be 0), undecided, largest |core double - device-libm double| in units of 2^-52 of the value
// over the decided ones (modes 0, 1, 8): the distance to the reference's double up to the 2 ulp64 between the two libms; the guard is 4096}
template <int S> __device__ void ft_site(float x, LdsTab AT, unsigned long long &n_ok, unsigned long long &n_bad, unsigned long long &n_und)
{
	bool ok;
	(void)trig_fast<S>(x, atan_tab(AT), ok);
	if (ok) ++n_ok; else ++n_und;
	if (__float_as_uint(trig_at<S>(x, AT)) != __float_as_uint(trig_prev<S>(x))) ++n_bad;
}
__global__ __launch_bounds__(BLOCK) void k_fast_trig_selftest(long long n, int mode, uint32_t first, uint32_t seed, unsigned long long *counters)
{
	__shared__ double s_atan[16];
	const LdsTab AT = atan_tab_to_lds(s_atan, threadIdx.x);
	__syncthreads();
	const lds_f64p T = atan_tab(AT);
	unsigned long long n_ok = 0, n_bad = 0, n_und = 0, worst = 0;
	const long long stride = (long long)gridDim.x * BLOCK;
	const double r2d = (mode == 8) ? 1.0 : D(F(180.0 / DJB_PI));
	for (long long k = (long long)blockIdx.x * BLOCK + threadIdx.x; k < n; k += stride) {
		if (mode == 9) {                                   // float(tan(double x)): sincos_fast quotient against the device libm's tan, every float
			const float x = __uint_as_float(first + (uint32_t)k);
			bool ok;
			(void)tan_fast_f32(x, ok);
			if (ok) ++n_ok; else ++n_und;
			if (__float_as_uint(tan_fast_f(x)) != __float_as_uint(tan_f(x))) ++n_bad;
			continue;
		}
		if (mode >= 2 && mode <= 7) {
			const float x = __uint_as_float(first + (uint32_t)k);
			switch (mode) {
			case 2: ft_site<FT_ACOS>(x, AT, n_ok, n_bad, n_und); break;
			case 3: ft_site<FT_ACOS_U>(x, AT, n_ok, n_bad, n_und); break;
			case 4: ft_site<FT_ACOS_U32>(x, AT, n_ok, n_bad, n_und); break;
			case 5: ft_site<FT_ATAN_U>(x, AT, n_ok, n_bad, n_und); break;
			case 6: ft_site<FT_ATAN_SQU>(x, AT, n_ok, n_bad, n_und); break;
			default: ft_site<FT_ATAN_SQRT>(x, AT, n_ok, n_bad, n_und); break;
			}
			continue;
		}
		bool ok; float got, want; double dl, dd;
		if (mode == 0) {
			const float z = __uint_as_float(first + (uint32_t)k);
			got = utia_acos_deg_t1(z, T, ok, &dd); want = acos_deg_f(z); dl = r2d * acos(D(z));
		} else {
			const uint32_t h0 = hash_u32(seed, (uint64_t)k, 1u), h1 = hash_u32(seed, (uint64_t)k, 2u), h2 = hash_u32(seed, (uint64_t)k, 3u);
			float y, x;
			const unsigned int fam = h2 & 7u;
			if (fam < 3u) {                                   // a point of the unit circle scaled by sin(theta): what a direction's (y, x) is
				const float ph = 6.2831853f * (float)(h0 >> 8) * 0x1p-24f, sc = sqrtf((float)(h1 >> 8) * 0x1p-24f);
				y = sc * sinf(ph); x = sc * cosf(ph);
			} else if (fam == 3u) { y = __uint_as_float(h0); x = __uint_as_float(h1); }                                   // any two floats
			else if (fam == 4u) { y = __uint_as_float((h0 & 0x807fffffu) | 0x3f000000u); x = y * (1.0f + (float)(int)(h1 & 15u) * 0x1p-23f) * ((h2 & 8u) ? -1.0f : 1.0f); }   // |y| ~ |x|
			else if (fam == 5u) { y = (float)(int)(h0 & 0xffu) * 0x1p-20f * ((h2 & 8u) ? -1.0f : 1.0f); x = __uint_as_float((h1 & 0x807fffffu) | 0x3f000000u); }   // near the x axis, +-0 included
			else if (fam == 6u) { x = (float)(int)(h0 & 0xffu) * 0x1p-20f * ((h2 & 8u) ? -1.0f : 1.0f); y = __uint_as_float((h1 & 0x807fffffu) | 0x3f000000u); }   // near the y axis, +-0 included
			else { const float t7 = tanf(0.13089969f * (float)(h0 % 49u)); y = t7 * __uint_as_float((h1 & 0x007fffffu) | 0x3f000000u); x = __uint_as_float((h1 & 0x007fffffu) | 0x3f000000u); if (h2 & 8u) x = -x; if (h2 & 16u) y = -y; }   // on the 7.5-degree grid lines
			got = atan2_fast_f32(y, x, r2d, T, ok, &dd); want = atan2_to_f32(y, x, r2d); dl = r2d * atan2(D(y), D(x));
		}
		if (!ok) { ++n_und; continue; }
		++n_ok;
		if (__float_as_uint(got) != __float_as_uint(want)) ++n_bad;
		const double adl = fabs(dl);
		const unsigned long long q = (unsigned long long)(fabs(dd - dl) / (adl * 0x1p-52 + 1e-300));
		worst = q > worst ? q : worst;
	}
	atomicAdd(&counters[0], n_ok); atomicAdd(&counters[1], n_bad); atomicAdd(&counters[2], n_und); atomicMax(&counters[3], worst);
}
template <int WANT>
__global__ __launch_bounds__(BLOCK) void k_eval_utia_fix(Brdf b, long long n, View vi, View vo, View vout, float *out_pdf,
                                                         const unsigned int *list, unsigned int cap, const unsigned int *count)
{
	const unsigned int c = *count;
	const bool all = c > cap;                                    // overflow: redo the whole batch
	const long long m = all ? n : (long long)c;
	const Params none = {};
	const long long stride = (long long)gridDim.x * BLOCK;
	for (long long j = (long long)blockIdx.x * BLOCK + threadIdx.x; j < m; j += stride) {
		const long long k = all ? j : (long long)list[j];
		v3 i = load3(vi, k), o = load3(vo, k), fr = mk(0, 0, 0); float pdf = 0.0f;
		eval_one<KIND_UTIA, WANT>(b, none, i, o, fr, pdf);
		if (WANT & 3) store3(vout, k, fr);
		if (WANT & 4) out_pdf[k] = pdf;
	}
}
template <int WANT>
hipError_t launch_utia_tt(hipStream_t s, const Brdf &b, long long n, const View &i, const View &o, const View &out,
                          float *out_pdf, unsigned int *list, unsigned int cap, unsigned int *count, bool contract)
{
	hipError_t e = hipMemsetAsync(count, 0, 16, s);
	if (e != hipSuccess) return e;
	dim3 g(grid_for(n)), t(BLOCK);
	const bool dn = dense(i) && dense(o) && dense(out);
	if (contract) {
		if (dn) hipLaunchKernelGGL((k_utia_v2<WANT, true, true>), g, t, 0, s, b, n, i, o, out, out_pdf, list, cap, count);
		else hipLaunchKernelGGL((k_utia_v2<WANT, true, false>), g, t, 0, s, b, n, i, o, out, out_pdf, list, cap, count);
	} else {
		if (dn) hipLaunchKernelGGL((k_utia_v2<WANT, false, true>), g, t, 0, s, b, n, i, o, out, out_pdf, list, cap, count);
		else hipLaunchKernelGGL((k_utia_v2<WANT, false, false>), g, t, 0, s, b, n, i, o, out, out_pdf, list, cap, count);
	}
	if ((e = hipGetLastError()) != hipSuccess) return e;
	hipLaunchKernelGGL((k_eval_utia_fix<WANT>), dim3(64), t, 0, s, b, n, i, o, out, out_pdf, list, cap, count);
	return hipGetLastError();
}

} // namespace

namespace djbk {

hipError_t launch_utia_twotier(hipStream_t s, const Brdf &b, long long n, const View &i, const View &o, const View &out,
                               float *out_pdf, int want, unsigned int *list, unsigned int cap, unsigned int *count, bool contract)
{
	if (n <= 0) return hipSuccess;
	switch (want) {
	case 1: return launch_utia_tt<1>(s, b, n, i, o, out, out_pdf, list, cap, count, contract);
	case 2: return launch_utia_tt<2>(s, b, n, i, o, out, out_pdf, list, cap, count, contract);
	case 5: return launch_utia_tt<5>(s, b, n, i, o, out, out_pdf, list, cap, count, contract);
	case 6: return launch_utia_tt<6>(s, b, n, i, o, out, out_pdf, list, cap, count, contract);
	}
	return hipErrorInvalidValue;
}

hipError_t launch_fast_trig_selftest(hipStream_t s, long long n, int mode, uint32_t first, uint32_t seed, unsigned long long *counters4)
{
	if (n <= 0) return hipSuccess;
	hipLaunchKernelGGL(k_fast_trig_selftest, dim3(grid_for(n)), dim3(BLOCK), 0, s, n, mode, first, seed, counters4);
	return hipGetLastError();
}

} // namespace djbk
