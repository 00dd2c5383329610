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
	const long long cap = 256LL * 16;   // 256 CUs x 16 resident workgroups' worth, grid-stride beyond
	if (blocks > cap) blocks = cap;
	if (blocks < 1) blocks = 1;
	return (int)blocks;
}
inline bool dense(const View &v) { return v.stride == 1 || v.x == nullptr; }

// utia::eval, two-tier.  The exact fall-back of the azimuths (glibc's atan2, djb_device.hpp atan2_to_f32) kept inside
// k_eval<UTIA> as a rarely taken branch doubles the kernel's time (2.8 -> 5.3 ms per 1e8; inline or as a call: its
// registers and constants land in the loop).  Tier 1 runs the same per-pair code without it (utia_eval_t<true>) and
// appends the index of every pair with an azimuth that was not decided away from a float rounding boundary (8e-6 of
// them) to a worklist; tier 2 re-evaluates those with utia_eval and overwrites the result.  If the list overflows,
// tier 2 redoes the whole batch, so the result never depends on the capacity.
//
// Tried and dropped (round 4, profiles/r04/NOTES.md): a wave-cooperative record fetch -- eight neighbouring lanes load the
// eight 16-byte chunks of one record, 8 lines per load instruction instead of 64, data to their owner lanes through LDS.
// Bit-identical, but 4.0-4.2 ms per 1e8 against 2.83: the 16 bpermutes, 16 LDS writes and 12 LDS reads per pair-set cost
// more than the line look-ups they save.
template <int WANT, bool DENSE>
__global__ __launch_bounds__(BLOCK, DJB_UTIA_MIN_WAVES) void k_eval_utia_t1(Brdf b, long long n, View vi, View vo, View vout, float *out_pdf,
                                                                        unsigned int *list, unsigned int cap, unsigned int *count)
{
	const long long stride = (long long)gridDim.x * BLOCK;
	const unsigned int t = threadIdx.x;
	for (long long k0 = (long long)blockIdx.x * BLOCK; k0 < n; k0 += stride) {     // k0: workgroup-uniform
		const long long k = k0 + t;
		if (k >= n) continue;
		v3 i = DENSE ? load3_dense(vi, k0, t) : load3(vi, k), o = DENSE ? load3_dense(vo, k0, t) : load3(vo, k);
		bool ok;
		v3 e = utia_eval_t<true>(b, i, o, ok);
		v3 fr = (WANT & 2) ? scale(i.z, e) : e;                                        // brdf::evalp, dj_brdf.h:803-806
		if (DENSE) store3_dense(vout, k0, t, fr); else store3(vout, k, fr);
		if (WANT & 4) { float pdf = F(D(i.z) / DJB_PI); if (DENSE) (out_pdf + k0)[t] = pdf; else out_pdf[k] = pdf; }   // dj_brdf.h:842-845
		if (__builtin_expect(!ok, 0)) {
			const unsigned int slot = atomicAdd(count, 1u);
			if (slot < cap) list[slot] = (unsigned int)k;
		}
	}
}
// Tier 1, round 6 (k_utia_v2): the record fetch starts from ESTIMATED cells and the reference's angles are computed under it
// (djb_device_tables.inc: utia_cells_estimate / utia_weights / utia_decode_*).  CT = DJB_OPT_CONTRACT_1E5 (fast sRGB power only:
// the 16-tap sums stay the reference's bits).  COOP: the two 96-byte payloads of a pair's records are fetched wave-cooperatively,
// straight into LDS (global_load_lds_dwordx4): 384 chunks of 16 bytes per record set = 6 wave-instructions, chunk g = 64 s + lane
// belongs to lane g / 6 and lands at float4 slot g of the tile, so a wave-instruction touches ~11 table lines instead of 64 and the
// owner reads its six chunks back from slots 6 L .. 6 L + 5.  TILES = 2: both record sets in flight at once (12 KB of LDS per wave).
#ifndef DJB_UTIA_V2_WAVES
#define DJB_UTIA_V2_WAVES 4
#endif
#ifndef DJB_UTIA_FORM_DEFAULT
#define DJB_UTIA_FORM_DEFAULT 2
#endif
template <int WANT, bool CT, int COOP, bool DENSE>
__global__ __launch_bounds__(BLOCK, DJB_UTIA_V2_WAVES) void k_utia_v2(Brdf b, long long n, View vi, View vo, View vout, float *out_pdf,
                                                                     unsigned int *list, unsigned int cap, unsigned int *count)
{
	constexpr int TILES = COOP ? COOP : 1;
	__shared__ float4 s_tile[COOP ? BLOCK / 64 : 1][COOP ? TILES * 384 : 1];
	const long long stride = (long long)gridDim.x * BLOCK;
	const unsigned int t = threadIdx.x, wave = t >> 6, lane = t & 63u;
	typedef __attribute__((address_space(3))) void lds_void;
	typedef __attribute__((address_space(1))) const void glb_void;
	float4 *tile = s_tile[COOP ? wave : 0];
#if DJB_UTIA_V2_NOLOOP
	(void)stride;
	{ const long long k0 = (long long)blockIdx.x * BLOCK;
#else
	for (long long k0 = (long long)blockIdx.x * BLOCK; k0 < n; k0 += stride) {     // k0: workgroup-uniform
#endif
		const long long k = k0 + t;
		const bool live = k < n;
		if (!COOP && !live) return;
		v3 i = mk(0, 0, 1), o = mk(0, 0, 1);
		if (live) { i = DENSE ? load3_dense(vi, k0, t) : load3(vi, k); o = DENSE ? load3_dense(vo, k0, t) : load3(vo, k); }
		const UtiaCells c = utia_cells_estimate(i, o);
		int e[2];
		utia_record_index(c, e);
		float4 q0[6], q1[6];
		auto fetch = [&](int a, float4 *dst) {
#pragma unroll
			for (unsigned int s = 0; s < 6u; ++s) {
				const unsigned int g = s * 64u + lane, r = (g * 10923u) >> 16, chunk = g - 6u * r;     // r = g / 6 for g < 384
				const int e_src = __shfl(e[a], (int)r);
				const float4 *src = b.utia + 8 * (size_t)e_src + chunk;
				__builtin_amdgcn_global_load_lds((glb_void *)src, (lds_void *)(dst + s * 64u), 16, 0, 0);
			}
		};
		auto take = [&](const float4 *src, float4 (&q)[6]) {
#pragma unroll
			for (unsigned int j = 0; j < 6u; ++j) q[j] = src[lane * 6u + j];
			asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                      // read before the tile is overwritten
		};
		if (COOP) {
			fetch(0, tile);
			if (TILES == 2) fetch(1, tile + 384);
		} else {
			const float4 *r0 = b.utia + 8 * (size_t)e[0], *r1 = b.utia + 8 * (size_t)e[1];
#pragma unroll
			for (int j = 0; j < 6; ++j) { q0[j] = r0[j]; q1[j] = r1[j]; }
		}
		UtiaTaps u;
		bool ok = utia_weights(i, o, c, u);                                         // under the fetch
		float acc[3] = { 0.0f, 0.0f, 0.0f };
		if (COOP) {
			if (TILES == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			take(tile, q0);
			if (TILES == 1) fetch(1, tile);
		}
		utia_accumulate(u, 0, q0, acc);
		if (COOP) {
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			take(TILES == 2 ? tile + 384 : tile, q1);
		}
		utia_accumulate(u, 1, q1, acc);
		const v3 ev = CT ? utia_decode_ct(u, acc, ok) : utia_decode_t1(u, acc, ok);
		if (live) {
			v3 fr = (WANT & 2) ? scale(i.z, ev) : ev;                                   // brdf::evalp, dj_brdf.h:803-806
			if (DENSE) store3_dense(vout, k0, t, fr); else store3(vout, k, fr);
			if (WANT & 4) { float pdf = F(D(i.z) / DJB_PI); if (DENSE) (out_pdf + k0)[t] = pdf; else out_pdf[k] = pdf; }   // dj_brdf.h:842-845
			if (__builtin_expect(!ok, 0)) {
				const unsigned int slot = atomicAdd(count, 1u);
				if (slot < cap) list[slot] = (unsigned int)k;
			}
		}
	}
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
template <int WANT, bool CT, int COOP>
void launch_utia_v2(hipStream_t s, dim3 g, dim3 t, bool dn, const Brdf &b, long long n, const View &i, const View &o, const View &out,
                    float *out_pdf, unsigned int *list, unsigned int cap, unsigned int *count)
{
	if (dn) hipLaunchKernelGGL((k_utia_v2<WANT, CT, COOP, true>), g, t, 0, s, b, n, i, o, out, out_pdf, list, cap, count);
	else hipLaunchKernelGGL((k_utia_v2<WANT, CT, COOP, false>), g, t, 0, s, b, n, i, o, out, out_pdf, list, cap, count);
}
template <int WANT>
hipError_t launch_utia_tt(hipStream_t s, const Brdf &b, long long n, const View &i, const View &o, const View &out,
                          float *out_pdf, unsigned int *list, unsigned int cap, unsigned int *count, bool contract)
{
	hipError_t e = hipMemsetAsync(count, 0, 16, s);
	if (e != hipSuccess) return e;
	dim3 g(grid_for(n)), t(BLOCK);
	const bool dn = dense(i) && dense(o) && dense(out);
	// tier-1 form (A/B switch of round 6, profiles/r06/utia_v2.txt): 0 = k_eval_utia_t1 (angles first), 1 = k_utia_v2 lane-private fetch,
	// 2 / 3 = k_utia_v2 with the wave-cooperative fetch through one / two LDS tiles
	static const int form = getenv("DJB_UTIA_FORM") ? atoi(getenv("DJB_UTIA_FORM")) : DJB_UTIA_FORM_DEFAULT;
	if (form == 0 && !contract) {
		if (dn) hipLaunchKernelGGL((k_eval_utia_t1<WANT, true>), g, t, 0, s, b, n, i, o, out, out_pdf, list, cap, count);
		else hipLaunchKernelGGL((k_eval_utia_t1<WANT, false>), g, t, 0, s, b, n, i, o, out, out_pdf, list, cap, count);
	} else if (contract) {
		if (form == 2) launch_utia_v2<WANT, true, 1>(s, g, t, dn, b, n, i, o, out, out_pdf, list, cap, count);
		else if (form == 3) launch_utia_v2<WANT, true, 2>(s, g, t, dn, b, n, i, o, out, out_pdf, list, cap, count);
		else launch_utia_v2<WANT, true, 0>(s, g, t, dn, b, n, i, o, out, out_pdf, list, cap, count);
	} else {
		if (form == 2) launch_utia_v2<WANT, false, 1>(s, g, t, dn, b, n, i, o, out, out_pdf, list, cap, count);
		else if (form == 3) launch_utia_v2<WANT, false, 2>(s, g, t, dn, b, n, i, o, out, out_pdf, list, cap, count);
		else launch_utia_v2<WANT, false, 0>(s, g, t, dn, b, n, i, o, out, out_pdf, list, cap, count);
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

} // namespace djbk
