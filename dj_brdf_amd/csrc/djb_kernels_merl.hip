// djb_kernels_merl.hip -- two-tier, bit-exact MERL lookup (merl::eval, dj_brdf.h:987-1024).
//
// k_eval<MERL> (djb_kernels_eval.hip) reproduces the reference operation by operation: 8 fp64
// libm-class calls per pair, ~770 VALU instructions, VALU-bound at 17 % of the HBM roofline.
// Here the same result comes from two tiers inside ONE kernel:
//   tier 1   every pair: fp32 closed-form angles + guard bands (merl_index_fast).  Certain pairs (~99.6 %) gather their
//            table entry and are done; an ambiguous pair's {k, i, o} goes to a per-wave queue in LDS (one wave, in-order
//            LDS: no barrier, no atomics).
//   tier 2   whenever a wave has queued 64 pairs -- and once more at its end for what is left -- it runs the exact fp64
//            path (merl_index) on them as one dense wave and overwrites their placeholders.
// Until round 4 tier 2 was a second kernel fed through a sharded worklist in HBM (0.43-0.52 ms per 1e9 pairs, bound by its
// scattered result stores, plus a memset, a launch and an adaptive list capacity on the host).  The drain costs the kernel the
// exact path's registers (~140 VGPRs: 3 waves per SIMD instead of 6), which tier 1 does not notice -- it is bound by the
// memory system at any occupancy from 2 to 7 waves (profiles/r04/merl_occupancy.txt).
// Bit-exactness argument and calibration of the bands: DESIGN.md 4.2.
#include "djb_internal.hpp"
#include "djb_worklist.hpp"
#include <stdlib.h>

using namespace djbdev;

namespace {

constexpr int BLOCK = 256;
#ifndef DJB_MERL_GRID_CAP
#define DJB_MERL_GRID_CAP (256LL * 64)
#endif

inline int grid_for(long long n, long long cap = 256LL * 16)
{
	long long blocks = (n + BLOCK - 1) / BLOCK;
	if (blocks > cap) blocks = cap;
	if (blocks < 1) blocks = 1;
	return (int)blocks;
}

template <int WANT>
DJB_DEV void merl_emit(const Brdf &b, int idx, v3 i, long long k, const View &vout, float *out_pdf)
{
	if (WANT & 3) {
		MerlTexel t = b.merl[idx];
		v3 e = mk(t.x, t.y, t.z);
		store3(vout, k, (WANT & 2) ? scale(i.z, e) : e);          // brdf::evalp, dj_brdf.h:803-806
	}
	if (WANT & 4) out_pdf[k] = F(D(i.z) / DJB_PI);                // brdf::pdf,   dj_brdf.h:842-845
}

// ---- djb_merl_bin_keys_batch: the 21-bit bin key of every pair from TIER-1 arithmetic only (fp32 closed forms, ~120 VALU, no fp64
// path, no table access) -- for callers that order a batch before the look-up: the look-up's rate is set by how many distinct table
// lines a launch touches per unit time (0.23 of the roofline on look-ups uniform over the bins, 0.40 on random directions, 0.67 on a
// renderer-coherent batch).  The key IS merl_index(i, o) for every pair tier 1 is certain of (99.6 % of random directions) and the
// neighbouring bin the estimate falls into otherwise: good for ordering, not a substitute for djb_merl_index_batch.
__global__ __launch_bounds__(BLOCK) void k_merl_keys(long long n, View vi, View vo, uint32_t *keys, MerlGuard g)
{
	const long long stride = (long long)gridDim.x * BLOCK;
	for (long long k = (long long)blockIdx.x * BLOCK + threadIdx.x; k < n; k += stride) {
		int idx;
		(void)merl_index_fast(load3(vi, k), load3(vo, k), g, idx);
		idx = idx < 0 ? 0 : idx > 90 * 90 * 180 - 1 ? 90 * 90 * 180 - 1 : idx;      // NaN / stray directions: any valid key
		keys[k] = (uint32_t)idx;
	}
}

// ---- the per-wave queue of ambiguous pairs and its drain.  QCAP: fewer than 64 pairs wait when an iteration starts and an
// iteration adds at most 4 x 64, so 320 slots always suffice; there is ONE drain site per kernel (the exact path is ~6 000
// instructions and ~140 registers: every further inlined copy costs both)
constexpr unsigned int QCAP = 320;
typedef unsigned int MerlQueue[7][QCAP];
DJB_DEV void merl_queue(MerlQueue &q, unsigned int &qn, int lane, bool amb, unsigned int k, v3 i, v3 o)
{
	const unsigned long long mask = __ballot(amb);
	if (!mask) return;
	if (amb) {
		const unsigned int j = qn + (unsigned int)__popcll(mask & ((1ull << lane) - 1ull));
		q[0][j] = k;
		q[1][j] = __float_as_uint(i.x); q[2][j] = __float_as_uint(i.y); q[3][j] = __float_as_uint(i.z);
		q[4][j] = __float_as_uint(o.x); q[5][j] = __float_as_uint(o.y); q[6][j] = __float_as_uint(o.z);
	}
	qn += (unsigned int)__popcll(mask);
}
// tier 2 on `cnt` queued pairs starting at slot `first`, one per lane: the operation-by-operation fp64 path
template <int WANT>
DJB_DEV void merl_drain(const Brdf &b, MerlQueue &q, unsigned int first, unsigned int cnt, int lane, long long k_base, const View &vout, float *out_pdf)
{
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
	__builtin_amdgcn_wave_barrier();
	// the placeholders of these very pairs were stored by this wave as parts of float4 stores: they must have left the wave
	// before the scalar stores below are issued
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	if ((unsigned int)lane < cnt) {
		const unsigned int j = first + (unsigned int)lane;
		const long long k = k_base + (long long)q[0][j];
		const v3 i = mk(__uint_as_float(q[1][j]), __uint_as_float(q[2][j]), __uint_as_float(q[3][j]));
		const v3 o = mk(__uint_as_float(q[4][j]), __uint_as_float(q[5][j]), __uint_as_float(q[6][j]));
		merl_emit<WANT>(b, merl_index(i, o), i, k, vout, out_pdf);
	}
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
	__builtin_amdgcn_wave_barrier();
}
template <int WANT>
__global__ __launch_bounds__(BLOCK) void k_merl_fast(Brdf b, long long k_begin, long long n, View vi, View vo,
                                                     View vout, float *out_pdf, MerlGuard g)
{
	__shared__ MerlQueue wbuf[BLOCK / 64];
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	unsigned int qn = 0;                                         // wave-uniform
	const long long stride = (long long)gridDim.x * BLOCK;
	for (long long k0 = k_begin + (long long)blockIdx.x * BLOCK; ; k0 += stride) {   // block-uniform trip count; one extra trip flushes the queue
		const bool last = k0 >= n;
		if (!last) {
			long long k = k0 + threadIdx.x;
			bool amb = false;
			v3 i = mk(0, 0, 1), o = mk(0, 0, 1);
			if (k < n) {
				i = load3(vi, k); o = load3(vo, k);
				int idx;
				if (merl_index_fast(i, o, g, idx)) merl_emit<WANT>(b, idx, i, k, vout, out_pdf);
				else amb = true;                                  // tier 2 finishes this pair
			}
			merl_queue(wbuf[wave], qn, lane, amb, (unsigned int)k, i, o);   // pair indices travel as uint32 (the host chunks at 2^31)
		}
		while (qn >= 64u || (last && qn)) {                       // tier 2: a full wave of waiting pairs, or what is left at the end
			const unsigned int cnt = qn < 64u ? qn : 64u;
			qn -= cnt;
			merl_drain<WANT>(b, wbuf[wave], qn, cnt, lane, 0, vout, out_pdf);
		}
		if (last) break;
	}
}

// Same kernel for dense SoA input (stride 1, 16-byte aligned): four consecutive pairs per lane,
// float4 loads/stores, so each wave keeps 6 x 1 KiB loads and four independent table gathers in
// flight -- the scalar version is latency-bound (one load -> compute -> gather -> store chain per
// wave).  Ambiguous pairs get a placeholder in the float4 store; tier 2 overwrites it from the same wave, later.
#ifdef DJB_EXP_MERL_WAVES          // timing experiment: tier 1 at a forced occupancy (waves per SIMD)
#define DJB_MERL_V4_ATTR __attribute__((amdgpu_waves_per_eu(DJB_EXP_MERL_WAVES, DJB_EXP_MERL_WAVES)))
#else
#define DJB_MERL_V4_ATTR
#endif
template <int WANT>
__global__ __launch_bounds__(BLOCK) DJB_MERL_V4_ATTR void k_merl_fast_v4(Brdf b, long long n4, View vi, View vo, View vout,
                                                                          float *out_pdf, MerlGuard g)
{
	__shared__ MerlQueue wbuf[BLOCK / 64];
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	unsigned int qn = 0;
	const float4 *ix4 = (const float4 *)vi.x, *iy4 = (const float4 *)vi.y, *iz4 = (const float4 *)vi.z;
	const float4 *ox4 = (const float4 *)vo.x, *oy4 = (const float4 *)vo.y, *oz4 = (const float4 *)vo.z;
	long long stride = (long long)gridDim.x * BLOCK;
	for (long long q0 = (long long)blockIdx.x * BLOCK; ; q0 += stride) {       // one extra trip flushes the queue
		const bool last = q0 >= n4;
		long long q = q0 + threadIdx.x;
		bool amb[4] = { false, false, false, false };
		float ixs[4], iys[4], izs[4], oxs[4], oys[4], ozs[4];
		if (!last && q < n4) {
			// the 36 B/pair streams are touched once: non-temporal, so they do not evict the table from L2
			float4 ax = nt_load4(ix4 + q), ay = nt_load4(iy4 + q), az = nt_load4(iz4 + q),
			       bx = nt_load4(ox4 + q), by = nt_load4(oy4 + q), bz = nt_load4(oz4 + q);
			nt_load_wait6(ax, ay, az, bx, by, bz);
			ixs[0] = ax.x; ixs[1] = ax.y; ixs[2] = ax.z; ixs[3] = ax.w;
			iys[0] = ay.x; iys[1] = ay.y; iys[2] = ay.z; iys[3] = ay.w;
			izs[0] = az.x; izs[1] = az.y; izs[2] = az.z; izs[3] = az.w;
			oxs[0] = bx.x; oxs[1] = bx.y; oxs[2] = bx.z; oxs[3] = bx.w;
			oys[0] = by.x; oys[1] = by.y; oys[2] = by.z; oys[3] = by.w;
			ozs[0] = bz.x; ozs[1] = bz.y; ozs[2] = bz.z; ozs[3] = bz.w;
			int idx[4];
#pragma unroll
			for (int j = 0; j < 4; ++j)
				amb[j] = !merl_index_fast(mk(ixs[j], iys[j], izs[j]), mk(oxs[j], oys[j], ozs[j]), g, idx[j]);
			if (WANT & 3) {
				MerlTexel t[4];
#pragma unroll
				for (int j = 0; j < 4; ++j) t[j] = b.merl[amb[j] ? 0 : idx[j]];
				float r[4], gg[4], bb[4];
#pragma unroll
				for (int j = 0; j < 4; ++j) {
					float s = (WANT & 2) ? izs[j] : 1.0f;                      // evalp = eval * i.z
					r[j] = (WANT & 2) ? s * t[j].x : t[j].x;
					gg[j] = (WANT & 2) ? s * t[j].y : t[j].y;
					bb[j] = (WANT & 2) ? s * t[j].z : t[j].z;
				}
				// ambiguous pairs get a placeholder here; tier 2 overwrites it
				nt_store4(r[0], r[1], r[2], r[3], (float4 *)vout.x + q);
				nt_store4(gg[0], gg[1], gg[2], gg[3], (float4 *)vout.y + q);
				nt_store4(bb[0], bb[1], bb[2], bb[3], (float4 *)vout.z + q);
			}
			if (WANT & 4)
				((float4 *)out_pdf)[q] = make_float4(F(D(izs[0]) / DJB_PI), F(D(izs[1]) / DJB_PI),
				                                     F(D(izs[2]) / DJB_PI), F(D(izs[3]) / DJB_PI));
		}
		if (__ballot(amb[0] | amb[1] | amb[2] | amb[3])) {
#pragma unroll
			for (int j = 0; j < 4; ++j)
				merl_queue(wbuf[wave], qn, lane, amb[j], (unsigned int)(4 * q + j), mk(ixs[j], iys[j], izs[j]), mk(oxs[j], oys[j], ozs[j]));
		}
		while (qn >= 64u || (last && qn)) {                           // tier 2: a full wave of waiting pairs, or what is left at the end
			const unsigned int cnt = qn < 64u ? qn : 64u;
			qn -= cnt;
			merl_drain<WANT>(b, wbuf[wave], qn, cnt, lane, 0, vout, out_pdf);
		}
		if (last) break;
	}
}

// ---- calibration: how far is the fp32 estimate from the reference's own value, in units of its
// guard band?  stats: [0..2] max ratio (theta_h, theta_d, phi_d) as float bits; counters follow.
__global__ __launch_bounds__(BLOCK) void k_merl_guard_stats(long long n, View vi, View vo, MerlGuard g,
                                                            unsigned int *max_bits,
                                                            unsigned long long *counters)
{
	long long stride = (long long)gridDim.x * BLOCK;
	float mh = 0, md = 0, mp = 0;
	unsigned long long n_special = 0, n_amb = 0, n_mismatch = 0, n_sure = 0;
	for (long long k = (long long)blockIdx.x * BLOCK + threadIdx.x; k < n; k += stride) {
		v3 i = load3(vi, k), o = load3(vo, k);
		MerlFast f = merl_fast_coords(i, o, g);
		int idx_fast;
		bool sure = merl_index_fast(i, o, g, idx_fast);
		float th, td, pd;
		merl_angles_exact(i, o, th, td, pd);
		int idx_ref = phi_diff_index(pd) + theta_diff_index(td) * 180 + theta_half_index(th) * 16200;
		if (f.special) { ++n_special; continue; }
		// the reference's continuous coordinates (dj_brdf.h:906-957)
		float Th = D(th) <= 0.0 ? 0.0f : sqrtf(F((D(th) / (DJB_PI / 2.0)) * 90) * 90.0f);
		float Xd = F(D(td) / (DJB_PI * 0.5) * 90);
		float pw = D(pd) < 0.0 ? F(D(pd) + DJB_PI) : pd;
		float Xp = F(D(pw) / DJB_PI * 360 / 2);
		float dh = fabsf(f.t_h - Th), dd = fabsf(f.x_d - Xd), dp = fabsf(f.x_p - Xp);
		dp = fminf(dp, 180.0f - dp);
		if (f.m_h < 0.45f) mh = fmaxf(mh, dh / f.m_h);
		if (f.m_d < 0.45f) { md = fmaxf(md, dd / f.m_d); mp = fmaxf(mp, dp / f.m_p); }
		if (sure) { ++n_sure; if (idx_fast != idx_ref) ++n_mismatch; } else ++n_amb;
	}
	atomicMax(&max_bits[0], __float_as_uint(mh));
	atomicMax(&max_bits[1], __float_as_uint(md));
	atomicMax(&max_bits[2], __float_as_uint(mp));
	atomicAdd(&counters[0], n_special); atomicAdd(&counters[1], n_amb);
	atomicAdd(&counters[2], n_mismatch); atomicAdd(&counters[3], n_sure);
}

// ---- directed search: every lane hill-climbs ONE pair over the bit patterns of its six input floats, maximising
// |fp32 estimate - the reference's own value| / guard band (the quantity k_merl_guard_stats samples).  A move adds
// +-2^e units in the last place (e = 0..20, hash-drawn) to one coordinate; it is kept if the ratio grows.  Every
// evaluated pair that tier 1 calls CERTAIN is also checked against the exact index: counters[1] must stay 0.
DJB_DEV float merl_guard_ratio(v3 i, v3 o, const MerlGuard g, unsigned long long &n_mismatch)
{
	MerlFast f = merl_fast_coords(i, o, g);
	int idx_fast;
	const bool sure = merl_index_fast(i, o, g, idx_fast);
	float th, td, pd;
	merl_angles_exact(i, o, th, td, pd);
	if (sure) {
		const int idx_ref = phi_diff_index(pd) + theta_diff_index(td) * 180 + theta_half_index(th) * 16200;
		if (idx_fast != idx_ref) ++n_mismatch;
	}
	if (f.special) return 0.0f;
	float Th = D(th) <= 0.0 ? 0.0f : sqrtf(F((D(th) / (DJB_PI / 2.0)) * 90) * 90.0f);
	float Xd = F(D(td) / (DJB_PI * 0.5) * 90);
	float pw = D(pd) < 0.0 ? F(D(pd) + DJB_PI) : pd;
	float Xp = F(D(pw) / DJB_PI * 360 / 2);
	float dh = fabsf(f.t_h - Th), dd = fabsf(f.x_d - Xd), dp = fabsf(f.x_p - Xp);
	dp = fminf(dp, 180.0f - dp);
	float r = 0.0f;
	if (f.m_h < 0.45f) r = fmaxf(r, dh / f.m_h);
	if (f.m_d < 0.45f) { r = fmaxf(r, dd / f.m_d); r = fmaxf(r, dp / f.m_p); }
	return r == r ? r : 0.0f;
}
__global__ __launch_bounds__(BLOCK) void k_merl_guard_attack(long long n, View vi, View vo, MerlGuard g, int iters, uint32_t seed,
                                                             float *best, unsigned long long *counters)
{
	const long long stride = (long long)gridDim.x * BLOCK;
	unsigned long long n_eval = 0, n_mis = 0, n_acc = 0;
	for (long long k = (long long)blockIdx.x * BLOCK + threadIdx.x; k < n; k += stride) {
		float c[6];
		{ v3 i = load3(vi, k), o = load3(vo, k); c[0] = i.x; c[1] = i.y; c[2] = i.z; c[3] = o.x; c[4] = o.y; c[5] = o.z; }
		float cur = merl_guard_ratio(mk(c[0], c[1], c[2]), mk(c[3], c[4], c[5]), g, n_mis);
		++n_eval;
		for (int it = 0; it < iters; ++it) {
			const uint32_t h = hash_u32(seed, (uint64_t)k * 4096ull + (uint64_t)it, 7u);
			const int w = (int)(h % 6u), e = (int)((h >> 3) % 21u);
			const int delta = (h & 0x80000000u) ? (1 << e) : -(1 << e);
			const float old = c[w];
			const float cand = __uint_as_float(__float_as_uint(old) + (uint32_t)delta);       // +-2^e ulps (sign-magnitude walk)
			if (!(fabsf(cand) < 16.0f)) continue;                                            // stay finite and near the family
			c[w] = cand;
			const float r = merl_guard_ratio(mk(c[0], c[1], c[2]), mk(c[3], c[4], c[5]), g, n_mis);
			++n_eval;
			if (r > cur) { cur = r; ++n_acc; } else c[w] = old;
		}
		store3(vi, k, mk(c[0], c[1], c[2])); store3(vo, k, mk(c[3], c[4], c[5]));
		best[k] = cur;
	}
	atomicAdd(&counters[0], n_eval); atomicAdd(&counters[1], n_mis); atomicAdd(&counters[2], n_acc);
}

template <int WANT>
hipError_t launch_tt(hipStream_t s, const Brdf &b, long long n, const View &i, const View &o,
                     const View &out, float *out_pdf, const MerlGuard &g)
{
	auto al16 = [](const void *p) { return ((uintptr_t)p & 15) == 0; };
	bool dense = i.stride == 1 && o.stride == 1 && (!(WANT & 3) || out.stride == 1) &&
	             al16(i.x) && al16(i.y) && al16(i.z) && al16(o.x) && al16(o.y) && al16(o.z) &&
	             (!(WANT & 3) || (al16(out.x) && al16(out.y) && al16(out.z))) && (!(WANT & 4) || al16(out_pdf));
	long long n4 = dense ? n / 4 : 0;
	// a batch too small to give every CU a workgroup of four-pair lanes runs one pair per lane: four times the waves, a quarter
	// of the dependent chain per wave -- 5.6-7.4 instead of 7.1-10.8 us per call below 2^17 pairs, equal at 2^18, slower above
	// (profiles/r04/merl_small_batches.txt)
	long long v4_min = 1LL << 18;
#ifdef DJB_EXPERIMENT
	if (const char *e = getenv("DJB_MERL_V4_MIN")) v4_min = atoll(e);
#endif
	if (n < v4_min) n4 = 0;
	long long gcap = DJB_MERL_GRID_CAP;
#ifdef DJB_EXPERIMENT
	if (const char *e = getenv("DJB_MERL_GRID_CAP_ENV")) gcap = atoll(e);
#endif
	if (n4 > 0)
		hipLaunchKernelGGL((k_merl_fast_v4<WANT>), dim3(grid_for(n4, gcap)), dim3(BLOCK), 0, s, b, n4, i, o, out, out_pdf, g);
	if (4 * n4 < n)   // strided / unaligned input, or the < 4-pair tail of a dense batch
		hipLaunchKernelGGL((k_merl_fast<WANT>), dim3(grid_for(n - 4 * n4)), dim3(BLOCK), 0, s, b, 4 * n4, n, i, o, out, out_pdf, g);
	return hipGetLastError();
}

} // namespace

namespace djbk {

hipError_t launch_merl_twotier(hipStream_t s, const Brdf &b, long long n, const View &i, const View &o,
                               const View &out, float *out_pdf, int want)
{
	if (n <= 0) return hipSuccess;
	const MerlGuard g = MERL_GUARD_DEFAULT;
	switch (want) {
	case 1: return launch_tt<1>(s, b, n, i, o, out, out_pdf, g);
	case 2: return launch_tt<2>(s, b, n, i, o, out, out_pdf, g);
	case 5: return launch_tt<5>(s, b, n, i, o, out, out_pdf, g);
	case 6: return launch_tt<6>(s, b, n, i, o, out, out_pdf, g);
	}
	return hipErrorInvalidValue;
}

hipError_t launch_merl_keys(hipStream_t s, long long n, const View &i, const View &o, uint32_t *keys)
{
	if (n <= 0) return hipSuccess;
	const MerlGuard g = MERL_GUARD_DEFAULT;
	hipLaunchKernelGGL(k_merl_keys, dim3(grid_for(n, 256LL * 32)), dim3(BLOCK), 0, s, n, i, o, keys, g);
	return hipGetLastError();
}

hipError_t launch_merl_guard_stats(hipStream_t s, long long n, const View &i, const View &o,
                                   const float *guard6, unsigned int *max_bits, unsigned long long *counters)
{
	MerlGuard g = MERL_GUARD_DEFAULT;
	if (guard6) { g.a_h = guard6[0]; g.b_h = guard6[1]; g.a_d = guard6[2]; g.b_d = guard6[3]; g.c_d = guard6[4]; g.a_p = guard6[5]; }
	hipLaunchKernelGGL(k_merl_guard_stats, dim3(grid_for(n)), dim3(BLOCK), 0, s, n, i, o, g, max_bits, counters);
	return hipGetLastError();
}

hipError_t launch_merl_guard_attack(hipStream_t s, long long n, const View &i, const View &o, const float *guard6, int iters, uint32_t seed,
                                    float *best, unsigned long long *counters)
{
	MerlGuard g = MERL_GUARD_DEFAULT;
	if (guard6) { g.a_h = guard6[0]; g.b_h = guard6[1]; g.a_d = guard6[2]; g.b_d = guard6[3]; g.c_d = guard6[4]; g.a_p = guard6[5]; }
	hipLaunchKernelGGL(k_merl_guard_attack, dim3(grid_for(n)), dim3(BLOCK), 0, s, n, i, o, g, iters, seed, best, counters);
	return hipGetLastError();
}

} // namespace djbk
