// djb_worklist.hpp -- device helpers shared by the two-tier kernels (djb_kernels_merl.hip, djb_kernels_contract.hip):
// 16-byte non-temporal stream accesses and the per-wave LDS staging of the tier-2 worklist.
#pragma once
#include "djb_device.hpp"

namespace djbdev {

constexpr unsigned int WBUF = 128;   // per-wave LDS staging slots for worklist records (7 dwords each)

// 16-byte accesses for the streams that are touched exactly once.  DJB_STREAM_LOAD_POLICY / DJB_STREAM_STORE_POLICY pick
// the cache-policy bits (0 = nt through the compiler's builtin; 1 = sc1, 2 = sc0 sc1, 3 = sc0 sc1 nt, 4 = plain: inline asm):
// what the streams leave behind in the XCD's L2 decides how much of it the table gathers keep (profiles/r04/merl_stream_policy.txt)
#ifndef DJB_STREAM_LOAD_POLICY
#define DJB_STREAM_LOAD_POLICY 0
#endif
#ifndef DJB_STREAM_STORE_POLICY
#define DJB_STREAM_STORE_POLICY 0
#endif
typedef float nt_v4f __attribute__((ext_vector_type(4)));
#if DJB_STREAM_LOAD_POLICY == 0
DJB_DEV float4 nt_load4(const float4 *p)
{
	nt_v4f v = __builtin_nontemporal_load((const nt_v4f *)p);
	return make_float4(v.x, v.y, v.z, v.w);
}
DJB_DEV void nt_load_wait6(float4 &, float4 &, float4 &, float4 &, float4 &, float4 &) {}
#else
DJB_DEV float4 nt_load4(const float4 *p)
{
	nt_v4f v;
#if DJB_STREAM_LOAD_POLICY == 1
	asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
#elif DJB_STREAM_LOAD_POLICY == 2
	asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
#elif DJB_STREAM_LOAD_POLICY == 3
	asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt" : "=v"(v) : "v"(p) : "memory");
#else
	asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
#endif
	return make_float4(v.x, v.y, v.z, v.w);
}
// the compiler does not count loads issued from inline asm: wait for them here, with the six results as operands so
// that no use can be scheduled ahead of the wait
DJB_DEV void nt_load_wait6(float4 &a, float4 &b, float4 &c, float4 &d, float4 &e, float4 &f)
{
	asm volatile("s_waitcnt vmcnt(0)" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w), "+v"(b.x), "+v"(b.y), "+v"(b.z), "+v"(b.w),
	             "+v"(c.x), "+v"(c.y), "+v"(c.z), "+v"(c.w) :: "memory");
	asm volatile("" : "+v"(d.x), "+v"(d.y), "+v"(d.z), "+v"(d.w), "+v"(e.x), "+v"(e.y), "+v"(e.z), "+v"(e.w),
	             "+v"(f.x), "+v"(f.y), "+v"(f.z), "+v"(f.w) :: "memory");
}
#endif
DJB_DEV void nt_store4(float a, float b, float c, float d, float4 *p)
{
	nt_v4f v = { a, b, c, d };
#if DJB_STREAM_STORE_POLICY == 0
	__builtin_nontemporal_store(v, (nt_v4f *)p);
#elif DJB_STREAM_STORE_POLICY == 1
	asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
#elif DJB_STREAM_STORE_POLICY == 2
	asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
#elif DJB_STREAM_STORE_POLICY == 3
	asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
#else
	asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
#endif
}

// ---- per-wave worklist staging.  Ambiguous pairs are staged per wave in LDS (no barrier needed: one
// wave, in-order LDS) and flushed with ONE global atomic per flush: a returning atomic per ambiguous
// lane (~8e6 per 1e9 pairs on one address) costs more than the whole kernel.  A record is
// {k, i.xyz, o.xyz, pad} = two uint4, so the fix-up kernel streams its inputs instead of gathering them.
typedef unsigned int WaveBuf[7][WBUF];

DJB_DEV void wl_flush(WaveBuf &wb, unsigned int &wcount, int lane, uint4 *list, unsigned int cap,
                      unsigned int *count)
{
	unsigned int base = 0;
	if (lane == 0) base = atomicAdd(count, wcount);
	base = __shfl(base, 0);
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
	for (unsigned int j = lane; j < wcount; j += 64)
		if (base + j < cap) {                                    // beyond cap: fix-up kernel rescans
			list[2 * (size_t)(base + j)] = make_uint4(wb[0][j], wb[1][j], wb[2][j], wb[3][j]);
			list[2 * (size_t)(base + j) + 1] = make_uint4(wb[4][j], wb[5][j], wb[6][j], 0u);
		}
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
	wcount = 0;
}

DJB_DEV void wl_push(WaveBuf &wb, unsigned int &wcount, int lane, uint4 *list, unsigned int cap,
                     unsigned int *count, bool amb, unsigned int k, v3 i, v3 o)
{
	unsigned long long mask = __ballot(amb);
	if (!mask) return;
	unsigned int c = (unsigned int)__popcll(mask);
	if (wcount + c > WBUF) wl_flush(wb, wcount, lane, list, cap, count);
	if (amb) {
		unsigned int slot = wcount + (unsigned int)__popcll(mask & ((1ull << lane) - 1ull));
		wb[0][slot] = k;
		wb[1][slot] = __float_as_uint(i.x); wb[2][slot] = __float_as_uint(i.y); wb[3][slot] = __float_as_uint(i.z);
		wb[4][slot] = __float_as_uint(o.x); wb[5][slot] = __float_as_uint(o.y); wb[6][slot] = __float_as_uint(o.z);
	}
	wcount += c;
}

} // namespace djbdev
