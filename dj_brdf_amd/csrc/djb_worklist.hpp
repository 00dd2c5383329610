// djb_worklist.hpp -- device helpers shared by the two-tier kernels (djb_kernels_merl.hip, djb_kernels_contract.hip):
// 16-byte non-temporal stream accesses and the per-wave LDS staging of the tier-2 worklist.
#pragma once
#include "djb_device.hpp"

namespace djbdev {

constexpr unsigned int WBUF = 128;   // per-wave LDS staging slots for worklist records (7 dwords each)

// 16-byte non-temporal accesses for the streams that are touched exactly once
typedef float nt_v4f __attribute__((ext_vector_type(4)));
DJB_DEV float4 nt_load4(const float4 *p)
{
	nt_v4f v = __builtin_nontemporal_load((const nt_v4f *)p);
	return make_float4(v.x, v.y, v.z, v.w);
}
DJB_DEV void nt_store4(float a, float b, float c, float d, float4 *p)
{
	nt_v4f v = { a, b, c, d };
	__builtin_nontemporal_store(v, (nt_v4f *)p);
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
