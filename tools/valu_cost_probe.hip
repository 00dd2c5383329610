// valu_cost_probe.hip -- issue cost of the VALU instructions the bit-exact kernels are made of, on gfx950.
//   hipcc --offload-arch=gfx950 -O2 -o tools/bin/valu_cost_probe tools/valu_cost_probe.hip && tools/bin/valu_cost_probe
// Every probe runs 8 independent dependency chains per lane (latency hidden), 4 waves per SIMD on every CU, and reports
// the time per wave-instruction relative to v_mul_f32 (= 1 issue slot).  What the kernels' "VALU instructions per unit"
// counts hide: fp64 transcendentals (v_rsq_f64, v_rcp_f64, v_sqrt_f64) and the integer multiply cost several slots.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <string>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int REP = 16;      // instructions per chain per loop iteration
constexpr int CHAINS = 8;

#define R16(X) X X X X X X X X X X X X X X X X

// 32-bit destination / accumulator %0..%7 ("+v"), second source %8
#define PROBE32(NAME, ASM)                                                                              \
__global__ __launch_bounds__(256) void NAME(float *out, int iters, float seed)                          \
{                                                                                                       \
	float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
	float c = 1.0000001f;                                                                               \
	for (int it = 0; it < iters; ++it) {                                                                \
		R16(asm volatile(ASM(%0) "\n" ASM(%1) "\n" ASM(%2) "\n" ASM(%3) "\n" ASM(%4) "\n" ASM(%5) "\n" ASM(%6) "\n" ASM(%7)    \
		                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "vcc");) \
	}                                                                                                   \
	out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                         \
}
#define PROBE64(NAME, ASM)                                                                              \
__global__ __launch_bounds__(256) void NAME(float *out, int iters, float seed)                          \
{                                                                                                       \
	double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
	double c = 1.0000001;                                                                               \
	for (int it = 0; it < iters; ++it) {                                                                \
		R16(asm volatile(ASM(%0) "\n" ASM(%1) "\n" ASM(%2) "\n" ASM(%3) "\n" ASM(%4) "\n" ASM(%5) "\n" ASM(%6) "\n" ASM(%7)    \
		                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "vcc");) \
	}                                                                                                   \
	out[blockIdx.x * 256 + threadIdx.x] = (float)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);              \
}
// 64 -> 32 and 32 -> 64 conversions need both register classes: a pair of accumulators per chain
#define PROBEMIX(NAME, ASM)                                                                             \
__global__ __launch_bounds__(256) void NAME(float *out, int iters, float seed)                          \
{                                                                                                       \
	double d0 = seed + threadIdx.x, d1 = d0 + 1, d2 = d0 + 2, d3 = d0 + 3;                              \
	float f0 = seed, f1 = seed + 1, f2 = seed + 2, f3 = seed + 3;                                        \
	for (int it = 0; it < iters; ++it) {                                                                \
		R16(asm volatile(ASM(%0, %4) "\n" ASM(%1, %5) "\n" ASM(%2, %6) "\n" ASM(%3, %7) "\n" ASM(%0, %4) "\n" ASM(%1, %5) "\n" ASM(%2, %6) "\n" ASM(%3, %7) \
		                 : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : : "vcc");) \
	}                                                                                                   \
	out[blockIdx.x * 256 + threadIdx.x] = (float)(d0 + d1 + d2 + d3) + f0 + f1 + f2 + f3;               \
}

// v_cndmask with its mask in an SGPR pair (%9) instead of vcc, and a cmp + cndmask pair as the compiler emits selects
#define PROBE32S(NAME, ASM)                                                                             \
__global__ __launch_bounds__(256) void NAME(float *out, int iters, float seed)                          \
{                                                                                                       \
	float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
	float c = 1.0000001f;                                                                               \
	unsigned long long m = 0x5555555555555555ull + (unsigned long long)iters;                           \
	for (int it = 0; it < iters; ++it) {                                                                \
		R16(asm volatile(ASM(%0) "\n" ASM(%1) "\n" ASM(%2) "\n" ASM(%3) "\n" ASM(%4) "\n" ASM(%5) "\n" ASM(%6) "\n" ASM(%7)    \
		                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "s"(m) : "vcc");) \
	}                                                                                                   \
	out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                         \
}
#define A_CNDMASK_S(r) "v_cndmask_b32_e64 " #r ", " #r ", %8, %9"
#define A_CMP_CND(r) "v_cmp_lt_f32 vcc, " #r ", %8\n s_nop 1\n v_cndmask_b32 " #r ", " #r ", %8, vcc"
#define A_CMP_S(r) "v_cmp_lt_f32_e64 %9, " #r ", %8"
#define A_MAX_F32(r) "v_max_f32 " #r ", " #r ", %8"
#define A_MUL_F32(r) "v_mul_f32 " #r ", " #r ", %8"
#define A_ADD_F32(r) "v_add_f32 " #r ", " #r ", %8"
#define A_FMA_F32(r) "v_fma_f32 " #r ", " #r ", %8, %8"
#define A_RCP_F32(r) "v_rcp_f32 " #r ", " #r
#define A_RSQ_F32(r) "v_rsq_f32 " #r ", " #r
#define A_SQRT_F32(r) "v_sqrt_f32 " #r ", " #r
#define A_EXP_F32(r) "v_exp_f32 " #r ", " #r
#define A_LOG_F32(r) "v_log_f32 " #r ", " #r
#define A_DIVSCALE_F32(r) "v_div_scale_f32 " #r ", vcc, " #r ", %8, " #r
#define A_DIVFMAS_F32(r) "v_div_fmas_f32 " #r ", " #r ", %8, %8"
#define A_DIVFIXUP_F32(r) "v_div_fixup_f32 " #r ", " #r ", %8, %8"
#define A_LDEXP_F32(r) "v_ldexp_f32 " #r ", " #r ", 1"
#define A_CVT_F32_U32(r) "v_cvt_f32_u32 " #r ", " #r
#define A_CNDMASK(r) "v_cndmask_b32 " #r ", " #r ", %8, vcc"
#define A_CMP(r) "v_cmp_lt_f32 vcc, " #r ", %8"
#define A_MUL_LO_U32(r) "v_mul_lo_u32 " #r ", " #r ", %8"
#define A_MUL_HI_U32(r) "v_mul_hi_u32 " #r ", " #r ", %8"
#define A_MUL_U24(r) "v_mul_u32_u24 " #r ", " #r ", %8"
#define A_MAD_U24(r) "v_mad_u32_u24 " #r ", " #r ", %8, %8"
#define A_LSHL(r) "v_lshlrev_b32 " #r ", 3, " #r
#define A_ADD_U32(r) "v_add_u32 " #r ", " #r ", %8"
#define A_XOR(r) "v_xor_b32 " #r ", " #r ", %8"
#define A_MOV(r) "v_mov_b32 " #r ", %8"
#define A_BFE(r) "v_bfe_u32 " #r ", " #r ", 3, 7"
#define A_MBCNT(r) "v_mbcnt_lo_u32_b32 " #r ", -1, " #r

#define A_MUL_F64(r) "v_mul_f64 " #r ", " #r ", %8"
#define A_ADD_F64(r) "v_add_f64 " #r ", " #r ", %8"
#define A_FMA_F64(r) "v_fma_f64 " #r ", " #r ", %8, %8"
#define A_RCP_F64(r) "v_rcp_f64 " #r ", " #r
#define A_RSQ_F64(r) "v_rsq_f64 " #r ", " #r
#define A_SQRT_F64(r) "v_sqrt_f64 " #r ", " #r
#define A_LDEXP_F64(r) "v_ldexp_f64 " #r ", " #r ", 1"
#define A_DIVSCALE_F64(r) "v_div_scale_f64 " #r ", vcc, " #r ", %8, " #r
#define A_DIVFMAS_F64(r) "v_div_fmas_f64 " #r ", " #r ", %8, %8"
#define A_DIVFIXUP_F64(r) "v_div_fixup_f64 " #r ", " #r ", %8, %8"
#define A_PK_MUL_F32(r) "v_pk_mul_f32 " #r ", " #r ", %8"
#define A_PK_FMA_F32(r) "v_pk_fma_f32 " #r ", " #r ", %8, %8"
#define A_PK_ADD_F32(r) "v_pk_add_f32 " #r ", " #r ", %8"
#define A_LSHL_B64(r) "v_lshlrev_b64 " #r ", 3, " #r
#define A_LSHL_ADD_U64(r) "v_lshl_add_u64 " #r ", " #r ", 1, %8"
#define A_MOV_B64(r) "v_mov_b64 " #r ", %8"
#define A_CMP_F64(r) "v_cmp_lt_f64 vcc, " #r ", %8"
#define A_FREXP_F64(r) "v_frexp_mant_f64 " #r ", " #r

#define A_CVT_F64_F32(d, f) "v_cvt_f64_f32 " #d ", " #f
#define A_CVT_F32_F64(d, f) "v_cvt_f32_f64 " #f ", " #d
#define A_CVT_F64_I32(d, f) "v_cvt_f64_i32 " #d ", " #f
#define A_CVT_I32_F64(d, f) "v_cvt_i32_f64 " #f ", " #d

PROBE32(p_mul_f32, A_MUL_F32) PROBE32(p_add_f32, A_ADD_F32) PROBE32(p_fma_f32, A_FMA_F32) PROBE32(p_rcp_f32, A_RCP_F32)
PROBE32(p_rsq_f32, A_RSQ_F32) PROBE32(p_sqrt_f32, A_SQRT_F32) PROBE32(p_exp_f32, A_EXP_F32) PROBE32(p_log_f32, A_LOG_F32)
PROBE32(p_divscale_f32, A_DIVSCALE_F32) PROBE32(p_divfmas_f32, A_DIVFMAS_F32) PROBE32(p_divfixup_f32, A_DIVFIXUP_F32)
PROBE32(p_ldexp_f32, A_LDEXP_F32) PROBE32(p_cvt_f32_u32, A_CVT_F32_U32) PROBE32(p_cndmask, A_CNDMASK) PROBE32(p_cmp, A_CMP)
PROBE32(p_mul_lo_u32, A_MUL_LO_U32) PROBE32(p_mul_hi_u32, A_MUL_HI_U32) PROBE32(p_mul_u24, A_MUL_U24) PROBE32(p_mad_u24, A_MAD_U24)
PROBE32(p_lshl, A_LSHL) PROBE32(p_add_u32, A_ADD_U32) PROBE32(p_xor, A_XOR) PROBE32(p_mov, A_MOV) PROBE32(p_bfe, A_BFE) PROBE32(p_mbcnt, A_MBCNT)
PROBE32S(p_cndmask_sgpr, A_CNDMASK_S) PROBE32(p_cmp_cnd_pair, A_CMP_CND) PROBE32(p_max_f32, A_MAX_F32)
PROBE64(p_mul_f64, A_MUL_F64) PROBE64(p_add_f64, A_ADD_F64) PROBE64(p_fma_f64, A_FMA_F64) PROBE64(p_rcp_f64, A_RCP_F64)
PROBE64(p_rsq_f64, A_RSQ_F64) PROBE64(p_sqrt_f64, A_SQRT_F64) PROBE64(p_ldexp_f64, A_LDEXP_F64) PROBE64(p_divscale_f64, A_DIVSCALE_F64)
PROBE64(p_divfmas_f64, A_DIVFMAS_F64) PROBE64(p_divfixup_f64, A_DIVFIXUP_F64) PROBE64(p_pk_mul_f32, A_PK_MUL_F32) PROBE64(p_pk_fma_f32, A_PK_FMA_F32)
PROBE64(p_pk_add_f32, A_PK_ADD_F32) PROBE64(p_lshl_b64, A_LSHL_B64) PROBE64(p_lshl_add_u64, A_LSHL_ADD_U64) PROBE64(p_mov_b64, A_MOV_B64)
PROBE64(p_cmp_f64, A_CMP_F64) PROBE64(p_frexp_f64, A_FREXP_F64)
PROBEMIX(p_cvt_f64_f32, A_CVT_F64_F32) PROBEMIX(p_cvt_f32_f64, A_CVT_F32_F64) PROBEMIX(p_cvt_f64_i32, A_CVT_F64_I32) PROBEMIX(p_cvt_i32_f64, A_CVT_I32_F64)

typedef void (*kern_t)(float *, int, float);
struct Probe { const char *name; kern_t k; };

int main()
{
	std::vector<Probe> P = {
#define E(n) { #n, p_##n }
		E(mul_f32), E(add_f32), E(fma_f32), E(mov), E(max_f32), E(cndmask), E(cndmask_sgpr), E(cmp_cnd_pair), E(cmp), E(add_u32), E(xor), E(lshl), E(bfe), E(mbcnt), E(ldexp_f32), E(cvt_f32_u32),
		E(mul_u24), E(mad_u24), E(mul_lo_u32), E(mul_hi_u32),
		E(rcp_f32), E(rsq_f32), E(sqrt_f32), E(exp_f32), E(log_f32), E(divscale_f32), E(divfmas_f32), E(divfixup_f32),
		E(pk_mul_f32), E(pk_add_f32), E(pk_fma_f32),
		E(mul_f64), E(add_f64), E(fma_f64), E(ldexp_f64), E(frexp_f64), E(cmp_f64), E(mov_b64), E(lshl_b64), E(lshl_add_u64),
		E(cvt_f64_f32), E(cvt_f32_f64), E(cvt_f64_i32), E(cvt_i32_f64),
		E(rcp_f64), E(rsq_f64), E(sqrt_f64), E(divscale_f64), E(divfmas_f64), E(divfixup_f64),
	};
	hipDeviceProp_t prop; CHK(hipGetDeviceProperties(&prop, 0));
	const int cus = prop.multiProcessorCount, blocks = cus * 4;    // 4 workgroups of 4 waves per CU = 4 waves per SIMD
	float *out; CHK(hipMalloc(&out, sizeof(float) * blocks * 256));
	hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
	const int iters = 2000;
	double base = 0;
	printf("# %s, %d CUs, clock %d MHz; %d workgroups x 256 threads, %d chains x %d instructions x %d iterations per lane\n", prop.gcnArchName, cus,
	       prop.clockRate / 1000, blocks, CHAINS, REP, iters);
	printf("# %-16s %10s %14s %8s\n", "instruction", "ms", "ns/wave-instr", "slots");
	for (size_t i = 0; i < P.size(); ++i) {
		for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(P[i].k, dim3(blocks), dim3(256), 0, 0, out, iters / 4, 1.0f);
		CHK(hipDeviceSynchronize());
		float best = 1e30f;
		for (int rep = 0; rep < 3; ++rep) {
			CHK(hipEventRecord(e0));
			hipLaunchKernelGGL(P[i].k, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
			CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
			float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
			if (ms < best) best = ms;
		}
		// wave-instructions per SIMD: 4 waves x iters x REP x CHAINS
		const double per = best * 1e6 / (4.0 * iters * REP * CHAINS);
		if (i == 0) base = per;
		printf("  %-16s %10.3f %14.3f %8.2f\n", P[i].name, best, per, per / base);
	}
	return 0;
}
