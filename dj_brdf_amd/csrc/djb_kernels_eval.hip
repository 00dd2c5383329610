// djb_kernels_eval.hip -- batch eval / evalp / pdf / sample / evalp_is kernels for gfx950.
//
// One (i, o) pair per lane, grid-stride over the batch; directions stream from HBM as
// coalesced dword loads (SoA, stride 1) and results stream back the same way.  The BRDF
// object and the microfacet parameters arrive as kernel arguments (SGPRs).  There is no
// inter-lane communication: the path is elementwise (+ one cache-resident gather for MERL/UTIA).
#include "djb_internal.hpp"

using namespace djbdev;

namespace {

constexpr int BLOCK = 256;

inline int grid_for(long long n)
{
	long long blocks = (n + BLOCK - 1) / BLOCK;
	const long long cap = 256LL * 16;   // 256 CUs x 16 resident workgroups' worth, grid-stride beyond
	if (blocks > cap) blocks = cap;
	if (blocks < 1) blocks = 1;
	return (int)blocks;
}

// The VALU-bound kernels of the analytic lobes (eval, eval_pp, sample of beckmann / ggx) get one
// workgroup per 256 units: the hardware dispatcher balances the data-dependent work (Beckmann's
// Newton loop, dead lanes) better than a persistent grid-stride grid does -- measured -3 % (GGX
// eval+pdf) and -8 % (Beckmann sample).  The table-driven kinds are faster on the persistent grid
// (tabular eval: 2.25 vs 3.86 ms per 1e8).  The loop stays for batches beyond 2^31 - 1 workgroups.
inline int grid_full(long long n)
{
	long long blocks = (n + BLOCK - 1) / BLOCK;
	if (blocks > 0x7fffffffLL) blocks = 0x7fffffffLL;
	if (blocks < 1) blocks = 1;
	return (int)blocks;
}

// load3_dense / store3_dense (djb_device_units.inc): dense batches address their arrays as uniform base + lane offset
inline bool dense(const View &v) { return v.stride == 1 || v.x == nullptr; }

// min-waves hint per kind, measured (round 2; ms per 1e8 pairs at 1 / 4 / 8 waves; the rates themselves are bench.py legs now): the analytic /
// tabulated microfacet kernels fit 128 VGPRs (4); utia 3.34 / 2.93 / 10.5 and sgd 4.46 / 4.17 / 8.3 want 4
// (left alone they take 172 VGPRs = 2 waves, too few to hide the table gathers; at 8 they spill);
// abc 1.58 / 1.34 / 1.30 is light enough for 8; the operation-by-operation merl kernel stays unconstrained.
#ifndef DJB_UTIA_MIN_WAVES
#define DJB_UTIA_MIN_WAVES 4
#endif
constexpr int eval_min_waves(int kind)
{
	return kind == KIND_UTIA ? DJB_UTIA_MIN_WAVES
	     : (kind <= KIND_TABULAR || kind == KIND_TABULAR_ANISO || kind == KIND_SGD) ? 4
	     : kind == KIND_ABC ? 8 : 1;
}
// workgroup size: 1024 for tabular_anisotropic -- its two elev x azim grids (2 x 32 KB at 90 x 90) are staged in LDS, and sixteen waves
// sharing one copy keep two workgroups = 8 waves per SIMD resident (256-thread workgroups with 65 KB each: 2 per SIMD, 3.7 ms per 1e8 pairs
// against 2.8 with sigma's grid alone and 3.06 without LDS)
constexpr int eval_block(int kind) { return kind == KIND_TABULAR_ANISO ? 1024 : BLOCK; }
template <int KIND, int WANT, int FRK, bool DENSE>
__global__ __launch_bounds__(eval_block(KIND), eval_min_waves(KIND)) void k_eval(Brdf b, Params p, long long n, View vi, View vo,
                                                   View vout, float *out_pdf)
{
	constexpr int BS = eval_block(KIND);
	// Beckmann evaluates the fp64 exp of glibc (djb_device.hpp) three times per pair: its 2 KB table goes to LDS
	// (sgd: 9 exp + 9 pow per pair, abc: one pow -- both tables)
	// (sgd also calls glibc's acos twice per pair: its 21 KB of tables)
	// (round 6: sgd's polar angles come from the arctangent core below; glibc's acos -- its 21 KB of tables stay in global memory -- only
	// answers for the units the decided fast tier leaves, djb_fast_models.inc; a row outside that tier's domain pays for it: 5.2 ms per 1e8
	// pairs instead of 4.0.  Occupancy: 87 VGPRs = 5 waves per SIMD; 6 (16 B of scratch): the same time, 8 (80 B): 3.59 against 3.12 ms)
	constexpr bool EXPT = KIND == KIND_BECKMANN || KIND == KIND_SGD || KIND == KIND_ABC, POWT = KIND == KIND_SGD || KIND == KIND_ABC,
	               ACOST = false;
	__shared__ unsigned long long s_exp[EXPT ? 256 : 1];
	__shared__ double s_pow[POWT ? 384 : 1];
	__shared__ double s_acos[ACOST ? 2568 + 128 : 1];
	if (EXPT) b.exp_lds = glibc_exp_tab_to_lds(s_exp, threadIdx.x, BS);
	if (POWT) b.pow_lds = glibc_pow_tab_to_lds(s_pow, threadIdx.x, BS);
	if (ACOST) b.acos_lds = glibc_acos_tab_to_lds(s_acos, threadIdx.x, BS);
	// the tabulated lobes' table coordinates (acos / atan / atan2 of a float, rounded to float) from the arctangent core (djb_device.hpp)
	constexpr bool ATANT = KIND == KIND_TABULAR || KIND == KIND_TABULAR_ANISO || KIND == KIND_SGD || FRK == FR_SPLINE;      // ... and the Fresnel spline's (dj_brdf.h:1341)
	__shared__ double s_atan[ATANT ? 16 : 1];
	b.atan_lds = ATANT ? atan_tab_to_lds(s_atan, threadIdx.x) : 0u;
	// a fitted lobe's tables in LDS (north star: "LDS-staged tiles of the tables"): the slope-pdf and sigma tables of tabular (float[res]
	// each), the sigma and slope-pdf grids of tabular_anisotropic (elev x azim each), the Fresnel spline's points -- 12-20
	// divergent 4-byte look-ups per pair otherwise go through the texture path.  Staged when they fit TAB_LDS floats; the per-unit code
	// reads through the same pointers (the address space is inferred after inlining)
	constexpr int TAB_LDS = KIND == KIND_TABULAR ? 3072 : KIND == KIND_TABULAR_ANISO ? 16384 + 768 : FRK == FR_SPLINE ? 768 : 0;
	__shared__ float s_tab[TAB_LDS ? TAB_LDS : 1];
	if (TAB_LDS) {
		int used = 0;
		auto stage = [&](const float *&src, int count) {
			if (src == nullptr || count <= 0 || used + count > TAB_LDS) return;
			float *dst = s_tab + used;
			for (int k = threadIdx.x; k < count; k += BS) dst[k] = src[k];
			src = dst; used += count;
		};
		if (KIND == KIND_TABULAR) { stage(b.p22, b.n_p22); stage(b.sigma, b.n_sigma); }
		if (KIND == KIND_TABULAR_ANISO) { stage(b.sigma, b.elev * b.azim); stage(b.p22, b.elev * b.azim); }
		if (FRK == FR_SPLINE) stage(b.fr.pts, 3 * b.fr.npts);
	}
	if (EXPT || POWT || ATANT || TAB_LDS) __syncthreads();
	const long long stride = (long long)gridDim.x * BS;
	const unsigned int t = threadIdx.x;
	constexpr bool STDK = KIND == KIND_BECKMANN || KIND == KIND_GGX || KIND == KIND_TABULAR;     // same-box A/B: GGX eval+pdf 1.060 -> 1.051 ms per 1e8, tabular 1.804 -> 1.769
	const bool std_frame = STDK && p.rho == 0.0f && p.s == 1.0f && p.tx == 0.0f && p.ty == 0.0f && p.nx == 0.0f && p.ny == 0.0f && p.nz == 1.0f;
	for (long long k0 = (long long)blockIdx.x * BS; k0 < n; k0 += stride) {     // k0: workgroup-uniform
		const long long k = k0 + t;
		// SOFF (Beckmann, GGX, tabular; same-box -1.3 % each; abc was 5 % slower with it): the tile's live lanes from a scalar bound and the
		// accesses as SGPR base + 32-bit lane offset (djb_device_units.inc: lane_byte_offset)
		constexpr bool SOFF = DENSE && KIND <= KIND_TABULAR;
		if (SOFF) { const unsigned int rem = n - k0 >= (long long)BS ? (unsigned int)BS : (unsigned int)(n - k0); if (t >= rem) continue; }
		else if (k >= n) continue;
		const unsigned int toff = SOFF ? lane_byte_offset(t) : (t << 2);
		v3 i = DENSE ? load3_dense_off(vi, k0, toff) : load3(vi, k), o = DENSE ? load3_dense_off(vo, k0, toff) : load3(vo, k);
		v3 fr = mk(0, 0, 0); float pdf = 0.0f;
		if (STDK) {
			// a standard frame (launch-uniform) and finite x, y of both directions (per lane; v_max ignores a NaN, which both forms propagate):
			// the parameter arithmetic without its zero terms (djb_device_microfacet.inc, STD) -- the same floats
			const bool fin = fmaxf(fmaxf(fabsf(i.x), fabsf(i.y)), fmaxf(fabsf(o.x), fabsf(o.y))) < __builtin_inff();
			if (__builtin_expect(std_frame & fin, 1)) eval_one<KIND, WANT, FRK, true>(b, p, i, o, fr, pdf);
			else eval_one<KIND, WANT, FRK>(b, p, i, o, fr, pdf);
		} else eval_one<KIND, WANT, FRK>(b, p, i, o, fr, pdf);
		const unsigned int soff = SOFF ? lane_byte_offset(t) : (t << 2);           // again: the stores sit in another block than the loads
		if (WANT & 3) { if (DENSE) store3_dense_off(vout, k0, soff, fr); else store3(vout, k, fr); }
		if (WANT & 4) { if (DENSE) (*dense_off(out_pdf + k0, soff)) = pdf; else out_pdf[k] = pdf; }
	}
}

// ---- Beckmann eval / evalp / pdf of a SHARP lobe, bit-identical, two paths.  exp(-r^2) makes most pairs of a sharp lobe exact
// zeros -- float(exp(-r^2) / pi) IS zero from r^2 = 102.83 on (half of the smallest denormal), and r^2 = tan^2(theta_h) / alpha^2 passes 104
// at theta_h = 27 deg for alpha = 0.05 -- but k_eval pays the whole evaluation for them: both sigmas (an fp64 exp and an erf each), the
// shadowing term, the Fresnel term, ~430 VALU.  Here every lane computes only what decides that: h and the stretched slope radius, with
// the reference's own operations (the same functions eval_one calls, so r^2 is its r^2 bit for bit).  A pair is TRIVIAL when the
// reference's result is a zero that can be written down:
//   (a) o below the horizon (or i, with shadowing): g1 = 0, G = 0, eval and pdf keep their initial +0 (dj_brdf.h:1529-1555, 1633-1665,
//       1713-1730) whatever else the pair holds, NaNs included (0 * NaN is NaN, and `G > 0` is false for it); eval divides that
//       vec3(0) by i.z all the same (-0 for i below the horizon, NaN on it);
//   (b) both directions finite, |component| < 8, z > 1e-4, and h.z <= 1e-4 (ndf returns 0) or r^2 >= 104 (p22 is +0): then D = +0; G
//       is finite -- sigma >= k.z / 2 > 0, so g1 <= ~1, and g1i + g1o - g1i g1o > 0 -- or not positive, F is finite and not negative
//       (ideal; schlick with f0 in [0, 1]; unpolarized with ior > 1: beckmann_sharp_supported), so eval = evalp = (+0, +0, +0);
//       pdf = (+0) / (4 dot(i, h)) = +0 for dot(i, h) > 0 -- pairs with dot(i, h) <= 0 are not taken.
// Everything else is queued per wave in LDS and evaluated by eval_one in dense waves of 64, as the deferred samples of k_sample_bk are.
// Only launched for lobes sharp enough to make that pay (beckmann_sharp_supported): the prefix alone takes 0.76 ms per 1e8 pairs, a full
// pair about two thirds of what it takes in k_eval; eval + pdf of 1e8 bench pairs 1.87 -> 1.01 ms at alpha = 0.02 (91 % zeros), 1.89 -> 1.27 at
// 0.05 (59 %), 1.82 -> 1.57 at 0.08, even at 0.1, slower above (profiles/r04/beckmann_sharp.txt).
constexpr unsigned int SHQ = 128;       // queue slots per wave: < 64 waiting + <= 64 new per iteration
template <int WANT, int FRK, bool DENSE>
__global__ __launch_bounds__(BLOCK) void k_eval_bk_sharp(Brdf b, Params p, long long n, View vi, View vo, View vout, float *out_pdf)
{
	__shared__ unsigned long long s_exp[256];
	__shared__ unsigned int s_q[BLOCK / 64][8][SHQ];      // {k lo, k hi, i.xyz, o.xyz}
	b.exp_lds = glibc_exp_tab_to_lds(s_exp, threadIdx.x, BLOCK);
	__syncthreads();
	const unsigned int t = threadIdx.x, wave = t >> 6, lane = t & 63u;
	unsigned int (&q)[8][SHQ] = s_q[wave];
	unsigned int qn = 0;                                   // wave-uniform
	auto drain = [&](unsigned int first, unsigned int cnt) {
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
		__builtin_amdgcn_wave_barrier();
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the placeholders of these pairs were stored by other lanes of this wave: they leave first
		if (lane < cnt) {
			const unsigned int j = first + lane;
			const long long k = (long long)(((unsigned long long)q[1][j] << 32) | q[0][j]);
			const v3 i = mk(__uint_as_float(q[2][j]), __uint_as_float(q[3][j]), __uint_as_float(q[4][j]));
			const v3 o = mk(__uint_as_float(q[5][j]), __uint_as_float(q[6][j]), __uint_as_float(q[7][j]));
			v3 fr = mk(0, 0, 0); float pdf = 0.0f;
			eval_one<KIND_BECKMANN, WANT, FRK>(b, p, i, o, fr, pdf);
			if (WANT & 3) store3(vout, k, fr);
			if (WANT & 4) out_pdf[k] = pdf;
		}
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
		__builtin_amdgcn_wave_barrier();
	};
	const long long stride = (long long)gridDim.x * BLOCK;
	// the next tile's directions are requested before this tile's are looked at: a persistent loop has no other workgroup's loads to hide behind
	v3 i_next = mk(0, 0, 1), o_next = mk(0, 0, 1);
	{
		const long long k0 = (long long)blockIdx.x * BLOCK, k = k0 + t;
		if (k < n) { i_next = DENSE ? load3_dense(vi, k0, t) : load3(vi, k); o_next = DENSE ? load3_dense(vo, k0, t) : load3(vo, k); }
	}
	for (long long k0 = (long long)blockIdx.x * BLOCK; k0 < n; k0 += stride) {     // k0: workgroup-uniform
		const long long k = k0 + t;
		const bool live = k < n;
		const v3 i = i_next, o = o_next;
		{
			const long long k0n = k0 + stride, kn = k0n + t;
			i_next = mk(0, 0, 1); o_next = mk(0, 0, 1);
			if (kn < n) { i_next = DENSE ? load3_dense(vi, k0n, t) : load3(vi, kn); o_next = DENSE ? load3_dense(vo, k0n, t) : load3(vo, kn); }
		}
		// (a): dot(k, m_n) = k.x 0 + k.y 0 + k.z for the mean normal (0, 0, 1) of a lobe without offset: not positive, or NaN
		const bool fin_o = (fabsf(o.x) < 3e38f) & (fabsf(o.y) < 3e38f), fin_i = (fabsf(i.x) < 3e38f) & (fabsf(i.y) < 3e38f);
		const bool below = (fin_o & !(o.z > 0.0f)) | ((b.shadow != 0) & fin_i & !(i.z > 0.0f));
		// (b)
		// each component on its own: fmaxf drops a NaN operand, and a NaN in i.x / i.y (with fine z's) must reach the exact path --
		// without shadowing G = g1(o) > 0 there and F(sat(NaN)) = NaN for the Schlick / unpolarized terms, so the reference returns NaN * 0
		const bool sane = (fabsf(i.x) < 8.0f) & (fabsf(i.y) < 8.0f) & (fabsf(i.z) < 8.0f) & (fabsf(o.x) < 8.0f) & (fabsf(o.y) < 8.0f) &
		                  (fabsf(o.z) < 8.0f) & (i.z > 1e-4f) & (o.z > 1e-4f);
		const v3 h = normalize(add(i, o));                                          // as mf_eval_pdf
		const bool facing = h.z > 1e-4f;                                            // mf_ndf's cut
		const float r2 = mf_p22_rsqr(-h.x / h.z, -h.y / h.z, p);                    // mf_ndf / mf_p22's slope radius
		const float ih = dot(i, h);
		bool trivial = below | (sane & (!facing | (r2 >= 104.0f)));
		// (b)'s pdf is (+0) / (4 dot(i, h)) when G > 0 and the initial +0 when it is not (a sigma can be NaN for a direction that the
		// stretch rounds onto the pole): the same +0 as long as dot(i, h) > 0, which is all that is taken here
		if (WANT & 4) trivial &= below | (ih > 0.0f);
		if (live) {
			// every lane stores -- whole lines leave the wave; a queued pair's value is a placeholder that the drain overwrites (after
			// waiting for these stores: drain).  eval = evalp / i.z = (1 / i.z) * vec3(0): +0 in (b); in (a) -0 for i below the horizon,
			// NaN on it (dj_brdf.h:1551-1555)
			const v3 z = (WANT & 1) ? divs(mk(0, 0, 0), i.z) : mk(0, 0, 0);
			if (WANT & 3) { if (DENSE) store3_dense(vout, k0, t, z); else store3(vout, k, z); }
			if (WANT & 4) { if (DENSE) (*dense_at(out_pdf + k0, t)) = 0.0f; else out_pdf[k] = 0.0f; }
		}
		const bool full = live & !trivial;
		const unsigned long long mask = __ballot(full);
		if (mask) {
			if (full) {
				const unsigned int j = qn + (unsigned int)__popcll(mask & ((1ull << lane) - 1ull));
				q[0][j] = (unsigned int)(unsigned long long)k; q[1][j] = (unsigned int)((unsigned long long)k >> 32);
				q[2][j] = __float_as_uint(i.x); q[3][j] = __float_as_uint(i.y); q[4][j] = __float_as_uint(i.z);
				q[5][j] = __float_as_uint(o.x); q[6][j] = __float_as_uint(o.y); q[7][j] = __float_as_uint(o.z);
			}
			qn += (unsigned int)__popcll(mask);
			if (qn >= 64u) { qn -= 64u; drain(qn, 64u); }
		}
	}
	if (qn) drain(0u, qn);
}

// k_eval_bk_sharp's domain: no mean-normal offset (the (a) / (b) arguments use m_n = +z), a Fresnel term that cannot be negative or NaN,
// and a lobe sharp enough for the trivial pairs to pay for the prefix (DJB_BK_SHARP_ALPHA: profiles/r04/beckmann_sharp.txt)
#ifndef DJB_BK_SHARP_ALPHA
#define DJB_BK_SHARP_ALPHA 0.10f
#endif
inline bool beckmann_sharp_supported(const Brdf &b, const Params &p)
{
	static const float alpha_max = getenv("DJB_BK_SHARP_ALPHA") ? (float)atof(getenv("DJB_BK_SHARP_ALPHA")) : DJB_BK_SHARP_ALPHA;
	if (b.kind != KIND_BECKMANN) return false;
	if (!(p.tx == 0.0f && p.ty == 0.0f && p.nx == 0.0f && p.ny == 0.0f && p.nz == 1.0f)) return false;
	if (!(p.ax >= 1e-3f && p.ay >= 1e-3f && p.ax <= alpha_max && p.ay <= alpha_max && fabsf(p.rho) <= 0.99f)) return false;
	if (b.fr.kind == FR_SCHLICK) { for (int c = 0; c < 3; ++c) if (!(b.fr.a[c] >= 0.0f && b.fr.a[c] <= 1.0f)) return false; }
	else if (b.fr.kind == FR_UNPOLARIZED) { for (int c = 0; c < 3; ++c) if (!(b.fr.a[c] > 1.0f && b.fr.a[c] < 1e6f)) return false; }
	else if (b.fr.kind != FR_IDEAL) return false;
	return true;
}

template <int KIND, int FRK>
hipError_t launch_eval_kind_fr(hipStream_t s, const Brdf &b, const Params &p, long long n,
                               const View &i, const View &o, const View &out, float *out_pdf, int want)
{
	dim3 g((KIND == KIND_BECKMANN || KIND == KIND_GGX) ? grid_full(n) : grid_for(n)), t(eval_block(KIND));
	if (eval_block(KIND) != BLOCK) { long long bl = (n + eval_block(KIND) - 1) / eval_block(KIND); g = dim3((unsigned int)(bl < 1 ? 1 : bl > 2048 ? 2048 : bl)); }
	const bool dn = dense(i) && dense(o) && dense(out);
	if constexpr (KIND == KIND_BECKMANN && (FRK == FR_IDEAL || FRK == FR_SCHLICK || FRK == FR_UNPOLARIZED)) {
		if (beckmann_sharp_supported(b, p) && n >= (1LL << 16)) {
			// persistent grid, ~32 tiles per workgroup: the queue needs iterations to fill
			long long tiles = (n + BLOCK - 1) / BLOCK, blocks = (tiles + 31) / 32;
			if (blocks < 4096) blocks = tiles < 4096 ? tiles : 4096;
			const dim3 gs((unsigned int)(blocks > 0x7fffffffLL ? 0x7fffffffLL : blocks));
#define DJB_LAUNCH_SHARP(W_) do { if (dn) hipLaunchKernelGGL((k_eval_bk_sharp<W_, FRK, true>), gs, t, 0, s, b, p, n, i, o, out, out_pdf); \
                                  else hipLaunchKernelGGL((k_eval_bk_sharp<W_, FRK, false>), gs, t, 0, s, b, p, n, i, o, out, out_pdf); } while (0)
			switch (want) {
			case 1: DJB_LAUNCH_SHARP(1); break;
			case 2: DJB_LAUNCH_SHARP(2); break;
			case 4: DJB_LAUNCH_SHARP(4); break;
			case 5: DJB_LAUNCH_SHARP(5); break;
			case 6: DJB_LAUNCH_SHARP(6); break;
			default: return hipErrorInvalidValue;
			}
#undef DJB_LAUNCH_SHARP
			return hipGetLastError();
		}
	}
#define DJB_LAUNCH_EVAL(W_) do { if (dn) hipLaunchKernelGGL((k_eval<KIND, W_, FRK, true>), g, t, 0, s, b, p, n, i, o, out, out_pdf); \
                                 else hipLaunchKernelGGL((k_eval<KIND, W_, FRK, false>), g, t, 0, s, b, p, n, i, o, out, out_pdf); } while (0)
	switch (want) {
	case 1: DJB_LAUNCH_EVAL(1); break;
	case 2: DJB_LAUNCH_EVAL(2); break;
	case 4: DJB_LAUNCH_EVAL(4); break;
	case 5: DJB_LAUNCH_EVAL(5); break;
	case 6: DJB_LAUNCH_EVAL(6); break;
	default: return hipErrorInvalidValue;
	}
#undef DJB_LAUNCH_EVAL
	return hipGetLastError();
}

template <int KIND>
hipError_t launch_eval_kind(hipStream_t s, const Brdf &b, const Params &p, long long n,
                            const View &i, const View &o, const View &out, float *out_pdf, int want)
{
	// the analytic lobes get kernels specialised for the ideal / schlick Fresnel terms; the pdf-only
	// output (want == 4) never evaluates Fresnel, so it uses the ideal instantiation too
	if constexpr (KIND == KIND_BECKMANN || KIND == KIND_GGX) {
		if (b.fr.kind == FR_IDEAL || want == 4)
			return launch_eval_kind_fr<KIND, FR_IDEAL>(s, b, p, n, i, o, out, out_pdf, want);
		if (b.fr.kind == FR_SCHLICK)
			return launch_eval_kind_fr<KIND, FR_SCHLICK>(s, b, p, n, i, o, out, out_pdf, want);
		if (b.fr.kind == FR_UNPOLARIZED)
			return launch_eval_kind_fr<KIND, FR_UNPOLARIZED>(s, b, p, n, i, o, out, out_pdf, want);
	}
	// the fitted lobes carry a spline Fresnel (or the ideal one while under construction): the generic
	// instantiation would drag the sgd term's pow -- glibc_pow, ~60 VGPRs -- into a kernel that never calls it
	if constexpr (KIND == KIND_TABULAR || KIND == KIND_TABULAR_ANISO) {
		if (b.fr.kind == FR_SPLINE)
			return launch_eval_kind_fr<KIND, FR_SPLINE>(s, b, p, n, i, o, out, out_pdf, want);
	}
	return launch_eval_kind_fr<KIND, -1>(s, b, p, n, i, o, out, out_pdf, want);
}

// ------------------------------------------------------------------ per-pair parameters (LEAN / LEADR)
// The batch form of "build microfacet::params per hit, then evalp(i, o, &params)"
// (mitsuba/dj_beckmannconductor.cpp:291-319).  MODE 0: pdfparams records (ax, ay, rho, tx, ty) are
// read per pair.  MODE 1: per-pair LEAN texel moments (E1..E5) are composed on the fly with the base
// lobe, params = lrep_to_params(lrep(lean_k) * dmapscale + params_to_lrep(base)), and optionally written back.
template <int KIND, int WANT, int MODE, int FRK = -1>
__global__ __launch_bounds__(BLOCK) void k_eval_pp(Brdf b, long long n, View vi, View vo, const float *rec,
                                                   LeanCfg base, View vout, float *out_pdf, float *out_pp)
{
	__shared__ unsigned long long s_exp[KIND == KIND_BECKMANN ? 256 : 1];     // as in k_eval
	constexpr bool ATANT = KIND == KIND_TABULAR || KIND == KIND_TABULAR_ANISO || KIND == KIND_SGD || FRK == FR_SPLINE;
	__shared__ double s_atan[ATANT ? 16 : 1];
	if (KIND == KIND_BECKMANN) b.exp_lds = glibc_exp_tab_to_lds(s_exp, threadIdx.x, BLOCK);
	b.atan_lds = ATANT ? atan_tab_to_lds(s_atan, threadIdx.x) : 0u;
	if (KIND == KIND_BECKMANN || ATANT) __syncthreads();
	long long stride = (long long)gridDim.x * BLOCK;
	for (long long k = (long long)blockIdx.x * BLOCK + threadIdx.x; k < n; k += stride) {
		v3 fr = mk(0, 0, 0); float pdf = 0.0f;
		pp_one<KIND, WANT, MODE, FRK>(b, load3(vi, k), load3(vo, k), rec + 5 * k, base, out_pp ? out_pp + 5 * k : nullptr, fr, pdf);
		if (WANT & 3) store3(vout, k, fr);
		if (WANT & 4) out_pdf[k] = pdf;
	}
}

template <int KIND, int MODE, int FRK>
hipError_t launch_eval_pp_kind_fr(hipStream_t s, const Brdf &b, long long n, const View &i, const View &o,
                                  const float *rec, const LeanCfg &base, const View &out, float *out_pdf,
                                  float *out_pp, int want)
{
	dim3 g((KIND == KIND_BECKMANN || KIND == KIND_GGX) ? grid_full(n) : grid_for(n)), t(BLOCK);
	switch (want) {
	case 1: hipLaunchKernelGGL((k_eval_pp<KIND, 1, MODE, FRK>), g, t, 0, s, b, n, i, o, rec, base, out, out_pdf, out_pp); break;
	case 2: hipLaunchKernelGGL((k_eval_pp<KIND, 2, MODE, FRK>), g, t, 0, s, b, n, i, o, rec, base, out, out_pdf, out_pp); break;
	case 4: hipLaunchKernelGGL((k_eval_pp<KIND, 4, MODE, FRK>), g, t, 0, s, b, n, i, o, rec, base, out, out_pdf, out_pp); break;
	case 5: hipLaunchKernelGGL((k_eval_pp<KIND, 5, MODE, FRK>), g, t, 0, s, b, n, i, o, rec, base, out, out_pdf, out_pp); break;
	case 6: hipLaunchKernelGGL((k_eval_pp<KIND, 6, MODE, FRK>), g, t, 0, s, b, n, i, o, rec, base, out, out_pdf, out_pp); break;
	default: return hipErrorInvalidValue;
	}
	return hipGetLastError();
}

template <int KIND, int MODE>
hipError_t launch_eval_pp_kind(hipStream_t s, const Brdf &b, long long n, const View &i, const View &o,
                               const float *rec, const LeanCfg &base, const View &out, float *out_pdf,
                               float *out_pp, int want)
{
	// as launch_eval_kind: the analytic lobes get kernels specialised for the ideal / schlick Fresnel terms
	if constexpr (KIND == KIND_BECKMANN || KIND == KIND_GGX) {
		if (b.fr.kind == FR_IDEAL || want == 4)
			return launch_eval_pp_kind_fr<KIND, MODE, FR_IDEAL>(s, b, n, i, o, rec, base, out, out_pdf, out_pp, want);
		if (b.fr.kind == FR_SCHLICK)
			return launch_eval_pp_kind_fr<KIND, MODE, FR_SCHLICK>(s, b, n, i, o, rec, base, out, out_pdf, out_pp, want);
		if (b.fr.kind == FR_UNPOLARIZED)
			return launch_eval_pp_kind_fr<KIND, MODE, FR_UNPOLARIZED>(s, b, n, i, o, rec, base, out, out_pdf, out_pp, want);
	}
	if constexpr (KIND == KIND_TABULAR || KIND == KIND_TABULAR_ANISO) {
		if (b.fr.kind == FR_SPLINE)
			return launch_eval_pp_kind_fr<KIND, MODE, FR_SPLINE>(s, b, n, i, o, rec, base, out, out_pdf, out_pp, want);
	}
	return launch_eval_pp_kind_fr<KIND, MODE, -1>(s, b, n, i, o, rec, base, out, out_pdf, out_pp, want);
}

// ------------------------------------------------------------------ sample / evalp_is
#include "djb_contract_device.inc"   // ct_is_tail: the evalp_is tail under DJB_OPT_CONTRACT_1E5
// evalp_is of a GGX lobe under the contract: the EXACT sampled direction (mf_sample), weight and pdf by ct_is_tail; the few
// pairs it declines run the exact tail in place (a cold branch: 0.05 % of the bench distribution)
template <bool RNG, int FRK, bool DENSE>
__global__ __launch_bounds__(BLOCK) void k_evalp_is_ggx_ct(Brdf b, Params p, djbk::CtParams ct, long long n, const float *u1a,
                                                           const float *u2a, uint32_t seed1, uint32_t seed2,
                                                           unsigned long long start, View vo, View vi_out,
                                                           View vw_out, float *out_pdf)
{
	const GlibcTabs gt = glibc_tabs_global();
	const long long stride = (long long)gridDim.x * BLOCK;
	const unsigned int t = threadIdx.x;
	for (long long k0 = (long long)blockIdx.x * BLOCK; k0 < n; k0 += stride) {
		const long long k = k0 + t;
		if (k >= n) continue;
		float u1 = RNG ? gen_uniform(seed1, start + (unsigned long long)k) : (DENSE ? (*dense_at(u1a + k0, t)) : u1a[k]);
		float u2 = RNG ? gen_uniform(seed2, start + (unsigned long long)k) : (DENSE ? (*dense_at(u2a + k0, t)) : u2a[k]);
		const v3 o = DENSE ? load3_dense(vo, k0, t) : load3(vo, k);
		const v3 i_ = mf_sample<KIND_GGX>(b, p, u1, u2, o, gt);
		v3 w, i_out; float pdf; bool live;
		if (ct_is_tail<KIND_GGX, FRK>(ct, i_, o, w, pdf, live)) i_out = live ? i_ : mk(0, 0, 0);
		else { i_out = mk(0, 0, 0); w = mf_evalp_is_tail<KIND_GGX, FRK>(b, p, i_, o, i_out, pdf); }
		if (DENSE) { store3_dense(vi_out, k0, t, i_out); store3_dense(vw_out, k0, t, w); (*dense_at(out_pdf + k0, t)) = pdf; }
		else { store3(vi_out, k, i_out); store3(vw_out, k, w); out_pdf[k] = pdf; }
	}
}
// FRK: Fresnel kind fixed at compile time for evalp_is of the analytic lobes (as in k_eval)
template <int KIND, bool IS, bool RNG, int FRK = -1, bool DENSE = false>
__global__ __launch_bounds__(BLOCK) void k_sample(Brdf b, Params p, long long n, const float *u1a,
                                                  const float *u2a, uint32_t seed1, uint32_t seed2,
                                                  unsigned long long start, View vo, View vi_out,
                                                  View vw_out, float *out_pdf)
{
	// Beckmann's quantile functions call glibc's logf / expf / powf restatement: its tables go to LDS
	__shared__ double s_glibc[KIND == KIND_BECKMANN ? GLIBC_LDS_WORDS : 1];
	GlibcTabs gt = glibc_tabs_global();
	__shared__ unsigned long long s_exp[KIND == KIND_BECKMANN ? 256 : 1];     // the fp64 exp table, as in k_eval
	// evalp_is evaluates the lobe at the sampled direction: the table coordinates of a tabulated lobe / of a Fresnel spline (as in k_eval)
	constexpr bool ATANT = IS && (KIND == KIND_TABULAR || KIND == KIND_TABULAR_ANISO || FRK == FR_SPLINE);
	__shared__ double s_atan[ATANT ? 16 : 1];
	if (KIND == KIND_BECKMANN) {
		gt = glibc_tabs_to_lds(s_glibc, threadIdx.x, BLOCK);
		gt.exp64 = b.exp_lds = glibc_exp_tab_to_lds(s_exp, threadIdx.x, BLOCK);
	}
	b.atan_lds = ATANT ? atan_tab_to_lds(s_atan, threadIdx.x) : 0u;
	// the quantile tables of a fitted lobe in LDS (as k_eval does for its tables): qf of tabular, qf1 and the qf2 grid of tabular_anisotropic
	constexpr int TAB_LDS = KIND == KIND_TABULAR ? 2048 : KIND == KIND_TABULAR_ANISO ? 8192 + 1024 : 0;
	__shared__ float s_tab[TAB_LDS ? TAB_LDS : 1];
	if (TAB_LDS) {
		int used = 0;
		auto stage = [&](const float *&src, int count) {
			if (src == nullptr || count <= 0 || used + count > TAB_LDS) return;
			float *dst = s_tab + used;
			for (int k = threadIdx.x; k < count; k += BLOCK) dst[k] = src[k];
			src = dst; used += count;
		};
		if (KIND == KIND_TABULAR) stage(b.qf, b.n_qf);
		if (KIND == KIND_TABULAR_ANISO) { stage(b.a_qf2, b.elev * b.azim); stage(b.a_qf1, b.n_a_qf1); }
	}
	if (KIND == KIND_BECKMANN || ATANT || TAB_LDS) __syncthreads();
	const long long stride = (long long)gridDim.x * BLOCK;
	const unsigned int t = threadIdx.x;
	for (long long k0 = (long long)blockIdx.x * BLOCK; k0 < n; k0 += stride) {     // k0: workgroup-uniform
		const long long k = k0 + t;
		// scalar tile bound, SGPR-base dense accesses (as in k_eval / k_sample_bk; same-box: tabular sample -2.5 %, GGX evalp_is -1.7 %;
		// tabular_anisotropic +0.6 % with the opaque offsets: plain ones there)
		const unsigned int rem = n - k0 >= (long long)BLOCK ? (unsigned int)BLOCK : (unsigned int)(n - k0);
		if (t >= rem) continue;
		const unsigned int toff = KIND == KIND_TABULAR_ANISO ? (t << 2) : lane_byte_offset(t);
		float u1 = RNG ? gen_uniform(seed1, start + (unsigned long long)k) : (DENSE ? (*dense_off(u1a + k0, toff)) : u1a[k]);
		float u2 = RNG ? gen_uniform(seed2, start + (unsigned long long)k) : (DENSE ? (*dense_off(u2a + k0, toff)) : u2a[k]);
		v3 o = DENSE ? load3_dense_off(vo, k0, toff) : load3(vo, k), i_out, w; float pdf;
		sample_one<KIND, IS, FRK>(b, p, u1, u2, o, gt, i_out, w, pdf);
		const unsigned int soff = KIND == KIND_TABULAR_ANISO ? (t << 2) : lane_byte_offset(t);
		if (DENSE) store3_dense_off(vi_out, k0, soff, i_out); else store3(vi_out, k, i_out);
		if (IS) {
			if (DENSE) { store3_dense_off(vw_out, k0, soff, w); (*dense_off(out_pdf + k0, soff)) = pdf; }
			else { store3(vw_out, k, w); out_pdf[k] = pdf; }
		}
	}
}

template <int KIND>
hipError_t launch_sample_kind(hipStream_t s, const Brdf &b, const Params &p, long long n,
                              const float *u1, const float *u2, uint32_t s1, uint32_t s2,
                              unsigned long long start, const View &o, const View &out_i,
                              const View *out_w, float *out_pdf)
{
	dim3 g((KIND == KIND_BECKMANN || KIND == KIND_GGX) ? grid_full(n) : grid_for(n)), t(BLOCK);
	View w = out_w ? *out_w : View{ nullptr, nullptr, nullptr, 0 };
	const bool is = out_w != nullptr, rng = u1 == nullptr;
	const bool dn = dense(o) && dense(out_i) && dense(w);
	// evalp_is evaluates the Fresnel term of the sampled pair: specialised like k_eval
	constexpr bool analytic = KIND == KIND_BECKMANN || KIND == KIND_GGX;
	constexpr bool fitted = KIND == KIND_TABULAR || KIND == KIND_TABULAR_ANISO;
#define DJB_LAUNCH_S(IS_, RNG_, FRK_, DN_) hipLaunchKernelGGL((k_sample<KIND, IS_, RNG_, FRK_, DN_>), g, t, 0, s, b, p, n, u1, u2, s1, s2, start, o, out_i, w, out_pdf)
#define DJB_LAUNCH_S2(IS_, FRK_) do { if (rng) { if (dn) DJB_LAUNCH_S(IS_, true, FRK_, true); else DJB_LAUNCH_S(IS_, true, FRK_, false); } \
                                      else { if (dn) DJB_LAUNCH_S(IS_, false, FRK_, true); else DJB_LAUNCH_S(IS_, false, FRK_, false); } \
                                      return hipGetLastError(); } while (0)
	if (!is) DJB_LAUNCH_S2(false, -1);
	if constexpr (analytic) {
		if (b.fr.kind == FR_IDEAL) DJB_LAUNCH_S2(true, FR_IDEAL);
		if (b.fr.kind == FR_SCHLICK) DJB_LAUNCH_S2(true, FR_SCHLICK);
		if (b.fr.kind == FR_UNPOLARIZED) DJB_LAUNCH_S2(true, FR_UNPOLARIZED);
	}
	if constexpr (fitted) {
		if (b.fr.kind == FR_SPLINE) DJB_LAUNCH_S2(true, FR_SPLINE);
	}
	DJB_LAUNCH_S2(true, -1);
#undef DJB_LAUNCH_S2
#undef DJB_LAUNCH_S
}

// sample / evalp_is with per-pair parameters (records as in k_eval_pp): one Newton inversion per lane on the general path --
// the two-path Beckmann sampler of djb_kernels_sample.hip needs wave-uniform parameters
template <int KIND, bool IS, int MODE, int FRK = -1>
__global__ __launch_bounds__(BLOCK) void k_sample_pp(Brdf b, long long n, const float *u1a, const float *u2a, View vo,
                                                     const float *rec, LeanCfg base, View vi_out, View vw_out,
                                                     float *out_pdf, float *out_pp)
{
	__shared__ double s_glibc[KIND == KIND_BECKMANN ? GLIBC_LDS_WORDS : 1];
	GlibcTabs gt = glibc_tabs_global();
	__shared__ unsigned long long s_exp[KIND == KIND_BECKMANN ? 256 : 1];
	if (KIND == KIND_BECKMANN) {
		gt = glibc_tabs_to_lds(s_glibc, threadIdx.x, BLOCK);
		gt.exp64 = b.exp_lds = glibc_exp_tab_to_lds(s_exp, threadIdx.x, BLOCK);
		__syncthreads();
	}
	const long long stride = (long long)gridDim.x * BLOCK;
	for (long long k = (long long)blockIdx.x * BLOCK + threadIdx.x; k < n; k += stride) {
		v3 i_out, w; float pdf;
		pp_sample_one<KIND, IS, MODE, FRK>(b, u1a[k], u2a[k], load3(vo, k), rec + 5 * k, base, out_pp ? out_pp + 5 * k : nullptr,
		                                   gt, i_out, w, pdf);
		store3(vi_out, k, i_out);
		if (IS) { store3(vw_out, k, w); out_pdf[k] = pdf; }
	}
}

template <int KIND>
hipError_t launch_sample_pp_kind(hipStream_t s, const Brdf &b, long long n, const float *u1, const float *u2, const View &o,
                                 const float *rec, int mode, const LeanCfg &base, const View &out_i, const View *out_w,
                                 float *out_pdf, float *out_pp)
{
	dim3 g((KIND == KIND_BECKMANN || KIND == KIND_GGX) ? grid_full(n) : grid_for(n)), t(BLOCK);
	View w = out_w ? *out_w : View{ nullptr, nullptr, nullptr, 0 };
#define DJB_SPP(IS_, FRK_) do { if (mode == 0) hipLaunchKernelGGL((k_sample_pp<KIND, IS_, 0, FRK_>), g, t, 0, s, b, n, u1, u2, o, rec, base, out_i, w, out_pdf, out_pp); \
                                else hipLaunchKernelGGL((k_sample_pp<KIND, IS_, 1, FRK_>), g, t, 0, s, b, n, u1, u2, o, rec, base, out_i, w, out_pdf, out_pp); \
                                return hipGetLastError(); } while (0)
	if (!out_w) DJB_SPP(false, -1);
	if constexpr (KIND == KIND_BECKMANN || KIND == KIND_GGX) {
		if (b.fr.kind == FR_IDEAL) DJB_SPP(true, FR_IDEAL);
		if (b.fr.kind == FR_SCHLICK) DJB_SPP(true, FR_SCHLICK);
	}
	if constexpr (KIND == KIND_TABULAR || KIND == KIND_TABULAR_ANISO) {
		if (b.fr.kind == FR_SPLINE) DJB_SPP(true, FR_SPLINE);
	}
	DJB_SPP(true, -1);
#undef DJB_SPP
}

// ------------------------------------------------------------------ microfacet / radial queries
template <int KIND>
__global__ __launch_bounds__(BLOCK) void k_query(Brdf b, Params p, int which, long long n, View va, View vb,
                                                 View vc, View vout)
{
	long long stride = (long long)gridDim.x * BLOCK;
	for (long long k = (long long)blockIdx.x * BLOCK + threadIdx.x; k < n; k += stride)
		store3(vout, k, query_one<KIND>(b, p, which, k, va, vb, vc));
}

// sgd::{ndf, gaf, g1, fresnel} and abc::{ndf, gaf, fresnel} (dj_brdf.h:505-509, 530-533)
template <int KIND>
__global__ __launch_bounds__(BLOCK) void k_model_query(Brdf b, int which, long long n, View va, View vb, View vc, View vout)
{
	long long stride = (long long)gridDim.x * BLOCK;
	for (long long k = (long long)blockIdx.x * BLOCK + threadIdx.x; k < n; k += stride)
		store3(vout, k, model_query_one<KIND>(b, which, k, va, vb, vc));
}

// ------------------------------------------------------------------ small utilities
template <bool INVERSE>
__global__ __launch_bounds__(BLOCK) void k_io_hd(long long n, View a, View bb, View c, View d)
{
	long long stride = (long long)gridDim.x * BLOCK;
	for (long long k = (long long)blockIdx.x * BLOCK + threadIdx.x; k < n; k += stride) {
		v3 r1, r2;
		if (!INVERSE) io_to_hd(load3(a, k), load3(bb, k), r1, r2);
		else hd_to_io(load3(a, k), load3(bb, k), r1, r2);
		store3(c, k, r1); store3(d, k, r2);
	}
}

__global__ __launch_bounds__(BLOCK) void k_merl_index(long long n, View vi, View vo, int32_t *idx)
{
	long long stride = (long long)gridDim.x * BLOCK;
	for (long long k = (long long)blockIdx.x * BLOCK + threadIdx.x; k < n; k += stride)
		idx[k] = merl_index(load3(vi, k), load3(vo, k));
}

// dj_brdf.h:1010-1023 applied once per table entry instead of once per lookup
__global__ __launch_bounds__(BLOCK) void k_merl_convert(const double *s, long long n, MerlTexel *table)
{
	long long stride = (long long)gridDim.x * BLOCK;
	for (long long k = (long long)blockIdx.x * BLOCK + threadIdx.x; k < n; k += stride) table[k] = merl_convert_one(s, n, k);
}

// utia::normalize (dj_brdf.h:1162-1177) then the (float_t) cast of dj_brdf.h:1144
// n = 3*288*288 samples (three planes) -> 288*288 records of eight float4 (128 bytes = one L2 line):
// the RGB of the eight taps (theta_v + c, phi_i + k, phi_v + l), c, k, l in {0, 1}, tap order
// 4c + 2k + l, azimuths wrapped at 48; theta_v + 1 is clamped to the last row for itv = 5 (never read:
// utia_eval clamps itv0 <= 4).  See utia_eval.
__global__ __launch_bounds__(BLOCK) void k_utia_convert(const double *s, long long n, float4 *table)
{
	long long stride = (long long)gridDim.x * BLOCK;
	for (long long e = (long long)blockIdx.x * BLOCK + threadIdx.x; e < n / 3; e += stride) utia_convert_one(s, n, e, table);
}

__global__ __launch_bounds__(BLOCK) void k_gen_dir(long long n, uint32_t seed, unsigned long long start, View out)
{
	long long stride = (long long)gridDim.x * BLOCK;
	for (long long k = (long long)blockIdx.x * BLOCK + threadIdx.x; k < n; k += stride)
		store3(out, k, gen_direction(seed, start + (unsigned long long)k));
}
__global__ __launch_bounds__(BLOCK) void k_gen_uni(long long n, uint32_t seed, unsigned long long start, float *out)
{
	long long stride = (long long)gridDim.x * BLOCK;
	for (long long k = (long long)blockIdx.x * BLOCK + threadIdx.x; k < n; k += stride)
		out[k] = gen_uniform(seed, start + (unsigned long long)k);
}

// self-test of the guarded fast paths of djb_device.hpp (inversesqrt_, recip_to_f32) against the
// exact double sequences they stand in for, on hash-generated inputs across the exponent range.
// counters: [0] inversesqrt mismatches, [1] reciprocal mismatches (both must be 0),
//           [2] inversesqrt exact-path fallbacks, [3] reciprocal fallbacks,
//           [4] sRGB-decode (pow 2.4) mismatches (must be 0), [5] sRGB-decode fallbacks,
//           [6] fdiv_r(a, b, 1/b) != a / b (must be 0), [7] its IEEE fallbacks (sub-normal quotients),
//           [8] float(sqrt(a)) mismatches, [9] its fallbacks, [10] div_to_f32(num, den) != float(num / den) (must be 0), [11] its fallbacks.
__global__ __launch_bounds__(BLOCK) void k_guard_selftest(long long n, uint32_t seed, unsigned long long *counters)
{
	unsigned long long bad_r = 0, bad_d = 0, fb_r = 0, fb_d = 0, bad_p = 0, fb_p = 0, bad_q = 0, fb_q = 0, bad_s = 0, fb_s = 0, bad_v = 0, fb_v = 0;
	long long stride = (long long)gridDim.x * BLOCK;
	for (long long k = (long long)blockIdx.x * BLOCK + threadIdx.x; k < n; k += stride) {
		uint32_t h0 = hash_u32(seed, (uint64_t)k, 0), h1 = hash_u32(seed, (uint64_t)k, 1), h2 = hash_u32(seed, (uint64_t)k, 2);
		// float x with a random mantissa and an exponent in [2^-40, 2^40); every 8th near 1 (unit vectors)
		int ex = (k & 7) ? (int)(h1 % 80u) - 40 : (int)(h1 % 3u) - 1;
		float x = ldexpf(1.0f + (float)(h0 >> 9) * 1.1920928955078125e-07f, ex);
		if (inversesqrt_(x) != F(1.0 / sqrt(D(x)))) ++bad_r;
		if (near_f32_midpoint(inversesqrt_fast(x))) ++fb_r;
		// double q = pi * t * t (the GGX slope pdf denominator) or 1 + p*x (erf), random low bits
		double q = (h2 & 1) ? DJB_PI * D(x) * D(x) : 1.0 + D(x) * 0.3275911;
		q = __longlong_as_double(__double_as_longlong(q) ^ (long long)(h2 >> 3));
		if (recip_to_f32(q) != F(1.0 / q)) ++bad_d;
		if (near_f32_midpoint(recip_fast(q))) ++fb_d;
		// float(sqrt(a)) of a double that is not a float: 1 - c^2 with c in [0, 1] (three quarters of the draws) or the random q
		{
			const float cc = (float)(h0 >> 8) * 5.9604644775390625e-08f;
			const double a = (k & 3) ? 1.0 - D(cc * cc) : (q < 0 ? -q : q);
			if (sqrt_to_f32(a) != F(sqrt(a))) ++bad_s;
			if (near_f32_midpoint(sqrt_fast(a))) ++fb_s;
		}
		// interpolated UTIA value above the sRGB knee: mostly (0.0375, 1.2], every 8th in [2^-4, 2^4)
		float v = (k & 7) ? 0.0375f + (float)(h2 >> 8) * 5.9604644775390625e-08f * 1.1625f
		                  : ldexpf(1.0f + (float)(h0 >> 9) * 1.1920928955078125e-07f, (int)(h1 % 8u) - 4);
		if (D(v) > 0.0375) {
			bool ok;
			(void)srgb_decode_fast(v, ok);
			if (!ok) ++fb_p;
			if (srgb_decode(v) != srgb_decode_exact(v)) ++bad_p;
		}
		// exact division through a double reciprocal (fdiv_r): numerator x, denominator from a second hash draw
		// (any exponent, edge mantissas every 8th), R as the host computes it
		{
			uint32_t h3 = hash_u32(seed, (uint64_t)k, 3);
			uint32_t mb = (h3 & 7u) == 1u ? 0u : (h3 & 7u) == 2u ? 0x7fffffu : (h3 >> 9);
			float bden = __uint_as_float(((h3 >> 8 & 1u) << 31) | ((1u + (h2 >> 8) % 253u) << 23) | mb);
			float a_num = (k & 1) ? x : __uint_as_float(h0 ^ (h1 << 5));
			float want = a_num / bden, got = fdiv_r(a_num, bden, 1.0 / D(bden));
			if (!(got == want || (got != got && want != want))) ++bad_q;
			float q0 = F(D(a_num) * (1.0 / D(bden)));
			if (!(fabsf(q0) >= 1.17549435e-38f) && a_num != 0.0f) ++fb_q;
		}
		// float(num / den) of two doubles: the operands of ggx_qf2_radial's addition forms (quotients a, b in [0, 1.001]:
		// -(a + b) / (1 - a b), (1 + a b) / (a - b)) three times out of four, else the random q above over a random double
		{
			const float qa = (float)(h0 >> 8) * 5.9664249e-08f, qb = (float)(h2 >> 8) * 5.9664249e-08f;
			double num, den;
			if ((k & 3) == 1) { num = D(-(qa + qb)); den = 1.0 - D(qa * qb); }
			else if ((k & 3) == 2) { num = 1.0 + D(qa * qb); den = D(qa - qb); }
			else if ((k & 3) == 3) { num = D(qa + qb); den = 1.0 - D(qa * qb); }
			else { num = q; den = __longlong_as_double(__double_as_longlong(D(x)) ^ (long long)(h1 >> 2)); }
			const float want = F(num / den), got = div_to_f32(num, den);
			if (!(got == want || (got != got && want != want))) ++bad_v;
			const double qq = num * recip_fast(den), aqq = qq < 0 ? -qq : qq;
			if (near_f32_midpoint(qq) || !(aqq > 1e-30 && aqq < 1e30)) ++fb_v;
		}
	}
	atomicAdd(&counters[0], bad_r); atomicAdd(&counters[1], bad_d);
	atomicAdd(&counters[2], fb_r); atomicAdd(&counters[3], fb_d);
	atomicAdd(&counters[4], bad_p); atomicAdd(&counters[5], fb_p);
	atomicAdd(&counters[6], bad_q); atomicAdd(&counters[7], fb_q);
	atomicAdd(&counters[8], bad_s); atomicAdd(&counters[9], fb_s);
	atomicAdd(&counters[10], bad_v); atomicAdd(&counters[11], fb_v);
}

// bins x bins histogram over [-1,1]^2: LDS atomics, one global flush per workgroup
__global__ __launch_bounds__(BLOCK) void k_hist_xy(long long n, View v, int bins, unsigned long long *counts)
{
	extern __shared__ unsigned int lds_hist[];
	int nb = bins * bins;
	for (int t = threadIdx.x; t < nb; t += BLOCK) lds_hist[t] = 0;
	__syncthreads();
	long long stride = (long long)gridDim.x * BLOCK;
	for (long long k = (long long)blockIdx.x * BLOCK + threadIdx.x; k < n; k += stride) {
		long long off = k * v.stride;
		float x = v.x[off], y = v.y[off];
		int bx = (int)((x + 1.0f) * 0.5f * (float)bins), by = (int)((y + 1.0f) * 0.5f * (float)bins);
		bx = bx < 0 ? 0 : (bx >= bins ? bins - 1 : bx);
		by = by < 0 ? 0 : (by >= bins ? bins - 1 : by);
		atomicAdd(&lds_hist[by * bins + bx], 1u);
	}
	__syncthreads();
	for (int t = threadIdx.x; t < nb; t += BLOCK)
		if (lds_hist[t]) atomicAdd(&counts[t], (unsigned long long)lds_hist[t]);
}

} // namespace

namespace djbk {

hipError_t launch_eval(hipStream_t s, const Brdf &b, const Params &p, long long n, const View &i,
                       const View &o, const View &out, float *out_pdf, int want)
{
	if (n <= 0) return hipSuccess;
	switch (b.kind) {
	case KIND_BECKMANN: return launch_eval_kind<KIND_BECKMANN>(s, b, p, n, i, o, out, out_pdf, want);
	case KIND_GGX:      return launch_eval_kind<KIND_GGX>(s, b, p, n, i, o, out, out_pdf, want);
	case KIND_TABULAR:  return launch_eval_kind<KIND_TABULAR>(s, b, p, n, i, o, out, out_pdf, want);
	case KIND_TABULAR_ANISO: return launch_eval_kind<KIND_TABULAR_ANISO>(s, b, p, n, i, o, out, out_pdf, want);
	case KIND_MERL:     return launch_eval_kind<KIND_MERL>(s, b, p, n, i, o, out, out_pdf, want);
	case KIND_UTIA:     return launch_eval_kind<KIND_UTIA>(s, b, p, n, i, o, out, out_pdf, want);
	case KIND_LAMBERT:  return launch_eval_kind<KIND_LAMBERT>(s, b, p, n, i, o, out, out_pdf, want);
	case KIND_SGD:      return launch_eval_kind<KIND_SGD>(s, b, p, n, i, o, out, out_pdf, want);
	case KIND_ABC:      return launch_eval_kind<KIND_ABC>(s, b, p, n, i, o, out, out_pdf, want);
	}
	return hipErrorInvalidValue;
}

hipError_t launch_sample(hipStream_t s, const Brdf &b, const Params &p, long long n, const float *u1,
                         const float *u2, uint32_t s1, uint32_t s2, unsigned long long start,
                         const View &o, const View &out_i, const View *out_w, float *out_pdf, bool contract)
{
	if (n <= 0) return hipSuccess;
	switch (b.kind) {
	case KIND_BECKMANN:
		// sample, and evalp_is with a Fresnel term fixed at compile time: the two-path kernel (djb_kernels_sample.hip).  evalp_is with
		// a run-time Fresnel kind (spline, sgd) keeps the one-kernel form: both paths inlined would need 185 VGPRs (2 waves per SIMD)
		if (!out_w || b.fr.kind == FR_IDEAL || b.fr.kind == FR_SCHLICK || b.fr.kind == FR_UNPOLARIZED)
			return launch_sample_beckmann(s, b, p, n, u1, u2, s1, s2, start, o, out_i, out_w, out_pdf, contract);
		return launch_sample_kind<KIND_BECKMANN>(s, b, p, n, u1, u2, s1, s2, start, o, out_i, out_w, out_pdf);
	case KIND_GGX: {
		if (contract && !out_w && sample_contract_supported(b, p))              // sample under DJB_OPT_CONTRACT_1E5: directions within 1e-5
			return launch_sample_ggx_contract(s, b, p, n, u1, u2, s1, s2, start, o, out_i);
		CtParams ct;
		if (contract && out_w && contract_params(b, p, nullptr, &ct)) {      // evalp_is under DJB_OPT_CONTRACT_1E5: exact direction, contract tail
			const dim3 g(grid_full(n)), t(BLOCK);
			const bool rng = u1 == nullptr, dn = dense(o) && dense(out_i) && dense(*out_w);
#define DJB_IS_CT(FRK_) do { \
			if (rng) { if (dn) hipLaunchKernelGGL((k_evalp_is_ggx_ct<true, FRK_, true>), g, t, 0, s, b, p, ct, n, u1, u2, s1, s2, start, o, out_i, *out_w, out_pdf); \
			           else hipLaunchKernelGGL((k_evalp_is_ggx_ct<true, FRK_, false>), g, t, 0, s, b, p, ct, n, u1, u2, s1, s2, start, o, out_i, *out_w, out_pdf); } \
			else { if (dn) hipLaunchKernelGGL((k_evalp_is_ggx_ct<false, FRK_, true>), g, t, 0, s, b, p, ct, n, u1, u2, s1, s2, start, o, out_i, *out_w, out_pdf); \
			       else hipLaunchKernelGGL((k_evalp_is_ggx_ct<false, FRK_, false>), g, t, 0, s, b, p, ct, n, u1, u2, s1, s2, start, o, out_i, *out_w, out_pdf); } \
			return hipGetLastError(); } while (0)
			if (b.fr.kind == FR_IDEAL) DJB_IS_CT(FR_IDEAL);
			if (b.fr.kind == FR_SCHLICK) DJB_IS_CT(FR_SCHLICK);
			if (b.fr.kind == FR_UNPOLARIZED) DJB_IS_CT(FR_UNPOLARIZED);
#undef DJB_IS_CT
		}
		return launch_sample_kind<KIND_GGX>(s, b, p, n, u1, u2, s1, s2, start, o, out_i, out_w, out_pdf);
	}
	case KIND_TABULAR:  return launch_sample_kind<KIND_TABULAR>(s, b, p, n, u1, u2, s1, s2, start, o, out_i, out_w, out_pdf);
	case KIND_TABULAR_ANISO: return launch_sample_kind<KIND_TABULAR_ANISO>(s, b, p, n, u1, u2, s1, s2, start, o, out_i, out_w, out_pdf);
	case KIND_MERL:     return launch_sample_kind<KIND_MERL>(s, b, p, n, u1, u2, s1, s2, start, o, out_i, out_w, out_pdf);
	case KIND_UTIA:     return launch_sample_kind<KIND_UTIA>(s, b, p, n, u1, u2, s1, s2, start, o, out_i, out_w, out_pdf);
	case KIND_LAMBERT:  return launch_sample_kind<KIND_LAMBERT>(s, b, p, n, u1, u2, s1, s2, start, o, out_i, out_w, out_pdf);
	case KIND_SGD:      return launch_sample_kind<KIND_SGD>(s, b, p, n, u1, u2, s1, s2, start, o, out_i, out_w, out_pdf);
	case KIND_ABC:      return launch_sample_kind<KIND_ABC>(s, b, p, n, u1, u2, s1, s2, start, o, out_i, out_w, out_pdf);
	}
	return hipErrorInvalidValue;
}

hipError_t launch_eval_pp(hipStream_t s, const Brdf &b, long long n, const View &i, const View &o,
                          const float *rec, int mode, const float *base5, float scale, int lean_flags, const View &out,
                          float *out_pdf, float *out_pp, int want)
{
	if (n <= 0) return hipSuccess;
	const LeanCfg base = lean_cfg(base5, scale, lean_flags);
#define DJB_PP(K) (mode == 0 ? launch_eval_pp_kind<K, 0>(s, b, n, i, o, rec, base, out, out_pdf, out_pp, want) \
                             : launch_eval_pp_kind<K, 1>(s, b, n, i, o, rec, base, out, out_pdf, out_pp, want))
	switch (b.kind) {
	case KIND_BECKMANN: return DJB_PP(KIND_BECKMANN);
	case KIND_GGX:      return DJB_PP(KIND_GGX);
	case KIND_TABULAR:  return DJB_PP(KIND_TABULAR);
	case KIND_TABULAR_ANISO: return DJB_PP(KIND_TABULAR_ANISO);
	}
#undef DJB_PP
	return hipErrorInvalidValue;
}

hipError_t launch_sample_pp(hipStream_t s, const Brdf &b, long long n, const float *u1, const float *u2, const View &o,
                            const float *rec, int mode, const float *base5, float scale, int lean_flags, const View &out_i,
                            const View *out_w, float *out_pdf, float *out_pp)
{
	if (n <= 0) return hipSuccess;
	const LeanCfg base = lean_cfg(base5, scale, lean_flags);
	switch (b.kind) {
	case KIND_BECKMANN: return launch_sample_pp_kind<KIND_BECKMANN>(s, b, n, u1, u2, o, rec, mode, base, out_i, out_w, out_pdf, out_pp);
	case KIND_GGX:      return launch_sample_pp_kind<KIND_GGX>(s, b, n, u1, u2, o, rec, mode, base, out_i, out_w, out_pdf, out_pp);
	case KIND_TABULAR:  return launch_sample_pp_kind<KIND_TABULAR>(s, b, n, u1, u2, o, rec, mode, base, out_i, out_w, out_pdf, out_pp);
	case KIND_TABULAR_ANISO: return launch_sample_pp_kind<KIND_TABULAR_ANISO>(s, b, n, u1, u2, o, rec, mode, base, out_i, out_w, out_pdf, out_pp);
	}
	return hipErrorInvalidValue;
}

hipError_t launch_query(hipStream_t s, const Brdf &b, const Params &p, int which, long long n, const View &a,
                        const View &bb, const View &c, const View &out)
{
	if (n <= 0) return hipSuccess;
	dim3 g(grid_for(n)), t(BLOCK);
	switch (b.kind) {
	case KIND_BECKMANN: hipLaunchKernelGGL((k_query<KIND_BECKMANN>), g, t, 0, s, b, p, which, n, a, bb, c, out); break;
	case KIND_GGX:      hipLaunchKernelGGL((k_query<KIND_GGX>), g, t, 0, s, b, p, which, n, a, bb, c, out); break;
	case KIND_TABULAR:  hipLaunchKernelGGL((k_query<KIND_TABULAR>), g, t, 0, s, b, p, which, n, a, bb, c, out); break;
	case KIND_TABULAR_ANISO: hipLaunchKernelGGL((k_query<KIND_TABULAR_ANISO>), g, t, 0, s, b, p, which, n, a, bb, c, out); break;
	case KIND_SGD: hipLaunchKernelGGL((k_model_query<KIND_SGD>), g, t, 0, s, b, which, n, a, bb, c, out); break;
	case KIND_ABC: hipLaunchKernelGGL((k_model_query<KIND_ABC>), g, t, 0, s, b, which, n, a, bb, c, out); break;
	default: return hipErrorInvalidValue;
	}
	return hipGetLastError();
}

hipError_t launch_io_to_hd(hipStream_t s, long long n, const View &a, const View &b, const View &c,
                           const View &d, bool inverse)
{
	if (n <= 0) return hipSuccess;
	dim3 g(grid_for(n)), t(BLOCK);
	if (!inverse) hipLaunchKernelGGL((k_io_hd<false>), g, t, 0, s, n, a, b, c, d);
	else hipLaunchKernelGGL((k_io_hd<true>), g, t, 0, s, n, a, b, c, d);
	return hipGetLastError();
}

hipError_t launch_merl_index(hipStream_t s, long long n, const View &i, const View &o, int32_t *idx)
{
	if (n <= 0) return hipSuccess;
	hipLaunchKernelGGL(k_merl_index, dim3(grid_for(n)), dim3(BLOCK), 0, s, n, i, o, idx);
	return hipGetLastError();
}

hipError_t launch_merl_convert(hipStream_t s, const double *samples, long long n, MerlTexel *table)
{
	hipLaunchKernelGGL(k_merl_convert, dim3(grid_for(n)), dim3(BLOCK), 0, s, samples, n, table);
	return hipGetLastError();
}

hipError_t launch_utia_convert(hipStream_t s, const double *samples, long long n, float4 *table)
{
	hipLaunchKernelGGL(k_utia_convert, dim3(grid_for(n)), dim3(BLOCK), 0, s, samples, n, table);
	return hipGetLastError();
}

hipError_t launch_gen_directions(hipStream_t s, long long n, uint32_t seed, unsigned long long start,
                                 const View &out)
{
	if (n <= 0) return hipSuccess;
	hipLaunchKernelGGL(k_gen_dir, dim3(grid_for(n)), dim3(BLOCK), 0, s, n, seed, start, out);
	return hipGetLastError();
}

hipError_t launch_gen_uniforms(hipStream_t s, long long n, uint32_t seed, unsigned long long start,
                               float *out)
{
	if (n <= 0) return hipSuccess;
	hipLaunchKernelGGL(k_gen_uni, dim3(grid_for(n)), dim3(BLOCK), 0, s, n, seed, start, out);
	return hipGetLastError();
}

// the device restatements of glibc's libm functions, evaluated as-is for the test-suite (djb_selftest_libm)
__global__ __launch_bounds__(BLOCK) void k_libm_probe(int fn, long long n, const double *x, const double *y, double *out)
{
	const GlibcTabs gt = glibc_tabs_global();
	long long stride = (long long)gridDim.x * BLOCK;
	for (long long k = (long long)blockIdx.x * BLOCK + threadIdx.x; k < n; k += stride) {
		double r;
		switch (fn) {
		case 0: r = glibc_exp(x[k]); break;
		case 1: r = glibc_pow(x[k], y[k]); break;
		case 2: r = D(glibc_logf(F(x[k]), gt)); break;
		case 3: r = D(glibc_expf(F(x[k]), gt)); break;
		case 4: r = D(glibc_powf(F(x[k]), F(y[k]), gt)); break;
		case 5: r = glibc_atan2(x[k], y[k]); break;
		case 6: r = D(atan2_to_f32(F(x[k]), F(y[k]), 1.0)); break;
		case 7: r = D(atan2_to_f32(F(x[k]), F(y[k]), D(F(180.0 / DJB_PI)))); break;
		case 8: r = glibc_sin(x[k]); break;
		case 9: r = glibc_cos(x[k]); break;
		case 10: r = glibc_tan(x[k]); break;
		default: r = glibc_acos(x[k], 0u); break;
		}
		out[k] = r;
	}
}
hipError_t launch_libm_probe(hipStream_t s, int fn, long long n, const double *x, const double *y, double *out)
{
	if (n <= 0) return hipSuccess;
	hipLaunchKernelGGL(k_libm_probe, dim3(grid_for(n)), dim3(BLOCK), 0, s, fn, n, x, y, out);
	return hipGetLastError();
}

// trig site fn on the floats with bit patterns first .. first + n - 1 (djb_selftest_trig_sweep): float sites write
// floats, the double sites (fn >= TRIG_DOUBLE) doubles
__global__ __launch_bounds__(BLOCK) void k_trig_sweep(int fn, uint32_t first, long long n, void *out)
{
	long long stride = (long long)gridDim.x * BLOCK;
	for (long long k = (long long)blockIdx.x * BLOCK + threadIdx.x; k < n; k += stride) {
		float x = __uint_as_float(first + (uint32_t)k);
		if (fn >= TRIG_DOUBLE) ((double *)out)[k] = trig_site_d(fn, x);
		else ((float *)out)[k] = trig_site(fn, x);
	}
}
hipError_t launch_trig_sweep(hipStream_t s, int fn, uint32_t first, long long n, void *out)
{
	if (n <= 0) return hipSuccess;
	hipLaunchKernelGGL(k_trig_sweep, dim3(grid_for(n)), dim3(BLOCK), 0, s, fn, first, n, out);
	return hipGetLastError();
}

// djb_selftest_model_fast: the decided fast tier of the sgd / abc models (djb_fast_models.inc) against the reference's chains, on the device.
// seed == 0: unit k takes the float whose bit pattern is first + k (every float polar cosine of (0, 1] is bits 1 .. 0x3f800000); else unit k
// draws a polar cosine (family k & 3: uniform; hugging the wall theta_k = theta0 of channel (k >> 2) % 3 [sgd]; grazing; next to the
// normal) and evaluates sgd::g1 and sgd::ndf (abc: ndf) of the direction (0, 0, z) twice: through the product's functions (fast tier, what
// it leaves to the exact chain) and through the exact chains alone.  counters = {values, values the fast tier left undecided, values
// whose two floats differ (must be 0)} for g1 and for ndf.
__global__ __launch_bounds__(BLOCK) void k_model_fast_selftest(Brdf b, long long n, uint32_t seed, uint32_t first, unsigned long long *counters)
{
	__shared__ unsigned long long s_exp[256];
	__shared__ double s_pow[384];
	__shared__ double s_atan[16];
	b.exp_lds = glibc_exp_tab_to_lds(s_exp, threadIdx.x, BLOCK);
	b.pow_lds = glibc_pow_tab_to_lds(s_pow, threadIdx.x, BLOCK);
	b.atan_lds = atan_tab_to_lds(s_atan, threadIdx.x);
	__syncthreads();
	const double *m = b.model;
	unsigned long long c[6] = { 0, 0, 0, 0, 0, 0 };
	const long long stride = (long long)gridDim.x * BLOCK;
	for (long long k = (long long)blockIdx.x * BLOCK + threadIdx.x; k < n; k += stride) {
		const uint32_t h0 = hash_u32(seed, (uint64_t)k, 1u), h1 = hash_u32(seed, (uint64_t)k, 2u);
		const float u = (float)(h0 >> 8) * 0x1p-24f, v = (float)(h1 >> 8) * 0x1p-24f;
		const unsigned int fam = (unsigned int)k & 3u, ch = ((unsigned int)k >> 2) % 3u;
		float z;
		if (seed == 0u) z = __uint_as_float(first + (uint32_t)k);            // exhaustive mode: the float with bit pattern first + k
		else if (fam == 0u) z = u;
		else if (fam == 1u) z = b.kind == KIND_SGD ? F(cos(m[30 + ch] + (D(u) - 0.3) * exp2(-30.0 * D(v)))) : sqrtf(u);
		else if (fam == 2u) z = 0.05f * u;
		else z = 1.0f - u * exp2f(-24.0f * v);
		if (!(z > 0.0f)) z = 1e-3f;
		if (z > 1.0f) z = 1.0f;
		const v3 d = mk(0.0f, 0.0f, z);
		if (b.kind == KIND_SGD) {
			const v3 got = sgd_g1_rgb(b, d);
			const double theta_k = glibc_acos(D(z), 0u);
			const float gg[3] = { got.x, got.y, got.z };
			bool in;
			const double th = acos_fast(z, atan_tab(b.atan_lds), in), dth = 1.01 * 0x1p-48 * th;
#pragma unroll
			for (int j = 0; j < 3; ++j) {
				const float want = F(sgd_g1(b, theta_k, m[30 + j], m[24 + j], m[27 + j], m[21 + j]));
				bool dec;
				(void)sgd_g1_fast<true>(m, j, th - m[30 + j], dth, b.pow_lds, b.exp_lds, dec);
				++c[0]; if (!(dec & in) || m[SGD_FAST_FLAG] == 0.0) ++c[1];
				if (__float_as_uint(gg[j]) != __float_as_uint(want)) ++c[2];
			}
			const v3 gn = sgd_ndf_rgb(b, d);
			const float nn[3] = { gn.x, gn.y, gn.z };
			const double chd = D(z), c2 = chd * chd, rc = recip_fast(c2);
#pragma unroll
			for (int j = 0; j < 3; ++j) {
				const float want = sgd_ndf(b, chd, m[6 + j], m[9 + j], m[18 + j]);
				bool dec;
				(void)sgd_ndf_fast(m, j, c2, (1.0 - c2) * rc, rc * rc, b.pow_lds, b.exp_lds, dec);
				++c[3]; if (!dec || m[SGD_FAST_FLAG] == 0.0) ++c[4];
				if (__float_as_uint(nn[j]) != __float_as_uint(want)) ++c[5];
			}
		} else {
			const v3 gn = abc_ndf_rgb(b, d);
			const float nn[3] = { gn.x, gn.y, gn.z };
			const double w = 1.0 + m[6] * (1.0 - D(z));
			const double den = glibc_pow(w, m[7], 0u, 0u);
			float fv[3]; bool dec[3];
			abc_ndf_fast(m, w, b.pow_lds, b.exp_lds, fv, dec);
#pragma unroll
			for (int j = 0; j < 3; ++j) {
				const float want = F(m[3 + j] / den);
				++c[3]; if (!dec[j]) ++c[4];
				if (__float_as_uint(nn[j]) != __float_as_uint(want)) ++c[5];
			}
		}
	}
#pragma unroll
	for (int j = 0; j < 6; ++j) if (c[j]) atomicAdd(&counters[j], c[j]);
}
hipError_t launch_model_fast_selftest(hipStream_t s, const Brdf &b, long long n, uint32_t seed, uint32_t first, unsigned long long *counters6)
{
	if (n <= 0) return hipSuccess;
	hipLaunchKernelGGL(k_model_fast_selftest, dim3(grid_for(n)), dim3(BLOCK), 0, s, b, n, seed, first, counters6);
	return hipGetLastError();
}

hipError_t launch_guard_selftest(hipStream_t s, long long n, uint32_t seed, unsigned long long *counters)
{
	hipLaunchKernelGGL(k_guard_selftest, dim3(grid_for(n)), dim3(BLOCK), 0, s, n, seed, counters);
	return hipGetLastError();
}

hipError_t launch_histogram_xy(hipStream_t s, long long n, const View &v, int bins,
                               unsigned long long *counts)
{
	if (n <= 0) return hipSuccess;
	size_t lds = sizeof(unsigned int) * (size_t)bins * bins;
	hipLaunchKernelGGL(k_hist_xy, dim3(grid_for(n)), dim3(BLOCK), lds, s, n, v, bins, counts);
	return hipGetLastError();
}

} // namespace djbk
