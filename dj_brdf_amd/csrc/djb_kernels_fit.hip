// djb_kernels_fit.hip -- the power-iteration fitter on gfx950:
//   djb::tabular::tabular(brdf, res, shadow)                        dj_brdf.h:2215-2236
//   tabular::fit_beckmann_parameters / fit_ggx_parameters           dj_brdf.h:3133-3184
//
// One 1024-thread workgroup (16 wave64s) per material; materials are independent, so a batch of M
// materials is a grid of M workgroups and a multi-GPU batch is a partition of the material list.
//
// The reference accumulates its quadratures in float, in loop order.  Float addition does not
// commute with a tree reduction, so to reproduce the reference's tables (and the %.3f digits of
// params.txt) every *sum* keeps the reference's order: the expensive per-term work (libm-class
// transcendentals, MERL look-ups, spline fetches) is spread over all 1024 lanes and staged in LDS,
// then the lane that owns a row adds its terms front to back.  Rows (theta_k, theta_d, theta_o)
// are independent and map to lanes; wave64 lanes of one row-owner wave broadcast-read the shared
// per-node tables from LDS.  No MFMA: the only matrix product is an (res-1)^2 matvec in double,
// four times (dj_brdf.h:2467-2480), whose summation order is likewise kept.
#include <cstdlib>
#include "djb_internal.hpp"
#include <stdio.h>

using namespace djbdev;

namespace {

constexpr int FIT_BLOCK = 1024;
constexpr int NTHETA_SIGMA = 90, NPHI_SIGMA = 180;      // dj_brdf.h:2350-2351
constexpr int NNODE_SIGMA = NTHETA_SIGMA * NPHI_SIGMA;
constexpr int NTHETA_FIT = 128;                          // dj_brdf.h:2279, 3135, 3162
constexpr int MAX_PHI_STEPS = 512;
// columns of the sigma quadrature staged per step (two LDS buffers of cnt x (tile + 1) floats)
__host__ __device__ inline int sig_tile(int cnt) { return cnt <= 100 ? 64 : 32; }
constexpr int FRES_REC = 5;             // floats per record of FitSplit::fres_dirs
#ifndef DJB_FIT_FRESNEL_SPLIT_MIN
#define DJB_FIT_FRESNEL_SPLIT_MIN 4     // slices per material from which the Fresnel-ratio pass is sliced as well (measured: profiles/r03/fit_phases.txt)
#endif
#ifndef DJB_FIT_WIDE_TILES
#define DJB_FIT_WIDE_TILES 1      // 0: measurement builds (the tile width of a sliced sigma pass stays sig_tile)
#endif

struct LdsPlan {   // byte offsets into dynamic LDS (doubles first: 8-byte aligned)
	int v0, v1, cphid, cthd, sphid;          // doubles
	int p22, sigma, cdf, qf, fres;           // floats (fres: 3 per entry)
	int theta, cosv, tanv, kji;              // floats [cnt]
	int cphi;                                // floats [MAX_PHI_STEPS]
	int ndf;                                 // floats [16200]
	int sh, ui, cthf;                        // floats [90]
	int terms;                               // floats [2*128]
	int qprobe;                              // floats [8*cnt]
	int skv, ckv;                            // floats [cnt]: sin / cos of theta_k (sigma rows)
	int stile;                               // floats [2][cnt][sig_tile + 4]: double-buffered tiles of sigma terms (rows 16-byte aligned)
	int total;
};

__host__ __device__ inline LdsPlan make_plan(int res)
{
	LdsPlan p; int cnt = res - 1, off = 0;
	auto take = [&](int bytes) { int o = off; off += (bytes + 15) & ~15; return o; };
	p.v0 = take(8 * cnt); p.v1 = take(8 * cnt);
	p.cphid = take(8 * NPHI_SIGMA); p.cthd = take(8 * NTHETA_SIGMA); p.sphid = take(8 * NPHI_SIGMA);
	p.p22 = take(4 * res); p.sigma = take(4 * res); p.cdf = take(4 * res); p.qf = take(4 * res);
	p.fres = take(12 * res);
	p.theta = take(4 * cnt); p.cosv = take(4 * cnt); p.tanv = take(4 * cnt); p.kji = take(4 * cnt);
	p.cphi = take(4 * MAX_PHI_STEPS);
	p.ndf = take(4 * NNODE_SIGMA);
	p.sh = take(4 * NTHETA_SIGMA); p.ui = take(4 * NTHETA_SIGMA); p.cthf = take(4 * NTHETA_SIGMA);
	p.terms = take(4 * 2 * NTHETA_FIT);
	p.qprobe = take(4 * 8 * cnt);
	p.skv = take(4 * cnt); p.ckv = take(4 * cnt);
	p.stile = take(4 * 2 * cnt * (sig_tile(cnt) + 4));
	p.total = off;
	return p;
}

// slot: the query slot of this look-up (djb_device.hpp: fit_merl_slot_count); a sparse MERL source holds one texel per slot.
// slot_bin (optional): the MERL bin of every slot (fit_merl_slot_index: the fit's directions depend on the resolution only, so their
// half / difference angle transforms -- fp64 atan2 / acos, the most expensive thing a dense MERL source does here -- are tabulated
// once per context with the rest of FitSplit::fres_dirs)
template <int SRC>
DJB_DEV v3 src_eval(const Brdf &src, const Params &std_p, v3 i, v3 o, int slot, const int *slot_bin = nullptr)
{
	v3 fr = mk(0, 0, 0); float pdf;
	if (SRC <= KIND_TABULAR || SRC == KIND_TABULAR_ANISO) mf_eval_pdf<SRC, 1>(src, std_p, i, o, fr, pdf);
	else if (SRC == KIND_MERL) {
		if (src.merl_sparse) { MerlTexel t = src.merl[slot]; fr = mk(t.x, t.y, t.z); }
		else if (slot_bin) { MerlTexel t = src.merl[slot_bin[slot]]; fr = mk(t.x, t.y, t.z); }
		else fr = merl_eval(src, i, o);
	}
	else if (SRC == KIND_UTIA) fr = utia_eval(src, i, o);
	else if (SRC == KIND_SGD) fr = sgd_eval(src, i, o);
	else if (SRC == KIND_ABC) fr = abc_eval(src, i, o);
	else fr = divs(mk(1, 1, 1), F(DJB_PI));
	(void)pdf;
	return fr;
}

#ifdef DJB_EXP_FIT_TS            // measurement builds only: wall-clock stamps (100 MHz) of the phase boundaries, last workgroup's thread 0
__device__ unsigned long long g_fit_ts[16];
#define DJB_FIT_TS(k_) do { if (tid == 0 && blockIdx.x == gridDim.x - 1) g_fit_ts[k_] = wall_clock64(); } while (0)
#else
#define DJB_FIT_TS(k_) do { } while (0)
#endif
template <int SRC>
__global__ __launch_bounds__(FIT_BLOCK) void k_fit(const Brdf *srcs, Params std_p, int n_mat, int res, int shadow,
                                                   double *km_scratch, float *ratio_scratch,
                                                   djbk::FitOut out, djbk::FitSplit split)
{
	extern __shared__ __align__(16) unsigned char lds[];
	// blocks [0, n_mat * (parts - 1)) are the helpers (dispatched first: they never wait), the last n_mat blocks
	// run the whole fit of material m and pick the helpers' sigma rows up
	const int parts = split.parts, n_help = n_mat * (parts - 1);
	const int part = (int)blockIdx.x < n_help ? 1 + (int)blockIdx.x / n_mat : 0;
	const int m = (int)blockIdx.x < n_help ? (int)blockIdx.x % n_mat : (int)blockIdx.x - n_help;
	const int tid = threadIdx.x, cnt = res - 1;
	// Slicing the Fresnel-ratio pass as well pays when few materials run (4+ slices each): with ~200 workgroups in
	// flight the extra release / acquire pairs (L2 write-backs of everybody's scratch) cost what the slicing gains.
	const bool fresnel_split = parts >= DJB_FIT_FRESNEL_SPLIT_MIN;
	const LdsPlan P = make_plan(res);
	double *v0 = (double *)(lds + P.v0), *v1 = (double *)(lds + P.v1);
	double *cphid = (double *)(lds + P.cphid), *cthd = (double *)(lds + P.cthd), *sphid = (double *)(lds + P.sphid);
	float *p22 = (float *)(lds + P.p22), *sigma = (float *)(lds + P.sigma);
	float *cdf = (float *)(lds + P.cdf), *qf = (float *)(lds + P.qf), *fres = (float *)(lds + P.fres);
	float *theta = (float *)(lds + P.theta), *cosv = (float *)(lds + P.cosv);
	float *tanv = (float *)(lds + P.tanv), *kji = (float *)(lds + P.kji);
	float *cphi = (float *)(lds + P.cphi), *ndf_tab = (float *)(lds + P.ndf);
	float *sh = (float *)(lds + P.sh), *ui = (float *)(lds + P.ui), *cthf = (float *)(lds + P.cthf);
	float *terms = (float *)(lds + P.terms), *qprobe = (float *)(lds + P.qprobe);
	float *skv = (float *)(lds + P.skv), *ckv = (float *)(lds + P.ckv), *stile = (float *)(lds + P.stile);
	__shared__ int s_nphi, s_nqf, s_have;
	__shared__ float s_scale;

	const Brdf src = srcs[m];
	// kmT[theta_h][theta_o], private to the workgroup.  It is dead before the sigma pass fills ndf_tab, so when it fits there
	// (res 90: 63 368 of 64 800 bytes) it lives in LDS: the four matvecs of the power iteration -- 89 lanes, 89 dependent
	// fp64 multiply-adds each -- read it at LDS latency instead of the L2's.
	const bool km_lds = (size_t)cnt * cnt * sizeof(double) <= (size_t)NNODE_SIGMA * sizeof(float) && (P.ndf & 7) == 0;
	double *kmT = km_lds ? (double *)(lds + P.ndf) : km_scratch + (size_t)blockIdx.x * cnt * cnt;
	float *ratio = ratio_scratch + (size_t)m * cnt * (cnt + 1) * 3;

	// the object under construction: tabular NDF, ideal Fresnel until compute_fresnel finishes
	Brdf self;
	self.kind = KIND_TABULAR; self.shadow = shadow;
	self.fr.kind = FR_IDEAL; self.fr.pts = nullptr; self.fr.npts = 0;
	self.p22 = p22; self.sigma = sigma; self.cdf = cdf; self.qf = qf;
	self.n_p22 = res; self.n_sigma = res; self.n_cdf = res; self.n_qf = res;
	self.merl = nullptr; self.merl_sparse = 0; self.utia = nullptr; self.exp_lds = 0u; self.pow_lds = 0u; self.atan_lds = 0u;

	DJB_FIT_TS(0);
	// ================================================================ compute_p22_smith (dj_brdf.h:2482-2522)
	const float dtheta_k = F(sqrt(DJB_PI * 0.5) / D((float)cnt));
	const float dphi_h = F(DJB_PI / 180.0);
	// the phi integral of every (theta_o, theta_h) entry of the K matrix depends on the resolution only: tabulated once per context
	// (k_fit_smith_nint, the same loops as below), so that building K is one multiplication chain per entry
	const float *nint_tab = split.fres_dirs ? split.fres_dirs + FRES_REC * (size_t)cnt * (cnt + 1) + 1 + res + NNODE_SIGMA : nullptr;
	const int *slot_bin = nint_tab ? (const int *)(nint_tab + (size_t)cnt * cnt) : nullptr;      // [cnt + cnt (cnt + 1)]
	if (tid == 0 && !nint_tab) {   // the float-stepped phi loop (361 steps for dphi = pi/180): same phi values
		int c = 0;
		for (float phi = 0.0f; D(phi) < 2.0 * DJB_PI && c < MAX_PHI_STEPS; phi += dphi_h) cphi[c++] = phi;
		s_nphi = c;
	}
	for (int k = tid; k < cnt; k += FIT_BLOCK) {
		float th = fit_backscatter_theta(k, cnt);
		float th2 = th * th;
		float c = cos_f(th2), t = tan_f(th2);
		theta[k] = th; cosv[k] = c; tanv[k] = t;
		v3 w = from_angles(th2, 0.0f);
		float fr_i = intensity(src_eval<SRC>(src, std_p, w, w, k, slot_bin));
		kji[k] = F((D(dtheta_k) * glibc_pow(D(c), D(6.0f))) * (8.0 * D(fr_i)));
		v0[k] = 1.0;
	}
	__syncthreads();
	if (nint_tab) {
		for (int e = tid; e < cnt * cnt; e += FIT_BLOCK) {
			const int io = e / cnt, jh = e - io * cnt;
			const float ch = cosv[jh];
			kmT[(size_t)jh * cnt + io] = D(theta[jh] * kji[io] * nint_tab[e] * tanv[jh] / (ch * ch));
		}
	} else {
	const int nphi = s_nphi;
	for (int k = tid; k < nphi; k += FIT_BLOCK) cphi[k] = cos_f(cphi[k]);
	__syncthreads();
	{
		// Entry (io, jh) integrates max(1, tan(theta_h) tan(theta_o) cos(phi)) over the phi steps.  Where the product of the
		// tangents is <= 1 every term is exactly 1.0f and the sum exactly nphi (an integer below 2^24): no loop.  tanv grows
		// with the index, so for a given io the entries that need the loop are a suffix jh >= jmin(io); they are enumerated
		// (prefix sums over io, a 7-step search per work item) so that every lane of the loop has one.
		int *hoff = (int *)stile, *hmin = hoff + cnt + 1;        // the sigma tiles are not in use yet
		auto store = [&](int io, int jh, float nint) {
			nint *= dphi_h;
			const float ch = cosv[jh];
			kmT[(size_t)jh * cnt + io] = D(theta[jh] * kji[io] * nint * tanv[jh] / (ch * ch));
		};
		for (int io = tid; io < cnt; io += FIT_BLOCK) {
			int jm = 0;
			while (jm < cnt && !(tanv[jm] * tanv[io] > 1.0f)) ++jm;
			hmin[io] = jm; hoff[io + 1] = cnt - jm;
		}
		__syncthreads();
		if (tid == 0) { hoff[0] = 0; for (int io = 0; io < cnt; ++io) hoff[io + 1] += hoff[io]; }
		for (int e = tid; e < cnt * cnt; e += FIT_BLOCK) {       // the entries without a loop
			const int io = e / cnt, jh = e - io * cnt;
			if (jh < hmin[io]) store(io, jh, (float)nphi);
		}
		__syncthreads();
		const int n_heavy = hoff[cnt];
		for (int w = tid; w < n_heavy; w += FIT_BLOCK) {
			int lo = 0, hi = cnt - 1;                            // the io with hoff[io] <= w < hoff[io + 1]
			while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (hoff[mid] <= w) lo = mid; else hi = mid - 1; }
			const int io = lo, jh = hmin[io] + (w - hoff[io]);
			const float tan_product = tanv[jh] * tanv[io];
			float nint = 0.0f;
			for (int q = 0; q < nphi; ++q) nint += fmax_(1.0f, tan_product * cphi[q]);
			store(io, jh, nint);
		}
	}
	}
	if (!km_lds) __threadfence_block();
	__syncthreads();
	// matrix::eigenvector(4): 4 un-normalised matvecs from ones, row sums in index order
	for (int it = 0; it < 4; ++it) {
		double *vin = (it & 1) ? v1 : v0, *vout = (it & 1) ? v0 : v1;
		for (int j = tid; j < cnt; j += FIT_BLOCK) {
			double acc = 0.0;
			for (int i = 0; i < cnt; ++i) acc += kmT[(size_t)i * cnt + j] * vin[i];
			vout[j] = acc;
		}
		__syncthreads();
	}
	for (int k = tid; k < res; k += FIT_BLOCK) p22[k] = k < cnt ? F(1e-2 * v0[k]) : 0.0f;
	__syncthreads();

	DJB_FIT_TS(1);
	// ================================================================ normalize_p22 (dj_brdf.h:2277-2304)
	for (int k = tid; k < NTHETA_FIT; k += FIT_BLOCK) {
		float u = (float)k / (float)NTHETA_FIT;
		float th = F(D(u * u) * DJB_PI * 0.5);
		float r = tan_f(th), c = cos_f(th);
		float pr = p22_radial<KIND_TABULAR>(self, r * r);
		terms[k] = (u * pr * r) / (c * c);
	}
	__syncthreads();
	if (tid == 0) {
		float nint = 0.0f;
		for (int k = 0; k < NTHETA_FIT; ++k) nint += terms[k];
		nint *= F(DJB_PI / D((float)NTHETA_FIT)) * F(2.0 * DJB_PI);
		s_scale = F(1.0 / D(nint));
	}
	__syncthreads();
	for (int k = tid; k < res; k += FIT_BLOCK) p22[k] *= s_scale;
	__syncthreads();

	DJB_FIT_TS(2);
	// ================================================================ compute_sigma (dj_brdf.h:2348-2386)
	for (int k = tid; k < NPHI_SIGMA; k += FIT_BLOCK) {
		const double phi_h = D(F(D((float)k / (float)NPHI_SIGMA) * 2.0 * DJB_PI));
		cphid[k] = glibc_cos(phi_h); sphid[k] = glibc_sin(phi_h);
	}
	for (int k = tid; k < NTHETA_SIGMA; k += FIT_BLOCK) {
		float u = (float)k / (float)NTHETA_SIGMA;
		float th = F(D(u * u) * DJB_PI * 0.5);
		ui[k] = u; sh[k] = sin_f(th); cthd[k] = glibc_cos(D(th)); cthf[k] = cos_f(th);
	}
	__syncthreads();
	// ndf(vec3(theta_h, phi_h)) of the 180 x 90 quadrature nodes (theta_k-independent).  vec3(theta, phi) is
	// (float(s cos phi), float(s sin phi), cos_f(theta)) with s = sin_f(theta) (dj_brdf.h:589-595): the four trigonometric values
	// of a node come from the 90 + 180 entries above instead of four fp64 libm calls per node
	for (int e = tid; e < NNODE_SIGMA; e += FIT_BLOCK) {
		const int j2 = e / NTHETA_SIGMA, j1 = e - j2 * NTHETA_SIGMA;
		const float s1 = sh[j1];
		const float *u_node = split.fres_dirs ? split.fres_dirs + FRES_REC * (size_t)cnt * (cnt + 1) + 1 + res + e : nullptr;   // tabulated once per context
		ndf_tab[e] = mf_ndf<KIND_TABULAR>(self, mk(F(D(s1) * cphid[j2]), F(D(s1) * sphid[j2]), cthf[j1]), std_p, u_node);
	}
	__syncthreads();
	{
		// sigma(theta_k) = max(cos theta_k, sum over the 180 x 90 nodes) for cnt rows: 16 200 dependent
		// float adds per row in the reference's node order.  Producer waves compute the terms of one
		// tile of nodes (thread = node x row group: the node's table values stay in registers, sin / cos
		// of theta_k are LDS broadcasts) into one LDS buffer while the waves that own the rows add the
		// previous tile's terms front to back from the other buffer.  Same terms, same order as the
		// one-lane-per-row loop this replaces (1.14 ms of the 1.57 ms kernel).  When CUs are idle (fewer
		// materials than CUs) the rows are sliced over `parts` workgroups per material, which shortens the
		// producers' share of every tile: 100 materials 0.76 -> 0.72 ms (2 slices); one material 0.71 -> 0.44 ms
		// (8 slices, together with the sliced Fresnel pass below).
		const float dth = F(DJB_PI / D((float)NTHETA_SIGMA));
		const float dph = F(2.0 * DJB_PI / D((float)NPHI_SIGMA));
		for (int k = tid; k < cnt; k += FIT_BLOCK) {
			float tmp = (float)k / (float)cnt;
			float theta_k = F(D(tmp) * 0.5 * DJB_PI);
			ckv[k] = cos_f(theta_k); skv[k] = sin_f(theta_k);
		}
		// A workgroup that owns a slice of the rows (parts > 1) needs the tile buffers for that many rows only: the same LDS holds
		// tiles of two or four times the nodes, i.e. a half or a quarter of the producer / consumer hand-overs (barriers) per row.  Rows are stored relative to
		// the slice; the bounded-wait fallback below (partners late) walks the other rows in chunks of at most `rcap`.
		const int rcap = parts > 1 ? (cnt + parts - 1) / parts : cnt;
		const int T0 = sig_tile(cnt);
		const int fit4 = rcap * (4 * T0 + 4) <= cnt * (T0 + 4), fit2 = rcap * (2 * T0 + 4) <= cnt * (T0 + 4);
		const int T = (DJB_FIT_WIDE_TILES && parts > 1) ? (fit4 ? 4 * T0 : fit2 ? 2 * T0 : T0) : T0;
		const int TS = T + 4;                                // row stride: 16-byte aligned rows for the owners' float4 reads
		const int BUF = rcap * TS;                           // floats per tile buffer (two of them fit LdsPlan::stile by construction)
		const int n_sum = ((cnt + 63) / 64) * 64;            // threads [0, n_sum): row owners (whole waves)
		const int n_prod = FIT_BLOCK - n_sum;                // threads [n_sum, FIT_BLOCK): producers
		const int a = tid - n_sum, col = a % T, grp = a / T, ngrp = n_prod / T;
		const int ntiles = (NNODE_SIGMA + T - 1) / T;
		// rows [r0, r1) of the quadrature; every workgroup of the material runs the same code on its slice
		auto sigma_rows = [&](int r0, int r1) {
			// a producer lane serves the same rows k = r0 + grp + j ngrp in every tile: their sin / cos(theta_k) are read once
			// (registers) instead of twice per row and tile from LDS, and the rows' chains are independent of each other
			constexpr int RMAX = 8;
			const int nrows = (r1 - r0 + ngrp - 1) / (ngrp > 0 ? ngrp : 1);              // rows per producer lane, workgroup-uniform
			const bool held = nrows <= RMAX;
			float skr[RMAX]; double ckr[RMAX];
			__syncthreads();                                     // skv / ckv (and, on a re-run, the previous rows' tiles) are complete
			// (a lane whose j-th row would lie past the slice repeats the slice's last row: it stores the same value to the same slot
			// as the lane that owns that row, and the row loop needs no per-row predicate -- seven straight-line chains)
#pragma unroll
			for (int j = 0; j < RMAX; ++j) {
				int k = r0 + (grp > 0 ? grp : 0) + j * ngrp;
				k = k < r1 ? k : r1 - 1;
				skr[j] = held ? skv[k] : 0.0f; ckr[j] = held ? D(ckv[k]) : 0.0;
			}
			auto produce = [&](int t) {
				const int e = t * T + col;
				if (a < 0 || grp >= ngrp) return;
				float *buf = stile + (t & 1) * BUF - r0 * TS;         // row k of the slice at (k - r0) * TS
				if (e >= NNODE_SIGMA) {                              // pad the last tile: x + 0.0f == x
					for (int k = r0 + grp; k < r1; k += ngrp) buf[k * TS + col] = 0.0f;
					return;
				}
				const int j2 = e / NTHETA_SIGMA, j1 = e - j2 * NTHETA_SIGMA;
				const double cp = cphid[j2], ct = cthd[j1];
				const float s1 = sh[j1], nd = ndf_tab[e], w = ui[j1];
				if (held) {
#pragma unroll
					for (int j = 0; j < RMAX; ++j) {
						if (j >= nrows) break;                       // uniform
						int k = r0 + grp + j * ngrp;
						k = k < r1 ? k : r1 - 1;
						float kh = F(D(skr[j] * s1) * cp + ckr[j] * ct);
						buf[k * TS + col] = fmax_(0.0f, kh) * nd * w * s1;
					}
				} else {
					for (int k = r0 + grp; k < r1; k += ngrp) {
						float kh = F(D(skv[k] * s1) * cp + D(ckv[k]) * ct);
						buf[k * TS + col] = fmax_(0.0f, kh) * nd * w * s1;
					}
				}
			};
			__syncthreads();
			produce(0);
			__syncthreads();
			float nint = 0.0f;                                   // row accumulator of thread tid (r0 <= tid < r1)
#ifdef DJB_EXP_FIT_TS
			unsigned long long acc_work = 0, t_loop0 = wall_clock64();
#endif
			for (int t = 0; t < ntiles; ++t) {
#ifdef DJB_EXP_FIT_TS
				const unsigned long long t_w0 = wall_clock64();
#endif
				if (tid >= n_sum) { if (t + 1 < ntiles) produce(t + 1); }
				else if (tid >= r0 && tid < r1) {
					// the tile's T terms of this row: all loads first (8 x 16 bytes in flight), then the adds in
					// the reference's order -- a load-add-load-add chain exposes the LDS latency 64 times per tile
					const float4 *row = (const float4 *)(stile + (t & 1) * BUF + (tid - r0) * TS);
					for (int c = 0; c < T / 4; c += 8) {
						float4 v[8];
#pragma unroll
						for (int j = 0; j < 8; ++j) v[j] = row[c + j];
#pragma unroll
						for (int j = 0; j < 8; ++j) { nint += v[j].x; nint += v[j].y; nint += v[j].z; nint += v[j].w; }   // zero-padded last tile: x + 0.0f == x
					}
				}
#ifdef DJB_EXP_FIT_TS
				acc_work += wall_clock64() - t_w0;
#endif
				__syncthreads();
			}
#ifdef DJB_EXP_FIT_TS
			if (blockIdx.x == gridDim.x - 1) {
				if (tid == n_sum) { g_fit_ts[10] = acc_work; g_fit_ts[12] = wall_clock64() - t_loop0; }      // first producer lane
				if (tid == r0) g_fit_ts[11] = acc_work;                                                     // first row owner
			}
#endif
			if (tid >= r0 && tid < r1) { nint *= dth * dph; sigma[tid] = fmax_(ckv[tid], nint); }
		};
		const int r0 = (cnt * part) / parts, r1 = (cnt * (part + 1)) / parts;
		DJB_FIT_TS(13);
		sigma_rows(r0, r1);
		DJB_FIT_TS(14);
		if (parts > 1) {
			// every workgroup of the material publishes its rows; with few materials (fresnel_split) all of them
			// pick the others' up and go on to take a slice of the Fresnel-ratio pass, which needs the whole
			// table; otherwise only the main workgroup does and the helpers leave.  All waits are bounded: a
			// workgroup whose partners are late (not resident yet, e.g. the device is shared) computes the
			// missing rows itself -- same arithmetic, same values -- so neither progress nor the result
			// depends on scheduling.
			float *sx = split.sig_x + (size_t)m * res;
			unsigned int *done = split.sig_done + 2 * m;
			// One release per workgroup, not one fence per thread: the barrier orders the row owners' stores before thread 0's
			// release (release is cumulative), and an agent-scope release writes the XCD's dirty L2 lines back -- with 200
			// workgroups' register spills in there, 1 024 of them per workgroup cost the material's main workgroup a 78 us wait
			if (tid >= r0 && tid < r1) sx[tid] = sigma[tid];
			__syncthreads();
			if (tid == 0) __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
			if (!fresnel_split && part > 0) return;          // sigma-only slicing: the helper is done
			if (tid == 0) {
				int spins = 0;
				while (__hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned int)parts && ++spins < (1 << 14))
					__builtin_amdgcn_s_sleep(8);
				s_have = spins < (1 << 14);
				// one fence; the barrier extends it to the workgroup.  A full one (write-back, then invalidate): this workgroup's own
				// rows share cache lines with the rows it is about to read, and its L2 holds them dirty
				__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
			}
			__syncthreads();
			if (s_have) {
				if (tid < cnt && !(tid >= r0 && tid < r1))
					sigma[tid] = __uint_as_float(__hip_atomic_load((const unsigned int *)sx + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
			} else {
				for (int q = 0; q < r0; q += rcap) sigma_rows(q, q + rcap < r0 ? q + rcap : r0);
				for (int q = r1; q < cnt; q += rcap) sigma_rows(q, q + rcap < cnt ? q + rcap : cnt);
			}
		}
	}
	__syncthreads();
	if (tid == 0) sigma[cnt] = sigma[cnt - 1];
	__syncthreads();

	DJB_FIT_TS(3);
	// ================================================================ compute_fresnel (dj_brdf.h:2583-2641)
	// pair (i, j) runs iff theta_h(j-1) < pi/2 - theta_d(i)   (theta_h(-1) := 0): 5 456 of the 8 010 pairs at res 90, and for a
	// given i the valid j are a prefix [0, nj(i)).
	const int n_pairs = cnt * (cnt + 1);
	auto fresnel_pair = [&](int i, int j, int e) {
		const float qnan = __builtin_nanf("");
		float rx = qnan, ry = qnan, rz = qnan;
		v3 dir_i, dir_o;
		bool valid;
		EvalHints hints; const EvalHints *hp = nullptr;
		if (split.fres_dirs) {                                   // tabulated once per context and resolution (k_fit_fresnel_dirs)
			const float *d = split.fres_dirs + FRES_REC * (size_t)e;
			dir_o = mk(d[0], d[1], d[2]); dir_i = mk(0, 0, 1);
			valid = dir_o.x == dir_o.x;
			hints.u_sigma_o = d[3]; hints.u_ndf_h = d[4]; hints.u_sigma_i = split.fres_dirs[FRES_REC * (size_t)n_pairs];
			hp = &hints;
		} else valid = fit_fresnel_dirs(i, j, cnt, dir_i, dir_o);
		if (valid) {
			v3 fr1 = src_eval<SRC>(src, std_p, dir_i, dir_o, cnt + e, slot_bin);
			v3 fr2; float pdf;
			mf_eval_pdf<KIND_TABULAR, 1>(self, std_p, dir_i, dir_o, fr2, pdf, hp);
			if (D(fr2.x) > 1e-4) rx = fr1.x / fr2.x;
			if (D(fr2.y) > 1e-4) ry = fr1.y / fr2.y;
			if (D(fr2.z) > 1e-4) rz = fr1.z / fr2.z;
		}
		ratio[3 * (size_t)e] = rx; ratio[3 * (size_t)e + 1] = ry; ratio[3 * (size_t)e + 2] = rz;
	};
	if (!fresnel_split) {
		// One workgroup does the whole pass (eight pairs per lane at res 90): the valid pairs are enumerated -- prefix sums over
		// i, a 7-step search per work item -- so that every lane evaluates one; a third of the lanes used to idle through
		// skipped pairs (100 materials: 161 -> 135 us).  The skipped pairs read as "no sample" (NaN) in the row sums below.
		int *foff = (int *)stile;                                // [cnt + 1]; the sigma tiles are done with
		if (split.fres_dirs) {                                   // tabulated with the directions
			const int *g = (const int *)(split.fres_dirs + FRES_REC * (size_t)n_pairs + 1);
			for (int i = tid; i <= cnt; i += FIT_BLOCK) foff[i] = g[i];
		} else {
			for (int i = tid; i < cnt; i += FIT_BLOCK) {
				int nj = 0; float td, th;
				while (nj <= cnt && fit_fresnel_valid(i, nj, cnt, td, th)) ++nj;
				foff[i + 1] = nj;
			}
			__syncthreads();
			if (tid == 0) { foff[0] = 0; for (int i = 0; i < cnt; ++i) foff[i + 1] += foff[i]; }
		}
		__syncthreads();
		const int n_valid = foff[cnt];
		DJB_FIT_TS(8);
		const float qnan = __builtin_nanf("");
		for (int e = tid; e < n_pairs; e += FIT_BLOCK) {
			const int i = e / (cnt + 1), j = e - i * (cnt + 1);
			if (j >= foff[i + 1] - foff[i]) { ratio[3 * (size_t)e] = qnan; ratio[3 * (size_t)e + 1] = qnan; ratio[3 * (size_t)e + 2] = qnan; }
		}
		for (int q = tid; q < n_valid; q += FIT_BLOCK) {
			int lo = 0, hi = cnt - 1;                            // the row i with foff[i] <= q < foff[i + 1]
			while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (foff[mid] <= q) lo = mid; else hi = mid - 1; }
			const int j = q - foff[lo];
			fresnel_pair(lo, j, lo * (cnt + 1) + j);
		}
		__threadfence_block();
		__syncthreads();
		DJB_FIT_TS(9);
	} else {
		// this workgroup's slice of the (theta_d, theta_h) pairs (one pair per lane); helpers hand theirs over and leave
		auto fresnel_pairs = [&](int e0, int e1) {
			for (int e = e0 + tid; e < e1; e += FIT_BLOCK) { const int i = e / (cnt + 1); fresnel_pair(i, e - i * (cnt + 1), e); }
		};
		const int e0 = (int)(((long long)n_pairs * part) / parts), e1 = (int)(((long long)n_pairs * (part + 1)) / parts);
		fresnel_pairs(e0, e1);
		__syncthreads();                                         // (as above: thread 0's release below covers the workgroup's stores)
		unsigned int *done = split.sig_done + 2 * m + 1;
		if (part > 0) {
			if (tid == 0) __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
			return;
		}
		if (tid == 0) {
			int spins = 0;
			while (__hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned int)(parts - 1) && ++spins < (1 << 14))
				__builtin_amdgcn_s_sleep(8);
			s_have = spins < (1 << 14);
			__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");     // as above: the slices' ratios meet inside cache lines
		}
		__syncthreads();
		if (!s_have) { fresnel_pairs(e1, n_pairs); __threadfence_block(); }
		__syncthreads();
	}
	for (int i = tid; i < cnt; i += FIT_BLOCK) {
		float fx = 0, fy = 0, fz = 0; int cx = 0, cy = 0, cz = 0;
		// ten steps' ratios are requested together (the adds keep their order): one L2 round trip per ten steps instead of per step
		const float *row = ratio + 3 * (size_t)i * (cnt + 1);
		for (int j0 = 0; j0 <= cnt; j0 += 10) {
			float v[30];
#pragma unroll
			for (int t = 0; t < 10; ++t) {
				const float *r = row + 3 * (j0 + t <= cnt ? j0 + t : cnt);
				v[3 * t] = r[0]; v[3 * t + 1] = r[1]; v[3 * t + 2] = r[2];
			}
#pragma unroll
			for (int t = 0; t < 10; ++t) {
				const bool on = j0 + t <= cnt;
				const float rx = v[3 * t], ry = v[3 * t + 1], rz = v[3 * t + 2];
				if (on && rx == rx) { fx += rx; ++cx; }
				if (on && ry == ry) { fy += ry; ++cy; }
				if (on && rz == rz) { fz += rz; ++cz; }
			}
		}
		fres[3 * i] = cx == 0 ? 1.0f : fmin_(1.0f, fx / (float)cx);
		fres[3 * i + 1] = cy == 0 ? 1.0f : fmin_(1.0f, fy / (float)cy);
		fres[3 * i + 2] = cz == 0 ? 1.0f : fmin_(1.0f, fz / (float)cz);
	}
	__syncthreads();
	if (tid < 3) fres[3 * cnt + tid] = fres[3 * (cnt - 1) + tid];
	__syncthreads();

	DJB_FIT_TS(4);
	// ================================================================ compute_cdf (dj_brdf.h:2705-2727)
	for (int k = tid; k < cnt; k += FIT_BLOCK) {
		float u = (float)k / (float)cnt;
		float th = F(D(u * u) * DJB_PI * 0.5);
		float c = cos_f(th), r = tan_f(th);
		float pr = p22_radial<KIND_TABULAR>(self, r * r);
		qprobe[k] = (u * r * pr) / (c * c);       // staging (qprobe has 8*cnt slots)
	}
	__syncthreads();
	if (tid == 0) {
		const float dth = F(DJB_PI / D((float)cnt));
		float nint = 0.0f;
		for (int k = 0; k < cnt; ++k) { nint += qprobe[k]; cdf[k] = F(D(nint * dth) * (2.0 * DJB_PI)); }
		cdf[cnt] = 1.0f;
	}
	__syncthreads();

	DJB_FIT_TS(5);
	// ================================================================ compute_qf (dj_brdf.h:2731-2762)
	const int qres = cnt * 8;
	for (int j = tid; j < qres; j += FIT_BLOCK) {
		float u = (float)j / (float)qres;
		float th = F(D(u) * DJB_PI * 0.5);
		qprobe[j] = tab_cdf_radial(self, tan_f(th));
	}
	if (tid == 0) s_nqf = 0;
	__syncthreads();
	// The reference scans forward with a j that persists across i (dj_brdf.h:2735): entry i is the first j >= j(i-1) with
	// probe[j] >= i / cnt.  The thresholds grow with i, so the sets {j : probe[j] >= c_i} shrink and that first j is simply the
	// first j of set i -- no dependence on i - 1 -- and the i that find one are a prefix 1 .. I (an i without one leaves j at
	// the end for all later i).  One lane per i scans the 8 cnt probes (independent LDS reads) instead of one lane doing all.
	if (tid >= 1 && tid < cnt) {
		const float c = (float)tid / (float)cnt;
		int j = 0;
		for (; j + 16 <= qres; j += 16) {                        // 16 probes per step: the reads do not wait for each other
			bool any = false;
#pragma unroll
			for (int t = 0; t < 16; ++t) any |= qprobe[j + t] >= c;
			if (any) break;
		}
		while (j < qres && !(qprobe[j] >= c)) ++j;
		if (j < qres) { qf[tid] = (float)j / (float)qres; atomicAdd(&s_nqf, 1); }
	}
	__syncthreads();
	if (tid == 0) {
		const int found = s_nqf;                                 // entries 1 .. found are set
		qf[0] = 0.0f; qf[found + 1] = 1.0f;
		s_nqf = found + 2;
		for (int k = found + 2; k < res; ++k) qf[k] = 0.0f;
	}
	__syncthreads();

	DJB_FIT_TS(6);
	// ================================================================ fits (dj_brdf.h:3133-3184)
	for (int k = tid; k < NTHETA_FIT; k += FIT_BLOCK) {
		float u = (float)k / (float)NTHETA_FIT;
		float th = F(D(u * u) * DJB_PI * 0.5);
		float c = cos_f(th), r = tan_f(th);
		float r2 = r * r;
		float pr = p22_radial<KIND_TABULAR>(self, r2);
		terms[k] = (u * r2 * r * pr) / (c * c);
		terms[NTHETA_FIT + k] = (u * r2 * pr) / (c * c);
	}
	__syncthreads();
	if (tid == 0) {
		const float dth = F(DJB_PI / D((float)NTHETA_FIT));
		float nb = 0.0f, ng = 0.0f;
		for (int k = 0; k < NTHETA_FIT; ++k) { nb += terms[k]; ng += terms[NTHETA_FIT + k]; }
		nb = F(D(nb) * (D(dth) * DJB_PI));
		ng = F(D(ng) * (D(dth) * 4.0));
		out.alpha_beckmann[m] = F(sqrt(2.0 * D(nb)));
		out.alpha_ggx[m] = ng;
		out.n_qf[m] = s_nqf;
	}
	for (int k = tid; k < res; k += FIT_BLOCK) {
		size_t o = (size_t)m * res + k;
		out.p22[o] = p22[k]; out.sigma[o] = sigma[k]; out.cdf[o] = cdf[k]; out.qf[o] = qf[k];
		out.fresnel[3 * o] = fres[3 * k]; out.fresnel[3 * o + 1] = fres[3 * k + 1];
		out.fresnel[3 * o + 2] = fres[3 * k + 2];
	}
	DJB_FIT_TS(7);
}

// nint[io * cnt + jh] = dphi * sum over the float-stepped phi of max(1, tan(theta_h) tan(theta_o) cos(phi)) (compute_p22_smith,
// dj_brdf.h:2482-2522): the same operations in the same order as k_fit's own loops, one workgroup
__global__ __launch_bounds__(1024) void k_fit_smith_nint(int res, float *nint)
{
	__shared__ float s_tan[256], s_cphi[MAX_PHI_STEPS];
	__shared__ int s_n;
	const int cnt = res - 1, tid = threadIdx.x;
	const float dphi_h = F(DJB_PI / 180.0);
	if (tid == 0) {
		int c = 0;
		for (float phi = 0.0f; D(phi) < 2.0 * DJB_PI && c < MAX_PHI_STEPS; phi += dphi_h) s_cphi[c++] = phi;
		s_n = c;
	}
	for (int k = tid; k < cnt && k < 256; k += 1024) { const float th = fit_backscatter_theta(k, cnt); s_tan[k] = tan_f(th * th); }
	__syncthreads();
	const int nphi = s_n;
	for (int k = tid; k < nphi; k += 1024) s_cphi[k] = cos_f(s_cphi[k]);
	__syncthreads();
	for (int e = tid; e < cnt * cnt; e += 1024) {
		const int io = e / cnt, jh = e - io * cnt;
		const float tan_product = s_tan[jh] * s_tan[io];
		float acc = 0.0f;
		for (int q = 0; q < nphi; ++q) acc += fmax_(1.0f, tan_product * s_cphi[q]);
		nint[e] = acc * dphi_h;
	}
}

// One record per pair of the Fresnel-ratio pass: dir_o (dir_i is (0, 0, 1) for all of them, dj_brdf.h:2609; x = NaN: the reference
// skips the pair) and the two table coordinates of the fitted lobe that the pair's geometry fixes -- sigma's for dir_o, the NDF's for
// the half vector -- plus, after the last record, sigma's coordinate for dir_i, the res prefix sums that enumerate the valid pairs and
// the NDF's coordinate at the 16 200 nodes of the sigma quadrature; then the (res - 1)^2 phi integrals of the K matrix (k_fit_smith_nint)
// and the MERL bin of each of the (res - 1) (res + 1) query slots.
// Everything in them is independent of the material.
__global__ __launch_bounds__(256) void k_fit_fresnel_dirs(int res, Params std_p, float *recs)
{
	const int cnt = res - 1, n_pairs = cnt * (cnt + 1), e = blockIdx.x * 256 + threadIdx.x;
	if (e == 0) {
		recs[FRES_REC * (size_t)n_pairs] = mf_sigma_table_u(mk(0, 0, 1), std_p);
		// prefix sums of the number of valid theta_h per theta_d (for a given theta_d they are a prefix): the enumeration of the valid pairs
		int *foff = (int *)(recs + FRES_REC * (size_t)n_pairs + 1);
		foff[0] = 0;
		for (int i2 = 0; i2 < cnt; ++i2) {
			int nj = 0; float td, th;
			while (nj <= cnt && fit_fresnel_valid(i2, nj, cnt, td, th)) ++nj;
			foff[i2 + 1] = foff[i2] + nj;
		}
	}
	if (e < NNODE_SIGMA) {           // the NDF's table coordinate at each node of the sigma quadrature (the nodes do not depend on res)
		const int j2 = e / NTHETA_SIGMA, j1 = e - j2 * NTHETA_SIGMA;
		const float phi_h = F(D((float)j2 / (float)NPHI_SIGMA) * 2.0 * DJB_PI), u = (float)j1 / (float)NTHETA_SIGMA;
		recs[FRES_REC * (size_t)n_pairs + 1 + res + e] = mf_ndf_table_u(from_angles(F(D(u * u) * DJB_PI * 0.5), phi_h), std_p);
	}
	if (e < cnt + n_pairs) {         // the MERL bin of every query slot of a fit at this resolution (-1: a pair the reference skips)
		int *bins = (int *)(recs + FRES_REC * (size_t)n_pairs + 1 + res + NNODE_SIGMA + (size_t)cnt * cnt);
		const int b = fit_merl_slot_index(e, res);
		bins[e] = b < 0 ? 0 : b;
	}
	if (e >= n_pairs) return;
	const int i = e / (cnt + 1), j = e - i * (cnt + 1);
	v3 dir_i, dir_o;
	float *r = recs + FRES_REC * (size_t)e;
	if (!fit_fresnel_dirs(i, j, cnt, dir_i, dir_o)) { r[0] = __builtin_nanf(""); r[1] = r[2] = r[3] = r[4] = 0.0f; return; }
	r[0] = dir_o.x; r[1] = dir_o.y; r[2] = dir_o.z;
	r[3] = mf_sigma_table_u(dir_o, std_p);
	r[4] = mf_ndf_table_u(normalize(add(dir_i, dir_o)), std_p);
}

// the MERL table index each query slot reads (the file pipeline gathers exactly these entries on the host)
__global__ __launch_bounds__(256) void k_fit_merl_slots(int res, int n_slots, int32_t *idx)
{
	int s = blockIdx.x * 256 + threadIdx.x;
	if (s < n_slots) idx[s] = fit_merl_slot_index(s, res);
}

template <int SRC>
hipError_t launch_fit_kind(hipStream_t s, const Brdf *srcs, const Params &std_p, int n_mat, int res,
                           int shadow, double *km, float *ratio, const djbk::FitOut &out, const djbk::FitSplit &split)
{
	size_t lds = (size_t)make_plan(res).total;
	hipError_t e = hipFuncSetAttribute((const void *)k_fit<SRC>,
	                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
	if (e != hipSuccess) return e;
	if (split.parts > 1 && (e = hipMemsetAsync(split.sig_done, 0, sizeof(unsigned int) * 2 * n_mat, s)) != hipSuccess) return e;
	hipLaunchKernelGGL((k_fit<SRC>), dim3(n_mat * split.parts), dim3(FIT_BLOCK), lds, s, srcs, std_p, n_mat, res, shadow,
	                   km, ratio, out, split);
#ifdef DJB_EXP_FIT_TS
	{
		unsigned long long h[16];
		(void)hipStreamSynchronize(s); (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_fit_ts), sizeof h);
		fprintf(stderr, "djb_exp: k_fit phases (us, last workgroup of %d x %d): p22_smith %.1f normalize %.1f sigma %.1f fresnel %.1f cdf %.1f qf %.1f fits %.1f\n",
		        n_mat, split.parts, (h[1] - h[0]) * 0.01, (h[2] - h[1]) * 0.01, (h[3] - h[2]) * 0.01, (h[4] - h[3]) * 0.01, (h[5] - h[4]) * 0.01,
		        (h[6] - h[5]) * 0.01, (h[7] - h[6]) * 0.01);
		fprintf(stderr, "djb_exp:   [%d x %d] sigma: tables + NDF fill %.1f us, own rows %.1f, exchange with the other slices %.1f\n", n_mat, split.parts, (h[13] - h[2]) * 0.01, (h[14] - h[13]) * 0.01, (h[3] - h[14]) * 0.01);
		fprintf(stderr, "djb_exp:   [%d x %d] sigma tile loop %.1f us: a producer wave busy %.1f, a row owner busy %.1f (the rest of each: waiting at the barrier)\n",
		        n_mat, split.parts, h[12] * 0.01, h[10] * 0.01, h[11] * 0.01);
		fprintf(stderr, "djb_exp:   fresnel (unsplit): prefix %.1f pairs %.1f row sums %.1f\n", (h[8] - h[3]) * 0.01, (h[9] - h[8]) * 0.01, (h[4] - h[9]) * 0.01);
	}
#endif
	return hipGetLastError();
}

} // namespace

namespace djbk {

size_t fit_lds_bytes(int res) { return (size_t)make_plan(res).total; }

int fit_merl_slots(int res) { return fit_merl_slot_count(res); }
size_t fit_fresnel_dirs_floats(int res) { return (size_t)FRES_REC * (res - 1) * res + 1 + (size_t)res + NNODE_SIGMA + (size_t)(res - 1) * (res - 1) + (size_t)(res - 1) * (res + 1); }
hipError_t launch_fit_fresnel_dirs(hipStream_t s, int res, const Params &std_p, float *recs)
{
	const int n = (res - 1) * (res + 1) > NNODE_SIGMA ? (res - 1) * (res + 1) : NNODE_SIGMA;
	hipLaunchKernelGGL(k_fit_fresnel_dirs, dim3((n + 255) / 256), dim3(256), 0, s, res, std_p, recs);
	if (res - 1 > 256) return hipErrorInvalidValue;        // k_fit_smith_nint's tangent table (the fit itself is built for res <= ~100)
	hipLaunchKernelGGL(k_fit_smith_nint, dim3(1), dim3(1024), 0, s, res, recs + FRES_REC * (size_t)(res - 1) * res + 1 + res + NNODE_SIGMA);
	return hipGetLastError();
}
hipError_t launch_fit_merl_slots(hipStream_t s, int res, int32_t *idx)
{
	const int n = fit_merl_slot_count(res);
	hipLaunchKernelGGL(k_fit_merl_slots, dim3((n + 255) / 256), dim3(256), 0, s, res, n, idx);
	return hipGetLastError();
}

// One workgroup occupies a CU (124 KB of LDS at res 90); each extra slice shortens the producers' share of a tile,
// the row owners' 64 dependent adds per tile stay.
int fit_parts(int n_mat, int n_cus)
{
	if (const char *e = getenv("DJB_FIT_PARTS")) { int v = atoi(e); if (v >= 1 && v <= 8) return v; }   // experiments
	int p = n_mat > 0 ? n_cus / n_mat : 1;
	return p < 1 ? 1 : p > 8 ? 8 : p;
}

hipError_t launch_fit(hipStream_t s, const Brdf *srcs, int src_kind, const Params &std_p, int n_mat,
                      int res, int shadow, double *km, float *ratio, const FitOut &out, const FitSplit &split)
{
	switch (src_kind) {
	case KIND_BECKMANN: return launch_fit_kind<KIND_BECKMANN>(s, srcs, std_p, n_mat, res, shadow, km, ratio, out, split);
	case KIND_GGX:      return launch_fit_kind<KIND_GGX>(s, srcs, std_p, n_mat, res, shadow, km, ratio, out, split);
	case KIND_TABULAR:  return launch_fit_kind<KIND_TABULAR>(s, srcs, std_p, n_mat, res, shadow, km, ratio, out, split);
	case KIND_TABULAR_ANISO: return launch_fit_kind<KIND_TABULAR_ANISO>(s, srcs, std_p, n_mat, res, shadow, km, ratio, out, split);
	case KIND_MERL:     return launch_fit_kind<KIND_MERL>(s, srcs, std_p, n_mat, res, shadow, km, ratio, out, split);
	case KIND_UTIA:     return launch_fit_kind<KIND_UTIA>(s, srcs, std_p, n_mat, res, shadow, km, ratio, out, split);
	case KIND_LAMBERT:  return launch_fit_kind<KIND_LAMBERT>(s, srcs, std_p, n_mat, res, shadow, km, ratio, out, split);
	case KIND_SGD:      return launch_fit_kind<KIND_SGD>(s, srcs, std_p, n_mat, res, shadow, km, ratio, out, split);
	case KIND_ABC:      return launch_fit_kind<KIND_ABC>(s, srcs, std_p, n_mat, res, shadow, km, ratio, out, split);
	}
	return hipErrorInvalidValue;
}

} // namespace djbk
