// djb_kernels_fit_aniso.hip -- djb::tabular_anisotropic on gfx950 (dj_brdf.h:428-478):
//   ctor = compute_p22_smith (power iteration on an (w*h)^2 Smith kernel) -> normalize_p22 ->
//   compute_sigma -> compute_fresnel -> pdf1/cdf1/qf1 -> pdf2/cdf2/qf2; fit_{beckmann,ggx}_parameters.
//
// At the reference's usual 90 x 90 resolution the kernel matrix is 8010^2 doubles = 513 MB on the
// CPU (dj_brdf.h:2531-2532).  Here it is never stored: K(a, b) = k1(a) * k2(a, b) is recomputed
// inside the matvec from 8 floats per row/column, and every sum keeps the
// reference's order -- rows (theta_o/phi_o nodes, theta_k nodes, azimuths) are the parallel
// dimension, one lane per row, exactly as in djb_kernels_fit.hip.  A fit is a short sequence of
// launches on one stream (phases need a grid-wide dependency); each launch is a few microseconds
// to a few hundred microseconds.
#include "djb_internal.hpp"

using namespace djbdev;

namespace {

constexpr int BLOCK = 256;
constexpr int NT_SIG = 45, NP_SIG = 90;        // compute_sigma quadrature (dj_brdf.h:2390-2391)
constexpr int NT_NORM = 128, NP_NORM = 256;    // normalize_p22            (dj_brdf.h:2308-2309)
constexpr int NT_FIT = 128, NP_FIT = 512;      // moment fits              (dj_brdf.h:3189-3190)

using djbk::AnisoScratch;

inline int blocks_for(long long n) { long long b = (n + BLOCK - 1) / BLOCK; return (int)(b < 1 ? 1 : b); }

DJB_DEV Brdf self_view(const AnisoScratch &S, int shadow)
{
	Brdf b;
	b.kind = KIND_TABULAR_ANISO; b.shadow = shadow;
	b.fr.kind = FR_IDEAL; b.fr.pts = nullptr; b.fr.npts = 0;
	b.p22 = S.p22; b.sigma = S.sigma; b.cdf = nullptr; b.qf = nullptr;
	b.n_p22 = b.n_sigma = S.elev * S.azim; b.n_cdf = b.n_qf = 0;
	b.merl = nullptr; b.merl_sparse = 0; b.utia = nullptr; b.model = nullptr; b.exp_lds = 0u; b.pow_lds = 0u; b.atan_lds = 0u;
	b.a_pdf1 = S.pdf1; b.a_cdf1 = S.cdf1; b.a_qf1 = S.qf1; b.a_pdf2 = S.pdf2; b.a_cdf2 = S.cdf2; b.a_qf2 = S.qf2;
	b.elev = S.elev; b.azim = S.azim; b.n_a_cdf1 = S.azim; b.n_a_qf1 = S.azim;
	return b;
}

// slot: the query slot of this look-up (djb_cpu_aniso.inc: aniso_query_dirs); a per-slot source (Brdf::merl_sparse: the samples of a
// user-defined brdf, djb_brdf_create_tabular_anisotropic_from_samples) holds one rgb per slot
template <int SRC>
DJB_DEV v3 src_eval(const Brdf &src, const Params &std_p, v3 i, v3 o, int slot)
{
	v3 fr = mk(0, 0, 0); float pdf;
	if (SRC <= KIND_TABULAR || SRC == KIND_TABULAR_ANISO) mf_eval_pdf<SRC, 1>(src, std_p, i, o, fr, pdf);
	else if (SRC == KIND_MERL) {
		if (src.merl_sparse) { MerlTexel t = src.merl[slot]; fr = mk(t.x, t.y, t.z); }
		else fr = merl_eval(src, i, o);
	}
	else if (SRC == KIND_UTIA) fr = utia_eval(src, i, o);
	else if (SRC == KIND_SGD) fr = sgd_eval(src, i, o);
	else if (SRC == KIND_ABC) fr = abc_eval(src, i, o);
	else fr = divs(mk(1, 1, 1), F(DJB_PI));
	(void)pdf;
	return fr;
}

// ---- compute_p22_smith: per-node factors (dj_brdf.h:2535-2566) ------------------------------
template <int SRC>
__global__ __launch_bounds__(BLOCK) void ka_setup(Brdf src, Params std_p, AnisoScratch S)
{
	const int w = S.elev - 1, h = S.azim, N = w * h;
	int a = blockIdx.x * BLOCK + threadIdx.x;
	if (a >= N) return;
	int i2 = a / w, i1 = a - i2 * w;
	const float dtheta = F(sqrt(DJB_PI * 0.5) / D((float)w)), dphi = F(2.0 * DJB_PI / D((float)h));
	float theta = F(D((float)i1 / (float)w) * 0.5 * DJB_PI), phi = F(D((float)i2 / (float)h) * 2.0 * DJB_PI);
	float st = sin_f(theta);
	float zo = cos_f(theta);
	S.zo[a] = zo; S.xo[a] = F(D(st) * glibc_cos(D(phi))); S.yo[a] = F(D(st) * glibc_sin(D(phi)));
	v3 wv = from_angles(theta, phi);
	float fr_i = intensity(src_eval<SRC>(src, std_p, wv, wv, a));
	S.k1[a] = F(D(dtheta * dphi) * (4.0 * D(fr_i) * glibc_pow(D(zo), D(5.0f))));
	float tt = tan_f(theta);
	S.tn[a] = tt; S.dn[a] = zo * zo;               // cos_theta * cos_theta (same float as zo)
	S.s1[a] = F(D(-tt) * glibc_cos(D(phi))); S.s2[a] = F(D(-tt) * glibc_sin(D(phi)));
	S.v0[a] = 1.0;
}

// ---- one step of matrix::eigenvector: out[a] = sum_b double(float(k1[a]*k2(a,b))) * v[b] -------
// The N x N kernel matrix (8010 x 8010 at 90 x 90) is never stored: its entries are recomputed from
// 8 floats per node.  A workgroup owns MV_ROWS rows.  Fourteen producer wave-halves compute the
// products of the next MV_TILE columns (thread = row x column group, the row's factors stay in
// registers, an IEEE division per entry) into one LDS buffer while the row owners add the previous
// tile's products in column order from the other buffer: the same products in the same order as
// the one-lane-per-row loop this replaces (1.23 ms per step, latency-bound on one wave per SIMD).
constexpr int MV_BLOCK = 512, MV_ROWS = 32, MV_GROUPS = (MV_BLOCK - 64) / MV_ROWS, MV_PER = 4,
              MV_TILE = MV_GROUPS * MV_PER;

__global__ __launch_bounds__(MV_BLOCK) void ka_matvec(AnisoScratch S, const double *vin, double *vout)
{
	__shared__ double tile[2][MV_ROWS][MV_TILE + 1];
	const int N = (S.elev - 1) * S.azim;
	const int tid = threadIdx.x, row0 = blockIdx.x * MV_ROWS;
	const bool producer = tid >= 64, consumer = tid < MV_ROWS;
	const int p = tid - 64, r = producer ? p % MV_ROWS : tid, g = producer ? p / MV_ROWS : 0;
	const int a = row0 + r;
	const bool valid = a < N;
	float zo = 0, xo = 0, yo = 0, k1 = 0;
	if (producer && valid) { zo = S.zo[a]; xo = S.xo[a]; yo = S.yo[a]; k1 = S.k1[a]; }
	const int ntiles = (N + MV_TILE - 1) / MV_TILE;
	auto produce = [&](int t) {
#pragma unroll
		for (int q = 0; q < MV_PER; ++q) {
			const int c = g * MV_PER + q, b = t * MV_TILE + c;
			double v = 0.0;                                            // padding: acc + 0.0 == acc
			if (valid && b < N) {
				float m_dot_o = zo - xo * S.s1[b] - yo * S.s2[b];
				float k2 = S.tn[b] * fmax_(0.0f, m_dot_o) / S.dn[b];
				v = D(k1 * k2) * vin[b];
			}
			tile[t & 1][r][c] = v;
		}
	};
	if (producer) produce(0);
	__syncthreads();
	double acc = 0.0;
	for (int t = 0; t < ntiles; ++t) {
		if (producer) { if (t + 1 < ntiles) produce(t + 1); }
		else if (consumer) {
			const double *row = tile[t & 1][tid];
			for (int c = 0; c < MV_TILE; ++c) acc += row[c];
		}
		__syncthreads();
	}
	if (consumer && valid) vout[a] = acc;
}

__global__ __launch_bounds__(BLOCK) void ka_p22_grid(AnisoScratch S, const double *v)
{
	const int E = S.elev, w = E - 1, h = S.azim;
	int e = blockIdx.x * BLOCK + threadIdx.x;
	if (e >= E * h) return;
	int j = e / E, i = e - j * E;
	S.p22[e] = i < w ? F(v[j * w + i]) : 0.0f;
}

// ---- normalize_p22 (dj_brdf.h:2306-2338): 256 x 128 terms in parallel, one ordered sum --------
__global__ __launch_bounds__(BLOCK) void ka_norm_terms(AnisoScratch S, int shadow)
{
	int e = blockIdx.x * BLOCK + threadIdx.x;
	if (e >= NP_NORM * NT_NORM) return;
	int j = e / NT_NORM, i = e - j * NT_NORM;
	const Brdf self = self_view(S, shadow);
	float phi = F(D((float)j / (float)NP_NORM) * 2.0 * DJB_PI);
	float theta = F(D((float)i / (float)NT_NORM) * sqrt(DJB_PI * 0.5));
	float ts = theta * theta;
	float c = cos_f(ts);
	float weight = F(D(theta) * glibc_tan(D(ts)) / D(c * c));
	S.terms[e] = weight * aniso_p22_theta_phi(self, ts, phi);
}
// ---- ordered float sums of NACC arrays of M terms each (arrays are M apart, starting at `terms`).
// The reference accumulates these quadratures front to back in one float; a single lane reading HBM
// term by term is latency-bound (14 ns per add).  Here waves 1.. of the block stream tiles of
// OS_TILE terms per array into double-buffered LDS while lane k < NACC of wave 0 adds array k's
// terms in index order (loads batched 16 at a time, then 16 dependent adds).  Returns the sum in
// threads k < NACC.  Launch with BLOCK threads; `buf` holds 2 * NACC * OS_STRIDE floats.
constexpr int OS_TILE = 512, OS_STRIDE = OS_TILE + 1;   // +1: lanes k read array k's row from distinct LDS banks
template <int NACC>
DJB_DEV float ordered_sums(const float *terms, size_t M, float *buf)
{
	const int tid = threadIdx.x;
	constexpr int NLOAD = BLOCK - 64, PER = (OS_TILE + NLOAD - 1) / NLOAD;   // launched with BLOCK threads
	const int ntiles = (int)((M + OS_TILE - 1) / OS_TILE);
	auto fetch = [&](int t) {
		float *dst = buf + (size_t)(t & 1) * NACC * OS_STRIDE;
		float v[NACC][PER];
#pragma unroll
		for (int k = 0; k < NACC; ++k)                                       // all loads first: one latency per tile
#pragma unroll
			for (int j = 0; j < PER; ++j) {
				const int c = tid - 64 + j * NLOAD;
				const size_t e = (size_t)t * OS_TILE + c;
				v[k][j] = (c < OS_TILE && e < M) ? terms[k * M + e] : 0.0f;   // padding: n + 0.0f == n
			}
#pragma unroll
		for (int k = 0; k < NACC; ++k)
#pragma unroll
			for (int j = 0; j < PER; ++j) {
				const int c = tid - 64 + j * NLOAD;
				if (c < OS_TILE) dst[k * OS_STRIDE + c] = v[k][j];
			}
	};
	if (tid >= 64) fetch(0);
	__syncthreads();
	float n = 0.0f;
	for (int t = 0; t < ntiles; ++t) {
		if (tid >= 64) { if (t + 1 < ntiles) fetch(t + 1); }
		else if (tid < NACC) {
			const float *row = buf + (size_t)(t & 1) * NACC * OS_STRIDE + tid * OS_STRIDE;
			for (int c = 0; c < OS_TILE; c += 16) {
				float v[16];
#pragma unroll
				for (int q = 0; q < 16; ++q) v[q] = row[c + q];
#pragma unroll
				for (int q = 0; q < 16; ++q) n += v[q];
			}
		}
		__syncthreads();
	}
	return n;
}

__global__ __launch_bounds__(BLOCK) void ka_norm_apply(AnisoScratch S)
{
	__shared__ float s_k;
	__shared__ float s_buf[2 * OS_STRIDE];
	float k = ordered_sums<1>(S.terms, (size_t)NP_NORM * NT_NORM, s_buf);
	if (threadIdx.x == 0) {
		const float dtheta = F(sqrt(0.5 * DJB_PI) / D((float)NT_NORM)), dphi = F(2.0 * DJB_PI / D((float)NP_NORM));
		k = F(D(k) * (2.0 * D(dtheta) * D(dphi)));
		s_k = F(1.0 / D(k));
	}
	__syncthreads();
	for (int e = threadIdx.x; e < S.elev * S.azim; e += BLOCK) S.p22[e] *= s_k;
}

// ---- compute_sigma (dj_brdf.h:2388-2432) ------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void ka_sigma_tables(AnisoScratch S, Params std_p, int shadow)
{
	int e = blockIdx.x * BLOCK + threadIdx.x;
	const Brdf self = self_view(S, shadow);
	if (e < NT_SIG * NP_SIG) {                       // ndf(vec3(theta^2, phi)) does not depend on k
		int j2 = e / NT_SIG, j1 = e - j2 * NT_SIG;
		float phi = F(D((float)j2 / (float)NP_SIG) * 2.0 * DJB_PI);
		float theta = F(D((float)j1 / (float)NT_SIG) * sqrt(DJB_PI * 0.5));
		S.ndf_tab[e] = mf_ndf<KIND_TABULAR_ANISO>(self, from_angles(theta * theta, phi), std_p);
	}
	if (e < S.azim * NP_SIG) {                       // cos(phi - phi_k): float subtraction, double cosine
		int i2 = e / NP_SIG, j2 = e - i2 * NP_SIG;
		float phi_k = F(D((float)i2 / (float)S.azim) * 2.0 * DJB_PI);
		float phi = F(D((float)j2 / (float)NP_SIG) * 2.0 * DJB_PI);
		S.cosd[e] = glibc_cos(D(phi - phi_k));
	}
	if (e < NT_SIG) {
		float theta = F(D((float)e / (float)NT_SIG) * sqrt(DJB_PI * 0.5));
		float ts = theta * theta;
		S.sig_theta[e] = theta; S.sig_sin[e] = sin_f(ts); S.sig_cosd[e] = glibc_cos(D(ts));
	}
}
__global__ __launch_bounds__(BLOCK) void ka_sigma_rows(AnisoScratch S)
{
	const int E = S.elev, w = E - 1, h = S.azim;
	int a = blockIdx.x * BLOCK + threadIdx.x;
	if (a >= w * h) return;
	int i2 = a / w, i1 = a - i2 * w;
	const float dtheta = F(sqrt(DJB_PI * 0.5) / D((float)NT_SIG)), dphi = F(2.0 * DJB_PI / D((float)NP_SIG));
	float theta_k = F(D((float)i1 / (float)w) * 0.5 * DJB_PI);
	float cos_k = cos_f(theta_k);
	double sin_kd = glibc_sin(D(theta_k));
	float nint = 0.0f;
	for (int j2 = 0; j2 < NP_SIG; ++j2) {
		double cp = S.cosd[i2 * NP_SIG + j2];
		for (int j1 = 0; j1 < NT_SIG; ++j1) {
			float sin_t = S.sig_sin[j1];
			float m_dot_k = F(sin_kd * D(sin_t) * cp + D(cos_k) * S.sig_cosd[j1]);
			float weight = S.sig_theta[j1] * sin_t;
			float masking = fmax_(0.0f, m_dot_k) * S.ndf_tab[j2 * NT_SIG + j1];
			nint += weight * masking;
		}
	}
	nint = F(D(nint) * (2.0 * D(dtheta) * D(dphi)));
	float v = fmax_(cos_k, nint);
	S.sigma[i1 + E * i2] = v;
	if (i1 == w - 1) S.sigma[w + E * i2] = v;        // m_sigma.push_back(m_sigma.back())
}

// ---- compute_fresnel (dj_brdf.h:2643-2701), as in the isotropic kernel -------------------------
template <int SRC>
__global__ __launch_bounds__(BLOCK) void ka_fres_pairs(Brdf src, Params std_p, AnisoScratch S, int shadow)
{
	const int cnt = S.elev - 1;
	int e = blockIdx.x * BLOCK + threadIdx.x;
	if (e >= cnt * (cnt + 1)) return;
	int i = e / (cnt + 1), j = e - i * (cnt + 1);
	const Brdf self = self_view(S, shadow);
	float theta_d = F(D((float)i / (float)cnt) * DJB_PI * 0.5);
	float prev = 0.0f;
	if (j > 0) { float t1 = (float)(j - 1) / (float)cnt; prev = F(D(t1 * t1) * DJB_PI * 0.5); }
	float t1 = (float)j / (float)cnt;
	float theta_h = F(D(t1 * t1) * DJB_PI * 0.5);
	const float qnan = __builtin_nanf("");
	float rx = qnan, ry = qnan, rz = qnan;
	if (D(prev) < DJB_PI * 0.5 - D(theta_d) && !(D(theta_h) > DJB_PI * 0.5)) {
		v3 dir_h = from_angles(theta_h, 0.0f), dir_d = from_angles(theta_d, F(DJB_PI * 0.5));
		v3 dir_i, dir_o;
		hd_to_io(dir_h, dir_d, dir_i, dir_o);
		dir_i = mk(0, 0, 1);
		v3 fr1 = src_eval<SRC>(src, std_p, dir_i, dir_o, cnt * S.azim + e);
		v3 fr2; float pdf;
		mf_eval_pdf<KIND_TABULAR_ANISO, 1>(self, std_p, dir_i, dir_o, fr2, pdf);
		if (D(fr2.x) > 1e-4) rx = fr1.x / fr2.x;
		if (D(fr2.y) > 1e-4) ry = fr1.y / fr2.y;
		if (D(fr2.z) > 1e-4) rz = fr1.z / fr2.z;
	}
	S.ratio[3 * (size_t)e] = rx; S.ratio[3 * (size_t)e + 1] = ry; S.ratio[3 * (size_t)e + 2] = rz;
}
__global__ __launch_bounds__(BLOCK) void ka_fres_rows(AnisoScratch S)
{
	const int cnt = S.elev - 1;
	int i = blockIdx.x * BLOCK + threadIdx.x;
	if (i >= cnt) return;
	float fx = 0, fy = 0, fz = 0; int cx = 0, cy = 0, cz = 0;
	for (int j = 0; j <= cnt; ++j) {
		const float *r = S.ratio + 3 * ((size_t)i * (cnt + 1) + j);
		float rx = r[0], ry = r[1], rz = r[2];
		if (rx == rx) { fx += rx; ++cx; }
		if (ry == ry) { fy += ry; ++cy; }
		if (rz == rz) { fz += rz; ++cz; }
	}
	float ox = cx == 0 ? 1.0f : fmin_(1.0f, fx / (float)cx);
	float oy = cy == 0 ? 1.0f : fmin_(1.0f, fy / (float)cy);
	float oz = cz == 0 ? 1.0f : fmin_(1.0f, fz / (float)cz);
	S.fres[3 * i] = ox; S.fres[3 * i + 1] = oy; S.fres[3 * i + 2] = oz;
	if (i == cnt - 1) { S.fres[3 * cnt] = ox; S.fres[3 * cnt + 1] = oy; S.fres[3 * cnt + 2] = oz; }
}

// nint += (val * tan(theta)) / (cos_theta * cos_theta): sum carried in double, rounded per step
DJB_DEV float acc_tan_over_cos2(float nint, float val, double tan_d, float c2) { return F(D(nint) + (D(val) * tan_d) / D(c2)); }

// ---- compute_pdf1 + normalize_pdf1 (dj_brdf.h:2849-2875, 3038-3058); one block --------------
__global__ __launch_bounds__(BLOCK) void ka_pdf1(AnisoScratch S, int shadow)
{
	__shared__ double s_tan[256];
	__shared__ float s_c2[256], s_theta[256];
	__shared__ float s_k;
	const Brdf self = self_view(S, shadow);
	const int A = S.azim;
	{
		int j = threadIdx.x;                          // ntheta = 256 == BLOCK
		float theta = F(D((float)j / 256.0f) * 0.5 * DJB_PI);
		float c = cos_f(theta);
		s_theta[j] = theta; s_tan[j] = glibc_tan(D(theta)); s_c2[j] = c * c;
	}
	__syncthreads();
	const float dtheta = F(0.5 * DJB_PI / D(256.0f));
	for (int i = threadIdx.x; i < A; i += BLOCK) {
		float phi = F(D((float)i / (float)A) * 2.0 * DJB_PI);
		float nint = 0.0f;
		for (int j = 0; j < 256; ++j)
			nint = acc_tan_over_cos2(nint, aniso_p22_theta_phi(self, s_theta[j], phi), s_tan[j], s_c2[j]);
		S.pdf1[i] = nint * dtheta;
	}
	__threadfence_block();
	__syncthreads();
	if (threadIdx.x == 0) {
		const int cnt = 512;
		float dphi = F(2.0 * DJB_PI / D((float)cnt)), nint = 0.0f;
		for (int i = 0; i < cnt; ++i) nint += aniso_pdf1(self, F(D((float)i / (float)cnt) * 2.0 * DJB_PI));
		nint *= dphi;
		s_k = F(1.0 / D(nint));
	}
	__syncthreads();
	for (int i = threadIdx.x; i < A; i += BLOCK) S.pdf1[i] *= s_k;
}

// ---- compute_cdf1 + compute_qf1 (dj_brdf.h:2879-2936); one block ------------------------------
__global__ __launch_bounds__(BLOCK) void ka_cdf1_qf1(AnisoScratch S, int shadow)
{
	const Brdf self = self_view(S, shadow);
	const int A = S.azim, cnt = A - 1, res = cnt * 8;
	if (threadIdx.x == 0) {
		float dphi = F(2.0 * DJB_PI / D((float)cnt)), nint = 0.0f;
		S.cdf1[0] = 0.0f;
		for (int i = 1; i < cnt; ++i) {
			nint += aniso_pdf1(self, F(D((float)i / (float)cnt) * 2.0 * DJB_PI));
			S.cdf1[i] = nint * dphi;
		}
		S.cdf1[cnt] = 1.0f;
	}
	__threadfence_block();
	__syncthreads();
	for (int j = threadIdx.x; j < res; j += BLOCK)
		S.probes[j] = aniso_cdf1(self, F(D((float)j / (float)res) * 2.0 * DJB_PI));
	__threadfence_block();
	__syncthreads();
	if (threadIdx.x == 0) {
		int nq = 0, j = 0;
		S.qf1[nq++] = 0.0f;
		for (int i = 1; i < cnt; ++i) {
			float c = (float)i / (float)cnt;
			for (; j < res; ++j)
				if (S.probes[j] >= c) { S.qf1[nq++] = (float)j / (float)res; break; }
		}
		S.qf1[nq++] = 1.0f;
		S.counts[0] = nq;
		for (int k = nq; k < A; ++k) S.qf1[k] = 1.0f;
	}
}

// ---- compute_pdf2 + normalize_pdf2 (dj_brdf.h:2945-2970, 3062-3094) ---------------------------
__global__ __launch_bounds__(BLOCK) void ka_pdf2_grid(AnisoScratch S, int shadow)
{
	const Brdf self = self_view(S, shadow);
	const int E = S.elev, w = E - 1, A = S.azim;
	int e = blockIdx.x * BLOCK + threadIdx.x;
	if (e >= E * A) return;
	int i = e / E, j = e - i * E;                     // i: azimuth row, j: elevation
	float phi = F(D((float)i / (float)A) * 2.0 * DJB_PI);
	float theta = F(D((float)j / (float)w) * 0.5 * DJB_PI);
	S.pdf2[e] = j < w ? aniso_p22_theta_phi(self, theta, phi) / aniso_pdf1(self, phi) : 0.0f;
}
__global__ __launch_bounds__(BLOCK) void ka_pdf2_norm(AnisoScratch S, int shadow)
{
	const Brdf self = self_view(S, shadow);
	int j = blockIdx.x * BLOCK + threadIdx.x;
	if (j >= S.azim) return;
	const float dtheta = F(0.5 * DJB_PI / D(256.0f));
	float phi = F(D((float)j / (float)S.azim) * 2.0 * DJB_PI), nint = 0.0f;
	for (int i = 0; i < 256; ++i) {
		float theta = F(D((float)i / 256.0f) * 0.5 * DJB_PI);
		float c = cos_f(theta);
		nint = acc_tan_over_cos2(nint, aniso_pdf2(self, theta, phi), glibc_tan(D(theta)), c * c);
	}
	nint *= dtheta;
	S.rowk[j] = F(1.0 / D(nint));
}
__global__ __launch_bounds__(BLOCK) void ka_pdf2_scale(AnisoScratch S)
{
	int e = blockIdx.x * BLOCK + threadIdx.x;
	if (e >= S.elev * S.azim) return;
	S.pdf2[e] *= S.rowk[e / S.elev];
}

// ---- compute_cdf2 / compute_qf2 (dj_brdf.h:2974-3034) ------------------------------------------
__global__ __launch_bounds__(BLOCK) void ka_cdf2(AnisoScratch S, int shadow)
{
	const Brdf self = self_view(S, shadow);
	const int E = S.elev, w = E - 1;
	int i = blockIdx.x * BLOCK + threadIdx.x;
	if (i >= S.azim) return;
	const float dtheta = F(0.5 * DJB_PI / D((float)w));
	float phi = F(D((float)i / (float)S.azim) * 2.0 * DJB_PI), nint = 0.0f;
	for (int j = 0; j < w; ++j) {
		float theta = F(D((float)j / (float)w) * 0.5 * DJB_PI);
		float c = cos_f(theta);
		nint = acc_tan_over_cos2(nint, aniso_pdf2(self, theta, phi), glibc_tan(D(theta)), c * c);
		S.cdf2[j + E * i] = nint * dtheta;
	}
	S.cdf2[w + E * i] = 1.0f;
}
__global__ __launch_bounds__(BLOCK) void ka_qf2_probes(AnisoScratch S, int shadow)
{
	const Brdf self = self_view(S, shadow);
	const int res = (S.elev - 1) * 8;
	int e = blockIdx.x * BLOCK + threadIdx.x;
	if (e >= S.azim * res) return;
	int k = e / res, j = e - k * res;
	float phi = F(D((float)k / (float)S.azim) * 2.0 * DJB_PI);
	S.probes[e] = aniso_cdf2(self, F(D((float)j / (float)res) * 0.5 * DJB_PI), phi);
}
// compute_qf2's forward scan (dj_brdf.h:3005-3034): for i = 1 .. w-1 the probe index j only moves forward and
// stops at the first probe >= i / w.  Because the thresholds increase, that is the FIRST probe of the
// whole row that reaches the threshold (a probe before the previous stop that reached i / w would have
// reached (i-1) / w too and stopped the previous search earlier), and once a threshold is never reached
// none of the later ones is.  So every (row, i) searches independently; one block per azimuth row.
__global__ __launch_bounds__(128) void ka_qf2_merge(AnisoScratch S)
{
	const int E = S.elev, w = E - 1, res = w * 8;
	const int k = blockIdx.x;
	const float *pr = S.probes + (size_t)k * res;
	float *row = S.qf2_rows + (size_t)E * k;
	__shared__ int s_missing;
	if (threadIdx.x == 0) { s_missing = w; row[0] = 0.0f; }
	__syncthreads();
	for (int i = 1 + threadIdx.x; i < w; i += 128) {
		const float c = (float)i / (float)w;
		int j = 0;
		while (j < res && !(pr[j] >= c)) ++j;
		if (j < res) row[i] = (float)j / (float)res;
		else atomicMin(&s_missing, i);
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		int nq = s_missing;                            // entries the scan pushed: row[0 .. nq-1]
		row[nq++] = 1.0f;                              // the closing push_back(1.0) of the row (dj_brdf.h:3028)
		S.qf2_len[k] = nq;                             // == E unless the conditional CDF could not be inverted for every quantile
		if (nq != E) atomicAdd(&S.counts[1], 1);
	}
}
// m_qf2 as the reference lays it out (dj_brdf.h:3005-3034): the rows are push_back'ed one after the other, so a
// row that came up short (a conditional CDF that stays below (w-1)/w at the last probe: grazing-heavy data)
// shifts every later row, and spline::eval2d then reads the vector with a row stride of `elev` all the same
// (dj_brdf.h:2814-2824) -- misaligned rows, and indices past the end of the vector (undefined in the
// reference; 1.0 here).  `aligned` != 0 keeps every row at its own offset instead, padded with 1.0
// (DJB_OPT_ANISO_QF2_ALIGNED: what the table was meant to be; not what the reference computes).
__global__ __launch_bounds__(1024) void ka_qf2_layout(AnisoScratch S, int aligned)
{
	const int E = S.elev, A = S.azim, G = E * A;
	__shared__ int s_off[1025];
	if (threadIdx.x == 0) {
		int o = 0;
		for (int k = 0; k < A; ++k) { s_off[k] = aligned ? E * k : o; o += S.qf2_len[k]; }
		s_off[A] = o;
		S.counts[2] = o;                               // entries the reference's vector holds
	}
	for (int e = threadIdx.x; e < G; e += 1024) S.qf2[e] = 1.0f;
	__syncthreads();
	for (int k = 0; k < A; ++k) {
		const int len = S.qf2_len[k];
		const float *row = S.qf2_rows + (size_t)E * k;
		for (int i = threadIdx.x; i < len; i += 1024) S.qf2[s_off[k] + i] = row[i];
	}
}

// ---- fit_beckmann_parameters / fit_ggx_parameters (dj_brdf.h:3186-3307) ------------------------
__global__ __launch_bounds__(BLOCK) void ka_fit_terms(AnisoScratch S, int shadow)
{
	const Brdf self = self_view(S, shadow);
	int e = blockIdx.x * BLOCK + threadIdx.x;
	if (e >= NP_FIT * NT_FIT) return;
	int j = e / NT_FIT, i = e - j * NT_FIT;
	float phi = F(D((float)j / (float)NP_FIT) * 2.0 * DJB_PI);
	float cp = cos_f(phi), sp = sin_f(phi);
	float theta = F(D((float)i / (float)NT_FIT) * sqrt(DJB_PI * 0.5));
	float ts = theta * theta;
	float p22 = aniso_p22_theta_phi(self, ts, phi);
	float tt = tan_f(ts), ct = cos_f(ts);
	float tt2 = tt * tt;
	float tmp2 = theta * p22 * tt / (ct * ct);
	float e1 = -tt * cp, e2 = -tt * sp;
	const size_t M = (size_t)NP_FIT * NT_FIT;
	S.terms[e] = tmp2 * e1;
	S.terms[M + e] = tmp2 * e2;
	S.terms[2 * M + e] = tmp2 * (tt2 * (cp * cp));
	S.terms[3 * M + e] = tmp2 * (tt2 * (sp * sp));
	S.terms[4 * M + e] = tmp2 * (tt2 * cp * sp);
	S.terms[5 * M + e] = tmp2 * fabsf(e1);
	S.terms[6 * M + e] = tmp2 * fabsf(e2);
}
__global__ __launch_bounds__(BLOCK) void ka_fit_sum(AnisoScratch S)
{
	__shared__ float s_n[7];
	__shared__ float s_buf[2 * 7 * OS_STRIDE];
	const size_t M = (size_t)NP_FIT * NT_FIT;
	float n = ordered_sums<7>(S.terms, M, s_buf);
	if (threadIdx.x < 7) {
		const float dtheta = F(sqrt(DJB_PI * 0.5) / D((float)NT_FIT)), dphi = F(2.0 * DJB_PI / D((float)NP_FIT));
		s_n[threadIdx.x] = F(D(n) * (2.0 * D(dtheta) * D(dphi)));
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		float mux = s_n[0], muy = s_n[1];
		float ax = F(sqrt(D(2.0f * (s_n[2] - mux * mux))));
		float ay = F(sqrt(D(2.0f * (s_n[3] - muy * muy))));
		S.fit[0] = ax; S.fit[1] = ay;
		S.fit[2] = F(2.0 * D(s_n[4] - mux * muy) / D(ax * ay));
		S.fit[3] = mux; S.fit[4] = muy;
		S.fit[5] = F(sqrt(D(s_n[5] * s_n[5] - mux * mux)));
		S.fit[6] = F(sqrt(D(s_n[6] * s_n[6] - muy * muy)));
		S.fit[7] = 0.0f; S.fit[8] = mux; S.fit[9] = muy;
	}
}

template <int SRC>
hipError_t run_kind(hipStream_t s, const Brdf &src, const Params &std_p, const AnisoScratch &S, int shadow)
{
	const int E = S.elev, A = S.azim, w = E - 1, N = w * A, cnt = E - 1;
	hipLaunchKernelGGL((ka_setup<SRC>), dim3(blocks_for(N)), dim3(BLOCK), 0, s, src, std_p, S);
	double *va = S.v0, *vb = S.v1;
	for (int it = 0; it < 4; ++it) {
		hipLaunchKernelGGL(ka_matvec, dim3((N + MV_ROWS - 1) / MV_ROWS), dim3(MV_BLOCK), 0, s, S, va, vb);
		double *t = va; va = vb; vb = t;
	}
	hipLaunchKernelGGL(ka_p22_grid, dim3(blocks_for(E * A)), dim3(BLOCK), 0, s, S, va);
	hipLaunchKernelGGL(ka_norm_terms, dim3(blocks_for(NP_NORM * NT_NORM)), dim3(BLOCK), 0, s, S, shadow);
	hipLaunchKernelGGL(ka_norm_apply, dim3(1), dim3(BLOCK), 0, s, S);
	long long tabn = NT_SIG * NP_SIG > A * NP_SIG ? NT_SIG * NP_SIG : A * NP_SIG;
	hipLaunchKernelGGL(ka_sigma_tables, dim3(blocks_for(tabn)), dim3(BLOCK), 0, s, S, std_p, shadow);
	hipLaunchKernelGGL(ka_sigma_rows, dim3(blocks_for(N)), dim3(BLOCK), 0, s, S);
	hipLaunchKernelGGL((ka_fres_pairs<SRC>), dim3(blocks_for((long long)cnt * (cnt + 1))), dim3(BLOCK), 0, s, src, std_p, S, shadow);
	hipLaunchKernelGGL(ka_fres_rows, dim3(blocks_for(cnt)), dim3(BLOCK), 0, s, S);
	hipLaunchKernelGGL(ka_pdf1, dim3(1), dim3(BLOCK), 0, s, S, shadow);
	hipLaunchKernelGGL(ka_cdf1_qf1, dim3(1), dim3(BLOCK), 0, s, S, shadow);
	hipLaunchKernelGGL(ka_pdf2_grid, dim3(blocks_for(E * A)), dim3(BLOCK), 0, s, S, shadow);
	hipLaunchKernelGGL(ka_pdf2_norm, dim3(blocks_for(A)), dim3(BLOCK), 0, s, S, shadow);
	hipLaunchKernelGGL(ka_pdf2_scale, dim3(blocks_for(E * A)), dim3(BLOCK), 0, s, S);
	hipLaunchKernelGGL(ka_cdf2, dim3(blocks_for(A)), dim3(BLOCK), 0, s, S, shadow);
	hipLaunchKernelGGL(ka_qf2_probes, dim3(blocks_for((long long)A * w * 8)), dim3(BLOCK), 0, s, S, shadow);
	hipLaunchKernelGGL(ka_qf2_merge, dim3(A), dim3(128), 0, s, S);
	hipLaunchKernelGGL(ka_qf2_layout, dim3(1), dim3(1024), 0, s, S, S.qf2_aligned);
	hipLaunchKernelGGL(ka_fit_terms, dim3(blocks_for(NP_FIT * NT_FIT)), dim3(BLOCK), 0, s, S, shadow);
	hipLaunchKernelGGL(ka_fit_sum, dim3(1), dim3(BLOCK), 0, s, S);
	return hipGetLastError();
}

} // namespace

namespace djbk {

size_t aniso_terms_count() { return (size_t)7 * NP_FIT * NT_FIT; }   // >= NP_NORM * NT_NORM
size_t aniso_ndf_count() { return (size_t)NT_SIG * NP_SIG; }
size_t aniso_cosd_count(int azim) { return (size_t)azim * NP_SIG; }
size_t aniso_sig_nodes() { return NT_SIG; }

hipError_t launch_fit_aniso(hipStream_t s, const Brdf &src, const Params &std_p, const AnisoScratch &S, int shadow)
{
	switch (src.kind) {
	case KIND_BECKMANN: return run_kind<KIND_BECKMANN>(s, src, std_p, S, shadow);
	case KIND_GGX:      return run_kind<KIND_GGX>(s, src, std_p, S, shadow);
	case KIND_TABULAR:  return run_kind<KIND_TABULAR>(s, src, std_p, S, shadow);
	case KIND_MERL:     return run_kind<KIND_MERL>(s, src, std_p, S, shadow);
	case KIND_UTIA:     return run_kind<KIND_UTIA>(s, src, std_p, S, shadow);
	case KIND_LAMBERT:  return run_kind<KIND_LAMBERT>(s, src, std_p, S, shadow);
	case KIND_SGD:      return run_kind<KIND_SGD>(s, src, std_p, S, shadow);
	case KIND_ABC:      return run_kind<KIND_ABC>(s, src, std_p, S, shadow);
	case KIND_TABULAR_ANISO: return run_kind<KIND_TABULAR_ANISO>(s, src, std_p, S, shadow);
	}
	return hipErrorInvalidValue;
}

} // namespace djbk
