// stream_probe.hip -- microbenchmark behind DESIGN.md's "achievable bandwidth" figure: the access
// pattern of the batch kernels (6 SoA float streams in, 3 out, 4 consecutive pairs per lane) with
// a configurable amount of dependent VALU work per pair, non-temporal or plain accesses, and
// different grid shapes.  Build: hipcc --offload-arch=gfx950 -O3 tools/stream_probe.hip -o /tmp/stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));

template <bool NT> __device__ inline v4f ld(const v4f *p)
{
	return NT ? __builtin_nontemporal_load(p) : *p;
}
template <bool NT> __device__ inline void st(v4f v, v4f *p)
{
	if (NT) __builtin_nontemporal_store(v, p); else *p = v;
}

template <bool NT, int WORK, int UNROLL>
__global__ __launch_bounds__(256) void k_probe(const v4f *a0, const v4f *a1, const v4f *a2, const v4f *b0,
                                               const v4f *b1, const v4f *b2, v4f *c0, v4f *c1, v4f *c2,
                                               long long n4)
{
	long long stride = (long long)gridDim.x * 256 * UNROLL;
	for (long long q0 = (long long)blockIdx.x * 256 * UNROLL + threadIdx.x; q0 < n4; q0 += stride) {
		v4f x[UNROLL], y[UNROLL], z[UNROLL], u[UNROLL], v[UNROLL], w[UNROLL];
#pragma unroll
		for (int j = 0; j < UNROLL; ++j) {
			long long q = q0 + j * 256;
			if (q < n4) {
				x[j] = ld<NT>(a0 + q); y[j] = ld<NT>(a1 + q); z[j] = ld<NT>(a2 + q);
				u[j] = ld<NT>(b0 + q); v[j] = ld<NT>(b1 + q); w[j] = ld<NT>(b2 + q);
			}
		}
#pragma unroll
		for (int j = 0; j < UNROLL; ++j) {
			long long q = q0 + j * 256;
			if (q < n4) {
				v4f r = x[j] + u[j], g = y[j] + v[j], b = z[j] + w[j];
#pragma unroll 8
				for (int k = 0; k < WORK; ++k) {   // 3 dependent vec4 FMAs = 12 VALU ops per 4 pairs
					r = r * g + b; g = g * b + r; b = b * r + g;
				}
				st<NT>(r, c0 + q); st<NT>(g, c1 + q); st<NT>(b, c2 + q);
			}
		}
	}
}

// ---- random 12-byte gathers from a table of `entries` texels (uniform over the table), 4 per lane
// per iteration, with or without the 36 B/pair streams alongside: what the L2 / Infinity Cache sustain.
struct Texel { float x, y, z; };
__device__ inline unsigned int pcg(unsigned int v)
{
	unsigned int s = v * 747796405u + 2891336453u;
	unsigned int w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
	return (w >> 22u) ^ w;
}
template <int POL> __device__ inline void gload3(float &a, float &b, float &c, const Texel *p)
{
	typedef float v3f __attribute__((ext_vector_type(3)));
	v3f v;
	if (POL == 0) asm volatile("global_load_dwordx3 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
	if (POL == 1) asm volatile("global_load_dwordx3 %0, %1, off sc0" : "=v"(v) : "v"(p) : "memory");
	if (POL == 2) asm volatile("global_load_dwordx3 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
	if (POL == 3) asm volatile("global_load_dwordx3 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
	if (POL == 4) asm volatile("global_load_dwordx3 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
	if (POL == 5) asm volatile("global_load_dwordx3 %0, %1, off sc0 sc1 nt" : "=v"(v) : "v"(p) : "memory");
	if (POL == 6) asm volatile("global_load_dwordx3 %0, %1, off sc0 nt" : "=v"(v) : "v"(p) : "memory");
	if (POL == 7) asm volatile("global_load_dwordx3 %0, %1, off sc1 nt" : "=v"(v) : "v"(p) : "memory");
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	a = v.x; b = v.y; c = v.z;
}
template <int POL>
__global__ __launch_bounds__(256) void k_gather_pol(const Texel *tab, unsigned int entries, v4f *c0, long long n4)
{
	long long stride = (long long)gridDim.x * 256;
	float acc = 0;
	for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < n4; q += stride) {
		unsigned int h = pcg((unsigned int)q * 4u);
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			h = pcg(h + j);
			float a, b, c;
			gload3<POL>(a, b, c, tab + (unsigned int)(((unsigned long long)h * entries) >> 32));
			acc += a + b + c;
		}
	}
	if (acc == 123.456f) c0[0] = v4f{ acc, 0, 0, 0 };
}
template <int POL>
static void run_pol(unsigned int entries, float **d, Texel *tab, long long n)
{
	hipEvent_t e0, e1;
	(void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	auto launch = [&]() { hipLaunchKernelGGL((k_gather_pol<POL>), dim3(16384), dim3(256), 0, 0, tab, entries, (v4f *)d[6], n / 4); };
	launch(); launch();
	(void)hipEventRecord(e0);
	for (int k = 0; k < 3; ++k) launch();
	(void)hipEventRecord(e1);
	(void)hipEventSynchronize(e1);
	float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 3;
	printf("gather(serial waits) policy=%d table=%6.2f MB : %7.3f ms  %7.1f G lookups/s\n", POL, entries * 12.0 / 1048576.0, ms, n / ms / 1e6);
}

template <bool STREAMS>
__global__ __launch_bounds__(256) void k_gather(const Texel *tab, unsigned int entries, const v4f *a0, const v4f *a1,
                                                const v4f *a2, const v4f *b0, const v4f *b1, const v4f *b2,
                                                v4f *c0, v4f *c1, v4f *c2, long long n4)
{
	long long stride = (long long)gridDim.x * 256;
	v4f accr = { 0, 0, 0, 0 };
	for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < n4; q += stride) {
		v4f x = { 0, 0, 0, 0 }, y = x, z = x, u = x, v = x, w = x;
		if (STREAMS) {
			x = ld<true>(a0 + q); y = ld<true>(a1 + q); z = ld<true>(a2 + q);
			u = ld<true>(b0 + q); v = ld<true>(b1 + q); w = ld<true>(b2 + q);
		}
		unsigned int h = pcg((unsigned int)q * 4u + (STREAMS ? (unsigned int)(x.x + u.x) : 0u));
		Texel t[4];
#pragma unroll
		for (int j = 0; j < 4; ++j) { h = pcg(h + j); t[j] = tab[(unsigned int)(((unsigned long long)h * entries) >> 32)]; }
		v4f r = { t[0].x, t[1].x, t[2].x, t[3].x }, g = { t[0].y, t[1].y, t[2].y, t[3].y }, b = { t[0].z, t[1].z, t[2].z, t[3].z };
		if (STREAMS) {
			r += y + v; g += z + w;
			st<true>(r, c0 + q); st<true>(g, c1 + q); st<true>(b, c2 + q);
		} else accr += r + g + b;
	}
	if (!STREAMS && accr.x == 123.456f) c0[0] = accr;
}

// ---- round 3: table layout vs the 64-byte sector a miss fetches (profiles/r03/gather_miss_calibration.txt).  LAYOUT 0: packed
// 12-byte texels (shipped: 17 % of them straddle a 64-byte boundary and can miss twice); 1: five texels per 64-byte sector
// (60 B + 4 B pad, +6.7 % footprint, none straddles); 2: padded to 16 bytes (+33 %).  Look-ups skewed like the bench
// distribution: SKEW % of them into the first eighth of the table.
template <int LAYOUT, int SKEW>
__global__ __launch_bounds__(256) void k_gather_layout(const float *tab, unsigned int entries, const v4f *a0, const v4f *a1,
                                                       const v4f *a2, const v4f *b0, const v4f *b1, const v4f *b2,
                                                       v4f *c0, v4f *c1, v4f *c2, long long n4)
{
	long long stride = (long long)gridDim.x * 256;
	for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < n4; q += stride) {
		v4f x = ld<true>(a0 + q), y = ld<true>(a1 + q), z = ld<true>(a2 + q);
		v4f u = ld<true>(b0 + q), v = ld<true>(b1 + q), w = ld<true>(b2 + q);
		unsigned int h = pcg((unsigned int)q * 4u + (unsigned int)(x.x + u.x));
		float tx[4], ty[4], tz[4];
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			h = pcg(h + j);
			const unsigned int h2 = pcg(h ^ 0x85ebca6bu);
			const bool hot = (h >> 7) % 100u < (unsigned int)SKEW;
			const unsigned int range = hot ? entries / 8u : entries;
			const unsigned int idx = (unsigned int)(((unsigned long long)h2 * range) >> 32);
			unsigned int dw;                                             // dword offset of the texel
			if (LAYOUT == 0) dw = 3u * idx;
			else if (LAYOUT == 1) { const unsigned int s5 = idx / 5u; dw = 16u * s5 + 3u * (idx - 5u * s5); }
			else dw = 4u * idx;
			const float *t = tab + dw;
			tx[j] = t[0]; ty[j] = t[1]; tz[j] = t[2];
		}
		v4f r = { tx[0], tx[1], tx[2], tx[3] }, g = { ty[0], ty[1], ty[2], ty[3] }, b = { tz[0], tz[1], tz[2], tz[3] };
		r += y + v; g += z + w;
		st<true>(r, c0 + q); st<true>(g, c1 + q); st<true>(b, c2 + q);
	}
}
template <int LAYOUT, int SKEW>
static void run_gather_layout(unsigned int entries, float **d, Texel *tab, long long n)
{
	hipEvent_t e0, e1;
	(void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	auto launch = [&]() {
		hipLaunchKernelGGL((k_gather_layout<LAYOUT, SKEW>), dim3(16384), dim3(256), 0, 0, (const float *)tab, entries, (const v4f *)d[0],
		                   (const v4f *)d[1], (const v4f *)d[2], (const v4f *)d[3], (const v4f *)d[4],
		                   (const v4f *)d[5], (v4f *)d[6], (v4f *)d[7], (v4f *)d[8], n / 4);
	};
	launch(); launch();
	(void)hipEventRecord(e0);
	for (int k = 0; k < 5; ++k) launch();
	(void)hipEventRecord(e1);
	(void)hipEventSynchronize(e1);
	float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
	static const char *names[3] = { "packed 12 B        ", "5 per 64-B sector  ", "padded 16 B        " };
	printf("streams+gather layout %s skew %2d %% into 1/8 of %u entries: %7.3f ms  %7.1f G pairs/s\n", names[LAYOUT], SKEW, entries, ms, n / ms / 1e6);
}

template <bool STREAMS>
static void run_gather(unsigned int entries, float **d, Texel *tab, long long n)
{
	long long n4 = n / 4;
	hipEvent_t e0, e1;
	(void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	auto launch = [&]() {
		hipLaunchKernelGGL((k_gather<STREAMS>), dim3(16384), dim3(256), 0, 0, tab, entries, (const v4f *)d[0],
		                   (const v4f *)d[1], (const v4f *)d[2], (const v4f *)d[3], (const v4f *)d[4],
		                   (const v4f *)d[5], (v4f *)d[6], (v4f *)d[7], (v4f *)d[8], n4);
	};
	launch(); launch();
	(void)hipEventRecord(e0);
	for (int k = 0; k < 5; ++k) launch();
	(void)hipEventRecord(e1);
	(void)hipEventSynchronize(e1);
	float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
	printf("gather streams=%d table=%6.2f MB : %7.3f ms  %7.1f G lookups/s\n", (int)STREAMS, entries * 12.0 / 1048576.0, ms,
	       n / ms / 1e6);
}

// ---- streams + gather structure variants (MODE): 0 = 4 gathers in flight after the loads (the shipped
// kernel's shape); 1 = gathers serialised by a data dependence; 2 = next iteration's streams
// prefetched before this iteration's gathers; 3 = one pair per lane (dword streams)
template <int MODE, int MINW>
__global__ __launch_bounds__(256, MINW) void k_sg(const Texel *tab, unsigned int entries, const v4f *a0, const v4f *a1,
                                            const v4f *a2, const v4f *b0, const v4f *b1, const v4f *b2,
                                            v4f *c0, v4f *c1, v4f *c2, long long n4)
{
	long long stride = (long long)gridDim.x * 256;
	if (MODE == 3) {
		const float *fa0 = (const float *)a0, *fa1 = (const float *)a1, *fa2 = (const float *)a2;
		const float *fb0 = (const float *)b0, *fb1 = (const float *)b1, *fb2 = (const float *)b2;
		float *fc0 = (float *)c0, *fc1 = (float *)c1, *fc2 = (float *)c2;
		for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < 4 * n4; q += stride) {
			float x = __builtin_nontemporal_load(fa0 + q), y = __builtin_nontemporal_load(fa1 + q), z = __builtin_nontemporal_load(fa2 + q);
			float u = __builtin_nontemporal_load(fb0 + q), v = __builtin_nontemporal_load(fb1 + q), w = __builtin_nontemporal_load(fb2 + q);
			unsigned int h = pcg((unsigned int)q + (unsigned int)(x + u));
			Texel t = tab[(unsigned int)(((unsigned long long)h * entries) >> 32)];
			__builtin_nontemporal_store(t.x + y + v, fc0 + q); __builtin_nontemporal_store(t.y + z + w, fc1 + q);
			__builtin_nontemporal_store(t.z, fc2 + q);
		}
		return;
	}
	long long q = (long long)blockIdx.x * 256 + threadIdx.x;
	v4f x, y, z, u, v, w;
	if (MODE == 2 && q < n4) {
		x = ld<true>(a0 + q); y = ld<true>(a1 + q); z = ld<true>(a2 + q);
		u = ld<true>(b0 + q); v = ld<true>(b1 + q); w = ld<true>(b2 + q);
	}
	for (; q < n4; q += stride) {
		if (MODE != 2) {
			x = ld<true>(a0 + q); y = ld<true>(a1 + q); z = ld<true>(a2 + q);
			u = ld<true>(b0 + q); v = ld<true>(b1 + q); w = ld<true>(b2 + q);
		}
		unsigned int h = pcg((unsigned int)q * 4u + (unsigned int)(x.x + u.x));
		v4f yy = y + v, zz = z + w;
		if (MODE == 2 && q + stride < n4) {
			long long qn = q + stride;
			x = ld<true>(a0 + qn); y = ld<true>(a1 + qn); z = ld<true>(a2 + qn);
			u = ld<true>(b0 + qn); v = ld<true>(b1 + qn); w = ld<true>(b2 + qn);
		}
		Texel t[4];
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			h = pcg(h + j);
			unsigned int idx = (unsigned int)(((unsigned long long)h * entries) >> 32);
			if (MODE == 1 && j > 0) idx += (unsigned int)(t[j - 1].x * 0.0f);
			t[j] = tab[idx];
		}
		v4f r = { t[0].x, t[1].x, t[2].x, t[3].x }, g = { t[0].y, t[1].y, t[2].y, t[3].y }, b = { t[0].z, t[1].z, t[2].z, t[3].z };
		r += yy; g += zz;
		st<true>(r, c0 + q); st<true>(g, c1 + q); st<true>(b, c2 + q);
	}
}
template <int MODE, int MINW>
static void run_sg(unsigned int entries, float **d, Texel *tab, long long n, int blocks)
{
	hipEvent_t e0, e1;
	(void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	auto launch = [&]() {
		hipLaunchKernelGGL((k_sg<MODE, MINW>), dim3(blocks), dim3(256), 0, 0, tab, entries, (const v4f *)d[0],
		                   (const v4f *)d[1], (const v4f *)d[2], (const v4f *)d[3], (const v4f *)d[4],
		                   (const v4f *)d[5], (v4f *)d[6], (v4f *)d[7], (v4f *)d[8], n / 4);
	};
	launch(); launch();
	(void)hipEventRecord(e0);
	for (int k = 0; k < 3; ++k) launch();
	(void)hipEventRecord(e1);
	(void)hipEventSynchronize(e1);
	float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 3;
	printf("streams+gather mode=%d minwaves=%d blocks=%6d table=%6.2f MB : %7.3f ms  %7.1f G pairs/s\n", MODE, MINW, blocks,
	       entries * 12.0 / 1048576.0, ms, n / ms / 1e6);
}

// ---- round 3, the last untested lever of profiles/archive/r03_NOTES.md 4.2: serve the hottest table lines from LDS.  HOTPCT % of the look-ups
// go to the first HOT_N texels of the table (128 KB: what one workgroup can hold next to nothing else), the rest
// uniformly to the remainder; LDS = true answers the hot ones from a per-workgroup LDS copy, false from the table (where
// they hit L2).  1024-thread workgroups, one per CU (the LDS copy allows no more), grid-stride.  The bench distribution
// puts 11 % of its look-ups into its hottest 160 KB of lines (CPU histogram, profiles/archive/r03_NOTES.md 4.2); 30 % is an optimistic case.
constexpr unsigned int HOT_N = 10922;      // 128 KB of 12-byte texels
template <bool LDS, int HOTPCT>
__global__ __launch_bounds__(1024) void k_sg_hot(const Texel *tab, unsigned int entries, const v4f *a0, const v4f *a1,
                                               const v4f *a2, const v4f *b0, const v4f *b1, const v4f *b2,
                                               v4f *c0, v4f *c1, v4f *c2, long long n4)
{
	extern __shared__ float hot_lds[];
	Texel *hot = (Texel *)hot_lds;
	if (LDS) {
		for (unsigned int t = threadIdx.x; t < HOT_N; t += 1024) hot[t] = tab[t];
		__syncthreads();
	}
	const long long stride = (long long)gridDim.x * 1024;
	for (long long q = (long long)blockIdx.x * 1024 + threadIdx.x; q < n4; q += stride) {
		v4f x = ld<true>(a0 + q), y = ld<true>(a1 + q), z = ld<true>(a2 + q);
		v4f u = ld<true>(b0 + q), v = ld<true>(b1 + q), w = ld<true>(b2 + q);
		unsigned int h = pcg((unsigned int)q * 4u + (unsigned int)(x.x + u.x));
		v4f yy = y + v, zz = z + w;
		Texel t[4];
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			h = pcg(h + j);
			const unsigned int h2 = pcg(h ^ 0x9e3779b9u);
			const bool is_hot = (h >> 7) % 100u < (unsigned int)HOTPCT;
			const unsigned int ih = (unsigned int)(((unsigned long long)h2 * HOT_N) >> 32);
			const unsigned int ic = HOT_N + (unsigned int)(((unsigned long long)h2 * (entries - HOT_N)) >> 32);
			if (LDS) { if (is_hot) t[j] = hot[ih]; else t[j] = tab[ic]; }
			else t[j] = tab[is_hot ? ih : ic];
		}
		v4f r = { t[0].x, t[1].x, t[2].x, t[3].x }, g = { t[0].y, t[1].y, t[2].y, t[3].y }, b = { t[0].z, t[1].z, t[2].z, t[3].z };
		r += yy; g += zz;
		st<true>(r, c0 + q); st<true>(g, c1 + q); st<true>(b, c2 + q);
	}
}
template <bool LDS, int HOTPCT>
static void run_sg_hot(unsigned int entries, float **d, Texel *tab, long long n, int blocks)
{
	hipEvent_t e0, e1;
	(void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	const size_t lds = LDS ? sizeof(Texel) * HOT_N : 0;
	if (LDS) (void)hipFuncSetAttribute((const void *)k_sg_hot<LDS, HOTPCT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
	auto launch = [&]() {
		hipLaunchKernelGGL((k_sg_hot<LDS, HOTPCT>), dim3(blocks), dim3(1024), lds, 0, tab, entries, (const v4f *)d[0],
		                   (const v4f *)d[1], (const v4f *)d[2], (const v4f *)d[3], (const v4f *)d[4],
		                   (const v4f *)d[5], (v4f *)d[6], (v4f *)d[7], (v4f *)d[8], n / 4);
	};
	launch(); launch();
	if (hipDeviceSynchronize() != hipSuccess) { printf("hot: launch failed: %s\n", hipGetErrorString(hipGetLastError())); return; }
	(void)hipEventRecord(e0);
	for (int k = 0; k < 3; ++k) launch();
	(void)hipEventRecord(e1);
	(void)hipEventSynchronize(e1);
	float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 3;
	printf("streams+gather, %2d %% of the look-ups in the hottest 128 KB, served from %s, %4d workgroups x 1024: %7.3f ms  %7.1f G pairs/s\n",
	       HOTPCT, LDS ? "LDS  " : "table", blocks, ms, n / ms / 1e6);
}

// ---- MODE 4 of the streams + gather experiment: the table look-ups go through the SCALAR memory
// path (v_readlane -> s_load_dwordx4 -> v_writelane), leaving the vector memory pipeline to the streams
template <int LANES_PER_BATCH>
__global__ __launch_bounds__(256) void k_sg_scalar(const float *tab, unsigned int entries, const v4f *a0, const v4f *a1,
                                                   const v4f *a2, const v4f *b0, const v4f *b1, const v4f *b2,
                                                   v4f *c0, v4f *c1, v4f *c2, long long n4)
{
	long long stride = (long long)gridDim.x * 256;
	for (long long q0 = (long long)blockIdx.x * 256; q0 < n4; q0 += stride) {
		long long q = q0 + threadIdx.x;
		bool live = q < n4;
		long long qq = live ? q : 0;
		v4f x = ld<true>(a0 + qq), y = ld<true>(a1 + qq), z = ld<true>(a2 + qq);
		v4f u = ld<true>(b0 + qq), v = ld<true>(b1 + qq), w = ld<true>(b2 + qq);
		unsigned int h = pcg((unsigned int)qq * 4u + (unsigned int)(x.x + u.x));
		unsigned int idx[4];
#pragma unroll
		for (int j = 0; j < 4; ++j) { h = pcg(h + j); idx[j] = (unsigned int)(((unsigned long long)h * entries) >> 32) * 3u; }
		float tr[4], tg[4], tb[4];
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			float r = 0, g = 0, b = 0;
			for (int l0 = 0; l0 < 64; l0 += LANES_PER_BATCH) {
				float sr[LANES_PER_BATCH], sg[LANES_PER_BATCH], sb[LANES_PER_BATCH];
#pragma unroll
				for (int l = 0; l < LANES_PER_BATCH; ++l) {
					unsigned int si = __builtin_amdgcn_readlane(idx[j], l0 + l);     // uniform -> s_load
					typedef const __attribute__((address_space(4))) float *cptr;   // constant address space -> s_load
					cptr p = (cptr)(tab + si);
					sr[l] = p[0]; sg[l] = p[1]; sb[l] = p[2];
				}
#pragma unroll
				for (int l = 0; l < LANES_PER_BATCH; ++l) {
					bool me = (int)(threadIdx.x & 63) == l0 + l;      // v_cmp + 3 v_cndmask (no writelane builtin)
					r = me ? sr[l] : r; g = me ? sg[l] : g; b = me ? sb[l] : b;
				}
			}
			tr[j] = r; tg[j] = g; tb[j] = b;
		}
		v4f r = { tr[0], tr[1], tr[2], tr[3] }, g = { tg[0], tg[1], tg[2], tg[3] }, b = { tb[0], tb[1], tb[2], tb[3] };
		r += y + v; g += z + w;
		if (live) { st<true>(r, c0 + q); st<true>(g, c1 + q); st<true>(b, c2 + q); }
	}
}
template <int LPB>
static void run_sg_scalar(unsigned int entries, float **d, Texel *tab, long long n, int blocks)
{
	hipEvent_t e0, e1;
	(void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	auto launch = [&]() {
		hipLaunchKernelGGL((k_sg_scalar<LPB>), dim3(blocks), dim3(256), 0, 0, (const float *)tab, entries, (const v4f *)d[0],
		                   (const v4f *)d[1], (const v4f *)d[2], (const v4f *)d[3], (const v4f *)d[4],
		                   (const v4f *)d[5], (v4f *)d[6], (v4f *)d[7], (v4f *)d[8], n / 4);
	};
	launch(); launch();
	(void)hipEventRecord(e0);
	for (int k = 0; k < 3; ++k) launch();
	(void)hipEventRecord(e1);
	(void)hipEventSynchronize(e1);
	float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 3;
	printf("streams+SCALAR-path gather lanes/batch=%2d blocks=%6d table=%6.2f MB : %7.3f ms  %7.1f G pairs/s\n", LPB, blocks,
	       entries * 12.0 / 1048576.0, ms, n / ms / 1e6);
}

// ---- cache-policy bits on the STREAM accesses while random gathers compete for L2 (table 9 MB):
// does any policy keep the streams from evicting table lines?  SPOL/WPOL: 0 none, 1 nt, 2 sc0 sc1 nt, 3 sc1 nt, 4 sc0 sc1
template <int POL> __device__ inline v4f ld_pol(const v4f *p)
{
	v4f v;
	if (POL == 0) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
	if (POL == 1) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
	if (POL == 2) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt" : "=v"(v) : "v"(p) : "memory");
	if (POL == 3) asm volatile("global_load_dwordx4 %0, %1, off sc1 nt" : "=v"(v) : "v"(p) : "memory");
	if (POL == 4) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
	return v;
}
template <int POL> __device__ inline void st_pol(v4f v, v4f *p)
{
	if (POL == 0) asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(p), "v"(v) : "memory");
	if (POL == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(p), "v"(v) : "memory");
	if (POL == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" :: "v"(p), "v"(v) : "memory");
	if (POL == 3) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" :: "v"(p), "v"(v) : "memory");
	if (POL == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
}
template <int SPOL, int WPOL>
__global__ __launch_bounds__(256) void k_sg_pol(const Texel *tab, unsigned int entries, const v4f *a0, const v4f *a1,
                                                const v4f *a2, const v4f *b0, const v4f *b1, const v4f *b2,
                                                v4f *c0, v4f *c1, v4f *c2, long long n4)
{
	long long stride = (long long)gridDim.x * 256;
	for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < n4; q += stride) {
		v4f x = ld_pol<SPOL>(a0 + q), y = ld_pol<SPOL>(a1 + q), z = ld_pol<SPOL>(a2 + q);
		v4f u = ld_pol<SPOL>(b0 + q), v = ld_pol<SPOL>(b1 + q), w = ld_pol<SPOL>(b2 + q);
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		unsigned int h = pcg((unsigned int)q * 4u + (unsigned int)(x.x + u.x));
		Texel t[4];
#pragma unroll
		for (int j = 0; j < 4; ++j) { h = pcg(h + j); t[j] = tab[(unsigned int)(((unsigned long long)h * entries) >> 32)]; }
		v4f r = { t[0].x, t[1].x, t[2].x, t[3].x }, g = { t[0].y, t[1].y, t[2].y, t[3].y }, b = { t[0].z, t[1].z, t[2].z, t[3].z };
		r += y + v; g += z + w;
		st_pol<WPOL>(r, c0 + q); st_pol<WPOL>(g, c1 + q); st_pol<WPOL>(b, c2 + q);
	}
}
template <int SPOL, int WPOL>
static void run_sg_pol(unsigned int entries, float **d, Texel *tab, long long n)
{
	hipEvent_t e0, e1;
	(void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	auto launch = [&]() {
		hipLaunchKernelGGL((k_sg_pol<SPOL, WPOL>), dim3(16384), dim3(256), 0, 0, tab, entries, (const v4f *)d[0],
		                   (const v4f *)d[1], (const v4f *)d[2], (const v4f *)d[3], (const v4f *)d[4],
		                   (const v4f *)d[5], (v4f *)d[6], (v4f *)d[7], (v4f *)d[8], n / 4);
	};
	launch(); launch();
	(void)hipEventRecord(e0);
	for (int k = 0; k < 3; ++k) launch();
	(void)hipEventRecord(e1);
	(void)hipEventSynchronize(e1);
	float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 3;
	printf("streams(load policy %d, store policy %d) + gather table=%6.2f MB : %7.3f ms\n", SPOL, WPOL, entries * 12.0 / 1048576.0, ms);
}

// ---- cost of the IEEE fp32 division (v_div_scale x2, v_rcp, 4 fma, v_div_fmas, v_div_fixup) and of
// sqrtf / fast reciprocal inside the stream pattern: NDIV operations of kind OP per pair
// OP 0: a / b (IEEE)   1: a * __builtin_amdgcn_rcpf(b)   2: sqrtf (IEEE)   3: fma chain of the same length (10 per "division")
template <int OP, int NDIV>
__global__ __launch_bounds__(256) void k_divs(const v4f *a0, const v4f *a1, const v4f *a2, const v4f *b0, const v4f *b1,
                                              const v4f *b2, v4f *c0, v4f *c1, v4f *c2, long long n4)
{
	long long stride = (long long)gridDim.x * 256;
	for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < n4; q += stride) {
		v4f x = ld<true>(a0 + q), y = ld<true>(a1 + q), z = ld<true>(a2 + q);
		v4f u = ld<true>(b0 + q), v = ld<true>(b1 + q), w = ld<true>(b2 + q);
		v4f r = x + u, g = y + v + 1.5f, b = z + w + 2.5f;
#pragma unroll
		for (int k = 0; k < NDIV; ++k) {
#pragma unroll
			for (int c = 0; c < 4; ++c) {
				if (OP == 0) r[c] = g[c] / (r[c] + b[c]);
				if (OP == 1) r[c] = g[c] * __builtin_amdgcn_rcpf(r[c] + b[c]);
				if (OP == 2) r[c] = __builtin_sqrtf(r[c] + b[c]);
				if (OP == 3) { float t = r[c] + b[c];
#pragma unroll
					for (int j = 0; j < 10; ++j) t = __builtin_fmaf(t, g[c], b[c]);
					r[c] = t; }
			}
		}
		st<true>(r, c0 + q); st<true>(g, c1 + q); st<true>(b, c2 + q);
	}
}
template <int OP, int NDIV>
static void run_divs(float **d, long long n)
{
	hipEvent_t e0, e1;
	(void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	auto launch = [&]() {
		hipLaunchKernelGGL((k_divs<OP, NDIV>), dim3(4096), dim3(256), 0, 0, (const v4f *)d[0], (const v4f *)d[1], (const v4f *)d[2],
		                   (const v4f *)d[3], (const v4f *)d[4], (const v4f *)d[5], (v4f *)d[6], (v4f *)d[7], (v4f *)d[8], n / 4);
	};
	launch(); launch();
	(void)hipEventRecord(e0);
	for (int k = 0; k < 3; ++k) launch();
	(void)hipEventRecord(e1);
	(void)hipEventSynchronize(e1);
	float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 3;
	const char *names[] = { "IEEE a/b", "a*rcp(b)", "IEEE sqrtf", "10-fma chain" };
	printf("%-13s x %2d per pair : %7.3f ms per 1e9 pairs, %8.1f G ops/s\n", names[OP], NDIV, ms * 1e9 / n,
	       (double)NDIV * n / ms / 1e6);
}

// same work, ONE pair per lane per iteration (dword accesses), as k_eval does it
template <int OP, int NDIV, int BLOCKS_PER_CU>
__global__ __launch_bounds__(256) void k_divs1(const float *a0, const float *a1, const float *a2, const float *b0, const float *b1,
                                               const float *b2, float *c0, float *c1, float *c2, long long n, long long sa)
{
	long long stride = (long long)gridDim.x * 256;
	for (long long k = (long long)blockIdx.x * 256 + threadIdx.x; k < n; k += stride) {
		long long q = k * sa;                       // runtime stride, like View::stride
		float x = a0[q], y = a1[q], z = a2[q], u = b0[q], v = b1[q], w = b2[q];
		float r = x + u, g = y + v + 1.5f, b = z + w + 2.5f;
#pragma unroll
		for (int j = 0; j < NDIV; ++j) {
			if (OP == 0) r = g / (r + b);
			if (OP == 3) { float t = r + b;
#pragma unroll
				for (int i = 0; i < 10; ++i) t = __builtin_fmaf(t, g, b);
				r = t; }
		}
		c0[q] = r; c1[q] = g; c2[q] = b;
	}
}
template <int OP, int NDIV>
static void run_divs1(float **d, long long n, int blocks)
{
	hipEvent_t e0, e1;
	(void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
	auto launch = [&]() {
		hipLaunchKernelGGL((k_divs1<OP, NDIV, 0>), dim3(blocks), dim3(256), 0, 0, d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7], d[8], n, 1LL);
	};
	launch(); launch();
	(void)hipEventRecord(e0);
	for (int k = 0; k < 3; ++k) launch();
	(void)hipEventRecord(e1);
	(void)hipEventSynchronize(e1);
	float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 3;
	printf("1 pair/lane, dword accesses, op %d x %2d, blocks %7d : %7.3f ms per 1e9 pairs\n", OP, NDIV, blocks, ms * 1e9 / n);
}

template <bool NT, int WORK, int UNROLL>
static void run(const char *name, int blocks, float **d, long long n)
{
	long long n4 = n / 4;
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	auto launch = [&]() {
		hipLaunchKernelGGL((k_probe<NT, WORK, UNROLL>), dim3(blocks), dim3(256), 0, 0, (const v4f *)d[0],
		                   (const v4f *)d[1], (const v4f *)d[2], (const v4f *)d[3], (const v4f *)d[4],
		                   (const v4f *)d[5], (v4f *)d[6], (v4f *)d[7], (v4f *)d[8], n4);
	};
	launch(); launch();
	hipEventRecord(e0);
	for (int k = 0; k < 5; ++k) launch();
	hipEventRecord(e1);
	hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
	printf("%-28s nt=%d work=%3d (VALU/pair=%4d) unroll=%d blocks=%7d : %7.3f ms  %7.1f GB/s\n", name, (int)NT, WORK,
	       WORK * 3, UNROLL, blocks, ms, 36.0 * n / ms / 1e6);
}

// ---- do the two halves of the MERL kernel hurt each other because they share each CU's memory pipeline?
// The stream kernel (6 in / 3 out, no gather) and the gather kernel (no streams) run CONCURRENTLY on two HIP streams,
// each over the full 1e9 units: if the time is ~max(stream, gather) a split design (index pass + gather pass on
// disjoint CUs) could beat the fused kernel's stream + gather sum; if it is ~the sum, the fused kernel already is
// what the memory system gives.
static void run_concurrent(unsigned int entries, float **d, Texel *tab, long long n)
{
	long long n4 = n / 4;
	hipStream_t s1, s2;
	hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
	hipEvent_t e0, e1, e2;
	hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2);
	auto copy = [&](hipStream_t st, int blocks) {
		hipLaunchKernelGGL((k_probe<true, 0, 1>), dim3(blocks), dim3(256), 0, st, (const v4f *)d[0], (const v4f *)d[1], (const v4f *)d[2],
		                   (const v4f *)d[3], (const v4f *)d[4], (const v4f *)d[5], (v4f *)d[6], (v4f *)d[7], (v4f *)d[8], n4);
	};
	auto gather = [&](hipStream_t st, int blocks) {
		hipLaunchKernelGGL((k_gather<false>), dim3(blocks), dim3(256), 0, st, tab, entries, (const v4f *)d[0], (const v4f *)d[1], (const v4f *)d[2],
		                   (const v4f *)d[3], (const v4f *)d[4], (const v4f *)d[5], (v4f *)d[6], (v4f *)d[7], (v4f *)d[8], n4);
	};
	for (int blocks : { 2048, 4096, 16384 }) {
		float ms_c, ms_g, ms_both;
		copy(s1, blocks); gather(s2, blocks); hipDeviceSynchronize();
		hipEventRecord(e0, s1); for (int k = 0; k < 3; ++k) copy(s1, blocks); hipEventRecord(e1, s1); hipEventSynchronize(e1);
		hipEventElapsedTime(&ms_c, e0, e1); ms_c /= 3;
		hipEventRecord(e0, s2); for (int k = 0; k < 3; ++k) gather(s2, blocks); hipEventRecord(e1, s2); hipEventSynchronize(e1);
		hipEventElapsedTime(&ms_g, e0, e1); ms_g /= 3;
		hipDeviceSynchronize();
		hipEventRecord(e0, s1);
		hipStreamWaitEvent(s2, e0, 0);
		for (int k = 0; k < 3; ++k) { copy(s1, blocks); gather(s2, blocks); }
		hipEventRecord(e1, s1); hipEventRecord(e2, s2);
		hipStreamWaitEvent(s1, e2, 0);
		hipEventRecord(e1, s1);
		hipEventSynchronize(e1);
		hipEventElapsedTime(&ms_both, e0, e1); ms_both /= 3;
		printf("table %6.2f MB, %5d workgroups each: streams alone %6.3f ms, gathers alone %6.3f ms, both concurrently %6.3f ms (sum %6.3f, max %6.3f)\n",
		       entries * 12.0 / 1048576.0, blocks, ms_c, ms_g, ms_both, ms_c + ms_g, ms_c > ms_g ? ms_c : ms_g);
	}
}

int main(int argc, char **argv)
{
	long long n = argc > 1 ? atoll(argv[1]) : 1000000000LL;
	float *d[9];
	for (int k = 0; k < 9; ++k) {
		if (hipMalloc((void **)&d[k], n * sizeof(float)) != hipSuccess) { printf("alloc failed\n"); return 1; }
		hipMemset(d[k], 0, n * sizeof(float));
	}
	int full = (int)((n / 4 + 255) / 256);
	if (argc > 2) {   // gather mode
		Texel *tab;
		if (hipMalloc((void **)&tab, 1458000 * 12 * 4) != hipSuccess) return 1;
		(void)hipMemset(tab, 0, 1458000 * 12 * 4);
		const unsigned int sizes[] = { 65536, 262144, 524288, 786432, 1458000, 2916000, 5832000 };
		if (argv[2][0] == 'd') {
			run_divs<0, 0>(d, n); run_divs<0, 8>(d, n); run_divs<0, 16>(d, n); run_divs<0, 32>(d, n);
			run_divs1<0, 0>(d, n, 4096); run_divs1<0, 16>(d, n, 4096); run_divs1<3, 16>(d, n, 4096); run_divs1<3, 32>(d, n, 4096);
			run_divs1<0, 16>(d, n, 16384); run_divs1<0, 16>(d, n, (int)((n + 255) / 256)); run_divs1<3, 32>(d, n, 16384);
			run_divs<1, 16>(d, n); run_divs<1, 32>(d, n); run_divs<2, 16>(d, n); run_divs<3, 16>(d, n); run_divs<3, 32>(d, n);
			return 0;
		}
		if (argv[2][0] == 'w') {
			for (unsigned int e : { 786432u, 1458000u }) {
				run_sg_pol<0, 0>(e, d, tab, n); run_sg_pol<1, 1>(e, d, tab, n); run_sg_pol<2, 2>(e, d, tab, n);
				run_sg_pol<3, 3>(e, d, tab, n); run_sg_pol<4, 4>(e, d, tab, n); run_sg_pol<1, 2>(e, d, tab, n);
				run_sg_pol<2, 1>(e, d, tab, n); run_sg_pol<1, 0>(e, d, tab, n); run_sg_pol<0, 1>(e, d, tab, n);
			}
			return 0;
		}
		if (argv[2][0] == 'c') {
			for (unsigned int e : { 262144u, 1458000u }) {
				run_sg<0, 1>(e, d, tab, n, 16384);
				run_sg_scalar<8>(e, d, tab, n, 16384); run_sg_scalar<16>(e, d, tab, n, 16384); run_sg_scalar<32>(e, d, tab, n, 16384);
				run_sg_scalar<16>(e, d, tab, n, 4096);
			}
			return 0;
		}
		if (argv[2][0] == 's') {
			for (unsigned int e : { 262144u, 1458000u }) {
				run_sg<0, 1>(e, d, tab, n, 16384); run_sg<1, 1>(e, d, tab, n, 16384); run_sg<2, 1>(e, d, tab, n, 16384);
				run_sg<3, 1>(e, d, tab, n, 16384); run_sg<0, 1>(e, d, tab, n, 2048); run_sg<0, 1>(e, d, tab, n, 1024);
				run_sg<0, 1>(e, d, tab, n, 512); run_sg<2, 1>(e, d, tab, n, 2048); run_sg<2, 1>(e, d, tab, n, 1024);
				run_sg<3, 1>(e, d, tab, n, 65536);
			}
			return 0;
		}
		if (argv[2][0] == 'l') {
			const unsigned int e = 1458000u;
			run_gather_layout<0, 0>(e, d, tab, n); run_gather_layout<1, 0>(e, d, tab, n); run_gather_layout<2, 0>(e, d, tab, n);
			run_gather_layout<0, 60>(e, d, tab, n); run_gather_layout<1, 60>(e, d, tab, n); run_gather_layout<2, 60>(e, d, tab, n);
			run_gather_layout<0, 85>(e, d, tab, n); run_gather_layout<1, 85>(e, d, tab, n); run_gather_layout<2, 85>(e, d, tab, n);
			return 0;
		}
		if (argv[2][0] == 'h') {
			const unsigned int e = 1458000u;
			run_sg<0, 1>(e, d, tab, n, 16384);                      // the shipped shape, uniform look-ups, for scale
			for (int blocks : { 256, 512 }) {
				run_sg_hot<false, 11>(e, d, tab, n, blocks); run_sg_hot<true, 11>(e, d, tab, n, blocks);
				run_sg_hot<false, 30>(e, d, tab, n, blocks); run_sg_hot<true, 30>(e, d, tab, n, blocks);
			}
			return 0;
		}
		if (argv[2][0] == 'x') {
			for (unsigned int e : { 262144u, 786432u, 1458000u }) run_concurrent(e, d, tab, n);
			return 0;
		}
		if (argv[2][0] == 'p') {
			for (unsigned int e : { 262144u, 1458000u }) {
				run_pol<0>(e, d, tab, n); run_pol<1>(e, d, tab, n); run_pol<2>(e, d, tab, n); run_pol<3>(e, d, tab, n);
				run_pol<4>(e, d, tab, n); run_pol<5>(e, d, tab, n); run_pol<6>(e, d, tab, n); run_pol<7>(e, d, tab, n);
			}
			return 0;
		}
		for (unsigned int e : sizes) run_gather<false>(e, d, tab, n);
		for (unsigned int e : sizes) run_gather<true>(e, d, tab, n);
		return 0;
	}
	run<true, 0, 1>("copy", 4096, d, n);
	run<false, 0, 1>("copy", 4096, d, n);
	run<true, 0, 1>("copy", full, d, n);
	run<false, 0, 1>("copy", full, d, n);
	run<true, 0, 1>("copy", 2048, d, n);
	run<true, 0, 1>("copy", 8192, d, n);
	run<true, 0, 1>("copy", 16384, d, n);
	run<true, 0, 2>("copy", 4096, d, n);
	run<false, 0, 2>("copy", 4096, d, n);
	run<true, 0, 4>("copy", 2048, d, n);
	run<true, 20, 1>("valu", 4096, d, n);
	run<true, 40, 1>("valu", 4096, d, n);
	run<true, 60, 1>("valu", 4096, d, n);
	run<true, 80, 1>("valu", 4096, d, n);
	run<true, 120, 1>("valu", 4096, d, n);
	run<false, 60, 1>("valu", 4096, d, n);
	run<true, 60, 1>("valu", full, d, n);
	run<true, 60, 2>("valu", 4096, d, n);
	return 0;
}
