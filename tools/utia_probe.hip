// utia_probe.hip -- microbenchmark behind DESIGN.md section 4.4 (UTIA eval): what the 16-tap interpolation of
// utia::eval costs on this memory system for different table layouts and gather shapes, without the arithmetic.
// Every variant streams 6 floats per pair in and 3 out (36 B/pair, like k_eval<UTIA>), derives the node indices
// (theta_i, phi_i, theta_v, phi_v) from the pair with the distribution uniform hemisphere directions give, fetches the
// 16 RGB taps and adds them up.
//   p  lane-private: 2 records of 128 B per pair (8 taps each, 10.6 MB table)      -- the round-1/2 kernel
//   c  wave-cooperative: same records, 8 lanes x 16 B per record, transposed through LDS
//   h  wave-cooperative: 4 records of 64 B per pair (4 taps each, 5.3 MB table)
//   q  wave-cooperative: 8 records of 32 B per pair (2 taps each, 2.65 MB table)
//   r  lane-private raw texels: 8 segments of 24 B per pair out of the 1 MB RGB table
// Build: hipcc --offload-arch=gfx950 -O3 tools/utia_probe.hip -o /tmp/utia_probe ; run: /tmp/utia_probe [n]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ inline unsigned int pcg(unsigned int v)
{
	unsigned int s = v * 747796405u + 2891336453u;
	unsigned int w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
	return (w >> 22u) ^ w;
}
__global__ void k_fill(float *p, long long n, unsigned int seed)
{
	long long k = (long long)blockIdx.x * 256 + threadIdx.x;
	if (k < n) p[k] = (pcg((unsigned int)k ^ seed) >> 8) * (1.0f / 16777216.0f);
}

struct Node { int ti, pi, tv, pv; };
__device__ inline Node node_of(float ix, float iz, float ox, float oz)
{
	Node nd;
	int ti = (int)(acosf(iz) * (57.29578f / 15.0f)), tv = (int)(acosf(oz) * (57.29578f / 15.0f));
	nd.ti = ti > 4 ? 4 : ti; nd.tv = tv > 4 ? 4 : tv;
	nd.pi = (int)(ix * 48.0f); nd.pv = (int)(ox * 48.0f);
	return nd;
}

// ---- p: lane-private 128-B records
__global__ __launch_bounds__(256) void k_private(const float *in, float *out, const v4f *tab, long long n)
{
	long long stride = (long long)gridDim.x * 256;
	for (long long k = (long long)blockIdx.x * 256 + threadIdx.x; k < n; k += stride) {
		float ix = in[k], iy = in[n + k], iz = in[2 * n + k], ox = in[3 * n + k], oy = in[4 * n + k], oz = in[5 * n + k];
		Node nd = node_of(ix, iz, ox, oz);
		v4f acc = { iy, oy, 0, 0 };
#pragma unroll
		for (int a = 0; a < 2; ++a) {
			const v4f *rec = tab + 8 * (size_t)(288 * (48 * (nd.ti + a) + nd.pi) + 48 * nd.tv + nd.pv);
#pragma unroll
			for (int j = 0; j < 6; ++j) acc += rec[j];
		}
		out[k] = acc.x; out[n + k] = acc.y; out[2 * n + k] = acc.z + acc.w;
	}
}

// ---- cooperative: REC_B bytes per record, RECS records per pair; LANES = REC_B / 16 lanes fetch one record
// record r of a pair whose (a = 0, c = 0, k = 0) node is e0; dk = node offset of phi_i + 1 (wrapped)
template <int RECS> __device__ inline int rec_index(int e0, int dk, int r)
{
	if (RECS == 2) return e0 + r * (288 * 48);                                       // a
	if (RECS == 4) return e0 + (r >> 1) * (288 * 48) + (r & 1) * 48;                 // a, c
	return e0 + (r >> 2) * (288 * 48) + ((r >> 1) & 1) * dk + (r & 1) * 48;          // a, k, c
}
template <int REC_B, int RECS, int WRAP = 0>   // WRAP: record indices folded into a table of WRAP records (what an L2-resident table would do)
__global__ __launch_bounds__(256) void k_coop(const float *in, float *out, const v4f *tab, long long n)
{
	constexpr int LANES = REC_B / 16;               // lanes per record
	constexpr int PER_INSTR = 64 / LANES;           // records per load instruction
	constexpr int ROUNDS = 64 * RECS / PER_INSTR;   // load instructions per wave
	constexpr int PITCH = REC_B * RECS + 16;        // bytes of LDS per lane (padded against bank conflicts)
	__shared__ __attribute__((aligned(16))) char lds_all[4 * 64 * PITCH];
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	char *lds = lds_all + wave * 64 * PITCH;
	long long stride = (long long)gridDim.x * 256;
	long long n_round = (n + 63) & ~63LL;
	for (long long k = (long long)blockIdx.x * 256 + threadIdx.x; k < n_round; k += stride) {
		bool live = k < n;
		long long kk = live ? k : n - 1;
		float ix = in[kk], iy = in[n + kk], iz = in[2 * n + kk], ox = in[3 * n + kk], oy = in[4 * n + kk], oz = in[5 * n + kk];
		Node nd = node_of(ix, iz, ox, oz);
		const int e0 = 288 * (48 * nd.ti + nd.pi) + 48 * nd.tv + nd.pv, dk = nd.pi == 47 ? -47 * 288 : 288;
		// record r of lane L sits at lds + L * PITCH + r * REC_B; load instruction s fetches records
		// (L, r) with L * RECS + r = s * PER_INSTR + lane / LANES
#pragma unroll
		for (int s = 0; s < ROUNDS; ++s) {
			int slot = s * PER_INSTR + lane / LANES;        // which (L, r)
			int L = slot / RECS, r = slot % RECS;           // RECS is a power of two
			int se0 = __shfl(e0, L), sdk = RECS == 8 ? __shfl(dk, L) : 0;
			size_t e = (size_t)rec_index<RECS>(se0, sdk, r);
			if (WRAP) e = e % WRAP;
			v4f v = tab[e * LANES + (lane % LANES)];
			*(v4f *)(lds + L * PITCH + r * REC_B + 16 * (lane % LANES)) = v;
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		v4f acc = { iy, oy, 0, 0 };
#pragma unroll
		for (int j = 0; j < REC_B * RECS / 16; ++j) acc += *(const v4f *)(lds + lane * PITCH + 16 * j);
		__builtin_amdgcn_wave_barrier();
		if (live) { out[k] = acc.x; out[n + k] = acc.y; out[2 * n + k] = acc.z + acc.w; }
	}
}

// ---- c2: cooperative 128-B records in two phases (theta_i, then theta_i + 1): 144 B of LDS per lane
__global__ __launch_bounds__(256) void k_coop2(const float *in, float *out, const v4f *tab, long long n)
{
	constexpr int PITCH = 128 + 16;
	__shared__ __attribute__((aligned(16))) char lds_all[4 * 64 * PITCH];
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	char *lds = lds_all + wave * 64 * PITCH;
	long long stride = (long long)gridDim.x * 256;
	long long n_round = (n + 63) & ~63LL;
	for (long long k = (long long)blockIdx.x * 256 + threadIdx.x; k < n_round; k += stride) {
		bool live = k < n;
		long long kk = live ? k : n - 1;
		float ix = in[kk], iy = in[n + kk], iz = in[2 * n + kk], ox = in[3 * n + kk], oy = in[4 * n + kk], oz = in[5 * n + kk];
		Node nd = node_of(ix, iz, ox, oz);
		const int e0 = 288 * (48 * nd.ti + nd.pi) + 48 * nd.tv + nd.pv;
		v4f acc = { iy, oy, 0, 0 };
#pragma unroll
		for (int a = 0; a < 2; ++a) {
#pragma unroll
			for (int s = 0; s < 8; ++s) {
				int L = s * 8 + (lane >> 3);
				size_t e = (size_t)(__shfl(e0, L) + a * (288 * 48));
				*(v4f *)(lds + L * PITCH + 16 * (lane & 7)) = tab[e * 8 + (lane & 7)];
			}
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			__builtin_amdgcn_wave_barrier();
#pragma unroll
			for (int j = 0; j < 6; ++j) acc += *(const v4f *)(lds + lane * PITCH + 16 * j);
			__builtin_amdgcn_wave_barrier();
		}
		if (live) { out[k] = acc.x; out[n + k] = acc.y; out[2 * n + k] = acc.z + acc.w; }
	}
}

// ---- g: cooperative 128-B records straight into LDS (global_load_lds_dwordx4): linear destination, the source piece
// and the read position swizzled with the same involution (piece ^ (record & 7)); two phases, 8 KB of LDS per wave
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_glds(const float *in, float *out, const v4f *tab, long long n)
{
	__shared__ __attribute__((aligned(1024))) char lds_all[WAVES * 64 * 128];
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	char *lds = lds_all + wave * 64 * 128;
	long long stride = (long long)gridDim.x * 64 * WAVES;
	long long n_round = (n + 63) & ~63LL;
	for (long long k = (long long)blockIdx.x * 64 * WAVES + threadIdx.x; k < n_round; k += stride) {
		bool live = k < n;
		long long kk = live ? k : n - 1;
		float ix = in[kk], iy = in[n + kk], iz = in[2 * n + kk], ox = in[3 * n + kk], oy = in[4 * n + kk], oz = in[5 * n + kk];
		Node nd = node_of(ix, iz, ox, oz);
		const int e0 = 288 * (48 * nd.ti + nd.pi) + 48 * nd.tv + nd.pv;
		v4f acc = { iy, oy, 0, 0 };
#pragma unroll
		for (int a = 0; a < 2; ++a) {
#pragma unroll
			for (int s = 0; s < 8; ++s) {
				int L = s * 8 + (lane >> 3);
				size_t e = (size_t)(__shfl(e0, L) + a * (288 * 48));
				const v4f *src = tab + e * 8 + ((lane & 7) ^ (L & 7));
				__builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)src,
				                                 (void __attribute__((address_space(3))) *)(lds + s * 1024), 16, 0, 0);
			}
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			__builtin_amdgcn_wave_barrier();
#pragma unroll
			for (int j = 0; j < 6; ++j) acc += *(const v4f *)(lds + lane * 128 + 16 * (j ^ (lane & 7)));
			__builtin_amdgcn_wave_barrier();
		}
		if (live) { out[k] = acc.x; out[n + k] = acc.y; out[2 * n + k] = acc.z + acc.w; }
	}
}

// ---- r: raw RGB texels, 8 segments of 24 B
__global__ __launch_bounds__(256) void k_raw(const float *in, float *out, const float *tab, long long n)
{
	long long stride = (long long)gridDim.x * 256;
	for (long long k = (long long)blockIdx.x * 256 + threadIdx.x; k < n; k += stride) {
		float ix = in[k], iy = in[n + k], iz = in[2 * n + k], ox = in[3 * n + k], oy = in[4 * n + k], oz = in[5 * n + k];
		Node nd = node_of(ix, iz, ox, oz);
		v2f acc = { iy, oy };
		int pv = nd.pv == 47 ? 46 : nd.pv;   // (the wrap would be two 12-B reads; same cost class)
#pragma unroll
		for (int r = 0; r < 8; ++r) {
			int pi = nd.pi + ((r >> 1) & 1); if (pi == 48) pi = 0;
			const float *p = tab + 3 * (size_t)(288 * (48 * (nd.ti + (r >> 2)) + pi) + 48 * (nd.tv + (r & 1)) + pv);
			const v2f *q = (const v2f *)p;   // 24 B = 3 x 8 B (8-byte aligned: texel index * 12 with even pv ... not always; use dwords)
			acc.x += p[0] + p[1] + p[2]; acc.y += p[3] + p[4] + p[5];
			(void)q;
		}
		out[k] = acc.x; out[n + k] = acc.y; out[2 * n + k] = acc.x - acc.y;
	}
}

template <class F> static float time_ms(F f, int reps)
{
	hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
	f(); f(); CHECK(hipDeviceSynchronize());
	CHECK(hipEventRecord(a));
	for (int r = 0; r < reps; ++r) f();
	CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
	float ms; CHECK(hipEventElapsedTime(&ms, a, b));
	return ms / reps;
}

int main(int argc, char **argv)
{
	long long n = argc > 1 ? atoll(argv[1]) : 100000000LL;
	float *in, *out; v4f *tab;
	const size_t tab_bytes = (size_t)288 * 288 * 128;
	CHECK(hipMalloc((void **)&in, sizeof(float) * 6 * n));
	CHECK(hipMalloc((void **)&out, sizeof(float) * 3 * n));
	CHECK(hipMalloc((void **)&tab, tab_bytes));
	hipLaunchKernelGGL(k_fill, dim3((unsigned)((6 * n + 255) / 256)), dim3(256), 0, 0, in, 6 * n, 17u);
	hipLaunchKernelGGL(k_fill, dim3((unsigned)((tab_bytes / 4 + 255) / 256)), dim3(256), 0, 0, (float *)tab, (long long)(tab_bytes / 4), 99u);
	CHECK(hipDeviceSynchronize());
	int grids[] = { 256 * 8, 256 * 16, 256 * 32 };
	for (int g : grids) {
		float p = time_ms([&] { hipLaunchKernelGGL(k_private, dim3(g), dim3(256), 0, 0, in, out, tab, n); }, 5);
		float c = time_ms([&] { hipLaunchKernelGGL((k_coop<128, 2>), dim3(g), dim3(256), 0, 0, in, out, tab, n); }, 5);
		float h = time_ms([&] { hipLaunchKernelGGL((k_coop<64, 4>), dim3(g), dim3(256), 0, 0, in, out, tab, n); }, 5);
		float q = time_ms([&] { hipLaunchKernelGGL((k_coop<32, 8>), dim3(g), dim3(256), 0, 0, in, out, tab, n); }, 5);
		float c2 = time_ms([&] { hipLaunchKernelGGL(k_coop2, dim3(g), dim3(256), 0, 0, in, out, tab, n); }, 5);
		float g4 = time_ms([&] { hipLaunchKernelGGL(k_glds<4>, dim3(g), dim3(256), 0, 0, in, out, tab, n); }, 5);
		float g8 = time_ms([&] { hipLaunchKernelGGL(k_glds<8>, dim3(g / 2), dim3(512), 0, 0, in, out, tab, n); }, 5);
		printf("n %lld grid %5d: coop 2x128B two-phase %.3f ms | direct-to-LDS two-phase, 256 threads %.3f | 512 threads %.3f\n", n, g, c2, g4, g8);
		float w2 = time_ms([&] { hipLaunchKernelGGL((k_coop<128, 2, 17280>), dim3(g), dim3(256), 0, 0, in, out, tab, n); }, 5);
		float w4 = time_ms([&] { hipLaunchKernelGGL((k_coop<128, 4, 17280>), dim3(g), dim3(256), 0, 0, in, out, tab, n); }, 5);
		float w2b = time_ms([&] { hipLaunchKernelGGL((k_coop<128, 2, 41472>), dim3(g), dim3(256), 0, 0, in, out, tab, n); }, 5);
		printf("n %lld grid %5d: coop 128-B records folded into 2.2 MB: 2 per pair %.3f ms, 4 per pair %.3f | into 5.3 MB: 2 per pair %.3f\n", n, g, w2, w4, w2b);
		float r = time_ms([&] { hipLaunchKernelGGL(k_raw, dim3(g), dim3(256), 0, 0, in, out, (const float *)tab, n); }, 5);
		printf("n %lld grid %5d: private 2x128B (10.6 MB) %.3f ms | coop 2x128B %.3f | coop 4x64B (5.3 MB) %.3f | coop 8x32B (2.65 MB) %.3f | raw 8x24B (1 MB) %.3f\n",
		       n, g, p, c, h, q, r);
	}
	return 0;
}
