// pcie_probe.hip -- what the host <-> HBM boundary can sustain on this box: pageable hipMemcpy,
// hipHostRegister cost + registered copies, pinned staging copies, and host memcpy bandwidth
// with 1..16 threads.  Build: hipcc --offload-arch=gfx950 -O2 tools/pcie_probe.hip -o tools/bin/pcie_probe -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
	const size_t B = 1200ull << 20;   // 1.2 GB
	char *h = (char *)aligned_alloc(4096, B), *h2 = (char *)aligned_alloc(4096, B);
	memset(h, 1, B); memset(h2, 2, B);
	char *d; hipMalloc((void **)&d, B);
	hipStream_t s; hipStreamCreate(&s);
	double t;
	for (int r = 0; r < 2; ++r) {
		t = now(); hipMemcpy(d, h, B, hipMemcpyHostToDevice); printf("pageable H2D   %6.2f GB/s\n", B / (now() - t) / 1e9);
		t = now(); hipMemcpy(h2, d, B, hipMemcpyDeviceToHost); printf("pageable D2H   %6.2f GB/s\n", B / (now() - t) / 1e9);
	}
	t = now(); hipError_t e = hipHostRegister(h, B, hipHostRegisterDefault); double treg = now() - t;
	printf("hipHostRegister 1.2 GB: %s, %.1f ms (%.2f GB/s)\n", hipGetErrorString(e), treg * 1e3, B / treg / 1e9);
	t = now(); hipHostRegister(h2, B, hipHostRegisterDefault); printf("hipHostRegister #2: %.1f ms\n", (now() - t) * 1e3);
	for (int r = 0; r < 2; ++r) {
		t = now(); hipMemcpyAsync(d, h, B, hipMemcpyHostToDevice, s); hipStreamSynchronize(s); printf("registered H2D %6.2f GB/s\n", B / (now() - t) / 1e9);
		t = now(); hipMemcpyAsync(h2, d, B, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); printf("registered D2H %6.2f GB/s\n", B / (now() - t) / 1e9);
	}
	{   // both directions at once on two streams
		hipStream_t s2; hipStreamCreate(&s2);
		char *d2; hipMalloc((void **)&d2, B);
		t = now(); hipMemcpyAsync(d, h, B, hipMemcpyHostToDevice, s); hipMemcpyAsync(h2, d2, B, hipMemcpyDeviceToHost, s2);
		hipStreamSynchronize(s); hipStreamSynchronize(s2); printf("registered H2D+D2H concurrent %6.2f GB/s total\n", 2.0 * B / (now() - t) / 1e9);
	}
	t = now(); hipHostUnregister(h); printf("hipHostUnregister: %.1f ms\n", (now() - t) * 1e3);
	hipHostUnregister(h2);
	char *p; hipHostMalloc((void **)&p, B, hipHostMallocDefault);
	for (int nt : { 1, 2, 4, 8, 16 }) {
		std::vector<std::thread> th;
		t = now();
		for (int k = 0; k < nt; ++k) th.emplace_back([=]() { size_t lo = B / nt * k; memcpy(p + lo, h + lo, B / nt); });
		for (auto &x : th) x.join();
		printf("host memcpy pageable->pinned, %2d threads: %6.2f GB/s\n", nt, B / (now() - t) / 1e9);
	}
	printf("hardware_concurrency %u\n", std::thread::hardware_concurrency());
	return 0;
}
