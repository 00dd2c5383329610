/* Exhaustive check that the two-instruction form of near_f32_midpoint (dj_brdf_amd/csrc/djb_device.hpp) equals its
 * definition |(lo & 0x1FFFFFFF) - 2^28| <= width for every low word, for the widths the kernels use.
 * gcc -O3 -o /tmp/nmc tools/near_midpoint_check.c && /tmp/nmc [width ...]   (default: 256 1024 1 0 4096) */
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
int main(int argc, char **argv)
{
	const int defaults[] = { 256, 1024, 1, 0, 4096 };
	const int nw = argc > 1 ? argc - 1 : (int)(sizeof defaults / sizeof defaults[0]);
	for (int w = 0; w < nw; ++w) {
		const int width = argc > 1 ? atoi(argv[w + 1]) : defaults[w];
		unsigned long long bad = 0, hits = 0;
		uint32_t lo = 0;
		do {
			int d0 = (int)(lo & 0x1FFFFFFFu) - 0x10000000;
			int want = (d0 < 0 ? -d0 : d0) <= width;
			uint32_t d = (lo << 3) + (0u - ((0x10000000u - (uint32_t)width) << 3));
			int got = d <= ((uint32_t)width << 4);
			bad += want != got; hits += want;
		} while (++lo != 0);
		printf("width %d: %llu mismatches over 2^32 low words (%llu inside the band)\n", width, bad, hits);
		if (bad) return 1;
	}
	return 0;
}
