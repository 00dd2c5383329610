// tools/l2_sim.py's cache model: a set-associative LRU cache (16 ways, 128-byte lines) under a stream of 12-byte texel look-ups.
// A miss fills the whole line, as gfx950's L2 does for a gather (profiles/r03/gather_miss_calibration.txt: the compulsory misses of a
// small table equal its number of 128-byte lines); a texel that straddles two lines is two accesses.
//   l2_sim <u32 texel indices> <capacity bytes> <8100 bytes: hot flag per (theta_h, theta_d) row> <mode>
// mode 0: every access allocates; 1: look-ups of rows that are not hot bypass the cache on a miss; 2: they are inserted at the LRU position
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
int main(int argc, char **argv)
{
	if (argc < 5) return 2;
	FILE *f = fopen(argv[1], "rb"); const long cap_bytes = atol(argv[2]); const int ways = 16, mode = atoi(argv[4]);
	fseek(f, 0, SEEK_END); const long n = ftell(f) / 4; fseek(f, 0, SEEK_SET);
	uint32_t *idx = malloc(n * 4); if (fread(idx, 4, n, f) != (size_t)n) return 1;
	uint8_t hot[8100]; FILE *h = fopen(argv[3], "rb"); if (fread(hot, 1, 8100, h) != 8100) return 1;
	const long lines = cap_bytes / 128, sets = lines / ways;
	uint64_t *tag = malloc(lines * 8), *age = calloc(lines, 8);
	for (long i = 0; i < lines; ++i) tag[i] = ~0ull;
	long miss = 0, t = 1000;
	for (int pass = 0; pass < 2; ++pass) {                    // pass 0 warms the cache
		miss = 0;
		for (long k = 0; k < n; ++k) {
			const uint64_t a0 = (uint64_t)idx[k] * 12, a1 = a0 + 11; const int cold = !hot[idx[k] / 180];
			for (uint64_t line = a0 / 128; line <= a1 / 128; ++line) {
				const long set = (long)((line * 0x9E3779B97F4A7C15ull >> 20) % (uint64_t)sets), base = set * ways; long w = -1, lru = base; ++t;
				for (long j = base; j < base + ways; ++j) { if (tag[j] == line) { w = j; break; } if (age[j] < age[lru]) lru = j; }
				if (w >= 0) { if (!(cold && mode)) age[w] = t; continue; }
				++miss;
				if (cold && mode == 1) continue;
				tag[lru] = line; if (!(cold && mode == 2)) age[lru] = t;
			}
		}
	}
	printf("misses per look-up %.3f\n", (double)miss / n);
	return 0;
}
