#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
// as sim.c, plus: rows flagged cold (hot[row]==0) either bypass (mode 1: no allocation on miss) or are inserted at LRU position (mode 2)
int main(int argc, char **argv)
{
	FILE *f = fopen(argv[1], "rb"); long cap_bytes = atol(argv[2]); int ways = 16; int mode = atoi(argv[4]);
	fseek(f, 0, SEEK_END); long n = ftell(f) / 4; fseek(f, 0, SEEK_SET);
	uint32_t *idx = malloc(n * 4); if (fread(idx, 4, n, f) != (size_t)n) return 1;
	uint8_t hot[8100]; FILE *h = fopen(argv[3], "rb"); if (fread(hot, 1, 8100, h) != 8100) return 1;
	long lines = cap_bytes / 128, sets = lines / ways;
	uint64_t *tag = calloc(lines, 8); uint8_t *val = calloc(lines, 1); uint64_t *age = calloc(lines, 8);
	for (long i = 0; i < lines; ++i) tag[i] = ~0ull;
	long ml = 0, ms = 0, hit = 0, t = 1000, acc = 0;
	for (int pass = 0; pass < 2; ++pass) { if (pass == 1) { ml = ms = hit = acc = 0; }
	for (long k = 0; k < n; ++k) {
		uint64_t a0 = (uint64_t)idx[k] * 12, a1 = a0 + 11; int cold = !hot[idx[k] / 180];
		for (uint64_t sec = a0 / 64; sec <= a1 / 64; ++sec) {
			uint64_t line = sec / 2; int s = sec & 1; long set = (line * 0x9E3779B97F4A7C15ull >> 20) % sets; ++t; ++acc;
			long base = set * ways, w = -1, lru = base;
			for (long j = base; j < base + ways; ++j) { if (tag[j] == line) { w = j; break; } if (age[j] < age[lru]) lru = j; }
			if (w >= 0) { if (val[w] >> s & 1) ++hit; else { ++ms; val[w] |= 1 << s; } if (!(cold && mode)) age[w] = t; }
			else { ++ml; if (cold && mode == 1) continue; tag[lru] = line; val[lru] = 1 << s; age[lru] = (cold && mode == 2) ? age[lru] : t; }
		}
	} }
	printf("%s mode %d cap %ld KB: hit %.3f line-miss %.3f sector-miss %.3f\n", argv[3], mode, cap_bytes >> 10, (double)hit / acc, (double)ml / acc, (double)ms / acc);
	return 0;
}
