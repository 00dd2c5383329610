// djb_merl_file.hpp -- reading what a tabular(merl, res) fit needs out of a MERL file, shared by the GPU file
// pipeline (djb_loader.hip) and the host path (djb_cpu.cpp).  Plain C++ (no HIP).
//
// djb::tabular(merl, res) evaluates its source at a fixed set of directions (djb_device.hpp: fit_merl_slot_count):
// cnt back-scattering configurations + the (theta_d, theta_h) Fresnel pairs -- 5 545 of a file's 4 374 000 doubles
// at res 90.  The reference loads all 35 MB to read them (0.135 s per file).  Here the caller
// computes the table index of every query slot once (the same code the fit runs), and each file is mapped and only
// those entries are turned into per-slot texels: float(double(sample) * channel scale) with below-horizon bins
// zeroed, exactly what k_merl_convert / merl_convert_one write for them (dj_brdf.h:1010-1023).
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "../../include/djb_hip.h"

namespace djbfile {

constexpr long long MERL_N = 90LL * 90 * 180;
constexpr size_t PAYLOAD = sizeof(double) * 3 * MERL_N;   // 34 992 000 bytes after the 12-byte header

struct SlotPlan { std::vector<int32_t> slot, idx; };     // used slots sorted by table index (monotone walk over the mapping)

// idx[s]: table index query slot s reads, or < 0 for a slot the fit never evaluates
inline SlotPlan make_plan(const std::vector<int32_t> &idx)
{
	std::vector<int32_t> order;
	for (int s = 0; s < (int)idx.size(); ++s) if (idx[s] >= 0 && idx[s] < MERL_N) order.push_back(s);
	std::sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return idx[a] < idx[b] || (idx[a] == idx[b] && a < b); });
	SlotPlan plan;
	for (int32_t s : order) { plan.slot.push_back(s); plan.idx.push_back(idx[s]); }
	return plan;
}

// out: 3 floats (r, g, b) per slot, slot-major.  Same checks and messages as djb::merl::merl (dj_brdf.h:963-983); the
// header is untrusted (64-bit product of positive dims, MERL shape only); the reference reads the whole payload and
// fails if the file is short (dj_brdf.h:979-982): same verdict here, from the file size.
// `keep` (optional): the mapping is handed back instead of unmapped -- see the caller (djb_loader.hip: unmapping inside
// the gather loop is what stopped the gather from scaling with reader threads).
inline djb_status gather_file(const char *path, const SlotPlan &plan, float *out, std::string *err,
                              std::vector<std::pair<void *, size_t>> *keep = nullptr)
{
	char buf[256];
	int fd = open(path, O_RDONLY);
	if (fd < 0) { snprintf(buf, sizeof buf, "djb_error: Failed to open %s\n", path); *err = buf; return DJB_ERR_OPEN_FAILED; }
	int32_t dims[3] = { 0, 0, 0 };
	ssize_t got = pread(fd, dims, 12, 0);
	const bool positive = got == 12 && dims[0] > 0 && dims[1] > 0 && dims[2] > 0;
	long long n = positive ? (long long)dims[0] * (long long)dims[1] * (long long)dims[2] : 0;
	if (n <= 0) { close(fd); *err = "djb_error: Failed to read MERL header\n"; return DJB_ERR_BAD_HEADER; }
	if (n != MERL_N) {
		close(fd);
		snprintf(buf, sizeof buf, "djb_error: MERL table has %lld samples per channel, expected %lld\n", n, MERL_N);
		*err = buf; return DJB_ERR_BAD_HEADER;
	}
	struct stat sb;
	if (fstat(fd, &sb) != 0 || (size_t)sb.st_size < 12 + PAYLOAD) {
		close(fd);
		snprintf(buf, sizeof buf, "djb_error: Reading %s failed\n", path); *err = buf; return DJB_ERR_READ_FAILED;
	}
	void *map = mmap(nullptr, 12 + PAYLOAD, PROT_READ, MAP_PRIVATE, fd, 0);
	close(fd);
	if (map == MAP_FAILED) { snprintf(buf, sizeof buf, "djb_error: Reading %s failed\n", path); *err = buf; return DJB_ERR_READ_FAILED; }
	const char *base = (const char *)map + 12;                   // the payload is 4 bytes off 8-byte alignment: memcpy each double
	const size_t m = plan.slot.size();
	for (size_t k = 0; k < m; ++k) {
		const long long i = plan.idx[k];
		double s[3];
		memcpy(&s[0], base + 8 * (size_t)i, 8);
		memcpy(&s[1], base + 8 * (size_t)(i + MERL_N), 8);
		memcpy(&s[2], base + 8 * (size_t)(i + 2 * MERL_N), 8);
		// merl_convert_one on this entry (same expressions; host IEEE arithmetic == the device's)
		float r = (float)(s[0] * (1.00 / 1500.0)), g = (float)(s[1] * (1.15 / 1500.0)), b = (float)(s[2] * (1.66 / 1500.0));
		if ((double)r < 0.0 || (double)g < 0.0 || (double)b < 0.0) r = g = b = 0.0f;
		float *o = out + 3 * (size_t)plan.slot[k];
		o[0] = r; o[1] = g; o[2] = b;
	}
	if (keep) keep->emplace_back(map, 12 + PAYLOAD);
	else munmap(map, 12 + PAYLOAD);
	return DJB_OK;
}

} // namespace djbfile
