#!/usr/bin/env python3
"""Exhaustive check of the sites of the fp64 trig family: GPU (ROCm ocml double functions, rounded to
float as the kernels do) against the library's host instantiation (glibc, what the reference links), over ALL 2^32
float inputs of every site (csrc/djb_device.hpp TRIG_*).  Needs a GPU:

    python tools/exhaustive_trig.py [--sites cos,acos] [--out gpurun_out/exhaustive_trig.json] [--chunk-log2 27]

Prints one line per site and writes the list of differing inputs (input bits, device bits, host bits).  The *_d
sites compare the double itself; there the count is informational: it says how often ROCm's and glibc's double
functions differ in the last place on float arguments (the places that keep the double run glibc's own algorithms).
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dj_brdf_amd import djb  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sites", default=",".join(djb.TRIG_SITES + djb.TRIG_DOUBLE_SITES))
    ap.add_argument("--out", default="gpurun_out/exhaustive_trig.json")
    ap.add_argument("--chunk-log2", type=int, default=27)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--first", type=lambda v: int(v, 0), default=0)
    ap.add_argument("--count", type=lambda v: int(v, 0), default=1 << 32)
    a = ap.parse_args()
    chunk = 1 << a.chunk_log2
    res = {"inputs_per_site": a.count, "first_bits": a.first, "sites": {}}
    for site in a.sites.split(","):
        t0 = time.time()
        total, rows = 0, []
        pos, end = a.first, a.first + a.count
        while pos < end:
            n = min(chunk, end - pos)
            nb, bad = djb.selftest_trig_sweep(site, pos, n, threads=a.threads, cap=4096)
            total += nb
            rows += bad
            pos += n
        dt = time.time() - t0
        if site in djb.TRIG_DOUBLE_SITES:   # rows: (input bits, difference in ulps of the double, 0)
            res["sites"][site] = {"differ": total, "seconds": round(dt, 1), "listed": len(rows),
                                  "max_ulp_listed": max([d for _, d, _ in rows], default=0),
                                  "first_inputs": ["0x%08x" % x for x, _, _ in rows[:32]]}
        else:
            res["sites"][site] = {"differ": total, "seconds": round(dt, 1),
                                  "inputs": [["0x%08x" % x, "0x%08x" % d, "0x%08x" % h] for x, d, h in rows]}
        print("%-10s %d inputs differ of %d  (%.0f s)" % (site, total, a.count, dt), flush=True)
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        with open(a.out, "w") as f:
            json.dump(res, f, indent=1)
    return 0


if __name__ == "__main__":
    sys.exit(main())
