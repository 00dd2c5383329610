#!/usr/bin/env python3
"""Exhaustive check of the float -> float sites of the fp64 trig family: GPU (ROCm ocml double functions, rounded to
float as the kernels do) against the library's host instantiation (glibc, what the reference links), over ALL 2^32
float inputs of every site (csrc/djb_device.hpp TRIG_*).  Needs a GPU:

    python tools/exhaustive_trig.py [--sites cos,acos] [--out gpurun_out/exhaustive_trig.json] [--chunk-log2 27]

Prints one line per site and writes the list of differing inputs (input bits, device bits, host bits).
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
    ap.add_argument("--sites", default=",".join(djb.TRIG_SITES))
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
        res["sites"][site] = {"differ": total, "seconds": round(dt, 1),
                              "inputs": [["0x%08x" % x, "0x%08x" % d, "0x%08x" % h] for x, d, h in rows]}
        print("%-10s %d inputs differ of %d  (%.0f s)" % (site, total, a.count, dt), flush=True)
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        with open(a.out, "w") as f:
            json.dump(res, f, indent=1)
    return 0


if __name__ == "__main__":
    sys.exit(main())
