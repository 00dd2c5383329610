#!/usr/bin/env python3
"""Memory-side requests per second of each profiled workload: TCC_EA0_RDREQ (= FETCH_SIZE / 64 B: a 128-byte stream read and a 64-byte
gather miss are one request each) + write requests (WRITE_SIZE / 64 B: the L2 writes back in requests of at most 64 bytes), divided by
the launch time of profiles/<round>/<workload>_bench_plain.json.   python tools/fabric_request_rate.py [round=r04]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r04"
print("%-26s %9s %14s %15s %12s" % ("workload", "ms", "read requests", "write requests", "G requests/s"))
for w in ("merl_eval", "merl_eval_uniform_bins", "merl_eval_coherent", "ggx_eval_pdf_contract", "ggx_eval_pdf", "beckmann_sample", "utia_eval"):
    d = json.load(open(os.path.join(ROOT, "profiles", f"pmc_{w}.json")))
    b = json.loads(open(os.path.join(ROOT, "profiles", rnd, f"{w}_bench_plain.json")).read().strip().splitlines()[-1])
    ms = b["ms_per_step"]
    rd = d["FETCH_SIZE_KB_per_launch"] * 1024 / 64
    wr = d["WRITE_SIZE_KB_per_launch"] * 1024 / 64
    print("%-26s %9.3f %12.3fe9 %13.3fe9 %12.1f" % (w, ms, rd / 1e9, wr / 1e9, (rd + wr) / ms / 1e6))
