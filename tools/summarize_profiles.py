#!/usr/bin/env python3
"""gpurun_out/prof/<workload>/... (tools/profile_bench.sh) -> profiles/<round>/*.csv and
profiles/pmc_<workload>.json (read by bench.py for roofline.traffic)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RND = sys.argv[1] if len(sys.argv) > 1 else "r01"
# the tree the passes were taken on: this script runs in the build container right after the gpurun call that profiled the working tree
# (the GPU box has no .git); bench.py prints it next to the figures it reads from these summaries
try:
    TREE = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or "unknown"
    if subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--", "dj_brdf_amd/csrc", "bench.py"], capture_output=True, text=True).stdout.strip():
        TREE += "+uncommitted"
except Exception:
    TREE = "unknown"
SRC = os.path.join(ROOT, "gpurun_out", "prof")
DST = os.path.join(ROOT, "profiles", RND)
os.makedirs(DST, exist_ok=True)
sys.path.insert(0, ROOT)
import bench          # LAUNCHES: the kernels each workload launches in its timed region (the same table bench.py checks a summary against)


def dominant(w, kernel):
    """is `kernel` (a rocprofv3 Kernel_Name) one of the kernels workload w times?"""
    return bench.kernel_short_name(kernel) in bench.LAUNCHES.get(w, ())


def counters(path):
    fs = glob.glob(os.path.join(path, "*", "*_counter_collection.csv"))
    fs = sorted(fs, key=os.path.getmtime)[-1:]       # gpurun merges runs into the same folder: newest only
    acc, meta = collections.defaultdict(list), {}
    for f in fs:
        for r in csv.DictReader(open(f)):
            acc[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
            meta[r["Kernel_Name"]] = (r["VGPR_Count"], r["SGPR_Count"], r["Workgroup_Size"], r["Grid_Size"], r["LDS_Block_Size"])
    return acc, meta


# gpurun merges every session's gpurun_out/prof/<workload> into the same tree: only the workloads named on the command line (the ones the
# last session profiled) are summarised -- anything else there is an older kernel's passes
ONLY = sys.argv[2:]
if not ONLY:
    sys.exit("usage: summarize_profiles.py <round> <workload> [<workload> ...]   (the workloads the last profile session ran)")
for w in sorted(os.listdir(SRC)):
    if w not in ONLY:
        continue
    d = os.path.join(SRC, w)
    for f in sorted(glob.glob(os.path.join(d, "trace", "*", "*_kernel_stats.csv")), key=os.path.getmtime)[-1:]:
        shutil.copy(f, os.path.join(DST, f"{w}_kernel_stats.csv"))
    for name in ("bench_plain.json", "bench_trace.json"):
        if os.path.exists(os.path.join(d, name)):
            shutil.copy(os.path.join(d, name), os.path.join(DST, f"{w}_{name}"))
    rows, per_kernel = [], collections.defaultdict(dict)
    for p in ("fetch", "write", "sq", "l2"):
        acc, meta = counters(os.path.join(d, p))
        for (k, c), v in sorted(acc.items()):
            rows.append([p, k, c, len(v), sum(v) / len(v)] + list(meta[k]))
            per_kernel[k][c] = sum(v) / len(v)
    with open(os.path.join(DST, f"{w}_pmc_summary.csv"), "w", newline="") as f:
        wr = csv.writer(f)
        wr.writerow(["pass", "kernel", "counter", "dispatches", "avg_per_dispatch", "vgpr", "sgpr", "workgroup", "grid", "lds"])
        wr.writerows(rows)
    fetch_kb = sum(v.get("FETCH_SIZE", 0) for k, v in per_kernel.items() if dominant(w, k))
    write_kb = sum(v.get("WRITE_SIZE", 0) for k, v in per_kernel.items() if dominant(w, k))
    if fetch_kb or write_kb:
        out = {
            "workload": w, "round": "round %d (profiles/%s)" % (int(RND.lstrip("r") or 0), RND), "tree": TREE,
            "kernels": [k for k in per_kernel if dominant(w, k)],
            "FETCH_SIZE_KB_per_launch": fetch_kb, "WRITE_SIZE_KB_per_launch": write_kb,
            # MI355X_MICROARCH.md section HBM: FETCH_SIZE counts 128-B streaming requests as 64 B on gfx950
            # (verified here on k_eval<GGX>: 1.2 GB reported for 2.4 GB read) -> x2; WRITE_SIZE is exact
            # (verified on k_gen_dir / k_eval: 12 B per element).  Units are KB (x1024).
            "hbm_bytes_per_launch": (2 * fetch_kb + write_kb) * 1024,
            "note": "FETCH_SIZE x2 (gfx950: 128-byte read requests are tallied at 64) + WRITE_SIZE",
        }
        # Table gathers (the MERL legs).  An L2 miss of a 12-byte gather is ONE memory-side request that fills the whole 128-byte line:
        # the compulsory misses of a table that fits the L2 equal its number of 128-byte lines per XCD (0.75 MB: 6 159 per XCD for
        # 6 144 lines; 3 MB: 24 588 for 24 576 -- profiles/r03/gather_miss_calibration.txt), and there are no 32-byte requests.  FETCH_SIZE
        # tallies every request at 64 bytes, so the x2 applies to them as it does to the streams.  (Round 3 read the same table as "64
        # bytes per miss" -- FETCH_SIZE / TCC_MISS is 64 by construction -- and reported the gather misses at half their size; round 4
        # corrects it: tools/exp/r04/merl_pair_lines.sh asked for both halves of each line explicitly and changed no counter.)
        # Their stream bytes being known (24 B per pair, read once), the gathers' share is split out.
        n_units = None
        try:
            n_units = json.loads(open(os.path.join(d, "bench_plain.json")).read().strip().splitlines()[-1])["config"]["units_per_gpu_per_step"]
            out["units_per_launch"] = n_units
            if bench.WORKLOADS[w][1]:
                out["algorithmic_bytes_per_launch"] = bench.WORKLOADS[w][1] * n_units
                out["traffic_over_algorithmic"] = out["hbm_bytes_per_launch"] / out["algorithmic_bytes_per_launch"]
        except Exception as e:  # pragma: no cover
            print("units per launch unknown:", e)
        if w.startswith("merl_eval") and n_units:
            try:
                requests = max(fetch_kb * 1024 / 64.0 - 24.0 * n_units / 128.0, 0.0)
                out["gather_miss_requests_per_launch"] = requests
                out["gather_miss_bytes_per_launch"] = 128.0 * requests
                out["units_per_launch"] = n_units
                out["note"] = ("memory-side bytes = 2 x FETCH_SIZE (every read request of this kernel moves 128 bytes -- the 24 B/pair streams and the "
                               "line fills of the table-gather misses alike -- and is tallied at 64) + WRITE_SIZE; the gather misses (gather_miss_bytes) "
                               "are served by the Infinity Cache (17.5 MB table), the streams by HBM")
            except Exception as e:  # pragma: no cover
                print("gather split skipped:", e)
        hit = sum(v.get("TCC_HIT_sum", 0) for k, v in per_kernel.items() if dominant(w, k))
        miss = sum(v.get("TCC_MISS_sum", 0) for k, v in per_kernel.items() if dominant(w, k))
        if hit + miss > 0:
            out["l2_hit_rate"] = hit / (hit + miss)          # TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum), MI355X_MICROARCH.md section L2
        json.dump(out, open(os.path.join(ROOT, "profiles", f"pmc_{w}.json"), "w"), indent=1)
        print(w, out["hbm_bytes_per_launch"] / 1e9, "GB per launch")
# the instruction-mix summaries of tools/valu_report.py (written on the GPU box into gpurun_out/): tracked copies, stamped with the tree
for f in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "valu_*.json"))):
    if os.path.basename(f)[len("valu_"):-len(".json")] not in ONLY:
        continue
    v = json.load(open(f))
    v["tree"] = TREE
    v["round"] = "round %d (profiles/%s)" % (int(RND.lstrip("r") or 0), RND)
    json.dump(v, open(os.path.join(ROOT, "profiles", os.path.basename(f)), "w"), indent=1)
    print(os.path.basename(f), "%.1f instr / %.1f slots per unit" % (v["insts_per_unit"], v["slots_per_unit"]), [k["kernel"] for k in v["kernels"]])
print(sorted(os.listdir(DST)))
