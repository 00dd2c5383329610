#!/usr/bin/env python3
"""VALU issue load of a workload's kernels next to its HBM fraction: reads the SQ_INSTS_* passes that
`tools/instmix.sh <workload>` left under gpurun_out/mix/<workload>/ and the per-class issue costs measured by
tools/valu_cost_probe.hip (profiles/r03/valu_issue_cost.txt), writes profiles/valu_<workload>.json, which bench.py
attaches to its line as roofline.valu (static, labelled like roofline.traffic).

    instructions per unit = SQ_INSTS_VALU (wave instructions per launch) / (units per launch / 64)
    slots per unit        = sum over classes of (instructions of the class) x (issue slots of the class), one slot = the
                            issue time of v_mul_f32 = 1.155 ns per wave-instruction per SIMD (2 cycles at the ~1.73 GHz the
                            chip sustains with every SIMD issuing VALU)
    issue time per launch = slots per unit x units / 64 x 1.155 ns / 1024 SIMDs
    frac_of_issue         = issue time / the kernel time of the rocprofv3 kernel trace of the same passes

Usage (on the GPU box, after tools/instmix.sh W [tag]):  python tools/valu_report.py W units_per_launch [round-label [tag]]
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import LAUNCHES, kernel_short_name as short_name      # the kernels each workload times (the table bench.py checks a summary against)
SLOT_NS = 1.155          # v_mul_f32 wave-instruction per SIMD, profiles/r03/valu_issue_cost.txt
N_SIMD = 1024            # 256 CUs x 4
# issue slots per instruction class (profiles/r03/valu_issue_cost.txt); OTHER = what the SQ class counters leave of
# SQ_INSTS_VALU: compares, selects, min / max, moves, bit-field and shift instructions: 1.5 for all but v_mov (0.85)
COST = {"ADD_F32": 0.94, "MUL_F32": 1.00, "FMA_F32": 1.36, "TRANS_F32": 2.96,
        "ADD_F64": 1.57, "MUL_F64": 1.67, "FMA_F64": 1.68, "TRANS_F64": 5.89,
        "CVT": 1.50, "INT32": 1.20, "INT64": 1.60, "OTHER": 1.45}


def kernel_id(full):
    """'void (anonymous namespace)::k_merl_fast_v4<1>(djbdev::Brdf, ...)' -> 'k_merl_fast_v4<1>' (round 4 cut at the first '(',
    which is the one of '(anonymous namespace)': every name came out as 'void ')"""
    import re
    m = re.search(r"(k_\w+(?:<[^()]*>)?)\s*\(", full.replace("(anonymous namespace)::", ""))
    return m.group(1) if m else full


def main():
    w = sys.argv[1]
    units = float(sys.argv[2])
    label = sys.argv[3] if len(sys.argv) > 3 else "round 4 (profiles/r04)"
    tag = sys.argv[4] if len(sys.argv) > 4 else ""          # tools/instmix.sh's directory tag (library variants)
    base = os.path.join(ROOT, "gpurun_out", "mix", w + tag)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    dur_pmc = collections.defaultdict(list)
    for f in glob.glob(base + "/*/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r.get("Start_Timestamp") and r.get("End_Timestamp"):      # serialised dispatch under the counters
                dur_pmc[r["Kernel_Name"]].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-6)
    for f in glob.glob(base + "/*/*/*kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            dur[r["Kernel_Name"]].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-6)
    kernels = []
    for k, v in agg.items():
        if any(s in k for s in ("gen_dir", "gen_uni", "rocclr", "convert", "at::native", "k_coherent", "k_uniform")):
            continue
        if LAUNCHES.get(w) and short_name(k) not in LAUNCHES[w]:       # set-up kernels of the leg (the fit that builds a tabular object ...)
            continue
        c = {n: sum(x) / len(x) for n, x in v.items()}
        if c.get("SQ_INSTS_VALU", 0) < 1e3:
            continue
        cls = {n[len("SQ_INSTS_VALU_"):]: x for n, x in c.items() if n.startswith("SQ_INSTS_VALU_") and n != "SQ_INSTS_VALU_IOPS"}
        named = sum(cls.values())
        cls["OTHER"] = max(c["SQ_INSTS_VALU"] - named, 0.0)
        slots = sum(cls[n] * COST.get(n, 1.5) for n in cls)
        d = dur.get(k) or dur_pmc.get(k)
        ms = sorted(d)[len(d) // 2] if d else None
        issue_ms = slots * SLOT_NS * 1e-6 / N_SIMD
        kernels.append({"kernel": kernel_id(k), "valu_wave_insts_per_launch": c["SQ_INSTS_VALU"],
                        "salu_wave_insts_per_launch": c.get("SQ_INSTS_SALU"),
                        "insts_per_unit": c["SQ_INSTS_VALU"] / (units / 64.0), "slots_per_unit": slots / (units / 64.0),
                        "issue_ms_per_launch": issue_ms, "kernel_ms_under_counters": ms,
                        "frac_of_issue": (issue_ms / ms) if ms else None,
                        "classes_per_unit": {n: x / (units / 64.0) for n, x in sorted(cls.items())}})
    kernels.sort(key=lambda r: -r["valu_wave_insts_per_launch"])
    tot_issue = sum(r["issue_ms_per_launch"] for r in kernels)
    tot_ms = sum(r["kernel_ms_under_counters"] or 0.0 for r in kernels)
    out = {"workload": w, "round": label, "units_per_launch": units,
           "insts_per_unit": sum(r["insts_per_unit"] for r in kernels), "slots_per_unit": sum(r["slots_per_unit"] for r in kernels),
           "issue_ms_per_launch": tot_issue, "kernel_ms_under_counters": tot_ms, "frac_of_issue": tot_issue / tot_ms if tot_ms else None,
           "slot": "issue time of one v_mul_f32 wave-instruction per SIMD = 1.155 ns (profiles/r03/valu_issue_cost.txt); 1024 SIMDs",
           "source": "tools/instmix.sh + tools/valu_report.py: SQ_INSTS_VALU_* passes of rocprofv3 --pmc, per-class issue costs of tools/valu_cost_probe.hip",
           "kernels": kernels}
    p = os.path.join(ROOT, "gpurun_out", f"valu_{w}{tag}.json")
    json.dump(out, open(p, "w"), indent=1)
    print(json.dumps({k: out[k] for k in ("workload", "insts_per_unit", "slots_per_unit", "issue_ms_per_launch", "kernel_ms_under_counters", "frac_of_issue")}))


if __name__ == "__main__":
    main()
