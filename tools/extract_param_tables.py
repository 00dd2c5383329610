#!/usr/bin/env python3
"""Extract the published SGD (Bagher et al. 2012) and ABC (Low et al. 2012) per-material parameter
tables that the reference embeds as data (dj_brdf.h:3312-3413, 3505-3606) into plain CSV files
under dj_brdf_amd/data/.  Build-container only (needs /root/reference); the CSVs are committed.
Only numbers and material names are extracted -- no code."""
import csv
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/dj_brdf.h"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dj_brdf_amd", "data")
src = open(REF).read()
NUM = r"[-+]?(?:\d+\.?\d*|\.\d+)(?:[eE][-+]?\d+)?"


def rows_of(table_decl):
    start = src.index(table_decl)
    body = src[src.index("{", start) + 1: src.index("\n};", start)]
    for line in body.split("\n"):
        line = line.strip()
        if line.startswith("{"):
            names = re.findall(r'"([^"]*)"', line)
            nums = re.findall(NUM, re.sub(r'"[^"]*"', "", line))
            yield names, nums


sgd_fields = ["rhoD", "rhoS", "alpha", "p", "f0", "f1", "kap", "lambda", "c", "k", "theta0", "error"]
with open(os.path.join(OUT, "sgd_params.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["name", "other_name"] + [f"{fld}_{ch}" for fld in sgd_fields for ch in "rgb"])
    n = 0
    for names, nums in rows_of("const sgd::data sgd::s_data[]"):
        assert len(names) == 2 and len(nums) == 36, (names, len(nums))
        w.writerow(names + nums); n += 1
    assert n == 100, n
with open(os.path.join(OUT, "abc_params.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["name", "kD_r", "kD_g", "kD_b", "A_r", "A_g", "A_b", "B", "C", "ior"])
    n = 0
    for names, nums in rows_of("const abc::data abc::s_data[]"):
        assert len(names) == 1 and len(nums) == 9, (names, len(nums))
        w.writerow(names + nums); n += 1
    assert n == 100, n
print("wrote", OUT)
