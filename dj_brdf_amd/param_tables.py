"""The published per-material parameter tables of the SGD (Bagher et al. 2012) and ABC
(Low et al. 2012) models, as shipped in dj_brdf_amd/data/*.csv (extracted by
tools/extract_param_tables.py from the data tables of the reference, dj_brdf.h:3312-3413,
3505-3606).  Rows are keyed by MERL material name; SGD rows also answer to their alias."""
from __future__ import annotations

import csv
import os
from functools import lru_cache

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
SGD_FIELDS = ["rhoD", "rhoS", "alpha", "p", "f0", "f1", "kap", "lambda", "c", "k", "theta0"]


@lru_cache(maxsize=None)
def _sgd():
    rows = {}
    with open(os.path.join(_DATA, "sgd_params.csv")) as f:
        for r in csv.DictReader(f):
            vals = [float(r[f"{fld}_{ch}"]) for fld in SGD_FIELDS for ch in "rgb"]
            rows[r["name"]] = vals
            if r["other_name"]:
                rows.setdefault(r["other_name"], vals)
    return rows


@lru_cache(maxsize=None)
def _abc():
    rows = {}
    with open(os.path.join(_DATA, "abc_params.csv")) as f:
        for r in csv.DictReader(f):
            rows[r["name"]] = [float(r[k]) for k in ("kD_r", "kD_g", "kD_b", "A_r", "A_g", "A_b", "B", "C", "ior")]
    return rows


def sgd_params(name: str):
    """33 doubles: rhoD rhoS alpha p f0 f1 kap lambda c k theta0 (x RGB), or KeyError."""
    return _sgd()[name]


def abc_params(name: str):
    """9 doubles: kD[3] A[3] B C ior, or KeyError."""
    return _abc()[name]


def sgd_names():
    return sorted(_sgd())


def abc_names():
    return sorted(_abc())
