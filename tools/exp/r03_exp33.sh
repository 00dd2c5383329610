#!/bin/bash
# round 3, GPU call 33: one agent-scope release / acquire per workgroup in k_fit's slice exchange: fit tests, full suite, and a repeat-stress
# (300 x 100 materials, 300 x 13, 300 x 1 at res 90 and 64: every table of every repeat against the first)
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; O=gpurun_out/r03; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
PYTHONPATH=. timeout 900 python - <<'PY'
import ctypes as C
import numpy as np
from dj_brdf_amd import djb, synth, _lib
lib = _lib.load()
ctx = djb.Context(0)
bad = 0; total = 0
for n, res in ((100, 90), (13, 90), (1, 90), (37, 64), (7, 90)):
    mats = [djb.merl.from_table(synth.merl_table(*synth.material_recipe(k)), ctx=ctx) for k in range(n)]
    ptrs = (C.c_void_p * n)(*[b._h.value for b in mats])
    ref = None
    for rep in range(300):
        ab, ag = np.zeros(n, np.float32), np.zeros(n, np.float32)
        tabs = [np.zeros(n * res, np.float32) for _ in range(4)] + [np.zeros(3 * n * res, np.float32)]
        _lib.check(lib.djb_fit_brdf_batch(ctx._h, C.c_int(n), ptrs, C.c_int(res), C.c_int(1), C.c_void_p(ab.ctypes.data), C.c_void_p(ag.ctypes.data),
                                          *[C.c_void_p(t.ctypes.data) for t in tabs]))
        flat = np.concatenate([x.view(np.uint32) for x in [ab, ag] + tabs])
        if ref is None: ref = flat
        elif not np.array_equal(ref, flat): bad += 1
        total += 1
    print(f"n={n} res={res}: 300 repeats x {ref.size} values")
print("repeats that differ from the first:", bad, "of", total)
PY
