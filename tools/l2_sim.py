#!/usr/bin/env python3
"""What an XCD's L2 can do for the MERL table gathers: the look-ups of the bench distribution (uniform hemisphere pairs through the
product's host path) replayed through a model cache -- 16-way LRU, 128-byte lines filled whole on a miss, texels of 12 bytes --
at several capacities, next to the coverage of the statically hottest lines (what no replacement policy can beat).
    python tools/l2_sim.py > profiles/r04/merl_l2_sim.txt        (CPU only; compiles tools/l2_sim.c)
mode 1 / 2: the rows outside the hottest `hot` MB (by prior) bypass the cache / are inserted at the LRU position -- the retention
hint tried with nt / sc1 gathers in round 4 (profiles/r04/merl_cold_policy.txt: the hardware does not reward it)."""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dj_brdf_amd import djb, synth  # noqa: E402

n = 1 << 24
ctx = djb.cpu_context()
idx = np.concatenate([np.asarray(djb.merl_index(synth.directions_aos(1 << 22, synth.SEED_I, lo), synth.directions_aos(1 << 22, synth.SEED_O, lo), ctx=ctx))
                      for lo in range(0, n, 1 << 22)]).astype(np.uint32)
tmp = tempfile.mkdtemp()
exe = os.path.join(tmp, "l2_sim")
subprocess.run(["gcc", "-O2", "-o", exe, os.path.join(ROOT, "tools", "l2_sim.c")], check=True)
idx.tofile(os.path.join(tmp, "idx.bin"))
line = (idx.astype(np.int64) * 12) // 128
c = np.sort(np.bincount(line))[::-1].astype(np.float64); cs = np.cumsum(c) / c.sum()
print("look-ups: %d, distinct texels %d, 128-byte lines touched %d (%.1f MB)" % (n, len(np.unique(idx)), int((c > 0).sum()), (c > 0).sum() * 128 / 2**20))
print("coverage of the statically hottest lines: " + ", ".join("%d MB %.3f" % (mb, cs[mb * 8192 - 1]) for mb in (1, 2, 3, 4, 6, 8)))
row = idx.astype(np.int64) // 180
p = np.bincount(row, minlength=8100).astype(np.float64); order = np.argsort(-p)
allhot = np.ones(8100, np.uint8); allhot.tofile(os.path.join(tmp, "hot_all.bin"))
print("LRU, every gather allocates (the shipped kernel; the counters say 0.314 misses per look-up -- profiles/pmc_merl_eval.json):")
for cap in (2, 3, 3.5, 4):
    print("  %.1f MB: " % cap + subprocess.run([exe, os.path.join(tmp, "idx.bin"), str(int(cap * 2**20)), os.path.join(tmp, "hot_all.bin"), "0"], capture_output=True, text=True).stdout.strip())
for mb in (3.0, 3.5):
    hot = np.zeros(8100, np.uint8); hot[order[:int(mb * 2**20 / 2160)]] = 1
    f = os.path.join(tmp, "hot.bin"); hot.tofile(f)
    for mode in (1, 2):
        print("hottest %.1f MB of rows kept, the others %s, 4 MB: " % (mb, "bypass" if mode == 1 else "at the LRU position") +
              subprocess.run([exe, os.path.join(tmp, "idx.bin"), str(4 << 20), f, str(mode)], capture_output=True, text=True).stdout.strip())
