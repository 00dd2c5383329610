#!/usr/bin/env python3
"""fuzz_diff_seeds.py REF_EXE OUR_EXE ref.txt ours.txt [extra args...]: which seeds of two whole-program fuzz outputs differ, and for each of
them whether the REFERENCE itself is reproducible -- the reference reads past the end of a short conditional-quantile table in
tabular_anisotropic::qf2 (dj_brdf.h:2819 on an m_qf2 that compute_qf2, :3005-3037, left shorter than elev x azim): what it returns there is
whatever the heap holds, i.e. it depends on the seeds run before.  A seed whose reference output changes between "alone" and "in sequence",
with our (reproducible) output differing from it on exactly those lines, is the reference's undefined behaviour, not a difference."""
import subprocess, sys, os

ref_exe, our_exe, ref_txt, our_txt, extra = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5:]


def blocks(path):
    out, cur, key = {}, [], None
    for line in open(path):
        if line.startswith("== seed "):
            if key is not None: out[key] = "".join(cur)
            key, cur = int(line.split()[2]), []
        cur.append(line)
    if key is not None: out[key] = "".join(cur)
    return out


R, G = blocks(ref_txt), blocks(our_txt)
bad = sorted(k for k in R if R[k] != G.get(k))
print(f"{len(R)} seeds, {len(bad)} differ: {bad[:20]}")
real = 0
for k in bad:
    env = dict(os.environ, DJB_QUIET="1")
    alone_ref = subprocess.run([ref_exe, str(k), "1"] + extra, capture_output=True, text=True).stdout
    alone_our = subprocess.run([our_exe, str(k), "1"] + extra, capture_output=True, text=True, env=env).stdout
    rs, ra, go = R[k].splitlines(), alone_ref.splitlines(), G[k].splitlines()
    unstable = set(n for n, (a, b) in enumerate(zip(rs, ra)) if a != b) if len(rs) == len(ra) else None     # where the reference disagrees with ITSELF
    ours_off = set(n for n, (a, b) in enumerate(zip(rs, go)) if a != b) | set(n for n, (a, b) in enumerate(zip(ra, go)) if a != b) if len(rs) == len(go) else None
    if alone_our != G[k]:
        real += 1; print(f"  seed {k}: OUR output depends on the seeds run before it")
    elif unstable and ours_off is not None and ours_off <= unstable:
        print(f"  seed {k}: the reference is not reproducible (alone vs in sequence, lines {sorted(unstable)}: {rs[min(unstable)].split()[0]}); we differ from it on those lines only: reference UB")
    else:
        real += 1
        print(f"  seed {k}: A REAL DIFFERENCE (reference reproducible: {alone_ref == R[k]}; lines {sorted(ours_off or [])[:6]})")
sys.exit(1 if real else 0)
