#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; O=gpurun_out/r03; mkdir -p $O
python - <<'PY' 2>&1 | grep -v amdgpu
import sys, os
sys.path.insert(0, os.getcwd())
from dj_brdf_amd import djb
ctx = djb.default_context(0)
for ndf in ("ggx", "beckmann"):
    for a in (1.0, 0.3, 0.05):
        g = getattr(djb, ndf)(djb.fresnel.ideal(), True, ctx=ctx)
        for fam in (0, 1, 3):
            print(ndf, a, fam, djb.selftest_contract(g, djb.microfacet.params.isotropic(a), n=1 << 24, seed=5, family=fam, ctx=ctx))
PY
DJB_KIND_RATES_CONTRACT=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ct -- python tools/kind_rates.py > $O/prof_ct.log 2>&1
python - <<'PY'
import glob
for f in glob.glob("gpurun_out/r03/prof_ct/**/*kernel_stats.csv", recursive=True):
    for l in open(f).read().splitlines()[:14]: print(l[:230])
PY
