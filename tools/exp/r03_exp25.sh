#!/bin/bash
# round 3, GPU call 25: divisions by pi / float(pi) replaced by guarded reciprocal multiplies (Beckmann p22, brdf::pdf, the tabular
# lobes' acos_u / acos_u32 / atan_u / atan_squ sites): exhaustive sweep of the four sites, full suite, rates
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; O=gpurun_out/r03; mkdir -p $O
PYTHONPATH=. timeout 900 python tools/exhaustive_trig.py --sites acos_u,acos_u32,atan_squ,atan_u --out $O/exhaustive_trig_sites.json > $O/exhaustive_trig_sites.log 2>&1; cat $O/exhaustive_trig_sites.log | tail -5
timeout 2400 python -m pytest tests -m gpu -x -q > $O/gpu_suite.txt 2>&1; echo "rc=$?" >> $O/gpu_suite.txt; tail -3 $O/gpu_suite.txt
PYTHONPATH=. timeout 600 python tools/kind_rates.py 2>/dev/null | grep -v amdgpu
