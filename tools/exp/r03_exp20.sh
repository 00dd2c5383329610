#!/bin/bash
# round 3, GPU call 20: full GPU suite + smoke + default bench after the two-path sampler
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; O=gpurun_out/r03; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/gpu_suite.txt 2>&1; echo "rc=$?" >> $O/gpu_suite.txt; tail -4 $O/gpu_suite.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -2
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 3000 $O/bench_default.json
