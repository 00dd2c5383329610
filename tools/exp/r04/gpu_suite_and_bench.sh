#!/bin/bash
# round 4: smoke(), the whole GPU suite and the default bench line of the tree
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; tail -2 gpurun_out/smoke.log
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/pytest_gpu.log
tail -4 gpurun_out/pytest_gpu.log
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -c 600 gpurun_out/bench_default.json; tail -3 gpurun_out/bench_default.err
