#!/bin/bash
# round 4: GPU suite + default bench line after the tier-1 MERL rewrite and the contract-mode sampler
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/pytest_gpu.log
tail -8 gpurun_out/pytest_gpu.log
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -c 1500 gpurun_out/bench_default.json; tail -3 gpurun_out/bench_default.err
