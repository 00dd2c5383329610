#!/bin/bash
# round 5: the full GPU suite, the sampler rates (evalp_is keeps its trip loop rolled: does the peeled last trip pay there too?), the default bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; tail -5 gpurun_out/pytest_gpu.log
PYTHONPATH=. timeout 600 python tools/sample_rates.py > gpurun_out/sample_rates.txt 2>&1; tail -12 gpurun_out/sample_rates.txt
timeout 1200 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -c 600 gpurun_out/bench_default.json
