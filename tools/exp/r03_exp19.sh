#!/bin/bash
# round 3, GPU call 19: Beckmann sample two-path kernel: adversarial parity test + the sampling tests, rocprofv3 passes, instruction mix
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; O=gpurun_out/r03; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q -k "sample or histogram or evalp_is or scalar or facade or golden" > $O/sample_tests.txt 2>&1; echo "rc=$?" >> $O/sample_tests.txt; tail -4 $O/sample_tests.txt
WORKLOADS=beckmann_sample timeout 900 bash tools/profile_bench.sh > gpurun_out/profile_bench.log 2>&1
timeout 300 bash tools/instmix.sh beckmann_sample > gpurun_out/instmix_beckmann_sample.txt 2>&1
cat gpurun_out/prof/beckmann_sample/bench_plain.json | tail -1
head -30 gpurun_out/instmix_beckmann_sample.txt
