#!/bin/bash
# round 3, GPU call 6: how the sparse file gather scales with reader threads, by mapping mode
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; O=gpurun_out/r03; mkdir -p $O
for m in 0 1 2 3; do FIT_RATES_SPARSE_ONLY=1 DJB_GATHER_MODE=$m timeout 600 python tools/fit_files_rates.py > $O/fit_gather_mode_$m.txt 2>&1; done
cat /sys/kernel/mm/transparent_hugepage/enabled /proc/sys/vm/overcommit_memory 2>/dev/null; uname -r; nproc; numactl -H 2>/dev/null | head -5; df -h /tmp | tail -1
for m in 0 1 2 3; do cat $O/fit_gather_mode_$m.txt | grep -v amdgpu.ids; done
