#!/bin/bash
# round 3, GPU call 28: randomised differential run of the final round-3 build against the oracle (tests/fuzz_parity.py), three seeds
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; O=gpurun_out/r03; mkdir -p $O
: > $O/fuzz_round3.txt
for seed in ${SEEDS:-9401 9402 9403}; do
  PYTHONPATH=. timeout 1500 python tests/fuzz_parity.py ${ROUNDS:-20} 8e6 $seed 2>&1 | grep -v amdgpu | tail -3 >> $O/fuzz_round3.txt
done
cat $O/fuzz_round3.txt
