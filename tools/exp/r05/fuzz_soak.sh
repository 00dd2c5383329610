#!/bin/bash
# round 5: the two whole-program fuzzers over thousands of seeds on the GPU box (the reference binaries of oracle/_ref run on its host cores,
# in 8 processes; the facade side from 8 threads on one GPU context)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out /tmp/soak; O=gpurun_out/fuzz_soak.txt; : > $O
ref_par() { # program first count extra...: the reference side split over 8 processes, concatenated in seed order
  local prog=$1 first=$2 count=$3; shift 3; local per=$((count / 8))
  for p in 0 1 2 3 4 5 6 7; do mkdir -p /tmp/soak/r$p; ( ./oracle/_ref/$prog $((first + p * per)) $per "${@//rX/r$p}" > /tmp/soak/ref_$p.txt ) & done; wait
  cat /tmp/soak/ref_[0-7].txt
}
mkdir -p /tmp/soak/g
ref_par api_fuzz 100000 3200 /tmp/soak/rX > /tmp/soak/r.txt
./examples/api_fuzz 100000 3200 /tmp/soak/g threads=8 | cmp - /tmp/soak/r.txt && echo "api_fuzz 3200 seeds, GPU, 8 threads: IDENTICAL ($(wc -l < /tmp/soak/r.txt) lines)" >> $O
DJB_SCALAR_ON_DEVICE=1 ./examples/api_fuzz 100000 3200 /tmp/soak/g threads=8 | cmp - /tmp/soak/r.txt && echo "api_fuzz 3200 seeds, one-pair calls through the kernels: IDENTICAL" >> $O
DJB_DEVICE=cpu ./examples/api_fuzz 100000 3200 /tmp/soak/g threads=8 | cmp - /tmp/soak/r.txt && echo "api_fuzz 3200 seeds, host path: IDENTICAL" >> $O
ref_par api_fuzz 200000 240 /tmp/soak/rX merl > /tmp/soak/r.txt
./examples/api_fuzz 200000 240 /tmp/soak/g merl threads=8 | cmp - /tmp/soak/r.txt && echo "api_fuzz 240 seeds with MERL files, GPU: IDENTICAL" >> $O
ref_par custom_brdf_fuzz 100000 3200 > /tmp/soak/r.txt
./examples/custom_brdf_fuzz 100000 3200 threads=8 | cmp - /tmp/soak/r.txt && echo "custom_brdf_fuzz 3200 seeds, GPU, 8 threads: IDENTICAL ($(wc -l < /tmp/soak/r.txt) lines)" >> $O
DJB_DEVICE=cpu ./examples/custom_brdf_fuzz 100000 3200 threads=8 | cmp - /tmp/soak/r.txt && echo "custom_brdf_fuzz 3200 seeds, host path: IDENTICAL" >> $O
cat $O
