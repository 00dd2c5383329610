#!/bin/bash
# round 5: the two whole-program fuzzers over thousands of seeds on the GPU box (the reference binaries of oracle/_ref run on its host cores,
# in 8 processes; the facade side from 8 threads on one context).  A difference is looked at seed by seed (fuzz_diff_seeds.py): the reference
# reads past the end of a short m_qf2 table in tabular_anisotropic::qf2 -- its output for such a seed depends on the seeds run before it.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out /tmp/soak/g; O=gpurun_out/fuzz_soak.txt; : > $O
PARTS=${PARTS:-"api custom"}; FIRST=${FIRST:-100000}; FIRST_MERL=$((FIRST + 100000))   # FIRST: another seed range (a re-run on a later tree)
ref_par() { # program first count extra...: the reference side split over 8 processes, concatenated in seed order (rX -> a scratch of the process's own)
  local prog=$1 first=$2 count=$3; shift 3; local per=$((count / 8))
  for p in 0 1 2 3 4 5 6 7; do mkdir -p /tmp/soak/r$p; ( ./oracle/_ref/$prog $((first + p * per)) $per "${@//rX/r$p}" > /tmp/soak/ref_$p.txt ) & done; wait
  cat /tmp/soak/ref_[0-7].txt
}
check() { # label prog extra... ; stdin = our output
  local label=$1 prog=$2; shift 2
  cat > /tmp/soak/o.txt
  if cmp -s /tmp/soak/o.txt /tmp/soak/r.txt; then echo "$label: IDENTICAL ($(wc -l < /tmp/soak/r.txt) lines)" >> $O
  else echo "$label: $(python3 tools/exp/r05/fuzz_diff_seeds.py ./oracle/_ref/$prog ./examples/$prog /tmp/soak/r.txt /tmp/soak/o.txt "$@" | tr '\n' ' ')" >> $O; fi
}
if [[ $PARTS == *api* ]]; then
ref_par api_fuzz $FIRST 3200 /tmp/soak/rX > /tmp/soak/r.txt
./examples/api_fuzz $FIRST 3200 /tmp/soak/g threads=8 | check "api_fuzz 3200 seeds, GPU, 8 threads" api_fuzz /tmp/soak/g
DJB_SCALAR_ON_DEVICE=1 ./examples/api_fuzz $FIRST 3200 /tmp/soak/g threads=8 | check "api_fuzz 3200 seeds, one-pair calls through the kernels" api_fuzz /tmp/soak/g
DJB_DEVICE=cpu ./examples/api_fuzz $FIRST 3200 /tmp/soak/g threads=8 | check "api_fuzz 3200 seeds, host path" api_fuzz /tmp/soak/g
ref_par api_fuzz $FIRST_MERL 240 /tmp/soak/rX merl > /tmp/soak/r.txt
./examples/api_fuzz $FIRST_MERL 240 /tmp/soak/g merl threads=8 | check "api_fuzz 240 seeds with MERL files, GPU" api_fuzz /tmp/soak/g merl
fi
if [[ $PARTS == *custom* ]]; then
ref_par custom_brdf_fuzz $FIRST 3200 > /tmp/soak/r.txt
./examples/custom_brdf_fuzz $FIRST 3200 threads=8 | check "custom_brdf_fuzz 3200 seeds, GPU, 8 threads" custom_brdf_fuzz
DJB_SCALAR_ON_DEVICE=1 ./examples/custom_brdf_fuzz $FIRST 3200 threads=8 | check "custom_brdf_fuzz 3200 seeds, one-pair calls through the kernels" custom_brdf_fuzz
DJB_DEVICE=cpu ./examples/custom_brdf_fuzz $FIRST 3200 threads=8 | check "custom_brdf_fuzz 3200 seeds, host path" custom_brdf_fuzz
fi
cat $O
