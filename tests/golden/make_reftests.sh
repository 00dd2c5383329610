#!/bin/bash
# Expected outputs of the reference's own test / example programs, produced by the REAL reference
# (compiled in place from /root/reference, run on the CPU) -> tests/golden/reftests/.
# tests/test_gpu_golden.py::test_reference_programs_unchanged compares the same programs, compiled
# unchanged against include/dj_brdf.h and run on the GPU, with these files byte for byte.
set -e
REF=${REF:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=$HERE/reftests
W=$(mktemp -d)
mkdir -p "$OUT"
for f in tests/plot_cdf tests/plot_qf tests/nrm_utia; do
	g++ -O2 -DNVERBOSE -I"$REF" -o "$W/$(basename $f)" "$REF/$f.cpp"
done
(cd "$W" && ./plot_cdf && ./plot_qf && cp eval_*.txt "$OUT/")
# nrm_utia on a UTIA-format file that violates the white furnace at the first outgoing direction
# (all samples = 140 * 0.9): the program stops after one row of the quadrature
python3 - "$W/furnace_fail.bin" <<'PY'
import sys, numpy as np
np.full(3 * 288 * 288, 140.0 * 0.9).tofile(sys.argv[1])
PY
(cd "$W" && { ./nrm_utia furnace_fail.bin > out.txt && echo "exit=0" >> out.txt || echo "exit=$?" >> out.txt; } && cp out.txt "$OUT/nrm_utia_fail.txt")
# a user program with classes DERIVED from djb::brdf and djb::fresnel::impl (examples/custom_brdf.cpp, written against the
# reference's interface only): what the real reference prints for it
g++ -O2 -ffp-contract=off -DNVERBOSE -w -I"$REF" -o "$W/custom_brdf" "$HERE/../../examples/custom_brdf.cpp" -lm
"$W/custom_brdf" > "$OUT/custom_brdf.txt"
# its randomised companion (random user lobes / Fresnel terms / NDFs, random resolutions and directions): seeds 1..6
g++ -O2 -ffp-contract=off -DNVERBOSE -w -I"$REF" -o "$W/custom_brdf_fuzz" "$HERE/../../examples/custom_brdf_fuzz.cpp" -lm
"$W/custom_brdf_fuzz" 1 6 > "$OUT/custom_brdf_fuzz.txt"
# the library's own classes through the reference's public interface with random arguments (examples/api_fuzz.cpp): seeds 1..4
g++ -O2 -ffp-contract=off -DNVERBOSE -w -I"$REF" -o "$W/api_fuzz" "$HERE/../../examples/api_fuzz.cpp" -lm
"$W/api_fuzz" 1 4 "$W" > "$OUT/api_fuzz.txt"
rm -rf "$W"
ls -la "$OUT"
