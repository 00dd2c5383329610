#!/bin/bash
# round 5: the product's host path (dj_brdf_amd/csrc/djb_cpu.cpp: the per-unit code with DJB_HOST_MATH, both fitters, the file gather) built
# with AddressSanitizer + UndefinedBehaviorSanitizer (gcc), the whole-program fuzzers and the example programs built likewise and run on a
# CPU context; outputs must be the reference's and the sanitizers silent (LeakSanitizer included).  Runs in the build container.
set -e
cd "$(dirname "$0")/../../.."; ROOT=$PWD; W=${W:-/tmp/djb_san}; mkdir -p $W/s $W/r gpurun_variants
make -s -C dj_brdf_amd/csrc BUILD=build_asan OUT=../../gpurun_variants/libdjb_asan.so CXX="g++ -fsanitize=address,undefined -fno-omit-frame-pointer -g" -j16
SAN="-O1 -g -fsanitize=address,undefined -std=c++14 -DNVERBOSE -I$ROOT/include -L$ROOT/gpurun_variants -l:libdjb_asan.so -Wl,-rpath,$ROOT/gpurun_variants -pthread"
for p in api_fuzz custom_brdf_fuzz custom_brdf facade_check merl_params; do g++ -o $W/$p examples/$p.cpp $SAN; done
export DJB_DEVICE=cpu DJB_QUIET=1 ASAN_OPTIONS=detect_leaks=1
R=oracle/_ref
$W/api_fuzz 1 20 $W/s merl 2> $W/e1 | cmp - <($R/api_fuzz 1 20 $W/r merl) && echo "api_fuzz 20 seeds + MERL files: reference's bytes"
$W/api_fuzz 100 64 $W/s threads=8 2> $W/e2 | cmp - <($R/api_fuzz 100 64 $W/r) && echo "api_fuzz 64 seeds from 8 threads: reference's bytes"
$W/custom_brdf_fuzz 1 100 2> $W/e3 | cmp - <($R/custom_brdf_fuzz 1 100) && echo "custom_brdf_fuzz 100 seeds: reference's bytes"
$W/custom_brdf_fuzz 200 64 threads=8 2> $W/e4 | cmp - <($R/custom_brdf_fuzz 200 64) && echo "custom_brdf_fuzz 64 seeds from 8 threads: reference's bytes"
$W/custom_brdf 2> $W/e5 | cmp - tests/golden/reftests/custom_brdf.txt && echo "custom_brdf: reference's bytes"
$W/facade_check > /dev/null 2> $W/e6 && echo "facade_check: ok"
python3 -c "
import sys; sys.path.insert(0, '$ROOT')
from dj_brdf_amd import synth
for k, a in enumerate((0.3, 0.1, 0.05)): synth.write_merl_binary('$W/s/m%d.binary' % k, synth.merl_table(alpha=a))"
(cd $W/s && for mode in "" "-s"; do ../merl_params $mode m0.binary m1.binary m2.binary > /dev/null 2>> $W/e7 && cp params.txt p$mode.txt; done && $ROOT/examples/merl_params m0.binary m1.binary m2.binary > /dev/null && cmp params.txt p.txt && cmp params.txt p-s.txt && echo "merl_params (file pipeline, both modes): the unsanitized build's params.txt")
echo "sanitizer reports: $(cat $W/e1 $W/e2 $W/e3 $W/e4 $W/e5 $W/e6 $W/e7 | wc -c) bytes"
# ThreadSanitizer: the same host path, the two fuzzers from 8 threads on the shared CPU context
make -s -C dj_brdf_amd/csrc BUILD=build_tsan OUT=../../gpurun_variants/libdjb_tsan.so CXX="g++ -fsanitize=thread -fno-omit-frame-pointer -g" -j16
TSAN="-O1 -g -fsanitize=thread -std=c++14 -DNVERBOSE -I$ROOT/include -L$ROOT/gpurun_variants -l:libdjb_tsan.so -Wl,-rpath,$ROOT/gpurun_variants -pthread"
g++ -o $W/api_fuzz_t examples/api_fuzz.cpp $TSAN; g++ -o $W/custom_brdf_fuzz_t examples/custom_brdf_fuzz.cpp $TSAN
$W/api_fuzz_t 100 32 $W/s threads=8 2> $W/t1 | cmp - <($R/api_fuzz 100 32 $W/r) && echo "TSan: api_fuzz 32 seeds from 8 threads: reference's bytes"
$W/custom_brdf_fuzz_t 200 32 threads=8 2> $W/t2 | cmp - <($R/custom_brdf_fuzz 200 32) && echo "TSan: custom_brdf_fuzz 32 seeds from 8 threads: reference's bytes"
echo "ThreadSanitizer reports: $(cat $W/t1 $W/t2 | wc -c) bytes"
rm -rf dj_brdf_amd/csrc/build_asan dj_brdf_amd/csrc/build_tsan gpurun_variants/libdjb_asan.so gpurun_variants/libdjb_tsan.so
