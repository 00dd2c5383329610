#!/bin/bash
# round 4: table gathers that fetch whole 128-byte lines (neighbouring lanes read a texel and a word of the sibling sector in the same
# load instruction; -DDJB_EXP_MERL_PAIR_LINES variant in gpurun_variants/libdjb_pl.so) -> profiles/r04/merl_pair_lines.txt
# variant: the tree of the commit "experiment: MERL gathers that ask for both halves of each 128-byte line", make BUILD=build_pl OUT=../../gpurun_variants/libdjb_pl.so EXTRA=-DDJB_EXP_MERL_PAIR_LINES
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; mkdir -p gpurun_out
O=$R/gpurun_out/merl_pair_lines.txt; : > $O
V=gpurun_variants/libdjb_pl.so; S=dj_brdf_amd/lib/libdjb_hip.so
DJB_LIB_PATH=$V timeout 900 python -m pytest tests/test_gpu_verification.py tests/test_gpu_golden.py -m gpu -q -x 2>&1 | tail -1 >> $O
run() { # label lib workload
  A=""; case $3 in merl_eval_*) A="--n 250000000";; esac
  DJB_LIB_PATH=$2 timeout 300 python bench.py --workload $3 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary $A 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('%-12s %-24s %8.3f ms/step  frac %.3f' % ('$1', '$3', d['ms_per_step'], d['roofline']['frac']))" >> $O
}
for rep in 1 2 3; do for w in merl_eval merl_eval_uniform_bins merl_eval_coherent; do run shipped $S $w; run pair-lines $V $w; done; done
# counters: read requests and L2 hit rate of the look-up kernel
for v in shipped pair-lines; do
  lib=$S; [ $v = pair-lines ] && lib=$V
  for pmc in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
    D=$R/gpurun_out/pl_prof/$v/$(echo $pmc | tr ' ' '_'); rm -rf $D; mkdir -p $D
    DJB_LIB_PATH=$lib rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $D -- python bench.py --workload merl_eval --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > /dev/null 2>&1
    python - "$D" "$v" >> $O <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_merl_fast_v4" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()): print("%-12s %-26s %.4g per launch" % (sys.argv[2], k, sum(v) / len(v)))
PY
  done
done
cat $O
