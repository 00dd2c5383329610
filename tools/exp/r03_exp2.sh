#!/bin/bash
# round 3, GPU call 2: Newton-loop divergence bound (fixed trip counts), contract grid sweep, kernel split of the uniform-bins MERL leg
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; O=gpurun_out/r03; mkdir -p $O
B="python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-secondary"
for v in "" nf3 nf4; do
  lib=$([ -z "$v" ] && echo "" || echo gpurun_variants/libdjb_$v.so)
  DJB_LIB_PATH=$lib timeout 300 $B --workload beckmann_sample --n 250000000 > $O/newton_${v:-ship}.json 2>$O/newton_${v:-ship}.err
done
for cap in 2048 4096 8192 16384 32768 10000000; do
  DJB_LIB_PATH=gpurun_variants/libdjb_exp.so DJB_CT_GRID_CAP=$cap timeout 300 $B --workload ggx_eval_pdf_contract > $O/ctgrid_$cap.json 2>$O/ctgrid_$cap.err
done
DJB_LIB_PATH=gpurun_variants/libdjb_exp.so DJB_CT_NOFIX=1 timeout 300 $B --workload ggx_eval_pdf_contract > $O/ctgrid_nofix.json 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_uniform -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --workload merl_eval_uniform_bins --n 250000000 > $O/prof_uniform.log 2>&1
python - <<'PY'
import json,glob,csv
for f in sorted(glob.glob("gpurun_out/r03/newton_*.json")+glob.glob("gpurun_out/r03/ctgrid_*.json")):
    try:
        r=json.loads(open(f).read().strip().splitlines()[-1])
        print("%-44s %8.3f ms  %8.2f G/s  frac %.3f" % (f.split('/')[-1], r["ms_per_step"], r["value"]/1e9, r["roofline"]["frac"] or 0))
    except Exception as e: print(f, "ERR", e)
for f in glob.glob("gpurun_out/r03/prof_uniform/**/*kernel_stats.csv", recursive=True):
    print(open(f).read()[:1500])
PY
