#!/bin/bash
# round 3, GPU call 1: contract-mode tests + rates, SLP on/off A/B of the bit-exact kernels, the two extra MERL legs
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/r03; mkdir -p $O
B="python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-secondary"
timeout 900 python -m pytest tests/test_gpu_contract.py -x -q -s > $O/contract_tests.txt 2>&1; echo "contract tests rc=$?" | tee -a $O/contract_tests.txt
for w in ggx_eval_pdf ggx_eval_pdf_contract; do timeout 300 $B --workload $w > $O/bench_$w.json 2>$O/bench_$w.err; done
timeout 300 $B --workload ggx_eval_pdf_contract --fresnel schlick > $O/bench_ggx_contract_schlick.json 2>&1
for lib in "" gpurun_variants/libdjb_noslp.so; do
  tag=$([ -z "$lib" ] && echo ship || echo noslp)
  for w in ggx_eval_pdf beckmann_sample merl_eval utia_eval; do
    n=$([ $w = ggx_eval_pdf -o $w = utia_eval ] && echo 100000000 || echo 250000000)
    DJB_LIB_PATH=$lib timeout 300 $B --workload $w --n $n > $O/ab_${tag}_$w.json 2>$O/ab_${tag}_$w.err
  done
done
for w in merl_eval_uniform_bins merl_eval_coherent; do timeout 600 $B --workload $w --n 250000000 > $O/bench_$w.json 2>$O/bench_$w.err; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03/*.json")):
    try:
        r=json.loads(open(f).read().strip().splitlines()[-1])
        print("%-44s %8.3f ms  %8.2f G/s  frac %.3f" % (f.split('/')[-1], r["ms_per_step"], r["value"]/1e9, r["roofline"]["frac"] or 0))
    except Exception as e: print(f, "ERR", e)
PY
tail -30 $O/contract_tests.txt
