#!/bin/bash
# round 3, GPU call 5: f32-seeded guarded rsq/rcp/sqrt A/B, whole gpu suite, N-rank fit self-test (shared GPU), scalar break-even
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; O=gpurun_out/r03; mkdir -p $O
B="python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-secondary"
for rep in 1 2; do
for lib in gpurun_variants/libdjb_prev.so ""; do
  tag=$([ -z "$lib" ] && echo new || echo prev)
  DJB_LIB_PATH=$lib timeout 300 $B --workload beckmann_sample --n 250000000 > $O/g_${tag}_beckmann_sample_$rep.json 2>$O/g.err
  DJB_LIB_PATH=$lib timeout 300 $B --workload ggx_eval_pdf > $O/g_${tag}_ggx_eval_pdf_$rep.json 2>>$O/g.err
done; done
for lib in gpurun_variants/libdjb_prev.so ""; do
  tag=$([ -z "$lib" ] && echo new || echo prev)
  DJB_LIB_PATH=$lib timeout 300 python tools/kind_rates.py > $O/kind_rates_$tag.txt 2>&1
  DJB_LIB_PATH=$lib timeout 300 python tools/sample_rates.py > $O/sample_rates_$tag.txt 2>&1
done
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu_suite2.txt 2>&1; echo "gpu suite rc=$?" >> $O/gpu_suite2.txt
nproc > $O/fit_ranks.txt
for N in 1 2 4 8; do
  if [ $N = 1 ]; then
    timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --selftest-n 10000000 > $O/fit_ranks_$N.json 2>$O/fit_ranks_$N.err
  else
    DJB_BENCH_SHARE_GPU=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 2950$N bench.py --gpus $N --steps 2 --warmup 1 --no-cpu-baseline --selftest-n 10000000 > $O/fit_ranks_$N.json 2>$O/fit_ranks_$N.err
  fi
done
timeout 600 python tools/scalar_breakeven.py > $O/scalar_breakeven.txt 2>&1
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03/g_*.json")):
    try:
        r=json.loads(open(f).read().strip().splitlines()[-1]); print("%-44s %8.3f ms" % (f.split('/')[-1], r["ms_per_step"]))
    except Exception as e: print(f, "ERR", e)
for N in (1,2,4,8):
    try:
        r=json.loads(open(f"gpurun_out/r03/fit_ranks_{N}.json").read().strip().splitlines()[-1]); s=r["secondary"]
        print("ranks", N, "files->alphas %.2f ms (load %.2f fit %.2f)  dense %.1f ms  compute-only %.3f ms" % (s["merl_fit_files_100"]["wall_ms"], s["merl_fit_files_100"]["pipeline_ms"]["load"], s["merl_fit_files_100"]["pipeline_ms"]["fit"], s["merl_fit_files_100"]["dense_upload"]["wall_ms"], s["merl_fit_100"]["wall_ms"]))
    except Exception as e: print(N, "ERR", e)
PY
tail -4 $O/gpu_suite2.txt; cat $O/kind_rates_prev.txt $O/kind_rates_new.txt $O/sample_rates_prev.txt $O/sample_rates_new.txt; cat $O/scalar_breakeven.txt
