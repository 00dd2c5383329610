#!/bin/bash
# round 3, GPU call 27: the N-rank path of bench.py on one GPU (every rank on device 0) with the pooled file pipeline
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; O=gpurun_out/r03; mkdir -p $O
for N in 1 2 4 8; do
  if [ $N = 1 ]; then
    timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --selftest-n 10000000 > $O/fit_ranks_$N.json 2>$O/fit_ranks_$N.err
  else
    DJB_BENCH_SHARE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 2950$N bench.py --gpus $N --steps 2 --warmup 1 --no-cpu-baseline --selftest-n 10000000 > $O/fit_ranks_$N.json 2>$O/fit_ranks_$N.err
  fi
done
python - <<'PY'
import json
for N in (1,2,4,8):
    try:
        r=json.loads(open(f"gpurun_out/r03/fit_ranks_{N}.json").read().strip().splitlines()[-1]); f=r["secondary"]["merl_fit_files_100"]
        print("ranks %d: files->alphas %.2f ms (first call %.2f; load %.2f, fit %.2f)   dense upload %.1f ms   compute only %.3f ms   merl_eval %.1f G/s" % (N, f["wall_ms"], f["first_call_wall_ms"], f["pipeline_ms"]["load"], f["pipeline_ms"]["fit"], f["dense_upload"]["wall_ms"], r["secondary"]["merl_fit_100"]["wall_ms"], r["value"]/1e9))
    except Exception as e: print(N, "ERR", e)
PY
