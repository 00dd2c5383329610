#!/bin/bash
# round 5 (bench.py gained the cold-cache leg and the profile checks): bench.py's N-rank path on one GPU (DJB_BENCH_SHARE_GPU=1: all ranks on device 0, gloo control plane; labelled, not a scaling
# measurement): placement records, CPU pinning, reader-pool sizing, scaling_model_ms, first_call_wall_ms
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
for N in 2 8; do
  DJB_BENCH_SHARE_GPU=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29500 + N)) bench.py --gpus $N --steps 3 --warmup 1 --selftest-n 20000000 > gpurun_out/bench_share_$N.json 2> gpurun_out/bench_share_$N.err
  echo "N=$N rc=$?"; tail -c 2500 gpurun_out/bench_share_$N.json; tail -3 gpurun_out/bench_share_$N.err
done
