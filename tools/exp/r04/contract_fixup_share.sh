cd /tmp && export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
rm -rf gpurun_out/ctprof; mkdir -p gpurun_out/ctprof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ctprof/sgd -- python bench.py --workload sgd_eval_contract --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ctprof/bk -- python tools/contract_beckmann_share.py > /dev/null 2>&1
for d in sgd bk; do echo "== $d"; cat gpurun_out/ctprof/$d/*/*kernel_stats.csv | cut -d, -f1-4 | cut -c1-110 | head -8; done
