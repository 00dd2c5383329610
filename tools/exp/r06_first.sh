#!/bin/bash
# round 6, session 1: GPU suite on the new on-chip generator, the default bench line, kernel names of the new plugin-operator legs
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=$R/gpurun_out/r06; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; echo "pytest rc $?" >> $O/gputests.log
tail -5 $O/gputests.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"
for w in tabular_eval_pdf tabular_sample tabular_abc_sample ggx_evalp_is beckmann_evalp_is lean_evalp_pdf abc_evalp tabular_aniso_eval_pdf tabular_aniso_sample fit_tabular_90 fit_aniso_90x90; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$w -- python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $O/bench_$w.json 2> $O/bench_$w.err
  f=$(ls $O/trace_$w/*/*_kernel_stats.csv | tail -1); cp $f $O/kernel_stats_$w.csv; rm -rf $O/trace_$w
done
head -c 1500 $O/bench_default.json
