#!/bin/bash
# round 6, session 2: new GPU tests (keys, multi-context fits, full-size sweeps), order-lever leg, profile passes of the plugin-operator legs
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=$R/gpurun_out/r06; mkdir -p $O
python -m pytest tests -m gpu -x -q -k "bin_keys or multi_contexts or other_distributions or native_file or histogram" > $O/gputests2.log 2>&1; echo "pytest rc $?" >> $O/gputests2.log
tail -4 $O/gputests2.log
python bench.py --no-cpu-baseline > $O/bench_default2.json 2> $O/bench_default2.err; echo "bench rc $?"
WORKLOADS="tabular_eval_pdf tabular_sample tabular_abc_sample ggx_evalp_is beckmann_evalp_is lean_evalp_pdf abc_evalp tabular_aniso_eval_pdf tabular_aniso_sample" bash tools/exp/r06_profiles.sh
