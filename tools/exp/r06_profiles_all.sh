#!/bin/bash
# round 6: the whole GPU suite, then the profile passes + instruction mixes of EVERY workload bench.py reports (final kernels)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; O=$R/gpurun_out/r06; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/gputests_all.log 2>&1; echo "pytest rc $?" >> $O/gputests_all.log
tail -4 $O/gputests_all.log
WORKLOADS="${ALL:-merl_eval merl_eval_uniform_bins merl_eval_coherent ggx_eval_pdf ggx_eval_pdf_contract ggx_unpolarized_eval_pdf ggx_unpolarized_eval_pdf_contract sgd_eval sgd_eval_contract beckmann_sample beckmann_sample_contract utia_eval utia_eval_contract merl_fit tabular_eval_pdf tabular_sample tabular_abc_sample ggx_evalp_is beckmann_evalp_is lean_evalp_pdf abc_evalp tabular_aniso_eval_pdf tabular_aniso_sample fit_tabular_90 fit_aniso_90x90}" bash tools/exp/r06_profiles.sh
