#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 passes over bench.py for every workload.
#   pass 1: --kernel-trace --stats            (per-kernel durations; must agree with bench.py's HIP events)
#   pass 2: --pmc FETCH_SIZE                  (TCC read requests;  gfx950: x2 for streaming reads)
#   pass 3: --pmc WRITE_SIZE                  (separate pass: FETCH_SIZE + WRITE_SIZE exceed the 4 TCC slots)
#   pass 4: --pmc SQ_* GRBM_GUI_ACTIVE        (VALU issue / busy / waits)
# Outputs land in gpurun_out/prof/<workload>/{trace,fetch,write,sq}; summarise with tools/summarize_profiles.py.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"
#   pass 5: --pmc TCC_HIT_sum TCC_MISS_sum    (L2 hit rate of the table gathers: the MERL legs)
for w in ${WORKLOADS:-merl_eval merl_eval_uniform_bins merl_eval_coherent ggx_eval_pdf ggx_eval_pdf_contract beckmann_sample beckmann_sample_contract utia_eval merl_fit}; do
  O=$R/gpurun_out/prof/$w; rm -rf $O; mkdir -p $O
  A="--workload $w --steps 5 --warmup 1 --no-cpu-baseline --no-secondary"
  case $w in merl_eval_*) A="$A --n 250000000";; esac
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python bench.py $A > $O/bench_trace.json 2> $O/trace.err
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python bench.py $A > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python bench.py $A > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE \
            --kernel-trace --output-format csv -d $O/sq -- python bench.py $A > /dev/null 2>&1
  case $w in merl_eval*) rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/l2 -- python bench.py $A > /dev/null 2>&1;; esac
  python bench.py $A --steps 10 --warmup 2 > $O/bench_plain.json 2> $O/bench_plain.err
done
ls $R/gpurun_out/prof
