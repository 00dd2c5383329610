#!/bin/bash
# round 4: (a) how much of the MERL tier-1 kernel's time is its arithmetic?  timing-only variants with part of the VALU work removed
# (-DDJB_EXP_MERL_NOGUARD: no guard bands / snap test; -DDJB_EXP_MERL_ATAN_CHEAP: the three atan2 replaced by 5 instructions), each
# with its SQ_INSTS_VALU count -> profiles/r04/merl_valu_sensitivity.txt;  (b) instruction mixes -> profiles/valu_<workload>.json
# (as run on the tree of commit 452eecc: the timing-only flags -DDJB_EXP_MERL_NOGUARD / -DDJB_EXP_MERL_ATAN_CHEAP lived in djb_device_tables.inc
#  until the tier was rewritten; variants built with `make -C dj_brdf_amd/csrc BUILD=build_x OUT=../../gpurun_variants/libdjb_x.so EXTRA=-D...`)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
O=gpurun_out/merl_valu_sensitivity.txt; : > $O
BASE=dj_brdf_amd/lib/libdjb_hip.so
run() {
  local line=$(DJB_LIB_PATH=$2 timeout 300 python bench.py --workload merl_eval --steps 10 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1)
  python - "$1" "$line" >> $O <<'PY'
import sys, json
try:
    d = json.loads(sys.argv[2]); print("%-6s %8.3f ms/step  launch %8.3f ms" % (sys.argv[1], d["ms_per_step"], d["roofline"]["launch_ms"]))
except Exception as e:
    print(sys.argv[1], "FAILED", e, sys.argv[2][:200])
PY
}
for rep in 1 2; do
  run base $BASE; for v in ng ac ngac; do run $v gpurun_variants/libdjb_$v.so; done
done
run base $BASE
bash tools/instmix.sh merl_eval > gpurun_out/instmix_merl_eval.txt 2>&1
python tools/valu_report.py merl_eval 1e9 >> $O
for v in ng ac ngac; do
  DJB_LIB_PATH=gpurun_variants/libdjb_$v.so bash tools/instmix.sh merl_eval _$v > gpurun_out/instmix_merl_eval_$v.txt 2>&1
  python tools/valu_report.py merl_eval 1e9 "round 4 timing-only variant $v" _$v >> $O
done
bash tools/instmix.sh beckmann_sample > gpurun_out/instmix_beckmann_sample.txt 2>&1
python tools/valu_report.py beckmann_sample 1e9 >> $O
bash tools/instmix.sh ggx_eval_pdf > gpurun_out/instmix_ggx_eval_pdf.txt 2>&1
python tools/valu_report.py ggx_eval_pdf 1e8 >> $O
bash tools/instmix.sh ggx_eval_pdf_contract > gpurun_out/instmix_ggx_eval_pdf_contract.txt 2>&1
python tools/valu_report.py ggx_eval_pdf_contract 1e8 >> $O
cat $O
