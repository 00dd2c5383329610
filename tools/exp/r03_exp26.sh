#!/bin/bash
# round 3, GPU call 26: what bounds the tabular / sgd / abc / beckmann eval kernels (SQ counters over tools/kind_rates.py)
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; O=gpurun_out/r03/kind_pmc; rm -rf $O; mkdir -p $O
PYTHONPATH=. timeout 280 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/a -- python tools/kind_rates.py > $O/a.log 2>&1
PYTHONPATH=. timeout 280 rocprofv3 --pmc SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 --kernel-trace --output-format csv -d $O/b -- python tools/kind_rates.py > $O/b.log 2>&1
python - <<'PY'
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list)); meta={}
for f in glob.glob("gpurun_out/r03/kind_pmc/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"])); meta[r["Kernel_Name"][:70]]=(r["VGPR_Count"],r["LDS_Block_Size"],r["Grid_Size"])
for k,v in agg.items():
    if "k_eval" not in k: continue
    print(k, meta[k])
    print("   " + "  ".join("%s=%.3g" % (c.replace("SQ_",""), sum(x)/len(x)) for c,x in sorted(v.items())))
PY
