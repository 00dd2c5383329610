#!/bin/bash
# instruction mix of the dominant kernel of a workload (run on the GPU box): tools/instmix.sh <workload>
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"
# optional second argument: a tag for the output directory (a DJB_LIB_PATH variant of the library measured next to the shipped one)
w=$1; O=$R/gpurun_out/mix/$w$2; rm -rf $O; mkdir -p $O
A="--workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-secondary"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SALU --kernel-trace --output-format csv -d $O/a -- python bench.py $A >/dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT --kernel-trace --output-format csv -d $O/b -- python bench.py $A >/dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_IOPS SQ_INSTS_BRANCH SQ_INSTS_VSKIPPED SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/c -- python bench.py $A >/dev/null 2>&1
python - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items():
    if "gen_dir" in k or "rocclr" in k or "convert" in k: continue
    print(k)
    for c,x in sorted(v.items()): print("   %-28s %.4g" % (c, sum(x)/len(x)))
PY
