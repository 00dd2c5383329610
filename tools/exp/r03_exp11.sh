#!/bin/bash
# round 3, GPU call 11: how many bytes does one L2 miss of a 12-byte table gather move?  (calibrates FETCH_SIZE for the gather part of the MERL kernel)
# every rocprofv3 invocation under its own timeout: an unknown counter makes it abort and then hang in its signal handler
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; O=gpurun_out/r03/gather_calib; rm -rf $O; mkdir -p $O
timeout 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/a -- tools/bin/stream_probe 250000000 g > $O/probe.txt 2>&1
timeout 150 rocprofv3 --pmc TCC_MISS_sum TCC_HIT_sum --kernel-trace --output-format csv -d $O/b -- tools/bin/stream_probe 250000000 g > /dev/null 2>&1
timeout 150 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace --output-format csv -d $O/c -- tools/bin/stream_probe 250000000 g > $O/c.txt 2>&1
python - <<'PY'
import csv,glob,collections
acc=collections.OrderedDict()
for d in ("a","b","c"):
    for f in glob.glob(f"gpurun_out/r03/gather_calib/{d}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "k_gather" not in r["Kernel_Name"]: continue
            key=int(r["Dispatch_Id"])
            acc.setdefault(key,{"k":r["Kernel_Name"][28:60]})[r["Counter_Name"]]=float(r["Counter_Value"])
for k,v in sorted(acc.items()):
    if k % 7 == 3: print(k, v)
PY
grep "^gather" $O/probe.txt | head -16
