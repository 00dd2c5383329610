#!/bin/bash
# round 5: the two rows the MERL why-not table lacked.
#  (i)  write side: how large are the L2's write requests to the fabric?  TCC_EA0_WRREQ vs TCC_EA0_WRREQ_64B for k_merl_fast_v4 and,
#       as a reference point, for a pure streaming store (djb_gen_directions: 12 B per unit, nothing else)
#  (ii) stream pollution: the same kernel with its 36 B/pair streams in UNCACHED memory (hipExtMallocWithFlags) -- time, and the
#       memory-side read requests, from which the gathers' line fills follow (requests - 0.1875 stream reads per pair)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd "$R"; mkdir -p gpurun_out
O=$R/gpurun_out/merl_write_side; rm -rf $O; mkdir -p $O
rocprofv3 -L 2>/dev/null | grep -o "TCC_EA0_[A-Z0-9_]*" | sort -u | tr '\n' ' ' > $O/counters_available.txt
A="--workload merl_eval --n 250000000 --steps 4 --warmup 1 --no-cpu-baseline --no-secondary"
rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace --output-format csv -d $O/bench -- python bench.py $A > /dev/null 2> $O/bench.err
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum --kernel-trace --output-format csv -d $O/bench_l2 -- python bench.py $A > /dev/null 2>> $O/bench.err
for cfg in hipMalloc,hipMalloc uncached,uncached uncached,hipMalloc hipMalloc,uncached; do
  t=${cfg/,/_}
  DJB_MTYPE_ONLY=$cfg PYTHONPATH=. rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace --output-format csv -d $O/mt_$t -- python tools/merl_mtype_probe.py > $O/mt_$t.txt 2>&1
  DJB_MTYPE_ONLY=$cfg PYTHONPATH=. rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum --kernel-trace --output-format csv -d $O/mtl2_$t -- python tools/merl_mtype_probe.py > /dev/null 2>&1
done
python - <<PY > $R/gpurun_out/merl_write_side.txt
import csv, glob, collections, os
O = "$O"
print("counters on this box:", open(O + "/counters_available.txt").read().strip()[:600])
def table(d):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{O}/{d}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg
for d, n in [("bench", 2.5e8), ("bench_l2", 2.5e8)] + [(f"{p}_{t}", 5e8) for t in ("hipMalloc_hipMalloc", "uncached_uncached", "uncached_hipMalloc", "hipMalloc_uncached") for p in ("mt", "mtl2")]:
    for k, v in table(d).items():
        if "merl_fast" not in k and "gen_dir" not in k: continue
        row = {c: sum(x) / len(x) for c, x in v.items()}
        print(f"{d:28s} {k[:40]:40s} " + "  ".join(f"{c}={row[c]:.4g} ({row[c] / n:.4f}/unit)" for c in sorted(row)))
for f in sorted(glob.glob(O + "/mt_*.txt")):
    print(os.path.basename(f), [l for l in open(f).read().splitlines() if "ms per" in l])
PY
cat $R/gpurun_out/merl_write_side.txt
