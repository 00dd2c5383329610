#!/bin/bash
# kernel_resources.sh <file.hip> [filter] [extra flags]: VGPRs / spills / scratch / occupancy / LDS of every kernel of one translation unit
# (hipcc -Rpass-analysis=kernel-resource-usage with the Makefile's flags; add e.g. "-mllvm -disable-machine-licm=false" as the third argument to see the
# hoisting MachineLICM would do), demangled, one line per kernel.  Build container; no GPU.
cd "$(dirname "$0")/../dj_brdf_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -fno-slp-vectorize -mllvm -disable-machine-licm $3 \
  -Rpass-analysis=kernel-resource-usage -c "$1" -o /tmp/kres.o 2>&1 | python3 -c "
import sys, re, subprocess
rows, cur = [], None
for l in sys.stdin:
    m = re.search(r'remark: (.*?) \[-Rpass', l)
    if not m:
        if 'error' in l: print(l.rstrip())
        continue
    t = m.group(1).strip()
    if t.startswith('Function Name:'):
        cur = {'name': t.split(':', 1)[1].strip()}; rows.append(cur)
    elif cur is not None and ':' in t:
        k, v = t.split(':', 1); cur[k.strip()] = v.strip()
names = subprocess.run(['c++filt'] + [r['name'] for r in rows], capture_output=True, text=True).stdout.splitlines()
for r, n in zip(rows, names):
    n = re.sub(r'\(anonymous namespace\)::', '', n); n = re.sub(r'\(.*', '', n).replace('void ', '')
    if '$2' and not re.search('$2', n): continue
    print(f\"{n:44s} VGPR {r.get('VGPRs','?'):>4s} AGPR {r.get('AGPRs','?'):>3s} spill {r.get('VGPR Spill','?'):>4s} scratch {r.get('ScratchSize [bytes/lane]','?'):>5s} occ {r.get('Occupancy [waves/SIMD]','?'):>2s} LDS {r.get('LDS Size [bytes/block]','?'):>6s}\")
"
