#!/bin/bash
# round 5: the Beckmann sampler's last Newton trip shortened (bk_qf2_common / bk_sample_contract: erfinv(b3) only; that the reference leaves
# its loop there is SHOWN from the Newton step that led to b3 instead of being computed).
#  1. the decision against the value it stands in for (DJB_EXP_TRIP4_CHECK build: gpurun_variants/libdjb_t4chk.so)
#  2. A/B on one box: t4full (-DDJB_BK_TRIP4_FULL = the round-4 form) vs the shipped library, exact and contract mode, 1e9 samples
#  3. parity: the sampler tests (bit-identical to the oracle) and the contract tests
# builds: make -C dj_brdf_amd/csrc BUILD=build_t4full OUT=../../gpurun_variants/libdjb_t4full.so EXTRA=-DDJB_BK_TRIP4_FULL
#         make -C dj_brdf_amd/csrc BUILD=build_t4chk OUT=../../gpurun_variants/libdjb_t4chk.so EXTRA="-DDJB_EXP_RARE_COUNT -DDJB_EXP_TRIP4_CHECK"
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
O=gpurun_out/beckmann_trip4.txt; : > $O
echo "== 1. decision check (cumulative counters; 2.5e8 samples per lobe, bench directions)" >> $O
DJB_LIB_PATH=gpurun_variants/libdjb_t4chk.so PYTHONPATH=. timeout 900 python - >> $O 2>&1 <<'PY'
import torch
from dj_brdf_amd import djb, synth
ctx = djb.default_context(0)
b = djb.beckmann(ctx=ctx); P = djb.microfacet.params
n = 250_000_000
o = djb.gen_directions(n, synth.SEED_O, ctx=ctx)
for name, p in (("elliptic(0.2,0.5,0.7)", P.elliptic(0.2, 0.5, 0.7)), ("isotropic(0.3)", P.isotropic(0.3)), ("isotropic(1.0)", P.isotropic(1.0)),
                ("isotropic(0.02)", P.isotropic(0.02)), ("elliptic(0.05,0.8,0.3)", P.elliptic(0.05, 0.8, 0.3)),
                ("pdfparams(0.4,0.25,0.6,0.1,-0.2)", P.pdfparams(0.4, 0.25, 0.6, 0.1, -0.2)), ("pdfparams(3,5,-0.9,0,0)", P.pdfparams(3.0, 5.0, -0.9, 0.0, 0.0))):
    print(f"-- {name}", flush=True)
    keep = b.sample_rng(synth.SEED_U1 + 7, synth.SEED_U2 + 7, o, p); del keep
    torch.cuda.synchronize()
# grazing and near-normal views (the families of the contract selftest): directions squeezed towards the horizon / the normal
g = o.clone(); g[2] *= 0.02; g /= g.norm(dim=0, keepdim=True)
print("-- elliptic(0.2,0.5,0.7), grazing views", flush=True)
keep = b.sample_rng(3, 4, g, P.elliptic(0.2, 0.5, 0.7)); del keep; torch.cuda.synchronize()
g = o.clone(); g[:2] *= 0.02; g /= g.norm(dim=0, keepdim=True)
print("-- elliptic(0.2,0.5,0.7), near-normal views", flush=True)
keep = b.sample_rng(5, 6, g, P.elliptic(0.2, 0.5, 0.7)); del keep; torch.cuda.synchronize()
PY
echo "== 2. A/B, ms per 1e9 samples (bench.py, 10 steps)" >> $O
for rep in 1 2 3; do for v in t4full short; do
  lib=gpurun_variants/libdjb_$v.so; [ $v = short ] && lib=dj_brdf_amd/lib/libdjb_hip.so
  for w in beckmann_sample beckmann_sample_contract; do
    DJB_LIB_PATH=$lib timeout 300 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('%-7s %-26s %8.3f ms/step  frac %.3f' % ('$v', '$w', d['ms_per_step'], d['roofline']['frac']))" >> $O
  done; done; done
echo "== 3. parity" >> $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_contract.py tests/test_gpu_golden.py -m gpu -x -q -k "sample or beckmann or contract or hostile" 2>&1 | tail -4 >> $O
cat $O
