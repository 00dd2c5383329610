#!/bin/bash
# round 4: why samples leave the contract-mode sampler's fast path (DJB_EXP_RARE_COUNT build: cumulative counts by cause after each launch)
# (needs the counting build: `make -C dj_brdf_amd/csrc BUILD=build_rc OUT=../../gpurun_variants/libdjb_rc.so EXTRA=-DDJB_EXP_RARE_COUNT`)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
DJB_LIB_PATH=gpurun_variants/libdjb_rc.so PYTHONPATH=. timeout 600 python - > gpurun_out/contract_sample_causes.txt 2>&1 <<'PY'
import torch
from dj_brdf_amd import djb, synth
ctx = djb.default_context(0)
b = djb.beckmann(ctx=ctx); P = djb.microfacet.params
n = 250_000_000
o = djb.gen_directions(n, synth.SEED_O, ctx=ctx)
for name, p in (("elliptic(0.2,0.5,0.7)", P.elliptic(0.2, 0.5, 0.7)), ("isotropic(0.3)", P.isotropic(0.3)), ("isotropic(1.0)", P.isotropic(1.0))):
    for on in (False, True):
        djb.set_contract_1e5(ctx, on)
        print(f"== {name} contract {on} (counts are cumulative over launches)", flush=True)
        keep = b.sample_rng(synth.SEED_U1, synth.SEED_U2, o, p); del keep
        torch.cuda.synchronize()
PY
cat gpurun_out/contract_sample_causes.txt
