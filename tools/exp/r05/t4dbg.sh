cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
DJB_LIB_PATH=gpurun_variants/libdjb_t4chk.so PYTHONPATH=. timeout 900 python - > gpurun_out/t4dbg.txt 2>&1 <<'PY'
import torch
from dj_brdf_amd import djb, synth
ctx = djb.default_context(0)
b = djb.beckmann(ctx=ctx); P = djb.microfacet.params
n = 250_000_000
o = djb.gen_directions(n, synth.SEED_O, ctx=ctx)
keep = b.sample_rng(synth.SEED_U1 + 7, synth.SEED_U2 + 7, o, P.pdfparams(0.4, 0.25, 0.6, 0.1, -0.2)); del keep
torch.cuda.synchronize()
PY
cat gpurun_out/t4dbg.txt
