"""Verification holes named by the round-1 review, closed on the GPU (all through the C ABI):

* the widest batch fit -- all 100 materials of BASELINE configs[4] in ONE launch, with the default helper
  slicing (`fit_parts` = 2 on a 256-CU device: 200 workgroups with a bounded-wait hand-off) and with eight slices
  forced (800 workgroups on 256 CUs: helpers that are not resident when their partner looks) -- against the CPU
  oracle run on host threads: alphas bit for bit and the params.txt bytes;
* the two-tier MERL lookup on every adversarial input family of tools/calibrate_merl_guard.py, against the
  operation-by-operation kernel, with the guard-band margin asserted (djb_merl_guard_stats);
* a two-process run of the real HIP path (both ranks may share device 0): pair-range eval and round-robin fit
  reassembled and compared with the unsharded result.
"""
import os
import socket
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from dj_brdf_amd import djb, merl_params, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------------------------------------- (a) 100-material fit
def test_fit_all_100_materials_one_launch(gpu_ctx, oracle, monkeypatch):
    n_mat = 100
    mats, futures = [], []
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 4)

    def oracle_fit(tab):
        t = oracle.tabular(oracle.merl_from_table(tab), 90, True)
        w = oracle.tabular_tables(t)
        return np.float32(w["alpha_beckmann"]), np.float32(w["alpha_ggx"])

    with ThreadPoolExecutor(max_workers=max(2, min(32, cores))) as pool:     # ctypes releases the GIL
        for k in range(n_mat):
            tab = synth.merl_table(*synth.material_recipe(k))
            mats.append(djb.merl.from_table(tab, ctx=gpu_ctx))
            futures.append(pool.submit(oracle_fit, tab))
            del tab
        want = np.array([f.result() for f in futures], np.float32)          # [100, 2]
    monkeypatch.delenv("DJB_FIT_PARTS", raising=False)
    ab, ag = djb.fit_brdf_batch(mats, 90, True, ctx=gpu_ctx)                 # default slicing (2 per material on 256 CUs)
    got = np.stack([ab, ag], 1)
    bad = np.nonzero((got.view(np.uint32) != want.view(np.uint32)).any(axis=1))[0]
    assert bad.size == 0, f"materials {bad[:8].tolist()}: fitted alphas differ from the oracle's: {got[bad[:4]]} vs {want[bad[:4]]}"
    names = [f"/x/{synth.MERL_NAMES[k]}.binary" for k in range(n_mat)]
    txt = merl_params.format_params_txt(names, [(float(a), float(b)) for a, b in got])
    assert txt.encode() == merl_params.format_params_txt(names, [(float(a), float(b)) for a, b in want]).encode()
    assert txt.count("\n") == n_mat + 1
    for parts in ("8", "1", "3"):       # 800 workgroups (late helpers: the bounded wait must fall back), none, odd
        monkeypatch.setenv("DJB_FIT_PARTS", parts)
        ab2, ag2 = djb.fit_brdf_batch(mats, 90, True, ctx=gpu_ctx)
        assert np.array_equal(ab2.view(np.uint32), ab.view(np.uint32)) and np.array_equal(ag2.view(np.uint32), ag.view(np.uint32)), parts
    # a few complete table sets of the batch against the oracle (first, last, middle)
    monkeypatch.delenv("DJB_FIT_PARTS", raising=False)
    lib_tabs = djb.fit_merl_batch([m.get_samples() for m in (mats[0], mats[57], mats[99])], 90, True, ctx=gpu_ctx, return_tables=True)[2]
    for j, k in enumerate((0, 57, 99)):
        w = oracle.tabular_tables(oracle.tabular(oracle.merl_from_table(synth.merl_table(*synth.material_recipe(k))), 90, True))
        for name in ("p22", "sigma", "cdf", "fresnel"):
            a, b = np.asarray(lib_tabs[name][j], np.float32), np.asarray(w[name], np.float32).reshape(lib_tabs[name][j].shape)
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (k, name)


# ------------------------------------------------------------------------------------------- (b) MERL guard families
def _merl_families(m, dev):
    """(name, i, o) device tensors [3, m]: the adversarial families of tools/calibrate_merl_guard.py"""
    import torch
    g = torch.Generator(device=dev); g.manual_seed(1234)
    R = lambda *shape: torch.rand(*shape, device=dev, generator=g)
    unit = lambda v: (v / v.norm(dim=0, keepdim=True)).contiguous()
    ones, zeros = torch.ones(m, device=dev), torch.zeros(m, device=dev)

    def on_cone(axis, theta_deg, phi):
        t = torch.deg2rad(theta_deg)
        ref = torch.zeros_like(axis); ref[0] = 1.0
        ref = torch.where((axis[0].abs() > 0.9).unsqueeze(0), torch.stack([zeros, ones, zeros]), ref)
        e1 = unit(torch.linalg.cross(axis, ref, dim=0)); e2 = torch.linalg.cross(axis, e1, dim=0)
        return unit(torch.cos(t) * axis + torch.sin(t) * (torch.cos(phi) * e1 + torch.sin(phi) * e2))

    base = djb.gen_directions(m, 99)
    yield "bench_distribution", djb.gen_directions(m, synth.SEED_I), djb.gen_directions(m, synth.SEED_O)
    yield "backscatter", base, unit(base + 1e-3 * djb.gen_directions(m, 5))
    yield "backscatter_tiny", base, unit(base + 1e-5 * djb.gen_directions(m, 6))
    yield "mirror", base, unit(torch.stack([-base[0], -base[1], base[2]]) + 1e-3 * djb.gen_directions(m, 7))
    yield "grazing", base, unit(torch.stack([base[0], base[1], 1e-3 * base[2]]))
    yield "identical", base, base.clone()
    yield "near_normal", unit(torch.stack([1e-2 * (R(m) - .5), 1e-2 * (R(m) - .5), ones])), \
        unit(torch.stack([3e-2 * (R(m) - .5), 3e-2 * (R(m) - .5), ones]))
    sph = lambda: unit(torch.randn(3, m, device=dev, generator=g))
    yield "full_sphere", sph(), sph()
    yield "lengths_0.5_to_2", (base * (0.5 + 1.5 * R(m))).contiguous(), (djb.gen_directions(m, 11) * (0.5 + 1.5 * R(m))).contiguous()
    h = djb.gen_directions(m, 21)
    td = torch.randint(1, 89, (m,), device=dev).float() + (R(m) - .5) * 2e-5
    ii = on_cone(h, td, 2 * torch.pi * R(m))
    yield "theta_d_on_bin_edges", ii, unit(2 * (ii * h).sum(0, keepdim=True) * h - ii)
    k = torch.randint(1, 89, (m,), device=dev).float()
    th = k * k / 90.0 + (R(m) - .5) * 2e-5
    hh = on_cone(torch.stack([zeros, zeros, ones]), th, 2 * torch.pi * R(m))
    ii = on_cone(hh, 5 + 70 * R(m), 2 * torch.pi * R(m))
    yield "theta_h_on_bin_edges", ii, unit(2 * (ii * hh).sum(0, keepdim=True) * hh - ii)
    pd = torch.randint(0, 180, (m,), device=dev).float() + (R(m) - .5) * 2e-5           # phi_d on whole degrees
    hh = djb.gen_directions(m, 23)
    ii = on_cone(hh, 5 + 70 * R(m), torch.deg2rad(pd))
    yield "phi_d_near_bin_edges", ii, unit(2 * (ii * hh).sum(0, keepdim=True) * hh - ii)


def test_merl_two_tier_on_adversarial_families(gpu_ctx):
    """Tier 1 (fp32 estimates + guard bands) must agree with the operation-by-operation fp64 kernel on every pair
    it calls certain -- on the bench distribution AND on inputs built to sit on bin edges, at the poles, below the
    horizon, un-normalised.  Asserted per family: 0 index mismatches among certain pairs, composite == exact kernel
    bit for bit, and the measured |estimate - reference| stays below half of the guard band (max_ratio < 0.5)."""
    import torch
    m = 20_000_000
    tab = synth.merl_table_hashed()
    mobj = djb.merl.from_table(tab, ctx=gpu_ctx)
    report = []
    for name, i, o in _merl_families(m, f"cuda:{gpu_ctx.device}"):
        s = djb.merl_guard_stats(i, o, ctx=gpu_ctx)
        a = mobj.eval(i, o)
        # ... and once more with the worklist knob set: until round 4 that forced tier 2's overflow rescan (a second kernel re-taking
        # tier 1's decision pair by pair); the look-up now drains its ambiguous pairs from per-wave LDS queues inside the same kernel,
        # no capacity is involved any more, and this second evaluation checks that the result does not depend on the knob or the run
        djb.set_test_worklist_cap(gpu_ctx, 64)
        try:
            a_rescan = mobj.eval(i, o)
        finally:
            djb.set_test_worklist_cap(gpu_ctx, -1)
        djb.set_merl_exact_only(gpu_ctx, True)
        try:
            b = mobj.eval(i, o)
        finally:
            djb.set_merl_exact_only(gpu_ctx, False)
        # NaN inputs (un-normalisable) produce NaN-free table values either way; compare bits
        diff = int((a.view(torch.int32) != b.view(torch.int32)).any(dim=0).sum())
        report.append((name, s, diff))
        assert s["mismatch"] == 0, (name, s)
        assert diff == 0, f"{name}: {diff} of {m} two-tier results differ from the exact kernel ({s})"
        diff_rescan = int((a_rescan.view(torch.int32) != b.view(torch.int32)).any(dim=0).sum())
        assert diff_rescan == 0, f"{name}: {diff_rescan} of {m} results differ from the exact kernel when the worklist overflows"
        del a_rescan
        assert max(s["max_ratio"]) < 0.5, f"{name}: guard-band margin below 2x: {s}"
        assert s["special"] + s["ambiguous"] + s["certain"] == m
        del i, o, a, b
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "merl_guard_families.txt"), "w") as f:
            for name, s, diff in report:
                f.write(f"{name}: {s} two_tier_vs_exact_mismatches={diff}\n")


def test_merl_guard_attack(gpu_ctx):
    """Directed search instead of sampling (tools/merl_guard_attack.py, shorter): candidates from every family hill-climb
    over the bit patterns of their inputs to maximise |tier-1 estimate - reference| / guard band.  The two-tier kernel is
    exact while that ratio stays below 1; asserted: < 0.5 under attack, and 0 index mismatches among the pairs tier 1
    called certain over everything the search visited (~2e8 evaluations here, 4.8e9 in profiles/r03)."""
    m = 1 << 15
    worst, evals = 0.0, 0
    for name, i, o in _merl_families(m, f"cuda:{gpu_ctx.device}"):
        i, o = i.clone().contiguous(), o.clone().contiguous()
        best, c = djb.merl_guard_attack(i, o, iters=512, seed=5, ctx=gpu_ctx)
        assert c["mismatch"] == 0, (name, c)
        assert float(best.max()) < 0.5, f"{name}: attacked ratio {float(best.max()):.3f}"
        worst, evals = max(worst, float(best.max())), evals + c["evaluations"]
    print(f"\nmerl guard attack: {evals:.2e} directed evaluations, worst ratio {worst:.3f}")


# ------------------------------------------------------------------------------------------- (2) two ranks, real HIP path
def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _rank_main(rank, world, port, n_pairs, n_mat, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from dj_brdf_amd import djb, shard, synth
    dist.init_process_group("gloo", rank=rank, world_size=world)     # control plane only: no data-path collective exists
    dev = rank % max(1, djb.device_count())                            # ranks share device 0 when one GPU is visible
    torch.cuda.set_device(dev)
    ctx = djb.Context(dev)
    # pair-range eval: MERL two-tier lookup + GGX eval on this rank's block, inputs generated on the device
    lo, hi = shard.block_range(n_pairs, world, rank)
    i = djb.gen_directions(hi - lo, synth.SEED_I, start=lo, ctx=ctx)
    o = djb.gen_directions(hi - lo, synth.SEED_O, start=lo, ctx=ctx)
    m = djb.merl.from_table(synth.merl_table_hashed(), ctx=ctx)
    g = djb.ggx(djb.fresnel.schlick((1.0, 0.71, 0.29)), True, ctx=ctx)
    part = torch.cat([m.eval(i, o), g.eval(i, o, djb.microfacet.params.isotropic(0.3))]).cpu().numpy()
    # round-robin fit: this rank's materials in one launch
    mine = shard.round_robin(n_mat, world, rank)
    mats = [djb.merl.from_table(synth.merl_table(*synth.material_recipe(k)), ctx=ctx) for k in mine]
    ab, ag = djb.fit_brdf_batch(mats, 90, True, ctx=ctx)
    rows = shard.gather_rows([(k, (float(a), float(b))) for k, a, b in zip(mine, ab, ag)], world, rank, n_mat)
    parts = [None] * world
    dist.all_gather_object(parts, (lo, part))
    q.put((rank, [p for _, p in sorted(parts, key=lambda x: x[0])] if rank == 0 else None, rows))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_ranks_run_the_hip_path(gpu_ctx, oracle):
    import torch
    import torch.multiprocessing as mp
    world, n_pairs, n_mat = 2, 1_000_003, 7
    port = _free_port()
    mpctx = mp.get_context("spawn")
    q = mpctx.Queue()
    procs = [mpctx.Process(target=_rank_main, args=(r, world, port, n_pairs, n_mat, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {r: (parts, rows) for r, parts, rows in (q.get(timeout=500) for _ in range(world))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # unsharded, in this process
    i = djb.gen_directions(n_pairs, synth.SEED_I, ctx=gpu_ctx); o = djb.gen_directions(n_pairs, synth.SEED_O, ctx=gpu_ctx)
    m = djb.merl.from_table(synth.merl_table_hashed(), ctx=gpu_ctx)
    g = djb.ggx(djb.fresnel.schlick((1.0, 0.71, 0.29)), True, ctx=gpu_ctx)
    want = torch.cat([m.eval(i, o), g.eval(i, o, djb.microfacet.params.isotropic(0.3))]).cpu().numpy()
    got = np.concatenate(res[0][0], axis=1)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), "sharded pair ranges != unsharded batch"
    # ... and the sharded result against the ORACLE (not only against another HIP run): every 97th pair
    sub = np.arange(0, n_pairs, 97)
    ih = np.ascontiguousarray(i.cpu().numpy().T[sub]); oh = np.ascontiguousarray(o.cpu().numpy().T[sub])
    om = oracle.merl_from_table(synth.merl_table_hashed())
    og = oracle.microfacet("ggx", ("schlick", 1.0, 0.71, 0.29), True)
    want_o = np.concatenate([oracle.eval(om, ih, oh).T, oracle.eval(og, ih, oh, ("elliptic", 0.3, 0.3, 0.0)).T])
    assert np.array_equal(np.ascontiguousarray(got[:, sub]).view(np.uint32), np.ascontiguousarray(want_o).view(np.uint32)), "sharded HIP result != oracle"
    mats = [djb.merl.from_table(synth.merl_table(*synth.material_recipe(k)), ctx=gpu_ctx) for k in range(n_mat)]
    ab, ag = djb.fit_brdf_batch(mats, 90, True, ctx=gpu_ctx)
    for r in range(world):
        assert res[r][1] == [(float(a), float(b)) for a, b in zip(ab, ag)], f"rank {r}: sharded fit rows != unsharded"


# ------------------------------------------------------------------------------------------- torch stream following
def test_context_follows_torch_current_stream(gpu_ctx):
    """Device-tensor calls run on torch's CURRENT stream (read per call), so outputs that torch allocated under
    `with torch.cuda.stream(s)` are written on s, where their consumers are ordered (the stream used to
    be captured once at construction)."""
    import ctypes as C
    import torch
    from dj_brdf_amd import _lib
    lib = _lib.load()
    n = 4_000_000
    g = djb.ggx(djb.fresnel.schlick((1.0, 0.71, 0.29)), True, ctx=gpu_ctx)
    p = djb.microfacet.params.isotropic(0.3)
    i = djb.gen_directions(n, synth.SEED_I, ctx=gpu_ctx); o = djb.gen_directions(n, synth.SEED_O, ctx=gpu_ctx)
    want = g.eval(i, o, p)
    torch.cuda.synchronize()
    base = lib.djb_ctx_stream(gpu_ctx._h)
    s2 = torch.cuda.Stream()
    with torch.cuda.stream(s2):
        got = g.eval(i, o, p)
        assert (lib.djb_ctx_stream(gpu_ctx._h) or 0) == s2.cuda_stream != (base or 0)
        total = got.sum()                       # consumer on s2: ordered after the kernel only if it ran on s2
    s2.synchronize()
    assert torch.equal(got, want) and torch.isfinite(total)
    again = g.eval(i, o, p)                     # back on the original stream
    assert (lib.djb_ctx_stream(gpu_ctx._h) or 0) == (base or 0)
    torch.cuda.synchronize()
    assert torch.equal(again, want)


# ---- fp64 trig family: every float -> float site, GPU (ocml) against the library's host instantiation (glibc)
# tools/exhaustive_trig.py sweeps all 2^32 inputs per site (profiles/r02/exhaustive_trig.json: 0 differences); the
# suite repeats 2^26 inputs per site over the ranges the path feeds them and proves the comparison can fail.
_TRIG_RANGES = [
    (0x3C000000, 1 << 24),   # 2^-7 .. : small angles / cosines
    (0x3F000000, 1 << 24),   # 0.5 .. 2 (covers the acos pole at 1, theta up to pi/2 ...)
    (0x40000000, 1 << 24),   # 2 .. 8 (phi up to 2 pi, tan poles)
    (0xBF000000, 1 << 24),   # -0.5 .. -2
    (0x41000000, 3 << 24),   # 8 .. 512: utia's degrees (grid cells of 15 / 7.5 degrees)
]


@pytest.mark.parametrize("site", djb.TRIG_SITES)
def test_trig_float_sites_match_glibc(gpu_ctx, site):
    for first, count in _TRIG_RANGES:
        n_bad, rows = djb.selftest_trig_sweep(site, first, count, ctx=gpu_ctx)
        assert n_bad == 0, (site, hex(first), rows[:4])


def test_trig_sweep_negative_control(gpu_ctx):
    # 2 acos(c) / pi with pi as a double against pi rounded to float: different float results on many inputs
    n_bad, rows = djb.selftest_trig_sweep("acos_u", 0x3F000000, 1 << 20, host_fn="acos_u32", ctx=gpu_ctx)
    assert n_bad > 1000 and len(rows) == 64
    x, d, h = rows[0]
    assert d != h and 0x3F000000 <= x < 0x3F000000 + (1 << 20)
    with pytest.raises(djb.exc):
        djb.selftest_trig_sweep("cos", 0x3F000000, 16, host_fn="cos_d", ctx=gpu_ctx)


def test_utia_two_tier_equals_the_one_kernel_form(gpu_ctx):
    """utia eval / evalp batches run two tiers (tier 1 without glibc's atan2 behind the azimuths, a worklist of the pairs
    next to a float rounding boundary, tier 2 = the exact form); DJB_OPT_UTIA_EXACT_ONLY runs the one-kernel form.  Same
    bits on 5e7 pairs, eval and evalp + pdf, dense and strided views."""
    import torch
    n = 50_000_000
    i = djb.gen_directions(n, synth.SEED_I, ctx=gpu_ctx); o = djb.gen_directions(n, synth.SEED_O, ctx=gpu_ctx)
    u = djb.utia.from_table(synth.utia_table_smooth(), ctx=gpu_ctx)
    try:
        for op in ("eval", "evalp"):
            djb.set_utia_exact_only(gpu_ctx, False)
            two = getattr(u, op)(i, o)
            djb.set_utia_exact_only(gpu_ctx, True)
            one = getattr(u, op)(i, o)
            assert torch.equal(two.view(torch.int32), one.view(torch.int32)), op
            assert float(two.abs().sum()) > 0
        # strided (array-of-vec3) views take the non-dense instantiation
        m = 1_000_000
        ia = i[:, :m].t().contiguous(); oa = o[:, :m].t().contiguous()
        djb.set_utia_exact_only(gpu_ctx, False)
        two = u.eval(ia, oa)
        djb.set_utia_exact_only(gpu_ctx, True)
        one = u.eval(ia, oa)
        assert torch.equal(two.view(torch.int32), one.view(torch.int32))
    finally:
        djb.set_utia_exact_only(gpu_ctx, False)


def test_utia_worklist_overflow_redoes_the_batch(gpu_ctx, monkeypatch):
    """A worklist too small for the undecided pairs (capacity forced to 0) makes tier 2 redo the whole batch: same bits."""
    import torch
    n = 20_000_000
    i = djb.gen_directions(n, synth.SEED_I, ctx=gpu_ctx); o = djb.gen_directions(n, synth.SEED_O, ctx=gpu_ctx)
    u = djb.utia.from_table(synth.utia_table_smooth(), ctx=gpu_ctx)
    want = u.eval(i, o)
    djb.set_test_worklist_cap(gpu_ctx, 0)
    try:
        got = u.eval(i, o)
    finally:
        djb.set_test_worklist_cap(gpu_ctx, -1)
    assert torch.equal(got.view(torch.int32), want.view(torch.int32))
    try:
        djb.set_utia_exact_only(gpu_ctx, True)
        one = u.eval(i, o)
    finally:
        djb.set_utia_exact_only(gpu_ctx, False)
    assert torch.equal(one.view(torch.int32), want.view(torch.int32))


def test_two_tier_overflow_and_in_place_calls(gpu_ctx):
    """(1) Contract-mode worklist forced to overflow: tier 2 redoes the batch, same results (the MERL look-up has no worklist
    since round 4 -- its tier 2 runs in-kernel -- and must simply not notice the knob).  (2) In-place
    evalp (the output arrays ARE the input arrays of i): the two-tier kernels must not be used -- their second tier
    re-reads inputs the first has overwritten -- and the result must equal the out-of-place call bit for bit."""
    import ctypes as C
    import torch
    n = 3_000_001
    i = djb.gen_directions(n, synth.SEED_I, ctx=gpu_ctx); o = djb.gen_directions(n, synth.SEED_O, ctx=gpu_ctx)
    m = djb.merl.from_table(synth.merl_table_hashed(), ctx=gpu_ctx)
    u = djb.utia.from_table(synth.utia_table_smooth(), ctx=gpu_ctx)
    g = djb.ggx(djb.fresnel.schlick((1.0, 0.71, 0.29)), True, ctx=gpu_ctx)
    p = djb.microfacet.params.isotropic(0.3)
    want_m, want_u = m.evalp(i, o), u.evalp(i, o)
    djb.set_contract_1e5(gpu_ctx, True)
    try:
        want_g = g.evalp(i, o, p)
        djb.set_test_worklist_cap(gpu_ctx, 3)
        try:
            assert torch.equal(m.evalp(i, o).view(torch.int32), want_m.view(torch.int32)), "merl: worklist overflow changed the result"
            assert torch.equal(g.evalp(i, o, p).view(torch.int32), want_g.view(torch.int32)), "contract mode: worklist overflow changed the result"
        finally:
            djb.set_test_worklist_cap(gpu_ctx, -1)
        lib = djb._lib.load()
        for obj, want, par in ((m, want_m, None), (u, want_u, None), (g, None, p)):
            ii = i.clone()
            vi, vo = djb._Vec(ii), djb._Vec(o)
            djb._lib.check(lib.djb_evalp_batch(gpu_ctx._h, obj._h, C.c_int64(n), C.byref(vi.view), C.byref(vo.view),
                                               C.byref(par._p) if par is not None else None, C.byref(vi.view), C.c_int(0)))
            gpu_ctx.synchronize()
            if want is None:          # contract mode must have stepped aside: the in-place call is the bit-exact kernel's
                djb.set_contract_1e5(gpu_ctx, False)
                want = g.evalp(i, o, p)
                djb.set_contract_1e5(gpu_ctx, True)
            assert torch.equal(ii.view(torch.int32), want.view(torch.int32)), f"in-place evalp differs ({obj.__class__.__name__})"
    finally:
        djb.set_contract_1e5(gpu_ctx, False)
    # (3) in-place sample (the sampled directions overwrite o): the Beckmann kernel defers some samples to a later dense pass
    # (djb_kernels_sample.hip) -- their inputs must come from its queue, not from the overwritten array
    bk = djb.beckmann(djb.fresnel.ideal(), True, ctx=gpu_ctx)
    pe = djb.microfacet.params.elliptic(0.2, 0.5, 0.7)
    u1 = djb.gen_uniforms(n, synth.SEED_U1, ctx=gpu_ctx); u2 = djb.gen_uniforms(n, synth.SEED_U2, ctx=gpu_ctx)
    want_s = bk.sample(u1, u2, o, pe)
    oo = o.clone()
    vo = djb._Vec(oo)
    djb._lib.check(lib.djb_sample_batch(gpu_ctx._h, bk._h, C.c_int64(n), C.c_void_p(u1.data_ptr()), C.c_void_p(u2.data_ptr()),
                                        C.byref(vo.view), C.byref(pe._p), C.byref(vo.view), C.c_int(0)))
    gpu_ctx.synchronize()
    assert torch.equal(oo.view(torch.int32), want_s.view(torch.int32)), "in-place Beckmann sample differs"
    # (4) in-place evalp of a SHARP Beckmann lobe, over i and over o: k_eval_bk_sharp stores placeholders for the pairs it queues and
    # evaluates them later -- from its queue, not from the arrays it has already written
    ps = djb.microfacet.params.isotropic(0.05)
    want_e = bk.evalp(i, o, ps)
    for over_i in (True, False):
        ii, oo = i.clone(), o.clone()
        vi, vo = djb._Vec(ii), djb._Vec(oo)
        dst = vi if over_i else vo
        djb._lib.check(lib.djb_evalp_batch(gpu_ctx._h, bk._h, C.c_int64(n), C.byref(vi.view), C.byref(vo.view), C.byref(ps._p), C.byref(dst.view), C.c_int(0)))
        gpu_ctx.synchronize()
        assert torch.equal((ii if over_i else oo).view(torch.int32), want_e.view(torch.int32)), f"in-place sharp-lobe evalp differs (over {'i' if over_i else 'o'})"


def test_merl_two_tier_every_output_set_and_launch_shape(gpu_ctx):
    """The fused two-tier look-up against the operation-by-operation kernel for every output set the operator surface asks for
    (eval, evalp, eval + pdf, evalp + pdf) and every launch shape: dense SoA above 2^18 pairs (four pairs per lane + the < 4-pair
    tail), dense SoA below (one pair per lane), an array of vec3 (strided).  Inputs: bench directions mixed with pairs that sit on
    phi_d bin edges, so that the in-kernel drain of ambiguous pairs runs in every shape.  Bit for bit; pdf = float(double(i.z) / pi)."""
    import numpy as np
    import torch
    dev = f"cuda:{gpu_ctx.device}"
    tab = synth.merl_table_hashed()
    m = djb.merl.from_table(tab, ctx=gpu_ctx)
    for n in ((1 << 20) + 3, (1 << 16) + 1):
        i = djb.gen_directions(n, synth.SEED_I, ctx=gpu_ctx); o = djb.gen_directions(n, synth.SEED_O, ctx=gpu_ctx)
        fam = dict((name, (a, b)) for name, a, b in _merl_families(n // 2, dev))
        ei, eo = fam["phi_d_near_bin_edges"]
        i[:, : n // 2] = ei; o[:, : n // 2] = eo              # half of the batch is adversarial
        for layout in ("soa", "aos"):
            ii, oo = (i, o) if layout == "soa" else (i.t().contiguous(), o.t().contiguous())
            got = {"eval": m.eval(ii, oo), "evalp": m.evalp(ii, oo), "eval_pdf": m.eval_pdf(ii, oo), "evalp_pdf": m.eval_pdf(ii, oo, cos=True)}
            djb.set_merl_exact_only(gpu_ctx, True)
            try:
                want = {"eval": m.eval(ii, oo), "evalp": m.evalp(ii, oo), "eval_pdf": m.eval_pdf(ii, oo), "evalp_pdf": m.eval_pdf(ii, oo, cos=True)}
            finally:
                djb.set_merl_exact_only(gpu_ctx, False)
            bits = lambda t: t.contiguous().view(torch.int32)
            for k in ("eval", "evalp"):
                assert torch.equal(bits(got[k]), bits(want[k])), (n, layout, k)
            for k in ("eval_pdf", "evalp_pdf"):
                assert torch.equal(bits(got[k][0]), bits(want[k][0])) and torch.equal(bits(got[k][1]), bits(want[k][1])), (n, layout, k)
            iz = i[2].double().cpu().numpy()
            pdf = (iz / np.pi).astype(np.float32)
            assert np.array_equal(got["eval_pdf"][1].cpu().numpy().view(np.uint32), pdf.view(np.uint32)), (n, layout, "pdf")
            assert torch.equal(bits(got["eval_pdf"][0]), bits(got["eval"])) and torch.equal(bits(got["evalp_pdf"][0]), bits(got["evalp"]))


def test_fast_trig_sites_against_their_previous_forms(gpu_ctx):
    """The table-driven kinds take their trig sites from one branch-free fp64 arctangent core and keep a value only where it is decided
    (further from a float rounding boundary than the core's error); otherwise the previous form answers.  utia's polar angle for every
    float cosine in [1e-17, 1] against acos_deg_f (identical to the host's by exhaustion); its azimuth and xyz_to_theta_phi's for 2^33
    generated (y, x) pairs against atan2_to_f32 (glibc's atan2 behind a guard); the six one-argument sites over ALL 2^32 floats against the
    forms the exhaustive sweeps (tools/exhaustive_trig.py) were run on: not one different float."""
    first = int(np.float32(1e-17).view(np.uint32))
    n = 0x3F800000 - first + 1                                                      # every float in [1e-17, 1]
    r = djb.selftest_fast_trig(n, 0, first=first, ctx=gpu_ctx)
    print("utia acos_deg:", r)
    assert r["mismatch"] == 0 and r["decided"] > 0.9999 * n and r["worst_ulp64"] < 256          # the guard is 4096
    for mode in (1, 8):
        tot = {"decided": 0, "mismatch": 0, "undecided": 0, "worst_ulp64": 0}
        for seed in range(1, 5):
            r = djb.selftest_fast_trig(1 << 31, mode, seed=seed, ctx=gpu_ctx)
            for k in ("decided", "mismatch", "undecided"): tot[k] += r[k]
            tot["worst_ulp64"] = max(tot["worst_ulp64"], r["worst_ulp64"])
        print("atan2 (scale %s):" % ("r2d" if mode == 1 else "1"), tot)
        assert tot["mismatch"] == 0 and tot["decided"] > 0.8 * (1 << 33) and tot["worst_ulp64"] < 256
    for mode, name in ((2, "acos_f"), (3, "acos_u_f"), (4, "acos_u32_f"), (5, "atan_u_f"), (6, "atan_squ_f"), (7, "atan_sqrt_f"), (9, "tan_f")):
        tot = {"decided": 0, "mismatch": 0, "undecided": 0}
        for half in (0, 1):                                                         # all 2^32 bit patterns, two launches
            r = djb.selftest_fast_trig(1 << 31, mode, first=half << 31, ctx=gpu_ctx)
            for k in tot: tot[k] += r[k]
        print("%-12s" % name, tot)
        assert tot["mismatch"] == 0 and tot["decided"] + tot["undecided"] == 1 << 32 and tot["decided"] > 0.2 * (1 << 32)   # (the rest: |x| > 1, NaN, ...)


def test_utia_two_tier_on_ragged_batch_sizes(gpu_ctx):
    """k_utia_v2 fetches its records wave-cooperatively (every lane of a wave takes part, also the ones past the end of the batch): device
    batches of 1 .. 100003 pairs, eval and evalp, against the one-kernel form; under DJB_OPT_CONTRACT_1E5 within the contract; the fused
    eval + pdf call gives eval's bits."""
    import torch
    u = djb.utia.from_table(np.random.default_rng(3).uniform(-5, 120, 3 * 288 * 288), ctx=gpu_ctx)
    try:
        for n in (1, 2, 63, 64, 65, 255, 256, 257, 1000, 4097, 100003):
            i = djb.gen_directions(n, 7, ctx=gpu_ctx); o = djb.gen_directions(n, 8, ctx=gpu_ctx)
            for op in ("eval", "evalp"):
                djb.set_utia_exact_only(gpu_ctx, True)
                one = getattr(u, op)(i, o)
                djb.set_utia_exact_only(gpu_ctx, False)
                two = getattr(u, op)(i, o)
                assert torch.equal(two.view(torch.int32), one.view(torch.int32)), (n, op)
                djb.set_contract_1e5(gpu_ctx, True)
                fast = getattr(u, op)(i, o)
                djb.set_contract_1e5(gpu_ctx, False)
                assert bool(((fast == 0) == (one == 0)).all()), (n, op)
                rel = ((fast - one).abs() / one.abs().clamp_min(1e-30)).masked_fill(one == 0, 0.0)
                assert float(rel.max()) <= 1e-5, (n, op, float(rel.max()))
            fr, _ = u.eval_pdf(i, o)
            assert torch.equal(fr.view(torch.int32), u.eval(i, o).view(torch.int32)), n
    finally:
        djb.set_utia_exact_only(gpu_ctx, False); djb.set_contract_1e5(gpu_ctx, False)
