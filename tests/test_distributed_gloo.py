"""The N>1 path on CPU: two gloo ranks shard the units (materials / pair ranges) of THE PRODUCT -- each rank runs its
share through the library's own host execution path (djb.Context("cpu"), the same per-unit code as the kernels) --
with no data-path collective, and the result rows are reassembled in input order.  The oracle is only
the checker: the reassembled results must equal its unsharded values bit for bit.  On GPUs each rank runs the HIP
kernels on its own device the same way (bench.py --gpus N, tests/test_gpu_verification.py::test_two_ranks_run_the_hip_path)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_PAIRS, N_MAT = 10007, 5


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), DJB_CPU_THREADS="2")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dj_brdf_amd import djb, shard, synth
    ctx = djb.Context("cpu")                                  # the product's host path: this rank's "device"
    # (1) batch fit, materials dealt round-robin (BASELINE configs[4]): this rank's materials in one call
    mine = shard.round_robin(N_MAT, world, rank)
    mats = [djb.merl.from_table(synth.merl_table(*synth.material_recipe(k)), ctx=ctx) for k in mine]
    ab, ag = djb.fit_brdf_batch(mats, 90, True, ctx=ctx)
    rows = shard.gather_rows([(k, (float(a), float(b))) for k, a, b in zip(mine, ab, ag)], world, rank, N_MAT)
    # (2) pair-range sharding: each rank evaluates its block with inputs generated for that block
    lo, hi = shard.block_range(N_PAIRS, world, rank)
    i = synth.directions_aos(hi - lo, synth.SEED_I, start=lo)
    o = synth.directions_aos(hi - lo, synth.SEED_O, start=lo)
    g = djb.ggx(djb.fresnel.schlick((1.0, 0.71, 0.29)), True, ctx=ctx)
    m = djb.merl.from_table(synth.merl_table_hashed(), ctx=ctx)
    part = np.concatenate([g.eval(i, o, djb.microfacet.params.isotropic(0.3)), m.eval(i, o)], axis=1)
    # timing reduction used by bench.py: MAX over ranks
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    parts = [None] * world
    dist.all_gather_object(parts, (lo, part))
    full = np.concatenate([p for _, p in sorted(parts, key=lambda x: x[0])])
    q.put((rank, rows, full if rank == 0 else None, float(t.item())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_sharding_gloo(oracle):
    from dj_brdf_amd import synth
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=500) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # the checker: unsharded oracle values
    i, o = synth.directions_aos(N_PAIRS, synth.SEED_I), synth.directions_aos(N_PAIRS, synth.SEED_O)
    want = np.concatenate([oracle.eval(oracle.microfacet("ggx", ("schlick", 1.0, 0.71, 0.29), True), i, o, ("elliptic", 0.3, 0.3, 0.0)),
                           oracle.eval(oracle.merl_from_table(synth.merl_table_hashed()), i, o)], axis=1)
    fits = []
    for k in range(N_MAT):
        t = oracle.tabular_tables(oracle.tabular(oracle.merl_from_table(synth.merl_table(*synth.material_recipe(k))), 90, True))
        fits.append((float(np.float32(t["alpha_beckmann"])), float(np.float32(t["alpha_ggx"]))))
    for rank, rows, full, tmax in res:
        assert rows == fits, f"rank {rank}: sharded fits {rows} != unsharded oracle {fits}"
        assert tmax == float(world)
        if rank == 0:
            assert np.array_equal(full.view(np.uint32), want.view(np.uint32)), "sharded pair ranges do not concatenate to the unsharded result"


def _placement_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    os.environ.pop("DJB_READER_THREADS", None)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    before = sorted(os.sched_getaffinity(0))
    ranks = bench.rank_placement(torch, dist, rank, 0, world, pin=True)
    q.put((rank, before, sorted(os.sched_getaffinity(0)), ranks, os.environ.get("DJB_READER_THREADS")))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_bench_rank_placement_splits_the_cpus():
    """bench.py --gpus N: every rank reports where it runs and confines its host threads (the file pipeline's reader pool)
    to its own share of the CPUs -- here two gloo ranks without GPUs, i.e. one (unknown) NUMA node split in two."""
    import bench
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_placement_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=200) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, all0, mine0, ranks0, rt0), (_, all1, mine1, ranks1, rt1) = res
    assert all0 == all1 and ranks0 == ranks1 and [r["rank"] for r in ranks0] == [0, 1]
    if len(all0) >= 2:
        assert not set(mine0) & set(mine1), "the two ranks share CPUs"
        assert sorted(mine0 + mine1) == all0, "CPUs were dropped"
    assert int(rt0) == max(2, min(32, len(mine0))) and ranks0[1]["reader_threads"] == int(rt1)
    m = bench.scaling_model_ms(8)
    assert set(m["merl_fit_files_100.wall_ms"]) == {"1", "2", "4", "8"}
