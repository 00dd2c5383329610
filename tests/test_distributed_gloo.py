"""The N>1 path on CPU: two gloo ranks shard the units (materials / pair ranges) with no
data-path collective and reassemble the result rows in input order.
The per-unit work is stubbed with the CPU oracle here (test infrastructure); on GPUs each rank
runs the HIP kernels on its own device (bench.py --gpus N)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n_units, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dj_brdf_amd import shard, synth
    import oraclelib
    O = oraclelib.oracle()
    g = O.microfacet("ggx")
    # (1) material-style sharding: round robin, rows gathered in input order on every rank
    mine = shard.round_robin(n_units, world, rank)
    rows = [(k, float(k) * 0.5 + 1.0) for k in mine]
    allrows = shard.gather_rows(rows, world, rank, n_units)
    ok1 = allrows == [float(k) * 0.5 + 1.0 for k in range(n_units)]
    # (2) pair-range sharding: each rank evaluates its block; blocks concatenate to the full batch
    n = 10007
    lo, hi = shard.block_range(n, world, rank)
    i = synth.directions_aos(hi - lo, synth.SEED_I, start=lo)
    o = synth.directions_aos(hi - lo, synth.SEED_O, start=lo)
    part = O.eval(g, i, o, ("elliptic", 0.3, 0.3, 0.0))
    # timing reduction used by bench.py: MAX over ranks
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    parts = [None] * world
    dist.all_gather_object(parts, (lo, part))
    full = np.concatenate([p for _, p in sorted(parts, key=lambda x: x[0])])
    want = O.eval(g, synth.directions_aos(n, synth.SEED_I), synth.directions_aos(n, synth.SEED_O),
                  ("elliptic", 0.3, 0.3, 0.0))
    ok2 = np.array_equal(full.view(np.uint32), want.view(np.uint32))
    q.put((rank, ok1, ok2, float(t.item())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_sharding_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, 11, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok1, ok2, tmax in res:
        assert ok1, f"rank {rank}: material rows not reassembled in input order"
        assert ok2, f"rank {rank}: sharded pair ranges do not concatenate to the unsharded result"
        assert tmax == float(world)
