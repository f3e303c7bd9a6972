"""CPU test of the N > 1 path: 2 ranks over gloo shard an MSM by point range,
all-gather their partial sums and combine; the per-rank MSM and the combine are
oracle-backed stand-ins here (the device versions are covered by the GPU tests)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import bp_oracle as o
from conftest import ROOT


def test_shard_ranges_cover():
    from bulletproofs_amd.sharded import shard_range

    for n in (0, 1, 7, 8, 1000, 1 << 20):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import bp_oracle as oo
    from bulletproofs_amd.sharded import ShardedMsm, shard_range

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    c = oo.SECP256K1
    n = 50
    rng = oo.SplitMix64(77)
    pts = oo.gen_points_arith(n, 5, 9, c)
    ks = rng.scalars(n, c.n)
    b, e = shard_range(n, rank, world)

    def local_windows(d_x, d_y, d_inf, d_s, n_local, out):
        # stand-in for bp_msm_windows_dev: one "window" holding this rank's partial sum (XYZZ, zz=zzz=1)
        P = oo.msm_naive(pts[b:e], ks[b:e], c)
        vals = [0, 0, 0, 0] if P is None else [P[0], P[1], 1, 1]
        arr = oo.pack_field(vals, c.p).reshape(-1).view(np.int64)
        out[:16] = torch.from_numpy(arr.copy())
        return 1, 16

    def combine(gathered, w, nw, cb):
        acc = None
        for r in range(w):
            x, y, zz, zzz = oo.unpack_field(gathered[r, :16].numpy().view(np.uint64).reshape(4, 4), c.p)
            acc = oo.add(acc, None if zz == 0 else (x, y), c)
        return acc

    sm = ShardedMsm(None, c.id, dist=dist, device="cpu", local_windows=local_windows, combine=combine)
    got = sm.msm(None, None, None, None, e - b)
    q.put((rank, got == oo.msm_naive(pts, ks, c)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_shard_gather_combine():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(results) == [(0, True), (1, True)]
