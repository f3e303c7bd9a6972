"""GPU tests of the multi-GPU MSM path on one GPU: shards at a common window width through the
C entry points, the torch.distributed driver over a real one-rank RCCL group, and the native
example that uses the C ABI alone.  The result of every split equals the C oracle's MSM of the
whole input (bit-exact)."""
import json
import os
import subprocess

import numpy as np
import pytest

import bp_oracle as o
from conftest import ROOT
from test_gpu_parity import _dev, seeded_points

pytestmark = pytest.mark.gpu

CURVES = [o.SECP256K1, o.VESTA]
ids = lambda c: c.name  # noqa: E731


def raw_scalars(c, n, seed):
    """uniform values below both group orders, used directly as the in-memory (Montgomery) form"""
    rng = np.random.default_rng(seed)
    raw = rng.integers(0, 1 << 64, (max(n, 1), 4), dtype=np.uint64)
    raw[:, 3] &= np.uint64((1 << 62) - 1)
    return raw[:n]


@pytest.mark.parametrize("c", CURVES, ids=ids)
@pytest.mark.parametrize("n,world", [(1023, 2), (3, 4), (4096, 8), ((1 << 18) - 1, 2), ((1 << 19) - 1, 2), (1 << 20, 8)])
def test_shards_at_the_common_width_add_up(ctx, coracle, c, n, world):
    """shard sizes on both sides of a width threshold (512 | 511 and 2^18 | 2^18 - 1), empty shards
    (3 points over 4 ranks) and the eight-way split of the headline size"""
    import torch

    from bulletproofs_amd.api import shard_window_bits
    from bulletproofs_amd.sharded import shard_range

    if c.id == 1 and n > (1 << 19):
        pytest.skip("headline size is quoted on secp256k1")
    px, py = seeded_points(coracle, c, n, 3, 5)
    K = raw_scalars(c, n, n + world)
    dx, dy, dk = _dev(px), _dev(py), _dev(K)
    cb = shard_window_bits(n, world)
    gathered = torch.zeros((world, 64 * 16), dtype=torch.int64, device="cuda")
    nws = set()
    for r in range(world):
        b, e = shard_range(n, r, world)
        nws.add(ctx.msm_shard_windows_dev(c.id, dx.data_ptr() + 32 * b, dy.data_ptr() + 32 * b, None, dk.data_ptr() + 32 * b,
                                          e - b, cb, gathered[r].data_ptr()))
    assert len(nws) == 1
    nw = nws.pop()
    packed = gathered[:, :nw * 16].contiguous()
    got = ctx.msm_combine_windows_dev(c.id, packed.data_ptr(), world, nw, cb)
    assert np.array_equal(got, coracle.msm(c.id, (px, py, None), K, 16))


@pytest.mark.parametrize("c", CURVES, ids=ids)
def test_resident_shards(ctx, coracle, c):
    """every shard holds its range of the bases as a registered set (the CRS of a multi-GPU prover)"""
    import torch

    from bulletproofs_amd.api import shard_window_bits
    from bulletproofs_amd.sharded import shard_range

    n, world = 3000, 3
    px, py = seeded_points(coracle, c, n, 9, 11)
    K = raw_scalars(c, n, 5)
    cb = shard_window_bits(n, world)
    gathered = torch.zeros((world, 64 * 16), dtype=torch.int64, device="cuda")
    for r in range(world):
        b, e = shard_range(n, r, world)
        bases = ctx.register_bases(c.id, (px[b:e], py[b:e], None))
        nw = bases.shard_windows(K[b:e], cb, gathered[r].data_ptr())
        bases.free()
    packed = gathered[:, :nw * 16].contiguous()
    got = ctx.msm_combine_windows_dev(c.id, packed.data_ptr(), world, nw, cb)
    assert np.array_equal(got, coracle.msm(c.id, (px, py, None), K, 8))


def test_shard_width_out_of_range_is_refused(ctx):
    import torch

    import bulletproofs_amd as bp

    buf = torch.zeros(64 * 16, dtype=torch.int64, device="cuda")
    for bad in (0, 1, 21):
        with pytest.raises(bp.BpError):
            ctx.msm_shard_windows_dev(0, None, None, None, None, 0, bad, buf.data_ptr())


def test_sharded_driver_over_a_one_rank_rccl_group(ctx, coracle):
    """ShardedMsm with a real process group (backend nccl = RCCL, world size 1): the all_gather
    runs on the device; result against the oracle"""
    import torch
    import torch.distributed as dist

    from bulletproofs_amd.sharded import ShardedMsm

    c = o.SECP256K1
    n = 1 << 16
    px, py = seeded_points(coracle, c, n, 21, 23)
    K = raw_scalars(c, n, 99)
    dx, dy, dk = _dev(px), _dev(py), _dev(K)
    torch.cuda.set_device(0)
    port = 29600 + os.getpid() % 300
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        sm = ShardedMsm(ctx, c.id, dist=dist, n_total=n)
        assert sm.collective and sm.world == 1
        got = sm.msm(dx.data_ptr(), dy.data_ptr(), None, dk.data_ptr(), n)
        bases = ctx.register_bases(c.id, (px, py, None))
        got_res = sm.msm_resident(bases, K)
        bases.free()
    finally:
        dist.destroy_process_group()
    exp = coracle.msm(c.id, (px, py, None), K, 16)
    assert np.array_equal(got, exp)
    assert np.array_equal(got_res, exp)


@pytest.mark.parametrize("n,shards", [((1 << 19) - 1, 2), (100000, 5)])
def test_native_sharded_example(n, shards):
    """bulletproofs_amd/host/sharded_msm: the sharded MSM from C++ through the C ABI alone (contexts
    on as many devices as there are), checked inside the program against bp_msm on one context"""
    exe = os.path.join(ROOT, "bulletproofs_amd", "host", "sharded_msm")
    assert os.path.exists(exe), "build the host examples first (__graft_entry__.build)"
    res = subprocess.run([exe, str(n), str(shards)], capture_output=True, text=True, timeout=240)
    assert res.returncode == 0, res.stdout + res.stderr
    line = json.loads(res.stdout.strip().splitlines()[-1])
    assert line["equal"] is True and line["shards"] == shards and line["n_total"] == n


@pytest.mark.parametrize("c", CURVES, ids=ids)
@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("n,world", [(64, 4), (8, 8), (32, 1)])
def test_sharded_ipa_rounds_in_lockstep(ctx, c, variant, n, world):
    """ShardedIpa: the vectors spread over `world` sessions by residue class (here all on one GPU, driven
    in lockstep); every round's L, R, c_l, c_r equal the oracle's for the whole vectors, u terms included,
    through the switch to the replicated tail, down to the final a, b, G, H"""
    from gen_golden import ipa_inputs
    from util import xyz_to_affine

    from bulletproofs_amd.sharded import ShardedIpa

    g, h, a, b, ch = ipa_inputs(c, n, 4100 + n + variant)
    u = o.scalar_mul(424242, c.g, c)
    ux, uy, _ = o.pack_points([u], c)
    G, H = o.pack_points(g, c), (o.pack_points(h, c) if variant == 0 else None)
    A, B = o.pack_field(a, c.n), o.pack_field(b, c.n)
    shards = [ShardedIpa(ctx, c.id, variant, G, H, A, B, rank=r, world=world, gather=(lambda p: p) if world > 1 else None,
                         u_xy=np.concatenate([ux[0], uy[0]])) for r in range(min(world, n))]
    if shards[0].wants_tail:  # vectors as long as the world size: the tail starts before the first round
        packets = np.stack([s.local_tail() for s in shards])
        for s in shards:
            s.begin_tail(packets)
    gs, hs, av, bv = list(g), list(h), list(a), list(b)
    k = n.bit_length() - 1
    for j in range(k):
        if shards[0].tail is None:
            packets = np.stack([s.local_lr() for s in shards])
            results = [s.combine_lr(packets) for s in shards]
        else:
            results = [s.round_lr() for s in shards]
        eL, eR, ecl, ecr = o.ipa_round_lr(variant, gs, hs, av, bv, c)
        eL = o.add(eL, o.scalar_mul(ecl, u, c), c)
        eR = o.add(eR, o.scalar_mul(ecr, u, c), c)
        for L, R, cl, cr in results:
            assert xyz_to_affine(L, c) == eL and xyz_to_affine(R, c) == eR, j
            assert o.unpack_field(cl, c.n)[0] == ecl and o.unpack_field(cr, c.n)[0] == ecr
        x = ch[j]
        X, XI = o.pack_field([x], c.n), o.pack_field([pow(x, -1, c.n)], c.n)
        need_tail = [s.local_fold(X, XI) for s in shards]
        if any(need_tail):
            assert all(need_tail)
            packets = np.stack([s.local_tail() for s in shards])
            for s in shards:
                s.begin_tail(packets)
        gs, hs, av, bv = o.ipa_round_fold(variant, gs, hs, av, bv, x, c)
    for s in shards:
        Gf, Hf, Af, Bf = s.read()
        assert o.unpack_points(*Gf, c) == gs and o.unpack_field(Af, c.n) == av and o.unpack_field(Bf, c.n) == bv
        if variant == 0:
            assert o.unpack_points(*Hf, c) == hs
        s.free()


@pytest.mark.parametrize("c", CURVES, ids=ids)
@pytest.mark.parametrize("k,world", [(10, 8), (3, 8), (12, 2), (6, 1)])
def test_fold_todos_witness_shards_interleave_to_the_whole(ctx, c, k, world):
    """fold_todos_witness_shard on the device: rank r's slice is elements r, r + world, ... of the fold of
    the whole deferred checks (bp_fold_s_vectors over all k challenges, itself checked against the oracle)"""
    from bulletproofs_amd.sharded import fold_todos_witness_shard

    rng = o.SplitMix64(0x70D0 + 16 * k + world + c.id)
    m = 3
    ch = [[rng.field(c.n) or 1 for _ in range(k)] for _ in range(m)]
    alpha = rng.field(c.n)
    w = [pow(alpha, i, c.n) for i in range(m)]
    CH = np.stack([o.pack_field(r, c.n) for r in ch])
    CHI = np.stack([o.pack_field([pow(x, -1, c.n) for x in r], c.n) for r in ch])
    W = o.pack_field(w, c.n)
    full = ctx.fold_s_vectors(c.id, CH, CHI, W)
    expect = [0] * (1 << k)
    for i in range(m):
        expect = [(a + w[i] * s) % c.n for a, s in zip(expect, o.s_vector(ch[i], c.n))]
    assert o.unpack_field(full, c.n) == expect
    for rank in range(world):
        got = fold_todos_witness_shard(ctx, c.id, CH, CHI, W, rank, world)
        assert np.array_equal(got, full[rank::world]), (rank, world)


# ---- the polynomial commitment over several ranks (bp_host_sharded.hpp) ---------------------------

def _poly_inputs(coracle, c, k, n_coeffs, seed):
    from test_gpu_configs import gen_points, rand_scalars, xy8

    rng = o.SplitMix64(seed)
    G = gen_points(coracle, c, rng, 1 << k)
    hg = np.stack([xy8(o.scalar_mul(rng.field(c.n), c.g, c), c), xy8(c.g, c)])
    coeffs = rand_scalars(seed + 1, n_coeffs)
    r, x = rand_scalars(seed + 2, 2)
    return G, hg, coeffs, r, x


def _assert_same_flow(got, exp):
    for key in ("L", "R", "a", "r", "g", "commitment", "evaluation"):
        assert np.array_equal(got[key], exp[key]), key
    assert got["verify_ok"] == exp["verify_ok"] and got["lazy_ok"] == exp["lazy_ok"] and got["halo_ok"] == exp["halo_ok"]
    if "second" in exp:
        for part in ("second", "folded"):
            for key, v in exp[part].items():
                assert np.array_equal(got[part][key], v), (part, key)


@pytest.mark.parametrize("c", CURVES, ids=ids)
@pytest.mark.parametrize("k,n_coeffs,world", [(12, (1 << 12) - 37, 4), (12, 1 << 12, 8), (3, 8, 8), (5, 20, 2), (4, 16, 1)])
def test_sharded_poly_commit_equals_the_single_gpu_flow(ctx, coracle, c, k, n_coeffs, world):
    """commit -> open -> verify -> lazy_verify -> fold of two deferred checks -> opening of the folded statement
    (src/poly_commit/mod.rs:45-262, src/poly_commit/types.rs:137-151) with the vectors spread over `world`
    ranks (threads with a context each, here all on one GPU): every message -- the commitments and
    evaluations, all L_j, R_j, a, r, the folded generators, the folded witness and statement -- equals the
    single-GPU driver's, the first opening also the C oracle's; world = 8 on 8 elements starts with the
    replicated tail"""
    from bulletproofs_amd import host_mirror

    G, hg, coeffs, r, x = _poly_inputs(coracle, c, k, n_coeffs, 0x5A4D + 16 * k + world + c.id)
    exp = host_mirror.poly_commit_roundtrip(ctx, c.id, G, hg, coeffs, r, x, seed_rand=5, seed_chal=6)
    assert exp["verify_ok"] and exp["lazy_ok"] and exp["halo_ok"]
    got = host_mirror.sharded_poly_roundtrip_threads(c.id, G, hg, coeffs, r, x, world, seed_rand=5, seed_chal=6)
    _assert_same_flow(got, exp)
    ora = coracle.poly_prove(c.id, G, hg, coeffs, r, x, 5, 6, threads=coracle.max_threads())
    for key in ("L", "R", "a", "r", "g"):
        assert np.array_equal(got[key], ora[key]), key
    # a wrong evaluation and proofs of another shape are rejected by every rank
    for mode in (1, 2, 3):
        bad = host_mirror.sharded_poly_roundtrip_threads(c.id, G, hg, coeffs, r, x, world, seed_rand=5, seed_chal=6, tamper=mode)
        assert not bad["verify_ok"] and not bad["lazy_ok"]


def test_config5_sharded_poly_commit_2_20_over_8_ranks(ctx, coracle):
    """the degree-2^20 - 1 opening with the halo fold over eight ranks (BASELINE config 5's split) against the
    single-GPU flow, which test_gpu_configs.py checks against the C oracle message by message"""
    from bulletproofs_amd import host_mirror

    c = o.SECP256K1
    G, hg, coeffs, r, x = _poly_inputs(coracle, c, 20, 1 << 20, 0x5A4D20)
    exp = host_mirror.poly_commit_roundtrip(ctx, c.id, G, hg, coeffs, r, x, seed_rand=5, seed_chal=6)
    assert exp["verify_ok"] and exp["lazy_ok"] and exp["halo_ok"]
    got = host_mirror.sharded_poly_roundtrip_threads(c.id, G, hg, coeffs, r, x, 8, seed_rand=5, seed_chal=6)
    _assert_same_flow(got, exp)


def _spoly_worker(rank, world, port, q):
    try:
        import sys

        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import torch.distributed as dist

        import bulletproofs_amd as bp
        import c_oracle
        from bulletproofs_amd import host_mirror
        from bulletproofs_amd.sharded import host_gather

        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        c = o.SECP256K1
        k = 10
        n = 1 << k
        G, hg, coeffs, r, x = _poly_inputs(c_oracle.COracle(), c, k, n - 5, 0x90100)
        full = np.zeros((n, 4), dtype=np.uint64)
        full[:n - 5] = coeffs
        rev = np.zeros((n, 4), dtype=np.uint64)
        rev[:n - 5] = coeffs[::-1]
        ctx = bp.Context(0)
        sl = slice(rank, None, world)
        got = host_mirror.sharded_poly_roundtrip(ctx, c.id, (G[0][sl], G[1][sl], None), hg, full[sl], rev[sl], n, r, x, rank, world,
                                                 host_gather(dist), seed_rand=5, seed_chal=6)
        exp = host_mirror.poly_commit_roundtrip(ctx, c.id, G, hg, coeffs, r, x, seed_rand=5, seed_chal=6)
        exp["folded"]["witness"] = exp["folded"]["witness"][sl]  # a rank returns its slice of the folded witness
        _assert_same_flow(got, exp)
        ctx.close()
        q.put((rank, True))
        dist.barrier()
        dist.destroy_process_group()
    except BaseException as exc:
        q.put((rank, repr(exc)))
        raise


def test_sharded_poly_commit_two_processes_over_gloo():
    """the same flow with the ranks as two processes that share this GPU and exchange their packets over a
    real gloo group (host buffers, all_gather): each rank's result equals the single-GPU flow's"""
    import torch.multiprocessing as mp

    mctx = mp.get_context("spawn")
    q = mctx.Queue()
    port = 32500 + (os.getpid() % 2000)
    procs = [mctx.Process(target=_spoly_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    assert results == [(0, True), (1, True)], results


@pytest.mark.parametrize("k,ranks", [(14, 4), (16, 8)])
def test_native_sharded_poly_example(k, ranks):
    """bulletproofs_amd/host/sharded_poly: the sharded polynomial-commitment flow from C++ without Python
    (threads with a context each on as many devices as there are), compared inside the program with the same
    flow on one context"""
    exe = os.path.join(ROOT, "bulletproofs_amd", "host", "sharded_poly")
    assert os.path.exists(exe), "build the host examples first (__graft_entry__.build)"
    res = subprocess.run([exe, str(k), str(ranks)], capture_output=True, text=True, timeout=240)
    assert res.returncode == 0, res.stdout + res.stderr
    line = json.loads(res.stdout.strip().splitlines()[-1])
    assert line["equal"] is True and line["verified"] is True and line["ranks"] == ranks and line["log2n"] == k
