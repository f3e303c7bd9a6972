"""CPU tests of the N > 1 path: 2 ranks over gloo shard an MSM by point range, all-gather
their per-window partial sums and combine.  The per-rank window sums and the combine are
oracle-backed stand-ins here that follow the device format (n_windows XYZZ points of 16 u64 at
the common window width); the device versions are covered by the GPU tests."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

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


def test_shard_window_width_is_a_function_of_the_total_alone():
    """every rank derives the same width without talking to the others, also where the shard
    sizes straddle a threshold of the single-GPU choice (511 | 512, 2^17 - 1 | 2^17)"""
    from bulletproofs_amd.api import shard_window_bits

    assert shard_window_bits(1 << 20, 1) == 16
    assert shard_window_bits(1 << 21, 8) == 16
    assert shard_window_bits((1 << 18) - 1, 2) == 16   # shards of 2^17 and 2^17 - 1 points
    assert shard_window_bits((1 << 18) - 2, 2) == 8
    assert shard_window_bits(1023, 2) == 8             # shards of 512 and 511 points
    assert shard_window_bits(1022, 2) == 4
    assert shard_window_bits(3, 4) == 4                # some shards are empty
    assert shard_window_bits(0, 2) >= 2


def _worker(rank, world, port, n, q):
    try:
        _worker_body(rank, world, port, n, q)
    except BaseException as exc:  # the parent would otherwise wait for its timeout
        q.put((rank, repr(exc), -1))
        raise


def _worker_body(rank, world, port, n, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import bp_oracle as oo
    import c_oracle
    from bulletproofs_amd.sharded import ShardedMsm, shard_range

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    c = oo.SECP256K1
    co = c_oracle.COracle()
    rng = oo.SplitMix64(77)
    ks = rng.scalars(n, c.n)
    g = oo.pack_field([c.g[0], c.g[1]], c.p).reshape(8)
    px, py = co.gen_points(c.id, g, g, max(n, 1))
    b, e = shard_range(n, rank, world)
    one = oo.pack_field([1], c.p).reshape(4)
    seen = {}

    def local_windows(d_x, d_y, d_inf, d_s, n_local, window_bits, out):
        # stand-in for bp_msm_shard_windows_dev: window w holds sum_i digit_w(k_i) P_i over this rank's range
        assert n_local == e - b
        seen["bits"] = window_bits
        nw = (256 + window_bits - 1) // window_bits
        for w in range(nw):
            digits = [(k >> (window_bits * w)) & ((1 << window_bits) - 1) for k in ks[b:e]]
            xyz = co.msm(c.id, (px[b:e], py[b:e], None), oo.pack_field(digits, c.n), 1) if e > b else np.zeros(12, np.uint64)
            win = np.zeros((4, 4), dtype=np.uint64)
            if xyz[8:].any():
                win[0], win[1], win[2], win[3] = xyz[0:4], xyz[4:8], one, one
            out[w * 16:(w + 1) * 16] = torch.from_numpy(win.reshape(-1).view(np.int64).copy())
        return nw

    def combine(gathered, w, nw, cb):
        acc = None
        for win in reversed(range(nw)):
            for _ in range(cb):
                acc = oo.add(acc, acc, c)
            for r in range(w):
                x, y, zz, zzz = oo.unpack_field(gathered[r, win * 16:(win + 1) * 16].numpy().view(np.uint64).reshape(4, 4), c.p)
                acc = oo.add(acc, None if zz == 0 else (x, y), c)
        return acc

    sm = ShardedMsm(None, c.id, dist=dist, device="cpu", local_windows=local_windows, combine=combine, n_total=n)
    got = sm.msm(None, None, None, None, e - b)
    exp = co.msm(c.id, (px[:n], py[:n], None), oo.pack_field(ks, c.n), 2)
    exp_pt = None if not exp[8:].any() else tuple(oo.unpack_field(exp[:8].reshape(2, 4), c.p))
    q.put((rank, got == exp_pt, seen["bits"]))
    dist.barrier()
    dist.destroy_process_group()


def _run(world, n):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(results)


def test_two_rank_gloo_shard_gather_combine():
    assert _run(2, 50) == [(0, True, 4), (1, True, 4)]


def test_two_rank_gloo_shards_straddling_a_width_threshold():
    # 512 and 511 points: alone, the ranks would pick 8- and 4-bit windows
    assert _run(2, 1023) == [(0, True, 8), (1, True, 8)]


# ---- sharded inner-product argument over gloo, sessions and sums backed by the Python oracle ------

class _OracleSession:
    """stand-in for api.IpaSession on CPU: the same calls, answered by the Python oracle"""

    def __init__(self, oo, c, variant, gs, hs, a, b):
        self.oo, self.c, self.variant = oo, c, variant
        self.g = oo.unpack_points(*gs, c)
        self.h = oo.unpack_points(*hs, c) if hs is not None else list(self.g)
        self.a, self.b = oo.unpack_field(a, c.n), oo.unpack_field(b, c.n)
        self.u = None

    def set_u(self, u_xy):
        x, y = self.oo.unpack_field(np.asarray(u_xy).reshape(2, 4), self.c.p)
        self.u = (x, y)

    def _xyz(self, P):
        oo, c = self.oo, self.c
        out = np.zeros(12, dtype=np.uint64)
        one = oo.pack_field([1], c.p).reshape(4)
        out[0:4] = out[4:8] = one
        if P is not None:
            out[0:8] = oo.pack_field([P[0], P[1]], c.p).reshape(8)
            out[8:12] = one
        return out

    def round_lr(self):
        oo, c = self.oo, self.c
        L, R, cl, cr = oo.ipa_round_lr(self.variant, self.g, self.h, self.a, self.b, c)
        if self.u is not None:
            L = oo.add(L, oo.scalar_mul(cl, self.u, c), c)
            R = oo.add(R, oo.scalar_mul(cr, self.u, c), c)
        return self._xyz(L), self._xyz(R), oo.pack_field([cl], c.n).reshape(4), oo.pack_field([cr], c.n).reshape(4)

    def round_fold(self, x, x_inv):
        xv = self.oo.unpack_field(np.asarray(x).reshape(1, 4), self.c.n)[0]
        self.g, self.h, self.a, self.b = self.oo.ipa_round_fold(self.variant, self.g, self.h, self.a, self.b, xv, self.c)

    def read(self):
        oo, c = self.oo, self.c
        return (oo.pack_points(self.g, c), oo.pack_points(self.h, c) if self.variant == 0 else None,
                oo.pack_field(self.a, c.n), oo.pack_field(self.b, c.n))

    def free(self):
        pass


class _OracleCtx:
    def __init__(self, oo, c):
        self.oo, self.c = oo, c

    def ipa_begin(self, curve, variant, gs, hs, a, b):
        return _OracleSession(self.oo, self.c, variant, gs, hs, a, b)

    def msm_unchecked(self, curve, bases, scalars):
        oo, c = self.oo, self.c
        pts = oo.unpack_points(bases[0], bases[1], bases[2], c)
        P = oo.msm_naive(pts, oo.unpack_field(scalars, c.n), c)
        return _OracleSession._xyz(self, P)

    def sum(self, curve, a):
        return self.oo.pack_field([self.oo.vsum(self.oo.unpack_field(a, self.c.n), self.c.n)], self.c.n).reshape(4)


def _ipa_worker(rank, world, port, variant, q):
    try:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import bp_oracle as oo
        from bulletproofs_amd.sharded import ShardedIpa, host_gather
        from gen_golden import ipa_inputs

        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        c = oo.SECP256K1
        n = 8
        g, h, a, b, ch = ipa_inputs(c, n, 5150 + variant)
        u = oo.scalar_mul(31337, c.g, c)
        u_xy = oo.pack_field([u[0], u[1]], c.p).reshape(8)
        s = ShardedIpa(_OracleCtx(oo, c), c.id, variant, oo.pack_points(g, c), oo.pack_points(h, c) if variant == 0 else None,
                       oo.pack_field(a, c.n), oo.pack_field(b, c.n), rank=rank, world=world, gather=host_gather(dist),
                       u_xy=u_xy)
        gs, hs, av, bv = list(g), list(h), list(a), list(b)
        ok = True
        for j in range(3):
            L, R, cl, cr = s.round_lr()
            eL, eR, ecl, ecr = oo.ipa_round_lr(variant, gs, hs, av, bv, c)
            eL = oo.add(eL, oo.scalar_mul(ecl, u, c), c)
            eR = oo.add(eR, oo.scalar_mul(ecr, u, c), c)
            got_l = tuple(oo.unpack_field(L[:8].reshape(2, 4), c.p))
            got_r = tuple(oo.unpack_field(R[:8].reshape(2, 4), c.p))
            ok = ok and got_l == eL and got_r == eR and oo.unpack_field(cl.reshape(1, 4), c.n)[0] == ecl
            x = ch[j]
            s.round_fold(oo.pack_field([x], c.n), oo.pack_field([pow(x, -1, c.n)], c.n))
            gs, hs, av, bv = oo.ipa_round_fold(variant, gs, hs, av, bv, x, c)
        G, H, A, B = s.read()
        ok = ok and oo.unpack_points(*G, c) == gs and oo.unpack_field(A, c.n) == av and oo.unpack_field(B, c.n) == bv
        q.put((rank, bool(ok)))
        dist.barrier()
        dist.destroy_process_group()
    except BaseException as exc:
        q.put((rank, repr(exc)))
        raise


def test_two_rank_gloo_sharded_ipa():
    """ShardedIpa's exchange (one all_gather of host packets per round, the switch to the replicated tail)
    over a real two-rank group; the sessions are the Python oracle's, the sums of the ranks' shares the
    library's host sums (bph_points_sum, bph_scalars_sum)"""
    for variant in (0, 1):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = 31500 + (os.getpid() % 2000) + variant
        procs = [ctx.Process(target=_ipa_worker, args=(r, 2, port, variant, q)) for r in range(2)]
        for p in procs:
            p.start()
        results = sorted(q.get(timeout=240) for _ in range(2))
        for p in procs:
            p.join(timeout=60)
        assert results == [(0, True), (1, True)], results


# ---- the fold of deferred checks by residue class of the index ------------------------------------

class _OracleFoldCtx:
    """stand-in for api.Context.fold_s_vectors on CPU: the weighted sum of s-vectors by the Python oracle"""

    def __init__(self, oo, c):
        self.oo, self.c = oo, c

    def fold_s_vectors(self, curve, challenges, challenge_inverses, weights):
        oo, c = self.oo, self.c
        m, k = challenges.shape[0], challenges.shape[1]
        acc = [0] * (1 << k)
        for i in range(m):
            ch = oo.unpack_field(challenges[i], c.n) if k else []
            assert (oo.unpack_field(challenge_inverses[i], c.n) if k else []) == [pow(x, -1, c.n) for x in ch]
            w = oo.unpack_field(weights[i:i + 1], c.n)[0]
            acc = [(a + w * s) % c.n for a, s in zip(acc, oo.s_vector(ch, c.n))]
        return oo.pack_field(acc, c.n)


def test_fold_todos_witness_by_residue_class():
    """fold_todos_witness_shard: the slices of all ranks, interleaved, are fold_todos_witness
    (src/poly_commit/mod.rs:228-244) of the whole deferred checks, for every world size up to the
    vector length; the rank's constant is folded into the weights on the host"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import bp_oracle as oo
    from bulletproofs_amd.sharded import fold_todos_witness_shard

    for c in (oo.SECP256K1, oo.VESTA):
        rng = oo.SplitMix64(0x70D05 + c.id)
        m, k = 3, 4
        ch = [[rng.field(c.n) or 1 for _ in range(k)] for _ in range(m)]
        alpha = rng.field(c.n)
        w = [pow(alpha, i, c.n) for i in range(m)]
        full = [0] * (1 << k)
        for i in range(m):
            full = [(a + w[i] * s) % c.n for a, s in zip(full, oo.s_vector(ch[i], c.n))]
        CH = np.stack([oo.pack_field(r, c.n) for r in ch])
        CHI = np.stack([oo.pack_field([pow(x, -1, c.n) for x in r], c.n) for r in ch])
        W = oo.pack_field(w, c.n)
        ctx = _OracleFoldCtx(oo, c)
        for world in (1, 2, 4, 16):
            for rank in range(world):
                got = fold_todos_witness_shard(ctx, c.id, CH, CHI, W, rank, world)
                assert oo.unpack_field(got, c.n) == full[rank::world], (c.name, world, rank)


def test_host_sums_of_rank_shares():
    """bph_points_sum / bph_scalars_sum (the combination of a sharded round's shares on the host) against
    the Python oracle: sums with the identity, of a point and its negative, of equal points"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import bp_oracle as oo
    from bulletproofs_amd.host_mirror import points_sum, scalars_sum

    for c in (oo.SECP256K1, oo.VESTA):
        rng = oo.SplitMix64(0x5A3 + c.id)
        one = oo.pack_field([1], c.p)[0]

        def xyz(P):
            out = np.zeros(12, dtype=np.uint64)
            if P is None:
                out[0:4], out[4:8] = one, one
            else:
                out[0:4], out[4:8], out[8:12] = oo.pack_field([P[0]], c.p)[0], oo.pack_field([P[1]], c.p)[0], one
            return out

        pts = [oo.scalar_mul(rng.field(c.n), c.g, c) for _ in range(5)]
        neg = (pts[0][0], (c.p - pts[0][1]) % c.p)
        cases = [pts, [pts[0], None, pts[1]], [pts[0], neg], [pts[2], pts[2]], [None, None], [pts[3]], []]
        for case in cases:
            want = None
            for P in case:
                want = oo.add(want, P, c)
            got = points_sum(c.id, np.stack([xyz(P) for P in case]) if case else np.zeros((0, 12), dtype=np.uint64))
            assert np.array_equal(got, xyz(want)), (c.name, case)
        vals = [rng.field(c.n) for _ in range(7)] + [c.n - 1, c.n - 1]
        assert oo.unpack_field(scalars_sum(c.id, oo.pack_field(vals, c.n)).reshape(1, 4), c.n)[0] == sum(vals) % c.n
