"""GPU tests of the C++ host mirror (bulletproofs_amd/host): the Msm accumulator and
the IPA prove -> verify loops of the reference, driven through the C ABI."""
import numpy as np
import pytest

import bp_oracle as o
from util import xyz_to_affine

pytestmark = pytest.mark.gpu
CURVES = [o.SECP256K1, o.VESTA]


def seeded_challenge(seed, j, mod):
    return o.SplitMix64(seed + j).field(mod)


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_msm_accumulator_merges_duplicates(ctx, c):
    """src/msm.rs:33-38,84-87"""
    from bulletproofs_amd import host_mirror

    pts = o.gen_points_arith(12, 5, 9, c)
    rng = o.SplitMix64(1)
    bases = pts + pts[:5] + [None, None] + [pts[3]]
    ks = rng.scalars(len(bases), c.n)
    ks[12] = (-ks[0]) % c.n  # cancels pts[0]
    got, distinct = host_mirror.msm_accumulate(ctx, c.id, o.pack_points(bases, c), o.pack_field(ks, c.n))
    assert distinct == 13  # 12 points + the identity
    m = o.Msm(c)
    m.upsert_batch(zip(bases, ks))
    assert xyz_to_affine(got, c) == m.execute() == o.msm_naive(bases, ks, c)


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("n", [1, 2, 16, 128])
def test_ipa_prove_verify_roundtrip(ctx, c, n):
    """src/ipa/mod.rs:246-288 (prove -> verify) and :290-327 (batch of proofs), with the
    transcript replaced by seeded challenges; every L_j, R_j, a, b checked against the oracle"""
    from bulletproofs_amd import host_mirror

    rng = o.SplitMix64(4242 + n + c.id)
    g = o.gen_points_arith(n, rng.field(c.n), rng.field(c.n), c)
    h = o.gen_points_arith(n, rng.field(c.n), rng.field(c.n), c)
    u = o.scalar_mul(rng.field(c.n), c.g, c)
    n_proofs = 3
    a = [rng.scalars(n, c.n) for _ in range(n_proofs)]
    b = [rng.scalars(n, c.n) for _ in range(n_proofs)]
    seed = 99
    u_xy = np.concatenate([o.pack_field([u[0]], c.p)[0], o.pack_field([u[1]], c.p)[0]])
    A = o.pack_field([v for w in a for v in w], c.n)
    B = o.pack_field([v for w in b for v in w], c.n)
    res = host_mirror.ipa_roundtrip(ctx, c.id, o.pack_points(g, c), o.pack_points(h, c), u_xy, A, B, n_proofs, seed)
    assert res["single_ok"] and res["batch_ok"]
    # oracle replay of proof 0
    k = n.bit_length() - 1
    ch = [seeded_challenge(seed, j, c.n) for j in range(k)]
    rounds, (gf, hf, af, bf) = o.ipa_prove(o.BULLETPROOF, g, h, a[0], b[0], ch, c)
    for j, r in enumerate(rounds):
        eL = o.add(r["L"], o.scalar_mul(r["cL"], u, c), c)
        eR = o.add(r["R"], o.scalar_mul(r["cR"], u, c), c)
        assert xyz_to_affine(res["L"][j], c) == eL and xyz_to_affine(res["R"][j], c) == eR
    assert o.unpack_field(res["a"], c.n) == af and o.unpack_field(res["b"], c.n) == bf
    P = o.add(o.add(o.msm_naive(g, a[0], c), o.msm_naive(h, b[0], c), c),
              o.scalar_mul(o.inner_product(a[0], b[0], c.n), u, c), c)
    assert xyz_to_affine(res["P"], c) == P
    # a corrupted proof must be rejected, alone and inside the batch
    bad = host_mirror.ipa_roundtrip(ctx, c.id, o.pack_points(g, c), o.pack_points(h, c), u_xy, A, B, n_proofs, seed,
                                    tamper=True)
    assert not bad["single_ok"] and not bad["batch_ok"]


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("m,n_bits", [(1, 64), (2, 8), (4, 16)])
def test_aggregated_range_proof(ctx, c, m, n_bits):
    """src/range/aggregate.rs:313-344 (prove -> verify) with seeded randomness and
    challenges: every prover message equals the oracle's, the verifier's deferred MSM
    evaluates to the identity, and a corrupted proof is rejected"""
    from bulletproofs_amd import host_mirror

    n = m * n_bits
    rng = o.SplitMix64(31337 + n + c.id)
    gs = o.gen_points_arith(n, rng.field(c.n), rng.field(c.n), c)
    hs = o.gen_points_arith(n, rng.field(c.n), rng.field(c.n), c)
    u, g, h = (o.scalar_mul(rng.field(c.n), c.g, c) for _ in range(3))
    values = [rng.next() & ((1 << n_bits) - 1) for _ in range(m)]
    values[0] = (1 << n_bits) - 1  # all ones
    gammas = rng.scalars(m, c.n)
    ugh = np.stack([np.concatenate([o.pack_field([P[0]], c.p)[0], o.pack_field([P[1]], c.p)[0]]) for P in (u, g, h)])
    res = host_mirror.range_aggregate_roundtrip(ctx, c.id, o.pack_points(gs, c), o.pack_points(hs, c), ugh, values,
                                                o.pack_field(gammas, c.n), n_bits, seed_rand=11, seed_chal=22)
    assert res["ok"]
    exp = o.range_aggregate_prove(c, gs, hs, u, g, h, values, gammas, n_bits, 11, 22)
    for key in ("A", "S", "TT1", "TT2"):
        assert xyz_to_affine(res[key], c) == exp[key], key
    for key in ("tao_x", "mu", "t_hat", "a", "b"):
        assert o.unpack_field(res[key], c.n)[0] == exp[key], key
    for j in range(len(exp["L"])):
        assert xyz_to_affine(res["L"][j], c) == exp["L"][j] and xyz_to_affine(res["R"][j], c) == exp["R"][j]
    bad = host_mirror.range_aggregate_roundtrip(ctx, c.id, o.pack_points(gs, c), o.pack_points(hs, c), ugh, values,
                                                o.pack_field(gammas, c.n), n_bits, seed_rand=11, seed_chal=22, tamper=True)
    assert not bad["ok"]


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("n,deg", [(1, 1), (2, 2), (16, 13), (64, 64)])
def test_poly_commit_open_verify_halo(ctx, c, n, deg):
    """src/poly_commit/mod.rs: prove (45-119) equals the oracle message by message (L, R, a, r,
    folded generator), verify (121-173) and lazy_verify (264-315) accept, a wrong evaluation
    is rejected, and two deferred checks folded with fold_todos_* (228-262) verify as one
    opening (the identities of the reference's poly_commit tests, src/poly_commit/mod.rs:317-420)"""
    from bulletproofs_amd import host_mirror

    rng = o.SplitMix64(4242 + n + c.id)
    pts = o.gen_points_arith(n + 1, rng.field(c.n), rng.field(c.n), c)
    gs, h = pts[:n], pts[n]
    coeffs = rng.scalars(deg, c.n)
    r, x = rng.field(c.n), rng.field(c.n)
    hg = np.stack([np.concatenate([o.pack_field([P[0]], c.p)[0], o.pack_field([P[1]], c.p)[0]]) for P in (h, c.g)])
    res = host_mirror.poly_commit_roundtrip(ctx, c.id, o.pack_points(gs, c), hg, o.pack_field(coeffs, c.n),
                                            o.pack_field([r], c.n), o.pack_field([x], c.n), seed_rand=5, seed_chal=6)
    com, ev = o.poly_commit_statement(c, gs, h, coeffs, r, x)
    Ls, Rs, a, rr, g, ui = o.poly_commit_prove(c, gs, h, coeffs, r, x, 5, 6)
    assert xyz_to_affine(res["commitment"], c) == com
    assert o.unpack_field(res["evaluation"], c.n)[0] == ev
    for j in range(len(Ls)):
        assert xyz_to_affine(res["L"][j], c) == Ls[j] and xyz_to_affine(res["R"][j], c) == Rs[j], j
    assert o.unpack_field(res["a"], c.n)[0] == a and o.unpack_field(res["r"], c.n)[0] == rr
    got_g = None if res["g"][8] else (o.unpack_field(res["g"][0:4], c.p)[0], o.unpack_field(res["g"][4:8], c.p)[0])
    assert got_g == g
    assert o.poly_commit_check(c, gs, h, com, ev, x, Ls, Rs, a, rr, 6)[0]  # the oracle's verifier accepts the same proof
    assert res["verify_ok"] and res["lazy_ok"] and res["halo_ok"]
    bad = host_mirror.poly_commit_roundtrip(ctx, c.id, o.pack_points(gs, c), hg, o.pack_field(coeffs, c.n),
                                            o.pack_field([r], c.n), o.pack_field([x], c.n), seed_rand=5, seed_chal=6,
                                            tamper=True, halo=False)
    assert not bad["verify_ok"] and not bad["lazy_ok"]


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_range_proof_batch_verification_on_the_device(ctx, coracle, c):
    """verify_batch_aux (src/msm.rs:91-115) over ten range proofs of four different statements: the random
    linear combination kept on the device (bp_batch_*) and the host's agree, accept valid batches and
    reject a batch whose last proof is corrupted.  The batch primitives themselves against the C oracle:
    acc = sum_k w_k * s_k, then one MSM."""
    from bulletproofs_amd import host_mirror
    from bulletproofs_amd._lib import lib
    import ctypes as C

    m, n_bits, distinct = 2, 8, 4
    n = m * n_bits
    rng = o.SplitMix64(777 + c.id)
    gs = o.gen_points_arith(n, rng.field(c.n), rng.field(c.n), c)
    hs = o.gen_points_arith(n, rng.field(c.n), rng.field(c.n), c)
    u, g, h = (o.scalar_mul(rng.field(c.n), c.g, c) for _ in range(3))
    values = np.array([[rng.next() & 0xFF for _ in range(m)] for _ in range(distinct)], dtype=np.uint64)
    gammas = np.stack([o.pack_field(rng.scalars(m, c.n), c.n) for _ in range(distinct)])
    ugh = np.stack([np.concatenate([o.pack_field([P[0]], c.p)[0], o.pack_field([P[1]], c.p)[0]]) for P in (u, g, h)])
    G, H = o.pack_points(gs, c), o.pack_points(hs, c)
    good = host_mirror.range_aggregate_verify_batch(ctx, c.id, G, H, ugh, values, gammas, n_bits, 10, 5, 6)
    assert good["ok_device"] and good["ok_host"] and good["ok_direct"]
    bad = host_mirror.range_aggregate_verify_batch(ctx, c.id, G, H, ugh, values, gammas, n_bits, 10, 5, 6, tamper=True)
    assert not bad["ok_device"] and not bad["ok_host"] and not bad["ok_direct"]

    # the primitives: three weighted scalar vectors over 300 registered points plus two outside points
    npts = 300
    px, py = coracle.gen_points(c.id, np.concatenate([G[0][0], G[1][0]]), np.concatenate([G[0][1], G[1][1]]), npts + 2)
    bases = ctx.register_bases(c.id, (px[:npts], py[:npts], None))
    vecs = [o.pack_field(rng.scalars(npts, c.n), c.n) for _ in range(3)]
    ws = rng.scalars(3, c.n)
    extra_s = rng.scalars(2, c.n)
    hb = C.c_void_p()
    assert lib.bp_batch_begin(ctx.h, bases.h, npts, C.byref(hb)) == 0
    for v, w in zip(vecs, ws):
        wa = o.pack_field([w], c.n)
        assert lib.bp_batch_axpy(hb, 0, v.ctypes.data, npts, wa.ctypes.data) == 0
    # a partial range too: acc[100 .. 150) += w0 * vecs[1][0 .. 50)
    w0 = o.pack_field([ws[0]], c.n)
    assert lib.bp_batch_axpy(hb, 100, vecs[1].ctypes.data, 50, w0.ctypes.data) == 0
    assert lib.bp_batch_axpy(hb, 290, vecs[1].ctypes.data, 11, w0.ctypes.data) != 0  # outside the accumulator
    out = np.zeros(12, dtype=np.uint64)
    es = o.pack_field(extra_s, c.n)
    ex, ey = np.ascontiguousarray(px[npts:]), np.ascontiguousarray(py[npts:])
    assert lib.bp_batch_msm(hb, ex.ctypes.data, ey.ctypes.data, None, es.ctypes.data, 2, out.ctypes.data) == 0
    lib.bp_batch_free(hb)
    bases.free()
    ints = [o.unpack_field(v, c.n) for v in vecs]
    acc = [sum(w * iv[i] for w, iv in zip(ws, ints)) % c.n for i in range(npts)]
    for i in range(50):
        acc[100 + i] = (acc[100 + i] + ws[0] * ints[1][i]) % c.n
    exp = coracle.msm(c.id, (px, py, None), o.pack_field(acc + extra_s, c.n), 4)
    assert np.array_equal(out, exp)
