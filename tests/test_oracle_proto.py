"""CPU tests of the C oracle's protocol drivers (oracle/bp_oracle_proto.c): every prover
message equals the big-integer Python oracle's on small sizes, the restated verifiers
accept what the restated provers emit and reject corrupted proofs, and the identities the
reference's own tests assert hold (prove -> verify round trips: src/ipa/mod.rs:246-288,
src/range/mod.rs:281-306, src/range/aggregate.rs:313-344, src/poly_commit/mod.rs:383-464)."""
import numpy as np
import pytest

import bp_oracle as o
from util import xyz_to_affine

CURVES = [o.SECP256K1, o.VESTA]
IDENTITY = None


def xy8(P, c):
    return np.concatenate([o.pack_field([P[0]], c.p)[0], o.pack_field([P[1]], c.p)[0]])


def chal(seed, j, mod):
    return o.SplitMix64(seed + j).field(mod)


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("n", [1, 2, 8])
def test_ipa_prove_verify(coracle, c, n):
    rng = o.SplitMix64(500 + n + c.id)
    g = o.gen_points_arith(n, rng.field(c.n), rng.field(c.n), c)
    h = o.gen_points_arith(n, rng.field(c.n), rng.field(c.n), c)
    u = o.scalar_mul(rng.field(c.n), c.g, c)
    a, b = rng.scalars(n, c.n), rng.scalars(n, c.n)
    G, H, A, B = o.pack_points(g, c), o.pack_points(h, c), o.pack_field(a, c.n), o.pack_field(b, c.n)
    seed = 77
    P = coracle.ipa_statement(c.id, G, H, xy8(u, c), A, B)
    eP = o.add(o.add(o.msm_naive(g, a, c), o.msm_naive(h, b, c), c), o.scalar_mul(o.inner_product(a, b, c.n), u, c), c)
    assert xyz_to_affine(P, c) == eP
    got = coracle.ipa_prove(c.id, G, H, xy8(u, c), A, B, seed, threads=2)
    k = n.bit_length() - 1
    chs = [chal(seed, j, c.n) for j in range(k)]
    rounds, (_, _, af, bf) = o.ipa_prove(o.BULLETPROOF, g, h, a, b, chs, c)
    for j, r in enumerate(rounds):
        assert xyz_to_affine(got["L"][j], c) == o.add(r["L"], o.scalar_mul(r["cL"], u, c), c)
        assert xyz_to_affine(got["R"][j], c) == o.add(r["R"], o.scalar_mul(r["cR"], u, c), c)
    assert o.unpack_field(got["a"], c.n) == af and o.unpack_field(got["b"], c.n) == bf
    # ipa::verify: the deferred MSM of a valid proof is the identity; the Python Msm agrees on a corrupted one
    v = coracle.ipa_verify(c.id, G, H, xy8(u, c), P, n, got["L"], got["R"], got["a"], got["b"], seed)
    assert xyz_to_affine(v, c) is IDENTITY
    bad_a = o.pack_field([(af[0] + 1) % c.n], c.n)[0]
    v = coracle.ipa_verify(c.id, G, H, xy8(u, c), P, n, got["L"], got["R"], bad_a, got["b"], seed)
    gs_s, hs_s = o.ipa_verify_scalars(chs, (af[0] + 1) % c.n, bf[0], c)
    m = o.Msm(c)
    m.upsert(u, (af[0] + 1) * bf[0] % c.n)
    m.upsert_batch(zip(g, gs_s))
    m.upsert_batch(zip(h, hs_s))
    m.upsert(eP, c.n - 1)
    for j in range(k):
        m.upsert(xyz_to_affine(got["L"][j], c), (-chs[j] * chs[j]) % c.n)
        m.upsert(xyz_to_affine(got["R"][j], c), (-pow(chs[j], -2, c.n)) % c.n)
    assert xyz_to_affine(v, c) == m.execute() and m.execute() is not IDENTITY


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("m,n_bits", [(1, 8), (2, 4), (1, 64)])
def test_range_aggregate_prove_verify(coracle, c, m, n_bits):
    """m = 1 is also the single-value proof of src/range/mod.rs (same expressions term by term)"""
    n = m * n_bits
    rng = o.SplitMix64(900 + n + c.id)
    gs = o.gen_points_arith(n, rng.field(c.n), rng.field(c.n), c)
    hs = o.gen_points_arith(n, rng.field(c.n), rng.field(c.n), c)
    u, g, h = (o.scalar_mul(rng.field(c.n), c.g, c) for _ in range(3))
    values = [rng.next() & ((1 << n_bits) - 1) for _ in range(m)]
    gammas = rng.scalars(m, c.n)
    G, H = o.pack_points(gs, c), o.pack_points(hs, c)
    ugh = np.stack([xy8(P, c) for P in (u, g, h)])
    gam = o.pack_field(gammas, c.n)
    got = coracle.range_agg_prove(c.id, G, H, ugh, values, gam, n_bits, 11, 22, threads=2)
    if n <= 16:
        exp = o.range_aggregate_prove(c, gs, hs, u, g, h, values, gammas, n_bits, 11, 22)
        for key in ("A", "S", "TT1", "TT2"):
            assert xyz_to_affine(got[key], c) == exp[key], key
        for key in ("tao_x", "mu", "t_hat", "a", "b"):
            assert o.unpack_field(got[key], c.n)[0] == exp[key], key
        for j in range(len(exp["L"])):
            assert xyz_to_affine(got["L"][j], c) == exp["L"][j] and xyz_to_affine(got["R"][j], c) == exp["R"][j]
    vcom = coracle.range_value_commitments(c.id, ugh, values, gam)
    for j in range(m):
        assert xyz_to_affine(vcom[j], c) == o.add(o.scalar_mul(values[j], g, c), o.scalar_mul(gammas[j], h, c), c)
    ok = coracle.range_agg_verify(c.id, G, H, ugh, values, gam, n_bits, got["points"], got["scalars"], got["lr"], 33, 22)
    assert xyz_to_affine(ok, c) is IDENTITY
    bad = got["scalars"].copy()
    bad[2] = o.pack_field([(o.unpack_field(bad[2], c.n)[0] + 1) % c.n], c.n)[0]  # t_hat + 1
    v = coracle.range_agg_verify(c.id, G, H, ugh, values, gam, n_bits, got["points"], bad, got["lr"], 33, 22)
    assert xyz_to_affine(v, c) is not IDENTITY
    # a value outside the range cannot be proved: the proof of v + 2^n_bits' bits is for another statement
    if n_bits < 64:
        wrong = list(values)
        wrong[0] ^= 1
        v = coracle.range_agg_verify(c.id, G, H, ugh, wrong, gam, n_bits, got["points"], got["scalars"], got["lr"], 33, 22)
        assert xyz_to_affine(v, c) is not IDENTITY


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("n,deg", [(1, 1), (2, 2), (8, 5), (16, 16)])
def test_poly_commit(coracle, c, n, deg):
    rng = o.SplitMix64(1300 + n + c.id)
    pts = o.gen_points_arith(n + 1, rng.field(c.n), rng.field(c.n), c)
    gs, h = pts[:n], pts[n]
    coeffs = rng.scalars(deg, c.n)
    r, x = rng.field(c.n), rng.field(c.n)
    G = o.pack_points(gs, c)
    hg = np.stack([xy8(h, c), xy8(c.g, c)])
    C_, R_, X_ = o.pack_field(coeffs, c.n), o.pack_field([r], c.n)[0], o.pack_field([x], c.n)[0]
    com, ev = coracle.poly_statement(c.id, G, hg, C_, R_, X_)
    ecom, eev = o.poly_commit_statement(c, gs, h, coeffs, r, x)
    assert xyz_to_affine(com, c) == ecom and o.unpack_field(ev, c.n)[0] == eev
    got = coracle.poly_prove(c.id, G, hg, C_, R_, X_, 5, 6, threads=2)
    Ls, Rs, a, rr, g, ui = o.poly_commit_prove(c, gs, h, coeffs, r, x, 5, 6)
    for j in range(len(Ls)):
        assert xyz_to_affine(got["L"][j], c) == Ls[j] and xyz_to_affine(got["R"][j], c) == Rs[j]
    assert o.unpack_field(got["a"], c.n)[0] == a and o.unpack_field(got["r"], c.n)[0] == rr
    assert o.unpack_field(got["ui"], c.n) == ui
    gg = None if got["g"][8] else (o.unpack_field(got["g"][0:4], c.p)[0], o.unpack_field(got["g"][4:8], c.p)[0])
    assert gg == g
    res, ui2 = coracle.poly_verify(c.id, G, hg, com, X_, ev, got["L"], got["R"], got["a"], got["r"], 6)
    assert xyz_to_affine(res, c) is IDENTITY and o.unpack_field(ui2, c.n) == ui
    res, _ = coracle.poly_verify(c.id, G, hg, com, X_, ev, got["L"], got["R"], got["a"], got["r"], 6, g_assumed=got["g"])
    assert xyz_to_affine(res, c) is IDENTITY
    bad_ev = o.pack_field([(eev + 1) % c.n], c.n)[0]
    for ga in (None, got["g"]):
        res, _ = coracle.poly_verify(c.id, G, hg, com, X_, bad_ev, got["L"], got["R"], got["a"], got["r"], 6, g_assumed=ga)
        assert xyz_to_affine(res, c) is not IDENTITY
    # fold of two deferred checks (src/poly_commit/mod.rs:228-262) against the Python restatement
    coeffs2 = list(reversed(coeffs))
    got2 = coracle.poly_prove(c.id, G, hg, o.pack_field(coeffs2, c.n), R_, X_, 15, 16, threads=2)
    alpha, x2 = rng.field(c.n), rng.field(c.n)
    uis = np.stack([got["ui"], got2["ui"]]) if n > 1 else np.zeros((2, 0, 4), dtype=np.uint64)
    fc, fcom, fev = coracle.poly_fold_todos(c.id, uis, np.stack([got["g"], got2["g"]]), o.pack_field([alpha], c.n)[0],
                                            o.pack_field([x2], c.n)[0])
    g2 = None if got2["g"][8] else (o.unpack_field(got2["g"][0:4], c.p)[0], o.unpack_field(got2["g"][4:8], c.p)[0])
    ec, ecm, eev2 = o.fold_todos(c, [(g, ui), (g2, o.unpack_field(got2["ui"], c.n) if n > 1 else [])], alpha, x2)
    assert o.unpack_field(fc, c.n) == ec and xyz_to_affine(fcom, c) == ecm and o.unpack_field(fev, c.n)[0] == eev2
    # the halo identity: the folded witness commits (r = 0) to the folded statement
    wcom, wev = coracle.poly_statement(c.id, G, hg, fc, o.pack_field([0], c.n)[0], o.pack_field([x2], c.n)[0])
    assert np.array_equal(wcom, fcom) and np.array_equal(wev, fev)


def test_fold_generators_shared_doublings_matches_kernel_oracle(coracle):
    """the protocol drivers' joint double-and-add fold equals the kernel oracle's two scalar
    multiplications (bpo_fold_generators), through one IPA round at n = 4 with an identity generator"""
    c = o.SECP256K1
    rng = o.SplitMix64(99)
    n = 4
    g = o.gen_points_arith(n, rng.field(c.n), rng.field(c.n), c)
    h = o.gen_points_arith(n, rng.field(c.n), rng.field(c.n), c)
    u = o.scalar_mul(rng.field(c.n), c.g, c)
    a, b = rng.scalars(n, c.n), rng.scalars(n, c.n)
    got = coracle.ipa_prove(c.id, o.pack_points(g, c), o.pack_points(h, c), xy8(u, c), o.pack_field(a, c.n),
                            o.pack_field(b, c.n), 3)
    x = chal(3, 0, c.n)
    xi = pow(x, -1, c.n)
    G = o.pack_points(g, c)
    fx, fy, fi = coracle.fold_generators(c.id, tuple(v[:2] for v in G), tuple(v[2:] for v in G), o.pack_field([xi], c.n)[0],
                                         o.pack_field([x], c.n)[0])
    assert o.unpack_points(fx, fy, fi, c) == o.fold_generators(g[:2], g[2:], xi, x, c)
    # second-round L uses the folded generators: checked through the Python replay above; here the final scalars
    rounds, (_, _, af, bf) = o.ipa_prove(o.BULLETPROOF, g, h, a, b, [x, chal(3, 1, c.n)], c)
    assert o.unpack_field(got["a"], c.n) == af and o.unpack_field(got["b"], c.n) == bf


@pytest.mark.parametrize("n_bits", [8, 64])
def test_single_value_range_proof_is_the_one_value_aggregate(coracle, n_bits):
    """src/range/mod.rs:26-159 restated from its own expressions (Python) equals the C oracle's
    aggregated prover at m = 1, message by message: the single-value protocol (benches/range_bench.rs,
    n_bits = 64) is driven through the aggregated code path"""
    c = o.SECP256K1
    rng = o.SplitMix64(4000 + n_bits)
    gs = o.gen_points_arith(n_bits, rng.field(c.n), rng.field(c.n), c)
    hs = o.gen_points_arith(n_bits, rng.field(c.n), rng.field(c.n), c)
    u, g, h = (o.scalar_mul(rng.field(c.n), c.g, c) for _ in range(3))
    v = rng.next() & ((1 << n_bits) - 1)
    gamma = rng.field(c.n)
    exp = o.range_prove(c, gs, hs, u, g, h, v, gamma, n_bits, 11, 22)
    ugh = np.stack([xy8(P, c) for P in (u, g, h)])
    got = coracle.range_agg_prove(c.id, o.pack_points(gs, c), o.pack_points(hs, c), ugh, [v], o.pack_field([gamma], c.n),
                                  n_bits, 11, 22, threads=2)
    for key in ("A", "S", "TT1", "TT2"):
        assert xyz_to_affine(got[key], c) == exp[key], key
    for key in ("tao_x", "mu", "t_hat", "a", "b"):
        assert o.unpack_field(got[key], c.n)[0] == exp[key], key
    for j in range(len(exp["L"])):
        assert xyz_to_affine(got["L"][j], c) == exp["L"][j] and xyz_to_affine(got["R"][j], c) == exp["R"][j]
