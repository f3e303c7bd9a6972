"""CPU tests: the Python oracle against literal expectations, standards' values
and algebraic properties; the C oracle against the committed golden vectors."""
import numpy as np
import pytest

import bp_oracle as o
from conftest import load_golden
from gen_golden import MSM_CASES, ipa_inputs, msm_case_inputs
from util import digest, pt_from_json, xyz_to_affine

CURVES = [o.SECP256K1, o.VESTA]


def test_vector_ops_literals():
    """src/vector_ops.rs:238-314 -- the only literal expectations the reference holds."""
    n = o.SECP256K1.n
    lit = load_golden("field_kats")["vector_ops_literals"]
    assert o.vector_add(lit["add"]["a"], lit["add"]["b"], n) == lit["add"]["out"]
    assert o.inner_product(lit["inner_product"]["a"], lit["inner_product"]["b"], n) == 32
    c = lit["composed"]
    assert o.inner_product(o.vector_add(c["v1"], c["v2"], n), o.scale(c["v3"], c["c"], n), n) == 75
    ml = lit["mat_mul_l"]
    cols = [o.inner_product(ml["vector"], [row[j] for row in ml["matrix"]], n) for j in range(2)]
    assert cols == ml["out"]
    mr = lit["mat_mul_r"]
    assert [o.inner_product(row, mr["vector"], n) for row in mr["matrix"]] == mr["out"]


def test_sec2_generator_multiples():
    c = o.SECP256K1
    assert o.on_curve(c.g, c)
    two_g = o.scalar_mul(2, c.g, c)
    assert two_g == (0xC6047F9441ED7D6D3045406E95C07CD85C778E4B8CEF3CA7ABAC09B95C709EE5,
                     0x1AE168FEA63DC339A3C58419466CEAEEF7F632653266D0E1236431A950CFE52A)
    assert o.scalar_mul(3, c.g, c)[0] == 0xF9308A019258C31049344F85F89D5229B531C845836F99B08601F113BCE036F9
    assert o.scalar_mul(c.n - 1, c.g, c) == o.neg(c.g, c)
    assert o.scalar_mul(c.n, c.g, c) is None


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_group_laws_and_msm_agreement(c):
    rng = o.SplitMix64(42 + c.id)
    P = o.scalar_mul(rng.field(c.n), c.g, c)
    Q = o.scalar_mul(rng.field(c.n), c.g, c)
    assert o.on_curve(P, c) and o.on_curve(o.add(P, Q, c), c)
    assert o.add(P, o.neg(P, c), c) is None and o.add(P, None, c) == P
    assert o.add(P, P, c) == o.scalar_mul(2, P, c)
    ks = rng.scalars(24, c.n)
    pts = o.gen_points_arith(24, 11, 13, c)
    assert o.msm_naive(pts, ks, c) == o.msm_pippenger(pts, ks, c) == o.msm_closed_form(ks, 11, 13, c)


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_statement_additivity(c):
    """src/ipa/types.rs:187-203: commitments are additive in the witness."""
    rng = o.SplitMix64(7 + c.id)
    n = 8
    g = o.gen_points_arith(n, 3, 5, c)
    a1, a2 = rng.scalars(n, c.n), rng.scalars(n, c.n)
    lhs = o.msm_naive(g, o.vector_add(a1, a2, c.n), c)
    assert lhs == o.add(o.msm_naive(g, a1, c), o.msm_naive(g, a2, c), c)


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_hpoly_and_final_generator(c):
    """src/poly_commit/mod.rs:324-346 (evaluate == <coeffs, powers>) and :108-113."""
    rng = o.SplitMix64(9 + c.id)
    for k in (1, 3, 5):
        ch = rng.scalars(k, c.n)
        x = rng.field(c.n)
        ss = o.s_vector(ch, c.n)
        assert o.hpoly_evaluate(ch, x, c.n) == o.inner_product(ss, o.powers(x, 1 << k, c.n), c.n)
        inv = o.batch_inversion(ss, c.n)
        assert inv == o.s_vector([pow(v, -1, c.n) for v in ch], c.n)  # src/ipa/mod.rs:150-154
    n = 16
    g, h, a, b, ch = ipa_inputs(c, n, 1234)
    for variant in (o.BULLETPROOF, o.POLYCOMMIT):
        _, (gf, hf, af, bf) = o.ipa_prove(variant, g, h, a, b, ch, c)
        ss = o.s_vector(ch, c.n)
        assert gf[0] == o.msm_naive(g, ss, c)
        if variant == o.BULLETPROOF:
            assert hf[0] == o.msm_naive(h, o.batch_inversion(ss, c.n), c)
            assert af[0] == o.inner_product(a, o.batch_inversion(ss, c.n), c.n) and bf[0] == o.inner_product(b, ss, c.n)


def test_msm_accumulator_semantics():
    """src/msm.rs:33-38,73-87: duplicate bases merge by adding scalars."""
    c = o.SECP256K1
    pts = o.gen_points_arith(6, 5, 9, c)
    m = o.Msm(c)
    m.upsert_batch(zip(pts, [1, 2, 3, 4, 5, 6]))
    m.upsert(pts[2], 10)
    m2 = o.Msm(c)
    m2.upsert(pts[0], c.n - 1)
    m.batch(m2)
    bases, scalars = m.bases_and_scalars()
    assert len(bases) == 6 and dict(zip(bases, scalars))[pts[2]] == 13 and dict(zip(bases, scalars))[pts[0]] == 0
    assert m.execute() == o.msm_naive(pts, [0, 2, 13, 4, 5, 6], c)


# ---- C oracle against the golden vectors -----------------------------------------


def _curve(name):
    return o.CURVES[name]


def test_c_oracle_msm_golden(coracle):
    for fx in load_golden("msm"):
        c = _curve(fx["curve"])
        if fx["case"] == "closed_form":
            rng = o.SplitMix64(fx["seed"])
            r0, q = rng.field(c.n), rng.field(c.n)
            pts = o.gen_points_arith(fx["n"], r0, q, c)
            ks = rng.scalars(fx["n"], c.n)
        else:
            pts, ks = msm_case_inputs(fx["case"], fx["n"], c, fx["seed"])
        P, K = o.pack_points(pts, c), o.pack_field(ks, c.n)
        got = xyz_to_affine(coracle.msm(c.id, P, K, 4), c)
        assert got == pt_from_json(fx["expect"]), (fx["curve"], fx["case"], fx["n"])
        if fx["n"] <= 64:
            assert xyz_to_affine(coracle.msm_naive(c.id, P, K), c) == got


def test_c_oracle_vectors_golden(coracle):
    for fx in load_golden("vectors"):
        c = _curve(fx["curve"])
        rng = o.SplitMix64(fx["seed"])
        n = fx["n"]
        a, b = rng.scalars(n, c.n), rng.scalars(n, c.n)
        x, y = rng.field(c.n), rng.field(c.n)
        A, B, X, Y = (o.pack_field(v, c.n) for v in (a, b, [x], [y]))
        assert hex(o.unpack_field(coracle.inner_product(c.id, A, B), c.n)[0]) == fx["inner_product"]
        assert hex(o.unpack_field(coracle.sum(c.id, A), c.n)[0]) == fx["sum"]
        assert digest(o.unpack_field(coracle.fold_scalars(c.id, A, B, X, Y, 2), c.n)) == fx["fold_scalars_sha256"]
        for k in (0, 1, 4, 10):
            ch = o.SplitMix64(fx["seed"] + 1000 + k).scalars(k, c.n)
            chi = [pow(v, -1, c.n) for v in ch]
            s = o.unpack_field(coracle.s_vector(c.id, o.pack_field(ch, c.n), o.pack_field(chi, c.n), 2), c.n)
            assert digest(s) == fx["s_vector"][str(k)]["sha256"]
        xe = rng.field(c.n)
        assert hex(xe) == fx["hpoly_eval"]["x"]
        pts = o.gen_points_arith(24, rng.field(c.n), rng.field(c.n), c)
        PX, PY, PI = o.pack_points(pts, c)
        ox, oy, oi = coracle.fold_generators(c.id, (PX[:12], PY[:12], PI[:12]), (PX[12:], PY[12:], PI[12:]), X, Y, 4)
        assert o.unpack_points(ox, oy, oi, c) == [pt_from_json(p) for p in fx["fold_generators"]]
        ysc = o.pack_field(o.hs_prime_scalars(12, y, c.n), c.n)
        ox, oy, oi = coracle.scale_points(c.id, (PX[:12], PY[:12], PI[:12]), ysc, 4)
        assert o.unpack_points(ox, oy, oi, c) == [pt_from_json(p) for p in fx["hs_prime"]]


def test_golden_field_and_curve_kats_selfconsistent():
    for name, blk in load_golden("field_kats").items():
        if name == "vector_ops_literals":
            continue
        mod = int(blk["modulus"], 16)
        for r in blk["rows"]:
            a, b = int(r["a"], 16), int(r["b"], 16)
            assert (a + b) % mod == int(r["add"], 16) and a * b % mod == int(r["mul"], 16)
            assert o.limbs_to_int([int(v, 16) for v in r["mont"]]) == o.to_mont(a, mod)
            if a:
                assert a * int(r["inv"], 16) % mod == 1
    for name, blk in load_golden("curve_kats").items():
        c = _curve(name)
        assert blk["n_times_g"] == "inf"
        for m in blk["multiples"]:
            assert o.on_curve(pt_from_json(m["P"]), c)
