#!/usr/bin/env python3
"""Writes tests/golden/*.json from the big-integer Python oracle (bp_oracle.py).

The reference holds no golden vectors for this path (its tests are randomised
round trips), so these fixtures are the committed known answers: inputs are
either literal or regenerated from the recorded SplitMix64 seeds, expected
outputs are canonical integers (hex).  Run from the repo root:
    python oracle/gen_golden.py
"""
from __future__ import annotations

import hashlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bp_oracle as o  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
SEED = 0xB0117E7


def hx(v):
    return "inf" if v is None else hex(v)


def pt(P):
    return None if P is None else [hex(P[0]), hex(P[1])]


def digest(vals):
    h = hashlib.sha256()
    for v in vals:
        h.update(int(v).to_bytes(32, "little"))
    return h.hexdigest()


def field_kats():
    out = {}
    for c in (o.SECP256K1, o.VESTA):
        for fname, mod in (("fq", c.p), ("fr", c.n)):
            rng = o.SplitMix64(SEED + 100 + c.id)
            vals = [0, 1, 2, mod - 1, 1 << 255 if mod >> 255 else 1 << 253] + rng.scalars(12, mod)
            rows = []
            for i, a in enumerate(vals):
                b = vals[(i * 7 + 3) % len(vals)]
                rows.append(
                    dict(a=hex(a), b=hex(b), add=hex((a + b) % mod), sub=hex((a - b) % mod), mul=hex(a * b % mod),
                         neg=hex(-a % mod), inv=hex(0 if a == 0 else pow(a, -1, mod)),
                         mont=[hex(l) for l in o.int_to_limbs(o.to_mont(a, mod))]))
            out[f"{c.name}.{fname}"] = dict(modulus=hex(mod), rows=rows)
    # the literal expectations of src/vector_ops.rs:238-314
    out["vector_ops_literals"] = dict(
        add=dict(a=[1, 2, 3], b=[4, 5, 6], out=[5, 7, 9]),
        inner_product=dict(a=[1, 2, 3], b=[4, 5, 6], out=32),
        composed=dict(v1=[1, 2, 3], v2=[2, 3, 4], v3=[1, 1, 1], c=5, out=75),
        mat_mul_l=dict(vector=[2, 3], matrix=[[1, 4], [2, 5], [3, 6]], out=[8, 23]),
        mat_mul_r=dict(matrix=[[1, 2, 3], [4, 5, 6]], vector=[1, 2, 3], out=[14, 32]),
    )
    return out


def curve_kats():
    out = {}
    for c in (o.SECP256K1, o.VESTA):
        rng = o.SplitMix64(SEED + 200 + c.id)
        ks = [1, 2, 3, 7, c.n - 1, c.n - 2, (c.n + 1) // 2] + rng.scalars(6, c.n)
        mult = [dict(k=hex(k), P=pt(o.scalar_mul(k, c.g, c))) for k in ks]
        P = o.scalar_mul(ks[-1], c.g, c)
        Q = o.scalar_mul(ks[-2], c.g, c)
        out[c.name] = dict(g=pt(c.g), multiples=mult, n_times_g="inf" if o.scalar_mul(c.n, c.g, c) is None else "BAD",
                           add=dict(P=pt(P), Q=pt(Q), sum=pt(o.add(P, Q, c)), dbl=pt(o.add(P, P, c)),
                                    p_minus_p=pt(o.add(P, o.neg(P, c), c))))
    return out


def msm_case_inputs(name, n, c, seed):
    """Deterministic inputs for a named MSM case (also used by the tests)."""
    rng = o.SplitMix64(seed)
    r0, q = rng.field(c.n), rng.field(c.n)
    pts = o.gen_points_arith(n, r0, q, c)
    ks = rng.scalars(n, c.n)
    if name == "zeros":
        ks = [0] * n
    elif name == "ones":
        ks = [1] * n
    elif name == "minus_ones":
        ks = [c.n - 1] * n
    elif name == "bits":  # the a_L || a_R shape of src/range/aggregate.rs:50-64
        ks = [(rng.next() & 1) for _ in range(n // 2)]
        ks = ks + [(k - 1) % c.n for k in ks] + [0] * (n - 2 * (n // 2))
    elif name == "identity_base":
        pts[n // 2] = None
        pts[0] = None
    elif name == "dup_bases":
        for i in range(1, n, 3):
            pts[i] = pts[i - 1]
    elif name == "neg_pairs":  # P, -P with the same scalar cancel
        for i in range(1, n, 2):
            pts[i] = o.neg(pts[i - 1], c)
            ks[i] = ks[i - 1]
    elif name == "big_scalars":
        ks = [c.n - 1 - (rng.next() & 0xFFFF) for _ in range(n)]
    elif name == "same_scalar":
        ks = [ks[0]] * n
    return pts, ks


MSM_CASES = [("random", 1), ("random", 2), ("random", 33), ("random", 256), ("zeros", 17), ("ones", 40),
             ("minus_ones", 40), ("bits", 128), ("identity_base", 20), ("dup_bases", 30), ("neg_pairs", 32),
             ("big_scalars", 25), ("same_scalar", 64)]


def msm_fixtures():
    out = []
    for c in (o.SECP256K1, o.VESTA):
        for idx, (name, n) in enumerate(MSM_CASES):
            seed = SEED + 300 + 16 * idx + c.id
            pts, ks = msm_case_inputs(name, n, c, seed)
            exp = o.msm_naive(pts, ks, c)
            assert exp == o.msm_pippenger(pts, ks, c, window=5)
            out.append(dict(curve=c.name, case=name, n=n, seed=seed, expect=pt(exp)))
        # closed-form larger cases
        for n in (1 << 10, 1 << 12):
            seed = SEED + 900 + n + c.id
            rng = o.SplitMix64(seed)
            r0, q = rng.field(c.n), rng.field(c.n)
            ks = rng.scalars(n, c.n)
            out.append(dict(curve=c.name, case="closed_form", n=n, seed=seed, expect=pt(o.msm_closed_form(ks, r0, q, c))))
    return out


def ipa_inputs(c, n, seed):
    rng = o.SplitMix64(seed)
    g = o.gen_points_arith(n, rng.field(c.n), rng.field(c.n), c)
    h = o.gen_points_arith(n, rng.field(c.n), rng.field(c.n), c)
    a = rng.scalars(n, c.n)
    b = rng.scalars(n, c.n)
    k = n.bit_length() - 1
    ch = rng.scalars(k, c.n)
    return g, h, a, b, ch


def ipa_fixtures():
    out = []
    for c in (o.SECP256K1, o.VESTA):
        for n in (8, 64):
            for variant in (o.BULLETPROOF, o.POLYCOMMIT):
                seed = SEED + 500 + n + 2 * variant + c.id
                g, h, a, b, ch = ipa_inputs(c, n, seed)
                rounds, (gf, hf, af, bf) = o.ipa_prove(variant, g, h, a, b, ch, c)
                ss = o.s_vector(ch, c.n)
                assert gf[0] == o.msm_naive(g, ss, c)  # src/poly_commit/mod.rs:108-113
                out.append(
                    dict(curve=c.name, n=n, variant=variant, seed=seed,
                         rounds=[dict(L=pt(r["L"]), R=pt(r["R"]), cL=hex(r["cL"]), cR=hex(r["cR"])) for r in rounds],
                         final=dict(g=pt(gf[0]), h=pt(hf[0]) if hf else None, a=hex(af[0]), b=hex(bf[0]))))
    return out


def vector_fixtures():
    out = []
    for c in (o.SECP256K1, o.VESTA):
        seed = SEED + 700 + c.id
        rng = o.SplitMix64(seed)
        n = 300
        a, b = rng.scalars(n, c.n), rng.scalars(n, c.n)
        x, y = rng.field(c.n), rng.field(c.n)
        sv = {}
        for k in range(0, 11):
            ch = o.SplitMix64(seed + 1000 + k).scalars(k, c.n)
            s = o.s_vector(ch, c.n)
            sv[str(k)] = dict(first=[hex(v) for v in s[:4]], sha256=digest(s))
        xe = rng.field(c.n)
        ch5 = o.SplitMix64(seed + 1005).scalars(5, c.n)
        pts = o.gen_points_arith(24, rng.field(c.n), rng.field(c.n), c)
        fold = o.fold_generators(pts[:12], pts[12:], x, y, c)
        ysc = o.hs_prime_scalars(12, y, c.n)
        hsp = o.scale_points(pts[:12], ysc, c)
        out.append(
            dict(curve=c.name, seed=seed, n=n, inner_product=hex(o.inner_product(a, b, c.n)), sum=hex(o.vsum(a, c.n)),
                 fold_scalars_sha256=digest(o.fold_scalars(a, b, x, y, c.n)),
                 hadamard_sha256=digest(o.hadamard(a, b, c.n)), add_sha256=digest(o.vector_add(a, b, c.n)),
                 sub_sha256=digest(o.vector_sub(a, b, c.n)), scale_sha256=digest(o.scale(a, x, c.n)),
                 powers_sha256=digest(o.powers(x, 777, c.n)), batch_inverse_sha256=digest(o.batch_inversion(a, c.n)),
                 s_vector=sv, hpoly_eval=dict(x=hex(xe), value=hex(o.hpoly_evaluate(ch5, xe, c.n))),
                 fold_generators=[pt(P) for P in fold], hs_prime=[pt(P) for P in hsp]))
    return out


def main():
    os.makedirs(OUT, exist_ok=True)
    for name, fn in (("field_kats", field_kats), ("curve_kats", curve_kats), ("msm", msm_fixtures),
                     ("ipa", ipa_fixtures), ("vectors", vector_fixtures)):
        data = fn()
        with open(os.path.join(OUT, name + ".json"), "w") as f:
            json.dump(data, f, indent=1)
        print("wrote", name)


if __name__ == "__main__":
    main()
