"""Randomised parity run of the inner-product argument and of polynomial-commitment openings against the C
oracle: random curve and size (n = 2^k, k <= 14; ragged coefficient counts), every message compared
(tests/test_gpu_configs.py: check_poly and the checks of the IPA configuration test).  Stops at the first
failure (exit 1) or after the time given.  Usage (GPU box): python tools/fuzz_proto.py [seconds] [seed]"""
import json
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import bp_oracle as o  # noqa: E402
import c_oracle  # noqa: E402
import bulletproofs_amd as bp  # noqa: E402
from bulletproofs_amd import host_mirror  # noqa: E402
from test_gpu_configs import check_poly, gen_points, is_identity, rand_scalars, xy8  # noqa: E402


def ipa_case(ctx, coracle, c, k, n_proofs, seed):
    n = 1 << k
    rng = o.SplitMix64(seed)
    threads = coracle.max_threads()
    G, H = gen_points(coracle, c, rng, n), gen_points(coracle, c, rng, n)
    u_xy = xy8(o.scalar_mul(rng.field(c.n), c.g, c), c)
    A, B = rand_scalars(seed + 1, n * n_proofs), rand_scalars(seed + 2, n * n_proofs)
    res = host_mirror.ipa_roundtrip(ctx, c.id, G, H, u_xy, A, B, n_proofs, seed)
    assert res["single_ok"] and res["batch_ok"]
    assert np.array_equal(res["P"], coracle.ipa_statement(c.id, G, H, u_xy, A[:n], B[:n], threads=threads))
    exp = coracle.ipa_prove(c.id, G, H, u_xy, A[:n], B[:n], seed, threads=threads)
    assert len(res["L"]) == k
    if k:
        assert np.array_equal(res["L"], exp["L"]) and np.array_equal(res["R"], exp["R"])
    assert np.array_equal(res["a"], exp["a"]) and np.array_equal(res["b"], exp["b"])
    assert is_identity(coracle.ipa_verify(c.id, G, H, u_xy, res["P"], n, res["L"], res["R"], res["a"], res["b"], seed, threads=threads))
    bad = host_mirror.ipa_roundtrip(ctx, c.id, G, H, u_xy, A, B, n_proofs, seed, tamper=1)
    assert not bad["single_ok"] and not bad["batch_ok"]


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    coracle = c_oracle.COracle()
    ctx = bp.Context(0)
    t_end = time.time() + budget
    cases = 0
    while time.time() < t_end:
        c = (o.SECP256K1, o.VESTA)[int(rng.integers(0, 2))]
        if rng.random() < 0.5:
            case = {"kind": "ipa", "curve": c.name, "k": int(rng.integers(1, 15)), "n_proofs": int(rng.integers(1, 4)),
                    "seed": int(rng.integers(1, 1 << 20))}
            run = lambda: ipa_case(ctx, coracle, c, case["k"], case["n_proofs"], case["seed"])  # noqa: E731
        else:
            k = int(rng.integers(2, 14))  # at k = 1 check_poly's seeds make the blinder equal the first coefficient
            case = {"kind": "poly", "curve": c.name, "k": k, "n_coeffs": int(rng.integers(1, (1 << k) + 1))}
            run = lambda: check_poly(ctx, coracle, c, case["k"], case["n_coeffs"], reprove_all=True)  # noqa: E731
        try:
            run()
        except Exception:
            print(json.dumps({"failed": True, **case}), flush=True)
            traceback.print_exc()
            sys.exit(1)
        cases += 1
        if cases % 10 == 0:
            print(json.dumps({"cases": cases, "last": case}), flush=True)
    print(json.dumps({"ok": True, "cases": cases, "seed": seed}))


if __name__ == "__main__":
    main()
