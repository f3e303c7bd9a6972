"""Randomised parity run of the MSM against the C oracle: sizes across every front end and their boundaries,
both curves, full-range scalars (above and below r/2) with special values and repeats mixed in, identity
bases, default and overridden window widths.  Stops at the first mismatch (exit 1) or after the time given.
Usage (GPU box): python tools/fuzz_parity.py [seconds] [seed]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import bp_oracle as o  # noqa: E402
import c_oracle  # noqa: E402
import bulletproofs_amd as bp  # noqa: E402

BOUNDARIES = [1, 2, 63, 64, 65, 511, 512, 513, 17407, 17408, 17409, 65535, 65536, 65537, 131071, 131072, 131073]


def points(coracle, c, n, r0, q):
    st, sp = o.scalar_mul(r0, c.g, c), o.scalar_mul(q, c.g, c)
    sxy = np.concatenate([o.pack_field([st[0]], c.p)[0], o.pack_field([st[1]], c.p)[0]])
    pxy = np.concatenate([o.pack_field([sp[0]], c.p)[0], o.pack_field([sp[1]], c.p)[0]])
    return coracle.gen_points(c.id, sxy, pxy, n)


def scalars(rng, c, n):
    raw = rng.integers(0, 1 << 64, (n, 5), dtype=np.uint64)
    ints = [(int(a) | int(b) << 64 | int(cc) << 128 | int(d) << 192 | int(e) << 256) % c.n for a, b, cc, d, e in raw]
    kind = rng.integers(0, 6)
    if kind == 1:  # special values sprinkled in
        sp = [0, 1, 2, c.n - 1, c.n - 2, (c.n - 1) // 2, (c.n + 1) // 2, 1 << 128, (1 << 255) % c.n, 255, 256, 65535, 65536]
        for i in rng.integers(0, n, max(1, n // 50)):
            ints[int(i)] = sp[int(rng.integers(0, len(sp)))]
    elif kind == 2:  # one value repeated over a run
        a, b = sorted(int(v) for v in rng.integers(0, n + 1, 2))
        for i in range(a, b):
            ints[i] = ints[a]
    elif kind == 3:  # short scalars (a crowded low window, empty high ones)
        bits = int(rng.integers(1, 200))
        ints = [v >> (256 - bits) for v in ints]
    return o.pack_field(ints, c.n)


def run(ctx, coracle, budget, seed, log=None):
    """runs cases until `budget` seconds have passed; returns (cases, mismatch or None)"""
    rng = np.random.default_rng(seed)
    t_end = time.time() + budget
    cases = 0
    biggest = 0
    while time.time() < t_end:
        c = (o.SECP256K1, o.VESTA)[int(rng.integers(0, 2))]
        r = rng.random()
        if r < 0.3:
            n = BOUNDARIES[int(rng.integers(0, len(BOUNDARIES)))]
        elif r < 0.9:
            n = int(2 ** rng.uniform(0, 16.5))
        else:
            n = int(2 ** rng.uniform(16.5, 17.6))
        w = 0 if rng.random() < 0.6 else int(rng.integers(2, 17))
        px, py = points(coracle, c, n, int(rng.integers(1, 1 << 30)), int(rng.integers(1, 1 << 30)))
        K = scalars(rng, c, n)
        inf = None
        if rng.random() < 0.4:
            inf = (rng.random(n) < rng.uniform(0, 0.3)).astype(np.uint8)
        exp = coracle.msm(c.id, (px, py, inf), K, 8)
        try:
            ctx.set_msm_window(w)
            got = ctx.msm_unchecked(c.id, (px, py, inf), K)
        finally:
            ctx.set_msm_window(0)
        cases += 1
        biggest = max(biggest, n)
        if not np.array_equal(got, exp):
            return cases, {"mismatch": True, "curve": c.id, "n": n, "window": w, "seed": seed, "case": cases}
        if log and cases % 25 == 0:
            log(json.dumps({"cases": cases, "largest_n": biggest}))
    return cases, None


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    cases, bad = run(bp.Context(0), c_oracle.COracle(), budget, seed, lambda line: print(line, flush=True))
    if bad:
        print(json.dumps(bad), flush=True)
        sys.exit(1)
    print(json.dumps({"ok": True, "cases": cases, "seed": seed}))


if __name__ == "__main__":
    main()
