"""Run by test_gpu_rebase.py in a process of its own (the thresholds are read once per process):
an inner-product argument whose session replaces its generators by the folded ones every two
rounds, checked round by round against the Python oracle, with reads of the folded generators in
between and at the end; once with the generators handed over with the call and once over a registered
base set with window tables (whose first replacement reads the tables).  usage: BP_IPA_REBASE_MIN=16 BP_IPA_REBASE_M=4 python tests/rebase_check.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bp_oracle as o  # noqa: E402
import bulletproofs_amd as bp  # noqa: E402
from gen_golden import ipa_inputs  # noqa: E402
from util import xyz_to_affine  # noqa: E402

ctx = bp.Context(0)
n, k = 128, 7
for c, variant, resident in [(c, v, r) for c in (o.SECP256K1, o.VESTA) for v in (0, 1) for r in (False, True)]:
    if True:
        g, h, a, b, ch = ipa_inputs(c, n, 991 + variant + 7 * c.id)
        while len(ch) < k:
            ch.append((ch[-1] * 7 + 3) % c.n)
        g[5] = None  # an identity among the generators
        bases = None
        if resident:
            both = o.pack_points(g + h, c)
            bases = ctx.register_bases(c.id, both)
            bases.precompute()
            s = ctx.ipa_begin_resident(bases, variant, 0, n, o.pack_field(a, c.n), o.pack_field(b, c.n))
        else:
            s = ctx.ipa_begin(c.id, variant, o.pack_points(g, c), o.pack_points(h, c) if variant == 0 else None,
                              o.pack_field(a, c.n), o.pack_field(b, c.n))
        gs, hs, av, bv = list(g), list(h), list(a), list(b)
        for j in range(k):
            L, R, cl, cr = s.round_lr()
            eL, eR, ecl, ecr = o.ipa_round_lr(variant, gs, hs, av, bv, c)
            assert xyz_to_affine(L, c) == eL and xyz_to_affine(R, c) == eR, (c.name, variant, j)
            assert o.unpack_field(cl, c.n)[0] == ecl and o.unpack_field(cr, c.n)[0] == ecr
            x = ch[j]
            s.round_fold(o.pack_field([x], c.n), o.pack_field([pow(x, -1, c.n)], c.n))
            gs, hs, av, bv = o.ipa_round_fold(variant, gs, hs, av, bv, x, c)
            if j in (1, 2, 3, 4):  # right after the first rebase (j = 1), one fold later, after the second (j = 3), and later
                G, H, A, B = s.read()
                assert o.unpack_points(*G, c) == gs, (c.name, variant, j, "G")
                if variant == 0:
                    assert o.unpack_points(*H, c) == hs, (c.name, variant, j, "H")
                assert o.unpack_field(A, c.n) == av and o.unpack_field(B, c.n) == bv
        G, H, A, B = s.read()
        assert o.unpack_points(*G, c) == gs
        if variant == 0:
            assert o.unpack_points(*H, c) == hs
        s.free()
        if bases is not None:
            bases.free()
        print("ok", c.name, variant, "resident" if resident else "direct", flush=True)
print("rebase check passed")
