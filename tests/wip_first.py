import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import bp_oracle as o
import bulletproofs_amd as bp

ctx = bp.Context(0)
print("int mad peak MAC/s: %.3e" % ctx.int_mad_peak(2000), flush=True)

def xyz_to_aff(xyz, c):
    x, y, z = [o.limbs_to_int(xyz[4*i:4*i+4]) for i in range(3)]
    x, y, z = [o.from_mont(v, c.p) for v in (x, y, z)]
    if z == 0: return None
    assert z == 1
    return (x, y)

for c in (o.SECP256K1, o.VESTA):
    rng = o.SplitMix64(0xB0117E7 + c.id)
    # field vectors
    n = 1000
    a = rng.scalars(n, c.n); b = rng.scalars(n, c.n)
    A, B = o.pack_field(a, c.n), o.pack_field(b, c.n)
    got = o.unpack_field(ctx.inner_product(c.id, A, B), c.n)[0]
    assert got == o.inner_product(a, b, c.n), "dot"
    assert o.unpack_field(ctx.sum(c.id, A), c.n)[0] == o.vsum(a, c.n), "sum"
    x, y = rng.field(c.n), rng.field(c.n)
    X, Y = o.pack_field([x], c.n), o.pack_field([y], c.n)
    assert o.unpack_field(ctx.fold_scalars(c.id, A, B, X, Y), c.n) == o.fold_scalars(a, b, x, y, c.n), "fold_scalars"
    assert o.unpack_field(ctx.vector_add(c.id, A, B), c.n) == o.vector_add(a, b, c.n)
    assert o.unpack_field(ctx.vector_sub(c.id, A, B), c.n) == o.vector_sub(a, b, c.n)
    assert o.unpack_field(ctx.hadamard(c.id, A, B), c.n) == o.hadamard(a, b, c.n)
    assert o.unpack_field(ctx.scale(c.id, A, X), c.n) == o.scale(a, x, c.n)
    ch = rng.scalars(11, c.n); chi = [pow(v, -1, c.n) for v in ch]
    assert o.unpack_field(ctx.s_vector(c.id, o.pack_field(ch, c.n), o.pack_field(chi, c.n)), c.n) == o.s_vector(ch, c.n), "svec"
    assert o.unpack_field(ctx.powers(c.id, X, 777), c.n) == o.powers(x, 777, c.n), "powers"
    a0 = list(a[:100]); a0[3] = 0
    assert o.unpack_field(ctx.batch_inversion(c.id, o.pack_field(a0, c.n)), c.n) == o.batch_inversion(a0, c.n), "binv"
    print(c.name, "field vectors ok", flush=True)
    # curve
    npts = 300
    pts = o.gen_points_arith(npts, 12345, 678, c)
    pts[7] = None
    ks = rng.scalars(npts, c.n); ks[5] = 0; ks[6] = 1; ks[8] = c.n - 1
    PX, PY, PI = o.pack_points(pts, c)
    K = o.pack_field(ks, c.n)
    t = time.time(); got = xyz_to_aff(ctx.msm_unchecked(c.id, (PX, PY, PI), K), c); t = time.time() - t
    exp = o.msm_pippenger(pts, ks, c)
    assert got == exp, ("msm", got, exp)
    print(c.name, "msm ok", t, flush=True)
    h = 40
    L = pts[:h]; Rr = pts[h:2*h]
    ox, oy, oi = ctx.fold_generators(c.id, (PX[:h], PY[:h], PI[:h]), (PX[h:2*h], PY[h:2*h], PI[h:2*h]), X, Y)
    assert o.unpack_points(ox, oy, oi, c) == o.fold_generators(L, Rr, x, y, c), "foldgen"
    ox, oy, oi = ctx.scale_points(c.id, (PX[:h], PY[:h], PI[:h]), K[:h])
    assert o.unpack_points(ox, oy, oi, c) == o.scale_points(L, ks[:h], c), "scale_points"
    print(c.name, "curve vectors ok", flush=True)
    # IPA
    n = 16
    g = o.gen_points_arith(n, 3, 5, c); hh = o.gen_points_arith(n, 1000, 7, c)
    a = rng.scalars(n, c.n); b = rng.scalars(n, c.n); chs = rng.scalars(4, c.n)
    for variant in (0, 1):
        rounds, (gf, hf, af, bf) = o.ipa_prove(variant, g, hh, a, b, chs, c)
        s = ctx.ipa_begin(c.id, variant, o.pack_points(g, c), o.pack_points(hh, c) if variant == 0 else None,
                          o.pack_field(a, c.n), o.pack_field(b, c.n))
        for r, xch in zip(rounds, chs):
            Lx, Rx, cl, cr = s.round_lr()
            assert xyz_to_aff(Lx, c) == r["L"] and xyz_to_aff(Rx, c) == r["R"], "ipa LR"
            assert o.unpack_field(cl, c.n)[0] == r["cL"] and o.unpack_field(cr, c.n)[0] == r["cR"]
            s.round_fold(o.pack_field([xch], c.n), o.pack_field([pow(xch, -1, c.n)], c.n))
        G, H, A2, B2 = s.read()
        assert o.unpack_points(*G, c) == gf and o.unpack_field(A2, c.n) == af and o.unpack_field(B2, c.n) == bf
        if variant == 0: assert o.unpack_points(*H, c) == hf
    print(c.name, "ipa ok", flush=True)

# bigger MSM with closed form
c = o.SECP256K1
import ctypes
for logn in (12, 16):
    n = 1 << logn
    rng = o.SplitMix64(99)
    t=time.time(); pts = o.gen_points_arith(n, 777, 31337, c); ks = rng.scalars(n, c.n); print("gen", time.time()-t, flush=True)
    P = o.pack_points(pts, c); K = o.pack_field(ks, c.n)
    for rep in range(2):
        t = time.time(); got = xyz_to_aff(ctx.msm_unchecked(c.id, P, K), c); t = time.time() - t
        print("msm 2^%d: %.4fs  acc %.3f ms  dev %.3f ms" % (logn, t, ctx.last_kernel_ms(0), ctx.last_kernel_ms(1)), flush=True)
    assert got == o.msm_closed_form(ks, 777, 31337, c)
print("ALL OK")
