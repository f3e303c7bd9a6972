"""Wire format of points (ark-serialize 0.5 compressed encoding, 33 bytes on both curves).
CPU: the Python restatement against values that can be written down by hand (the secp256k1
generator of SEC 2, whose y is the smaller root; the identity; the square-root routine on both base
fields).  GPU: bp_points_compress / bp_points_decompress against that restatement, bit-exact,
including the identity, both signs of y, and bytes that are no point.  The reference holds no byte
vector of this encoding (its types derive it from the arkworks crates), so parity with arkworks
itself is unpinned; what is pinned is the format as published and the algebra (y^2 = x^3 + b)."""
import numpy as np
import pytest

import bp_oracle as o

CURVES = [o.SECP256K1, o.VESTA]
ids = lambda c: c.name  # noqa: E731


def sample_points(c, n, seed):
    rng = o.SplitMix64(seed)
    return o.gen_points_arith(n, rng.field(c.n), rng.field(c.n), c)


def test_generator_bytes():
    c = o.SECP256K1
    data = o.point_compress(c.g, c)
    assert len(data) == 33
    assert data[:32] == bytes.fromhex("79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798")[::-1]
    assert data[32] == 0x00  # Gy = 0x483A... is below p/2: the "positive" root
    neg = (c.g[0], c.p - c.g[1])
    assert o.point_compress(neg, c)[32] == 0x80
    assert o.point_compress(None, c) == bytes(32) + b"\x40"


@pytest.mark.parametrize("c", CURVES, ids=ids)
def test_sqrt_and_round_trip(c):
    rng = o.SplitMix64(0x5152 + c.id)
    for _ in range(40):
        a = rng.field(c.p)
        r = o.field_sqrt(a * a % c.p, c.p)
        assert r in (a, c.p - a)
    non_residues = [a for a in range(2, 40) if pow(a, (c.p - 1) // 2, c.p) == c.p - 1]
    assert non_residues and all(o.field_sqrt(a, c.p) is None for a in non_residues)
    for P in sample_points(c, 24, 0x77 + c.id) + [None, c.g, (c.g[0], c.p - c.g[1])]:
        ok, Q = o.point_decompress(o.point_compress(P, c), c)
        assert ok and Q == P


@pytest.mark.parametrize("c", CURVES, ids=ids)
def test_rejections(c):
    good = bytearray(o.point_compress(c.g, c))
    both = bytes(good[:32]) + b"\xc0"
    assert o.point_decompress(both, c) == (False, None)
    too_big = c.p.to_bytes(32, "little") + b"\x00"
    assert o.point_decompress(too_big, c) == (False, None)
    x = 1
    while o.field_sqrt((x ** 3 + c.b) % c.p, c.p) is not None:
        x += 1
    assert o.point_decompress(x.to_bytes(32, "little") + b"\x00", c) == (False, None)
    # the six low bits of the flag byte are not part of the integer
    noisy = bytes(good[:32]) + bytes([good[32] | 0x2A])
    assert o.point_decompress(noisy, c) == (True, c.g)


@pytest.mark.gpu
@pytest.mark.parametrize("c", CURVES, ids=ids)
@pytest.mark.parametrize("n", [1, 5, 300, 4099])
def test_gpu_compress_and_decompress(ctx, c, n):
    pts = sample_points(c, n, 0x9000 + n + c.id)
    # both signs of y, the identity, the generator
    pts = [(P[0], c.p - P[1]) if i % 3 == 1 else P for i, P in enumerate(pts)]
    if n >= 5:
        pts[2] = None
        pts[4] = c.g
    X, Y, INF = o.pack_points(pts, c)
    got = ctx.points_compress(c.id, (X, Y, INF))
    expect = np.frombuffer(b"".join(o.point_compress(P, c) for P in pts), dtype=np.uint8).reshape(n, 33)
    assert np.array_equal(got, expect)
    (x2, y2, inf2), ok = ctx.points_decompress(c.id, got)
    assert ok.all()
    assert o.unpack_points(x2, y2, inf2, c) == pts
    assert np.array_equal(x2, X) and np.array_equal(y2, Y) and np.array_equal(inf2, INF)


@pytest.mark.gpu
@pytest.mark.parametrize("c", CURVES, ids=ids)
def test_gpu_decompress_rejects_what_is_no_point(ctx, c):
    rows, expect_ok = [], []
    g = o.point_compress(c.g, c)
    rows.append(g); expect_ok.append(1)
    rows.append(g[:32] + b"\xc0"); expect_ok.append(0)                        # both flags
    rows.append(c.p.to_bytes(32, "little") + b"\x00"); expect_ok.append(0)    # x = p
    rows.append(((1 << 256) - 1).to_bytes(32, "little") + b"\x80"); expect_ok.append(0)
    x = 1
    while o.field_sqrt((x ** 3 + c.b) % c.p, c.p) is not None:
        x += 1
    rows.append(x.to_bytes(32, "little") + b"\x00"); expect_ok.append(0)      # x^3 + b is no square
    rows.append(g[:32] + bytes([g[32] | 0x15])); expect_ok.append(1)          # low bits of the flag byte ignored
    rows.append(bytes(32) + b"\x40"); expect_ok.append(1)                     # the identity
    for xx in range(1, 12):                                                   # small x either way
        ok, _ = o.point_decompress(xx.to_bytes(32, "little") + b"\x80", c)
        rows.append(xx.to_bytes(32, "little") + b"\x80"); expect_ok.append(1 if ok else 0)
    data = np.frombuffer(b"".join(rows), dtype=np.uint8).reshape(len(rows), 33)
    (x2, y2, inf2), ok = ctx.points_decompress(c.id, data)
    assert ok.tolist() == expect_ok
    pts = o.unpack_points(x2, y2, inf2, c)
    for row, good, P in zip(rows, expect_ok, pts):
        if good:
            assert (True, P) == o.point_decompress(row, c)
        else:
            assert P == (0, 0) or P is None


@pytest.mark.gpu
@pytest.mark.parametrize("c", CURVES, ids=ids)
def test_gpu_base_set_from_compressed_points(ctx, c):
    """bp_bases_register_compressed: an MSM over the decoded base set equals the oracle's over the points;
    one bad element refuses the whole set and names it"""
    from bulletproofs_amd.api import BpError
    from util import xyz_to_affine

    n = 200
    pts = sample_points(c, n, 0xBA5E + c.id)
    pts[7] = None
    pts = [(P[0], c.p - P[1]) if P is not None and i % 2 else P for i, P in enumerate(pts)]
    data = np.frombuffer(b"".join(o.point_compress(P, c) for P in pts), dtype=np.uint8).reshape(n, 33).copy()
    rng = o.SplitMix64(0x5CA1 + c.id)
    ks = rng.scalars(n, c.n)
    bases = ctx.register_bases_compressed(c.id, data)
    assert len(bases) == n
    got = bases.msm(o.pack_field(ks, c.n))
    assert xyz_to_affine(got, c) == o.msm_naive(pts, ks, c)
    bases.free()
    data[11, 32] = 0xC0
    with pytest.raises(BpError) as err:
        ctx.register_bases_compressed(c.id, data)
    assert "point 11" in str(err.value)


@pytest.mark.gpu
@pytest.mark.parametrize("c", CURVES, ids=ids)
def test_gpu_scalars_on_the_wire(ctx, c):
    """bp_scalars_to_bytes / bp_scalars_from_bytes: canonical little-endian integers, values that are not
    below the group order refused"""
    rng = o.SplitMix64(0x5CA1A + c.id)
    vals = [0, 1, c.n - 1, c.n // 2] + rng.scalars(300, c.n)
    got = ctx.scalars_to_bytes(c.id, o.pack_field(vals, c.n))
    expect = np.frombuffer(b"".join(v.to_bytes(32, "little") for v in vals), dtype=np.uint8).reshape(len(vals), 32)
    assert np.array_equal(got, expect)
    back, ok = ctx.scalars_from_bytes(c.id, got)
    assert ok.all() and o.unpack_field(back, c.n) == vals
    bad = np.frombuffer(b"".join(v.to_bytes(32, "little") for v in (c.n, c.n + 5, (1 << 256) - 1, c.n - 1)),
                        dtype=np.uint8).reshape(4, 32)
    back, ok = ctx.scalars_from_bytes(c.id, bad)
    assert ok.tolist() == [0, 0, 0, 1]
    assert o.unpack_field(back, c.n) == [0, 0, 0, c.n - 1]


@pytest.mark.gpu
def test_gpu_range_proof_as_bytes(ctx):
    """a range proof written as the byte string of its prover messages and read back: every message equal,
    length as the message list gives it, and a damaged element or a wrong length refused"""
    import bulletproofs_amd as bp
    from bulletproofs_amd import host_mirror, wire

    m, n_bits = 4, 8
    gs, hs, ugh, values, gamma = host_mirror.range_bench_inputs(ctx, m, n_bits, 0xB17E5)
    proof = host_mirror.range_aggregate_roundtrip(ctx, bp.SECP256K1, gs, hs, ugh, values, gamma, n_bits, 11, 12)
    assert proof["ok"]
    k = (m * n_bits).bit_length() - 1
    data = wire.range_proof_to_bytes(ctx, bp.SECP256K1, proof)
    assert len(data) == wire.range_proof_len(k) == (4 + 2 * k) * 33 + 5 * 32
    back, ok = wire.range_proof_from_bytes(ctx, bp.SECP256K1, data, k)
    assert ok
    for key in ("A", "S", "TT1", "TT2", "tao_x", "mu", "t_hat", "L", "R", "a", "b"):
        assert np.array_equal(np.asarray(back[key]), np.asarray(proof[key])), key
    # the first point's x replaced by the field modulus: no point
    p = (1 << 256) - (1 << 32) - 977
    bad = p.to_bytes(32, "little") + data[32:]
    assert wire.range_proof_from_bytes(ctx, bp.SECP256K1, bad, k) == (None, False)
    # a scalar that is not below the group order
    off = 4 * 33
    bad = data[:off] + b"\xff" * 32 + data[off + 32:]
    assert wire.range_proof_from_bytes(ctx, bp.SECP256K1, bad, k) == (None, False)
    assert wire.range_proof_from_bytes(ctx, bp.SECP256K1, data, k + 1) == (None, False)
    assert wire.range_proof_from_bytes(ctx, bp.SECP256K1, data[:-1], k) == (None, False)
