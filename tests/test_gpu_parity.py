"""GPU parity tests: every entry point of the C ABI against the committed golden
vectors and the CPU oracles on the same seeded inputs.  Integer work: bit-exact."""
import ctypes as C

import numpy as np
import pytest

import bp_oracle as o
from conftest import load_golden
from gen_golden import ipa_inputs, msm_case_inputs
from util import digest, pt_from_json, xyz_to_affine

pytestmark = pytest.mark.gpu
CURVES = [o.SECP256K1, o.VESTA]
ids = lambda c: c.name  # noqa: E731


def _curve(name):
    return o.CURVES[name]


def _dev(arr):
    import torch

    a = np.ascontiguousarray(arr)
    if a.dtype == np.uint64:
        a = a.view(np.int64)
    return torch.from_numpy(a.copy()).cuda()


def seeded_points(coracle, c, n, r0=777, q=31337):
    """P_i = (r0 + i*q) G, generated with the C oracle (fast for large n)."""
    st, sp = o.scalar_mul(r0, c.g, c), o.scalar_mul(q, c.g, c)
    sxy = np.concatenate([o.pack_field([st[0]], c.p)[0], o.pack_field([st[1]], c.p)[0]])
    pxy = np.concatenate([o.pack_field([sp[0]], c.p)[0], o.pack_field([sp[1]], c.p)[0]])
    return coracle.gen_points(c.id, sxy, pxy, n)


def random_scalars(c, n, seed):
    """uniform canonical scalars below the modulus as Montgomery limbs (numpy path for large n)"""
    rng = np.random.default_rng(seed)
    raw = rng.integers(0, 1 << 64, (n, 4), dtype=np.uint64)
    raw[:, 3] &= np.uint64((1 << (62 if c.id else 63)) - 1)  # below both moduli
    ints = [o.limbs_to_int(r) for r in raw]
    return ints, o.pack_field(ints, c.n)


# ---- field vectors ---------------------------------------------------------------


def test_vector_kernels_golden(ctx):
    for fx in load_golden("vectors"):
        c = _curve(fx["curve"])
        rng = o.SplitMix64(fx["seed"])
        n = fx["n"]
        a, b = rng.scalars(n, c.n), rng.scalars(n, c.n)
        x, y = rng.field(c.n), rng.field(c.n)
        A, B, X, Y = (o.pack_field(v, c.n) for v in (a, b, [x], [y]))
        assert hex(o.unpack_field(ctx.inner_product(c.id, A, B), c.n)[0]) == fx["inner_product"]
        assert hex(o.unpack_field(ctx.sum(c.id, A), c.n)[0]) == fx["sum"]
        assert digest(o.unpack_field(ctx.fold_scalars(c.id, A, B, X, Y), c.n)) == fx["fold_scalars_sha256"]
        assert digest(o.unpack_field(ctx.hadamard(c.id, A, B), c.n)) == fx["hadamard_sha256"]
        assert digest(o.unpack_field(ctx.vector_add(c.id, A, B), c.n)) == fx["add_sha256"]
        assert digest(o.unpack_field(ctx.vector_sub(c.id, A, B), c.n)) == fx["sub_sha256"]
        assert digest(o.unpack_field(ctx.scale(c.id, A, X), c.n)) == fx["scale_sha256"]
        assert digest(o.unpack_field(ctx.powers(c.id, X, 777), c.n)) == fx["powers_sha256"]
        assert digest(o.unpack_field(ctx.batch_inversion(c.id, A), c.n)) == fx["batch_inverse_sha256"]
        axpy = o.unpack_field(ctx.axpy(c.id, A, X, B), c.n)
        assert axpy == [(u + x * v) % c.n for u, v in zip(a, b)]
        for k, blk in fx["s_vector"].items():
            k = int(k)
            ch = o.SplitMix64(fx["seed"] + 1000 + k).scalars(k, c.n)
            chi = [pow(v, -1, c.n) for v in ch]
            s = o.unpack_field(ctx.s_vector(c.id, o.pack_field(ch, c.n), o.pack_field(chi, c.n)), c.n)
            assert digest(s) == blk["sha256"] and [hex(v) for v in s[:4]] == blk["first"]
        # HPoly::evaluate == <coeffs, powers>  (src/poly_commit/mod.rs:324-346)
        xe = int(fx["hpoly_eval"]["x"], 16)
        ch5 = o.SplitMix64(fx["seed"] + 1005).scalars(5, c.n)
        ss = ctx.s_vector(c.id, o.pack_field(ch5, c.n), o.pack_field([pow(v, -1, c.n) for v in ch5], c.n))
        pw = ctx.powers(c.id, o.pack_field([xe], c.n), 32)
        assert hex(o.unpack_field(ctx.inner_product(c.id, ss, pw), c.n)[0]) == fx["hpoly_eval"]["value"]


def test_vector_ops_reference_literals(ctx):
    """src/vector_ops.rs:238-314 through the GPU path."""
    c = o.SECP256K1
    f = lambda v: o.pack_field(v, c.n)  # noqa: E731
    assert o.unpack_field(ctx.vector_add(c.id, f([1, 2, 3]), f([4, 5, 6])), c.n) == [5, 7, 9]
    assert o.unpack_field(ctx.inner_product(c.id, f([1, 2, 3]), f([4, 5, 6])), c.n) == [32]
    s = ctx.vector_add(c.id, f([1, 2, 3]), f([2, 3, 4]))
    assert o.unpack_field(ctx.inner_product(c.id, s, ctx.scale(c.id, f([1, 1, 1]), f([5]))), c.n) == [75]


@pytest.mark.parametrize("c", CURVES, ids=ids)
def test_vector_edge_cases(ctx, c):
    empty = np.zeros((0, 4), dtype=np.uint64)
    assert o.unpack_field(ctx.inner_product(c.id, empty, empty), c.n) == [0]
    assert o.unpack_field(ctx.sum(c.id, empty), c.n) == [0]
    assert ctx.fold_scalars(c.id, empty, empty, o.pack_field([3], c.n), o.pack_field([4], c.n)).shape == (0, 4)
    # ragged: zip to the shorter, as the reference's iterator adaptors do
    a, b = [5, 6, 7, 8, 9], [1, 2, 3]
    assert o.unpack_field(ctx.inner_product(c.id, o.pack_field(a, c.n), o.pack_field(b, c.n)), c.n) == [38]
    # extremes
    m = c.n - 1
    A = o.pack_field([m, m, 0, 1], c.n)
    assert o.unpack_field(ctx.hadamard(c.id, A, A), c.n) == [1, 1, 0, 1]
    assert o.unpack_field(ctx.vector_add(c.id, A, A), c.n) == [c.n - 2, c.n - 2, 0, 2]
    assert o.unpack_field(ctx.batch_inversion(c.id, A), c.n) == [m, m, 0, 1]
    # k = 0 s-vector is [1]
    assert o.unpack_field(ctx.s_vector(c.id, empty, empty), c.n) == [1]
    # large dot against the C oracle is covered in test_large_vectors_vs_c_oracle


@pytest.mark.parametrize("c", CURVES, ids=ids)
def test_large_vectors_vs_c_oracle(ctx, coracle, c):
    n = (1 << 17) + 13
    _, A = random_scalars(c, n, 11)
    _, B = random_scalars(c, n, 12)
    X, Y = A[5:6].copy(), B[7:8].copy()
    assert np.array_equal(ctx.inner_product(c.id, A, B), coracle.inner_product(c.id, A, B))
    assert np.array_equal(ctx.sum(c.id, A), coracle.sum(c.id, A))
    assert np.array_equal(ctx.fold_scalars(c.id, A, B, X, Y), coracle.fold_scalars(c.id, A, B, X, Y, 8))
    ch, chi = A[:15].copy(), ctx.batch_inversion(c.id, A[:15])
    assert np.array_equal(ctx.s_vector(c.id, ch, chi), coracle.s_vector(c.id, ch, chi, 8))
    fq = ctx.batch_inversion(c.id, A[:1000], field=1)  # base-field inversion
    vals = o.unpack_field(A[:1000], c.p) if c.id else None
    if vals is not None:  # Vesta: every 254-bit value is below p as well
        assert o.unpack_field(fq, c.p) == o.batch_inversion(vals, c.p)


@pytest.mark.parametrize("c", CURVES, ids=ids)
@pytest.mark.parametrize("k,m", [(0, 1), (1, 3), (5, 1), (9, 4), (13, 7)])
def test_fold_s_vectors(ctx, c, k, m):
    """sum_i alpha^i * coeffs(h_poly_i), src/poly_commit/mod.rs:228-244, against one s-vector
    per deferred check folded with axpy, and against the big-integer oracle at small sizes."""
    rng = o.SplitMix64(0x5EC + 16 * k + m)
    ch = [rng.scalars(k, c.n) for _ in range(m)]
    chi = [[pow(v, -1, c.n) for v in row] for row in ch]
    w = rng.scalars(m, c.n)
    CH = np.stack([o.pack_field(r, c.n).reshape(k, 4) for r in ch])
    CHI = np.stack([o.pack_field(r, c.n).reshape(k, 4) for r in chi])
    got = ctx.fold_s_vectors(c.id, CH, CHI, o.pack_field(w, c.n))
    acc = ctx.scale(c.id, ctx.s_vector(c.id, CH[0], CHI[0]), o.pack_field([w[0]], c.n))
    for i in range(1, m):
        acc = ctx.axpy(c.id, acc, o.pack_field([w[i]], c.n), ctx.s_vector(c.id, CH[i], CHI[i]))
    assert np.array_equal(got, acc)
    if k <= 9:
        ss = [o.s_vector(ch[i], c.n) for i in range(m)]
        exp = [sum(w[i] * ss[i][j] for i in range(m)) % c.n for j in range(1 << k)]
        assert o.unpack_field(got, c.n) == exp


@pytest.mark.parametrize("c", CURVES, ids=ids)
@pytest.mark.parametrize("m,n_bits", [(1, 8), (3, 5), (2, 64), (64, 64), (5, 33)])
def test_range_agg_vector_polynomials(ctx, c, m, n_bits):
    """l(X), r(X), t1, t2, then l, r, t_hat of the aggregated range proof
    (src/range/aggregate.rs:50-64, 88-127, 143-146; src/range/types.rs:98-129) against the
    big-integer restatement; includes the all-zero and all-ones values."""
    rng = o.SplitMix64(0xA66 + 64 * m + n_bits)
    mask = (1 << n_bits) - 1
    values = [rng.next() & mask for _ in range(m)]
    values[0] = mask
    if m > 1:
        values[1] = 0
    n = m * n_bits
    sl, sr = rng.scalars(n, c.n), rng.scalars(n, c.n)
    y, z, x = rng.field(c.n), rng.field(c.n), rng.field(c.n)
    l0, r0, r1 = [], [], []
    for j in range(m):
        for i in range(n_bits):
            k = j * n_bits + i
            bit = (values[j] >> i) & 1
            l0.append((bit - z) % c.n)
            r0.append((pow(y, k, c.n) * ((bit - 1) + z) + pow(z, 2 + j, c.n) * (1 << i)) % c.n)
            r1.append(pow(y, k, c.n) * sr[k] % c.n)
    dot = lambda a, b: sum(u * v for u, v in zip(a, b)) % c.n  # noqa: E731
    f = lambda v: o.pack_field(v, c.n)  # noqa: E731
    t1, t2 = ctx.range_agg_poly(c.id, np.array(values, dtype=np.uint64), n_bits, f(sl), f(sr), f([y]), f([z]))
    assert o.unpack_field(t1, c.n) == [(dot(l0, r1) + dot(sl, r0)) % c.n]
    assert o.unpack_field(t2, c.n) == [dot(sl, r1)]
    l, r, t_hat = ctx.range_agg_eval(c.id, f([x]), n)
    el = [(a + x * b) % c.n for a, b in zip(l0, sl)]
    er = [(a + x * b) % c.n for a, b in zip(r0, r1)]
    assert o.unpack_field(l, c.n) == el and o.unpack_field(r, c.n) == er
    assert o.unpack_field(t_hat, c.n) == [dot(el, er)]
    # t(x) = t0 + t1 x + t2 x^2 with t0 = <l0, r0> (the identity the verifier relies on)
    t1i, t2i = o.unpack_field(t1, c.n)[0], o.unpack_field(t2, c.n)[0]
    assert (dot(l0, r0) + t1i * x + t2i * x * x) % c.n == dot(el, er)
    # a second evaluation of another size is a state error
    with pytest.raises(Exception):
        ctx.range_agg_eval(c.id, f([x]), n + 1)


@pytest.mark.parametrize("c", CURVES, ids=ids)
@pytest.mark.parametrize("m,n_bits", [(1, 1), (1, 8), (2, 4), (4, 16), (64, 64)])
def test_range_agg_verifier_scalars(ctx, c, m, n_bits):
    """gs_i: a*s_i + z; hs_i: y^-i (b*s_i^-1 - z^(2+j) 2^i) - z (src/ipa/mod.rs:134-166,
    src/range/aggregate.rs:239-261,272) against the big-integer restatement"""
    n = m * n_bits
    k = n.bit_length() - 1
    rng = o.SplitMix64(0xE1F + n)
    ch = rng.scalars(k, c.n)
    chi = [pow(v, -1, c.n) for v in ch]
    a, b, y, z = (rng.field(c.n) for _ in range(4))
    yi = pow(y, -1, c.n)
    f = lambda v: o.pack_field(v, c.n)  # noqa: E731
    g, h = ctx.range_agg_verify_scalars(c.id, f(ch).reshape(k, 4), f(chi).reshape(k, 4), f([a]), f([b]), f([yi]), f([z]),
                                        m, n_bits)
    ss = o.s_vector(ch, c.n) if k else [1]
    eg = [(a * ss[i] + z) % c.n for i in range(n)]
    eh = [(pow(yi, i, c.n) * (b * pow(ss[i], -1, c.n) - pow(z, 2 + i // n_bits, c.n) * (1 << (i % n_bits))) - z) % c.n
          for i in range(n)]
    assert o.unpack_field(g, c.n) == eg and o.unpack_field(h, c.n) == eh


# ---- MSM ----------------------------------------------------------------------------


def test_msm_golden(ctx):
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
        got = xyz_to_affine(ctx.msm_unchecked(c.id, P, K), c)
        assert got == pt_from_json(fx["expect"]), (fx["curve"], fx["case"], fx["n"])


def test_msm_without_event_markers(ctx, coracle):
    """bp_set_timing(0): same results on the small, mid-size and chunked paths, and the last
    recorded durations stay as they were"""
    c = o.SECP256K1
    try:
        for n in (300, 40000, 300000):
            px, py = seeded_points(coracle, c, n, 3, 5)
            _, K = random_scalars(c, n, n)
            ctx.set_timing(True)
            exp = ctx.msm_unchecked(c.id, (px, py, None), K)
            before = ctx.last_kernel_ms(1)
            assert before > 0
            ctx.set_timing(False)
            assert np.array_equal(ctx.msm_unchecked(c.id, (px, py, None), K), exp), n
            assert ctx.last_kernel_ms(1) == before
        assert np.array_equal(exp, coracle.msm(c.id, (px, py, None), K, 8))
    finally:
        ctx.set_timing(True)


@pytest.mark.parametrize("c", CURVES, ids=ids)
def test_msm_edge_shapes(ctx, c):
    empty_f = np.zeros((0, 4), dtype=np.uint64)
    assert xyz_to_affine(ctx.msm_unchecked(c.id, (empty_f, empty_f, None), empty_f), c) is None
    pts = o.gen_points_arith(10, 5, 7, c)
    ks = o.SplitMix64(3).scalars(10, c.n)
    P, K = o.pack_points(pts, c), o.pack_field(ks, c.n)
    # unequal lengths: msm_unchecked zips to the shorter slice
    got = xyz_to_affine(ctx.msm_unchecked(c.id, P, K[:6]), c)
    assert got == o.msm_naive(pts[:6], ks[:6], c)
    # all bases the identity
    inf = np.ones(10, dtype=np.uint8)
    assert xyz_to_affine(ctx.msm_unchecked(c.id, (P[0], P[1], inf), K), c) is None


@pytest.mark.parametrize("c", CURVES, ids=ids)
def test_msm_window_sweep(ctx, coracle, c):
    """every window width gives the same group element"""
    n = 1500
    px, py = seeded_points(coracle, c, n)
    ints, K = random_scalars(c, n, 5)
    exp = coracle.msm(c.id, (px, py, None), K, 8)
    try:
        for w in (2, 3, 5, 8, 9, 10, 11, 13, 16):
            ctx.set_msm_window(w)
            assert np.array_equal(ctx.msm_unchecked(c.id, (px, py, None), K), exp), w
    finally:
        ctx.set_msm_window(0)


@pytest.mark.parametrize("c", CURVES, ids=ids)
def test_window_reduction_equal_and_opposite_buckets(ctx, c):
    """The same base point in neighbouring buckets: the sum tree of the window reduction meets
    P + P (doubling inside an addition) and P + (-P) (the identity), on the one-block reduction
    (64..256 buckets), the row/column reduction (from 512 buckets) and the segment scheme."""
    g = c.g
    pts = [g, g, g, g, o.scalar_mul(11, g, c)]
    ks = [2, 3, 5, c.n - 7, 1]
    exp = o.msm_naive(pts, ks, c)
    assert exp == o.scalar_mul(14, g, c)
    P, K = o.pack_points(pts, c), o.pack_field(ks, c.n)
    try:
        for w in (4, 7, 8, 9, 10, 12, 16):
            ctx.set_msm_window(w)
            xyz = ctx.msm_unchecked(c.id, P, K)
            assert xyz_to_affine(xyz, c) == exp, w
    finally:
        ctx.set_msm_window(0)


@pytest.mark.parametrize("c", CURVES, ids=ids)
def test_msm_degenerate_distributions(ctx, coracle, c):
    """range-proof A-commitment scalars ({0,1,-1}: src/range/aggregate.rs:50-64) and a
    single repeated scalar put every point of a window into one bucket"""
    _degenerate_case(ctx, coracle, c, 8192)


@pytest.mark.parametrize("c", CURVES, ids=ids)
def test_msm_degenerate_distributions_sliced_path(ctx, coracle, c):
    """the same scalar sets at a size that takes the sliced counting sort"""
    _degenerate_case(ctx, coracle, c, 30000)


@pytest.mark.parametrize("c", CURVES, ids=ids)
def test_msm_degenerate_distributions_two_level_sort(ctx, coracle, c):
    """the same at a size that takes the two-level counting sort: every entry of a window in one
    partition and one bucket (wave-uniform ranking, a partition larger than a block keeps)"""
    _degenerate_case(ctx, coracle, c, (1 << 18) + 77)


def _degenerate_case(ctx, coracle, c, n):
    px, py = seeded_points(coracle, c, n, 99, 101)
    rng = np.random.default_rng(8)
    bits = rng.integers(0, 2, n // 2)
    ks = [int(b) for b in bits] + [(int(b) - 1) % c.n for b in bits]
    K = o.pack_field(ks, c.n)
    assert np.array_equal(ctx.msm_unchecked(c.id, (px, py, None), K), coracle.msm(c.id, (px, py, None), K, 8))
    K2 = np.repeat(o.pack_field([ks[-1] * 12345 % c.n], c.n), n, axis=0)
    assert np.array_equal(ctx.msm_unchecked(c.id, (px, py, None), K2), coracle.msm(c.id, (px, py, None), K2, 8))


@pytest.mark.parametrize("c", CURVES, ids=ids)
@pytest.mark.parametrize("n", [12000, 40000, 1 << 19])
def test_msm_crowded_top_window(ctx, coracle, c, n):
    """scalars below 2^250 leave the top window a handful of buckets that hold every point:
    crowded buckets of each size class (block within the launch, queued block, one wave per
    bucket on the chunked path) against the closed form"""
    r0, q = 777, 4242
    px, py = seeded_points(coracle, c, n, r0, q)
    rng = np.random.default_rng(n)
    limbs = rng.integers(0, 1 << 64, (n, 4), dtype=np.uint64)
    limbs[:, 3] &= np.uint64((1 << 58) - 1)
    ints = [int(a) | int(b) << 64 | int(cc) << 128 | int(d) << 192 for a, b, cc, d in limbs]
    K = o.pack_field(ints, c.n)
    w = np.arange(n, dtype=object) * q + r0
    exp = o.scalar_mul(int(np.dot(np.array(ints, dtype=object), w)) % c.n, c.g, c)
    assert xyz_to_affine(ctx.msm_unchecked(c.id, (px, py, None), K), c) == exp


@pytest.mark.parametrize("stages", [2, 3, 8])
def test_msm_window_batches(ctx, coracle, stages, monkeypatch):
    """BP_MSM_STAGES runs the windows of a large MSM in batches on three streams; same sum"""
    c = o.SECP256K1
    n = 1 << 18
    r0, q = 5150, 2112
    px, py = seeded_points(coracle, c, n, r0, q)
    ints, K = random_scalars(c, n, 77)
    w = np.arange(n, dtype=object) * q + r0
    exp = o.scalar_mul(int(np.dot(np.array(ints, dtype=object), w)) % c.n, c.g, c)
    monkeypatch.setenv("BP_MSM_STAGES", str(stages))
    for _ in range(2):  # twice: the second call reuses streams, events and workspace
        assert xyz_to_affine(ctx.msm_unchecked(c.id, (px, py, None), K), c) == exp


@pytest.mark.parametrize("c", CURVES, ids=ids)
@pytest.mark.parametrize("logn", [16, 18])
def test_msm_large_closed_form(ctx, coracle, c, logn):
    """sum k_i (r0 + i q) G has a closed form at any size"""
    n = 1 << logn
    r0, q = 424242, 31337
    px, py = seeded_points(coracle, c, n, r0, q)
    ints, K = random_scalars(c, n, logn)
    acc = 0
    for i, k in enumerate(ints):
        acc += k * (r0 + i * q)
    exp = o.scalar_mul(acc % c.n, c.g, c)
    assert xyz_to_affine(ctx.msm_unchecked(c.id, (px, py, None), K), c) == exp


def test_msm_full_size_properties(ctx, coracle):
    """BASELINE size (2^20): closed form + linearity, device-resident inputs"""
    c = o.SECP256K1
    n = 1 << 20
    r0, q = 987654321, 1234577
    px, py = seeded_points(coracle, c, n, r0, q)
    ints1, K1 = random_scalars(c, n, 21)
    ints2, K2 = random_scalars(c, n, 22)
    dx, dy, d1, d2 = _dev(px), _dev(py), _dev(K1), _dev(K2)
    r1 = xyz_to_affine(ctx.msm_dev(c.id, dx.data_ptr(), dy.data_ptr(), None, d1.data_ptr(), n), c)
    r2 = xyz_to_affine(ctx.msm_dev(c.id, dx.data_ptr(), dy.data_ptr(), None, d2.data_ptr(), n), c)
    w = np.arange(n, dtype=object) * q + r0
    e1 = int(np.dot(np.array(ints1, dtype=object), w)) % c.n
    e2 = int(np.dot(np.array(ints2, dtype=object), w)) % c.n
    assert r1 == o.scalar_mul(e1, c.g, c) and r2 == o.scalar_mul(e2, c.g, c)
    K3 = ctx.vector_add(c.id, K1, K2)
    d3 = _dev(K3)
    r3 = xyz_to_affine(ctx.msm_dev(c.id, dx.data_ptr(), dy.data_ptr(), None, d3.data_ptr(), n), c)
    assert r3 == o.add(r1, r2, c)


@pytest.mark.parametrize("c", CURVES, ids=ids)
@pytest.mark.parametrize("n", [17407, 17408, 17409, 40000])
def test_msm_path_boundaries(ctx, coracle, c, n):
    """sizes on both sides of the switch between the one-block counting sort (<= 17408
    points) and the sliced one, with identity bases and zero / +-1 scalars mixed in"""
    px, py = seeded_points(coracle, c, n, 31 + n, 977)
    ints, K = random_scalars(c, n, n)
    special = o.pack_field([0, 1, c.n - 1, 2, c.n - 2], c.n)
    K[100:105] = special
    K[n - 5:] = special
    inf = np.zeros(n, dtype=np.uint8)
    inf[[7, 8192, n - 1]] = 1
    exp = coracle.msm(c.id, (px, py, inf), K, 8)
    assert np.array_equal(ctx.msm_unchecked(c.id, (px, py, inf), K), exp)
    try:
        for w in (7, 9):  # short top windows: many entries in few buckets
            ctx.set_msm_window(w)
            assert np.array_equal(ctx.msm_unchecked(c.id, (px, py, inf), K), exp), w
    finally:
        ctx.set_msm_window(0)


@pytest.mark.parametrize("c", CURVES, ids=ids)
@pytest.mark.parametrize("n", [40000, 70001])
def test_msm_spread_scalars(ctx, coracle, c, n):
    """Window widths that do not divide 256 on the two-level counting sort: every scalar is taken as
    k + t r (r P is the identity), so that the top window fills like the others.  Same group element as
    the oracle's, with identity bases and 0, +-1, +-2 among the scalars, and with the spreading off."""
    px, py = seeded_points(coracle, c, n, 17 + n, 1201)
    ints, K = random_scalars(c, n, n + 1)
    special = o.pack_field([0, 1, c.n - 1, 2, c.n - 2], c.n)
    K[200:205] = special
    K[n - 5:] = special
    inf = np.zeros(n, dtype=np.uint8)
    inf[[3, 20000, n - 2]] = 1
    exp = coracle.msm(c.id, (px, py, inf), K, 8)
    try:
        for w in (10, 11, 12, 13, 14, 15):
            ctx.set_msm_window(w)
            assert np.array_equal(ctx.msm_unchecked(c.id, (px, py, inf), K), exp), w
    finally:
        ctx.set_msm_window(0)


def test_msm_beyond_baseline_size(ctx, coracle):
    """2^23 points (8 copies of 2^20): sum_j <K_j, P> == <sum_j K_j, P>.  At this size the
    accumulation chunks are longer than at 2^20, whose result the closed form pins."""
    import torch

    c = o.SECP256K1
    n, reps = 1 << 20, 8
    px, py = seeded_points(coracle, c, n, 5550123, 7)
    rng = np.random.default_rng(23)
    K = rng.integers(0, 1 << 64, (n * reps, 4), dtype=np.uint64)
    K[:, 3] &= np.uint64((1 << 62) - 1)  # any value below the modulus is a valid element
    dx, dy, dk = _dev(px).repeat(reps, 1), _dev(py).repeat(reps, 1), _dev(K)
    big = ctx.msm_dev(c.id, dx.data_ptr(), dy.data_ptr(), None, dk.data_ptr(), n * reps)
    ksum = K[:n]
    for j in range(1, reps):
        ksum = ctx.vector_add(c.id, ksum, K[j * n:(j + 1) * n])
    ds = _dev(ksum)
    small = ctx.msm_dev(c.id, dx.data_ptr(), dy.data_ptr(), None, ds.data_ptr(), n)
    assert np.array_equal(big, small)
    del dx, dy, dk
    torch.cuda.empty_cache()


@pytest.mark.parametrize("c", CURVES, ids=ids)
def test_msm_resident_and_two_part(ctx, coracle, c):
    n = 3000
    px, py = seeded_points(coracle, c, n, 31, 17)
    ints, K = random_scalars(c, n, 9)
    inf = np.zeros(n, dtype=np.uint8)
    inf[5] = 1
    bases = ctx.register_bases(c.id, (px, py, inf))
    assert len(bases) == n
    for off, cnt in ((0, n), (100, 1024), (2999, 1), (7, 0)):
        exp = coracle.msm(c.id, (px[off:off + cnt], py[off:off + cnt], inf[off:off + cnt]), K[:cnt], 4)
        assert np.array_equal(bases.msm(K[:cnt], off, cnt), exp), (off, cnt)
    with pytest.raises(Exception):
        bases.msm(K[:10], n - 5, 10)
    # two sums over the same points in one pass (A and S of the range proofs: one vector of
    # {0, 1, -1}, one uniform), at sizes on either side of the one-launch front end
    _, K2 = random_scalars(c, n, 10)
    deg = o.pack_field([(0, 1, c.n - 1)[i % 3] for i in range(n)], c.n)
    for off, cnt, A, B in ((0, n, K, K2), (50, 2048, deg, K2), (0, 1, K, K2), (3, 0, K, K2)):
        pts = (px[off:off + cnt], py[off:off + cnt], inf[off:off + cnt])
        ga, gb = bases.msm_pair(A[:cnt], B[:cnt], off, cnt)
        assert np.array_equal(ga, coracle.msm(c.id, pts, A[:cnt], 4)), (off, cnt)
        assert np.array_equal(gb, coracle.msm(c.id, pts, B[:cnt], 4)), (off, cnt)
    # registered points plus uploaded ones = the concatenated MSM (Msm::execute, src/msm.rs:84-87)
    ex, ey = seeded_points(coracle, c, 37, 91, 7)
    einf = np.zeros(37, dtype=np.uint8)
    einf[3] = 1
    for off, cnt, k in ((0, n, 37), (10, 500, 1), (0, 0, 37), (5, 100, 0)):
        exp = coracle.msm(c.id, (np.concatenate([px[off:off + cnt], ex[:k]]), np.concatenate([py[off:off + cnt], ey[:k]]),
                                 np.concatenate([inf[off:off + cnt], einf[:k]])), np.concatenate([K[:cnt], K2[:k]]), 4)
        assert np.array_equal(bases.msm_plus(K[:cnt], (ex[:k], ey[:k], einf[:k]), K2[:k], off, cnt), exp), (off, cnt, k)
    bases.free()
    big = 12000  # 2 x 12000 points: beyond the one-launch front end
    bx, by = seeded_points(coracle, c, big, 5, 3)
    _, KA = random_scalars(c, big, 21)
    _, KB = random_scalars(c, big, 22)
    bases = ctx.register_bases(c.id, (bx, by, None))
    ga, gb = bases.msm_pair(KA, KB)
    assert np.array_equal(ga, coracle.msm(c.id, (bx, by, None), KA, 8))
    assert np.array_equal(gb, coracle.msm(c.id, (bx, by, None), KB, 8))
    bases.free()
    # two-part form = concatenated form (src/ipa/mod.rs:44-54)
    dx, dy, dk = _dev(px), _dev(py), _dev(K)
    h = 1000
    part0 = (dx.data_ptr() + 32 * h, dy.data_ptr() + 32 * h, None, dk.data_ptr(), h)
    part1 = (dx.data_ptr(), dy.data_ptr(), None, dk.data_ptr() + 32 * h, h)
    exp = coracle.msm(c.id, (np.concatenate([px[h:2 * h], px[:h]]), np.concatenate([py[h:2 * h], py[:h]]), None),
                      K[:2 * h], 4)
    assert np.array_equal(ctx.msm2_dev(c.id, part0, part1), exp)


@pytest.mark.parametrize("c", CURVES, ids=ids)
def test_msm_window_partials_combine(ctx, coracle, c):
    """the multi-GPU path on one GPU: per-shard window sums, gathered and combined"""
    import torch

    n = 4096
    px, py = seeded_points(coracle, c, n, 3, 5)
    ints, K = random_scalars(c, n, 4)
    dx, dy, dk = _dev(px), _dev(py), _dev(K)
    parts = 4
    m = n // parts
    gathered = torch.zeros((parts, 64 * 16), dtype=torch.int64, device="cuda")
    shape = None
    for r in range(parts):
        nw, cb = ctx.msm_windows_dev(c.id, dx.data_ptr() + 32 * m * r, dy.data_ptr() + 32 * m * r, None,
                                     dk.data_ptr() + 32 * m * r, m, gathered[r].data_ptr())
        assert shape in (None, (nw, cb))
        shape = (nw, cb)
    nw, cb = shape
    packed = gathered[:, :nw * 16].contiguous()
    got = ctx.msm_combine_windows_dev(c.id, packed.data_ptr(), parts, nw, cb)
    assert np.array_equal(got, coracle.msm(c.id, (px, py, None), K, 8))


# ---- curve vectors --------------------------------------------------------------------


def test_curve_vectors_golden(ctx):
    for fx in load_golden("vectors"):
        c = _curve(fx["curve"])
        rng = o.SplitMix64(fx["seed"])
        n = fx["n"]
        rng.scalars(n, c.n), rng.scalars(n, c.n)
        x, y = rng.field(c.n), rng.field(c.n)
        rng.field(c.n)
        pts = o.gen_points_arith(24, rng.field(c.n), rng.field(c.n), c)
        PX, PY, PI = o.pack_points(pts, c)
        X, Y = o.pack_field([x], c.n), o.pack_field([y], c.n)
        out = ctx.fold_generators(c.id, (PX[:12], PY[:12], PI[:12]), (PX[12:], PY[12:], PI[12:]), X, Y)
        assert o.unpack_points(*out, c) == [pt_from_json(p) for p in fx["fold_generators"]]
        ysc = o.pack_field(o.hs_prime_scalars(12, y, c.n), c.n)
        assert np.array_equal(ysc, ctx.powers(c.id, o.pack_field([pow(y, -1, c.n)], c.n), 12))
        out = ctx.scale_points(c.id, (PX[:12], PY[:12], PI[:12]), ysc)
        assert o.unpack_points(*out, c) == [pt_from_json(p) for p in fx["hs_prime"]]


@pytest.mark.parametrize("c", CURVES, ids=ids)
def test_curve_vectors_edge_and_large(ctx, coracle, c):
    n = 2048
    px, py = seeded_points(coracle, c, n, 41, 43)
    inf = np.zeros(n, dtype=np.uint8)
    inf[[0, 3, 1024 + 3, 1500]] = 1          # identity on the left, both sides, right only
    px[1030], py[1030] = px[6], py[6]        # L == R: doubling inside L + R
    py_neg = o.pack_field([(-v) % c.p for v in o.unpack_field(py[7:8], c.p)], c.p)
    px[1031], py[1031] = px[7], py_neg[0]    # L == -R
    ints, K = random_scalars(c, 4, 77)
    X, Y = K[0:1], K[1:2]
    L, R = (px[:1024], py[:1024], inf[:1024]), (px[1024:], py[1024:], inf[1024:])
    got = ctx.fold_generators(c.id, L, R, X, Y)
    exp = coracle.fold_generators(c.id, L, R, X, Y, 8)
    assert all(np.array_equal(g, e) for g, e in zip(got, exp))
    # x = y: both scalars equal; x = 0; y = 1
    one, zero = o.pack_field([1], c.n), o.pack_field([0], c.n)
    for xs, ys in ((X, X), (zero, one), (zero, zero)):
        got = ctx.fold_generators(c.id, (px[:64], py[:64], inf[:64]), (px[1024:1088], py[1024:1088], inf[1024:1088]), xs, ys)
        exp = coracle.fold_generators(c.id, (px[:64], py[:64], inf[:64]), (px[1024:1088], py[1024:1088], inf[1024:1088]), xs, ys, 8)
        assert all(np.array_equal(g, e) for g, e in zip(got, exp))
    _, S = random_scalars(c, 512, 78)
    S[3] = 0
    S[4] = o.pack_field([c.n - 1], c.n)[0]
    got = ctx.scale_points(c.id, (px[:512], py[:512], inf[:512]), S)
    exp = coracle.scale_points(c.id, (px[:512], py[:512], inf[:512]), S, 8)
    assert all(np.array_equal(g, e) for g, e in zip(got, exp))
    # normalize_batch: Jacobian (X*z^2, Y*z^3, z) -> the affine inputs
    zs = o.SplitMix64(5).scalars(16, c.p)
    xs, ys_ = o.unpack_field(px[:16], c.p), o.unpack_field(py[:16], c.p)
    jac = np.zeros((16, 12), dtype=np.uint64)
    for i in range(16):
        z = zs[i] if i != 2 else 0
        jac[i, 0:4] = o.pack_field([xs[i] * z * z], c.p)[0]
        jac[i, 4:8] = o.pack_field([ys_[i] * z * z * z], c.p)[0]
        jac[i, 8:12] = o.pack_field([z], c.p)[0]
    ox, oy, oi = ctx.normalize_batch(c.id, jac)
    assert oi[2] == 1 and oi.sum() == 1
    keep = [i for i in range(16) if i != 2]
    assert np.array_equal(ox[keep], px[:16][keep]) and np.array_equal(oy[keep], py[:16][keep])


# ---- IPA sessions ------------------------------------------------------------------------


def test_ipa_rounds_golden(ctx):
    for fx in load_golden("ipa"):
        c = _curve(fx["curve"])
        n, variant = fx["n"], fx["variant"]
        g, h, a, b, ch = ipa_inputs(c, n, fx["seed"])
        s = ctx.ipa_begin(c.id, variant, o.pack_points(g, c), o.pack_points(h, c) if variant == 0 else None,
                          o.pack_field(a, c.n), o.pack_field(b, c.n))
        for r, x in zip(fx["rounds"], ch):
            L, R, cl, cr = s.round_lr()
            assert xyz_to_affine(L, c) == pt_from_json(r["L"]) and xyz_to_affine(R, c) == pt_from_json(r["R"])
            assert hex(o.unpack_field(cl, c.n)[0]) == r["cL"] and hex(o.unpack_field(cr, c.n)[0]) == r["cR"]
            s.round_fold(o.pack_field([x], c.n), o.pack_field([pow(x, -1, c.n)], c.n))
        assert len(s) == 1
        G, H, A, B = s.read()
        assert o.unpack_points(*G, c) == [pt_from_json(fx["final"]["g"])]
        assert [hex(v) for v in o.unpack_field(A, c.n)] == [fx["final"]["a"]]
        assert [hex(v) for v in o.unpack_field(B, c.n)] == [fx["final"]["b"]]
        if variant == 0:
            assert o.unpack_points(*H, c) == [pt_from_json(fx["final"]["h"])]
        with pytest.raises(Exception):
            s.round_lr()  # already folded to length 1: BP_E_STATE
        s.free()


@pytest.mark.parametrize("variant", [0, 1])
def test_ipa_final_generator_property(ctx, coracle, variant):
    """n = 2^11: the folded generator equals MSM(gs, s-vector) (src/poly_commit/mod.rs:108-113)
    and every round's L/R match the C oracle's MSM of the same slices"""
    c = o.SECP256K1
    n, k = 1 << 11, 11
    gx, gy = seeded_points(coracle, c, n, 7, 11)
    hx, hy = seeded_points(coracle, c, n, 70001, 13)
    _, A = random_scalars(c, n, 1)
    _, B = random_scalars(c, n, 2)
    _, CH = random_scalars(c, k, 3)
    CHI = ctx.batch_inversion(c.id, CH)
    s = ctx.ipa_begin(c.id, variant, (gx, gy, None), (hx, hy, None) if variant == 0 else None, A, B)
    g_cur, h_cur, a_cur, b_cur = (gx, gy), (hx, hy), A, B
    for j in range(k):
        L, R, cl, cr = s.round_lr()
        if j < 3:  # spot-check the first rounds against the C oracle on the host copies
            (g2, h2, a_cur, b_cur) = s.read() if j else ((gx, gy, None), (hx, hy, None), A, B)
            m = len(a_cur) // 2
            if variant == 0:
                bases = (np.concatenate([g2[0][m:], h2[0][:m]]), np.concatenate([g2[1][m:], h2[1][:m]]), None)
                sc = np.concatenate([a_cur[:m], b_cur[m:]])
            else:
                bases, sc = (g2[0][m:], g2[1][m:], None), a_cur[:m]
            assert np.array_equal(L, coracle.msm(c.id, bases, sc, 8))
            assert np.array_equal(cl, coracle.inner_product(c.id, a_cur[:m], b_cur[m:]))
            assert np.array_equal(cr, coracle.inner_product(c.id, a_cur[m:], b_cur[:m]))
        s.round_fold(CH[j:j + 1], CHI[j:j + 1])
    G, H, Af, Bf = s.read()
    ss = ctx.s_vector(c.id, CH, CHI)
    exp = ctx.msm_unchecked(c.id, (gx, gy, None), ss)
    ex, ey, _ = ctx.normalize_batch(c.id, exp.reshape(1, 12))
    assert np.array_equal(G[0], ex) and np.array_equal(G[1], ey)
    assert np.array_equal(exp, coracle.msm(c.id, (gx, gy, None), ss, 8))
    s.free()


@pytest.mark.parametrize("c", CURVES, ids=ids)
@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("mode", ["point", "index", "tables"])
def test_ipa_resident_scaled_with_u(ctx, c, variant, mode):
    """generators from a registered base set, H scaled by y^-i, u folded into L and R:
    every round equals the oracle's ipa round over (gs, hs' = y^-i hs) plus c*u
    (src/range/aggregate.rs:148-166, src/ipa/mod.rs:40-73)"""
    from bulletproofs_amd import BpError

    n, k = 32, 5
    g, h, a, b, ch = ipa_inputs(c, n, 4242 + variant)
    u0, factor = o.scalar_mul(987654321, c.g, c), 0x1234567890ABCDEF1234567890ABCDEF % c.n
    u = o.scalar_mul(factor, u0, c)  # mode "point": u itself; otherwise factor * bases[2n]
    yinv = pow(31337, -1, c.n)
    scale = [pow(yinv, i, c.n) for i in range(n)]
    hsp = o.scale_points(h, scale, c) if variant == 0 else None
    bases = ctx.register_bases(c.id, o.pack_points(g + h + [u0], c))
    if mode == "tables":
        bases.precompute()  # window tables: same results through one bucket set per sum
    s = ctx.ipa_begin_resident(bases, variant, 0, n, o.pack_field(a, c.n), o.pack_field(b, c.n))
    if variant == 0:
        s.scale_h(o.pack_field(scale, c.n))
    else:
        with pytest.raises(BpError):
            s.scale_h(o.pack_field(scale, c.n))
    if mode == "point":
        ux, uy, _ = o.pack_points([u], c)
        s.set_u(np.concatenate([ux[0], uy[0]]))
    else:
        s.set_u_resident(2 * n, o.pack_field([factor], c.n))
    gs_cur, hs_cur, a_cur, b_cur = list(g), (list(hsp) if hsp else list(h)), list(a), list(b)
    for j in range(k):
        L, R, cl, cr = s.round_lr()
        eL, eR, ecl, ecr = o.ipa_round_lr(variant, gs_cur, hs_cur, a_cur, b_cur, c)
        eL = o.add(eL, o.scalar_mul(ecl, u, c), c)
        eR = o.add(eR, o.scalar_mul(ecr, u, c), c)
        assert xyz_to_affine(L, c) == eL and xyz_to_affine(R, c) == eR, j
        assert o.unpack_field(cl, c.n)[0] == ecl and o.unpack_field(cr, c.n)[0] == ecr
        x = ch[j]
        if j == 1 and variant == 0:
            with pytest.raises(BpError):
                s.scale_h(o.pack_field(scale, c.n))  # only before the first fold
        s.round_fold(o.pack_field([x], c.n), o.pack_field([pow(x, -1, c.n)], c.n))
        gs_cur, hs_cur, a_cur, b_cur = o.ipa_round_fold(variant, gs_cur, hs_cur, a_cur, b_cur, x, c)
    G, H, A, B = s.read()
    assert o.unpack_points(*G, c) == gs_cur
    if variant == 0:
        assert o.unpack_points(*H, c) == hs_cur
    assert o.unpack_field(A, c.n) == a_cur and o.unpack_field(B, c.n) == b_cur
    s.free()
    bases.free()


@pytest.mark.parametrize("variant", [0, 1])
def test_ipa_mid_size_rounds_with_u(ctx, coracle, variant):
    """n = 2^14 (the sliced counting sort with two output groups and the extra point u):
    L and R of the first rounds equal the C oracle's MSM over the folded generators plus
    c*u, and the final generators equal <s-vector, gs>, <s-vector^-1, hs>"""
    _ipa_rounds_with_u(ctx, coracle, variant, 14, 2)


def test_ipa_large_first_round_with_u(ctx, coracle):
    """n = 2^17: 2^18 + 2 points in two groups, the largest input of the sliced counting sort;
    the first round's L and R equal the C oracle's MSMs plus c*u"""
    _ipa_rounds_with_u(ctx, coracle, 0, 17, 1, full=False)


@pytest.mark.parametrize("c", CURVES, ids=ids)
@pytest.mark.parametrize("variant", [0, 1])
def test_ipa_first_round_16_bit_windows(ctx, coracle, c, variant):
    """n = 2^18 (variant 1: 2^19): two groups of 2^18 points and the extra point u take the
    16-bit-window pipeline with the two-level counting sort; L and R of the first round equal
    the C oracle's MSMs plus c*u"""
    _ipa_rounds_with_u(ctx, coracle, variant, 18 + variant, 1, full=False, c=c)


def _ipa_rounds_with_u(ctx, coracle, variant, k, checked_rounds, full=True, c=o.SECP256K1):
    n = 1 << k
    gx, gy = seeded_points(coracle, c, n, 17, 19)
    hx, hy = seeded_points(coracle, c, n, 900001, 23)
    _, A = random_scalars(c, n, 41)
    _, B = random_scalars(c, n, 42)
    _, CH = random_scalars(c, k, 43)
    CHI = ctx.batch_inversion(c.id, CH)
    u = o.scalar_mul(1234567, c.g, c)
    ux, uy, _ = o.pack_points([u], c)
    bases = ctx.register_bases(c.id, (np.concatenate([gx, hx]), np.concatenate([gy, hy]), None))
    s = ctx.ipa_begin_resident(bases, variant, 0, n, A, B)
    s.set_u(np.concatenate([ux[0], uy[0]]))
    for j in range(k if full else checked_rounds):
        L, R, cl, cr = s.round_lr()
        if j < checked_rounds:
            (g2, h2, a_cur, b_cur) = s.read() if j else ((gx, gy, None), (hx, hy, None), A, B)
            m = len(a_cur) // 2
            if variant == 0:
                lb = (np.concatenate([g2[0][m:], h2[0][:m]]), np.concatenate([g2[1][m:], h2[1][:m]]), None)
                ls = np.concatenate([a_cur[:m], b_cur[m:]])
                rb = (np.concatenate([g2[0][:m], h2[0][m:]]), np.concatenate([g2[1][:m], h2[1][m:]]), None)
                rs = np.concatenate([a_cur[m:], b_cur[:m]])
            else:
                lb, ls = (g2[0][m:], g2[1][m:], None), a_cur[:m]
                rb, rs = (g2[0][:m], g2[1][:m], None), a_cur[m:]
            ecl = coracle.inner_product(c.id, a_cur[:m], b_cur[m:])
            ecr = coracle.inner_product(c.id, a_cur[m:], b_cur[:m])
            assert np.array_equal(cl, ecl) and np.array_equal(cr, ecr)
            eL = o.add(xyz_to_affine(coracle.msm(c.id, lb, ls, 8), c), o.scalar_mul(o.unpack_field(ecl, c.n)[0], u, c), c)
            eR = o.add(xyz_to_affine(coracle.msm(c.id, rb, rs, 8), c), o.scalar_mul(o.unpack_field(ecr, c.n)[0], u, c), c)
            assert xyz_to_affine(L, c) == eL and xyz_to_affine(R, c) == eR, j
        s.round_fold(CH[j:j + 1], CHI[j:j + 1])
    if not full:
        s.free()
        bases.free()
        return
    G, H, Af, Bf = s.read()
    ss = ctx.s_vector(c.id, CH, CHI)
    eg = xyz_to_affine(coracle.msm(c.id, (gx, gy, None), ss, 8), c)
    assert o.unpack_points(*G, c) == [eg]
    if variant == 0:
        ssi = ctx.s_vector(c.id, CHI, CH)
        eh = xyz_to_affine(coracle.msm(c.id, (hx, hy, None), ssi, 8), c)
        assert o.unpack_points(*H, c) == [eh]
    s.free()
    bases.free()


# ---- error behaviour -----------------------------------------------------------------------


def test_error_codes(ctx):
    from bulletproofs_amd import BpError

    f = np.zeros((4, 4), dtype=np.uint64)
    with pytest.raises(BpError) as e:
        ctx.inner_product(7, f, f)
    assert e.value.code == -2
    with pytest.raises(BpError) as e:
        ctx.ipa_begin(0, 0, (f[:3], f[:3], None), (f[:3], f[:3], None), f[:3], f[:3])  # n not a power of two
    assert e.value.code == -1
    with pytest.raises(BpError) as e:
        ctx.ipa_begin(0, 0, (f, f, None), None, f, f)  # bulletproof variant needs hs
    assert e.value.code == -1
    with pytest.raises(BpError):
        ctx.set_msm_window(1)


@pytest.mark.parametrize("c", CURVES, ids=ids)
@pytest.mark.parametrize("n", [1, 5, 300, 8192, 20000])
def test_msm_window_tables(ctx, coracle, c, n):
    """bp_bases_precompute: the MSM over a base set with window tables (one bucket set for all
    windows, no doublings) equals the C oracle's, for whole sets and sub-ranges, with
    identity bases and zero scalars"""
    px, py = seeded_points(coracle, c, n, 555 + n, 13)
    _, K = random_scalars(c, n, 7 * n)
    inf = np.zeros(n, dtype=np.uint8)
    if n > 4:
        inf[3] = 1
        K[1] = 0
    b = ctx.register_bases(c.id, (px, py, inf)).precompute()
    assert np.array_equal(b.msm(K), coracle.msm(c.id, (px, py, inf), K, 8))
    if n > 10:
        assert np.array_equal(b.msm(K[2:9], offset=2, n=7), coracle.msm(c.id, (px[2:9], py[2:9], inf[2:9]), K[2:9], 8))
    b.free()


@pytest.mark.parametrize("n", [1, 5, 64])
def test_window_tables_on_a_fresh_context(coracle, n):
    """the table path's per-slice histograms need a buffer that does not shrink with n: a context
    whose workspace has never grown (no earlier large MSM) runs register -> precompute -> msm, and a
    two-sum IPA session of 4 generators over the same tables, against the oracle"""
    import bulletproofs_amd as bp

    c = o.SECP256K1
    fresh = bp.Context(0)
    try:
        px, py = seeded_points(coracle, c, max(n, 8), 91 + n, 17)
        _, K = random_scalars(c, n, 3 * n + 1)
        b = fresh.register_bases(c.id, (px, py, None)).precompute()
        assert np.array_equal(b.msm(K, n=n), coracle.msm(c.id, (px[:n], py[:n], None), K, 4))
        ints_a, A = random_scalars(c, 4, 11)
        ints_b, B = random_scalars(c, 4, 12)
        s = fresh.ipa_begin_resident(b, bp.IPA_BULLETPROOF, 0, 4, A, B)
        L, R, cl, cr = s.round_lr()
        pts = o.unpack_points(px[:8], py[:8], np.zeros(8, dtype=np.uint8), c)
        eL, eR, ecl, ecr = o.ipa_round_lr(o.BULLETPROOF, pts[:4], pts[4:8], ints_a, ints_b, c)
        assert xyz_to_affine(L, c) == eL and xyz_to_affine(R, c) == eR
        assert o.unpack_field(cl, c.n)[0] == ecl and o.unpack_field(cr, c.n)[0] == ecr
        s.free()
        b.free()
    finally:
        fresh.close()


@pytest.mark.parametrize("c", CURVES, ids=ids)
@pytest.mark.parametrize("n", [1, 8, 1024])
def test_ipa_session_with_powers_built_on_the_device(ctx, c, n):
    """bp_ipa_begin_resident_powers: b_i = x^i (src/poly_commit/mod.rs:65) built on the device gives the
    same rounds as the session that was handed the powers: first L, R, c_l, c_r against the oracle, and
    the vector b itself read back after one fold (at n = 1: as given)"""
    import bulletproofs_amd as bp

    g, _, a, _, ch = ipa_inputs(c, n, 77 + n)
    x = 0xDEADBEEFCAFEF00D123456789ABCDEF0FEDCBA9876543210 % c.n
    b = [pow(x, i, c.n) for i in range(n)]
    bases = ctx.register_bases(c.id, o.pack_points(g, c))
    s = ctx.ipa_begin_resident_powers(bases, bp.IPA_POLYCOMMIT, 0, 0, o.pack_field(a, c.n), o.pack_field([x], c.n))
    if n > 1:
        L, R, cl, cr = s.round_lr()
        eL, eR, ecl, ecr = o.ipa_round_lr(1, g, g, a, b, c)
        assert xyz_to_affine(L, c) == eL and xyz_to_affine(R, c) == eR
        assert o.unpack_field(cl, c.n)[0] == ecl and o.unpack_field(cr, c.n)[0] == ecr
        s.round_fold(o.pack_field([ch[0]], c.n), o.pack_field([pow(ch[0], -1, c.n)], c.n))
        _, _, a2, b2 = o.ipa_round_fold(1, g, g, a, b, ch[0], c)
    else:
        a2, b2 = a, b
    if n <= 8:
        _, _, A, B = s.read()
        assert o.unpack_field(A, c.n) == a2 and o.unpack_field(B, c.n) == b2
    s.free()


@pytest.mark.parametrize("c", CURVES, ids=ids)
@pytest.mark.parametrize("layout", ["gs|hs|u", "hs|gap|gs|gap|u"])
def test_ipa_verify_msm_against_the_oracle(ctx, c, layout):
    """bp_ipa_verify_msm: sum a s_i G_i + b s_i^-1 H_i + u_scalar U + extras (src/ipa/mod.rs:125-190) with the
    s-vectors built on the device, against the oracle's s-vector and naive MSM; also with gs, hs, u apart"""
    k, n = 4, 16
    rng = o.SplitMix64(0x1FA + c.id + len(layout))
    pts = o.gen_points_arith(2 * n + 8, rng.field(c.n), rng.field(c.n), c)
    g, h, u, extra_pts = pts[:n], pts[n:2 * n], pts[2 * n], pts[2 * n + 1:2 * n + 4]
    pad = pts[2 * n + 4:2 * n + 8]
    if layout == "gs|hs|u":
        all_pts, g_off, h_off, u_idx = g + h + [u], 0, n, 2 * n
    else:
        all_pts, g_off, h_off, u_idx = h + pad[:2] + g + pad[2:] + [u], n + 2, 0, 2 * n + 4
    ch = [rng.field(c.n) or 1 for _ in range(k)]
    chi = [pow(x, -1, c.n) for x in ch]
    a, b, us = rng.field(c.n), rng.field(c.n), rng.field(c.n)
    es = rng.scalars(3, c.n)
    bases = ctx.register_bases(c.id, o.pack_points(all_pts, c))
    got = bases.ipa_verify_msm(g_off, h_off, u_idx, o.pack_field(ch, c.n), o.pack_field(chi, c.n), o.pack_field([a], c.n),
                               o.pack_field([b], c.n), o.pack_field([us], c.n), o.pack_points(extra_pts, c),
                               o.pack_field(es, c.n))
    s = o.s_vector(ch, c.n)
    si = o.s_vector(chi, c.n)
    assert si == [pow(v, -1, c.n) for v in s]
    scalars = [a * v % c.n for v in s] + [b * v % c.n for v in si] + [us] + es
    assert xyz_to_affine(got, c) == o.msm_naive(g + h + [u] + extra_pts, scalars, c)
    bases.free()


@pytest.mark.parametrize("c", CURVES, ids=ids)
def test_range_agg_verify_msm_against_the_two_step_form(ctx, c):
    """bp_range_agg_verify_msm equals bp_range_agg_verify_scalars (checked against the big-integer restatement
    elsewhere) followed by the oracle's MSM over gs | hs | u and the extra terms"""
    from bulletproofs_amd._lib import lib

    m, n_bits, k = 2, 8, 4
    n = m * n_bits
    rng = o.SplitMix64(0x2A66 + c.id)
    pts = o.gen_points_arith(2 * n + 4, rng.field(c.n), rng.field(c.n), c)
    crs, extra_pts = pts[:2 * n + 1], pts[2 * n + 1:]
    ch = [rng.field(c.n) or 1 for _ in range(k)]
    chi = [pow(x, -1, c.n) for x in ch]
    a, b, y_inv, z, us = (rng.field(c.n) for _ in range(5))
    es = rng.scalars(3, c.n)
    F = lambda v: o.pack_field(v if isinstance(v, list) else [v], c.n)  # noqa: E731
    gsc, hsc = ctx.range_agg_verify_scalars(c.id, F(ch), F(chi), F(a), F(b), F(y_inv), F(z), m, n_bits)
    bases = ctx.register_bases(c.id, o.pack_points(crs, c))
    ex, ey, ei = o.pack_points(extra_pts, c)
    out = np.zeros(12, dtype=np.uint64)
    CH, CHI, A, B, YI, Z, US, ES = F(ch), F(chi), F(a), F(b), F(y_inv), F(z), F(us), F(es)
    rc = lib.bp_range_agg_verify_msm(ctx.h, bases.h, CH.ctypes.data, CHI.ctypes.data, k, A.ctypes.data, B.ctypes.data,
                                     YI.ctypes.data, Z.ctypes.data, m, n_bits, US.ctypes.data, ex.ctypes.data, ey.ctypes.data,
                                     ei.ctypes.data, ES.ctypes.data, 3, out.ctypes.data)
    assert rc == 0
    scalars = o.unpack_field(gsc, c.n) + o.unpack_field(hsc, c.n) + [us] + es
    assert xyz_to_affine(out, c) == o.msm_naive(crs + extra_pts, scalars, c)
    bases.free()
