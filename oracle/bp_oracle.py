"""CPU oracle (TEST INFRASTRUCTURE ONLY) for the Bulletproofs hot path.

Pure-Python big-integer restatement of the arithmetic that the reference
(l-adic/bulletproofs, Rust) reaches through arkworks: prime-field arithmetic in
Montgomery form, short-Weierstrass curve arithmetic, multi-scalar
multiplication, the inner-product-argument vector kernels and round loops.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product path (bulletproofs_amd/) never does.

PARITY UNPINNED by reference byte vectors: the reference holds no golden
vectors for this path (all of its tests are prove->verify round trips on OsRng
inputs) and its arithmetic lives in ark-ff / ark-ec 0.5.0, registry crates that
are not vendored.  What pins this oracle instead:
  * the six literal expectations of src/vector_ops.rs:238-314,
  * the SEC2 secp256k1 generator multiples (G, 2G, (n-1)G = -G, nG = O),
  * the algebraic properties the reference's own tests assert
    (src/ipa/types.rs:187-203 additivity, src/poly_commit/mod.rs:324-346
    HPoly evaluate == <coeffs, powers>, src/poly_commit/mod.rs:108-113
    final generator == MSM(gs, s-vector)).
Every output is a canonical value (a reduced residue or an affine point), so
"equal to the mathematics" is the same statement as "equal to arkworks".

Data convention (matches the arkworks in-memory form the C ABI takes):
  field element  = 4 little-endian u64 limbs of a*2^256 mod p  (Montgomery form)
  affine point   = (x limbs, y limbs, infinity byte)
"""
from __future__ import annotations

import numpy as np

R_BITS = 256
R = 1 << R_BITS
MASK64 = (1 << 64) - 1

# ---------------------------------------------------------------------------
# Curves.  secp256k1: SEC2.  Vesta: Pasta curves (y^2 = x^3 + 5, generator (-1, 2)).
# ---------------------------------------------------------------------------


class Curve:
    def __init__(self, cid, name, p, n, b, gx, gy):
        self.id = cid
        self.name = name
        self.p = p  # base field modulus  (coordinates)
        self.n = n  # scalar field modulus (group order)
        self.b = b
        self.g = (gx, gy)


SECP256K1 = Curve(
    0,
    "secp256k1",
    (1 << 256) - (1 << 32) - 977,
    0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141,
    7,
    0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798,
    0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8,
)
_VESTA_P = 0x40000000000000000000000000000000224698FC0994A8DD8C46EB2100000001
_VESTA_N = 0x40000000000000000000000000000000224698FC094CF91B992D30ED00000001
VESTA = Curve(1, "vesta", _VESTA_P, _VESTA_N, 5, _VESTA_P - 1, 2)
CURVES = {0: SECP256K1, 1: VESTA, "secp256k1": SECP256K1, "vesta": VESTA}

# ---------------------------------------------------------------------------
# Montgomery encode/decode and limb packing
# ---------------------------------------------------------------------------


def to_mont(a: int, p: int) -> int:
    return (a << R_BITS) % p


def from_mont(a: int, p: int) -> int:
    return (a * pow(R, -1, p)) % p


def int_to_limbs(a: int):
    return [(a >> (64 * i)) & MASK64 for i in range(4)]


def limbs_to_int(l) -> int:
    return int(l[0]) | (int(l[1]) << 64) | (int(l[2]) << 128) | (int(l[3]) << 192)


def pack_field(vals, p: int) -> np.ndarray:
    """list of canonical ints -> (n,4) uint64 array of Montgomery limbs."""
    out = np.zeros((len(vals), 4), dtype=np.uint64)
    for i, v in enumerate(vals):
        m = to_mont(v % p, p)
        for k in range(4):
            out[i, k] = (m >> (64 * k)) & MASK64
    return out


def unpack_field(arr: np.ndarray, p: int):
    arr = np.asarray(arr, dtype=np.uint64).reshape(-1, 4)
    rinv = pow(R, -1, p)
    return [(limbs_to_int(row) * rinv) % p for row in arr]


def pack_points(pts, curve: Curve):
    """list of affine points (None = identity) -> (x (n,4), y (n,4), inf (n,)) arrays."""
    n = len(pts)
    xs = [0 if P is None else P[0] for P in pts]
    ys = [0 if P is None else P[1] for P in pts]
    inf = np.array([1 if P is None else 0 for P in pts], dtype=np.uint8).reshape(n)
    return pack_field(xs, curve.p), pack_field(ys, curve.p), inf


def unpack_points(x, y, inf, curve: Curve):
    xs = unpack_field(x, curve.p)
    ys = unpack_field(y, curve.p)
    inf = np.asarray(inf).reshape(-1)
    return [None if inf[i] else (xs[i], ys[i]) for i in range(len(xs))]


# ---------------------------------------------------------------------------
# Seeded input generator (SplitMix64), shared by the C oracle and the benches
# ---------------------------------------------------------------------------


class SplitMix64:
    def __init__(self, seed: int):
        self.s = seed & MASK64

    def next(self) -> int:
        self.s = (self.s + 0x9E3779B97F4A7C15) & MASK64
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
        return z ^ (z >> 31)

    def field(self, mod: int) -> int:
        """4 consecutive outputs as little-endian limbs, rejected if >= mod."""
        while True:
            v = 0
            for k in range(4):
                v |= self.next() << (64 * k)
            if v < mod:
                return v

    def scalars(self, count: int, mod: int):
        return [self.field(mod) for _ in range(count)]


# ---------------------------------------------------------------------------
# Curve arithmetic (a = 0 short Weierstrass).  Affine points are (x, y) tuples,
# identity is None.  Jacobian (X, Y, Z), identity has Z == 0.
# ---------------------------------------------------------------------------


def on_curve(P, c: Curve) -> bool:
    if P is None:
        return True
    x, y = P
    return (y * y - x * x * x - c.b) % c.p == 0


def neg(P, c: Curve):
    return None if P is None else (P[0], (-P[1]) % c.p)


def add(P, Q, c: Curve):
    if P is None:
        return Q
    if Q is None:
        return P
    p = c.p
    x1, y1 = P
    x2, y2 = Q
    if x1 == x2:
        if (y1 + y2) % p == 0:
            return None
        lam = (3 * x1 * x1) * pow(2 * y1, -1, p) % p
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, p) % p
    x3 = (lam * lam - x1 - x2) % p
    y3 = (lam * (x1 - x3) - y1) % p
    return (x3, y3)


def jac_double(P, p):
    X, Y, Z = P
    if Z == 0:
        return P
    A = X * X % p
    B = Y * Y % p
    C = B * B % p
    D = 2 * ((X + B) * (X + B) - A - C) % p
    E = 3 * A % p
    F = E * E % p
    X3 = (F - 2 * D) % p
    Y3 = (E * (D - X3) - 8 * C) % p
    Z3 = 2 * Y * Z % p
    return (X3, Y3, Z3)


def jac_add_affine(P, Q, p):
    """Jacobian P + affine Q (Q not identity)."""
    X1, Y1, Z1 = P
    if Z1 == 0:
        return (Q[0], Q[1], 1)
    Z1Z1 = Z1 * Z1 % p
    U2 = Q[0] * Z1Z1 % p
    S2 = Q[1] * Z1 * Z1Z1 % p
    if U2 == X1:
        if S2 == Y1:
            return jac_double(P, p)
        return (1, 1, 0)
    H = (U2 - X1) % p
    HH = H * H % p
    I = 4 * HH % p
    J = H * I % p
    r = 2 * (S2 - Y1) % p
    V = X1 * I % p
    X3 = (r * r - J - 2 * V) % p
    Y3 = (r * (V - X3) - 2 * Y1 * J) % p
    Z3 = ((Z1 + H) * (Z1 + H) - Z1Z1 - HH) % p
    return (X3, Y3, Z3)


def jac_to_affine(P, p):
    X, Y, Z = P
    if Z == 0:
        return None
    zi = pow(Z, -1, p)
    zi2 = zi * zi % p
    return (X * zi2 % p, Y * zi2 * zi % p)


def scalar_mul(k: int, P, c: Curve):
    k %= c.n
    if P is None or k == 0:
        return None
    acc = (1, 1, 0)
    for bit in bin(k)[2:]:
        acc = jac_double(acc, c.p)
        if bit == "1":
            acc = jac_add_affine(acc, P, c.p)
    return jac_to_affine(acc, c.p)


def msm_naive(points, scalars, c: Curve):
    """sum_i k_i * P_i, the group element G::msm_unchecked returns
    (reference call sites: src/msm.rs:84-87, src/ipa/mod.rs:54,70,206)."""
    acc = None
    for P, k in zip(points, scalars):
        acc = add(acc, scalar_mul(k, P, c), c)
    return acc


def msm_pippenger(points, scalars, c: Curve, window: int = 8):
    """Bucket method (published Pippenger algorithm that ark-ec 0.5.0's
    VariableBaseMSM implements); same group element as msm_naive."""
    n_win = (256 + window - 1) // window
    total = None
    for w in reversed(range(n_win)):
        for _ in range(window):
            total = add(total, total, c)
        buckets = [None] * (1 << window)
        for P, k in zip(points, scalars):
            d = ((k % c.n) >> (w * window)) & ((1 << window) - 1)
            if d and P is not None:
                buckets[d] = add(buckets[d], P, c)
        run = None
        acc = None
        for d in range((1 << window) - 1, 0, -1):
            run = add(run, buckets[d], c)
            acc = add(acc, run, c)
        total = add(total, acc, c)
    return total


def gen_points_arith(count: int, r0: int, q: int, c: Curve):
    """P_i = (r0 + i*q)*G, generated incrementally (P_{i+1} = P_i + q*G).
    Gives any-size MSM a closed-form answer: sum k_i P_i = (sum k_i (r0+i q)) G."""
    P = scalar_mul(r0, c.g, c)
    Q = scalar_mul(q, c.g, c)
    out = []
    for _ in range(count):
        out.append(P)
        P = add(P, Q, c)
    return out


def msm_closed_form(scalars, r0: int, q: int, c: Curve):
    s = 0
    for i, k in enumerate(scalars):
        s = (s + k * (r0 + i * q)) % c.n
    return scalar_mul(s, c.g, c)


def gen_points_hashed(count: int, rng: SplitMix64, c: Curve):
    """'Unstructured' points: try-and-increment on a seeded x (p = 3 mod 4 for
    secp256k1; Tonelli-Shanks via pow for Vesta is avoided by using multiples)."""
    out = []
    p = c.p
    while len(out) < count:
        x = rng.field(p)
        while True:
            rhs = (x * x * x + c.b) % p
            if p % 4 == 3:
                y = pow(rhs, (p + 1) // 4, p)
                if y * y % p == rhs:
                    break
            else:
                y = _sqrt_ts(rhs, p)
                if y is not None:
                    break
            x = (x + 1) % p
        if rng.next() & 1:
            y = p - y
        out.append((x, y))
    return out


def _sqrt_ts(a, p):
    if a == 0:
        return 0
    if pow(a, (p - 1) // 2, p) != 1:
        return None
    q, s = p - 1, 0
    while q % 2 == 0:
        q //= 2
        s += 1
    z = 2
    while pow(z, (p - 1) // 2, p) != p - 1:
        z += 1
    m, cc, t, r = s, pow(z, q, p), pow(a, q, p), pow(a, (q + 1) // 2, p)
    while t != 1:
        i, t2 = 0, t
        while t2 != 1:
            t2 = t2 * t2 % p
            i += 1
        b = pow(cc, 1 << (m - i - 1), p)
        m, cc, t, r = i, b * b % p, t * b * b % p, r * b % p
    return r


# ---------------------------------------------------------------------------
# Vector kernels (src/vector_ops.rs)
# ---------------------------------------------------------------------------


def inner_product(a, b, mod: int) -> int:
    """src/vector_ops.rs:159-171: left fold of a_i*b_i, zip to the shorter."""
    acc = 0
    for x, y in zip(a, b):
        acc = (acc + x * y) % mod
    return acc


def vsum(a, mod: int) -> int:
    """src/vector_ops.rs:173-179."""
    acc = 0
    for x in a:
        acc = (acc + x) % mod
    return acc


def vector_add(a, b, mod):  # src/vector_ops.rs:5-24
    return [(x + y) % mod for x, y in zip(a, b)]


def vector_sub(a, b, mod):  # src/vector_ops.rs:26-45
    return [(x - y) % mod for x, y in zip(a, b)]


def hadamard(a, b, mod):  # src/vector_ops.rs:47-66, 149-155
    return [(x * y) % mod for x, y in zip(a, b)]


def scale(a, s, mod):  # src/vector_ops.rs:68-83
    return [(s * x) % mod for x in a]


def fold_scalars(left, right, x, y, mod):
    """src/ipa/mod.rs:229-234: out_i = l_i*x + r_i*y."""
    return [(l * x + r * y) % mod for l, r in zip(left, right)]


def fold_generators(left, right, x, y, c: Curve):
    """src/ipa/mod.rs:215-227: out_i = x*L_i + y*R_i, normalised to affine."""
    return [add(scalar_mul(x, l, c), scalar_mul(y, r, c), c) for l, r in zip(left, right)]


def scale_points(points, scalars, c: Curve):
    """h'_i = s_i * h_i -> affine  (src/range/mod.rs:136-141, src/range/aggregate.rs:148-153)."""
    return [scalar_mul(s, P, c) for P, s in zip(points, scalars)]


def hs_prime_scalars(count: int, y: int, mod: int):
    """src/range/utils.rs:22-29: 1, y^-1, y^-2, ..."""
    yi = pow(y, -1, mod)
    out, cur = [], 1
    for _ in range(count):
        out.append(cur)
        cur = cur * yi % mod
    return out


def batch_inversion(v, mod):
    """ark-ff batch_inversion semantics: zeros stay zero, others inverted."""
    return [0 if x % mod == 0 else pow(x, -1, mod) for x in v]


def s_vector(challenges, mod):
    """src/ipa/mod.rs:134-148 and src/poly_commit/mod.rs:194-219:
    s_i = prod_j (bit j of i ? x[k-1-j] : x[k-1-j]^-1),  i < 2^k."""
    k = len(challenges)
    inv = [pow(x, -1, mod) for x in challenges]
    out = []
    for i in range(1 << k):
        acc = 1
        for j in range(k):
            idx = k - j - 1
            acc = acc * (challenges[idx] if (i >> j) & 1 else inv[idx]) % mod
        out.append(acc)
    return out


def hpoly_evaluate(ui, x, mod):
    """src/poly_commit/mod.rs:181-192."""
    acc = 1
    for i, u in enumerate(reversed(ui)):
        acc = acc * (pow(u, -1, mod) + u * pow(x, 1 << i, mod)) % mod
    return acc


def powers(x, count, mod):
    out, cur = [], 1
    for _ in range(count):
        out.append(cur)
        cur = cur * x % mod
    return out


# ---------------------------------------------------------------------------
# Msm accumulator (src/msm.rs)
# ---------------------------------------------------------------------------


class Msm:
    """src/msm.rs:8-88: map base -> scalar; duplicate bases merge by adding scalars."""

    def __init__(self, c: Curve):
        self.c = c
        self.m = {}

    def upsert(self, base, scalar):  # :33-38
        self.m[base] = (self.m.get(base, 0) + scalar) % self.c.n

    def upsert_batch(self, items):  # :41-48
        for b, s in items:
            self.upsert(b, s)

    def scale(self, s):  # :50-54
        for k in self.m:
            self.m[k] = self.m[k] * s % self.c.n

    def scale_elems(self, items):  # :62-71
        for b, f in items:
            if b in self.m:
                self.m[b] = self.m[b] * f % self.c.n

    def batch(self, rhs):  # :73-77
        for b, s in rhs.m.items():
            self.upsert(b, s)

    def bases_and_scalars(self):  # :79-82
        return list(self.m.keys()), list(self.m.values())

    def execute(self):  # :84-87
        b, s = self.bases_and_scalars()
        return msm_naive(b, s, self.c) if len(b) < 64 else msm_pippenger(b, s, self.c)


def verify_batch_aux(msms, alpha, c: Curve):
    """src/msm.rs:91-115 with the random alpha supplied by the caller."""
    acc = None
    cur = 1
    for m in msms:
        m.scale(cur)
        if acc is None:
            acc = m
        else:
            acc.batch(m)
        cur = cur * alpha % c.n
    return acc.execute() is None


# ---------------------------------------------------------------------------
# IPA round loops with caller-supplied challenges (transcript is out of scope)
# ---------------------------------------------------------------------------

BULLETPROOF = 0
POLYCOMMIT = 1


def ipa_round_lr(variant, g, h, a, b, c: Curve, msm=msm_naive):
    """One round's cross terms.
    BULLETPROOF (src/ipa/mod.rs:40-73): L_msm = MSM(g_hi||h_lo ; a_lo||b_hi), cL = <a_lo,b_hi>;
                                        R_msm = MSM(g_lo||h_hi ; a_hi||b_lo), cR = <a_hi,b_lo>.
    POLYCOMMIT  (src/poly_commit/mod.rs:76-87): L_msm = MSM(g_hi ; a_lo), cL = <a_lo,b_hi>;
                                        R_msm = MSM(g_lo ; a_hi), cR = <a_hi,b_lo>.
    The host adds u*c (and h*l_j) afterwards."""
    n = len(a) // 2
    a_lo, a_hi = a[:n], a[n:]
    b_lo, b_hi = b[:n], b[n:]
    g_lo, g_hi = g[:n], g[n:]
    cL = inner_product(a_lo, b_hi, c.n)
    cR = inner_product(a_hi, b_lo, c.n)
    if variant == BULLETPROOF:
        h_lo, h_hi = h[:n], h[n:]
        L = msm(g_hi + h_lo, a_lo + b_hi, c)
        Rr = msm(g_lo + h_hi, a_hi + b_lo, c)
    else:
        L = msm(g_hi, a_lo, c)
        Rr = msm(g_lo, a_hi, c)
    return L, Rr, cL, cR


def ipa_round_fold(variant, g, h, a, b, x, c: Curve):
    """BULLETPROOF (src/ipa/mod.rs:80-92): G' = x^-1 G_lo + x G_hi, H' = x H_lo + x^-1 H_hi,
                                           a' = a_lo x + a_hi x^-1, b' = b_lo x^-1 + b_hi x.
    POLYCOMMIT (src/poly_commit/mod.rs:95-102): a' = a_hi u^-1 + a_lo u, b' = b_lo u^-1 + b_hi u,
                                           g' = g_lo u^-1 + g_hi u."""
    n = len(a) // 2
    xi = pow(x, -1, c.n)
    a_lo, a_hi = a[:n], a[n:]
    b_lo, b_hi = b[:n], b[n:]
    g_lo, g_hi = g[:n], g[n:]
    if variant == BULLETPROOF:
        h_lo, h_hi = h[:n], h[n:]
        g2 = fold_generators(g_lo, g_hi, xi, x, c)
        h2 = fold_generators(h_lo, h_hi, x, xi, c)
        a2 = fold_scalars(a_lo, a_hi, x, xi, c.n)
        b2 = fold_scalars(b_lo, b_hi, xi, x, c.n)
    else:
        g2 = fold_generators(g_lo, g_hi, xi, x, c)
        h2 = []
        a2 = fold_scalars(a_hi, a_lo, xi, x, c.n)
        b2 = fold_scalars(b_lo, b_hi, xi, x, c.n)
    return g2, h2, a2, b2


def ipa_prove(variant, g, h, a, b, challenges, c: Curve, msm=msm_naive):
    """Runs the whole `while n != 1` loop; returns per-round records and the finals."""
    rounds = []
    g, h, a, b = list(g), list(h), list(a), list(b)
    for x in challenges:
        L, Rr, cL, cR = ipa_round_lr(variant, g, h, a, b, c, msm)
        g, h, a, b = ipa_round_fold(variant, g, h, a, b, x, c)
        rounds.append(dict(L=L, R=Rr, cL=cL, cR=cR))
    return rounds, (g, h, a, b)


def ipa_verify_scalars(challenges, a, b, c: Curve):
    """src/ipa/mod.rs:127-167: scalars multiplying gs (a*s_i) and hs (b*s_i^-1)."""
    ss = s_vector(challenges, c.n)
    ss_inv = batch_inversion(ss, c.n)
    return scale(ss, a, c.n), scale(ss_inv, b, c.n)


# ---------------------------------------------------------------------------
# Aggregated range proof, prover side (src/range/aggregate.rs:23-170) with seeded
# randomness and challenges: the caller of the hot path for BASELINE config 4.
# ---------------------------------------------------------------------------


def range_aggregate_prove(c: Curve, gs, hs, u, g, h, values, gammas, n_bits, rand_seed, chal_seed):
    """Returns every prover message.  Randomness: one SplitMix64 stream (alpha, rho,
    s_L, s_R, tau1, tau2 in the reference's order); challenge number j is the first
    field element of SplitMix64(chal_seed + j)."""
    mod = c.n
    m = len(values)
    n = n_bits * m
    rnd = SplitMix64(rand_seed)
    counter = [0]

    def chal():
        v = SplitMix64(chal_seed + counter[0]).field(mod)
        counter[0] += 1
        return v

    gs, hs = list(gs[:n]), list(hs[:n])
    a_l = [(v >> i) & 1 for v in values for i in range(n_bits)]          # :50-58
    a_r = [(b - 1) % mod for b in a_l]                                     # :60-64
    alpha, rho = rnd.field(mod), rnd.field(mod)                            # :66-67
    s_l = [rnd.field(mod) for _ in range(n)]                               # :68-70
    s_r = [rnd.field(mod) for _ in range(n)]                               # :71-73
    msm = msm_pippenger if n > 16 else msm_naive
    A = add(scalar_mul(alpha, h, c), msm(gs + hs, a_l + a_r, c), c)        # :81
    S = add(scalar_mul(rho, h, c), msm(gs + hs, s_l + s_r, c), c)          # :82
    y, z = chal(), chal()                                                  # :86
    y_vec = powers(y, n, mod)
    l0 = [(a - z) % mod for a in a_l]                                      # :93-101
    zz = z * z % mod
    r0, r1 = [], []
    for j in range(m):                                                     # :103-119
        zp = zz * pow(z, j, mod) % mod
        for i in range(n_bits):
            k = j * n_bits + i
            r0.append((y_vec[k] * ((a_r[k] + z) % mod) + zp * pow(2, i, mod)) % mod)
            r1.append(y_vec[k] * s_r[k] % mod)
    tao1, tao2 = rnd.field(mod), rnd.field(mod)                            # :121-122
    t1 = (inner_product(l0, r1, mod) + inner_product(s_l, r0, mod)) % mod  # src/range/types.rs:98-117
    t2 = inner_product(s_l, r1, mod)
    TT1 = add(scalar_mul(t1, g, c), scalar_mul(tao1, h, c), c)             # :126
    TT2 = add(scalar_mul(t2, g, c), scalar_mul(tao2, h, c), c)             # :127
    x = chal()                                                             # :133
    sigma = sum(zz * pow(z, j, mod) * gammas[j] for j in range(m)) % mod   # :136-139
    tao_x = (tao1 * x + tao2 * x * x + sigma) % mod                        # :140
    mu = (alpha + rho * x) % mod                                           # :142
    l = [(a + b * x) % mod for a, b in zip(l0, s_l)]                       # :143
    r = [(a + b * x) % mod for a, b in zip(r0, r1)]                        # :144
    t_hat = inner_product(l, r, mod)
    hs_p = scale_points(hs, hs_prime_scalars(n, y, mod), c)                # :148-153
    xe = chal()                                                            # src/ipa/extended.rs:24
    u2 = scalar_mul(xe, u, c)                                              # :32
    k = n.bit_length() - 1
    chs = [chal() for _ in range(k)]
    rounds, (gf, hf, af, bf) = ipa_prove(BULLETPROOF, gs, hs_p, l, r, chs, c, msm)
    Ls = [add(rd["L"], scalar_mul(rd["cL"], u2, c), c) for rd in rounds]   # src/ipa/mod.rs:43
    Rs = [add(rd["R"], scalar_mul(rd["cR"], u2, c), c) for rd in rounds]
    return dict(A=A, S=S, TT1=TT1, TT2=TT2, tao_x=tao_x, mu=mu, t_hat=t_hat, L=Ls, R=Rs, a=af[0], b=bf[0])


def range_prove(c: Curve, gs, hs, u, g, h, v, gamma, n_bits, rand_seed, chal_seed):
    """Single-value range proof, src/range/mod.rs:26-159, written from that file's own
    expressions (two_vec scaled by z^2, tao_x with z^2 * gamma).  Same stand-ins for
    randomness and challenges as range_aggregate_prove."""
    mod = c.n
    n = n_bits
    rnd = SplitMix64(rand_seed)
    counter = [0]

    def chal():
        val = SplitMix64(chal_seed + counter[0]).field(mod)
        counter[0] += 1
        return val

    gs, hs = list(gs[:n]), list(hs[:n])
    two_vec = [pow(2, i, mod) for i in range(n)]                           # :46-50
    a_l = [(v >> i) & 1 for i in range(n)]                                 # :52-56
    a_r = [(b - 1) % mod for b in a_l]                                     # :58
    alpha, rho = rnd.field(mod), rnd.field(mod)                            # :60-61
    s_l = [rnd.field(mod) for _ in range(n)]                               # :62-64
    s_r = [rnd.field(mod) for _ in range(n)]                               # :65-67
    msm = msm_pippenger if n > 16 else msm_naive
    A = add(scalar_mul(alpha, h, c), msm(gs + hs, a_l + a_r, c), c)        # :77
    S = add(scalar_mul(rho, h, c), msm(gs + hs, s_l + s_r, c), c)          # :78
    y, z = chal(), chal()                                                  # :82
    y_vec = powers(y, n, mod)                                              # :83-86
    l0 = [(a - z) % mod for a in a_l]                                      # :90-92
    r0 = [(yy * ((ar + z) % mod) + z * z * tw) % mod for yy, ar, tw in zip(y_vec, a_r, two_vec)]   # :100-109
    r1 = hadamard(y_vec, s_r, mod)                                         # :110
    tao1, tao2 = rnd.field(mod), rnd.field(mod)                            # :115-116
    t1 = (inner_product(l0, r1, mod) + inner_product(s_l, r0, mod)) % mod  # :119
    t2 = inner_product(s_l, r1, mod)
    TT1 = add(scalar_mul(t1, g, c), scalar_mul(tao1, h, c), c)             # :120
    TT2 = add(scalar_mul(t2, g, c), scalar_mul(tao2, h, c), c)             # :121
    x = chal()                                                             # :127
    tao_x = (tao2 * x * x + tao1 * x + z * z * gamma) % mod                # :129
    mu = (alpha + rho * x) % mod                                           # :130
    l = [(a + b * x) % mod for a, b in zip(l0, s_l)]                       # :131
    r = [(a + b * x) % mod for a, b in zip(r0, r1)]                        # :132
    t_hat = inner_product(l, r, mod)
    hs_p = scale_points(hs, hs_prime_scalars(n, y, mod), c)                # :136-141
    xe = chal()                                                            # src/ipa/extended.rs:24
    u2 = scalar_mul(xe, u, c)
    k = n.bit_length() - 1
    chs = [chal() for _ in range(k)]
    rounds, (gf, hf, af, bf) = ipa_prove(BULLETPROOF, gs, hs_p, l, r, chs, c, msm)
    Ls = [add(rd["L"], scalar_mul(rd["cL"], u2, c), c) for rd in rounds]
    Rs = [add(rd["R"], scalar_mul(rd["cR"], u2, c), c) for rd in rounds]
    return dict(A=A, S=S, TT1=TT1, TT2=TT2, tao_x=tao_x, mu=mu, t_hat=t_hat, L=Ls, R=Rs, a=af[0], b=bf[0])


# ---- polynomial commitment (src/poly_commit) ---------------------------------------------

def poly_commit_statement(c: Curve, gs, h, coeffs, r, x):
    """Witness::statement, src/poly_commit/types.rs:137-151: (commitment, evaluation)."""
    mod = c.n
    padded = list(coeffs) + [0] * (len(gs) - len(coeffs))
    msm = msm_pippenger if len(gs) > 16 else msm_naive
    commitment = add(msm(gs, padded, c), scalar_mul(r, h, c), c)
    evaluation = inner_product(padded, powers(x, len(gs), mod), mod)
    return commitment, evaluation


def poly_commit_prove(c: Curve, gs, h, coeffs, r, x, rand_seed, chal_seed):
    """poly_commit::prove, src/poly_commit/mod.rs:45-119.  Randomness: (l_j, r_j) per round
    from one SplitMix64 stream; challenge number j is SplitMix64(chal_seed + j).field:
    u_coeff first, then one u_j per round.  Returns (L list, R list, a, r, g_final, ui)."""
    mod = c.n
    n = len(gs)
    rnd = SplitMix64(rand_seed)
    counter = [0]

    def chal():
        v = SplitMix64(chal_seed + counter[0]).field(mod)
        counter[0] += 1
        return v

    u = scalar_mul(chal(), c.g, c)                                         # :55-58
    g = list(gs)
    a = list(coeffs) + [0] * (n - len(coeffs))
    b = powers(x, n, mod)                                                  # :65
    msm = msm_pippenger if n > 32 else msm_naive
    Ls, Rs, ui = [], [], []
    while n != 1:
        n //= 2
        g_lo, g_hi, a_lo, a_hi, b_lo, b_hi = g[:n], g[n:], a[:n], a[n:], b[:n], b[n:]
        l_j, r_j = rnd.field(mod), rnd.field(mod)                          # :75
        left = add(add(msm(g_hi, a_lo, c), scalar_mul(l_j, h, c), c),
                   scalar_mul(inner_product(a_lo, b_hi, mod), u, c), c)    # :78-80
        right = add(add(msm(g_lo, a_hi, c), scalar_mul(r_j, h, c), c),
                    scalar_mul(inner_product(a_hi, b_lo, mod), u, c), c)   # :83-85
        Ls.append(left)
        Rs.append(right)
        u_j = chal()                                                       # :91
        ui.append(u_j)
        u_inv = pow(u_j, -1, mod)
        a = fold_scalars(a_hi, a_lo, u_inv, u_j, mod)                      # :96
        b = fold_scalars(b_lo, b_hi, u_inv, u_j, mod)                      # :97
        g = fold_generators(g_lo, g_hi, u_inv, u_j, c)                     # :102
        r = (r + l_j * u_j * u_j + r_j * u_inv * u_inv) % mod              # :104
    return Ls, Rs, a[0], r, g[0], ui


def poly_commit_check(c: Curve, gs, h, commitment, evaluation, x, Ls, Rs, a, r, chal_seed, g_assumed=None):
    """poly_commit::verify (src/poly_commit/mod.rs:121-173), or lazy_verify (:264-315) when
    g_assumed is given: returns (accepted, ui)."""
    mod = c.n
    counter = [0]

    def chal():
        v = SplitMix64(chal_seed + counter[0]).field(mod)
        counter[0] += 1
        return v

    u = scalar_mul(chal(), c.g, c)
    q = add(commitment, scalar_mul(evaluation, u, c), c)                   # :137
    ui = []
    for L, R in zip(Ls, Rs):
        u_j = chal()
        ui.append(u_j)
        u_inv = pow(u_j, -1, mod)
        q = add(add(q, scalar_mul(u_j * u_j % mod, L, c), c), scalar_mul(u_inv * u_inv % mod, R, c), c)  # :147-150
    if g_assumed is None:
        ss = s_vector(ui, mod)                                             # :152-157
        msm = msm_pippenger if len(gs) > 32 else msm_naive
        g = msm(gs, ss, c)                                                 # :159
        b = inner_product(ss, powers(x, len(gs), mod), mod)                # :160-163
    else:
        g = g_assumed
        b = hpoly_evaluate(ui, x, mod)                                     # :301
    res = add(add(scalar_mul(a, g, c), scalar_mul(r, h, c), c), scalar_mul(a * b % mod, u, c), c)
    return res == q, ui                                                    # :166-172


def fold_todos(c: Curve, todos, alpha, x):
    """fold_todos_witness / fold_todos_statement (src/poly_commit/mod.rs:228-262).
    todos: list of (g point, ui).  Returns (coeffs, commitment, evaluation)."""
    mod = c.n
    coeffs, commitment, evaluation, ai = None, None, 0, 1
    for g, ui in todos:
        ss = s_vector(ui, mod)
        coeffs = [ai * v % mod for v in ss] if coeffs is None else [(u + ai * v) % mod for u, v in zip(coeffs, ss)]
        commitment = add(commitment, scalar_mul(ai, g, c), c)
        evaluation = (evaluation + ai * hpoly_evaluate(ui, x, mod)) % mod
        ai = ai * alpha % mod
    return coeffs, commitment, evaluation


# ---- wire format of points (ark-serialize 0.5, Compress::Yes) ------------------------------------
# Not in the reference tree: the proof and CRS types derive their encoding from the arkworks crates
# (src/ipa/types.rs:48,140).  Restated from the published format; no reference byte vector pins it.

def field_sqrt(a, p):
    """a square root of a mod p, or None (Tonelli-Shanks; p = 3 mod 4 takes the short cut)"""
    a %= p
    if a == 0:
        return 0
    if pow(a, (p - 1) // 2, p) != 1:
        return None
    if p % 4 == 3:
        return pow(a, (p + 1) // 4, p)
    s, t = 0, p - 1
    while t % 2 == 0:
        s, t = s + 1, t // 2
    g = 2
    while pow(g, (p - 1) // 2, p) != p - 1:
        g += 1
    z, x, b, v = pow(g, t, p), pow(a, (t + 1) // 2, p), pow(a, t, p), s
    while b != 1:
        k, b2 = 0, b
        while b2 != 1:
            b2, k = b2 * b2 % p, k + 1
        w = pow(z, 1 << (v - k - 1), p)
        z, b, x, v = w * w % p, b * w * w % p, x * w % p, k
    return x


def point_compress(P, c):
    """33 bytes: x little-endian, flag byte (0x80: y > p - y; 0x40: the identity, x = 0)"""
    if P is None:
        return bytes(32) + b"\x40"
    x, y = P
    return x.to_bytes(32, "little") + (b"\x80" if y > (c.p - y) % c.p else b"\x00")


def point_decompress(data, c):
    """-> (ok, point): the smaller root unless bit 7 is set; x >= p, both flags or a non-residue fail"""
    x, flags = int.from_bytes(data[:32], "little"), data[32]
    neg, inf = bool(flags & 0x80), bool(flags & 0x40)
    if x >= c.p or (neg and inf):
        return False, None
    if inf:
        return True, None
    y = field_sqrt((x * x * x + c.b) % c.p, c.p)
    if y is None:
        return False, None
    small, large = min(y, c.p - y), max(y, c.p - y)
    return True, (x, large if neg else small)
