"""GPU parity tests at the full sizes of the benchmark configurations, checked message by
message against the C oracle's restatement of the reference's provers and verifiers
(oracle/bp_oracle_proto.c):

  config 1  single 64-bit range proof          src/range/mod.rs, benches/range_bench.rs
  config 3  IPA prove / verify, n = 2^16        src/ipa/mod.rs, benches/ipa_bench.rs
  config 4  aggregated range proof 64 x 64 bit  src/range/aggregate.rs, benches/aggregate_range_bench.rs
  config 5  polynomial commitment, n = 2^20,    src/poly_commit/mod.rs, benches/poly_commit_bench.rs
            with the fold of deferred checks

(config 2, the standalone MSM at 2^16..2^20, is in test_gpu_parity.py.)  Transcript and
randomness are the seeded stand-ins both sides share; inputs come from SplitMix64 /
numpy seeds 0xB0117E7 + config index."""
import numpy as np
import pytest

import bp_oracle as o
from util import xyz_to_affine

pytestmark = pytest.mark.gpu
SEED = 0xB0117E7
IDENTITY_XYZ_Z = slice(8, 12)


def xy8(P, c):
    return np.concatenate([o.pack_field([P[0]], c.p)[0], o.pack_field([P[1]], c.p)[0]])


def gen_points(coracle, c, rng, count):
    """count points (r0 + i q) G as planes, through the C oracle's generator"""
    start = xy8(o.scalar_mul(rng.field(c.n), c.g, c), c)
    step = xy8(o.scalar_mul(rng.field(c.n), c.g, c), c)
    x, y = coracle.gen_points(c.id, start, step, count)
    return x, y, None


def rand_scalars(seed, count):
    """uniform 254-bit values: valid Montgomery limbs for both curves' scalar fields"""
    a = np.random.default_rng(seed).integers(0, 1 << 64, (count, 4), dtype=np.uint64)
    a[:, 3] &= np.uint64((1 << 62) - 1)
    return a


def is_identity(xyz):
    return not np.any(xyz[IDENTITY_XYZ_Z])


def assert_range_proof_equal(got, exp):
    for key in ("A", "S", "TT1", "TT2", "tao_x", "mu", "t_hat", "a", "b"):
        assert np.array_equal(got[key], exp[key]), key
    assert np.array_equal(got["L"], exp["L"]) and np.array_equal(got["R"], exp["R"])


@pytest.mark.parametrize("n_bits", [8, 16, 32, 64])
def test_config1_single_value_range_proof(ctx, coracle, n_bits):
    """range::prove / verify (src/range/mod.rs:26-268) at the sizes of benches/range_bench.rs:92:
    every prover message equals the oracle's, the verifier's deferred MSM evaluates to the identity,
    on a corrupted proof to the very point the oracle's verifier computes, and proofs of another
    shape are refused"""
    from bulletproofs_amd import host_mirror

    c = o.SECP256K1
    rng = o.SplitMix64(SEED + 1 + n_bits)
    threads = coracle.max_threads()
    G, H = gen_points(coracle, c, rng, n_bits), gen_points(coracle, c, rng, n_bits)
    u, g, h = (o.scalar_mul(rng.field(c.n), c.g, c) for _ in range(3))
    ugh = np.stack([xy8(P, c) for P in (u, g, h)])
    v = rng.next() & ((1 << n_bits) - 1)
    gamma_int = rng.field(c.n)
    gamma = o.pack_field([gamma_int], c.n)
    res = host_mirror.range_aggregate_roundtrip(ctx, c.id, G, H, ugh, [v], gamma, n_bits, seed_rand=11, seed_chal=22, single=True)
    exp = coracle.range_agg_prove(c.id, G, H, ugh, [v], gamma, n_bits, 11, 22, threads=threads)
    assert_range_proof_equal(res, exp)
    assert res["ok"] and is_identity(res["value"])
    # the restatement of src/range/mod.rs from its own expressions (Python big integers)
    gs = o.unpack_points(G[0], G[1], np.zeros(n_bits, dtype=np.uint8), c)
    hs = o.unpack_points(H[0], H[1], np.zeros(n_bits, dtype=np.uint8), c)
    py = o.range_prove(c, gs, hs, u, g, h, v, gamma_int, n_bits, 11, 22)
    assert xyz_to_affine(res["A"], c) == py["A"] and xyz_to_affine(res["TT2"], c) == py["TT2"]
    assert o.unpack_field(res["t_hat"], c.n)[0] == py["t_hat"] and o.unpack_field(res["a"], c.n)[0] == py["a"]
    assert xyz_to_affine(res["L"][-1], c) == py["L"][-1]
    # the oracle's verifier accepts the GPU's proof
    ov = coracle.range_agg_verify(c.id, G, H, ugh, [v], gamma, n_bits, res["points"], res["scalars"], res["lr"], 11 ^ 0x5EED, 22,
                                  threads=threads)
    assert is_identity(ov)
    bad = host_mirror.range_aggregate_roundtrip(ctx, c.id, G, H, ugh, [v], gamma, n_bits, seed_rand=11, seed_chal=22, tamper=1,
                                                single=True)
    assert not bad["ok"] and not is_identity(bad["value"])
    sc = res["scalars"].copy()
    sc[2] = o.pack_field([(o.unpack_field(sc[2], c.n)[0] + 1) % c.n], c.n)[0]
    ov = coracle.range_agg_verify(c.id, G, H, ugh, [v], gamma, n_bits, res["points"], sc, res["lr"], 11 ^ 0x5EED, 22, threads=threads)
    assert np.array_equal(bad["value"], ov)
    for mode in (2, 3):
        assert not host_mirror.range_aggregate_roundtrip(ctx, c.id, G, H, ugh, [v], gamma, n_bits, seed_rand=11, seed_chal=22,
                                                         tamper=mode, single=True)["ok"]


@pytest.mark.parametrize("c", [o.SECP256K1, o.VESTA], ids=lambda c: c.name)
@pytest.mark.parametrize("m,n_bits", [(64, 64), (8, 32)])
def test_config4_aggregated_range_proof_full_size(ctx, coracle, c, m, n_bits):
    """range::aggregate::prove / verify (src/range/aggregate.rs:23-300) at m = 64 values of 64 bits
    (N = 4096, benches/aggregate_range_bench.rs:93-97): A, S, T1, T2, tao_x, mu, t_hat, all 12
    (L_j, R_j), a, b equal the oracle's; the oracle's verifier accepts the GPU's proof; the GPU
    verifier's deferred MSM equals the oracle's on the valid and on a corrupted proof"""
    from bulletproofs_amd import host_mirror

    if c is o.VESTA and m == 64:
        pytest.skip("the 64 x 64 case is run on the curve the reference benches use; Vesta runs 8 x 32")
    n = m * n_bits
    rng = o.SplitMix64(SEED + 4 + c.id)
    threads = coracle.max_threads()
    G, H = gen_points(coracle, c, rng, n), gen_points(coracle, c, rng, n)
    ugh = np.stack([xy8(o.scalar_mul(rng.field(c.n), c.g, c), c) for _ in range(3)])
    values = np.random.default_rng(SEED + 4).integers(0, 1 << n_bits, m, dtype=np.uint64)
    values[0] = (1 << n_bits) - 1
    values[1] = 0
    gamma = rand_scalars(SEED + 40, m)
    res = host_mirror.range_aggregate_roundtrip(ctx, c.id, G, H, ugh, values, gamma, n_bits, seed_rand=31, seed_chal=32)
    exp = coracle.range_agg_prove(c.id, G, H, ugh, values, gamma, n_bits, 31, 32, threads=threads)
    assert_range_proof_equal(res, exp)
    assert len(res["L"]) == n.bit_length() - 1
    assert res["ok"] and is_identity(res["value"])
    ov = coracle.range_agg_verify(c.id, G, H, ugh, values, gamma, n_bits, res["points"], res["scalars"], res["lr"], 31 ^ 0x5EED, 32,
                                  threads=threads)
    assert is_identity(ov)
    bad = host_mirror.range_aggregate_roundtrip(ctx, c.id, G, H, ugh, values, gamma, n_bits, seed_rand=31, seed_chal=32, tamper=1)
    assert not bad["ok"]
    sc = res["scalars"].copy()
    sc[2] = o.pack_field([(o.unpack_field(sc[2], c.n)[0] + 1) % c.n], c.n)[0]
    ov = coracle.range_agg_verify(c.id, G, H, ugh, values, gamma, n_bits, res["points"], sc, res["lr"], 31 ^ 0x5EED, 32, threads=threads)
    assert not is_identity(ov) and np.array_equal(bad["value"], ov)
    # a statement about other values is not proved by this proof
    other = values.copy()
    other[3] ^= np.uint64(1)
    assert not is_identity(coracle.range_agg_verify(c.id, G, H, ugh, other, gamma, n_bits, res["points"], res["scalars"], res["lr"],
                                                    31 ^ 0x5EED, 32, threads=threads))
    for mode in (2, 3):
        assert not host_mirror.range_aggregate_roundtrip(ctx, c.id, G, H, ugh, values, gamma, n_bits, seed_rand=31, seed_chal=32,
                                                         tamper=mode)["ok"]


@pytest.mark.parametrize("c", [o.VESTA, o.SECP256K1], ids=lambda c: c.name)
def test_config3_ipa_2_16_prove_verify(ctx, coracle, c):
    """ipa::prove -> verify at n = 2^16 (the largest size of src/ipa/mod.rs:246-288, run there on
    Vesta, and of benches/ipa_bench.rs:102-105 on secp256k1): the statement, all 16 (L_j, R_j), a and b
    equal the oracle's; the GPU verifier (resident CRS and keyed accumulator) and the oracle's verifier
    accept; a corrupted proof and proofs of another shape are rejected, alone and in a batch"""
    from bulletproofs_amd import host_mirror

    n = 1 << 16
    rng = o.SplitMix64(SEED + 3 + c.id)
    threads = coracle.max_threads()
    G, H = gen_points(coracle, c, rng, n), gen_points(coracle, c, rng, n)
    u_xy = xy8(o.scalar_mul(rng.field(c.n), c.g, c), c)
    n_proofs = 2
    A, B = rand_scalars(SEED + 30 + c.id, n * n_proofs), rand_scalars(SEED + 31 + c.id, n * n_proofs)
    seed = 4141
    res = host_mirror.ipa_roundtrip(ctx, c.id, G, H, u_xy, A, B, n_proofs, seed)
    assert res["single_ok"] and res["batch_ok"]
    P = coracle.ipa_statement(c.id, G, H, u_xy, A[:n], B[:n], threads=threads)
    assert np.array_equal(res["P"], P)
    exp = coracle.ipa_prove(c.id, G, H, u_xy, A[:n], B[:n], seed, threads=threads)
    assert len(res["L"]) == 16
    assert np.array_equal(res["L"], exp["L"]) and np.array_equal(res["R"], exp["R"])
    assert np.array_equal(res["a"], exp["a"]) and np.array_equal(res["b"], exp["b"])
    assert is_identity(coracle.ipa_verify(c.id, G, H, u_xy, res["P"], n, res["L"], res["R"], res["a"], res["b"], seed, threads=threads))
    bad_a = o.pack_field([(o.unpack_field(res["a"], c.n)[0] + 1) % c.n], c.n)[0]
    assert not is_identity(coracle.ipa_verify(c.id, G, H, u_xy, res["P"], n, res["L"], res["R"], bad_a, res["b"], seed, threads=threads))
    bad = host_mirror.ipa_roundtrip(ctx, c.id, G, H, u_xy, A, B, n_proofs, seed, tamper=1)
    assert not bad["single_ok"] and not bad["batch_ok"]


@pytest.mark.parametrize("c", [o.SECP256K1, o.VESTA], ids=lambda c: c.name)
@pytest.mark.parametrize("n", [2, 128])
@pytest.mark.parametrize("mode", [2, 3, 4])
def test_proofs_of_another_shape_are_refused(ctx, coracle, c, n, mode):
    """a proof with a round dropped, a round appended or one more R than L never reaches a kernel:
    the reference reads exactly log2(n) pairs (src/ipa/mod.rs:110-119, src/poly_commit/mod.rs:129-145)"""
    from bulletproofs_amd import host_mirror

    rng = o.SplitMix64(SEED + 99 + n + c.id)
    G, H = gen_points(coracle, c, rng, n), gen_points(coracle, c, rng, n)
    u_xy = xy8(o.scalar_mul(rng.field(c.n), c.g, c), c)
    A, B = rand_scalars(1, n), rand_scalars(2, n)
    bad = host_mirror.ipa_roundtrip(ctx, c.id, G, H, u_xy, A, B, 1, 5, tamper=mode)
    assert not bad["single_ok"] and not bad["batch_ok"]
    hg = np.stack([u_xy, xy8(c.g, c)])
    bad = host_mirror.poly_commit_roundtrip(ctx, c.id, G, hg, A, B[0], B[1], seed_rand=5, seed_chal=6, tamper=mode, halo=False)
    assert not bad["verify_ok"] and not bad["lazy_ok"]


def check_poly(ctx, coracle, c, k, n_coeffs, reprove_all):
    from bulletproofs_amd import host_mirror

    n = 1 << k
    rng = o.SplitMix64(SEED + 5 + k + c.id)
    threads = coracle.max_threads()
    G = gen_points(coracle, c, rng, n)
    hg = np.stack([xy8(o.scalar_mul(rng.field(c.n), c.g, c), c), xy8(c.g, c)])
    coeffs = rand_scalars(SEED + 50 + k, n_coeffs)
    r, x = rand_scalars(SEED + 51, 2)
    res = host_mirror.poly_commit_roundtrip(ctx, c.id, G, hg, coeffs, r, x, seed_rand=5, seed_chal=6)
    assert res["verify_ok"] and res["lazy_ok"] and res["halo_ok"]
    # statement and opening proof, message by message
    com, ev = coracle.poly_statement(c.id, G, hg, coeffs, r, x, threads=threads)
    assert np.array_equal(res["commitment"], com) and np.array_equal(res["evaluation"], ev)
    exp = coracle.poly_prove(c.id, G, hg, coeffs, r, x, 5, 6, threads=threads)   # asserts g[0] == <s, gs> as :108-113
    assert len(res["L"]) == k
    assert np.array_equal(res["L"], exp["L"]) and np.array_equal(res["R"], exp["R"])
    assert np.array_equal(res["a"], exp["a"]) and np.array_equal(res["r"], exp["r"])
    assert np.array_equal(res["g"], exp["g"])
    # the oracle's verify and lazy_verify accept the GPU's proof, and reject it for another evaluation
    ores, ui = coracle.poly_verify(c.id, G, hg, res["commitment"], x, res["evaluation"], res["L"], res["R"], res["a"], res["r"], 6,
                                   threads=threads)
    assert is_identity(ores) and np.array_equal(ui, exp["ui"])
    ores, _ = coracle.poly_verify(c.id, G, hg, res["commitment"], x, res["evaluation"], res["L"], res["R"], res["a"], res["r"], 6,
                                  g_assumed=res["g"], threads=threads)
    assert is_identity(ores)
    ores, _ = coracle.poly_verify(c.id, G, hg, res["commitment"], x, r, res["L"], res["R"], res["a"], res["r"], 6, g_assumed=res["g"],
                                  threads=threads)
    assert not is_identity(ores)
    bad = host_mirror.poly_commit_roundtrip(ctx, c.id, G, hg, coeffs, r, x, seed_rand=5, seed_chal=6, tamper=1, halo=False)
    assert not bad["verify_ok"] and not bad["lazy_ok"]
    # second opening (reversed coefficients) and the fold of the two deferred checks
    sec, fol = res["second"], res["folded"]
    coeffs2 = np.ascontiguousarray(coeffs[::-1])
    com2, ev2 = coracle.poly_statement(c.id, G, hg, coeffs2, r, x, threads=threads)
    assert np.array_equal(sec["commitment"], com2) and np.array_equal(sec["evaluation"], ev2)
    if reprove_all:
        exp2 = coracle.poly_prove(c.id, G, hg, coeffs2, r, x, 5 + 7777, 6 + 7777, threads=threads)
        for key in ("L", "R", "a", "r", "g"):
            assert np.array_equal(sec[key], exp2[key]), key
    ores, ui2 = coracle.poly_verify(c.id, G, hg, sec["commitment"], x, sec["evaluation"], sec["L"], sec["R"], sec["a"], sec["r"],
                                    6 + 7777, threads=threads)   # full verify: also checks sec["g"] below
    assert is_identity(ores)
    ores, _ = coracle.poly_verify(c.id, G, hg, sec["commitment"], x, sec["evaluation"], sec["L"], sec["R"], sec["a"], sec["r"],
                                  6 + 7777, g_assumed=sec["g"], threads=threads)
    assert is_identity(ores)
    # fold_todos_witness / fold_todos_statement (src/poly_commit/mod.rs:228-262)
    alpha = o.pack_field([o.SplitMix64((6 ^ 0xA1FA) + 0).field(c.n)], c.n)[0]
    x2 = o.pack_field([o.SplitMix64((6 ^ 0xA1FA) + 1).field(c.n)], c.n)[0]
    assert np.array_equal(fol["alpha"], alpha) and np.array_equal(fol["x"], x2)
    uis = np.stack([ui, ui2])
    fc, fcom, fev = coracle.poly_fold_todos(c.id, uis, np.stack([res["g"], sec["g"]]), alpha, x2, threads=threads)
    assert np.array_equal(fol["witness"], fc)
    assert np.array_equal(fol["commitment"], fcom) and np.array_equal(fol["evaluation"], fev)
    # the folded witness commits (r = 0) to the folded statement: the identity behind the halo trick
    wcom, wev = coracle.poly_statement(c.id, G, hg, fc, np.zeros(4, dtype=np.uint64), x2, threads=threads)
    assert np.array_equal(wcom, fcom) and np.array_equal(wev, fev)
    # one opening discharges both deferred checks
    if reprove_all:
        exp3 = coracle.poly_prove(c.id, G, hg, fc, np.zeros(4, dtype=np.uint64), x2, 5 + 9999, 6 + 9999, threads=threads)
        for key in ("L", "R", "a", "r"):
            assert np.array_equal(fol[key], exp3[key]), key
    ores, _ = coracle.poly_verify(c.id, G, hg, fol["commitment"], x2, fol["evaluation"], fol["L"], fol["R"], fol["a"], fol["r"],
                                  6 + 9999, threads=threads)
    assert is_identity(ores)


@pytest.mark.parametrize("c", [o.SECP256K1, o.VESTA], ids=lambda c: c.name)
def test_poly_commit_2_12_every_opening_reproved_by_the_oracle(ctx, coracle, c):
    """the three openings of the halo flow (benches/poly_commit_bench.rs:136-158 shape) at n = 2^12,
    each re-proved by the oracle and compared message by message"""
    check_poly(ctx, coracle, c, 12, (1 << 12) - 37, reprove_all=True)


def test_config5_poly_commit_2_20_with_halo_fold(ctx, coracle):
    """poly_commit prove / verify / lazy_verify / fold at degree 2^20 - 1 on secp256k1: the first
    opening (all 20 L_j, R_j, a, r and the folded generator, which the oracle asserts to equal
    <h_poly.coeffs(), gs> as src/poly_commit/mod.rs:108-113) equals the oracle's; the second opening
    and the opening of the folded statement are accepted by the oracle's verifier
    (src/poly_commit/mod.rs:121-173), whose n-point MSM also pins their folded generators; the
    folded witness and statement equal the oracle's"""
    check_poly(ctx, coracle, o.SECP256K1, 20, 1 << 20, reprove_all=False)


def test_config4_verify_batch_of_100_at_full_size(ctx, coracle):
    """verify_batch at BATCH_SIZE 100 over 64 x 64-bit aggregated range proofs
    (benches/aggregate_range_bench.rs:65-84, src/msm.rs:91-115): four distinct proofs cycled through the
    batch; the device-side combination, the combination of host accumulators on the device and the
    reference's host merge all accept; every distinct proof is the one the single-proof driver
    produces from the same seeds and is accepted by the oracle's verifier; with the last proof of
    the batch corrupted all three reject"""
    from bulletproofs_amd import host_mirror

    c = o.SECP256K1
    m, n_bits, distinct, batch = 64, 64, 4, 100
    n = m * n_bits
    rng = o.SplitMix64(SEED + 44)
    threads = coracle.max_threads()
    G, H = gen_points(coracle, c, rng, n), gen_points(coracle, c, rng, n)
    ugh = np.stack([xy8(o.scalar_mul(rng.field(c.n), c.g, c), c) for _ in range(3)])
    values = np.random.default_rng(SEED + 45).integers(0, 1 << 63, (distinct, m), dtype=np.uint64)
    gamma = rand_scalars(SEED + 46, distinct * m).reshape(distinct, m, 4)
    res = host_mirror.range_aggregate_verify_batch(ctx, c.id, G, H, ugh, values, gamma, n_bits, batch, seed_rand=51, seed_chal=52)
    assert res["ok_direct"] and res["ok_device"] and res["ok_host"]
    for p in range(distinct):
        one = host_mirror.range_aggregate_roundtrip(ctx, c.id, G, H, ugh, values[p], gamma[p], n_bits, seed_rand=51 + p, seed_chal=52 + p)
        assert one["ok"]
        ov = coracle.range_agg_verify(c.id, G, H, ugh, values[p], gamma[p], n_bits, one["points"], one["scalars"], one["lr"],
                                      (51 + p) ^ 0x5EED, 52 + p, threads=threads)
        assert is_identity(ov), p
    bad = host_mirror.range_aggregate_verify_batch(ctx, c.id, G, H, ugh, values, gamma, n_bits, batch, seed_rand=51, seed_chal=52,
                                                   tamper=True)
    assert not bad["ok_direct"] and not bad["ok_device"] and not bad["ok_host"]
