"""ctypes access to libbp_host.so: the C++ host mirror of the reference's `Msm`
accumulator (src/msm.rs) and IPA prover/verifier loops (src/ipa/mod.rs), which
drive libbp_gpu through the C ABI.  Used by the parity tests."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib  # loads the HIP runtime and libbp_gpu.so first
from .api import Context, _f, _p, _pts

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libbp_host.so")
if not os.path.exists(_PATH):
    raise ImportError(f"{_PATH} not found: build it with `make -C bulletproofs_amd/host`")
hlib = C.CDLL(_PATH)
vp, sz, ci = C.c_void_p, C.c_size_t, C.c_int
hlib.bph_msm_accumulate.restype = ci
hlib.bph_msm_accumulate.argtypes = [vp, ci, vp, vp, vp, vp, sz, vp, C.POINTER(sz)]
hlib.bph_ipa_roundtrip.restype = ci
hlib.bph_ipa_roundtrip.argtypes = [vp, ci, vp, vp, vp, vp, vp, vp, vp, sz, sz, C.c_uint64, ci, vp, vp, vp, vp,
                                   C.POINTER(ci), C.POINTER(ci)]


def msm_accumulate(ctx: Context, curve, bases, scalars):
    """Msm::from_iter + execute: duplicate bases merge by adding scalars."""
    x, y, inf = _pts(bases)
    s = _f(scalars)
    out = np.zeros(12, dtype=np.uint64)
    distinct = sz()
    rc = hlib.bph_msm_accumulate(ctx.h, curve, _p(x), _p(y), _p(inf), _p(s), x.shape[0], _p(out), C.byref(distinct))
    if rc:
        raise RuntimeError(f"bph_msm_accumulate rc={rc}")
    return out, distinct.value


def ipa_roundtrip(ctx: Context, curve, gs, hs, u_xy, a, b, n_proofs=1, seed=1, tamper=False):
    """statement + ipa::prove + ipa::verify (+ verify_batch_aux) with the seeded transcript stand-in."""
    gx, gy, _ = _pts(gs)
    hx, hy, _ = _pts(hs)
    a, b = _f(a), _f(b)
    n = gx.shape[0]
    assert a.shape[0] == n * n_proofs and b.shape[0] == n * n_proofs
    k = n.bit_length() - 1
    u_xy = np.ascontiguousarray(u_xy, dtype=np.uint64).reshape(8)
    out_l = np.zeros((max(k, 1), 12), dtype=np.uint64)
    out_r = np.zeros((max(k, 1), 12), dtype=np.uint64)
    out_ab = np.zeros(8, dtype=np.uint64)
    out_p = np.zeros(12, dtype=np.uint64)
    ok1, okb = ci(), ci()
    rc = hlib.bph_ipa_roundtrip(ctx.h, curve, _p(gx), _p(gy), _p(hx), _p(hy), _p(u_xy), _p(a), _p(b), n, n_proofs,
                                seed, int(tamper), _p(out_l), _p(out_r), _p(out_ab), _p(out_p), C.byref(ok1),
                                C.byref(okb))
    if rc:
        raise RuntimeError(f"bph_ipa_roundtrip rc={rc}")
    return dict(L=out_l[:k], R=out_r[:k], a=out_ab[:4], b=out_ab[4:], P=out_p, single_ok=bool(ok1.value),
                batch_ok=bool(okb.value))


hlib.bph_range_agg_roundtrip.restype = ci
hlib.bph_range_agg_roundtrip.argtypes = [vp, ci, vp, vp, vp, vp, vp, vp, vp, sz, C.c_uint, C.c_uint64, C.c_uint64, ci, ci, ci,
                                         vp, vp, vp, vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(ci)]


def range_aggregate_roundtrip(ctx: Context, curve, gs, hs, ugh_xy, values, gamma, n_bits, seed_rand=1, seed_chal=2,
                              tamper=False, reps=1, single=False):
    """aggregated range proof (src/range/aggregate.rs): prove + verify through the GPU path.
    ugh_xy: (3, 8) uint64 = affine u, g, h.  values: m integers < 2^n_bits.  tamper: False/0 none,
    True/1 t_hat + 1, 2 a round dropped, 3 a round appended.  single=True (one value) drives the
    single-value types of src/range/mod.rs.  `value` is the verifier's deferred MSM evaluated."""
    gx, gy, _ = _pts(gs)
    hx, hy, _ = _pts(hs)
    values = np.ascontiguousarray(values, dtype=np.uint64)
    gamma = _f(gamma)
    m = values.shape[0]
    n = m * n_bits
    assert n & (n - 1) == 0 and gx.shape[0] >= n and hx.shape[0] >= n
    k = n.bit_length() - 1
    ugh_xy = np.ascontiguousarray(ugh_xy, dtype=np.uint64).reshape(24)
    pts = np.zeros((4, 12), dtype=np.uint64)
    sc = np.zeros((5, 4), dtype=np.uint64)
    lr = np.zeros((max(k, 1), 24), dtype=np.uint64)
    value = np.zeros(12, dtype=np.uint64)
    tp, tv, ok = C.c_double(), C.c_double(), ci()
    rc = hlib.bph_range_agg_roundtrip(ctx.h, curve, _p(gx), _p(gy), _p(hx), _p(hy), _p(ugh_xy), _p(values), _p(gamma), m,
                                      n_bits, seed_rand, seed_chal, int(tamper), reps, int(single), _p(pts), _p(sc), _p(lr),
                                      _p(value), C.byref(tp), C.byref(tv), C.byref(ok))
    if rc:
        raise RuntimeError(f"bph_range_agg_roundtrip rc={rc}")
    return dict(A=pts[0], S=pts[1], TT1=pts[2], TT2=pts[3], tao_x=sc[0], mu=sc[1], t_hat=sc[2], a=sc[3], b=sc[4],
                L=lr[:k, :12], R=lr[:k, 12:], points=pts, scalars=sc, lr=lr, value=value, prove_ms=tp.value,
                verify_ms=tv.value, ok=bool(ok.value))


hlib.bph_range_agg_prove_batch.restype = ci
hlib.bph_range_agg_prove_batch.argtypes = [vp, ci, vp, vp, vp, vp, vp, vp, vp, sz, sz, C.c_uint, C.c_uint64, C.c_uint64, ci, ci, ci,
                                           vp, vp, vp, C.POINTER(C.c_double), C.POINTER(ci)]


def range_aggregate_prove_batch(ctx: Context, curve, gs, hs, ugh_xy, values, gamma, n_bits, seed_rand=1, seed_chal=2, reps=1,
                                host_blinders=False, verify=False):
    """B aggregated range proofs proved together from one host thread (RangeAggregate::prove_batch over
    bp_range_batch_*): values (B, m), gamma (B, m, 4); proof b uses the seeds seed_rand + b, seed_chal + b.
    Returns one dict per proof (keys as range_aggregate_roundtrip's) and the wall time of a batch."""
    gx, gy, _ = _pts(gs)
    hx, hy, _ = _pts(hs)
    values = np.ascontiguousarray(values, dtype=np.uint64)
    gamma = np.ascontiguousarray(gamma, dtype=np.uint64)
    batch, m = values.shape
    assert gamma.shape == (batch, m, 4)
    n = m * n_bits
    assert n & (n - 1) == 0 and gx.shape[0] >= n and hx.shape[0] >= n
    k = n.bit_length() - 1
    ugh_xy = np.ascontiguousarray(ugh_xy, dtype=np.uint64).reshape(24)
    pts = np.zeros((batch, 4, 12), dtype=np.uint64)
    sc = np.zeros((batch, 5, 4), dtype=np.uint64)
    lr = np.zeros((batch, max(k, 1), 24), dtype=np.uint64)
    tb, ok = C.c_double(), ci()
    rc = hlib.bph_range_agg_prove_batch(ctx.h, curve, _p(gx), _p(gy), _p(hx), _p(hy), _p(ugh_xy), _p(values), _p(gamma), batch, m,
                                        n_bits, seed_rand, seed_chal, reps, int(bool(host_blinders)), int(bool(verify)), _p(pts),
                                        _p(sc), _p(lr), C.byref(tb), C.byref(ok))
    if rc:
        raise RuntimeError(f"bph_range_agg_prove_batch rc={rc}: {_lib.lib.bp_last_error(ctx.h).decode()}")
    proofs = [dict(A=pts[b, 0], S=pts[b, 1], TT1=pts[b, 2], TT2=pts[b, 3], tao_x=sc[b, 0], mu=sc[b, 1], t_hat=sc[b, 2], a=sc[b, 3],
                   b=sc[b, 4], L=lr[b, :k, :12], R=lr[b, :k, 12:], points=pts[b], scalars=sc[b], lr=lr[b]) for b in range(batch)]
    return dict(proofs=proofs, batch_ms=tb.value, per_s=batch / tb.value * 1e3, ok=bool(ok.value))


hlib.bph_range_agg_throughput.restype = ci
hlib.bph_range_agg_throughput.argtypes = [ci, ci, vp, vp, vp, vp, vp, vp, vp, sz, C.c_uint, C.c_uint64, C.c_uint64, ci, ci,
                                          C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(ci)]


def range_aggregate_throughput(device, curve, gs, hs, ugh_xy, values, gamma, n_bits, workers, reps, seed_rand=1, seed_chal=2):
    """`workers` host threads (one GPU context each) prove and then verify `reps` aggregated
    range proofs each, concurrently on one GPU.  Returns proofs/s for each phase."""
    gx, gy, _ = _pts(gs)
    hx, hy, _ = _pts(hs)
    values = np.ascontiguousarray(values, dtype=np.uint64)
    gamma = _f(gamma)
    m = values.shape[0]
    n = m * n_bits
    assert n & (n - 1) == 0 and gx.shape[0] >= n and hx.shape[0] >= n
    ugh_xy = np.ascontiguousarray(ugh_xy, dtype=np.uint64).reshape(24)
    tp, tv, ok = C.c_double(), C.c_double(), ci()
    rc = hlib.bph_range_agg_throughput(device, curve, _p(gx), _p(gy), _p(hx), _p(hy), _p(ugh_xy), _p(values), _p(gamma), m,
                                       n_bits, seed_rand, seed_chal, workers, reps, C.byref(tp), C.byref(tv), C.byref(ok))
    if rc:
        raise RuntimeError(f"bph_range_agg_throughput rc={rc}")
    total = workers * reps
    return dict(ok=bool(ok.value), prove_wall_ms=tp.value, verify_wall_ms=tv.value,
                prove_per_s=total / tp.value * 1e3, verify_per_s=total / tv.value * 1e3)


hlib.bph_range_agg_verify_batch.restype = ci
hlib.bph_range_agg_verify_batch.argtypes = [vp, ci, vp, vp, vp, vp, vp, vp, vp, sz, C.c_uint, C.c_uint64, C.c_uint64, ci, ci, ci,
                                            C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                            C.POINTER(ci), C.POINTER(ci), C.POINTER(C.c_double), C.POINTER(ci)]


def range_aggregate_verify_batch(ctx: Context, curve, gs, hs, ugh_xy, values, gamma, n_bits, batch, seed_rand=1, seed_chal=2,
                                 tamper=False):
    """verify_batch of the range benches (benches/aggregate_range_bench.rs:65-84): values (distinct, m) and
    gamma (distinct, m, 4) give `distinct` proofs; a batch of `batch` verifications cycling through them is
    collected (verify_aux each) and combined by verify_batch_aux (src/msm.rs:91-115), on the device and on
    the host; and once more with every proof's CRS scalars built and added on the device
    (RangeAggregate::verify_batch: direct_ms, ok_direct).  tamper: the last proof of the batch is corrupted
    and all three must reject."""
    gx, gy, _ = _pts(gs)
    hx, hy, _ = _pts(hs)
    values = np.ascontiguousarray(values, dtype=np.uint64)
    gamma = np.ascontiguousarray(gamma, dtype=np.uint64)
    distinct, m = values.shape
    assert gamma.shape == (distinct, m, 4)
    n = m * n_bits
    assert n & (n - 1) == 0 and gx.shape[0] >= n and hx.shape[0] >= n
    ugh_xy = np.ascontiguousarray(ugh_xy, dtype=np.uint64).reshape(24)
    tc, td, th = C.c_double(), C.c_double(), C.c_double()
    okd, okh, okx, tx = ci(), ci(), ci(), C.c_double()
    rc = hlib.bph_range_agg_verify_batch(ctx.h, curve, _p(gx), _p(gy), _p(hx), _p(hy), _p(ugh_xy), _p(values), _p(gamma), m,
                                         n_bits, seed_rand, seed_chal, distinct, batch, int(bool(tamper)), C.byref(tc),
                                         C.byref(td), C.byref(th), C.byref(okd), C.byref(okh), C.byref(tx), C.byref(okx))
    if rc:
        raise RuntimeError(f"bph_range_agg_verify_batch rc={rc}")
    return dict(ok_device=bool(okd.value), ok_host=bool(okh.value), ok_direct=bool(okx.value), collect_ms=tc.value,
                device_merge_ms=td.value, host_merge_ms=th.value, direct_ms=tx.value)


hlib.bph_poly_roundtrip.restype = ci
hlib.bph_poly_roundtrip.argtypes = [vp, ci, vp, vp, vp, vp, sz, sz, vp, C.c_uint64, C.c_uint64, ci, ci, vp, vp, vp, vp, vp, vp,
                                    C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(ci), C.POINTER(ci), C.POINTER(ci),
                                    vp, vp]


def poly_commit_roundtrip(ctx: Context, curve, gs, h_gen_xy, coeffs, r, x, seed_rand=1, seed_chal=2, tamper=False, halo=True):
    """polynomial commitment opening (src/poly_commit): statement, prove, verify, lazy_verify and
    the fold of two deferred checks, through the GPU path.  h_gen_xy: (2, 8) uint64 = affine h and
    the group generator.  coeffs: up to n = len(gs) scalars; r, x: one scalar each."""
    gx, gy, _ = _pts(gs)
    n = gx.shape[0]
    assert n & (n - 1) == 0
    k = n.bit_length() - 1
    coeffs = _f(coeffs)
    rx = np.concatenate([_f(r).reshape(4), _f(x).reshape(4)])
    h_gen_xy = np.ascontiguousarray(h_gen_xy, dtype=np.uint64).reshape(16)
    L = np.zeros((max(k, 1), 12), dtype=np.uint64)
    R = np.zeros((max(k, 1), 12), dtype=np.uint64)
    ar = np.zeros((2, 4), dtype=np.uint64)
    g = np.zeros(9, dtype=np.uint64)
    com = np.zeros(12, dtype=np.uint64)
    ev = np.zeros(4, dtype=np.uint64)
    tp, tv = C.c_double(), C.c_double()
    okv, okl, okh = ci(), ci(), ci()
    want_halo = bool(halo) and not tamper
    ho = np.zeros(48 * k + 65, dtype=np.uint64) if want_halo else None
    hw = np.zeros((n, 4), dtype=np.uint64) if want_halo else None
    rc = hlib.bph_poly_roundtrip(ctx.h, curve, _p(gx), _p(gy), _p(h_gen_xy), _p(coeffs), coeffs.shape[0], n, _p(rx), seed_rand,
                                 seed_chal, int(tamper), int(halo), _p(L), _p(R), _p(ar), _p(g), _p(com), _p(ev),
                                 C.byref(tp), C.byref(tv), C.byref(okv), C.byref(okl), C.byref(okh), _p(ho), _p(hw))
    if rc:
        raise RuntimeError(f"bph_poly_roundtrip rc={rc}")
    res = dict(L=L[:k], R=R[:k], a=ar[0], r=ar[1], g=g, commitment=com, evaluation=ev, prove_ms=tp.value,
               verify_ms=tv.value, verify_ok=bool(okv.value), lazy_ok=bool(okl.value), halo_ok=bool(okh.value))
    if want_halo:
        pos = [0]

        def take(cnt, shape=None):
            v = ho[pos[0]:pos[0] + cnt]
            pos[0] += cnt
            return v if shape is None else v.reshape(shape)

        second = dict(L=take(12 * k, (k, 12)), R=take(12 * k, (k, 12)), a=take(4), r=take(4), g=take(9), commitment=take(12),
                      evaluation=take(4))
        folded = dict(alpha=take(4), x=take(4), commitment=take(12), evaluation=take(4))
        folded.update(L=take(12 * k, (k, 12)), R=take(12 * k, (k, 12)), a=take(4), r=take(4), witness=hw)
        res.update(second=second, folded=folded)
    return res


hlib.bph_poly_verify_batch.restype = ci
hlib.bph_poly_verify_batch.argtypes = [vp, ci, vp, vp, vp, vp, sz, sz, vp, sz, sz, C.c_uint64, C.c_uint64, ci, sz, ci, vp, vp, vp, vp, vp,
                                       C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(ci), C.POINTER(ci)]


def poly_commit_verify_batch(ctx: Context, curve, gs, h_gen_xy, coeffs, r, x, batch=None, seed_rand=1, seed_chal=2, tamper=0,
                             tamper_at=0, reps=1):
    """`count` openings over one CRS (coeffs (count, n_coeffs, 4), r and x (count, 4); opening i uses the seeds + i)
    proved, then checked together (PolyCommit::verify_batch: the MSM over the CRS once for the batch, its scalars
    added up on the device) and one by one.  batch > count cycles through the openings.  tamper: 1 evaluation + 1,
    2 a round dropped, at opening tamper_at of the batch."""
    gx, gy, _ = _pts(gs)
    n = gx.shape[0]
    k = n.bit_length() - 1
    coeffs = np.ascontiguousarray(coeffs, dtype=np.uint64)
    count, n_coeffs = coeffs.shape[0], coeffs.shape[1]
    batch = count if batch is None else batch
    rx = np.ascontiguousarray(np.concatenate([np.asarray(r, dtype=np.uint64).reshape(count, 4),
                                              np.asarray(x, dtype=np.uint64).reshape(count, 4)], axis=1))
    hg = np.ascontiguousarray(h_gen_xy, dtype=np.uint64).reshape(16)
    L = np.zeros((count, max(k, 1), 12), dtype=np.uint64)
    R = np.zeros((count, max(k, 1), 12), dtype=np.uint64)
    ar = np.zeros((count, 8), dtype=np.uint64)
    com, ev = np.zeros((count, 12), dtype=np.uint64), np.zeros((count, 4), dtype=np.uint64)
    tb, te, okb, oke = C.c_double(), C.c_double(), ci(), ci()
    rc = hlib.bph_poly_verify_batch(ctx.h, curve, _p(gx), _p(gy), _p(hg), _p(coeffs), n_coeffs, n, _p(rx), count, batch, seed_rand,
                                    seed_chal, int(tamper), tamper_at, reps, _p(L), _p(R), _p(ar), _p(com), _p(ev), C.byref(tb),
                                    C.byref(te), C.byref(okb), C.byref(oke))
    if rc:
        raise RuntimeError(f"bph_poly_verify_batch rc={rc}")
    return dict(L=L[:, :k], R=R[:, :k], a=ar[:, :4], r=ar[:, 4:], commitment=com, evaluation=ev, batch_ms=tb.value,
                one_by_one_ms=te.value, ok_batch=bool(okb.value), ok_each=bool(oke.value))


def range_bench_inputs(ctx: Context, m, n_bits, seed):
    """Synthetic inputs of the aggregated range proof bench (benches/aggregate_range_bench.rs:30-52):
    a CRS of 2*m*n_bits + 3 random multiples of the secp256k1 generator (gs, hs, then u, g, h),
    m uniform n_bits-bit values and their blinding factors, all from one numpy seed."""
    from .api import SECP256K1
    n = m * n_bits
    p = (1 << 256) - (1 << 32) - 977
    gx = 0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798
    gy = 0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8
    r = 1 << 256
    lim = lambda v: [(v >> (64 * i)) & ((1 << 64) - 1) for i in range(4)]  # noqa: E731
    cnt = 2 * n + 3
    X = np.tile(np.array(lim(gx * r % p), dtype=np.uint64), (cnt, 1))
    Y = np.tile(np.array(lim(gy * r % p), dtype=np.uint64), (cnt, 1))
    rng = np.random.default_rng(seed)
    S = rng.integers(0, 1 << 64, (cnt, 4), dtype=np.uint64)
    S[:, 3] &= np.uint64((1 << 63) - 1)
    px, py, _ = ctx.scale_points(SECP256K1, (X, Y, None), S)
    gs = (px[:n], py[:n], None)
    hs = (px[n:2 * n], py[n:2 * n], None)
    ugh = np.concatenate([px[2 * n:], py[2 * n:]], axis=1)
    hi = (1 << n_bits) if n_bits < 64 else (1 << 64)
    values = rng.integers(0, hi, m, dtype=np.uint64)
    gamma = rng.integers(0, 1 << 64, (m, 4), dtype=np.uint64)
    gamma[:, 3] &= np.uint64((1 << 63) - 1)
    return gs, hs, ugh, values, gamma


# ---- polynomial commitment over several ranks (bp_host_sharded.hpp) ---------------------------------
_GATHER = C.CFUNCTYPE(ci, vp, C.POINTER(C.c_uint64), sz, C.POINTER(C.c_uint64))
hlib.bph_spoly_roundtrip.restype = ci
hlib.bph_spoly_roundtrip.argtypes = [vp, ci, sz, sz, _GATHER, vp, vp, vp, vp, vp, vp, sz, vp, C.c_uint64, C.c_uint64, ci, vp, vp, vp, vp,
                                     vp, vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(ci), C.POINTER(ci),
                                     C.POINTER(ci), vp, vp]
hlib.bph_spoly_roundtrip_threads.restype = ci
hlib.bph_spoly_roundtrip_threads.argtypes = [ci, ci, sz, ci, vp, vp, vp, vp, sz, sz, vp, C.c_uint64, C.c_uint64, ci, ci, vp, vp, vp, vp,
                                             vp, vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(ci), C.POINTER(ci),
                                             C.POINTER(ci), vp, vp]


def _poly_outputs(k, n_w, want_halo):
    L = np.zeros((max(k, 1), 12), dtype=np.uint64)
    R = np.zeros((max(k, 1), 12), dtype=np.uint64)
    ar, g = np.zeros((2, 4), dtype=np.uint64), np.zeros(9, dtype=np.uint64)
    com, ev = np.zeros(12, dtype=np.uint64), np.zeros(4, dtype=np.uint64)
    ho = np.zeros(48 * k + 65, dtype=np.uint64) if want_halo else None
    hw = np.zeros((n_w, 4), dtype=np.uint64) if want_halo else None
    return L, R, ar, g, com, ev, ho, hw


def _poly_result(k, L, R, ar, g, com, ev, ho, hw, tp, tv, okv, okl, okh):
    res = dict(L=L[:k], R=R[:k], a=ar[0], r=ar[1], g=g, commitment=com, evaluation=ev, prove_ms=tp.value, verify_ms=tv.value,
               verify_ok=bool(okv.value), lazy_ok=bool(okl.value), halo_ok=bool(okh.value))
    if ho is not None:
        pos = [0]

        def take(cnt, shape=None):
            v = ho[pos[0]:pos[0] + cnt]
            pos[0] += cnt
            return v if shape is None else v.reshape(shape)

        second = dict(L=take(12 * k, (k, 12)), R=take(12 * k, (k, 12)), a=take(4), r=take(4), g=take(9), commitment=take(12),
                      evaluation=take(4))
        folded = dict(alpha=take(4), x=take(4), commitment=take(12), evaluation=take(4))
        folded.update(L=take(12 * k, (k, 12)), R=take(12 * k, (k, 12)), a=take(4), r=take(4), witness=hw)
        res.update(second=second, folded=folded)
    return res


def sharded_poly_roundtrip_threads(curve, gs, h_gen_xy, coeffs, r, x, world, seed_rand=1, seed_chal=2, tamper=False, halo=True,
                                   first_device=0, n_devices=1):
    """poly_commit_roundtrip with the vectors spread over `world` ranks by residue class of the index, the ranks
    being threads of this process with a context each on devices first_device .. first_device + n_devices - 1
    (round robin): same arguments and same result dictionary; `witness` of the folded statement is the whole
    vector, interleaved from the ranks' slices."""
    gx, gy, _ = _pts(gs)
    n = gx.shape[0]
    assert n & (n - 1) == 0
    k = n.bit_length() - 1
    coeffs = _f(coeffs)
    rx = np.concatenate([_f(r).reshape(4), _f(x).reshape(4)])
    h_gen_xy = np.ascontiguousarray(h_gen_xy, dtype=np.uint64).reshape(16)
    want_halo = bool(halo) and not tamper
    L, R, ar, g, com, ev, ho, hw = _poly_outputs(k, n, want_halo)
    tp, tv, okv, okl, okh = C.c_double(), C.c_double(), ci(), ci(), ci()
    rc = hlib.bph_spoly_roundtrip_threads(first_device, n_devices, world, curve, _p(gx), _p(gy), _p(h_gen_xy), _p(coeffs),
                                          coeffs.shape[0], n, _p(rx), seed_rand, seed_chal, int(tamper), int(want_halo), _p(L), _p(R),
                                          _p(ar), _p(g), _p(com), _p(ev), C.byref(tp), C.byref(tv), C.byref(okv), C.byref(okl),
                                          C.byref(okh), _p(ho), _p(hw))
    if rc:
        raise RuntimeError(f"bph_spoly_roundtrip_threads rc={rc}")
    return _poly_result(k, L, R, ar, g, com, ev, ho, hw, tp, tv, okv, okl, okh)


def sharded_poly_roundtrip(ctx: Context, curve, gs_slice, h_gen_xy, coeffs_slice, coeffs2_slice, n_total, r, x, rank, world, gather,
                           seed_rand=1, seed_chal=2, tamper=False):
    """One rank of the same flow, the ranks being processes: gs_slice and the coefficient slices hold the
    elements rank, rank + world, ... of the whole vectors (n_total / world each, zero-padded); coeffs2_slice
    may be None (no second opening, no fold).  gather(packet: uint64 array) -> (world, len) uint64 array is
    the caller's all-gather of small host packets (see sharded.host_gather).  Every rank returns the same
    dictionary, except `witness` of the folded statement: this rank's slice."""
    gx, gy, _ = _pts(gs_slice)
    nl = gx.shape[0]
    assert nl * world == n_total and n_total & (n_total - 1) == 0
    k = n_total.bit_length() - 1
    c1 = _f(coeffs_slice)
    c2 = None if coeffs2_slice is None else _f(coeffs2_slice)
    assert c1.shape[0] == nl and (c2 is None or c2.shape[0] == nl)
    rx = np.concatenate([_f(r).reshape(4), _f(x).reshape(4)])
    h_gen_xy = np.ascontiguousarray(h_gen_xy, dtype=np.uint64).reshape(16)
    want_halo = c2 is not None and not tamper
    L, R, ar, g, com, ev, ho, hw = _poly_outputs(k, nl, want_halo)
    tp, tv, okv, okl, okh = C.c_double(), C.c_double(), ci(), ci(), ci()
    failure = []

    def cb(_user, mine, length, out):
        try:
            packet = np.ctypeslib.as_array(mine, shape=(length,)).copy()
            got = np.ascontiguousarray(gather(packet), dtype=np.uint64).reshape(world * length)
            np.ctypeslib.as_array(out, shape=(world * length,))[:] = got
            return 0
        except Exception as e:  # an exception must not cross the C frames
            failure.append(e)
            return 1

    cfn = _GATHER(cb)
    rc = hlib.bph_spoly_roundtrip(ctx.h, curve, rank, world, cfn, None, _p(gx), _p(gy), _p(h_gen_xy), _p(c1), _p(c2), n_total, _p(rx),
                                  seed_rand, seed_chal, int(tamper), _p(L), _p(R), _p(ar), _p(g), _p(com), _p(ev), C.byref(tp),
                                  C.byref(tv), C.byref(okv), C.byref(okl), C.byref(okh), _p(ho), _p(hw))
    if failure:
        raise failure[0]
    if rc:
        raise RuntimeError(f"bph_spoly_roundtrip rc={rc}")
    return _poly_result(k, L, R, ar, g, com, ev, ho, hw, tp, tv, okv, okl, okh)


hlib.bph_points_sum.restype = ci
hlib.bph_points_sum.argtypes = [ci, vp, sz, vp]
hlib.bph_scalars_sum.restype = ci
hlib.bph_scalars_sum.argtypes = [ci, vp, sz, vp]


def points_sum(curve, xyz):
    """sum of group results (count x 12 u64, normalised Jacobian) on the host, normalised; needs no GPU"""
    xyz = np.ascontiguousarray(xyz, dtype=np.uint64).reshape(-1, 12)
    out = np.zeros(12, dtype=np.uint64)
    rc = hlib.bph_points_sum(curve, _p(xyz), xyz.shape[0], _p(out))
    if rc:
        raise RuntimeError(f"bph_points_sum rc={rc}")
    return out


def scalars_sum(curve, v):
    """sum of scalars (count x 4 u64, Montgomery limbs) on the host; needs no GPU"""
    v = np.ascontiguousarray(v, dtype=np.uint64).reshape(-1, 4)
    out = np.zeros(4, dtype=np.uint64)
    rc = hlib.bph_scalars_sum(curve, _p(v), v.shape[0], _p(out))
    if rc:
        raise RuntimeError(f"bph_scalars_sum rc={rc}")
    return out
