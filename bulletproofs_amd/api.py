"""Host-side Python mirror of the reference's hot-path functions on top of the
C ABI.  Names follow the reference: msm_unchecked (ark-ec, called from
src/msm.rs:84-87), inner_product / sum (src/vector_ops.rs:159-179),
fold_scalars / fold_generators (src/ipa/mod.rs:215-234), normalize_batch,
batch_inversion, HPoly::coeffs (s_vector).

Values are numpy arrays in the arkworks memory form: a field vector is (n,4)
uint64 Montgomery limbs; a point vector is a tuple (x (n,4), y (n,4), inf (n,)
uint8); a group result is 12 uint64 (Jacobian X,Y,Z, normalised).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import lib

SECP256K1, VESTA = 0, 1
VEC_ADD, VEC_SUB, VEC_HADAMARD, VEC_SCALE, VEC_AXPY = 0, 1, 2, 3, 4
IPA_BULLETPROOF, IPA_POLYCOMMIT = 0, 1
POINT_WIRE_BYTES = 33  # ark-serialize, Compress::Yes, on both curves


class BpError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"{_lib.ERR_NAMES.get(code, code)}: {msg}")
        self.code = code


def _f(a):
    """field vector -> contiguous (n,4) uint64"""
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return a.reshape(-1, 4)


def _p(a):
    return None if a is None else a.ctypes.data


def _pts(P):
    x, y, inf = P
    x, y = _f(x), _f(y)
    if x.shape != y.shape:
        raise ValueError("x/y planes differ in length")
    if inf is not None:
        inf = np.ascontiguousarray(inf, dtype=np.uint8).reshape(-1)
        if inf.shape[0] != x.shape[0]:
            raise ValueError("infinity plane length mismatch")
    return x, y, inf


AOS_STRIDE = 72


def pack_aos(bases) -> np.ndarray:
    """(x, y, inf) planes -> (n, 72)-byte records {x: 32, y: 32, infinity: 1, padding: 7}, the
    memory image of a slice of arkworks short-Weierstrass Affine points over a 4-limb field"""
    x, y, inf = _pts(bases)
    n = x.shape[0]
    rec = np.zeros((n, AOS_STRIDE), dtype=np.uint8)
    rec[:, 0:32] = x.view(np.uint8).reshape(n, 32)
    rec[:, 32:64] = y.view(np.uint8).reshape(n, 32)
    if inf is not None:
        rec[:, 64] = (inf != 0)
    return rec


def shard_window_bits(n_total: int, n_shards: int) -> int:
    """window width every shard of an n_total-point MSM over n_shards ranges uses (host only)"""
    c = lib.bp_msm_shard_window_bits(n_total, n_shards)
    if c < 2:
        raise BpError(c, "bp_msm_shard_window_bits: bad argument")
    return c


class Context:
    """One GPU + one stream (bp_ctx)."""

    def __init__(self, device: int = 0):
        h = C.c_void_p()
        rc = lib.bp_init(device, C.byref(h))
        if rc != 0:
            raise BpError(rc, "bp_init failed (no usable gfx950 device?)")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            lib.bp_shutdown(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise BpError(rc, lib.bp_last_error(self.h).decode())

    def set_msm_window(self, c: int):
        self._chk(lib.bp_set_msm_window(self.h, c))

    def sync(self):
        self._chk(lib.bp_sync(self.h))

    # ---- MSM ------------------------------------------------------------
    def msm_unchecked(self, curve, bases, scalars):
        x, y, inf = _pts(bases)
        s = _f(scalars)
        n = min(x.shape[0], s.shape[0])  # msm_unchecked zips to the shorter length
        out = np.zeros(12, dtype=np.uint64)
        self._chk(lib.bp_msm(self.h, curve, _p(x), _p(y), _p(inf), _p(s), n, _p(out)))
        return out

    def msm_aos(self, curve, records, scalars):
        """bases as an (n, 72)-byte array of {x, y, infinity, padding} records (pack_aos)"""
        rec = np.ascontiguousarray(records, dtype=np.uint8)
        if rec.ndim != 2 or rec.shape[1] != AOS_STRIDE:
            raise ValueError("records must be (n, 72) bytes")
        s = _f(scalars)
        n = min(rec.shape[0], s.shape[0])
        out = np.zeros(12, dtype=np.uint64)
        self._chk(lib.bp_msm_aos(self.h, curve, _p(rec), _p(s), n, _p(out)))
        return out

    def register_bases_aos(self, curve, records):
        return Bases(self, curve, None, records=records)

    def msm_dev(self, curve, d_x, d_y, d_inf, d_s, n):
        out = np.zeros(12, dtype=np.uint64)
        self._chk(lib.bp_msm_dev(self.h, curve, d_x, d_y, d_inf, d_s, n, _p(out)))
        return out

    def msm2_dev(self, curve, part0, part1):
        out = np.zeros(12, dtype=np.uint64)
        self._chk(lib.bp_msm2_dev(self.h, curve, *part0, *part1, _p(out)))
        return out

    def register_bases(self, curve, bases):
        return Bases(self, curve, bases)

    def register_bases_compressed(self, curve, data):
        """base set from (n, 33) compressed points, decoded on the device (bp_bases_register_compressed)"""
        return Bases(self, curve, None, compressed=data)

    def msm_windows_dev(self, curve, d_x, d_y, d_inf, d_s, n, d_windows):
        nw, cb = C.c_int(), C.c_int()
        self._chk(lib.bp_msm_windows_dev(self.h, curve, d_x, d_y, d_inf, d_s, n, d_windows, C.byref(nw), C.byref(cb)))
        return nw.value, cb.value

    def msm_shard_windows_dev(self, curve, d_x, d_y, d_inf, d_s, n, window_bits, d_windows):
        """window sums of one shard of a multi-GPU MSM, at the width shard_window_bits gave"""
        nw = C.c_int()
        self._chk(lib.bp_msm_shard_windows_dev(self.h, curve, d_x, d_y, d_inf, d_s, n, window_bits, d_windows, C.byref(nw)))
        return nw.value

    def msm_combine_windows_dev(self, curve, d_windows, n_parts, n_windows, window_bits):
        out = np.zeros(12, dtype=np.uint64)
        self._chk(lib.bp_msm_combine_windows_dev(self.h, curve, d_windows, n_parts, n_windows, window_bits, _p(out)))
        return out

    # ---- field vectors ----------------------------------------------------
    def inner_product(self, curve, a, b):
        a, b = _f(a), _f(b)
        n = min(a.shape[0], b.shape[0])
        out = np.zeros(4, dtype=np.uint64)
        self._chk(lib.bp_inner_product(self.h, curve, _p(a), _p(b), n, _p(out)))
        return out

    def sum(self, curve, a):
        a = _f(a)
        out = np.zeros(4, dtype=np.uint64)
        self._chk(lib.bp_sum(self.h, curve, _p(a), a.shape[0], _p(out)))
        return out

    def fold_scalars(self, curve, left, right, x, y):
        l, r = _f(left), _f(right)
        h = min(l.shape[0], r.shape[0])
        x, y = _f(x), _f(y)
        out = np.zeros((h, 4), dtype=np.uint64)
        self._chk(lib.bp_fold_scalars(self.h, curve, _p(l), _p(r), h, _p(x), _p(y), _p(out)))
        return out

    def vec_op(self, curve, op, a, b=None, s=None):
        a = _f(a)
        n = a.shape[0]
        if b is not None:
            b = _f(b)
            n = min(n, b.shape[0])
        if s is not None:
            s = _f(s)
        out = np.zeros((n, 4), dtype=np.uint64)
        self._chk(lib.bp_vec_op(self.h, curve, op, _p(a), _p(b), _p(s), n, _p(out)))
        return out

    def vector_add(self, curve, a, b):
        return self.vec_op(curve, VEC_ADD, a, b)

    def vector_sub(self, curve, a, b):
        return self.vec_op(curve, VEC_SUB, a, b)

    def hadamard(self, curve, a, b):
        return self.vec_op(curve, VEC_HADAMARD, a, b)

    def scale(self, curve, a, s):
        return self.vec_op(curve, VEC_SCALE, a, None, s)

    def axpy(self, curve, a, s, b):
        return self.vec_op(curve, VEC_AXPY, a, b, s)

    def s_vector(self, curve, challenges, challenge_inverses):
        ch, chi = _f(challenges), _f(challenge_inverses)
        k = ch.shape[0]
        if chi.shape[0] != k:
            raise ValueError("challenge / inverse count mismatch")
        out = np.zeros((1 << k, 4), dtype=np.uint64)
        self._chk(lib.bp_s_vector(self.h, curve, _p(ch), _p(chi), k, _p(out)))
        return out

    def fold_s_vectors(self, curve, challenges, challenge_inverses, weights):
        """sum_i weights[i] * s_vector(challenges[i], challenge_inverses[i]); challenges: (m, k, 4)"""
        ch = np.ascontiguousarray(challenges, dtype=np.uint64)
        chi = np.ascontiguousarray(challenge_inverses, dtype=np.uint64)
        w = _f(weights)
        if ch.ndim != 3 or ch.shape != chi.shape or ch.shape[2] != 4 or w.shape[0] != ch.shape[0] or ch.shape[0] == 0:
            raise ValueError("challenges must be (m, k, 4) with m >= 1, one weight per row")
        m, k = ch.shape[0], ch.shape[1]
        out = np.zeros((1 << k, 4), dtype=np.uint64)
        self._chk(lib.bp_fold_s_vectors(self.h, curve, _p(ch), _p(chi), _p(w), k, m, _p(out)))
        return out

    def range_agg_poly(self, curve, values, n_bits, s_l, s_r, y, z):
        """vector polynomials of the aggregated range proof (src/range/aggregate.rs:88-127) built on
        the device; returns (t1, t2) and leaves l0, l1, r0, r1 in the context for range_agg_eval"""
        v = np.ascontiguousarray(values, dtype=np.uint64)
        sl, sr, y, z = _f(s_l), _f(s_r), _f(y), _f(z)
        n = v.shape[0] * n_bits
        if sl.shape[0] != n or sr.shape[0] != n:
            raise ValueError("one blinder per bit of every value")
        t1 = np.zeros((1, 4), dtype=np.uint64)
        t2 = np.zeros((1, 4), dtype=np.uint64)
        self._chk(lib.bp_range_agg_poly(self.h, curve, _p(v), v.shape[0], n_bits, _p(sl), _p(sr), _p(y), _p(z),
                                        _p(t1), _p(t2)))
        return t1, t2

    def range_agg_eval(self, curve, x, n):
        """l = l0 + x*l1, r = r0 + x*r1, t_hat = <l, r> (src/range/aggregate.rs:143-146)"""
        x = _f(x)
        l = np.zeros((n, 4), dtype=np.uint64)
        r = np.zeros((n, 4), dtype=np.uint64)
        t = np.zeros((1, 4), dtype=np.uint64)
        self._chk(lib.bp_range_agg_eval(self.h, curve, _p(x), n, _p(l), _p(r), _p(t)))
        return l, r, t

    def range_agg_verify_scalars(self, curve, challenges, challenge_inverses, a, b, y_inv, z, m, n_bits):
        """scalars of gs_i and hs_i in the verifier's single deferred MSM (see bp_gpu.h)"""
        ch, chi = _f(challenges), _f(challenge_inverses)
        k = ch.shape[0]
        n = 1 << k
        og = np.zeros((n, 4), dtype=np.uint64)
        oh = np.zeros((n, 4), dtype=np.uint64)
        self._chk(lib.bp_range_agg_verify_scalars(self.h, curve, _p(ch), _p(chi), k, _p(_f(a)), _p(_f(b)), _p(_f(y_inv)),
                                                  _p(_f(z)), m, n_bits, _p(og), _p(oh)))
        return og, oh

    def powers(self, curve, x, n):
        x = _f(x)
        out = np.zeros((n, 4), dtype=np.uint64)
        self._chk(lib.bp_powers(self.h, curve, _p(x), n, _p(out)))
        return out

    def batch_inversion(self, curve, v, field=0):
        v = _f(v).copy()
        self._chk(lib.bp_batch_inverse(self.h, curve, field, _p(v), v.shape[0]))
        return v

    # ---- curve vectors ------------------------------------------------------
    def fold_generators(self, curve, left, right, x, y):
        lx, ly, li = _pts(left)
        rx, ry, ri = _pts(right)
        h = min(lx.shape[0], rx.shape[0])
        x, y = _f(x), _f(y)
        ox = np.zeros((h, 4), dtype=np.uint64)
        oy = np.zeros((h, 4), dtype=np.uint64)
        oi = np.zeros(h, dtype=np.uint8)
        self._chk(
            lib.bp_fold_generators(self.h, curve, _p(lx), _p(ly), _p(li), _p(rx), _p(ry), _p(ri), h, _p(x), _p(y),
                                   _p(ox), _p(oy), _p(oi)))
        return ox, oy, oi

    def scale_points(self, curve, points, scalars):
        px, py, pi = _pts(points)
        s = _f(scalars)
        n = min(px.shape[0], s.shape[0])
        ox = np.zeros((n, 4), dtype=np.uint64)
        oy = np.zeros((n, 4), dtype=np.uint64)
        oi = np.zeros(n, dtype=np.uint8)
        self._chk(lib.bp_scale_points(self.h, curve, _p(px), _p(py), _p(pi), _p(s), n, _p(ox), _p(oy), _p(oi)))
        return ox, oy, oi

    def scalars_to_bytes(self, curve, v):
        """(n, 32) bytes: canonical little-endian integers of scalar-field elements (bp_scalars_to_bytes)"""
        v = _f(v)
        out = np.zeros((v.shape[0], 32), dtype=np.uint8)
        self._chk(lib.bp_scalars_to_bytes(self.h, curve, _p(v), v.shape[0], _p(out)))
        return out

    def scalars_from_bytes(self, curve, data):
        """(n, 32) bytes -> Montgomery limbs, ok (0 where the integer is not below the group order)"""
        data = np.ascontiguousarray(data, dtype=np.uint8).reshape(-1, 32)
        n = data.shape[0]
        out = np.zeros((n, 4), dtype=np.uint64)
        ok = np.zeros(n, dtype=np.uint8)
        self._chk(lib.bp_scalars_from_bytes(self.h, curve, _p(data), n, _p(out), _p(ok)))
        return out, ok

    def points_compress(self, curve, points):
        """ark-serialize's compressed encoding of affine points: (n, 33) bytes (bp_points_compress)"""
        px, py, pi = _pts(points)
        n = px.shape[0]
        out = np.zeros((n, POINT_WIRE_BYTES), dtype=np.uint8)
        self._chk(lib.bp_points_compress(self.h, curve, _p(px), _p(py), _p(pi), n, _p(out)))
        return out

    def points_decompress(self, curve, data):
        """(n, 33) bytes -> (x, y, inf), ok: y recomputed on the device; ok[i] = 0 where the bytes are no point"""
        data = np.ascontiguousarray(data, dtype=np.uint8).reshape(-1, POINT_WIRE_BYTES)
        n = data.shape[0]
        ox = np.zeros((n, 4), dtype=np.uint64)
        oy = np.zeros((n, 4), dtype=np.uint64)
        oi = np.zeros(n, dtype=np.uint8)
        ok = np.zeros(n, dtype=np.uint8)
        self._chk(lib.bp_points_decompress(self.h, curve, _p(data), n, _p(ox), _p(oy), _p(oi), _p(ok)))
        return (ox, oy, oi), ok

    def normalize_batch(self, curve, xyz):
        xyz = np.ascontiguousarray(xyz, dtype=np.uint64).reshape(-1, 12)
        n = xyz.shape[0]
        ox = np.zeros((n, 4), dtype=np.uint64)
        oy = np.zeros((n, 4), dtype=np.uint64)
        oi = np.zeros(n, dtype=np.uint8)
        self._chk(lib.bp_normalize_batch(self.h, curve, _p(xyz), n, _p(ox), _p(oy), _p(oi)))
        return ox, oy, oi

    # ---- measurement ----------------------------------------------------------
    def int_mad_peak(self, iters=2000):
        v = C.c_double()
        self._chk(lib.bp_bench_int_mad_peak(self.h, iters, C.byref(v)))
        return v.value

    def set_timing(self, on):
        """event markers around MSMs (bp_last_kernel_ms) on or off"""
        self._chk(lib.bp_set_timing(self.h, 1 if on else 0))

    def last_kernel_ms(self, group):
        v = C.c_float()
        self._chk(lib.bp_last_kernel_ms(self.h, group, C.byref(v)))
        return v.value

    def ipa_begin(self, curve, variant, gs, hs, a, b):
        return IpaSession(self, curve, variant, gs, hs, a, b)

    def ipa_begin_resident(self, bases, variant, g_offset, h_offset, a, b):
        return IpaSession(self, bases.curve, variant, None, None, a, b, resident=bases, g_offset=g_offset, h_offset=h_offset)

    def ipa_begin_resident_powers(self, bases, variant, g_offset, h_offset, a, x):
        """as ipa_begin_resident with b = (1, x, x^2, ...) built on the device"""
        return IpaSession(self, bases.curve, variant, None, None, a, None, resident=bases, g_offset=g_offset,
                          h_offset=h_offset, b_power=x)


class Bases:
    """Device-resident base points (bp_bases): the CRS uploaded once."""

    def __init__(self, ctx: Context, curve, bases, records=None, compressed=None):
        h = C.c_void_p()
        if compressed is not None:
            data = np.ascontiguousarray(compressed, dtype=np.uint8).reshape(-1, POINT_WIRE_BYTES)
            ctx._chk(lib.bp_bases_register_compressed(ctx.h, curve, _p(data), data.shape[0], C.byref(h)))
        elif records is not None:
            rec = np.ascontiguousarray(records, dtype=np.uint8)
            if rec.ndim != 2 or rec.shape[1] != AOS_STRIDE:
                raise ValueError("records must be (n, 72) bytes")
            ctx._chk(lib.bp_bases_register_aos(ctx.h, curve, _p(rec), rec.shape[0], C.byref(h)))
        else:
            x, y, inf = _pts(bases)
            ctx._chk(lib.bp_bases_register(ctx.h, curve, _p(x), _p(y), _p(inf), x.shape[0], C.byref(h)))
        self.ctx, self.h, self.curve = ctx, h, curve

    def __len__(self):
        return lib.bp_bases_len(self.h)

    def ipa_verify_msm(self, g_offset, h_offset, u_index, challenges, challenge_inverses, a, b, u_scalar, extra, extra_scalars):
        """the inner-product verifier's deferred MSM with its scalars built on the device
        (bp_ipa_verify_msm; src/ipa/mod.rs:102-193): returns the group element, the identity iff valid"""
        ch = np.ascontiguousarray(challenges, dtype=np.uint64).reshape(-1, 4)
        chi = np.ascontiguousarray(challenge_inverses, dtype=np.uint64).reshape(-1, 4)
        ex, ey, ei = _pts(extra)
        es, a, b, us = _f(extra_scalars), _f(a), _f(b), _f(u_scalar)  # named: the arrays must outlive the call
        out = np.zeros(12, dtype=np.uint64)
        self.ctx._chk(lib.bp_ipa_verify_msm(self.ctx.h, self.h, g_offset, h_offset, u_index, _p(ch), _p(chi), ch.shape[0],
                                            _p(a), _p(b), _p(us), _p(ex), _p(ey), _p(ei), _p(es), ex.shape[0], _p(out)))
        return out

    def precompute(self):
        """window tables for this base set (bp_bases_precompute)"""
        self.ctx._chk(lib.bp_bases_precompute(self.h))
        return self

    def msm_s_vector(self, challenges, challenge_inverses, offset=0):
        """<s-vector of the challenges, bases[offset, offset + 2^k)> with the scalars built on the device
        (bp_msm_resident_s_vector; src/poly_commit/mod.rs:109-111,152-159)"""
        ch = np.ascontiguousarray(challenges, dtype=np.uint64).reshape(-1, 4)
        chi = np.ascontiguousarray(challenge_inverses, dtype=np.uint64).reshape(-1, 4)
        out = np.zeros(12, dtype=np.uint64)
        self.ctx._chk(lib.bp_msm_resident_s_vector(self.ctx.h, self.h, offset, _p(ch), _p(chi), ch.shape[0], _p(out)))
        return out

    def msm(self, scalars, offset=0, n=None):
        s = _f(scalars)
        if n is None:
            n = s.shape[0]
        out = np.zeros(12, dtype=np.uint64)
        self.ctx._chk(lib.bp_msm_resident(self.ctx.h, self.h, offset, n, _p(s), _p(out)))
        return out

    def msm_plus(self, scalars, extra_points, extra_scalars, offset=0, n=None):
        """sum over registered points plus a few uploaded ones (a verifier's deferred MSM)"""
        s = _f(scalars)
        if n is None:
            n = s.shape[0]
        ex, ey, einf = _pts(extra_points)
        es = _f(extra_scalars)
        k = min(ex.shape[0], es.shape[0])
        out = np.zeros(12, dtype=np.uint64)
        self.ctx._chk(lib.bp_msm_resident_plus(self.ctx.h, self.h, offset, n, _p(s), _p(ex), _p(ey), _p(einf), _p(es), k,
                                               _p(out)))
        return out

    def shard_windows(self, scalars, window_bits, d_windows, offset=0, n=None):
        """window sums of this GPU's range of a sharded base set (bp_msm_shard_windows_resident)"""
        s = _f(scalars)
        if n is None:
            n = s.shape[0]
        nw = C.c_int()
        self.ctx._chk(lib.bp_msm_shard_windows_resident(self.ctx.h, self.h, offset, n, _p(s), window_bits, d_windows,
                                                        C.byref(nw)))
        return nw.value

    def msm_pair(self, scalars_a, scalars_b, offset=0, n=None):
        """two sums over the same points in one pass (the A and S commitments of the range proofs)"""
        a, b = _f(scalars_a), _f(scalars_b)
        if n is None:
            n = min(a.shape[0], b.shape[0])
        if a.shape[0] < n or b.shape[0] < n:
            raise ValueError("fewer scalars than points")
        oa = np.zeros(12, dtype=np.uint64)
        ob = np.zeros(12, dtype=np.uint64)
        self.ctx._chk(lib.bp_msm_resident_pair(self.ctx.h, self.h, offset, n, _p(a), _p(b), _p(oa), _p(ob)))
        return oa, ob

    def free(self):
        if self.h:
            lib.bp_bases_free(self.h)
            self.h = None

    def __del__(self):
        try:
            if self.ctx.h:
                self.free()
        except Exception:
            pass


class IpaSession:
    """Device-resident G, H, a, b across the rounds of ipa::prove
    (src/ipa/mod.rs:33-95) or poly_commit::prove (src/poly_commit/mod.rs:69-105)."""

    def __init__(self, ctx: Context, curve, variant, gs, hs, a, b, resident=None, g_offset=0, h_offset=0, b_power=None):
        if resident is not None and b_power is not None:
            # b_i = b_power^i built on the device (bp_ipa_begin_resident_powers; src/poly_commit/mod.rs:65)
            a, x = _f(a), _f(b_power)
            h = C.c_void_p()
            ctx._chk(lib.bp_ipa_begin_resident_powers(ctx.h, resident.h, variant, g_offset, h_offset, _p(a), _p(x),
                                                      a.shape[0], C.byref(h)))
            self.ctx, self.h, self.has_h, self._bases = ctx, h, variant == 0, resident
            return
        a, b = _f(a), _f(b)
        n = a.shape[0]
        if resident is not None:
            # generators come from a registered base set (bp_ipa_begin_resident), not copied
            h = C.c_void_p()
            ctx._chk(lib.bp_ipa_begin_resident(ctx.h, resident.h, variant, g_offset, h_offset, _p(a), _p(b), n, C.byref(h)))
            self.ctx, self.h, self.has_h, self._bases = ctx, h, variant == 0, resident
            return
        gx, gy, gi = _pts(gs)
        if hs is not None:
            hx, hy, hi = _pts(hs)
        else:
            hx = hy = hi = None
        h = C.c_void_p()
        ctx._chk(
            lib.bp_ipa_begin(ctx.h, curve, variant, _p(gx), _p(gy), _p(gi), _p(hx), _p(hy), _p(hi), _p(a), _p(b), n,
                             C.byref(h)))
        self.ctx, self.h, self.has_h = ctx, h, hs is not None

    def __len__(self):
        return lib.bp_ipa_len(self.h)

    def round_lr(self):
        L = np.zeros(12, dtype=np.uint64)
        R = np.zeros(12, dtype=np.uint64)
        cl = np.zeros(4, dtype=np.uint64)
        cr = np.zeros(4, dtype=np.uint64)
        self.ctx._chk(lib.bp_ipa_round_lr(self.h, _p(L), _p(R), _p(cl), _p(cr)))
        return L, R, cl, cr

    def scale_h(self, scale):
        """session continues as if H_i were scale_i * H_i (hs' of src/range/aggregate.rs:148-153)"""
        sc = _f(scale)
        self.ctx._chk(lib.bp_ipa_scale_h(self.h, _p(sc)))

    def set_u(self, u_xy):
        """u as 8 u64 (x|y): round_lr then returns L, R including c_l*u, c_r*u"""
        u = None if u_xy is None else np.ascontiguousarray(u_xy, dtype=np.uint64).reshape(8)
        self.ctx._chk(lib.bp_ipa_set_u(self.h, _p(u)))

    def set_u_resident(self, u_index, factor):
        """u = factor * bases[u_index] (bp_ipa_set_u_resident)"""
        f = _f(factor)
        self.ctx._chk(lib.bp_ipa_set_u_resident(self.h, u_index, _p(f)))

    def round_fold(self, x, x_inv):
        x, xi = _f(x), _f(x_inv)
        self.ctx._chk(lib.bp_ipa_round_fold(self.h, _p(x), _p(xi)))

    def read(self):
        n = len(self)
        gx = np.zeros((n, 4), dtype=np.uint64)
        gy = np.zeros((n, 4), dtype=np.uint64)
        gi = np.zeros(n, dtype=np.uint8)
        a = np.zeros((n, 4), dtype=np.uint64)
        b = np.zeros((n, 4), dtype=np.uint64)
        if self.has_h:
            hx = np.zeros((n, 4), dtype=np.uint64)
            hy = np.zeros((n, 4), dtype=np.uint64)
            hi = np.zeros(n, dtype=np.uint8)
        else:
            hx = hy = hi = None
        self.ctx._chk(lib.bp_ipa_read(self.h, _p(gx), _p(gy), _p(gi), _p(hx), _p(hy), _p(hi), _p(a), _p(b)))
        return (gx, gy, gi), (None if hx is None else (hx, hy, hi)), a, b

    def free(self):
        if self.h:
            lib.bp_ipa_free(self.h)
            self.h = None

    def __del__(self):
        try:
            if self.ctx.h:
                self.free()
        except Exception:
            pass
