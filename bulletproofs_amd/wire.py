"""A proof as the byte string of its prover messages, in the order the reference sends them, with the
point and scalar encodings of `bp_points_compress` / `bp_scalars_to_bytes` (ark-serialize, compressed):

    aggregated / single range proof (src/range/aggregate.rs:85,129,159; src/range/mod.rs:81,123,148):
        A, S | T1, T2 | tao_x, mu, t_hat | inner-product argument
    inner-product argument (src/ipa/mod.rs:75,97):   (L_j, R_j) for every round | a, b
    polynomial-commitment opening (src/poly_commit/mod.rs:89,107):   (L_j, R_j) for every round | a, r

The messages are concatenated without framing; the Fiat-Shamir sponge that the reference runs over the
same bytes (spongefish) is not part of this package, and no byte vector of the reference pins the
layout: it restates the published encodings of the arkworks crates."""
from __future__ import annotations

import numpy as np

from .api import POINT_WIRE_BYTES

SCALAR_WIRE_BYTES = 32


def _planes(xyz):
    """normalised Jacobian results (n, 12) -> affine planes; Z = 0 marks the identity"""
    xyz = np.ascontiguousarray(xyz, dtype=np.uint64).reshape(-1, 12)
    inf = (~xyz[:, 8:12].any(axis=1)).astype(np.uint8)
    x, y = np.ascontiguousarray(xyz[:, 0:4]), np.ascontiguousarray(xyz[:, 4:8])
    x[inf == 1] = 0
    y[inf == 1] = 0
    return x, y, inf


def _xyz(x, y, inf, one):
    out = np.zeros((x.shape[0], 12), dtype=np.uint64)
    out[:, 0:4], out[:, 4:8] = x, y
    out[:, 8:12] = one
    idle = inf == 1
    out[idle, 0:4] = one   # the identity as the library returns it: (1, 1, 0)
    out[idle, 4:8] = one
    out[idle, 8:12] = 0
    return out


def points_to_bytes(ctx, curve, xyz) -> bytes:
    return ctx.points_compress(curve, _planes(xyz)).tobytes()


def scalars_to_bytes(ctx, curve, v) -> bytes:
    return ctx.scalars_to_bytes(curve, np.ascontiguousarray(v, dtype=np.uint64).reshape(-1, 4)).tobytes()


def ipa_proof_to_bytes(ctx, curve, L, R, a, b) -> bytes:
    """(L_j, R_j) per round, then the two final scalars (a, b of ipa::prove; a, r of poly_commit::prove)"""
    L = np.ascontiguousarray(L, dtype=np.uint64).reshape(-1, 12)
    R = np.ascontiguousarray(R, dtype=np.uint64).reshape(-1, 12)
    lr = np.stack([L, R], axis=1).reshape(-1, 12)  # L_0, R_0, L_1, R_1, ...
    return points_to_bytes(ctx, curve, lr) + scalars_to_bytes(ctx, curve, np.stack([a, b]))


def range_proof_to_bytes(ctx, curve, proof) -> bytes:
    """proof: the dict of host_mirror.range_aggregate_roundtrip (A, S, TT1, TT2, tao_x, mu, t_hat, L, R, a, b)"""
    head = points_to_bytes(ctx, curve, np.stack([proof["A"], proof["S"], proof["TT1"], proof["TT2"]]))
    mid = scalars_to_bytes(ctx, curve, np.stack([proof["tao_x"], proof["mu"], proof["t_hat"]]))
    return head + mid + ipa_proof_to_bytes(ctx, curve, proof["L"], proof["R"], proof["a"], proof["b"])


def _mont_one(curve, field_modulus):
    v = (1 << 256) % field_modulus
    return np.array([(v >> (64 * i)) & ((1 << 64) - 1) for i in range(4)], dtype=np.uint64)


_FQ = {0: (1 << 256) - (1 << 32) - 977, 1: 0x40000000000000000000000000000000224698FC0994A8DD8C46EB2100000001}


def _take_points(ctx, curve, data, count):
    raw = np.frombuffer(data[:count * POINT_WIRE_BYTES], dtype=np.uint8).reshape(count, POINT_WIRE_BYTES)
    (x, y, inf), ok = ctx.points_decompress(curve, raw)
    return _xyz(x, y, inf, _mont_one(curve, _FQ[curve])), bool(ok.all()), data[count * POINT_WIRE_BYTES:]


def _take_scalars(ctx, curve, data, count):
    raw = np.frombuffer(data[:count * SCALAR_WIRE_BYTES], dtype=np.uint8).reshape(count, SCALAR_WIRE_BYTES)
    v, ok = ctx.scalars_from_bytes(curve, raw)
    return v, bool(ok.all()), data[count * SCALAR_WIRE_BYTES:]


def range_proof_len(rounds: int) -> int:
    return (4 + 2 * rounds) * POINT_WIRE_BYTES + 5 * SCALAR_WIRE_BYTES


def range_proof_from_bytes(ctx, curve, data: bytes, rounds: int):
    """-> (proof dict as range_proof_to_bytes takes it, ok).  ok is False when the length does not fit the
    number of rounds of the statement or any element is no point / no scalar (the reference's
    deserialisation error); the dict is then None."""
    if len(data) != range_proof_len(rounds):
        return None, False
    head, ok1, data = _take_points(ctx, curve, data, 4)
    mid, ok2, data = _take_scalars(ctx, curve, data, 3)
    lr, ok3, data = _take_points(ctx, curve, data, 2 * rounds) if rounds else (np.zeros((0, 12), dtype=np.uint64), True, data)
    ab, ok4, data = _take_scalars(ctx, curve, data, 2)
    if not (ok1 and ok2 and ok3 and ok4):
        return None, False
    lr = lr.reshape(rounds, 2, 12)
    return dict(A=head[0], S=head[1], TT1=head[2], TT2=head[3], tao_x=mid[0], mu=mid[1], t_hat=mid[2],
                L=np.ascontiguousarray(lr[:, 0]), R=np.ascontiguousarray(lr[:, 1]), a=ab[0], b=ab[1]), True
