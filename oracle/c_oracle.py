"""ctypes loader for oracle/_build/libbp_oracle.so (the C restatement in
bp_oracle.c).  TEST INFRASTRUCTURE ONLY -- see the header of bp_oracle.c."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libbp_oracle.so")


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def load():
    if not os.path.exists(_SO):
        build()
    lib = C.CDLL(_SO)
    for name in ("bpo_inner_product", "bpo_sum", "bpo_fold_scalars", "bpo_s_vector", "bpo_fold_generators",
                 "bpo_scale_points", "bpo_msm", "bpo_msm_naive", "bpo_gen_points", "bpo_max_threads",
                 "bpo_ipa_statement", "bpo_ipa_prove", "bpo_ipa_verify", "bpo_range_agg_prove", "bpo_range_agg_verify",
                 "bpo_range_value_commitments", "bpo_poly_statement", "bpo_poly_prove", "bpo_poly_verify",
                 "bpo_poly_fold_todos"):
        getattr(lib, name).restype = C.c_int
    return lib


def _p(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


class COracle:
    def __init__(self):
        self.lib = load()

    def max_threads(self):
        return self.lib.bpo_max_threads()

    def inner_product(self, curve, a, b):
        out = np.zeros(4, dtype=np.uint64)
        n = min(len(a), len(b))
        assert self.lib.bpo_inner_product(curve, _p(a), _p(b), C.c_size_t(n), _p(out)) == 0
        return out

    def sum(self, curve, a):
        out = np.zeros(4, dtype=np.uint64)
        assert self.lib.bpo_sum(curve, _p(a), C.c_size_t(len(a)), _p(out)) == 0
        return out

    def fold_scalars(self, curve, l, r, x, y, threads=1):
        h = min(len(l), len(r))
        out = np.zeros((h, 4), dtype=np.uint64)
        assert self.lib.bpo_fold_scalars(curve, _p(l), _p(r), C.c_size_t(h), _p(x), _p(y), _p(out), threads) == 0
        return out

    def s_vector(self, curve, ch, chi, threads=1):
        k = len(ch)
        out = np.zeros((1 << k, 4), dtype=np.uint64)
        assert self.lib.bpo_s_vector(curve, _p(ch), _p(chi), C.c_size_t(k), _p(out), threads) == 0
        return out

    def fold_generators(self, curve, L, R, x, y, threads=1):
        h = min(len(L[0]), len(R[0]))
        ox = np.zeros((h, 4), dtype=np.uint64)
        oy = np.zeros((h, 4), dtype=np.uint64)
        oi = np.zeros(h, dtype=np.uint8)
        assert self.lib.bpo_fold_generators(curve, _p(L[0]), _p(L[1]), _p(L[2]), _p(R[0]), _p(R[1]), _p(R[2]),
                                            C.c_size_t(h), _p(x), _p(y), _p(ox), _p(oy), _p(oi), threads) == 0
        return ox, oy, oi

    def scale_points(self, curve, P, s, threads=1):
        n = min(len(P[0]), len(s))
        ox = np.zeros((n, 4), dtype=np.uint64)
        oy = np.zeros((n, 4), dtype=np.uint64)
        oi = np.zeros(n, dtype=np.uint8)
        assert self.lib.bpo_scale_points(curve, _p(P[0]), _p(P[1]), _p(P[2]), _p(s), C.c_size_t(n), _p(ox), _p(oy),
                                         _p(oi), threads) == 0
        return ox, oy, oi

    def msm(self, curve, P, s, threads=1):
        n = min(len(P[0]), len(s))
        out = np.zeros(12, dtype=np.uint64)
        assert self.lib.bpo_msm(curve, _p(P[0]), _p(P[1]), _p(P[2]), _p(s), C.c_size_t(n), threads, _p(out)) == 0
        return out

    def msm_naive(self, curve, P, s):
        n = min(len(P[0]), len(s))
        out = np.zeros(12, dtype=np.uint64)
        assert self.lib.bpo_msm_naive(curve, _p(P[0]), _p(P[1]), _p(P[2]), _p(s), C.c_size_t(n), _p(out)) == 0
        return out

    def gen_points(self, curve, start_xy, step_xy, count):
        ox = np.zeros((count, 4), dtype=np.uint64)
        oy = np.zeros((count, 4), dtype=np.uint64)
        assert self.lib.bpo_gen_points(curve, _p(start_xy), _p(step_xy), C.c_size_t(count), _p(ox), _p(oy)) == 0
        return ox, oy

    # ---- protocol drivers (bp_oracle_proto.c) ------------------------------------------------
    # Points are (x, y, inf) plane triples as produced by bp_oracle.pack_points; single points
    # are 8 u64 (x | y); group results 12 u64 (X, Y, Z with Z = 1, identity = (1, 1, 0)).

    def ipa_statement(self, curve, G, H, u_xy, a, b, threads=1):
        n = len(a)
        out = np.zeros(12, dtype=np.uint64)
        assert self.lib.bpo_ipa_statement(curve, _p(G[0]), _p(G[1]), _p(H[0]), _p(H[1]), _p(u_xy), _p(a), _p(b), C.c_size_t(n),
                                          threads, _p(out)) == 0
        return out

    def ipa_prove(self, curve, G, H, u_xy, a, b, chal_seed, threads=1):
        n = len(a)
        k = max(n.bit_length() - 1, 1)
        L, R, ab = np.zeros((k, 12), dtype=np.uint64), np.zeros((k, 12), dtype=np.uint64), np.zeros(8, dtype=np.uint64)
        assert self.lib.bpo_ipa_prove(curve, _p(G[0]), _p(G[1]), _p(H[0]), _p(H[1]), _p(u_xy), _p(a), _p(b), C.c_size_t(n),
                                      C.c_uint64(chal_seed), threads, _p(L), _p(R), _p(ab)) == 0
        k = n.bit_length() - 1
        return dict(L=L[:k], R=R[:k], a=ab[:4], b=ab[4:])

    def ipa_verify(self, curve, G, H, u_xy, p_xyz, n, L, R, a, b, chal_seed, threads=1):
        ab = np.concatenate([np.asarray(a, dtype=np.uint64).reshape(4), np.asarray(b, dtype=np.uint64).reshape(4)])
        L, R = np.ascontiguousarray(L, dtype=np.uint64), np.ascontiguousarray(R, dtype=np.uint64)
        out = np.zeros(12, dtype=np.uint64)
        assert self.lib.bpo_ipa_verify(curve, _p(G[0]), _p(G[1]), _p(H[0]), _p(H[1]), _p(u_xy), _p(p_xyz), C.c_size_t(n),
                                       _p(L), _p(R), _p(ab), C.c_uint64(chal_seed), threads, _p(out)) == 0
        return out

    def range_agg_prove(self, curve, G, H, ugh_xy, values, gamma, n_bits, seed_rand, seed_chal, threads=1):
        values = np.ascontiguousarray(values, dtype=np.uint64)
        m = len(values)
        n = m * n_bits
        k = n.bit_length() - 1
        ugh = np.ascontiguousarray(ugh_xy, dtype=np.uint64).reshape(24)
        pts, sc = np.zeros((4, 12), dtype=np.uint64), np.zeros((5, 4), dtype=np.uint64)
        lr = np.zeros((max(k, 1), 24), dtype=np.uint64)
        assert self.lib.bpo_range_agg_prove(curve, _p(G[0]), _p(G[1]), _p(H[0]), _p(H[1]), _p(ugh), _p(values), _p(gamma),
                                            C.c_size_t(m), C.c_uint(n_bits), C.c_uint64(seed_rand), C.c_uint64(seed_chal),
                                            threads, _p(pts), _p(sc), _p(lr)) == 0
        return dict(A=pts[0], S=pts[1], TT1=pts[2], TT2=pts[3], tao_x=sc[0], mu=sc[1], t_hat=sc[2], a=sc[3], b=sc[4],
                    L=lr[:k, :12], R=lr[:k, 12:], points=pts, scalars=sc, lr=lr)

    def range_agg_verify(self, curve, G, H, ugh_xy, values, gamma, n_bits, points, scalars, lr, seed_rand, seed_chal, threads=1):
        """Value of the verifier's deferred MSM (identity iff accepted)."""
        values = np.ascontiguousarray(values, dtype=np.uint64)
        ugh = np.ascontiguousarray(ugh_xy, dtype=np.uint64).reshape(24)
        points, scalars, lr = (np.ascontiguousarray(v, dtype=np.uint64) for v in (points, scalars, lr))
        out = np.zeros(12, dtype=np.uint64)
        assert self.lib.bpo_range_agg_verify(curve, _p(G[0]), _p(G[1]), _p(H[0]), _p(H[1]), _p(ugh), _p(values), _p(gamma),
                                             C.c_size_t(len(values)), C.c_uint(n_bits), _p(points), _p(scalars), _p(lr),
                                             C.c_uint64(seed_rand), C.c_uint64(seed_chal), threads, _p(out)) == 0
        return out

    def range_value_commitments(self, curve, ugh_xy, values, gamma):
        values = np.ascontiguousarray(values, dtype=np.uint64)
        ugh = np.ascontiguousarray(ugh_xy, dtype=np.uint64).reshape(24)
        out = np.zeros((len(values), 12), dtype=np.uint64)
        assert self.lib.bpo_range_value_commitments(curve, _p(ugh), _p(values), _p(gamma), C.c_size_t(len(values)), _p(out)) == 0
        return out

    def poly_statement(self, curve, G, h_gen_xy, coeffs, r, x, threads=1):
        n = len(G[0])
        hg = np.ascontiguousarray(h_gen_xy, dtype=np.uint64).reshape(16)
        rx = np.concatenate([np.asarray(r, dtype=np.uint64).reshape(4), np.asarray(x, dtype=np.uint64).reshape(4)])
        com, ev = np.zeros(12, dtype=np.uint64), np.zeros(4, dtype=np.uint64)
        assert self.lib.bpo_poly_statement(curve, _p(G[0]), _p(G[1]), _p(hg), _p(coeffs), C.c_size_t(len(coeffs)), C.c_size_t(n),
                                           _p(rx), threads, _p(com), _p(ev)) == 0
        return com, ev

    def poly_prove(self, curve, G, h_gen_xy, coeffs, r, x, seed_rand, seed_chal, threads=1):
        n = len(G[0])
        k = n.bit_length() - 1
        hg = np.ascontiguousarray(h_gen_xy, dtype=np.uint64).reshape(16)
        rx = np.concatenate([np.asarray(r, dtype=np.uint64).reshape(4), np.asarray(x, dtype=np.uint64).reshape(4)])
        L, R = np.zeros((max(k, 1), 12), dtype=np.uint64), np.zeros((max(k, 1), 12), dtype=np.uint64)
        ar, g, ui = np.zeros(8, dtype=np.uint64), np.zeros(9, dtype=np.uint64), np.zeros((max(k, 1), 4), dtype=np.uint64)
        rc = self.lib.bpo_poly_prove(curve, _p(G[0]), _p(G[1]), _p(hg), _p(coeffs), C.c_size_t(len(coeffs)), C.c_size_t(n), _p(rx),
                                     C.c_uint64(seed_rand), C.c_uint64(seed_chal), threads, _p(L), _p(R), _p(ar), _p(g), _p(ui))
        assert rc == 0, rc
        return dict(L=L[:k], R=R[:k], a=ar[:4], r=ar[4:], g=g, ui=ui[:k])

    def poly_verify(self, curve, G, h_gen_xy, commit_xyz, x, evaluation, L, R, a, r, seed_chal, g_assumed=None, threads=1):
        """res of poly_commit::verify (or lazy_verify when g_assumed is given) and the challenges."""
        n = len(G[0])
        k = n.bit_length() - 1
        hg = np.ascontiguousarray(h_gen_xy, dtype=np.uint64).reshape(16)
        xe = np.concatenate([np.asarray(x, dtype=np.uint64).reshape(4), np.asarray(evaluation, dtype=np.uint64).reshape(4)])
        ar = np.concatenate([np.asarray(a, dtype=np.uint64).reshape(4), np.asarray(r, dtype=np.uint64).reshape(4)])
        L, R = np.ascontiguousarray(L, dtype=np.uint64), np.ascontiguousarray(R, dtype=np.uint64)
        ga = None if g_assumed is None else np.ascontiguousarray(g_assumed, dtype=np.uint64)
        res, ui = np.zeros(12, dtype=np.uint64), np.zeros((max(k, 1), 4), dtype=np.uint64)
        assert self.lib.bpo_poly_verify(curve, _p(G[0]), _p(G[1]), _p(hg), C.c_size_t(n), _p(commit_xyz), _p(xe), _p(L), _p(R),
                                        _p(ar), _p(ga), C.c_uint64(seed_chal), threads, _p(res), _p(ui)) == 0
        return res, ui[:k]

    def poly_fold_todos(self, curve, uis, gs_xyi, alpha, x, threads=1):
        uis = np.ascontiguousarray(uis, dtype=np.uint64)
        count, k = uis.shape[0], uis.shape[1]
        gs_xyi = np.ascontiguousarray(gs_xyi, dtype=np.uint64).reshape(count, 9)
        ax = np.concatenate([np.asarray(alpha, dtype=np.uint64).reshape(4), np.asarray(x, dtype=np.uint64).reshape(4)])
        coeffs, com, ev = np.zeros((1 << k, 4), dtype=np.uint64), np.zeros(12, dtype=np.uint64), np.zeros(4, dtype=np.uint64)
        assert self.lib.bpo_poly_fold_todos(curve, _p(uis), _p(gs_xyi), C.c_size_t(count), C.c_size_t(k), _p(ax), threads,
                                            _p(coeffs), _p(com), _p(ev)) == 0
        return coeffs, com, ev
