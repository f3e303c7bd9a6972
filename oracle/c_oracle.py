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
                 "bpo_scale_points", "bpo_msm", "bpo_msm_naive", "bpo_gen_points", "bpo_max_threads"):
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
