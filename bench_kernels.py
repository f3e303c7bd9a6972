#!/usr/bin/env python3
"""Per-kernel measurements: streaming kernels in GB/s of
algorithmic bytes, curve kernels in outputs/s and MAC/s, MSM over sizes, one IPA
prove of n = 2^16.  Prints one JSON object per line.  Inputs are resident in HBM;
timings are HIP-event durations reported by the library (bp_last_kernel_ms) or
wall time around the synchronous call where noted."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bulletproofs_amd as bp  # noqa: E402
from bulletproofs_amd._lib import lib  # noqa: E402

ctx = bp.Context(0)
C = bp.SECP256K1
rng = np.random.default_rng(7)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()


def rand_field(n):
    a = rng.integers(0, 1 << 64, (n, 4), dtype=np.uint64)
    a[:, 3] &= np.uint64((1 << 63) - 1)
    return a


def gen_points(n):
    p = (1 << 256) - (1 << 32) - 977
    order = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
    gx = 0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798
    gy = 0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8
    r = 1 << 256
    lim = lambda v: [(v >> (64 * i)) & ((1 << 64) - 1) for i in range(4)]  # noqa: E731
    X = np.tile(np.array(lim(gx * r % p), dtype=np.uint64), (n, 1))
    Y = np.tile(np.array(lim(gy * r % p), dtype=np.uint64), (n, 1))
    rm = r % order
    vals = [((i + 1) * rm) % order for i in range(n)]
    S = np.zeros((n, 4), dtype=np.uint64)
    for k in range(4):
        S[:, k] = np.array([(v >> (64 * k)) & ((1 << 64) - 1) for v in vals], dtype=np.uint64)
    ox, oy, _ = ctx.scale_points(C, (X, Y, None), S)
    return ox, oy


def best(fn, reps=5):
    ts = []
    for _ in range(reps):
        fn()
        ts.append(ctx.last_kernel_ms(2))
    return min(ts), float(np.median(ts))


ctx.int_mad_peak(12000)  # warm-up launch: the clock the GPU holds under load
peak = max(ctx.int_mad_peak(12000) for _ in range(3))  # launches of >= 50 ms, as bench.py's probe
print(json.dumps({"kernel": "v_mad_u64_u32 peak", "mac_per_s": peak}), flush=True)

# ---- what this GPU's memory delivers to the simplest kernels (torch's own copy, fill and sum over
# 512 MiB buffers, HIP-event timed): the rates the streaming kernels below are priced against ----
def torch_rate(fn, nbytes, reps=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best_ms = 1e30
    for _ in range(reps):
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        best_ms = min(best_ms, e0.elapsed_time(e1))
    return nbytes / best_ms / 1e6  # GB/s


_src = torch.randint(0, 1 << 62, (1 << 26,), dtype=torch.int64, device="cuda")  # 512 MiB
_dst = torch.empty_like(_src)
calib = {"copy_GBps": torch_rate(lambda: _dst.copy_(_src), 2 * _src.numel() * 8),
         "write_GBps": torch_rate(lambda: _dst.fill_(7), _src.numel() * 8),
         "read_GBps": torch_rate(lambda: _src.sum(), _src.numel() * 8)}
del _src, _dst
print(json.dumps({"kernel": "memory calibration (torch copy_, fill_, sum over 512 MiB)", **calib}), flush=True)

# ---- the scalar-field multiply in registers: the other ceiling of the streaming kernels ----
import ctypes as _C  # noqa: E402


def _mb(i, blocks, iters):
    ms = _C.c_float()
    ctx._chk(lib.bp_microbench(ctx.h, i, blocks, iters, _C.byref(ms)))
    return ms.value


_blocks, _it = 8192, 200
fr_mul_rate = _blocks * 256 * _it * 2 / min(_mb(9, _blocks, _it) for _ in range(3)) * 1e3
fr_mul2_rate = _blocks * 256 * _it * 2 / min(_mb(10, _blocks, _it) for _ in range(3)) * 1e3
print(json.dumps({"kernel": "secp256k1 Fr multiply in registers (bp_microbench 9, 10)", "mul_per_s": fr_mul_rate,
                  "dual_products_per_s": fr_mul2_rate,
                  "s_vector_ceiling_GBps": fr_mul_rate * 32 / 1e9, "fold_scalars_ceiling_GBps": fr_mul2_rate * 96 / 1e9,
                  "inner_product_ceiling_GBps": fr_mul2_rate * 128 / 1e9}), flush=True)

# ---- streaming kernels ----
for logn in (20, 24):
    n = 1 << logn
    a, b, out = dev(rand_field(n)), dev(rand_field(n)), dev(np.zeros((n, 4), dtype=np.uint64))
    x, y = rand_field(1), rand_field(1)
    fn = lambda: ctx._chk(lib.bp_fold_scalars_dev(ctx.h, C, a.data_ptr(), b.data_ptr(), n, x.ctypes.data, y.ctypes.data, out.data_ptr()))  # noqa: E731
    mn, med = best(fn)
    print(json.dumps({"kernel": "fold_scalars", "n": n, "ms": mn, "GBps": 96 * n / mn / 1e6, "frac_hbm_8TBs": 96 * n / mn / 1e6 / 8000,
                      "frac_of_copy_rate": 96 * n / mn / 1e6 / calib["copy_GBps"],
                      "frac_of_multiply_ceiling": n / mn * 1e3 / fr_mul2_rate, "TMACps": 272 * n / mn / 1e9, "frac_int_peak": 272 * n / mn * 1e3 / peak}), flush=True)
    res = np.zeros(4, dtype=np.uint64)
    fn = lambda: ctx._chk(lib.bp_inner_product_dev(ctx.h, C, a.data_ptr(), b.data_ptr(), n, res.ctypes.data))  # noqa: E731
    mn, med = best(fn)
    print(json.dumps({"kernel": "inner_product", "n": n, "ms": mn, "GBps": 64 * n / mn / 1e6, "frac_hbm_8TBs": 64 * n / mn / 1e6 / 8000,
                      "frac_of_read_rate": 64 * n / mn / 1e6 / calib["read_GBps"], "TMACps": 136 * n / mn / 1e9, "frac_int_peak": 136 * n / mn * 1e3 / peak}), flush=True)
    k = logn
    ch = rand_field(k)
    chi = ctx.batch_inversion(C, ch)
    fn = lambda: ctx._chk(lib.bp_s_vector_dev(ctx.h, C, ch.ctypes.data, chi.ctypes.data, k, out.data_ptr()))  # noqa: E731
    mn, med = best(fn)
    print(json.dumps({"kernel": "s_vector", "n": n, "ms": mn, "GBps": 32 * n / mn / 1e6, "frac_hbm_8TBs": 32 * n / mn / 1e6 / 8000,
                      "frac_of_write_rate": 32 * n / mn / 1e6 / calib["write_GBps"],
                      "frac_of_multiply_ceiling": n / mn * 1e3 / fr_mul_rate, "TMACps": 136 * n / mn / 1e9}), flush=True)
    del a, b, out

# ---- curve kernels ----
npts = 1 << 17
px, py = gen_points(npts)
dx, dy = dev(px), dev(py)
h = npts // 2
ox, oy, oi = dev(np.zeros((h, 4), dtype=np.uint64)), dev(np.zeros((h, 4), dtype=np.uint64)), torch.zeros(h, dtype=torch.uint8, device="cuda")
x, y = rand_field(1), rand_field(1)
fn = lambda: ctx._chk(lib.bp_fold_generators_dev(ctx.h, C, dx.data_ptr(), dy.data_ptr(), None, dx.data_ptr() + 32 * h, dy.data_ptr() + 32 * h, None, h,  # noqa: E731
                                                 x.ctypes.data, y.ctypes.data, ox.data_ptr(), oy.data_ptr(), oi.data_ptr()))
mn, med = best(fn, 3)
alg_mul = 256 * 7 + 192 * 11  # doublings + expected mixed additions of the joint double-and-add
print(json.dumps({"kernel": "fold_generators", "h": h, "ms": mn, "outputs_per_s": h / mn * 1e3,
                  "TMACps": alg_mul * 136 * h / mn / 1e9, "frac_int_peak": alg_mul * 136 * h / mn * 1e3 / peak}), flush=True)

# ---- MSM over sizes ----
big = 1 << 22
bx, by = gen_points(1 << 16)
reps = big // (1 << 16)
BX, BY = dev(np.tile(bx, (reps, 1))), dev(np.tile(by, (reps, 1)))
K = dev(rand_field(big))
for logn in (16, 17, 18, 19, 20, 21, 22):
    n = 1 << logn
    ts = []
    for _ in range(4):
        t0 = time.perf_counter()
        ctx.msm_dev(C, BX.data_ptr(), BY.data_ptr(), None, K.data_ptr(), n)
        ts.append(time.perf_counter() - t0)
    print(json.dumps({"kernel": "msm", "log2n": logn, "wall_ms": min(ts) * 1e3, "points_per_s": n / min(ts),
                      "device_ms": ctx.last_kernel_ms(1), "accumulate_ms": ctx.last_kernel_ms(0)}), flush=True)
del BX, BY, K

# ---- IPA prove, n = 2^16 (benches/ipa_bench.rs:102-105), seeded challenges ----
n = 1 << 16
gx, gy = px[:n], py[:n]
hx, hy = px[n:2 * n], py[n:2 * n]
a, b = rand_field(n), rand_field(n)
ch = rand_field(16)
chi = ctx.batch_inversion(C, ch)
t0 = time.perf_counter()
s = ctx.ipa_begin(C, bp.IPA_BULLETPROOF, (gx, gy, None), (hx, hy, None), a, b)
t_begin = time.perf_counter() - t0
t_lr = t_fold = 0.0
for j in range(16):
    t0 = time.perf_counter()
    s.round_lr()
    t_lr += time.perf_counter() - t0
    t0 = time.perf_counter()
    s.round_fold(ch[j:j + 1], chi[j:j + 1])
    t_fold += time.perf_counter() - t0
print(json.dumps({"kernel": "ipa_prove_rounds", "n": n, "upload_ms": t_begin * 1e3, "lr_ms": t_lr * 1e3, "fold_ms": t_fold * 1e3,
                  "total_ms": (t_lr + t_fold) * 1e3, "proofs_per_s": 1.0 / (t_lr + t_fold)}), flush=True)

# ---- IPA verify, n = 2^16 (src/ipa/mod.rs:102-212): the s-vector and its inverse, their multiples
# by a and b, and the deferred MSM over gs || hs || u (registered once) plus P, L_j, R_j ----
# (u and the proof's points are stand-ins taken from the same table: values do not change the work)
bases = ctx.register_bases(C, (np.concatenate([gx, hx, px[:1]]), np.concatenate([gy, hy, py[:1]]), None))
extra = (px[1:34], py[1:34], None)
extra_s = rand_field(33)
pa, pb = rand_field(1), rand_field(1)
ts = []
for _ in range(3):
    t0 = time.perf_counter()
    ss = ctx.s_vector(C, ch, chi)
    ssi = ctx.s_vector(C, chi, ch)
    sc = np.concatenate([ctx.scale(C, ss, pa), ctx.scale(C, ssi, pb), ctx.hadamard(C, pa, pb)])
    bases.msm_plus(sc, extra, extra_s)
    ts.append(time.perf_counter() - t0)
print(json.dumps({"kernel": "ipa_verify", "n": n, "msm_points": 2 * n + 34, "total_ms": min(ts) * 1e3,
                  "verifies_per_s": 1 / min(ts)}), flush=True)
# the same check with the scalars built where the MSM reads them (bp_ipa_verify_msm)
ts = []
for _ in range(3):
    t0 = time.perf_counter()
    bases.ipa_verify_msm(0, n, 2 * n, ch, chi, pa, pb, ctx.hadamard(C, pa, pb), extra, extra_s)
    ts.append(time.perf_counter() - t0)
print(json.dumps({"kernel": "ipa_verify_fused", "n": n, "msm_points": 2 * n + 34, "total_ms": min(ts) * 1e3,
                  "verifies_per_s": 1 / min(ts)}), flush=True)
bases.free()

# ---- points from bytes: a CRS of 2^20 compressed points decoded into a base set (host bytes in) ----
n = 1 << 20
wx, wy = gen_points(1 << 16)
wire = np.tile(ctx.points_compress(C, (wx, wy, None)), (n >> 16, 1))
ts = []
for _ in range(3):
    t0 = time.perf_counter()
    wb = ctx.register_bases_compressed(C, wire)
    ts.append(time.perf_counter() - t0)
    wb.free()
print(json.dumps({"kernel": "bases_register_compressed", "n": n, "total_ms": min(ts) * 1e3, "points_per_s": n / min(ts),
                  "note": "33 bytes per point uploaded, one square root (a^((p+1)/4)) per point on the device, base set allocated"}),
      flush=True)
