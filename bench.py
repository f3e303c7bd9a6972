#!/usr/bin/env python3
"""Headline benchmark: MSM throughput (points/s) at 2^20 secp256k1 points per GPU.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU)

A step is one multi-scalar multiplication over synthetic random points/scalars
that are already resident in HBM when the timed region starts; the result is
brought back to the host (it is what G::msm_unchecked returns).  At N > 1 every
rank owns a 2^20-point shard of one N*2^20-point MSM (weak scaling): it computes
the per-window partial sums of its shard, the partials are all-gathered over
RCCL, and every rank combines them on its GPU.

Rank 0 prints ONE JSON line.  `roofline` prices the dominant kernel (bucket
accumulation) against the measured v_mad_u64_u32 issue rate; `cpu_baseline` is
the C oracle's Pippenger timed on the host cores over a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

LOG2_N = 20
SEED = 0xB0117E7 + 2  # config index 2 of BASELINE.json ("Standalone MSM 2^16..2^20")
MADD_MULS = 10        # XYZZ mixed addition: 8M + 2S
MAC_PER_MUL = 136     # 8-limb Montgomery multiply: 2*8*8 + 8 32x32 multiply-accumulates (algorithmic credit)
# v_mad_u64_u32 the generated device multiply really issues per field product (tools/gen_fe_mul.py):
# secp256k1's pipeline works on plain residues mod 2^256 - 2^32 - 977 (64 partial products, 8 folds by 977
# and 9 additions of a 32-bit limb done as multiply-accumulates with 1)
MAD_ISSUED_PER_MUL = 81
# whole-MSM arithmetic the bucket method needs per point at n = 2^20, c = 16: 16 windows x one mixed addition
# priced at 11 M, plus the bucket reduction 2 * 2^15 * 16 full additions of 16 M amortised over n, 136 MAC per M
WHOLE_STEP_MULS = lambda n, nw, cb: nw * 11 + 2 * (1 << (cb - 1)) * nw * 16 / n  # noqa: E731
PEAK_PROBES = 6
# one v_mad_u64_u32 per lane and cycle: 256 CUs x 4 SIMDs x 16 lanes at the 2.4 GHz peak engine clock.  The probe
# kernel reaches 32-36 TMAC/s of it, by the box and by how warm it is; the roofline is priced against this figure.
NOMINAL_MAD_PEAK = 256 * 4 * 16 * 2.4e9
PEAK_PROBE_ITERS = 12000  # one probe launch runs >= 50 ms
# Memory-side bytes of one k_chunk_sums launch at the default size, from separate
# `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs of this script (averages over the
# launches; profiles/r02_v5_pmc_summary.csv).  The counters read KiB; a
# scattered read of 64-byte records with a known byte count (tools/calib_gather.py,
# profiles/r01_gather64_calibration.csv) reads back exactly, so no factor is applied.
PMC_TRAFFIC_2_20 = {"fetch_bytes": 1222888.0 * 1024, "write_bytes": 125880.1 * 1024}
CPU_MSM_SAMPLES = 5


def make_inputs(n, rank):
    """P_i = (r0 + i*q) G by repeated addition on the device-independent generator
    in the product library's own scale kernel is not needed here: values do not
    change the work, so a small table of valid points is tiled; scalars are
    uniform below 2^255 (below the group order)."""
    import bulletproofs_amd as bp  # noqa: F401  (fails loudly if the library is missing)

    rng = np.random.default_rng(SEED + 1000 * rank)
    ks = rng.integers(0, 1 << 64, (n, 4), dtype=np.uint64)
    ks[:, 3] &= np.uint64((1 << 63) - 1)
    return ks


def gen_points_on_gpu(ctx, n, rank, stride=1, first=None):
    """n distinct valid points: (first + stride * i) * G for i < n (first = rank * n + 1 unless given) computed by
    the library's own per-point scaling kernel from the generator (setup, untimed)."""
    # secp256k1 generator in Montgomery form
    p = (1 << 256) - (1 << 32) - 977
    gx = 0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798
    gy = 0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8
    r = 1 << 256

    def limbs(v):
        return [(v >> (64 * i)) & ((1 << 64) - 1) for i in range(4)]

    order = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
    X = np.tile(np.array(limbs(gx * r % p), dtype=np.uint64), (n, 1))
    Y = np.tile(np.array(limbs(gy * r % p), dtype=np.uint64), (n, 1))
    # scalars rank*n + i + 1 in Montgomery form: (v * R) mod order, vectorised via Python ints per chunk
    base = rank * n + 1 if first is None else first
    rm = r % order
    S = np.zeros((n, 4), dtype=np.uint64)
    vals = [((base + stride * i) * rm) % order for i in range(n)]
    for k in range(4):
        S[:, k] = np.array([(v >> (64 * k)) & ((1 << 64) - 1) for v in vals], dtype=np.uint64)
    ox, oy, oi = ctx.scale_points(0, (X, Y, None), S)
    assert not oi.any()
    return ox, oy


def _oracle_and_cores():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import c_oracle

    co = c_oracle.COracle()
    # the GPU box exposes the whole machine's CPUs; its share is OMP_NUM_THREADS (16 there)
    cores = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
    return co, max(1, min(cores, co.max_threads()))


def cpu_ipa_baseline(G, H, u_xy, wa, wb, samples=3):
    """C oracle's ipa::prove and ipa::verify (oracle/bp_oracle_proto.c) at the size of config 3 on the
    host cores: the CPU figure beside other_configs.ipa_2_16.  A C restatement, not the arkworks binary."""
    co, cores = _oracle_and_cores()
    n = wa.shape[0]
    P = co.ipa_statement(0, G, H, u_xy, wa, wb, threads=cores)
    tp, tv = [], []
    for _ in range(samples):
        t0 = time.perf_counter()
        pr = co.ipa_prove(0, G, H, u_xy, wa, wb, SEED, threads=cores)
        t1 = time.perf_counter()
        res = co.ipa_verify(0, G, H, u_xy, P, n, pr["L"], pr["R"], pr["a"], pr["b"], SEED, threads=cores)
        t2 = time.perf_counter()
        if res[8:].any():
            raise SystemExit("C oracle rejected its own inner-product argument")
        tp.append(t1 - t0)
        tv.append(t2 - t1)
    mp, mv = float(np.median(tp)), float(np.median(tv))
    return dict(prove_ms=mp * 1e3, verify_ms=mv * 1e3, proofs_per_s=1.0 / mp, verifies_per_s=1.0 / mv, cores=cores, kind="port",
                sample=f"median of {samples} prove + verify runs at n = 2^{n.bit_length() - 1} (prove min {min(tp):.3f} s, max "
                       f"{max(tp):.3f} s), C restatement of src/ipa/mod.rs:19-212 (oracle/bp_oracle_proto.c, OpenMP inside the MSMs "
                       f"and folds); not the arkworks binary")


def cpu_poly_baseline(G, hg, coeffs, r, x, samples=1):
    """C oracle's poly_commit::prove and verify at the size of config 5 on the host cores, once each (a
    prove is several seconds of CPU work): the CPU figure beside other_configs.poly_commit_2_20."""
    co, cores = _oracle_and_cores()
    n = coeffs.shape[0]
    com, ev = co.poly_statement(0, G, hg, coeffs, r, x, threads=cores)
    tp, tv = [], []
    for _ in range(samples):
        t0 = time.perf_counter()
        pr = co.poly_prove(0, G, hg, coeffs, r, x, SEED, SEED + 1, threads=cores)
        t1 = time.perf_counter()
        res, _ = co.poly_verify(0, G, hg, com, x, ev, pr["L"], pr["R"], pr["a"], pr["r"], SEED + 1, threads=cores)
        t2 = time.perf_counter()
        if res[8:].any():
            raise SystemExit("C oracle rejected its own polynomial-commitment opening")
        tp.append(t1 - t0)
        tv.append(t2 - t1)
    mp, mv = float(np.median(tp)), float(np.median(tv))
    return dict(prove_ms=mp * 1e3, verify_ms=mv * 1e3, cores=cores, kind="port",
                sample=f"{samples} prove + verify run(s) at n = 2^{n.bit_length() - 1}, C restatement of "
                       f"src/poly_commit/mod.rs:45-173 (oracle/bp_oracle_proto.c, OpenMP inside the MSMs and folds; the prover "
                       f"includes the full-size MSM of :108-113 as the reference's does); not the arkworks binary")


def other_configs(ctx, px, py, cpu=True):
    """BASELINE configs 3 and 5 beside the headline figure, on the points of the MSM leg: the rounds of
    an inner-product argument over 2 x 2^16 generators (benches/ipa_bench.rs:102-105, seeded challenges)
    and a polynomial-commitment opening of degree 2^20 - 1 with its verification and the fold of two
    deferred checks (benches/poly_commit_bench.rs), each through the C++ drivers / the session API, and
    beside each the C oracle's prover and verifier timed on the host cores."""
    import bulletproofs_amd as bp
    from bulletproofs_amd import host_mirror

    C = bp.SECP256K1
    rng = np.random.default_rng(SEED + 77)

    def rand_field(cnt):
        a = rng.integers(0, 1 << 64, (cnt, 4), dtype=np.uint64)
        a[:, 3] &= np.uint64((1 << 63) - 1)
        return a

    out = {}
    n = 1 << 16
    if px.shape[0] >= 2 * n + 1:
        ch = rand_field(16)
        chi = ctx.batch_inversion(C, ch)
        # CRS gs | hs | u registered once; the statement P = <a, gs> + <b, hs> + <a, b> u
        bases = ctx.register_bases(C, (px[:2 * n + 1], py[:2 * n + 1], None))
        bases.precompute()  # window tables of the CRS (2 KiB per point, once): the session's first replacement of its generators reads them
        u_xy = np.concatenate([px[2 * n], py[2 * n]])
        wa, wb = rand_field(n), rand_field(n)
        ab = ctx.inner_product(C, wa, wb).reshape(1, 4)
        P = bases.msm(np.concatenate([wa, wb, ab]))
        best = None
        for _ in range(2):
            s = ctx.ipa_begin_resident(bases, bp.IPA_BULLETPROOF, 0, n, wa, wb)
            s.set_u(u_xy)
            Ls, Rs = [], []
            t0 = time.perf_counter()
            for j in range(16):
                L, R, _, _ = s.round_lr()
                Ls.append(L)
                Rs.append(R)
                s.round_fold(ch[j:j + 1], chi[j:j + 1])
            dt = time.perf_counter() - t0
            _, _, fa, fb = s.read()
            s.free()
            best = dt if best is None else min(best, dt)
        # ipa::verify_aux (src/ipa/mod.rs:102-193) as one deferred MSM: -P - sum x_j^2 L_j - sum x_j^-2 R_j + ...
        one = ctx.hadamard(C, ch[:1], chi[:1])  # x * x^-1 = 1 in the library's own form
        minus = ctx.vector_sub(C, np.zeros((33, 4), dtype=np.uint64),
                               np.concatenate([one, ctx.hadamard(C, ch, ch), ctx.hadamard(C, chi, chi)]))
        pts = np.stack([P] + Ls + Rs)
        extra = (np.ascontiguousarray(pts[:, 0:4]), np.ascontiguousarray(pts[:, 4:8]),
                 (~pts[:, 8:12].any(axis=1)).astype(np.uint8))
        vbest, valid = None, True
        for _ in range(3):
            t0 = time.perf_counter()
            value = bases.ipa_verify_msm(0, n, 2 * n, ch, chi, fa, fb, ctx.hadamard(C, fa, fb), extra, minus)
            dt = time.perf_counter() - t0
            valid = valid and not value[8:12].any()
            vbest = dt if vbest is None else min(vbest, dt)
        bases.free()
        if not valid:
            raise SystemExit("inner-product argument at n = 2^16 failed to verify")
        out["ipa_2_16"] = {"workload": "inner-product argument, n = 2^16 (2 x 2^16 generators), 16 rounds, seeded challenges; "
                                       "verification as one deferred MSM with its scalars built on the device",
                           "prove_rounds_ms": best * 1e3, "proofs_per_s": 1.0 / best, "verify_ms": vbest * 1e3,
                           "verifies_per_s": 1.0 / vbest, "verified": bool(valid)}
        if cpu:
            out["ipa_2_16"]["cpu_baseline"] = cpu_ipa_baseline((px[:n], py[:n], None), (px[n:2 * n], py[n:2 * n], None), u_xy, wa, wb)
    n = px.shape[0]
    if n >= 1 << 20:
        n = 1 << 20
        # CRS: the MSM leg's points; h and the generator of u: two more multiples of G
        hx, hy, _ = ctx.scale_points(C, (px[:2], py[:2], None), rand_field(2))
        hg = np.stack([np.concatenate([hx[0], hy[0]]), np.concatenate([hx[1], hy[1]])])
        gs = (px[:n], py[:n], None)
        coeffs, r, x = rand_field(n), rand_field(1), rand_field(1)
        host_mirror.poly_commit_roundtrip(ctx, C, gs, hg, coeffs, r, x, SEED, SEED + 1, halo=False)  # warm-up
        res = host_mirror.poly_commit_roundtrip(ctx, C, gs, hg, coeffs, r, x, SEED, SEED + 1, halo=True)
        if not (res["verify_ok"] and res["lazy_ok"] and res["halo_ok"]):
            raise SystemExit("polynomial-commitment opening at degree 2^20 - 1 failed to verify")
        out["poly_commit_2_20"] = {"workload": "polynomial commitment opening, degree 2^20 - 1, CRS resident",
                                   "prove_ms": res["prove_ms"], "verify_ms": res["verify_ms"],
                                   "all_verified": bool(res["verify_ok"] and res["lazy_ok"] and res["halo_ok"])}
        # 16 openings checked together: two distinct openings cycled through the batch, the MSM over the 2^20
        # generators once for all of them (PolyCommit::verify_batch), against the same 16 verified one by one
        vb = host_mirror.poly_commit_verify_batch(ctx, C, gs, hg, np.stack([coeffs, np.ascontiguousarray(coeffs[::-1])]),
                                                  np.stack([r[0], r[0]]), np.stack([x[0], x[0]]), batch=16, seed_rand=SEED,
                                                  seed_chal=SEED + 1, reps=3)
        if not (vb["ok_batch"] and vb["ok_each"]):
            raise SystemExit("batch of polynomial-commitment openings at degree 2^20 - 1 failed to verify")
        out["poly_commit_2_20"]["verify_batch_16"] = {"workload": "16 openings of degree 2^20 - 1 verified together, CRS resident",
                                                      "ms": vb["batch_ms"], "openings_per_s": 16.0 / vb["batch_ms"] * 1e3,
                                                      "one_by_one_ms": vb["one_by_one_ms"]}
        if cpu:
            out["poly_commit_2_20"]["cpu_baseline"] = cpu_poly_baseline(gs, hg, coeffs, r[0], x[0])
    return out


# Memory-side bytes of the streaming kernels at 2^24 elements from separate `rocprofv3 --pmc FETCH_SIZE` /
# `--pmc WRITE_SIZE` runs of tools/stream_kernels.py (tools/prof_stream.sh; FETCH_SIZE doubled as coalesced
# 16-byte-per-lane reads on gfx950 require): read + written bytes per launch, profiles/r02_v5_stream_pmc_summary.csv
PMC_STREAM_2_24 = {"fold_scalars": 1124257587 + 607618253, "inner_product": 1278147584 + 32768, "s_vector": 8525824 + 571458150}
HBM_PEAK_GBPS = 8000.0


def streaming_kernels(ctx, torch):
    """fold_scalars (src/ipa/mod.rs:229-234), inner_product (src/vector_ops.rs:159-171) and the s-vector
    (src/ipa/mod.rs:134-148) at 2^20 and 2^24 elements, inputs resident in HBM: algorithmic bytes per
    element (96 / 64 / 32) over the HIP-event duration of the launch(es), best of five, against the 8 TB/s
    of HBM3E.  At 2^20 elements (32-96 MB, the size BASELINE's configurations use) a launch is some 25 us
    and what is measured is launch latency, not memory."""
    import bulletproofs_amd as bp
    from bulletproofs_amd._lib import lib

    C = bp.SECP256K1
    rng = np.random.default_rng(SEED + 5)

    def rand_field(cnt):
        a = rng.integers(0, 1 << 64, (cnt, 4), dtype=np.uint64)
        a[:, 3] &= np.uint64((1 << 63) - 1)
        return a

    def best(fn, reps=5):
        fn()
        ts = []
        for _ in range(reps):
            fn()
            ts.append(ctx.last_kernel_ms(2))
        return min(ts)

    out = {"peak_GBps": HBM_PEAK_GBPS, "bound": "hbm (nominal 8 TB/s); the field multiply per element keeps fold_scalars and "
                                                "inner_product near 0.6 of it and the s-vector, one product per 32 bytes, near 0.35",
           "traffic_source": "profiles/r02_v5_stream_pmc_summary.csv (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs)"}
    to_dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()  # noqa: E731
    for logn in (20, 24):
        n = 1 << logn
        a, b = to_dev(rand_field(n)), to_dev(rand_field(n))
        o = torch.empty((n, 4), dtype=torch.int64, device="cuda")
        x, y = rand_field(1), rand_field(1)
        res = np.zeros(4, dtype=np.uint64)
        ch = rand_field(logn)
        chi = ctx.batch_inversion(C, ch)
        runs = {
            "fold_scalars": (96, lambda: ctx._chk(lib.bp_fold_scalars_dev(ctx.h, C, a.data_ptr(), b.data_ptr(), n, x.ctypes.data,
                                                                          y.ctypes.data, o.data_ptr()))),
            "inner_product": (64, lambda: ctx._chk(lib.bp_inner_product_dev(ctx.h, C, a.data_ptr(), b.data_ptr(), n, res.ctypes.data))),
            "s_vector": (32, lambda: ctx._chk(lib.bp_s_vector_dev(ctx.h, C, ch.ctypes.data, chi.ctypes.data, logn, o.data_ptr()))),
        }
        for name, (bytes_per, fn) in runs.items():
            ms = best(fn)
            gbps = bytes_per * n / ms / 1e6
            out[f"{name}_2_{logn}"] = {"ms": ms, "algorithmic_bytes": bytes_per * n, "achieved_GBps": gbps, "frac": gbps / HBM_PEAK_GBPS,
                                       "traffic": PMC_STREAM_2_24[name] if logn == 24 else None}
        del a, b, o
    torch.cuda.empty_cache()
    return out


def cpu_baseline(n_sample):
    """C oracle (oracle/bp_oracle.c, a port of the bucket method) on all host cores."""
    co, cores = _oracle_and_cores()
    rng = np.random.default_rng(SEED)
    ks = rng.integers(0, 1 << 64, (n_sample, 4), dtype=np.uint64)
    ks[:, 3] &= np.uint64((1 << 63) - 1)
    # valid points: a tiled table of 4096 multiples of G from the oracle's own generator
    p = (1 << 256) - (1 << 32) - 977
    gx = 0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798
    gy = 0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8
    r = 1 << 256
    lim = lambda v: np.array([(v >> (64 * i)) & ((1 << 64) - 1) for i in range(4)], dtype=np.uint64)  # noqa: E731
    gxy = np.concatenate([lim(gx * r % p), lim(gy * r % p)])
    px, py = co.gen_points(0, gxy, gxy, n_sample)
    co.msm(0, (px, py, None), ks, cores)  # untimed: pages in, spins the OpenMP team up
    times = []
    for _ in range(CPU_MSM_SAMPLES):
        t0 = time.perf_counter()
        co.msm(0, (px, py, None), ks, cores)
        times.append(time.perf_counter() - t0)
    dt = float(np.median(times))
    return dict(value=n_sample / dt, unit="points/s", cores=cores, kind="port",
                sample=f"median of {CPU_MSM_SAMPLES} runs of one 2^{n_sample.bit_length() - 1}-point secp256k1 MSM "
                       f"(min {min(times):.3f} s, median {dt:.3f} s, max {max(times):.3f} s), C restatement of the bucket "
                       f"method (oracle/bp_oracle.c, OpenMP over windows); not the arkworks binary")


def cpu_range_baseline(gs, hs, ugh, values, gamma, n_bits, seed, samples):
    """C oracle's range prover and verifier (oracle/bp_oracle_proto.c) on the host cores: the CPU
    figure beside the range-proof half of the metric.  A C restatement, not the arkworks binary."""
    co, cores = _oracle_and_cores()
    tp, tv = [], []
    for _ in range(samples):
        t0 = time.perf_counter()
        pr = co.range_agg_prove(0, gs, hs, ugh, values, gamma, n_bits, seed, seed + 1, threads=cores)
        t1 = time.perf_counter()
        res = co.range_agg_verify(0, gs, hs, ugh, values, gamma, n_bits, pr["points"], pr["scalars"], pr["lr"],
                                  seed ^ 0x5EED, seed + 1, threads=cores)
        t2 = time.perf_counter()
        if int(res[8:].any()) != 0:
            raise SystemExit("C oracle rejected its own range proof")
        tp.append(t1 - t0)
        tv.append(t2 - t1)
    mp, mv = float(np.median(tp)), float(np.median(tv))
    return dict(prove_per_s=1.0 / mp, verify_per_s=1.0 / mv, prove_ms=mp * 1e3, verify_ms=mv * 1e3, cores=cores, kind="port",
                sample=f"median of {samples} prove + verify runs, C restatement of src/range/aggregate.rs "
                       f"(oracle/bp_oracle_proto.c, OpenMP inside the MSMs and folds); not the arkworks binary")


def range_proofs(ctx, device, workers=8, per_worker=8, cpu=True, extras=True):
    """second half of the metric: aggregated range proof, 64 values x 64 bits
    (benches/aggregate_range_bench.rs:93-97), proved and verified through the GPU path by the
    host mirror with the seeded transcript stand-in; measured outside the timed MSM steps.  Beside
    it the single-value 64-bit proof of benches/range_bench.rs:37-62 (the reference's own
    CPU-runnable case) and, for both, the C oracle timed on the host cores."""
    import bulletproofs_amd as bp
    from bulletproofs_amd import host_mirror

    seed = SEED + 4
    gs, hs, ugh, values, gamma = host_mirror.range_bench_inputs(ctx, 64, 64, seed)
    host_mirror.range_aggregate_roundtrip(ctx, bp.SECP256K1, gs, hs, ugh, values, gamma, 64, seed, seed + 1, reps=1)
    one = host_mirror.range_aggregate_roundtrip(ctx, bp.SECP256K1, gs, hs, ugh, values, gamma, 64, seed, seed + 1, reps=5)
    thr = host_mirror.range_aggregate_throughput(device, bp.SECP256K1, gs, hs, ugh, values, gamma, 64, workers,
                                                 per_worker, seed, seed + 1)
    if not (one["ok"] and thr["ok"]):
        raise SystemExit("aggregated range proof failed to verify")
    # 64 proofs over different values through shared launches from ONE host thread (bp_range_batch_*); every proof
    # of the first batch is verified
    rng = np.random.default_rng(seed + 8)
    pvals = rng.integers(0, 1 << 64, (64, 64), dtype=np.uint64)
    pgam = rng.integers(0, 1 << 64, (64, 64, 4), dtype=np.uint64)
    pgam[:, :, 3] &= np.uint64((1 << 63) - 1)
    pb = host_mirror.range_aggregate_prove_batch(ctx, bp.SECP256K1, gs, hs, ugh, pvals, pgam, 64, seed, seed + 1, reps=1, verify=True)
    if not pb["ok"]:
        raise SystemExit("a proof of the batch of 64 failed to verify")
    pb = host_mirror.range_aggregate_prove_batch(ctx, bp.SECP256K1, gs, hs, ugh, pvals, pgam, 64, seed, seed + 1, reps=5)
    batched = {"workload": "64 proofs of 64 x 64 bits proved together, one host thread, one context", "batch_ms": pb["batch_ms"],
               "prove_per_s": pb["per_s"]}
    # the largest batch the library takes: twice the latency of a batch, a fifth more proofs per second
    pvals2 = np.concatenate([pvals, pvals[::-1]])
    pgam2 = np.concatenate([pgam, pgam[::-1]])
    pb2 = host_mirror.range_aggregate_prove_batch(ctx, bp.SECP256K1, gs, hs, ugh, pvals2, pgam2, 64, seed, seed + 1, reps=1, verify=True)
    if not pb2["ok"]:
        raise SystemExit("a proof of the batch of 128 failed to verify")
    pb2 = host_mirror.range_aggregate_prove_batch(ctx, bp.SECP256K1, gs, hs, ugh, pvals2, pgam2, 64, seed, seed + 1, reps=3)
    batched["batch_128"] = {"batch_ms": pb2["batch_ms"], "prove_per_s": pb2["per_s"]}
    # two host threads, a context and a batch of 128 each: one batch's short launches run beside the other's long ones
    import threading

    flight = [None, None]
    ctxs = [bp.Context(device) for _ in flight]

    def in_flight(k):
        host_mirror.range_aggregate_prove_batch(ctxs[k], bp.SECP256K1, gs, hs, ugh, pvals2, pgam2, 64, seed, seed + 1, reps=1)
        flight[k] = host_mirror.range_aggregate_prove_batch(ctxs[k], bp.SECP256K1, gs, hs, ugh, pvals2, pgam2, 64, seed, seed + 1, reps=4)

    workers_ = [threading.Thread(target=in_flight, args=(k,)) for k in range(len(flight))]
    for w_ in workers_:
        w_.start()
    for w_ in workers_:
        w_.join()
    if any(f is None for f in flight):
        raise SystemExit("a batch in flight failed")
    batched["two_batches_of_128_in_flight"] = {"host_threads": 2, "batch_ms_each": [f["batch_ms"] for f in flight],
                                               "prove_per_s": sum(128.0 / f["batch_ms"] * 1e3 for f in flight),
                                               "note": "each thread's own clock over its timed batches; the threads start together "
                                                       "and run the same number of batches"}
    for c_ in ctxs:
        c_.close()
    if not extras:  # the figures that are summed over the GPUs of a node
        return {"workload": "m = 64 values x 64 bits (N = 4096), secp256k1, prove and verify, 1 GPU", "prove_per_s": thr["prove_per_s"],
                "verify_per_s": thr["verify_per_s"], "host_threads": workers, "prove_batch_64": batched,
                "one_at_a_time": {"prove_ms": one["prove_ms"], "verify_ms": one["verify_ms"]}}
    g1, h1 = tuple(None if v is None else v[:64] for v in gs), tuple(None if v is None else v[:64] for v in hs)
    host_mirror.range_aggregate_roundtrip(ctx, bp.SECP256K1, g1, h1, ugh, values[:1], gamma[:1], 64, seed, seed + 1, reps=1,
                                          single=True)
    single = host_mirror.range_aggregate_roundtrip(ctx, bp.SECP256K1, g1, h1, ugh, values[:1], gamma[:1], 64, seed, seed + 1,
                                                   reps=5, single=True)
    if not single["ok"]:
        raise SystemExit("single-value range proof failed to verify")
    # 128 single-value proofs through the same shared launches
    svals = rng.integers(0, 1 << 64, (128, 1), dtype=np.uint64)
    sgam = rng.integers(0, 1 << 64, (128, 1, 4), dtype=np.uint64)
    sgam[:, :, 3] &= np.uint64((1 << 63) - 1)
    sb = host_mirror.range_aggregate_prove_batch(ctx, bp.SECP256K1, g1, h1, ugh, svals, sgam, 64, seed, seed + 1, reps=1, verify=True)
    if not sb["ok"]:
        raise SystemExit("a single-value proof of the batch of 128 failed to verify")
    sb = host_mirror.range_aggregate_prove_batch(ctx, bp.SECP256K1, g1, h1, ugh, svals, sgam, 64, seed, seed + 1, reps=5)
    out = {
        "metric": "aggregated 64-bit range proofs/sec",
        "workload": "m = 64 values x 64 bits (N = 4096), secp256k1, prove and verify, 1 GPU",
        "prove_per_s": thr["prove_per_s"],
        "verify_per_s": thr["verify_per_s"],
        "host_threads": workers,
        "prove_batch_64": batched,
        "one_at_a_time": {"prove_ms": one["prove_ms"], "verify_ms": one["verify_ms"]},
        "note": "host threads with one context each on the same GPU; every proof verified; transcript is a seeded "
                "stand-in for the reference's Fiat-Shamir sponge",
        "single_value_64_bit": {"workload": "one 64-bit value (benches/range_bench.rs:37-62 shape), one at a time, 1 GPU",
                                "prove_ms": single["prove_ms"], "verify_ms": single["verify_ms"],
                                "prove_batch_128": {"batch_ms": sb["batch_ms"], "prove_per_s": sb["per_s"]}},
    }
    # verify_batch at BATCH_SIZE 100 (benches/aggregate_range_bench.rs:65-84): four distinct proofs cycled through a
    # batch of 100, every proof's CRS scalars built and added on the device, one MSM (RangeAggregate::verify_batch)
    rng = np.random.default_rng(seed + 9)
    bvals = rng.integers(0, 1 << 64, (4, 64), dtype=np.uint64)
    bgam = rng.integers(0, 1 << 64, (4, 64, 4), dtype=np.uint64)
    bgam[:, :, 3] &= np.uint64((1 << 63) - 1)
    host_mirror.range_aggregate_verify_batch(ctx, bp.SECP256K1, gs, hs, ugh, bvals, bgam, 64, 100, seed, seed + 1)  # warm-up
    vb = host_mirror.range_aggregate_verify_batch(ctx, bp.SECP256K1, gs, hs, ugh, bvals, bgam, 64, 100, seed, seed + 1)
    if not (vb["ok_direct"] and vb["ok_device"] and vb["ok_host"]):
        raise SystemExit("verify_batch of 100 aggregated range proofs rejected valid proofs")
    out["verify_batch_100"] = {"workload": "verify_batch, 100 proofs of 64 x 64 bits (benches/aggregate_range_bench.rs:65-84)",
                               "ms": vb["direct_ms"], "proofs_per_s": 100.0 / vb["direct_ms"] * 1e3,
                               "host_collect_then_device_merge_ms": vb["collect_ms"] + vb["device_merge_ms"],
                               "host_collect_then_host_merge_ms": vb["collect_ms"] + vb["host_merge_ms"]}
    if cpu:
        out["cpu_baseline"] = cpu_range_baseline(gs, hs, ugh, values, gamma, 64, seed, 3)
        out["single_value_64_bit"]["cpu_baseline"] = cpu_range_baseline(g1, h1, ugh, values[:1], gamma[:1], 64, seed, 5)
    return out


def distributed_legs(ctx, bp, torch, dist, rank, world, local_rank, sharded_cls, args):
    """What the configurations quoted on several GPUs look like over `world` ranks, beside the weak-scaling MSM:
    a 2^20-point MSM cut over the ranks (strong scaling); aggregated range proofs as independent proofs per GPU
    (the scaling that applies to them) and, as the number SURVEY's "expect < 1x" stands for, the 8192-point
    commitment MSM of one proof cut over the ranks; the degree-2^20 - 1 polynomial-commitment flow with its
    vectors spread over the ranks by residue class (bp_host_sharded.hpp).  Every rank takes part; rank 0 reports."""
    from bulletproofs_amd import host_mirror
    from bulletproofs_amd.sharded import host_gather, shard_range

    out = {}
    to_dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()  # noqa: E731

    def timed_sharded_msm(n_total, steps):
        b, e = shard_range(n_total, rank, world)
        n = e - b
        px, py = gen_points_on_gpu(ctx, max(n, 1), rank)
        ks = make_inputs(max(n, 1), rank)
        dx, dy, dk = to_dev(px), to_dev(py), to_dev(ks)
        sm = sharded_cls(ctx, bp.SECP256K1, dist=dist, n_total=n_total)
        for _ in range(3):
            sm.msm(dx.data_ptr(), dy.data_ptr(), None, dk.data_ptr(), n)
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            sm.msm(dx.data_ptr(), dy.data_ptr(), None, dk.data_ptr(), n)
        dist.barrier()
        torch.cuda.synchronize()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item()) / steps * 1e3, sm.window_bits()

    # (1) strong scaling of the headline size
    ms, cb = timed_sharded_msm(1 << LOG2_N, args.steps)
    out["msm_strong_2_20"] = {"workload": f"one secp256k1 MSM of 2^{LOG2_N} points cut over {world} GPU(s) by point range, window sums "
                                          f"all-gathered over RCCL, combined on every GPU", "ms_per_step": ms,
                              "points_per_s": (1 << LOG2_N) / ms * 1e3, "window_bits": cb, "steps": args.steps}
    # (2) the commitment MSM of one 64 x 64-bit range proof (2 x 4096 points) cut the same way
    ms, cb = timed_sharded_msm(8192, 50)
    out["range_commitment_msm_8192_sharded"] = {"workload": f"one MSM of 8192 points (the A or S commitment of a 64 x 64-bit range proof, "
                                                            f"src/range/aggregate.rs:80-83) cut over {world} GPU(s)", "ms_per_msm": ms,
                                                "window_bits": cb,
                                                "note": "a chain of short launches per rank plus one collective: cutting it does not "
                                                        "shorten it; range proofs scale as independent proofs per GPU (range_proofs)"}
    # (3) aggregated range proofs: every GPU proves and verifies its own proofs
    if not args.no_range:
        mine = range_proofs(ctx, local_rank, cpu=False, extras=False)
        t = torch.tensor([mine["prove_per_s"], mine["verify_per_s"], mine["prove_batch_64"]["batch_128"]["prove_per_s"]],
                         dtype=torch.float64, device="cuda")
        per_gpu = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(per_gpu, t)
        out["range_proofs"] = {"metric": "aggregated 64-bit range proofs/sec", "workload": mine["workload"].replace("1 GPU", f"{world} GPU(s), "
                               "independent proofs on every GPU"),
                               "prove_per_s": float(sum(v[0].item() for v in per_gpu)), "verify_per_s": float(sum(v[1].item() for v in per_gpu)),
                               "prove_in_batches_of_128_per_s": float(sum(v[2].item() for v in per_gpu)),
                               "per_gpu_prove_in_batches_of_128_per_s": [float(v[2].item()) for v in per_gpu],
                               "per_gpu_prove_per_s": [float(v[0].item()) for v in per_gpu],
                               "per_gpu_verify_per_s": [float(v[1].item()) for v in per_gpu], "host_threads_per_gpu": mine["host_threads"],
                               "one_at_a_time": mine["one_at_a_time"]}
    # (4) polynomial commitment of degree 2^20 - 1 over the ranks: rank r holds generators and coefficients r, r + world, ...
    if not args.no_range and world & (world - 1) == 0:
        n = 1 << 20
        nl = n // world
        px, py = gen_points_on_gpu(ctx, nl, rank, stride=world, first=rank + 1)
        hx, hy = gen_points_on_gpu(ctx, 2, 0, first=n + 1)
        hg = np.stack([np.concatenate([hx[0], hy[0]]), np.concatenate([hx[1], hy[1]])])
        rng = np.random.default_rng(SEED + 55)  # the same whole vectors on every rank, each keeps its slice
        coeffs = rng.integers(0, 1 << 64, (n, 4), dtype=np.uint64)
        coeffs[:, 3] &= np.uint64((1 << 63) - 1)
        rx = rng.integers(0, 1 << 64, (2, 4), dtype=np.uint64)
        rx[:, 3] &= np.uint64((1 << 63) - 1)
        sl = slice(rank, None, world)
        c1, c2 = np.ascontiguousarray(coeffs[sl]), np.ascontiguousarray(coeffs[::-1][sl])
        group = dist.new_group(backend="gloo") if world > 1 else None  # host packets: 24 u64 per rank and round
        gather = host_gather(dist, group) if world > 1 else None
        host_mirror.sharded_poly_roundtrip(ctx, bp.SECP256K1, (px, py, None), hg, c1, None, n, rx[0], rx[1], rank, world, gather,
                                           SEED, SEED + 1)  # warm-up
        res = host_mirror.sharded_poly_roundtrip(ctx, bp.SECP256K1, (px, py, None), hg, c1, c2, n, rx[0], rx[1], rank, world, gather,
                                                 SEED, SEED + 1)
        if not (res["verify_ok"] and res["lazy_ok"] and res["halo_ok"]):
            raise SystemExit("sharded polynomial-commitment opening failed to verify")
        t = torch.tensor([res["prove_ms"], res["verify_ms"]], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        out["poly_commit_2_20_sharded"] = {"workload": f"polynomial commitment opening, degree 2^20 - 1, vectors spread over {world} GPU(s) by "
                                                       f"residue class of the index, CRS slices resident; with the second opening and the fold "
                                                       f"of the two deferred checks verified", "prove_ms": float(t[0].item()),
                                           "verify_ms": float(t[1].item()), "all_verified": True,
                                           "exchange": "24 u64 per rank and round between host buffers (gloo all_gather)"}
    return out


def msm_in_flight(bp, device, dx, dy, dk, n, steps, workers=2):
    """MSM throughput with `workers` host threads, one context (stream and workspace) each, over the
    same resident inputs: the tail of one MSM (bucket reduction, a dependent chain on few lanes)
    overlaps the front of the next.  Reported beside the headline figure, never as it."""
    import threading

    import torch

    ctxs = [bp.Context(device) for _ in range(workers)]
    for c in ctxs:
        c.msm_dev(bp.SECP256K1, dx.data_ptr(), dy.data_ptr(), None, dk.data_ptr(), n)
    torch.cuda.synchronize()
    gate = threading.Barrier(workers + 1)

    def run(c):
        gate.wait()
        for _ in range(steps):
            c.msm_dev(bp.SECP256K1, dx.data_ptr(), dy.data_ptr(), None, dk.data_ptr(), n)

    threads = [threading.Thread(target=run, args=(c,)) for c in ctxs]
    for t in threads:
        t.start()
    gate.wait()
    t0 = time.perf_counter()
    for t in threads:
        t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"value": workers * steps * n / dt, "unit": "points/s", "workers": workers, "steps_each": steps,
            "ms_per_msm": dt / (workers * steps) * 1e3}


def msm_sizes(ctx, bp, dx, dy, dk, n_max, steps):
    """the five sizes of benches/msm_bench.rs (2^16 .. 2^20 points) on prefixes of the headline inputs: wall time per
    MSM over `steps` calls after a warm-up, inputs resident in HBM"""
    out = {}
    for lg in range(16, 21):
        n = 1 << lg
        if n > n_max:
            break
        for _ in range(3):
            ctx.msm_dev(bp.SECP256K1, dx.data_ptr(), dy.data_ptr(), None, dk.data_ptr(), n)
        t0 = time.perf_counter()
        for _ in range(steps):
            ctx.msm_dev(bp.SECP256K1, dx.data_ptr(), dy.data_ptr(), None, dk.data_ptr(), n)
        dt = (time.perf_counter() - t0) / steps
        out[f"2^{lg}"] = {"ms": dt * 1e3, "points_per_s": n / dt, "device_phase_ms": ctx.last_kernel_ms(1)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log2n", type=int, default=LOG2_N)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-range", action="store_true", help="skip the aggregated range proof figures")
    ap.add_argument("--no-in-flight", action="store_true", help="skip the two-MSMs-in-flight figure")
    ap.add_argument("--no-streaming", action="store_true", help="skip the fold / dot / s-vector bandwidth figures")
    ap.add_argument("--no-sizes", action="store_true", help="skip the 2^16 .. 2^20 MSM figures (profiles of the headline kernel alone)")
    ap.add_argument("--no-dist-legs", action="store_true", help="N > 1: only the headline MSM, none of the other multi-GPU legs")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: 2^log2n points on every GPU; strong: 2^log2n points in total, split by shard_range")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import bulletproofs_amd as bp

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    torch.cuda.set_device(local_rank)
    # BP_BENCH_FORCE_DIST=1 runs the collective path even with one rank (rehearsal on a 1-GPU box)
    use_dist = world > 1 or os.environ.get("BP_BENCH_FORCE_DIST") == "1"
    saved_stdout = None
    if use_dist:
        # RCCL prints a version banner on stdout when its first communicator comes up: while the collectives
        # run, file descriptor 1 points at stderr, so that the JSON line is the only thing on stdout
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from bulletproofs_amd.sharded import ShardedMsm, shard_range

    n_total = (1 << args.log2n) * (world if args.scaling == "weak" else 1)
    begin, end = shard_range(n_total, rank, world)
    n = end - begin
    ctx = bp.Context(local_rank)
    px, py = gen_points_on_gpu(ctx, n, rank)
    ks = make_inputs(n, rank)
    to_dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()  # noqa: E731
    dx, dy, dk = to_dev(px), to_dev(py), to_dev(ks)
    sharded = ShardedMsm(ctx, bp.SECP256K1, dist=dist if use_dist else None, n_total=n_total)
    win = sharded.win

    def step():
        if not use_dist:
            return ctx.msm_dev(bp.SECP256K1, dx.data_ptr(), dy.data_ptr(), None, dk.data_ptr(), n)
        # per-rank window sums -> one RCCL all_gather -> window-wise combine on every GPU
        return sharded.msm(dx.data_ptr(), dy.data_ptr(), None, dk.data_ptr(), n)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # A GPU that has been idle runs its first work while its clocks still ramp: about thirty 2^20-point MSMs,
    # 6 % slower than the ones after them (1.40 ms per step measured with W = 3 right after start-up, 1.32 ms
    # with W = 30 or 300).  The probes of the multiplier's peak rate, which the roofline needs anyway, therefore
    # come first, on every rank (0.4 s at full load; one call before them sets up the workspaces), so that the
    # warm-up steps and the timed steps run at the clocks a prover that keeps proving sees, whatever W is.  The
    # probes are repeated after the timed steps and the highest rate of all of them is the peak.
    step()
    ctx.int_mad_peak(PEAK_PROBE_ITERS)  # warm-up: clocks up
    probes = [ctx.int_mad_peak(PEAK_PROBE_ITERS) for _ in range(PEAK_PROBES)]
    for _ in range(args.warmup):
        step()
    acc_ms, dev_ms = [], []
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
        acc_ms.append(ctx.last_kernel_ms(0))
        dev_ms.append(ctx.last_kernel_ms(1))
    barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # the legs every rank takes part in (collectives inside): strong-scaling MSM, range proofs per GPU, sharded polynomial commitment
    legs = None
    if use_dist and args.scaling == "weak" and args.log2n == LOG2_N and not args.no_dist_legs:
        legs = distributed_legs(ctx, bp, torch, dist, rank, world, local_rank, ShardedMsm, args)
    if rank == 0:
        # plan facts: window width and count chosen by the library for this n
        if use_dist:
            cb = sharded.window_bits()
            nw = ctx.msm_shard_windows_dev(bp.SECP256K1, dx.data_ptr(), dy.data_ptr(), None, dk.data_ptr(), n, cb, win.data_ptr())
        else:
            nw, cb = ctx.msm_windows_dev(bp.SECP256K1, dx.data_ptr(), dy.data_ptr(), None, dk.data_ptr(), n, win.data_ptr())
        probes += [ctx.int_mad_peak(PEAK_PROBE_ITERS) for _ in range(PEAK_PROBES)]
        measured_peak = max(probes)
        peak = max(NOMINAL_MAD_PEAK, measured_peak)
        acc_avg_s = float(np.mean(acc_ms)) * 1e-3
        muls_per_launch = float(n) * nw * MADD_MULS
        achieved = muls_per_launch * MAC_PER_MUL / acc_avg_s
        issued = muls_per_launch * MAD_ISSUED_PER_MUL / acc_avg_s
        step_s = elapsed / args.steps
        whole_mac_per_point = WHOLE_STEP_MULS(n, nw, cb) * MAC_PER_MUL
        out = {
            "metric": "MSM points/sec at 2^20",
            "value": n_total * args.steps / elapsed,
            "unit": "points/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": step_s * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "u32 limbs (256-bit integers mod p)",
            "data": "synthetic",
            "config": {
                "workload": (f"standalone secp256k1 MSM, 2^{args.log2n} points/GPU" if args.scaling == "weak" else
                             f"standalone secp256k1 MSM, 2^{args.log2n} points in total") +
                            ", uniform scalars, inputs resident in HBM",
                "points_per_gpu": n,
                "points_total": n_total,
                "curve": "secp256k1",
                "window_bits": cb,
                "windows": nw,
                "parallelism": "1 GPU" if world == 1 else f"{world} GPUs, point-range shards, all-gather of window sums",
                "seed": hex(SEED),
                "device_phase_ms": float(np.mean(dev_ms)),
            },
            "roofline": {
                "bound": "valu-int32-mad (v_mad_u64_u32 issue rate; not HBM, not MFMA)",
                "kernel": "k_chunk_sums (bucket accumulation)",
                "achieved": achieved / 1e12,
                "peak": peak / 1e12,
                "unit": "TMAC/s",
                "frac": achieved / peak,
                "frac_issued": issued / peak,
                "frac_whole_step": float(n) * whole_mac_per_point / step_s / peak,
                "peak_basis": "256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz, one v_mad_u64_u32 per lane and cycle",
                "peak_measured": measured_peak / 1e12,
                "frac_of_measured_peak": achieved / measured_peak,
                "peak_probes": {"count": len(probes), "min": min(probes) / 1e12, "max": measured_peak / 1e12,
                                "note": "a kernel of nothing but dependent-free v_mad_u64_u32, one launch of >= 50 ms per probe; half of them run before the warm-up "
                                        "steps (after a warm-up launch of their own), so that the timed steps see steady "
                                        "clocks, half after the timed steps"},
                "avg_launch_ms": acc_avg_s * 1e3,
                "algorithmic_mac_per_point": nw * MADD_MULS * MAC_PER_MUL,
                "issued_mad_per_point": nw * MADD_MULS * MAD_ISSUED_PER_MUL,
                "whole_step_mac_per_point": whole_mac_per_point,
                "note": "frac credits 136 multiply-accumulates per field product (the 8-limb Montgomery count); the "
                        "kernel works on plain residues and issues 81 v_mad_u64_u32 per product: frac_issued is the "
                        "share of the multiplier's issue slots really used; frac_whole_step prices the whole MSM "
                        "(all launches and the host's part, ms_per_step) at the bucket method's count per point",
                "traffic": (PMC_TRAFFIC_2_20["fetch_bytes"] + PMC_TRAFFIC_2_20["write_bytes"]) if args.log2n == LOG2_N and args.scaling == "weak" else None,
                "traffic_unit": "bytes per launch (FETCH_SIZE + WRITE_SIZE of separate rocprofv3 --pmc runs of this command, taken from "
                                "profiles/r03_v5_pmc_summary.csv, the 2^20-point launches: a constant from profiles/, not counted in this run)",
                "algorithmic_bytes": float(n) * nw * (64 + 4),
            },
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(n)
        if not args.no_range and world == 1:
            out["range_proofs"] = range_proofs(ctx, local_rank, cpu=not args.no_cpu_baseline)
        if world == 1 and not args.no_sizes:
            out["msm_sizes"] = msm_sizes(ctx, bp, dx, dy, dk, n, args.steps)
        if world == 1 and not args.no_in_flight:
            out["msm_two_in_flight"] = msm_in_flight(bp, local_rank, dx, dy, dk, n, args.steps)
        if world == 1 and not args.no_streaming:
            out["streaming"] = streaming_kernels(ctx, torch)
        if legs is not None:
            out["distributed"] = legs
        if world == 1 and not args.no_range:
            out["other_configs"] = other_configs(ctx, px, py, cpu=not args.no_cpu_baseline)
        assert res is not None
        if saved_stdout is not None:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
