"""Runs `reps` secp256k1 MSMs of 2^log2n points (inputs resident in HBM) and prints the wall time of one;
a target for rocprofv3.  Usage: python tools/msm_size.py <log2n> [reps] [window bits]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import bulletproofs_amd as bp  # noqa: E402


def main():
    lg = int(sys.argv[1])
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    n = 1 << lg
    ctx = bp.Context(0)
    if len(sys.argv) > 3:
        ctx.set_msm_window(int(sys.argv[3]))
    px, py = bench.gen_points_on_gpu(ctx, n, 0)
    ks = bench.make_inputs(n, 0)
    to_dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()  # noqa: E731
    dx, dy, dk = to_dev(px), to_dev(py), to_dev(ks)
    for _ in range(3):
        ctx.msm_dev(bp.SECP256K1, dx.data_ptr(), dy.data_ptr(), None, dk.data_ptr(), n)
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.msm_dev(bp.SECP256K1, dx.data_ptr(), dy.data_ptr(), None, dk.data_ptr(), n)
    dt = (time.perf_counter() - t0) / reps
    print(json.dumps({"log2n": lg, "c": int(sys.argv[3]) if len(sys.argv) > 3 else None, "ms": dt * 1e3, "device_phase_ms": ctx.last_kernel_ms(1), "points_per_s": n / dt}))


if __name__ == "__main__":
    main()
