import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import bp_oracle as o
import bulletproofs_amd as bp
ctx = bp.Context(0)
c = o.SECP256K1
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n = 1 << logn
rng = np.random.default_rng(1)
# synthetic: replicate a small set of valid points (values irrelevant for timing), random scalars below 2^255
pts = o.gen_points_arith(1024, 777, 31337, c)
PX, PY, PI = o.pack_points(pts, c)
idx = rng.integers(0, 1024, n)
PX, PY, PI = PX[idx].copy(), PY[idx].copy(), PI[idx].copy()
K = rng.integers(0, 1 << 63, (n, 4), dtype=np.uint64)
for rep in range(5):
    t = time.time(); ctx.msm_unchecked(c.id, (PX, PY, PI), K); t = time.time() - t
    print("msm 2^%d: %.4fs  acc %.3f ms  dev %.3f ms" % (logn, t, ctx.last_kernel_ms(0), ctx.last_kernel_ms(1)), flush=True)
