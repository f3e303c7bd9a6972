import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import bulletproofs_amd as bp
ctx = bp.Context(0)
for it in (200, 1000, 4000):
    print(it, "peak %.4e MAC/s" % ctx.int_mad_peak(it))
