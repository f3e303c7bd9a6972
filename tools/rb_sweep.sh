#!/bin/bash
# usage (GPU box): bash tools/rb_sweep.sh -- batched range-proof throughput under a few tuning variables
for lg in 4 5 3; do
  echo "BP_MSM_REDUCE_LPW_LOG=$lg"
  BP_MSM_REDUCE_LPW_LOG=$lg timeout -k 10 120 python3 -u tools/range_batch_bench.py 64 2>/dev/null || exit 1
done
