#!/bin/bash
# usage (GPU box): bash tools/msm_c_sweep.sh <log2n> <c> [c ...] -- one MSM timing per window width
lg=$1; shift
for c in "$@"; do
  timeout -k 10 60 python3 -u tools/msm_size.py $lg 30 $c 2>/dev/null || echo "{\"log2n\": $lg, \"c\": $c, \"failed\": true}"
done
