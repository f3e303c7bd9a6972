#!/bin/bash
# usage (GPU box): bash tools/msm_cc_sweep.sh <log2n> "<c ...>" "<chunk ...>" -- one MSM timing per (window width, chunk length)
lg=$1
for c in $2; do
  for ch in $3; do
    echo -n "chunk $ch: "
    BP_MSM_CHUNK=$ch timeout -k 10 60 python3 -u tools/msm_size.py $lg 30 $c 2>/dev/null || echo "{\"log2n\": $lg, \"c\": $c, \"failed\": true}"
  done
done
