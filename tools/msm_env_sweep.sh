#!/bin/bash
# usage (GPU box): bash tools/msm_env_sweep.sh <log2n> <VAR> <value> [value ...] -- one MSM timing per value of a tuning variable
lg=$1; var=$2; shift 2
for v in "$@"; do
  echo "$var=$v"
  env $var=$v timeout -k 10 60 python3 -u tools/msm_size.py $lg 40 2>/dev/null || echo failed
done
