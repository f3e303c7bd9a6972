# usage (GPU box): bash tools/sweep_chunk.sh <log2n> <chunk> [chunk ...] -- MSM device time per chunk length
l=$1; shift
for c in "$@"; do echo "chunk=$c"; BP_MSM_CHUNK=$c SWEEP_C=0 timeout -k 10 120 python tools/sweep_msm.py $l || exit 1; done
