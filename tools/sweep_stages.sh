for s in 1 2 4 8; do echo "stages=$s"; BP_MSM_STAGES=$s SWEEP_C=0 timeout -k 10 200 python tools/sweep_msm.py 18 19 20 21 22 || exit 1; done
