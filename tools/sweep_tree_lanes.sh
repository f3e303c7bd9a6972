# usage (GPU box): bash tools/sweep_tree_lanes.sh -- aggregated range proof rates against the lane budget of k_bucket_tree
for cap in 65536 32768 16384 8192; do echo "cap=$cap"; BP_MSM_TREE_LANES=$cap timeout -k 10 100 python bench_range.py 64 16 | grep -o '"prove_ms": [0-9.]*, "verify_ms": [0-9.]*\|"workers": 16, "prove_per_s": [0-9.]*, "verify_per_s": [0-9.]*' || exit 1; done
