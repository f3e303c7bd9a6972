# usage (GPU box): bash tools/range_multiproc.sh <processes> <threads per process>
# Aggregated range proof throughput with several processes (each has its own hardware queues).
p=$1; t=$2
pids=""
for i in $(seq 1 $p); do
  timeout -k 10 200 python bench_range.py 64 $t > gpurun_out/range_mp_$i.json 2> gpurun_out/range_mp_$i.err &
  pids="$pids $!"
done
rc=0
for q in $pids; do wait $q || rc=1; done
python - <<PY
import json, glob
tot_p = tot_v = 0
for f in sorted(glob.glob('gpurun_out/range_mp_*.json'))[:$p]:
    d = json.load(open(f))["concurrent"][0]
    tot_p += d["prove_per_s"]; tot_v += d["verify_per_s"]
print("processes=$p threads=$t  prove/s=%.0f verify/s=%.0f" % (tot_p, tot_v))
PY
exit $rc
