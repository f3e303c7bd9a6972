#!/bin/bash
# usage (on the GPU box, via gpurun): bash tools/prof_round.sh <tag>
# Three separate rocprofv3 runs of the default bench.py command:
#   1. --kernel-trace --stats          -> per-kernel average durations (default command, then the MSM alone)
#   2. --pmc FETCH_SIZE                -> L2 memory-side read bytes per dispatch
#   3. --pmc WRITE_SIZE                -> L2 memory-side write bytes per dispatch
# Summaries land in gpurun_out/<tag>_*; copy the ones to keep into profiles/.
set -o pipefail
tag=$1
out=/root/repo/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_trace -- python3 -u /root/repo/bench.py > $out/${tag}_trace.log 2>&1 || exit 1
grep '^{"metric"' $out/${tag}_trace.log > $out/${tag}_bench_line.json
cp $out/${tag}_trace/*/*kernel_stats.csv $out/${tag}_kernel_stats.csv
# the MSM alone (no range-proof threads, no second context): the averages DESIGN.md quotes per kernel
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_msm_only -- python3 -u /root/repo/bench.py --no-cpu-baseline --no-range --no-in-flight > $out/${tag}_msm_only.log 2>&1 || exit 1
grep '^{"metric"' $out/${tag}_msm_only.log > $out/${tag}_msm_only_bench_line.json
cp $out/${tag}_msm_only/*/*kernel_stats.csv $out/${tag}_msm_only_kernel_stats.csv
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/${tag}_fetch -- python3 -u /root/repo/bench.py --no-cpu-baseline --no-range --no-in-flight --steps 5 --warmup 1 > $out/${tag}_fetch.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/${tag}_write -- python3 -u /root/repo/bench.py --no-cpu-baseline --no-range --no-in-flight --steps 5 --warmup 1 > $out/${tag}_write.log 2>&1 || exit 1
python3 - <<PY
import csv, glob, collections
out = "$out"; tag = "$tag"
rows = []
for name in ("fetch", "write"):
    f = glob.glob(f"{out}/{tag}_{name}/*/*counter_collection.csv")[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "rocprim" in k:
            k = "rocprim:" + k.split("detail::")[-1][:60]
        acc[(k.split("(")[0][:90], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(acc.items()):
        rows.append((k, c, len(v), sum(v) / len(v)))
with open(f"{out}/{tag}_pmc_summary.csv", "w") as fh:
    fh.write("kernel,counter,dispatches,avg_value_KB_raw\n")
    for k, c, n, a in rows:
        fh.write('"%s",%s,%d,%.3f\n' % (k, c, n, a))
        if "bp::" in k:
            print(k[:70].ljust(70), c, n, "%.1f" % a)
PY
python3 - <<PY
import csv
for r in csv.DictReader(open("$out/${tag}_kernel_stats.csv")):
    print(r['Name'][:60].ljust(60), r['Calls'].rjust(5), "%12.1f" % float(r['AverageNs']), r['Percentage'])
PY
cut -c1-400 $out/${tag}_bench_line.json
