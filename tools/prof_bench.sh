#!/bin/bash
# usage (on the GPU box, via gpurun): bash tools/prof_bench.sh <tag> [bench args...]
# Runs bench.py under rocprofv3 --kernel-trace --stats and prints the per-kernel table.
tag=$1; shift
out=/root/repo/gpurun_out/prof_$tag
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 -u /root/repo/bench.py --no-cpu-baseline --no-range --no-in-flight "$@" > $out.log 2>&1
tail -1 $out.log | cut -c1-200
python3 - <<PY
import csv, glob
f = glob.glob('$out/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    print(r['Name'][:60].ljust(60), r['Calls'].rjust(5), "%12.1f" % float(r['AverageNs']), r['Percentage'])
PY
