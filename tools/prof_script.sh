#!/bin/bash
# usage (on the GPU box): bash tools/prof_script.sh <tag> <script.py> [args...]
# rocprofv3 kernel trace + stats of an arbitrary script of this repo; prints the per-kernel table.
tag=$1; shift
script=$1; shift
out=/root/repo/gpurun_out/prof_$tag
rm -rf $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 -u /root/repo/$script "$@" > $out.log 2>&1
tail -1 $out.log | cut -c1-300
python3 - <<PY
import csv, glob
f = glob.glob('$out/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    print(r['Name'][:70].ljust(70), r['Calls'].rjust(5), "%12.1f" % float(r['AverageNs']), "%12.1f" % (float(r['TotalDurationNs'])/1e3), r['Percentage'])
PY
