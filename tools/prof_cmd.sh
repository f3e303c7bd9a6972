#!/bin/bash
# usage (on the GPU box, via gpurun): bash tools/prof_cmd.sh <tag> <python script under the repo> [args...]
# Runs the script under rocprofv3 --kernel-trace --stats, prints the per-kernel table and leaves the
# trace under gpurun_out/prof_<tag>/ (the last launches as a timeline: tools/trace_tail.py).
tag=$1; shift
script=$1; shift
out=/root/repo/gpurun_out/prof_$tag
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 -u /root/repo/$script "$@" > $out.log 2>&1
tail -2 $out.log | cut -c1-300
python3 - <<PY
import csv, glob
f = glob.glob('$out/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    print(r['Name'][:60].ljust(60), r['Calls'].rjust(5), "%12.1f" % float(r['AverageNs']), r['Percentage'])
PY
