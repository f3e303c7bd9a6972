#!/bin/bash
# usage (on the GPU box, via gpurun): bash tools/pmc_cmd.sh <tag> "<counters>" <python script under the repo> [args...]
# One rocprofv3 --pmc pass (counters within one pass's limits) over the script; prints, per kernel, the launches
# and the mean of every counter per launch.
tag=$1; shift
ctrs=$1; shift
script=$1; shift
out=/root/repo/gpurun_out/pmc_$tag
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --pmc $ctrs --output-format csv -d $out -- python3 -u /root/repo/$script "$@" > $out.log 2>&1 || { tail -5 $out.log; exit 1; }
tail -1 $out.log | cut -c1-200
python3 - <<PY
import csv, glob, collections
f = glob.glob('$out/*/*counter_collection.csv')[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
seen = set()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].replace('void bp::', '')[:44]
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    key = (r['Dispatch_Id'])
    if key not in seen:
        seen.add(key); calls[k] += 1
for k in sorted(acc, key=lambda k: -sum(acc[k].values())):
    print(k.ljust(44), str(calls[k]).rjust(5), ' '.join('%s=%.4g' % (c, v / calls[k]) for c, v in sorted(acc[k].items())))
PY
