#!/bin/bash
# usage (on the GPU box, via gpurun): bash tools/prof_stream.sh <tag> [k=24]
# Memory-side bytes of the streaming kernels: two rocprofv3 runs of tools/stream_kernels.py, one per
# counter (FETCH_SIZE, WRITE_SIZE: raw values in KiB), summarised per kernel next to the algorithmic
# bytes into gpurun_out/<tag>_stream_pmc_summary.csv.
set -o pipefail
tag=$1
k=${2:-24}
out=/root/repo/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/${tag}_sfetch -- python3 -u /root/repo/tools/stream_kernels.py $k > $out/${tag}_sfetch.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/${tag}_swrite -- python3 -u /root/repo/tools/stream_kernels.py $k > $out/${tag}_swrite.log 2>&1 || exit 1
python3 - <<PY
import csv, glob, collections
out = "$out"; tag = "$tag"; n = 1 << $k
alg = {"k_fold_scalars": (64 * n, 32 * n), "k_dot_partial": (64 * n, 0), "k_svec_expand": (0, 32 * n)}
acc = collections.defaultdict(list)
for name in ("sfetch", "swrite"):
    f = glob.glob(f"{out}/{tag}_{name}/*/*counter_collection.csv")[0]
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"].split("(")[0].replace("void bp::", "").split("<")[0]
        if kn in alg:
            acc[(kn, r["Counter_Name"])].append(float(r["Counter_Value"]))
with open(f"{out}/{tag}_stream_pmc_summary.csv", "w") as fh:
    fh.write("kernel,elements,counter,dispatches,avg_value_KiB_raw,corrected_bytes,algorithmic_bytes,ratio\n")
    for (kn, c), v in sorted(acc.items()):
        # the first launch of a kernel may include one-off traffic: all are kept, the average is reported.
        # On gfx950 FETCH_SIZE tallies the 128-byte requests of a wide coalesced read (16 bytes per lane,
        # what these kernels issue) at 64 bytes: doubled here; WRITE_SIZE reads such stores exactly.
        a = sum(v) / len(v)
        fetch = c == "FETCH_SIZE"
        want = alg[kn][0 if fetch else 1]
        got = a * 1024 * (2 if fetch else 1)
        line = "%s,%d,%s,%d,%.1f,%.0f,%d,%s" % (kn, n, c, len(v), a, got, want, ("%.3f" % (got / want)) if want else "")
        fh.write(line + "\n")
        print(line)
PY
grep '^{' $out/${tag}_sfetch.log
