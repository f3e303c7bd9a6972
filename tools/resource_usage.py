#!/usr/bin/env python3
"""Registers, scratch, LDS and occupancy of every kernel as the compiler reports them
(hipcc -Rpass-analysis=kernel-resource-usage, gfx950), one CSV row per kernel.
usage: python tools/resource_usage.py > profiles/<name>.csv     (no GPU needed; compiles each source)"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "bulletproofs_amd", "csrc")
SOURCES = sys.argv[1:] or ["bp_msm.hip", "bp_reduce.hip", "bp_vec.hip", "bp_curvevec.hip", "bp_wire.hip"]
FIELDS = [("VGPRs", "VGPRs"), ("AGPRs", "AGPRs"), ("SGPRs", "TotalSGPRs"), ("scratch_bytes_per_lane", r"ScratchSize \[bytes/lane\]"),
          ("VGPR_spills", "VGPRs Spill"), ("LDS_bytes_per_block", r"LDS Size \[bytes/block\]"),
          ("occupancy_waves_per_SIMD", r"Occupancy \[waves/SIMD\]")]
print("source,kernel," + ",".join(f[0] for f in FIELDS))
for src in SOURCES:
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-gpu-rdc", "--cuda-device-only",
           "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
    text = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True).stderr
    for block in text.split("remark: Function Name: ")[1:]:
        mangled = block.split()[0]
        name = subprocess.run(["c++filt", mangled], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(.*$", "", name).replace("void ", "")
        row = []
        for _, pat in FIELDS:
            m = re.search(r"remark:\s+" + pat + r": (\d+)", block)
            row.append(m.group(1) if m else "")
        print('%s,"%s",%s' % (src, name, ",".join(row)))
