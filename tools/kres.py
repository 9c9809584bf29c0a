#!/usr/bin/env python3
"""Kernel resource usage (VGPRs / SGPRs / scratch / LDS / occupancy) of one csrc/*.hip file, from hipcc's
-Rpass-analysis=kernel-resource-usage:   python tools/kres.py sinkhorn [regex]"""
import os, re, subprocess, sys
f = sys.argv[1]
pat = re.compile(sys.argv[2] if len(sys.argv) > 2 else ".")
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from repconc_amd.build import FLAGS
r = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, "-Rpass-analysis=kernel-resource-usage", "-c",
                    os.path.join(root, "repconc_amd", "csrc", f + ".hip"), "-o", f"/tmp/kres_{f}.o"],
                   capture_output=True, text=True, cwd="/tmp")
cur, rows = None, {}
for line in r.stderr.splitlines():
    if "error" in line:
        print(line)
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1); rows[cur] = {}; continue
    m = re.search(r"remark:\s+(\S[^:]*): (\S+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = m.group(2)
for k, v in rows.items():
    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip().split("(")[0]
    if pat.search(name):
        print("%-64s vgpr %s agpr %s sgpr %s scratch %s occ %s lds %s" % (
            name[-64:], v.get("VGPRs"), v.get("AGPRs"), v.get("TotalSGPRs"), v.get("ScratchSize [bytes/lane]"),
            v.get("Occupancy [waves/SIMD]"), v.get("LDS Size [bytes/block]")))
