#!/usr/bin/env python3
"""Compact per-kernel resource table (VGPR/AGPR/SGPR/spills/LDS/occupancy) from hipcc's -Rpass-analysis=kernel-resource-usage.
   python tools/resusage.py resnet_amd/csrc/kernels_gemm.hip [filter]"""
import re, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + ROOT + "/include", "-I" + ROOT + "/resnet_amd/csrc",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None; rows = []
for l in err.splitlines():
    m = re.search(r"remark: .*?(Function Name|Name): (\S+)", l)
    if m:
        cur = {"name": subprocess.run(["c++filt", m.group(2)], capture_output=True, text=True).stdout.strip()}
        rows.append(cur); continue
    m = re.search(r"remark: .*?\s+(\w[\w ]*?): (\d+)", l)
    if m and cur is not None: cur[m.group(1).strip()] = int(m.group(2))
print("%-6s %-6s %-6s %-7s %-7s %-4s %s" % ("VGPR", "AGPR", "SGPR", "spillV", "LDS", "occ", "kernel"))
for r in rows:
    if flt and flt not in r["name"]: continue
    print("%-6s %-6s %-6s %-7s %-7s %-4s %s" % (r.get("VGPRs"), r.get("AGPRs"), r.get("TotalSGPRs"), r.get("VGPR Spill", r.get("VGPRs Spill")),
          r.get("LDS Size"), r.get("Occupancy"), re.sub(r"\(.*", "", r["name"])[:110]))
