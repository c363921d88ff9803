#!/usr/bin/env python3
"""Print VGPR/SGPR/LDS/occupancy per kernel of csrc/cgx_kernels.hip (hipcc -Rpass-analysis)."""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "conjugate-gradient_amd", "csrc", os.environ.get("CGX_KERNEL_FILE", "cgx_kernels.hip"))
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I/opt/rocm/include", "-c", src,
       "-o", "/tmp/_cgx_k.o", "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = {}
rows = []
for line in out.splitlines():
    m = re.search(r"remark:\s+([A-Za-z][A-Za-z \[\]/]*?):\s*(.*?) \[-Rpass", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if "Name" in k:
        cur = {"name": v}; rows.append(cur)
    else:
        cur[k] = v
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(.*", "", name)
    if len(sys.argv) > 1 and sys.argv[1] not in name:
        continue
    print("%-60s vgpr=%-4s sgpr=%-4s spill=%s/%s lds=%-6s occ=%s" % (
        name[-44:], r.get("VGPRs"), r.get("TotalSGPRs", r.get("SGPRs")), r.get("VGPR Spill", r.get("VGPRs Spill")),
        r.get("SGPRs Spill", r.get("SGPR Spill")), r.get("LDS Size [bytes/block]"), r.get("Occupancy [waves/SIMD]")))
