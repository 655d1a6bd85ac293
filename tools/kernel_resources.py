#!/usr/bin/env python3
"""Compact per-kernel resource table (VGPR/AGPR/SGPR/spill/LDS/occupancy) from hipcc's
-Rpass-analysis=kernel-resource-usage remarks. Usage: tools/kernel_resources.py file.hip [filter]"""
import re, subprocess, sys
src = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", "/dev/null",
                      "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
cur = None; rows = []
for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        name = t.split(":", 1)[1].strip()
        d = re.sub(r"^_ZN3bnb12_GLOBAL__N_1\d+", "", name)
        d = d.replace("DF16b", "bf16,").replace("DF16_", "f16,").replace("Li", "").replace("Lb", "b").replace("E", ",")
        d = re.sub(r",+vNS.*$|,+v$", "", d)
        cur = {"name": d}; rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1); cur[k.strip()] = v.strip()
print(f"{'kernel':70s} VGPR AGPR SGPR spillV scratch   LDS occ")
for r in rows:
    if flt and flt not in r["name"]: continue
    print(f"{r['name'][:70]:70s} {r.get('VGPRs','?'):>4} {r.get('AGPRs','?'):>4} {r.get('TotalSGPRs','?'):>4} "
          f"{r.get('VGPRs Spill','?'):>6} {r.get('ScratchSize [bytes/lane]','?'):>7} {r.get('LDS Size [bytes/block]','?'):>5} "
          f"{r.get('Occupancy [waves/SIMD]','?'):>3}")
