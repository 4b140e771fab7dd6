#!/usr/bin/env python3
"""Summarise hipcc -Rpass-analysis=kernel-resource-usage remarks: one line per kernel."""
import re, subprocess, sys
src = sys.argv[1]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=off",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"] + sys.argv[2:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"remark: [^:]+:\d+:\d+:\s+(.*?)\s*\[-Rpass", line) or re.search(r":\d+:\d+: remark:\s+(.*?)\s*\[-Rpass", line)
    if not m:
        continue
    t = m.group(1)
    if t.startswith("Function Name:") or t.startswith("Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(.*", "", name.replace("(anonymous namespace)::", ""))
    print(f"{name:55s} VGPR {r.get('VGPRs','?'):>4} AGPR {r.get('AGPRs','?'):>4} scratch {r.get('ScratchSize [bytes/lane]','?'):>5} "
          f"occ {r.get('Occupancy [waves/SIMD]','?'):>2} spillV {r.get('VGPRs Spill','?'):>3} LDS {r.get('LDS Size [bytes/block]','?')}")
