#!/bin/bash
# usage (GPU box): tools/prof_any.sh <tag> <script.py> [args]  -> gpurun_out/tl_<tag>.txt: kernel trace of the LAST 40 kernels
tag=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/prof_$tag
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o k -- python3 $R/$@ > $O.log 2>&1
python3 - $O <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/k_kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
rows = rows[-44:]
t0 = int(rows[0]['Start_Timestamp']); prev = None
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - prev) / 1e3 if prev else 0.0
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {gap:6.1f}  {r['Kernel_Name'][:90]}")
    prev = e
PY
