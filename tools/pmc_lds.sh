#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O="$R/gpurun_out/pmc_lds_${1:?tag}"; rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
if [ -n "$2" ]; then export ENSLAM_LIB=$R/$2 ENSLAM_LIB_ALLOW_MISSING=1; fi
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $O -o p -- python3 $R/bench.py --steps 10 --warmup 3 --eager --no-secondary --no-cpu-baseline --no-kernel-events > $O.log 2>&1
python3 - $O <<'PY'
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + '/**/p_counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    import re; m = re.search(r"(render_fwd_ring_kernel|decoder_bwd_split_kernel|step_kernel|sample_kernel|composite_fwd_kernel)", r["Kernel_Name"]); k = m.group(1) if m else "other"
    acc[k][r['Counter_Name']] += float(r['Counter_Value']); 
    if r['Counter_Name'] == 'SQ_WAVE_CYCLES': n[k] += 1
for k in acc:
    if 'render_fwd' in k or 'decoder_bwd' in k:
        print(k, n[k], {c: round(v / max(n[k], 1)) for c, v in acc[k].items()})
PY
