#!/bin/bash
# usage (GPU box): tools/pmc_icache.sh <tag> [lib]  -- instruction-cache counters of the bench step's kernels
R=${GRAFT_REPO_ROOT:-/root/repo}; O="$R/gpurun_out/pmc_ic_${1:?tag}"; rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
if [ -n "$2" ]; then export ENSLAM_LIB=$R/$2 ENSLAM_LIB_ALLOW_MISSING=1; fi
rocprofv3 --list-avail > $O/avail.txt 2>&1
grep -o -i -E "SQC?_[A-Z_]*(ICACHE|IFETCH|INST_LEVEL|WAIT_INST)[A-Z_]*" $O/avail.txt | sort -u > $O/names.txt
cat $O/names.txt
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --output-format csv -d $O/$tag -o p -- python3 $R/bench.py --steps 10 --warmup 3 --eager --no-secondary --no-cpu-baseline --no-kernel-events > $O/$tag.log 2>&1
  python3 - $O/$tag <<'PY'
import csv, sys, glob, collections, re
fs = glob.glob(sys.argv[1] + '/**/p_counter_collection.csv', recursive=True)
if not fs:
    print("no counters in", sys.argv[1]); sys.exit(0)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for r in csv.DictReader(open(fs[0])):
    m = re.search(r"(render_fwd_ring_kernel|decoder_bwd_split_kernel|decoder_chain_kernel|decoder_dw_kernel|step_kernel|sample_kernel|composite_fwd_kernel)", r["Kernel_Name"])
    if not m: continue
    acc[m.group(1)][r["Counter_Name"]] += float(r["Counter_Value"]); n[m.group(1)][r["Counter_Name"]] += 1
for k in acc:
    print(k, {c: round(v / n[k][c]) for c, v in acc[k].items()})
PY
done
