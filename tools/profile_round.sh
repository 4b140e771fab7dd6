#!/bin/bash
# Run on the GPU box (through gpurun): kernel-time stats of the bench's config-2 step (hipGraph replay) and of an eager
# run, then HBM traffic / SQ counters in separate --pmc passes.  Everything lands in gpurun_out/prof_round/; the summaries
# worth keeping are copied into profiles/ by hand.  All passes measure the default workload only (--no-secondary: the extra
# config-4 measurement launches the same kernels on office0 / 5000 rays and would mix into the per-kernel means;
# --no-kernel-events: the event passes include launches that walk every tile).
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}; O="$R/gpurun_out/prof_round"; rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
ARGS="--no-secondary --no-cpu-baseline --no-kernel-events --no-api"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/graph -o g -- python3 $R/bench.py --steps 100 --warmup 10 $ARGS > $O/graph.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/eager -o e -- python3 $R/bench.py --steps 50 --warmup 10 --eager $ARGS > $O/eager.log 2>&1
# Counter passes run the HEADLINE mode (hipGraph replays: no dense zero-fill, persistent gradients) unless PMC_MODE=--eager.
# Infinity-Cache (MALL) counters: whatever this rocprofv3 lists under that name gets a pass of its own; none listed = none exists.
PMC_MODE=${PMC_MODE:-}
rocprofv3 --list-avail > $O/avail.txt 2>&1 || true
MALL=$(grep -o -E "\b[A-Z0-9_]*(MALL|INFINITY_CACHE|L3_HIT|L3_MISS)[A-Z0-9_]*\b" $O/avail.txt | sort -u | head -4 | tr '\n' ' ')
echo "Infinity-Cache (MALL) counters in rocprofv3 --list-avail: ${MALL:-none (the TCC_EA0_RDREQ_DRAM / _GMI / _IO counters classify the DESTINATION of a request, not its residency: a hit rate of the Infinity Cache cannot be read on this stack)}" > $O/mall.txt
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "$MALL"; do
  [ -z "$(echo $c | tr -d ' ')" ] && continue
  tag=$(echo $c | cut -d' ' -f1)
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_$tag -o p -- python3 $R/bench.py --steps 10 --warmup 3 $PMC_MODE $ARGS > $O/pmc_$tag.log 2>&1 || echo "pass $tag failed" >> $O/mall.txt
done
PMC_LAST=${PMC_LAST:-10} python3 $R/tools/pmc_summary.py $O $O/pmc_summary.json > $O/pmc_summary.txt
python3 $R/tools/timeline.py $(find $O/graph -name 'g_kernel_trace.csv' | head -1) > $O/timeline.txt
cat $O/timeline.txt
cat $O/mall.txt
grep -h '"metric"' $O/graph.log | cut -c1-260
