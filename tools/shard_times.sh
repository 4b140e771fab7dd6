#!/bin/bash
# usage (GPU box): tools/shard_times.sh  -> gpurun_out/shard_times.txt: single-GPU step time of the shard sizes of a 1/2/4/8-GPU split
# (config 2 weak: 1000 rays of room0 per GPU; config 4 strong: office0, 5000 rays / N) and the bucket size of the whole batch
# (1-rank RCCL rehearsal, ENSLAM_BENCH_FORCE_COMM=1).  Input of the PROJECTED scaling table in DESIGN.md (not a measurement of N > 1).
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/shard_times.txt; : > $O
cd $R
one() {  # scene rays extra-env...
  scene=$1; rays=$2; shift 2
  env "$@" python bench.py --scene $scene --rays $rays --scaling weak --steps 100 --warmup 10 --no-secondary --no-cpu-baseline --no-kernel-events --no-api 2>/dev/null \
    | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d.get('comm') or {}; print('$scene', $rays, '%.4f ms/step' % d['ms_per_step'], '%.3f Mrays/s' % (d['value']/1e6), 'mode', d['mode'], 'bucket_bytes', c.get('bucket_bytes'), 'pre_ms', c.get('pre_ms'), 'allreduce_ms', c.get('gradient_allreduce_ms'))" >> $O
}
for n in 1000 500 250 125; do one room0 $n A=1; done
for n in 5000 2500 1250 625; do one office0 $n A=1; done
one room0 1000 ENSLAM_BENCH_FORCE_COMM=1
one office0 5000 ENSLAM_BENCH_FORCE_COMM=1
one office0 625 ENSLAM_BENCH_FORCE_COMM=1
cat $O
