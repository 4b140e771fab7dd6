#!/bin/bash
# usage (GPU box): tools/ab_kernels.sh <tag>[=<lib.so>] ...   -- per-kernel average durations (rocprofv3 --stats over 110 graph
# replays of the default bench step) of each library on the SAME box, one line per library.  Step times of differently linked
# libraries differ by a few us for reasons unrelated to the change under test; per-kernel averages do not.
for spec in "$@"; do
  tag=${spec%%=*}; lib=""; [ "$spec" != "$tag" ] && lib=${spec#*=}
  tools/prof_timeline.sh $tag $lib > /dev/null 2>&1
  python3 - $tag <<'PY'
import csv, sys
t = sys.argv[1]
d = {}
for r in csv.DictReader(open(f'gpurun_out/ks_{t}.csv')):
    for k in ('sample_prepare', 'step_kernel', 'render_fwd_ring', 'composite_fwd', 'decoder_bwd_split', 'decoder_chain', 'decoder_scatter', 'decoder_dw', 'grid_scatter'):
        if k in r['Name']: d[k] = float(r['AverageNs']) / 1e3
print(f"{t:10s}", "  ".join(f"{k} {v:6.1f}" for k, v in d.items()), f"  sum {sum(d.values()):6.1f}")
PY
done
