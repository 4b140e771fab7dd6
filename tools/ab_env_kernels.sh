#!/bin/bash
# usage: ab_env2.sh VAR tag=value ...  -- per-kernel averages with env VAR set per run (same library)
var=$1; shift
for spec in "$@"; do
  tag=${spec%%=*}; val=${spec#*=}
  env $var=$val tools/prof_timeline.sh $tag > /dev/null 2>&1
  python3 - $tag <<'PY'
import csv, sys
t = sys.argv[1]
d = {}
for r in csv.DictReader(open(f'gpurun_out/ks_{t}.csv')):
    for k in ('sample_kernel', 'sample_prepare', 'step_kernel', 'step_rec', 'render_fwd_ring', 'composite_fwd', 'decoder_bwd_split', 'decoder_chain', 'decoder_scatter', 'decoder_dw'):
        if k in r['Name']: d[k] = (float(r['AverageNs']) / 1e3, int(r['Calls']))
tot = sum(v[0] * v[1] for v in d.values()) / max(d.get('render_fwd_ring', (0, 1))[1], 1)
print(f"{t:8s}", "  ".join(f"{k} {v[0]:6.1f}" for k, v in d.items()), f"  per step {tot:6.1f}")
PY
done
