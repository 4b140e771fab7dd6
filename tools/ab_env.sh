#!/bin/bash
# usage (GPU box): tools/ab_env.sh "<tag>:<VAR=val ...>" ...   -- the default bench (config 2 only) once per environment, same box
for spec in "$@"; do
  tag=${spec%%:*}; envs=${spec#*:}
  env $envs python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-secondary > gpurun_out/ab_$tag.json 2> gpurun_out/ab_$tag.err || { echo "$tag FAILED"; tail -3 gpurun_out/ab_$tag.err; continue; }
  python - "$tag" <<'PY'
import json, sys
tag = sys.argv[1]
d = json.loads(open(f"gpurun_out/ab_{tag}.json").read().strip().splitlines()[-1])
r = d.get("roofline", {})
print(f"{tag:14s} {d['ms_per_step']*1e3:8.1f} us/step  {d['value']/1e6:6.3f} Mrays/s  bwd {r.get('avg_launch_us', 0):7.1f} us (dense {r.get('avg_launch_us_dense', 0):7.1f})  frac {r.get('frac', 0):.3f} step_frac {r.get('step_frac', 0):.3f} loss {d['loss']:.6f}")
PY
done
