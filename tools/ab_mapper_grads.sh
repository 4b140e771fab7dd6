#!/bin/bash
# usage (GPU box): tools/ab_mapper_grads.sh  -- per-kernel averages of the config-2 step with the reference mapper's gradient set
# (grids + colour decoder) under the backward's launch policies, same box.
R=${GRAFT_REPO_ROOT:-/root/repo}
run() { tag=$1; shift; ( export "$@"; $R/tools/kstats.sh $tag bench.py --variant mapper_grads --steps 100 --warmup 10 --no-secondary --no-cpu-baseline --no-api --no-kernel-events ); }
run mg_default ENS_X=0
run mg_defer ENSLAM_DEFER_SCATTER=1
