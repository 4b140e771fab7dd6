#!/bin/bash
# usage (GPU box): tools/ab_defer.sh [lib.so ...]  -- per-kernel averages of the config-2 step (all gradients) and of the mapper's gradient
# set with ENSLAM_DEFER_SCATTER=1, default library and each given library, same box.
R=${GRAFT_REPO_ROOT:-/root/repo}
run() { tag=$1; shift; ( export "$@"; $R/tools/kstats.sh $tag bench.py --steps 100 --warmup 10 --no-secondary --no-cpu-baseline --no-api --no-kernel-events $VARIANT ) | grep -E "^==|decoder_bwd|grid_scatter"; }
for lib in "" "$@"; do
  L=""; [ -n "$lib" ] && L="ENSLAM_LIB=$R/$lib"
  VARIANT="" run all_$(basename "$lib" .so) ENSLAM_DEFER_SCATTER=1 ENSLAM_LIB_ALLOW_MISSING=1 $L
  VARIANT="--variant mapper_grads" run mg_$(basename "$lib" .so) ENSLAM_DEFER_SCATTER=1 ENSLAM_LIB_ALLOW_MISSING=1 $L
done
