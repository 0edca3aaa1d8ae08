#!/bin/bash
# A/B of the fused kernels of ONE build on the headline bench: tools/ab_mode.sh <mode> [<mode> ...]   (C4_FUSED_MODE values;
# extra bench args via AB_ARGS).  Each mode runs twice, interleaved.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for rep in 1 2; do
  for mode in "$@"; do
    out=$(C4_FUSED_MODE=$mode python3 bench.py --steps 60 --warmup 5 --no-cpu-baseline --precise-compare 0 --profile-steps 0 ${AB_ARGS:-} 2>/dev/null | tail -1)
    echo "$mode rep$rep $(echo "$out" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("exp/s %.1fM sims/s %.1fM games/s %.0f hit %.3f depth %.2f net %.1fM" % (d["value"]/1e6, d["sims_per_sec"]/1e6, d["games_per_sec"], d["eval_cache_hit_rate"], d["mean_leaf_depth"], d["net_evals_per_sec"]/1e6))')"
  done
done
