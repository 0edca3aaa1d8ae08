#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3c13
mkdir -p $O
line() { python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("exp/s %.1fM sims/s %.1fM games/s %.0f hit %.3f net %.1fM preroll %.2fs dtype %s" % (d["value"]/1e6, d["sims_per_sec"]/1e6, d["games_per_sec"], d["eval_cache_hit_rate"], d["net_evals_per_sec"]/1e6, d["preroll_s"], d["dtype"]))'; }
B="python3 bench.py --steps 60 --warmup 5 --no-cpu-baseline --precise-compare 0"
for rep in 1 2; do
  for bits in 29 30 28; do
    echo "f32x3 cache 2^$bits rep$rep $($B --eval-cache $bits 2>/dev/null | tail -1 | line)"
  done
done 2>&1 | tee $O/cache.txt
echo "f32x3 cache 2^30 steps 400 $(python3 bench.py --steps 400 --warmup 5 --no-cpu-baseline --precise-compare 0 --eval-cache 30 2>/dev/null | tail -1 | line)" | tee -a $O/cache.txt
echo "f32x3 cache 2^29 steps 400 $(python3 bench.py --steps 400 --warmup 5 --no-cpu-baseline --precise-compare 0 --eval-cache 29 2>/dev/null | tail -1 | line)" | tee -a $O/cache.txt
