#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3c9
mkdir -p $O
line() { python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("exp/s %.1fM sims/s %.1fM games/s %.0f hit %.3f net %.1fM spec %.1fM dtype %s" % (d["value"]/1e6, d["sims_per_sec"]/1e6, d["games_per_sec"], d["eval_cache_hit_rate"], d["net_evals_per_sec"]/1e6, d["speculative_evals_per_sec"]/1e6, d["dtype"]))'; }
B="python3 bench.py --steps 60 --warmup 5 --no-cpu-baseline --precise-compare 0"
for rep in 1 2; do
  for tw in 4 3 2; do
    echo "f32x3 TW=$tw rep$rep $(C4_SPLIT_TW=$tw $B 2>/dev/null | tail -1 | line)"
  done
done 2>&1 | tee $O/tw.txt
echo "f32x3 8192 $($B --slots 8192 2>/dev/null | tail -1 | line)" | tee -a $O/tw.txt
echo "f32x3 8192x3200 $($B --slots 8192 --sims 3200 2>/dev/null | tail -1 | line)" | tee -a $O/tw.txt
