#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3c4
mkdir -p $O
line() { python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("exp/s %.1fM sims/s %.1fM games/s %.0f hit %.3f net %.1fM spec %.1fM dtype %s" % (d["value"]/1e6, d["sims_per_sec"]/1e6, d["games_per_sec"], d["eval_cache_hit_rate"], d["net_evals_per_sec"]/1e6, d["speculative_evals_per_sec"]/1e6, d["dtype"]))'; }
B="python3 bench.py --steps 60 --warmup 5 --no-cpu-baseline --precise-compare 0"
for rep in 1 2; do
  echo "f32x3 base rep$rep $($B 2>/dev/null | tail -1 | line)"
  echo "f32x3 seg rep$rep $(C4_SPLIT_SEG=1 $B 2>/dev/null | tail -1 | line)"
  echo "f32x3 tw3 rep$rep $(C4_SPLIT_TW=3 $B 2>/dev/null | tail -1 | line)"
  echo "f16 base rep$rep $($B --net-precision f16 2>/dev/null | tail -1 | line)"
  echo "f16 seg rep$rep $(C4_SPLIT_SEG=1 $B --net-precision f16 2>/dev/null | tail -1 | line)"
done 2>&1 | tee $O/seg.txt
echo "f32x3 8192 base $($B --slots 8192 2>/dev/null | tail -1 | line)" | tee -a $O/seg.txt
echo "f32x3 8192 seg $(C4_SPLIT_SEG=1 $B --slots 8192 2>/dev/null | tail -1 | line)" | tee -a $O/seg.txt
python3 -m pytest tests/test_gpu_api.py -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest.txt
