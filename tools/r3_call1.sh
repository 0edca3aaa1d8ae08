#!/bin/bash
# Round-3 first measurement call (GPU box, from the repo root): reference-precision baseline at HEAD, tree-wave count,
# phases of the f32x3 network pass (inside the split kernel and alone), then the GPU test suite on the new defaults.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3c1
mkdir -p $O
line() { python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("exp/s %.1fM sims/s %.1fM games/s %.0f hit %.3f depth %.2f net %.1fM spec %.1fM dtype %s" % (d["value"]/1e6, d["sims_per_sec"]/1e6, d["games_per_sec"], d["eval_cache_hit_rate"], d["mean_leaf_depth"], d["net_evals_per_sec"]/1e6, d["speculative_evals_per_sec"]/1e6, d["dtype"]))'; }
B="python3 bench.py --steps 60 --warmup 5 --no-cpu-baseline --precise-compare 0"
for rep in 1 2; do
  for tw in 4 3 2; do
    echo "f32x3 TW=$tw rep$rep $(C4_SPLIT_TW=$tw $B 2>/dev/null | tail -1 | line)"
  done
done 2>&1 | tee $O/tw.txt
echo "f16 TW=4 $($B --net-precision f16 2>/dev/null | tail -1 | line)" | tee -a $O/tw.txt
echo "f16 TW=3 $(C4_SPLIT_TW=3 $B --net-precision f16 2>/dev/null | tail -1 | line)" | tee -a $O/tw.txt
echo "f32x3 8192 TW=4 $($B --slots 8192 2>/dev/null | tail -1 | line)" | tee -a $O/tw.txt
for tw in 4 3; do
  C4_SPLIT_TW=$tw C4_ENGINE_LIB=$PWD/build/libc4_netstamps.so python3 tools/split_stamps.py 4096 2>&1 | grep -v amdgpu.ids | tee -a $O/stamps_tw$tw.txt
done
for act in 8 4 1; do
  C4_NET_WAVE_ACTIVE=$act C4_NET_STAMPS=1 python3 tools/bench_net.py --precision f32x3 --wave 1 --n 2048 2>&1 | grep -v amdgpu.ids | tee -a $O/net_alone.txt
done
C4_NET_WAVE_ACTIVE=4 C4_NET_STAMPS=1 python3 tools/bench_net.py --precision f16 --wave 1 --n 2048 2>&1 | grep -v amdgpu.ids | tee -a $O/net_alone.txt
python3 -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $O/pytest.txt
