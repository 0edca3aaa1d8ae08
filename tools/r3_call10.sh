#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3c10
mkdir -p $O
python3 tools/compare_net_builds.py --precision f16 build/libc4_head.so build/libc4_f16lean_b3.so build/libc4_f16lean_b2.so 2>&1 | grep -v amdgpu.ids | tee $O/identity.txt
for lib in head f16lean_b3 f16lean_b2; do
  for act in 1 4; do
    echo "== $lib active $act" | tee -a $O/net_alone.txt
    C4_ENGINE_LIB=$PWD/build/libc4_$lib.so C4_NET_WAVE_ACTIVE=$act C4_NET_STAMPS=1 python3 tools/bench_net.py --precision f16 --wave 1 --n 2048 2>&1 | grep -E "fused net|wave 0" | tee -a $O/net_alone.txt
  done
done
AB_ARGS="--net-precision f16" bash tools/ab_bench.sh build/libc4_head.so build/libc4_f16lean_b3.so build/libc4_f16lean_b2.so 2>&1 | tee $O/ab_f16.txt
AB_ARGS="--net-precision f16 --slots 8192" bash tools/ab_bench.sh build/libc4_head.so build/libc4_f16lean_b3.so 2>&1 | tee $O/ab_f16_8192.txt
C4_NET_PRECISION=f16 C4_ENGINE_LIB=$PWD/build/libc4_f16lean_stamps.so python3 tools/split_stamps.py 4096 2>&1 | grep -v amdgpu.ids | tee $O/stamps.txt
