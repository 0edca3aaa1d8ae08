#!/bin/bash
# Round-3 call 2: lean reference-precision forward (buffer-load weight stream, pass-start fragments in LDS, group barriers,
# packed epilogue) against the round-2 forward: bit-identity, phases alone and inside the split kernel, headline A/B.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3c2
mkdir -p $O
python3 tools/compare_net_builds.py build/libc4_old.so build/libc4_lean.so build/libc4_lean_nosgb.so build/libc4_lean_nopk.so 2>&1 | grep -v amdgpu.ids | tee $O/identity.txt
for lib in old lean lean_nosgb lean_nopk; do
  for act in 1 4; do
    echo "== $lib active $act" | tee -a $O/net_alone.txt
    C4_ENGINE_LIB=$PWD/build/libc4_$lib.so C4_NET_WAVE_ACTIVE=$act C4_NET_STAMPS=1 python3 tools/bench_net.py --precision f32x3 --wave 1 --n 2048 2>&1 | grep -E "fused net|wave 0" | tee -a $O/net_alone.txt
  done
done
AB_ARGS="" bash tools/ab_bench.sh build/libc4_old.so build/libc4_lean.so build/libc4_lean_nosgb.so build/libc4_lean_nopk.so 2>&1 | tee $O/ab.txt
for lib in old_stamps lean_stamps; do
  echo "== $lib" | tee -a $O/stamps.txt
  C4_ENGINE_LIB=$PWD/build/libc4_$lib.so python3 tools/split_stamps.py 4096 2>&1 | grep -v amdgpu.ids | tee -a $O/stamps.txt
done
python3 -m pytest tests/test_gpu_fused_net.py tests/test_gpu_training.py tests/test_gpu_api.py -m gpu -x -q -s 2>&1 | grep -E "passed|failed|Error|error|GPU train|precise" | tee $O/pytest.txt
