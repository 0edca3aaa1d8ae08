#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3c6
mkdir -p $O
timeout -k 10 300 python3 -m pytest tests/test_gpu_api.py tests/test_gpu_fused_net.py -m gpu -x -q 2>&1 | tail -4 | tee $O/pytest_quick.txt
grep -q passed $O/pytest_quick.txt || exit 1
AB_ARGS="" bash tools/ab_bench.sh build/libc4_single.so build/libc4_pair.so 2>&1 | tee $O/ab.txt
AB_ARGS="--slots 8192" bash tools/ab_bench.sh build/libc4_single.so build/libc4_pair.so 2>&1 | tee $O/ab8192.txt
AB_ARGS="--net-precision f16" bash tools/ab_bench.sh build/libc4_single.so 2>&1 | tee $O/ab_f16.txt
C4_NET_PRECISION=f32x3 C4_ENGINE_LIB=$PWD/build/libc4_pair_phases.so python3 tools/split_phases.py 4096 2>&1 | grep -v amdgpu.ids | tee $O/phases.txt
C4_ENGINE_LIB=$PWD/build/libc4_pair_stamps.so python3 tools/split_stamps.py 4096 2>&1 | grep -v amdgpu.ids | tee $O/stamps.txt
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $O/pytest.txt
