#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3c5
mkdir -p $O
for prec in f32x3 f16; do
  echo "== $prec 4096" | tee -a $O/phases.txt
  C4_NET_PRECISION=$prec C4_ENGINE_LIB=$PWD/build/libc4_phases.so python3 tools/split_phases.py 4096 2>&1 | grep -v amdgpu.ids | tee -a $O/phases.txt
done
echo "== f32x3 8192" | tee -a $O/phases.txt
C4_NET_PRECISION=f32x3 C4_ENGINE_LIB=$PWD/build/libc4_phases.so python3 tools/split_phases.py 8192 2>&1 | grep -v amdgpu.ids | tee -a $O/phases.txt
