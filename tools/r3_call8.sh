#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3c8
mkdir -p $O
AB_ARGS="" bash tools/ab_bench.sh build/libc4_spec_always.so build/libc4_idle1.so build/libc4_idle2.so build/libc4_idle3.so build/libc4_nospec.so 2>&1 | tee $O/ab.txt
AB_ARGS="--net-precision f16" bash tools/ab_bench.sh build/libc4_spec_always.so build/libc4_idle1.so build/libc4_idle2.so build/libc4_nospec.so 2>&1 | tee $O/ab_f16.txt
