#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3c7
mkdir -p $O
AB_ARGS="" bash tools/ab_bench.sh build/libc4_base.so build/libc4_cap2.so build/libc4_cap3.so build/libc4_nospec.so 2>&1 | tee $O/ab.txt
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $O/pytest.txt
