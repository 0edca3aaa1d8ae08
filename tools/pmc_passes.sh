#!/bin/bash
# rocprofv3 PMC passes of the headline bench (one counter set per pass: --pmc is never combined with
# a trace domain other than --kernel-trace).  Usage (on the GPU box, from the repo root):
#   bash tools/pmc_passes.sh <out_dir> [extra bench.py args]
# Summaries: python tools/pmc_summary.py <out_dir>
set -u
OUT=${1:-gpurun_out/pmc}
shift || true
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p "$OUT"
BENCH="python3 bench.py --steps 12 --warmup 2 --no-cpu-baseline --precise-compare 0 --profile-steps 0 $*"
run() {  # name, counters...
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- $BENCH > "$OUT/$name.json" 2> "$OUT/$name.err"
  echo "$name rc=$?"
}
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD
run sq2 SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA
run sq3 SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
