#!/bin/bash
# rocprofv3 counter passes over tools/pmc_workload.py, one counter group per run (SQ: 8 slots per pass; FETCH_SIZE and WRITE_SIZE do not
# fit one pass; no trace domain besides the kernel trace -- MI355X_MICROARCH.md "rocprofv3 PMC slots").  Usage (on the GPU box):
#   bash tools/pmc_passes.sh gpurun_out/pmc_r02      then      python3 tools/pmc_summarise.py gpurun_out/pmc_r02
set -e
OUT=${1:-gpurun_out/pmc}
mkdir -p "$OUT"
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
run() {   # name, counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/$name" -- python3 tools/pmc_workload.py --out "$OUT/workload.json" --osd-counts "$OUT/osd_counts.json" > "$OUT/$name.log" 2>&1
  echo "pass $name done"
}
run sq_a SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES
run sq_b SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run sq_c SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F32
run fetch FETCH_SIZE
run write WRITE_SIZE
run grbm GRBM_GUI_ACTIVE
