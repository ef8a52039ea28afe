#!/bin/bash
# counter passes over the lab binaries of the 4-wave GEMM (one --pmc set per run; kernel trace only)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_g4
mkdir -p $OUT
cd $R/scratch/gemm_lab
for v in "$@"; do
  rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/${v}_p1 -- ./g4_$v 4096 4096 > $OUT/${v}_p1.log 2>&1 || exit 1
  rocprofv3 --pmc SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/${v}_p2 -- ./g4_$v 4096 4096 > $OUT/${v}_p2.log 2>&1 || exit 1
done
cd $R
python3 tools/pmc_raw.py $OUT/*_p1 $OUT/*_p2 --match gemm4w
