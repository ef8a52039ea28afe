#!/bin/bash
# rocprofv3 counter passes over the pass-B lab executables (one PMC set per run, kernel trace only; the program itself after `--`)
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_dq2
mkdir -p $OUT
cd $R/scratch/r04/dq2_lab
export LCV_ATTN_BWD_DQ_WAVES=4
for exe in ${EXES:-dq2_lab_full dq3_lab_d2t2 dq2_lab_no_dma dq2_lab_no_valu dq2_lab_no_reads dq2_lab_mfma_only}; do
  run() { name=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/${exe}_$name -- ./$exe 2 $exe > $OUT/${exe}_$name.log 2>&1; }
  run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
  run sq2 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU
  run sq3 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_BF16
  echo "$exe done"
done
python3 $R/tools/pmc_raw.py $OUT/*_sq1 $OUT/*_sq2 $OUT/*_sq3 --match attn_bwd_dq2 > $OUT/raw.txt
cat $OUT/raw.txt
