#!/bin/bash
# rocprofv3 counter passes over scratch/prof_attn_r03.py (one PMC set per run, kernel trace only)
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_r03
mkdir -p $OUT
cd $R
run() { name=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -- python3 scratch/prof_attn_r03.py 2 fwd,bwd > $OUT/$name.log 2>&1; echo "$name done"; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run sq2 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU
run fetch FETCH_SIZE
run write WRITE_SIZE
python3 tools/pmc_raw.py $OUT/sq1 $OUT/sq2 $OUT/fetch $OUT/write --match attn > $OUT/raw.txt
cat $OUT/raw.txt
