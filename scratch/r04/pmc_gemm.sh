#!/bin/bash
# rocprofv3 counter passes over scratch/r04/pmc_gemm.py: ours vs the vendor kernel, SQ + cache-path counters
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_pmc_gemm
rm -rf $OUT; mkdir -p $OUT
cd $R
rocprofv3 -L > $OUT/counters_list.txt 2>&1 || true
SHAPE=${1:-proj}
run() { name=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -- python3 scratch/r04/pmc_gemm.py $SHAPE > $OUT/$name.log 2>&1 || { echo "$name FAILED"; tail -5 $OUT/$name.log; }; echo "$name done"; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run sq2 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU
run sq3 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_MFMA
run tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
run tcc2 TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_TAG_STALL_sum TCC_BUBBLE_sum
run tcp1 TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
run tcp2 TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TD_TCP_STALL_CYCLES_sum
run fetch FETCH_SIZE
python3 tools/pmc_raw.py $OUT/sq1 $OUT/sq2 $OUT/sq3 $OUT/tcc1 $OUT/tcc2 $OUT/tcp1 $OUT/tcp2 $OUT/fetch --match "gemm,Cijk" > $OUT/raw_$SHAPE.txt 2>&1
cat $OUT/raw_$SHAPE.txt
