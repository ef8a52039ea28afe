#!/bin/bash
# cache-path counters only (L2 requests / misses / fabric reads), ours vs the vendor kernel; $1 = shape
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_pmc_gemm_l2
rm -rf $OUT; mkdir -p $OUT
cd $R
SHAPE=${1:-qkv}
run() { name=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -- python3 scratch/r04/pmc_gemm.py $SHAPE > $OUT/$name.log 2>&1 || { echo "$name FAILED"; tail -5 $OUT/$name.log; }; echo "$name done"; }
run sq2 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
run tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
run tcc2 TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_TAG_STALL_sum
run tcp1 TCP_TCC_READ_REQ_sum
python3 scratch/r04/pmc_by_launch.py $OUT/sq2 $OUT/tcc1 $OUT/tcc2 $OUT/tcp1 > $OUT/raw_$SHAPE.txt 2>&1
cat $OUT/raw_$SHAPE.txt
