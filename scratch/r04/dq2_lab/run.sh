#!/bin/bash
# all variants back to back in one process sequence on one box (clock spread between boxes is +-4 %: compare within one call only)
cd "$(dirname "$0")"
for v in full no_dma no_kvread no_tr no_reads no_valu no_valu_no_dma no_barrier no_wait mfma_only mfma_bare full; do
  LCV_ATTN_BWD_DQ_WAVES=${WAVES:-4} timeout -k 10 120 ./dq2_lab_$v 5 $v || exit 1
done
