#!/bin/bash
cd "$(dirname "$0")"
for v in dq2_lab_full dq3_lab_d1t1 dq3_lab_d1t2 dq3_lab_d2t1 dq3_lab_d2t2 dq3_lab_d2t3 dq3_lab_d3t2 dq2_lab_full; do
  LCV_ATTN_BWD_DQ_WAVES=4 timeout -k 10 120 ./$v 5 $v || exit 1
done
