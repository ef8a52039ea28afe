#!/bin/bash
cd "$(dirname "$0")"
for w in 4 8; do for x in 0 1; do for v in dq2_lab_full dq3_lab_d2t2 dq3_lab_d1t2; do
  LCV_ATTN_BWD_DQ_WAVES=$w LCV_ATTN_BWD_XCD=$x timeout -k 10 120 ./$v 5 "$v waves=$w xcd=$x" || exit 1
done; done; done
