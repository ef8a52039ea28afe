#!/bin/bash
cd "$(dirname "$0")"
for v in ${VARIANTS:-full skipa1 skipa2 skipa3 skipa4 skipw3 skip33 full}; do LCV_GEMM_TILE=k timeout -k 10 200 ./gemm_lab_$v $v || exit 1; done
