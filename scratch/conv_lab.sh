#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for lab in ${LABS:-0 8}; do echo "LAB=$lab"; LCV_CONV_LAB=$lab timeout -k 10 100 python scratch/pmc_conv.py 5 2>&1 | grep conv_rows; done
