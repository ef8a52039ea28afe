#!/bin/bash
# builds dq2_lab_<variant> executables (pass B alone at the K3-TTA shapes); run.sh runs them all in one call
set -e
cd "$(dirname "$0")"
python3 make_lab.py
F="--offload-arch=gfx950 -O3 -std=c++17 -fno-honor-nans -mno-amdgpu-ieee -fno-slp-vectorize -fno-gpu-rdc -I ../../../include -I ../../../longcat-video-tta_amd/csrc"
/opt/rocm/bin/hipcc $F -x hip -c dq2_main.cpp -o /tmp/dq2_main.o
build() {  # name, defines
  /opt/rocm/bin/hipcc $F $2 -c dq2_lab.hip -o /tmp/dq2_$1.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 /tmp/dq2_$1.o /tmp/dq2_main.o -o dq2_lab_$1
}
build full "" &
build delta_acc "-DDQ2_DELTA_ACC" &
build no_dma "-DDQ2_NO_DMA" &
build no_kvread "-DDQ2_NO_KVREAD" &
build no_tr "-DDQ2_NO_TR" &
wait
build no_valu "-DDQ2_NO_VALU" &
build no_barrier "-DDQ2_NO_BARRIER" &
build no_wait "-DDQ2_NO_WAIT -DDQ2_NO_DMA" &
build no_reads "-DDQ2_NO_KVREAD -DDQ2_NO_TR" &
wait
build no_valu_no_dma "-DDQ2_NO_VALU -DDQ2_NO_DMA" &
build mfma_only "-DDQ2_NO_KVREAD -DDQ2_NO_TR -DDQ2_NO_VALU -DDQ2_NO_DMA" &
build mfma_bare "-DDQ2_NO_KVREAD -DDQ2_NO_TR -DDQ2_NO_VALU -DDQ2_NO_DMA -DDQ2_NO_BARRIER" &
wait
ls dq2_lab_*
# (the half-tile form tried as dq3_lab.hip is the product kernel since; the dq3 builds below compare prefetch depths)
build3() { /opt/rocm/bin/hipcc $F $2 -c dq3_lab.hip -o /tmp/dq3_$1.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 /tmp/dq3_$1.o /tmp/dq2_main.o -o dq3_lab_$1; }
build3 d2t1 "-DDQ3_DEPTH=2 -DDQ3_TRD=1" &
build3 d2t2 "-DDQ3_DEPTH=2 -DDQ3_TRD=2" &
build3 d2t3 "-DDQ3_DEPTH=2 -DDQ3_TRD=3" &
build3 d1t2 "-DDQ3_DEPTH=1 -DDQ3_TRD=2" &
wait
build3 d3t2 "-DDQ3_DEPTH=3 -DDQ3_TRD=2" &
build3 d1t1 "-DDQ3_DEPTH=1 -DDQ3_TRD=1" &
wait
