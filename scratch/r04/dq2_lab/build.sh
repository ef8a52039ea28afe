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
