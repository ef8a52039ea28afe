#!/bin/bash
# stamp build (pipe_lab) and ablation builds (pipe_lab_<name>) of the software-pipelined forward kernel
set -e
cd "$(dirname "$0")"
F="--offload-arch=gfx950 -O3 -std=c++17 -fno-honor-nans -mno-amdgpu-ieee -fno-slp-vectorize -fno-gpu-rdc -I ../../include -I ../../longcat-video-tta_amd/csrc"
/opt/rocm/bin/hipcc $F -c ../../longcat-video-tta_amd/csrc/attn_fwd.hip -o /tmp/fwd_plain.o
/opt/rocm/bin/hipcc $F -x hip -c pipe_main.cpp -o /tmp/pipe_main.o
build() {  # name, extra defines
  /opt/rocm/bin/hipcc $F -DLCV_ATTN_STAMPS $2 -c ../../longcat-video-tta_amd/csrc/attn_fwd_pipe.hip -o /tmp/pipe_$1.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 /tmp/pipe_$1.o /tmp/fwd_plain.o /tmp/pipe_main.o -o pipe_lab$3
}
build stamp "" ""
if [ "$1" = "ablation" ]; then
  build no_valu "-DLCV_PIPE_NO_VALU" _no_valu
  build no_kread "-DLCV_PIPE_NO_KREAD" _no_kread
  build half_kread "-DLCV_PIPE_HALF_KREAD" _half_kread
  build no_vread "-DLCV_PIPE_NO_VREAD" _no_vread
  build half_vread "-DLCV_PIPE_HALF_VREAD" _half_vread
  build half_reads "-DLCV_PIPE_HALF_KREAD -DLCV_PIPE_HALF_VREAD" _half_reads
  build no_dma "-DLCV_PIPE_NO_DMA" _no_dma
  build no_reads "-DLCV_PIPE_NO_KREAD -DLCV_PIPE_NO_VREAD" _no_reads
  build mfma_only "-DLCV_PIPE_NO_KREAD -DLCV_PIPE_NO_VREAD -DLCV_PIPE_NO_DMA -DLCV_PIPE_NO_VALU" _mfma_only
fi
