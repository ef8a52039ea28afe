#!/bin/bash
# stamp build (w64_lab) and ablation builds (w64_lab_<name>) of the 64-rows-per-wave forward kernel; run with LCV_ATTN_FWD_W64=1
set -e
cd "$(dirname "$0")"
F="--offload-arch=gfx950 -O3 -std=c++17 -fno-honor-nans -mno-amdgpu-ieee -fno-slp-vectorize -fno-gpu-rdc -I ../../include -I ../../longcat-video-tta_amd/csrc"
/opt/rocm/bin/hipcc $F -c ../../longcat-video-tta_amd/csrc/attn_fwd.hip -o /tmp/fwd_plain.o
/opt/rocm/bin/hipcc $F -c ../../longcat-video-tta_amd/csrc/attn_fwd_pipe.hip -o /tmp/fwd_pipe_plain.o
/opt/rocm/bin/hipcc $F -x hip -c w64_main.cpp -o /tmp/w64_main.o
build() {  # name, extra defines, suffix
  /opt/rocm/bin/hipcc $F -DLCV_ATTN_STAMPS $2 -c ../../longcat-video-tta_amd/csrc/attn_fwd_w64.hip -o /tmp/w64_$1.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 /tmp/w64_$1.o /tmp/fwd_plain.o /tmp/fwd_pipe_plain.o /tmp/w64_main.o -o w64_lab$3
}
build stamp "" ""
if [ "$1" = "variants" ]; then
  build pd3 "-DW64_PD=3" _pd3
  build dma2 "-DW64_DMA_STRIDE=2" _dma2
  build dma3 "-DW64_DMA_STRIDE=3" _dma3
  build pd3dma2 "-DW64_PD=3 -DW64_DMA_STRIDE=2" _pd3dma2
fi
if [ "$1" = "ablation" ]; then
  build no_valu "-DLCV_W64_NO_VALU" _no_valu
  build no_kread "-DLCV_W64_NO_KREAD" _no_kread
  build no_vread "-DLCV_W64_NO_VREAD" _no_vread
  build no_dma "-DLCV_W64_NO_DMA" _no_dma
  build mfma_only "-DLCV_W64_NO_KREAD -DLCV_W64_NO_VREAD -DLCV_W64_NO_DMA -DLCV_W64_NO_VALU" _mfma_only
fi
