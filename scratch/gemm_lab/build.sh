#!/bin/bash
set -e
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fgpu-rdc -DLCV_GEMM_STAMPS -I ../../include -I ../../longcat-video-tta_amd/csrc -x hip ../../longcat-video-tta_amd/csrc/gemm.hip -x hip main.cpp -o gemm_lab 2>&1 | grep -E " error|undefined" || true
