#!/bin/bash
set -e
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-honor-nans -mno-amdgpu-ieee -fgpu-rdc -I ../../include -I ../../longcat-video-tta_amd/csrc -x hip attn_pp.hip -x hip main.cpp -o attn_lab
