#!/bin/bash
set -e
cd "$(dirname "$0")"
F="--offload-arch=gfx950 -O3 -std=c++17 -fno-honor-nans -mno-amdgpu-ieee -fno-slp-vectorize -fno-gpu-rdc -I ../../include -I ../../longcat-video-tta_amd/csrc"
/opt/rocm/bin/hipcc $F -DLCV_DKV3_STAMPS -c ../../longcat-video-tta_amd/csrc/attn_bwd_dkv3.hip -o /tmp/dkv3_stamp.o
/opt/rocm/bin/hipcc $F -x hip -c dkv3_main.cpp -o /tmp/dkv3_main.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 /tmp/dkv3_stamp.o /tmp/dkv3_main.o -o dkv3_lab
