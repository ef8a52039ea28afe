#!/bin/bash
set -e
cd "$(dirname "$0")"
F="--offload-arch=gfx950 -O3 -std=c++17 -fno-honor-nans -mno-amdgpu-ieee -fno-slp-vectorize -fno-gpu-rdc -I ../../include -I ../../longcat-video-tta_amd/csrc"
/opt/rocm/bin/hipcc $F -DLCV_ATTN_STAMPS -c ../../longcat-video-tta_amd/csrc/attn_fwd_pipe.hip -o /tmp/pipe_stamp.o
/opt/rocm/bin/hipcc $F -c ../../longcat-video-tta_amd/csrc/attn_fwd.hip -o /tmp/fwd_plain.o
/opt/rocm/bin/hipcc $F -x hip -c pipe_main.cpp -o /tmp/pipe_main.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 /tmp/pipe_stamp.o /tmp/fwd_plain.o /tmp/pipe_main.o -o pipe_lab
