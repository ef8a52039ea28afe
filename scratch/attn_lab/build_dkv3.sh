#!/bin/bash
# stamp build of pass A, third form (attn_bwd_dkv3.hip) + timing-only ablations of its merged phase (results are WRONG in those:
# they only tell which ingredient the cycles go to)
set -e
cd "$(dirname "$0")"
F="--offload-arch=gfx950 -O3 -std=c++17 -fno-honor-nans -mno-amdgpu-ieee -fno-slp-vectorize -fno-gpu-rdc -I ../../include -I ../../longcat-video-tta_amd/csrc"
/opt/rocm/bin/hipcc $F -x hip -c dkv3_main.cpp -o /tmp/dkv3_main.o
for v in full NO_VALU NO_TR NO_ROW NO_DMA "NO_VALU -DLCV_DKV3_NO_DMA" "NO_VALU -DLCV_DKV3_NO_DMA -DLCV_DKV3_NO_TR -DLCV_DKV3_NO_ROW"; do
  name=$(echo "$v" | sed 's/ -DLCV_DKV3_/_/g' | tr 'A-Z' 'a-z')
  D="-DLCV_DKV3_$v"; [ "$v" = full ] && D=""
  /opt/rocm/bin/hipcc $F -DLCV_DKV3_STAMPS $D -c ../../longcat-video-tta_amd/csrc/attn_bwd_dkv3.hip -o /tmp/dkv3_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 /tmp/dkv3_$name.o /tmp/dkv3_main.o -o dkv3_lab_$name
done
