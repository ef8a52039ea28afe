#!/bin/bash
# gemm_lab_<variant>: the product's gemm.hip + lib.hip with lab switches in gemm4k.h (make_lab.py); run.sh runs them in one call
set -e
cd "$(dirname "$0")"
python3 make_lab.py
C=../../../longcat-video-tta_amd/csrc
F="--offload-arch=gfx950 -O3 -std=c++17 -fno-gpu-rdc -I . -I ../../../include -I $C"
/opt/rocm/bin/hipcc $F -x hip -c gemm_main.cpp -o /tmp/g4l_main.o 2>/dev/null &
/opt/rocm/bin/hipcc $F -c $C/lib.hip -o /tmp/g4l_lib.o &
wait
build() { /opt/rocm/bin/hipcc $F $2 -c gemm_lab.hip -o /tmp/g4l_$1.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 /tmp/g4l_$1.o /tmp/g4l_lib.o /tmp/g4l_main.o -o gemm_lab_$1; }
for v in "$@"; do
  name=${v%%:*}; defs=${v#*:}
  build $name "$defs" &
done
wait
ls gemm_lab_*
