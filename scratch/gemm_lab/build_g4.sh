#!/bin/bash
# builds scratch/gemm_lab/g4_<variant> for every ingredient combination of csrc/gemm4w.h (see g4_main.cpp)
set -e
cd "$(dirname "$0")"
CS=../../longcat-video-tta_amd/csrc
L=$(grep -n "void gemm16_nt_kernel" $CS/gemm.hip | cut -d: -f1); sed -n "1,$((L-2))p" $CS/gemm.hip > /tmp/g4pre.h
b() { name=$1; shift; /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -Wno-pass-failed -I $CS -I ../../include "-DG4_LAB_NAME=\"$name\"" "$@" g4_main.cpp -o g4_$name & }
b full
b trivial_epi -DG4_LAB_TRIVIAL_EPI
b no_dma -DG4_LAB_TRIVIAL_EPI -DG4_LAB_NO_DMA
b no_barrier -DG4_LAB_TRIVIAL_EPI -DG4_LAB_NO_BARRIER
b no_reads -DG4_LAB_TRIVIAL_EPI -DG4_LAB_NO_READS
b w8_full -DG4_LAB_NW=8
b w8_trivial_epi -DG4_LAB_NW=8 -DG4_LAB_TRIVIAL_EPI
b w8_no_dma -DG4_LAB_NW=8 -DG4_LAB_TRIVIAL_EPI -DG4_LAB_NO_DMA
b mfma_only -DG4_LAB_TRIVIAL_EPI -DG4_LAB_NO_READS -DG4_LAB_NO_DMA -DG4_LAB_NO_BARRIER
wait
ls g4_*
