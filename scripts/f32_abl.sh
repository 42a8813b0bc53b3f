#!/bin/bash
# timing ablations of the 64x64 fp32 GEMM loop (needs the DTF_GEMM_EXPERIMENTS build)
for abl in 0 1 2 4 8 6 7 14 15; do
  echo "ABL=$abl"
  DTF_F32_ABL=$abl timeout -k 10 100 python scripts/f32_kscan.py 1000 || exit 1
done
