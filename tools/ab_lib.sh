#!/bin/bash
# A/B of two library builds on config 3 without EQ, interleaved in one gpurun call:
#   gpurun -- 'bash tools/ab_lib.sh open_headstage_amd/libohs_hip_nt.so'   (the product library is the other arm)
variant=$1
for i in 1 2 3; do
  python tools/bench_configs.py c3noeq 2>&1 | grep config | sed 's/^/product /'
  OHS_LIB=$variant python tools/bench_configs.py c3noeq 2>&1 | grep config | sed "s|^|$(basename $variant) |"
done
