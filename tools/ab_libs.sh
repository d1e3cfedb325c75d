#!/bin/bash
# Interleaved timing of several library builds on config 3 without EQ (one gpurun call, three rounds):
#   gpurun -- 'bash tools/ab_libs.sh "" ko_mirror ko_mirror2'      ("" = the product library; name = OHS_BUILD_TAG)
for i in 1 2 3; do
  for v in "$@"; do
    if [ -z "$v" ]; then lib=$PWD/open_headstage_amd/libohs_hip.so; else lib=$PWD/open_headstage_amd/libohs_hip_$v.so; fi
    OHS_LIB=$lib TUNE_P1_TARGETS=8192 python3 tools/tune_p1.py 2>&1 | grep median | sed "s|^|${v:-product} |"
  done
done
