#!/bin/bash
# EQ kernel variants (OHS_BUILD_TAG names; "" = product): bit-exactness tests, then ns per sample of one launch
#   gpurun -- 'bash tools/ab_eq_libs.sh "" vd1 vd2'
for v in "$@"; do
  if [ -z "$v" ]; then lib=$PWD/open_headstage_amd/libohs_hip.so; else lib=$PWD/open_headstage_amd/libohs_hip_$v.so; fi
  echo "== ${v:-product}: $(OHS_LIB=$lib timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -q -x -k 'eq_bit_exact or eq_band_counts or eq_long_stream or sample_routing or partial_last_wave' 2>&1 | tail -1)"
done
for i in 1 2 3; do
  for v in "$@"; do
    if [ -z "$v" ]; then lib=$PWD/open_headstage_amd/libohs_hip.so; else lib=$PWD/open_headstage_amd/libohs_hip_$v.so; fi
    echo "${v:-product} $(OHS_LIB=$lib python3 tools/eq_launch_cost.py 2>&1 | grep -E 'blocks  938|fit')" | tr '\n' ' '; echo
  done
done
