#!/bin/bash
# EQ-kernel builds (OHS_BUILD_TAG names; "" = product) in the overlapped headline step: bit-exactness tests, then step and
# EQ-span times of the 256-stream batch step, three interleaved rounds
#   gpurun -- 'bash tools/ab_eq_step.sh "" w4'
for v in "$@"; do
  if [ -z "$v" ]; then lib=$PWD/open_headstage_amd/libohs_hip.so; else lib=$PWD/open_headstage_amd/libohs_hip_$v.so; fi
  echo "== ${v:-product}: $(OHS_LIB=$lib timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -q -x -k 'eq_bit_exact or eq_band_counts or eq_long_stream or sample_routing or partial_last_wave' 2>&1 | tail -1)"
done
for i in 1 2 3; do
  for v in "$@"; do
    if [ -z "$v" ]; then lib=$PWD/open_headstage_amd/libohs_hip.so; else lib=$PWD/open_headstage_amd/libohs_hip_$v.so; fi
    echo "${v:-product} $(OHS_LIB=$lib python3 tools/ab_eq_share.py 256 2>&1 | grep streams | cut -c1-120)"
  done
done
