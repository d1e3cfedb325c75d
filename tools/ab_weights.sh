#!/bin/bash
# k_conv_p1 chunk-length weights (OHS_P1_WEIGHTS, read once per process): config 3 without EQ, interleaved
#   gpurun -- 'bash tools/ab_weights.sh "1425,1128,825,622" "1465,1146,823,598"'
for i in 1 2 3; do
  for w in "$@"; do
    echo "$w $(OHS_P1_WEIGHTS=$w TUNE_P1_TARGETS=4096 python3 tools/tune_p1.py ${STREAMS:-256} 2>&1 | grep median | cut -d' ' -f5-)"
  done
done
