#!/bin/bash
# EQ || convolution cut points at high stream counts: step time by OHS_OVERLAP_CUTS
#   gpurun -- 'STREAMS="1024 1536" bash tools/ab_cuts_many.sh "0.25,0.5,0.75,0.92,0.98" "0.2,0.37,..."'
for c in "$@"; do
  for S in ${STREAMS:-1024}; do
    echo "$c $(OHS_OVERLAP_CUTS=$c python3 tools/ab_eq_share.py $S 2>&1 | grep streams | cut -c1-100)"
  done
done
