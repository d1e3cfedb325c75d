#!/bin/bash
# XCD partition experiment: EQ launches on XCDs [0, x), overlapped convolution on [x, 8) (OHS_XCD_SPLIT=x)
#   gpurun --timeout 900 -- 'bash tools/ab_xcd_split.sh > gpurun_out/xcd_split.txt 2>&1'
run() { python3 tools/ab_eq_share.py "$@" 2>&1 | grep streams; }
run 256 512 1024 1280 1536
OHS_XCD_SPLIT=1 OHS_EQ_WG_WAVES=4 run 256
OHS_XCD_SPLIT=2 OHS_EQ_WG_WAVES=2 run 256
OHS_XCD_SPLIT=4 OHS_EQ_WG_WAVES=1 run 256
OHS_XCD_SPLIT=2 OHS_EQ_WG_WAVES=4 run 512
OHS_XCD_SPLIT=4 OHS_EQ_WG_WAVES=2 run 512
OHS_XCD_SPLIT=4 OHS_EQ_WG_WAVES=4 run 1024
OHS_XCD_SPLIT=5 OHS_EQ_WG_WAVES=4 run 1024 1280
OHS_XCD_SPLIT=6 OHS_EQ_WG_WAVES=4 OHS_FORCE_OVERLAP=1 run 1280 1536
