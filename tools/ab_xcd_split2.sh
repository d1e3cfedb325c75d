#!/bin/bash
run() { python3 tools/ab_eq_share.py "$@" 2>&1 | grep streams; }
export OHS_EQ_WG_WAVES=4
run 1024
OHS_P1_XCD=4,4 run 1024
OHS_EQ_XCD=0,4 run 1024
OHS_P1_XCD=4,4 OHS_EQ_XCD=0,4 run 1024
OHS_NO_OVERLAP=1 run 1024
OHS_NO_OVERLAP=1 OHS_P1_XCD=4,4 OHS_EQ_XCD=0,4 run 1024
OHS_NO_OVERLAP=1 OHS_P1_XCD=0,4 OHS_EQ_XCD=0,4 run 1024
