#!/bin/bash
OUT=gpurun_out/${1:-r02i}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python tools/abn_bench.py cur st2c:-DTG_STATE_ST=2:-DTG_COLS_PLAIN st1c:-DTG_STATE_ST=1:-DTG_COLS_PLAIN nont:-DTG_NO_NT cur2 > $OUT/abn1.log 2>&1; echo "abn1 rc=$?"; cat $OUT/abn1.log
