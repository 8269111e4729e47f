#!/bin/bash
OUT=gpurun_out/${1:-r02r}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python tools/abn_bench.py cur wbb4:-DTG_WBB=4 wbb7:-DTG_WBB=7 cur2 > $OUT/abn.log 2>&1; echo "abn rc=$?"; cat $OUT/abn.log
