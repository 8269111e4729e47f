#!/bin/bash
OUT=gpurun_out/${1:-r02z}; mkdir -p $OUT; export TMPDIR=/tmp
AB_OBS=rgb AB_STEPS=128 timeout 1500 python tools/abn_bench.py cur "e36:-DTG_RGB_ELOG(E)=36" "e40:-DTG_RGB_ELOG(E)=40" "e28:-DTG_RGB_ELOG(E)=28" > $OUT/abn_rgb.log 2>&1; echo "abn rc=$?"; cat $OUT/abn_rgb.log
