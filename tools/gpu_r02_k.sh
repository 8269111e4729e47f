#!/bin/bash
OUT=gpurun_out/${1:-r02k}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python tools/abn_bench.py cur stag1:-DTG_WG_STAGGER=1 stag2:-DTG_WG_STAGGER=2 stag3:-DTG_WG_STAGGER=3 stag4:-DTG_WG_STAGGER=4 stg2@TG_STAGGER=2 stg8@TG_STAGGER=8 cur2 > $OUT/abn1.log 2>&1; echo "abn1 rc=$?"; cat $OUT/abn1.log
