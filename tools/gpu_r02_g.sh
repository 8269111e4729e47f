#!/bin/bash
OUT=gpurun_out/${1:-r02g}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python tools/abn_bench.py r01=_ab/r01 cur st1:-DTG_STATE_ST=1 st2:-DTG_STATE_ST=2 stag2:-DTG_WG_STAGGER=2 stag4:-DTG_WG_STAGGER=4 stag6:-DTG_WG_STAGGER=6 > $OUT/abn1.log 2>&1; echo "abn1 rc=$?"; cat $OUT/abn1.log
timeout 1200 python tools/abn_bench.py cur sk256@TG_ALLOC_SKEW=256 sk1k@TG_ALLOC_SKEW=1024 sk4k@TG_ALLOC_SKEW=4096 sk68k@TG_ALLOC_SKEW=69632 sk1m@TG_ALLOC_SKEW=1052672 > $OUT/abn2.log 2>&1; echo "abn2 rc=$?"; cat $OUT/abn2.log
