#!/bin/bash
OUT=gpurun_out/${1:-r02x}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python tools/abn_bench.py head=_ab/head cur head2=_ab/head cur2 > $OUT/abn.log 2>&1; echo "abn rc=$?"; cat $OUT/abn.log
TG_FUSE=8 timeout 1200 python tools/abn_bench.py head=_ab/head cur > $OUT/abn_fused.log 2>&1; cat $OUT/abn_fused.log
timeout 300 python tools/phase_timing_bits.py 2>&1 | tail -18 | head -8
