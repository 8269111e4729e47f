#!/bin/bash
OUT=gpurun_out/${1:-r02d}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python tools/abn_bench.py r01=_ab/r01 rngc=_ab/rngc early=_ab/early cur > $OUT/abn.log 2>&1; echo "abn rc=$?"; cat $OUT/abn.log
for v in "_ab/rngc 1024" "_ab/early 128" "tetris_gymnasium_amd/csrc 1024"; do set -- $v; echo "=== timeline $1"; TL_SRC=$1 TL_RNG_ENVS=$2 timeout 600 python tools/phase_timeline_bits.py -DTG_STAMP_REALTIME 2>&1 | tee -a $OUT/timeline.log | head -24; done
