#!/bin/bash
OUT=gpurun_out/${1:-r02t}; mkdir -p $OUT; export TMPDIR=/tmp
AB_OBS=dict AB_STEPS=256 timeout 1200 python tools/abn_bench.py cur nt:-DTG_DICT_NT > $OUT/abn_dict.log 2>&1; echo "abn rc=$?"; cat $OUT/abn_dict.log
timeout 1200 python tools/abn_bench.py cur obsplain:-DTG_NO_NT > $OUT/abn_board.log 2>&1; cat $OUT/abn_board.log
