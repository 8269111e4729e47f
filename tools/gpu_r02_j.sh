#!/bin/bash
OUT=gpurun_out/${1:-r02j}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python tools/abn_bench.py r01=_ab/r01 cur colsnt:-DTG_COLS_NT recplain:-DTG_REC_PLAIN wbplain:-DTG_WB_PLAIN ringplain:-DTG_RING_PLAIN perlane:-DTG_WB_PER_LANE > $OUT/abn1.log 2>&1; echo "abn1 rc=$?"; cat $OUT/abn1.log
echo "=== timeline"; timeout 600 python tools/phase_timeline_bits.py -DTG_STAMP_REALTIME 2>&1 | tee -a $OUT/timeline.log | head -19
