#!/bin/bash
OUT=gpurun_out/${1:-r02f}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_gpu.log
timeout 900 python tools/abn_bench.py r01=_ab/r01 cur > $OUT/abn.log 2>&1; echo "abn rc=$?"; cat $OUT/abn.log
echo "=== timeline"; timeout 600 python tools/phase_timeline_bits.py -DTG_STAMP_REALTIME 2>&1 | tee -a $OUT/timeline.log | head -24
echo "=== phases"; timeout 300 python tools/phase_timing_bits.py > $OUT/phase.log 2>&1; tail -20 $OUT/phase.log
