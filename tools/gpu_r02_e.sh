#!/bin/bash
OUT=gpurun_out/${1:-r02e}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python tools/abn_bench.py r01=_ab/r01 cur reps1:-DTG_RNG_REPS=1 reps1p1:-DTG_RNG_REPS=1:-DTG_RNG_PRIO=1 reps1p2:-DTG_RNG_REPS=1:-DTG_RNG_PRIO=2 reps1p3:-DTG_RNG_REPS=1:-DTG_RNG_PRIO=3 p2:-DTG_RNG_PRIO=2 > $OUT/abn.log 2>&1; echo "abn rc=$?"; cat $OUT/abn.log
echo "=== timeline reps1"; timeout 600 python tools/phase_timeline_bits.py -DTG_STAMP_REALTIME -DTG_RNG_REPS=1 2>&1 | tee -a $OUT/timeline.log | head -24
