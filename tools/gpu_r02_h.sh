#!/bin/bash
OUT=gpurun_out/${1:-r02h}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu.log
timeout 1200 python tools/abn_bench.py r01=_ab/r01 cur perlane:-DTG_WB_PER_LANE "kpre:-mllvm:-amdgpu-kernarg-preload-count=16" stg2@TG_STAGGER=2 stg6@TG_STAGGER=6 > $OUT/abn1.log 2>&1; echo "abn1 rc=$?"; cat $OUT/abn1.log
echo "=== timeline"; timeout 600 python tools/phase_timeline_bits.py -DTG_STAMP_REALTIME 2>&1 | tee -a $OUT/timeline.log | head -19
