#!/bin/bash
OUT=gpurun_out/r02b; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python tools/phase_timeline_bits.py > $OUT/timeline.log 2>&1; echo "timeline rc=$?"; cat $OUT/timeline.log
AB_RULESET=functional timeout 600 python tools/ab_bench.py -- -DTG_NO_PRIO_SWAP > $OUT/ab_fn.log 2>&1; tail -4 $OUT/ab_fn.log
timeout 900 python -m pytest tests/test_gpu_surfaces.py -m gpu -x -q -k "final_observation or wrapper_class_grouped" > $OUT/pytest_final.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest_final.log
