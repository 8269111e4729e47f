#!/bin/bash
OUT=gpurun_out/${1:-r02c}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_gpu.log
timeout 600 python bench.py --no-cpu-baseline --no-blocks > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; python3 -c "
import json;d=json.load(open('$OUT/bench.json'));print('us/step',round(d['ms_per_step']*1e3,3),'kernel reps',d['roofline']['kernel_us_per_step_reps'],'copy',d.get('hbm_copy_gbps'))"
timeout 600 python tools/phase_timeline_bits.py -DTG_STAMP_REALTIME > $OUT/timeline.log 2>&1; echo "timeline rc=$?"; cat $OUT/timeline.log
timeout 300 python tools/phase_timing_bits.py > $OUT/phase.log 2>&1; tail -20 $OUT/phase.log
