#!/bin/bash
OUT=gpurun_out/${1:-r02m}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_gpu.log
python bench.py --no-cpu-baseline --no-blocks --obs dict --steps 512 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('dict us/step', d['ms_per_step']*1e3, d['roofline']['kernel_us_per_step_reps'], d['roofline']['frac'])"
echo "=== dict phases"; PT_OBS=dict timeout 300 python tools/phase_timing_bits.py > $OUT/phase_dict.log 2>&1; tail -18 $OUT/phase_dict.log
