#!/bin/bash
OUT=gpurun_out/${1:-r02s}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_surfaces.py -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -12 $OUT/pytest_gpu.log
python bench.py --no-cpu-baseline --no-blocks --obs rgb --steps 256 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('rgb us/step', d['ms_per_step']*1e3, d['roofline']['kernel'], d['roofline']['kernel_us_per_step_reps'])"
