#!/bin/bash
OUT=gpurun_out/${1:-r02v}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu.log
for K in quad wave; do TG_LIB=$PWD/tetris_gymnasium_amd/libtetris_gym_hip_legacy.so TG_KERNEL=$K timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $OUT/pytest_$K.log 2>&1; echo "legacy $K rc=$?"; tail -2 $OUT/pytest_$K.log; done
TG_KERNEL=quad python -c "
from tetris_gymnasium_amd import Engine
try:
    Engine(64)
    print('UNEXPECTED: product library accepted TG_KERNEL=quad')
except Exception as e: print('ok, refused:', str(e)[:120])"
