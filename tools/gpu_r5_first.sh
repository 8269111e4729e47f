#!/bin/bash
# Runs ON THE GPU BOX: round-5 first pass — the driver's own bench command (compact last line), the default bench run, the
# GPU parity suite of the default kernel.   outputs → gpurun_out/<tag>/
TAG=${1:-r5a}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
TG_BENCH_BLOCKS_FILE=$OUT/bench_blocks_driver.json timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.log 2> $OUT/bench_driver.err; echo "bench driver-style rc=$?"
tail -1 $OUT/bench_driver.log | wc -c; tail -1 $OUT/bench_driver.log | cut -c1-1500
if [[ " $* " != *" notest "* ]]; then
  timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_gpu.log
fi
