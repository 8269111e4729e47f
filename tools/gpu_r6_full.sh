#!/bin/bash
# Runs ON THE GPU BOX: round 6's verification set -> gpurun_out/<tag>/.
#   1. the whole `pytest -m gpu` suite (timed; the driver runs the same command)
#   2. the driver-style bench (`python bench.py`, all blocks) — the line BENCH_r06.json will hold
#   3. rocprofv3 profile set of the blocks whose kernel changed this round (tools/gpu_profile.sh: kernel trace >= 1 000 dispatches,
#      FETCH_SIZE / WRITE_SIZE in separate --pmc passes, SQ counters)
# usage: tools/gpu_r6_full.sh <tag> [profile blocks...]
TAG=${1:-r6full}; shift; BLOCKS=${@:-config5 config3}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
t0=$(date +%s)
timeout 2400 python -m pytest tests -m gpu -x -q --durations=15 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$? in $(( $(date +%s) - t0 )) s"; tail -25 $OUT/pytest_gpu.log
t0=$(date +%s)
TG_BENCH_BLOCKS_FILE=$OUT/bench_blocks.json timeout 900 python3 bench.py > $OUT/bench_default.log 2> $OUT/bench_default.err; echo "bench rc=$? in $(( $(date +%s) - t0 )) s"
tail -1 $OUT/bench_default.log | cut -c1-3000
bash tools/gpu_profile.sh $TAG/prof $BLOCKS 2>&1 | tail -40
