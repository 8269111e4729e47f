#!/bin/bash
# Runs ON THE GPU BOX: round 6's closing set on the FINAL binary -> gpurun_out/<tag>/.
#   1. the whole `pytest -m gpu` suite
#   2. the driver's bench command and the default bench run (all blocks)
#   3. tools/gpu_profile.sh for EVERY block (kernel trace, FETCH_SIZE / WRITE_SIZE passes, SQ counters) -> prof/latest.json
#   4. long runs of the three differential fuzzers and the parity soak (summaries only are committed)
# usage: tools/gpu_r6_final.sh <tag>
TAG=${1:-r6final}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
t0=$(date +%s)
timeout 2400 python -m pytest tests -m gpu -q --durations=10 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$? in $(( $(date +%s) - t0 )) s"; tail -4 $OUT/pytest_gpu.log
TG_BENCH_BLOCKS_FILE=$OUT/bench_blocks_driver_style.json timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_style.log 2> $OUT/bench_driver_style.err; echo "driver-style bench rc=$?"
tail -1 $OUT/bench_driver_style.log | cut -c1-600
TG_BENCH_BLOCKS_FILE=$OUT/bench_blocks_default.json timeout 900 python3 bench.py > $OUT/bench_default.log 2> $OUT/bench_default.err; echo "default bench rc=$?"
tail -1 $OUT/bench_default.log | cut -c1-600
bash tools/gpu_profile.sh $TAG/prof > $OUT/profile.log 2>&1; tail -3 $OUT/profile.log
for s in 61 62 63 64; do timeout 900 python3 tests/fuzz_parity.py 120 $s > $OUT/fuzz_parity_$s.log 2>&1; echo "fuzz_parity seed $s rc=$?: $(tail -1 $OUT/fuzz_parity_$s.log)"; done
timeout 900 python3 tests/fuzz_drop_in.py 80 61 > $OUT/fuzz_drop_in_61.log 2>&1; echo "fuzz_drop_in rc=$?: $(tail -1 $OUT/fuzz_drop_in_61.log)"
timeout 900 python3 tests/fuzz_vector_env.py 150 61 > $OUT/fuzz_vector_env_61.log 2>&1; echo "fuzz_vector_env rc=$?: $(tail -1 $OUT/fuzz_vector_env_61.log)"
timeout 900 python3 tests/soak_parity.py 6000 > $OUT/soak_parity.log 2>&1; echo "soak rc=$?: $(tail -2 $OUT/soak_parity.log | tr '\n' ' ')"
