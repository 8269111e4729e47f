#!/bin/bash
# Runs ON THE GPU BOX: long runs of the differential fuzzers (first failure of a seed ends that seed).  usage: tools/gpu_fuzz_long.sh <tag> <seeds...>
TAG=$1; shift; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for s in "$@"; do
  timeout 1200 python3 tests/fuzz_parity.py 120 $s > $OUT/fuzz_parity_$s.log 2>&1; echo "fuzz_parity seed $s rc=$?: $(tail -1 $OUT/fuzz_parity_$s.log | cut -c1-300)"
done
for s in "$@"; do
  timeout 600 python3 tests/fuzz_drop_in.py 40 $s > $OUT/fuzz_drop_in_$s.log 2>&1; echo "fuzz_drop_in seed $s rc=$?: $(tail -1 $OUT/fuzz_drop_in_$s.log | cut -c1-300)"
  timeout 600 python3 tests/fuzz_vector_env.py 60 $s > $OUT/fuzz_vector_env_$s.log 2>&1; echo "fuzz_vector_env seed $s rc=$?: $(tail -1 $OUT/fuzz_vector_env_$s.log | cut -c1-300)"
done
