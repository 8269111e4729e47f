#!/bin/bash
# Runs ON THE GPU BOX: round-2 baseline — GPU tests of the default kernel, bench (N=1 and the self-launched N=2 on one
# GPU), phase timing, copy ceiling.  outputs → gpurun_out/r02a/
OUT=gpurun_out/r02a; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_n1_k20.json 2> $OUT/bench_n1_k20.err; echo "bench k20 rc=$?"; cut -c1-1200 $OUT/bench_n1_k20.json
timeout 600 python bench.py --gpus 1 --no-cpu-baseline > $OUT/bench_n1.json 2> $OUT/bench_n1.err; echo "bench rc=$?"; cut -c1-600 $OUT/bench_n1.json
timeout 600 python bench.py --gpus 2 --steps 64 --warmup 8 > $OUT/bench_n2.json 2> $OUT/bench_n2.err; echo "bench n2 rc=$?"; cut -c1-900 $OUT/bench_n2.json; tail -3 $OUT/bench_n2.err
timeout 300 python tools/phase_timing_bits.py > $OUT/phase.log 2>&1; echo "phase rc=$?"; cat $OUT/phase.log | tail -22
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/copy_ceiling tools/ubench/copy_ceiling.hip 2>/dev/null && timeout 300 /tmp/copy_ceiling > $OUT/copy_ceiling.log 2>&1; cat $OUT/copy_ceiling.log
