#!/bin/bash
# Runs ON THE GPU BOX: kernel-trace stats of the secondary surfaces (config 5, DICT, RGB, placement boards).
OUT=gpurun_out/${1:-aux}; mkdir -p $OUT; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/c5 -o c5 -- python3 bench.py --config5 --steps 64 --warmup 8 --no-cpu-baseline > $OUT/c5.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/dict -o dict -- python3 bench.py --obs dict --steps 64 --warmup 8 --no-cpu-baseline > $OUT/dict.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/rgb -o rgb -- python3 bench.py --obs rgb --steps 64 --warmup 8 --no-cpu-baseline > $OUT/rgb.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/pb -o pb -- python3 tools/bench_placements.py > $OUT/pb.log 2>&1
for f in c5 dict rgb pb; do python3 profiles/summarize_rocprof.py $OUT/$f/${f}_results.db step_kernel > $OUT/${f}_summary.md 2>&1; echo "== $f"; head -12 $OUT/${f}_summary.md | cut -c1-160; tail -2 $OUT/$f.log | cut -c1-300; done
rm -f $OUT/*/*.db
