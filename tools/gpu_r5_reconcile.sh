#!/bin/bash
# Runs ON THE GPU BOX: three clocks on the same launches, for the headline and the functional block —
#   (a) bench.py alone: wall clock per launch (barrier + sync on both sides) and HIP events on the engine's stream
#   (b) the same command under rocprofv3 --kernel-trace, >= 10 000 launches: per-dispatch mean / median / min and the dispatch period
#   (c) tools/phase_timeline_bits.py: s_memrealtime stamps inside the kernel — the span of one launch on the device
# usage: tools/gpu_r5_reconcile.sh <tag>     outputs → gpurun_out/<tag>/
TAG=${1:-rec}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
declare -A ARGS=( [config3]="" [functional]="--only-block functional" )
COMMON="--no-cpu-baseline --no-blocks --steps 2048 --warmup 64 --reps 5"
for b in config3 functional; do
  env | grep '^TG_' > $OUT/${b}_env.txt   # (the TG_* environment of both runs: none unless the caller set some)
  python3 bench.py $COMMON ${ARGS[$b]} > $OUT/${b}_plain.log 2>&1
  rocprofv3 --kernel-trace -d $OUT/${b}_kt -o kt -- python3 bench.py $COMMON ${ARGS[$b]} > $OUT/${b}_kt.log 2>&1
  python3 profiles/kt_stats.py $OUT/${b}_kt/kt_results.db > $OUT/${b}_kt_stats.md 2>&1
  python3 - $OUT/${b}_plain.log $OUT/${b}_kt.log <<'PY'
import json, sys
for f in sys.argv[1:]:
    d = json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1]); r = d["roofline"]
    print(f"{f}: wall {d['ms_per_step']*1e3:.3f} us/launch  hip-events {r['kernel_us_hip_events']:.3f}  reps {r.get('kernel_us_per_step_reps')}  {r['kernel']}")
PY
  head -8 $OUT/${b}_kt_stats.md | cut -c1-160; tail -2 $OUT/${b}_kt_stats.md
  rm -rf $OUT/${b}_kt
done
python3 tools/phase_timeline_bits.py > $OUT/timeline_config3.log 2>&1; grep "span\|stamped" $OUT/timeline_config3.log | head -12
TL_RULESET=functional python3 tools/phase_timeline_bits.py > $OUT/timeline_functional.log 2>&1; grep "span\|stamped" $OUT/timeline_functional.log | head -12
