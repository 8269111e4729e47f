#!/bin/bash
# Runs ON THE GPU BOX: A/B of the kFeatSeq instantiations (round 6, config 5) — selected parity tests, then config 5 three times with
# TG_SEQ=1 (default) and TG_SEQ=0 (round 5's launch), then the batch sweep at config 5's geometry for both.
#   usage: tools/gpu_r6_seq.sh <tag> ["pytest -k expression"]
TAG=${1:-r6seq}; K=${2:-}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
if [[ -n "$K" ]]; then
  timeout 1500 python -m pytest tests -m gpu -x -q -k "$K" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $OUT/pytest.log
fi
one() {  # $1 = TG_SEQ, $2 = batch, $3 = log
  TG_SEQ=$1 timeout 300 python3 bench.py --config5 --no-blocks --no-cpu-baseline --steps 1024 --warmup 64 --batch $2 > $3 2>$3.err
  python3 - $3 $1 $2 <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print(f"TG_SEQ={sys.argv[2]} config5 geometry, {int(sys.argv[3]):6d} envs: wall {d['ms_per_step']*1e3:.2f} us/launch  hip-events {r['kernel_us_hip_events']:.2f}  -> {d['value']:.3e} env steps/s  frac {r.get('frac_hip_events', 0):.3f}  {r['kernel']}")
PY
}
for i in 1 2 3; do
  for sq in 1 0; do one $sq 65536 $OUT/bench_seq${sq}_$i.log; done
done | tee $OUT/ab_config5.log
for B in 8192 16384 32768 36864 40960 49152 57344 65536 73728 81920 98304 131072; do
  for sq in 1 0; do one $sq $B $OUT/sweep_seq${sq}_$B.log; done
done | tee $OUT/config5_batch_sweep.log
