#!/bin/bash
# Runs ON THE GPU BOX: selected GPU tests + the headline launch three times (a quick A/B after a kernel change).
#   usage: tools/gpu_r5_quick.sh <tag> "<pytest -k expression or empty>" [bench args...]
TAG=${1:-q}; K=$2; shift 2
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
if [[ -n "$K" ]]; then
  timeout 1500 python -m pytest tests -m gpu -x -q -k "$K" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $OUT/pytest.log
fi
for i in 1 2 3; do
  timeout 300 python3 bench.py --no-blocks --no-cpu-baseline --steps 1024 --warmup 64 "$@" > $OUT/bench_$i.log 2>$OUT/bench_$i.err
  python3 - $OUT/bench_$i.log <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print(f"value {d['value']:.4e}  wall {d['ms_per_step']*1e3:.3f} us  hip {r['kernel_us_hip_events']:.3f} us  {r['kernel']}")
PY
done
