#!/bin/bash
# Runs ON THE GPU BOX: batch sweep of the headline block (NUMPY rule-set) and of the functional rule-set on one GPU
# (kernel time by HIP events, median of 7 repetitions).  usage: tools/gpu_sweep.sh [numpy|functional ...]
for RS in ${@:-numpy functional}; do
  EXTRA=""; [ $RS = functional ] && EXTRA="--ruleset functional --obs fn"
  echo "## $RS rule-set, one transition per launch"
  for B in 64 4096 16384 32768 65536 131072 262144 524288; do
    python bench.py --no-cpu-baseline --no-blocks --batch $B --steps 256 --warmup 32 $EXTRA | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('B=%7d  kernel %.2f us  %.3e env steps/s  frac %.3f%s  resident %s  %s' % ($B, r['kernel_ms']*1e3, $B/(r['kernel_ms']*1e-3), r['frac'], (' (by the nibble format\'s own bytes: %.3f)' % r['frac_packed_bytes']) if 'frac_packed_bytes' in r else '', r['resident'], r['kernel'].split('<')[1].split('>')[0]))"
  done
done
