#!/bin/bash
# Runs ON THE GPU BOX: batch sweep of the step kernel on one GPU — both rule-sets, one and sixteen transitions per launch (the fused ones
# through tg_rollout with trajectory slots) — HIP-event median of bench.py per size.  What to look for: sizes where a transition costs
# more per env than at its neighbours (a build picked past its residency, a spilling instantiation).  usage: tools/gpu_sweep.sh [numpy|functional ...]
SIZES=${SWEEP_SIZES:-"64 4096 16384 32768 49152 65536 81920 98304 114688 131072 196608 262144 524288"}
for RS in ${@:-numpy functional}; do
  EXTRA=""; [ $RS = functional ] && EXTRA="--ruleset functional --obs fn"
  for FUSE in 1 16; do
    FX=""; [ $FUSE = 16 ] && FX="--fuse 16 --traj --no-secondary"
    echo "## $RS rule-set, $FUSE transition(s) per launch"
    for B in $SIZES; do
      python3 bench.py --no-cpu-baseline --no-blocks --batch $B --steps 256 --warmup 32 --reps 5 $EXTRA $FX 2>/dev/null | python3 -c "
import sys,json
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); r=d['roofline']
us=r['kernel_us_hip_events']/$FUSE
print('B=%7d  %7.2f us per transition  %6.2f ns per 1000 envs  %.3e env steps/s  frac %.3f  %s  <%s>' % ($B, us, us*1e6/$B, $B/(us*1e-6), r['frac_hip_events'], r['resident'], r['kernel'].split('<')[1].split('>')[0]))"
    done
  done
done
