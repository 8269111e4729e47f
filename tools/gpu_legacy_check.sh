#!/bin/bash
# Runs ON THE GPU BOX: the core parity tests (reference traces, NUMPY and FUNCTIONAL rule-sets against the oracle, line clears,
# hold) through each of the earlier step kernels in the test-only library (make -C tetris_gymnasium_amd/csrc legacy):
# wave = one env per wavefront (north_star's mapping), lanes / quad = one env per lane / per four lanes on uint8 probes,
# block = two-phase workgroup kernel.  Built-in board sizes only (a run-time compiled size is refused by tg_create under these kernels).  (The in-step dict pack of the reference traces is the bits kernel's own: not part of this.)  usage: tools/gpu_legacy_check.sh <tag>
TAG=${1:-legacy}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
export TG_LIB=$PWD/tetris_gymnasium_amd/libtetris_gym_hip_legacy.so
K_EXPR="(replays_reference_trace and not in_step_observation and not custom_mappings and not 7x14) or numpy_rules_engine_vs_oracle or functional_rules_engine_vs_oracle or line_clear_dense or swap_and_hold or post_game_over"
{
  echo "# earlier step kernels (csrc/legacy/, test-only library) on the core parity tests: pytest tests/test_gpu_parity.py -k \"$K_EXPR\""
  for K in wave lanes quad block; do
    echo "## TG_KERNEL=$K"
    TG_KERNEL=$K timeout 1200 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "$K_EXPR" 2>&1 | tail -4
    TG_KERNEL=$K timeout 300 python bench.py --steps 128 --warmup 16 --no-cpu-baseline --no-blocks 2>/dev/null | python3 -c "
import sys,json
try:
    d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('bench: %.2f us per 65 536-env step, %.3e env steps/s, kernel %s' % (d['roofline']['kernel_us_hip_events'], d['value'], d['roofline']['kernel']))
except Exception as e: print('bench failed', e)"
  done
  echo "## a board size that is not built in, under an earlier kernel: refused, not launched"
  TG_KERNEL=lanes timeout 120 python3 -c "
from tetris_gymnasium_amd import Engine, EngineError
try:
    Engine(8, 7, 14); print('NOT refused')
except EngineError as e:
    print('refused:', str(e)[:160])"
} > $OUT/legacy_kernels_check.log 2>&1
cat $OUT/legacy_kernels_check.log
