#!/bin/bash
# Runs ON THE GPU BOX: functional rule-set check + A/B of the working tree against a snapshot (round 4).
# usage: tools/gpu_r4_fn.sh <tag> [pytest -k expression] [variants...]
TAG=${1:-fn}; shift; KEXPR=${1:-functional}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
VARS=("$@"); [ ${#VARS[@]} -eq 0 ] && VARS=(r03=_ab/r03 cur)
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "$KEXPR" > $OUT/pytest.log 2>&1; echo "== pytest -k '$KEXPR'"; tail -4 $OUT/pytest.log
AB_RULESET=functional timeout 900 python tools/abn_bench.py "${VARS[@]}" > $OUT/abn_fn.log 2>&1; echo "== functional, one transition per launch"; cat $OUT/abn_fn.log
AB_RULESET=functional TG_FUSE=16 timeout 900 python tools/abn_bench.py "${VARS[@]}" > $OUT/abn_fn_fused.log 2>&1; echo "== functional, 16 transitions per launch"; cat $OUT/abn_fn_fused.log
