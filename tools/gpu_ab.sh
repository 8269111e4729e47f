#!/bin/bash
# Runs ON THE GPU BOX: in-process A/B/n of step-kernel builds (tools/abn_bench.py) for the three launch kinds.
# usage: tools/gpu_ab.sh <tag> variant...      e.g.  tools/gpu_ab.sh try1 head=_ab/head cur "x=_ab/x:-DTG_PHASE_TIMING"
TAG=${1:-ab}; shift; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python tools/abn_bench.py "$@" > $OUT/abn.log 2>&1; echo "== one transition per launch"; cat $OUT/abn.log
TG_FUSE=16 timeout 1200 python tools/abn_bench.py "$@" > $OUT/abn_fused.log 2>&1; echo "== 16 transitions per launch"; cat $OUT/abn_fused.log
AB_RULESET=functional timeout 1200 python tools/abn_bench.py "$@" > $OUT/abn_fn.log 2>&1; echo "== functional rule-set"; cat $OUT/abn_fn.log
