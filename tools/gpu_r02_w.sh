#!/bin/bash
OUT=gpurun_out/${1:-r02w}; mkdir -p $OUT; export TMPDIR=/tmp
AB_RULESET=functional timeout 1200 python tools/abn_bench.py head=_ab/head cur noprio:-DTG_NO_PRIO_SWAP > $OUT/abn_fn.log 2>&1; echo "abn rc=$?"; cat $OUT/abn_fn.log
