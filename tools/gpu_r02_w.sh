#!/bin/bash
OUT=gpurun_out/${1:-r02w}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python tools/abn_bench.py cur noearly:-DTG_NO_EARLY_OBS cur2 noearly2:-DTG_NO_EARLY_OBS > $OUT/abn.log 2>&1; echo "abn rc=$?"; cat $OUT/abn.log
AB_RULESET=functional timeout 1200 python tools/abn_bench.py cur noearly:-DTG_NO_EARLY_OBS > $OUT/abn_fn.log 2>&1; cat $OUT/abn_fn.log
