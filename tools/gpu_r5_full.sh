#!/bin/bash
# Runs ON THE GPU BOX: the round's evidence set — GPU suite, the driver's bench command, the three-clock reconciliation, the profile set.
TAG=${1:-r5full}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
tools/gpu_r5_first.sh $TAG
tools/gpu_r5_reconcile.sh ${TAG}_rec
tools/gpu_profile.sh ${TAG}_prof > $OUT/profile.log 2>&1; tail -5 $OUT/profile.log
