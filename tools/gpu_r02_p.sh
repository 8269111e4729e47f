#!/bin/bash
OUT=gpurun_out/${1:-r02p}; mkdir -p $OUT; export TMPDIR=/tmp
AB_OBS=dict timeout 1200 python tools/abn_bench.py cur full:-DTG_DICT_FULL nosplit:-DTG_DICT_NOSPLIT fullnosplit:-DTG_DICT_FULL:-DTG_DICT_NOSPLIT > $OUT/abn_dict.log 2>&1; echo "abn rc=$?"; cat $OUT/abn_dict.log
python bench.py --no-cpu-baseline --no-blocks --obs dict --steps 512 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench.py dict us/step', d['ms_per_step']*1e3, d['roofline']['kernel_us_per_step_reps'])"
