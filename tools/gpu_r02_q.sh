#!/bin/bash
OUT=gpurun_out/${1:-r02q}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu.log
python bench.py --no-cpu-baseline --steps 512 > $OUT/bench.json 2> $OUT/bench.err; python3 -c "
import json;d=json.load(open('$OUT/bench.json'));print('head us/step',round(d['ms_per_step']*1e3,3),d['roofline']['kernel_us_per_step_reps'],'frac',round(d['roofline']['frac'],3),'copy',d.get('hbm_copy_gbps'))
for k,b in d['blocks'].items(): print(k, round(b['ms_per_step']*1e3,2),'us', b['roofline']['kernel'], 'frac', round(b['roofline']['frac'],3), b['roofline']['kernel_us_per_step_reps'])"
