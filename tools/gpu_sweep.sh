#!/bin/bash
# Runs ON THE GPU BOX: bench sweep over TG_BLOCKS_PER_CU (persistent grid size)
OUT=gpurun_out/${1:-sweep}; mkdir -p $OUT
for b in ${2:-2 4 6 7 8}; do
  TG_BLOCKS_PER_CU=$b timeout 300 python bench.py --steps 512 --warmup 64 --no-cpu-baseline > $OUT/bench_bpc$b.json 2>$OUT/err_$b.log
  python - <<PY
import json
d=json.load(open("$OUT/bench_bpc$b.json"))
print("blocks/CU=$b  value=%.3e  ms/step=%.4f  kernel_ms=%.4f frac=%.3f" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"]))
PY
done
