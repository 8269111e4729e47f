#!/bin/bash
# Runs ON THE GPU BOX: quick bench of the default kernel.  usage: tools/gpu_try.sh "VAR=val ... [: bench args]" ...
for cfg in "$@"; do
  envs="${cfg%%:*}"; args=""; [[ "$cfg" == *:* ]] && args="${cfg#*:}"
  echo "== $cfg"; env $envs python bench.py --steps 512 --warmup 64 --no-cpu-baseline $args | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,2),'us/step', round(d['value']/1e9,3),'e9 steps/s  kernel_ms',round(d['roofline']['kernel_ms']*1e3,2))"
done
