#!/bin/bash
# VGPRs / spills / occupancy of every step-kernel instantiation of a source tree (compile only, no GPU needed)
# usage: tools/kernel_resources.sh [srcdir] [extra -D flags...]
SRC=${1:-tetris_gymnasium_amd/csrc}; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-kernarg-preload-count=16 -mllvm -amdgpu-sched-strategy=max-ilp "-DTG_FOR_EACH_SIZE(X)=X(10,20)" -I$SRC -Itetris_gymnasium_amd/csrc "$@" --cuda-device-only -c -o /tmp/tg_res.o $SRC/tg_engine.hip -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys, re
name = None; row = {}
for l in sys.stdin:
    m = re.search(r'Function Name: (\S+)', l)
    if m:
        name = m.group(1); row = {}
    for k in ('    VGPRs:', 'VGPRs Spill:', 'Occupancy [waves/SIMD]:', 'ScratchSize [bytes/lane]:'):
        if k in l and name and 'step_kernel_bits' in name:
            row[k.strip()] = l.split(k)[1].split('[')[0].strip()
            if len(row) == 4:
                f = re.search(r'ILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)E', name).groups()
                print(f'{f[0]}x{f[1]} ruleset {f[2]} feat {f[3]:>2}: VGPRs {row[\"VGPRs:\"]:>3} spill {row[\"VGPRs Spill:\"]:>2} scratch {row[\"ScratchSize [bytes/lane]:\"]:>3} waves/SIMD {row[\"Occupancy [waves/SIMD]:\"]}')
"
