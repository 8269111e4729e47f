#!/bin/bash
# Static ISA instruction counts of the headline kernel step_kernel_bits<10,20,0,0> of a source tree (compile only):
# whole kernel and the logic wave's code in front of its first s_barrier (rare blocks included), by instruction class.
# usage: tools/isa_count.sh [srcdir] [extra flags...]
SRC=${1:-tetris_gymnasium_amd/csrc}; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "-DTG_FOR_EACH_SIZE(X)=X(10,20)" -I$SRC -Itetris_gymnasium_amd/csrc "$@" -S --cuda-device-only -o /tmp/tg_isa.s $SRC/tg_engine.hip 2>/dev/null || exit 1
python3 - <<'PY'
import re
lines = open('/tmp/tg_isa.s').read().split('\n')
start = next(i for i, l in enumerate(lines) if re.match(r'^_ZN2tg16step_kernel_bitsILi10ELi20ELi0ELi0EEE.*:', l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
body = [l.strip() for l in lines[start + 1:end]]
ins = [l.split()[0] for l in body if l and not l.startswith((';', '.', '_')) and not l.endswith(':')]
def classes(v):
    c = {'VALU': 0, 'SALU': 0, 'LDS': 0, 'VMEM': 0, 'SMEM': 0, 'branch': 0, 'wait/other': 0}
    for i in v:
        if i.startswith('s_cbranch') or i.startswith('s_branch'): c['branch'] += 1
        elif i.startswith('s_load') or i.startswith('s_buffer'): c['SMEM'] += 1
        elif i.startswith(('s_waitcnt', 's_nop', 's_barrier', 's_setprio', 's_sleep', 's_endpgm', 's_sendmsg')): c['wait/other'] += 1
        elif i.startswith('s_'): c['SALU'] += 1
        elif i.startswith('ds_'): c['LDS'] += 1
        elif i.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): c['VMEM'] += 1
        elif i.startswith('v_'): c['VALU'] += 1
        else: c['wait/other'] += 1
    return c
print('whole kernel      :', len(ins), classes(ins))
idx = [k for k, l in enumerate(body) if l.startswith('s_setprio 3')]
if idx:
    seg = body[idx[0]:]
    stop = next(k for k, l in enumerate(seg) if l.startswith('s_barrier'))
    v = [l.split()[0] for l in seg[:stop] if l and not l.startswith((';', '.', '_')) and not l.endswith(':')]
    print('logic pre-barrier :', len(v), classes(v))
PY
