#!/bin/bash
# Developer helper: compile csrc/ppo_mlp_walk.hip, print registers / spills per kernel and, for every K loop, its instruction mix and first waits
# (what to look for: no spills, no v_accvgpr / scratch traffic in the loops, vmcnt(n > 0) at the loop heads)
set -e
C=/root/repo/leibnizgym_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Rpass-analysis=kernel-resource-usage -c -o $C/_obj/ppo_mlp_walk.o $C/ppo_mlp_walk.hip 2>&1 \
  | grep -E "error|warning|Function Name|VGPRs:|AGPRs:|Spill|ScratchSize|Occupancy" | sed 's/.*remark: *//; s/ *\[-Rpass.*//' | paste - - - - - - - - | cut -c1-300
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -S --cuda-device-only -o /tmp/walk.s $C/ppo_mlp_walk.hip 2>/dev/null
python3 - <<'PY'
import re
lines=open('/tmp/walk.s').read().split('\n')
blocks=[];cur=None
for i,l in enumerate(lines):
    m=re.match(r'^(\.LBB\d+_\d+):',l)
    if m:
        cur={'name':m.group(1),'start':i,'ops':{}, 'br':[], 'waits':[]}
        blocks.append(cur)
    elif cur is not None:
        t=l.strip().split()
        if not t or t[0].startswith(';'): continue
        op=t[0]
        key = 'mfma' if op.startswith('v_mfma') else 'vload' if (op.startswith('global_load') or op.startswith('buffer_load')) else 'vstore' if 'store' in op and not op.startswith('ds_') else 'ds' if op.startswith('ds_') else 'acc' if op.startswith('v_accvgpr') else 'scratch' if op.startswith('scratch') else 'valu' if op.startswith('v_') else 'wait' if op=='s_waitcnt' else 'nop' if op=='s_nop' else 'salu'
        cur['ops'][key]=cur['ops'].get(key,0)+1
        if op=='s_waitcnt' and len(cur['waits'])<3: cur['waits'].append(' '.join(t[1:]))
        if op.startswith('s_cbranch') or op=='s_branch': cur['br'].append(t[-1])
for b in blocks:
    if b['ops'].get('mfma',0)>=8 and b['name'] in b['br']:
        print(' LOOP',b['name'],b['start'],b['ops'],b['waits'])
print('flat ops:', sum(1 for l in lines if 'flat_load' in l or 'flat_store' in l))
PY
