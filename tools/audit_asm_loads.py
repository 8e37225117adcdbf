#!/usr/bin/env python3
"""Audit of kernels that issue buffer loads from inline asm (fdet_wgrad3x3_x3.hip pipelined kernel,
fdet_stem_x3.hip pipelined forward): hipcc treats an asm load's destination as written when the asm
statement ends, so under register pressure it may copy / spill / read it BEFORE the data has landed.
This scans the device assembly of one kernel and reports every instruction that touches the destination
registers of an asm buffer_load before the next hand-written `s_waitcnt vmcnt` (within 150 lines).

  hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -Iinclude -S --cuda-device-only <src>.hip -o /tmp/k.s
  python tools/audit_asm_loads.py <mangled-kernel-name-prefix> /tmp/k.s        # expect: touches ... : 0
"""
import re,sys
fn=sys.argv[1]; path=sys.argv[2]
lines=open(path).read().split('\n')
i=[k for k,l in enumerate(lines) if l.startswith(fn)][0]
body=lines[i:]
end=[k for k,l in enumerate(body) if 's_endpgm' in l][0]
body=body[:end]
def regs(tok):
    out=set()
    for m in re.finditer(r'\b([va])\[(\d+):(\d+)\]',tok):
        for r in range(int(m.group(2)),int(m.group(3))+1): out.add(m.group(1)+str(r))
    for m in re.finditer(r'\b([va])(\d+)\b',tok): out.add(m.group(1)+m.group(2))
    return out
# walk linearly; a manual wait (inside ASM block) ends the danger window of all loads before it in LINEAR order only if
# it is the wait that covers them -- we approximate: window = until the next manual s_waitcnt OR a label (block boundary) + 200 lines
inasm=False; bad=0; nload=0; pending=[]
for k,l in enumerate(body):
    t=l.strip()
    if t.startswith(';;#ASMSTART'): inasm=True; continue
    if t.startswith(';;#ASMEND'): inasm=False; continue
    if not t or t.startswith(';'): continue
    if t.startswith('.LBB'): continue
    if inasm and t.startswith('buffer_load'):
        pending.append((regs(t.split(',')[0]),k)); nload+=1; continue
    if inasm and 's_waitcnt vmcnt' in t:
        pending=[]; continue
    if t.startswith('s_'): continue
    pending=[(d,k0) for d,k0 in pending if k-k0<150]
    used=regs(t)
    for d,k0 in pending:
        if used & d:
            bad+=1
            if bad<=8: print('TOUCH',k,t[:80],' <- load at',k0)
print(fn[-40:],'loads',nload,'touches within 150 lines before next manual wait:',bad)
