#!/usr/bin/env python3
"""Audit of kernels that issue buffer loads from inline asm (fdet_wgrad3x3_x3.hip pipelined kernel,
fdet_stem_x3.hip pipelined forward): hipcc treats an asm load's destination as written when the asm
statement ends, so under register pressure it may copy / spill / read it BEFORE the data has landed.
This scans the device assembly of one kernel and reports every instruction that touches the destination
registers of an asm buffer_load before the next hand-written `s_waitcnt vmcnt` (within 150 lines).

  hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -Iinclude -S --cuda-device-only <src>.hip -o /tmp/k.s
  python tools/audit_asm_loads.py <mangled-kernel-name-prefix> /tmp/k.s        # expect: touches ... : 0
"""
import re, sys


def _regs(tok):
    out = set()
    for m in re.finditer(r'\b([va])\[(\d+):(\d+)\]', tok):
        for r in range(int(m.group(2)), int(m.group(3)) + 1): out.add(m.group(1) + str(r))
    for m in re.finditer(r'\b([va])(\d+)\b', tok): out.add(m.group(1) + m.group(2))
    return out


def audit(fn_prefix, path, window=150, verbose=True):
    """returns (asm buffer loads found, touches of in-flight destination registers)"""
    lines = open(path).read().split('\n')
    starts = [k for k, l in enumerate(lines) if l.startswith(fn_prefix)]
    if not starts:
        raise KeyError(f"{fn_prefix} not found in {path}")
    body = lines[starts[0]:]
    end = [k for k, l in enumerate(body) if 's_endpgm' in l][0]
    body = body[:end]
    inasm = False; bad = 0; nload = 0; pending = []
    for k, l in enumerate(body):
        t = l.strip()
        if t.startswith(';;#ASMSTART'): inasm = True; continue
        if t.startswith(';;#ASMEND'): inasm = False; continue
        if not t or t.startswith(';') or t.startswith('.LBB'): continue
        if inasm and t.startswith('buffer_load'):
            pending.append((_regs(t.split(',')[0]), k)); nload += 1; continue
        if inasm and 's_waitcnt vmcnt' in t:
            pending = []; continue
        if t.startswith('s_'): continue
        pending = [(d, k0) for d, k0 in pending if k - k0 < window]
        used = _regs(t)
        for d, k0 in pending:
            if used & d:
                bad += 1
                if verbose and bad <= 8: print('TOUCH', k, t[:80], ' <- load at', k0)
    return nload, bad


if __name__ == "__main__":
    n, b = audit(sys.argv[1], sys.argv[2])
    print(sys.argv[1][-40:], 'loads', n, 'touches within 150 lines before next manual wait:', b)
