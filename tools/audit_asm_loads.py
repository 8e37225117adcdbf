#!/usr/bin/env python3
"""Audit of kernels that issue buffer loads from inline asm (fdet_wgrad3x3_x3.hip pipelined kernel,
fdet_stem_x3.hip pipelined kernels): hipcc treats an asm load's destination as written when the asm
statement ends, so under register pressure it may copy / spill / read it BEFORE the data has landed, and it
knows nothing of the hand-counted `s_waitcnt vmcnt(N)` that retires the loads.

The scan walks the device assembly of one kernel and models the VM counter the way the hardware does:

  * every asm `buffer_load` joins a FIFO of in-flight loads (destination registers, line);
  * an asm `s_waitcnt vmcnt(N)` retires all but the N youngest entries (counts are in issue order);
  * between an asm load and the wait that retires it
      - any instruction (compiler or asm) that reads or writes an in-flight destination register is a TOUCH;
      - any vector-memory instruction issued by the COMPILER (buffer_/global_/scratch_/flat_ outside the asm
        blocks: a spill, a store, its own loads) is reported as a FOREIGN op.  It also counts on vmcnt, but only in
        the safe direction: `vmcnt(N)` leaves at most N operations outstanding, loads complete in order among
        themselves, so extra operations in the queue can only force MORE of the asm loads to have landed than the
        hand count intended (an over-wait: performance, not correctness).  Foreign ops are therefore counted and
        printed, not failed -- a scratch_ one (a spill inside the pipeline) is worth a look;
  * control flow is followed, not text order: basic blocks, both successors of every conditional branch, loops
    re-entered until the set of in-flight FIFOs seen at each block entry stops growing -- so a load issued at the loop
    tail and consumed at the loop head is seen, and a block that the layout puts ahead of its predecessors is not
    mis-read;
  * nothing expires by distance.

  hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -Iinclude -S --cuda-device-only <src>.hip -o /tmp/k.s
  python tools/audit_asm_loads.py <mangled-kernel-name-prefix> /tmp/k.s        # expect: touches 0, foreign 0
"""
import re
import sys

VMEM = ("buffer_", "global_", "scratch_", "flat_")


def _regs(tok):
    out = set()
    for m in re.finditer(r'\b([va])\[(\d+):(\d+)\]', tok):
        for r in range(int(m.group(2)), int(m.group(3)) + 1):
            out.add(m.group(1) + str(r))
    for m in re.finditer(r'\b([va])(\d+)\b', tok):
        out.add(m.group(1) + m.group(2))
    return out


def _blocks(body):
    """Basic blocks of the kernel text: [(first line, last line + 1)], label -> block index."""
    is_label = [bool(re.match(r'\s*\.LBB\d+_\d+:', l)) for l in body]
    is_branch = [l.strip().split()[0] in ('s_branch', 's_endpgm') or l.strip().startswith('s_cbranch') if l.strip() else False
                 for l in body]
    starts = {0}
    for k in range(len(body)):
        if is_label[k]:
            starts.add(k)
        if is_branch[k] and k + 1 < len(body):
            starts.add(k + 1)
    order = sorted(starts)
    blocks = [(lo, hi) for lo, hi in zip(order, order[1:] + [len(body)])]
    label_of = {}
    for i, (lo, hi) in enumerate(blocks):
        m = re.match(r'\s*(\.LBB\d+_\d+):', body[lo])
        if m:
            label_of[m.group(1)] = i
    return blocks, label_of


def _run_block(body, lo, hi, state, found, verbose):
    """Simulates body[lo:hi] from the in-flight FIFO `state` (tuple of (frozenset regs, load line)); records touches
    and foreign ops in `found`; returns (FIFO at the end, successor kind, branch target or None)."""
    pending = list(state)
    inasm = False
    kind, target = 'fall', None
    for k in range(lo, hi):
        t = body[k].strip()
        if t.startswith(';;#ASMSTART'):
            inasm = True
            continue
        if t.startswith(';;#ASMEND'):
            inasm = False
            continue
        if not t or t.startswith(';') or t.startswith('.'):
            continue
        op = t.split()[0]
        if inasm and op.startswith('buffer_load'):
            pending.append((frozenset(_regs(t.split(',')[0])), k))
            del pending[:max(0, len(pending) - 63)]
            found['loads'].add(k)
            continue
        m = re.search(r's_waitcnt\b.*vmcnt\((\d+)\)', t)
        if m:                                   # hand-written or compiler-inserted: all but the N youngest have landed
            del pending[:max(0, len(pending) - int(m.group(1)))]
            continue
        if op == 's_endpgm':
            return tuple(pending), 'end', None
        if op == 's_branch':
            return tuple(pending), 'jump', t.split()[-1]
        if op.startswith('s_cbranch'):
            kind, target = 'cond', t.split()[-1]
            continue
        if op.startswith('s_'):
            continue
        if not inasm and op.startswith(VMEM):
            # a compiler-issued vector-memory op sits in the same in-order queue: it takes one of the N slots a later
            # vmcnt(N) leaves outstanding (the stem forward kernel's hand count includes its 16 output stores per row)
            if any(d for d, _ in pending):
                if k not in found['foreign']:
                    found['foreign'].add(k)
                    if verbose and len(found['foreign']) <= 8:
                        print('FOREIGN', k, t[:90], ' with', sum(1 for d, _ in pending if d), 'asm loads in flight')
            pending.append((frozenset(), -1))        # anonymous: which store it was does not matter
            del pending[:max(0, len(pending) - 63)]      # the counter holds 6 bits: issue stalls at 63 outstanding
            used = _regs(t)
            for d, k0 in pending:
                if used & d and (k, k0) not in found['touch']:
                    found['touch'].add((k, k0))
            continue
        used = _regs(t)
        for d, k0 in pending:
            if used & d and (k, k0) not in found['touch']:
                found['touch'].add((k, k0))
                if verbose and len(found['touch']) <= 8:
                    print('TOUCH', k, t[:90], ' <- load at', k0)
    return tuple(pending), kind, target


def audit(fn_prefix, path, verbose=True, max_states=4096):
    """returns (asm buffer loads found, touches of in-flight destination registers, foreign vector-memory ops).
    Walks the control-flow graph (basic blocks, both successors of a conditional branch, loops until the set of
    in-flight FIFOs seen at each block entry stops growing)."""
    lines = open(path).read().split('\n')
    starts = [k for k, l in enumerate(lines) if l.startswith(fn_prefix)]
    if not starts:
        raise KeyError(f"{fn_prefix} not found in {path}")
    body = lines[starts[0]:]
    end = [k for k, l in enumerate(body) if 's_endpgm' in l][0]
    body = body[:end + 1]
    blocks, label_of = _blocks(body)
    found = {'loads': set(), 'touch': set(), 'foreign': set()}
    seen = [set() for _ in blocks]
    work = [(0, ())]
    while work:
        b, state = work.pop()
        if state in seen[b]:
            continue
        if len(seen[b]) >= max_states:
            raise RuntimeError(f"audit: more than {max_states} distinct in-flight states at block {b} (line {blocks[b][0]})")
        seen[b].add(state)
        out, kind, target = _run_block(body, blocks[b][0], blocks[b][1], state, found, verbose)
        if kind == 'end':
            continue
        if kind in ('jump', 'cond') and target in label_of:
            work.append((label_of[target], out))
        if kind in ('fall', 'cond') and b + 1 < len(blocks):
            work.append((b + 1, out))
    found_touches = sorted(found['touch'])
    audit.last_touches = found_touches               # [(touching line, load line)] for callers that want distances
    return len(found['loads']), len(found_touches), len(found['foreign'])


if __name__ == "__main__":
    n, b, f = audit(sys.argv[1], sys.argv[2])
    print(sys.argv[1][-40:], 'asm loads', n, '; touches of in-flight destinations:', b, '; foreign vector-memory ops (over-wait only):', f)
