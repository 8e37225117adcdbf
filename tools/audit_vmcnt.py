#!/usr/bin/env python3
"""Audit of the HAND-COUNTED `s_waitcnt vmcnt(K)` of the kernels that stage operands with LDS-DMA issued from inline asm
(fdet_conv3x3_ps.hip, fdet_wgrad3x3_ps.hip, fdet_chain_x3.hip; fdet_ldsdma.h).

Those kernels open an LDS buffer with `s_waitcnt vmcnt(K)` + `s_barrier`, where K is the number of vector-memory
operations the wave has issued AFTER the DMA pieces it waits for (operations retire in order, so "at most K outstanding"
means "the pieces have landed").  K is a hand count of compiler-issued instructions -- the epilogue's buffer stores, the
prefetch loads of the next epilogue -- and hipcc knows nothing of it: if it ever emitted FEWER such operations than
counted (two stores merged, a dead load dropped, a loop not unrolled), the wait would pass with pieces still in flight and
stale LDS would be read silently.

The scan works on the device assembly of one kernel (cross-compiled, no GPU needed):

  * DMA piece      = an asm `buffer_load ... lds`;
  * counted op     = any other vector-memory instruction (buffer_/global_/scratch_/flat_), compiler- or asm-issued;
  * hand wait      = an `s_waitcnt ... vmcnt(K)` between `;;#ASMSTART` and `;;#ASMEND`.

For every hand wait with K > 0 it walks the control-flow graph BACKWARDS from the wait (every predecessor, loops until
no new (block, count) pair appears) to the nearest DMA piece on each path and collects the number of counted ops in
between (loads count too: operations retire in order, so ANY K operations behind a piece make `vmcnt(K)` sufficient).
Which of several waits a wave takes is decided at run time from a counter the kernel keeps next to its stores
(`young`), so the walk cannot know which paths are feasible; check() therefore proves two things per wait: a run of at
least K operations reaches it, and every shorter run that can reach it has a length the source dispatches on (another
branch of the same `young` chain) -- a store that disappears, doubles or merges changes a run length and fails the audit.

  hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -Iinclude [-DPS_TU=n] -S --cuda-device-only src.hip -o k.s
  python tools/audit_vmcnt.py <mangled-kernel-name-prefix> k.s
"""
import re
import sys

VMEM = ("buffer_", "global_", "scratch_", "flat_")


def _kernel_body(lines, fn_prefix):
    starts = [k for k, l in enumerate(lines) if l.startswith(fn_prefix) and l.rstrip().split(";")[0].strip().endswith(":")]
    if not starts:
        raise KeyError(f"{fn_prefix} not found")
    st = starts[0]
    en = next(k for k in range(st, len(lines)) if lines[k].startswith(".Lfunc_end"))
    return lines[st:en]


def _events(body):
    """Per line: None or ('dma',) / ('op',) / ('wait', K, hand) / ('label', name) / ('jump', target) / ('cond', target) /
    ('end',)."""
    ev = [None] * len(body)
    inasm = False
    for k, raw in enumerate(body):
        t = raw.strip()
        if t.startswith(";;#ASMSTART"):
            inasm = True
            continue
        if t.startswith(";;#ASMEND"):
            inasm = False
            continue
        m = re.match(r"(\.LBB\d+_\d+):", t)
        if m:
            ev[k] = ("label", m.group(1))
            continue
        if not t or t.startswith(";") or t.startswith("."):
            continue
        op = t.split()[0]
        if op.startswith("buffer_load") and re.search(r"\blds\b", t.split(";")[0]):
            ev[k] = ("dma",)
        elif op.startswith(VMEM):
            ev[k] = ("op",)
        elif op == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", t)
            if m:
                ev[k] = ("wait", int(m.group(1)), inasm)
        elif op == "s_endpgm":
            ev[k] = ("end",)
        elif op == "s_branch":
            ev[k] = ("jump", t.split()[1])
        elif op.startswith("s_cbranch"):
            ev[k] = ("cond", t.split()[1])
    return ev


def _cfg(ev):
    """Basic blocks [(lo, hi)], predecessor lists."""
    n = len(ev)
    starts = {0}
    for k, e in enumerate(ev):
        if e and e[0] == "label":
            starts.add(k)
        if e and e[0] in ("jump", "cond", "end") and k + 1 < n:
            starts.add(k + 1)
    order = sorted(starts)
    blocks = list(zip(order, order[1:] + [n]))
    of_label = {}
    for i, (lo, hi) in enumerate(blocks):
        if ev[lo] and ev[lo][0] == "label":
            of_label[ev[lo][1]] = i
    preds = [[] for _ in blocks]
    for i, (lo, hi) in enumerate(blocks):
        last = next((ev[k] for k in range(hi - 1, lo - 1, -1) if ev[k] and ev[k][0] in ("jump", "cond", "end")), None)
        # a terminator is the last instruction of its block by construction
        term = ev[hi - 1] if ev[hi - 1] and ev[hi - 1][0] in ("jump", "cond", "end") else None
        if term is None or term[0] == "cond":
            if i + 1 < len(blocks):
                preds[i + 1].append(i)
        if term is not None and term[0] in ("jump", "cond") and term[1] in of_label:
            preds[of_label[term[1]]].append(i)
    return blocks, preds


def path_counts(ev, blocks, preds, line, cap=200):
    """Counts of vector-memory ops between the nearest DMA piece and `line`, over every backward path."""
    blk = next(i for i, (lo, hi) in enumerate(blocks) if lo <= line < hi)
    found = set()
    seen = set()
    work = [(blk, line - 1, 0)]
    while work:
        b, k, c = work.pop()
        lo = blocks[b][0]
        hit = False
        while k >= lo:
            e = ev[k]
            if e:
                if e[0] == "dma":
                    found.add(c)
                    hit = True
                    break
                if e[0] == "op":
                    c += 1
                    if c > cap:
                        found.add(c)
                        hit = True
                        break
            k -= 1
        if hit:
            continue
        if not preds[b]:
            found.add(("entry", c))                      # reached the kernel entry without meeting a DMA piece
            continue
        for p in preds[b]:
            if (p, c) not in seen:
                seen.add((p, c))
                work.append((p, blocks[p][1] - 1, c))
    return found


def audit(fn_prefix, path):
    """-> ([(line, K, sorted path counts)] for every hand-written vmcnt(K) with K > 0, number of DMA pieces found)."""
    lines = open(path).read().split("\n")
    body = _kernel_body(lines, fn_prefix)
    ev = _events(body)
    blocks, preds = _cfg(ev)
    ndma = sum(1 for e in ev if e and e[0] == "dma")
    out = []
    for k, e in enumerate(ev):
        if e and e[0] == "wait" and e[2] and e[1] > 0:
            counts = path_counts(ev, blocks, preds, k)
            out.append((k, e[1], sorted(c for c in counts if isinstance(c, int))))
    return out, ndma


def check(res):
    """The two rules (see the module text).  D = the run lengths the source dispatches on = {0} + every K of the kernel's
    hand waits, closed under addition (a path that skips a conditional DMA group adds two runs up).  For every wait(K):
      B  some path reaches it with >= K operations behind the nearest DMA piece (the run the count stands for exists);
      A  every SHORTER run that can reach it is itself a member of D (another branch of the same dispatch takes it) --
         a store that disappears, doubles or merges produces a run length outside D.
    -> list of (line, K, reason) violations."""
    base = {0} | {K for _, K, _ in res}
    D = set(base)
    grew = True
    while grew:
        grew = False
        for a in list(D):
            for b in base:
                if a + b <= 256 and a + b not in D:
                    D.add(a + b)
                    grew = True
    bad = []
    for line, K, S in res:
        if not any(c >= K for c in S):
            bad.append((line, K, f"no path carries {K} operations behind the DMA (runs: {S[:12]})"))
        short = [c for c in S if c < K and c not in D]
        if short:
            bad.append((line, K, f"runs {short[:8]} are shorter than {K} and not a count the source dispatches on {sorted(base)}"))
    return bad


if __name__ == "__main__":
    res, ndma = audit(sys.argv[1], sys.argv[2])
    bad = check(res)
    print(f"{sys.argv[1][-48:]}: {ndma} DMA pieces, {len(res)} hand-counted waits, {len(bad)} violations")
    for line, K, S in res:
        print(f"  line {line}: vmcnt({K})  runs reaching it: {S[:16]}{' ...' if len(S) > 16 else ''}")
    for b in bad:
        print("  VIOLATION", b)
    sys.exit(0 if not bad and ndma else 1)
