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
  python tools/audit_vmcnt.py <mangled-kernel-name-prefix> k.s [stores]
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
            ev[k] = ("op", "store" if "_store" in op or "_atomic" in op else "load")
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


def dma_bypass_edges(ev, blocks, preds):
    """Forward edges (P -> B, B later in the layout than P + 1) whose skipped blocks hold DMA pieces: `if (this wave moves
    a piece) { pieces }`.  A wave that takes such an edge has issued no piece there, so nothing it waits for starts at that
    point -- and in the precision16 kernels the lo-plane waves skip EVERY piece: followed backwards, such paths run on to the
    kernel entry and collect counts that belong to no DMA.  The walk ignores these edges."""
    has_dma = [any(ev[k] and ev[k][0] == "dma" for k in range(lo, hi)) for lo, hi in blocks]
    of_label = {ev[lo][1]: i for i, (lo, hi) in enumerate(blocks) if ev[lo] and ev[lo][0] == "label"}

    def term(i):
        e = ev[blocks[i][1] - 1]
        return e if e and e[0] in ("jump", "cond", "end") else None
    out = {}
    for b, ps in enumerate(preds):
        for p in ps:
            # a simple if-then: P branches to B, the (few) blocks in between hold the pieces and only lead on to B
            if not (p + 1 < b <= p + 5 and any(has_dma[p + 1:b])):
                continue
            tp = term(p)
            if tp is None or tp[0] != "cond" or of_label.get(tp[1]) != b:
                continue
            inner_ok = True
            for i in range(p + 1, b):
                ti = term(i)
                if ti is not None and (ti[0] == "end" or not (p < of_label.get(ti[1], -1) <= b)):
                    inner_ok = False
            if inner_ok:
                out.setdefault(b, set()).add(p)
    return out


def path_counts(ev, blocks, preds, line, cap=200, stores_only=False, bypass=None):
    """Counts of vector-memory ops (stores_only: of stores) between the nearest DMA piece and `line`, over every backward
    path."""
    blk = next(i for i, (lo, hi) in enumerate(blocks) if lo <= line < hi)
    bypass = bypass or {}
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
                if e[0] == "op" and (not stores_only or e[1] == "store"):
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
            if p in bypass.get(b, ()):
                continue                                     # an edge that jumps over DMA pieces: that wave issued none
            if (p, c) not in seen:
                seen.add((p, c))
                work.append((p, blocks[p][1] - 1, c))
    return found


def audit(fn_prefix, path, stores_only=False):
    """-> ([(line, K, sorted path counts)] for every hand-written vmcnt(K) with K > 0, number of DMA pieces found).
    stores_only: count what the SOURCE counts when its K is a number of stores (the fused pooled forward: its epilogue's
    loads all precede its stores and are not in K -- they only add slack, operations retire in order)."""
    lines = open(path).read().split("\n")
    body = _kernel_body(lines, fn_prefix)
    ev = _events(body)
    blocks, preds = _cfg(ev)
    ndma = sum(1 for e in ev if e and e[0] == "dma")
    bypass = dma_bypass_edges(ev, blocks, preds)
    out = []
    for k, e in enumerate(ev):
        if e and e[0] == "wait" and e[2] and e[1] > 0:
            counts = path_counts(ev, blocks, preds, k, stores_only=stores_only, bypass=bypass)
            out.append((k, e[1], sorted(c for c in counts if isinstance(c, int))))
    return out, ndma


def check(res, rule_a=True):
    """The two rules (see the module text).  D = the run lengths the source dispatches on = {0} + every K of the kernel's
    hand waits, closed under addition (a path that skips a conditional DMA group adds two runs up).  For every wait(K):
      B  some path reaches it with >= K operations behind the nearest DMA piece (the run the count stands for exists);
      A  every SHORTER run that can reach it is itself a member of D (another branch of the same dispatch takes it) --
         a store that disappears, doubles or merges produces a run length outside D.
    rule_a=False keeps rule B only: for a kernel whose stores hipcc duplicates under complementary EXEC masks (`off = ok ?
    address : out-of-range` compiled as two predicated stores with an `s_cbranch_execz` around each), a walk that cannot
    pair the two masks sees every subset of them as a path.
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
        short = [c for c in S if c < K and c not in D] if rule_a else []
        if short:
            bad.append((line, K, f"runs {short[:8]} are shorter than {K} and not a count the source dispatches on {sorted(base)}"))
    return bad


if __name__ == "__main__":
    res, ndma = audit(sys.argv[1], sys.argv[2], stores_only=len(sys.argv) > 3 and sys.argv[3] == "stores")
    bad = check(res)
    print(f"{sys.argv[1][-48:]}: {ndma} DMA pieces, {len(res)} hand-counted waits, {len(bad)} violations")
    for line, K, S in res:
        print(f"  line {line}: vmcnt({K})  runs reaching it: {S[:16]}{' ...' if len(S) > 16 else ''}")
    for b in bad:
        print("  VIOLATION", b)
    sys.exit(0 if not bad and ndma else 1)
