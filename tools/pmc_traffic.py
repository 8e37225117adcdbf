#!/usr/bin/env python3
"""HBM traffic per launch from two rocprofv3 --pmc passes over bench.py (FETCH_SIZE, WRITE_SIZE).
usage: pmc_traffic.py <fetch_dir> <fetch_prefix> <write_dir> <write_prefix> <out.json>
FETCH_SIZE is doubled: on gfx950 it tallies 128-B requests of wide coalesced reads at 64 B
(MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact for 16-B-per-lane and dword-per-lane stores."""
import csv, json, sys, collections
fd, fp, wd, wp, out = sys.argv[1:6]

def load(d, p, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f"{d}/{p}_counter_collection.csv")):
        if r["Counter_Name"] != counter: continue
        import re
        nm = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        nm = re.sub(r"^void ", "", nm).split("(")[0]
        acc[(nm[:60], r["Grid_Size"])].append(float(r["Counter_Value"]))
    return acc
F = load(fd, fp, "FETCH_SIZE"); W = load(wd, wp, "WRITE_SIZE")
res = {}
for k in sorted(F, key=lambda k: -sum(F[k])):
    f = sum(F[k]) / len(F[k]); w = sum(W.get(k, [0])) / max(1, len(W.get(k, [0])))
    res[f"{k[0]} grid={k[1]}"] = {"launches": len(F[k]), "fetch_kb_raw": round(f), "write_kb": round(w),
                                   "hbm_bytes_per_launch": int((2 * f + w) * 1024)}
json.dump(res, open(out, "w"), indent=1)
for k, v in list(res.items())[:14]:
    print(k, v)
