#!/usr/bin/env python3
"""SQ ratios per kernel from two rocprofv3 --pmc passes (tools/collect_profiles.sh: s1, s2):
   wait_any = SQ_WAIT_ANY / SQ_WAVE_CYCLES (parked at s_waitcnt / barrier), wait_inst = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES
   (issue stalled), active = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES, mfma = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_WAVE_CYCLES)
   (matrix-pipe utilisation at one wave per SIMD, half of it at two), lds_conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE.
   usage: pmc_sq_ratios.py <dir1> <prefix1> <dir2> <prefix2>"""
import csv, sys, collections, re
d1, p1, d2, p2 = sys.argv[1:5]

def load(d, p):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f"{d}/{p}_counter_collection.csv")):
        nm = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        nm = re.sub(r"^void ", "", nm).split("(")[0][:60]
        acc[nm][r["Counter_Name"]] += float(r["Counter_Value"])
    return acc
A, B = load(d1, p1), load(d2, p2)
rows = []
for k in A:
    a, b = A[k], B.get(k, {})
    wc = a.get("SQ_WAVE_CYCLES", 0)
    if wc <= 0: continue
    wc2 = b.get("SQ_WAVE_CYCLES", 0) or 1
    rows.append((wc, f"{k:62s} wait_any={a.get('SQ_WAIT_ANY',0)/wc:.2f} wait_inst={a.get('SQ_WAIT_INST_ANY',0)/wc:.2f} "
                     f"active={a.get('SQ_ACTIVE_INST_ANY',0)/wc:.2f} mfma={b.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/(4*wc2):.2f} "
                     f"lds_conflict={(b.get('SQ_LDS_BANK_CONFLICT',0)/b['SQ_LDS_IDX_ACTIVE']) if b.get('SQ_LDS_IDX_ACTIVE') else 0:.3f}"))
for _, line in sorted(rows, reverse=True)[:24]:
    print(line)
