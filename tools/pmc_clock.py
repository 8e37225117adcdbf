#!/usr/bin/env python3
"""Joins rocprofv3 counter_collection + kernel_trace: per kernel mean duration, effective clock
(GRBM_GUI_ACTIVE / 8 / duration) and MFMA pipe utilisation (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * cycles)).
GRBM_GUI_ACTIVE also counts the cycles the graphics block is busy before / after the kernel's own timestamps (dispatch,
cache write-back), a fixed cost of a few microseconds: for launches under 100 us it inflated the quotient beyond the
2.4 GHz maximum, so the clock column is printed only from 100 us up ("n/a" below); the MFMA
utilisation is a ratio of two cycle counts and does not depend on the timestamps."""
import csv, sys, collections
d = sys.argv[1]; pre = sys.argv[2]
tr = {r["Dispatch_Id"]: r for r in csv.DictReader(open(f"{d}/{pre}_kernel_trace.csv"))}
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f"{d}/{pre}_counter_collection.csv")):
    t = tr.get(r["Dispatch_Id"])
    if t is None: continue
    dur = (int(t["End_Timestamp"]) - int(t["Start_Timestamp"])) / 1e3
    k = r["Kernel_Name"][:70]
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    agg[k]["_dur_us"].append(dur)
for k, v in agg.items():
    dur = sum(v["_dur_us"]) / len(v["_dur_us"])
    if dur < 20: continue
    gui = sum(v.get("GRBM_GUI_ACTIVE", [0])) / max(1, len(v.get("GRBM_GUI_ACTIVE", [0])))
    mf = sum(v.get("SQ_VALU_MFMA_BUSY_CYCLES", [0])) / max(1, len(v.get("SQ_VALU_MFMA_BUSY_CYCLES", [0])))
    clk = gui / 8 / dur / 1e3 if gui else 0           # GHz
    util = mf / (1024 * gui / 8) if gui else 0
    clk_s = f"{clk:5.2f} GHz" if dur >= 100 else "  n/a    "
    print(f"{k:70s} dur={dur:9.1f} us  clock={clk_s}  mfma_util={util:5.2f}")
