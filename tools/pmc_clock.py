#!/usr/bin/env python3
"""Joins rocprofv3 counter_collection + kernel_trace: per kernel mean duration, effective clock
(GRBM_GUI_ACTIVE / 8 / duration) and MFMA pipe utilisation (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * cycles))."""
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
    print(f"{k:70s} dur={dur:9.1f} us  clock={clk:5.2f} GHz  mfma_util={util:5.2f}")
