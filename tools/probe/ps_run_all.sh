#!/bin/bash
# tests + timing + ablation of the PS kernels (GPU box)
timeout -k 10 300 python -m pytest tests/test_gpu_ps.py -x -q 2>&1 | tail -8 || exit 1
timeout -k 10 300 python tools/probe/ps_conv_time.py > gpurun_out/ps_time.json 2>&1
python - <<PY
import json
t=open("gpurun_out/ps_time.json").read(); d=json.loads(t[t.index("{"):t.index("WGRAD")] if "WGRAD" in t else t[t.index("{"):])
for k,v in d.items(): print(k,{a:b["median_ms"] for a,b in v.items()})
if "WGRAD" in t:
    w=json.loads(t[t.index("WGRAD")+6:])
    for k,v in w.items(): print("wgrad",k,{a:b["median_ms"] for a,b in v.items()})
PY
PS_DBG_LIST="${PS_DBG_LIST:-1 5}" bash tools/probe/ps_dbg_run.sh > gpurun_out/ps_dbg.txt 2>&1; cat gpurun_out/ps_dbg.txt
