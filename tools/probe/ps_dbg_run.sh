for n in ${PS_DBG_LIST:-1 2 3 4 5 6}; do echo "== PS_DBG=$n"; FDET_LIB_PATH=$PWD/pytorch-face-detection-from-scratch_amd/lib/dbg/libfdet_ps_dbg$n.so python tools/probe/ps_conv_time.py 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d.items(): print(k, {a:b['median_ms'] for a,b in v.items() if a.endswith('_ps')})
"; done
