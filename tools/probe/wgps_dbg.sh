for n in 1 2 3 4 5 6; do echo "WG_DBG=$n"; FDET_LIB_PATH=$PWD/pytorch-face-detection-from-scratch_amd/lib/dbg/libfdet_wgrad3x3_ps_dbg$n.so timeout -k 10 300 python tools/probe/ps_conv_time.py 2>/dev/null | grep WGRAD | python -c "
import json,sys
d=json.loads(sys.stdin.read()[6:])
print({k:v['wgrad_ps']['median_ms'] for k,v in d.items()})"; done
