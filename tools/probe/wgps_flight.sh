timeout -k 10 300 python -m pytest tests/test_gpu_ps.py -x -q -k wgrad 2>&1 | tail -3
for f in 1 2; do echo "FLIGHT=$f"; FDET_WGPS_FLIGHT=$f timeout -k 10 300 python tools/probe/ps_conv_time.py 2>/dev/null | grep WGRAD; done
