# timing ablations of the PS block chain (tools/probe/ps_dbg_build.sh fdet_chain_x3 CH_DBG 1 2 3 4 5)
echo "full: $(python tools/probe/chain_time.py 2>/dev/null | tail -1)"
for n in 1 2 3 4 5; do echo "CH_DBG=$n: $(FDET_LIB_PATH=$PWD/pytorch-face-detection-from-scratch_amd/lib/dbg/libfdet_chain_x3_dbg$n.so python tools/probe/chain_time.py 2>/dev/null | tail -1)"; done
