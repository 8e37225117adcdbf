#!/bin/bash
# Ablation builds of the PS conv kernel (timing only, wrong results): lib/dbg/libfdet_ps_dbg<N>.so with -DPS_DBG=N
# (1 = no epilogue, 2 = no MFMAs, 4 = no DMA; sums combine).  Use: FDET_LIB_PATH=<so> python tools/probe/ps_conv_time.py
set -e
cd "$(dirname "$0")/../../pytorch-face-detection-from-scratch_amd/csrc"
mkdir -p ../lib/dbg
for n in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-result -ffp-contract=off -I../../include -DPS_DBG=$n -c fdet_conv3x3_ps.hip -o ../lib/dbg/ps_dbg$n.o
  objs=$(ls ../lib/obj/*.o | grep -v fdet_conv3x3_ps.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/dbg/libfdet_ps_dbg$n.so $objs ../lib/dbg/ps_dbg$n.o
done
