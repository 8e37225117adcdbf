#!/bin/bash
# Ablation builds of the PS kernels (timing only, wrong results): lib/dbg/libfdet_<tag>.so with one source recompiled
# with -D<MACRO>=<N>.   usage: ps_dbg_build.sh <source stem> <MACRO> <N> [<N> ...]
#   fdet_conv3x3_ps PS_DBG: 1 no epilogue, 2 no MFMAs, 4 no DMA, 8 activations from image 0, 16 no weight DMA
#   fdet_wgrad3x3_ps WG_DBG: 1 no DMA in the band loop, 2 no MFMAs, 4 no fragment reads
# Run with FDET_LIB_PATH=<so> python tools/probe/ps_conv_time.py
set -e
cd "$(dirname "$0")/../../pytorch-face-detection-from-scratch_amd/csrc"
mkdir -p ../lib/dbg
stem=$1; macro=$2; shift 2
for n in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-result -ffp-contract=off -I../../include -D$macro=$n -c $stem.hip -o ../lib/dbg/${stem}_$n.o
  objs=$(ls ../lib/obj/*.o | grep -v "/$stem.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/dbg/lib${stem}_dbg$n.so $objs ../lib/dbg/${stem}_$n.o
done
