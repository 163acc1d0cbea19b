#!/bin/bash
# Build timing-only variants of the N=32 sweep kernels (PDE_ABL=1: no re-layouts, 2: no state/gradient
# work in the backward) into lib/libpdecnn_abl<k>.so.  Results are WRONG by construction; only bench
# timings are meaningful (PDECNN_LIB=... python bench.py --no-cpu-baseline --no-secondary).
set -e
cd "$(dirname "$0")/../cnn-with-pde_amd/csrc"
for k in ${ABLS:-1 2}; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fno-slp-vectorize --offload-arch=gfx950 -I../../include -I. -DPDE_WAVES=8 -DPDE_ABL=$k -DPDE_INST_N=32 -c pde_adi_inst.hip -o ../lib/obj/abl_$k.o
  objs=$(ls ../lib/obj/*.o | grep -v "abl_" | grep -v "inst_32.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs ../lib/obj/abl_$k.o -o ../lib/libpdecnn_abl$k.so
done
