#!/bin/bash
# Diagnostic: build libpdecnn_stamp.so (N=32 kernels with cycle stamps in one wave of the backward).
set -e
cd "$(dirname "$0")/../cnn-with-pde_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fno-slp-vectorize --offload-arch=gfx950 -I../../include -I. -DPDE_WAVES=8 -DPDE_STAMP=1 -DPDE_INST_N=32 -c pde_adi_inst.hip -o ../lib/obj/stamp_32.o
objs=$(ls ../lib/obj/*.o | grep -v "abl_" | grep -v "stamp_" | grep -v "inst_32.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs ../lib/obj/stamp_32.o -o ../lib/libpdecnn_stamp.so
