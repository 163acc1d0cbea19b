#!/bin/bash
# Diagnostic: build cnn-with-pde_amd/lib/libpdecnn_<tag>.so with pde_rh.hip compiled under extra -D flags (PDE_RH_PF,
# PDE_RH_ROT), every other object from the stock build (run `make` in csrc first).  Select it with PDECNN_LIB.
# usage: tools/variant_rh.sh <tag> [flags...]
set -e
tag=$1; shift
cd "$(dirname "$0")/../cnn-with-pde_amd/csrc"
mkdir -p ../lib/obj/var
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fno-slp-vectorize --offload-arch=gfx950 -I../../include -I. "$@" -c pde_rh.hip -o ../lib/obj/var/${tag}_rh.o
objs=$(ls ../lib/obj/*.o | grep -v "/pde_rh.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs ../lib/obj/var/${tag}_rh.o -o ../lib/libpdecnn_${tag}.so
echo built ../lib/libpdecnn_${tag}.so
