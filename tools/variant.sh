#!/bin/bash
# Diagnostic: build cnn-with-pde_amd/lib/libpdecnn_<tag>.so — the N = 32 sweep kernels and pde_adi.hip compiled with extra
# -D flags (PDE_ABL, PDE_WAVES, PDE_JF, PDE_JB ...), every other object from the stock build.  Select it with PDECNN_LIB.
# usage: tools/variant.sh <tag> [flags...]
set -e
tag=$1; shift
cd "$(dirname "$0")/../cnn-with-pde_amd/csrc"
# (run `make` in csrc first: the stock objects are linked in)
mkdir -p ../lib/obj/var
BASE="-O3 -std=c++17 -fPIC -fno-slp-vectorize --offload-arch=gfx950 -I../../include -I. -DPDE_PACK=0 -DPDE_SKEW=1 -DPDE_PRIO=1"
case " $* " in *PDE_WAVES*) ;; *) BASE="$BASE -DPDE_WAVES=8";; esac
/opt/rocm/bin/hipcc $BASE "$@" -DPDE_INST_N=32 -c pde_adi_inst.hip -o ../lib/obj/var/${tag}_inst32.o &
/opt/rocm/bin/hipcc $BASE "$@" -c pde_adi.hip -o ../lib/obj/var/${tag}_adi.o &
wait
objs=$(ls ../lib/obj/*.o | grep -v "inst_32.o" | grep -v "/pde_adi.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs ../lib/obj/var/${tag}_inst32.o ../lib/obj/var/${tag}_adi.o -o ../lib/libpdecnn_${tag}.so
echo built ../lib/libpdecnn_${tag}.so
