#!/bin/bash
# Diagnostic: registers / spills / occupancy hipcc reports for the N = 32 fp32 Strang sweep kernels under extra flags.
# usage: tools/resource_usage.sh <tag> [flags...]       (objects go to cnn-with-pde_amd/lib/obj/var/)
tag=$1; shift
cd "$(dirname "$0")/../cnn-with-pde_amd/csrc"
mkdir -p ../lib/obj/var
BASE="-O3 -std=c++17 -fPIC -fno-slp-vectorize --offload-arch=gfx950 -I../../include -I. -DPDE_PACK=0 -DPDE_SKEW=1 -DPDE_PRIO=1 -DPDE_INST_N=32"
case " $* " in *PDE_WAVES*) ;; *) BASE="$BASE -DPDE_WAVES=8";; esac
/opt/rocm/bin/hipcc $BASE "$@" -Rpass-analysis=kernel-resource-usage -c pde_adi_inst.hip -o ../lib/obj/var/ru_$tag.o 2>&1 \
  | grep -E "Function Name|VGPRs:|VGPRs Spill|ScratchSize|Occupancy" | paste - - - - - \
  | grep -E "adi_(bwd|fwd)_kernelILi32ELi[0-9]EfLi1E" | sed -E 's/[^ ]*remark: *//g; s/\[-Rpass[^]]*\]//g; s/ +/ /g' | sed "s/^/$tag: /"
