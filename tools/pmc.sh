#!/bin/bash
# usage: tools/pmc.sh <outdir> "<counter list>" -- runs the bench under rocprofv3 --pmc (kernel-trace only)
out=$1; shift
ctr=$1; shift
export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $out.log 2>&1
echo "pmc $ctr exit=$?"
