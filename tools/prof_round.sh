#!/bin/bash
# Kernel-trace statistics quoted in profiles/README.md (one rocprofv3 --kernel-trace --stats run per workload):
#   tools/prof_round.sh <tag>  -> gpurun_out/prof_<tag>_{bench,legs,rh,anysize}/ ; copy the *_kernel_stats.csv into profiles/
tag=${1:-run}
export TMPDIR=/tmp
run() { name=$1; shift; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_${name} -- "$@" > gpurun_out/prof_${tag}_${name}.log 2>&1; echo "prof $name exit=$?"; }
run bench python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-configs
run legs python bench.py --steps 20 --warmup 5 --no-cpu-baseline
run rh python tools/perf_rh.py 128
run anysize python tools/perf_anysize.py
