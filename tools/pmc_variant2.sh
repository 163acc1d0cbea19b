#!/bin/bash
# one SQ counter pass (given list) of the bench workload with a variant build: tools/pmc_variant2.sh <tag> <pass-name> "<counters>"
tag=$1; pass=$2; ctr=$3
export TMPDIR=/tmp
export PDECNN_LIB=$PWD/cnn-with-pde_amd/lib/libpdecnn_${tag}.so
rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d gpurun_out/pmcv_${tag}_${pass} -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-configs > gpurun_out/pmcv_${tag}_${pass}.log 2>&1; echo "pmc $tag $pass exit=$?"
python - <<PY
import csv,collections,glob
for f in glob.glob("gpurun_out/pmcv_${tag}_${pass}/**/*counter_collection.csv", recursive=True):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "adi_bwd" in k or "adi_fwd" in k:
            acc[(k.split("<")[0].split("::")[-1], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k,v in sorted(acc.items()):
        print("${tag}", k[0], k[1], "%.3f M"%(sum(v)/len(v)/1e6))
PY
