#!/bin/bash
# SQ counter passes of the bench workload with a variant build: tools/pmc_variant.sh <tag>
tag=$1
export TMPDIR=/tmp
export PDECNN_LIB=$PWD/cnn-with-pde_amd/lib/libpdecnn_${tag}.so
run() { rocprofv3 --kernel-trace --pmc $2 --output-format csv -d gpurun_out/pmcv_${tag}_$1 -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-configs > gpurun_out/pmcv_${tag}_$1.log 2>&1; echo "pmc $tag $1 exit=$?"; }
run sq1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY"
run sq2 "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS"
python - <<PY
import csv,collections,glob
for p in ("sq1","sq2"):
    for f in glob.glob("gpurun_out/pmcv_${tag}_%s/**/*counter_collection.csv"%p, recursive=True):
        acc=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"]
            if "adi_bwd" in k or "adi_fwd" in k:
                acc[(k.split("<")[0].split("::")[-1], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for k,v in sorted(acc.items()):
            print("${tag}", k[0], k[1], "%.1f M"%(sum(v)/len(v)/1e6))
PY
