#!/bin/bash
# All PMC passes of the bench workload (one counter group per run, kernel-trace only; never together with other traces):
#   tools/pmc_all.sh <tag>      -> gpurun_out/pmc_<tag>_{fetch,write,sq1,sq2}/ ; summarise with tools/pmc_summary.py
tag=${1:-run}
export TMPDIR=/tmp
run() { rocprofv3 --kernel-trace --pmc $2 --output-format csv -d gpurun_out/pmc_${tag}_$1 -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-configs > gpurun_out/pmc_${tag}_$1.log 2>&1; echo "pmc $1 exit=$?"; }
run fetch "FETCH_SIZE"
run write "WRITE_SIZE"
run sq1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY"
run sq2 "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
