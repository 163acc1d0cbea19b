#!/bin/bash
# All counter passes quoted by bench.py / DESIGN.md, one counter group per run (kernel-trace only, never with other traces):
#   tools/pmc_round.sh <tag>   -> gpurun_out/pmc_<tag>_{fetch,write,sq1,sq2}  (bench workload: the sweep kernels)
#                                 gpurun_out/pmc_<tag>_tiny_{fetch,write}      (cfg5: the explicit 5-point kernels)
#                                 gpurun_out/pmc_<tag>_{secondary,cfg4}_{fetch,write}  (layers with a channel operator)
# then, in the repository (git available):  python tools/pmc_to_json.py <tag>
tag=${1:-run}
export TMPDIR=/tmp
bench() { rocprofv3 --kernel-trace --pmc $2 --output-format csv -d gpurun_out/pmc_${tag}_$1 -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-configs > gpurun_out/pmc_${tag}_$1.log 2>&1; echo "pmc $1 exit=$?"; }
tiny() { rocprofv3 --kernel-trace --pmc $2 --output-format csv -d gpurun_out/pmc_${tag}_tiny_$1 -- python tools/prof_tiny.py > gpurun_out/pmc_${tag}_tiny_$1.log 2>&1; echo "pmc tiny $1 exit=$?"; }
bench fetch "FETCH_SIZE"
bench write "WRITE_SIZE"
bench sq1 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY"
bench sq2 "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
tiny fetch "FETCH_SIZE"
tiny write "WRITE_SIZE"
# the layers WITH a channel operator (round 4): cfg2 with mixing (tools/prof_secondary.py, 6 steps) and cfg4 (tools/prof_cfg4.py, 3 steps)
other() { rocprofv3 --kernel-trace --pmc $3 --output-format csv -d gpurun_out/pmc_${tag}_$1_$2 -- python tools/prof_$1.py > gpurun_out/pmc_${tag}_$1_$2.log 2>&1; echo "pmc $1 $2 exit=$?"; }
other secondary fetch "FETCH_SIZE"
other secondary write "WRITE_SIZE"
other cfg4 fetch "FETCH_SIZE"
other cfg4 write "WRITE_SIZE"
