#!/bin/bash
TAG=${1:-run}; shift; R=$PWD; cd /tmp; export TMPDIR=/tmp
for set in "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE SQC_DCACHE_BUSY_CYCLES SQC_TC_DATA_READ_REQ SQC_TC_STALL SQC_DCACHE_INPUT_VALID_READYB" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_ICACHE_BUSY_CYCLES SQC_TC_INST_REQ SQ_INST_CYCLES_SMEM SQ_INST_LEVEL_SMEM" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_BUSY_CYCLES"; do
  n=$(echo $set | cut -d" " -f1)
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc3_${TAG}_$n -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > $R/gpurun_out/pmc3_${TAG}_$n.log 2>&1 || echo "set $n failed"
  python3 $R/profiles/pmc_summary.py $R/gpurun_out/pmc3_${TAG}_$n
done
