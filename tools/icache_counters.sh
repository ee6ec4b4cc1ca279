#!/bin/bash
# Instruction-fetch counters of one bench.py launch (is a kernel's code size costing it?).  usage: tools/icache_counters.sh KERNEL_SUBSTR [bench args]
K=${1:-render_tiles_packet}; shift; R=$PWD; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_DATA_READ_REQ SQC_TC_STALL"; do
  n=$(echo $set | cut -d" " -f1)
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/ic_$n -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extension "$@" > $R/gpurun_out/ic_$n.log 2>&1 || echo "set $n failed"
  python3 $R/profiles/pmc_summary.py $R/gpurun_out/ic_$n $K
  rm -rf $R/gpurun_out/ic_$n
done
