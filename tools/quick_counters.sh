#!/bin/bash
# Two quick PMC passes (instruction mix, waits) of one bench.py launch.  usage: tools/quick_counters.sh KERNEL_SUBSTR [bench args]
K=${1:-render_tiles_packet}; shift; R=$PWD; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_BRANCH"; do
  n=$(echo $set | cut -d" " -f1)
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/qc_$n -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extension "$@" > $R/gpurun_out/qc_$n.log 2>&1 || echo "set $n failed"
  python3 $R/profiles/pmc_summary.py $R/gpurun_out/qc_$n $K
  rm -rf $R/gpurun_out/qc_$n
done
