#!/bin/bash
# usage: tools/gpu_pmc_any.sh TAG KERNEL_SUBSTR [bench.py args...]  -- instruction-mix / wait / memory counters of one kernel
TAG=${1:-run}; K=${2:-render_}; shift 2; R=$PWD; cd /tmp; export TMPDIR=/tmp
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_BRANCH SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  n=$(echo $set | cut -d" " -f1)
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmca_${TAG}_$n -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extension "$@" > $R/gpurun_out/pmca_${TAG}_$n.log 2>&1 || echo "set $n failed"
  python3 $R/profiles/pmc_summary.py $R/gpurun_out/pmca_${TAG}_$n $K
done
