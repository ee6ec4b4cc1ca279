#!/bin/bash
# usage: tools/gpu_pmc_cmd.sh TAG KERNEL_SUBSTR script.py [args...]  -- counters of one kernel of an arbitrary python script
TAG=${1:-run}; K=${2:-render_}; shift 2; R=$PWD; cd /tmp; export TMPDIR=/tmp
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE"; do
  n=$(echo $set | cut -d" " -f1)
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmcc_${TAG}_$n -- python3 $R/"$@" > $R/gpurun_out/pmcc_${TAG}_$n.log 2>&1 || echo "set $n failed"
  python3 $R/profiles/pmc_summary.py $R/gpurun_out/pmcc_${TAG}_$n $K
done
