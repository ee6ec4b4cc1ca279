#!/bin/bash
# usage: tools/gpu_pmc.sh TAG  -- SQ instruction-mix and wait counters for one bench step (rocprofv3 --pmc, separate passes)
TAG=${1:-run}; R=$PWD; cd /tmp; export TMPDIR=/tmp
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM" "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS"; do
  n=$(echo $set | cut -d" " -f1)
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_${TAG}_$n -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmc_${TAG}_$n.log 2>&1 || echo "set $n failed"
  python3 $R/profiles/pmc_summary.py $R/gpurun_out/pmc_${TAG}_$n
done
