#!/bin/bash
TAG=${1:-run}; R=$PWD; cd /tmp; export TMPDIR=/tmp
for set in "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_IFETCH SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64" "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_CVT SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU2"; do
  n=$(echo $set | cut -d" " -f1)
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc2_${TAG}_$n -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmc2_${TAG}_$n.log 2>&1 || echo "set $n failed"
  python3 $R/profiles/pmc_summary.py $R/gpurun_out/pmc2_${TAG}_$n
done
