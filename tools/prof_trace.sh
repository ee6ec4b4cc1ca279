#!/bin/bash
# PMC profile of ONE trace_rays_kernel dispatch on bounce-like rays (tools/prof_trace.py): each counter set in its own rocprofv3
# run (never combined with tracing), the last matching dispatch of each run is reported.  usage: tools/prof_trace.sh TAG [n_rays]
TAG=${1:-trace}; N=${2:-4000000}; R=$PWD; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
: > $R/gpurun_out/prof_${TAG}.txt
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "GRBM_GUI_ACTIVE TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_TAG_STALL_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/prof_${TAG}_$i -- python3 $R/tools/prof_trace.py $N > $R/gpurun_out/prof_${TAG}_$i.log 2>&1 || echo "set $i failed" >> $R/gpurun_out/prof_${TAG}.txt
  python3 - $R/gpurun_out/prof_${TAG}_$i >> $R/gpurun_out/prof_${TAG}.txt <<'PY'
import csv, glob, sys, collections
rows = []
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    rows += [r for r in csv.DictReader(open(f)) if "trace_rays_kernel" in r["Kernel_Name"]]
if rows:
    last = max(int(r["Dispatch_Id"]) for r in rows)
    for r in rows:
        if int(r["Dispatch_Id"]) == last:
            print(f'{r["Counter_Name"]:36s} {float(r["Counter_Value"]):.6g}')
PY
  rm -rf $R/gpurun_out/prof_${TAG}_$i
done
cat $R/gpurun_out/prof_${TAG}.txt
