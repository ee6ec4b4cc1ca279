#!/bin/bash
# usage: tools/gpu_counters.sh TAG KERNEL_SUBSTR [bench.py args...]
# One workload = one TAG: kernel-trace stats of a short bench run, then rocprofv3 PMC passes (each counter set in its own run of
# `bench.py --steps 1 --warmup 0`, --pmc never combined with tracing) for the kernel whose name contains KERNEL_SUBSTR.
# Results: gpurun_out/cnt_TAG_*.txt (summed per counter), gpurun_out/cnt_TAG_kernel_stats.csv.  tools/collect_counters.py turns
# them into profiles/r03_counters.json, which bench.py reads for roofline.traffic / valu_issue_frac.
TAG=${1:-run}; K=${2:-render_}; shift 2; R=$PWD; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/cnt_${TAG}_trace -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extension "$@" > $R/gpurun_out/cnt_${TAG}_bench.log 2>&1 || echo "trace run failed"
cp $R/gpurun_out/cnt_${TAG}_trace/*/*_kernel_stats.csv $R/gpurun_out/cnt_${TAG}_kernel_stats.csv 2>/dev/null
grep '^{' $R/gpurun_out/cnt_${TAG}_bench.log > $R/gpurun_out/cnt_${TAG}_bench.json
: > $R/gpurun_out/cnt_${TAG}.txt
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  n=$(echo $set | cut -d" " -f1)
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/cnt_${TAG}_$n -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extension "$@" > $R/gpurun_out/cnt_${TAG}_$n.log 2>&1 || echo "set $n failed"
  python3 $R/profiles/pmc_summary.py $R/gpurun_out/cnt_${TAG}_$n $K >> $R/gpurun_out/cnt_${TAG}.txt
  rm -rf $R/gpurun_out/cnt_${TAG}_$n
done
rm -rf $R/gpurun_out/cnt_${TAG}_trace
echo "== $TAG"; cat $R/gpurun_out/cnt_${TAG}.txt; head -4 $R/gpurun_out/cnt_${TAG}_kernel_stats.csv; cut -c1-400 $R/gpurun_out/cnt_${TAG}_bench.json
