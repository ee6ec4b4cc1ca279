#!/bin/bash
# usage: tools/gpu_profile.sh TAG : bench + rocprofv3 kernel-trace stats + HBM traffic PMC (separate passes) for profiles/
TAG=${1:-r01}; R=$PWD; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
python3 $R/bench.py --steps 10 --warmup 2 --check > $R/gpurun_out/${TAG}_bench.log 2>&1; grep '^{' $R/gpurun_out/${TAG}_bench.log > $R/gpurun_out/${TAG}_bench.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_trace -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $R/gpurun_out/${TAG}_bench_rocprof.log 2>&1
grep '^{' $R/gpurun_out/${TAG}_bench_rocprof.log > $R/gpurun_out/${TAG}_bench_under_rocprof.json
cp $R/gpurun_out/${TAG}_trace/*/*_kernel_stats.csv $R/gpurun_out/${TAG}_kernel_stats.csv
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  n=$(echo $c | cut -d" " -f1)
  timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/${TAG}_pmc_$n -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/${TAG}_pmc_$n.log 2>&1 || echo "pmc $n failed"
  python3 $R/profiles/pmc_summary.py $R/gpurun_out/${TAG}_pmc_$n render_tiles | tee -a $R/gpurun_out/${TAG}_traffic_counters.txt
done
cat $R/gpurun_out/${TAG}_kernel_stats.csv | head -4
cat $R/gpurun_out/${TAG}_bench.json | cut -c1-300
