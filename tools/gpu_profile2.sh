#!/bin/bash
# kernel-trace stats of the secondary workloads (atrium stand-in, path extension) for profiles/
R=$PWD; mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r01_atrium_trace -- python3 $R/bench.py --scene atrium --spp 64 --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r01_atrium_bench.log 2>&1
cp $R/gpurun_out/r01_atrium_trace/*/*_kernel_stats.csv $R/gpurun_out/r01_atrium_kernel_stats.csv
grep '^{' $R/gpurun_out/r01_atrium_bench.log > $R/gpurun_out/r01_atrium_bench.json
head -3 $R/gpurun_out/r01_atrium_kernel_stats.csv
python3 $R/bench.py --scene atrium --spp 64 --steps 3 --warmup 1 > $R/gpurun_out/r01_atrium_bench_full.json 2> /dev/null
cut -c1-200 $R/gpurun_out/r01_atrium_bench_full.json
