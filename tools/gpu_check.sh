#!/bin/bash
# Runs on the GPU box: parity suite, then a short bench with the oracle spot-check.  usage: tools/gpu_check.sh [tag] [bench args...]
TAG=${1:-run}; shift
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_$TAG.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_$TAG.log
tail -4 gpurun_out/pytest_$TAG.log
timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --check "$@" > gpurun_out/bench_$TAG.log 2>&1
grep '^{' gpurun_out/bench_$TAG.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('Mrays/s', round(d['value'],1), 'ms', round(d['ms_per_step'],2), 'kernel_ms', round(d['roofline']['kernel_ms'],2), 'mismatch', d.get('check_mismatches'))" || tail -5 gpurun_out/bench_$TAG.log
