#!/bin/bash
# Runs on the GPU box: parity suite, then a short bench with the oracle spot-check.  usage: tools/gpu_check.sh [tag] [bench args...]
TAG=${1:-run}; shift
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_$TAG.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_$TAG.log
tail -6 gpurun_out/pytest_$TAG.log
timeout -k 10 400 python bench.py --steps 5 --warmup 1 --check "$@" > gpurun_out/bench_$TAG.log 2>&1
grep '^{' gpurun_out/bench_$TAG.log > gpurun_out/bench_$TAG.json
python - <<PY || tail -5 gpurun_out/bench_$TAG.log
import json
d = json.load(open("gpurun_out/bench_$TAG.json"))
print("Mrays/s", round(d["value"], 1), "ms", round(d["ms_per_step"], 2), "kernel_ms", round(d["roofline"]["kernel_ms"], 2), "mismatch", d.get("check_mismatches"))
for k in ("paths_depth8",):
    if k in d: print(k, round(d[k]["value"], 1), "Mrays/s", round(d[k]["ms_per_step"], 1), "ms")
if "teapot_c2" in d:
    for k, v in d["teapot_c2"].items(): print("teapot", k, round(v["value"], 1), "Mrays/s", round(v["ms_per_step"], 2), "ms")
if "cpu_baseline" in d: print("cpu", round(d["cpu_baseline"]["value"], 2), d["cpu_baseline"]["cores"], "threads")
PY
