#!/bin/bash
# usage: tools/gpu_paths.sh  -- path-extension parity tests + depth-8 throughput on atrium (16 spp) and teapot (64 spp)
python -m pytest tests -m gpu -x -q -k "path or sharded or trace" 2>&1 | tail -2
python bench.py --scene atrium --spp 16 --depth 8 --steps 2 --warmup 1 --no-cpu-baseline --no-extension | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('atrium d8 Mrays/s', round(d['value']), 'ms', round(d['ms_per_step'],2))"
python bench.py --spp 64 --depth 8 --steps 2 --warmup 1 --no-cpu-baseline --no-extension | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('teapot d8 Mrays/s', round(d['value']), 'ms', round(d['ms_per_step'],2))"
