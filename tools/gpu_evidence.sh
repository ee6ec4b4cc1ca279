#!/bin/bash
# The LAST GPU action of a round: evidence from HEAD, mechanically.  Runs on the GPU box (gpurun):
#   kernel-trace stats + PMC counter passes of the four workloads bench.py reports -> gpurun_out/cnt_*; afterwards, on the build
#   machine:  tools/gpu_evidence.sh collect  turns them into profiles/r03_counters.json (stamped with the SHA-256 of the device
#   sources) and copies the kernel-stats CSVs / the default bench line into profiles/r03_*.
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
cd "$R"
collect_counters() {
  python3 tools/collect_counters.py "atrium256=atrium 1920x1080 256spp tile64 depth0 packets" "atrium_d8=atrium 1920x1080 64spp tile64 depth8 packets" \
      "teapot256=teapot.obj 1920x1080 256spp tile64 depth0 packets" "teapot_d8=teapot.obj 1920x1080 256spp tile64 depth8 packets" > /dev/null
}
if [ "$1" = "collect" ]; then
  collect_counters
  for t in atrium256 atrium_d8 teapot256 teapot_d8; do
    cp gpurun_out/cnt_${t}_kernel_stats.csv profiles/r03_${t}_kernel_stats.csv
    cp gpurun_out/cnt_${t}.txt profiles/r03_${t}_counters.txt
  done
  cp gpurun_out/bench_r03_final.json profiles/r03_bench.json
  tools/isa_meta.sh > profiles/r03_isa_meta.txt
  echo "profiles/r03_* written"; exit 0
fi
bash tools/gpu_counters.sh atrium256 render_tiles_packet_kernel
bash tools/gpu_counters.sh atrium_d8 render_paths --spp 64 --depth 8
bash tools/gpu_counters.sh teapot256 render_tiles_packet_kernel --scene teapot
bash tools/gpu_counters.sh teapot_d8 render_paths --scene teapot --depth 8
collect_counters   # on the box too: the default bench line below then carries the counter-derived figures of THIS build
python3 bench.py > gpurun_out/bench_r03_final.log 2>&1
grep '^{' gpurun_out/bench_r03_final.log > gpurun_out/bench_r03_final.json
cut -c1-600 gpurun_out/bench_r03_final.json
