#!/usr/bin/env python3
"""Builds profiles/r03_counters.json from the gpurun_out/cnt_TAG.txt files tools/gpu_counters.sh wrote.

usage: collect_counters.py TAG=WORKLOAD_KEY ...   (WORKLOAD_KEY as bench.py's workload_key(), e.g.
       "atrium 1920x1080 256spp tile64 depth0 packets")

hbm_bytes_per_launch = FETCH_SIZE x 2 + WRITE_SIZE (KB -> bytes): MI355X_MICROARCH.md "HBM": on gfx950 FETCH_SIZE tallies
128-byte requests at 64 bytes (double it), WRITE_SIZE reads exactly; both from their own --pmc pass, one launch each."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # DEVICE_SOURCES / device_sources_sha256: the counters are evidence for exactly these sources

out_path = os.path.join(ROOT, "profiles", "r03_counters.json")
try:
    out = json.load(open(out_path))
except Exception:
    out = {}
for arg in sys.argv[1:]:
    tag, key = arg.split("=", 1)
    c = {}
    for line in open(os.path.join(ROOT, "gpurun_out", f"cnt_{tag}.txt")):
        p = line.split()
        if len(p) >= 2:
            c[p[0]] = float(p[1])
            c[p[0] + "_rows"] = int(p[2].strip("(")) if len(p) > 2 else 1
    e = {"source": f"profiles/r03_counters.json <- tools/gpu_counters.sh {tag} (rocprofv3 --pmc, one launch per pass)"}
    for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM", "SQ_INSTS_LDS", "SQ_WAVES", "SQ_ACTIVE_INST_VALU",
              "SQ_THREAD_CYCLES_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES",
              "TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum", "TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum", "FETCH_SIZE", "WRITE_SIZE"):
        if k in c:
            e[k] = c[k] / max(c.get(k + "_rows", 1), 1)  # per launch
    if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
        e["hbm_bytes_per_launch"] = int((e["FETCH_SIZE"] * 2 + e["WRITE_SIZE"]) * 1024)
    if e.get("SQ_ACTIVE_INST_VALU") and e.get("SQ_THREAD_CYCLES_VALU"):
        e["valu_lane_utilisation"] = e["SQ_THREAD_CYCLES_VALU"] / (64 * e["SQ_ACTIVE_INST_VALU"])
    if e.get("TCC_HIT_sum") is not None and e.get("TCC_MISS_sum") is not None and (e["TCC_HIT_sum"] + e["TCC_MISS_sum"]) > 0:
        e["tcc_hit_rate"] = e["TCC_HIT_sum"] / (e["TCC_HIT_sum"] + e["TCC_MISS_sum"])
    out[key] = e
sha = bench.device_sources_sha256()
if out.get("_meta", {}).get("device_sources_sha256") not in (None, sha):
    # counters of other sources must not survive next to fresh ones: keep only what this call collected
    out = {k: v for k, v in out.items() if k in [a.split("=", 1)[1] for a in sys.argv[1:]]}
out["_meta"] = {"device_sources_sha256": sha, "device_sources": bench.DEVICE_SOURCES,
                "note": "bench.py uses these counters only while the SHA-256 over device_sources matches the tree's"}
json.dump(out, open(out_path, "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
