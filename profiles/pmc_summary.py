#!/usr/bin/env python3
"""Sums rocprofv3 --pmc counter_collection.csv rows per kernel and counter.  usage: pmc_summary.py DIR [kernel-substr]"""
import csv, glob, sys, collections
d = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else "render_tiles"
acc = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in sorted(acc):
    print(f"{k:28s} {acc[k]:.6g}  ({n[k]} rows)")
