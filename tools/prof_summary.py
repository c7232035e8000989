#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats + per-dispatch counters) into a small text summary.
usage: prof_summary.py <rocprof-output-dir> [...]  (prints to stdout)"""
import collections
import csv
import glob
import os
import sys

csv.field_size_limit(1 << 30)
for d in sys.argv[1:]:
    for f in sorted(glob.glob(os.path.join(d, "**", "*_kernel_stats.csv"), recursive=True)):
        print(f"== {f}")
        for i, row in enumerate(csv.reader(open(f))):
            if i < 8:
                print("  " + ",".join(c[:70] for c in row))
    for f in sorted(glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)):
        print(f"== {f}")
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        meta = {}
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:90]
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta[k] = {m: r.get(m) for m in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size", "Grid_Size")}
        for k, v in agg.items():
            print("  kernel:", k, meta[k])
            for c, vals in sorted(v.items()):
                print(f"     {c:28s} dispatches={len(vals):4d} mean={sum(vals) / len(vals):.6g}")
