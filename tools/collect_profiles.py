#!/usr/bin/env python3
"""Copies one round's tools/prof.sh outputs from gpurun_out/ (scratch) into profiles/ (tracked) and merges the per-config
traffic files into profiles/traffic_<round>.json, which bench.py reads for `roofline.traffic`.
usage: tools/collect_profiles.py r03 cfg1 cfg1chain:cfg1_chain cfg2 cfg3 cfg4 cfg5 tw44 tw52   (tag[:traffic key])"""
import json, os, shutil, sys
rnd, tags = sys.argv[1], sys.argv[2:]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles")
tfile = os.path.join(dst, f"traffic_{rnd}.json")
traffic = json.load(open(tfile)) if os.path.exists(tfile) else {}
for spec in tags:
    tag, _, key = spec.partition(":")
    out = tag.replace("cfg1chain", "cfg1_chain")
    for a, b in ((f"prof_{rnd}_{tag}.txt", f"{rnd}_{out}_rocprofv3_summary.txt"), (f"kernel_stats_{rnd}_{tag}.csv", f"{rnd}_{out}_kernel_stats.csv"),
                 (f"prof_{rnd}_{tag}_bench.json", f"{rnd}_{out}_bench_under_rocprof.json"),
                 (f"prof_{rnd}_{tag}_bench_unprofiled.json", f"{rnd}_{out}_bench.json")):
        shutil.copyfile(os.path.join(src, a), os.path.join(dst, b))
    t = json.load(open(os.path.join(src, f"traffic_{rnd}_{tag}.json")))
    (wl, val), = t.items()
    traffic[key or (wl if tag.startswith("cfg") else tag)] = val
json.dump(traffic, open(tfile, "w"), indent=1)
print("merged", sorted(k for k in traffic if k != "_doc"))
