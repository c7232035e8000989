#!/usr/bin/env python3
"""Per-kernel launch durations from a rocprofv3 --kernel-trace CSV, WITHOUT the warm-up dispatches.
usage: kernel_times.py <rocprof-output-dir> <dispatches_per_kernel_to_skip>  → CSV on stdout
rocprofv3's own --stats averages every dispatch of a run, the warm-up launches included (cold instruction cache, clocks
still ramping): round 2's committed average (1.655 ms) exceeded the bench line's ms_per_step (1.589 ms) for that reason.
Columns: kernel, calls (after skipping), mean, median, min, max in ns."""
import collections
import csv
import glob
import os
import statistics
import sys

csv.field_size_limit(1 << 30)
d, skip = sys.argv[1], int(sys.argv[2])
rows = collections.defaultdict(list)
for f in glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
print("Name,Calls,MeanNs,MedianNs,MinNs,MaxNs,SkippedWarmupCalls")
out = []
for k, v in rows.items():
    v.sort()
    dur = [x[1] for x in v[skip:]] if len(v) > skip else [x[1] for x in v]
    out.append((sum(dur), k, dur, min(skip, len(v) - len(dur) if len(v) > skip else 0)))
for _, k, dur, sk in sorted(out, reverse=True):
    print('"%s",%d,%.1f,%.1f,%d,%d,%d' % (k.replace('"', "'"), len(dur), statistics.mean(dur), statistics.median(dur), min(dur), max(dur), skip if len(dur) else 0))
