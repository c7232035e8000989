#!/usr/bin/env python3
"""Turn the FETCH_SIZE / WRITE_SIZE passes of tools/prof.sh into the profiles/traffic_rNN.json entry bench.py reads.
usage: traffic_from_pmc.py <workload> <variant> <batch> <steps_in_pmc_run> <fetch_pmc_dir> <write_pmc_dir>  → JSON on stdout
Every pdog kernel of a step is listed with its per-launch bytes and launches per step; FETCH_SIZE is doubled for
kernels that stream with 16 B/lane loads (MI355X_MICROARCH.md §HBM: gfx950 counts those at half), left as counted for
the others (uncalibrated widths: dword / byte loads)."""
import collections
import csv
import glob
import json
import os
import sys

csv.field_size_limit(1 << 30)
WIDE = ("dog_roll_kernel", "dog_chain_kernel")   # global_load_dwordx4 staging


def counters(d, name):
    out = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name and "pdog::" in r["Kernel_Name"]:
                out[r["Kernel_Name"].split("(")[0].replace("void pdog::", "")].append(float(r["Counter_Value"]))
    return out


def main():
    wl, variant, batch, steps, fdir, wdir = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], sys.argv[6]
    fetch, write = counters(fdir, "FETCH_SIZE"), counters(wdir, "WRITE_SIZE")
    kernels = {}
    for k, vals in fetch.items():
        kernels[k] = {"fetch_size_kib": sum(vals) / len(vals), "write_size_kib": (sum(write[k]) / len(write[k])) if write.get(k) else 0.0,
                      "fetch_correction": 2.0 if k.startswith(WIDE) else 1.0, "launches_per_step": len(vals) / steps}
    print(json.dumps({wl: {"variant": variant, "batch": batch, "kernels": kernels}}, indent=1))


if __name__ == "__main__":
    main()
