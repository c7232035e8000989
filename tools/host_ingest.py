#!/usr/bin/env python3
"""PCIe-inclusive throughput of the host-batch path (pdog_detect_batch_host): frames in pageable host memory,
tiles packed by host threads, chunked uploads overlapping the kernels.  Printed beside the two alternatives a
caller has: upload whole frames and run the device batch, or the one-window host functor per frame.
Never bench.py's `value` (that is HBM-resident); DESIGN.md quotes these numbers.

    python tools/host_ingest.py [--workload cfg3] [--batch 4096] [--reps 3]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg3")
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    import numpy as np
    import torch
    import bench
    import pawsometracker_jl_amd as pt

    fh, fw, tw, ws, _, desc = bench.WORKLOADS[args.workload]
    ws = pt.fix_window_size(ws if not isinstance(ws, tuple) else (ws[1], ws[0]))
    radii = (ws[0] // 2, ws[1] // 2)
    dev = torch.device("cuda", 0)
    frames_d, guesses, centres = bench.make_frames(torch, args.batch, fh, fw, tw, radii, seed=0, noise=3, device=dev)
    frames = frames_d.cpu().numpy()
    fill = pt.mode(frames[0])
    bt = pt.BatchTracker(fh, fw, tw, ws, True, fill)
    info = bt.info()
    ref = bt.detect(frames_d, torch.from_numpy(guesses).to(dev)).cpu().numpy()
    del frames_d
    torch.cuda.empty_cache()
    res = {"workload": f"{args.workload}: {desc}", "batch": args.batch, "tile_bytes": int(info.algorithmic_bytes_per_window) - 8,
           "frame_bytes": fh * fw, "host_threads": os.environ.get("PDOG_HOST_THREADS", "default min(16, cores)")}
    got = bt.detect_host(frames, guesses)                      # warm: allocates staging, pins, first touch
    assert np.array_equal(got, ref), "host-batch positions differ from the device batch"
    ts = []
    for _ in range(args.reps):
        t0 = time.perf_counter()
        got = bt.detect_host(frames, guesses)
        ts.append(time.perf_counter() - t0)
    res["host_batch"] = {"windows_per_s": args.batch / min(ts), "ms": min(ts) * 1e3, "all_ms": [t * 1e3 for t in ts],
                         "tile_GBps": args.batch * res["tile_bytes"] / min(ts) / 1e9}
    # alternative 1: whole frames over PCIe (pageable -> device), then the device batch
    g_d = torch.from_numpy(guesses).to(dev)
    ts = []
    for _ in range(max(1, args.reps - 1)):
        t0 = time.perf_counter()
        fd = torch.from_numpy(frames).to(dev)
        out = bt.detect(fd, g_d)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
        del fd
    res["whole_frame_upload"] = {"windows_per_s": args.batch / min(ts), "ms": min(ts) * 1e3,
                                 "frame_GBps": args.batch * fh * fw / min(ts) / 1e9}
    # alternative 2: the host functor, one window per call
    tr = pt.Tracker(frames[0], tw, ws, True)
    m = min(args.batch, 500)
    t0 = time.perf_counter()
    for b in range(m):
        tr.img.data = frames[b]
        tr((int(guesses[b, 0]), int(guesses[b, 1])))
    dt = time.perf_counter() - t0
    res["functor_per_call"] = {"windows_per_s": m / dt, "us_per_call": dt / m * 1e6}
    print(json.dumps(res))
    bt.close()


if __name__ == "__main__":
    main()
