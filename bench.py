#!/usr/bin/env python3
"""bench.py — DoG + argmax frames/s on synthetic window batches (BASELINE.json metric).

A "step" is one pass of the hot path (pdog_detect_batch: fused separable DoG + argmax kernel,
then the tiny strip-combine/clamp kernel) over one batch of windows whose frames are already
resident in HBM, plus — for N > 1 — the gather of the int32 positions to rank 0 (the only
collective on the path).  Default workload = BASELINE.json configs[2]: 1080p frames,
256x256 search windows (-> 257x257 outputs), batch 4096 per GPU, target_width 25.

One JSON line on stdout (rank 0).  `roofline` prices the fused kernel against HBM as the
contract asks (algorithmic bytes = input tile u8 + 8 B result per window) and also
reports the FP32-VALU fraction, which is the bound that actually binds.
`cpu_baseline` times the oracle's dense Float64 statement of the reference algorithm
(kind "port": Julia is not available) on a bounded sample on all host cores, threaded inside a
window (the reference's CPUThreads model) and across windows (value = the faster), plus the
oracle's separable Float64 statement for the algorithmic-vs-hardware split.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (frame_h, frame_w, target_width, window_size, batch_per_gpu, description)
    "cfg1": (240, 320, 25, 45, 100, "240x320, default 45x45 window, 100 frames, tw=25"),
    "cfg2": (1080, 1920, 25, (270, 480), 64, "1080p auto-detect window sz.÷4 = 271x481, tw=25"),
    "cfg3": (1080, 1920, 25, 256, 4096, "1080p, 256x256 windows (257x257 outputs), batch 4096, tw=25"),
    "cfg4": (2160, 3840, 25, 512, 1024, "4K, 512x512 windows, 1024 frames per GPU (8192 over 8), tw=25"),
    "cfg5": (1080, 1920, 120, 205, 4096, "1080p, tw=120 (l=293), default 205x205 window, batch 4096"),
}
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_PEAK_FMA = 78.6e12        # 157.3 TFLOP/s FP32 vector = 78.6 T FMA/s (v_pk_fma_f32)


def make_frames(torch, n, h, w, tw, radii, seed, noise, device):
    """Synthetic frames on the device (recipe of test/test-basic-test.jl:65-68, raw u8):
    background 128, one dark disc of radius tw÷2 per frame, optional ±noise levels."""
    import numpy as np
    rng = np.random.Generator(np.random.PCG64(seed))
    ci = rng.integers(1, h + 1, n)
    cj = rng.integers(1, w + 1, n)
    di = rng.integers(-(radii[0] // 2), radii[0] // 2 + 1, n)
    dj = rng.integers(-(radii[1] // 2), radii[1] // 2 + 1, n)
    guesses = np.stack([np.clip(ci + di, 1, h), np.clip(cj + dj, 1, w)], 1).astype(np.int32)
    centres = np.stack([ci, cj], 1).astype(np.int32)
    frames = torch.full((n, h, w), 128, dtype=torch.uint8, device=device)
    rad = int(tw) // 2
    yy, xx = torch.meshgrid(torch.arange(-rad, rad + 1, device=device), torch.arange(-rad, rad + 1, device=device), indexing="ij")
    disc = (yy * yy + xx * xx) <= rad * rad
    for b in range(n):
        i, j = int(ci[b]) - 1, int(cj[b]) - 1
        i0, i1, j0, j1 = max(0, i - rad), min(h - 1, i + rad), max(0, j - rad), min(w - 1, j + rad)
        m = disc[i0 - i + rad:i1 - i + rad + 1, j0 - j + rad:j1 - j + rad + 1]
        frames[b, i0:i1 + 1, j0:j1 + 1].masked_fill_(m, 0)
    if noise:
        gen = torch.Generator(device=device)
        gen.manual_seed(seed + 1)
        step = max(1, (256 << 20) // (h * w))
        for b0 in range(0, n, step):
            blk = frames[b0:b0 + step]
            nz = torch.randint(-noise, noise + 1, blk.shape, dtype=torch.int16, device=device, generator=gen)
            blk.copy_((blk.to(torch.int16) + nz).clamp_(0, 255).to(torch.uint8))
    return frames, guesses, centres


def stored_traffic(workload, variant, batch):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/traffic_r01.json);
    None when no counter run exists for this workload/variant/batch (counters cannot be read from inside a
    timed run: rocprofv3 wraps the process)."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic_r01.json")) as f:
            rec = json.load(f).get(workload)
        if not rec or rec["variant"] != variant or rec["batch"] != batch:
            return None
        return int((rec["fetch_size_kib"] * rec["fetch_correction"] + rec["write_size_kib"]) * 1024)
    except (OSError, KeyError, ValueError):
        return None


def cpu_quota():
    """CPUs this process may use at once under its cgroup (None when unlimited/unknown): the box exposes all host
    cores to OpenMP but may schedule only a share of them."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else round(int(q) / int(per), 2)
    except (OSError, ValueError):
        return None


def cpu_baseline(frames_host, guesses_host, fill, tw, radii, budget_s=8.0):
    """The oracle's dense Float64 correlation + first-max argmax (oracle/dog_oracle.c, -Ofast build for timing)
    on a bounded sample, threaded two ways: inside one window, as the reference's CPUThreads call does
    (src/PawsomeTracker.jl:57), and across windows (one window per core), which is the stronger baseline for
    a batch.  The separable Float64 statement is timed across windows on the same sample to separate the
    algorithmic gain (rank-2 separable, what the GPU kernels compute) from the hardware gain.
    Returns a dict; positions of every mode are checked against each other."""
    import numpy as np
    from oracle.dog_oracle import Oracle, build
    build()
    o = Oracle(fast=True)
    strict = Oracle(fast=False)
    sig = strict.sigma(tw)
    K = strict.dog_kernel(sig, True)
    cores = o.max_threads()
    n_avail = len(frames_host)

    def timed_chunks(fn, chunk):
        done, outs = 0, []
        t0 = time.perf_counter()
        while done < n_avail:
            outs.append(fn(done, min(n_avail, done + chunk)))
            done = min(n_avail, done + chunk)
            if time.perf_counter() - t0 > budget_s:
                break
        return done / (time.perf_counter() - t0), done, np.concatenate(outs, 0)

    o.detect(frames_host[0], fill, K, radii, guesses_host[0])          # warm threads/caches
    within, n_w, pos_w = timed_chunks(
        lambda lo, hi: np.array([o.detect(frames_host[b], fill, K, radii, guesses_host[b]) for b in range(lo, hi)], np.int32), 4)
    across, n_a, pos_a = timed_chunks(
        lambda lo, hi: o.detect_batch_par(frames_host[lo:hi], fill, K, sig, True, radii, guesses_host[lo:hi], False), cores)
    sep, n_s, pos_s = timed_chunks(
        lambda lo, hi: o.detect_batch_par(frames_host[lo:hi], fill, K, sig, True, radii, guesses_host[lo:hi], True), cores)
    m = min(n_w, n_a)
    assert np.array_equal(pos_w[:m], pos_a[:m]) and np.array_equal(pos_s[:min(n_s, n_a)], pos_a[:min(n_s, n_a)]), \
        "oracle: threading modes / separable statement disagree on the sample"
    pos = pos_a if n_a >= n_w else pos_w
    return {"value": max(within, across), "cores": cores, "pos": pos,
            "within": (within, n_w), "across": (across, n_a), "separable": (sep, n_s)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="override windows per GPU")
    ap.add_argument("--noise", type=int, default=3, help="± uniform noise levels on the synthetic frames")
    ap.add_argument("--variant", type=int, default=-1, help="force a kernel specialisation")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--data-rank", type=int, default=-1, help="generate the synthetic data of this rank (checks the per-rank seeds on one GPU)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    import pawsometracker_jl_amd as pt

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus or world == 1 and args.gpus == 1, "launch with torch.distributed.run for --gpus > 1"
    # PDOG_BENCH_BACKEND=gloo rehearses the N > 1 control flow where RCCL cannot run (several ranks on ONE GPU of a
    # development box): ranks then share devices round-robin and the gather goes through host memory
    backend = os.environ.get("PDOG_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    fh, fw, tw, ws, batch, desc = WORKLOADS[args.workload]
    if args.batch:
        batch = args.batch
    ws = pt.fix_window_size(ws if not isinstance(ws, tuple) else (ws[1], ws[0]))   # (w,h) -> (h,w), :70
    radii = (ws[0] // 2, ws[1] // 2)
    fill = 128
    frames, guesses_h, centres = make_frames(torch, batch, fh, fw, tw, radii, seed=1000 * (args.data_rank if args.data_rank >= 0 else rank),
                                             noise=args.noise, device=dev)
    if args.noise:
        fill = pt.mode(frames[0].cpu().numpy())                 # mode of the first frame, :47
    guesses = torch.from_numpy(guesses_h).to(dev)
    bt = pt.BatchTracker(fh, fw, tw, ws, True, fill, device=dev_index)
    if args.variant >= 0:
        bt.set_variant(args.variant)
    bt.reserve(batch)
    bt.use_torch_stream()
    info = bt.info()
    out = torch.empty((batch, 2), dtype=torch.int32, device=dev)
    n_total = batch * world

    def step():
        bt.detect(frames, guesses, out=out)
        if world > 1:
            return pt.gather_positions(out, n_total)
        return out

    for _ in range(args.warmup):
        step()
    if world > 1:
        pt.gather_positions(out, n_total)   # RCCL sets up its point-to-point channels on first use: not part of any timed step
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    pending = []
    for k in range(args.steps):
        ev[k][0].record()
        bt.detect(frames, guesses, out=out)
        ev[k][1].record()
        if world > 1:   # the 8 B/window gather of step k runs beside step k+1's kernels; all of them finish inside the timed region
            pending.append(pt.gather_positions(out, n_total, async_op=True))
    for h in pending:
        gathered = h.wait()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))   # HIP events on the launch stream

    got = out.cpu().numpy()
    # sanity inside the bench: a disc fully inside frame+window must be found within 1 px of its
    # centre (exactly, when noise-free) — a wrong-but-fast kernel must not produce a number
    rad = tw // 2 + 1
    inside = ((centres[:, 0] > rad) & (centres[:, 0] <= fh - rad) & (centres[:, 1] > rad) & (centres[:, 1] <= fw - rad)
              & (np.abs(centres - guesses_h) <= np.array(radii) - rad).all(1))
    err = np.abs(got[inside] - centres[inside]).max() if inside.any() else 0
    if not os.environ.get("PDOG_BENCH_NOCHECK"):   # timing-only ablation builds produce wrong positions
        assert err <= (1 if args.noise else 0), f"tracking sanity failed: max |pos - centre| = {err}"

    if rank == 0:
        value = n_total * args.steps / dt
        abytes = int(info.algorithmic_bytes_per_window)
        afma = int(info.algorithmic_fma_per_window)
        ach_gbs = abytes * batch / (kern_ms * 1e-3) / 1e9
        fma_rate = afma * batch / (kern_ms * 1e-3)
        res = {
            "metric": "DoG+argmax frames/s, 1080p 256x256 windows batch 4096; % HBM roofline @1/8 GPU",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {desc}", "frame": [fh, fw], "window": [info.win_h, info.win_w],
                       "batch_per_gpu": batch, "target_width": tw, "kernel_len": info.kernel_len,
                       "noise_levels": args.noise, "variant": info.variant, "kernel_for_this_batch": bt.kernel_for_batch(batch),
                       "strips": info.n_strips,
                       "sharding": (f"frames x{world}, gather int32[n,2] to rank 0" + ("" if backend == "nccl" else f" ({backend} rehearsal)"))
                                   if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "achieved": ach_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach_gbs / HBM_PEAK_GBS, "traffic": stored_traffic(args.workload, info.variant, batch),
                         "kernel_ms": kern_ms, "algorithmic_bytes_per_window": abytes,
                         # what `frac` would be with the FP32 vector ALUs 100 % busy on algorithmic FMAs: the ceiling of this path
                         "frac_at_fp32_vector_peak": abytes / (afma / VALU_PEAK_FMA) / 1e9 / HBM_PEAK_GBS,
                         "valu": {"achieved_fma_per_s": fma_rate, "peak_fma_per_s": VALU_PEAK_FMA,
                                  "frac": fma_rate / VALU_PEAK_FMA, "algorithmic_fma_per_window": afma},
                         "note": "path is FP32-VALU bound (375 flop/B vs ridge 19.7, DESIGN.md); traffic = FETCH_SIZE x2 (gfx950 "
                                 "16 B/lane correction) + WRITE_SIZE from profiles/traffic_r01.json, bytes per launch; the chip "
                                 "sustains ~2.0 GHz under this kernel, valu.peak is quoted at the nominal 2.4 GHz"},
        }
        if world == 1 and not args.no_cpu:
            ns = min(256, batch)
            cb = cpu_baseline(frames[:ns].cpu().numpy(), guesses_h[:ns], fill, tw, radii)
            cpos = cb["pos"]
            assert np.array_equal(cpos, got[:len(cpos)]), "GPU positions differ from the CPU oracle on the sample"
            L = info.kernel_len
            res["cpu_baseline"] = {
                "value": cb["value"], "unit": "frames/s", "cores": cb["cores"], "cgroup_cpu_quota": cpu_quota(), "kind": "port",
                "sample": f"first {cb['across'][1]} windows of the same batch (time-bounded), dense {L}x{L} Float64 correlation + "
                          "first-max argmax (oracle/dog_oracle.c, -Ofast, OpenMP); value = the faster of the two threadings; "
                          f"positions equal to the GPU's on all {len(cpos)} sampled windows",
                "threads_across_windows": {"value": cb["across"][0], "windows": cb["across"][1],
                                           "note": "one window per core, each window single-threaded"},
                "threads_within_window": {"value": cb["within"][0], "windows": cb["within"][1],
                                          "note": "how one reference Tracker call threads (CPUThreads splits the window)"},
                "separable_f64": {"value": cb["separable"][0], "windows": cb["separable"][1], "unit": "frames/s",
                                  "note": "threads across windows; the oracle's separable Float64 statement "
                                          "(the arithmetic the GPU kernels do, in double): algorithmic vs hardware gain"}}
        print(json.dumps(res))
    bt.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
