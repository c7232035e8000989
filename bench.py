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
    """HBM bytes per STEP (every kernel a step launches, summed) from the committed PMC passes
    (profiles/traffic_r03.json, else traffic_r02.json; written by tools/traffic_from_pmc.py from separate FETCH_SIZE / WRITE_SIZE runs);
    None when no counter run exists for this workload/variant/batch (counters cannot be read from inside a timed
    run: rocprofv3 wraps the process)."""
    for name in ("traffic_r03.json", "traffic_r02.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                rec = json.load(f).get(workload)
            if not rec or rec["variant"] != variant or rec["batch"] != batch:
                continue
            total = 0.0
            for k in rec["kernels"].values():
                total += (k["fetch_size_kib"] * k["fetch_correction"] + k["write_size_kib"]) * k["launches_per_step"]
            return int(total * 1024)
        except (OSError, KeyError, ValueError, TypeError):
            continue
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
    # threads = what the box will actually schedule: the cgroup CPU quota when there is one (128 OpenMP threads
    # under a 16-CPU quota mostly measure oversubscription), else every host thread
    quota = cpu_quota()
    cores = max(1, min(o.max_threads(), int(quota))) if quota else o.max_threads()
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
        lambda lo, hi: np.array([o.detect(frames_host[b], fill, K, radii, guesses_host[b], nthreads=cores) for b in range(lo, hi)], np.int32), 4)
    across, n_a, pos_a = timed_chunks(
        lambda lo, hi: o.detect_batch_par(frames_host[lo:hi], fill, K, sig, True, radii, guesses_host[lo:hi], False, nthreads=cores), cores)
    sep, n_s, pos_s = timed_chunks(
        lambda lo, hi: o.detect_batch_par(frames_host[lo:hi], fill, K, sig, True, radii, guesses_host[lo:hi], True, nthreads=cores), cores)
    m = min(n_w, n_a)
    assert np.array_equal(pos_w[:m], pos_a[:m]) and np.array_equal(pos_s[:min(n_s, n_a)], pos_a[:min(n_s, n_a)]), \
        "oracle: threading modes / separable statement disagree on the sample"
    pos = pos_a if n_a >= n_w else pos_w
    return {"value": max(within, across), "cores": cores, "host_threads": o.max_threads(), "pos": pos,
            "within": (within, n_w), "across": (across, n_a), "separable": (sep, n_s)}


class native_stdout_to_stderr:
    """RCCL prints a version banner on the process's stdout when a communicator is created; the contract is ONE JSON
    line there.  While this is active, file descriptor 1 points at stderr (native code included)."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="override windows per GPU")
    ap.add_argument("--noise", type=int, default=3, help="± uniform noise levels on the synthetic frames")
    ap.add_argument("--variant", type=int, default=-1, help="force a kernel specialisation")
    ap.add_argument("--target-width", type=float, default=0.0, help="override the workload's target_width (tuning: other kernel lengths)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--tuning", action="append", default=[], metavar="KEY=0|1",
                    help="pin one of the library's alternative code paths for a same-session A/B (pdog_set_tuning), e.g. --tuning no_fold=1")
    ap.add_argument("--no-exact", action="store_true", help="exact mode off (pdog_set_exact(t, 0)): the raw FP32 ranking, for the A/B of its cost")
    ap.add_argument("--data-rank", type=int, default=-1, help="generate the synthetic data of this rank (checks the per-rank seeds on one GPU)")
    ap.add_argument("--chain", action="store_true",
                    help="the SERIAL chain of src/PawsomeTracker.jl:163-169 instead of independent windows: the workload's batch is ONE clip of "
                         "that many frames, frame k searched around frame k-1's answer (pdog_detect_chain: one launch per clip); a step = one clip; "
                         "cpu_baseline = the oracle walking the same chain")
    ap.add_argument("--group-total", type=int, default=0,
                    help="--group only: total windows over the group (default batch x gpus); a total that is not a multiple of the group "
                         "size gives shards that differ by one window and exercises the compaction of the gathered blocks")
    ap.add_argument("--group", action="store_true",
                    help="N GPUs from ONE process through the C ABI's pdog_group_* (in-process RCCL ncclGather) instead of one process per GPU")
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE: start the N ranks ourselves, exactly as the driver
    would (torch.distributed.run, one process per GPU, RCCL), as a CHILD process started before this process has
    touched the GPU; relay rank 0's JSON line and the exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    p = subprocess.run(cmd, env=env)
    return p.returncode


def tracking_sanity(np, got, centres, guesses_h, fh, fw, tw, radii, noise):
    """A disc fully inside frame+window must be found within 1 px of its centre (exactly, when noise-free) — a
    wrong-but-fast kernel must not produce a number."""
    rad = tw // 2 + 1
    inside = ((centres[:, 0] > rad) & (centres[:, 0] <= fh - rad) & (centres[:, 1] > rad) & (centres[:, 1] <= fw - rad)
              & (np.abs(centres - guesses_h) <= np.array(radii) - rad).all(1))
    err = np.abs(got[inside] - centres[inside]).max() if inside.any() else 0
    if not os.environ.get("PDOG_BENCH_NOCHECK"):   # timing-only ablation builds produce wrong positions
        assert err <= (1 if noise else 0), f"tracking sanity failed: max |pos - centre| = {err}"


def result_line(args, desc, world, dt, kern_ms, batch, info, kernel_for_batch, fh, fw, tw, sharding, traffic_key=None):
    n_total = batch * world
    value = n_total * args.steps / dt
    abytes = int(info.algorithmic_bytes_per_window)
    afma = int(info.algorithmic_fma_per_window)
    ach_gbs = abytes * batch / (kern_ms * 1e-3) / 1e9
    fma_rate = afma * batch / (kern_ms * 1e-3)
    return {
        "metric": "DoG+argmax frames/s, 1080p 256x256 windows batch 4096; % HBM roofline @1/8 GPU",
        "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.workload}: {desc}", "frame": [fh, fw], "window": [info.win_h, info.win_w],
                   "batch_per_gpu": batch, "target_width": tw, "kernel_len": info.kernel_len,
                   "noise_levels": args.noise, "variant": info.variant, "kernel_for_this_batch": kernel_for_batch,
                   "strips": info.n_strips, "sharding": sharding},
        "roofline": {"bound": "hbm", "achieved": ach_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": ach_gbs / HBM_PEAK_GBS, "traffic": stored_traffic(traffic_key or args.workload, info.variant, batch),
                     "kernel_ms": kern_ms, "algorithmic_bytes_per_window": abytes,
                     # what `frac` would be with the FP32 vector ALUs 100 % busy on algorithmic FMAs: the ceiling of this path
                     "frac_at_fp32_vector_peak": abytes / (afma / VALU_PEAK_FMA) / 1e9 / HBM_PEAK_GBS,
                     "valu": {"achieved_fma_per_s": fma_rate, "peak_fma_per_s": VALU_PEAK_FMA,
                              "frac": fma_rate / VALU_PEAK_FMA, "algorithmic_fma_per_window": afma},
                     "note": "path is FP32-VALU bound (375 flop/B vs ridge 19.7, DESIGN.md); achieved = algorithmic bytes of one "
                             "step / the step's kernel time (HIP events on the launch stream: every kernel of the step); "
                             "traffic = FETCH_SIZE (x2 gfx950 correction where the kernel reads 16 B/lane) + WRITE_SIZE of every "
                             "kernel of a step, profiles/traffic_r03.json; the chip sustains ~2.0 GHz under this kernel, "
                             "valu.peak is quoted at the nominal 2.4 GHz"},
    }


def run_group(args):
    """N GPUs from one process: pdog_group_* (one tracker + stream per device, contiguous shards, one in-process RCCL
    ncclGather of the positions to device 0 per step).  Weak scaling like the multi-process path: `batch` windows per GPU."""
    import numpy as np
    import torch
    import pawsometracker_jl_amd as pt
    world = args.gpus
    fh, fw, tw, ws, batch, desc = WORKLOADS[args.workload]
    if args.batch:
        batch = args.batch
    ws = pt.fix_window_size(ws if not isinstance(ws, tuple) else (ws[1], ws[0]))
    radii = (ws[0] // 2, ws[1] // 2)
    n_total = args.group_total if args.group_total > 0 else batch * world
    sizes = [pt.shard_range(n_total, r, world) for r in range(world)]
    frames, guesses, guesses_h, centres = [], [], [], []
    fill = 128
    for r in range(world):
        dev = torch.device("cuda", r)
        f, g_h, c = make_frames(torch, sizes[r][1] - sizes[r][0], fh, fw, tw, radii, seed=1000 * r, noise=args.noise, device=dev)
        if r == 0 and args.noise:
            fill = pt.mode(f[0].cpu().numpy())
        frames.append(f)
        guesses.append(torch.from_numpy(g_h).to(dev))
        guesses_h.append(g_h)
        centres.append(c)
    with native_stdout_to_stderr():
        gt = pt.GroupTracker(list(range(world)), fh, fw, tw, ws, True, fill)
    assert all(gt.shard(n_total, r) == tuple(sizes[r]) for r in range(world))
    gt.reserve(n_total)
    info = gt.info(0)
    out = torch.empty((n_total, 2), dtype=torch.int32, device="cuda:0")
    # the trackers launch on their own streams; the kernel time of a step is taken with HIP events on rank 0's
    root_stream = torch.cuda.ExternalStream(gt.stream(0), device=0)
    for r in range(world):
        torch.cuda.synchronize(r)
    with native_stdout_to_stderr():
        for _ in range(args.warmup):
            gt.detect(frames, guesses, n_total, out)
        gt.sync()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    for r in range(world):
        torch.cuda.synchronize(r)
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record(root_stream)
        gt.detect(frames, guesses, n_total, out)
        ev[k][1].record(root_stream)
    gt.sync()
    for r in range(world):
        torch.cuda.synchronize(r)
    dt = time.perf_counter() - t0
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    got = out.cpu().numpy()
    tracking_sanity(np, got, np.concatenate(centres), np.concatenate(guesses_h), fh, fw, tw, radii, args.noise)
    res = result_line(args, desc, world, dt, kern_ms, batch, info, gt.kernel_for_batch(batch), fh, fw, tw,
                      f"frames x{world} from one process (pdog_group_*), in-process RCCL ncclGather of int32[n,2] to device 0")
    res["roofline"]["note"] += "; group mode: kernel_ms = rank 0's kernels + its part of the gather"
    if args.group_total > 0:
        res["value"] = n_total * args.steps / dt
        res["config"]["group_total"] = n_total
        res["config"]["shard_sizes"] = [hi - lo for lo, hi in sizes]
    print(json.dumps(res))
    gt.close()
    return 0


def make_clip(np, n, h, w, tw, radii, seed, noise):
    """One synthetic clip: a dark disc walking an Archimedean spiral out of the frame centre in steps of a few pixels
    (the reference's test trajectory, test/test-basic-test.jl:23-41, without its unseeded jitter), +-noise levels."""
    rng = np.random.Generator(np.random.PCG64(seed))
    rad = int(tw) // 2
    k = np.arange(n)
    step = min(radii) / 4.0                                     # arc length per frame: well inside the search window
    theta = np.sqrt(2.0 * step * k / 3.0 + 1.0)                  # r = 3·theta: ds ≈ r·dtheta
    r = np.minimum(3.0 * theta, min(h, w) / 2.0 - rad - 2)
    ci = np.rint(h / 2 + r * np.sin(theta)).astype(np.int64)
    cj = np.rint(w / 2 + r * np.cos(theta)).astype(np.int64)
    yy, xx = np.mgrid[0:h, 0:w]
    frames = np.full((n, h, w), 128, np.uint8)
    for f in range(n):
        frames[f][(yy - ci[f]) ** 2 + (xx - cj[f]) ** 2 <= rad * rad] = 0
    if noise:
        frames = np.clip(frames.astype(np.int16) + rng.integers(-noise, noise + 1, frames.shape), 0, 255).astype(np.uint8)
    return frames, np.stack([ci + 1, cj + 1], 1).astype(np.int32)   # 1-based centres


def run_chain(args):
    """cfg1 as BASELINE.json states it: ONE clip walked serially (src/PawsomeTracker.jl:163-169) — frame k's window is
    centred on frame k-1's answer.  Frames resident in HBM, one pdog_detect_chain launch per clip; the CPU port walks the same
    chain with the oracle's dense Float64 functor, threaded inside the window like the reference's CPUThreads call."""
    import numpy as np
    import torch
    import pawsometracker_jl_amd as pt
    fh, fw, tw, ws, n_frames, desc = WORKLOADS[args.workload]
    if args.batch:
        n_frames = args.batch
    if args.target_width:
        tw, desc = args.target_width, desc + f" [target_width overridden: {args.target_width}]"
    ws = pt.fix_window_size(ws if not isinstance(ws, tuple) else (ws[1], ws[0]))
    radii = (ws[0] // 2, ws[1] // 2)
    frames_h, centres = make_clip(np, n_frames, fh, fw, tw, radii, seed=0, noise=args.noise)
    fill = pt.mode(frames_h[0])
    dev = torch.device("cuda", 0)
    frames = torch.from_numpy(frames_h).to(dev)
    start = (int(centres[0, 0]), int(centres[0, 1]))
    bt = pt.BatchTracker(fh, fw, tw, ws, True, fill, device=0)
    for kv in args.tuning:
        key, _, val = kv.partition("=")
        bt.set_tuning(key, int(val or 1))
    if args.no_exact:
        bt.set_exact(0)
    bt.use_torch_stream()
    info = bt.info()
    out = torch.empty((n_frames, 2), dtype=torch.int32, device=dev)
    for _ in range(args.warmup):
        bt.detect_chain(frames, start, out=out)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    torch.cuda.synchronize()
    refined0 = bt.exact_stats()[2]
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record()
        bt.detect_chain(frames, start, out=out)
        ev[k][1].record()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    got = out.cpu().numpy()
    err = np.abs(got - centres).max()
    if not os.environ.get("PDOG_BENCH_NOCHECK"):
        assert err <= (1 if args.noise else 0), f"chain lost the target: max |pos - centre| = {err}"
    res = result_line(args, desc + f" [--chain: one clip of {n_frames} frames walked serially]", 1, dt, kern_ms, n_frames, info, "chain", fh, fw, tw,
                      "single GPU, one clip (a serial chain does not shard: replicas only)", traffic_key=args.workload + "_chain")
    res["config"]["chain"] = True
    res["roofline"]["note"] = ("serial chain: one workgroup walks the clip, so this line is LATENCY (us per frame = ms_per_step * 1000 / frames), "
                               "not throughput; achieved/valu are the clip's algorithmic bytes and FMAs over the clip's time. " + res["roofline"]["note"])
    res["us_per_frame"] = dt / args.steps / n_frames * 1e6
    on, thr, refined1 = bt.exact_stats()
    res["exact"] = {"on": on, "threshold_2delta": thr, "refined_windows_per_step": (refined1 - refined0) / args.steps}
    if not args.no_cpu:
        from oracle.dog_oracle import Oracle, build
        build()
        o, strict = Oracle(fast=True), Oracle(fast=False)
        K = strict.dog_kernel(strict.sigma(tw), True)
        quota = cpu_quota()
        cores = max(1, min(o.max_threads(), int(quota))) if quota else o.max_threads()
        best = {}
        for label, nt in (("threads_within_window", cores), ("single_thread", 1)):
            g, pos = start, []
            o.detect(frames_h[0], fill, K, radii, g, nthreads=nt)   # warm the thread pool
            t1 = time.perf_counter()
            for f in range(n_frames):
                g = o.detect(frames_h[f], fill, K, radii, g, nthreads=nt)
                pos.append(g)
            best[label] = (n_frames / (time.perf_counter() - t1), np.array(pos, np.int32))
        for label, (_, pos) in best.items():
            assert np.array_equal(pos, got), f"GPU chain differs from the CPU oracle's chain ({label})"
        L = info.kernel_len
        val = max(v for v, _ in best.values())
        res["cpu_baseline"] = {
            "value": val, "unit": "frames/s", "cores": cores if best["threads_within_window"][0] >= best["single_thread"][0] else 1,
            "cgroup_cpu_quota": quota, "host_threads_visible": o.max_threads(), "kind": "port",
            "sample": f"the same {n_frames}-frame clip walked serially by the oracle (oracle/dog_oracle.c, -Ofast): dense {L}x{L} Float64 correlation + "
                      "first-max argmax per frame, each frame searched around the previous answer; value = the faster of threading inside the window "
                      f"(the reference's CPUThreads model, {cores} threads) and one thread; positions equal to the GPU's on every frame",
            "threads_within_window": {"value": best["threads_within_window"][0], "threads": cores},
            "single_thread": {"value": best["single_thread"][0], "threads": 1}}
    print(json.dumps(res))
    bt.close()
    return 0


def main():
    args = parse_args()
    if args.gpus < 1:
        print("bench.py: --gpus must be >= 1", file=sys.stderr)
        return 2
    env_world = os.environ.get("WORLD_SIZE")
    import torch   # device_count() does not initialise the GPU on this image (safe before starting child processes)
    backend = os.environ.get("PDOG_BENCH_BACKEND", "nccl")
    have = torch.cuda.device_count()
    if backend == "nccl" and have < args.gpus:
        print(f"bench.py: --gpus {args.gpus} requested but only {have} GPU(s) are visible on this node "
              "(RCCL needs one device per rank; PDOG_BENCH_BACKEND=gloo rehearses the control flow on fewer)", file=sys.stderr)
        return 2
    if args.chain:
        if args.gpus != 1 or env_world not in (None, "1"):
            print("bench.py: --chain walks ONE clip serially; a serial chain does not shard (replicas only): use --gpus 1", file=sys.stderr)
            return 2
        return run_chain(args)
    if args.group:
        if env_world not in (None, "1"):
            print("bench.py: --group drives every GPU from ONE process; do not start it under torch.distributed.run", file=sys.stderr)
            return 2
        return run_group(args)
    if env_world is None and args.gpus > 1:
        return self_launch(args)
    if env_world is not None and int(env_world) != args.gpus:
        print(f"bench.py: WORLD_SIZE={env_world} but --gpus {args.gpus}: start N ranks for --gpus N "
              "(or run plain `python bench.py --gpus N`, which starts them itself)", file=sys.stderr)
        return 2

    import numpy as np
    import torch.distributed as dist
    import pawsometracker_jl_amd as pt

    rank = int(os.environ.get("RANK", "0"))
    world = int(env_world or "1")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # PDOG_BENCH_BACKEND=gloo rehearses the N > 1 control flow where RCCL cannot run (several ranks on ONE GPU of a
    # development box): ranks then share devices round-robin and the gather goes through host memory
    dev_index = local_rank if backend == "nccl" else local_rank % max(1, have)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        with native_stdout_to_stderr():
            if backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            else:
                dist.init_process_group(backend, rank=rank, world_size=world)

    fh, fw, tw, ws, batch, desc = WORKLOADS[args.workload]
    if args.batch:
        batch = args.batch
    if args.target_width:
        tw, desc = args.target_width, desc + f" [target_width overridden: {args.target_width}]"
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
    for kv in args.tuning:
        key, _, val = kv.partition("=")
        bt.set_tuning(key, int(val or 1))
    if args.no_exact:
        bt.set_exact(0)
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

    with native_stdout_to_stderr():
        for _ in range(args.warmup):
            step()
        if world > 1:
            pt.gather_positions(out, n_total)   # RCCL sets up its point-to-point channels on first use: not part of any timed step
            torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    torch.cuda.synchronize()
    refined0 = bt.exact_stats()[2]
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
    tracking_sanity(np, got, centres, guesses_h, fh, fw, tw, radii, args.noise)
    if world > 1 and rank == 0:   # the gathered list is the shards in rank order: rank 0's block must be its own answers
        g0 = gathered.cpu().numpy()
        assert g0.shape == (n_total, 2) and np.array_equal(g0[:batch], got), "gathered positions: rank 0's shard differs from its own results"

    if rank == 0:
        sharding = (f"frames x{world}, one process per GPU, gather int32[n,2] to rank 0" + ("" if backend == "nccl" else f" ({backend} rehearsal)")) \
            if world > 1 else "single GPU"
        res = result_line(args, desc, world, dt, kern_ms, batch, info, bt.kernel_for_batch(batch), fh, fw, tw, sharding,
                          traffic_key=(f"tw{args.target_width:g}" if args.target_width and args.workload == "cfg3" else None))
        on, thr, refined1 = bt.exact_stats()
        res["exact"] = {"on": on, "threshold_2delta": thr, "refined_windows_per_step": (refined1 - refined0) / args.steps,
                        "since_create": dict(zip(("windows", "column_blocks", "candidates", "sequential_chains"), bt.exact_detail())),
                        "note": "windows whose two best FP32 responses lay within 2*delta and were re-decided in the reference's Float64 "
                                "arithmetic inside the timed region (rank 0's shard)"}
        if world == 1 and not args.no_cpu:
            ns = min(256, batch)
            cb = cpu_baseline(frames[:ns].cpu().numpy(), guesses_h[:ns], fill, tw, radii)
            cpos = cb["pos"]
            assert np.array_equal(cpos, got[:len(cpos)]), "GPU positions differ from the CPU oracle on the sample"
            L = info.kernel_len
            res["cpu_baseline"] = {
                "value": cb["value"], "unit": "frames/s", "cores": cb["cores"], "effective_cores": cb["cores"],
                "host_threads_visible": cb["host_threads"], "cgroup_cpu_quota": cpu_quota(), "kind": "port",
                "sample": f"first {cb['across'][1]} windows of the same batch (time-bounded), dense {L}x{L} Float64 correlation + "
                          "first-max argmax (oracle/dog_oracle.c, -Ofast, OpenMP, threads = the cgroup CPU quota); value = the "
                          f"faster of the two threadings; positions equal to the GPU's on all {len(cpos)} sampled windows",
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
    return 0


if __name__ == "__main__":
    sys.exit(main())
