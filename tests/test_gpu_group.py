"""Multi-GPU leg on the one-GPU box: the C ABI's tracker group (pdog_group_*, in-process RCCL ncclGather) with a
single rank — RCCL accepts a 1-rank communicator, so ncclCommInitAll, the grouped ncclGather on the tracker's stream
and the result placement all run for real — plus torch.distributed's nccl (= RCCL) backend at world size 1 through
gather_positions, the bench's --group line, and the stream hand-over of pdog_set_stream.  More ranks need more
devices: the driver's 8-GPU run is the first time RCCL moves bytes between GPUs."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pt():
    import pawsometracker_jl_amd as m
    return m


def test_group_of_one_equals_detect_batch(pt, oracle):
    import torch
    from oracle import synth
    for (tw, ws, fh, fw, n) in ((25, (45, 45), 240, 320, 70), (25, (256, 256), 540, 960, 33), (40, (61, 61), 160, 200, 9)):
        radii = (ws[0] // 2, ws[1] // 2)
        frames, guesses, _ = synth.make_batch(n, fh, fw, tw, radii, True, seed=11, noise=3)
        fill = oracle.mode_u8(frames[0])
        d_f, d_g = torch.from_numpy(frames).cuda(), torch.from_numpy(guesses).cuda()
        bt = pt.BatchTracker(fh, fw, tw, ws, True, fill)
        want = bt.detect(d_f, d_g)
        bt.sync()
        want = want.cpu().numpy()
        bt.close()
        gt = pt.GroupTracker([0], fh, fw, tw, ws, True, fill)
        assert gt.size() == 1 and gt.shard(n, 0) == (0, n)
        out = torch.full((n, 2), -1, dtype=torch.int32, device="cuda:0")
        torch.cuda.synchronize()
        for _ in range(2):                     # second call: buffers and communicator are reused
            gt.detect([d_f], [d_g], n, out)
            gt.sync()
            assert np.array_equal(out.cpu().numpy(), want)
        # with a frame index: every window looks at frame 0
        fi = torch.zeros(n, dtype=torch.int32, device="cuda:0")
        torch.cuda.synchronize()
        gt.detect([d_f], [d_g], n, out, frame_index=[fi])
        gt.sync()
        bt = pt.BatchTracker(fh, fw, tw, ws, True, fill)
        want0 = bt.detect(d_f, d_g, frame_index=fi)
        bt.sync()
        assert np.array_equal(out.cpu().numpy(), want0.cpu().numpy())
        bt.close()
        gt.close()
        ref = oracle.detect_batch(frames[:8], fill, oracle.dog_kernel(oracle.sigma(tw), True), radii, guesses[:8])
        assert np.array_equal(want[:8], ref)


def test_unequal_shard_compaction_kernel(pt):
    """pdog_group_detect_batch gathers max-shard-sized blocks and, when n_total is not a multiple of the group size, compacts
    them on the root with a copy kernel.  With one rank the shards are always equal, so that kernel is driven directly
    here (pdog_group_test_compact) on synthetic gathered buffers of 2 … 8 ranks and checked against pdog_shard_owner /
    pdog_shard_range — the partition the multi-GPU path is built on."""
    import ctypes as C
    import torch
    L = pt.lib()
    for ndev, n_total in ((3, 10), (3, 11), (8, 8191), (8, 8193), (2, 1), (5, 3), (7, 4096), (4, 4098)):
        max_n = (n_total + ndev - 1) // ndev
        gathered = np.full((ndev, max_n, 2), -7, np.int32)            # padding entries must never reach the output
        want = np.empty((n_total, 2), np.int32)
        for w in range(n_total):
            r, k = C.c_int(), C.c_int()
            assert L.pdog_shard_owner(n_total, ndev, w, C.byref(r), C.byref(k)) == 0
            lo, hi = C.c_int(), C.c_int()
            assert L.pdog_shard_range(n_total, ndev, r.value, C.byref(lo), C.byref(hi)) == 0 and lo.value + k.value == w < hi.value
            gathered[r.value, k.value] = (w + 1, 3 * w + 2)
            want[w] = (w + 1, 3 * w + 2)
        d_g = torch.from_numpy(gathered).cuda()
        d_o = torch.full((n_total, 2), -1, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        pt._lib.check(L.pdog_group_test_compact(C.c_void_p(d_g.data_ptr()), n_total, ndev, C.c_void_p(d_o.data_ptr())))
        assert np.array_equal(d_o.cpu().numpy(), want), (ndev, n_total)


def test_group_checks_every_shard_pointer_before_launching(pt):
    """A null frames / guesses pointer for any rank with a non-empty shard is PDOG_E_ARG before anything is launched."""
    import ctypes as C
    import torch
    gt = pt.GroupTracker([0], 120, 160, 25, (45, 45), True, 128)
    L = pt.lib()
    out = torch.zeros((4, 2), dtype=torch.int32, device="cuda")
    frames = torch.zeros((4, 120, 160), dtype=torch.uint8, device="cuda")
    guesses = torch.full((4, 2), 60, dtype=torch.int32, device="cuda")
    ptrs = lambda t: (C.c_void_p * 1)(C.c_void_p(t.data_ptr()) if t is not None else None)
    nf = (C.c_int * 1)(4)
    for fr, gu in ((None, guesses), (frames, None)):
        rc = L.pdog_group_detect_batch(gt._h, ptrs(fr), 120 * 160, 160, nf, None, ptrs(gu), 4, C.c_void_p(out.data_ptr()))
        assert rc == pt._lib.PDOG_E_ARG and b"rank 0" in L.pdog_last_error()
    gt.sync()
    assert int(out.abs().sum()) == 0             # nothing ran
    gt.close()


def test_group_rejects_bad_devices(pt):
    import torch
    nd = torch.cuda.device_count()
    with pytest.raises(pt.PdogError) as e:
        pt.GroupTracker([0, nd], 64, 64, 25, (45, 45), True, 128)
    assert e.value.code == pt._lib.PDOG_E_NODEV
    with pytest.raises(pt.PdogError) as e:
        pt.GroupTracker([0, 0], 64, 64, 25, (45, 45), True, 128)
    assert e.value.code == pt._lib.PDOG_E_ARG and "twice" in str(e.value)


_NCCL_WORLD1 = r"""
import os, sys
sys.path.insert(0, {root!r})
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="{port}", RANK="0", WORLD_SIZE="1")
import torch, torch.distributed as dist
import pawsometracker_jl_amd as pt
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
local = torch.stack([torch.arange(7, dtype=torch.int32), torch.arange(7, dtype=torch.int32) * 3], 1).to(dev)
out = pt.gather_positions(local, 7)
hs = [pt.gather_positions(local + k, 7, async_op=True) for k in range(3)]
outs = [h.wait() for h in hs]
torch.cuda.synchronize()
assert out.is_cuda and torch.equal(out, local) and all(torch.equal(o, local + k) for k, o in enumerate(outs))
dist.destroy_process_group()
print("nccl-world1-ok")
"""


def test_gather_positions_on_the_nccl_backend_world1():
    """torch.distributed's nccl backend IS RCCL on ROCm: device tensors, device_id= init, blocking and asynchronous
    gather — the calls bench.py makes with N > 1 — at world size 1, in a child process."""
    import socket
    with socket.socket() as sk:          # a port nobody holds (a fixed one can sit in TIME_WAIT from an earlier run)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    p = subprocess.run([sys.executable, "-c", _NCCL_WORLD1.format(root=ROOT, port=port)], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "nccl-world1-ok" in p.stdout, (p.returncode, p.stderr[-1500:])


def test_bench_group_mode_and_refusal():
    import torch
    env = dict(os.environ)
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--group", "--steps", "3", "--warmup", "1", "--batch", "256"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), p.stdout[:400]      # ONE line on stdout: RCCL's banner goes to stderr
    r = json.loads(lines[0])
    assert r["n_gpus"] == 1 and "pdog_group" in r["config"]["sharding"] and r["value"] > 0 and r["roofline"]["kernel_ms"] > 0
    if torch.cuda.device_count() == 1:   # more ranks than devices: clear message, non-zero exit, nothing started
        for extra in ([], ["--group"]):
            p = subprocess.run([sys.executable, "bench.py", "--gpus", "2"] + extra, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
            assert p.returncode == 2 and "only 1 GPU(s) are visible" in p.stderr, (p.returncode, p.stderr[-500:])


def test_alternating_torch_streams_keep_the_scratch_ordered(pt, oracle):
    """pdog_set_stream hands the tracker's per-tracker scratch (strip partials, two-pass intermediate, counters)
    from one stream to the next with an event: detect() calls issued alternately under two torch streams must
    give the answers of serial calls (they raced on the partials before)."""
    import torch
    from oracle import synth
    cases = ((25, (256, 256), 540, 960, 600), (40, (61, 61), 160, 200, 40))   # roll + thin side stream; two-pass scratch
    for tw, ws, fh, fw, n in cases:
        radii = (ws[0] // 2, ws[1] // 2)
        frames, guesses, _ = synth.make_batch(n, fh, fw, tw, radii, True, seed=23, noise=3)
        fill = oracle.mode_u8(frames[0])
        ref = oracle.detect_batch(frames[:16], fill, oracle.dog_kernel(oracle.sigma(tw), True), radii, guesses[:16])
        d_f, d_g = torch.from_numpy(frames).cuda(), torch.from_numpy(guesses).cuda()
        g2 = torch.roll(d_g, 1, 0).contiguous()      # a second, different batch on the same frames
        bt = pt.BatchTracker(fh, fw, tw, ws, True, fill)
        bt.reserve(n)
        a_ref = bt.detect(d_f, d_g).clone()
        b_ref = bt.detect(d_f, g2, frame_index=torch.roll(torch.arange(n, dtype=torch.int32, device="cuda"), 1, 0).contiguous()).clone()
        torch.cuda.synchronize()
        fi2 = torch.roll(torch.arange(n, dtype=torch.int32, device="cuda"), 1, 0).contiguous()
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        torch.cuda.synchronize()
        outs = []
        for k in range(6):
            with torch.cuda.stream(s1 if k % 2 == 0 else s2):
                outs.append(bt.detect(d_f, d_g) if k % 2 == 0 else bt.detect(d_f, g2, frame_index=fi2))
        torch.cuda.synchronize()
        for k, o in enumerate(outs):
            assert torch.equal(o, a_ref if k % 2 == 0 else b_ref), (tw, k)
        assert np.array_equal(a_ref[:16].cpu().numpy(), ref)
        bt.close()


def test_device_guess_out_of_range_is_reported_by_sync(pt, oracle):
    """Device-resident guesses cannot be checked before the launch; where the reference raises BoundsError
    (guess outside [-l÷2, sz + l÷2 + 1], src/PawsomeTracker.jl:45-46) the kernels raise a flag that pdog_sync reports
    as PDOG_E_RANGE — once: the flag is cleared by the call that reports it.  Every kernel family."""
    import torch
    from oracle import synth
    tw, ws, fh, fw = 25, (45, 45), 120, 160
    radii = (22, 22)
    hw = oracle.kernel_len(oracle.sigma(tw)) // 2
    frames, guesses, _ = synth.make_batch(6, fh, fw, tw, radii, True, seed=3, noise=2)
    fill = oracle.mode_u8(frames[0])
    d_f = torch.from_numpy(frames).cuda()
    edge = guesses.copy()
    edge[0] = (-hw, fw + hw + 1)              # the outermost legal guess
    bad = guesses.copy()
    bad[3] = (fh + hw + 2, 10)                # one row too far
    for variant in (-1, 300, 100, 200, 13):
        bt = pt.BatchTracker(fh, fw, tw, ws, True, fill)
        if variant >= 0:
            bt.set_variant(variant)
        bt.detect(d_f, torch.from_numpy(edge).cuda())
        bt.sync()                              # legal: no error
        out = bt.detect(d_f, torch.from_numpy(bad).cuda())
        with pytest.raises(pt.PdogError) as e:
            bt.sync()
        assert e.value.code == pt._lib.PDOG_E_RANGE and "BoundsError" in str(e.value), variant
        bt.sync()                              # reported once
        ref = oracle.detect_batch(frames[:3], fill, oracle.dog_kernel(oracle.sigma(tw), True), radii, bad[:3])
        assert np.array_equal(out.cpu().numpy()[:3], ref)      # the in-range windows of that batch are still right
        # the serial chain checks its start guess the same way
        bt.detect_chains(d_f[:3].clone().unsqueeze(0), torch.tensor([[fh + hw + 2, 5]], dtype=torch.int32).cuda())
        with pytest.raises(pt.PdogError):
            bt.sync()
        bt.close()


def test_no_path_switch_comes_from_the_environment(pt, oracle, monkeypatch):
    """The product library reads resource limits from the environment and nothing else: the switches that used to select
    code paths (round 2: PDOG_NO_EXACT silently removed the position guarantee for every tracker of the process) do
    nothing; the same paths are pinned per tracker, explicitly, through pdog_set_exact / pdog_set_tuning."""
    from oracle import synth
    frames, guesses, _ = synth.make_batch(1, 120, 160, 25, (22, 22), True, seed=4, noise=2)
    for name in ("PDOG_NO_EXACT", "PDOG_HOST_COPY", "PDOG_NO_TILED", "PDOG_TWOPASS_4L"):
        monkeypatch.setenv(name, "1")
    t1 = pt.Tracker(frames[0], 25, (45, 45), True)
    assert t1.exact_stats()[0] is True
    g = (int(guesses[0, 0]), int(guesses[0, 1]))
    want = t1(g)
    t1.set_exact(0)
    assert t1.exact_stats()[0] is False and t1(g) == want
    for key in ("host_copy", "host_sync", "twopass_4l", "no_tiled", "no_roll_map", "no_fold", "fold_always"):
        t1.set_tuning(key, 1)
        assert t1(g) == want, key
        t1.set_tuning(key, 0)
    with pytest.raises(pt.PdogError):
        t1.set_tuning("no_such_key", 1)
    t1.close()

