"""Exact mode (csrc/dog_exact.hpp): returned positions ARE the Float64 reference's, by construction.

The reference ranks dense Float64 sums (src/PawsomeTracker.jl:57-59); the kernels rank FP32 separable sums and
re-decide every window whose two best responses lie within 2δ of each other in the reference's own arithmetic.
This file is the census the guarantee is checked by: a full cfg3 batch, ≥1024 windows of cfg4 and cfg5, and 20 000+
45×45 windows built to be hard (noise only, ±1-level contrast, two near-equal blobs, even-sized targets whose centre
falls between pixels) — every GPU position compared with the oracle's Float64 statements, every disagreement and a
random sample adjudicated by the DENSE oracle in the reference's accumulation order.  PARITY UNPINNED: the oracle is
this repo's restatement (oracle/dog_oracle.c); the reference holds no fixture for this path."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pt():
    import pawsometracker_jl_amd as m
    return m


def _hard_windows(n, fh, fw, tw, seed):
    """n frames fh×fw with one 45×45-window guess each, in five families built to make the top responses nearly
    tie.  Returns frames u8 [n, fh, fw], guesses int32 [n, 2], family ids."""
    rng = np.random.Generator(np.random.PCG64(seed))
    frames = np.full((n, fh, fw), 128, np.uint8)
    fam = rng.integers(0, 5, n)
    rad = tw // 2
    yy, xx = np.mgrid[-rad:rad + 1, -rad:rad + 1]
    disc = (yy * yy + xx * xx) <= rad * rad
    ci0, cj0 = fh // 2, fw // 2
    guesses = np.empty((n, 2), np.int32)

    def put(b, i, j, val, mask=disc):
        h, w = mask.shape
        i0, j0 = i - h // 2, j - w // 2
        sub = frames[b, max(i0, 0):i0 + h, max(j0, 0):j0 + w]
        m = mask[max(-i0, 0):max(-i0, 0) + sub.shape[0], max(-j0, 0):max(-j0, 0) + sub.shape[1]]
        sub[m] = val

    for b in range(n):
        f = fam[b]
        gi, gj = ci0 + rng.integers(-6, 7), cj0 + rng.integers(-6, 7)
        if f == 0:      # noise only: nothing to find, the maximum is whatever the noise makes it
            amp = int(rng.integers(1, 4))
            frames[b] = (128 + rng.integers(-amp, amp + 1, (fh, fw))).astype(np.uint8)
        elif f == 1:    # a ±1-level target, half the time under ±1 noise
            put(b, ci0 + rng.integers(-8, 9), cj0 + rng.integers(-8, 9), 127)
            if rng.integers(0, 2):
                frames[b] = (frames[b].astype(np.int16) + rng.integers(-1, 2, (fh, fw))).clip(0, 255).astype(np.uint8)
        elif f == 2:    # two blobs of (nearly) equal contrast, mirrored about the window centre
            di, dj = int(rng.integers(6, 15)), int(rng.integers(-14, 15))
            put(b, gi - di, gj - dj, 0)
            put(b, gi + di, gj + dj, int(rng.integers(0, 2)))
            if rng.integers(0, 2):
                frames[b] = (frames[b].astype(np.int16) + rng.integers(-1, 2, (fh, fw))).clip(0, 255).astype(np.uint8)
        elif f == 3:    # an even-sized square target: its centre falls between pixels, 2 or 4 pixels tie mathematically
            side = 2 * int(rng.integers(4, 12))
            mask = np.ones((side + int(rng.integers(0, 2)), side), bool)
            put(b, ci0 + rng.integers(-6, 7), cj0 + rng.integers(-6, 7), int(rng.integers(0, 100)), mask)
        else:           # an ordinary dark disc under ±3 noise (the bench's recipe) as the easy control
            put(b, ci0 + rng.integers(-10, 11), cj0 + rng.integers(-10, 11), 0)
            frames[b] = (frames[b].astype(np.int16) + rng.integers(-3, 4, (fh, fw))).clip(0, 255).astype(np.uint8)
        guesses[b] = (gi + 1, gj + 1)
    return frames, guesses, fam


def _gpu_positions(pt, frames, guesses, tw, ws, fill, exact, variant=-1):
    import torch
    bt = pt.BatchTracker(frames.shape[1], frames.shape[2], tw, ws, True, fill)
    if variant >= 0:
        bt.set_variant(variant)
    bt.set_exact(exact)
    out = bt.detect(torch.from_numpy(frames).cuda(), torch.from_numpy(guesses).cuda())
    bt.sync()
    got = out.cpu().numpy()
    stats = bt.exact_stats()
    bt.close()
    return got, stats


def _adjudicate(oracle, frames, guesses, fill, K, radii, idx):
    return np.array([oracle.detect(frames[b], fill, K, radii, guesses[b]) for b in idx], np.int32).reshape(-1, 2)


def test_hard_window_census_45x45(pt, oracle):
    tw, ws, radii = 25, (45, 45), (22, 22)
    n, fh, fw = 20480, 128, 128
    frames, guesses, fam = _hard_windows(n, fh, fw, tw, seed=77)
    fill = 128
    sig = oracle.sigma(tw)
    K = oracle.dog_kernel(sig, True)
    # the reference's arithmetic for EVERY window: dense 65×65 Float64 in kernel column-major order (8.6 M MAC each)
    dense = oracle.detect_batch_par(frames, fill, K, sig, True, radii, guesses, separable=False)
    for variant in (-1, 300, 100, 200, 13):  # automatic choice, fused (inline refinement), roll, two-pass, ring + finishing kernel
        got, (on, thr, refined) = _gpu_positions(pt, frames, guesses, tw, ws, fill, True, variant)
        bad = np.flatnonzero((got != dense).any(1))
        assert on and bad.size == 0, (variant, bad.size, fam[bad][:20], got[bad][:5], dense[bad][:5])
        assert refined > 0                  # the census does reach the refinement
        raw, _ = _gpu_positions(pt, frames, guesses, tw, ws, fill, False, variant)
        wrong = np.flatnonzero((raw != dense).any(1))
        print(f"variant {variant}: refined {refined} of {n} windows; FP32 ranking alone differs from the reference on {wrong.size} "
              f"(families {np.bincount(fam[wrong], minlength=5).tolist()}), exact mode on 0")


def test_refine_everything_equals_the_dense_oracle(pt, oracle):
    """pdog_set_exact(t, 2): every window is refined with an infinite threshold, i.e. EVERY pixel is evaluated as the
    reference evaluates it — the whole reference computation on the device.  Positions must equal the dense oracle's
    on noise-only windows, where nothing but the exact Float64 values decides."""
    import torch
    tw, ws, radii = 10, (21, 21), (10, 10)       # the reference's test defaults: l = 29
    rng = np.random.Generator(np.random.PCG64(5))
    n, fh, fw = 96, 64, 80
    frames = (128 + rng.integers(-2, 3, (n, fh, fw))).astype(np.uint8)
    guesses = np.stack([rng.integers(-5, fh + 6, n), rng.integers(-5, fw + 6, n)], 1).astype(np.int32)   # windows hanging over every border
    fill = 128
    sig = oracle.sigma(tw)
    K = oracle.dog_kernel(sig, True)
    dense = oracle.detect_batch_par(frames, fill, K, sig, True, radii, guesses, separable=False)
    for variant in (-1, 300, 129, 200, 0):
        bt = pt.BatchTracker(fh, fw, tw, ws, True, fill)
        if variant >= 0:
            bt.set_variant(variant)
        bt.set_exact(2)
        got = bt.detect(torch.from_numpy(frames).cuda(), torch.from_numpy(guesses).cuda())
        bt.sync()
        assert np.array_equal(got.cpu().numpy(), dense), variant
        assert bt.exact_stats()[2] == n
        bt.close()
    # the serial chain (inline refinement in the persistent kernels): frame k's guess is frame k-1's exact answer
    clip = frames[:40]
    want = []
    g = (30, 40)
    for k in range(len(clip)):
        g = oracle.detect(clip[k], fill, K, radii, g)
        want.append(g)
    for variant in (-1, 300, 129):
        bt = pt.BatchTracker(fh, fw, tw, ws, True, fill)
        if variant >= 0:
            bt.set_variant(variant)
        bt.set_exact(2)
        got = bt.detect_chain(torch.from_numpy(clip).cuda(), (30, 40))
        bt.sync()
        assert got.cpu().numpy().tolist() == [list(w) for w in want], variant
        bt.close()
    # many clips at once: the persistent roll chain, a workgroup per clip
    nc, nf = 24, 4
    clips = frames[:nc * nf].reshape(nc, nf, fh, fw)
    starts = np.stack([rng.integers(20, 44, nc), rng.integers(20, 60, nc)], 1).astype(np.int32)
    want = np.empty((nc, nf, 2), np.int32)
    for c in range(nc):
        g = tuple(starts[c])
        for k in range(nf):
            g = oracle.detect(clips[c, k], fill, K, radii, g)
            want[c, k] = g
    bt = pt.BatchTracker(fh, fw, tw, ws, True, fill)
    bt.set_variant(129)
    bt.set_exact(2)
    got = bt.detect_chains(torch.from_numpy(np.ascontiguousarray(clips)).cuda(), torch.from_numpy(starts).cuda())
    bt.sync()
    assert np.array_equal(got.cpu().numpy(), want)
    bt.close()


def test_host_functor_refines_before_it_answers(pt, oracle):
    """pdog_detect_host publishes a ticket the host polls; with a near-tie the ticket must not go out before the
    refinement has rewritten the answer (fused kernel: inline; two-pass: the refinement kernel publishes)."""
    rng = np.random.Generator(np.random.PCG64(9))
    for tw, ws, fh, fw in ((25, (45, 45), 160, 200), (25, (256, 256), 400, 500)):
        radii = (ws[0] // 2, ws[1] // 2)
        K = oracle.dog_kernel(oracle.sigma(tw), True)
        for trial in range(6):
            frame = (128 + rng.integers(-1, 2, (fh, fw))).astype(np.uint8)      # noise only: the refinement decides
            t = pt.Tracker(frame, tw, ws, True)
            guess = (int(rng.integers(1, fh + 1)), int(rng.integers(1, fw + 1)))
            want = oracle.detect(frame, t.img.fillvalue, K, radii, guess)
            assert t(guess) == want, (tw, ws, trial)
            assert t.exact_stats()[2] >= 0
            t.close()


@pytest.mark.parametrize("cfg", ["cfg3", "cfg4", "cfg5"])
def test_full_size_census(pt, oracle, cfg):
    """cfg3: the whole 4096-window headline batch; cfg4 / cfg5: 1024 windows each — the bench's own data (±3-level
    noise).  Every GPU position against the oracle's separable Float64 statement; every disagreement, and a random
    sample, against the dense oracle in the reference's accumulation order."""
    import sys, os
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    fh, fw, tw, ws, batch, _ = bench.WORKLOADS[cfg]
    n = 4096 if cfg == "cfg3" else 1024
    ws = pt.fix_window_size(ws)
    radii = (ws[0] // 2, ws[1] // 2)
    frames, guesses_h, _ = bench.make_frames(torch, n, fh, fw, tw, radii, seed=0, noise=3, device=torch.device("cuda", 0))
    fill = pt.mode(frames[0].cpu().numpy())
    bt = pt.BatchTracker(fh, fw, tw, ws, True, fill)
    got = bt.detect(frames, torch.from_numpy(guesses_h).cuda())
    bt.sync()
    got = got.cpu().numpy()
    on, thr, refined = bt.exact_stats()
    bt.set_exact(False)
    raw = bt.detect(frames, torch.from_numpy(guesses_h).cuda())
    bt.sync()
    raw = raw.cpu().numpy()
    bt.close()
    sig = oracle.sigma(tw)
    K = oracle.dog_kernel(sig, True)
    sep = np.empty((n, 2), np.int32)
    step = 256
    host = np.empty((step, fh, fw), np.uint8)
    sample_idx, sample_frames = [], {}
    rng = np.random.Generator(np.random.PCG64(1))
    pick = set(rng.choice(n, 24 if cfg == "cfg5" else 48, replace=False).tolist())
    disagree = []
    for b0 in range(0, n, step):
        m = min(step, n - b0)
        host[:m] = frames[b0:b0 + m].cpu().numpy()
        sep[b0:b0 + m] = oracle.detect_batch_par(host[:m], fill, K, sig, True, radii, guesses_h[b0:b0 + m], separable=True)
        for b in range(b0, b0 + m):
            if b in pick or (got[b] != sep[b]).any():
                if (got[b] != sep[b]).any():
                    disagree.append(b)
                dense = oracle.detect(host[b - b0], fill, K, radii, guesses_h[b])
                assert tuple(got[b]) == dense, (cfg, b, got[b], dense, sep[b])
    wrong_raw = int((raw != got).any(1).sum())
    print(f"{cfg}: {n} windows, refined {refined}, GPU vs separable-f64 disagreements {len(disagree)} (all adjudicated for the GPU by the "
          f"dense oracle), FP32 ranking alone differs from exact mode on {wrong_raw}")
    assert on and len(disagree) <= n // 100


def test_response_map_path_equals_rescan_path(pt, monkeypatch):
    """Two-pass path in exact mode: the candidates of a flagged window are read off the FP32 response map the column pass
    wrote (csrc/dog_exact.hpp, `map`); with PDOG_MAP_MB=0 (read once, at create) they are recomputed block by block as
    on the other paths.  Same candidates, same positions — on hard 45×45 windows (l = 65, two-pass forced) and on
    wide-target windows (tw = 120, l = 293: every window flagged)."""
    import torch
    def run(frames, guesses, tw, ws, variant):
        bt = pt.BatchTracker(frames.shape[1], frames.shape[2], tw, ws, True, 128)
        if variant >= 0:
            bt.set_variant(variant)
        out = bt.detect(torch.from_numpy(frames).cuda(), torch.from_numpy(guesses).cuda())
        bt.sync()
        res = out.cpu().numpy(), bt.exact_stats()[2], bt.exact_detail()
        bt.close()
        return res

    frames, guesses, _ = _hard_windows(4096, 128, 128, 25, seed=5)
    rng = np.random.Generator(np.random.PCG64(8))
    wide = (128 + rng.integers(-3, 4, (48, 480, 640))).astype(np.uint8)
    yy, xx = np.mgrid[0:480, 0:640]
    wg = np.empty((48, 2), np.int32)
    for b in range(48):
        ci, cj = int(rng.integers(150, 330)), int(rng.integers(200, 440))
        if b % 2 == 0:   # every other window holds nothing but noise: near-ties whatever the error bound (the two-pass kernels'
            wide[b][(yy - ci) ** 2 + (xx - cj) ** 2 <= 60 * 60] = 10   # blocked accumulation leaves most DISC windows unflagged since round 3)
        wg[b] = (ci + int(rng.integers(-20, 21)), cj + int(rng.integers(-20, 21)))
    cases = [(frames, guesses, 25, (45, 45), 200), (wide, wg, 120, None, -1)]
    with_map = [run(f, g, tw, ws if ws else (205, 205), v) for f, g, tw, ws, v in cases]
    monkeypatch.setenv("PDOG_MAP_MB", "0")
    without = [run(f, g, tw, ws if ws else (205, 205), v) for f, g, tw, ws, v in cases]
    for (pa, ra, da), (pb, rb, db) in zip(with_map, without):
        assert np.array_equal(pa, pb)
        assert ra == rb and ra > 0
        # (candidate counts differ by design: without a map the window's own |pixel − dc| bound is taken before the
        # candidates are recomputed, with a map only when the first scan finds many)


def test_hard_batches_large_windows_exact_and_not_slow(pt, oracle, monkeypatch):
    """257×257 windows with nothing to find (±2 noise) or a target one grey level darker under ±1 noise: every window is
    flagged, and under the a-priori V = 255 bound thousands of pixels per window lie "within T of the maximum" (81–206 ms
    per 4 096 windows when that went through the candidate list's overflow path).  The refinement takes the window's own
    max |pixel − dc| (δ is proportional to it) and batches after the first read their candidates off the response map:
    positions equal the dense oracle's, equal with and without the map, and the batch stays within 10× the raw ranking."""
    import time
    import torch
    tw, ws, h, w, n, nf = 25, (257, 257), 600, 800, 256, 8
    radii = (128, 128)
    sig = oracle.sigma(tw)
    K = oracle.dog_kernel(sig, True)
    rng = np.random.Generator(np.random.PCG64(31))
    for kind in ("noise", "faint"):
        amp = 2 if kind == "noise" else 1
        frames = (128 + rng.integers(-amp, amp + 1, (nf, h, w))).astype(np.uint8)
        if kind == "faint":
            yy, xx = np.ogrid[0:h, 0:w]
            for k in range(nf):
                ci, cj = int(rng.integers(200, 400)), int(rng.integers(250, 550))
                frames[k][(yy - ci) ** 2 + (xx - cj) ** 2 <= 144] -= 1
        fi = rng.integers(0, nf, n).astype(np.int32)
        guesses = np.stack([rng.integers(150, 450, n), rng.integers(200, 600, n)], 1).astype(np.int32)
        d_f, d_fi, d_g = torch.from_numpy(frames).cuda(), torch.from_numpy(fi).cuda(), torch.from_numpy(guesses).cuda()

        def run(exact, no_map=False):
            bt = pt.BatchTracker(h, w, tw, ws, True, 128)
            bt.set_variant(100)                      # the roll kernel, as a large batch would run
            bt.set_exact(exact)
            bt.set_tuning("no_roll_map", int(no_map))
            outs = []
            for _ in range(3):                       # the first batch recomputes its candidates, later ones use the map
                outs.append(bt.detect(d_f, d_g, d_fi).cpu().numpy())
            bt.sync()
            t0 = time.perf_counter()
            for _ in range(3):
                bt.detect(d_f, d_g, d_fi)
            bt.sync()
            dt = (time.perf_counter() - t0) / 3
            refined = bt.exact_stats()[2]
            bt.close()
            assert all(np.array_equal(o, outs[0]) for o in outs)
            return outs[0], dt, refined

        got, dt, refined = run(True)
        raw, dt_raw, _ = run(False)
        got_nomap, _, _ = run(True, no_map=True)
        assert refined >= 5 * n                                          # every window of every batch was flagged
        assert np.array_equal(got, got_nomap)
        for b in range(0, n, 16):                                        # 16 windows against the reference's own arithmetic
            assert tuple(got[b].tolist()) == oracle.detect(frames[fi[b]], 128, K, radii, tuple(guesses[b])), (kind, b)
        print(f"{kind}: exact {dt * 1e3:.2f} ms, raw {dt_raw * 1e3:.2f} ms per {n} windows; FP32 ranking alone differs on {(got != raw).any(1).sum()}")
        assert dt < 10 * dt_raw + 2e-3, (kind, dt, dt_raw)


@pytest.mark.parametrize("win_w", [100, 201, 481])
def test_peak_in_strip_overlap_is_not_its_own_runner_up(pt, oracle, win_w):
    """The roll kernel's last strip is shifted left over its predecessor; a peak inside the overlap reaches the strip
    combine twice (same index, same value).  Round 2 merged it with itself: runner-up = best, gap 0 ≤ T, so every
    well-tracked target of such a window (the centre column of window_size = 100 lies in the overlap) was refined
    although nothing was near a tie.  Clean discs (no noise: answer = disc centre by symmetry, SURVEY §8c i) with the
    target inside the overlap, widths 100 / 201 / 481, batch kernel (variant 100) and the persistent chain kernel:
    oracle positions, NO window refined, and exact mode on costs what exact mode off costs."""
    import torch
    tw, win_h = 25, 33
    ws = pt.fix_window_size((win_w, win_h))          # (w, h) → (h, w) like the reference (:70)
    radii = (ws[0] // 2, ws[1] // 2)
    n2 = 2 * radii[1] + 1
    assert n2 % 64 > 6, "widths whose remainder goes to a last, shifted strip"
    lo, hi = n2 - 64, (n2 // 64) * 64 - 1            # window columns the last two strips share
    n, fh, fw = 1024, 72, n2 + 80
    rng = np.random.Generator(np.random.PCG64(win_w))
    frames = np.full((n, fh, fw), 128, np.uint8)
    guesses = np.empty((n, 2), np.int32)
    centres = np.empty((n, 2), np.int32)
    yy, xx = np.mgrid[0:fh, 0:fw]
    for b in range(n):
        gi, gj = fh // 2 + int(rng.integers(-3, 4)), fw // 2 + int(rng.integers(-8, 9))   # 1-based guess
        x = int(rng.integers(lo + 2, hi - 1))                                             # window column of the disc centre, inside the overlap
        ci, cj = gi + int(rng.integers(-3, 4)), gj - radii[1] + x                          # 1-based frame position
        frames[b][(yy - (ci - 1)) ** 2 + (xx - (cj - 1)) ** 2 <= (tw // 2) ** 2] = 0
        guesses[b] = (gi, gj)
        centres[b] = (ci, cj)
    fill = 128
    K = oracle.dog_kernel(oracle.sigma(tw), True)
    sep = oracle.detect_batch_par(frames[:64], fill, K, oracle.sigma(tw), True, radii, guesses[:64], separable=True)
    assert np.array_equal(sep, centres[:64])
    d_frames, d_guesses = torch.from_numpy(frames).cuda(), torch.from_numpy(guesses).cuda()
    bt = pt.BatchTracker(fh, fw, tw, ws, True, fill)
    bt.set_variant(100)
    assert bt.kernel_for_batch(n) == 100

    def timed(reps=7):
        best = 1e9
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            bt.use_torch_stream()
            e0.record()
            for _ in range(4):
                out = bt.detect(d_frames, d_guesses)
            e1.record()
            bt.sync()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 4)
        return out.cpu().numpy(), best

    got, t_on = timed()
    assert np.array_equal(got, centres), win_w
    assert bt.exact_stats()[2] == 0, (win_w, bt.exact_stats())
    bt.set_exact(False)
    raw, t_off = timed()
    bt.set_exact(True)
    assert np.array_equal(raw, centres)
    print(f"width {win_w}: exact on {t_on:.4f} ms, off {t_off:.4f} ms per {n} windows")
    assert t_on <= 1.05 * t_off + 0.002
    # persistent chain kernel: 256 clips x 4 frames of the same windows, each frame searched around the previous answer
    clips = d_frames.view(256, 4, fh, fw)
    starts = d_guesses.view(256, 4, 2)[:, 0, :].contiguous()
    out = bt.detect_chains(clips, starts).cpu().numpy()
    bt.sync()
    assert bt.exact_stats()[2] == 0, (win_w, "chain", bt.exact_stats())
    for c in range(0, 256, 17):
        g = tuple(int(v) for v in guesses[4 * c])
        for k in range(4):
            ref = oracle.detect_batch_par(frames[4 * c + k][None], fill, K, oracle.sigma(tw), True, radii, np.array([g], np.int32), separable=True)[0]
            assert tuple(int(v) for v in out[c][k]) == tuple(int(v) for v in ref), (win_w, c, k)
            g = tuple(int(v) for v in ref)
    bt.close()
