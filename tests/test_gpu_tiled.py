"""The tiled kernel (csrc/dog_tiled.hpp): ONE large search window per frame, cut into sub-windows with a workgroup each —
the latency path of a single clip whose window does not fit the fused kernel (reference functor
src/PawsomeTracker.jl:55-62, frame loop :163-169).  Checked against the oracle's dense Float64 statement of the reference
(PARITY UNPINNED: the oracle is this repo's restatement) and against the two-pass launches it replaces (PDOG_NO_TILED=1)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FILL = 128


@pytest.fixture(scope="module")
def pt():
    import pawsometracker_jl_amd as m
    return m


def _disc(frame, ci, cj, rad, val):
    h, w = frame.shape
    yy, xx = np.ogrid[0:h, 0:w]
    frame[(yy - ci) ** 2 + (xx - cj) ** 2 <= rad * rad] = val


def _clip(n, h, w, seed, rad=12, step=9):
    rng = np.random.Generator(np.random.PCG64(seed))
    frames = (FILL + rng.integers(-3, 4, (n, h, w))).astype(np.uint8)
    pos = []
    ci, cj = h // 2, w // 2
    for k in range(n):
        ci = int(np.clip(ci + rng.integers(-step, step + 1), 5, h - 5))
        cj = int(np.clip(cj + rng.integers(-step, step + 1), 5, w - 5))
        _disc(frames[k], ci, cj, rad, 5)
        pos.append((ci + 1, cj + 1))
    return frames, pos


@pytest.mark.parametrize("ws", [(129, 129), (257, 257), (161, 301)])
def test_functor_and_small_batches_vs_dense_oracle(pt, oracle, ws):
    """Host functor (one window) and batches of one and two windows through the tiled kernel; guesses near every frame
    border so that sub-windows hang over the padded frame (:48)."""
    import torch
    tw, h, w = 25, 360, 480
    frames, _ = _clip(6, h, w, seed=3)
    sig = oracle.sigma(tw)
    K = oracle.dog_kernel(sig, True)
    radii = (ws[0] // 2, ws[1] // 2)
    guesses = [(180, 240), (1, 1), (h, w), (20, 470), (350, 10), (200, 300)]
    FILL = oracle.mode_u8(frames[0])       # mode(_img) of the first frame (:47) — under ±3 noise not the background level
    want = [oracle.detect(frames[k], FILL, K, radii, guesses[k]) for k in range(6)]
    t = pt.Tracker(frames[0], tw, ws, True)
    assert t.img.fillvalue == FILL
    for k in range(6):
        t.img.data[...] = frames[k]
        assert t(guesses[k]) == want[k], (k, guesses[k])
    t.close()
    bt = pt.BatchTracker(h, w, tw, ws, True, FILL)
    assert bt.kernel_for_batch(1) == 400 and bt.kernel_for_batch(2) == 400, "the window should be served by the tiled kernel"
    d_f = torch.from_numpy(frames).cuda()
    for lo in (0, 2, 4):
        g = torch.tensor(guesses[lo:lo + 2], dtype=torch.int32).cuda()
        out, resp = bt.detect(d_f[lo:lo + 2], g, want_resp=True)
        bt.sync()
        assert [tuple(r) for r in out.cpu().numpy().tolist()] == want[lo:lo + 2]
        # the response map of the tiled kernel against the oracle's Float64 one (FP32 evaluation error only)
        for b in range(2):
            _, ref = oracle.detect(frames[lo + b], FILL, K, radii, guesses[lo + b], want_resp=True)
            got = resp[b].cpu().numpy().T
            assert np.abs(got - ref).max() <= 3e-5 * max(1.0, np.abs(ref).max() / 0.09), (lo + b, np.abs(got - ref).max())
    one = bt.detect(d_f[:1], torch.tensor(guesses[:1], dtype=torch.int32).cuda())
    bt.sync()
    assert tuple(one.cpu().numpy()[0].tolist()) == want[0]
    bt.close()


def test_chain_equals_oracle_chain_and_two_pass_launches(pt, oracle):
    """ij[k] = trckr(ij[k-1]) (:167) over a clip: one cooperative launch of the tiled kernel, positions equal to the
    oracle's serial chain, to the same library with the tiled kernel pinned off (stream-ordered two-pass launches), and to the
    progress variant that publishes every frame."""
    import torch
    tw, ws, h, w, n = 25, (257, 257), 540, 720, 40
    frames, _ = _clip(n, h, w, seed=11)
    sig = oracle.sigma(tw)
    K = oracle.dog_kernel(sig, True)
    radii = (ws[0] // 2, ws[1] // 2)
    want, g = [], (h // 2, w // 2)
    for k in range(n):
        g = oracle.detect(frames[k], FILL, K, radii, g)
        want.append(g)
    d_f = torch.from_numpy(frames).cuda()

    def chain(no_tiled=False, generic=False):
        bt = pt.BatchTracker(h, w, tw, ws, True, FILL)
        bt.set_tuning("no_tiled", int(no_tiled))
        bt.set_tuning("no_fused_c", int(generic))  # l = 65: the runtime-length instance instead of the compile-time-length one
        out = bt.detect_chain(d_f, (h // 2, w // 2))
        bt.sync()
        res = [tuple(r) for r in out.cpu().numpy().tolist()]
        cp = bt.detect_chain_progress(d_f, (h // 2, w // 2))
        prog = [tuple(r) for r in cp.wait().tolist()]
        cp.close()
        tiled = bt.kernel_for_batch(1) == 400
        bt.close()
        return res, prog, tiled

    res, prog, tiled = chain()
    assert tiled and res == want and prog == want
    res2, prog2, tiled2 = chain(no_tiled=True)
    assert not tiled2 and res2 == want and prog2 == want
    res3, prog3, tiled3 = chain(generic=True)
    assert tiled3 and res3 == want and prog3 == want


def test_ties_across_sub_window_borders_and_refinement(pt, oracle):
    """A flat frame ties every response: the answer is the first pixel in column-major order (:59) whichever
    sub-window it lies in.  Two equal blobs in different sub-windows are a near-tie the FP32 ranking cannot settle:
    exact mode must flag the window and return the dense oracle's position."""
    import torch
    tw, ws, h, w = 25, (257, 257), 400, 400
    sig = oracle.sigma(tw)
    K = oracle.dog_kernel(sig, True)
    radii = (128, 128)
    flat = np.full((1, h, w), FILL, np.uint8)
    bt = pt.BatchTracker(h, w, tw, ws, True, FILL)
    g = torch.tensor([[200, 200]], dtype=torch.int32).cuda()
    out = bt.detect(torch.from_numpy(flat).cuda(), g)
    bt.sync()
    assert tuple(out.cpu().numpy()[0].tolist()) == oracle.detect(flat[0], FILL, K, radii, (200, 200)) == (72, 72)
    rng = np.random.Generator(np.random.PCG64(2))
    frames = np.full((8, h, w), FILL, np.uint8)
    guesses = np.empty((8, 2), np.int32)
    for b in range(8):
        di, dj = int(rng.integers(40, 100)), int(rng.integers(-90, 91))
        _disc(frames[b], 200 - di, 200 - dj, 12, 0)
        _disc(frames[b], 200 + di, 200 + dj, 12, int(rng.integers(0, 2)))
        guesses[b] = (201, 201)
    before = bt.exact_stats()[2]
    for b in range(8):
        out = bt.detect(torch.from_numpy(frames[b:b + 1]).cuda(), torch.from_numpy(guesses[b:b + 1]).cuda())
        bt.sync()
        assert tuple(out.cpu().numpy()[0].tolist()) == oracle.detect(frames[b], FILL, K, radii, tuple(guesses[b])), b
    assert bt.exact_stats()[2] > before     # the near-ties did go through the refinement inside the tiled kernel
    out = bt.detect_chain(torch.from_numpy(frames).cuda(), (201, 201))   # … and inside a chain: the refined answer is the next guess
    bt.sync()
    want, gg = [], (201, 201)
    for b in range(8):
        gg = oracle.detect(frames[b], FILL, K, radii, gg)
        want.append(gg)
    assert [tuple(r) for r in out.cpu().numpy().tolist()] == want
    bt.close()


def test_seeded_fuzz_large_windows_vs_oracle(pt, oracle):
    """Seeded random configurations in the tiled kernel's territory — windows 60…300 rows/columns (rectangular, odd and even),
    target widths 10…50 (l = 29…125), frames smaller and larger than the window, bright and dark targets, guesses up to
    l÷2 outside the frame — through the functor, a two-window batch and a three-frame chain, against the dense oracle."""
    import torch
    from oracle import synth
    rng = np.random.default_rng(20261004)
    ran = 0
    for case in range(40):
        fh, fw = int(rng.integers(60, 400)), int(rng.integers(60, 500))
        tw = float(rng.choice([10, 25, 25, 33, 40, 50]))
        ws = (int(rng.integers(60, 301)), int(rng.integers(60, 301)))
        darker = bool(rng.integers(0, 2))
        radii = (ws[0] // 2, ws[1] // 2)
        l = oracle.kernel_len(oracle.sigma(tw))
        hw = l // 2
        if (2 * radii[0] + 1) * (2 * radii[1] + 1) * l * l > 1.2e9:      # keep the dense oracle affordable
            continue
        frames = rng.integers(110, 146, (3, fh, fw)).astype(np.uint8)
        for b in range(3):
            disc = synth.disc_frame(fh, fw, (int(rng.integers(1, fh + 1)), int(rng.integers(1, fw + 1))), max(2, int(tw)), darker)
            mask = disc != 128
            frames[b][mask] = disc[mask]
        guesses = np.stack([rng.integers(-hw, fh + hw + 2, 3), rng.integers(-hw, fw + hw + 2, 3)], 1).astype(np.int32)
        fill = oracle.mode_u8(frames[0])
        K = oracle.dog_kernel(oracle.sigma(tw), darker)
        tag = (case, fh, fw, tw, ws, darker)
        bt = pt.BatchTracker(fh, fw, tw, ws, darker, fill)
        if bt.kernel_for_batch(1) != 400:
            bt.close()
            continue                                                   # small enough for the fused kernel, or no sub-window fits LDS
        ran += 1
        exp = [oracle.detect(frames[b], fill, K, radii, tuple(guesses[b])) for b in range(3)]
        got = bt.detect(torch.from_numpy(frames[:2]).cuda(), torch.from_numpy(guesses[:2]).cuda()).cpu().numpy()
        assert [tuple(r) for r in got.tolist()] == exp[:2], ("batch",) + tag
        g0 = (int(guesses[2, 0]), int(guesses[2, 1]))
        chain = bt.detect_chain(torch.from_numpy(np.repeat(frames[2:3], 3, 0)).cuda(), g0).cpu().numpy()
        g = g0
        for k in range(3):
            g = oracle.detect(frames[2], fill, K, radii, g)
            assert tuple(int(v) for v in chain[k]) == g, ("chain", k) + tag
        bt.close()
        tr = pt.Tracker(frames[0], tw, ws, darker)
        tr.img.data[...] = frames[1]
        assert tr((int(guesses[1, 0]), int(guesses[1, 1]))) == exp[1], ("functor",) + tag
        tr.close()
    assert ran >= 15


def test_few_clips_share_one_cooperative_launch(pt, oracle):
    """pdog_detect_chains with as many clips as the device keeps resident (257×257 windows: 81 workgroups per clip, three
    clips on 256 CUs) is ONE launch of the tiled kernel; a fourth clip falls back to the per-frame launches.  Every clip's
    positions equal the oracle's serial chain."""
    import torch
    tw, ws, h, w, nf = 25, (257, 257), 420, 560, 4
    sig = oracle.sigma(tw)
    K = oracle.dog_kernel(sig, True)
    radii = (128, 128)
    for n_clips in (3, 4):
        clips, want, starts = [], [], []
        for c in range(n_clips):
            frames, _ = _clip(nf, h, w, seed=100 + 10 * n_clips + c)
            clips.append(frames)
            g = (h // 2 + 3 * c, w // 2 - 5 * c)
            starts.append(g)
            chain = []
            for k in range(nf):
                g = oracle.detect(frames[k], FILL, K, radii, g)
                chain.append(g)
            want.append(chain)
        bt = pt.BatchTracker(h, w, tw, ws, True, FILL)
        out = bt.detect_chains(torch.from_numpy(np.stack(clips)).cuda(), torch.tensor(starts, dtype=torch.int32).cuda())
        bt.sync()
        got = out.cpu().numpy()
        for c in range(n_clips):
            assert [tuple(r) for r in got[c].tolist()] == want[c], (n_clips, c)
        bt.close()


def test_device_side_wait_gives_up_promptly_and_recovers(pt, oracle):
    """The tiled kernel's workgroups wait for each other between the frames of a clip (wait_counter, dog_kernels.hpp).  A peer
    that never arrives must not hang the GPU: the wait is bounded by wall time (≈1 s of s_memrealtime), the first wave to give
    up raises an abort word every other wait polls — so the whole launch ends about one timeout after the stall, not one
    timeout per waiting workgroup and frame — nothing is published after the fault, pdog_sync reports PDOG_E_HIP once and
    zeroes the control words, and the same tracker then walks the same clip correctly.  The stall is injected
    (pdog_set_tuning "fault_inject": sub-window 0 never delivers the partial of its second frame)."""
    import time
    import torch
    tw, ws, h, w, n = 25, (257, 257), 540, 720, 6
    frames, _ = _clip(n, h, w, seed=21)
    K = oracle.dog_kernel(oracle.sigma(tw), True)
    radii = (ws[0] // 2, ws[1] // 2)
    want, g = [], (h // 2, w // 2)
    for k in range(n):
        g = oracle.detect(frames[k], FILL, K, radii, g)
        want.append(g)
    d_f = torch.from_numpy(frames).cuda()
    bt = pt.BatchTracker(h, w, tw, ws, True, FILL)
    assert bt.kernel_for_batch(1) == 400
    bt.set_tuning("fault_inject", 1)
    t0 = time.perf_counter()
    cp = bt.detect_chain_progress(d_f, (h // 2, w // 2))
    with pytest.raises(pt.PdogError) as e:
        bt.sync()
    dt = time.perf_counter() - t0
    assert e.value.code == pt._lib.PDOG_E_HIP and "watchdog" in str(e.value)
    assert dt < 4.0, dt                      # one timeout (≈1 s), not one per workgroup and frame
    assert cp.done() <= 1                    # frame 0 finished before the stall; nothing was published after it
    cp.close()
    bt.sync()                                # reported once
    bt.set_tuning("fault_inject", 0)
    out = bt.detect_chain(d_f, (h // 2, w // 2))
    bt.sync()
    assert [tuple(r) for r in out.cpu().numpy().tolist()] == want
    bt.close()
