"""Parity tests proper: the HIP path, called through the C ABI (ctypes), against the CPU oracle
on the same inputs.  Bar: positions bit-exact; DoG response within 1e-5 relative Float32,
where "relative" is max|gpu - ref| / max|ref| over the window (BASELINE.json north_star;
pointwise relative error is meaningless where the response crosses zero).
PARITY UNPINNED: the oracle is our restatement, see oracle/dog_oracle.c."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RESP_RTOL = 1e-5       # relative to max|ref| over the window
RESP_ATOL = 1e-7       # for windows whose reference response is ~0 everywhere (flat: exact ties)


@pytest.fixture(scope="module")
def pt():
    import torch
    assert torch.cuda.is_available(), "gpu tests need a device"
    import pawsometracker_jl_amd as m
    return m


def _resp_err(got, ref):
    scale = np.abs(ref).max()
    err = np.abs(got.astype(np.float64) - ref).max()
    return err, scale


def _check_resp(got, ref, name):
    err, scale = _resp_err(got, ref)
    if scale < 1e-6:
        assert err <= RESP_ATOL, (name, err, scale)
    else:
        assert err / scale <= RESP_RTOL, (name, err / scale)


def test_golden_vectors_through_tracker(pt, golden):
    for c in golden:
        t = pt.Tracker(c["frame"], c["tw"], c["ws"], c["darker"])
        assert t.img.fillvalue == c["fill"]
        ij, resp = t(c["guess"], want_resp=True)
        assert ij == c["ij"], (c["name"], ij, c["ij"])
        _check_resp(resp, c["resp"], c["name"])
        assert t(c["guess"]) == c["ij"]          # the no-response kernel instantiation
        t.close()


@pytest.mark.parametrize("variant", [0, 1, 2, 10, 11, 12, 13, 14, 100, 200, 300])
def test_every_l65_variant(pt, golden, variant):
    n_run = 0
    for c in golden:
        if c["l"] != 65:
            continue
        t = pt.Tracker(c["frame"], c["tw"], c["ws"], c["darker"])
        try:
            t.set_variant(variant)
        except pt.PdogError:
            assert variant == 300   # the fused kernel needs the whole padded tile in LDS
            t.close()
            continue
        n_run += 1
        assert t.info().variant == variant
        ij, resp = t(c["guess"], want_resp=True)
        assert ij == c["ij"], (c["name"], variant, ij, c["ij"])
        _check_resp(resp, c["resp"], c["name"])
        assert t(c["guess"]) == c["ij"]
        t.close()
    assert n_run >= 3


def test_golden_via_the_copying_host_path(pt, golden):
    """The functor's fallback for windows that do not fit the fused kernel (tile uploaded with copy commands, batch
    kernels) on the golden cases that normally take the in-place path."""
    for c in golden:
        t = pt.Tracker(c["frame"], c["tw"], c["ws"], c["darker"])
        t.set_tuning("host_copy", 1)
        assert t(c["guess"]) == c["ij"], c["name"]
        t.close()


@pytest.mark.parametrize("variant", [0, 1, 2, 20, 200, 300])
def test_other_kernel_lengths(pt, golden, variant):
    for c in golden:
        if c["l"] == 65 or (variant == 20 and c["l"] != 29):
            continue
        t = pt.Tracker(c["frame"], c["tw"], c["ws"], c["darker"])
        try:
            t.set_variant(variant)
        except pt.PdogError:
            t.close()
            continue        # this variant's LDS footprint does not fit this kernel length
        ij, resp = t(c["guess"], want_resp=True)
        assert ij == c["ij"], (c["name"], variant, ij, c["ij"])
        _check_resp(resp, c["resp"], c["name"])
        t.close()


def _batch(pt, frames, guesses, tw, ws, darker, fill, frame_index=None, want_resp=False, variant=None, tuning=None):
    import torch
    bt = pt.BatchTracker(frames.shape[1], frames.shape[2], tw, ws, darker, fill)
    if variant is not None:
        bt.set_variant(variant)
    for key, val in (tuning or {}).items():
        bt.set_tuning(key, val)
    bt.use_torch_stream()
    d_frames = torch.from_numpy(frames).cuda()
    d_guess = torch.from_numpy(np.ascontiguousarray(guesses, np.int32)).cuda()
    d_fi = torch.from_numpy(np.ascontiguousarray(frame_index, np.int32)).cuda() if frame_index is not None else None
    out = bt.detect(d_frames, d_guess, d_fi, want_resp=want_resp)
    torch.cuda.synchronize()
    if want_resp:
        res = out[0].cpu().numpy(), out[1].cpu().numpy()
    else:
        res = out.cpu().numpy()
    bt.close()
    return res


def test_seeded_batch_vs_oracle(pt, oracle):
    from oracle import synth
    for tw, ws, darker, noise, seed in ((25, (45, 45), True, 3, 0), (25, (64, 96), False, 2, 1), (10, (21, 21), True, 3, 2)):
        radii = (ws[0] // 2, ws[1] // 2)
        frames, guesses, _ = synth.make_batch(24, 240, 320, tw, radii, darker, seed=seed, noise=noise)
        fill = oracle.mode_u8(frames[0])
        K = oracle.dog_kernel(oracle.sigma(tw), darker)
        ref = oracle.detect_batch(frames, fill, K, radii, guesses)
        got, resp = _batch(pt, frames, guesses, tw, ws, darker, fill, want_resp=True)
        assert np.array_equal(got, ref), (tw, ws, np.flatnonzero((got != ref).any(1)))
        for b in (0, 7, 23):
            _, r = oracle.detect(frames[b], fill, K, radii, guesses[b], want_resp=True)
            _check_resp(resp[b].T, r, f"batch{b}")


def test_frame_index_many_windows_per_frame(pt, oracle):
    from oracle import synth
    tw, ws = 25, (45, 45)
    frames, _, centres = synth.make_batch(3, 240, 320, tw, (22, 22), True, seed=5, noise=3)
    rng = np.random.default_rng(7)
    fi = rng.integers(0, 3, 40).astype(np.int32)
    guesses = np.stack([rng.integers(1, 241, 40), rng.integers(1, 321, 40)], 1).astype(np.int32)
    fill = oracle.mode_u8(frames[0])
    K = oracle.dog_kernel(oracle.sigma(tw), True)
    ref = np.array([oracle.detect(frames[fi[b]], fill, K, (22, 22), guesses[b]) for b in range(40)], np.int32)
    got = _batch(pt, frames, guesses, tw, ws, True, fill, frame_index=fi)
    assert np.array_equal(got, ref)


def test_serial_chain_matches_oracle_chain(pt, oracle):
    import torch
    from oracle import synth
    from oracle.dog_oracle import OracleTracker
    tw, h, w = 10, 100, 100            # the reference test defaults (test/test-basic-test.jl:1-10)
    rng = np.random.default_rng(3)
    pos = np.cumsum(rng.integers(-4, 5, (60, 2)), 0) + 50
    pos = np.clip(pos, 8, 92)
    frames = np.stack([synth.disc_frame(h, w, (int(p[0]), int(p[1])), tw, True) for p in pos])
    ot = OracleTracker(frames[0], tw, (21, 21), True, oracle)
    ref = [ot((50, 50))]
    for f in frames[1:]:
        ot.data[...] = f
        ref.append(ot(ref[-1]))
    # device chain (pdog_detect_chain)
    bt = pt.BatchTracker(h, w, tw, (21, 21), True, ot.fill)
    out = bt.detect_chain(torch.from_numpy(frames).cuda(), (50, 50))
    bt.sync()
    assert [tuple(int(v) for v in r) for r in out.cpu().numpy()] == ref
    bt.close()
    # host mirror of the reference loop (Tracker + trckr.img.data, :166-167)
    got = pt.track_frames(frames, target_width=tw, start_location=("ij", (50, 50)), window_size=21)
    assert got == ref
    assert ref == [(int(p[0]), int(p[1])) for p in pos]        # and it actually tracks the disc


def test_auto_detect_bootstrap(pt, oracle):
    # start_location === missing: window_size2 = sz .÷ 4 around the frame centre (:99-107)
    from oracle import synth
    h, w, tw = 240, 320, 25
    f = synth.disc_frame(h, w, (130, 170), tw, True)
    trckr, ij = pt.get_start_ij_and_tracker(None, f, tw, (45, 45), True)
    K = oracle.dog_kernel(oracle.sigma(tw), True)
    ref = oracle.detect(f, 128, K, ((h // 4) // 2, (w // 4) // 2), (h // 2, w // 2))
    assert ij == ref == (130, 170)
    assert trckr.radii == (22, 22)
    trckr.close()


def test_edge_cases(pt, oracle):
    from pawsometracker_jl_amd import _lib
    flat = np.full((64, 80), 77, np.uint8)
    # 1x1 window (window_size 1 -> radii 0): the answer is the guess, clamped
    t = pt.Tracker(flat, 25, (1, 1), True)
    assert t((10, 20)) == (10, 20) and t((0, 81)) == (1, 80)
    t.close()
    # window larger than the frame, flat frame: top-left of the window, clamped -> (1, 1)
    t = pt.Tracker(flat, 10, (201, 301), True)
    assert t((32, 40)) == (1, 1)
    t.close()
    # guess further outside than the reference's pad allows -> PDOG_E_RANGE (reference: BoundsError)
    t = pt.Tracker(flat, 25, (45, 45), True)
    l = t.info().kernel_len
    assert t((-(l // 2), 10)) == (1, 1)                  # last legal row
    with pytest.raises(pt.PdogError) as e:
        t((-(l // 2) - 1, 10))
    assert e.value.code == _lib.PDOG_E_RANGE
    t.close()
    # empty batch is a no-op
    import torch
    bt = pt.BatchTracker(64, 80, 25, (45, 45), True, 77)
    out = bt.detect(torch.zeros((1, 64, 80), dtype=torch.uint8, device="cuda"), torch.zeros((0, 2), dtype=torch.int32, device="cuda"))
    assert out.shape == (0, 2)
    bt.close()


def test_full_size_properties_1080p(pt, oracle):
    """BASELINE config 3 geometry (1080p, window 256 -> 257x257, tw=25) at a reduced batch:
    size-independent properties + a few windows against the dense oracle."""
    from oracle import synth
    tw, ws, n = 25, (256, 256), 48
    radii = (128, 128)
    frames, guesses, centres = synth.make_batch(n, 1080, 1920, tw, radii, True, seed=11, noise=0)
    fill = 128
    got = _batch(pt, frames, guesses, tw, ws, True, fill)
    # (i) noise-free disc fully inside frame and window -> exactly the disc centre
    inside = ((centres[:, 0] > 13) & (centres[:, 0] < 1080 - 13) & (centres[:, 1] > 13) & (centres[:, 1] < 1920 - 13)
              & (np.abs(centres - guesses) <= 128 - 13).all(1))
    assert inside.sum() > n // 2
    assert np.array_equal(got[inside], centres[inside])
    # (iii) adding a constant to frame and fill changes nothing (sum K = 0)
    got2 = _batch(pt, (frames.astype(np.int16) + 50).astype(np.uint8), guesses, tw, ws, True, fill + 50)
    assert np.array_equal(got, got2)
    # (iv) complement + bright target
    got3 = _batch(pt, 255 - frames, guesses, tw, ws, False, 255 - fill)
    assert np.array_equal(got, got3)
    # (ii) flat frames -> window top-left, clamped
    flat = np.full((2, 1080, 1920), 128, np.uint8)
    g = np.array([[540, 960], [50, 1900]], np.int32)
    assert np.array_equal(_batch(pt, flat, g, tw, ws, True, 128), np.array([[412, 832], [1, 1772]], np.int32))
    # every compiled l=65 variant gives the same positions
    for v in (10, 11, 12, 13, 14, 2, 100, 200):
        assert np.array_equal(_batch(pt, frames, guesses, tw, ws, True, fill, variant=v), got), v
    # noisy frames: a sample of windows against the dense Float64 oracle (279 M MAC each)
    nf, ng, _ = synth.make_batch(4, 1080, 1920, tw, radii, True, seed=12, noise=3)
    fill_n = oracle.mode_u8(nf[0])
    K = oracle.dog_kernel(oracle.sigma(tw), True)
    gotn, resp = _batch(pt, nf, ng, tw, ws, True, fill_n, want_resp=True)
    for b in range(4):
        ij, r = oracle.detect(nf[b], fill_n, K, radii, ng[b], want_resp=True)
        assert tuple(int(v) for v in gotn[b]) == ij
        _check_resp(resp[b].T, r, f"1080p{b}")


def test_random_geometries_vs_oracle(pt, oracle):
    """Seeded random frame sizes / window sizes / guesses (borders included) for the three kernel
    lengths the golden set exercises most: every strip layout of the roll kernel (partial strip,
    overlapping last strip, thin remainder columns 1..6) and the ring kernels' chunk/strip edges."""
    from oracle import synth
    rng = np.random.default_rng(20260104)
    cases = []
    for wcols in (1, 7, 63, 64, 65, 66, 70, 71, 100, 127, 128, 129, 134, 135, 193):   # window widths 2r+1 are odd: use r
        cases.append((25, int(rng.integers(3, 60)) | 1, wcols | 1))
    for _ in range(6):
        cases.append((int(rng.choice([10, 16, 25])), int(rng.integers(1, 90)) | 1, int(rng.integers(1, 150)) | 1))
    # every roll-kernel instance (l = 17 … 77) plus a runtime-l length, on strip layouts with and without remainder
    for tw in (5, 7, 8, 10, 12, 13, 15, 17, 18, 20, 22, 23, 25, 27, 28, 30, 33):
        cases.append((tw, int(rng.integers(20, 80)) | 1, int(rng.choice([45, 64, 67, 131]))| 1))
    for tw, wh, ww in cases:
        h, w = int(rng.integers(40, 200)), int(rng.integers(60, 260))
        centre = (int(rng.integers(1, h + 1)), int(rng.integers(1, w + 1)))
        f = synth.disc_frame(h, w, centre, tw, True)
        f = np.clip(f.astype(np.int16) + rng.integers(-4, 5, f.shape), 0, 255).astype(np.uint8)
        guess = (int(np.clip(centre[0] + rng.integers(-wh // 3 - 1, wh // 3 + 2), -5, h + 6)),
                 int(np.clip(centre[1] + rng.integers(-ww // 3 - 1, ww // 3 + 2), -5, w + 6)))
        fill = oracle.mode_u8(f)
        K = oracle.dog_kernel(oracle.sigma(tw), True)
        ref_ij, ref = oracle.detect(f, fill, K, (wh // 2, ww // 2), guess, want_resp=True)
        t = pt.Tracker(f, tw, (wh, ww), True)
        ij, resp = t(guess, want_resp=True)
        assert ij == ref_ij, (tw, wh, ww, h, w, guess, ij, ref_ij, t.info().variant)
        _check_resp(resp, ref, f"rand tw={tw} win=({wh},{ww}) frame=({h},{w})")
        assert t(guess) == ref_ij
        t.close()


def test_persistent_multi_clip_chains(pt, oracle):
    """pdog_detect_chains: several clips, each the serial chain of :163-169, in one persistent launch
    (one wave per strip, one workgroup per clip) — against the oracle's chain, for a one-strip and a
    multi-strip search window (the latter covers the overlapping last strip and the LDS exchange)."""
    import torch
    from oracle import synth
    from oracle.dog_oracle import OracleTracker
    rng = np.random.default_rng(11)
    for (h, w, tw, ws, nclips, nf) in ((120, 160, 25, (45, 45), 3, 12), (200, 260, 25, (90, 150), 2, 6), (100, 100, 10, (21, 21), 2, 10),
                                        (150, 200, 40, (61, 61), 1, 4)):
        clips, starts, refs = [], [], []
        for c in range(nclips):
            pos = np.cumsum(rng.integers(-5, 6, (nf, 2)), 0) + np.array([h // 2, w // 2])
            pos = np.clip(pos, 15, [h - 15, w - 15])
            fr = np.stack([synth.disc_frame(h, w, (int(p[0]), int(p[1])), tw, True) for p in pos])
            fr = np.clip(fr.astype(np.int16) + rng.integers(-3, 4, fr.shape), 0, 255).astype(np.uint8)
            clips.append(fr)
            starts.append((int(pos[0][0]) + 3, int(pos[0][1]) - 4))
        fill = oracle.mode_u8(clips[0][0])
        for c in range(nclips):
            ot = OracleTracker(clips[c][0], tw, ws, True, oracle)
            ot.fill = fill
            r = [ot(starts[c])]
            for f in clips[c][1:]:
                ot.data[...] = f
                r.append(ot(r[-1]))
            refs.append(r)
        d_clips = torch.from_numpy(np.stack(clips)).cuda()
        d_starts = torch.tensor(starts, dtype=torch.int32).cuda()
        # default: few clips run as small per-frame batches; with the kernel pinned (set_variant) the
        # persistent one-launch chain kernel runs — both must reproduce the oracle's chains
        for pin in (False, True, 300):
            bt = pt.BatchTracker(h, w, tw, ws, True, fill)
            if pin is True:
                if bt.info().variant < 100 or bt.info().variant == 200:
                    bt.close()
                    continue      # no roll instance (hence no persistent kernel) for this kernel length
                bt.set_variant(bt.info().variant)
            elif pin == 300:
                try:
                    bt.set_variant(300)   # fused kernel: one workgroup per clip, the frame loop inside the kernel
                except pt.PdogError:
                    bt.close()
                    continue
            out = bt.detect_chains(d_clips, d_starts)
            bt.sync()
            got = out.cpu().numpy()
            for c in range(nclips):
                assert [tuple(int(v) for v in r) for r in got[c]] == refs[c], (ws, c, pin)
            bt.close()


def test_extreme_parameters_do_not_break(pt, oracle):
    """Unusual but legal parameters: tiny and huge target widths, windows larger than the frame,
    one-pixel frames' worth of window, very wide windows — positions against the oracle where the
    dense oracle is affordable, otherwise the known answers (flat -> window top-left clamped)."""
    from oracle import synth
    rng = np.random.default_rng(5)
    # small enough for the dense oracle
    for tw, ws, (h, w) in ((1, (5, 5), (40, 50)), (2, (9, 7), (40, 50)), (3, (11, 11), (30, 30)), (200, (9, 9), (64, 64)),
                           (60, (31, 33), (90, 120)), (25, (3, 301), (60, 400)), (25, (201, 3), (300, 50))):
        f = rng.integers(0, 256, (h, w), dtype=np.uint8)
        fill = oracle.mode_u8(f)
        K = oracle.dog_kernel(oracle.sigma(tw), True)
        guess = (int(rng.integers(1, h + 1)), int(rng.integers(1, w + 1)))
        ref_ij, ref = oracle.detect(f, fill, K, (ws[0] // 2, ws[1] // 2), guess, want_resp=True)
        t = pt.Tracker(f, tw, ws, True)
        ij, resp = t(guess, want_resp=True)
        assert ij == ref_ij, (tw, ws, ij, ref_ij, t.info().variant)
        _check_resp(resp, ref, f"extreme tw={tw} ws={ws}")
        t.close()
    # too big for the dense oracle: flat frames have a known answer
    flat = np.full((1080, 1920), 200, np.uint8)
    for tw, ws, guess in ((25, (1001, 1921), (540, 960)), (120, (401, 401), (300, 300)), (25, (2161, 11), (10, 10))):
        t = pt.Tracker(flat, tw, ws, True)
        exp = (max(1, guess[0] - ws[0] // 2), max(1, guess[1] - ws[1] // 2))
        assert t(guess) == exp, (tw, ws, t.info().variant)
        t.close()


def test_device_mode_matches_statsbase_rule(pt, oracle):
    """pdog_mode_u8_device against the oracle's literal StatsBase scan, including engineered count ties."""
    import torch
    rng = np.random.default_rng(2)
    imgs = [rng.integers(0, 256, (37, 53), dtype=np.uint8), rng.integers(0, 3, (64, 64), dtype=np.uint8),
            np.full((20, 30), 7, np.uint8), rng.integers(100, 104, (1080, 1920), dtype=np.uint8)]
    tie = np.zeros((4, 6), np.uint8)           # 12 x value 0 ... make exact ties between 1, 2 and 3
    tie[:, 0] = 1; tie[:, 1] = 2; tie[:, 2] = 3; tie[:, 3] = [2, 1, 3, 9]; tie[:, 4] = [3, 2, 1, 9]; tie[:, 5] = 9
    imgs.append(tie)
    for _ in range(20):                        # small alphabets force ties
        imgs.append(rng.integers(0, 4, (int(rng.integers(2, 9)), int(rng.integers(2, 9))), dtype=np.uint8))
    for img in imgs:
        assert pt.mode_device(torch.from_numpy(img).cuda()) == oracle.mode_u8(img), img.shape
    view = torch.from_numpy(rng.integers(0, 5, (50, 80), dtype=np.uint8)).cuda()[:, 10:47]   # strided rows
    assert pt.mode_device(view) == oracle.mode_u8(view.cpu().numpy())


def test_two_pass_scratch_chunking(pt, oracle, monkeypatch):
    """The two-pass path keeps its intermediate in an HBM scratch buffer and walks large batches in chunks;
    force a tiny scratch so that a 40-window batch takes many chunks."""
    from oracle import synth
    tw, ws, radii = 40, (61, 61), (30, 30)
    frames, guesses, _ = synth.make_batch(40, 160, 200, tw, radii, True, seed=3, noise=3)
    fill = oracle.mode_u8(frames[0])
    K = oracle.dog_kernel(oracle.sigma(tw), True)
    ref = oracle.detect_batch(frames, fill, K, radii, guesses)
    monkeypatch.setenv("PDOG_SCRATCH_MB", "1")          # 61 x 161 x 8 B = 79 KB per window -> 13 windows per chunk
    assert np.array_equal(_batch(pt, frames, guesses, tw, ws, True, fill), ref)
    monkeypatch.delenv("PDOG_SCRATCH_MB")
    assert np.array_equal(_batch(pt, frames, guesses, tw, ws, True, fill), ref)
    # up to 16 windows the two-pass path runs as two launches (DC level inside the row pass, strip combine by the last
    # column-pass workgroup); pdog_set_tuning("twopass_4l") forces the four-launch form on the same windows.  Twice per tracker:
    # the per-window counters must be back at zero after a launch.
    import torch
    for n in (1, 5, 16, 17):
        for four in (False, True):
            bt = pt.BatchTracker(160, 200, tw, ws, True, fill)
            bt.set_tuning("twopass_4l", int(four))
            d_f, d_g = torch.from_numpy(frames[:n]).cuda(), torch.from_numpy(guesses[:n]).cuda()
            for _ in range(2):
                got, resp = bt.detect(d_f, d_g, want_resp=True)
                assert np.array_equal(got.cpu().numpy(), ref[:n]), (n, four)
            _, r0 = oracle.detect(frames[0], fill, K, radii, guesses[0], want_resp=True)
            _check_resp(resp[0].cpu().numpy().T, r0, f"two-pass n={n} four={four}")
            bt.close()


def test_trackers_with_different_geometry_coexist(pt, oracle):
    """Several live trackers share the compiled kernels (and their per-function dynamic-LDS limit): a small
    tracker created later must not break a large one created earlier (multi-video use, README.md:214 of the
    reference: concurrent `track` calls own separate Trackers)."""
    from oracle import synth
    rng = np.random.default_rng(9)
    big = np.clip(synth.disc_frame(400, 500, (200, 250), 120, True).astype(np.int16) + rng.integers(-3, 4, (400, 500)), 0, 255).astype(np.uint8)
    small = np.clip(synth.disc_frame(120, 160, (60, 80), 25, True).astype(np.int16) + rng.integers(-3, 4, (120, 160)), 0, 255).astype(np.uint8)
    tb = pt.Tracker(big, 120, (151, 151), True)       # two-pass kernels, large LDS rows
    first = tb((195, 258))
    ts = pt.Tracker(small, 25, (45, 45), True)         # same kernels at launch time (small batch), small LDS rows
    assert ts((55, 85)) == (60, 80)
    tw = pt.Tracker(small, 25, (257, 257), True)       # roll kernel family set up too
    assert tw((55, 85)) == (60, 80)
    assert tb((195, 258)) == first == (200, 250)
    for t in (tb, ts, tw):
        t.close()


def test_textured_frames_batch_vs_oracle(pt, oracle):
    """Frames that look like video rather than like the reference's flat test clips: smooth large-scale
    texture + sensor noise + a dark blob, so the window level sits far from the frame's mode (exercises the
    per-window DC level) and the peak competes with texture.  200 windows, positions exact."""
    from scipy.ndimage import gaussian_filter
    rng = np.random.default_rng(77)
    h, w, tw, ws, n = 360, 480, 25, (45, 45), 200
    base = gaussian_filter(rng.standard_normal((h, w)), 25)
    base = (base - base.min()) / (base.max() - base.min())            # 0..1 smooth texture
    frames = np.empty((8, h, w), np.uint8)
    centres = []
    for k in range(8):
        img = 60 + 150 * base + rng.normal(0, 4, (h, w))
        c = (int(rng.integers(30, h - 30)), int(rng.integers(30, w - 30)))
        yy, xx = np.ogrid[:h, :w]
        img[(yy - c[0]) ** 2 + (xx - c[1]) ** 2 <= 144] -= 70          # dark blob, radius 12
        frames[k] = np.clip(img, 0, 255).astype(np.uint8)
        centres.append(c)
    fi = rng.integers(0, 8, n).astype(np.int32)
    guesses = np.stack([rng.integers(1, h + 1, n), rng.integers(1, w + 1, n)], 1).astype(np.int32)
    near = rng.random(n) < 0.5                                          # half of the windows contain the blob
    for b in np.flatnonzero(near):
        guesses[b] = np.array(centres[fi[b]]) + rng.integers(-12, 13, 2)
    fill = oracle.mode_u8(frames[0])
    K = oracle.dog_kernel(oracle.sigma(tw), True)
    ref = np.array([oracle.detect(frames[fi[b]], fill, K, (22, 22), guesses[b]) for b in range(n)], np.int32)
    got, resp = _batch(pt, frames, guesses, tw, ws, True, fill, frame_index=fi, want_resp=True)
    assert np.array_equal(got, ref), np.flatnonzero((got != ref).any(1))
    for b in (0, 50, 199):
        _, r = oracle.detect(frames[fi[b]], fill, K, (22, 22), guesses[b], want_resp=True)
        _check_resp(resp[b].T, r, f"textured{b}")
    # large batch path (roll kernel) on the same data: 4 x the windows so that the batch is not "small"
    big = np.tile(guesses, (6, 1)); bfi = np.tile(fi, 6)
    got_big = _batch(pt, frames, big, tw, ws, True, fill, frame_index=bfi)
    assert np.array_equal(got_big, np.tile(ref, (6, 1)))


@pytest.mark.parametrize("cfg", ["cfg2", "cfg4", "cfg5"])
def test_full_size_properties_other_configs(pt, oracle, cfg):
    """The remaining BASELINE.json geometries at full frame/window size and a reduced batch: known answers
    that need no oracle (SURVEY §8c i–iv), for a batch large enough to take the batch kernels and for a
    single window (the small-batch kernels)."""
    from oracle import synth
    fh, fw, tw, ws = {"cfg2": (1080, 1920, 25, (270, 480)), "cfg4": (2160, 3840, 25, (512, 512)),
                      "cfg5": (1080, 1920, 120, (205, 205))}[cfg]
    radii = (ws[0] // 2, ws[1] // 2)
    n = 6
    frames, guesses, centres = synth.make_batch(n, fh, fw, tw, radii, True, seed=21, noise=0)
    rad = tw // 2 + 1
    inside = ((centres[:, 0] > rad) & (centres[:, 0] <= fh - rad) & (centres[:, 1] > rad) & (centres[:, 1] <= fw - rad)
              & (np.abs(centres - guesses) <= np.array(radii) - rad).all(1))
    assert inside.any()
    # replicate the batch so that it is not a "small" one (≥ 1000 strip-waves) and compare with the small path
    rep = 1 + 1000 // max(1, (2 * radii[1] + 1) // 64)
    fi = np.tile(np.arange(n, dtype=np.int32), rep)
    got_big = _batch(pt, frames, np.tile(guesses, (rep, 1)), tw, ws, True, 128, frame_index=fi)
    got_small = _batch(pt, frames, guesses, tw, ws, True, 128)
    assert np.array_equal(got_big[:n], got_small) and np.array_equal(got_big, np.tile(got_small, (rep, 1)))
    assert np.array_equal(got_small[inside], centres[inside])                         # (i) disc centre, exactly
    shifted = _batch(pt, (frames.astype(np.int16) + 40).astype(np.uint8), guesses, tw, ws, True, 168)
    assert np.array_equal(shifted, got_small)                                         # (iii) + constant
    bright = _batch(pt, 255 - frames, guesses, tw, ws, False, 127)
    assert np.array_equal(bright, got_small)                                          # (iv) complement
    flat = np.full((1, fh, fw), 128, np.uint8)
    g = np.array([[fh // 2, fw // 2]], np.int32)
    exp = np.array([[max(1, fh // 2 - radii[0]), max(1, fw // 2 - radii[1])]], np.int32)
    assert np.array_equal(_batch(pt, flat, g, tw, ws, True, 128), exp)                # (ii) flat window


@pytest.mark.parametrize("tw,ws,chunk", [(25, (256, 256), 64), (25, (45, 45), 0), (10, (21, 33), 16), (120, (205, 205), 8)])
def test_host_batch_ingest_equals_device_batch(pt, oracle, tw, ws, chunk, monkeypatch):
    """pdog_detect_batch_host (frames in host memory, only the window tiles uploaded, chunks rotating through
    three staging slots) returns the positions pdog_detect_batch returns for the same frames on the device;
    guesses up to l÷2 outside the frame, shared frames through frame_index, padded row stride."""
    from oracle import synth
    fh, fw = 300, 420
    radii = (ws[0] // 2, ws[1] // 2)
    nf = 40
    frames, guesses, _ = synth.make_batch(nf, fh, fw, tw, radii, True, seed=33, noise=3)
    wide = np.zeros((nf, fh, fw + 24), np.uint8)
    wide[:, :, :fw] = frames
    strided = wide[:, :, :fw]                                   # row stride fw + 24
    rng = np.random.default_rng(5)
    n = 200
    fi = rng.integers(0, nf, n).astype(np.int32)
    hw = oracle.kernel_len(oracle.sigma(tw)) // 2
    g = guesses[fi] + rng.integers(-radii[0], radii[0] + 1, (n, 2)).astype(np.int32)
    g[:8] = [[-hw, -hw], [fh + hw + 1, fw + hw + 1], [1, fw], [fh, 1], [-hw, fw + hw + 1], [fh + hw + 1, -hw], [1, 1], [fh, fw]]
    g = np.clip(g, [-hw, -hw], [fh + hw + 1, fw + hw + 1]).astype(np.int32)
    fill = oracle.mode_u8(frames[0])
    if chunk:
        monkeypatch.setenv("PDOG_INGEST_CHUNK", str(chunk))
    bt = pt.BatchTracker(fh, fw, tw, ws, True, fill)
    got = bt.detect_host(strided, g, fi)
    again = bt.detect_host(strided, g[:3], fi[:3])              # a second, smaller call reuses the slots
    no_index = bt.detect_host(strided, g[:nf])                  # window b -> frame b
    assert bt.detect_host(strided, g[:0]).shape == (0, 2)
    with pytest.raises(pt.PdogError):
        bad = g[:2].copy(); bad[1, 0] = fh + hw + 2
        bt.detect_host(strided, bad, fi[:2])
    bt.close()
    exp = _batch(pt, frames, g, tw, ws, True, fill, frame_index=fi)
    assert np.array_equal(got, exp)
    assert np.array_equal(again, exp[:3])
    assert np.array_equal(no_index, _batch(pt, frames, g[:nf], tw, ws, True, fill))
    K = oracle.dog_kernel(oracle.sigma(tw), True)
    for b in range(0, n, 37 if tw < 100 else 97):               # and a few against the oracle directly
        assert tuple(got[b]) == oracle.detect(frames[fi[b]], fill, K, radii, g[b])


def test_distinct_trackers_from_distinct_threads(pt, oracle):
    """SURVEY §8b threading: a handle is single-caller, distinct handles are usable from distinct threads
    (README.md:214 of the reference: concurrent `track` calls).  Four host threads, each with its own
    Tracker / BatchTracker of a different geometry, call concurrently (ctypes drops the GIL)."""
    import threading
    from oracle import synth
    cases = [(25, (45, 45)), (10, (21, 21)), (25, (96, 64)), (40, (81, 81))]
    results, errors = {}, []

    def run(k, tw, ws):
        try:
            import torch
            radii = (ws[0] // 2, ws[1] // 2)
            frames, guesses, _ = synth.make_batch(12, 240, 320, tw, radii, True, seed=50 + k, noise=2)
            tr = pt.Tracker(frames[0], tw, ws, True)
            fill = tr.img.fillvalue
            singles = []
            for b in range(len(frames)):
                tr.img.data[...] = frames[b]
                singles.append(tr((int(guesses[b, 0]), int(guesses[b, 1]))))
            bt = pt.BatchTracker(240, 320, tw, ws, True, fill)
            with torch.cuda.stream(torch.cuda.Stream()):
                out = bt.detect(torch.from_numpy(frames).cuda(), torch.from_numpy(guesses).cuda())
                torch.cuda.current_stream().synchronize()
                out = out.cpu().numpy()
            host = bt.detect_host(frames, guesses)
            bt.close()
            results[k] = (frames, guesses, fill, np.array(singles, np.int32), out, host)
        except Exception as e:                                   # noqa: BLE001 - reported by the main thread
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=run, args=(k, tw, ws)) for k, (tw, ws) in enumerate(cases)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for k, (tw, ws) in enumerate(cases):
        frames, guesses, fill, singles, out, host = results[k]
        K = oracle.dog_kernel(oracle.sigma(tw), True)
        exp = oracle.detect_batch(frames, fill, K, (ws[0] // 2, ws[1] // 2), guesses)
        assert np.array_equal(singles, exp) and np.array_equal(out, exp) and np.array_equal(host, exp)


def test_fused_kernel_batches_and_chains(pt, oracle):
    """dog_fused_kernel (one workgroup per window, tile in LDS, one launch): seeded batches of odd window shapes
    with the dense response against the oracle, and the in-kernel frame loop against the oracle's chain; both as
    the automatic choice for small batches and pinned (variant 300)."""
    import torch
    from oracle import synth
    from oracle.dog_oracle import OracleTracker
    rng = np.random.default_rng(77)
    for tw, ws, (h, w), n in ((25, (45, 45), (240, 320), 40), (25, (31, 57), (120, 90), 24), (10, (21, 21), (100, 140), 300),
                              (16, (5, 71), (80, 200), 16), (25, (63, 1), (150, 150), 9), (30, (33, 33), (200, 200), 12),
                              (40, (45, 45), (150, 170), 6)):   # l = 101: 155 KB of the 160 KB LDS
        radii = (ws[0] // 2, ws[1] // 2)
        frames, guesses, _ = synth.make_batch(n, h, w, tw, radii, True, seed=int(rng.integers(1 << 30)), noise=3)
        fill = oracle.mode_u8(frames[0])
        K = oracle.dog_kernel(oracle.sigma(tw), True)
        exp = oracle.detect_batch(frames, fill, K, radii, guesses)
        auto = _batch(pt, frames, guesses, tw, ws, True, fill)
        pinned, resp = _batch(pt, frames, guesses, tw, ws, True, fill, want_resp=True, variant=300)
        assert np.array_equal(auto, exp) and np.array_equal(pinned, exp), (tw, ws)
        for b in range(0, n, max(1, n // 5)):
            _, r = oracle.detect(frames[b], fill, K, radii, guesses[b], want_resp=True)
            _check_resp(resp[b].T, r, f"fused {ws} {b}")
        if True:
            # kernel lengths 29 … 101 have compile-time-length instances (the default above where their tile layout fits LDS;
            # interior and border tiles both occur in these batches); the runtime-length instance must give the same
            # positions and bit-identical responses
            generic, resp_g = _batch(pt, frames, guesses, tw, ws, True, fill, want_resp=True, variant=300, tuning={"no_fused_c": 1})
            assert np.array_equal(generic, exp) and np.array_equal(resp_g, resp), (tw, ws)
    # chains: 5 clips x 30 frames, default 45x45 window; the automatic choice for few clips is the fused kernel
    h, w, tw, ws, nclips, nf = 160, 220, 25, (45, 45), 5, 30
    clips, starts, refs = [], [], []
    for c in range(nclips):
        pos = np.clip(np.cumsum(rng.integers(-6, 7, (nf, 2)), 0) + np.array([h // 2, w // 2]), 5, [h - 5, w - 5])
        fr = np.stack([synth.disc_frame(h, w, (int(p[0]), int(p[1])), tw, True) for p in pos])
        clips.append(np.clip(fr.astype(np.int16) + rng.integers(-3, 4, fr.shape), 0, 255).astype(np.uint8))
        starts.append((int(pos[0][0]) - 2, int(pos[0][1]) + 5))
    fill = oracle.mode_u8(clips[0][0])
    for c in range(nclips):
        ot = OracleTracker(clips[c][0], tw, ws, True, oracle)
        ot.fill = fill
        r = [ot(starts[c])]
        for f in clips[c][1:]:
            ot.data[...] = f
            r.append(ot(r[-1]))
        refs.append(np.array(r, np.int32))
    bt = pt.BatchTracker(h, w, tw, ws, True, fill)
    got = bt.detect_chains(torch.from_numpy(np.stack(clips)).cuda(), torch.tensor(starts, dtype=torch.int32).cuda()).cpu().numpy()
    single = bt.detect_chain(torch.from_numpy(clips[2]).cuda(), starts[2]).cpu().numpy()
    bt.set_tuning("no_fused_c", 1)
    got_g = bt.detect_chains(torch.from_numpy(np.stack(clips)).cuda(), torch.tensor(starts, dtype=torch.int32).cuda()).cpu().numpy()
    bt.close()
    assert np.array_equal(got, np.stack(refs)) and np.array_equal(single, refs[2]) and np.array_equal(got_g, got)


def test_identical_targets_tie_goes_to_the_first_in_column_major_order(pt, oracle):
    """findmax (:59) returns the FIRST maximum in column-major order.  Several identical discs in one noise-free
    window give bit-identical responses at their centres only if every output sees its taps in the same order,
    whatever strip, sub-chunk, remainder column or workgroup computes it: the answer must be the disc with the
    smallest column (then row), for every kernel family, and equal to the oracle's."""
    from oracle import synth
    fh, fw = 400, 720

    def frame_with(discs, tw, darker=True):
        f = np.full((fh, fw), 128, np.uint8)
        rad = tw // 2
        yy, xx = np.mgrid[-rad:rad + 1, -rad:rad + 1]
        m = (yy * yy + xx * xx) <= rad * rad
        for (i, j) in discs:
            f[i - 1 - rad:i + rad, j - 1 - rad:j + rad][m] = 0 if darker else 255
        return f

    # window 181 x 451 centred at (200, 360): columns 135..585 -> 7 strips of 64 + 3 remainder columns (583..585)
    tw, ws, guess = 25, (181, 451), (200, 360)
    layouts = [[(250, 200), (150, 500), (200, 350)],            # different strips and row chunks; first = column 200
               [(150, 500), (260, 500), (130, 560)],            # same column twice: smaller row wins
               [(200, 584), (200, 300)],                        # one centre in the remainder columns (thin kernel)
               [(280, 170), (120, 170 + 64), (200, 170 + 128)]]  # same lane of three consecutive strips
    K = oracle.dog_kernel(oracle.sigma(tw), True)
    for discs in layouts:
        f = frame_with(discs, tw)
        exp = min(discs, key=lambda d: (d[1], d[0]))
        assert oracle.detect(f, 128, K, (ws[0] // 2, ws[1] // 2), guess) == exp
        for variant in (None, 100, 200, 2, 10):
            got = _batch(pt, f[None], np.array([guess], np.int32), tw, ws, True, 128, variant=variant)
            assert tuple(int(v) for v in got[0]) == exp, (discs, variant)
        # enough copies of the window to take the batch kernels' large-batch path as well
        got = _batch(pt, f[None], np.tile(np.array([guess], np.int32), (300, 1)), tw, ws, True, 128, frame_index=np.zeros(300, np.int32))
        assert (got == np.array(exp)).all()
    # a window that fits the fused kernel: two small bright discs (tw 6 -> l = 17)
    tw, ws, guess = 6, (45, 45), (100, 100)
    for discs in ([(90, 110), (108, 92)], [(95, 95), (95, 108)], [(112, 100), (88, 100)]):
        f = frame_with(discs, tw, darker=False)
        exp = min(discs, key=lambda d: (d[1], d[0]))
        K6 = oracle.dog_kernel(oracle.sigma(tw), False)
        assert oracle.detect(f, 128, K6, (22, 22), guess) == exp
        for variant in (None, 300, 200, 117):
            got = _batch(pt, f[None], np.array([guess], np.int32), tw, ws, False, 128, variant=variant)
            assert tuple(int(v) for v in got[0]) == exp, (discs, variant)
        tr = pt.Tracker(f, tw, ws, False)
        assert tr(guess) == exp
        tr.close()


def test_chain_progress_can_be_followed_from_the_host(pt, oracle):
    """pdog_detect_chain_progress: the chain's positions land in host-coherent memory frame by frame with a release-
    ordered counter, so a per-frame consumer (the reference's diagnostic overlay) can follow a device-side chain.
    Whatever prefix the host observes while the chain runs must already be final; the whole result equals
    pdog_detect_chain's.  Fused path (45x45) and per-frame-launch path (129x129 window, two-pass kernels)."""
    import time
    import torch
    from oracle import synth
    rng = np.random.default_rng(5)
    for ws, nf in (((45, 45), 400), ((129, 129), 120)):
        h, w, tw = 300, 400, 25
        pos = np.clip(np.cumsum(rng.integers(-6, 7, (nf, 2)), 0) + np.array([h // 2, w // 2]), 20, [h - 20, w - 20])
        fr = np.stack([synth.disc_frame(h, w, (int(p[0]), int(p[1])), tw, True) for p in pos])
        d_fr = torch.from_numpy(fr).cuda()
        bt = pt.BatchTracker(h, w, tw, ws, True, 128)
        start = (int(pos[0][0]) + 2, int(pos[0][1]) - 3)
        ref = bt.detect_chain(d_fr, start).cpu().numpy()
        assert np.array_equal(ref, pos.astype(np.int32))          # noise-free discs: the chain follows the centres exactly
        cp = bt.detect_chain_progress(d_fr, start)
        seen, snapshots = 0, 0
        t0 = time.perf_counter()
        while seen < nf and time.perf_counter() - t0 < 30:
            k = cp.done()
            assert seen <= k <= nf
            if k > seen:
                assert np.array_equal(cp.positions[:k], ref[:k]), (ws, k)     # every published prefix is final
                snapshots += 1
                seen = k
        assert seen == nf and np.array_equal(cp.wait(), ref)
        cp.close()
        bt.close()


def test_seeded_fuzz_every_entry_point_vs_oracle(pt, oracle):
    """120 seeded random configurations — frames from 1x1 to 90x120 (narrower than a dword, smaller than the kernel,
    smaller than the window), target widths 2…40, odd window shapes, guesses up to l÷2 outside the frame, random
    textures with discs, bright and dark targets — through every entry point that returns a position: the functor
    (in-place tile), the device batch (automatic kernel choice), the host batch and the single-clip chain."""
    import torch
    from oracle import synth
    rng = np.random.default_rng(424242)
    for case in range(120):
        fh, fw = int(rng.integers(1, 91)), int(rng.integers(1, 121))
        if case % 10 == 0:
            fw = int(rng.integers(1, 4))                         # narrower than one dword
        tw = float(rng.choice([2, 3, 5, 8, 10, 13, 16, 20, 25, 33, 40]))
        ws = (int(rng.integers(1, 60)), int(rng.integers(1, 80)))
        darker = bool(rng.integers(0, 2))
        radii = (ws[0] // 2, ws[1] // 2)
        l = oracle.kernel_len(oracle.sigma(tw))
        hw = l // 2
        if (2 * radii[0] + l) * (2 * radii[1] + l) * l * l > 4e9:   # keep the dense oracle affordable
            continue
        n = 5
        frames = rng.integers(100, 156, (n, fh, fw)).astype(np.uint8)
        for b in range(n):
            ci, cj = int(rng.integers(1, fh + 1)), int(rng.integers(1, fw + 1))
            disc = synth.disc_frame(fh, fw, (ci, cj), max(2, int(tw)), darker)
            mask = disc != 128
            frames[b][mask] = disc[mask]
        guesses = np.stack([rng.integers(-hw, fh + hw + 2, n), rng.integers(-hw, fw + hw + 2, n)], 1).astype(np.int32)
        fill = oracle.mode_u8(frames[0])
        K = oracle.dog_kernel(oracle.sigma(tw), darker)
        exp = oracle.detect_batch(frames, fill, K, radii, guesses)
        tag = (case, fh, fw, tw, ws, darker)
        bt = pt.BatchTracker(fh, fw, tw, ws, darker, fill)
        d_frames = torch.from_numpy(frames).cuda()
        got = bt.detect(d_frames, torch.from_numpy(guesses).cuda()).cpu().numpy()
        assert np.array_equal(got, exp), ("batch",) + tag
        assert np.array_equal(bt.detect_host(frames, guesses), exp), ("host batch",) + tag
        # chain on the first frame repeated: each step starts from the previous answer
        rep = np.repeat(frames[:1], 3, 0)
        chain = bt.detect_chain(torch.from_numpy(rep).cuda(), (int(guesses[0, 0]), int(guesses[0, 1]))).cpu().numpy()
        g = (int(guesses[0, 0]), int(guesses[0, 1]))
        for k in range(3):
            g = oracle.detect(frames[0], fill, K, radii, g)
            assert tuple(int(v) for v in chain[k]) == g, ("chain", k) + tag
        bt.close()
        tr = pt.Tracker(frames[0], tw, ws, darker)
        assert tr.img.fillvalue == fill
        for b in range(n):
            tr.img.data[...] = frames[b]
            assert tr((int(guesses[b, 0]), int(guesses[b, 1]))) == tuple(int(v) for v in exp[b]), ("functor", b) + tag
        tr.close()


def test_kernel_for_batch_reports_the_launch_time_switch(pt):
    bt = pt.BatchTracker(1080, 1920, 25, (45, 45), True, 128)
    # windows below 3000 pixels stay on the fused kernel at any batch size (a workgroup walks several windows)
    assert bt.info().variant == 100 and bt.kernel_for_batch(1) == 300 and bt.kernel_for_batch(999) == 300 and bt.kernel_for_batch(4096) == 300
    bt.set_variant(100)
    assert bt.kernel_for_batch(1) == 100
    bt.close()
    bt = pt.BatchTracker(1080, 1920, 25, (63, 63), True, 128)
    assert bt.kernel_for_batch(1) == 300 and bt.kernel_for_batch(999) == 300 and bt.kernel_for_batch(4096) == 100
    bt.close()
    bt = pt.BatchTracker(1080, 1920, 25, (256, 256), True, 128)
    # one or two windows too large for the fused kernel: the tiled kernel (a workgroup per sub-window, one launch)
    assert bt.kernel_for_batch(1) == 400 and bt.kernel_for_batch(2) == 400 and bt.kernel_for_batch(64) == 200 and bt.kernel_for_batch(4096) == 100
    bt.close()
    bt = pt.BatchTracker(1080, 1920, 120, (205, 205), True, 128)   # l = 293: no sub-window's halo fits LDS
    assert bt.info().variant == 200 and bt.kernel_for_batch(1) == 200 and bt.kernel_for_batch(4096) == 200
    bt.close()


@pytest.mark.parametrize("cfg", ["cfg2", "cfg3", "cfg4", "cfg5"])
def test_full_size_one_noisy_window_vs_dense_oracle(pt, oracle, cfg):
    """One noisy window of every BASELINE.json geometry at FULL frame, window and kernel size against the dense
    Float64 oracle (0.3 … 3.6 G MAC each): position exact, response within 1e-5 of max|ref| — through the batch
    kernels that the large batches of the bench use (kernel pinned) and through the small-batch path."""
    from oracle import synth
    fh, fw, tw, ws = {"cfg2": (1080, 1920, 25, (270, 480)), "cfg3": (1080, 1920, 25, (256, 256)),
                      "cfg4": (2160, 3840, 25, (512, 512)), "cfg5": (1080, 1920, 120, (205, 205))}[cfg]
    radii = (ws[0] // 2, ws[1] // 2)
    frames, guesses, _ = synth.make_batch(1, fh, fw, tw, radii, True, seed=77, noise=3)
    fill = oracle.mode_u8(frames[0])
    K = oracle.dog_kernel(oracle.sigma(tw), True)
    ij, ref = oracle.detect(frames[0], fill, K, radii, guesses[0], want_resp=True)
    bt = pt.BatchTracker(fh, fw, tw, ws, True, fill)
    main = bt.info().variant
    bt.close()
    for variant in (None, main):
        got, resp = _batch(pt, frames, guesses, tw, ws, True, fill, want_resp=True, variant=variant)
        assert tuple(int(v) for v in got[0]) == ij, (cfg, variant)
        _check_resp(resp[0].T, ref, f"{cfg} full size variant {variant}")


def test_translation_covariance_full_size(pt, oracle):
    """Shifting a frame and the guess by (di, dj) shifts the answer by (di, dj) and leaves the response bit-identical —
    whatever the new alignment of the tile to 4/16-byte loads, strips and sub-chunks.  1080p, 257x257 windows, noisy
    frames; windows kept clear of the frame border so that the wrap-around of np.roll never enters a tile."""
    import torch
    rng = np.random.default_rng(8)
    fh, fw, tw, ws, n = 1080, 1920, 25, (256, 256), 24
    frames = rng.integers(118, 139, (n, fh, fw)).astype(np.uint8)
    guesses = np.stack([rng.integers(400, fh - 400, n), rng.integers(400, fw - 400, n)], 1).astype(np.int32)
    from oracle import synth
    for b in range(n):
        c = guesses[b] + rng.integers(-60, 61, 2)
        disc = synth.disc_frame(fh, fw, (int(c[0]), int(c[1])), tw, True)
        frames[b][disc != 128] = 0
    fill = 128
    for variant in (100, 200, 10):          # roll kernel (what large batches run), two-pass, a ring kernel
        bt = pt.BatchTracker(fh, fw, tw, ws, True, fill)
        bt.set_variant(variant)
        base, base_resp = bt.detect(torch.from_numpy(frames).cuda(), torch.from_numpy(guesses).cuda(), want_resp=True)
        base, base_resp = base.cpu().numpy(), base_resp.cpu().numpy()
        for (di, dj) in ((1, 1), (0, 3), (7, -5), (-13, 16), (31, -33), (-64, 64), (2, 129)):
            shifted = np.roll(frames, (di, dj), axis=(1, 2))
            g2 = guesses + np.array([di, dj], np.int32)
            got, resp = bt.detect(torch.from_numpy(shifted).cuda(), torch.from_numpy(g2).cuda(), want_resp=True)
            assert np.array_equal(got.cpu().numpy(), base + np.array([di, dj], np.int32)), (variant, di, dj)
            assert np.array_equal(resp.cpu().numpy(), base_resp), (variant, di, dj)          # bit-identical response
        bt.close()
    # the same for the small-batch kernels (one window: two-pass; 45x45: fused)
    for ws2 in ((256, 256), (45, 45)):
        bt = pt.BatchTracker(fh, fw, tw, ws2, True, fill)
        b0, r0 = bt.detect(torch.from_numpy(frames[:1]).cuda(), torch.from_numpy(guesses[:1]).cuda(), want_resp=True)
        for (di, dj) in ((3, 2), (-17, 9)):
            sh = np.roll(frames[:1], (di, dj), axis=(1, 2))
            b1, r1 = bt.detect(torch.from_numpy(sh).cuda(), torch.from_numpy(guesses[:1] + np.array([di, dj], np.int32)).cuda(), want_resp=True)
            assert np.array_equal(b1.cpu().numpy(), b0.cpu().numpy() + np.array([di, dj], np.int32))
            assert torch.equal(r1, r0)
        bt.close()


def test_power_of_two_contrast_scales_the_response_exactly(pt):
    """The path is linear in (pixel - dc) and FP32 scaling by 2 is exact: doubling every pixel's distance from the
    fill value doubles the response bit for bit and leaves the position unchanged (as long as the window's DC level
    stays the fill value: sparse target, balanced noise).  No oracle involved."""
    import torch
    from oracle import synth
    rng = np.random.default_rng(12)
    for (fh, fw, tw, ws, n, variant) in ((600, 800, 25, (256, 256), 6, 100), (600, 800, 25, (256, 256), 6, 200),
                                         (300, 400, 25, (45, 45), 10, 300), (400, 500, 120, (101, 101), 3, 200)):
        fill = 128
        frames = (fill + rng.integers(-2, 3, (n, fh, fw))).astype(np.uint8)
        guesses = np.stack([rng.integers(fh // 3, 2 * fh // 3, n), rng.integers(fw // 3, 2 * fw // 3, n)], 1).astype(np.int32)
        for b in range(n):
            c = guesses[b] + rng.integers(-15, 16, 2)
            disc = synth.disc_frame(fh, fw, (int(c[0]), int(c[1])), tw, True)
            frames[b][disc != 128] = fill - 50
        doubled = (fill + 2 * (frames.astype(np.int16) - fill)).astype(np.uint8)
        bt = pt.BatchTracker(fh, fw, tw, ws, True, fill)
        bt.set_variant(variant)
        p1, r1 = bt.detect(torch.from_numpy(frames).cuda(), torch.from_numpy(guesses).cuda(), want_resp=True)
        p2, r2 = bt.detect(torch.from_numpy(doubled).cuda(), torch.from_numpy(guesses).cuda(), want_resp=True)
        assert torch.equal(p1, p2), variant
        assert torch.equal(r2, 2 * r1), variant
        bt.close()


def _tw_for_kernel_len(oracle, l):
    for tw10 in range(20, 700):
        if oracle.kernel_len(oracle.sigma(tw10 / 10)) == l:
            return tw10 / 10
    raise AssertionError(l)


@pytest.mark.parametrize("l", list(range(17, 150, 4)))
def test_every_roll_instance_pinned(pt, oracle, l):
    """One compiled roll-kernel instance per kernel length l = 17, 21, … 149 (dog_roll.hpp).  Small batches are
    switched to the fused / two-pass kernels at launch, so the instance is pinned here (pdog_set_variant) and run on
    window shapes that cover a partial strip, an overlapping last strip and remainder columns for the thin kernel —
    positions and the dense response against the oracle, plus the persistent chain kernel of the same instance."""
    import torch
    from oracle import synth
    from oracle.dog_oracle import OracleTracker
    tw = _tw_for_kernel_len(oracle, l)
    rng = np.random.default_rng(1000 + l)
    K = oracle.dog_kernel(oracle.sigma(tw), True)
    for ws in ((33, 41), (27, 100), (45, 131), (21, 193)):        # 41: partial strip, 100: overlapping last strip, 131/193: 64k + 3 / + 1
        radii = (ws[0] // 2, ws[1] // 2)
        n, h, w = 6, 150, 260
        frames, guesses, _ = synth.make_batch(n, h, w, max(2, int(tw)), radii, True, seed=int(rng.integers(1 << 30)), noise=3)
        fill = oracle.mode_u8(frames[0])
        bt = pt.BatchTracker(h, w, tw, ws, True, fill)
        vid = 100 if l == 65 else 100 + l
        assert bt.info().variant == vid and bt.info().kernel_len == l   # the roll instance is the tracker's batch kernel up to l = 149
        bt.set_variant(vid)
        assert bt.kernel_for_batch(n) == vid
        got, resp = bt.detect(torch.from_numpy(frames).cuda(), torch.from_numpy(guesses).cuda(), want_resp=True)
        got, resp = got.cpu().numpy(), resp.cpu().numpy()
        for b in range(n):
            ij, r = oracle.detect(frames[b], fill, K, radii, guesses[b], want_resp=True)
            assert tuple(int(v) for v in got[b]) == ij, (l, ws, b)
            if b < 2:
                _check_resp(resp[b].T, r, f"roll l={l} {ws} {b}")
        if ws != (33, 41):       # the persistent chain kernel of this instance (2, 3 and 4 strip-waves per clip): 2 clips x 5 frames
            clip = np.stack([frames[:5], frames[1:6]])
            starts = [(int(guesses[0, 0]), int(guesses[0, 1])), (int(guesses[1, 0]), int(guesses[1, 1]))]
            out = bt.detect_chains(torch.from_numpy(clip).cuda(), torch.tensor(starts, dtype=torch.int32).cuda()).cpu().numpy()
            for c in range(2):
                ot = OracleTracker(clip[c][0], tw, ws, True, oracle)
                ot.fill = fill
                g = starts[c]
                for k in range(5):
                    ot.data[...] = clip[c][k]
                    g = ot(g)
                    assert tuple(int(v) for v in out[c][k]) == g, (l, c, k)
        bt.close()


@pytest.mark.parametrize("l", list(range(29, 102, 4)))
def test_every_latency_instance_vs_oracle_and_runtime_length_instance(pt, oracle, l):
    """One compile-time-length instance of the fused and of the tiled kernel per kernel length l = 29 … 101 (lat_inst.hip,
    lat_lengths.def).  Small batches and a serial chain through each — a window the fused kernel holds in LDS, one it does
    not (tiled: 4 … 9 sub-windows, rows-of-4 and rows-of-8 row-pass tasks), tiles inside the frame and across its border —
    positions against the oracle, and positions + bit-identical responses against the runtime-length instance
    (`pdog_set_tuning "no_fused_c"`)."""
    import torch
    from oracle import synth
    tw = _tw_for_kernel_len(oracle, l)
    rng = np.random.default_rng(2000 + l)
    K = oracle.dog_kernel(oracle.sigma(tw), True)
    h, w = 190, 250
    for ws, kernel in (((23, 37), 300), ((96, 81), 400)):
        radii = (ws[0] // 2, ws[1] // 2)
        n = 2 if kernel == 400 else 6       # (the tiled kernel serves batches of one or two windows)
        frames, guesses, _ = synth.make_batch(n, h, w, max(2, int(tw)), radii, True, seed=int(rng.integers(1 << 30)), noise=3)
        fill = oracle.mode_u8(frames[0])
        d_f, d_g = torch.from_numpy(frames).cuda(), torch.from_numpy(guesses).cuda()
        res = {}
        for generic in (0, 1):
            bt = pt.BatchTracker(h, w, tw, ws, True, fill)
            bt.set_tuning("no_fused_c", generic)
            assert bt.kernel_for_batch(n) == kernel, (l, ws, bt.kernel_for_batch(n))
            got, resp = bt.detect(d_f, d_g, want_resp=True)
            chain = bt.detect_chain(d_f, (int(guesses[0, 0]), int(guesses[0, 1])))
            bt.sync()
            res[generic] = (got.cpu().numpy(), resp.cpu().numpy(), chain.cpu().numpy())
            bt.close()
        assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2]), (l, ws)
        for b in range(n):
            ij, r = oracle.detect(frames[b], fill, K, radii, guesses[b], want_resp=True)
            assert tuple(int(v) for v in res[0][0][b]) == ij, (l, ws, b)
            if b == 0:
                _check_resp(res[0][1][b].T, r, f"latency instance l={l} {ws}")
        g = (int(guesses[0, 0]), int(guesses[0, 1]))
        for k in range(n):
            g = oracle.detect(frames[k], fill, K, radii, g)
            assert tuple(int(v) for v in res[0][2][k]) == g, (l, ws, k)


@pytest.mark.parametrize("win_h", [256, 384, 512, 1024] + [70, 74, 78, 82, 86, 90, 94, 98, 102, 106, 110, 114, 118, 122, 126, 60, 62, 66])
def test_epilogue_height_classes(pt, oracle, win_h):
    """dog_roll_kernel<65, false, 0, EPI>: instances with statically shortened epilogue bodies, one per window-height class
    ((rows + 2) ÷ 4 mod 18): the common window sizes and one height for each of the 18 classes.  The target sits in the
    LAST rows of the window, where the shortened bodies run; positions against the dense oracle."""
    import torch
    from oracle import synth
    tw, ws = 25, (win_h, 70)
    radii = (ws[0] // 2, ws[1] // 2)
    n, fh, fw = 12, max(160, win_h + 40), 200
    rng = np.random.default_rng(win_h)
    guesses = np.stack([rng.integers(fh // 2 - 10, fh // 2 + 10, n), rng.integers(60, 140, n)], 1).astype(np.int32)
    frames = np.empty((n, fh, fw), np.uint8)
    for b in range(n):      # disc centre in the bottom 40 rows of the window (clipped by the frame for the tallest windows)
        ci = int(min(fh, guesses[b, 0] + radii[0] - rng.integers(0, 40)))
        synth.disc_frame(fh, fw, (ci, int(guesses[b, 1] + rng.integers(-20, 21))), tw, True, out=frames[b])
    frames = np.clip(frames.astype(np.int16) + rng.integers(-3, 4, frames.shape), 0, 255).astype(np.uint8)
    fill = oracle.mode_u8(frames[0])
    K = oracle.dog_kernel(oracle.sigma(tw), True)
    ref = oracle.detect_batch(frames, fill, K, radii, guesses)
    bt = pt.BatchTracker(fh, fw, tw, ws, True, fill)
    bt.set_variant(100)
    assert bt.kernel_for_batch(n) == 100
    got = bt.detect(torch.from_numpy(frames).cuda(), torch.from_numpy(guesses).cuda())
    bt.sync()
    assert np.array_equal(got.cpu().numpy(), ref), win_h
    bt.close()


def test_seeded_fuzz_two_pass_task_sizes_vs_oracle(pt, oracle):
    """The two-pass kernels pick their outputs per task per geometry (row pass P = 9 / 13, column pass P = 7 / 9, 8-tap
    blocks, register-ring windows): seeded random window shapes 20…300 on both sides — every combination of the choices —
    kernel lengths 29…125, bright and dark targets, windows hanging over the frame; positions and a response map per
    case against the dense oracle, batch kernels pinned to the two-pass family (large-batch form) and its small-batch form."""
    import torch
    from oracle import synth
    rng = np.random.default_rng(777)
    seen = set()
    for case in range(60):
        fh, fw = int(rng.integers(80, 360)), int(rng.integers(80, 420))
        tw = float(rng.choice([10, 16, 25, 33, 40, 50]))
        ws = (int(rng.integers(20, 301)), int(rng.integers(20, 301)))
        darker = bool(rng.integers(0, 2))
        radii = (ws[0] // 2, ws[1] // 2)
        n1, n2 = 2 * radii[0] + 1, 2 * radii[1] + 1
        l = oracle.kernel_len(oracle.sigma(tw))
        hw = l // 2
        if n1 * n2 * l * l > 6e8:
            continue
        ph1 = 13 if n2 / (-(-n2 // 208) * 208) > n2 / (-(-n2 // 144) * 144) + 0.05 else 9
        php = 9 if n1 / (-(-n1 // 288) * 288) > n1 / (-(-n1 // 224) * 224) + 0.05 else 7
        seen.add((ph1, php))
        n = 20                                        # > 16 windows: the four-launch form; the first two again as a small batch
        frames = rng.integers(118, 139, (n, fh, fw)).astype(np.uint8)
        for b in range(n):
            disc = synth.disc_frame(fh, fw, (int(rng.integers(1, fh + 1)), int(rng.integers(1, fw + 1))), max(2, int(tw)), darker)
            mask = disc != 128
            frames[b][mask] = disc[mask]
        guesses = np.stack([rng.integers(-hw, fh + hw + 2, n), rng.integers(-hw, fw + hw + 2, n)], 1).astype(np.int32)
        fill = oracle.mode_u8(frames[0])
        K = oracle.dog_kernel(oracle.sigma(tw), darker)
        exp = oracle.detect_batch_par(frames, fill, K, oracle.sigma(tw), darker, radii, guesses, separable=False)
        bt = pt.BatchTracker(fh, fw, tw, ws, darker, fill)
        bt.set_variant(200)
        d_f, d_g = torch.from_numpy(frames).cuda(), torch.from_numpy(guesses).cuda()
        got, resp = bt.detect(d_f, d_g, want_resp=True)
        assert np.array_equal(got.cpu().numpy(), exp), ("batch", case, fh, fw, tw, ws, darker)
        _, ref = oracle.detect(frames[0], fill, K, radii, tuple(guesses[0]), want_resp=True)
        err, _ = _resp_err(resp[0].cpu().numpy().T, ref)
        assert err <= 2.0 ** -24 * (6 * l + 4), ("response", case, err)   # the proven FP32 bound δ (csrc/dog_exact.hpp), not a relative one: these windows hold mostly noise
        small = bt.detect(d_f[:2], d_g[:2]).cpu().numpy()
        assert np.array_equal(small, exp[:2]), ("small batch", case, fh, fw, tw, ws, darker)
        bt.close()
    assert seen == {(9, 7), (9, 9), (13, 7), (13, 9)}, seen
