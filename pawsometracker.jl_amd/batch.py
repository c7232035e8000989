"""Device-resident batches and multi-GPU sharding for the DoG + argmax path.

`BatchTracker` is n independent applications of the reference functor
(/root/reference/src/PawsomeTracker.jl:55-62) on frames that already live in
HBM; torch is used only for device memory, streams and torch.distributed
(RCCL).  Sharding: contiguous window ranges per rank, no data-path
collective, one gather of the int32 (row, col) pairs to rank 0.
"""
import ctypes as C

from . import _lib


class BatchTracker:
    def __init__(self, frame_h, frame_w, target_width, window_size, darker_target, fill, device=0):
        h = C.c_void_p()
        _lib.check(_lib.lib().pdog_create(int(device), int(frame_h), int(frame_w), float(target_width),
                                          int(window_size[0]), int(window_size[1]), int(bool(darker_target)),
                                          int(fill), C.byref(h)))
        self._h = h
        self.device = int(device)
        self.frame_h, self.frame_w = int(frame_h), int(frame_w)

    def info(self):
        o = _lib.PdogInfo()
        _lib.check(_lib.lib().pdog_get_info(self._h, C.byref(o)))
        return o

    def set_variant(self, variant):
        _lib.check(_lib.lib().pdog_set_variant(self._h, int(variant)))

    def set_exact(self, on):
        """Exact mode (default on): near-ties of the FP32 ranking are re-decided in the reference's Float64 arithmetic."""
        _lib.check(_lib.lib().pdog_set_exact(self._h, int(on)))   # 0 off, 1 on, 2 re-evaluate everything (self-check)

    def set_tuning(self, key, value=1):
        """Pin one of the library's alternative code paths (pdog_set_tuning): tests and A/B only."""
        _lib.check(_lib.lib().pdog_set_tuning(self._h, key.encode(), int(value)))

    def exact_stats(self):
        """(on, threshold 2δ, windows re-evaluated so far) — pdog_get_exact."""
        on, thr, n = C.c_int(), C.c_double(), C.c_uint64()
        _lib.check(_lib.lib().pdog_get_exact(self._h, C.byref(on), C.byref(thr), C.byref(n)))
        return bool(on.value), thr.value, int(n.value)

    def exact_detail(self):
        """(windows refined, column blocks rescanned, candidates, sequential chains) — pdog_get_exact_detail."""
        out = (C.c_uint64 * 4)()
        _lib.check(_lib.lib().pdog_get_exact_detail(self._h, out))
        return tuple(int(v) for v in out)

    def kernel_for_batch(self, n):
        """Variant id of the kernel family a batch of n windows runs on (300 fused, 400 tiled, 200 two-pass, else info().variant)."""
        o = C.c_int()
        _lib.check(_lib.lib().pdog_kernel_for_batch(self._h, int(n), C.byref(o)))
        return o.value

    def reserve(self, n):
        _lib.check(_lib.lib().pdog_reserve(self._h, int(n)))

    def use_torch_stream(self):
        """Launch on torch's current stream.  Every detect* call does this: the kernels are asynchronous, and
        only work queued on the stream torch's caching allocator knows about is safe against a tensor
        (guesses, frames) being dropped by the caller and its memory reused before the kernels ran."""
        import torch
        _lib.check(_lib.lib().pdog_set_stream(self._h, C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))

    def sync(self):
        _lib.check(_lib.lib().pdog_sync(self._h))

    def detect(self, frames, guesses, frame_index=None, out=None, want_resp=False):
        """frames: uint8 cuda tensor [nf, h, w] (row stride may exceed w); guesses: int32 cuda [n, 2]
        1-based (row, col).  Returns int32 cuda [n, 2] (and float32 [n, win_w, win_h] — each
        window column-major, i.e. resp[b].T is the h x w response — when want_resp)."""
        import torch
        self.use_torch_stream()
        assert frames.is_cuda and frames.dtype == torch.uint8 and frames.dim() == 3
        assert frames.stride(2) == 1 and frames.shape[1] == self.frame_h and frames.shape[2] == self.frame_w
        assert guesses.is_cuda and guesses.dtype == torch.int32 and guesses.is_contiguous()
        n = guesses.shape[0]
        if out is None:
            out = torch.empty((n, 2), dtype=torch.int32, device=frames.device)
        resp = None
        if want_resp:
            info = self.info()
            resp = torch.empty((n, info.win_w, info.win_h), dtype=torch.float32, device=frames.device)
        fi = None
        if frame_index is not None:
            assert frame_index.is_cuda and frame_index.dtype == torch.int32 and frame_index.is_contiguous()
            fi = C.c_void_p(frame_index.data_ptr())
        _lib.check(_lib.lib().pdog_detect_batch(
            self._h, C.c_void_p(frames.data_ptr()), frames.stride(0), frames.stride(1), frames.shape[0], fi,
            C.c_void_p(guesses.data_ptr()), n, C.c_void_p(out.data_ptr()),
            C.c_void_p(resp.data_ptr()) if want_resp else None))
        return (out, resp) if want_resp else out

    def detect_host(self, frames, guesses, frame_index=None):
        """The same n applications on frames in HOST memory (numpy uint8 [nf, h, w], the layout
        `read!(vid, trckr.img.data)`, src/PawsomeTracker.jl:166, produces): only the window tiles are uploaded,
        in chunks that overlap the kernels.  guesses int32 [n, 2]; returns numpy int32 [n, 2].  Synchronous."""
        import numpy as np
        self.use_torch_stream()
        assert frames.dtype == np.uint8 and frames.ndim == 3 and frames.strides[2] == 1
        assert frames.shape[1] == self.frame_h and frames.shape[2] == self.frame_w
        g = np.ascontiguousarray(guesses, np.int32)
        n = g.shape[0]
        out = np.empty((n, 2), np.int32)
        fi = None
        if frame_index is not None:
            fi_arr = np.ascontiguousarray(frame_index, np.int32)
            assert fi_arr.shape == (n,)
            fi = C.c_void_p(fi_arr.ctypes.data)
        _lib.check(_lib.lib().pdog_detect_batch_host(
            self._h, C.c_void_p(frames.ctypes.data), frames.strides[0], frames.strides[1], frames.shape[0], fi,
            C.c_void_p(g.ctypes.data), n, C.c_void_p(out.ctypes.data)))
        return out

    def detect_chain(self, frames, start_guess, out=None):
        """The serial chain of src/PawsomeTracker.jl:163-169 on device-resident frames."""
        import torch
        self.use_torch_stream()
        assert frames.is_cuda and frames.dtype == torch.uint8 and frames.dim() == 3 and frames.stride(2) == 1
        n = frames.shape[0]
        if out is None:
            out = torch.empty((n, 2), dtype=torch.int32, device=frames.device)
        g = (C.c_int32 * 2)(int(start_guess[0]), int(start_guess[1]))
        _lib.check(_lib.lib().pdog_detect_chain(self._h, C.c_void_p(frames.data_ptr()), frames.stride(0),
                                                frames.stride(1), n, g, C.c_void_p(out.data_ptr())))
        return out

    def detect_chain_progress(self, frames, start_guess):
        """detect_chain whose positions land in host memory as the frames finish (pdog_detect_chain_progress): returns a
        ChainProgress to poll — what a per-frame consumer such as the reference's diagnostic overlay
        (src/diagnose.jl:30-38) needs from a device-side chain."""
        import torch
        self.use_torch_stream()
        assert frames.is_cuda and frames.dtype == torch.uint8 and frames.dim() == 3 and frames.stride(2) == 1
        cp = ChainProgress(self, frames.shape[0], frames)
        g = (C.c_int32 * 2)(int(start_guess[0]), int(start_guess[1]))
        try:
            _lib.check(_lib.lib().pdog_detect_chain_progress(self._h, C.c_void_p(frames.data_ptr()), frames.stride(0), frames.stride(1),
                                                             frames.shape[0], g, cp._out_ptr, cp._prog_ptr))
        except Exception:
            cp.close()      # nothing was queued: the pinned buffers go back at once
            raise
        return cp

    def detect_chains(self, frames, start_guesses, out=None):
        """Many clips at once (one persistent launch for l = 65): frames uint8 cuda [n_clips, n_frames, h, w],
        start_guesses int32 cuda [n_clips, 2]; returns int32 cuda [n_clips, n_frames, 2]."""
        import torch
        self.use_torch_stream()
        assert frames.is_cuda and frames.dtype == torch.uint8 and frames.dim() == 4 and frames.stride(3) == 1
        assert frames.stride(0) == frames.shape[1] * frames.stride(1), "clips must be stacked contiguously"
        assert start_guesses.is_cuda and start_guesses.dtype == torch.int32 and start_guesses.is_contiguous()
        nc, nf = frames.shape[0], frames.shape[1]
        if out is None:
            out = torch.empty((nc, nf, 2), dtype=torch.int32, device=frames.device)
        _lib.check(_lib.lib().pdog_detect_chains(self._h, C.c_void_p(frames.data_ptr()), frames.stride(1), frames.stride(2),
                                                 nf, nc, C.c_void_p(start_guesses.data_ptr()), C.c_void_p(out.data_ptr())))
        return out

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().pdog_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ChainProgress:
    """Host-visible state of a running chain: `done()` frames are finished and `positions[:done()]` are final
    (the library publishes the count with release order; numpy reads the pinned words afresh on every access)."""

    def __init__(self, bt, n_frames, keepalive=None):
        import numpy as np
        self._bt, self.n_frames, self._keepalive = bt, int(n_frames), keepalive
        self._out_ptr = self._prog_ptr = None
        out, prog = C.c_void_p(), C.c_void_p()
        _lib.check(_lib.lib().pdog_alloc_host(8 * self.n_frames, C.byref(out)))
        self._out_ptr = out
        try:
            _lib.check(_lib.lib().pdog_alloc_host(4, C.byref(prog)))
        except Exception:
            self.close()
            raise
        self._prog_ptr = prog
        self.positions = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_int32)), shape=(self.n_frames, 2))
        self._prog = np.ctypeslib.as_array(C.cast(prog, C.POINTER(C.c_int32)), shape=(1,))

    def done(self):
        return int(self._prog[0])

    def wait(self):
        self._bt.sync()
        return self.positions.copy()

    def close(self):
        """Give the pinned buffers back.  The chain that writes them must have finished: the tracker's stream is
        drained first — unless the tracker itself is already closed (pdog_destroy drains its stream)."""
        if self._out_ptr is None and self._prog_ptr is None:
            return
        try:
            if getattr(self._bt, "_h", None):
                self._bt.sync()          # may raise what the chain's kernels raised: the buffers go back regardless
        finally:
            self.positions = self._prog = None
            for ptr in (self._out_ptr, self._prog_ptr):
                if ptr is not None:
                    _lib.lib().pdog_free_host(ptr)
            self._out_ptr = self._prog_ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GroupTracker:
    """Several GPUs of one node behind one handle (pdog_group_*): contiguous window shards per device, frames and
    guesses resident on the owning device, one RCCL gather of the int32 positions to the root device per batch
    (the sharded unit is the functor src/PawsomeTracker.jl:55-62, the gathered result :173).
    One host process drives every device; torch only provides the device memory."""

    def __init__(self, devices, frame_h, frame_w, target_width, window_size, darker_target, fill):
        self.devices = [int(d) for d in devices]
        arr = (C.c_int * len(self.devices))(*self.devices)
        h = C.c_void_p()
        _lib.check(_lib.lib().pdog_group_create(len(self.devices), arr, int(frame_h), int(frame_w), float(target_width),
                                                int(window_size[0]), int(window_size[1]), int(bool(darker_target)),
                                                int(fill), C.byref(h)))
        self._h = h
        self.frame_h, self.frame_w = int(frame_h), int(frame_w)

    def size(self):
        return _lib.lib().pdog_group_size(self._h)

    def shard(self, n_total, rank):
        lo, hi = C.c_int(), C.c_int()
        _lib.check(_lib.lib().pdog_group_shard(self._h, int(n_total), int(rank), C.byref(lo), C.byref(hi)))
        return lo.value, hi.value

    def info(self, rank=0):
        t = C.c_void_p()
        _lib.check(_lib.lib().pdog_group_tracker(self._h, int(rank), C.byref(t)))
        o = _lib.PdogInfo()
        _lib.check(_lib.lib().pdog_get_info(t, C.byref(o)))
        return o

    def _tracker(self, rank):
        t = C.c_void_p()
        _lib.check(_lib.lib().pdog_group_tracker(self._h, int(rank), C.byref(t)))
        return t

    def stream(self, rank):
        """The hipStream_t (as an int) rank's tracker launches on."""
        st = C.c_void_p()
        _lib.check(_lib.lib().pdog_get_stream(self._tracker(rank), C.byref(st)))
        return st.value or 0

    def kernel_for_batch(self, n_per_rank, rank=0):
        o = C.c_int()
        _lib.check(_lib.lib().pdog_kernel_for_batch(self._tracker(rank), int(n_per_rank), C.byref(o)))
        return o.value

    def reserve(self, n_total):
        for r in range(len(self.devices)):
            lo, hi = self.shard(n_total, r)
            t = C.c_void_p()
            _lib.check(_lib.lib().pdog_group_tracker(self._h, r, C.byref(t)))
            _lib.check(_lib.lib().pdog_reserve(t, max(1, hi - lo)))

    def detect(self, frames, guesses, n_total, out, frame_index=None):
        """frames[r]: uint8 cuda tensor [nf_r, h, w] on device r (same strides on every rank); guesses[r]: int32 cuda
        [n_r, 2] on device r, n_r = the size of rank r's shard of n_total; out: int32 cuda [n_total, 2] on the root
        device.  Asynchronous (each rank's tracker stream); sync() waits.  The caller keeps the tensors alive until then."""
        import torch
        nd = len(self.devices)
        assert len(frames) == nd and len(guesses) == nd
        for r in range(nd):
            f, gq = frames[r], guesses[r]
            lo, hi = self.shard(n_total, r)
            assert f.is_cuda and f.dtype == torch.uint8 and f.dim() == 3 and f.stride(2) == 1 and f.device.index == self.devices[r]
            assert f.shape[1] == self.frame_h and f.shape[2] == self.frame_w
            assert f.stride(0) == frames[0].stride(0) and f.stride(1) == frames[0].stride(1)
            assert gq.is_cuda and gq.dtype == torch.int32 and gq.is_contiguous() and gq.shape == (hi - lo, 2) and gq.device.index == self.devices[r]
        assert out.is_cuda and out.dtype == torch.int32 and out.is_contiguous() and out.shape == (n_total, 2) and out.device.index == self.devices[0]
        P = C.c_void_p * nd
        fr = P(*[f.data_ptr() for f in frames])
        gs = P(*[q.data_ptr() for q in guesses])
        nf = (C.c_int * nd)(*[f.shape[0] for f in frames])
        fi = None
        if frame_index is not None:
            fi = P(*[(x.data_ptr() if x is not None else None) for x in frame_index])
        _lib.check(_lib.lib().pdog_group_detect_batch(self._h, fr, frames[0].stride(0), frames[0].stride(1), nf, fi, gs,
                                                      int(n_total), C.c_void_p(out.data_ptr())))
        return out

    def sync(self):
        _lib.check(_lib.lib().pdog_group_sync(self._h))

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().pdog_group_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def mode_device(frame):
    """mode(_img) (src/PawsomeTracker.jl:47, StatsBase tie rule) of a uint8 cuda tensor [h, w]."""
    import torch
    assert frame.is_cuda and frame.dtype == torch.uint8 and frame.dim() == 2 and frame.stride(1) == 1
    out = C.c_int()
    dev = frame.device.index if frame.device.index is not None else torch.cuda.current_device()
    _lib.check(_lib.lib().pdog_mode_u8_device(dev, C.c_void_p(frame.data_ptr()), frame.shape[0], frame.shape[1], frame.stride(0),
                                              C.c_void_p(torch.cuda.current_stream(dev).cuda_stream), C.byref(out)))
    return out.value


def shard_range(n, rank, world_size):
    """Contiguous window range [lo, hi) owned by `rank`: sizes differ by at most 1."""
    base, rem = divmod(int(n), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class _PendingGather:
    """Handle of an asynchronous gather_positions: wait() returns what the blocking call returns."""

    def __init__(self, work, bufs, sizes, is_dst):
        self._work, self._bufs, self._sizes, self._is_dst = work, bufs, sizes, is_dst

    def wait(self):
        import torch
        self._work.wait()
        if not self._is_dst:
            return None
        return torch.cat([self._bufs[r][: hi - lo] for r, (lo, hi) in enumerate(self._sizes)], 0)


def gather_positions(local_ij, n_total, group=None, dst=0, async_op=False):
    """Gather the per-rank int32 [n_local, 2] results to rank `dst` in shard order.
    The only collective on the path: 8 B per window (RCCL gather over xGMI with the
    nccl backend; gloo on CPU for tests).  Returns the [n_total, 2] tensor on dst, None elsewhere.
    With async_op=True the collective is only enqueued (it runs beside the next batch's kernels) and a
    handle is returned whose wait() gives that result."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = [shard_range(n_total, r, world) for r in range(world)]
    max_n = max(hi - lo for lo, hi in sizes)
    if local_ij.is_cuda and dist.get_backend(group) == "gloo":
        local_ij = local_ij.cpu()      # gloo has no device-memory gather: rehearsals and CPU tests go through host memory
    pad = torch.zeros((max_n, 2), dtype=torch.int32, device=local_ij.device)
    pad[: local_ij.shape[0]] = local_ij
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    work = dist.gather(pad, bufs, dst=dst, group=group, async_op=True)
    pending = _PendingGather(work, bufs, sizes, rank == dst)
    return pending if async_op else pending.wait()
