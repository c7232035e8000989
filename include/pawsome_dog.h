/*
 * pawsome_dog.h — C ABI of the MI355X-native DoG + argmax hot path.
 *
 * This is the drop-in boundary for PawsomeTracker's `Tracker`
 * (reference: /root/reference/src/PawsomeTracker.jl).  The reference has no
 * FFI of its own; the seam is the Julia callable-struct protocol
 *     Tracker(img, target_width, window_size, darker_target)   :39-52
 *     trckr(guess::NTuple{2,Int})::NTuple{2,Int}               :55-62
 *     read!(vid, trckr.img.data)  (frame ingest, in place)     :166
 * A Julia shim keeps those three spellings and `ccall`s the
 * entry points below.  Plain pointers and sizes only; every call returns an
 * int status (0 = ok) and never throws or aborts across the boundary;
 * pdog_last_error() gives the text the shim turns into `error(...)`.
 *
 * Conventions (all follow the reference):
 *  - frames are raw GRAY8, row-major h x w (what the PermutedDimsArray at
 *    :36 wraps), value = raw/255 (Gray{N0f8});
 *  - positions are 1-based (row, col) int32 pairs (CartesianIndex{2}, :173);
 *  - window_size is (h, w) AFTER fix_window_size (:70-72); radii = size .÷ 2
 *    (:44), so a window has 2r+1 rows/cols (:56);
 *  - everything outside the frame reads as the fill value (PaddedView, :48).
 *
 * Threading: a tracker handle is single-caller, like a `Tracker` (shared
 * buffers, :37); distinct handles may be used from distinct threads.
 */
#ifndef PAWSOME_DOG_H
#define PAWSOME_DOG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PDOG_ABI_VERSION 1

enum pdog_status {
    PDOG_OK = 0,
    PDOG_E_ARG = 1,    /* bad argument (null pointer, non-positive size, ...) */
    PDOG_E_HIP = 2,    /* a HIP runtime call failed; text in pdog_last_error() */
    PDOG_E_NODEV = 3,  /* no usable gfx950 device */
    PDOG_E_RANGE = 4,  /* a guess lies further outside the frame than the reference's pad allows */
    PDOG_E_ALLOC = 5
};

typedef struct pdog_tracker pdog_tracker; /* opaque; replaces `struct Tracker`, :32-53 */

/* Geometry the host side may want to read back (all derived as the reference derives them). */
typedef struct pdog_info {
    int32_t frame_h, frame_w; /* sz, :40 */
    int32_t radius_h, radius_w; /* radii = window_size .÷ 2, :44 */
    int32_t win_h, win_w;     /* 2r+1 per dim, :56 */
    int32_t kernel_len;       /* l of Kernel.DoG(sigma), :43 */
    int32_t fill;             /* mode of the first frame, :47 */
    int32_t darker_target;    /* :42 */
    int32_t n_strips;         /* column strips one window is split into on the GPU */
    int32_t strip_w;          /* output columns per strip */
    int32_t variant;          /* id of the compiled kernel specialisation in use */
    double sigma;             /* get_sigma(target_width), :30 */
    double target_width;
    int64_t algorithmic_bytes_per_window; /* (win_h+l-1)*(win_w+l-1) u8 + 8 B out */
    int64_t algorithmic_fma_per_window;   /* separable, both Gaussians, both passes */
} pdog_info;

int pdog_abi_version(void);
/* Text of the last failure on the calling thread ("" if none). */
const char *pdog_last_error(void);

/* ---- scalar helpers (host arithmetic, Float64 like the reference) ---- */
/* get_sigma, src/PawsomeTracker.jl:30 */
double pdog_sigma(double target_width);
/* guess_window_size, src/PawsomeTracker.jl:64-68 */
int pdog_default_window(double target_width);
/* length l of Kernel.DoG(get_sigma(target_width)), src/PawsomeTracker.jl:43 */
int pdog_kernel_len(double target_width);
/* The two normalised 1-D Gaussians whose outer products make Kernel.DoG (:43):
 * which = 0 -> sigma, which = 1 -> sqrt(2) sigma.  out has room for cap doubles. */
int pdog_gaussian_taps(double target_width, int which, double *out, int cap);
/* kernel = direction * Kernel.DoG(sigma), src/PawsomeTracker.jl:41-43: the dense l x l Float64 kernel, column-major,
 * products and difference rounded separately.  This is the table exact mode re-evaluates near-ties with; out has
 * room for cap (>= l*l) doubles. */
int pdog_dense_kernel(double target_width, int darker_target, double *out, int cap);
/* mode(_img), src/PawsomeTracker.jl:47 (StatsBase.mode tie rule: first value whose
 * count exceeds the running maximum while scanning column-major). Host pointer. */
int pdog_mode_u8(const uint8_t *img, int h, int w, int64_t row_stride, int *out_mode);

/* The same for a frame that already lives in device memory (one histogram pass on the GPU, 2 KB read
 * back; same tie rule: equal counts -> the value whose last occurrence comes first column-major).
 * Synchronous on hip_stream (NULL = the null stream). */
int pdog_mode_u8_device(int device, const uint8_t *d_img, int h, int w, int64_t row_stride,
                        void *hip_stream, int *out_mode);

/* ---- Tracker constructor, src/PawsomeTracker.jl:39-52 ----
 * win_h/win_w: window_size in (h, w) order; fill: the mode of the first frame
 * (pdog_mode_u8), frozen for the tracker's life like :47.  device: HIP ordinal. */
int pdog_create(int device, int frame_h, int frame_w, double target_width,
                int win_h, int win_w, int darker_target, int fill, pdog_tracker **out);
int pdog_destroy(pdog_tracker *t);
int pdog_get_info(const pdog_tracker *t, pdog_info *out);
int pdog_set_fill(pdog_tracker *t, int fill);
/* Launch on this hipStream_t instead of the tracker's own stream (NULL = HIP's null stream,
 * which is what torch.cuda.current_stream().cuda_stream reports for torch's default stream). */
int pdog_set_stream(pdog_tracker *t, void *hip_stream);
/* The hipStream_t the tracker launches on right now (its own stream unless pdog_set_stream changed it). */
int pdog_get_stream(const pdog_tracker *t, void **out_hip_stream);
/* Pre-size the per-window workspace so pdog_detect_batch never allocates. */
int pdog_reserve(pdog_tracker *t, int max_windows);
/* The kernel family a batch of n windows would run on (variant id: 300 = one workgroup per window,
 * 400 = one workgroup per sub-window of a large window (one or two windows, single-clip chains), 200 = two-pass,
 * otherwise the tracker's batch kernel as in pdog_info.variant): small batches are switched at launch to whatever
 * can fill the GPU.  For reporting. */
int pdog_kernel_for_batch(const pdog_tracker *t, int n, int *out_variant);
/* Force a kernel specialisation (tuning/tests); -1 = automatic. */
int pdog_set_variant(pdog_tracker *t, int variant);
/* Waits for the tracker's stream.  Returns PDOG_E_RANGE when a kernel of the work just finished met a
 * device-resident guess (pdog_detect_batch, pdog_detect_chains) further outside the frame than the reference's
 * pad allows — where `trckr(guess)` raises a BoundsError (src/PawsomeTracker.jl:45-46); the flag is cleared by
 * the call that reports it.  Host-side guesses are checked before anything is launched (pdog_detect_host …). */
int pdog_sync(pdog_tracker *t);

/* Exact mode (default on).  The reference ranks Float64 dense sums (src/PawsomeTracker.jl:57-59); the kernels rank
 * FP32 separable sums whose error is bounded by delta = 2^-24 * F * V/255, F following each kernel family's own operation
 * order (l = 65: F = 157 for the batch kernels; csrc/dog_exact.hpp).  Every kernel also tracks the runner-up
 * response of its window; when it lies within 2 delta of the maximum, the window's near-maximal pixels are
 * re-evaluated on the device as this repository's RESTATEMENT of the reference evaluates them (oracle/dog_oracle.c:
 * dense l x l Float64 correlation, kernel column-major accumulation order) and the first maximum of those values is
 * returned — so a returned position is the oracle's, not merely close to it.  The oracle is unpinned (no Julia here, no
 * numeric fixture in the reference); the guarantee is exactly as strong as its two last-bit assumptions: Float64(::N0f8)
 * is pixel / 255.0 (FixedPointNumbers 0.5-0.8 allowed by Project.toml: some versions multiply by a reciprocal), and
 * ImageFiltering's FIR inner loop is a strictly sequential `tmp += a*b`, products and sums rounded separately.
 * pdog_set_exact(t, 0) switches the re-evaluation off (the FP32 argmax
 * is returned as is); pdog_set_exact(t, 2) re-evaluates EVERY pixel of EVERY window that way (the whole restated
 * computation on the device: a self-check, orders of magnitude slower).  pdog_get_exact: state, the threshold 2 delta of the
 * tracker's batch kernels, and how many windows have been re-evaluated since the tracker was created (any of the out
 * pointers may be NULL; reading the count drains the stream and reports what its kernels raised, like pdog_sync).
 * pdog_create switches exact mode OFF for windows too tall for the refinement's LDS block (n1 + l beyond ~9000 rows:
 * pdog_set_exact(t, 1) then fails with PDOG_E_ARG); pdog_get_exact tells, and pdog_last_error() holds a note right after that
 * pdog_create (which still returns PDOG_OK). */
int pdog_set_exact(pdog_tracker *t, int on);
/* Pins one of the library's alternative code paths on a live tracker — for tests and same-session A/B, not for a host:
 * the defaults are the measured-best choices.  The library reads NO path switch from the environment (only resource
 * limits: PDOG_SCRATCH_MB, PDOG_MAP_MB, PDOG_HOST_THREADS, PDOG_INGEST_CHUNK, once at pdog_create).  Keys (value 0 / 1):
 *   "host_copy"    pdog_detect_host uploads the tile with copy commands instead of reading it in place
 *   "host_sync"    … reads in place but waits with a stream synchronise instead of the ticket
 *   "twopass_4l"   the two-pass kernels always in their four-launch form
 *   "no_tiled"     single large windows on the two-pass launches instead of the tiled kernel
 *   "no_fused_c"   the one-workgroup kernels' runtime-length instances also where a compile-time-length one exists (l = 29 … 101)
 *   "fault_inject" tests: one workgroup of a tiled chain skips an arrival (the bounded device-side waits must give up)
 *   "no_roll_map"  hard batches keep recomputing their refinement candidates
 *   "no_fold" / "fold_always"  a single remainder column always / never goes to the remainder-column kernel
 * Unknown key: PDOG_E_ARG.  Drains the tracker's stream. */
int pdog_set_tuning(pdog_tracker *t, const char *key, int value);
int pdog_get_exact(pdog_tracker *t, int *out_on, double *out_threshold, uint64_t *out_refined);
/* Where the re-evaluation's work went since the tracker was created: out[0] windows refined, out[1] column blocks
 * rescanned in FP32, out[2] candidates evaluated in separable Float64, out[3] sequential dense Float64 chains run
 * (0 for a window whose separable stage left a single survivor).  Drains the stream. */
int pdog_get_exact_detail(pdog_tracker *t, uint64_t out[4]);

/* ---- the functor, src/PawsomeTracker.jl:55-62, n independent applications ----
 * All pointers are DEVICE pointers.
 *  d_frames      n_frames frames, frame k at d_frames + k*frame_stride, rows row_stride bytes apart
 *  d_frame_index window b looks at frame d_frame_index[b]; NULL -> frame b
 *  d_guesses     n x 2 int32, 1-based (row, col); need not be clamped (a6)
 *  d_out_ij      n x 2 int32, 1-based (row, col), clamped to the frame (:61)
 *  d_out_resp    NULL, or n x win_h x win_w float32, column-major per window: the
 *                DoG response findmax sees (:58), for parity checks only
 * Asynchronous on the tracker's stream. */
int pdog_detect_batch(pdog_tracker *t, const uint8_t *d_frames, int64_t frame_stride,
                      int64_t row_stride, int n_frames, const int32_t *d_frame_index,
                      const int32_t *d_guesses, int n, int32_t *d_out_ij, float *d_out_resp);

/* One frame from HOST memory: ingest (:166) + functor (:167) + result back on
 * the host.  Synchronous.  h_resp may be NULL.  This is what the Julia shim's
 * `trckr(guess)` calls. Returns PDOG_E_RANGE where the reference would raise a
 * BoundsError (guess more than l - l÷2 - 1 outside the frame). */
int pdog_detect_host(pdog_tracker *t, const uint8_t *h_frame, int64_t row_stride,
                     const int32_t guess[2], int32_t out_ij[2], float *h_resp);

/* Host-only helper (no GPU): the padded tile the functor reads for one window — rows
 * guess[0] - win_h÷2 - l÷2 … guess[0] + win_h÷2 + l÷2 of the frame, likewise columns, (2·(win÷2) + l) per side —
 * with the PaddedView fill (src/PawsomeTracker.jl:48) materialised wherever it leaves the frame.  This is what
 * pdog_detect_host / pdog_detect_batch_host hand to the kernels.  Bytes past the tile width up to out_pitch
 * are set to fill. */
int pdog_window_tile(const uint8_t *h_frame, int frame_h, int frame_w, int64_t row_stride, int fill,
                     double target_width, int win_h, int win_w, const int32_t guess[2],
                     uint8_t *h_out, int64_t out_pitch);

/* n independent applications on frames in HOST memory — the batch form of the ingest step
 * `read!(vid, trckr.img.data)` (src/PawsomeTracker.jl:166) followed by the functor (:55-62).
 * All pointers are HOST pointers (pageable is fine); arguments as pdog_detect_batch.  Only each
 * window's padded tile crosses PCIe: host threads (PDOG_HOST_THREADS, default min(16, cores)) pack
 * tiles into pinned staging, chunked copies overlap the kernels.  Synchronous; positions are the
 * ones pdog_detect_batch returns for the same frames on the device.  PDOG_E_RANGE as pdog_detect_host. */
int pdog_detect_batch_host(pdog_tracker *t, const uint8_t *h_frames, int64_t frame_stride,
                           int64_t row_stride, int n_frames, const int32_t *h_frame_index,
                           const int32_t *h_guesses, int n, int32_t *h_out_ij);

/* The intended frame loop, src/PawsomeTracker.jl:163-169 (:167):
 * out[0] = functor(frame 0, start_guess); out[k] = functor(frame k, out[k-1]).
 * d_frames / d_out_ij are device pointers; start_guess is a host pointer. */
int pdog_detect_chain(pdog_tracker *t, const uint8_t *d_frames, int64_t frame_stride,
                      int64_t row_stride, int n_frames, const int32_t start_guess[2],
                      int32_t *d_out_ij);

/* Many clips at once, each the serial chain of :163-169: clip c's frame k is frame c*n_frames + k of
 * d_frames; d_start_guesses is n_clips x 2 (device), d_out_ij is n_clips x n_frames x 2 (device).
 * For short kernels (l = 65) and windows up to 512 columns this is ONE persistent launch: a workgroup
 * per clip walks its frames without any host involvement. */
int pdog_detect_chains(pdog_tracker *t, const uint8_t *d_frames, int64_t frame_stride,
                       int64_t row_stride, int n_frames, int n_clips,
                       const int32_t *d_start_guesses, int32_t *d_out_ij);

/* A chain whose positions can be consumed WHILE it runs — what the reference's diagnostic overlay
 * (src/diagnose.jl:30-38, called per frame inside the loop :163-169) needs from a device-side chain.
 * h_out_ij (n_frames x 2 int32) and h_progress (one int32) must come from pdog_alloc_host (pinned,
 * device-mapped, host-coherent; zero-initialised).  *h_progress counts finished frames with release
 * order: once an acquire load of it returns a value > k, h_out_ij[2k], h_out_ij[2k+1] hold frame k's
 * position.  Asynchronous on the tracker's stream; d_frames is a device pointer, start_guess a host
 * pointer.  Same positions as pdog_detect_chain. */
int pdog_alloc_host(size_t bytes, void **out);
int pdog_free_host(void *p);
int pdog_detect_chain_progress(pdog_tracker *t, const uint8_t *d_frames, int64_t frame_stride,
                               int64_t row_stride, int n_frames, const int32_t start_guess[2],
                               int32_t *h_out_ij, int32_t *h_progress);

/* ---- several GPUs of one node behind one handle ----
 * The functor is applied to n independent windows (src/PawsomeTracker.jl:55-62, "N independent
 * applications"), so a batch shards by contiguous window ranges: rank r of the group owns windows
 * [lo_r, hi_r) (pdog_group_shard; sizes differ by at most one), keeps ITS frames and guesses resident on
 * ITS device, and runs pdog_detect_batch there.  The only exchange is the result: one ncclGather (RCCL over
 * xGMI) of the int32 (row, col) pairs to the root device — the positions `track` returns as
 * CartesianIndex.(indices), src/PawsomeTracker.jl:173.  One host process drives all devices
 * (ncclCommInitAll); a group of size 1 is valid and runs the same code (RCCL accepts a 1-rank communicator).
 * A serial chain (src/PawsomeTracker.jl:167) does not shard: use one pdog_tracker per clip and device. */
typedef struct pdog_group pdog_group;

/* devices: ndev HIP ordinals (NULL = 0 … ndev-1); devices[0] is the root that receives the results.
 * Other arguments as pdog_create; every rank gets the same Tracker parameters. */
int pdog_group_create(int ndev, const int *devices, int frame_h, int frame_w, double target_width,
                      int win_h, int win_w, int darker_target, int fill, pdog_group **out);
int pdog_group_destroy(pdog_group *g);
int pdog_group_size(const pdog_group *g);
/* Borrowed handle of rank's tracker (pdog_get_info, pdog_set_variant, pdog_reserve, pdog_set_stream …);
 * owned by the group. */
int pdog_group_tracker(pdog_group *g, int rank, pdog_tracker **out);
/* The contiguous window range [*lo, *hi) of n_total windows that `rank` owns. */
int pdog_group_shard(const pdog_group *g, int n_total, int rank, int *lo, int *hi);
/* The same partition as pure host arithmetic (no group, no GPU): rank's range of n_total windows over ndev
 * ranks, and its inverse — the rank that owns `window` and the window's index inside that rank's shard (what
 * the root uses to put the gathered blocks back into window order). */
int pdog_shard_range(int n_total, int ndev, int rank, int *lo, int *hi);
int pdog_shard_owner(int n_total, int ndev, int window, int *rank, int *local_index);
/* n_total independent windows over the group.  Per-rank arrays (host arrays of ndev device pointers, each
 * pointer valid on that rank's device):
 *   d_frames[r]       rank r's frames (frame k at + k*frame_stride), n_frames[r] of them
 *   d_frame_index[r]  NULL, or hi_r - lo_r int32: shard-local window b looks at frame d_frame_index[r][b];
 *                     the array of pointers itself may be NULL
 *   d_guesses[r]      (hi_r - lo_r) x 2 int32, 1-based (row, col): the guesses of windows lo_r … hi_r-1
 * d_out_ij: n_total x 2 int32 on the ROOT device, window order.  Asynchronous: every rank's kernels run on its
 * tracker's stream, the gather is enqueued behind them; pdog_group_sync waits for all of it. */
int pdog_group_detect_batch(pdog_group *g, const uint8_t *const *d_frames, int64_t frame_stride,
                            int64_t row_stride, const int *n_frames, const int32_t *const *d_frame_index,
                            const int32_t *const *d_guesses, int n_total, int32_t *d_out_ij);
/* Waits for EVERY rank (each rank's raised flags are reported and cleared); returns the first error. */
int pdog_group_sync(pdog_group *g);
/* Test hook: runs the copy kernel that compacts the gathered blocks of unequal shards (d_gathered: int32[ndev][⌈n_total/ndev⌉][2],
 * d_out: int32[n_total][2], both on the current device) — the one step of pdog_group_detect_batch a one-GPU box cannot reach. */
int pdog_group_test_compact(const int32_t *d_gathered, int n_total, int ndev, int32_t *d_out);

#ifdef __cplusplus
}
#endif
#endif /* PAWSOME_DOG_H */
