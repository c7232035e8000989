// dog_kernels.hpp — gfx950 (CDNA4) device code for the DoG + argmax hot path: shared definitions
// (LaunchGeo, tap access, the per-window DC level, the strip combine) and the LDS-ring kernel, the first
// kernel of this repo.  Today the ring kernel serves kernel lengths l < 17 and is the second implementation
// the parity tests compare against; l = 17…97 runs dog_roll.hpp, longer kernels dog_twopass.hpp.
//
// Replaces, for a batch of independent search windows, the body of the
// reference functor /root/reference/src/PawsomeTracker.jl:55-62:
//   :57  imfilter!(…, kernel = ±Kernel.DoG(σ), NoPad(), window)   -> row pass + column pass
//   :58-59 findmax(view(buff, window))                              -> fused argmax
//   :60-61 index map + clamp                                        -> dog_finalize_kernel
// and the PaddedView fill semantics of :48 (reads outside the frame = fill).
//
// Algorithm (not the reference's: ImageFiltering runs the rank-2 DoG as ONE
// dense l×l Float64 kernel; here it is two separable Gaussians in FP32):
//   v      = pixel − dc                         (exact integer; dc = fill or the window mean: ΣK = 0)
//   R±[a,x]= Σ_k g±[k] · v[a, x+k]              row pass (contiguous direction)
//   D[y,x] = Σ_k (s·g+[k])·R+[y+k,x] + (−s·g−[k])·R−[y+k,x],  s = ±1/255   column pass
// One workgroup owns one column strip of one window and streams input rows
// through LDS in chunks: stage (u8 → f32 tile) → row pass (P outputs per lane,
// sliding register window) → ring of R rows in LDS → column pass (Q outputs
// per lane, sliding register window) → running (max, first col-major index)
// in registers → wave-shuffle + LDS reduction → one partial per strip.
// Both Gaussians ride in one v_pk_fma_f32 (pairs), taps come from SGPRs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pdog {

typedef float f2 __attribute__((ext_vector_type(2)));
// Filter taps are read through the CONSTANT address space: a load from it with a
// uniform address is always selected as a scalar load (s_load → SGPR operand).
typedef const f2 __attribute__((address_space(4))) *tap_ptr;
__device__ __forceinline__ tap_ptr as_taps(const f2 *p) { return (tap_ptr)(unsigned long long)p; }

// ---- running peak with runner-up (exact mode, dog_exact.hpp) ----
// The first maximum in column-major order (findmax, :59) and the largest response of any OTHER pixel: when the two
// are further apart than twice the FP32 error bound the FP32 argmax is provably the reference's.
struct Peak {
    float best;
    int idx;      // column-major index of the first maximum
    float second; // runner-up value (−inf if there is none)
};
__device__ __forceinline__ void peak_init(Peak &p)
{
    p.best = -__builtin_huge_valf();
    p.idx = 0x7fffffff;
    p.second = -__builtin_huge_valf();
}
__device__ __forceinline__ void peak_push(Peak &p, float v, int lin)
{
    p.second = __builtin_amdgcn_fmed3f(v, p.best, p.second); // second ≤ best always: the median is the new runner-up
    if (v > p.best || (v == p.best && lin < p.idx)) { p.best = v; p.idx = lin; }
}
// Merge the peak of another set of pixels.  The sets may OVERLAP (the last strip of the roll kernel is shifted left over
// its predecessor, dog_roll.hpp): a pixel both sets hold has the same index and — the strips being bit-identical — the
// same value in both, and must not become its own runner-up (that flagged every window whose peak lay in the overlap for
// a refinement that tighten() cannot withdraw: gap 0).  The true runner-up is then max(second_a, second_b).
__device__ __forceinline__ void peak_merge(Peak &p, float ov, int oi, float os)
{
    const float cross = (oi == p.idx) ? -__builtin_huge_valf() : fminf(p.best, ov);
    p.second = fmaxf(fmaxf(p.second, os), cross);
    if (ov > p.best || (ov == p.best && oi < p.idx)) { p.best = ov; p.idx = oi; }
}
// Cross-lane exchange without LDS traffic: DPP modifiers on the VALU (quad permutes, mirrors inside a row of 16 lanes)
// instead of ds_bpermute — a wave reduction is a chain of dependent steps on the latency path of every frame
// (dog_fused.hpp, dog_tiled.hpp: ≈0.5 µs per reduction with three bpermutes per step).
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_MIRROR = 0x140; // quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false); }
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) { return __builtin_bit_cast(float, dpp_i<CTRL>(__builtin_bit_cast(int, v))); }

// Sum of one int per lane, uniform on return (an SGPR): four DPP adds give every lane its row's total, four readlanes add the rows.
__device__ __forceinline__ int wave_sum(int v)
{
    v += dpp_i<DPP_XOR1>(v);
    v += dpp_i<DPP_XOR2>(v);
    v += dpp_i<DPP_HALF_MIRROR>(v);
    v += dpp_i<DPP_MIRROR>(v);
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48);
}

// The wave's peak, uniform on return.  Three plain reductions instead of one reduction of (value, index, runner-up) triples —
// on the latency path (one workgroup per window, its other waves waiting at a barrier) what costs is the LENGTH of the
// dependent chain (≈9 cycles per instruction for a lone wave), and a triple merge is ≈15 dependent instructions per step:
//   m      = max over lanes of best                                            (6 × v_max_f32 with a DPP operand)
//   idx    = min over the lanes with best = m of their index                    (6 × v_min_i32: the FIRST maximum, findmax :59)
//   second = max over lanes of (the lane holds the winning pixel ? its runner-up : its best)
// A pixel that several lanes hold (overlapping strips: same index, bit-identical value) is excluded from `second` in all of
// them — peak_merge's rule.  Steps: xor 1, xor 2, mirror in 8, mirror in 16 (every lane of a row then holds the row's result),
// row_bcast15 into rows 1 and 3, row_bcast31 into rows 2 and 3: lane 63 holds the wave's.
constexpr int DPP_BCAST15 = 0x142, DPP_BCAST31 = 0x143;
template <typename F>
__device__ __forceinline__ int wave_reduce_bits(int v, F op)
{
    v = op(v, dpp_i<DPP_XOR1>(v));
    v = op(v, dpp_i<DPP_XOR2>(v));
    v = op(v, dpp_i<DPP_HALF_MIRROR>(v));
    v = op(v, dpp_i<DPP_MIRROR>(v));
    v = op(v, __builtin_amdgcn_update_dpp(v, v, DPP_BCAST15, 0xA, 0xF, false));
    v = op(v, __builtin_amdgcn_update_dpp(v, v, DPP_BCAST31, 0xC, 0xF, false));
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ float wave_max(float v)
{
    return __builtin_bit_cast(float, wave_reduce_bits(__builtin_bit_cast(int, v), [](int a, int b) {
        return __builtin_bit_cast(int, fmaxf(__builtin_bit_cast(float, a), __builtin_bit_cast(float, b)));
    }));
}
__device__ __forceinline__ int wave_min(int v) { return wave_reduce_bits(v, [](int a, int b) { return a < b ? a : b; }); }
__device__ __forceinline__ void peak_wave_reduce(Peak &p, int width = 64)
{
    (void)width; // lanes that hold no peak hold peak_init's values
    const float m = wave_max(p.best);
    const int idx = wave_min(p.best == m ? p.idx : 0x7fffffff);
    const float second = wave_max(p.idx == idx ? p.second : p.best);
    p.best = m;
    p.idx = idx;
    p.second = second;
}

// Exact mode (dog_exact.hpp): a window whose two best FP32 responses lie within T = 2δ of each other is re-decided
// by a Float64 re-evaluation of its near-maximal pixels.
struct ExactCtl {
    unsigned long long *stat; // [4] since the tracker was created: windows refined, column blocks rescanned, candidates, sequential chains run (diagnostics)
    int *range_err;           // host-coherent words: [0] set when a guess lies where the reference raises BoundsError (:45-46) (2: a device-side wait gave up); [1] windows flagged for refinement by the finishing kernel, cumulative
    float T;                  // 2δ_main: the launched kernel family's own error bound (pawsome_dog.hip, exact_factors) for |pixel − dc| ≤ 255
    float T_rescan;           // δ_main + δ_rescan: the refinement's recomputed FP32 values (plain chains) against the main kernel's maximum
};
// The reference's PaddedView extends radii + l past the frame (:45-46) and the filter reads radii + l÷2 around the
// guess: a guess outside [−l÷2, sz + l÷2 + 1] raises BoundsError there.  Device-resident guesses cannot be checked
// before the launch, so the kernels raise a flag that pdog_sync reports.
// Bounded wait for a device-scope counter that other resident workgroups advance (dog_tiled.hpp): every wave that waits
// reaches an exit.  The bound is WALL time (s_memrealtime, 100 MHz): after ≈1 s without progress the wait gives up, raises
// `abort` (a device word every other wait polls, so the peers give up within microseconds instead of a second each) and the
// fault value 2 in the host-coherent word (pdog_sync then reports PDOG_E_HIP and zeroes the control words).  Returns false on
// give-up or when a peer has given up: the caller leaves its frame loop without publishing anything further.
constexpr unsigned long long WAIT_TICKS = 100000000ull; // 1 s of s_memrealtime
__device__ __forceinline__ bool wait_counter(const unsigned *ctr, unsigned target, const ExactCtl &x, unsigned *abort)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (unsigned spins = 1; __hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target; ++spins) {
        __builtin_amdgcn_s_sleep(1);
        if ((spins & 63u) == 0u) {
            if (abort && __hip_atomic_load(abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false; // a peer gave up
            if (__builtin_amdgcn_s_memrealtime() - t0 > WAIT_TICKS) {
                if (abort) __hip_atomic_store(abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (x.range_err) __hip_atomic_store(x.range_err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                return false;
            }
        }
    }
    return true; // (the counter reached its target: every peer arrived — no second, dependent read of the abort word on the frame's critical path)
}

__device__ __forceinline__ void range_check(const ExactCtl &x, int g1, int g2, int hw, int fh, int fw)
{
    if (x.range_err && (g1 < -hw || g1 > fh + hw + 1 || g2 < -hw || g2 > fw + hw + 1))
        __hip_atomic_store(x.range_err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

struct LaunchGeo {
    const uint8_t *__restrict__ frames;
    long long frame_stride, row_stride;
    const int *__restrict__ frame_index; // may be null
    const int *__restrict__ guesses;     // n x 2, 1-based (row, col)
    float *__restrict__ resp;            // only written by RESP instantiations
    float *__restrict__ part_val;        // n x nslots
    int *__restrict__ part_idx;          // n x nslots
    float *__restrict__ part_sec;        // n x nslots: runner-up value of each partial
    unsigned long long *__restrict__ part_mask; // n x nslots, roll kernel only: lanes (columns) of the strip whose maximum lies within ex.T of the strip's
    ExactCtl ex;
    int fh, fw, r1, r2, n1, n2, L, fill, nstrips, n;
    int RR, pitchA;                      // only read by runtime-L variants
    int nblocks;                         // n * nstrips
    int nslots;                          // partials per window: nstrips + thin columns
    int thin_x0, nthin;                  // window columns [thin_x0, thin_x0+nthin) go to dog_thin_kernel
    // FOLDED remainder column (dog_roll.hpp): nthin == 1 and fold_r != null ⇒ no dog_thin_kernel launch; the last strip of
    // every window also produces the row-pass outputs (R+, R−) of window column thin_x0 and leaves them here,
    // [n][n1 + L − 1]; dog_finish_kernel runs that one column's column pass.  Same operation order as a strip: the column's
    // values are bit-identical to what a strip would have produced.
    f2 *__restrict__ fold_r;
};

constexpr int FOLD_GO = 5; // output rows per lane and pass of a folded remainder column (257 rows = 64 lanes × 5 + 1 …)
// Column pass + peak of ONE window column whose row-pass outputs Rc[0 … n1 + L − 2] (+ FOLD_GO readable entries of padding) sit
// in LDS, by one wave: a lane owns Q = ⌈n1 / 64⌉ consecutive output rows, FOLD_GO at a time through a sliding register window —
// one LDS read and one scalar tap load per tap feed 2·FOLD_GO FMAs.  Per output: taps ascending, the g+ term then the g− term
// into one f32 — the strips' order.  The peak (best, first column-major index, runner-up) is valid in lane 0 on return.
__device__ __forceinline__ Peak fold_column_peak(const f2 *Rc, int n1, int L, tap_ptr tcol, int lin0, int lane)
{
    Peak tp;
    peak_init(tp);
    const int Q = (n1 + 63) / 64;
    for (int q0 = 0; q0 < Q; q0 += FOLD_GO) {
        const int y0 = lane * Q + q0;
        const int nv = max(0, min(min(FOLD_GO, Q - q0), n1 - y0)); // valid outputs of this lane in this group
        const f2 *rp = Rc + (nv > 0 ? y0 : 0);
        float acc[FOLD_GO];
        f2 win[FOLD_GO];
#pragma unroll
        for (int j = 0; j < FOLD_GO; ++j) { acc[j] = 0.f; win[j] = rp[j]; }
#pragma unroll 5
        for (int t = 0; t < L; ++t) {
            const f2 w = tcol[t];
            const f2 nxt = rp[t + FOLD_GO];
#pragma unroll
            for (int j = 0; j < FOLD_GO; ++j) {
                acc[j] = __builtin_fmaf(win[j].x, w.x, acc[j]);
                acc[j] = __builtin_fmaf(win[j].y, w.y, acc[j]);
            }
#pragma unroll
            for (int j = 0; j + 1 < FOLD_GO; ++j) win[j] = win[j + 1];
            win[FOLD_GO - 1] = nxt;
        }
#pragma unroll
        for (int j = 0; j < FOLD_GO; ++j)
            if (j < nv) peak_push(tp, acc[j], lin0 + y0 + j);
    }
    peak_wave_reduce(tp);
    return tp;
}

__host__ __device__ constexpr int round_up(int v, int m) { return (v + m - 1) / m * m; }
// LDS row pitches.  A (f32 input tile) is read by lanes that sit in consecutive
// rows (ds_read_b32, 32 banks): an odd pitch spreads them over all banks.
__host__ __device__ constexpr int pitch_a(int cols) { return cols | 1; }
// R ring (f2 per element) is written row-per-lane (ds_write_b64) and read
// column-per-lane (ds_read_b64, conflict free for any pitch); pitch in f2 units
// with 2*pitch ≡ 2 (mod 32) would be ideal for the writes; odd is a good compromise.
__host__ __device__ constexpr int pitch_r(int tw) { return tw | 1; }
__host__ __device__ constexpr int ring_rows(int CH, int L, int Q)
{
    // rows a col-pass span can touch: CH new + L-1 halo + (Q-1) slack when the
    // chunk cadence leaves a partial Q-group behind; multiple of 4 for the wrap logic
    return round_up(CH + L - 1 + (((CH % Q) == 0 && ((L - 1) % Q) == 0) ? 0 : Q - 1), 4);
}

// The fixed 32×32 sample grid over the window's padded tile that decides the DC level (see dog_window_kernel):
// thread `tid` of `nthreads` adds up its share; callers reduce and finish with dc_from_sum.
__device__ __forceinline__ int dc_sample_sum(const LaunchGeo &g, const uint8_t *__restrict__ frame, int ti0, int wj0,
                                             int L, int tid, int nthreads)
{
    const int tH = g.n1 + L - 1, tW = g.n2 + L - 1;
    int sum = 0;
    for (int k = tid; k < 1024; k += nthreads) {
        const int gi = ti0 + (int)(((long long)(k >> 5) * tH) >> 5);
        const int gj = wj0 + (int)(((long long)(k & 31) * tW) >> 5);
        int v = g.fill;
        if (gi >= 0 && gi < g.fh && gj >= 0 && gj < g.fw) v = frame[(long long)gi * g.row_stride + gj];
        sum += v;
    }
    return sum;
}
__device__ __forceinline__ int dc_from_sum(int total, int fill)
{
    int dc = (total + 512) >> 10;
    if (abs(dc - fill) <= 8) dc = fill;
    return dc;
}

__device__ __forceinline__ f2 fma_bcast(float a, f2 t, f2 c)
{
    f2 av = {a, a};
    return __builtin_elementwise_fma(av, t, c);
}
__device__ __forceinline__ f2 fma_pair(f2 a, f2 t, f2 c) { return __builtin_elementwise_fma(a, t, c); }

// acc[o] += Σ_{u<U} a[o+u] · taps[u]   (both Gaussians in the pair)
template <int P, int U>
__device__ __forceinline__ void row_block(f2 (&acc)[P], const float *a, tap_ptr taps)
{
    float in[P + U - 1];
#pragma unroll
    for (int i = 0; i < P + U - 1; ++i) in[i] = a[i];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const f2 t = taps[u];
#pragma unroll
        for (int o = 0; o < P; ++o) acc[o] = fma_bcast(in[o + u], t, acc[o]);
    }
}

// Ring reader: row offset `ro` (from the lane's first row y0) → element.
// The ring wraps at most once inside a span; slot0 and RR are multiples of WB
// (4 or 8), so a WB-row block never straddles the wrap and one select serves WB reads.
template <int WB>
struct RingLane {
    const f2 *pA, *pB;
    int wblk; // WB-row blocks before the wrap
    template <int PITCH>
    __device__ __forceinline__ f2 rd(int ro) const
    {
        const f2 *base = ((ro / WB) < wblk) ? pA : pB;
        return base[ro * PITCH];
    }
};

template <int Q, int U, int PITCH, int WB>
__device__ __forceinline__ void col_block(f2 (&acc)[Q], const RingLane<WB> &rl, int k0, tap_ptr taps)
{
    f2 in[Q + U - 1];
#pragma unroll
    for (int i = 0; i < Q + U - 1; ++i) in[i] = rl.template rd<PITCH>(k0 + i);
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const f2 t = taps[k0 + u];
#pragma unroll
        for (int o = 0; o < Q; ++o) acc[o] = fma_pair(in[o + u], t, acc[o]);
    }
}

// Compile-time-length FIR with a sliding register window, in tap blocks of U:
//   acc[o] += Σ_{u<L} ld(o+u) · taps[u],  o < NOUT
// Per block: U taps come in by scalar loads (SGPR operands), the next block's U
// inputs are requested from LDS before this block's NOUT·U packed FMAs run, and a
// scheduling barrier closes the block so that neither the SGPR nor the VGPR live
// ranges of later blocks are pulled forward (the unconstrained schedule keeps all
// L taps live and spills SGPRs through v_readlane).  Every output sees the taps in
// the same order u = 0..L-1, so equal inputs give bit-equal outputs (flat windows
// must tie exactly, like the reference's identical per-pixel loops).
template <int NOUT, int L, int U, typename T, typename Loader, typename Fma>
__device__ __forceinline__ void fir_sliding(f2 (&acc)[NOUT], Loader ld, Fma fma, tap_ptr taps)
{
    constexpr int NB = (L + U - 1) / U;
    T win[NOUT - 1 + U];
    T nxt[U];
    f2 tn[U];
#pragma unroll
    for (int j = 0; j < U; ++j)
        if (j < L) tn[j] = taps[j];
#pragma unroll
    for (int i = 0; i < NOUT - 1; ++i) win[i] = ld(i);
#pragma unroll
    for (int j = 0; j < U; ++j)
        if (j < L) nxt[j] = ld(NOUT - 1 + j);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int u0 = b * U;
        const int nu = (L - u0) < U ? (L - u0) : U;
        f2 t[U];
#pragma unroll
        for (int j = 0; j < U; ++j)
            if (j < nu) { win[NOUT - 1 + j] = nxt[j]; t[j] = tn[j]; }
        // requests for the next block go out before this block's FMAs
#pragma unroll
        for (int j = 0; j < U; ++j)
            if (u0 + U + j < L) { tn[j] = taps[u0 + U + j]; nxt[j] = ld(NOUT - 1 + u0 + U + j); }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (u < nu) {
#pragma unroll
                for (int o = 0; o < NOUT; ++o) acc[o] = fma(win[o + u], t[u], acc[o]);
            }
        }
#pragma unroll
        for (int i = 0; i < NOUT - 1; ++i) win[i] = win[i + nu];
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Hide a uniform pointer from loop-invariant code motion (keeps it in SGPRs).
__device__ __forceinline__ tap_ptr opaque_uniform(const f2 *p)
{
    unsigned long long u = (unsigned long long)p;
    asm volatile("" : "+s"(u));
    return (tap_ptr)u;
}

// P outputs/lane in the row pass, XG lane-groups across the strip (strip width
// TW = P*XG), Q outputs/lane in the column pass, CH input rows per chunk,
// LT = compile-time kernel length (0: runtime L), NT threads.
// taps_row: L pairs (g+[k], g−[k]); taps_col: L pairs (s·g+[k], −s·g−[k]).  They are
// top-level __restrict__ kernel arguments so that the uniform tap loads become scalar
// (s_load → SGPR operands of v_pk_fma_f32) instead of per-lane vector loads.
template <int P, int XG, int Q, int CH, int LT, int NT, bool RESP>
__global__ __launch_bounds__(NT, 2) void dog_window_kernel(const LaunchGeo g, const f2 *__restrict__ taps_row,
                                                           const f2 *__restrict__ taps_col)
{
    constexpr int TW = P * XG;
    constexpr int PR = pitch_r(TW);
    constexpr int NW = NT / 64;
    static_assert(CH % 4 == 0 && Q % 4 == 0, "wrap logic works on 4-row blocks");
    // 8-row wrap blocks when every y0 and the ring length are multiples of 8
    constexpr int WB = (LT && CH % 8 == 0 && Q % 8 == 0 && ring_rows(CH, LT ? LT : 1, Q) % 8 == 0) ? 8 : 4;
    const int L = LT ? LT : g.L;
    const int hw = L >> 1;
    const int TWin = TW + L - 1;
    const int PA = LT ? pitch_a(TW + LT - 1) : g.pitchA;
    const int RR = LT ? ring_rows(CH, LT, Q) : g.RR;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *A = reinterpret_cast<float *>(smem);
    f2 *ring = reinterpret_cast<f2 *>(smem + round_up(CH * PA * 4, 16));

    // XCD-aware block → (window, strip): blocks are dealt round-robin over the 8
    // XCDs, so give each XCD a contiguous range of logical ids; the strips of one
    // window (which share their halo columns) then meet in one L2.  Speed only.
    const int per_xcd = (g.nblocks + 7) >> 3;
    const int logical = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (logical >= g.nblocks) return;
    const int b = logical / g.nstrips;
    const int s = logical - b * g.nstrips;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g1 = g.guesses[2 * b], g2 = g.guesses[2 * b + 1];
    const int fidx = g.frame_index ? g.frame_index[b] : b;
    const uint8_t *__restrict__ frame = g.frames + (long long)fidx * g.frame_stride;
    const int x0 = s * TW;                 // first window column of this strip
    const int ti0 = g1 - g.r1 - 1 - hw;    // 0-based frame row of input row 0
    const int tj0 = g2 - g.r2 - 1 + x0 - hw; // 0-based frame col of input col 0
    const int NA = g.n1 + L - 1;           // input rows
    const int ws = min(TW, g.n2 - x0);     // valid output columns in this strip

    // ---- per-window DC level ----
    // ΣK = 0, so any constant may be subtracted from the pixels before filtering.  The
    // reference's fill value (mode of frame 1) is the natural one: flat background becomes
    // exactly 0 and flat windows tie exactly.  Where the window's content sits far from the
    // fill (textured scenes), the two ~DC-sized Gaussian sums would cancel in FP32 with an
    // error ∝ |DC|; there the rounded mean of a fixed 32×32 sample grid over the WINDOW's
    // padded tile is used instead.  Every strip of a window samples the same pixels, so all
    // its workgroups agree on the level (integer arithmetic, deterministic).
    int dc;
    {
        const int tH = g.n1 + L - 1, tW = g.n2 + L - 1;
        const int wj0 = g2 - g.r2 - 1 - hw; // 0-based frame col of the window tile's col 0
        int sum = 0;
        for (int k = tid; k < 1024; k += NT) {
            const int gi = ti0 + (int)(((long long)(k >> 5) * tH) >> 5);
            const int gj = wj0 + (int)(((long long)(k & 31) * tW) >> 5);
            int v = g.fill;
            if (gi >= 0 && gi < g.fh && gj >= 0 && gj < g.fw) v = frame[(long long)gi * g.row_stride + gj];
            sum += v;
        }
        sum = wave_sum(sum);
        int *ssum = reinterpret_cast<int *>(smem);
        if (lane == 0) ssum[wave] = sum;
        __syncthreads();
        int tot = 0;
        for (int w = 0; w < NW; ++w) tot += ssum[w];
        __syncthreads();
        dc = (tot + 512) >> 10;
        if (abs(dc - g.fill) <= 8) dc = g.fill;
    }

    Peak pk;
    peak_init(pk);
    int y_done = 0;

    for (int c0 = 0; c0 < NA; c0 += CH) {
        // ---- stage: rows [c0, c0+CH) × cols [0, TWin) as f32 (pixel − fill) ----
        for (int r = wave; r < CH; r += NW) {
            const int a = c0 + r, gi = ti0 + a;
            const bool rowok = (a < NA) && (gi >= 0) && (gi < g.fh);
            const uint8_t *src = frame + (long long)gi * g.row_stride;
            for (int c = lane; c < TWin; c += 64) {
                const int gj = tj0 + c;
                int v = g.fill;
                if (rowok && gj >= 0 && gj < g.fw) v = src[gj];
                A[r * PA + c] = (float)(v - dc);
            }
        }
        __syncthreads();
        // ---- row pass: task = (row r, lane-group gx), r fastest ----
        for (int t = tid; t < CH * XG; t += NT) {
            const int r = t % CH, gx = t / CH;
            const float *a = A + r * PA + gx * P;
            f2 acc[P];
#pragma unroll
            for (int o = 0; o < P; ++o) acc[o] = f2{0.f, 0.f};
            if (LT) {
                fir_sliding<P, (LT ? LT : 1), 8, float>(
                    acc, [&](int i) { return a[i]; },
                    [](float v, f2 t, f2 c) { return fma_bcast(v, t, c); }, opaque_uniform(taps_row));
            } else {
                int k0 = 0;
                for (; k0 + 16 <= L; k0 += 16) row_block<P, 16>(acc, a + k0, as_taps(taps_row) + k0);
                for (; k0 + 4 <= L; k0 += 4) row_block<P, 4>(acc, a + k0, as_taps(taps_row) + k0);
                for (; k0 < L; ++k0) row_block<P, 1>(acc, a + k0, as_taps(taps_row) + k0);
            }
            f2 *dst = ring + ((c0 + r) % RR) * PR + gx * P;
#pragma unroll
            for (int o = 0; o < P; ++o) dst[o] = acc[o];
        }
        __syncthreads();
        // ---- column pass over the output rows that became computable ----
        const bool last = (c0 + CH >= NA);
        const int y_avail = last ? g.n1 : (c0 + CH - (L - 1));
        if (y_avail > y_done) {
            const int y_hi = last ? y_avail : (y_done + (y_avail - y_done) / Q * Q);
            const int ngrp = (y_hi - y_done + Q - 1) / Q;
            for (int t = tid; t < ngrp * TW; t += NT) {
                const int yg = t / TW, x = t - yg * TW;
                const int y0 = y_done + yg * Q;
                const int slot0 = y0 % RR;
                RingLane<WB> rl;
                rl.pA = ring + slot0 * PR + x;
                rl.pB = rl.pA - RR * PR;
                rl.wblk = (RR - slot0) / WB;
                f2 acc[Q];
#pragma unroll
                for (int o = 0; o < Q; ++o) acc[o] = f2{0.f, 0.f};
                if (LT) {
                    fir_sliding<Q, (LT ? LT : 1), 8, f2>(
                        acc, [&](int ro) { return rl.template rd<PR>(ro); },
                        [](f2 v, f2 t, f2 c) { return fma_pair(v, t, c); }, opaque_uniform(taps_col));
                } else {
                    int k0 = 0;
                    for (; k0 + 16 <= L; k0 += 16) col_block<Q, 16, PR>(acc, rl, k0, as_taps(taps_col));
                    for (; k0 + 4 <= L; k0 += 4) col_block<Q, 4, PR>(acc, rl, k0, as_taps(taps_col));
                    for (; k0 < L; ++k0) col_block<Q, 1, PR>(acc, rl, k0, as_taps(taps_col));
                }
                {
                    const int lin0 = (x0 + x) * g.n1 + y0; // column-major index in the window
                    const bool colok = x < ws;
                    const int nvalid = colok ? (y_hi - y0) : 0; // outputs o < nvalid count
                    float v[Q];
                    float m = -__builtin_huge_valf();
#pragma unroll
                    for (int o = 0; o < Q; ++o) {
                        v[o] = acc[o].x + acc[o].y;
                        if (RESP && o < nvalid) g.resp[(long long)b * g.n1 * g.n2 + lin0 + o] = v[o];
                        v[o] = (o < nvalid) ? v[o] : -__builtin_huge_valf();
                        m = fmaxf(m, v[o]);
                    }
                    // first maximum in column-major order (findmax, :59); the index search only
                    // runs for a lane whose group reaches its running maximum (rare after warm-up)
                    if (m >= pk.best && nvalid > 0) {
#pragma unroll
                        for (int o = 0; o < Q; ++o)
                            if (o < nvalid) peak_push(pk, v[o], lin0 + o);
                    } else {
                        pk.second = fmaxf(pk.second, m); // every value of the group is below the lane's best
                    }
                }
            }
            y_done = y_hi;
        }
        __syncthreads();
    }

    // ---- peak: wave shuffle reduction, then across waves through LDS ----
    peak_wave_reduce(pk);
    float *sval = reinterpret_cast<float *>(smem);
    int *sidx = reinterpret_cast<int *>(smem + 64);
    float *ssec = reinterpret_cast<float *>(smem + 128);
    if (lane == 0) { sval[wave] = pk.best; sidx[wave] = pk.idx; ssec[wave] = pk.second; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < NW; ++w) peak_merge(pk, sval[w], sidx[w], ssec[w]);
        g.part_val[b * g.nslots + s] = pk.best;
        g.part_idx[b * g.nslots + s] = pk.idx;
        g.part_sec[b * g.nslots + s] = pk.second;
    }
}

#ifndef PDOG_ROLL_INST_ONLY
// mode(_img), src/PawsomeTracker.jl:47, for a frame that already lives on the device.  StatsBase.mode keeps
// the value whose count FIRST exceeds the running maximum while scanning the h×w view column-major.  Every
// value that ends with the maximum count M reaches M at its LAST occurrence, so the winner is: largest count,
// ties → the value whose last occurrence comes earliest in column-major order (index j·h + i).  One pass:
// per-workgroup LDS histogram + last-occurrence table, flushed with atomics; 2 KB go back to the host.
static __global__ __launch_bounds__(256) void dog_mode_kernel(const uint8_t *__restrict__ img, int h, int w, long long row_stride,
                                                       unsigned *__restrict__ hist, unsigned *__restrict__ last)
{
    __shared__ unsigned shist[256], slast[256];
    shist[threadIdx.x] = 0;
    slast[threadIdx.x] = 0;
    __syncthreads();
    const long long total = (long long)h * w;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += (long long)gridDim.x * blockDim.x) {
        const int i = (int)(p / w), j = (int)(p - (long long)i * w);
        const unsigned v = img[(long long)i * row_stride + j];
        atomicAdd(&shist[v], 1u);
        atomicMax(&slast[v], (unsigned)((long long)j * h + i) + 1u); // +1: 0 means "never seen"
    }
    __syncthreads();
    if (shist[threadIdx.x]) {
        atomicAdd(&hist[threadIdx.x], shist[threadIdx.x]);
        atomicMax(&last[threadIdx.x], slast[threadIdx.x]);
    }
}

#endif // PDOG_ROLL_INST_ONLY

} // namespace pdog
