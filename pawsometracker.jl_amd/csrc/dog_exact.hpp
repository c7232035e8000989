// dog_exact.hpp — positions that are the reference's BY CONSTRUCTION, not by observation.
//
// The reference ranks Float64 dense sums: buff[I] = Σ_J Float64(img[I+J])·K[J], accumulated sequentially in kernel
// column-major order, then findmax (/root/reference/src/PawsomeTracker.jl:57-59).  The kernels of this library
// rank FP32 separable sums.  Exact ties are settled identically by construction; NEAR ties are not: when the two
// best responses of a window differ by less than the FP32 evaluation error, FP32 may crown the wrong pixel.
//
// Guarantee.  Let F(p) be the reference's Float64 value of pixel p and f(p) the FP32 value of the kernel that ran.
// |f(p) − F(p)| ≤ δ for every p, with an a-priori bound that follows THE KERNEL'S OWN OPERATION ORDER (u = 2⁻²⁴,
// V = max |pixel − dc| ≤ 255).  An FMA chain ŝ_i = fl(ŝ_{i−1} + a_i·b̂_i) rounds each partial sum once:
//     |ŝ_n − s_n| ≤ u·(1 + u)·Σ_i |ŝ_i|,   |ŝ_i| ≤ V·W_i·(1 + nu),   W_i = Σ_{j ≤ i} |b_j|·max|a_j| / V
// — the cumulative tap weight in the order the taps are added.  The Gaussians sum to 1 and every kernel adds the kernel's
// edge taps first, so Σ_i W_i is a fraction of the chain length n that the order-blind bound n·u·V of round 2 charged:
//     row pass     symmetric pairs from the edge inwards, centre last, v_a + v_b exact:  W_i = Σ_{j ≤ i} 2g[j] (then + g[H]);
//                  the taps are rounded once (+u·V)                                            ⇒ |R̂± − R±| ≤ u·V·(Σ_i W_i + 1)
//     column pass  terms bounded by |ĉ±|·|R̂±| ≤ c±·V/255: the same sum over the column taps' cumulative weights — per channel
//                  where the two Gaussians keep separate chains and are added at the end (+2 for that addition and the rounded
//                  taps), both weights per step where one f32 takes the (+, −) terms alternately (roll kernels); plus the row
//                  errors times Σ|c±| = 1/255 each
//     ⇒ δ = u·(V/255)·F·1.02, F evaluated numerically over the tracker's own Float64 taps per kernel FAMILY (pawsome_dog.hip,
//       exact_factors): l = 65: F = 157 (roll), 94 (fused, tiled), 138 (ring) against 6l + 4 = 394; the two-pass kernels add
//       every chain of one register-ring trip from zero and sum the chains (dog_twopass.hpp): l = 293: F = 72 against 1762.
// The reference's own Float64 rounding (≤ l²·2⁻⁵³·2·V/255) is negligible beside it.  The FLAG uses V = 255 (δ(l = 65) = 9.5e-6
// for the roll kernels against a typical peak of 0.09 and a typical peak-to-neighbour gap of 4.7e-4) or, on the two-pass path,
// the window's own V, which its row pass collects.  The REFINEMENT of a flagged window may take the window's own
// V = max |pixel − dc| over its padded tile (refine_window, `tighten`): δ is proportional to it, so a window of ±2-level noise
// (V ≈ 3) has an 85× smaller T — a handful of candidates instead of thousands, or no near-tie at all (the flag is withdrawn
// and the FP32 argmax stands).  Values the refinement RECOMPUTES in FP32 (plain chains: F_rescan) are compared with the main
// kernel's maximum under δ_main + δ_rescan (ExactCtl::T_rescan), the main kernel's own values under 2δ_main (ExactCtl::T).
//   1. Every main kernel also tracks the RUNNER-UP value of its window (Peak, dog_kernels.hpp).  If best − runner-up > 2δ
//      the FP32 argmax is the reference's argmax (the true argmax p* has f(p*) ≥ F(p*) − δ ≥ F(p̂) − δ ≥ f(p̂) − 2δ,
//      so it is p̂ itself) and nothing else happens: ≈99 % of blob windows.
//   2. Otherwise the window is REFINED: its FP32 response is recomputed, every pixel with f ≥ max − 2δ (this set
//      contains p* and every pixel the reference ties with it) is re-evaluated as the reference does it — dense
//      l×l, Float64, no contraction, kernel column-major order, K = dir·(g₊⊗g₊ − g₋⊗g₋) built in Float64 on the
//      host, pixel/255.0 by division — and the first maximum in column-major order of THOSE values wins.
// Flat tiles (every response exactly equal in both arithmetics) would make every pixel a candidate; all their
// candidates evaluate to the same Float64 value, so the answer is right, only slow.  Cost is otherwise a few
// dozen dependent 4225-term chains per refined window.
#pragma once
#include "dog_kernels.hpp"
#include <type_traits>

namespace pdog {

typedef const double __attribute__((address_space(4))) *k64_ptr; // uniform Float64 kernel reads → scalar loads

// tmp + a·b with the product and the sum rounded separately (never an FMA): `tmp += a * b` as Julia evaluates it.
__device__ __forceinline__ double exact_mac(double tmp, double a, double b)
{
#pragma clang fp contract(off)
    const double prod = a * b;
    return tmp + prod;
}

// The reference's value of ONE pixel: dense l×l Float64 correlation over the patch whose first pixel is `patch`
// (rows `pitch` bytes apart), accumulated from 0.0 in kernel column-major order, products and sums rounded separately
// (:57).  lut[p] = p / 255.0 (FixedPointNumbers N0f8 → Float64).  Runs in whatever lanes call it.  The pixel and
// table reads of 8 terms are issued together; only the additions form the dependent chain.
// HIP's __dmul_rn/__dadd_rn are plain operators, and hipcc contracts a·b + c into an FMA by default: exact_mac keeps
// the product and the sum apart like the reference's `tmp += a*b`.
template <typename PixelPtr>
__device__ __forceinline__ double exact_patch(PixelPtr patch, long long pitch, int L, k64_ptr K, const double *lut)
{
    double tmp = 0.0;
    for (int kj = 0; kj < L; ++kj) {
        const PixelPtr col = patch + kj;
        const k64_ptr kc = K + (long long)L * kj;
        int ki = 0;
        for (; ki + 8 <= L; ki += 8) {
            double a[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] = lut[col[(long long)(ki + u) * pitch]];
#pragma unroll
            for (int u = 0; u < 8; ++u) tmp = exact_mac(tmp, a[u], kc[ki + u]);
        }
        for (; ki < L; ++ki) tmp = exact_mac(tmp, lut[col[(long long)ki * pitch]], kc[ki]);
    }
    return tmp;
}
// The same for a patch read from the frame itself (no staged tile): PaddedView semantics (:48) per pixel when the patch
// leaves the frame.  (i0, j0): 0-based frame coordinates of the patch's first row / column.
__device__ __forceinline__ double exact_pixel(const uint8_t *__restrict__ frame, long long row_stride, int fh, int fw, int fill,
                                              int i0, int j0, int L, k64_ptr K, const double *lut)
{
    if (i0 >= 0 && i0 + L <= fh && j0 >= 0 && j0 + L <= fw)
        return exact_patch(frame + (long long)i0 * row_stride + j0, row_stride, L, K, lut);
    double tmp = 0.0;
    for (int kj = 0; kj < L; ++kj) {
        const int gj = j0 + kj;
        const bool colok = gj >= 0 && gj < fw;
        const k64_ptr kc = K + (long long)L * kj;
        for (int ki = 0; ki < L; ++ki) {
            const int gi = i0 + ki;
            int px = fill; // PaddedView, :48
            if (colok && gi >= 0 && gi < fh) px = frame[(long long)gi * row_stride + gj];
            tmp = exact_mac(tmp, lut[px], kc[ki]);
        }
    }
    return tmp;
}

struct Peak64 {
    double best;
    int idx;
};
__device__ __forceinline__ void peak64_push(Peak64 &p, double v, int lin)
{
    if (v > p.best || (v == p.best && lin < p.idx)) { p.best = v; p.idx = lin; }
}
__device__ __forceinline__ void peak64_wave_reduce(Peak64 &p)
{
    for (int off = 32; off > 0; off >>= 1) {
        const double ov = __shfl_down(p.best, off, 64);
        const int oi = __shfl_down(p.idx, off, 64);
        peak64_push(p, ov, oi);
    }
}

// ---- refinement of one window by one workgroup ----
// Three stages, each cheaper to reach than the next is to run:
//   1. FP32 rescan of the column blocks that can hold a pixel within T = 2δ of the window's FP32 maximum M (the main
//      kernels' per-strip / per-block partial maxima say which: a block whose partial maximum is below M − T cannot):
//      a deliberately plain separable evaluation (any evaluation within δ serves, see the header) → the CANDIDATES.
//   2. Every candidate is evaluated in SEPARABLE Float64 (row sums of both Gaussians in double for the block's
//      columns, then 2l FMAs per candidate): no ordering constraint, so it parallelises, and its error against the
//      exact value — like the reference's own (l² sequential roundings) — is below δ64 = 2⁻⁵³·(2.1 l² + 8 l + 64).
//      A candidate more than T64 = 2δ64 below the best of these values cannot be the reference's maximum.
//   3. The SURVIVORS — one, unless the data holds a genuine near-tie — are evaluated exactly as the reference does it
//      (exact_pixel: dense l×l, sequential, kernel column-major order); a single survivor needs no evaluation at all.
// More candidates than the list holds (plateaus: every response exactly equal) → stage 2 is skipped and every
// candidate goes through stage 3 as it is found (right, only slow).
constexpr int REFINE_CAP = 512;    // candidate list entries
constexpr int REFINE_BLKCAP = 256; // column blocks listed for the rescan (more ⇒ every block is rescanned)
__host__ __device__ constexpr int refine_tile_pitch(int ncol, int L) { return (ncol + L - 1 + 3) / 4 * 4; }
// dynamic LDS of a refinement: fixed part (table p/255.0, candidate list, reductions), row-pass block (f2 for stage 1,
// reused as 2 doubles per element for stage 2), optional pixel tile
__host__ __device__ constexpr size_t refine_fixed_bytes() { return 256 * 8 + REFINE_CAP * 12 + 16 * 8 + 16 * 4 + 32 + REFINE_BLKCAP * 4 + 32; }
__host__ __device__ constexpr size_t refine_r_bytes(int n1, int L, int cbw) { return (size_t)(n1 + L - 1) * cbw * 16; }
__host__ __device__ constexpr size_t refine_tile_bytes(int rows, int L, int cbw) { return ((size_t)rows * refine_tile_pitch(cbw, L) + 15) / 16 * 16; }
// tile_rows: rows of the block's pixel tile resident at a time (n1 + l − 1 = all of them)
__host__ __device__ constexpr size_t refine_lds_bytes(int n1, int L, int cbw, int tile_rows)
{
    return refine_fixed_bytes() + refine_r_bytes(n1, L, cbw) + refine_tile_bytes(tile_rows, L, cbw);
}

// bits [a, b) of a 64-column strip's lane mask (clipped to the strip)
__device__ __forceinline__ unsigned long long column_bits(int a, int b)
{
    a = max(a, 0);
    b = min(b, 64);
    if (b <= a) return 0ull;
    const unsigned long long hi = b >= 64 ? ~0ull : ((1ull << b) - 1ull);
    return hi & ~((1ull << a) - 1ull);
}

// The refinement's constants, in device memory: the kernels that refine inline (fused, chain) carry ONE pointer to it and
// read the fields inside the rare branch — as kernel arguments they would sit in SGPRs across the whole frame loop.
struct RefineParams {
    const double *K64;   // dense Float64 kernel, l×l column-major (:41-43)
    const double *g64;   // [2][l] Float64 Gaussians
    double dir, T64;
};
typedef const RefineParams __attribute__((address_space(4))) *refine_params_ptr;

struct RefineCtx {
    tap_ptr trow, tcol;  // FP32 taps: (g₊, g₋)[k] and (s·g₊, −s·g₋)[k]
    k64_ptr K;           // dense Float64 kernel, l×l column-major (:41-43)
    k64_ptr g64;         // [2][l]: the normalised Gaussians σ and √2σ in Float64
    double dir;          // direction, :42
    double T64;          // 2δ64
    float T;             // 2δ for |pixel − dc| ≤ 255: the main kernel's own values against its own maximum (the flag, the response map)
    float T_rescan = 0.f; // δ_main + δ_plain: values RECOMPUTED here with plain chains against the main kernel's maximum; 0 = T (main kernel
                          // with plain chains too).  The two-pass kernels accumulate in blocks (dog_twopass.hpp): their δ is ≈13× smaller
    int vmax_known = -1;  // ≥ 0: the window's own V = max |pixel − dc| (the two-pass row pass collects it): thresholds scale with V/255 from the start
    float second;        // the window's FP32 runner-up value and the FP32 argmax (column-major index): with the window's own
    int fp32_idx;        // |pixel − dc| bound the flag may turn out unnecessary (fp32_idx < 0: not supplied, never withdrawn)
    int v_after = 64;    // map path: candidates of the first scan beyond which the window's own V is worth its pass over the tile
    int cbw, tile_rows;  // window columns per block; tile rows resident at a time
    unsigned char *lds;  // refine_lds_bytes(n1, l, cbw, tile_rows) bytes, 16-aligned
};

// `may(x0, x1, thr)`: false only if no pixel of window columns [x0, x1) can reach thr (M − T, or M − the window's own T).
// The block's pixels go through an LDS tile of c.tile_rows rows × refine_tile_pitch bytes, staged with coalesced dword
// loads: all NA rows at once when they fit (short kernels), otherwise slice by slice (a thread per row reading its own
// row from memory was 10× slower: 64 cache lines per load instruction).  UNR: loads in flight per thread while staging.
// Returns the window's answer (column-major index) in thread 0.
// `map`: null, or the window's FP32 response as the main kernel wrote it (n1·n2 floats, column-major — the two-pass path
// in exact mode): the candidates are then read off it instead of being recomputed (stage 1), and stage 2 only covers
// the candidates' columns and the tile rows their sums reach.
template <int UNR, typename May>
__device__ __forceinline__ int refine_window(const int NT, const LaunchGeo &g, const uint8_t *__restrict__ frame, int g1, int g2, float M,
                                             const RefineCtx &c, May may, const float *__restrict__ map = nullptr)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, NW = NT / 64;
    const int L = g.L, hw = L >> 1, NA = g.n1 + L - 1;
    const int ti0 = g1 - g.r1 - 1 - hw, wj0 = g2 - g.r2 - 1 - hw;
    double *lut = reinterpret_cast<double *>(c.lds);
    double *cand_val = lut + 256;
    double *dred = cand_val + REFINE_CAP;
    int *cand_lin = reinterpret_cast<int *>(dred + 16);
    int *ired = cand_lin + REFINE_CAP;
    int *cnt = ired + 16; // [0] candidates found, [1] list overflowed, [2] survivors, [3] answer, [4] column blocks to rescan
    int *blk = cnt + 8;   // [REFINE_BLKCAP] first columns of the blocks to rescan
    int *mm = blk + REFINE_BLKCAP; // [4] map path: candidates' column range, a column group's row range
    unsigned char *rbase = c.lds + refine_fixed_bytes();
    f2 *R32 = reinterpret_cast<f2 *>(rbase);
    double *R64 = reinterpret_cast<double *>(rbase);
    uint8_t *const tile = rbase + refine_r_bytes(g.n1, L, c.cbw);
    const int RS = min(c.tile_rows, NA);       // tile rows resident at a time
    const bool full = RS >= NA;                // the whole block's pixels stay in LDS

#ifdef PDOG_ABLATIONS
    unsigned long long stamp_prev = __builtin_readcyclecounter();
#define PDOG_STAMP(i) do { if (tid == 0) { const unsigned long long now_ = __builtin_readcyclecounter(); atomicAdd(g.ex.stat + 8 + (i), now_ - stamp_prev); stamp_prev = now_; } } while (0)
#else
#define PDOG_STAMP(i) do { } while (0)
#endif
    for (int p = tid; p < 256; p += NT) lut[p] = (double)p / 255.0;
    if (tid < 8) cnt[tid] = 0;
    // DC level: the same fixed sample grid as the main kernels (any level in 0…255 keeps the bound)
    int sum = dc_sample_sum(g, frame, ti0, wj0, L, tid, NT);
    sum = wave_sum(sum);
    if (lane == 0) ired[wave] = sum;
    __syncthreads();
    int tot = 0;
    for (int w = 0; w < NW; ++w) tot += ired[w];
    const int dc = dc_from_sum(tot, g.fill);
    // The window's own V = max |pixel − dc| over its padded tile.  δ is proportional to V (header), and the flag was raised
    // with V = 255: a window of ±2-level noise has V ≈ 3, its T is 85× smaller, and instead of thousands of pixels "within T
    // of the maximum" (each a 4 225-term chain in the worst case: 10–26 ms per window measured) it has a handful, or the
    // flag falls away altogether.  One pass over the tile (clamped dword loads, fill selected afterwards), ≈10 µs.
    // Taken up front where the candidates have to be recomputed (no response map); with a map only when the first scan
    // finds more than a few dozen candidates — on a long kernel's tile the pass costs as much as it saves otherwise (cfg5:
    // 247 KB per window, 36 candidates).
    const float T_r = c.T_rescan > 0.f ? c.T_rescan : c.T;
    float thr = M - c.T, thr_r = M - T_r; // thr: values of the main kernel (partials, response map); thr_r: values recomputed here
    bool have_v = false;
    if (c.vmax_known >= 0 && c.T < __builtin_huge_valf()) { // the window's own V is known: δ is proportional to it
        if (c.vmax_known == 0 && c.fp32_idx >= 0) return c.fp32_idx; // flat tile: every response exactly equal in both arithmetics
        const float sc = (float)c.vmax_known * (1.0f / 255.0f) * 1.00001f;
        thr = M - c.T * sc;
        thr_r = M - T_r * sc;
        have_v = true;
    }
    auto tighten = [&]() -> bool { // true: the flag is withdrawn, the FP32 argmax stands
        float T_eff = c.T;
        have_v = true;
        const int TW = g.n2 + L - 1, tq = (TW + 3) >> 2, total = NA * tq;
        int vmax = 0;
        if (g.fw >= 4) {
            int a = tid / tq, q = tid - a * tq;
            const int da = NT / tq, dq = NT - da * tq;
            for (int e0 = tid; e0 < total; e0 += 8 * NT) {
                uint32_t w[8];
                int ra[8], rq[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    ra[u] = a;
                    rq[u] = q;
                    const bool ok = e0 + u * NT < total;
                    const int gi = ti0 + (ok ? a : 0), gj = wj0 + 4 * (ok ? q : 0);
                    __builtin_memcpy(&w[u], frame + (long long)min(max(gi, 0), g.fh - 1) * g.row_stride + min(max(gj, 0), g.fw - 4), 4);
                    a += da;
                    q += dq;
                    if (q >= tq) { q -= tq; ++a; }
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (e0 + u * NT >= total) continue;
                    const int gi = ti0 + ra[u], gj = wj0 + 4 * rq[u], gjc = min(max(gj, 0), g.fw - 4);
                    const bool rowok = gi >= 0 && gi < g.fh;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int gjj = gj + i;
                        if (4 * rq[u] + i >= TW) continue;
                        const int px = (rowok && gjj >= 0 && gjj < g.fw) ? (int)((w[u] >> (8 * ((gjj - gjc) & 3))) & 0xffu) : g.fill;
                        vmax = max(vmax, abs(px - dc));
                    }
                }
            }
        } else {
            for (int e = tid; e < NA * TW; e += NT) {
                const int a = e / TW, cc = e - a * TW, gi = ti0 + a, gj = wj0 + cc;
                const int px = (gi >= 0 && gi < g.fh && gj >= 0 && gj < g.fw) ? (int)frame[(long long)gi * g.row_stride + gj] : g.fill;
                vmax = max(vmax, abs(px - dc));
            }
        }
        for (int off = 32; off > 0; off >>= 1) vmax = max(vmax, __shfl_xor(vmax, off, 64));
        __syncthreads(); // (ired: everyone has read the sample sums)
        if (lane == 0) ired[wave] = vmax;
        __syncthreads();
        for (int w = 0; w < NW; ++w) vmax = max(vmax, ired[w]);
        __syncthreads();
        if (c.T < __builtin_huge_valf()) T_eff = c.T * ((float)vmax * (1.0f / 255.0f)) * 1.00001f; // (pdog_set_exact(t, 2): T = ∞ stays)
        thr = M - T_eff;
        thr_r = c.T < __builtin_huge_valf() ? M - T_r * ((float)vmax * (1.0f / 255.0f)) * 1.00001f : thr;
        // flat tile (every pixel = dc: all responses exactly equal in both arithmetics) or a runner-up further than the
        // window's own T below the maximum: the FP32 argmax is the reference's
        if (c.fp32_idx >= 0 && c.T < __builtin_huge_valf() && (vmax == 0 || M - c.second > T_eff)) {
            if (tid == 0) atomicAdd(g.ex.stat + 4, 1ull);
            return true;
        }
        return false;
    };
    // which column blocks can hold a candidate: asked in parallel, once (a serial scan of a 513-column window's 257
    // blocks cost 90 µs); blocks beyond the list's capacity are all rescanned
    const int nblk_all = (g.n2 + c.cbw - 1) / c.cbw;
    auto list_blocks = [&]() {
        for (int cb = tid; cb < nblk_all; cb += NT)
            if (may(cb * c.cbw, min(g.n2, (cb + 1) * c.cbw), thr)) {
                const int k = atomicAdd(&cnt[4], 1);
                if (k < REFINE_BLKCAP) blk[k] = cb * c.cbw;
            }
        if (tid == 0) { mm[0] = 0x7fffffff; mm[1] = -1; }
        __syncthreads();
    };
    list_blocks();
    // no map, many blocks to recompute (a hard window: noise only, a faint target): the window's own V first, and the list
    // again under the tighter threshold — a window with one or two blocks listed (cfg4's one flagged window in 1 024) is
    // recomputed sooner than its 333 KB tile is scanned
    if (!map && cnt[4] > 4 && !have_v) {
        __syncthreads();
        if (tighten()) return c.fp32_idx;
        if (tid == 0) cnt[4] = 0;
        __syncthreads();
        list_blocks();
    }
    const bool blk_all = cnt[4] > REFINE_BLKCAP;
    const int nblk = blk_all ? nblk_all : cnt[4];
    // with a response map: the candidates are read straight off it — every pixel of the listed blocks with f ≥ M − T —
    // together with the column range they span
    constexpr int MAP_KPT = 4; // candidates per thread in the map path's stage 2
    bool use_map = map != nullptr && REFINE_CAP <= MAP_KPT * NT;
    auto scan_map = [&]() {
        const int npx = g.n1 * g.n2, bw = c.cbw * g.n1, tot = nblk * bw;
        for (int e0 = 0; e0 < tot; e0 += 8 * NT) {
            float v[8];
            int ee[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = e0 + u * NT + tid;
                const int bi = i / bw, r = i - bi * bw;
                const int e = (i < tot ? (blk_all ? bi * c.cbw : blk[bi]) : 0) * g.n1 + r;
                const bool ok = i < tot && e < npx; // (the last block of a window may be narrower)
                ee[u] = e;
                v[u] = ok ? map[ok ? e : 0] : -__builtin_huge_valf();
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (v[u] >= thr) {
                    const int k = atomicAdd(&cnt[0], 1);
                    if (k < REFINE_CAP) cand_lin[k] = ee[u]; else cnt[1] = 1;
                    const int x = ee[u] / g.n1;
                    atomicMin(&mm[0], x);
                    atomicMax(&mm[1], x);
                }
        }
        __syncthreads();
    };
    if (use_map) {
        scan_map();
        if ((cnt[1] || cnt[0] > c.v_after) && !have_v) { // many pixels within the V = 255 threshold: the window's own V, then once more
            __syncthreads();
            if (tighten()) return c.fp32_idx;
            if (tid == 0) { cnt[0] = 0; cnt[1] = 0; mm[0] = 0x7fffffff; mm[1] = -1; }
            __syncthreads();
            scan_map();
        }
        if (cnt[1]) { // still more candidates than the list holds (a plateau): the general path below deals with it
            __syncthreads();
            if (tid == 0) { cnt[0] = 0; cnt[1] = 0; }
            use_map = false;
            __syncthreads();
        }
    }
    PDOG_STAMP(0);

    // rows [a0, a0 + rows) of the block's tile → LDS: a dword (4 pixels) per item, loaded unconditionally at an address
    // clamped into the frame (a clamped dword still holds every in-frame byte its item needs, at a shifted position),
    // the PaddedView fill (:48) selected afterwards — no branch between two loads, UNR of a thread's loads in flight
    auto stage = [&](int x0, int tp, int a0, int rows) {
        const int tq = tp >> 2, total = rows * tq;
        const int gi0 = ti0 + a0, gj0 = wj0 + x0;
        // Batches of UNR items per thread: first every load (a surplus item of the last batch loads item 0's address), then
        // the stores — written as one loop with an exit test per item, each load waited for its own store (s_waitcnt vmcnt(0)
        // after every global_load: 57 µs for a 90 KB tile).  (16-byte items — unaligned `global_load_dwordx4`, a quarter of
        // the load instructions — were measured too: the finishing kernel went from 0.88 to 1.46 ms on cfg5.)
        if (g.fw >= 4) {
            const bool inside = gi0 >= 0 && gi0 + rows <= g.fh && gj0 >= 0 && gj0 + tp <= g.fw; // the usual case: a plain dword copy
            const uint8_t *base = frame + (long long)min(max(gi0, 0), g.fh - 1) * g.row_stride + min(max(gj0, 0), g.fw - 4);
            int a = tid / tq, q = tid - a * tq; // (row, dword) of this thread's next item, stepped without a division
            const int da = NT / tq, dq = NT - da * tq;
            for (int e0 = tid; e0 < total; e0 += UNR * NT) {
                uint32_t w[UNR];
                int ra[UNR], rq[UNR];
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    const bool ok = e0 + u * NT < total;
                    ra[u] = a;
                    rq[u] = q;
                    const uint8_t *src = base;
                    if (inside) {
                        if (ok) src = base + (long long)a * g.row_stride + 4 * q;
                    } else if (ok) { // clamped into the frame: a clamped dword still holds every in-frame byte its item needs, at a shifted position
                        const int gi = gi0 + a, gj = gj0 + 4 * q;
                        src = frame + (long long)min(max(gi, 0), g.fh - 1) * g.row_stride + min(max(gj, 0), g.fw - 4);
                    }
                    __builtin_memcpy(&w[u], src, 4);
                    a += da;
                    q += dq;
                    if (q >= tq) { q -= tq; ++a; }
                }
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    if (e0 + u * NT >= total) continue;
                    uint32_t o = w[u];
                    if (!inside) { // the PaddedView fill (:48) for the bytes outside the frame
                        const int gi = gi0 + ra[u], gj = gj0 + 4 * rq[u];
                        const int gjc = min(max(gj, 0), g.fw - 4);
                        const bool rowok = gi >= 0 && gi < g.fh;
                        o = 0;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int gjj = gj + i;
                            const uint32_t px = (rowok && gjj >= 0 && gjj < g.fw) ? ((w[u] >> (8 * ((gjj - gjc) & 3))) & 0xffu) : (uint32_t)g.fill;
                            o |= px << (8 * i);
                        }
                    }
                    *reinterpret_cast<uint32_t *>(tile + ra[u] * tp + 4 * rq[u]) = o;
                }
            }
        } else {
            for (int e = tid; e < rows * tp; e += NT) {
                const int a = e / tp, cc = e - a * tp;
                const int gi = ti0 + a0 + a, gj = wj0 + x0 + cc;
                tile[e] = (gi >= 0 && gi < g.fh && gj >= 0 && gj < g.fw) ? frame[(long long)gi * g.row_stride + gj] : (uint8_t)g.fill;
            }
        }
        __syncthreads();
    };

    // direct = false: stages 1 + 2 (candidates → list with separable Float64 values); direct = true: every candidate
    // straight to stage 3 (list overflow).  Returns the direct mode's running peak of this thread.
    auto sweep = [&](bool direct) {
        Peak64 pk;
        pk.best = -__builtin_huge_val();
        pk.idx = 0x7fffffff;
        for (int bi = 0; bi < nblk; ++bi) {
            const int x0 = blk_all ? bi * c.cbw : blk[bi];
            const int ncol = min(c.cbw, g.n2 - x0);
            if (tid == 0 && !direct) atomicAdd(g.ex.stat + 1, 1ull); // column blocks rescanned (diagnostics)
            const int tp = refine_tile_pitch(ncol, L);
            // stage 1, row pass: R±[a][x] = Σ_k ĝ±[k]·(pixel − dc), k ascending; 8 pixel reads and 8 taps requested
            // together, only the FMAs form a chain
            for (int a0 = 0; a0 < NA; a0 += RS) {
                const int rows = min(RS, NA - a0);
                stage(x0, tp, a0, rows);
                // a thread = FOUR adjacent outputs of a row (a lone wave per SIMD runs a dependent FMA chain at ≈9 cycles per step,
                // and one output per thread is one chain: 75 µs per refined 257×257 window at l = 109, 8× the arithmetic):
                // four independent chains, every pixel converted once per block for the outputs that share it.  Per output
                // still k ascending, one FMA per tap: the same values bit for bit.
                const int ngr = (ncol + 3) >> 2;
                for (int e = tid; e < rows * ngr; e += NT) {
                    const int a = e / ngr, x = 4 * (e - a * ngr);
                    const uint8_t *src = tile + a * tp + x; // (outputs x + j ≥ ncol are computed from whatever follows and dropped)
                    f2 acc[4] = {f2{0.f, 0.f}, f2{0.f, 0.f}, f2{0.f, 0.f}, f2{0.f, 0.f}};
                    int k = 0;
                    for (; k + 8 <= L; k += 8) {
                        float v[11];
#pragma unroll
                        for (int u = 0; u < 11; ++u) v[u] = (float)((int)src[k + u] - dc);
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const f2 t = c.trow[k + u];
#pragma unroll
                            for (int j = 0; j < 4; ++j) acc[j] = fma_bcast(v[u + j], t, acc[j]);
                        }
                    }
                    for (; k < L; ++k) {
                        const f2 t = c.trow[k];
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[j] = fma_bcast((float)((int)src[k + j] - dc), t, acc[j]);
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (x + j < ncol) R32[(a0 + a) * ncol + x + j] = acc[j];
                }
                __syncthreads();
            }
            PDOG_STAMP(2);
            const int first = min(cnt[0], REFINE_CAP);
            __syncthreads();
            // stage 1, column pass: the candidates
            // (four consecutive rows of a column per thread, as above: per output t ascending, the + term then the − term)
            const int ngy = (g.n1 + 3) >> 2;
            for (int e = tid; e < ngy * ncol; e += NT) {
                const int x = e / ngy, yb = 4 * (e - x * ngy);
                float acc4[4] = {0.f, 0.f, 0.f, 0.f};
                int t = 0;
                for (; t + 8 <= L; t += 8) {
                    f2 r[11];
#pragma unroll
                    for (int u = 0; u < 11; ++u) r[u] = R32[min(yb + t + u, NA - 1) * ncol + x];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const f2 w = c.tcol[t + u];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            acc4[j] = __builtin_fmaf(r[u + j].x, w.x, acc4[j]);
                            acc4[j] = __builtin_fmaf(r[u + j].y, w.y, acc4[j]);
                        }
                    }
                }
                for (; t < L; ++t) {
                    const f2 w = c.tcol[t];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f2 r = R32[min(yb + t + j, NA - 1) * ncol + x];
                        acc4[j] = __builtin_fmaf(r.x, w.x, acc4[j]);
                        acc4[j] = __builtin_fmaf(r.y, w.y, acc4[j]);
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                const int y = yb + j;
                const float acc = acc4[j];
                if (y < g.n1 && acc >= thr_r) {
                    const int lin = (x0 + x) * g.n1 + y;
                    if (direct) {
                        const double F = full ? exact_patch((const uint8_t *)(tile + y * tp + x), (long long)tp, L, c.K, lut)
                                              : exact_pixel(frame, g.row_stride, g.fh, g.fw, g.fill, ti0 + y, wj0 + x0 + x, L, c.K, lut);
                        peak64_push(pk, F, lin);
                    } else {
                        const int k = atomicAdd(&cnt[0], 1);
                        if (k < REFINE_CAP) cand_lin[k] = lin; else cnt[1] = 1;
                    }
                }
                }
            }
            __syncthreads();
            PDOG_STAMP(3);
            const int last = min(cnt[0], REFINE_CAP);
            if (!direct && last > first && !cnt[1]) {
                // stage 2: both Gaussians' row sums in Float64 for this block's columns (R32 is dead: same memory) …
                for (int a0 = 0; a0 < NA; a0 += RS) {
                    const int rows = min(RS, NA - a0);
                    if (!full) stage(x0, tp, a0, rows); // (a fully resident tile is still there)
                    // (two adjacent outputs of a row per thread: four independent Float64 chains instead of two; per output k ascending)
                    const int ngr2 = (ncol + 1) >> 1;
                    for (int e = tid; e < rows * ngr2; e += NT) {
                        const int a = e / ngr2, x = 2 * (e - a * ngr2);
                        const uint8_t *src = tile + a * tp + x;
                        double sp[2] = {0.0, 0.0}, sm[2] = {0.0, 0.0};
                        int k = 0;
                        for (; k + 8 <= L; k += 8) {
                            double v[9];
#pragma unroll
                            for (int u = 0; u < 9; ++u) v[u] = lut[src[k + u]];
#pragma unroll
                            for (int u = 0; u < 8; ++u) {
                                const double gp = c.g64[k + u], gm = c.g64[L + k + u];
#pragma unroll
                                for (int j = 0; j < 2; ++j) {
                                    sp[j] = __builtin_fma(gp, v[u + j], sp[j]);
                                    sm[j] = __builtin_fma(gm, v[u + j], sm[j]);
                                }
                            }
                        }
                        for (; k < L; ++k) {
                            const double gp = c.g64[k], gm = c.g64[L + k];
#pragma unroll
                            for (int j = 0; j < 2; ++j) {
                                const double v = lut[src[k + j]];
                                sp[j] = __builtin_fma(gp, v, sp[j]);
                                sm[j] = __builtin_fma(gm, v, sm[j]);
                            }
                        }
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            if (x + j < ncol) {
                                R64[2 * ((a0 + a) * ncol + x + j)] = sp[j];
                                R64[2 * ((a0 + a) * ncol + x + j) + 1] = sm[j];
                            }
                    }
                    __syncthreads();
                }
                PDOG_STAMP(4);
                // … and the new candidates' values dir·(Σ g₊[t]·R₊[y+t] − Σ g₋[t]·R₋[y+t])
                for (int k = first + tid; k < last; k += NT) {
                    const int lin = cand_lin[k];
                    const int xw = lin / g.n1, y = lin - xw * g.n1, x = xw - x0;
                    double sp = 0.0, sm = 0.0;
                    int t = 0;
                    for (; t + 8 <= L; t += 8) {
                        double rp[8], rm[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) { rp[u] = R64[2 * ((y + t + u) * ncol + x)]; rm[u] = R64[2 * ((y + t + u) * ncol + x) + 1]; }
#pragma unroll
                        for (int u = 0; u < 8; ++u) { sp = __builtin_fma(c.g64[t + u], rp[u], sp); sm = __builtin_fma(c.g64[L + t + u], rm[u], sm); }
                    }
                    for (; t < L; ++t) {
                        sp = __builtin_fma(c.g64[t], R64[2 * ((y + t) * ncol + x)], sp);
                        sm = __builtin_fma(c.g64[L + t], R64[2 * ((y + t) * ncol + x) + 1], sm);
                    }
                    cand_val[k] = c.dir * (sp - sm);
                }
                PDOG_STAMP(5);
            }
            __syncthreads();
        }
        return pk;
    };

    // map path, stage 2.  The candidates' columns in groups of ≤ 16; per group the tile rows its candidates' sums reach,
    // slice by slice: pixels → LDS once, Float64 row sums of both Gaussians for the group's columns (symmetric taps: the
    // two pixels of a tap pair are added as integers — exact — so a pair costs one conversion and two FMAs), and every
    // candidate of the group (≤ MAP_KPT per thread, sums in registers) takes up the slice's share of its column sums.
    // Raw pixel units; the N0f8 scale 1/255 is applied once at the end.  Error (4l + 8)·2⁻⁵³, inside δ64's allowance.
    auto from_map = [&]() {
        const int n = cnt[0], H = L >> 1;
        const int xmin = mm[0], xmax = mm[1];
        const size_t rbytes = refine_r_bytes(g.n1, L, c.cbw), tbytes = refine_tile_bytes(c.tile_rows, L, c.cbw);
        // TPC lanes share a candidate's taps (a few dozen candidates would otherwise keep a few dozen lanes busy)
        int TPC = 64;
        while (TPC > 1 && n * TPC > MAP_KPT * NT) TPC >>= 1;
        const int cpp = NT / TPC, sub = tid & (TPC - 1);
        double sp[MAP_KPT], sm[MAP_KPT];
        int cx[MAP_KPT], cy[MAP_KPT];
#pragma unroll
        for (int j = 0; j < MAP_KPT; ++j) {
            const int k = tid / TPC + j * cpp;
            cx[j] = -1; cy[j] = 0; sp[j] = 0.0; sm[j] = 0.0;
            if (k < n) { const int lin = cand_lin[k]; cx[j] = lin / g.n1; cy[j] = lin - cx[j] * g.n1; }
        }
        int CW = 16;
        while (CW > 4 && ((size_t)refine_tile_pitch(CW, L) > tbytes || (size_t)CW * 16 > rbytes)) CW -= 4;
        for (int xg = xmin; xg <= xmax; xg += CW) {
            const int ncol = min(CW, xmax + 1 - xg);
            __syncthreads();
            if (tid == 0) { mm[2] = 0x7fffffff; mm[3] = -1; }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < MAP_KPT; ++j)
                if (cx[j] >= xg && cx[j] < xg + ncol) { atomicMin(&mm[2], cy[j]); atomicMax(&mm[3], cy[j]); }
            __syncthreads();
            const int ymin = mm[2], ymax = mm[3];
            if (ymax < 0) continue; // no candidate in this group of columns
            if (tid == 0) atomicAdd(g.ex.stat + 1, 1ull);
            const int rows_tot = ymax - ymin + L, tp = refine_tile_pitch((ncol + 3) & ~3, L); // (whole groups of four columns)
            int RSm = max(1, (int)min((size_t)rows_tot, min(tbytes / tp, rbytes / ((size_t)ncol * 16))));
            {   // whole rounds of NT (row, group of four columns) items
                const int ng = (ncol + 3) >> 2;
                if ((RSm * ng) / NT >= 1 && RSm < rows_tot) RSm = ((RSm * ng) / NT * NT) / ng;
            }
            for (int a0 = 0; a0 < rows_tot; a0 += RSm) {
                const int rows = min(RSm, rows_tot - a0);
                stage(xg, tp, ymin + a0, rows);
                PDOG_STAMP(1);
                // Row sums: a thread takes one tile row and FOUR adjacent columns.  Symmetric taps: column c's pair k is
                // F[c + k] + F[c + l−1−k] (F: the row's bytes from the group's first column), so a block of four taps needs
                // seven consecutive bytes at the front and seven at the back — one new aligned dword per side and block,
                // each byte converted to double once and shared by the columns (read byte by byte per column, the LDS
                // pipe was the limit: 2 reads per 2 FMAs).  Eight FMA chains per thread; the two physical windows of a
                // side swap roles every block, so nothing is moved.
                const int ngrp = (ncol + 3) >> 2, nitem = rows * ngrp, M4 = (L - 1) >> 2; // l = 4·M4 + 1
                auto cvt4 = [](uint32_t w, double (&d)[4]) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) d[i] = (double)((w >> (8 * i)) & 0xffu);
                };
                for (int e = tid; e < nitem; e += NT) {
                    const int a = e / ngrp, cg = e - a * ngrp;
                    const uint8_t *fb = tile + a * tp + 4 * cg;
                    const uint32_t *fw4 = reinterpret_cast<const uint32_t *>(fb);
                    double rp[4], rm[4];
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) { rp[cc] = 0.0; rm[cc] = 0.0; }
                    double fa[4], fbk[4], ha[4], hb[4]; // front: dwords q, q+1; back: dwords M4−q−1 (ha), M4−q (hb), q = k0/4
                    cvt4(fw4[0], fa);
                    cvt4(fw4[1], fbk);
                    cvt4(fw4[M4 - 1], ha);
                    cvt4(fw4[M4], hb);
                    // one block of four taps k0 … k0+3: lo = (LO0, LO1) = bytes k0 … k0+7, back window (HA, HB) = bytes l−1−k0−4 … l−1−k0+3
                    auto block = [&](const double (&LO0)[4], const double (&LO1)[4], const double (&HA)[4], const double (&HB)[4], const double *gp, const double *gm) {
#pragma unroll
                        for (int u = 0; u < 4; ++u)
#pragma unroll
                            for (int cc = 0; cc < 4; ++cc) {
                                const int li = cc + u, hi = 4 + cc - u; // front byte k0 + li; back byte (l−1−k0−4) + hi
                                const double v = (li < 4 ? LO0[li] : LO1[li - 4]) + (hi < 4 ? HA[hi] : HB[hi - 4]);
                                rp[cc] = __builtin_fma(gp[u], v, rp[cc]);
                                rm[cc] = __builtin_fma(gm[u], v, rm[cc]);
                            }
                    };
                    int k0 = 0;
                    for (; k0 + 8 <= H; k0 += 8) {
                        double gp[8], gm[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) { gp[u] = c.g64[k0 + u]; gm[u] = c.g64[L + k0 + u]; }
                        const int q = k0 >> 2;
                        const uint32_t nf0 = fw4[q + 2], nh0 = fw4[M4 - q - 2], nf1 = fw4[q + 3], nh1 = fw4[max(M4 - q - 3, 0)];
                        block(fa, fbk, ha, hb, gp, gm);
                        cvt4(nf0, fa);  // fa ← dword q+2, hb ← dword M4−q−2: the next block's (lo1, ha)
                        cvt4(nh0, hb);
                        block(fbk, fa, hb, ha, gp + 4, gm + 4);
                        cvt4(nf1, fbk); // back to the first block's roles, two dwords on
                        cvt4(nh1, ha);
                    }
                    // the remaining pairs (fewer than 8) one at a time, then the centre tap (k = H) on its own
                    for (int k = k0; k <= H; ++k) {
                        const double gpk = c.g64[k], gmk = c.g64[L + k];
#pragma unroll
                        for (int cc = 0; cc < 4; ++cc) {
                            const double v = (double)(k < H ? (int)fb[cc + k] + (int)fb[cc + L - 1 - k] : (int)fb[cc + H]);
                            rp[cc] = __builtin_fma(gpk, v, rp[cc]);
                            rm[cc] = __builtin_fma(gmk, v, rm[cc]);
                        }
                    }
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc)
                        if (4 * cg + cc < ncol) { R64[2 * (a * ncol + 4 * cg + cc)] = rp[cc]; R64[2 * (a * ncol + 4 * cg + cc) + 1] = rm[cc]; }
                }
                __syncthreads();
                PDOG_STAMP(4);
#pragma unroll
                for (int j = 0; j < MAP_KPT; ++j) {
                    if (cx[j] < xg || cx[j] >= xg + ncol) continue;
                    const int x = cx[j] - xg, y = cy[j] - ymin;
                    const int t0 = max(0, a0 - y), t1 = min(L, a0 + rows - y); // taps whose rows lie in this slice
                    const double *r = R64 + 2 * ((y - a0) * ncol + x);
                    double p = sp[j], m = sm[j];
#pragma unroll 4
                    for (int t = t0 + sub; t < t1; t += TPC) {
                        p = __builtin_fma(c.g64[t], r[2 * t * ncol], p);
                        m = __builtin_fma(c.g64[L + t], r[2 * t * ncol + 1], m);
                    }
                    sp[j] = p; sm[j] = m;
                }
                PDOG_STAMP(5);
            }
        }
#pragma unroll
        for (int j = 0; j < MAP_KPT; ++j) {
            double p = sp[j], m = sm[j];
            for (int off = TPC >> 1; off > 0; off >>= 1) { p += __shfl_xor(p, off, 64); m += __shfl_xor(m, off, 64); }
            if (cx[j] >= 0 && sub == 0) cand_val[tid / TPC + j * cpp] = c.dir * (p - m) / 255.0;
        }
        __syncthreads();
    };

    Peak64 pk;
    pk.best = -__builtin_huge_val();
    pk.idx = 0x7fffffff;
    if (use_map) from_map(); else pk = sweep(false);
    bool overflow = cnt[1] != 0;
    if (overflow && !have_v) { // more candidates than the list holds under the V = 255 threshold: the window's own V, then once more
        __syncthreads();
        if (tighten()) return c.fp32_idx;
        if (tid == 0) { cnt[0] = 0; cnt[1] = 0; }
        __syncthreads();
        pk = sweep(false);
        overflow = cnt[1] != 0;
    }
    if (overflow) {
        __syncthreads();
        pk = sweep(true);
    } else {
        // stage 2 verdict: best separable value; survivors within T64 of it
        const int n = cnt[0];
        double m2 = -__builtin_huge_val();
        for (int k = tid; k < n; k += NT) m2 = fmax(m2, cand_val[k]);
        for (int off = 32; off > 0; off >>= 1) m2 = fmax(m2, __shfl_xor(m2, off, 64));
        if (lane == 0) dred[wave] = m2;
        __syncthreads();
        for (int w = 0; w < NW; ++w) m2 = fmax(m2, dred[w]);
        __syncthreads();
        const double keep = m2 - c.T64;
        // compact the survivors to the front of the list (order is irrelevant: ties are settled by index)
        for (int k0 = 0; k0 < n; k0 += NT) {
            const int k = k0 + tid;
            const bool sv = k < n && cand_val[k] >= keep;
            const int lin = sv ? cand_lin[k] : 0;
            __syncthreads();
            if (sv) cand_lin[atomicAdd(&cnt[2], 1)] = lin; // lands below k0 + NT: every entry there has been read already
            __syncthreads();
        }
        const int ns = cnt[2];
        if (tid == 0) { atomicAdd(g.ex.stat + 2, (unsigned long long)n); atomicAdd(g.ex.stat + 3, (unsigned long long)(ns == 1 ? 0 : ns)); }
        if (ns == 1) {
            if (tid == 0) cnt[3] = cand_lin[0];
        } else {
            for (int k = tid; k < ns; k += NT) {
                const int lin = cand_lin[k];
                const int xw = lin / g.n1, y = lin - xw * g.n1;
                peak64_push(pk, exact_pixel(frame, g.row_stride, g.fh, g.fw, g.fill, ti0 + y, wj0 + xw, L, c.K, lut), lin);
            }
        }
        PDOG_STAMP(6);
        if (ns == 1) { __syncthreads(); return cnt[3]; }
    }
    peak64_wave_reduce(pk);
    __syncthreads();
    if (lane == 0) { dred[wave] = pk.best; ired[wave] = pk.idx; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < NW; ++w) peak64_push(pk, dred[w], ired[w]);
        cnt[3] = pk.idx;
    }
    __syncthreads();
    return cnt[3];
}

// ---- the last kernel of a batch: strip combine + index map + clamp (:58-61), and the refinement of exact mode ----
// Four windows per workgroup: each wave combines one window's partial peaks (a handful of loads), writes the FP32
// answer, checks the guess's range and — the usual case — that is all: the runner-up lies further than 2δ below the
// maximum.  Otherwise the whole workgroup refines the window (above) and writes the reference's answer.
struct FinishGeo {
    LaunchGeo g;                 // frames, strides, frame_index, guesses, geometry, part_val/idx/sec, nslots, ex
    const double *K64;           // l×l, column-major, dir·(g₊⊗g₊ − g₋⊗g₋) (:41-43); null = exact mode off
    const double *g64;           // [2][l] Float64 Gaussians
    double dir, T64;
    int cbw, tile_rows;
    // how the partial slots map to window columns: slot s < nmain covers [min(s·slot_w, slot_last) … + slot_w) (the
    // roll kernel shifts its last strip left: slot_last = covered − slot_w; others: slot_last = huge), slots ≥ nmain
    // are single columns thin_x0 + (s − nmain)
    int slot_w, slot_last, nmain, thin_x0;
    int use_mask;                // the main slots carry per-column masks (roll kernel, slot_w = 64)
    int v_after;                 // RefineCtx::v_after
    const float *map;            // null, or the batch's FP32 responses [n][n2][n1] (two-pass path): refine_window reads the candidates off it
    const int *vmax;             // null, or [n]: each window's own V = max |pixel − dc| (two-pass row pass): the flag and the refinement scale T with V/255
    int32_t *out_ij;             // [n][2]
    int32_t *done_flag;          // NULL or host-coherent ticket word (see dog_fused.hpp): published with window 0's final answer
    int32_t done_value;
    int seq_windows;             // windows of this tracker's batches BEFORE this one (cumulative, wraps): published beside the flag count they produced
};

constexpr int REFINE_NT = 256;

constexpr int FINISH_WPB = REFINE_NT / 64; // windows per workgroup: a wave combines one window's partials

#ifndef PDOG_ROLL_INST_ONLY // the roll instantiation units (roll_inst.hip) do not need a private copy of this kernel each
static __global__ __launch_bounds__(REFINE_NT) void dog_finish_kernel(const FinishGeo fg, const f2 *__restrict__ taps_row,
                                                                      const f2 *__restrict__ taps_col)
{
    constexpr int NT = REFINE_NT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ int s_refine[FINISH_WPB], s_idx2[FINISH_WPB];
    __shared__ float s_max[FINISH_WPB], s_sec2[FINISH_WPB];
    const LaunchGeo &g = fg.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    {   // every wave: one window — the slots' loads go out together (one memory round trip), then a shuffle merge
        const int b = blockIdx.x * FINISH_WPB + wave;
        bool rf = false;
        float best = 0.f, sec = 0.f;
        int bidx = -1;
        if (b < g.n) {
            Peak pk;
            peak_init(pk);
            int nload = g.nslots;
            if (g.fold_r) {
                // Folded remainder column (dog_roll.hpp): the last strip left the row-pass outputs of window column thin_x0;
                // its column pass runs here, one output row per lane and round, in the strips' operation order (per tap t
                // ascending, the g+ term then the g− term into one f32) ⇒ bit-identical to what a strip would have produced.
                const int NA = g.n1 + g.L - 1, NAP = NA + FOLD_GO; // slice pitch: the sliding windows of the last lanes read a few entries past NA
                f2 *Rc = reinterpret_cast<f2 *>(smem) + (size_t)wave * NAP; // wave-private
                const f2 *src = g.fold_r + (long long)b * NA;
                for (int a = lane; a < NAP; a += 64) Rc[a] = a < NA ? src[a] : f2{0.f, 0.f};
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                Peak tp = fold_column_peak(Rc, g.n1, g.L, as_taps(taps_col), g.thin_x0 * g.n1, lane);
                nload = fg.nmain; // the column's slot is filled here, not read
                if (lane == 0) {
                    pk = tp;
                    g.part_val[b * g.nslots + fg.nmain] = tp.best; // the refinement reads the slots' maxima from memory
                    g.part_idx[b * g.nslots + fg.nmain] = tp.idx;
                    g.part_sec[b * g.nslots + fg.nmain] = tp.second;
                }
            }
            for (int s = lane; s < nload; s += 64) peak_merge(pk, g.part_val[b * g.nslots + s], g.part_idx[b * g.nslots + s], g.part_sec[b * g.nslots + s]);
            peak_wave_reduce(pk);
            if (lane == 0) {
                const int g1 = g.guesses[2 * b], g2 = g.guesses[2 * b + 1];
                const float Tb = (fg.vmax && g.ex.T < __builtin_huge_valf()) ? g.ex.T * ((float)fg.vmax[b] * (1.0f / 255.0f)) * 1.00001f : g.ex.T;
                rf = fg.K64 && (pk.best - pk.second <= Tb);
                best = pk.best;
                sec = pk.second;
                bidx = pk.idx;
                range_check(g.ex, g1, g2, g.L >> 1, g.fh, g.fw);
                if (rf) {
                    atomicAdd(g.ex.stat, 1ull);
                } else {
                    const int x = pk.idx / g.n1, y = pk.idx - x * g.n1;
                    fg.out_ij[2 * b] = min(max(g1 - g.r1 + y, 1), g.fh);       // :60-61
                    fg.out_ij[2 * b + 1] = min(max(g2 - g.r2 + x, 1), g.fw);
                    if (fg.done_flag && b == 0) __hip_atomic_store(fg.done_flag, fg.done_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
        }
        if (lane == 0) { s_refine[wave] = rf; s_max[wave] = best; s_sec2[wave] = sec; s_idx2[wave] = bidx; }
    }
    __syncthreads();
    // windows flagged so far (up to the batches before this one), where the host can see it without a copy or a wait: it switches
    // batches of hard windows to the response-map path (pawsome_dog.hip).  One plain store per batch — a system-scope
    // atomic per workgroup with flagged windows cost cfg5, where every window is flagged, 0.7 ms of PCIe atomics.
    if (blockIdx.x == 0 && tid == 0 && g.ex.range_err) {
        __hip_atomic_store(g.ex.range_err + 1, (int)__hip_atomic_load(g.ex.stat, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(g.ex.range_err + 2, fg.seq_windows, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); // … out of how many windows
    }
    // the windows of this workgroup that need the refinement (rare), one after the other, all threads on each
    constexpr int SLOT_CAP = 128;
    __shared__ float s_pv[SLOT_CAP];
    __shared__ unsigned long long s_pm[SLOT_CAP];
    for (int w = 0; w < FINISH_WPB; ++w) {
        if (!s_refine[w]) continue;
        const int b = blockIdx.x * FINISH_WPB + w;
        const int g1 = g.guesses[2 * b], g2 = g.guesses[2 * b + 1];
        const int fidx = g.frame_index ? g.frame_index[b] : b;
        const uint8_t *__restrict__ frame = g.frames + (long long)fidx * g.frame_stride;
        RefineCtx c;
        c.trow = as_taps(taps_row);
        c.tcol = as_taps(taps_col);
        c.K = (k64_ptr)(unsigned long long)fg.K64;
        c.g64 = (k64_ptr)(unsigned long long)fg.g64;
        c.dir = fg.dir;
        c.T64 = fg.T64;
        c.T = g.ex.T;
        c.second = s_sec2[w];
        c.fp32_idx = s_idx2[w];
        c.v_after = fg.v_after;
        c.T_rescan = g.ex.T_rescan;
        c.vmax_known = fg.vmax ? fg.vmax[b] : -1;
        c.cbw = fg.cbw;
        c.tile_rows = fg.tile_rows;
        c.lds = smem;
        // the slots' maxima and column masks once, into LDS: `may` is asked once per column block
        const bool slots_ok = g.nslots <= SLOT_CAP;
        __syncthreads();
        if (slots_ok)
            for (int s = tid; s < g.nslots; s += NT) {
                s_pv[s] = g.part_val[(long long)b * g.nslots + s];
                s_pm[s] = fg.use_mask ? g.part_mask[(long long)b * g.nslots + s] : ~0ull;
            }
        __syncthreads();
        auto may = [&](int x0, int x1, float thr) {
            if (!slots_ok) return true;
            // main slots: slot s covers [min(s·slot_w, slot_last), + slot_w); only those that can intersect [x0, x1) are looked at
            const int s_lo = max(0, x0 / fg.slot_w - 1), s_hi = min(fg.nmain, x1 / fg.slot_w + 2);
            for (int s = s_lo; s < s_hi; ++s) {
                const int lo = min(s * fg.slot_w, fg.slot_last), hi = lo + fg.slot_w;
                if (lo < x1 && hi > x0 && s_pv[s] >= thr && ((s_pm[s] & column_bits(x0 - lo, x1 - lo)) || fg.slot_w != 64)) return true;
            }
            if (fg.slot_last < (1 << 29) && fg.nmain > 0) { // the roll kernel's last strip, shifted left: overlaps its predecessors
                const int s = fg.nmain - 1, lo = fg.slot_last, hi = lo + fg.slot_w;
                if (lo < x1 && hi > x0 && s_pv[s] >= thr && (s_pm[s] & column_bits(x0 - lo, x1 - lo))) return true;
            }
            for (int s = fg.nmain; s < g.nslots; ++s) { // thin columns
                const int lo = fg.thin_x0 + (s - fg.nmain);
                if (lo < x1 && lo >= x0 && s_pv[s] >= thr) return true;
            }
            return false;
        };
        const int idx = refine_window<16>(NT, g, frame, g1, g2, s_max[w], c, may, fg.map ? fg.map + (long long)b * g.n1 * g.n2 : nullptr);
        if (tid == 0) {
            const int x = idx / g.n1, y = idx - x * g.n1;
            fg.out_ij[2 * b] = min(max(g1 - g.r1 + y, 1), g.fh);
            fg.out_ij[2 * b + 1] = min(max(g2 - g.r2 + x, 1), g.fw);
            if (fg.done_flag && b == 0) {
                __threadfence_system();
                __hip_atomic_store(fg.done_flag, fg.done_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        __syncthreads();
    }
}

#endif // PDOG_ROLL_INST_ONLY

} // namespace pdog
