// dog_exact.hpp — positions that are the reference's BY CONSTRUCTION, not by observation.
//
// The reference ranks Float64 dense sums: buff[I] = Σ_J Float64(img[I+J])·K[J], accumulated sequentially in kernel
// column-major order, then findmax (/root/reference/src/PawsomeTracker.jl:57-59).  The kernels of this library
// rank FP32 separable sums.  Exact ties are settled identically by construction; NEAR ties are not: when the two
// best responses of a window differ by less than the FP32 evaluation error, FP32 may crown the wrong pixel.
//
// Guarantee.  Let F(p) be the reference's Float64 value of pixel p and f(p) the FP32 value of any kernel here.
// |f(p) − F(p)| ≤ δ for every p, with the a-priori bound (u = 2⁻²⁴, V = max |pixel − dc| ≤ 255, l taps):
//     row pass     R̂± = Σ_k ĝ±[k]·v, one FMA chain of ≤ l terms, taps rounded once:  |R̂± − R±| ≤ (l + 1)·u·V
//     column pass  one chain of 2l FMAs over terms bounded by Σ|ĉ±||R̂±| ≤ 2V/255:     ≤ 2l·u·2V/255
//                  + the row errors times Σ|c±| = 1/255 each + the column taps' rounding 2u·V/255
//     ⇒ δ = u·(V/255)·(6l + 4)·(1 + ε)  + the reference's own Float64 rounding (≤ l²·2⁻⁵³·2·V/255, negligible)
// (any summation order, with or without the symmetric pre-add — so it covers every kernel family).  V = 255 is
// used: δ(l = 65) = 2.35e-5 against a typical peak of 0.09 and a typical peak-to-neighbour gap of 4.7e-4.
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

// ---- refinement of window columns [x0, x0 + ncol) by one workgroup of NT threads (a multiple of 64) ----
// A deliberately plain separable FP32 evaluation (any evaluation within δ serves: see the header), then the
// reference's arithmetic for the candidates.  Rlds: NA·ncol f2; tile: NULL, or NA rows of refine_tile_pitch(ncol, L)
// bytes — the block's pixels with the PaddedView fill materialised, which both passes then read instead of the
// frame (a candidate's 4225-term chain must not wait for memory 4225 times); lut: 256 doubles; ired/dred: NT/64
// entries each.  Returns the block's Float64 peak in thread 0 (best = −inf if the block holds no candidate).
__host__ __device__ constexpr int refine_tile_pitch(int ncol, int L) { return (ncol + L - 1 + 3) / 4 * 4; }
__device__ __forceinline__ Peak64 refine_columns(const int NT, const LaunchGeo &g, const uint8_t *__restrict__ frame, int g1, int g2, int x0, int ncol,
                                                 float thr, tap_ptr trow, tap_ptr tcol, k64_ptr K, f2 *Rlds, uint8_t *tile, double *lut, int *ired,
                                                 double *dred, bool fill_lut = true)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, NW = NT / 64;
    const int L = g.L, hw = L >> 1, NA = g.n1 + L - 1;
    const int ti0 = g1 - g.r1 - 1 - hw, wj0 = g2 - g.r2 - 1 - hw;
    const int tp = refine_tile_pitch(ncol, L), tw = ncol + L - 1;
    if (fill_lut)
        for (int p = tid; p < 256; p += NT) lut[p] = (double)p / 255.0;
    if (tile) {
        for (int e = tid; e < NA * tw; e += NT) {
            const int a = e / tw, c = e - a * tw;
            const int gi = ti0 + a, gj = wj0 + x0 + c;
            int px = g.fill;
            if (gi >= 0 && gi < g.fh && gj >= 0 && gj < g.fw) px = frame[(long long)gi * g.row_stride + gj];
            tile[a * tp + c] = (uint8_t)px;
        }
    }
    // DC level: the same fixed sample grid as the main kernels (any level in 0…255 keeps the bound)
    int sum = dc_sample_sum(g, frame, ti0, wj0, L, tid, NT);
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
    if (lane == 0) ired[wave] = sum;
    __syncthreads();
    int tot = 0;
    for (int w = 0; w < NW; ++w) tot += ired[w];
    __syncthreads();
    const int dc = dc_from_sum(tot, g.fill);
    // row pass: R±[a][x] = Σ_k ĝ±[k]·(pixel[a][x0+x+k] − dc), k ascending
    for (int e = tid; e < NA * ncol; e += NT) {
        const int a = e / ncol, x = e - a * ncol;
        f2 acc = f2{0.f, 0.f};
        if (tile) {
            const uint8_t *src = tile + a * tp + x;
            for (int k = 0; k < L; ++k) acc = fma_bcast((float)((int)src[k] - dc), trow[k], acc);
        } else {
            const int gi = ti0 + a, gj0 = wj0 + x0 + x;
            const bool rowok = gi >= 0 && gi < g.fh;
            const uint8_t *src = frame + (long long)gi * g.row_stride;
            for (int k = 0; k < L; ++k) {
                const int gj = gj0 + k;
                int px = g.fill;
                if (rowok && gj >= 0 && gj < g.fw) px = src[gj];
                acc = fma_bcast((float)(px - dc), trow[k], acc);
            }
        }
        Rlds[e] = acc;
    }
    __syncthreads();
    // column pass + candidates
    Peak64 pk;
    pk.best = -__builtin_huge_val();
    pk.idx = 0x7fffffff;
    for (int e = tid; e < g.n1 * ncol; e += NT) {
        const int x = e / g.n1, y = e - x * g.n1;
        float acc = 0.f;
        for (int t = 0; t < L; ++t) {
            const f2 r = Rlds[(y + t) * ncol + x];
            const f2 w = tcol[t];
            acc = __builtin_fmaf(r.x, w.x, acc);
            acc = __builtin_fmaf(r.y, w.y, acc);
        }
        if (acc >= thr) {
            const double F = tile ? exact_patch((const uint8_t *)(tile + y * tp + x), (long long)tp, L, K, lut)
                                  : exact_pixel(frame, g.row_stride, g.fh, g.fw, g.fill, ti0 + y, wj0 + x0 + x, L, K, lut);
            peak64_push(pk, F, (x0 + x) * g.n1 + y);
        }
    }
    peak64_wave_reduce(pk);
    __syncthreads();
    if (lane == 0) { dred[wave] = pk.best; ired[wave] = pk.idx; }
    __syncthreads();
    if (tid == 0)
        for (int w = 1; w < NW; ++w) peak64_push(pk, dred[w], ired[w]);
    __syncthreads();
    return pk;
}

// ---- the last kernel of a batch: strip combine + index map + clamp (:58-61), and the refinement of exact mode ----
// S workgroups per window.  Every one of them combines the window's partial peaks (a handful of loads); part 0
// writes the FP32 answer, checks the guess's range and — the usual case — that is all: the runner-up lies further
// than 2δ below the maximum.  Otherwise the S workgroups share the window's column blocks, re-evaluate the
// near-maximal pixels in the reference's arithmetic, and the last one to finish writes the reference's answer.
struct FinishGeo {
    LaunchGeo g;                 // frames, strides, frame_index, guesses, geometry, part_val/idx/sec, nslots, ex
    const double *K64;           // l×l, column-major, dir·(g₊⊗g₊ − g₋⊗g₋) (:41-43); null = exact mode off
    int cbw, nblk;               // window columns per column block; column blocks per window
    int use_tile;                // the block's pixels are staged in LDS behind the row-pass block
    int S;                       // workgroups per window
    double *part_val;            // [n][S] Float64 partial peaks
    int *part_idx;               // [n][S]
    int *part_done;              // [n] zero between launches
    int32_t *out_ij;             // [n][2]
    int32_t *done_flag;          // NULL or host-coherent ticket word (see dog_fused.hpp): published with window 0's final answer
    int32_t done_value;
};

constexpr int REFINE_NT = 256;
__host__ __device__ constexpr size_t refine_r_bytes(int n1, int L, int cbw) { return ((size_t)(n1 + L - 1) * cbw * sizeof(f2) + 15) / 16 * 16; }
__host__ __device__ constexpr size_t refine_tile_bytes(int n1, int L, int cbw) { return (size_t)(n1 + L - 1) * refine_tile_pitch(cbw, L); }
__host__ __device__ constexpr size_t refine_lds_bytes(int n1, int L, int cbw, bool tile = false)
{
    return refine_r_bytes(n1, L, cbw) + (tile ? refine_tile_bytes(n1, L, cbw) : 0);
}

static __global__ __launch_bounds__(REFINE_NT) void dog_finish_kernel(const FinishGeo fg, const f2 *__restrict__ taps_row,
                                                                      const f2 *__restrict__ taps_col)
{
    constexpr int NT = REFINE_NT, NW = NT / 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ double lut[256];
    __shared__ double dred[NW];
    __shared__ int ired[NW];
    __shared__ int s_refine, s_last;
    __shared__ float s_max;
    const LaunchGeo &g = fg.g;
    const int tid = threadIdx.x;
    const int b = blockIdx.x / fg.S, part = blockIdx.x - b * fg.S;
    const int g1 = g.guesses[2 * b], g2 = g.guesses[2 * b + 1];
    if (tid == 0) {
        Peak pk;
        peak_init(pk);
        for (int s = 0; s < g.nslots; ++s) peak_merge(pk, g.part_val[b * g.nslots + s], g.part_idx[b * g.nslots + s], g.part_sec[b * g.nslots + s]);
        const bool rf = fg.K64 && (pk.best - pk.second <= g.ex.T);
        if (part == 0) {
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
        s_refine = rf;
        s_max = pk.best;
    }
    __syncthreads();
    if (!s_refine) return;
    const int fidx = g.frame_index ? g.frame_index[b] : b;
    const uint8_t *__restrict__ frame = g.frames + (long long)fidx * g.frame_stride;
    const float thr = s_max - g.ex.T;
    Peak64 mine;
    mine.best = -__builtin_huge_val();
    mine.idx = 0x7fffffff;
    for (int cb = part; cb < fg.nblk; cb += fg.S) {
        const int x0 = cb * fg.cbw, ncol = min(fg.cbw, g.n2 - x0);
        const Peak64 pk = refine_columns(NT, g, frame, g1, g2, x0, ncol, thr, as_taps(taps_row), as_taps(taps_col), (k64_ptr)(unsigned long long)fg.K64,
                                         reinterpret_cast<f2 *>(smem), fg.use_tile ? smem + refine_r_bytes(g.n1, g.L, fg.cbw) : nullptr, lut, ired, dred);
        if (tid == 0) peak64_push(mine, pk.best, pk.idx);
    }
    if (tid == 0) {
        __hip_atomic_store(&fg.part_val[(long long)b * fg.S + part], mine.best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&fg.part_idx[(long long)b * fg.S + part], mine.idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int old = __hip_atomic_fetch_add(&fg.part_done[b], 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (old == fg.S - 1);
        if (s_last) { // this window's last workgroup: first Float64 maximum over all of them → position (:59-61)
            Peak64 w;
            w.best = -__builtin_huge_val();
            w.idx = 0x7fffffff;
            for (int k = 0; k < fg.S; ++k)
                peak64_push(w, __hip_atomic_load(&fg.part_val[(long long)b * fg.S + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                            __hip_atomic_load(&fg.part_idx[(long long)b * fg.S + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            const int x = w.idx / g.n1, y = w.idx - x * g.n1;
            fg.out_ij[2 * b] = min(max(g1 - g.r1 + y, 1), g.fh);
            fg.out_ij[2 * b + 1] = min(max(g2 - g.r2 + x, 1), g.fw);
            __hip_atomic_store(&fg.part_done[b], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (fg.done_flag && b == 0) {
                __threadfence_system();
                __hip_atomic_store(fg.done_flag, fg.done_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

} // namespace pdog
