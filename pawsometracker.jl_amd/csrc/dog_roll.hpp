// dog_roll.hpp — barrier-free "rolling accumulator" DoG + argmax kernel for gfx950.
//
// Same arithmetic contract as dog_kernels.hpp (reference functor
// /root/reference/src/PawsomeTracker.jl:55-62), different execution shape:
//
//   * one WAVE (64 lanes, one workgroup) owns a strip of 64 window columns for the
//     whole window height; nothing is shared between waves, so there is no
//     s_barrier anywhere and the only LDS is a wave-private 8 KB staging area;
//   * input rows stream through in sub-chunks of CH = 8 rows:
//       stage   8 rows × 128 u8 (16 B per lane, prefetched one sub-chunk ahead)
//               → (pixel − dc) as f32 in LDS
//       row pass  lane = (row r, group of P = 8 consecutive outputs), sliding register
//               window, symmetric taps: Σ_k g[k]·(a[x−k] + a[x+k]) — one v_add + one
//               v_pk_fma_f32 (both Gaussians) per tap pair instead of two FMAs
//       column pass  lane = column; the l partial sums a column has in flight live in
//               REGISTERS (S = l−1+CH f32 accumulator slots, output y in slot y mod S), so each
//               R row is read from LDS exactly once (8 ds_read_b64 per sub-chunk per lane)
//               instead of (Q+l−1)/Q times from a 64+ row LDS ring.  The slot ↔ tap
//               mapping rotates by CH per sub-chunk; S/CH statically unrolled bodies are
//               selected by a wave-uniform switch, so every register index is a constant.
//   * completed outputs feed a running (max, first column-major index) per lane;
//     wave-shuffle reduction at the end, one partial per strip.
//
// On gfx950 v_pk_fma_f32 issues at the same FLOP rate as v_fma_f32 (4 vs 2 cycles per
// wave64, tools/ubench_valu.hip), so what counts is the number of lane-operations, and one
// wave per SIMD already reaches ≈84 % of the packed-FMA rate: occupancy 2 waves/SIMD
// (≤ 256 VGPRs) is enough once there are no barriers.
#pragma once
#include "dog_kernels.hpp"
#include <type_traits>

namespace pdog {

constexpr int ROLL_CH = 8;   // rows per sub-chunk
constexpr int ROLL_P = 8;    // row-pass outputs per lane
constexpr int ROLL_TW = 64;  // strip width = lanes
constexpr int ROLL_PA = 129; // A pitch (f32): odd → conflict-free row-pass reads
constexpr int ROLL_PR = 65;  // R pitch (f2)

__host__ __device__ constexpr int roll_slots(int L) { return L - 1 + ROLL_CH; }
__host__ __device__ constexpr size_t roll_lds_bytes() { return (size_t)ROLL_CH * ROLL_PA * 4 + (size_t)ROLL_CH * ROLL_PR * 8; }

// Row pass for one lane: P outputs, symmetric taps.  a = &A[r][P*gx] (72 inputs for l = 65).
// out[o] = Σ_{k<H} T[k]·(a[o+k] + a[o+L-1-k]) + T[H]·a[o+H],  H = L/2, taps ascending.
template <int L>
__device__ __forceinline__ void roll_row_pass(f2 (&acc)[ROLL_P], const float *a, tap_ptr taps)
{
    constexpr int P = ROLL_P, H = L / 2, U = 8;
    static_assert(H % U == 0, "half length must be a multiple of the tap block");
    float wlo[P + U - 1], whi[P + U - 1];
#pragma unroll
    for (int j = 0; j < P + U - 1; ++j) {
        wlo[j] = a[j];
        whi[j] = a[L - 1 - (U - 1) + j]; // a[L-U+j]
    }
    f2 tn[U];
#pragma unroll
    for (int j = 0; j < U; ++j) tn[j] = taps[j];
#pragma unroll
    for (int k0 = 0; k0 < H; k0 += U) {
        f2 t[U];
#pragma unroll
        for (int j = 0; j < U; ++j) t[j] = tn[j];
        float nlo[U], nhi[U];
        if (k0 + U < H) {
#pragma unroll
            for (int j = 0; j < U; ++j) {
                tn[j] = taps[k0 + U + j];
                nlo[j] = a[k0 + U + (P - 1) + j];           // a[(k0+U) + 7 + j]
                nhi[j] = a[L - U - (k0 + U) + j];           // lower end of the next hi window
            }
        } else {
            tn[0] = taps[H]; // centre tap
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int o = 0; o < P; ++o) {
                // lo = a[o + k0+u], hi = a[o + L-1-(k0+u)]
                const float s = wlo[o + u] + whi[o + (U - 1) - u];
                acc[o] = fma_bcast(s, t[u], acc[o]);
            }
        }
        if (k0 + U < H) {
#pragma unroll
            for (int j = 0; j < P - 1; ++j) wlo[j] = wlo[j + U];
#pragma unroll
            for (int j = 0; j < U; ++j) wlo[P - 1 + j] = nlo[j];
#pragma unroll
            for (int j = P + U - 2; j >= U; --j) whi[j] = whi[j - U];
#pragma unroll
            for (int j = 0; j < U; ++j) whi[j] = nhi[j];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    // centre tap: a[o + H]; after the last block wlo = a[H-U .. H-U+P+U-2], so a[o+H] = wlo[o+U] for o+U <= P+U-2
    // and the last one (o = P-1) is whi[0+...]: whi = a[L-U-(H-U) + j] = a[H+1+j] → a[H+P-1] = whi[P-2]
#pragma unroll
    for (int o = 0; o < P; ++o) {
        const float c = (o + U <= P + U - 2) ? wlo[o + U] : whi[P - 2];
        acc[o] = fma_bcast(c, tn[0], acc[o]);
    }
}

// Column pass body for sub-chunk phase SC (rows a ≡ CH*SC + i mod S).  rv[i] = (R+, R−)[row i][x].
template <int L, int SC>
__device__ __forceinline__ void roll_col_body(float (&acc)[roll_slots(L)], const f2 (&rv)[ROLL_CH], tap_ptr taps)
{
    constexpr int S = roll_slots(L), CH = ROLL_CH, U = 8;
    constexpr int NB = (L + U - 1) / U;
    f2 tn[U];
#pragma unroll
    for (int j = 0; j < U; ++j) tn[j] = taps[j];
#pragma unroll
    for (int tb = 0; tb < NB; ++tb) {
        f2 t[U];
#pragma unroll
        for (int j = 0; j < U; ++j) t[j] = tn[j];
#pragma unroll
        for (int j = 0; j < U; ++j)
            if ((tb + 1) * U + j < L) tn[j] = taps[(tb + 1) * U + j];
        // One f32 accumulator per output: acc += (s·g+)·R+ then acc += (−s·g−)·R−.  Two
        // v_fma_f32 cost the same VALU cycles as one v_pk_fma_f32 on gfx950 and halve the
        // accumulator registers; the g+ terms of all slots are issued before the g− terms so
        // that no FMA waits on the one before it.
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
#pragma unroll
            for (int i = 0; i < CH; ++i) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int tap = tb * U + u;
                    if (tap < L) {
                        const int amod = CH * SC + i;
                        const int slot = ((amod - tap) % S + S) % S;
                        const float r = ch ? rv[i].y : rv[i].x;
                        const float w = ch ? t[u].y : t[u].x;
                        if (tap == 0 && ch == 0)
                            acc[slot] = r * w;             // first term of a new output: no stale accumulator
                        else
                            acc[slot] = __builtin_fmaf(r, w, acc[slot]);
                    }
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ABL: timing-only ablation bits (tools/tune): 1 = no column FMAs, 2 = no row pass, 4 = no staging
// conversion/LDS writes, 8 = no global loads.  ABL != 0 gives wrong results by design.
template <int LT, bool RESP, int ABL = 0>
__global__ __launch_bounds__(64, 2) void dog_roll_kernel(const LaunchGeo g, const f2 *__restrict__ taps_row,
                                                         const f2 *__restrict__ taps_col)
{
    constexpr int L = LT, hw = L / 2, S = roll_slots(L), CH = ROLL_CH, P = ROLL_P, TW = ROLL_TW;
    constexpr int NBODY = S / CH;
    static_assert(S % CH == 0, "slot count must be a multiple of the sub-chunk");
    constexpr int TWin = TW + L - 1; // 128 input columns
    static_assert(TWin == 128, "staging assumes 16 B per lane over 8 rows");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *A = reinterpret_cast<float *>(smem);
    f2 *Rb = reinterpret_cast<f2 *>(smem + CH * ROLL_PA * 4);

    const int per_xcd = (g.nblocks + 7) >> 3;
    const int logical = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (logical >= g.nblocks) return;
    const int b = logical / g.nstrips;
    const int s = logical - b * g.nstrips;
    const int lane = threadIdx.x;

    const int g1 = g.guesses[2 * b], g2 = g.guesses[2 * b + 1];
    const int fidx = g.frame_index ? g.frame_index[b] : b;
    const uint8_t *__restrict__ frame = g.frames + (long long)fidx * g.frame_stride;
    // strips are 64 wide; the last one is shifted left to stay inside the window (overlap
    // recomputes a few columns bit-identically), or is partial when the window is < 64 wide
    const int x0 = (g.n2 >= TW) ? min(s * TW, g.n2 - TW) : 0;
    const int ws = min(TW, g.n2);
    const int ti0 = g1 - g.r1 - 1 - hw;
    const int wj0 = g2 - g.r2 - 1 - hw; // frame col of the window tile's col 0
    const int tj0 = wj0 + x0;
    const int NA = g.n1 + L - 1;

    // ---- per-window DC level (see dog_kernels.hpp): same samples in every strip ----
    int dc;
    {
        const int tH = g.n1 + L - 1, tW = g.n2 + L - 1;
        int sum = 0;
#pragma unroll 4
        for (int k = lane; k < 1024; k += 64) {
            const int gi = ti0 + (int)(((long long)(k >> 5) * tH) >> 5);
            const int gj = wj0 + (int)(((long long)(k & 31) * tW) >> 5);
            int v = g.fill;
            if (gi >= 0 && gi < g.fh && gj >= 0 && gj < g.fw) v = frame[(long long)gi * g.row_stride + gj];
            sum += v;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
        dc = (sum + 512) >> 10;
        if (abs(dc - g.fill) <= 8) dc = g.fill;
    }

    // ---- staging geometry: lane → (row lane>>3, 16-byte segment lane&7) ----
    const int srow = lane >> 3, sseg = lane & 7;
    const int scol = tj0 + 16 * sseg;            // frame col of this lane's first byte
    const bool cols_in = (scol >= 0) && (scol + 16 <= g.fw);
    auto load16 = [&](int a_row, uint32_t (&w)[4]) {
        const int gi = ti0 + a_row;
        const bool rowok = (a_row < NA) && (gi >= 0) && (gi < g.fh);
        const uint32_t fill4 = (uint32_t)g.fill * 0x01010101u;
        w[0] = w[1] = w[2] = w[3] = fill4;
        if (rowok) {
            const uint8_t *src = frame + (long long)gi * g.row_stride + scol;
            if (cols_in) {
                __builtin_memcpy(w, src, 16);
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int gj = scol + i;
                    if (gj >= 0 && gj < g.fw) {
                        const uint32_t v = src[i];
                        w[i >> 2] = (w[i >> 2] & ~(0xffu << (8 * (i & 3)))) | (v << (8 * (i & 3)));
                    }
                }
            }
        }
    };

    float acc[S];
#pragma unroll
    for (int j = 0; j < S; ++j) acc[j] = 0.f;
    float best = -__builtin_huge_valf();
    int best_idx = 0x7fffffff;

    const tap_ptr trow = as_taps(taps_row);
    const tap_ptr tcol = as_taps(taps_col);

    uint32_t pre[4];
    load16(srow, pre);
    const int nsub = (NA + CH - 1) / CH;
    const int rr = lane & 7, rgx = lane >> 3; // row-pass task: row rr, output group rgx
    const long long resp_base = (long long)b * g.n1 * g.n2;

    for (int sc = 0; sc < nsub; ++sc) {
        // ---- stage this sub-chunk from the prefetched registers, request the next ----
        if (!(ABL & 4)) {
            float *dst = A + srow * ROLL_PA + 16 * sseg;
            const float fdc = (float)dc;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const uint32_t word = pre[i >> 2];
                const float v = (float)((word >> (8 * (i & 3))) & 0xffu); // v_cvt_f32_ubyteN
                dst[i] = v - fdc;
            }
            if (!(ABL & 8)) load16((sc + 1) * CH + srow, pre);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_barrier(); // single-wave workgroup: orders the LDS writes before the reads below
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        // ---- row pass: 8 rows × 8 groups of 8 outputs ----
        if (!(ABL & 2)) {
            f2 racc[P];
#pragma unroll
            for (int o = 0; o < P; ++o) racc[o] = f2{0.f, 0.f};
            roll_row_pass<L>(racc, A + rr * ROLL_PA + rgx * P, trow);
            f2 *dst = Rb + rr * ROLL_PR + rgx * P;
#pragma unroll
            for (int o = 0; o < P; ++o) dst[o] = racc[o];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        // ---- column pass: 8 new R rows into the rolling accumulators ----
        f2 rv[CH];
#pragma unroll
        for (int i = 0; i < CH; ++i) rv[i] = Rb[i * ROLL_PR + lane];
        const int phase = sc % NBODY;
        auto emit = [&](auto SCc) {
            constexpr int SC = decltype(SCc)::value;
            if (!(ABL & 1)) roll_col_body<L, SC>(acc, rv, tcol);
            // outputs y = a − (l−1) for the 8 rows of this sub-chunk are complete
            const int ybase = sc * CH - (L - 1);
            if (ybase + CH > 0 && ybase < g.n1) {
                float v[CH];
                float m = -__builtin_huge_valf();
                const bool colok = lane < ws;
                const int lin0 = (x0 + lane) * g.n1 + ybase;
#pragma unroll
                for (int i = 0; i < CH; ++i) {
                    constexpr int dummy = 0; (void)dummy;
                    const int slot = ((CH * SC + i - (L - 1)) % S + S) % S;
                    const bool ok = colok && (ybase + i >= 0) && (ybase + i < g.n1);
                    v[i] = acc[slot];
                    if (RESP && ok) g.resp[resp_base + lin0 + i] = v[i];
                    v[i] = ok ? v[i] : -__builtin_huge_valf();
                    m = fmaxf(m, v[i]);
                }
                if (m >= best && m > -__builtin_huge_valf()) {
#pragma unroll
                    for (int i = 0; i < CH; ++i)
                        if (v[i] > best || (v[i] == best && lin0 + i < best_idx)) { best = v[i]; best_idx = lin0 + i; }
                }
            }
        };
        switch (phase) {
        case 0: emit(std::integral_constant<int, 0>{}); break;
        case 1: emit(std::integral_constant<int, 1 % NBODY>{}); break;
        case 2: emit(std::integral_constant<int, 2 % NBODY>{}); break;
        case 3: emit(std::integral_constant<int, 3 % NBODY>{}); break;
        case 4: emit(std::integral_constant<int, 4 % NBODY>{}); break;
        case 5: emit(std::integral_constant<int, 5 % NBODY>{}); break;
        case 6: emit(std::integral_constant<int, 6 % NBODY>{}); break;
        case 7: emit(std::integral_constant<int, 7 % NBODY>{}); break;
        default: emit(std::integral_constant<int, 8 % NBODY>{}); break;
        }
        __builtin_amdgcn_s_barrier(); // A / Rb are rewritten by the next sub-chunk
    }

#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_down(best, off, 64);
        const int oi = __shfl_down(best_idx, off, 64);
        if (ov > best || (ov == best && oi < best_idx)) { best = ov; best_idx = oi; }
    }
    if (lane == 0) {
        g.part_val[logical] = best;
        g.part_idx[logical] = best_idx;
    }
}

} // namespace pdog
