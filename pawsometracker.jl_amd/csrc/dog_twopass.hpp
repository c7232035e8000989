// dog_twopass.hpp — DoG + argmax for LONG kernels (l ≳ 100, e.g. target_width 120 → l = 293).
//
// Same arithmetic contract as the other kernels (reference functor
// /root/reference/src/PawsomeTracker.jl:55-62).  With l−1 ≈ 300 rows of halo neither the LDS ring
// of dog_kernels.hpp (ring = (l−1+CH)·TW·8 B leaves room for ≤ 48 columns per CU) nor the register
// accumulators of dog_roll.hpp (l slots per lane) fit, so the two separable passes become two
// launches with the intermediate in HBM — affordable exactly because l is large: ≈590 FMA per
// intermediate element against 16 B of traffic.
//
//   dog_dc_kernel     per-window DC level (see dog_kernels.hpp), once per batch
//   dog_h1_kernel     ROW pass: 16 input rows → LDS once as f32 (pixel − dc); lane = (row, group of P
//                     outputs), sliding register windows, symmetric taps (v_add_f32 + one
//                     v_pk_fma_f32 for both Gaussians).  R is written TRANSPOSED, RT[x][a] (f2), so
//                     that the column pass becomes another pass along contiguous memory.  (A lane-
//                     per-column vertical pass over the strided u8 tile was tried first: 69 % of
//                     its wave cycles were memory waits on byte loads.)
//   dog_hpass_kernel  COLUMN pass + peak on RT: 16 RT rows (= 16 window columns x) → LDS once,
//                     lane = (x, group of P outputs y), sliding window along a,
//                     D = Σ (s·g₊)·R₊ + (−s·g₋)·R₋, running first maximum, workgroup reduction →
//                     one partial per 16-column block.
// Every output sees its taps in the same order (row pass: k ascending symmetric pairs then centre;
// column pass: k ascending), so equal inputs give bit-equal outputs and flat windows tie exactly.
#pragma once
#include "dog_roll.hpp"

namespace pdog {

struct TwoPassGeo {
    LaunchGeo g;
    f2 *__restrict__ RT;    // [windows in this chunk][n2][NA]   (transposed row-pass result)
    int *__restrict__ dc;   // [n]
    int TWin;               // n2 + L − 1 tile columns
    int NA;                 // n1 + L − 1 tile rows
    int win0;               // first window of this chunk
    int h1blocks_per_win;   // row-pass workgroups per window = ceil(NA / 16)
    int hblocks_per_win;    // column-pass workgroups per window = ceil(n2 / 16)
    int pitchA;             // LDS row pitch of the row pass, in floats
    int pitchV;             // LDS row pitch of the column pass, in f2
    // low-latency variants (small batches): no separate DC and strip-combine launches
    int *__restrict__ counter;   // [n] zero between launches: column-pass workgroups that have delivered their partial
    int32_t *__restrict__ out_ij; // [n][2] final positions, written by the last column-pass workgroup of a window
    int32_t *done_flag;          // NULL, or a word in host-coherent memory that receives done_value (system-scope release)
    int32_t done_value;          // right after window 0's answer: the host functor polls it (see dog_fused.hpp)
    int *__restrict__ vmax;      // NULL, or [n]: the window's own V = max |pixel − dc| over its padded tile, collected by the row pass
                                 // (exact mode's error bound is proportional to it: the finishing kernel flags with the window's V, not 255)
};

// BLOCKED ACCUMULATION (both passes).  Each pass adds its l (or l÷2 + 1 paired) terms in chains of one TRIP of the register ring
// (NB·U taps: 16–24) and adds the chains' sums up, instead of one chain of up to 293 FMAs: the rounding-error bound of a chain of
// n FMAs is ≈n·u·Σ|terms|, that of B chains of m plus B additions ≈(m + B)·u·Σ|terms| — 7× tighter at l = 293.  With the one-chain
// bound δ(l = 293) = 1.05e-4 exceeded the distance between NEIGHBOURING responses at σ = 51 (≈2e-5): exact mode flagged and
// re-decided every cfg5 window.  Cost: one v_pk_add_f32 + one v_mov_b64 per output and trip (≈+4 % instructions).
// Every output still sees the same operations in the same order: equal inputs give bit-equal outputs, flat windows tie exactly.
// FLUSH instances serve kernel lengths from TWOPASS_FLUSH_L on (the plain instances keep 70–96 VGPRs instead of 118–150 and serve the
// short kernels' small batches, where the plain chains' bound already flags next to nothing).
constexpr int TWOPASS_FLUSH_L = 101;
__host__ __device__ constexpr int twopass_ring(int P, int U) { return ((P + 2 * U - 1 + U - 1) / U) * U; } // register-ring slots = taps per trip

static __global__ __launch_bounds__(64) void dog_dc_kernel(const LaunchGeo g, int *__restrict__ dc, int *__restrict__ vmax)
{
    const int b = blockIdx.x, lane = threadIdx.x, hw = g.L >> 1;
    const int g1 = g.guesses[2 * b], g2 = g.guesses[2 * b + 1];
    const int fidx = g.frame_index ? g.frame_index[b] : b;
    const uint8_t *__restrict__ frame = g.frames + (long long)fidx * g.frame_stride;
    int sum = dc_sample_sum(g, frame, g1 - g.r1 - 1 - hw, g2 - g.r2 - 1 - hw, g.L, lane, 64);
    sum = wave_sum(sum);
    if (lane == 0) {
        dc[b] = dc_from_sum(sum, g.fill);
        if (vmax) vmax[b] = 0; // collected by the row pass that follows
    }
}

// ---- row pass (u8 rows → RT) ----
constexpr int HP_ROWS = 16; // rows per workgroup in both passes
// DCIN: the workgroup derives the window's DC level itself (the same 1024 integer samples, 4 per thread) instead of
// reading the result of dog_dc_kernel: one launch less where launches are what a small batch costs.
// One row-pass workgroup's work: tile rows 16·rb … of the window whose RT block is `b_local`, read from `frame` around
// guess (g1, g2).  DCIN: derive the DC level here; otherwise dc_in is used.  (The kernels below and the cooperative
// single-clip chain, dog_coop.hpp, share it.)
template <int P, int U, bool DCIN, bool FLUSH>
__device__ __forceinline__ void h1_block(const TwoPassGeo &tg, const f2 *__restrict__ taps_row, unsigned char *smem, int b_local, int rb,
                                         const uint8_t *__restrict__ frame, int g1, int g2, int dc_in)
{
    const LaunchGeo &g = tg.g;
    constexpr int NT = 256, NW = NT / 64, XG = NT / HP_ROWS;
    const int L = g.L, H = L >> 1, hw = L >> 1;
    float *A = reinterpret_cast<float *>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int a0 = rb * HP_ROWS;
    const int ti0 = g1 - g.r1 - 1 - hw;
    const int wj0 = g2 - g.r2 - 1 - hw;
    int dc;
    if (DCIN) {
        __shared__ int s_dcsum[NW];
        int sum = dc_sample_sum(g, frame, ti0, wj0, L, tid, NT);
        sum = wave_sum(sum);
        if (lane == 0) s_dcsum[wave] = sum;
        __syncthreads();
        int total = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) total += s_dcsum[w];
        dc = dc_from_sum(total, g.fill);
    } else {
        dc = dc_in;
    }
    // stage 16 tile rows as f32 (pixel − dc); outside the frame = fill − dc.  The row is staged out to the LDS pitch
    // (zeros past the tile): the sliding windows of the last, partly masked output group then read in-bounds
    // without any per-read clamp.  One unaligned dword per 4 pixels, loaded unconditionally at an address clamped
    // into the frame row (a clamped dword still holds every in-frame byte its group needs, at a shifted position);
    // the fill is selected afterwards.  (One byte per lane and iteration cost 2 100 issue slots per wave — a third
    // of them 64-bit scalar address arithmetic — against 4 600 for the whole tap loop.)
    int vm = 0; // max |pixel − dc| over the tile rows staged here
    if (g.fw >= 4) {
        for (int r = wave; r < HP_ROWS; r += NW) {
            const int a = a0 + r, gi = ti0 + a;
            const bool rowok = (a < tg.NA) && gi >= 0 && gi < g.fh;
            const uint8_t *src = frame + (long long)min(max(gi, 0), g.fh - 1) * g.row_stride;
            float *dst = A + r * tg.pitchA;
            for (int c0 = 4 * lane; c0 < tg.pitchA; c0 += 256) {
                const int gj0 = wj0 + c0, gj0c = min(max(gj0, 0), g.fw - 4);
                uint32_t w;
                __builtin_memcpy(&w, src + gj0c, 4);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int c = c0 + i, gj = gj0 + i;
                    const int px = (rowok && gj >= 0 && gj < g.fw) ? (int)((w >> (8 * ((gj - gj0c) & 3))) & 0xffu) : g.fill;
                    if (FLUSH) vm = max(vm, (c < tg.TWin && a < tg.NA) ? abs(px - dc) : 0); // (the plain instances do not collect V)
                    if (c < tg.pitchA) dst[c] = (c < tg.TWin) ? (float)(px - dc) : 0.f;
                }
            }
        }
    } else {
        for (int r = wave; r < HP_ROWS; r += NW) {
            const int a = a0 + r, gi = ti0 + a;
            const bool rowok = (a < tg.NA) && gi >= 0 && gi < g.fh;
            const uint8_t *src = frame + (long long)gi * g.row_stride;
            for (int c = lane; c < tg.pitchA; c += 64) {
                const int gj = wj0 + c;
                int v = g.fill;
                if (rowok && c < tg.TWin && gj >= 0 && gj < g.fw) v = src[gj];
                if (FLUSH && c < tg.TWin && a < tg.NA) vm = max(vm, abs(v - dc));
                A[r * tg.pitchA + c] = (c < tg.TWin) ? (float)(v - dc) : 0.f;
            }
        }
    }
    if (FLUSH && tg.vmax) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) vm = max(vm, __shfl_xor(vm, off, 64));
        if (lane == 0 && vm > 0) atomicMax(&tg.vmax[tg.win0 + b_local], vm);
    }
    __syncthreads();
    const tap_ptr taps = as_taps(taps_row);
    const int r = tid % HP_ROWS, gx = tid / HP_ROWS;
    const int a = a0 + r;
    for (int xb = gx * P; xb < g.n2; xb += XG * P) {
        const float *in = A + r * tg.pitchA + xb; // inputs in[0 .. P+L-2] (+ prefetch overrun: zero padding)
        auto ld = [&](int i) { return in[i]; };
        f2 acc[P];
#pragma unroll
        for (int o = 0; o < P; ++o) acc[o] = f2{0.f, 0.f};
        // The two sliding windows live in register RINGS of R slots (slot = input index mod R on the front side, counted
        // from the block's own origin on the back side): with NB = R/U blocks per trip every slot index is a constant, so
        // sliding costs nothing (shifting the windows by U per block took 52 v_mov_b32 beside 104 + 104 arithmetic
        // instructions: 14 % of the loop's issue time).
        constexpr int R = ((P + 2 * U - 1 + U - 1) / U) * U, NB = R / U;
        float lo[R], hi[R];
#pragma unroll
        for (int j = 0; j < P + U - 1; ++j) {
            lo[j] = ld(j);
            hi[j] = ld(L - U + j);
        }
        auto block = [&](int sb, int k0) { // sb = (k0 / U) mod NB: a constant after unrolling
#pragma unroll
            for (int j = 0; j < U; ++j) { // the next block's new ends, requested before this block's arithmetic
                lo[(sb * U + P + U - 1 + j) % R] = ld(k0 + U + (P - 1) + j);
                hi[(j + 2 * R - (sb + 1) * U) % R] = ld(L - U - (k0 + U) + j);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const f2 t = taps[k0 + u];
#pragma unroll
                for (int o = 0; o < P; ++o)
                    acc[o] = fma_bcast(lo[(sb * U + o + u) % R] + hi[(o + (U - 1) - u + R - sb * U) % R], t, acc[o]);
            }
        };
        const int nb = H / U;
        int bk = 0;
        f2 tot[FLUSH ? P : 1]; // the finished chains' sums (blocked accumulation, see the top of the file)
#pragma unroll
        for (int o = 0; o < (FLUSH ? P : 1); ++o) tot[o] = f2{0.f, 0.f};
        for (; bk + NB <= nb; bk += NB) {
#pragma unroll
            for (int sb = 0; sb < NB; ++sb) block(sb, (bk + sb) * U);
            if (FLUSH) {
#pragma unroll
                for (int o = 0; o < P; ++o) { tot[o] = tot[o] + acc[o]; acc[o] = f2{0.f, 0.f}; }
            }
        }
#pragma unroll
        for (int sb = 0; sb < NB - 1; ++sb)
            if (bk + sb < nb) block(sb, (bk + sb) * U);
        int k0 = nb * U;
        // remaining symmetric pairs (H − k0 < U) one tap at a time, then the centre tap.  (The tail as one more, partial block
        // from the register windows — 2 LDS reads per pair instead of 2·P — was measured: 94 → 112 VGPRs at P = 9 and the row
        // pass 7 % SLOWER at l = 109, 8 % on cfg5.)
        for (; k0 < H; ++k0) {
            const f2 t = taps[k0];
#pragma unroll
            for (int o = 0; o < P; ++o) acc[o] = fma_bcast(ld(o + k0) + ld(o + L - 1 - k0), t, acc[o]);
        }
        {
            const f2 t = taps[H];
#pragma unroll
            for (int o = 0; o < P; ++o) acc[o] = fma_bcast(ld(o + H), t, acc[o]);
        }
        if (FLUSH) {
#pragma unroll
            for (int o = 0; o < P; ++o) acc[o] = tot[o] + acc[o]; // the last (partial) chain joins the others
        }
        if (a < tg.NA) {
            f2 *dst = tg.RT + ((long long)b_local * g.n2 + xb) * tg.NA + a;
#pragma unroll
            for (int o = 0; o < P; ++o)
                if (xb + o < g.n2) dst[(long long)o * tg.NA] = acc[o];
        }
    }
}

template <int P, int U, bool DCIN = false, bool FLUSH = false>
__global__ __launch_bounds__(256) void dog_h1_kernel(const TwoPassGeo tg, const f2 *__restrict__ taps_row)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const LaunchGeo &g = tg.g;
    const int b_local = blockIdx.x / tg.h1blocks_per_win;
    const int rb = blockIdx.x - b_local * tg.h1blocks_per_win;
    const int b = tg.win0 + b_local;
    const int fidx = g.frame_index ? g.frame_index[b] : b;
    h1_block<P, U, DCIN, FLUSH>(tg, taps_row, smem, b_local, rb, g.frames + (long long)fidx * g.frame_stride, g.guesses[2 * b], g.guesses[2 * b + 1],
                         DCIN ? 0 : tg.dc[b]);
}

// ---- column pass + peak (on RT) ----
// FIN: the workgroup that delivers a window's last partial also combines them, maps the index and clamps
// (dog_finalize_kernel's job, :60-61) — no separate launch.  Partials cross workgroups through L2: release fence +
// device-scope counter on the writer side, device-scope loads on the reader side; the counter is left at zero.
// One column-pass workgroup's work on RT block `b_local`, window columns HR·rb …: the workgroup's peak, valid in thread 0
// (ends with a barrier: the caller may reuse the LDS).  b: the window's index for the optional response output.
template <int P, int U, bool RESP, int HR, bool FLUSH>
__device__ __forceinline__ Peak hpass_block(const TwoPassGeo &tg, const f2 *__restrict__ taps_col, unsigned char *smem, int b_local, int rb, int b)
{
    const LaunchGeo &g = tg.g;
    constexpr int NT = 256, NW = NT / 64, XG = NT / HR;
    const int L = g.L;
    f2 *Vs = reinterpret_cast<f2 *>(smem);
    __shared__ float sval[NW], ssec[NW];
    __shared__ int sidx[NW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r0 = rb * HR;                 // first window column x of this block
    const int nrows = min(HR, g.n2 - r0);
    // stage: nrows × NA f2, coalesced
    {
        const f2 *src = tg.RT + ((long long)b_local * g.n2 + r0) * tg.NA;
        for (int r = wave; r < HR; r += NW) {
            f2 *dst = Vs + r * tg.pitchV;
            if (r < nrows) {
                for (int c = lane; c < tg.pitchV; c += 64) dst[c] = (c < tg.NA) ? src[(long long)r * tg.NA + c] : f2{0.f, 0.f};
            } else {
                for (int c = lane; c < tg.pitchV; c += 64) dst[c] = f2{0.f, 0.f};
            }
        }
    }
    __syncthreads();
    const tap_ptr taps = as_taps(taps_col);
    const int r = tid % HR, gx = tid / HR;
    Peak pk;
    peak_init(pk);
    // a workgroup covers XG·P columns per round; wide windows take several rounds
    for (int xb = gx * P; xb < g.n1; xb += XG * P) { // xb: first output row y of this lane's group
        const f2 *a = Vs + r * tg.pitchV + xb;
        f2 acc[P];
#pragma unroll
        for (int o = 0; o < P; ++o) acc[o] = f2{0.f, 0.f};
        // rows are staged out to the LDS pitch (zeros past NA): no clamp on the window reads
        auto ld = [&](int i) { return a[i]; };
        // The sliding window is a register RING of R slots (slot = input index mod R; NB = R/U blocks per trip make every
        // slot index a constant — shifting the window took 22 v_mov_b64 + 15 v_mov_b32 per 112 FMAs), and the taps are
        // taken in whole blocks of U: the tap table ends in ≥ U zeros and the LDS rows in zeros, so the l mod U surplus
        // terms add 0·(finite) = 0 exactly (the one-tap-at-a-time tail they replace shifted the whole window per tap).
        constexpr int R = ((P + 2 * U - 1 + U - 1) / U) * U, NB = R / U;
        f2 win[R];
#pragma unroll
        for (int j = 0; j < P + U - 1; ++j) win[j] = ld(j);
        auto block = [&](int sb, int k0) { // sb = (k0 / U) mod NB: a constant after unrolling
#pragma unroll
            for (int j = 0; j < U; ++j) win[(sb * U + P + U - 1 + j) % R] = ld(k0 + U + (P - 1) + j);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const f2 t = taps[k0 + u];
#pragma unroll
                for (int o = 0; o < P; ++o) acc[o] = fma_pair(win[(sb * U + o + u) % R], t, acc[o]);
            }
        };
        const int nb = (L + U - 1) / U;
        int bk = 0;
        f2 tot[FLUSH ? P : 1]; // the finished chains' sums (blocked accumulation, see the top of the file)
#pragma unroll
        for (int o = 0; o < (FLUSH ? P : 1); ++o) tot[o] = f2{0.f, 0.f};
        for (; bk + NB <= nb; bk += NB) {
#pragma unroll
            for (int sb = 0; sb < NB; ++sb) block(sb, (bk + sb) * U);
            if (FLUSH) {
#pragma unroll
                for (int o = 0; o < P; ++o) { tot[o] = tot[o] + acc[o]; acc[o] = f2{0.f, 0.f}; }
            }
        }
#pragma unroll
        for (int sb = 0; sb < NB - 1; ++sb)
            if (bk + sb < nb) block(sb, (bk + sb) * U);
        if (FLUSH) {
#pragma unroll
            for (int o = 0; o < P; ++o) acc[o] = tot[o] + acc[o]; // the last (partial) chain joins the others
        }
        if (r < nrows) {
            const int x = r0 + r;
#pragma unroll
            for (int o = 0; o < P; ++o) {
                const int y = xb + o;
                if (y < g.n1) {
                    const float v = acc[o].x + acc[o].y;
                    const int lin = x * g.n1 + y;
                    if (RESP) g.resp[(long long)b * g.n1 * g.n2 + lin] = v;
                    peak_push(pk, v, lin);
                }
            }
        }
    }
    peak_wave_reduce(pk);
    if (lane == 0) { sval[wave] = pk.best; sidx[wave] = pk.idx; ssec[wave] = pk.second; }
    __syncthreads();
    if (tid == 0)
        for (int w = 1; w < NW; ++w) peak_merge(pk, sval[w], sidx[w], ssec[w]);
    __syncthreads();
    return pk;
}

template <int P, int U, bool RESP, int HR = HP_ROWS, bool FIN = false, bool FLUSH = false>
__global__ __launch_bounds__(256) void dog_hpass_kernel(const TwoPassGeo tg, const f2 *__restrict__ taps_col)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const LaunchGeo &g = tg.g;
    const int L = g.L;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b_local = blockIdx.x / tg.hblocks_per_win;
    const int rb = blockIdx.x - b_local * tg.hblocks_per_win;
    const int b = tg.win0 + b_local;
    Peak pk = hpass_block<P, U, RESP, HR, FLUSH>(tg, taps_col, smem, b_local, rb, b);
    __shared__ int s_last;
    if (tid == 0) {
        if (FIN) {
            __hip_atomic_store(&g.part_val[b * g.nslots + rb], pk.best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&g.part_idx[b * g.nslots + rb], pk.idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&g.part_sec[b * g.nslots + rb], pk.second, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int old = __hip_atomic_fetch_add(&tg.counter[b], 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            s_last = (old == tg.hblocks_per_win - 1);
        } else {
            g.part_val[b * g.nslots + rb] = pk.best;
            g.part_idx[b * g.nslots + rb] = pk.idx;
            g.part_sec[b * g.nslots + rb] = pk.second;
        }
    }
    if (FIN) {
        __syncthreads();
        if (s_last && wave == 0) {
            Peak w;
            peak_init(w);
            for (int sl = lane; sl < g.nslots; sl += 64)
                peak_merge(w, __hip_atomic_load(&g.part_val[b * g.nslots + sl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                           __hip_atomic_load(&g.part_idx[b * g.nslots + sl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                           __hip_atomic_load(&g.part_sec[b * g.nslots + sl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            peak_wave_reduce(w);
            if (lane == 0) {
                const int bi = w.idx;
                const int x = bi / g.n1, y = bi - x * g.n1;
                tg.out_ij[2 * b] = min(max(g.guesses[2 * b] - g.r1 + y, 1), g.fh);       // :60-61
                tg.out_ij[2 * b + 1] = min(max(g.guesses[2 * b + 1] - g.r2 + x, 1), g.fw);
                range_check(g.ex, g.guesses[2 * b], g.guesses[2 * b + 1], L >> 1, g.fh, g.fw);
                __hip_atomic_store(&tg.counter[b], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // (exact mode does not use this variant: dog_finish_kernel combines, refines and publishes there)
                if (tg.done_flag && b == 0) __hip_atomic_store(tg.done_flag, tg.done_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

} // namespace pdog
