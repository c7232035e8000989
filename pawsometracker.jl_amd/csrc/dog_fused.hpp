// dog_fused.hpp — DoG + argmax for windows whose whole padded tile fits in one CU's LDS, ONE workgroup per
// window and ONE launch per batch or per chain (reference functor /root/reference/src/PawsomeTracker.jl:55-62,
// frame loop :163-169).
//
// This is the latency kernel: the reference's own use is one `Tracker` call per decoded frame with the default
// window (45×45 for target_width 25, `guess_window_size` :64-68), where the batch kernels' four launches per
// frame cost more than the arithmetic (0.9 M FMA).  Here the DC level, both separable passes, the peak search,
// the index map and the clamp (:56-61) run inside a single 1024-thread workgroup with the row-pass result kept
// transposed in LDS, and a chain of frames (`ij[k] = trckr(ij[k-1])`, :167) is a loop inside the kernel: one
// barrier-separated pipeline per frame, no launch and no host round trip between frames.
//
//   tile (n1+l−1)×(n2+l−1) u8 → LDS f32 (pixel − dc)                     coalesced, fill outside the frame (:48)
//   row pass   task = (tile row a, group of PR outputs)  → RT[x][a] (f2: both Gaussians) in LDS
//   column pass task = (window column x, group of PC rows) on RT rows → D, running first maximum
//   workgroup reduction (ties → smallest column-major index, findmax :59) → clamped (row, col)
//
// The arithmetic per output is that of dog_twopass.hpp (row pass: symmetric pairs k ascending, then the centre
// tap; column pass: k ascending), identical for every output, so flat regions tie exactly.
#pragma once
#include "dog_twopass.hpp"
#include "dog_exact.hpp"
#include "dog_roll.hpp"

namespace pdog {

struct FusedGeo {
    LaunchGeo g;
    int NA, TWin;        // padded tile rows / columns
    int pitchA, pitchV;  // LDS pitches: tile rows (floats), RT rows (f2); odd
    int cshift;          // staging: 2^cshift ≥ ⌈TWin/4⌉ threads per tile row, 4 pixels each
    int pr, pc;          // outputs per task in the row / column pass, chosen per geometry so that the tasks fill the 1024 threads
    int chain_len;       // 1: n independent windows.  > 1: clip b = frames b·chain_len …, frame k > 0 starts at frame k−1's result
    int32_t *out_ij;     // [n][chain_len][2], clamped to the frame
    int32_t *done_flag;  // NULL, or a word in host-coherent memory that receives done_value (system-scope release) once
    int32_t done_value;  // window 0's answer is written: the host functor polls it instead of waiting for the kernel's end
    int progress;        // != 0: done_flag receives k + 1 after every frame k of clip 0 instead (a host consumer follows the chain)
    const RefineParams *rp; // exact mode's constants in device memory (dog_exact.hpp); null = off
    int ref_cbw, ref_rows; // the refinement's column-block width and resident tile rows (its scratch is this kernel's dynamic LDS)
    int dc_host;           // ≥ 0: the window's DC level, computed by the host from the same sample grid (the functor packs the tile
                           // itself: no sample loads, no reduction barrier); −1: sampled here
};

constexpr int FUSED_NT = 1024, FUSED_PMAX = 8, FUSED_U = 8;

// LDS pitches (odd ⇒ conflict-free strided reads) with room for the sliding windows of the last, partly masked
// output group of up to FUSED_PMAX outputs and their one-block prefetch
__host__ __device__ constexpr int fused_pitch_a(int n2, int L) { return ((n2 + FUSED_PMAX - 1) / FUSED_PMAX * FUSED_PMAX + L + FUSED_U + 1) | 1; }
__host__ __device__ constexpr int fused_pitch_v(int n1, int L) { return ((n1 + FUSED_PMAX - 1) / FUSED_PMAX * FUSED_PMAX + L + 24 + 1) | 1; } // + 24: the compile-time-l column pass reads whole 16-tap blocks
__host__ __device__ constexpr size_t fused_a_bytes(int n1, int n2, int L) { return ((size_t)(n1 + L - 1) * fused_pitch_a(n2, L) * 4 + 15) / 16 * 16; }
__host__ __device__ constexpr size_t fused_lds_bytes(int n1, int n2, int L) { return fused_a_bytes(n1, n2, L) + (size_t)n2 * fused_pitch_v(n1, L) * 8; }

// PR outputs of one tile row: out[o] = Σ_k (g₊[k], g₋[k]) · in[o+k], symmetric pairs first (k ascending), centre last.
// (The register-ring windows that removed the shifts from the two-pass kernels were measured here too: SLOWER — 10.8 →
// 11.6 µs per 45×45 frame, 13.8 → 14.7 µs per 257×257 frame — under the 128-VGPR budget of a 1024-thread workgroup the three
// unrolled blocks per trip spill, and five P instances of each task triple in code size.)
// (Round 3: prefetching frame k+1's pixels — the current tile grown by the radii, memory → LDS with global_load_lds_dword
// behind frame k's staging barrier, frame k+1 then staged and DC-sampled out of LDS — was built and measured on the
// 100-frame cfg1 chain: 11.84 µs per frame against 11.13 without.  The memory round trip it removes is shorter than the
// phase stamps suggested; issuing 96 wave-loads and unpacking bytes from LDS costs more.  Dropped.)
// Runtime kernel length, blocks of U taps.  (Compile-time-l instances with the tap loop fully unrolled were
// tried for l = 65: 400 SGPR + 300 VGPR spills under the 128-VGPR budget of a 1024-thread workgroup, 2× slower.)
template <int P, int U>
__device__ __forceinline__ void fused_row_task(const float *in, int L, tap_ptr taps, f2 (&acc)[P])
{
    const int H = L >> 1;
    auto ld = [&](int i) { return in[i]; };
#pragma unroll
    for (int o = 0; o < P; ++o) acc[o] = f2{0.f, 0.f};
    float lo[P + U - 1], hi[P + U - 1];
#pragma unroll
    for (int j = 0; j < P + U - 1; ++j) {
        lo[j] = ld(j);
        hi[j] = ld(L - U + j);
    }
    int k0 = 0;
    for (; k0 + U <= H; k0 += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const f2 t = taps[k0 + u];
#pragma unroll
            for (int o = 0; o < P; ++o) acc[o] = fma_bcast(lo[o + u] + hi[o + (U - 1) - u], t, acc[o]);
        }
#pragma unroll
        for (int j = 0; j < P - 1; ++j) lo[j] = lo[j + U];
#pragma unroll
        for (int j = 0; j < U; ++j) lo[P - 1 + j] = ld(k0 + U + (P - 1) + j);
#pragma unroll
        for (int j = P + U - 2; j >= U; --j) hi[j] = hi[j - U];
#pragma unroll
        for (int j = 0; j < U; ++j) hi[j] = ld(L - U - (k0 + U) + j);
    }
    for (; k0 < H; ++k0) {
        const f2 t = taps[k0];
#pragma unroll
        for (int o = 0; o < P; ++o) acc[o] = fma_bcast(ld(o + k0) + ld(o + L - 1 - k0), t, acc[o]);
    }
    const f2 t = taps[H];
#pragma unroll
    for (int o = 0; o < P; ++o) acc[o] = fma_bcast(ld(o + H), t, acc[o]);
}

// PC outputs of one window column: acc[o] = Σ_k (s·g₊[k], −s·g₋[k]) ∘ RT[x][o+k], k ascending
template <int P, int U>
__device__ __forceinline__ void fused_col_task(const f2 *a, int L, tap_ptr taps, f2 (&acc)[P])
{
    auto ld = [&](int i) { return a[i]; };
#pragma unroll
    for (int o = 0; o < P; ++o) acc[o] = f2{0.f, 0.f};
    f2 win[P + U - 1];
#pragma unroll
    for (int j = 0; j < P + U - 1; ++j) win[j] = ld(j);
    int k0 = 0;
    for (; k0 + U <= L; k0 += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const f2 t = taps[k0 + u];
#pragma unroll
            for (int o = 0; o < P; ++o) acc[o] = fma_pair(win[o + u], t, acc[o]);
        }
        // the next block's values are loaded straight into the window (P − 1 register moves per block instead of
        // P − 1 + U); the other waves of the SIMD cover the LDS latency
#pragma unroll
        for (int j = 0; j < P - 1; ++j) win[j] = win[j + U];
#pragma unroll
        for (int j = 0; j < U; ++j) win[P - 1 + j] = ld(k0 + U + (P - 1) + j);
    }
    for (; k0 < L; ++k0) {
        const f2 t = taps[k0];
#pragma unroll
        for (int o = 0; o < P; ++o) acc[o] = fma_pair(win[o], t, acc[o]);
#pragma unroll
        for (int j = 0; j < P + U - 2; ++j) win[j] = win[j + 1];
    }
}

// ---- compile-time kernel length (LT > 0): the default tracker's l = 65 and a few others (fused_has_instance) ----
// Round 2's attempt unrolled the generic tasks with every tap in SGPRs (400 SGPR + 300 VGPR spills).  This one takes the roll
// kernel's row pass (dog_roll.hpp: aligned register pairs out of ds_read_b128 quads, 4-tap blocks with their taps loaded
// beside them — ≈90 VGPRs, no shifts) and a column pass whose sliding window is a register ring with constant indices.
// The tile then lives in the roll kernel's layout: rows on a pitch of whole bank rows, skewed by {0, 1, 8, 9} 16-byte slots
// by (row & 3), so that the b128 reads of a lane group (rows r, r+1 × 8 output groups) are conflict free.  Same operation
// order per output as the runtime-length tasks above: the two kernels' responses are bit-identical.
__host__ __device__ constexpr bool fused_has_instance(int L) { return L % 4 == 1 && L >= 29 && L <= 101; } // lat_lengths.def
__host__ __device__ constexpr int fusedc_pitch_a(int n2, int L) { return (8 * ((n2 + 7) / 8) + L + 39 + 63) / 64 * 64; }
__host__ __device__ constexpr size_t fusedc_a_bytes(int n1, int n2, int L) { return ((size_t)(n1 + L - 1) * fusedc_pitch_a(n2, L) + 64) * 4; }
__host__ __device__ constexpr size_t fusedc_lds_bytes(int n1, int n2, int L) { return fusedc_a_bytes(n1, n2, L) + (size_t)n2 * fused_pitch_v(n1, L) * 8; }
// outputs per row-pass task of the compile-time-l instances: 8, or 4 where that loads the busiest SIMD less (waves of 64 tasks,
// four SIMDs; ≈51 / ≈53 instructions per output)
__host__ __device__ constexpr int fusedc_row_outputs(int NA, int n2)
{
    const int w8 = (NA * ((n2 + 7) / 8) + 63) / 64, w4 = (NA * ((n2 + 3) / 4) + 63) / 64;
    return ((w4 + 3) / 4) * 212 < ((w8 + 3) / 4) * 408 ? 4 : 8;
}
__device__ __forceinline__ int fusedc_row_base(int r, int pitch) { return r * pitch + 4 * ((r & 1) + ((r & 2) ? 8 : 0)); }

// P outputs of one window column, l known: taps in blocks of 8 (one s_load_dwordx16 each, requested a block ahead), the window
// entry e lives in register e mod (P − 1 + 16) — constant after unrolling, so nothing is ever shifted.
template <int L, int P>
__device__ __forceinline__ void fusedc_col_task(const f2 *a, tap_ptr taps, f2 (&acc)[P])
{
    // (taps read from an LDS copy instead of scalar loads — no lgkmcnt(0) drains — measured slower: 8.5 against 8.0 µs per frame, the tap
    // registers push the kernel over its 128 VGPRs;
    // a prefetch distance of two blocks measured equal to one; taps and window entries of block J + 1 are requested while
    // block J's FMAs issue — the compiler sinks the requests below the wait for block J's own)
    constexpr int U = 8, NB = L / U, R = L - U * NB, W = P - 1 + 2 * U;
    static_assert(R >= 1 && R < U, "kernel lengths are odd");
    f2 win[W];
    f2 tq[2][U];
#pragma unroll
    for (int o = 0; o < P; ++o) acc[o] = f2{0.f, 0.f};
    tap_ptr tb = pin_taps(taps);
#pragma unroll
    for (int e = 0; e < P - 1 + U; ++e) win[e % W] = a[e];
#pragma unroll
    for (int j = 0; j < U; ++j) tq[0][j] = tb[j];
#pragma unroll
    for (int J = 0; J < NB; ++J) {
        const int nb = J + 1;
        const int nnew = nb < NB ? U : R; // what block nb (or the tail of R taps) needs beyond what is there
        tb = pin_taps(tb);
#pragma unroll
        for (int j = 0; j < U; ++j)
            if (j < nnew) {
                tq[nb & 1][j] = tb[U * nb + j];
                win[(U * nb + P - 1 + j) % W] = a[U * nb + P - 1 + j];
            }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int o = 0; o < P; ++o) acc[o] = fma_pair(win[(U * J + u + o) % W], tq[J & 1][u], acc[o]);
#pragma unroll
        for (int o = 0; o < P; ++o) pin_acc(acc[o]);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int u = 0; u < R; ++u)
#pragma unroll
        for (int o = 0; o < P; ++o) acc[o] = fma_pair(win[(U * NB + u + o) % W], tq[NB & 1][u], acc[o]);
}

// DIAG != 0 (diagnostic builds only): thread 0 of block 0 stamps the phase boundaries of every frame into g.resp
// (16 floats per frame: shader cycles since the frame's start at 0 samples reduced (wave 0), 4 after the barrier,
// 1 tile staged, 2 row pass, 5 column pass + wave peak (wave 0), 6 after the barrier, 3 end of frame; then the same
// in 100 MHz ticks) instead of the response.
template <bool RESP, int DIAG = 0, int LT = 0>
__global__ __launch_bounds__(FUSED_NT) void dog_fused_kernel(const FusedGeo fg, const f2 *__restrict__ taps_row,
                                                             const f2 *__restrict__ taps_col)
{
    const LaunchGeo &g = fg.g;
    constexpr int NT = FUSED_NT, NW = NT / 64, U = FUSED_U;
    const int L = LT ? LT : g.L, hw = L >> 1, NA = fg.NA;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *A = reinterpret_cast<float *>(smem);
    f2 *Vs = reinterpret_cast<f2 *>(smem + (LT ? fusedc_a_bytes(g.n1, g.n2, L) : fused_a_bytes(g.n1, g.n2, L)));
    __shared__ int s_sum[NW];
    __shared__ float s_val[NW];
    __shared__ int s_idx[NW];
    __shared__ int s_guess[2];
    __shared__ float s_sec[NW];
    __shared__ int s_refine;
    __shared__ float s_vmax[NW]; // a flagged frame: max |pixel − dc| over the tile, per wave
    __shared__ float s_max, s_sec2;
    __shared__ int s_idx2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int b = blockIdx.x; // this workgroup's window (or clip): b, b + gridDim.x, … — the host launches no more workgroups than stay resident
    const tap_ptr trow = as_taps(taps_row), tcol = as_taps(taps_col);

    // tile columns c ≥ TWin and RT columns a ≥ NA are only ever read by the sliding windows of masked outputs: zero once
    // (and again after a refinement, which uses this LDS as its scratch)
    auto zero_padding = [&]() {
        if (LT) return; // the compile-time-l tasks read unstaged entries only into outputs that are masked out
        for (int r = wave; r < NA; r += NW)
            for (int c = fg.TWin + lane; c < fg.pitchA; c += 64) A[r * fg.pitchA + c] = 0.f;
        for (int x = wave; x < g.n2; x += NW)
            for (int c = NA + lane; c < fg.pitchV; c += 64) Vs[x * fg.pitchV + c] = f2{0.f, 0.f};
    };
    // Interior frames of the compile-time-l instances (the whole tile inside the frame — every frame of a clip but those at
    // the border): a thread's addresses relative to the tile's origin never change, so a frame issues its five loads off one
    // uniform base with no per-thread arithmetic, and unpacks with four conversions and a packed subtract per dword.
    constexpr int SU = 4; // rows per thread and batch: the default 45×45 window (109 tile rows, 32 per pass) needs exactly 4
    const int TW4 = (fg.TWin + 3) & ~3;

    // A workgroup walks its windows (clips) one after the other, like the frames of a clip: what a window costs beyond its frames —
    // workgroup dispatch, the kernel's prologue, the first tap loads — is paid once per workgroup (4096 windows of 45×45:
    // 12 → ≈9.5 µs per window and CU).
    for (; b < g.n; b += gridDim.x) {
    zero_padding(); // (the previous window may have been refined: the refinement uses this LDS as its scratch)
    int g1 = g.guesses[2 * b], g2 = g.guesses[2 * b + 1];
    for (int k = 0; k < fg.chain_len; ++k) {
        const long long fidx = fg.chain_len > 1 ? (long long)b * fg.chain_len + k : (g.frame_index ? g.frame_index[b] : b);
        const uint8_t *__restrict__ frame = g.frames + fidx * g.frame_stride;
        const int ti0 = g1 - g.r1 - 1 - hw, wj0 = g2 - g.r2 - 1 - hw;
        unsigned long long dc0 = 0, dr0 = 0;
        auto stamp = [&](int i) {
            if (DIAG && tid == 0 && b == 0) {
                g.resp[16 * k + i] = (float)(__builtin_amdgcn_s_memtime() - dc0);
                g.resp[16 * k + 8 + i] = (float)(__builtin_amdgcn_s_memrealtime() - dr0);
            }
        };
        if (DIAG) { dc0 = __builtin_amdgcn_s_memtime(); dr0 = __builtin_amdgcn_s_memrealtime(); }
        // chains (diagnostic build): every wave's own clock at four points of frame 20, behind the per-frame stamps
        auto wstamp = [&](int i) {
            if (DIAG && fg.chain_len > 20 && k == 20 && lane == 0 && b == 0)
                g.resp[16 * fg.chain_len + 16 * i + wave] = (float)(__builtin_amdgcn_s_memtime() - dc0);
        };
        // ---- per-window DC level (see dog_kernels.hpp) and staging.  Thread → (tile row tid >> cshift, 4-byte column
        // group tid & (2^cshift − 1)): one unaligned dword load per 4 pixels, no division.  The thread's DC sample
        // (one of the 32×32 grid of dc_sample_sum) and its first SU dwords are requested together — one memory round
        // trip — and the tile goes to LDS as (float)(pixel − dc) once the integer sample sum has been reduced.
        // Loads are unconditional at addresses clamped into the frame and the fill is selected afterwards (no branch
        // between two loads: all of a batch are in flight together); a clamped dword still holds every in-frame
        // byte its group needs, at a shifted position ----
        const bool interior = LT && ((NT >> fg.cshift) & 3) == 0 && ti0 >= 0 && ti0 + NA <= g.fh && wj0 >= 0 && wj0 + TW4 <= g.fw &&
                              !(fg.dc_host >= 0 && fg.chain_len == 1);
        // (the thread's offsets are derived again every frame from an opaque copy of its index: kept across the frame loop they
        // take nine registers out of the row pass, which has none to spare)
        int tid_f = tid;
        asm volatile("" : "+v"(tid_f));
        const int st_q = tid_f & ((1 << fg.cshift) - 1), st_r0 = tid_f >> fg.cshift, st_step = NT >> fg.cshift, st_c0 = 4 * st_q;
        if (interior) {
            const unsigned rs32 = (unsigned)g.row_stride;
            const unsigned st_off0 = (unsigned)st_r0 * rs32 + (unsigned)st_c0;                          // bytes from the tile's first pixel
            const unsigned st_soff = (unsigned)(((tid_f >> 5) * NA) >> 5) * rs32 + (unsigned)(((tid_f & 31) * fg.TWin) >> 5); // this thread's DC sample
            const int st_lds0 = fusedc_row_base(st_r0, fg.pitchA) + st_c0;                               // floats; st_step is a multiple of 4: one skew for all of a thread's rows
            const bool st_col = st_c0 < fg.TWin;
            const uint8_t *__restrict__ base = frame + ((long long)ti0 * g.row_stride + wj0); // uniform
            const int samp = base[st_soff];
            uint32_t v[SU];
            auto load_batch = [&](int it) {
#pragma unroll
                for (int u = 0; u < SU; ++u) {
                    // unconditional (a load inside a branch is waited for where the branch ends): idle threads read the tile's first dword
                    const int r = st_r0 + (it * SU + u) * st_step;
                    unsigned off = (st_col && r < NA) ? st_off0 + (unsigned)((it * SU + u) * st_step) * rs32 : 0u;
                    asm volatile("" : "+v"(off)); // (opaque: the compiler otherwise turns the select into a branch around a second, uniform-address load and waits for it with vmcnt(0))
                    __builtin_memcpy(&v[u], base + off, 4);
                }
            };
            load_batch(0);
            // (Tried: the NEXT frame's tile at this frame's position requested behind these loads into a dummy LDS area
            // (global_load_lds, no registers) so that its lines sit in this CU's vector cache a frame later — 8.9 against
            // 8.5 µs per frame: the sample phase did not get shorter (it is address arithmetic and issue, not the memory
            // round trip) and the workgroup barrier's fence waits for the extra loads.)
            const int wsum = wave_sum(samp);
            if (lane == 0) s_sum[wave] = wsum;
            stamp(0);
            __syncthreads();
            stamp(4);
            int total = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) total += s_sum[w];
            const float fdc = (float)dc_from_sum(total, g.fill);
            for (int it = 0; st_r0 + it * SU * st_step < NA; ++it) {
                if (it) load_batch(it);
#pragma unroll
                for (int u = 0; u < SU; ++u) {
                    const int r = st_r0 + (it * SU + u) * st_step;
                    if (st_col && r < NA) {
                        const f4 px = f4{(float)(v[u] & 0xffu), (float)((v[u] >> 8) & 0xffu), (float)((v[u] >> 16) & 0xffu), (float)(v[u] >> 24)};
                        *reinterpret_cast<f4 *>(A + st_lds0 + (it * SU + u) * st_step * fg.pitchA) = px - fdc;
                    }
                }
            }
        } else {
            const int q = st_q, sr0 = st_r0, srstep = st_step;
            const int c0 = 4 * q, gj0 = wj0 + c0, gj0c = min(max(gj0, 0), g.fw - 4);
            const uint8_t *colp = frame + gj0c;
            auto load_batch = [&](int r0, uint32_t (&v)[SU]) {
#pragma unroll
                for (int u = 0; u < SU; ++u) {
                    const int gi = ti0 + r0 + u * srstep;
                    __builtin_memcpy(&v[u], colp + (long long)min(max(gi, 0), g.fh - 1) * g.row_stride, 4);
                }
            };
            const bool host_dc = fg.dc_host >= 0 && fg.chain_len == 1;
            int samp = 0;
            if (!host_dc) {
                const int si = ti0 + (int)(((long long)(tid >> 5) * NA) >> 5), sj = wj0 + (int)(((long long)(tid & 31) * fg.TWin) >> 5);
                samp = frame[(long long)min(max(si, 0), g.fh - 1) * g.row_stride + min(max(sj, 0), g.fw - 1)];
                if (!(si >= 0 && si < g.fh && sj >= 0 && sj < g.fw)) samp = g.fill;
            }
            uint32_t v[SU];
            load_batch(sr0, v);
            int dc = fg.dc_host;
            if (!host_dc) {
                samp = wave_sum(samp);
                if (lane == 0) s_sum[wave] = samp;
                stamp(0);
                __syncthreads();
                stamp(4);
                int total = 0;
#pragma unroll
                for (int w = 0; w < NW; ++w) total += s_sum[w];
                dc = dc_from_sum(total, g.fill);
            }
            const bool colin = gj0 >= 0 && gj0 + 4 <= g.fw;
            const float fdc = (float)dc;
            for (int r0 = sr0; r0 < NA; r0 += srstep * SU) {
                if (r0 != sr0) load_batch(r0, v);
#pragma unroll
                for (int u = 0; u < SU; ++u) {
                    const int r = r0 + u * srstep, gi = ti0 + r;
                    const bool rowok = gi >= 0 && gi < g.fh;
                    if (LT) {
                        // four pixels → one 16-byte store: v_cvt_f32_ubyte0…3 and a packed subtract where the dword lies inside
                        // the frame (float(px) − float(dc) = float(px − dc) exactly), the per-pixel selection only at the border
                        if (r < NA && c0 < fg.TWin) {
                            f4 px;
                            if (rowok && colin) {
                                px = f4{(float)(v[u] & 0xffu), (float)((v[u] >> 8) & 0xffu), (float)((v[u] >> 16) & 0xffu), (float)(v[u] >> 24)};
                            } else {
#pragma unroll
                                for (int i = 0; i < 4; ++i) {
                                    const int gj = gj0 + i;
                                    px[i] = (float)((rowok && gj >= 0 && gj < g.fw) ? (int)((v[u] >> (8 * ((gj - gj0c) & 3))) & 0xffu) : g.fill);
                                }
                            }
                            *reinterpret_cast<f4 *>(A + fusedc_row_base(r, fg.pitchA) + c0) = px - fdc;
                        }
                        continue;
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int gj = gj0 + i;
                        const int px = (rowok && gj >= 0 && gj < g.fw) ? (int)((v[u] >> (8 * ((gj - gj0c) & 3))) & 0xffu) : g.fill;
                        if (c0 + i < fg.TWin && r < NA) A[r * fg.pitchA + c0 + i] = (float)(px - dc);
                    }
                }
            }
        }
        wstamp(3);
        __syncthreads();
        stamp(1);
        // ---- row pass → RT[x][a] ----
        if constexpr (LT > 0) {
            // task = (tile row a, group of PR = 8 or 4 outputs), dense over the threads; the division by the group count is a float
            // multiply (task + ½ never comes within 1/(2·ngx) of a multiple of ngx: exact for every task count that fits LDS).
            // PR is the host's choice (fusedc_row_outputs): 4 where tasks of 8 would leave most waves without one.
            auto run = [&](auto Pc) {
                constexpr int PR = decltype(Pc)::value;
                const int ngx = (g.n2 + PR - 1) / PR, ntask = NA * ngx;
                const float inv = 1.0f / (float)ngx;
                for (int task = tid; task < ntask; task += NT) {
                    const int a = (int)(((float)task + 0.5f) * inv), gx = task - a * ngx, xb = PR * gx;
                    f2 acc[PR];
#pragma unroll
                    for (int o = 0; o < PR; ++o) acc[o] = f2{0.f, 0.f};
                    roll_row_pass<LT, PR>(acc, A + fusedc_row_base(a, fg.pitchA) + xb, trow);
#pragma unroll
                    for (int o = 0; o < PR; ++o)
                        if (xb + o < g.n2) Vs[(xb + o) * fg.pitchV + a] = acc[o];
                }
            };
            if (fg.pr == 4) run(std::integral_constant<int, 4>{});
            else run(std::integral_constant<int, 8>{});
        } else {
            const int ngx = (g.n2 + fg.pr - 1) / fg.pr, ntask = NA * ngx;
            auto run = [&](auto Pc) {
                constexpr int PR = decltype(Pc)::value;
                for (int task = tid; task < ntask; task += NT) {
                    const int gx = task / NA, a = task - gx * NA, xb = gx * PR;
                    f2 acc[PR];
                    fused_row_task<PR, U>(A + a * fg.pitchA + xb, L, trow, acc);
#pragma unroll
                    for (int o = 0; o < PR; ++o)
                        if (xb + o < g.n2) Vs[(xb + o) * fg.pitchV + a] = acc[o];
                }
            };
            switch (fg.pr) {
            case 3: run(std::integral_constant<int, 3>{}); break;
            case 4: run(std::integral_constant<int, 4>{}); break;
            case 5: run(std::integral_constant<int, 5>{}); break;
            case 6: run(std::integral_constant<int, 6>{}); break;
            default: run(std::integral_constant<int, 8>{}); break;
            }
        }
        wstamp(0);
        __syncthreads();
        stamp(2);
        // ---- column pass + peak ----
        Peak pk;
        peak_init(pk);
        {
            const int ngy = (g.n1 + fg.pc - 1) / fg.pc, ntask = g.n2 * ngy;
            auto run = [&](auto Pc) {
                constexpr int PC = decltype(Pc)::value;
                const float inv = 1.0f / (float)g.n2;
                for (int task = tid; task < ntask; task += NT) {
                    const int gy = LT ? (int)(((float)task + 0.5f) * inv) : task / g.n2, x = task - gy * g.n2, yb = gy * PC;
                    f2 acc[PC];
                    if constexpr (LT > 0)
                        fusedc_col_task<LT, PC>(Vs + x * fg.pitchV + yb, tcol, acc);
                    else
                        fused_col_task<PC, U>(Vs + x * fg.pitchV + yb, L, tcol, acc);
#pragma unroll
                    for (int o = 0; o < PC; ++o) {
                        const int y = yb + o;
                        if (y < g.n1) {
                            const float v = acc[o].x + acc[o].y;
                            const int lin = x * g.n1 + y;
                            if (RESP && !DIAG) g.resp[(long long)b * g.n1 * g.n2 + lin] = v;
                            peak_push(pk, v, lin);
                        }
                    }
                }
            };
            switch (fg.pc) {
            case 2: run(std::integral_constant<int, 2>{}); break;
            case 3: run(std::integral_constant<int, 3>{}); break;
            case 4: run(std::integral_constant<int, 4>{}); break;
            case 6: run(std::integral_constant<int, 6>{}); break;
            default: run(std::integral_constant<int, 8>{}); break;
            }
        }
        wstamp(1);
        peak_wave_reduce(pk);
        if (lane == 0) { s_val[wave] = pk.best; s_idx[wave] = pk.idx; s_sec[wave] = pk.second; }
        wstamp(2);
        stamp(5);
        __syncthreads();
        stamp(6);
        const bool publish = fg.done_flag && b == 0 && (fg.progress || k == fg.chain_len - 1);
        int32_t *const o_ij = fg.out_ij + 2 * ((long long)b * fg.chain_len + k);
        if (wave == 0) { // the 16 wave peaks: lanes 0..15 of wave 0, same tie rule
            peak_init(pk);
            if (lane < NW) { pk.best = s_val[lane]; pk.idx = s_idx[lane]; pk.second = s_sec[lane]; }
            peak_wave_reduce(pk, NW);
            if (lane == 0) {
                // column = idx ÷ n1 by a float multiply (a window that fits LDS has < 2^15 pixels: idx + ½ stays further from a
                // multiple of n1 than the rounding of the product) — an integer division is ≈40 dependent instructions here
                const int x = (int)(((float)pk.idx + 0.5f) * (1.0f / (float)g.n1)), y = pk.idx - x * g.n1;
                const int i = min(max(g1 - g.r1 + y, 1), g.fh);   // :60-61
                const int j = min(max(g2 - g.r2 + x, 1), g.fw);
                o_ij[0] = i;
                o_ij[1] = j;
                s_guess[0] = i;
                s_guess[1] = j;
                if (k == 0) range_check(g.ex, g1, g2, hw, g.fh, g.fw);
                // exact mode (dog_exact.hpp): a runner-up within 2δ of the maximum → the reference's own arithmetic decides
                const bool rf = fg.rp && (pk.best - pk.second <= g.ex.T);
                s_refine = rf;
                s_max = pk.best;
                s_sec2 = pk.second;
                s_idx2 = pk.idx;
                if (!rf && publish)
                    __hip_atomic_store(fg.done_flag, fg.progress ? k + 1 : fg.done_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        __syncthreads();
        // A gap within the V = 255 bound.  The bound is proportional to the window's own V = max |pixel − dc| (dog_exact.hpp), and the
        // tile is still in LDS as exactly those differences: one pass over it (≈0.5 µs, flagged frames only) usually withdraws the
        // flag — a frame of ±2-level noise has V ≈ 3 — and a refinement that does start knows V (no pass over the pixels in memory).
        int vknown = -1;
        bool refine = s_refine != 0;
        if (refine && g.ex.T < __builtin_huge_valf()) {
            // (|x| compared as integers: no NaN logic; indices clamped instead of predicated — duplicates do not change a maximum —
            // so that 16 independent LDS reads are in flight per step)
            int vbits = 0;
            for (int r0 = wave; r0 < NA; r0 += 4 * NW)
                for (int cb = 0; cb < fg.TWin; cb += 256) {
                    int rd[4][4];
#pragma unroll
                    for (int kr = 0; kr < 4; ++kr) {
                        const int r = min(r0 + kr * NW, NA - 1);
                        const int *row = reinterpret_cast<const int *>(A + (LT ? fusedc_row_base(r, fg.pitchA) : r * fg.pitchA));
#pragma unroll
                        for (int j = 0; j < 4; ++j) rd[kr][j] = row[min(cb + lane + 64 * j, fg.TWin - 1)];
                    }
#pragma unroll
                    for (int kr = 0; kr < 4; ++kr)
#pragma unroll
                        for (int j = 0; j < 4; ++j) vbits = max(vbits, rd[kr][j] & 0x7fffffff);
                }
            const float vloc = __builtin_bit_cast(float, vbits);
            const float vw = wave_max(vloc);
            if (lane == 0) s_vmax[wave] = vw;
            __syncthreads();
            float vm = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) vm = fmaxf(vm, s_vmax[w]);
            vknown = (int)vm;
            refine = s_max - s_sec2 <= g.ex.T * (vm * (1.0f / 255.0f)) * 1.00001f;
            if (!refine && tid == 0 && publish) // withdrawn: the FP32 answer (already written) stands
                __hip_atomic_store(fg.done_flag, fg.progress ? k + 1 : fg.done_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        if (refine) {
            if (tid == 0) atomicAdd(g.ex.stat, 1ull);
            // A near-tie: the reference's own arithmetic decides (dog_exact.hpp: FP32 rescan → separable Float64 →
            // sequential dense chains for genuine ties).  The tile and RT are not needed any more: their LDS is the scratch.
            // (inlined: as an out-of-line call it made EVERY launch slower — 26 → 46 µs per functor call, the kernel then
            // carries a stack; inlined, its registers spill a little into the rare path only)
            // (An LDS copy of the window's responses for the map path of refine_window was tried as well: its stage 2 is
            // register-hungry — 82 VGPR spills, 332 B of scratch under this kernel's 128-VGPR budget — and every launch paid:
            // functor 26 → 35 µs, chain 10.8 → 13.0 µs per frame, hard frames no faster.)
            const refine_params_ptr rp = (refine_params_ptr)(unsigned long long)fg.rp; // read here, in the rare branch
            RefineCtx c;
            c.trow = trow;
            c.tcol = tcol;
            c.K = (k64_ptr)(unsigned long long)rp->K64;
            c.g64 = (k64_ptr)(unsigned long long)rp->g64;
            c.dir = rp->dir;
            c.T64 = rp->T64;
            c.T = g.ex.T;
            c.T_rescan = g.ex.T_rescan;
            c.vmax_known = vknown;
            c.second = s_sec2;
            c.fp32_idx = s_idx2;
            c.cbw = fg.ref_cbw;
            c.tile_rows = fg.ref_rows;
            c.lds = smem;
            const int idx = refine_window<4>(NT, g, frame, g1, g2, s_max, c, [](int, int, float) { return true; });
            if (tid == 0) {
                const int x = idx / g.n1, y = idx - x * g.n1;
                const int i = min(max(g1 - g.r1 + y, 1), g.fh);
                const int j = min(max(g2 - g.r2 + x, 1), g.fw);
                o_ij[0] = i;
                o_ij[1] = j;
                s_guess[0] = i;
                s_guess[1] = j;
                if (publish)
                    __hip_atomic_store(fg.done_flag, fg.progress ? k + 1 : fg.done_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            __syncthreads();
            if (k + 1 < fg.chain_len) { zero_padding(); }
        }
        stamp(3);
        g1 = s_guess[0];   // :167 — the next frame's guess
        g2 = s_guess[1];
    }
    __syncthreads(); // (s_guess has been read by everyone before the next window's wave 0 writes it)
    }
}

// frames finished so far, for paths whose own kernels do not publish it (see pdog_detect_chain_progress)
#ifndef PDOG_ROLL_INST_ONLY
static __global__ void dog_publish_kernel(int32_t *flag, int32_t value)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
#endif

} // namespace pdog
