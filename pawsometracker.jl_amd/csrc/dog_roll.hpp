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
//               REGISTERS (S = l−1+CH f32 accumulator slots, output y in slot y mod S, two
//               adjacent outputs per register pair / v_pk_fma_f32), so each
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
#include "dog_exact.hpp"
#include <type_traits>

namespace pdog {

#ifndef PDOG_ROLL_LMAX
#define PDOG_ROLL_LMAX 149
#endif
constexpr int ROLL_CH = 8;   // rows per sub-chunk (16 measured equal on cfg3: the kernel is VALU-bound, not latency-bound)
constexpr int ROLL_P = 8;    // row-pass outputs per lane
constexpr int ROLL_TW = 64;  // strip width = lanes
constexpr int ROLL_PR = 65;  // R pitch (f2)
constexpr int ROLL_LMIN = 17, ROLL_LMAX = PDOG_ROLL_LMAX; // kernel lengths with a roll instance (l = 4m+1): the l + 7 accumulators, the row-pass
                                              // windows and the loop state fit 168 VGPRs up to l = 81 (three waves per SIMD) and 256 up to l = 97
                                              // (two); longer kernels go to dog_twopass.hpp (round 2's l = 101 / 105 instances spilled and lost to it)
__host__ __device__ constexpr int roll_waves(int L) { return L <= 81 ? 3 : 2; } // waves per SIMD the instance is compiled for
// Kernel lengths whose instances can FOLD a single remainder column into the last strip (LaunchGeo::fold_r).  The second
// row-pass variant costs ≈18 VGPRs: l = 65 (the reference's default target_width) absorbs them inside its three-waves budget
// (163 of 168); the neighbouring lengths would lose a wave per SIMD to it and keep dog_thin_kernel instead.
__host__ __device__ constexpr bool roll_folds(int L) { return L == 65; }

// accumulator slots: the l outputs in flight plus the sub-chunk being emitted, rounded so that the
// slot ↔ tap mapping repeats after a whole number of sub-chunks
__host__ __device__ constexpr int roll_slots(int L) { return (L - 1 + ROLL_CH + ROLL_CH - 1) / ROLL_CH * ROLL_CH; }
// staging: 8 lanes per input row, each SB (multiple of 4) bytes.  A rows are 16-byte aligned and their bases are
// skewed by {0,1,8,9} 16-byte slots (row & 3) on a pitch that is a multiple of 256 B: the row pass reads ds_read_b128
// quads at (row base + 8·group + 4·q) floats, and with that skew the 16 lanes of every b128 lane group
// (MI355X_MICROARCH.md, LDS) land on 16 distinct slots of the 256-B bank row — conflict-free.
__host__ __device__ constexpr int roll_sb(int L) { return ((ROLL_TW + L - 1 + 7) / 8 + 3) / 4 * 4; }
__host__ __device__ constexpr int roll_rs(int L) { return (8 * roll_sb(L) + 40 + 63) / 64 * 64; } // A row pitch in floats (36 of skew + the 129th column of a folded strip)
__host__ __device__ constexpr int roll_row_base(int r, int L) { return r * roll_rs(L) + 4 * (((r & 1) ? 1 : 0) + ((r & 2) ? 8 : 0)); }
__host__ __device__ constexpr size_t roll_lds_bytes(int L) { return (size_t)ROLL_CH * roll_rs(L) * 4 + (size_t)ROLL_CH * ROLL_PR * 8; }

// Re-derive a tap pointer through an empty asm: the scalar loads that use it cannot be hoisted above
// this point (hoisted, every block's taps are live at once and the SGPRs spill through v_writelane).
// A fake use + redefinition of an accumulator pair: the FMAs that feed it cannot be sunk below this
// point and later ones cannot be hoisted above it (keeps each tap block's FMAs next to its taps).
__device__ __forceinline__ void pin_acc(f2 &a) { asm volatile("" : "+v"(a)); }

__device__ __forceinline__ tap_ptr pin_taps(tap_ptr p)
{
    unsigned long long u = (unsigned long long)p;
    asm volatile("" : "+s"(u));
    return (tap_ptr)u;
}

// Row pass for one lane: P = 8 outputs, symmetric taps, everything in 4-cycle packed ops
// (a lone wave issues one VALU instruction per ≈4.8 cycles whatever its width, so 2-cycle
// scalar ops would leave the pipe half empty at few waves per SIMD):
//   s2      = (a[o+k], a[o+1+k]) + (a[o+L-1-k], a[o+L-k])                 v_pk_add_f32
//   acc[o]  += s2.x · (g+[k], g−[k]);  acc[o+1] += s2.y · (g+[k], g−[k])   v_pk_fma_f32 (op_sel broadcast)
// VGPR pairs must be even-aligned and L − 1 is a multiple of 4, so o + k must be even: EVEN taps pair the outputs
// (0,1) (2,3) (4,5) (6,7), ODD taps pair (−1,0) (1,2) (3,4) (5,6) (7,8) — a fifth v_pk_add_f32 whose outer halves
// are not used, instead of the ≈50 v_pk_mov_b32 / v_mov_b32 per sub-chunk that assembled odd-aligned register pairs
// in round 2.  Every operand is then an even pair E[p] = (a[2p], a[2p+1]) of the lane's 16-byte aligned span: the
// inputs arrive as ds_read_b128 quads, each read once, and the sliding windows shrink to 3 + 3 quads (24 VGPRs
// instead of 40 + 16 staged) — what lets the kernel run three waves per SIMD.
//   tap k = 4J+u, block J:  lo pairs E[2J … 2J+5] = quads J … J+2,  hi pairs E[H−2J−2 … H−2J+3] = quads H/2−J−1 … H/2−J+1
// a = &A[r][P*gx] (16-byte aligned); inputs a[0 .. P+L-2].  Taps ascending k = 0..H-1, centre last: same order for
// every output, so equal inputs give bit-equal outputs.
typedef float f4 __attribute__((ext_vector_type(4)));
// NOUT = 9: the lane also computes output 8 — for the last lane group of a FOLDED strip that is window column 64·k of a
// window 64·k + 1 columns wide (257, 513, …), the column round 2 gave to dog_thin_kernel.  The odd taps already hold its
// pair sums (the outer half of their fifth v_pk_add_f32), the even taps add one: +49 packed instructions per sub-chunk
// in the one strip per window that folds, instead of a second kernel re-reading a 65-column patch per window.
// NOUT = 4: four outputs per lane (pairs (0,1) (2,3) / (−1,0) (1,2) (3,4), two-quad windows) for the latency kernels' small tiles,
// where tasks of 8 outputs leave most of a 1024-thread workgroup without one (dog_tiled.hpp).
template <int L, int NOUT = ROLL_P>
__device__ __forceinline__ void roll_row_pass(f2 (&acc)[NOUT == 9 ? ROLL_P + 2 : NOUT], const float *a, tap_ptr taps)
{
    constexpr int P = (NOUT == 9) ? 8 : NOUT, H = L / 2, U = 4, NB = (H + U - 1) / U, HQ = H / 2;
    constexpr int NLO = P / 4 + 1;                       // lo-window quads: pairs 2J … 2J + P/2 + 1
    constexpr int NHI = NLO + (NOUT == 9 ? 1 : 0);       // hi-window quads
    constexpr int ME = P / 2 + (NOUT == 9 ? 1 : 0);      // pair sums per even tap
    constexpr int MO = P / 2 + 1;                        // pair sums per odd tap
    static_assert((P == 8 || P == 4) && (NOUT == 4 || NOUT == 8 || NOUT == 9) && H % 2 == 0, "the pairing below is written for 4 or 8 (+1) outputs and L = 4m + 1");
    auto quad = [&](int q) { return *reinterpret_cast<const f4 *>(a + 4 * q); };
    auto half = [](const f4 &v, int h) { return h ? __builtin_shufflevector(v, v, 2, 3) : __builtin_shufflevector(v, v, 0, 1); };
    f4 lw[NLO], hw[NHI]; // quads J … J+NLO−1 and HQ−J−1 … HQ−J−2+NHI
#pragma unroll
    for (int j = 0; j < NLO; ++j) lw[j] = quad(j);
#pragma unroll
    for (int j = 0; j < NHI; ++j) hw[j] = quad(HQ - 1 + j);
    f2 tn[U];
    tap_ptr tb = pin_taps(taps); // ONE base re-pinned in place per block: the loads below keep constant offsets from it
#pragma unroll                   // (an opaque pointer per block made 9 loop-invariant address pairs, all spilled to VGPR lanes)
    for (int j = 0; j < U; ++j) tn[j] = tb[j];
#pragma unroll
    for (int J = 0; J < NB; ++J) {
        const int k0 = U * J;
        const int nu = (H - k0 < U) ? (H - k0) : U; // the last block holds two taps when H is not a multiple of 4
        const bool more = (J + 1 < NB);
        f2 t[U];
#pragma unroll
        for (int j = 0; j < U; ++j) t[j] = tn[j];
        f4 nl = lw[NLO - 1], nh = hw[0];
        tb = pin_taps(tb);
        const tap_ptr tnext = tb + (more ? k0 + U : H);
        if (more) {
#pragma unroll
            for (int j = 0; j < U; ++j) tn[j] = tnext[j]; // may run past tap H−1 on the last full load: those entries are never used
            nl = quad(J + NLO);    // new upper quad of the next lo window
            nh = quad(HQ - J - 2); // new lower quad of the next hi window
        } else {
            tn[0] = tnext[0]; // centre tap
        }
        auto LO = [&](int p) { return half(lw[(p >> 1) - J], p & 1); };
        auto HI = [&](int p) { return half(hw[(p >> 1) - (HQ - J - 1)], p & 1); };
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (u < nu) {
                const int k = k0 + u;
                if ((k & 1) == 0) {
#pragma unroll
                    for (int m = 0; m < ME; ++m) {
                        const f2 s2 = LO(k / 2 + m) + HI(H - k / 2 + m);
                        acc[2 * m] = fma_bcast(s2.x, t[u], acc[2 * m]);
                        if (m < P / 2) acc[2 * m + 1] = fma_bcast(s2.y, t[u], acc[2 * m + 1]);
                    }
                } else {
#pragma unroll
                    for (int m = 0; m < MO; ++m) {
                        const f2 s2 = LO((k - 1) / 2 + m) + HI(H - (k + 1) / 2 + m);
                        if (m >= 1) acc[2 * m - 1] = fma_bcast(s2.x, t[u], acc[2 * m - 1]);
                        if (m < P / 2 || NOUT == 9) acc[2 * m] = fma_bcast(s2.y, t[u], acc[2 * m]);
                    }
                }
            }
        }
        if (more) {
#pragma unroll
            for (int j = 0; j + 1 < NLO; ++j) lw[j] = lw[j + 1];
            lw[NLO - 1] = nl;
#pragma unroll
            for (int j = NHI - 1; j > 0; --j) hw[j] = hw[j - 1];
            hw[0] = nh;
        } else {
            // centre tap: a[o + H], the pairs E[HQ … HQ + P/2 − 1 (+1)] — inside the last hi window whatever H mod 4 is
#pragma unroll
            for (int m = 0; m < ME; ++m) {
                const f2 c2 = HI(HQ + m);
                acc[2 * m] = fma_bcast(c2.x, tn[0], acc[2 * m]);
                if (m < P / 2) acc[2 * m + 1] = fma_bcast(c2.y, tn[0], acc[2 * m + 1]);
            }
        }
#pragma unroll
        for (int o = 0; o < NOUT; ++o) pin_acc(acc[o]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Column-pass tap table for the rolling kernel (built on the host, pawsome_dog.hip): the f32
// accumulators of two adjacent outputs (slots j, j+1, j even) share one register pair and one
// v_pk_fma_f32 whose tap operand is the SGPR pair (T[t], T[t-1]).  Which t a row needs depends on
// its parity p = a & 1 (t ≡ p mod 2), so there are two pairings; T[-1] = T[l] = 0.
//   table[((qb*2 + p)*2 + c)*ROLL_QB + m] = (Tc[t], Tc[t-1]),  t = p + 2*(ROLL_QB*qb + m),
//   c = 0: s·g+, c = 1: −s·g−
constexpr int ROLL_QB = 2; // tap pairs per block (2 parities x 2 channels x QB pairs = 16 SGPRs, double-buffered)
__host__ __device__ constexpr int roll_col_blocks(int L) { return ((L + 1) / 2 + ROLL_QB - 1) / ROLL_QB; }
__host__ __device__ constexpr int roll_col_table_len(int L) { return roll_col_blocks(L) * 4 * ROLL_QB; } // in f2

// Column pass body for sub-chunk phase SC (rows a ≡ CH*SC + i mod S).  rv[i] = (R+, R−)[row i][x].
// Loop order (tap block, row, channel): every output receives its terms as t = 0: (+,−), 1: (+,−), …
// whatever its alignment to blocks and sub-chunks, so equal inputs give bit-equal outputs.
// QHI < roll_col_blocks(L): a prologue body.  The first input rows a < l−1 have no output above row 0 to feed: rows
// a … a+7 of sub-chunk sc only need taps t ≤ a+7, i.e. tap blocks 0 … 2sc+1 — the rest of the scatter would land in
// slots whose outputs do not exist (20 % of the column pass's FMAs over the first and last l−1 rows).  Up there the
// phase equals the sub-chunk number, so the shortened bodies are static instances with no branch inside.
__host__ __device__ constexpr int roll_prologue_blocks(int sc) { return 2 * sc + 2; }
// Bottom edge: rows a = 8 sc … need tap blocks qb ≥ qlo(sc) = 2 sc − c0, c0 = (n1 + 2) ÷ 4.  With the phase sc mod NBODY
// the pair (phase, qlo) of every epilogue sub-chunk is fixed by c0 mod 2·NBODY: the window's height class.
__host__ __device__ constexpr int roll_epi_c0(int n1) { return (n1 + 2) / 4; }
__host__ __device__ constexpr int roll_epi_class(int n1, int L) { return roll_epi_c0(n1) % (2 * (roll_slots(L) / ROLL_CH)); }
__host__ __device__ constexpr int roll_prologue_len(int L) { return (roll_col_blocks(L) - 2 + 1) / 2; } // sub-chunks with fewer blocks than the full body
// QLO > 0: an EPILOGUE body, blocks QLO … only: the last input rows a > n1−1 have no output below row n1−1 to feed,
// rows a … a+7 only need taps t ≥ a − (n1−1).  Down there sub-chunk and phase are tied through the window height, so
// these bodies exist per HEIGHT CLASS (roll_epi_class) — instances for the common window sizes only.
template <int L, int SC, int QHI = roll_col_blocks(L), int QLO = 0>
__device__ __forceinline__ void roll_col_body(f2 (&acc2)[roll_slots(L) / 2], const f2 (&rv)[ROLL_CH], tap_ptr table)
{
    constexpr int S = roll_slots(L), CH = ROLL_CH, QB = ROLL_QB, NQB = QHI;
    static_assert(QHI >= 1 && QHI <= roll_col_blocks(L) && QLO >= 0 && QLO < QHI, "tap-block bounds out of range");
    static_assert(S % 2 == 0 && (CH * SC) % 2 == 0, "pairing needs even slot counts");
    f2 tn[4 * QB];
    tap_ptr tb = pin_taps(table);
#pragma unroll
    for (int j = 0; j < 4 * QB; ++j) tn[j] = tb[QLO * 4 * QB + j];
#pragma unroll
    for (int qb = QLO; qb < NQB; ++qb) {
        f2 t[4 * QB];
#pragma unroll
        for (int j = 0; j < 4 * QB; ++j) t[j] = tn[j];
        if (qb + 1 < NQB) {
            tb = pin_taps(tb);
            const tap_ptr tnext = tb + (qb + 1) * 4 * QB;
#pragma unroll
            for (int j = 0; j < 4 * QB; ++j) tn[j] = tnext[j];
        }
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int par = i & 1;
            const int amod = CH * SC + i;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
#pragma unroll
                for (int m = 0; m < QB; ++m) {
                    const int tt = par + 2 * (QB * qb + m); // taps (tt, tt-1)
                    if (tt - 1 <= L - 1) {
                        const int slot = ((amod - tt) % S + S) % S; // even
                        const float r = c ? rv[i].y : rv[i].x;
                        // An even row's pair (T[0], T[−1] = 0) is the FIRST term of outputs (a, a+1) — the slots the
                        // sub-chunk S rows earlier emitted: a multiply starts them from scratch (no reset instructions;
                        // output a+1 starts at ±0 and meets its own first term, row a+1's T[0], later in this block).
                        if (tt == 0 && c == 0)
                            acc2[slot / 2] = f2{r, r} * t[(par * 2 + c) * QB + m];
                        else
                            acc2[slot / 2] = fma_bcast(r, t[(par * 2 + c) * QB + m], acc2[slot / 2]);
                    }
                }
            }
        }
        // pin exactly the pairs this block touched (pinning idle ones makes the allocator copy them)
#pragma unroll
        for (int i = 0; i < CH; ++i) {
#pragma unroll
            for (int m = 0; m < QB; ++m) {
                const int tt = (i & 1) + 2 * (QB * qb + m);
                if (tt - 1 <= L - 1) pin_acc(acc2[(((CH * SC + i - tt) % S + S) % S) / 2]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ABL: timing-only ablation bits (tools/tune): 1 = no column FMAs, 2 = no row pass, 4 = no staging
// conversion/LDS writes, 8 = no global loads.  ABL != 0 gives wrong results by design.
// One strip of one window, executed by ONE wave with wave-private LDS at `smem`: everything from the DC
// level to the wave-level peak reduction.  (best, best_idx) are valid in lane 0 on return.
// EPI ≥ 0: the instance for windows of height class EPI (roll_epi_class): statically shortened epilogue bodies too.
template <int LT, bool RESP, int ABL, int EPI = -1>
__device__ __forceinline__ void roll_strip(const LaunchGeo &g, const f2 *__restrict__ taps_row, const f2 *__restrict__ taps_col,
                                           unsigned char *smem, const uint8_t *__restrict__ frame, int g1, int g2, int s,
                                           int b, int logical, Peak &peak_out, unsigned long long &mask_out)
{
    constexpr int L = LT, hw = L / 2, S = roll_slots(L), CH = ROLL_CH, P = ROLL_P, TW = ROLL_TW;
    constexpr int NBODY = S / CH;
    constexpr int NPRO = (ABL == 0) ? roll_prologue_len(L) : 0; // sub-chunks at the top that run shortened column-pass bodies
    static_assert(S % CH == 0, "slot count must be a multiple of the sub-chunk");
    static_assert(CH == 8 && L % 4 == 1 && L >= ROLL_LMIN && L <= ROLL_LMAX, "roll kernel instance out of range");
    constexpr int SB = roll_sb(L);     // staged bytes per lane per sub-chunk (16 for l = 65)
    constexpr int SEGS = 64 / CH;      // lanes per row
    constexpr int RS = roll_rs(L);     // A row pitch in floats
    constexpr int RPASS = CH / 8;      // row-pass rounds: 8 rows × 8 groups of P = 8 outputs per round

    float *A = reinterpret_cast<float *>(smem);
    f2 *Rb = reinterpret_cast<f2 *>(smem + CH * RS * 4);

    const int lane = threadIdx.x & 63;
    // strips are 64 wide; the last one is shifted left to stay inside the window (overlap
    // recomputes a few columns bit-identically), or is partial when the window is < 64 wide
    const int ncols = g.nthin ? g.thin_x0 : g.n2; // columns covered by 64-wide strips
    const int x0 = (ncols >= TW) ? min(s * TW, ncols - TW) : 0;
    const int ws = min(TW, ncols);
    const int ti0 = g1 - g.r1 - 1 - hw;
    const int wj0 = g2 - g.r2 - 1 - hw; // frame col of the window tile's col 0
    const int tj0 = wj0 + x0;
    const int NA = g.n1 + L - 1;

    // ---- per-window DC level (see dog_kernels.hpp): same samples in every strip ----
    int dc;
    {
        int sum = dc_sample_sum(g, frame, ti0, wj0, L, lane, 64);
        sum = wave_sum(sum);
        dc = dc_from_sum(sum, g.fill);
    }

    // ---- staging geometry: lane → (row lane / SEGS, SB-byte segment lane % SEGS) ----
    const int srow = lane / SEGS, sseg = lane % SEGS;
    const int scol = tj0 + SB * sseg;            // frame col of this lane's first byte
    const bool cols_in = (scol >= 0) && (scol + SB <= g.fw);
    auto load16 = [&](int a_row, uint32_t (&w)[SB / 4]) {
        const int gi = ti0 + a_row;
        const bool rowok = (a_row < NA) && (gi >= 0) && (gi < g.fh);
        const uint32_t fill4 = (uint32_t)g.fill * 0x01010101u;
#pragma unroll
        for (int q = 0; q < SB / 4; ++q) w[q] = fill4;
        if (rowok) {
            const uint8_t *src = frame + (long long)gi * g.row_stride + scol;
            if (cols_in) {
                __builtin_memcpy(w, src, SB);
            } else {
#pragma unroll
                for (int i = 0; i < SB; ++i) {
                    const int gj = scol + i;
                    if (gj >= 0 && gj < g.fw) {
                        const uint32_t v = src[i];
                        w[i >> 2] = (w[i >> 2] & ~(0xffu << (8 * (i & 3)))) | (v << (8 * (i & 3)));
                    }
                }
            }
        }
    };

    // ---- folded remainder column (LaunchGeo::fold_r): the last strip's row pass also produces window column 64·nstrips.
    // Its last input is strip column TW + L − 1; where the 8·SB staged columns end one short of it (l = 33, 65, 97, 129)
    // the lanes of the last segment stage that one byte more, into the slack of the A row ----
    constexpr bool FOLD_OK = !RESP && ABL == 0 && roll_folds(L);
    constexpr bool FOLD_EXTRA = (8 * SB < TW + L);
    const bool fold = FOLD_OK && g.fold_r != nullptr && s == g.nstrips - 1; // wave-uniform
    const int xcol = tj0 + 8 * SB;                                          // frame col of the extra byte
    // (unconditional load at an address clamped into the frame, validity applied when the byte is used: a load inside a branch
    // is waited for where the branch ends — a full memory latency per sub-chunk in the folding waves)
    const int xcol_c = min(max(xcol, 0), g.fw - 1);
    auto load_extra = [&](int a_row) -> uint32_t {
        const int gi = min(max(ti0 + a_row, 0), g.fh - 1);
        return frame[(long long)gi * g.row_stride + xcol_c];
    };
    auto extra_ok = [&](int a_row) { const int gi = ti0 + a_row; return a_row < NA && gi >= 0 && gi < g.fh && xcol >= 0 && xcol < g.fw; };

    f2 acc2[S / 2];
#pragma unroll
    for (int j = 0; j < S / 2; ++j) acc2[j] = f2{0.f, 0.f};
    float best = -__builtin_huge_valf(), second = -__builtin_huge_valf();
    int best_y = 0;

    const tap_ptr trow = as_taps(taps_row);
    const tap_ptr tcol = as_taps(taps_col);

    unsigned long long stamp_c0 = 0, stamp_r0 = 0;
    if (ABL & 16) { stamp_c0 = __builtin_amdgcn_s_memtime(); stamp_r0 = __builtin_amdgcn_s_memrealtime(); }
    // (Tried: a fast staging path for strips wholly inside the frame — one running pointer per lane instead of load16's row
    // tests, fill pre-load and 64-bit row multiply, ≈10 VALU instructions fewer per sub-chunk: 0.4–0.6 % SLOWER in a three-way
    // same-session comparison.  The kernel does not feel a handful of VALU instructions beside its 900 packed ones.)
    const int nsub = (NA + CH - 1) / CH;
    uint32_t pre[SB / 4];
    load16(srow, pre);
    uint32_t pre_x = 0;
    if (FOLD_EXTRA && fold) pre_x = load_extra(srow);
    const int rr = lane & 7, rgx = lane >> 3; // row-pass task: row rr, output group rgx
    const long long resp_base = (long long)b * g.n1 * g.n2;

    for (int sc = 0; sc < nsub; ++sc) {
        // ---- stage this sub-chunk from the prefetched registers, request the next ----
        if (!(ABL & 4)) {
            float *dst = A + roll_row_base(srow, L) + SB * sseg; // 16-byte aligned: ds_write_b128
            const f2 ndc = f2{-(float)dc, -(float)dc};
#pragma unroll
            for (int q = 0; q < SB / 4; ++q) { // two pixels per v_pk_add_f32 (v_cvt_f32_ubyteN each): exact integers either way
                const uint32_t word = pre[q];
                const f2 v01 = f2{(float)(word & 0xffu), (float)((word >> 8) & 0xffu)} + ndc;
                const f2 v23 = f2{(float)((word >> 16) & 0xffu), (float)(word >> 24)} + ndc;
                *reinterpret_cast<f4 *>(dst + 4 * q) = f4{v01.x, v01.y, v23.x, v23.y};
            }
            if (FOLD_EXTRA && fold && sseg == SEGS - 1)
                A[roll_row_base(srow, L) + 8 * SB] = (float)(extra_ok(sc * CH + srow) ? (int)pre_x : g.fill) - (float)dc;
            if (!(ABL & 8)) {
                load16((sc + 1) * CH + srow, pre);
                if (FOLD_EXTRA && fold) pre_x = load_extra((sc + 1) * CH + srow);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier(); // the LDS traffic is wave-private: the fences' waits are all the ordering needed
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        // ---- row pass: rounds of 8 rows × 8 groups of 8 outputs ----
        if (!(ABL & 2)) {
            static_assert(RPASS == 1, "one row-pass round per sub-chunk");
            const float *arow = A + roll_row_base(rr, L) + rgx * P;
            f2 *dst = Rb + rr * ROLL_PR + rgx * P;
            if (FOLD_OK && fold) {
                f2 racc[P + 2];
#pragma unroll
                for (int o = 0; o < P + 2; ++o) racc[o] = f2{0.f, 0.f};
                roll_row_pass<L, P + 1>(racc, arow, trow);
#pragma unroll
                for (int o = 0; o < P; ++o) dst[o] = racc[o];
                // the last lane group's ninth output is window column 64·nstrips of input row sc·8 + rr
                if (rgx == TW / P - 1 && sc * CH + rr < NA) g.fold_r[(long long)b * NA + sc * CH + rr] = racc[P];
            } else {
                f2 racc[P];
#pragma unroll
                for (int o = 0; o < P; ++o) racc[o] = f2{0.f, 0.f};
                roll_row_pass<L>(racc, arow, trow);
#pragma unroll
                for (int o = 0; o < P; ++o) dst[o] = racc[o];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        // ---- column pass: 8 new R rows into the rolling accumulators ----
        f2 rv[CH];
#pragma unroll
        for (int i = 0; i < CH; ++i) rv[i] = Rb[i * ROLL_PR + lane];
        const int phase = sc % NBODY;
        const int sc_e = roll_epi_c0(g.n1) / 2 + 1; // first sub-chunk whose rows need no tap of the first block(s)
        auto emit = [&](auto SCc, auto QHIc, auto QLOc) {
            constexpr int SC = decltype(SCc)::value, QHI = decltype(QHIc)::value, QLO = decltype(QLOc)::value;
            if (!(ABL & 1)) roll_col_body<L, SC, QHI, QLO>(acc2, rv, tcol);
            // outputs y = a − (l−1) for the 8 rows of this sub-chunk are complete.  A lane owns ONE
            // column and meets its rows in increasing y (= increasing column-major index), so a strict
            // '>' keeps the first maximum of the lane (findmax, :59); ties between lanes and strips
            // are settled by index in the reductions.  Column validity is applied once at the end.
            const int ybase = sc * CH - (L - 1);
            if (ybase + CH > 0 && ybase < g.n1) {
                const bool full = (ybase >= 0) && (ybase + CH <= g.n1); // wave-uniform
#pragma unroll
                for (int i = 0; i < CH; ++i) {
                    const int slot = ((CH * SC + i - (L - 1)) % S + S) % S;
                    const float v = (slot & 1) ? acc2[slot / 2].y : acc2[slot / 2].x;
                    const bool rowok = full || ((ybase + i >= 0) && (ybase + i < g.n1));
                    if (RESP && rowok && lane < ws) g.resp[resp_base + (long long)(x0 + lane) * g.n1 + ybase + i] = v;
                    if (rowok) {
                        second = __builtin_amdgcn_fmed3f(v, best, second); // runner-up of the lane's column (exact mode)
                        if (v > best) { best = v; best_y = ybase + i; }
                    }
                }
            }
            // (no reset: the emitted slots are reused S rows later, and their first term is a multiply, see roll_col_body)
        };
        static_assert(NBODY <= 20 && NPRO < NBODY && NPRO <= 20, "extend the phase switches");
#define PDOG_FULL(k) case k: emit(std::integral_constant<int, (k) % NBODY>{}, std::integral_constant<int, roll_col_blocks(L)>{}, std::integral_constant<int, 0>{}); break;
#define PDOG_PRO(k)                                                                                                   \
    case k:                                                                                                           \
        if constexpr ((k) < NPRO) emit(std::integral_constant<int, (k) % NBODY>{}, std::integral_constant<int, ((k) < NPRO ? roll_prologue_blocks(k) : 1)>{}, std::integral_constant<int, 0>{}); \
        break;
        if (EPI >= 0 && sc >= sc_e) { // the last rows of a window of this height class: shortened bodies
            if constexpr (EPI >= 0) {
                // first epilogue sub-chunk sc_e = c0 ÷ 2 + 1: phase (EPI ÷ 2 + 1) mod NBODY (even class) …, qlo = 2 − (c0 mod 2), +2 per sub-chunk
                constexpr int PH0 = (EPI / 2 + 1) % NBODY, Q0 = 2 - (EPI & 1);
#define PDOG_EPI(e)                                                                                                          \
    case e:                                                                                                                  \
        if constexpr (Q0 + 2 * (e) < roll_col_blocks(L))                                                                     \
            emit(std::integral_constant<int, (PH0 + (e)) % NBODY>{}, std::integral_constant<int, roll_col_blocks(L)>{},      \
                 std::integral_constant<int, (Q0 + 2 * (e) < roll_col_blocks(L) ? Q0 + 2 * (e) : 0)>{});                     \
        break;
                switch (sc - sc_e) {
                    PDOG_EPI(0) PDOG_EPI(1) PDOG_EPI(2) PDOG_EPI(3) PDOG_EPI(4) PDOG_EPI(5) PDOG_EPI(6) PDOG_EPI(7)
                    PDOG_EPI(8) PDOG_EPI(9) PDOG_EPI(10) PDOG_EPI(11) PDOG_EPI(12) PDOG_EPI(13)
                default: break;
                }
#undef PDOG_EPI
            }
        } else if (sc < NPRO) { // the first rows: shortened bodies (sub-chunk = phase)
            switch (sc) {
                PDOG_PRO(0) PDOG_PRO(1) PDOG_PRO(2) PDOG_PRO(3) PDOG_PRO(4) PDOG_PRO(5) PDOG_PRO(6) PDOG_PRO(7)
                PDOG_PRO(8) PDOG_PRO(9) PDOG_PRO(10) PDOG_PRO(11) PDOG_PRO(12) PDOG_PRO(13) PDOG_PRO(14) PDOG_PRO(15)
                PDOG_PRO(16) PDOG_PRO(17) PDOG_PRO(18) PDOG_PRO(19)
            default: break;
            }
        } else {
            switch (phase) {
                PDOG_FULL(0) PDOG_FULL(1) PDOG_FULL(2) PDOG_FULL(3) PDOG_FULL(4) PDOG_FULL(5) PDOG_FULL(6) PDOG_FULL(7)
                PDOG_FULL(8) PDOG_FULL(9) PDOG_FULL(10) PDOG_FULL(11) PDOG_FULL(12) PDOG_FULL(13) PDOG_FULL(14) PDOG_FULL(15)
                PDOG_FULL(16) PDOG_FULL(17) PDOG_FULL(18)
            default: emit(std::integral_constant<int, 19 % NBODY>{}, std::integral_constant<int, roll_col_blocks(L)>{}, std::integral_constant<int, 0>{}); break;
            }
        }
#undef PDOG_FULL
#undef PDOG_PRO
        __builtin_amdgcn_wave_barrier(); // A / Rb are rewritten by the next sub-chunk
    }

    if (ABL & 16) { // diagnostic build only: shader cycles and 100 MHz ticks of the main loop, per wave
        const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        if (lane == 0 && g.resp) { g.resp[2 * logical] = (float)(c1 - stamp_c0); g.resp[2 * logical + 1] = (float)(r1 - stamp_r0); }
    }
    Peak pk;
    pk.best = best;
    pk.second = second;
    pk.idx = (x0 + lane) * g.n1 + best_y;
    if (lane >= ws) peak_init(pk);
    const float colbest = pk.best;
    peak_wave_reduce(pk);
    peak_out = pk;         // valid in lane 0
    // exact mode: which columns of the strip can hold a pixel within T of the window's maximum (≥ the strip's): the
    // refinement rescans only those
    const float sb = __shfl(pk.best, 0, 64);
    mask_out = __builtin_amdgcn_ballot_w64(colbest >= sb - g.ex.T);
}


// The strip as an out-of-line function with a register allocation of its own.  The persistent chain kernel calls it
// (inlined into that kernel's frame loop the accumulators, the row-pass windows and the loop state did not fit:
// 18–64 VGPRs went to scratch inside the hot loop, and the l = 105 instance returned wrong rows in round 1 while
// the batch kernel of the same length, compiled separately, was correct).  Arguments arrive in VGPRs by the calling
// convention: everything uniform goes back to SGPRs with readfirstlane (tap tables are read with scalar loads).
__device__ __forceinline__ unsigned long long uniform_u64(const void *p)
{
    const unsigned long long u = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
template <int LT>
__device__ __attribute__((noinline)) void roll_strip_call(const LaunchGeo *gp, const f2 *taps_row, const f2 *taps_col, unsigned char *smem,
                                                          const uint8_t *frame, int g1, int g2, int s, int b, Peak *out, unsigned long long *mask)
{
    LaunchGeo g;
    __builtin_memcpy(&g, (const void *)uniform_u64(gp), sizeof g);
    Peak pk;
    unsigned long long m;
    roll_strip<LT, false, 0>(g, (const f2 *)uniform_u64(taps_row), (const f2 *)uniform_u64(taps_col), smem, (const uint8_t *)uniform_u64(frame),
                             __builtin_amdgcn_readfirstlane(g1), __builtin_amdgcn_readfirstlane(g2), __builtin_amdgcn_readfirstlane(s),
                             __builtin_amdgcn_readfirstlane(b), 0, pk, m);
    *out = pk;
    *mask = m;
}

// One strip per workgroup (one wave).  Tried in round 3: up to four strips of ONE window per workgroup, a wave each (still no
// barrier), so that the strips of a window — which read each other's halo columns — run on one CU at one time.  The HBM
// traffic did not move (553.6 vs 557.6 MB per cfg3 step: the 1.31× over the algorithmic bytes is the fetch granularity on
// unaligned 321-byte rows, not re-fetched halos) and cfg3 ran 2 % slower (cfg4 0.6 % faster): dropped.
template <int LT, bool RESP, int ABL = 0, int EPI = -1>
__global__ __launch_bounds__(64, roll_waves(LT)) void dog_roll_kernel(const LaunchGeo g, const f2 *__restrict__ taps_row,
                                                                      const f2 *__restrict__ taps_col)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // XCD-aware block → (window, strip), see dog_kernels.hpp
    const int per_xcd = (g.nblocks + 7) >> 3;
    const int logical = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (logical >= g.nblocks) return;
    const int b = logical / g.nstrips;
    const int s = logical - b * g.nstrips;
    const int g1 = g.guesses[2 * b], g2 = g.guesses[2 * b + 1];
    const int fidx = g.frame_index ? g.frame_index[b] : b;
    const uint8_t *__restrict__ frame = g.frames + (long long)fidx * g.frame_stride;
    Peak pk;
    unsigned long long mask;
    roll_strip<LT, RESP, ABL, EPI>(g, taps_row, taps_col, smem, frame, g1, g2, s, b, logical, pk, mask);
    if (threadIdx.x == 0) {
        g.part_mask[b * g.nslots + s] = mask;
        g.part_val[b * g.nslots + s] = pk.best;
        g.part_idx[b * g.nslots + s] = pk.idx;
        g.part_sec[b * g.nslots + s] = pk.second;
    }
}

// ---- persistent serial chain (src/PawsomeTracker.jl:163-169, the intended loop :167) ----
// One workgroup per clip, one wave per strip of the search window; the workgroup walks the clip's frames,
// each frame searched around the previous frame's clamped answer, with no host round trip and no launch
// per frame.  Strips meet through 2 LDS words per wave and one workgroup barrier per frame.  Thin remainder
// columns are not used here: the last strip overlaps instead (bit-identical values either way).
struct ChainGeo {
    LaunchGeo g;                         // frames = first frame of clip 0; guesses/frame_index unused
    const RefineParams *rp;              // exact mode's constants in device memory (dog_exact.hpp); null = off
    int ref_cbw, ref_rows;               // refinement inside the kernel: window columns per block; tile rows resident at a time (the strips' LDS is its scratch)
    const f2 *taps_col_plain;            // (s·g₊[k], −s·g₋[k]) per tap: the refinement's column taps (taps_col is the roll kernel's paired table)
    const int *__restrict__ start;       // n_clips x 2, 1-based (row, col)
    int *__restrict__ out_ij;            // n_clips x n_frames x 2
    int n_frames;                        // per clip; clip c's frame k is frame c*n_frames + k
};
template <int LT>
__global__ __launch_bounds__(512) void dog_chain_kernel(const ChainGeo cg, const f2 *__restrict__ taps_row,
                                                        const f2 *__restrict__ taps_col)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ int cur[2];
    __shared__ float pv[8], ps[8];
    __shared__ int pi[8];
    __shared__ unsigned long long pm[8];
    __shared__ int s_refine;
    __shared__ float s_max, s_sec2;
    __shared__ int s_idx2;
    const LaunchGeo &g = cg.g;
    const int tid = threadIdx.x, wave = tid >> 6, nst = blockDim.x >> 6;
    const int c_ = blockIdx.x;
    if (tid == 0) {
        cur[0] = cg.start[2 * c_];
        cur[1] = cg.start[2 * c_ + 1];
        range_check(g.ex, cur[0], cur[1], LT / 2, g.fh, g.fw);
    }
    __syncthreads();
    for (int k = 0; k < cg.n_frames; ++k) {
        const int g1 = cur[0], g2 = cur[1];
        const uint8_t *__restrict__ frame = g.frames + ((long long)c_ * cg.n_frames + k) * g.frame_stride;
        Peak pk;
        unsigned long long mask;
        roll_strip_call<LT>(&g, taps_row, taps_col, smem + wave * roll_lds_bytes(LT), frame, g1, g2, wave, 0, &pk, &mask);
        if ((tid & 63) == 0) { pv[wave] = pk.best; pi[wave] = pk.idx; ps[wave] = pk.second; pm[wave] = mask; }
        __syncthreads();
        if (tid == 0) {
            Peak w;
            peak_init(w);
            for (int q = 0; q < nst; ++q) peak_merge(w, pv[q], pi[q], ps[q]);
            const int x = w.idx / g.n1, y = w.idx - x * g.n1;
            const int i = min(max(g1 - g.r1 + y, 1), g.fh);   // :60-61
            const int j = min(max(g2 - g.r2 + x, 1), g.fw);
            int *o = cg.out_ij + 2 * ((long long)c_ * cg.n_frames + k);
            o[0] = i; o[1] = j;
            cur[0] = i; cur[1] = j;
            s_refine = cg.rp && (w.best - w.second <= g.ex.T);
            s_max = w.best;
            s_sec2 = w.second;
            s_idx2 = w.idx;
            if (s_refine) atomicAdd(g.ex.stat, 1ull);
        }
        __syncthreads();
        if (s_refine) {
            // near-tie: the chain cannot go on before the reference's own arithmetic has decided (dog_exact.hpp);
            // the strips' LDS is free between frames and is the refinement's scratch
            RefineCtx c;
            c.trow = as_taps(taps_row);
            c.tcol = as_taps(cg.taps_col_plain);
            const refine_params_ptr rp = (refine_params_ptr)(unsigned long long)cg.rp;
            c.K = (k64_ptr)(unsigned long long)rp->K64;
            c.g64 = (k64_ptr)(unsigned long long)rp->g64;
            c.dir = rp->dir;
            c.T64 = rp->T64;
            c.T = g.ex.T;
            c.T_rescan = g.ex.T_rescan;
            c.second = s_sec2;
            c.fp32_idx = s_idx2;
            c.cbw = cg.ref_cbw;
            c.tile_rows = cg.ref_rows;
            c.lds = smem;
            auto may = [&](int x0, int x1, float thr) { // strip s covers 64 columns from min(64 s, n2 − 64) (one partial strip when n2 < 64)
                bool any = false;
                for (int q = 0; q < nst; ++q) {
                    const int lo = g.n2 >= ROLL_TW ? min(q * ROLL_TW, g.n2 - ROLL_TW) : 0, hi = lo + ROLL_TW;
                    if (lo < x1 && hi > x0 && pv[q] >= thr && (pm[q] & column_bits(x0 - lo, x1 - lo))) any = true;
                }
                return any;
            };
            const int idx = refine_window<8>(blockDim.x, g, frame, g1, g2, s_max, c, may);
            if (tid == 0) {
                const int x = idx / g.n1, y = idx - x * g.n1;
                const int i = min(max(g1 - g.r1 + y, 1), g.fh);
                const int j = min(max(g2 - g.r2 + x, 1), g.fw);
                int *o = cg.out_ij + 2 * ((long long)c_ * cg.n_frames + k);
                o[0] = i; o[1] = j;
                cur[0] = i; cur[1] = j;
            }
            __syncthreads();
        }
    }
}

#ifndef PDOG_ROLL_INST_ONLY
// Step of a multi-clip chain run as ordinary batches: file the step's answers under [clip][frame] and make
// them the next step's guesses.
static __global__ void dog_chain_step_kernel(const int *__restrict__ step_ij, int *__restrict__ cur, int *__restrict__ out_ij,
                                      int n_clips, int n_frames, int k)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_clips) return;
    const int i = step_ij[2 * c], j = step_ij[2 * c + 1];
    out_ij[2 * ((long long)c * n_frames + k)] = i;
    out_ij[2 * ((long long)c * n_frames + k) + 1] = j;
    cur[2 * c] = i;
    cur[2 * c + 1] = j;
}

#endif // PDOG_ROLL_INST_ONLY

// dynamic LDS of dog_thin_kernel: the R column (f2 per input row) and the column's input patch as bytes
__host__ __device__ constexpr size_t thin_lds_bytes(int n1, int L) { return ((size_t)(n1 + L - 1) * sizeof(f2) + 15) / 16 * 16 + (size_t)(n1 + L - 1) * ((L + 3) / 4 * 4); }

// ---- thin remainder ----
// A window whose width is 64·k + r with small r (257 = 4·64 + 1) would need a whole extra strip for
// r columns.  Instead those columns are done here, one 256-thread workgroup per (window, column):
// thread = input row for the row pass (l contiguous pixels, symmetric taps), then thread = output row
// for the column pass over the R column in LDS.  The arithmetic replays dog_roll_kernel's operation
// order exactly (row: k ascending pairs then centre, both Gaussians per v_pk_fma_f32; column: per tap
// t ascending, the g+ term then the g− term into one f32), so a pixel computed here is bit-identical
// to what a strip would have produced and flat windows still tie exactly.
template <int LT, bool RESP>
__global__ __launch_bounds__(256) void dog_thin_kernel(const LaunchGeo g, const f2 *__restrict__ taps_row,
                                                       const f2 *__restrict__ taps_col)
{
    constexpr int L = LT, hw = L / 2, H = L / 2, NT = 256, NW = NT / 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    f2 *Rc = reinterpret_cast<f2 *>(smem);            // NA entries
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // XCD-aware block → (window, column) like the strips (dog_kernels.hpp): the kernel runs BESIDE the roll kernel, and a
    // window's remainder column then meets its strips in one L2 — the patch it reads is the last strip's
    const int nb = g.n * g.nthin, per_xcd = (nb + 7) >> 3;
    const int logical = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (logical >= nb) return;
    const int b = logical / g.nthin;
    const int rc = logical - b * g.nthin;
    const int x = g.thin_x0 + rc;                     // window column
    const int g1 = g.guesses[2 * b], g2 = g.guesses[2 * b + 1];
    const int fidx = g.frame_index ? g.frame_index[b] : b;
    const uint8_t *__restrict__ frame = g.frames + (long long)fidx * g.frame_stride;
    const int ti0 = g1 - g.r1 - 1 - hw;
    const int wj0 = g2 - g.r2 - 1 - hw;
    const int NA = g.n1 + L - 1;
    const tap_ptr trow = as_taps(taps_row);
    const tap_ptr tcol = as_taps(taps_col);

    __shared__ int ssum[NW];
    __shared__ float sval[NW], ssec[NW];
    __shared__ int sidx[NW];
    int dc;
    {
        int sum = dc_sample_sum(g, frame, ti0, wj0, L, tid, NT);
        sum = wave_sum(sum);
        if (lane == 0) ssum[wave] = sum;
        __syncthreads();
        int tot = 0;
        for (int w = 0; w < NW; ++w) tot += ssum[w];
        dc = dc_from_sum(tot, g.fill);
    }
    const float fdc = (float)dc;

    // ---- the column's input patch, NA rows × l pixels, → LDS bytes: a dword per item, loaded unconditionally at an
    // address clamped into the frame (a clamped dword still holds every in-frame byte its item needs, at a shifted
    // position), PaddedView fill (:48) selected afterwards — coalesced along the rows, no per-pixel branches (the old
    // row pass read its own row straight from memory, 64 cache lines per load instruction, and its border path spilled
    // 360 SGPRs) ----
    constexpr int TP = (L + 3) / 4 * 4, TQ = TP / 4;  // tile pitch in bytes / dwords
    uint8_t *tile = smem + ((size_t)NA * sizeof(f2) + 15) / 16 * 16;
    const int gj0 = wj0 + x; // frame col of input k = 0
    if (ti0 >= 0 && ti0 + NA <= g.fh && gj0 >= 0 && gj0 + TP <= g.fw) { // the whole patch inside the frame (workgroup-uniform): a plain copy
#pragma unroll 4
        for (int e = tid; e < NA * TQ; e += NT) {
            const int a = e / TQ, q = e - a * TQ;
            uint32_t w;
            __builtin_memcpy(&w, frame + (long long)(ti0 + a) * g.row_stride + gj0 + 4 * q, 4);
            *reinterpret_cast<uint32_t *>(tile + a * TP + 4 * q) = w;
        }
    } else if (g.fw >= 4) {
#pragma unroll 4
        for (int e = tid; e < NA * TQ; e += NT) {
            const int a = e / TQ, q = e - a * TQ;
            const int gi = ti0 + a, gj = gj0 + 4 * q;
            const int gjc = min(max(gj, 0), g.fw - 4);
            uint32_t w;
            __builtin_memcpy(&w, frame + (long long)min(max(gi, 0), g.fh - 1) * g.row_stride + gjc, 4);
            const bool rowok = gi >= 0 && gi < g.fh;
            uint32_t o = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int gjj = gj + i;
                const uint32_t px = (rowok && gjj >= 0 && gjj < g.fw) ? ((w >> (8 * ((gjj - gjc) & 3))) & 0xffu) : (uint32_t)g.fill;
                o |= px << (8 * i);
            }
            *reinterpret_cast<uint32_t *>(tile + a * TP + 4 * q) = o;
        }
    } else {
        for (int e = tid; e < NA * TP; e += NT) {
            const int a = e / TP, cc = e - a * TP;
            const int gi = ti0 + a, gj = gj0 + cc;
            tile[e] = (gi >= 0 && gi < g.fh && gj >= 0 && gj < g.fw) ? frame[(long long)gi * g.row_stride + gj] : (uint8_t)g.fill;
        }
    }
    __syncthreads();
    // ---- row pass: R[a] for input rows a = tid, tid + 256, … (same operation order as the strips: pairs k ascending, centre last) ----
    for (int a = tid; a < NA; a += NT) {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(tile + a * TP);
        float v[L];
#pragma unroll
        for (int q = 0; q < TQ; ++q) {
            const uint32_t w = src[q];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (4 * q + i < L) v[4 * q + i] = (float)((w >> (8 * i)) & 0xffu) - fdc;
        }
        f2 acc = f2{0.f, 0.f};
        tap_ptr tb = trow;
#pragma unroll
        for (int k0 = 0; k0 < H; k0 += 8) { // taps in blocks of 8 behind a base pointer re-pinned per block (not hoisted out of the row loop)
            tb = pin_taps(tb);
#pragma unroll
            for (int k = k0; k < k0 + 8; ++k)
                if (k < H) acc = fma_bcast(v[k] + v[L - 1 - k], tb[k], acc);
        }
        tb = pin_taps(tb);
        acc = fma_bcast(v[H], tb[H], acc);
        Rc[a] = acc;
    }
    __syncthreads();
    // ---- column pass + argmax: outputs y = tid, tid + 256, … ----
    Peak pk;
    peak_init(pk);
    for (int y = tid; y < g.n1; y += NT) {
        float acc = 0.f;
#pragma unroll 13
        for (int t = 0; t < L; ++t) {
            const f2 r = Rc[y + t];
            const f2 w = tcol[t];
            acc = __builtin_fmaf(r.x, w.x, acc);
            acc = __builtin_fmaf(r.y, w.y, acc);
        }
        const int lin = x * g.n1 + y;
        if (RESP) g.resp[(long long)b * g.n1 * g.n2 + lin] = acc;
        peak_push(pk, acc, lin);
    }
    peak_wave_reduce(pk);
    if (lane == 0) { sval[wave] = pk.best; sidx[wave] = pk.idx; ssec[wave] = pk.second; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < NW; ++w) peak_merge(pk, sval[w], sidx[w], ssec[w]);
        g.part_val[b * g.nslots + g.nstrips + rc] = pk.best;
        g.part_idx[b * g.nslots + g.nstrips + rc] = pk.idx;
        g.part_sec[b * g.nslots + g.nstrips + rc] = pk.second;
    }
}

} // namespace pdog
