// dog_tiled.hpp — DoG + argmax for ONE large search window per frame (e.g. 257×257 around a 25-px target), the latency
// path of a single clip: the window is cut into sub-windows that each fit one CU's LDS, one 1024-thread workgroup per
// sub-window runs the whole pipeline of the fused kernel (dog_fused.hpp) on its piece, and the workgroup that delivers
// the last partial peak combines them (reference functor /root/reference/src/PawsomeTracker.jl:55-62, frame loop :163-169).
//
// Why: the two-pass kernels spread such a window over ≈50 workgroups but need the row-pass result to cross HBM and
// three stream-ordered launches per frame (row pass, column pass, finish: ≈27 µs per 257×257 frame, ≈50 µs per host
// functor call); one cooperative launch with grid barriers between the passes was slower still (dog_coop.hpp: a
// frame is ≈12 dependent memory round trips).  Here a frame is: samples + tile (one round trip) → both passes in LDS →
// one partial per sub-window → combine.  The price is the halo: each sub-window runs its row pass over l − 1 extra
// tile rows (3× the row-pass work at 32×32 sub-windows, l = 65) — irrelevant where latency, not throughput, is what a
// single clip sees.  Measured (MI355X, 1080p, l = 65): 257×257 chain 13.8 µs per frame (27.5 µs with the launches).
//
// Arithmetic per output: exactly the fused kernel's (same tasks, same tap order), with the DC level taken from the
// FULL window's sample grid, so every sub-window subtracts the same level and flat regions tie exactly across
// sub-window borders; indices are the full window's column-major indices, ties → the smallest (findmax, :59).
//
// Who combines.  Independent windows (chain_len = 1; an ordinary launch, any number of workgroups): the workgroup whose
// partial arrives LAST (an arrival counter; the counter and the frame flag are zero when a launch ends, as they were when it
// started).  Clips (chain_len > 1; a cooperative launch — every workgroup of a clip must be resident, guaranteed or refused):
// every workgroup polls the frame's partials — self-validating tagged words, no counter — and combines for itself, see the
// frame loop; sub-window 0's workgroup writes the answer out, refines and publishes.
#pragma once
#include "dog_fused.hpp"

namespace pdog {

struct TiledGeo {
    LaunchGeo g;            // full-window geometry; part_val / part_idx / part_sec: [n_clips][2][nsub] (two sets, by frame parity)
    int NA, TWin;           // the FULL window's padded tile (for the DC sample grid)
    int sn1, sn2;           // sub-window rows / columns (the last ones are smaller)
    int ns1, ns2;           // sub-windows per window column / row; nsub = ns1·ns2, sub-window s = s2·ns1 + s1
    int pitchA, pitchV;     // LDS pitches for the largest sub-window (fused_pitch_a/v)
    int cshift, pr, pc;     // staging threads per tile row (2^cshift), outputs per task (see dog_fused.hpp)
    int chain_len;          // frames per clip; clip c = frames c·chain_len …
    int32_t *out_ij;        // [n_clips][chain_len][2]
    int32_t *done_flag;     // NULL, or host-coherent word: done_value after clip 0's last frame, or k + 1 after each frame (progress)
    int32_t done_value;
    int progress;
    const RefineParams *rp; // exact mode (null = off)
    int ref_cbw, ref_rows;  // refinement scratch geometry inside this kernel's LDS
    int dc_host;            // ≥ 0: the window's DC level from the host (functor: see dog_fused.hpp); −1: sampled here
    int *cur;               // [n_clips][2]: the current guess, last arrival → everyone
    unsigned *sync;         // [n_clips][2], zero when a launch starts and when it ends: partial arrivals, frame flag (set after a refined frame only)
    unsigned *abort;        // one word, zero unless a device-side wait of this tracker gave up (wait_counter): every wait polls it
    unsigned long long *slots; // [n_clips][3][nsub][2]: chains — the sub-windows' partials as two self-validating 64-bit words each, two sets by
                               // frame parity; a third set carries each sub-window's max |pixel − dc| on flagged frames (see the frame loop)
    unsigned tag_base;      // chains: frame k's partials carry the tag tag_base + k + 1; the host advances it by chain_len + 1 per launch
    int fault_inject;       // tests: sub-window 0 of clip 0 never delivers the partial of its second frame (pdog_set_tuning "fault_inject")
};

constexpr int TILED_SLOT_CAP = 256; // sub-windows per window the combining wave handles (4 per lane)

// LT > 0: the compile-time-l tasks and tile layout of dog_fused.hpp (fusedc_*), same values bit for bit.
template <bool RESP, int LT = 0>
__global__ __launch_bounds__(FUSED_NT) void dog_tiled_kernel(const TiledGeo tg, const f2 *__restrict__ taps_row,
                                                             const f2 *__restrict__ taps_col)
{
    const LaunchGeo &g = tg.g;
    constexpr int NT = FUSED_NT, NW = NT / 64, U = FUSED_U;
    const int L = LT ? LT : g.L, hw = L >> 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *A = reinterpret_cast<float *>(smem);
    f2 *Vs = reinterpret_cast<f2 *>(smem + (LT ? fusedc_a_bytes(tg.sn1, tg.sn2, L) : fused_a_bytes(tg.sn1, tg.sn2, L)));
    __shared__ int s_sum[NW];
    __shared__ float s_val[NW], s_sec[NW];
    __shared__ int s_idx[NW];
    __shared__ int s_last, s_refine, s_abort;
    __shared__ float s_max, s_sec2;
    __shared__ int s_idx2;
    __shared__ float s_pv[TILED_SLOT_CAP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nsub = tg.ns1 * tg.ns2;
    const int clip = blockIdx.x / nsub, s = blockIdx.x - clip * nsub;
    const int s2 = s / tg.ns1, s1 = s - s2 * tg.ns1;
    const int sy = s1 * tg.sn1, sx = s2 * tg.sn2;                   // this sub-window's first row / column in the window
    const int m1 = min(tg.sn1, g.n1 - sy), m2 = min(tg.sn2, g.n2 - sx); // its size
    const int NAs = m1 + L - 1, TWs = m2 + L - 1;                   // its padded tile
    const tap_ptr trow = as_taps(taps_row), tcol = as_taps(taps_col);
    float *const pv = g.part_val + (long long)clip * 2 * nsub;
    int *const pi = g.part_idx + (long long)clip * 2 * nsub;
    float *const ps = g.part_sec + (long long)clip * 2 * nsub;
    unsigned *const arrive = tg.sync + 2 * clip, *const flag = arrive + 1;
    int *const cur = tg.cur + 2 * clip;

    // tile columns ≥ TWs and RT columns ≥ NAs are only read by the sliding windows of masked outputs: zero once (and after a refinement)
    auto zero_padding = [&]() {
        if (LT) return; // (the compile-time-l tasks read unstaged entries only into masked outputs)
        for (int r = wave; r < NAs; r += NW)
            for (int c = TWs + lane; c < tg.pitchA; c += 64) A[r * tg.pitchA + c] = 0.f;
        for (int x = wave; x < m2; x += NW)
            for (int c = NAs + lane; c < tg.pitchV; c += 64) Vs[x * tg.pitchV + c] = f2{0.f, 0.f};
    };
    if (tid == 0) s_abort = 0;
    zero_padding();

    // wave 0 of a clip's workgroup: poll a set of tagged slots (a lane per slot) until every sub-window's words carry `tag`.  Bounded like
    // wait_counter (dog_kernels.hpp): gives up when a peer has, or after 1 s of wall time, raising the abort and fault words.
    constexpr int SPL = TILED_SLOT_CAP / 64; // slots per lane
    auto poll_slots = [&](const unsigned long long *sl2, unsigned tag, unsigned long long (&a)[SPL], unsigned long long (&bq)[SPL]) -> int {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        for (unsigned spins = 1;; ++spins) {
            bool ok = true;
#pragma unroll
            for (int j = 0; j < SPL; ++j) {
                const int q = lane + 64 * j;
                a[j] = 0;
                bq[j] = 0;
                if (q < nsub) {
                    a[j] = __hip_atomic_load(&sl2[2 * q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    bq[j] = __hip_atomic_load(&sl2[2 * q + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = ok && (unsigned)(bq[j] >> 32) == tag && (unsigned)(a[j] >> 56) == (tag & 0xffu);
                }
            }
            if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) return 0;
            __builtin_amdgcn_s_sleep(1);
            if ((spins & 63u) == 0u) {
                if (__hip_atomic_load(tg.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return 1;
                if (__builtin_amdgcn_s_memrealtime() - t0 > WAIT_TICKS) {
                    __hip_atomic_store(tg.abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (g.ex.range_err) __hip_atomic_store(g.ex.range_err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    return 1;
                }
            }
        }
    };

    int g1 = g.guesses[2 * clip], g2 = g.guesses[2 * clip + 1];
    for (int k = 0; k < tg.chain_len; ++k) {
        const long long fidx = tg.chain_len > 1 ? (long long)clip * tg.chain_len + k : (g.frame_index ? g.frame_index[clip] : clip);
        const uint8_t *__restrict__ frame = g.frames + fidx * g.frame_stride;
        const int wi0 = g1 - g.r1 - 1 - hw, wj0 = g2 - g.r2 - 1 - hw; // the full window's tile origin
        const int ti0 = wi0 + sy, tj0 = wj0 + sx;                     // this sub-window's
        // ---- DC level of the FULL window (the fused kernel's 32×32 sample grid) and staging of the sub-window's tile: the
        // thread's sample and its first SU dwords are requested together (see dog_fused.hpp) ----
        {
            constexpr int SU = 4;
            const int q = tid & ((1 << tg.cshift) - 1), sr0 = tid >> tg.cshift, srstep = NT >> tg.cshift;
            const int c0 = 4 * q, gj0 = tj0 + c0, gj0c = min(max(gj0, 0), g.fw - 4);
            const uint8_t *colp = frame + gj0c;
            auto load_batch = [&](int r0, uint32_t (&v)[SU]) {
#pragma unroll
                for (int u = 0; u < SU; ++u) {
                    const int gi = ti0 + r0 + u * srstep;
                    __builtin_memcpy(&v[u], colp + (long long)min(max(gi, 0), g.fh - 1) * g.row_stride, 4);
                }
            };
            const bool host_dc = tg.dc_host >= 0 && tg.chain_len == 1;
            int samp = 0;
            if (!host_dc) {
                const int si = wi0 + (int)(((long long)(tid >> 5) * tg.NA) >> 5), sj = wj0 + (int)(((long long)(tid & 31) * tg.TWin) >> 5);
                samp = frame[(long long)min(max(si, 0), g.fh - 1) * g.row_stride + min(max(sj, 0), g.fw - 1)];
                if (!(si >= 0 && si < g.fh && sj >= 0 && sj < g.fw)) samp = g.fill;
            }
            uint32_t v[SU];
            load_batch(sr0, v);
            int dc = tg.dc_host;
            if (!host_dc) {
                samp = wave_sum(samp);
                if (lane == 0) s_sum[wave] = samp;
                __syncthreads();
                int total = 0;
#pragma unroll
                for (int w = 0; w < NW; ++w) total += s_sum[w];
                dc = dc_from_sum(total, g.fill);
            }
            const bool colin = gj0 >= 0 && gj0 + 4 <= g.fw;
            const float fdc = (float)dc;
            for (int r0 = sr0; r0 < NAs; r0 += srstep * SU) {
                if (r0 != sr0) load_batch(r0, v);
#pragma unroll
                for (int u = 0; u < SU; ++u) {
                    const int r = r0 + u * srstep, gi = ti0 + r;
                    const bool rowok = gi >= 0 && gi < g.fh;
                    if (LT) { // four pixels → one 16-byte store (dog_fused.hpp)
                        if (r < NAs && c0 < TWs) {
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
                            *reinterpret_cast<f4 *>(A + fusedc_row_base(r, tg.pitchA) + c0) = px - fdc;
                        }
                        continue;
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int gj = gj0 + i;
                        const int px = (rowok && gj >= 0 && gj < g.fw) ? (int)((v[u] >> (8 * ((gj - gj0c) & 3))) & 0xffu) : g.fill;
                        if (c0 + i < TWs && r < NAs) A[r * tg.pitchA + c0 + i] = (float)(px - dc);
                    }
                }
            }
        }
        __syncthreads();
        // ---- row pass → RT[x][a] ----
        if constexpr (LT > 0) {
            // task = (tile row a, group of PR = 8 or 4 outputs), dense over the threads; the division by the group count is a float
            // multiply (task + ½ never comes within 1/(2·ngx) of a multiple of ngx: exact for every task count that fits LDS).
            // PR is the host's choice (fusedc_row_outputs): 4 where tasks of 8 would leave most waves without one.
            auto run = [&](auto Pc) {
                constexpr int PR = decltype(Pc)::value;
                const int ngx = (m2 + PR - 1) / PR, ntask = NAs * ngx;
                const float inv = 1.0f / (float)ngx;
                for (int task = tid; task < ntask; task += NT) {
                    const int a = (int)(((float)task + 0.5f) * inv), gx = task - a * ngx, xb = PR * gx;
                    f2 acc[PR];
#pragma unroll
                    for (int o = 0; o < PR; ++o) acc[o] = f2{0.f, 0.f};
                    roll_row_pass<LT, PR>(acc, A + fusedc_row_base(a, tg.pitchA) + xb, trow);
#pragma unroll
                    for (int o = 0; o < PR; ++o)
                        if (xb + o < m2) Vs[(xb + o) * tg.pitchV + a] = acc[o];
                }
            };
            if (tg.pr == 4) run(std::integral_constant<int, 4>{});
            else run(std::integral_constant<int, 8>{});
        } else {
            const int ngx = (m2 + tg.pr - 1) / tg.pr, ntask = NAs * ngx;
            auto run = [&](auto Pc) {
                constexpr int PR = decltype(Pc)::value;
                for (int task = tid; task < ntask; task += NT) {
                    const int gx = task / NAs, a = task - gx * NAs, xb = gx * PR;
                    f2 acc[PR];
                    fused_row_task<PR, U>(A + a * tg.pitchA + xb, L, trow, acc);
#pragma unroll
                    for (int o = 0; o < PR; ++o)
                        if (xb + o < m2) Vs[(xb + o) * tg.pitchV + a] = acc[o];
                }
            };
            switch (tg.pr) {
            case 3: run(std::integral_constant<int, 3>{}); break;
            case 4: run(std::integral_constant<int, 4>{}); break;
            case 5: run(std::integral_constant<int, 5>{}); break;
            case 6: run(std::integral_constant<int, 6>{}); break;
            default: run(std::integral_constant<int, 8>{}); break;
            }
        }
        __syncthreads();
        // ---- column pass + peak (indices: the full window's column-major indices) ----
        Peak pk;
        peak_init(pk);
        {
            const int ngy = (m1 + tg.pc - 1) / tg.pc, ntask = m2 * ngy;
            auto run = [&](auto Pc) {
                constexpr int PC = decltype(Pc)::value;
                const float inv = 1.0f / (float)m2;
                for (int task = tid; task < ntask; task += NT) {
                    const int gy = LT ? (int)(((float)task + 0.5f) * inv) : task / m2, x = task - gy * m2, yb = gy * PC;
                    f2 acc[PC];
                    if constexpr (LT > 0)
                        fusedc_col_task<LT, PC>(Vs + x * tg.pitchV + yb, tcol, acc);
                    else
                        fused_col_task<PC, U>(Vs + x * tg.pitchV + yb, L, tcol, acc);
#pragma unroll
                    for (int o = 0; o < PC; ++o) {
                        const int y = yb + o;
                        if (y < m1) {
                            const float v = acc[o].x + acc[o].y;
                            const int lin = (sx + x) * g.n1 + sy + y;
                            if (RESP) g.resp[(long long)clip * g.n1 * g.n2 + lin] = v;
                            peak_push(pk, v, lin);
                        }
                    }
                }
            };
            switch (tg.pc) {
            case 2: run(std::integral_constant<int, 2>{}); break;
            case 3: run(std::integral_constant<int, 3>{}); break;
            case 4: run(std::integral_constant<int, 4>{}); break;
            case 6: run(std::integral_constant<int, 6>{}); break;
            default: run(std::integral_constant<int, 8>{}); break;
            }
        }
        peak_wave_reduce(pk);
        if (lane == 0) { s_val[wave] = pk.best; s_idx[wave] = pk.idx; s_sec[wave] = pk.second; }
        __syncthreads();
        // ---- this sub-window's partial → memory (two sets of slots, by frame parity: a workgroup can be at most one
        // frame ahead of the slowest reader).  Independent windows: the LAST arrival combines them.  Clips: EVERY
        // workgroup polls the partials and combines for itself — the same values in the same order give the same answer
        // everywhere; only sub-window 0's workgroup writes the answer out, and only a refinement (rare) goes through the frame flag. ----
        const bool chain = k + 1 < tg.chain_len; // (the clip's last frame has no successor to wait for: the last arrival alone combines it)
        const int par = (k & 1) * nsub;
        const bool publish = tg.done_flag && clip == 0 && (tg.progress || k == tg.chain_len - 1);
        int32_t *const o_ij = tg.out_ij + 2 * ((long long)clip * tg.chain_len + k);
        if (wave == 0) {
            peak_init(pk);
            if (lane < NW) { pk.best = s_val[lane]; pk.idx = s_idx[lane]; pk.second = s_sec[lane]; }
            peak_wave_reduce(pk, NW);
            int last = 0;
            // Round 3, clips: ONE memory round trip per frame.  A partial is two 64-bit words that validate themselves — (best | index and
            // the tag's low byte) and (runner-up | tag), the tag unique per frame and launch — stored by the sub-window's workgroup and
            // POLLED by every workgroup's wave 0 (a lane per slot) until all of the frame's are there.  No arrival counter, no second
            // trip to fetch what the counter announced: 11.2 → ≈9.5 µs per 257×257 frame.  Sub-window 0's workgroup plays the part the
            // last arrival plays for independent windows (writes the answer out, refines, publishes).  Two sets of slots by frame
            // parity as before: to leave frame k + 1 a workgroup needs everybody's partial of k + 1, so nobody still reads frame k's.
            Peak w;
            peak_init(w);
            bool combined = false;
            if (tg.chain_len > 1) {
                unsigned long long *const sl2 = tg.slots + ((size_t)clip * 3 + (k & 1)) * (size_t)nsub * 2;
                const unsigned tag = tg.tag_base + (unsigned)k + 1u;
                const bool skip = tg.fault_inject && clip == 0 && s == 0 && k == 1; // (tests: a peer that never delivers)
                if (lane == 0 && !skip) {
                    const unsigned long long w0 = (unsigned long long)__builtin_bit_cast(unsigned, pk.best) |
                                                  ((unsigned long long)((unsigned)pk.idx | ((tag & 0xffu) << 24)) << 32);
                    const unsigned long long w1 = (unsigned long long)__builtin_bit_cast(unsigned, pk.second) | ((unsigned long long)tag << 32);
                    __hip_atomic_store(&sl2[2 * s], w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&sl2[2 * s + 1], w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                last = (s == 0);
                if (chain || last) { // (the clip's last frame has no successor: only sub-window 0's workgroup combines it)
                    unsigned long long a[SPL], bq[SPL];
                    const int gave_up = poll_slots(sl2, tag, a, bq);
                    if (gave_up) {
                        if (lane == 0) s_abort = 1;
                    } else {
#pragma unroll
                        for (int j = 0; j < SPL; ++j) {
                            const int q = lane + 64 * j;
                            if (q < nsub) {
                                const float v = __builtin_bit_cast(float, (unsigned)a[j]);
                                s_pv[q] = v;
                                peak_merge(w, v, (int)((unsigned)(a[j] >> 32) & 0xffffffu), __builtin_bit_cast(float, (unsigned)bq[j]));
                            }
                        }
                        combined = true;
                    }
                }
            } else
            if (lane == 0) {
                __hip_atomic_store(&pv[par + s], pk.best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&pi[par + s], pk.idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&ps[par + s], pk.second, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned target = (unsigned)(k + 1) * (unsigned)nsub;
                const bool skip = tg.fault_inject && clip == 0 && s == 0 && k == 1; // (tests: a peer that never arrives)
                const unsigned old = skip ? 0u : __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
                last = !skip && (old == target - 1u);
                if ((chain && !last) || skip)
                    if (!wait_counter(arrive, skip ? ~0u : target, g.ex, tg.abort)) s_abort = 1; // gave up, or a peer did: leave the frame loop
            }
            last = __shfl(last, 0, 64);
            const int aborted = __shfl(lane == 0 ? s_abort : 0, 0, 64);
            if (lane == 0) { s_last = last; s_refine = 0; }
            if ((last || chain) && !aborted) {
                if (!combined) // independent windows: the last arrival fetches what the counter announced
                    for (int sl = lane; sl < nsub; sl += 64) {
                        const float v = __hip_atomic_load(&pv[par + sl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (sl < TILED_SLOT_CAP) s_pv[sl] = v;
                        peak_merge(w, v, __hip_atomic_load(&pi[par + sl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                                   __hip_atomic_load(&ps[par + sl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                    }
                peak_wave_reduce(w);
                if (lane == 0) {
                    const int x = w.idx / g.n1, y = w.idx - x * g.n1;
                    const int i = min(max(g1 - g.r1 + y, 1), g.fh);   // :60-61
                    const int j = min(max(g2 - g.r2 + x, 1), g.fw);
                    const bool rf = tg.rp && (w.best - w.second <= g.ex.T);
                    s_refine = rf;
                    s_max = w.best;
                    s_sec2 = w.second;
                    s_idx2 = w.idx;
                    s_idx[0] = i; // (the wave peaks have been consumed: the next guess travels through their slots)
                    s_idx[1] = j;
                    if (last) {
                        if (k == 0) range_check(g.ex, g1, g2, hw, g.fh, g.fw);
                        if (k == tg.chain_len - 1) { // nobody looks at the arrival count or the frame flag any more: zero for the next launch
                            if (tg.chain_len == 1) __hip_atomic_store(arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // (clips never count arrivals)
                            __hip_atomic_store(flag, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                        if (rf) {
                            if (!(chain && g.ex.T < __builtin_huge_valf())) atomicAdd(g.ex.stat, 1ull); // (clips' frames: counted once the window's own V has confirmed the flag)
                        } else {
                            o_ij[0] = i;
                            o_ij[1] = j;
                            if (publish) {
                                __threadfence_system();
                                __hip_atomic_store(tg.done_flag, tg.progress ? k + 1 : tg.done_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                            }
                        }
                    }
                }
            }
        }
        __syncthreads();
        if (s_abort) break; // a device-side wait gave up: nothing further is written or published; pdog_sync reports PDOG_E_HIP
        // A gap within the V = 255 bound on a clip's frame.  The bound is proportional to the window's own V = max |pixel − dc| (dog_exact.hpp) and
        // every workgroup still holds its sub-window's tile in LDS as exactly those differences: each takes its tile's maximum, the maxima
        // go round like the partials (third slot set, same tag), and everybody reaches the same verdict — for a frame of ±2-level noise
        // almost always "the FP32 answer stands" — in ≈2 µs instead of the ≈30 µs of a refinement that finds the same out from memory.
        int vknown = -1;
        bool refine = s_refine != 0;
        if (refine && chain && g.ex.T < __builtin_huge_valf()) {
            int vbits = 0;
            for (int r0 = wave; r0 < NAs; r0 += 4 * NW)
                for (int cb = 0; cb < TWs; cb += 256) {
                    int rd[4][4];
#pragma unroll
                    for (int kr = 0; kr < 4; ++kr) {
                        const int r = min(r0 + kr * NW, NAs - 1);
                        const int *row = reinterpret_cast<const int *>(A + (LT ? fusedc_row_base(r, tg.pitchA) : r * tg.pitchA));
#pragma unroll
                        for (int j = 0; j < 4; ++j) rd[kr][j] = row[min(cb + lane + 64 * j, TWs - 1)];
                    }
#pragma unroll
                    for (int kr = 0; kr < 4; ++kr)
#pragma unroll
                        for (int j = 0; j < 4; ++j) vbits = max(vbits, rd[kr][j] & 0x7fffffff);
                }
            const float vw = wave_max(__builtin_bit_cast(float, vbits));
            if (lane == 0) s_val[wave] = vw; // (the wave peaks have been consumed)
            __syncthreads();
            if (wave == 0) {
                float vm = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) vm = fmaxf(vm, s_val[w]);
                unsigned long long *const sl3 = tg.slots + ((size_t)clip * 3 + 2) * (size_t)nsub * 2;
                const unsigned tag = tg.tag_base + (unsigned)k + 1u;
                if (lane == 0) {
                    __hip_atomic_store(&sl3[2 * s], (unsigned long long)__builtin_bit_cast(unsigned, vm) | ((unsigned long long)((tag & 0xffu) << 24) << 32),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&sl3[2 * s + 1], (unsigned long long)tag << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                unsigned long long a[SPL], bq[SPL];
                if (poll_slots(sl3, tag, a, bq)) {
                    if (lane == 0) s_abort = 1;
                } else {
                    float vall = 0.f;
#pragma unroll
                    for (int j = 0; j < SPL; ++j)
                        if (lane + 64 * j < nsub) vall = fmaxf(vall, __builtin_bit_cast(float, (unsigned)a[j]));
                    vall = wave_max(vall);
                    if (lane == 0) s_sec[0] = vall;
                }
            }
            __syncthreads();
            if (s_abort) break;
            const float vm = s_sec[0];
            vknown = (int)vm;
            refine = s_max - s_sec2 <= g.ex.T * (vm * (1.0f / 255.0f)) * 1.00001f;
            if (s_last && tid == 0) {
                if (refine) {
                    atomicAdd(g.ex.stat, 1ull);
                } else { // withdrawn: the FP32 answer stands
                    o_ij[0] = s_idx[0];
                    o_ij[1] = s_idx[1];
                    if (publish) {
                        __threadfence_system();
                        __hip_atomic_store(tg.done_flag, tg.progress ? k + 1 : tg.done_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                }
            }
        }
        if (refine) {
            if (s_last) { // a near-tie: the reference's own arithmetic decides (dog_exact.hpp); this workgroup's LDS is the scratch
                const refine_params_ptr rp = (refine_params_ptr)(unsigned long long)tg.rp;
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
                c.cbw = tg.ref_cbw;
                c.tile_rows = tg.ref_rows;
                c.lds = smem;
                const bool slots_ok = nsub <= TILED_SLOT_CAP;
                auto may = [&](int x0, int x1, float thr) { // sub-window (s1, s2) covers window columns [s2·sn2, s2·sn2 + sn2)
                    if (!slots_ok) return true;
                    for (int c2 = x0 / tg.sn2; c2 <= (x1 - 1) / tg.sn2 && c2 < tg.ns2; ++c2)
                        for (int c1 = 0; c1 < tg.ns1; ++c1)
                            if (s_pv[c2 * tg.ns1 + c1] >= thr) return true;
                    return false;
                };
                const int idx = refine_window<4>(NT, g, frame, g1, g2, s_max, c, may);
                if (tid == 0) {
                    const int x = idx / g.n1, y = idx - x * g.n1;
                    const int i = min(max(g1 - g.r1 + y, 1), g.fh);
                    const int j = min(max(g2 - g.r2 + x, 1), g.fw);
                    o_ij[0] = i;
                    o_ij[1] = j;
                    __hip_atomic_store(&cur[0], i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&cur[1], j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (publish) {
                        __threadfence_system();
                        __hip_atomic_store(tg.done_flag, tg.progress ? k + 1 : tg.done_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                    __hip_atomic_store(flag, k + 1 < tg.chain_len ? (unsigned)(k + 1) : 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); // (0 again after the last frame)
                }
                __syncthreads();
                if (k + 1 < tg.chain_len) zero_padding();
            }
            if (k + 1 < tg.chain_len) { // the refined answer is the next guess (:167): it comes through the frame flag
                if (tid == 0) {
                    if (!wait_counter(flag, (unsigned)(k + 1), g.ex, tg.abort)) s_abort = 1;
                    s_idx[0] = __hip_atomic_load(&cur[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    s_idx[1] = __hip_atomic_load(&cur[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                __syncthreads();
                if (s_abort) break;
            }
        }
        if (k + 1 < tg.chain_len) { // :167
            g1 = s_idx[0];
            g2 = s_idx[1];
            __syncthreads();
        }
    }
}

} // namespace pdog
