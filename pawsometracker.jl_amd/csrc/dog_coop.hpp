// dog_coop.hpp — the serial chain of ONE clip whose search window is too large for the fused kernel (e.g. 257×257):
// one cooperative launch per clip instead of three stream-ordered launches per frame.
//
// The reference's real use is exactly this loop (/root/reference/src/PawsomeTracker.jl:163-169, the intended :167):
// frame k is searched around frame k−1's answer.  With a window of 257×257 the two-pass kernels spread one frame over
// ≈50 workgroups, and the frame-to-frame dependency was carried by stream order: row pass, column pass, finish = three
// launches and ≈28 µs per frame, most of it launch gaps.  Here G workgroups stay resident (cooperative launch: the
// runtime guarantees co-residency or refuses) and walk the frames together:
//     row pass (dog_twopass.hpp's h1_block, blocks dealt round-robin)      → grid barrier
//     column pass + partial peaks (hpass_block)                            → grid barrier
//     workgroup 0: combine, clamp (:58-61), exact-mode refinement, next guess → grid barrier
// Same arithmetic as the two-pass kernels (same block functions), same finishing logic as dog_finish_kernel.
//
// MEASURED (1080p, 257×257 window, MI355X): 48 µs per frame with cooperative_groups' grid sync, 31 µs with the barriers
// below, against 27.6 µs for the three stream-ordered launches — a frame's critical path is ≈12 dependent memory round
// trips (tile → RT → partials → guess → DC samples …), not launch gaps, and the launches overlap their own set-up with
// the previous kernel's tail.  The path is therefore OPT-IN (PDOG_COOP=1) and kept for the record and for its test.
#pragma once
#include "dog_twopass.hpp"
#include "dog_exact.hpp"

namespace pdog {

struct CoopGeo {
    TwoPassGeo tg;           // g.frames = the clip's first frame; RT / part_* sized for ONE window; hblocks_per_win = partial slots
    int n_frames;
    const int *start;        // device, 2 ints: the start guess (1-based row, col)
    int32_t *out_ij;         // [n_frames][2]: device memory, or host-coherent memory (pdog_detect_chain_progress)
    int *cur;                // device, 2 ints: the current guess (workgroup 0 → everyone, across the grid barrier)
    const RefineParams *rp;  // exact mode's constants (null = off)
    int ref_cbw, ref_rows;   // refinement: window columns per block; resident tile rows (its scratch is this kernel's LDS)
    int32_t *progress;       // NULL, or a host-coherent word that receives k + 1 after frame k (system-scope release)
    unsigned *sync;          // device, 3 words zeroed before the launch: barrier arrivals, column-pass arrivals, frame flag
};

constexpr int COOP_HR = 8; // window columns per column-pass block (as the two-pass kernels' 8-row form)

// Grid barrier for workgroups the cooperative launch keeps resident: a monotonic arrival counter (zeroed by the host
// before the launch), release on arrival, acquire spin until everyone of this round has arrived.  (HIP's
// cooperative_groups grid sync measured ≈12 µs per barrier here — more than the three launches it was to replace.)
__device__ __forceinline__ void coop_barrier(unsigned *ctr, unsigned target, const ExactCtl &x)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        (void)wait_counter(ctr, target, x);
    }
    __syncthreads();
}

static __global__ __launch_bounds__(256) void dog_coop_chain_kernel(const CoopGeo cg, const f2 *__restrict__ taps_row,
                                                                    const f2 *__restrict__ taps_col)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ int s_refine, s_last;
    __shared__ float s_max, s_sec2;
    __shared__ int s_idx2;
    const TwoPassGeo &tg = cg.tg;
    const LaunchGeo &g = tg.g;
    const int tid = threadIdx.x;
    const unsigned G = gridDim.x;
    unsigned *const bar = cg.sync, *const arrive = cg.sync + 1, *const flag = cg.sync + 2;
    for (int k = 0; k < cg.n_frames; ++k) {
        int g1, g2;
        if (k == 0) {
            g1 = cg.start[0];
            g2 = cg.start[1];
        } else { // written by the finishing workgroup of frame k − 1 before it released the frame flag
            g1 = __hip_atomic_load(&cg.cur[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            g2 = __hip_atomic_load(&cg.cur[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const uint8_t *__restrict__ frame = g.frames + (long long)k * g.frame_stride;
        // ---- row pass → RT (global scratch, transposed) ----
        for (int rb = blockIdx.x; rb < tg.h1blocks_per_win; rb += G) {
            h1_block<13, 8, true>(tg, taps_row, smem, 0, rb, frame, g1, g2, 0);
            __syncthreads(); // the LDS tile is rewritten by the next block
        }
        __threadfence();
        coop_barrier(bar, (unsigned)(k + 1) * G, g.ex); // every RT row is there (and nothing stale of frame k − 1 in this CU's cache)
        // ---- column pass + partial peaks ----
        for (int cb = blockIdx.x; cb < tg.hblocks_per_win; cb += G) {
            const Peak pk = hpass_block<7, 16, false, COOP_HR>(tg, taps_col, smem, 0, cb, 0);
            if (tid == 0) {
                __hip_atomic_store(&g.part_val[cb], pk.best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&g.part_idx[cb], pk.idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&g.part_sec[cb], pk.second, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        // ---- the workgroup that arrives last combines, clamps, refines and releases the frame; the others wait for it ----
        if (tid == 0) {
            const unsigned old = __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            s_last = (old == (unsigned)(k + 1) * G - 1u);
        }
        __syncthreads();
        if (s_last) {
            if (tid < 64) {
                Peak pk;
                peak_init(pk);
                for (int s = tid; s < tg.hblocks_per_win; s += 64)
                    peak_merge(pk, __hip_atomic_load(&g.part_val[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                               __hip_atomic_load(&g.part_idx[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                               __hip_atomic_load(&g.part_sec[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                peak_wave_reduce(pk);
                if (tid == 0) {
                    if (k == 0) range_check(g.ex, g1, g2, g.L >> 1, g.fh, g.fw);
                    const bool rf = cg.rp && (pk.best - pk.second <= g.ex.T);
                    if (rf) atomicAdd(g.ex.stat, 1ull);
                    s_refine = rf;
                    s_max = pk.best;
                    s_sec2 = pk.second;
                    s_idx2 = pk.idx;
                    const int x = pk.idx / g.n1, y = pk.idx - x * g.n1;
                    cg.cur[0] = min(max(g1 - g.r1 + y, 1), g.fh);   // :60-61 (overwritten below if refined)
                    cg.cur[1] = min(max(g2 - g.r2 + x, 1), g.fw);
                }
            }
            __syncthreads();
            if (s_refine) {
                const refine_params_ptr rp = (refine_params_ptr)(unsigned long long)cg.rp;
                RefineCtx c;
                c.trow = as_taps(taps_row);
                c.tcol = as_taps(taps_col);
                c.K = (k64_ptr)(unsigned long long)rp->K64;
                c.g64 = (k64_ptr)(unsigned long long)rp->g64;
                c.dir = rp->dir;
                c.T64 = rp->T64;
                c.T = g.ex.T;
                c.second = s_sec2;
                c.fp32_idx = s_idx2;
                c.cbw = cg.ref_cbw;
                c.tile_rows = cg.ref_rows;
                c.lds = smem;
                auto may = [&](int x0, int x1, float thr) { // partial slot s covers window columns [8s, 8s + 8)
                    for (int s = x0 / COOP_HR; s <= (x1 - 1) / COOP_HR && s < tg.hblocks_per_win; ++s)
                        if (__hip_atomic_load(&g.part_val[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= thr) return true;
                    return false;
                };
                const int idx = refine_window<8>(256, g, frame, g1, g2, s_max, c, may);
                if (tid == 0) {
                    const int x = idx / g.n1, y = idx - x * g.n1;
                    cg.cur[0] = min(max(g1 - g.r1 + y, 1), g.fh);
                    cg.cur[1] = min(max(g2 - g.r2 + x, 1), g.fw);
                }
                __syncthreads();
            }
            if (tid == 0) {
                const int i = cg.cur[0], j = cg.cur[1];
                cg.out_ij[2 * k] = i;
                cg.out_ij[2 * k + 1] = j;
                if (cg.progress) {
                    __threadfence_system();
                    __hip_atomic_store(cg.progress, k + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                __hip_atomic_store(flag, (unsigned)(k + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); // the next guess is out
            }
        } else if (tid == 0) {
            (void)wait_counter(flag, (unsigned)(k + 1), g.ex);
        }
        __syncthreads();
    }
}

} // namespace pdog
