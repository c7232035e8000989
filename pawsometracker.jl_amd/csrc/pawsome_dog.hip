// pawsome_dog.hip — host side of the C ABI declared in include/pawsome_dog.h.
//
// Mirrors the `Tracker` constructor (/root/reference/src/PawsomeTracker.jl:39-52):
// σ and the two Gaussian factors of Kernel.DoG are built in Float64 on the host
// exactly as the reference builds them, rounded to f32 once, and kept on the
// device; the functor (:55-62) becomes kernel launches on a HIP stream.
// There is NO CPU fallback: without a gfx950 device pdog_create fails.
#include "../../include/pawsome_dog.h"
#include "dog_kernels.hpp"
#include "dog_roll.hpp"
#include "dog_twopass.hpp"
#include "dog_fused.hpp"
#include "dog_exact.hpp"
#include "dog_tiled.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <atomic>
#include <emmintrin.h>
#include <chrono>
#include <map>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

using namespace pdog;

// compile-time-length instances of the latency kernels: built in lat_inst.hip, one translation unit per length
namespace pdog {
#define PDOG_LAT_L(LT)                                                                                          \
    extern template __global__ void dog_fused_kernel<false, 0, LT>(const FusedGeo, const f2 *, const f2 *);    \
    extern template __global__ void dog_fused_kernel<true, 0, LT>(const FusedGeo, const f2 *, const f2 *);     \
    extern template __global__ void dog_tiled_kernel<false, LT>(const TiledGeo, const f2 *, const f2 *);       \
    extern template __global__ void dog_tiled_kernel<true, LT>(const TiledGeo, const f2 *, const f2 *);
#include "lat_lengths.def"
#undef PDOG_LAT_L
} // namespace pdog

// the roll kernels are instantiated in roll_inst.hip (one translation unit per set of lengths, built in parallel)
namespace pdog {
#define PDOG_ROLL_L(LT)                                                                                          \
    extern template __global__ void dog_roll_kernel<LT, false, 0>(const LaunchGeo, const f2 *, const f2 *);      \
    extern template __global__ void dog_roll_kernel<LT, true, 0>(const LaunchGeo, const f2 *, const f2 *);       \
    extern template __global__ void dog_thin_kernel<LT, false>(const LaunchGeo, const f2 *, const f2 *);         \
    extern template __global__ void dog_thin_kernel<LT, true>(const LaunchGeo, const f2 *, const f2 *);          \
    extern template __global__ void dog_chain_kernel<LT>(const ChainGeo, const f2 *, const f2 *);
#include "roll_lengths.def"
#undef PDOG_ROLL_L
#define PDOG_EPI_CLASSES(X) X(10) X(2) X(16) X(14) X(6) X(4) X(0) X(1) X(3) X(5) X(7) X(8) X(9) X(11) X(12) X(13) X(15) X(17) // roll_inst.hip: every window-height class
#define PDOG_EPI_DECL(C) extern template __global__ void dog_roll_kernel<65, false, 0, C>(const LaunchGeo, const f2 *, const f2 *);
PDOG_EPI_CLASSES(PDOG_EPI_DECL)
#undef PDOG_EPI_DECL
} // namespace pdog

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}

// Run-time configuration.  The PRODUCT build reads four RESOURCE limits from the environment, once, when a tracker is
// created (never on the launch path: a functor call's whole budget is ≈23 µs).  Everything that selects a code path is
// either API (pdog_set_exact, pdog_set_variant, pdog_set_tuning — explicit, per tracker) or exists only in the diagnostic
// build (make diag, -DPDOG_ABLATIONS), which also reads the tuning switches from the environment for tools/.
struct Switches {
    // resource limits (environment, product build)
    size_t scratch_cap = (size_t)6 << 30; // PDOG_SCRATCH_MB: HBM scratch of the two-pass intermediate; larger batches go in chunks
    size_t map_cap = (size_t)4 << 30;     // PDOG_MAP_MB: largest FP32 response map the two-pass path keeps for exact mode (beyond: candidates are recomputed)
    int host_threads = 0;                 // PDOG_HOST_THREADS (0: min(16, cores))
    int ingest_chunk = 0;                 // PDOG_INGEST_CHUNK (0: ≈32 MB of tiles)
    // path selection (pdog_set_tuning; the tests exercise every alternative path of the product through it)
    bool host_copy = false;               // functor: upload the tile with copy commands and synchronise the stream instead of the in-place tile + ticket
    bool host_sync = false;               // functor: in-place tile, but wait with hipStreamSynchronize instead of polling the ticket
    bool twopass_4l = false;              // two-pass path always in its four-launch form
    bool no_tiled = false;                // single large windows stay on the two-pass launches
    bool no_roll_map = false;             // hard batches on the roll / ring kernels keep recomputing their candidates
    bool no_fold = false;                 // a single remainder column always goes to dog_thin_kernel
    bool fold_always = false;             // … always into the last strip, also below 8 strips per window
    bool no_fused_c = false;              // the fused kernel's runtime-length instance also where a compile-time-l instance exists
    bool fault_inject = false;            // tests: one sub-window of a tiled chain never delivers its second frame's partial (the device-side waits must give up)
    // tuning and diagnosis (diagnostic build only)
    bool host_trace = false, hpass16 = false, ingest_no_nt = false, ingest_trace = false, fused_diag = false;
    bool tiled_force = false;             // the tiled kernel also for windows the fused kernel serves
    bool no_host_dc = false;              // the functor's kernels sample the DC level themselves
    int tiled_sub = 0;                    // sub-window edge of the tiled kernel (0: chosen per geometry)
    int tp_ph1 = 0, tp_php = 0;           // outputs per task of the two-pass kernels (0: per geometry)
    int v_after = 64;                     // exact mode, map path: first-scan candidates beyond which the window's own |pixel − dc| bound is computed
    int h1_u = 4;                         // taps per block of the two-pass row pass (4, or 8)
    int hp_u = 8;                         // taps per block of the two-pass column pass (8 or 16)
    int tiled_batch = 2;                  // windows per batch up to which the tiled kernel is used
    int fused_pr = 0, fused_pc = 0;       // outputs per task of the fused kernel (0: chosen per geometry)
    int lds_pad = 0;                      // occupancy experiments
};
Switches read_switches()
{
    Switches w;
    if (const char *e = std::getenv("PDOG_SCRATCH_MB")) w.scratch_cap = (size_t)std::max(1, std::atoi(e)) << 20;
    if (const char *e = std::getenv("PDOG_MAP_MB")) w.map_cap = (size_t)std::max(0, std::atoi(e)) << 20;
    if (const char *e = std::getenv("PDOG_HOST_THREADS")) w.host_threads = std::max(1, std::min(64, std::atoi(e)));
    if (const char *e = std::getenv("PDOG_INGEST_CHUNK")) w.ingest_chunk = std::max(1, std::atoi(e));
#ifdef PDOG_ABLATIONS
    auto on = [](const char *n) { return std::getenv(n) != nullptr; };
    w.host_copy = on("PDOG_HOST_COPY");
    w.host_sync = on("PDOG_HOST_SYNC");
    w.host_trace = on("PDOG_HOST_TRACE");
    w.twopass_4l = on("PDOG_TWOPASS_4L");
    w.hpass16 = on("PDOG_HPASS16");
    w.ingest_no_nt = on("PDOG_INGEST_NO_NT");
    w.ingest_trace = on("PDOG_INGEST_TRACE");
    w.fused_diag = on("PDOG_FUSED_DIAG");
    w.no_tiled = on("PDOG_NO_TILED");
    w.no_fold = on("PDOG_NO_FOLD");
    w.fold_always = on("PDOG_FOLD_ALWAYS");
    w.no_roll_map = on("PDOG_NO_ROLL_MAP");
    w.no_host_dc = on("PDOG_NO_HOST_DC");
    w.tiled_force = on("PDOG_TILED_FORCE");
    if (const char *e = std::getenv("PDOG_TILED_SUB")) w.tiled_sub = std::max(0, std::min(96, std::atoi(e)));
    if (const char *e = std::getenv("PDOG_TILED_BATCH")) w.tiled_batch = std::max(0, std::atoi(e));
    if (const char *e = std::getenv("PDOG_V_AFTER")) w.v_after = std::max(0, std::atoi(e));
    if (const char *e = std::getenv("PDOG_H1_U")) w.h1_u = std::atoi(e) == 8 ? 8 : 4;
    if (const char *e = std::getenv("PDOG_HP_U")) w.hp_u = std::atoi(e) == 16 ? 16 : 8;
    if (const char *e = std::getenv("PDOG_TP_P")) {
        int a = 0, b = 0;
        if (std::sscanf(e, "%d,%d", &a, &b) == 2 && (a == 5 || a == 7 || a == 9 || a == 11 || a == 13 || a == 17) && (b == 5 || b == 7 || b == 9 || b == 13)) { w.tp_ph1 = a; w.tp_php = b; }
    }
    if (const char *e = std::getenv("PDOG_FUSED_P")) {
        int a = 0, b = 0;
        if (std::sscanf(e, "%d,%d", &a, &b) == 2 && (a == 3 || a == 4 || a == 5 || a == 6 || a == 8) &&
            (b == 2 || b == 3 || b == 4 || b == 6 || b == 8)) { w.fused_pr = a; w.fused_pc = b; }
    }
    if (const char *e = std::getenv("PDOG_LDS_PAD")) w.lds_pad = std::max(0, std::atoi(e));
#endif
    return w;
}

#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t e__ = (expr);                                                            \
        if (e__ != hipSuccess)                                                              \
            return fail(PDOG_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e__));    \
    } while (0)

// ---- Float64 host arithmetic, as the reference does it ----
double sigma_of(double tw) { return tw / (2.0 * std::sqrt(2.0 * std::log(2.0))); } // :30
int kernel_len_of_sigma(double s) { return 4 * (int)std::ceil(s * std::sqrt(2.0)) + 1; } // Kernel.DoG
void gaussian_1d(double s, int l, double *g)
{ // KernelFactors.gaussian: exp(-x²/2σ²) / sum
#pragma clang fp contract(off)
    const int w = l >> 1;
    for (int x = -w; x <= w; ++x) g[x + w] = std::exp(-((double)x * (double)x) / (2.0 * s * s));
    double sum = 0.0;
    for (int i = 0; i < l; ++i) sum += g[i];
    for (int i = 0; i < l; ++i) g[i] /= sum;
}

// The reference's dense kernel, K = dir·(g₊⊗g₊ − g₋⊗g₋) (src/PawsomeTracker.jl:41-43), column-major, every product and
// the difference rounded separately as Julia evaluates them: hipcc contracts a·b − c·d into an FMA by default, which
// changes last bits — and last bits are exactly what decides the ties exact mode exists for.
void dense_dog_kernel(const double *gp, const double *gm, int l, bool darker, double *K)
{
#pragma clang fp contract(off)
    const double dir = darker ? -1.0 : 1.0;
    for (int j = 0; j < l; ++j)
        for (int i = 0; i < l; ++i) {
            const double a = gp[i] * gp[j];
            const double b = gm[i] * gm[j];
            K[i + (size_t)l * j] = dir * (a - b);
        }
}

// MaxDynamicSharedMemorySize is a per-function (per-device) attribute shared by every tracker in the
// process: only ever raise it, so that a tracker with a small window cannot shrink the limit under a
// live tracker with a large one.
int raise_lds_limit(const void *fn, size_t bytes)
{
    static std::mutex mu;
    static std::map<std::pair<int, const void *>, size_t> limit;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return fail(PDOG_E_HIP, "hipGetDevice failed");
    std::lock_guard<std::mutex> lock(mu);
    size_t &cur = limit[{dev, fn}];
    if (bytes <= cur) return PDOG_OK;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return fail(PDOG_E_HIP, std::string("hipFuncSetAttribute(MaxDynamicSharedMemorySize): ") + hipGetErrorString(e));
    cur = bytes;
    return PDOG_OK;
}

// ---- compiled kernel specialisations ----
typedef void (*kernel_fn)(const LaunchGeo, const f2 *, const f2 *);
struct Variant {
    int id, P, XG, Q, CH, LT, NT;
    kernel_fn fn, fn_resp; // fn_resp also writes the dense response (parity checks)
    bool roll;             // dog_roll.hpp (one wave per 64-column strip) instead of dog_kernels.hpp
    kernel_fn thin, thin_resp; // remainder-column kernel of the roll variants (may be null)
    int twopass = 0;           // dog_twopass.hpp: vertical pass → HBM → horizontal pass (Q = this value)
    typedef void (*chain_fn)(const ChainGeo, const f2 *, const f2 *);
    chain_fn chain = nullptr;  // persistent serial-chain kernel of the roll variants
    bool fused = false;        // dog_fused.hpp: one workgroup per window, whole tile in LDS
    int tw() const { return P * XG; }
    int ring(int L) const { return LT ? ring_rows(CH, LT, Q) : ring_rows(CH, L, Q); }
    int pa(int L) const { return pitch_a(tw() + L - 1); }
    size_t lds(int L) const
    {
        if (twopass || fused) return 0; // sized per window at launch
        if (roll) return roll_lds_bytes(LT);
        return (size_t)round_up(CH * pa(L) * 4, 16) + (size_t)ring(L) * pitch_r(tw()) * sizeof(f2);
    }
};
#define PDOG_VARIANT(id, P, XG, Q, CH, LT, NT) \
    Variant { id, P, XG, Q, CH, LT, NT, (kernel_fn)dog_window_kernel<P, XG, Q, CH, LT, NT, false>, \
              (kernel_fn)dog_window_kernel<P, XG, Q, CH, LT, NT, true>, false, nullptr, nullptr }
#define PDOG_ROLL_VARIANT(id, LT) \
    Variant { id, ROLL_P, ROLL_TW / ROLL_P, ROLL_CH, ROLL_CH, LT, 64, (kernel_fn)dog_roll_kernel<LT, false>, \
              (kernel_fn)dog_roll_kernel<LT, true>, true, (kernel_fn)dog_thin_kernel<LT, false>, \
              (kernel_fn)dog_thin_kernel<LT, true>, 0, dog_chain_kernel<LT> }

const Variant kVariants[] = {
    // runtime-L (any target_width)
    PDOG_VARIANT(0, 4, 8, 8, 32, 0, 256),
    PDOG_VARIANT(1, 8, 8, 8, 32, 0, 256),
    PDOG_VARIANT(2, 11, 8, 8, 32, 0, 256),
    // l = 65 (target_width 25, the reference default)
    PDOG_VARIANT(10, 11, 8, 16, 32, 65, 256),
    PDOG_VARIANT(11, 11, 8, 8, 24, 65, 256),
    PDOG_VARIANT(12, 8, 8, 16, 32, 65, 256),
    PDOG_VARIANT(13, 8, 8, 8, 32, 65, 256),
    PDOG_VARIANT(14, 11, 8, 8, 32, 65, 256),
    // rolling-accumulator kernel: one instance per kernel length l = 4m+1 of roll_lengths.def (target_width ≈ 5 … 39); id = 100 + l, l = 65 → 100
#define PDOG_ROLL_L(LT) PDOG_ROLL_VARIANT((LT) == 65 ? 100 : 100 + (LT), LT),
#include "roll_lengths.def"
#undef PDOG_ROLL_L
    // any l: two launches with the intermediate in HBM (long kernels, target_width ≳ 40)
    Variant { 200, 13, 16, 16, 16, 0, 256, nullptr, nullptr, false, nullptr, nullptr, 16 },
    // any l, windows whose padded tile fits in LDS: one workgroup per window, one launch (latency path)
    Variant { 300, 1, 1, 1, 1, 0, FUSED_NT, nullptr, nullptr, false, nullptr, nullptr, 0, nullptr, true },
#ifdef PDOG_ABLATIONS
    Variant { 101, ROLL_P, 8, ROLL_CH, ROLL_CH, 65, 64, (kernel_fn)dog_roll_kernel<65, false, 1>, (kernel_fn)dog_roll_kernel<65, true>, true, nullptr, nullptr },
    Variant { 102, ROLL_P, 8, ROLL_CH, ROLL_CH, 65, 64, (kernel_fn)dog_roll_kernel<65, false, 2>, (kernel_fn)dog_roll_kernel<65, true>, true, nullptr, nullptr },
    Variant { 103, ROLL_P, 8, ROLL_CH, ROLL_CH, 65, 64, (kernel_fn)dog_roll_kernel<65, false, 3>, (kernel_fn)dog_roll_kernel<65, true>, true, nullptr, nullptr },
    Variant { 104, ROLL_P, 8, ROLL_CH, ROLL_CH, 65, 64, (kernel_fn)dog_roll_kernel<65, false, 12>, (kernel_fn)dog_roll_kernel<65, true>, true, nullptr, nullptr },
    Variant { 105, ROLL_P, 8, ROLL_CH, ROLL_CH, 65, 64, (kernel_fn)dog_roll_kernel<65, false, 15>, (kernel_fn)dog_roll_kernel<65, true>, true, nullptr, nullptr },
    Variant { 106, ROLL_P, 8, ROLL_CH, ROLL_CH, 65, 64, (kernel_fn)dog_roll_kernel<65, false, 0>, (kernel_fn)dog_roll_kernel<65, false, 16>, true, nullptr, nullptr },
#endif
    // l = 29 (target_width 10, the reference test default)
    PDOG_VARIANT(20, 8, 8, 4, 32, 29, 256),
};
constexpr int kNumVariants = sizeof(kVariants) / sizeof(kVariants[0]);
constexpr size_t kMaxLds = 160 * 1024;
constexpr int kThinMax = 6; // remainder columns done by dog_thin_kernel instead of one more strip

bool has_roll_instance(int L)
{
    for (int i = 0; i < kNumVariants; ++i)
        if (kVariants[i].roll && kVariants[i].LT == L && kVariants[i].id >= 100 && kVariants[i].id < 300) return true;
    return false;
}

const Variant *find_variant(int id)
{
    for (int i = 0; i < kNumVariants; ++i)
        if (kVariants[i].id == id) return &kVariants[i];
    return nullptr;
}

} // namespace

struct pdog_tracker {
    int device = 0;
    Switches sw;                   // environment switches as they were when the tracker was created
    hipEvent_t ev_switch = nullptr; // orders a new stream behind the work queued on the previous one (pdog_set_stream)
    int fh = 0, fw = 0, r1 = 0, r2 = 0, n1 = 0, n2 = 0, L = 0, fill = 0, darker = 0;
    double tw = 0, sigma = 0;
    const Variant *var = nullptr;
    int nstrips = 0;
    int nthin = 0, thin_x0 = 0; // window columns handled by the thin-remainder kernel
    bool forced_variant = false;   // pdog_set_variant pinned the kernel: no batch-size switching
    bool small_twopass = false;    // two-pass kernels are set up and may take over small batches
    bool fused_ok = false;         // the window's tile fits in LDS: the fused kernel takes small batches and short chains
    int fused_resident = 0;        // workgroups of the fused kernel the device keeps resident (its grid is capped there: a workgroup walks several windows)
    bool fused_c = false;          // … through its compile-time-l instance (dog_fused.hpp: l = 65, the default tracker's), whose tile layout is wider
    int hp_rows = HP_ROWS;         // RT rows (window columns) per column-pass workgroup: 16 (P = 13) or 8 (P = 7)
    int32_t *d_chain_tmp = nullptr; // [2][n_clips][2]: current guesses / step results of multi-clip chains
    int tp_ph1 = 13, tp_php = 7;   // outputs per task of the two-pass row / column pass (pick_twopass_p)
    // tiled kernel (dog_tiled.hpp): one large window cut into sub-windows, a workgroup each, one launch per batch / clip
    bool tiled_ok = false;
    int tiled_sn1 = 0, tiled_sn2 = 0, tiled_ns1 = 0, tiled_ns2 = 0, tiled_pr = 0, tiled_pc = 0, tiled_cshift = 0;
    int tiled_ref_cbw = 1, tiled_ref_rows = 8, tiled_resident = 0; // refinement scratch geometry; workgroups the device keeps resident
    size_t tiled_lds = 0;
    bool tiled_c = false;          // the tiled kernel's compile-time-l instance (l = 65) and tile layout
    int *d_tiled_ctl = nullptr;    // [cap][4]: current guess (2), partial arrivals, frame flag
    int tiled_ctl_cap = 0;
    unsigned long long *d_tiled_slots = nullptr; // [clips][2][nsub][2] tagged partials of the tiled kernel's clips (dog_tiled.hpp)
    long long tiled_slots_cap = 0; // in slots (clips × sub-windows)
    unsigned tiled_tag = 0;        // advanced by chain_len + 1 per clip launch: a frame's tag never repeats
    int chain_tmp_cap = 0;
    // two-pass path scratch
    f2 *d_V = nullptr;
    size_t v_bytes = 0;
    // exact mode on the batch kernels of short kernels (roll / ring): windows flagged per batch as the finishing kernel reports
    // them (h_pinned[6], cumulative); batches of HARD windows (noise only, ±1-level targets: every window flagged) switch to
    // the response-map refinement like the two-pass path — 81–206 ms per 4096 windows of 257×257 without it
    unsigned flag_last = 0, win_last = 0, win_launched = 0; // flagged / finished windows last seen (h_pinned[6], [7]); windows handed to finishing kernels so far
    int flag_calm = 0;
    bool roll_map = false;
    float *d_map = nullptr; // exact mode on the two-pass path: the batch's FP32 responses, where the refinement finds its candidates
    size_t map_bytes = 0;
    int *d_dc = nullptr;
    int dc_cap = 0;
    int *d_counter = nullptr;  // [kLowLatMax] zero between launches: delivered column-pass partials per window (low-latency two-pass)
    hipStream_t own_stream = nullptr, stream = nullptr;
    // side stream + fork/join events: the thin-remainder kernel runs beside the strips (its waves fit in
    // the registers the 2-waves-per-SIMD roll kernel leaves free) instead of after them
    hipStream_t aux_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    f2 *d_taps_row = nullptr, *d_taps_col = nullptr;
    f2 *d_taps_roll = nullptr; // paired column-tap table of dog_roll.hpp
    f2 *d_fold_r = nullptr;    // [n][n1 + l − 1] row-pass outputs of a folded remainder column (LaunchGeo::fold_r)
    size_t fold_bytes = 0;
    float *d_part_val = nullptr, *d_part_sec = nullptr;
    unsigned long long *d_part_mask = nullptr;
    int *d_part_idx = nullptr;
    int cap_windows = 0;
    // exact mode (dog_exact.hpp): windows whose two best FP32 responses lie within 2δ are re-decided in the
    // reference's own Float64 arithmetic
    bool exact = true;
    bool exact_all = false;           // pdog_set_exact(t, 2): refine every window with an infinite threshold (tests: the whole reference computation on the device)
    std::vector<double> h_gp, h_gm;   // the two Float64 Gaussians (host copy): the error bounds follow the kernels' operation order over THESE taps
    double F_sym_int = 0, F_sym_sep = 0, F_ring = 0, F_rescan = 0; // error-bound factors per kernel family (exact_factors)
    double *d_K64 = nullptr;          // dir·(g₊⊗g₊ − g₋⊗g₋), l×l column-major, Float64 (:41-43)
    double *d_g64 = nullptr;          // [2][l] the two normalised Gaussians in Float64 (the refinement's separable stage)
    RefineParams *d_rp = nullptr;     // {K64, g64, dir, T64} for the kernels that refine inline
    double exact_T64 = 0.0;           // 2δ64: separable Float64 against the reference's dense Float64
    unsigned long long *d_ref_stat = nullptr;
    int ref_cbw = 1, ref_rows = 8;    // refinement: window columns per block; rows of the block's pixel tile resident in LDS at a time
    int fused_ref_cbw = 1, fused_ref_rows = 8; // the same inside the fused kernel (its scratch is that kernel's LDS)
    // host-path staging (pdog_detect_host / chain seed)
    uint8_t *d_frame = nullptr;
    int32_t *d_small = nullptr; // [0..1] guess, [2..3] result
    int32_t *h_pinned = nullptr; // pinned, host-coherent mailbox: [0..1] guess, [2..3] result, [4] completion ticket of the functor
    int32_t ticket = 0;
    float *d_resp = nullptr;
    uint8_t *h_tile = nullptr;   // pinned, device-mapped: the functor's window tile when the fused kernel reads it in place
    uint8_t *d_tile_map = nullptr; // device addresses of h_tile and of the h_pinned mailbox
    int32_t *d_mail_map = nullptr;
    // host-batch ingest (pdog_detect_batch_host): rotating pinned staging / device tile slots
    static constexpr int kIngestSlots = 3;
    uint8_t *h_stage[kIngestSlots] = {nullptr, nullptr, nullptr};
    uint8_t *d_tiles[kIngestSlots] = {nullptr, nullptr, nullptr};
    size_t ingest_slot_bytes = 0;
    int32_t *d_ingest_guess = nullptr, *d_ingest_out = nullptr, *h_ingest_out = nullptr;
    int ingest_cap = 0, ingest_guess_cap = 0;
    hipStream_t h2d_stream = nullptr;
    hipEvent_t ev_h2d[kIngestSlots] = {nullptr, nullptr, nullptr}, ev_used[kIngestSlots] = {nullptr, nullptr, nullptr};
};

namespace {

void pack_tile(const pdog_tracker *t, const uint8_t *frame, int64_t row_stride, int g1, int g2, uint8_t *dst, int pitch);
typedef void (*fused_fn_t)(const FusedGeo, const f2 *, const f2 *);
fused_fn_t fused_kernel_for(const pdog_tracker *t, bool resp);

// LDS row pitches of the two-pass kernels: the sliding windows (and their one-block prefetch) of the last,
// partly masked group of 13 outputs must stay inside the zero-padded row.  nout outputs, l taps.
// `quantum` = outputs one round of a workgroup covers (groups × outputs per task); + l rounded up to a block of 16 taps + one block of prefetch
int twopass_pitch_q(int nout, int L, int quantum) { return (round_up(nout, quantum) + L + 48) | 1; }
int twopass_pitch(int nout, int L, int rows = HP_ROWS) { return twopass_pitch_q(nout, L, rows == 8 ? 32 * 7 : 16 * 13); }

// Outputs per task (P) of the two-pass ROW pass, per geometry.  A workgroup covers 16 × P outputs per round: a 257-wide
// window on P = 13 (208 per round) ran a second round 24 % full.  Measured per launch (4096 windows, same session):
//   257 wide, l = 109:  P = 5 1.94 ms, 7 1.83, 9 1.84, 11 2.20, 13 2.39, 17 2.18   (96 VGPRs at P = 9, 155 at 13)
//   205 wide, l = 293:  P = 5 4.35 ms, 7 4.17, 9 4.11, 11 4.61, 13 3.97            (13: one round, 98 % full)
// ⇒ P = 9 unless P = 13 fills its rounds better.  4-tap blocks (70 / 96 VGPRs instead of 94 / 155) gave another 2–5 %:
//   257 wide, l = 109:  P = 9 1.80 ms, 11 1.94, 13 2.17, 17 1.86;   205 wide, l = 293:  P = 9 4.20 ms, 13 3.95, 17 4.64
// COLUMN pass (32 × P outputs per round, 8 taps per block: 76 VGPRs against 125 with 16-tap blocks, −8 % on every config):
//   257 tall, l = 109:  P = 5 2.13 ms, 7 2.36, 9 1.96, 13 2.37
//   205 tall, l = 293:  P = 5 3.44 ms, 7 2.50, 9 2.57, 13 3.45
// ⇒ P = 7 unless P = 9 fills its rounds better.
typedef void (*tp_fn)(TwoPassGeo, const f2 *);
// flush: the blocked-accumulation instances (dog_twopass.hpp), for kernel lengths from TWOPASS_FLUSH_L on.  The product build
// holds the default block sizes only (row pass 4 taps, column pass 8); the diagnostic build adds the 8- / 16-tap instances.
tp_fn h1_kernel_for(int P, bool dcin, int U, bool flush)
{
#define PDOG_H1(PP, UU) (flush ? (dcin ? (tp_fn)dog_h1_kernel<PP, UU, true, true> : (tp_fn)dog_h1_kernel<PP, UU, false, true>) \
                               : (dcin ? (tp_fn)dog_h1_kernel<PP, UU, true, false> : (tp_fn)dog_h1_kernel<PP, UU, false, false>))
#ifdef PDOG_ABLATIONS
    if (U == 8) {
        switch (P) {
        case 5: return PDOG_H1(5, 8);
        case 7: return PDOG_H1(7, 8);
        case 9: return PDOG_H1(9, 8);
        case 11: return PDOG_H1(11, 8);
        case 17: return PDOG_H1(17, 8);
        default: return PDOG_H1(13, 8);
        }
    }
#endif
    (void)U;
    switch (P) {
    case 9: return PDOG_H1(9, 4);
    case 17: return PDOG_H1(17, 4);
    case 11: return PDOG_H1(11, 4);
    default: return PDOG_H1(13, 4);
    }
#undef PDOG_H1
}
tp_fn hpass8_kernel_for(int P, bool resp, bool fin, int U, bool flush)
{
#define PDOG_HP8F(PP, UU, FF) (fin ? (resp ? (tp_fn)dog_hpass_kernel<PP, UU, true, 8, true, FF> : (tp_fn)dog_hpass_kernel<PP, UU, false, 8, true, FF>) \
                                   : (resp ? (tp_fn)dog_hpass_kernel<PP, UU, true, 8, false, FF> : (tp_fn)dog_hpass_kernel<PP, UU, false, 8, false, FF>))
#define PDOG_HP8(PP, UU) (flush ? PDOG_HP8F(PP, UU, true) : PDOG_HP8F(PP, UU, false))
#ifdef PDOG_ABLATIONS
    if (U == 16) {
        switch (P) {
        case 5: return PDOG_HP8(5, 16);
        case 9: return PDOG_HP8(9, 16);
        default: return PDOG_HP8(7, 16);
        }
    }
#endif
    (void)U;
    switch (P) {
    case 5: return PDOG_HP8(5, 8);
    case 9: return PDOG_HP8(9, 8);
    case 13: return PDOG_HP8(13, 8);
    default: return PDOG_HP8(7, 8);
    }
#undef PDOG_HP8
#undef PDOG_HP8F
}
int pick_h1_outputs(int nout)
{
    auto fill = [&](int p) { return (double)nout / (double)(round_up(nout, 16 * p)); };
    return fill(13) > fill(9) + 0.05 ? 13 : 9;
}
int pick_hpass_outputs(int nout)
{
    auto fill = [&](int p) { return (double)nout / (double)(round_up(nout, 32 * p)); };
    return fill(9) > fill(7) + 0.05 ? 9 : 7;
}
// Exact mode's refinement (dog_exact.hpp) works on blocks of `cbw` window columns whose row-pass result (two doubles
// per element in the Float64 stage) fits ≈24 KB of LDS; the block's pixels go through an LDS tile — all n1 + l − 1 rows
// resident when that fits 100 KB in total (l ≲ 150: a candidate's exact chain then reads LDS), otherwise ≈24 KB slices.
int refine_slice_rows(int NA, int L, int cbw, size_t budget)
{
    return (int)std::max<size_t>(8, std::min<size_t>((size_t)NA, budget / (size_t)refine_tile_pitch(cbw, L)));
}
size_t fused_tile_lds(const pdog_tracker *t)
{
    return t->fused_c ? fusedc_lds_bytes(t->n1, t->n2, t->L) : fused_lds_bytes(t->n1, t->n2, t->L);
}
void setup_refine_geometry(pdog_tracker *t)
{
    const int NA = t->n1 + t->L - 1;
    t->ref_cbw = std::max(1, std::min({8, t->n2, (int)(24576 / ((size_t)NA * 16))}));
    t->ref_rows = refine_lds_bytes(t->n1, t->L, t->ref_cbw, NA) <= 100 * 1024 ? NA : refine_slice_rows(NA, t->L, t->ref_cbw, 24576);
    // inside the fused kernel the scratch is that kernel's own LDS (tile + RT, free by then): the widest block that fits
    // it with the whole pixel tile resident (a window whose tile fits as floats has room for it as bytes)
    t->fused_ref_cbw = 1;
    t->fused_ref_rows = NA;
    t->fused_c = fused_has_instance(t->L) && fusedc_lds_bytes(t->n1, t->n2, t->L) <= kMaxLds - 1024;
    if (t->sw.no_fused_c) t->fused_c = false;
    const size_t have = fused_tile_lds(t);
    for (int cbw = std::min(8, t->n2); cbw >= 1; --cbw)
        if (refine_lds_bytes(t->n1, t->L, cbw, NA) <= have) { t->fused_ref_cbw = cbw; break; }
    if (refine_lds_bytes(t->n1, t->L, 1, NA) > kMaxLds - 1024) t->fused_ref_rows = refine_slice_rows(NA, t->L, 1, 24576);
}
// dynamic LDS of the fused kernel: its tile + RT, or the scratch of the refinement it may run in the same memory
size_t fused_total_lds(const pdog_tracker *t)
{
    return std::max(fused_tile_lds(t), refine_lds_bytes(t->n1, t->L, t->fused_ref_cbw, t->fused_ref_rows));
}

int ensure_capacity(pdog_tracker *t, int n);
enum KernelFamily : int { kFamRoll, kFamRing, kFamFused, kFamTwoPass8, kFamTwoPass16 };
ExactCtl exact_ctl(const pdog_tracker *t, KernelFamily fam);

// Outputs per task of the fused / tiled kernels: fewest rounds of 1024 tasks, then least work per task (≈ P outputs + a fixed cost)
int pick_outputs_per_task(int lines, int nout, std::initializer_list<int> ps, double fixed)
{
    int best = 0;
    double best_cost = 0;
    for (int p : ps) {
        const long long tasks = (long long)lines * ((nout + p - 1) / p);
        const double cost = (double)((tasks + FUSED_NT - 1) / FUSED_NT) * (p + fixed);
        if (!best || cost < best_cost) { best = p; best_cost = cost; }
    }
    return best;
}

// Tiled kernel (dog_tiled.hpp): windows too large for the fused kernel, cut into sub-windows of ≈48 rows/columns (a
// 257×257 window: 6×6 of 43×43, each a tile of 107×107 like the default 45×45 window of the fused kernel).
const void *tiled_kernel_for(const pdog_tracker *t, bool resp)
{
    if (t->tiled_c) switch (t->L) {
#define PDOG_LAT_L(LT) case LT: return resp ? (const void *)dog_tiled_kernel<true, LT> : (const void *)dog_tiled_kernel<false, LT>;
#include "lat_lengths.def"
#undef PDOG_LAT_L
        default: break;
    }
    return resp ? (const void *)dog_tiled_kernel<true> : (const void *)dog_tiled_kernel<false>;
}
int setup_tiled(pdog_tracker *t)
{
    t->tiled_ok = false;
    if ((t->fused_ok && !t->sw.tiled_force) || t->sw.no_tiled || t->fw < 4) return PDOG_OK;
    // Sub-window edge: ≈32 measured best for a 257×257 window (81 workgroups: 13.9 µs per frame; 48: 15.3), but beyond
    // ≈128 workgroups the partial exchange costs more than smaller tiles save (513×513: 121 workgroups of 47 → 17.8 µs,
    // 169 of 40 → 21.9); long kernels need smaller sub-windows for their halo to fit LDS.
    int sn1 = 0, sn2 = 0, ns1 = 0, ns2 = 0;
    size_t need = 0;
    bool found = false;
    const int user = t->sw.tiled_sub;
    t->tiled_c = fused_has_instance(t->L) && !t->sw.no_fused_c;
    // (round 3, with the one-round-trip exchange: 129×129 at 24 → 36 workgroups 8.6–8.8 µs against 9.2–9.7 at 32 → 25; 257×257 at 20 / 24 /
    // 32: 9.8–10.1 / 10.2 / 10.2–10.5 — kept at 32 so that three clips stay resident; 513×513 at 36 → 225 workgroups 12.2–12.6 against 13.0 at 48)
    bool small_first = true;
    for (int target : {24, 32, 40, 48, 56, 64, 24, 16}) {
        const bool probe24 = small_first && target == 24; // small windows first try 24: only while that keeps the clip at ≤ 48 workgroups
        small_first = false;
        if (user) target = user;
        ns1 = (t->n1 + target - 1) / target;
        ns2 = (t->n2 + target - 1) / target;
        sn1 = (t->n1 + ns1 - 1) / ns1;
        sn2 = (t->n2 + ns2 - 1) / ns2;
        ns1 = (t->n1 + sn1 - 1) / sn1;
        ns2 = (t->n2 + sn2 - 1) / sn2;
        need = t->tiled_c ? fusedc_lds_bytes(sn1, sn2, t->L) : fused_lds_bytes(sn1, sn2, t->L);
        const bool fits = need <= kMaxLds - 1024 && sn2 + t->L - 1 <= 4 * FUSED_NT && (long long)t->n1 * t->n2 < (1 << 24) && // (a partial's index shares a word with 8 tag bits)
                          (long long)ns1 * ns2 <= (target >= 32 && !user ? 128 : TILED_SLOT_CAP);
        if (fits && !(probe24 && !user && (long long)ns1 * ns2 > 48)) { found = true; break; }
        if (user) break;
    }
    if (!found) return PDOG_OK;
    // the refinement's scratch is the kernel's own LDS: at least its smallest form must fit; then the widest block that does
    const size_t base = std::max(need, refine_lds_bytes(t->n1, t->L, 1, 8));
    if (base > kMaxLds - 1024) return PDOG_OK;
    const int NA = t->n1 + t->L - 1;
    t->tiled_ref_cbw = 1;
    t->tiled_ref_rows = 8;
    for (int cbw = std::min(t->n2, t->ref_cbw); cbw >= 1; --cbw) {
        const size_t fixed_r = refine_lds_bytes(t->n1, t->L, cbw, 0);
        if (fixed_r + (size_t)8 * refine_tile_pitch(cbw, t->L) > base) continue;
        t->tiled_ref_cbw = cbw;
        t->tiled_ref_rows = (int)std::min<size_t>((size_t)NA, (base - fixed_r) / (size_t)refine_tile_pitch(cbw, t->L));
        while (t->tiled_ref_rows > 8 && refine_lds_bytes(t->n1, t->L, cbw, t->tiled_ref_rows) > base) --t->tiled_ref_rows;
        break;
    }
    t->tiled_sn1 = sn1; t->tiled_sn2 = sn2; t->tiled_ns1 = ns1; t->tiled_ns2 = ns2;
    t->tiled_lds = base;
    t->tiled_cshift = 0;
    while ((1 << t->tiled_cshift) < (sn2 + t->L - 1 + 3) / 4) ++t->tiled_cshift;
    t->tiled_pr = t->tiled_c ? fusedc_row_outputs(sn1 + t->L - 1, sn2) : pick_outputs_per_task(sn1 + t->L - 1, sn2, {3, 4, 5, 6, 8}, 2.0);
    t->tiled_pc = pick_outputs_per_task(sn2, sn1, {2, 3, 4, 6, 8}, 1.5);
    for (bool resp : {false, true})
        if (int rc = raise_lds_limit(tiled_kernel_for(t, resp), base)) return rc;
    int per_cu = 0, cus = 0, coop = 0;
    t->tiled_resident = 0; // chains need every workgroup of a clip resident at once: a cooperative launch, if the device has them
    if (hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, t->device) == hipSuccess && coop &&
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, tiled_kernel_for(t, false), FUSED_NT, base) == hipSuccess && per_cu >= 1 &&
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, t->device) == hipSuccess)
        t->tiled_resident = per_cu * cus;
    t->tiled_ok = true;
    return PDOG_OK;
}

// n windows (chain_len = 1: independent; ordinary launch) or n clips of chain_len frames (cooperative launch: the
// workgroups of a clip wait for each other's partials frame by frame).  *launched = false: not taken, the caller goes on.
int launch_tiled(pdog_tracker *t, const uint8_t *d_frames, int64_t frame_stride, int64_t row_stride, const int32_t *d_frame_index,
                 const int32_t *d_guesses, int n, int chain_len, int32_t *d_out_ij, float *d_out_resp, int FH, int FW,
                 int32_t *d_done_flag, int32_t done_value, bool progress, bool *launched, int dc_host = -1)
{
    *launched = false;
    if (!t->tiled_ok || n < 1) return PDOG_OK;
    const int nsub = t->tiled_ns1 * t->tiled_ns2;
    if (chain_len > 1 && (long long)n * nsub > t->tiled_resident) return PDOG_OK;
    if (t->cap_windows < n || t->tiled_ctl_cap < n) {
        HIP_TRY(hipStreamSynchronize(t->stream));
        if (t->cap_windows < n) { if (int rc = ensure_capacity(t, n)) return rc; }
        if (t->tiled_ctl_cap < n) {
            if (t->d_tiled_ctl) (void)hipFree(t->d_tiled_ctl);
            t->d_tiled_ctl = nullptr; t->tiled_ctl_cap = 0;
            HIP_TRY(hipMalloc(&t->d_tiled_ctl, sizeof(int) * (4 * (size_t)n + 1)));
            HIP_TRY(hipMemset(t->d_tiled_ctl, 0, sizeof(int) * (4 * (size_t)n + 1))); // the kernel leaves its counters at zero; the last word is the abort word
            t->tiled_ctl_cap = n;
        }
    }
    TiledGeo tg;
    std::memset(&tg, 0, sizeof tg);
    LaunchGeo &g = tg.g;
    g.frames = d_frames;
    g.frame_stride = frame_stride;
    g.row_stride = row_stride;
    g.frame_index = d_frame_index;
    g.guesses = d_guesses;
    g.resp = d_out_resp;
    g.part_val = t->d_part_val;
    g.part_idx = t->d_part_idx;
    g.part_sec = t->d_part_sec;
    g.part_mask = t->d_part_mask;
    g.ex = exact_ctl(t, kFamFused);
    g.fh = FH; g.fw = FW; g.r1 = t->r1; g.r2 = t->r2; g.n1 = t->n1; g.n2 = t->n2;
    g.L = t->L; g.fill = t->fill; g.nstrips = nsub; g.nslots = nsub; g.n = n; g.nblocks = n * nsub;
    tg.NA = t->n1 + t->L - 1;
    tg.TWin = t->n2 + t->L - 1;
    tg.sn1 = t->tiled_sn1; tg.sn2 = t->tiled_sn2; tg.ns1 = t->tiled_ns1; tg.ns2 = t->tiled_ns2;
    tg.pitchA = t->tiled_c ? fusedc_pitch_a(t->tiled_sn2, t->L) : fused_pitch_a(t->tiled_sn2, t->L);
    tg.pitchV = fused_pitch_v(t->tiled_sn1, t->L);
    tg.cshift = t->tiled_cshift;
    tg.pr = t->tiled_pr;
    tg.pc = t->tiled_pc;
    tg.chain_len = chain_len;
    tg.out_ij = d_out_ij;
    tg.done_flag = d_done_flag;
    tg.done_value = done_value;
    tg.progress = progress ? 1 : 0;
    tg.rp = t->exact ? t->d_rp : nullptr;
    tg.ref_cbw = t->tiled_ref_cbw;
    tg.ref_rows = t->tiled_ref_rows;
    tg.dc_host = dc_host;
    tg.cur = t->d_tiled_ctl;
    tg.sync = reinterpret_cast<unsigned *>(t->d_tiled_ctl + 2 * (size_t)t->tiled_ctl_cap);
    tg.abort = reinterpret_cast<unsigned *>(t->d_tiled_ctl + 4 * (size_t)t->tiled_ctl_cap);
    tg.fault_inject = t->sw.fault_inject ? 1 : 0;
    tg.slots = nullptr;
    tg.tag_base = 0;
    if (chain_len > 1) {
        if (t->tiled_slots_cap < (long long)n * nsub) {
            HIP_TRY(hipStreamSynchronize(t->stream));
            if (t->d_tiled_slots) (void)hipFree(t->d_tiled_slots);
            t->d_tiled_slots = nullptr; t->tiled_slots_cap = 0;
            const size_t bytes = sizeof(unsigned long long) * (size_t)n * 3 * nsub * 2; // two partial sets by frame parity + the V set
            HIP_TRY(hipMalloc(&t->d_tiled_slots, bytes));
            HIP_TRY(hipMemset(t->d_tiled_slots, 0, bytes)); // tag 0 is never a frame's
            t->tiled_slots_cap = (long long)n * nsub;
        }
        tg.slots = t->d_tiled_slots;
        tg.tag_base = t->tiled_tag;
        t->tiled_tag += (unsigned)chain_len + 1u;
    }
    const void *fn = tiled_kernel_for(t, d_out_resp != nullptr);
    const f2 *tr = t->d_taps_row, *tc = t->d_taps_col;
    if (chain_len > 1) {
        void *args[] = {(void *)&tg, (void *)&tr, (void *)&tc};
        const hipError_t e = hipLaunchCooperativeKernel(fn, dim3(n * nsub), dim3(FUSED_NT), args, (unsigned)t->tiled_lds, t->stream);
        if (e != hipSuccess) { (void)hipGetLastError(); return PDOG_OK; } // refused: the caller's other paths
    } else {
        typedef void (*tiled_fn_t)(const TiledGeo, const f2 *, const f2 *);
        hipLaunchKernelGGL((tiled_fn_t)fn, dim3(n * nsub), dim3(FUSED_NT), t->tiled_lds, t->stream, tg, tr, tc);
        HIP_TRY(hipGetLastError());
    }
    *launched = true;
    return PDOG_OK;
}

// Longest kernel with a roll instance (dog_roll.hpp, roll_lengths.def).  Round 2 also built l = 101 / 105: they spilled in
// their loop and measured 3.95 / 4.15 ms per 4096 windows of 257×257 against 3.54 / 3.65 ms for the two-pass kernels, so
// they are gone; longer kernels take the two-pass path.

int choose_variant(pdog_tracker *t, int forced)
{
    const Variant *best = nullptr;
    double best_cost = 0;
    for (int i = 0; i < kNumVariants; ++i) {
        const Variant &v = kVariants[i];
        if (forced >= 0 && v.id != forced) continue;
        if (v.fused) { // never the tracker's batch kernel unless forced; small batches and chains reach it below
            if (forced == v.id && fused_total_lds(t) <= kMaxLds - 1024 && t->n2 + t->L - 1 <= 4 * FUSED_NT && t->fw >= 4) best = &v;
            continue;
        }
        if (v.LT != 0 && v.LT != t->L) continue;
        if (v.lds(t->L) > kMaxLds) continue;
        const bool roll_serves = has_roll_instance(t->L);
        if (forced < 0 && t->L >= ROLL_LMIN && !roll_serves && !v.twopass) continue; // long kernels without a roll instance: two-pass path (measured 1.7× the ring kernel at l = 293)
        if (v.twopass) {
            const size_t hl = (size_t)HP_ROWS * twopass_pitch(t->n1, t->L) * sizeof(f2);
            if (hl > kMaxLds - 1024) continue;
            if (forced < 0 && (roll_serves || t->L < ROLL_LMIN)) continue; // the ring/roll kernels win where a spill-free roll instance exists
            if (!best || forced >= 0) { best = &v; best_cost = 0.0; }
            continue;
        }
        // crude cost: columns computed × per-column efficiency guess; compile-time L wins
        const int strips = (t->n2 + v.tw() - 1) / v.tw();
        double cost = (double)strips * v.tw() * (v.LT ? 1.0 : 1.6);
        if (v.lds(t->L) > kMaxLds / 2) cost *= 1.3; // one workgroup per CU only
        if (v.roll) {
            // barrier-free rolling kernel: ≈2× the ring kernels per column (measured, cfg3); thin
            // remainder columns cost next to nothing
            const int r = t->n2 % v.tw();
            const int cols = (t->n2 > v.tw() && r > 0 && r <= kThinMax) ? t->n2 - r : strips * v.tw();
            cost = 0.5 * cols;
        }
        if (!best || cost < best_cost) { best = &v; best_cost = cost; }
    }
    if (!best) return fail(PDOG_E_ARG, "no kernel specialisation fits this target_width/window (LDS)");
    t->var = best;
    t->nstrips = (t->n2 + best->tw() - 1) / best->tw();
    t->nthin = 0;
    t->thin_x0 = 0;
    t->forced_variant = forced >= 0;
    t->small_twopass = false;
    t->fused_ok = fused_total_lds(t) <= kMaxLds - 1024 && t->n2 + t->L - 1 <= 4 * FUSED_NT && t->fw >= 4;
    if (t->fused_ok) {
        for (bool resp : {false, true}) {
            if (int rc = raise_lds_limit((const void *)fused_kernel_for(t, resp), fused_total_lds(t))) return rc;
        }
    }
    if (int rc = setup_tiled(t)) return rc;
    if (best->fused) { t->nstrips = 1; return PDOG_OK; }
    {
        // The two-pass kernels spread one window over dozens of workgroups, so they win whenever the batch
        // cannot fill the GPU with one wave per strip (single-frame tracking: 36 µs vs 144 µs for one
        // 257×257 window).  Set them up whenever their LDS tiles fit.
        t->tp_ph1 = t->sw.tp_ph1 ? t->sw.tp_ph1 : pick_h1_outputs(t->n2);
        t->tp_php = t->sw.tp_php ? t->sw.tp_php : pick_hpass_outputs(t->n1);
        const size_t hl = (size_t)HP_ROWS * twopass_pitch(t->n1, t->L) * sizeof(f2);
        const size_t h1l = (size_t)HP_ROWS * twopass_pitch_q(t->n2, t->L, 16 * t->tp_ph1) * sizeof(float);
        const size_t hl8 = (size_t)8 * twopass_pitch_q(t->n1, t->L, 32 * t->tp_php) * sizeof(f2);
        if (hl <= kMaxLds - 1024 && h1l <= kMaxLds - 1024 && hl8 <= kMaxLds - 1024) {
            for (const void *f : {(const void *)dog_hpass_kernel<13, 16, false>, (const void *)dog_hpass_kernel<13, 16, true>}) {
                if (int rc = raise_lds_limit(f, hl)) return rc;
            }
            for (bool resp : {false, true})
                for (bool fin : {false, true})
                    if (int rc = raise_lds_limit((const void *)hpass8_kernel_for(t->tp_php, resp, fin, t->sw.hp_u, t->L >= TWOPASS_FLUSH_L), hl8)) return rc;
            for (bool dcin : {false, true})
                if (int rc = raise_lds_limit((const void *)h1_kernel_for(t->tp_ph1, dcin, t->sw.h1_u, t->L >= TWOPASS_FLUSH_L), h1l)) return rc;
            t->small_twopass = true;
        }
    }
    if (best->twopass) {
        t->nstrips = (t->n2 + HP_ROWS - 1) / HP_ROWS; // partial slots = 16-column blocks
        if (!t->small_twopass) return fail(PDOG_E_ARG, "two-pass kernels: the window's rows do not fit LDS");
        return PDOG_OK;
    }
    if (best->roll && best->thin && t->n2 > best->tw()) {
        // width = 64·k + r: r ≤ kThinMax columns are cheaper one by one than as an extra strip
        const int r = t->n2 % best->tw();
        const size_t thin_lds = thin_lds_bytes(t->n1, t->L);
        if (r > 0 && r <= kThinMax && thin_lds <= kMaxLds - 1024) {
            for (kernel_fn f : {best->thin, best->thin_resp}) {
                if (int rc = raise_lds_limit((const void *)f, thin_lds)) return rc;
            }
            t->nthin = r;
            t->thin_x0 = t->n2 - r;
            t->nstrips = t->thin_x0 / best->tw();
        }
    }
    for (kernel_fn f : {best->fn, best->fn_resp}) {
        if (int rc = raise_lds_limit((const void *)f, best->lds(t->L))) return rc;
    }
    return PDOG_OK;
}

int ensure_capacity(pdog_tracker *t, int n)
{
    if (n <= t->cap_windows) return PDOG_OK;
    // worst case strips over all variants so a later pdog_set_variant never reallocates
    int max_strips = 1;
    for (int i = 0; i < kNumVariants; ++i)
        if (!kVariants[i].fused) max_strips = std::max(max_strips, (t->n2 + kVariants[i].tw() - 1) / kVariants[i].tw() + kThinMax);
    max_strips = std::max(max_strips, (t->n2 + 7) / 8);
    if (t->tiled_ok) max_strips = std::max(max_strips, 2 * t->tiled_ns1 * t->tiled_ns2); // the tiled kernel's partials: two sets per window
    for (void *p : {(void *)t->d_part_val, (void *)t->d_part_idx, (void *)t->d_part_sec, (void *)t->d_part_mask})
        if (p) (void)hipFree(p);
    t->d_part_val = t->d_part_sec = nullptr;
    t->d_part_idx = nullptr;
    t->d_part_mask = nullptr;
    t->cap_windows = 0;
    HIP_TRY(hipMalloc(&t->d_part_val, sizeof(float) * (size_t)n * max_strips));
    HIP_TRY(hipMalloc(&t->d_part_sec, sizeof(float) * (size_t)n * max_strips));
    HIP_TRY(hipMalloc(&t->d_part_idx, sizeof(int) * (size_t)n * max_strips));
    HIP_TRY(hipMalloc(&t->d_part_mask, sizeof(unsigned long long) * (size_t)n * max_strips));
    t->cap_windows = n;
    return PDOG_OK;
}

// fused-kernel instance (runtime kernel length)
fused_fn_t fused_kernel_for(const pdog_tracker *t, bool resp)
{
    if (t->fused_c) switch (t->L) {
#define PDOG_LAT_L(LT) case LT: return resp ? (fused_fn_t)dog_fused_kernel<true, 0, LT> : (fused_fn_t)dog_fused_kernel<false, 0, LT>;
#include "lat_lengths.def"
#undef PDOG_LAT_L
        default: break;
    }
    return resp ? (fused_fn_t)dog_fused_kernel<true> : (fused_fn_t)dog_fused_kernel<false>;
}

// Which kernel family a batch of n windows runs on (variant ids: 300 fused, 200 two-pass, otherwise the tracker's
// batch kernel).  A batch of fewer than ≈1000 strip-waves cannot fill 256 CUs × 8 waves with one wave per strip:
// windows that fit in LDS then go to the fused kernel (one workgroup per window, one launch), larger ones to the
// two-pass kernels (dozens of workgroups per window).  pdog_set_variant pins the tracker's kernel.
constexpr int kPathFused = 300, kPathTwoPass = 200, kPathTiled = 400;
int path_for_batch(const pdog_tracker *t, int n)
{
    const Variant &v = *t->var;
    if (v.fused) return kPathFused;
    if (t->forced_variant) return v.twopass ? kPathTwoPass : v.id;
    // (round 3, with the compile-time-l fused instances: 45×45 windows 1024 / 1536 / 2048 per batch: fused 53.8 / 77.3 / 101.9 µs, roll 79.2 / 75.7 / 103.6;
    // 63×63: 80.9 / 117.5 / 155.8 against 91.6 / 88.7 / 117.2)
    // (… and since a workgroup of the fused kernel walks several windows — dispatch, prologue and first tap loads once per workgroup — 45×45 windows:
    // 1024 / 2048 / 4096 per batch 44.5 / 83 / 155 µs against the roll kernel's 79 / 103 / 163: windows below 3000 pixels stay on it at any batch size)
    const bool few = v.twopass ? n <= 256 : ((long long)n * (t->nstrips + (t->nthin ? 1 : 0)) < 1200 || (long long)t->n1 * t->n2 < 3000);
    if (t->sw.tiled_force && t->tiled_ok && n <= t->sw.tiled_batch) return kPathTiled; // experiment switch
    if (few && t->fused_ok) return kPathFused;
    if (t->tiled_ok && n <= t->sw.tiled_batch) return kPathTiled; // one or two windows too large for the fused kernel: one launch (dog_tiled.hpp)
    if (v.twopass || (few && t->small_twopass)) return kPathTwoPass;
    return v.id;
}

// ---- exact mode's FP32 error bounds, per kernel family (dog_exact.hpp: the guarantee) ----
// An FMA chain ŝ_i = fl(ŝ_{i−1} + a_i·b̂_i) satisfies |ŝ_n − s_n| ≤ u·(1 + u)·Σ_i |ŝ_i| (each step rounds its own result once), and
// |ŝ_i| ≤ V·W_i·(1 + nu) where W_i = Σ_{j ≤ i} |b_j|·max|a_j|/V is the cumulative tap weight IN THE ORDER THE KERNEL ADDS THEM.  The
// Gaussians sum to 1, so Σ_i W_i is far below the chain length n that the order-blind bound n·u·V charges: the kernels add the
// smallest taps (the kernel's edges) first.  The factors below are Σ_i W_i evaluated numerically over the tracker's own Float64
// taps for each family's operation order (+1 per rounded tap table, +2 for a final channel addition); δ = u·(V/255)·F·1.02:
//   row pass, symmetric pairs from the edge inwards, centre last (roll, thin, fused, tiled, two-pass):  W_i = Σ_{j ≤ i} 2g[j]
//   row pass, plain chain over the l taps (ring kernels; the refinement's FP32 rescan):               W_i = Σ_{j ≤ i} g[j]
//   column pass, one f32 per output taking (+, −) terms alternately (roll, thin, folded column, rescan): both cumulative weights per step
//   column pass, the two Gaussians in separate chains, added at the end (ring, fused, tiled, two-pass)
//   two-pass: every chain is one trip of the register ring long, the chains' sums are added up (dog_twopass.hpp) — per chain its own
//   cumulative weights from zero, plus the running total's weight per addition
// l = 65: 157 (roll) / 94 (fused, tiled) / 138 (ring) against the order-blind 6l + 4 = 394;  l = 293 two-pass: 72 against 1762.
struct ExactFactors { double sym_int, sym_sep, ring, rescan; };
ExactFactors exact_factors(const std::vector<double> &gp, const std::vector<double> &gm)
{
    const int l = (int)gp.size(), H = l / 2;
    auto row_sym = [&](const std::vector<double> &g) { double W = 0, F = 0; for (int k = 0; k <= H; ++k) { W += (k < H ? 2.0 : 1.0) * g[k]; F += W; } return F + 1.0; };
    auto row_plain = [&](const std::vector<double> &g) { double W = 0, F = 0; for (int k = 0; k < l; ++k) { W += g[k]; F += W; } return F + 1.0; };
    double Gp = 0, Gm = 0, c_sep = 4.0, c_int = 2.0;
    for (int t = 0; t < l; ++t) {
        Gp += gp[t];
        c_int += Gp + Gm; // after the + term of tap t
        Gm += gm[t];
        c_int += Gp + Gm; // after the − term
        c_sep += Gp + Gm;
    }
    ExactFactors f;
    f.sym_int = row_sym(gp) + row_sym(gm) + c_int;
    f.sym_sep = row_sym(gp) + row_sym(gm) + c_sep;
    f.ring = row_plain(gp) + row_plain(gm) + c_sep;
    f.rescan = row_plain(gp) + row_plain(gm) + c_int;
    return f;
}
// the two-pass kernels' factor for the tracker's task sizes (hr8: the 8-row column-pass kernels, else the 16-row form <13, 16>)
double twopass_factor(const pdog_tracker *t, bool hr8)
{
    if (!hr8 || t->L < TWOPASS_FLUSH_L) return t->F_sym_sep; // the plain instances: symmetric row pass, separate column chains
    const int l = t->L, H = l / 2;
    const int U1 = t->sw.h1_u, m_r = twopass_ring(t->tp_ph1, U1), U2 = hr8 ? t->sw.hp_u : 16, m_c = hr8 ? twopass_ring(t->tp_php, U2) : twopass_ring(13, 16);
    auto blocked = [](const std::vector<double> &terms, int m, int n_full) { // chains of m terms (n_full of them), then ONE chain with the rest
        double F = 0, total = 0;
        const int n = (int)terms.size();
        for (int c = 0, a = 0; c <= n_full; ++c) {
            const int b = c < n_full ? a + m : n;
            double w = 0;
            for (int i = a; i < b; ++i) { w += terms[i]; F += w; } // the chain's own running sums, from zero
            total += w;
            F += total;                                             // the addition that takes the chain's sum into the running total
            a = b;
        }
        return F;
    };
    auto row = [&](const std::vector<double> &g) {
        std::vector<double> terms(H + 1);
        for (int k = 0; k <= H; ++k) terms[k] = (k < H ? 2.0 : 1.0) * g[k];
        return blocked(terms, m_r, (H / U1) / (m_r / U1)) + 1.0;
    };
    auto col = [&](const std::vector<double> &g) { return blocked(g, m_c, ((l + U2 - 1) / U2) / (m_c / U2)); };
    return row(t->h_gp) + row(t->h_gm) + col(t->h_gp) + col(t->h_gm) + 4.0;
}
ExactCtl exact_ctl(const pdog_tracker *t, KernelFamily fam)
{
    ExactCtl x;
    x.stat = t->d_ref_stat;
    x.range_err = t->d_mail_map ? t->d_mail_map + 5 : nullptr;
    const double F = fam == kFamRoll ? t->F_sym_int : fam == kFamRing ? t->F_ring : fam == kFamFused ? t->F_sym_sep : twopass_factor(t, fam == kFamTwoPass8);
    const double u = std::ldexp(1.0, -24);
    x.T = t->exact_all ? __builtin_huge_valf() : std::nextafter((float)(2.0 * u * F * 1.02 + 2e-9), 1.0f);
    x.T_rescan = t->exact_all ? __builtin_huge_valf() : std::nextafter((float)(u * (F + t->F_rescan) * 1.02 + 2e-9), 1.0f);
    return x;
}

// The last kernel of a batch (dog_exact.hpp): strip combine, index map and clamp (:58-61) for every window — one
// workgroup each — and the refinement of exact mode for the windows that need it.  `slot_w`, `slot_last`: how the
// partial slots the main kernels wrote map to window columns (the refinement only rescans column blocks whose slot
// can hold a near-maximal pixel).  With done_flag set the kernel also publishes the host functor's ticket with
// window 0's final answer.
int launch_finish(pdog_tracker *t, const LaunchGeo &g, int slot_w, int slot_last, int32_t *d_out_ij, int32_t *d_done_flag, int32_t done_value,
                  bool use_mask = false, const float *map = nullptr, const int *vmax = nullptr)
{
    FinishGeo fg;
    fg.g = g;
    fg.map = map;
    fg.vmax = vmax;
    fg.K64 = t->exact ? t->d_K64 : nullptr;
    fg.g64 = t->d_g64;
    fg.dir = t->darker ? -1.0 : 1.0;
    fg.T64 = t->exact_T64;
    fg.cbw = t->ref_cbw;
    fg.tile_rows = t->ref_rows;
    fg.slot_w = slot_w;
    fg.slot_last = slot_last;
    fg.nmain = g.nslots - g.nthin;
    fg.thin_x0 = g.thin_x0;
    fg.use_mask = use_mask ? 1 : 0;
    fg.v_after = t->sw.v_after;
    fg.out_ij = d_out_ij;
    fg.done_flag = d_done_flag;
    fg.done_value = done_value;
    fg.seq_windows = (int)t->win_launched;
    t->win_launched += (unsigned)g.n;
    size_t lds = t->exact ? refine_lds_bytes(t->n1, t->L, t->ref_cbw, t->ref_rows) : 0;
    if (g.fold_r) { // the folded remainder column's R values, one slice per wave (a window each)
        lds = std::max(lds, (size_t)FINISH_WPB * (t->n1 + t->L - 1 + FOLD_GO) * sizeof(f2));
        if (int rc = raise_lds_limit((const void *)dog_finish_kernel, lds)) return rc;
    }
    hipLaunchKernelGGL(dog_finish_kernel, dim3((g.n + FINISH_WPB - 1) / FINISH_WPB), dim3(REFINE_NT), lds, t->stream, fg, (const f2 *)t->d_taps_row,
                       (const f2 *)t->d_taps_col);
    HIP_TRY(hipGetLastError());
    return PDOG_OK;
}

// One workgroup per window (chain_len = 1) or per clip (chain_len frames, frame k > 0 starts at frame k−1's answer).
int launch_fused(pdog_tracker *t, const uint8_t *d_frames, int64_t frame_stride, int64_t row_stride,
                 const int32_t *d_frame_index, const int32_t *d_guesses, int n, int chain_len, int32_t *d_out_ij,
                 float *d_out_resp, int FH, int FW, int32_t *d_done_flag = nullptr, int32_t done_value = 0, bool progress = false, int dc_host = -1)
{
    FusedGeo fg;
    LaunchGeo &g = fg.g;
    std::memset(&g, 0, sizeof g);
    g.frames = d_frames;
    g.frame_stride = frame_stride;
    g.row_stride = row_stride;
    g.frame_index = d_frame_index;
    g.guesses = d_guesses;
    g.resp = d_out_resp;
    g.fh = FH; g.fw = FW; g.r1 = t->r1; g.r2 = t->r2; g.n1 = t->n1; g.n2 = t->n2;
    g.L = t->L; g.fill = t->fill; g.nstrips = 1; g.n = n; g.nslots = 1; g.nblocks = n;
    fg.NA = t->n1 + t->L - 1;
    fg.TWin = t->n2 + t->L - 1;
    fg.pitchA = t->fused_c ? fusedc_pitch_a(t->n2, t->L) : fused_pitch_a(t->n2, t->L);
    fg.pitchV = fused_pitch_v(t->n1, t->L);
    fg.cshift = 0;
    while ((1 << fg.cshift) < (fg.TWin + 3) / 4) ++fg.cshift;
    // outputs per task: fewest rounds of 1024 tasks, then least work per task (≈ P outputs + 2 of fixed cost)
    auto pick = [](int lines, int nout, std::initializer_list<int> ps, double fixed) {
        int best = 0;
        double best_cost = 0;
        for (int p : ps) {
            const long long tasks = (long long)lines * ((nout + p - 1) / p);
            const double cost = (double)((tasks + FUSED_NT - 1) / FUSED_NT) * (p + fixed);
            if (!best || cost < best_cost) { best = p; best_cost = cost; }
        }
        return best;
    };
    fg.pr = pick(fg.NA, t->n2, {3, 4, 5, 6, 8}, 2.0);
    fg.pc = pick(t->n2, t->n1, {2, 3, 4, 6, 8}, 1.5);
    if (t->fused_c) fg.pr = fusedc_row_outputs(fg.NA, t->n2);
    if (t->sw.fused_pr) { fg.pr = t->sw.fused_pr; fg.pc = t->sw.fused_pc; } // tuning switch PDOG_FUSED_P
    fg.chain_len = chain_len;
    fg.out_ij = d_out_ij;
    fg.done_flag = d_done_flag;
    fg.done_value = done_value;
    fg.progress = progress ? 1 : 0;
    fg.rp = t->exact ? t->d_rp : nullptr;
    fg.ref_cbw = t->fused_ref_cbw;
    fg.ref_rows = t->fused_ref_rows;
    fg.dc_host = dc_host;
    g.ex = exact_ctl(t, kFamFused);
    const size_t lds = fused_total_lds(t);
    typedef fused_fn_t fused_fn;
    fused_fn fn = fused_kernel_for(t, d_out_resp != nullptr);
#ifdef PDOG_ABLATIONS
    if (d_out_resp && t->sw.fused_diag) { // phase stamps instead of the response (tools/fused_phases.py)
        fn = (t->fused_c && t->L == 65) ? (fused_fn)dog_fused_kernel<true, 1, 65> : (fused_fn)dog_fused_kernel<true, 1>;
        if (int rc = raise_lds_limit((const void *)fn, lds)) return rc;
    }
#endif
#ifdef PDOG_ABLATIONS
    if (!d_out_resp && t->sw.fused_diag && chain_len > 8) { // a chain's steady state: stamps of every frame, the median printed per phase
        float *d_st = nullptr;
        HIP_TRY(hipMalloc(&d_st, sizeof(float) * (16 * chain_len + 64)));
        fg.g.resp = d_st;
        fn = (t->fused_c && t->L == 65) ? (fused_fn)dog_fused_kernel<true, 1, 65> : (fused_fn)dog_fused_kernel<true, 1>;
        if (int rc = raise_lds_limit((const void *)fn, lds)) return rc;
        hipLaunchKernelGGL(fn, dim3(n), dim3(FUSED_NT), lds, t->stream, fg, (const f2 *)t->d_taps_row, (const f2 *)t->d_taps_col);
        HIP_TRY(hipStreamSynchronize(t->stream));
        std::vector<float> st(16 * (size_t)chain_len + 64);
        HIP_TRY(hipMemcpy(st.data(), d_st, st.size() * sizeof(float), hipMemcpyDeviceToHost));
        HIP_TRY(hipFree(d_st));
        static const int order[7] = {0, 4, 1, 2, 5, 6, 3};
        static const char *names[7] = {"samples", "barrier", "staged", "row pass", "col+peak(w0)", "barrier", "finalize"};
        std::fprintf(stderr, "fused chain phases (median over frames 8.., 100 MHz ticks → us):");
        int prev = -1;
        for (int q = 0; q < 7; ++q) {
            std::vector<float> d;
            for (int k = 8; k < chain_len; ++k) d.push_back(st[16 * k + 8 + order[q]] - (prev < 0 ? 0.f : st[16 * k + 8 + prev]));
            std::sort(d.begin(), d.end());
            std::fprintf(stderr, " %s %.2f;", names[q], d[d.size() / 2] / 100.0);
            prev = order[q];
        }
        std::fprintf(stderr, "\n");
        static const char *wn[4] = {"row pass done", "col tasks done", "wave peak done", "staged"};
        for (int i : {3, 0, 1, 2}) {
            std::fprintf(stderr, "  frame 20, cycles since frame start, %s, waves 0-15:", wn[i]);
            for (int w = 0; w < 16; ++w) std::fprintf(stderr, " %.0f", st[16 * (size_t)chain_len + 16 * i + w]);
            std::fprintf(stderr, "\n");
        }
        return PDOG_OK;
    }
#endif
    if (t->fused_resident <= 0) { // once per tracker
        int per_cu = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)fn, FUSED_NT, lds) != hipSuccess || per_cu < 1) per_cu = 1;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, t->device) != hipSuccess || cus < 1) cus = 256;
        t->fused_resident = per_cu * cus;
    }
    hipLaunchKernelGGL(fn, dim3(std::min(n, t->fused_resident)), dim3(FUSED_NT), lds, t->stream, fg, (const f2 *)t->d_taps_row, (const f2 *)t->d_taps_col);
    HIP_TRY(hipGetLastError());
    return PDOG_OK;
}

int launch_detect(pdog_tracker *t, const uint8_t *d_frames, int64_t frame_stride, int64_t row_stride,
                  const int32_t *d_frame_index, const int32_t *d_guesses, int n, int32_t *d_out_ij,
                  float *d_out_resp, int fh_override = 0, int fw_override = 0, int32_t *d_done_flag = nullptr,
                  int32_t done_value = 0, bool *ticket_armed = nullptr, int dc_host = -1)
{
    const Variant &v = *t->var;
    if (ticket_armed) *ticket_armed = false; // set where the kernels that run will publish done_value (single-window paths)
    // frames of another size than the tracker's (the packed window tiles of pdog_detect_batch_host)
    const int FH = fh_override ? fh_override : t->fh, FW = fw_override ? fw_override : t->fw;
    LaunchGeo g;
    g.fold_r = nullptr;
    g.frames = d_frames;
    g.frame_stride = frame_stride;
    g.row_stride = row_stride;
    g.frame_index = d_frame_index;
    g.guesses = d_guesses;
    g.resp = d_out_resp;
    g.part_val = t->d_part_val;
    g.part_idx = t->d_part_idx;
    g.part_sec = t->d_part_sec;
    g.part_mask = t->d_part_mask;
    g.ex = exact_ctl(t, (v.roll ? kFamRoll : kFamRing));
    g.fh = FH; g.fw = FW; g.r1 = t->r1; g.r2 = t->r2; g.n1 = t->n1; g.n2 = t->n2;
    g.L = t->L; g.fill = t->fill; g.nstrips = t->nstrips; g.n = n;
    g.RR = v.ring(t->L);
    g.pitchA = v.pa(t->L);
    g.nblocks = n * t->nstrips;
    g.nslots = t->nstrips + t->nthin;
    g.thin_x0 = t->thin_x0;
    g.nthin = t->nthin;
    // windows that fit in LDS, in batches too small to fill the GPU any other way: one workgroup per window, one launch
    const int path = path_for_batch(t, n);
    if (path == kPathFused) {
        if (ticket_armed) *ticket_armed = d_done_flag != nullptr;
        return launch_fused(t, d_frames, frame_stride, row_stride, d_frame_index, d_guesses, n, 1, d_out_ij, d_out_resp, FH, FW,
                            d_done_flag, done_value, false, n == 1 ? dc_host : -1);
    }
    // one or a few windows too large for the fused kernel: the tiled kernel, one launch (dog_tiled.hpp)
    if (path == kPathTiled) {
        bool launched = false;
        if (int rc = launch_tiled(t, d_frames, frame_stride, row_stride, d_frame_index, d_guesses, n, 1, d_out_ij, d_out_resp, FH, FW,
                                  d_done_flag, done_value, false, &launched, n == 1 ? dc_host : -1)) return rc;
        if (launched) {
            if (ticket_armed) *ticket_armed = d_done_flag != nullptr;
            return PDOG_OK;
        }
    }
    // small batches: fewer than ≈1000 strip-waves cannot fill 256 CUs × 8 waves; the two-pass kernels can
    if (path == kPathTwoPass || path == kPathTiled) {
        const int hr = t->sw.hpass16 ? HP_ROWS : 8; // 8 RT rows per workgroup (32 KB LDS → 4 workgroups per CU): +3 % on cfg5 vs 16; env = tuning switch
        const int tp_slots = (t->n2 + hr - 1) / hr; // partial slots = hr-column blocks
        g.nstrips = tp_slots;
        g.nslots = tp_slots;
        g.nthin = 0;
        g.ex = exact_ctl(t, hr == 8 ? kFamTwoPass8 : kFamTwoPass16); // the two-pass kernels' own bound (blocked accumulation)
        // exact mode: the column pass also writes its responses (4 B per pixel), so that a window that needs the
        // refinement — every window, at the σ this path serves — reads its candidates off the map instead of recomputing them
        const float *map = d_out_resp;
        if (t->exact && !map) {
            const size_t need = sizeof(float) * (size_t)n * t->n1 * t->n2;
            if (need <= t->sw.map_cap) {
                if (t->map_bytes < need) {
                    HIP_TRY(hipStreamSynchronize(t->stream));
                    if (t->d_map) (void)hipFree(t->d_map);
                    t->d_map = nullptr; t->map_bytes = 0;
                    HIP_TRY(hipMalloc(&t->d_map, need));
                    t->map_bytes = need;
                }
                map = t->d_map;
                g.resp = t->d_map;
            }
        }
        const bool want_resp = g.resp != nullptr;
        TwoPassGeo tg;
        tg.g = g;
        tg.TWin = t->n2 + t->L - 1;
        tg.NA = t->n1 + t->L - 1;
        tg.h1blocks_per_win = (tg.NA + HP_ROWS - 1) / HP_ROWS;
        tg.hblocks_per_win = tp_slots;
        tg.pitchA = twopass_pitch_q(t->n2, t->L, 16 * t->tp_ph1);
        tg.pitchV = hr == 8 ? twopass_pitch_q(t->n1, t->L, 32 * t->tp_php) : twopass_pitch(t->n1, t->L, hr);
        const size_t per_win = (size_t)t->n2 * tg.NA * sizeof(f2);
        const size_t cap = t->sw.scratch_cap; // HBM scratch for the transposed intermediate; larger batches go in chunks
        const int chunk = (int)std::max<size_t>(1, std::min<size_t>((size_t)n, cap / per_win));
        if (t->v_bytes < per_win * chunk || t->dc_cap < n) {
            HIP_TRY(hipStreamSynchronize(t->stream));
            if (t->v_bytes < per_win * chunk) {
                if (t->d_V) (void)hipFree(t->d_V);
                t->d_V = nullptr; t->v_bytes = 0;
                HIP_TRY(hipMalloc(&t->d_V, per_win * chunk));
                t->v_bytes = per_win * chunk;
            }
            if (t->dc_cap < n) {
                if (t->d_dc) (void)hipFree(t->d_dc);
                t->d_dc = nullptr; t->dc_cap = 0;
                HIP_TRY(hipMalloc(&t->d_dc, sizeof(int) * 2 * (size_t)n)); // [n] DC levels, then [n] the windows' own V (exact mode)
                t->dc_cap = n;
            }
        }
        tg.RT = t->d_V;
        tg.dc = t->d_dc;
        tg.vmax = nullptr;
        tg.counter = nullptr;
        tg.out_ij = d_out_ij;
        tg.done_flag = nullptr;
        tg.done_value = 0;
        const size_t l1 = (size_t)HP_ROWS * tg.pitchA * sizeof(float);
        const size_t l2 = (size_t)hr * tg.pitchV * sizeof(f2);
        // A handful of windows (single-clip chains and functor calls with windows too large for the fused kernel,
        // the auto-detect pass): launches are what such a batch costs, so the DC level is derived inside the row pass
        // and the last column-pass workgroup of a window combines its partials — two launches instead of four.
        constexpr int kLowLatMax = 16; // measured crossover (257×257 and 271×481 windows): 2 launches win up to 16 windows, 4 launches beyond
        // (exact mode: the partials are combined by dog_finish_kernel, which also refines and publishes the ticket —
        // three launches; without it the last column-pass workgroup of a window combines them itself — two)
        const bool lowlat = n <= kLowLatMax && n <= chunk && hr == 8 && !t->sw.twopass_4l;
        if (lowlat && t->exact) {
            tg.win0 = 0;
            if (ticket_armed) *ticket_armed = d_done_flag != nullptr;
            hipLaunchKernelGGL(h1_kernel_for(t->tp_ph1, true, t->sw.h1_u, t->L >= TWOPASS_FLUSH_L), dim3(n * tg.h1blocks_per_win), dim3(256), l1, t->stream, tg, (const f2 *)t->d_taps_row);
            HIP_TRY(hipGetLastError());
            hipLaunchKernelGGL(hpass8_kernel_for(t->tp_php, want_resp, false, t->sw.hp_u, t->L >= TWOPASS_FLUSH_L), dim3(n * tg.hblocks_per_win), dim3(256), l2, t->stream, tg, (const f2 *)t->d_taps_col);
            HIP_TRY(hipGetLastError());
            return launch_finish(t, g, hr, 1 << 30, d_out_ij, d_done_flag, done_value, false, t->exact ? map : nullptr);
        }
        if (lowlat) {
            if (!t->d_counter) {
                HIP_TRY(hipMalloc(&t->d_counter, sizeof(int) * kLowLatMax));
                HIP_TRY(hipMemsetAsync(t->d_counter, 0, sizeof(int) * kLowLatMax, t->stream));
            }
            tg.counter = t->d_counter;
            tg.win0 = 0;
            tg.done_flag = d_done_flag;
            tg.done_value = done_value;
            if (ticket_armed) *ticket_armed = d_done_flag != nullptr;
            hipLaunchKernelGGL(h1_kernel_for(t->tp_ph1, true, t->sw.h1_u, t->L >= TWOPASS_FLUSH_L), dim3(n * tg.h1blocks_per_win), dim3(256), l1, t->stream, tg, (const f2 *)t->d_taps_row);
            HIP_TRY(hipGetLastError());
            hipLaunchKernelGGL(hpass8_kernel_for(t->tp_php, want_resp, true, t->sw.hp_u, t->L >= TWOPASS_FLUSH_L), dim3(n * tg.hblocks_per_win), dim3(256), l2, t->stream, tg, (const f2 *)t->d_taps_col);
            HIP_TRY(hipGetLastError());
            return PDOG_OK;
        }
        // exact mode: the row pass collects each window's own V = max |pixel − dc| (the error bound is proportional to it) and the
        // finishing kernel flags with it; the two-pass kernels' own bound (blocked accumulation) replaces the one-chain bound
        if (t->exact && !t->exact_all && hr == 8 && t->L >= TWOPASS_FLUSH_L) tg.vmax = t->d_dc + t->dc_cap; // (the blocked-accumulation instances collect it)
        hipLaunchKernelGGL(dog_dc_kernel, dim3(n), dim3(64), 0, t->stream, g, t->d_dc, tg.vmax);
        HIP_TRY(hipGetLastError());
        for (int w0 = 0; w0 < n; w0 += chunk) {
            const int nw = std::min(chunk, n - w0);
            tg.win0 = w0;
            hipLaunchKernelGGL(h1_kernel_for(t->tp_ph1, false, t->sw.h1_u, t->L >= TWOPASS_FLUSH_L), dim3(nw * tg.h1blocks_per_win), dim3(256), l1, t->stream, tg, (const f2 *)t->d_taps_row);
            HIP_TRY(hipGetLastError());
            if (hr == 8) {
                hipLaunchKernelGGL(hpass8_kernel_for(t->tp_php, want_resp, false, t->sw.hp_u, t->L >= TWOPASS_FLUSH_L), dim3(nw * tg.hblocks_per_win), dim3(256), l2, t->stream, tg, (const f2 *)t->d_taps_col);
            } else if (want_resp)
                hipLaunchKernelGGL((dog_hpass_kernel<13, 16, true>), dim3(nw * tg.hblocks_per_win), dim3(256), l2, t->stream, tg, (const f2 *)t->d_taps_col);
            else
                hipLaunchKernelGGL((dog_hpass_kernel<13, 16, false>), dim3(nw * tg.hblocks_per_win), dim3(256), l2, t->stream, tg, (const f2 *)t->d_taps_col);
            HIP_TRY(hipGetLastError());
        }
        return launch_finish(t, g, hr, 1 << 30, d_out_ij, nullptr, 0, false, t->exact ? map : nullptr, tg.vmax);
    }
    const int grid = round_up(g.nblocks, 8);
    // Exact mode: a batch whose predecessors flagged more than 8 % of their windows writes its responses (the kernels'
    // RESP instances: +15 % on the strips) and the finishing kernel reads the candidates off that map instead of recomputing
    // them per window; back to the plain instances after eight observations below 2 %.  The finishing kernels publish the
    // flagged count TOGETHER with the number of windows it came from, through host-coherent memory: nothing here waits for
    // the GPU, and the rate is right however many batches are in flight (round 2 compared the count's increase with ONE
    // batch's size: with three batches in flight 0.9 % looked like 2.7 % and one step in four paid for a map it did not need).
    const float *map = d_out_resp;
    if (t->exact && !t->exact_all) {
        const unsigned cur = (unsigned)__atomic_load_n(&t->h_pinned[6], __ATOMIC_ACQUIRE); // (the low 32 bits of a cumulative count: differences wrap correctly)
        const unsigned curw = (unsigned)__atomic_load_n(&t->h_pinned[7], __ATOMIC_ACQUIRE);
        const long long delta = (long long)(unsigned)(cur - t->flag_last), dwin = (long long)(unsigned)(curw - t->win_last);
        if (dwin > 0) {
            t->flag_last = cur;
            t->win_last = curw;
            if (delta * 12 > dwin) { t->roll_map = true; t->flag_calm = 0; }
            else if (delta * 50 < dwin) { if (++t->flag_calm >= 8) t->roll_map = false; }
            else t->flag_calm = 0;
        }
        const size_t need = sizeof(float) * (size_t)n * t->n1 * t->n2;
        if (!map && t->roll_map && !t->sw.no_roll_map && need <= t->sw.map_cap) {
            if (t->map_bytes < need) {
                HIP_TRY(hipStreamSynchronize(t->stream));
                if (t->d_map) (void)hipFree(t->d_map);
                t->d_map = nullptr; t->map_bytes = 0;
                HIP_TRY(hipMalloc(&t->d_map, need));
                t->map_bytes = need;
            }
            map = t->d_map;
            g.resp = t->d_map;
        }
    }
    const bool want_resp = g.resp != nullptr;
    // A single remainder column (widths 64·k + 1: 257, 513, …) can be FOLDED into the last strip: its row pass rides in that
    // strip (a ninth output in the last lane group), its R values go through d_fold_r to the finishing kernel, which runs the
    // column's column pass — no second kernel re-reading a 65-column patch per window, no side stream.  The response-writing
    // instances (parity checks, the response-map refinement) and the other kernel lengths keep dog_thin_kernel.
    const int NA = t->n1 + t->L - 1;
    const size_t fold_wave_lds = (size_t)(NA + FOLD_GO) * sizeof(f2);
    // Measured (same-session A/B against dog_thin_kernel beside the strips): the folded strip is the slowest of its window (+6 %:
    // 49 packed instructions per sub-chunk) and the finishing kernel gains the column pass — +2.7 % per cfg3 step (4 strips per
    // window), −0.8 % per cfg4 step (8 strips).  A variant that kept the R column in the strip's LDS and ran the column pass in
    // the strip's own wave cost a wave per SIMD (12.9 KB per wave) and +5 %.  So: folded from 8 strips per window on.
    const bool fold = v.roll && roll_folds(v.LT) && t->nthin == 1 && !want_resp && !t->sw.no_fold && (t->nstrips >= 8 || t->sw.fold_always) &&
                      (size_t)FINISH_WPB * fold_wave_lds <= kMaxLds - 1024;
    if (fold) {
        const size_t need = sizeof(f2) * (size_t)n * NA;
        if (t->fold_bytes < need) {
            HIP_TRY(hipStreamSynchronize(t->stream));
            if (t->d_fold_r) (void)hipFree(t->d_fold_r);
            t->d_fold_r = nullptr; t->fold_bytes = 0;
            HIP_TRY(hipMalloc(&t->d_fold_r, need));
            t->fold_bytes = need;
        }
        g.fold_r = t->d_fold_r;
    } else if (t->nthin) {
        // fork: the thin kernel only reads the frames and writes its own partial slots
        HIP_TRY(hipEventRecord(t->ev_fork, t->stream));
        HIP_TRY(hipStreamWaitEvent(t->aux_stream, t->ev_fork, 0));
        const size_t thin_lds = thin_lds_bytes(t->n1, t->L);
        hipLaunchKernelGGL(want_resp ? v.thin_resp : v.thin, dim3(round_up(n * t->nthin, 8)), dim3(256), thin_lds, t->aux_stream, g,
                           (const f2 *)t->d_taps_row, (const f2 *)t->d_taps_col);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(t->ev_join, t->aux_stream));
    }
    size_t lds_bytes = v.lds(t->L);
#ifdef PDOG_ABLATIONS
    if (t->sw.lds_pad) { // occupancy experiments: extra LDS per workgroup
        lds_bytes += (size_t)t->sw.lds_pad;
        (void)raise_lds_limit((const void *)(want_resp ? v.fn_resp : v.fn), lds_bytes);
    }
#endif
    kernel_fn fn = want_resp ? v.fn_resp : v.fn;
    if (!want_resp && v.roll && v.LT == 65 && v.id == 100) { // instances with statically shortened epilogue bodies for the common window heights
        const int cls = roll_epi_class(t->n1, 65);
#define PDOG_EPI_PICK(C) if (cls == C) fn = (kernel_fn)dog_roll_kernel<65, false, 0, C>;
        PDOG_EPI_CLASSES(PDOG_EPI_PICK)
#undef PDOG_EPI_PICK
    }
    hipLaunchKernelGGL(fn, dim3(grid), dim3(v.NT), lds_bytes, t->stream, g,
                       (const f2 *)t->d_taps_row, (const f2 *)(v.roll ? t->d_taps_roll : t->d_taps_col));
    HIP_TRY(hipGetLastError());
    if (t->nthin && !fold) HIP_TRY(hipStreamWaitEvent(t->stream, t->ev_join, 0)); // join before the strip combine
    // roll: 64-column strips over the first `covered` columns, the last one shifted left to stay inside; ring: tw() columns each
    const int covered = t->nthin ? t->thin_x0 : t->n2;
    return launch_finish(t, g, v.tw(), v.roll ? std::max(0, covered - v.tw()) : (1 << 30), d_out_ij, nullptr, 0, v.roll, t->exact ? map : nullptr);
}

} // namespace

extern "C" __attribute__((visibility("hidden"))) void pdog_set_error_text(const char *msg) { g_err = msg ? msg : ""; }

extern "C" {

int pdog_abi_version(void) { return PDOG_ABI_VERSION; }
const char *pdog_last_error(void) { return g_err.c_str(); }

double pdog_sigma(double target_width) { return sigma_of(target_width); }
int pdog_default_window(double target_width) { return 4 * (int)std::ceil(sigma_of(target_width)) + 1; } // :64-68
int pdog_kernel_len(double target_width) { return kernel_len_of_sigma(sigma_of(target_width)); }

int pdog_gaussian_taps(double target_width, int which, double *out, int cap)
{
    if (!out || (which != 0 && which != 1) || !(target_width > 0)) return fail(PDOG_E_ARG, "pdog_gaussian_taps: bad argument");
    const double s = sigma_of(target_width);
    const int l = kernel_len_of_sigma(s);
    if (cap < l) return fail(PDOG_E_ARG, "pdog_gaussian_taps: buffer too small");
    gaussian_1d(which ? s * std::sqrt(2.0) : s, l, out);
    return PDOG_OK;
}

int pdog_dense_kernel(double target_width, int darker_target, double *out, int cap)
{
    if (!out || !(target_width > 0)) return fail(PDOG_E_ARG, "pdog_dense_kernel: bad argument");
    const double s = sigma_of(target_width);
    const int l = kernel_len_of_sigma(s);
    if ((long long)cap < (long long)l * l) return fail(PDOG_E_ARG, "pdog_dense_kernel: buffer too small");
    std::vector<double> gp(l), gm(l);
    gaussian_1d(s, l, gp.data());
    gaussian_1d(s * std::sqrt(2.0), l, gm.data());
    dense_dog_kernel(gp.data(), gm.data(), l, darker_target != 0, out);
    return PDOG_OK;
}

int pdog_mode_u8(const uint8_t *img, int h, int w, int64_t row_stride, int *out_mode)
{
    if (!img || !out_mode || h <= 0 || w <= 0 || row_stride < w) return fail(PDOG_E_ARG, "pdog_mode_u8: bad argument");
    // StatsBase.mode over the h×w view, column-major scan (row index fastest): per-value
    // running counts; the winner is the value whose count FIRST exceeds the running maximum.
    // Equivalent single pass per column block: counts are order dependent only through ties,
    // so keep the literal scan order.
    int64_t cnt[256];
    std::memset(cnt, 0, sizeof cnt);
    int64_t mc = 0;
    int mv = img[0];
    for (int j = 0; j < w; ++j) {
        const uint8_t *p = img + j;
        for (int i = 0; i < h; ++i) {
            const int v = p[(int64_t)i * row_stride];
            const int64_t c = ++cnt[v];
            if (c > mc) { mc = c; mv = v; }
        }
    }
    *out_mode = mv;
    return PDOG_OK;
}

int pdog_mode_u8_device(int device, const uint8_t *d_img, int h, int w, int64_t row_stride, void *hip_stream, int *out_mode)
{
    if (!d_img || !out_mode || h <= 0 || w <= 0 || row_stride < w || (long long)h * w >= 0xffffffffLL)
        return fail(PDOG_E_ARG, "pdog_mode_u8_device: bad argument");
    HIP_TRY(hipSetDevice(device));
    hipStream_t stream = (hipStream_t)hip_stream;
    unsigned *d_tab = nullptr;
    HIP_TRY(hipMalloc(&d_tab, sizeof(unsigned) * 512));
    unsigned tab[512];
    hipError_t e = hipMemsetAsync(d_tab, 0, sizeof(unsigned) * 512, stream);
    if (e == hipSuccess) {
        const int blocks = (int)std::min<long long>(1024, ((long long)h * w + 255) / 256);
        hipLaunchKernelGGL(dog_mode_kernel, dim3(blocks), dim3(256), 0, stream, d_img, h, w, (long long)row_stride, d_tab, d_tab + 256);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(tab, d_tab, sizeof tab, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    (void)hipFree(d_tab);
    if (e != hipSuccess) return fail(PDOG_E_HIP, std::string("pdog_mode_u8_device: ") + hipGetErrorString(e));
    int best = 0;
    for (int v = 1; v < 256; ++v)
        if (tab[v] > tab[best] || (tab[v] == tab[best] && tab[256 + v] < tab[256 + best])) best = v;
    *out_mode = best;
    return PDOG_OK;
}

int pdog_create(int device, int frame_h, int frame_w, double target_width, int win_h, int win_w,
                int darker_target, int fill, pdog_tracker **out)
{
    if (!out) return fail(PDOG_E_ARG, "pdog_create: out is null");
    *out = nullptr;
    if (frame_h <= 0 || frame_w <= 0 || !(target_width > 0) || win_h < 0 || win_w < 0 || fill < 0 || fill > 255)
        return fail(PDOG_E_ARG, "pdog_create: bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(PDOG_E_NODEV, "pdog_create: no HIP device (this library has no CPU path)");
    if (device < 0 || device >= ndev) return fail(PDOG_E_ARG, "pdog_create: device ordinal out of range");
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(PDOG_E_NODEV, std::string("pdog_create: device is ") + prop.gcnArchName + ", kernels are built for gfx950 only");
    HIP_TRY(hipSetDevice(device));

    pdog_tracker *t = new pdog_tracker();
    t->device = device;
    t->sw = read_switches();
    t->fh = frame_h; t->fw = frame_w;
    t->tw = target_width;
    t->sigma = sigma_of(target_width);             // :41
    t->darker = darker_target ? 1 : 0;             // :42
    t->L = kernel_len_of_sigma(t->sigma);          // :43
    t->r1 = win_h / 2; t->r2 = win_w / 2;          // :44
    t->n1 = 2 * t->r1 + 1; t->n2 = 2 * t->r2 + 1;  // :56
    t->fill = fill;                                // :47
    if ((long long)t->n1 * t->n2 > 0x3fffffffLL) { delete t; return fail(PDOG_E_ARG, "pdog_create: window too large"); }

    setup_refine_geometry(t);
    int rc = choose_variant(t, -1);
    if (rc) { delete t; return rc; }

    // taps: Float64 on the host, one rounding to f32
    std::vector<double> gp(t->L), gm(t->L);
    gaussian_1d(t->sigma, t->L, gp.data());
    gaussian_1d(t->sigma * std::sqrt(2.0), t->L, gm.data());
    const double s = (t->darker ? -1.0 : 1.0) / 255.0; // direction (:42) and N0f8 scale
    constexpr int kTapPad = 16; // zero taps past the end: kernels that request taps one block ahead read them
    std::vector<f2> tr(t->L + kTapPad, f2{0.f, 0.f}), tc(t->L + kTapPad, f2{0.f, 0.f});
    for (int k = 0; k < t->L; ++k) {
        tr[k] = f2{(float)gp[k], (float)gm[k]};
        tc[k] = f2{(float)(s * gp[k]), (float)(-s * gm[k])};
    }
#define CREATE_TRY(expr)                                                                          \
    do {                                                                                          \
        hipError_t e__ = (expr);                                                                  \
        if (e__ != hipSuccess) {                                                                  \
            std::string m = std::string(#expr) + ": " + hipGetErrorString(e__);                   \
            pdog_destroy(t);                                                                      \
            return fail(PDOG_E_HIP, m);                                                           \
        }                                                                                         \
    } while (0)
    CREATE_TRY(hipStreamCreateWithFlags(&t->own_stream, hipStreamNonBlocking));
    t->stream = t->own_stream;
    CREATE_TRY(hipStreamCreateWithFlags(&t->aux_stream, hipStreamNonBlocking));
    CREATE_TRY(hipEventCreateWithFlags(&t->ev_fork, hipEventDisableTiming));
    CREATE_TRY(hipEventCreateWithFlags(&t->ev_join, hipEventDisableTiming));
    CREATE_TRY(hipEventCreateWithFlags(&t->ev_switch, hipEventDisableTiming));
    CREATE_TRY(hipMalloc(&t->d_taps_row, sizeof(f2) * tr.size()));
    CREATE_TRY(hipMalloc(&t->d_taps_col, sizeof(f2) * tc.size()));
    CREATE_TRY(hipMemcpy(t->d_taps_row, tr.data(), sizeof(f2) * tr.size(), hipMemcpyHostToDevice));
    CREATE_TRY(hipMemcpy(t->d_taps_col, tc.data(), sizeof(f2) * tc.size(), hipMemcpyHostToDevice));
    {
        // roll_col_body's table: (Tc[t], Tc[t-1]) pairs per parity, T outside 0..l-1 is 0
        const int nqb = roll_col_blocks(t->L);
        std::vector<f2> tab((size_t)roll_col_table_len(t->L), f2{0.f, 0.f});
        auto tapc = [&](int c, int k) -> float {
            if (k < 0 || k >= t->L) return 0.f;
            return c ? tc[k].y : tc[k].x;
        };
        for (int qb = 0; qb < nqb; ++qb)
            for (int p = 0; p < 2; ++p)
                for (int c = 0; c < 2; ++c)
                    for (int m = 0; m < ROLL_QB; ++m) {
                        const int tt = p + 2 * (ROLL_QB * qb + m);
                        tab[((qb * 2 + p) * 2 + c) * ROLL_QB + m] = f2{tapc(c, tt), tapc(c, tt - 1)};
                    }
        CREATE_TRY(hipMalloc(&t->d_taps_roll, sizeof(f2) * tab.size()));
        CREATE_TRY(hipMemcpy(t->d_taps_roll, tab.data(), sizeof(f2) * tab.size(), hipMemcpyHostToDevice));
    }
    CREATE_TRY(hipMalloc(&t->d_small, sizeof(int32_t) * 4));
    CREATE_TRY(hipHostMalloc(&t->h_pinned, sizeof(int32_t) * 8, hipHostMallocMapped | hipHostMallocCoherent));
    std::memset(t->h_pinned, 0, sizeof(int32_t) * 8);
    CREATE_TRY(hipHostGetDevicePointer((void **)&t->d_mail_map, t->h_pinned, 0));
    {
        // exact mode (dog_exact.hpp): the reference's dense kernel in Float64, built exactly as :41-43 builds it
        // (K = dir·(g₊⊗g₊ − g₋⊗g₋), column-major), and the decision threshold T = 2δ, δ = u·(6l + 4) for |pixel − dc| ≤ 255
        std::vector<double> K((size_t)t->L * t->L);
        dense_dog_kernel(gp.data(), gm.data(), t->L, t->darker != 0, K.data());
        CREATE_TRY(hipMalloc(&t->d_K64, sizeof(double) * K.size()));
        CREATE_TRY(hipMemcpy(t->d_K64, K.data(), sizeof(double) * K.size(), hipMemcpyHostToDevice));
        t->h_gp = gp;
        t->h_gm = gm;
        {
            const ExactFactors ef = exact_factors(gp, gm);
            t->F_sym_int = ef.sym_int; t->F_sym_sep = ef.sym_sep; t->F_ring = ef.ring; t->F_rescan = ef.rescan;
        }
        CREATE_TRY(hipMalloc(&t->d_ref_stat, sizeof(unsigned long long) * 16)); // [4..7] unused, [8..15]: phase cycles of the refinement (diagnostic build)
        CREATE_TRY(hipMemset(t->d_ref_stat, 0, sizeof(unsigned long long) * 16));
        {
            std::vector<double> g2(2 * (size_t)t->L);
            std::copy(gp.begin(), gp.end(), g2.begin());
            std::copy(gm.begin(), gm.end(), g2.begin() + t->L);
            CREATE_TRY(hipMalloc(&t->d_g64, sizeof(double) * g2.size()));
            CREATE_TRY(hipMemcpy(t->d_g64, g2.data(), sizeof(double) * g2.size(), hipMemcpyHostToDevice));
        }
        // separable Float64 vs the reference's dense Float64 (l² sequential roundings): both within δ64 of the exact value
        t->exact_T64 = 2.0 * std::ldexp(1.0, -53) * (2.1 * t->L * t->L + 8.0 * t->L + 64.0);
        {
            RefineParams rp;
            rp.K64 = t->d_K64;
            rp.g64 = t->d_g64;
            rp.dir = t->darker ? -1.0 : 1.0;
            rp.T64 = t->exact_T64;
            CREATE_TRY(hipMalloc(&t->d_rp, sizeof rp));
            CREATE_TRY(hipMemcpy(t->d_rp, &rp, sizeof rp, hipMemcpyHostToDevice));
        }
        t->exact = refine_lds_bytes(t->n1, t->L, 1, 8) <= kMaxLds - 8192;
        if (!t->exact) // (success all the same: pdog_last_error carries the note, pdog_get_exact reports the state)
            g_err = "pdog_create: window too tall for the refinement's LDS block (n1 + l beyond ~9000 rows): exact mode is OFF for this tracker";
        if (raise_lds_limit((const void *)dog_finish_kernel, refine_lds_bytes(t->n1, t->L, t->ref_cbw, t->ref_rows))) { pdog_destroy(t); return PDOG_E_HIP; }
    }
#undef CREATE_TRY
    rc = ensure_capacity(t, 1);
    if (rc) { pdog_destroy(t); return rc; }
    *out = t;
    return PDOG_OK;
}

int pdog_destroy(pdog_tracker *t)
{
    if (!t) return PDOG_OK;
    (void)hipSetDevice(t->device);
    if (t->stream) (void)hipStreamSynchronize(t->stream);
    if (t->d_taps_row) (void)hipFree(t->d_taps_row);
    if (t->d_taps_col) (void)hipFree(t->d_taps_col);
    if (t->d_taps_roll) (void)hipFree(t->d_taps_roll);
    for (void *p : {(void *)t->d_part_val, (void *)t->d_part_idx, (void *)t->d_part_sec, (void *)t->d_part_mask, (void *)t->d_K64, (void *)t->d_g64, (void *)t->d_rp,
                    (void *)t->d_ref_stat})
        if (p) (void)hipFree(p);
    if (t->d_fold_r) (void)hipFree(t->d_fold_r);
    if (t->d_frame) (void)hipFree(t->d_frame);
    if (t->d_small) (void)hipFree(t->d_small);
    if (t->h_pinned) (void)hipHostFree(t->h_pinned);
    if (t->d_resp) (void)hipFree(t->d_resp);
    if (t->h_tile) (void)hipHostFree(t->h_tile);
    if (t->d_V) (void)hipFree(t->d_V);
    if (t->d_map) (void)hipFree(t->d_map);
    if (t->d_dc) (void)hipFree(t->d_dc);
    if (t->d_counter) (void)hipFree(t->d_counter);
    if (t->d_chain_tmp) (void)hipFree(t->d_chain_tmp);
    if (t->d_tiled_slots) (void)hipFree(t->d_tiled_slots);
    if (t->d_tiled_ctl) (void)hipFree(t->d_tiled_ctl);
    if (t->h2d_stream) { (void)hipStreamSynchronize(t->h2d_stream); (void)hipStreamDestroy(t->h2d_stream); }
    for (int k = 0; k < pdog_tracker::kIngestSlots; ++k) {
        if (t->h_stage[k]) (void)hipHostFree(t->h_stage[k]);
        if (t->d_tiles[k]) (void)hipFree(t->d_tiles[k]);
        if (t->ev_h2d[k]) (void)hipEventDestroy(t->ev_h2d[k]);
        if (t->ev_used[k]) (void)hipEventDestroy(t->ev_used[k]);
    }
    if (t->d_ingest_guess) (void)hipFree(t->d_ingest_guess);
    if (t->d_ingest_out) (void)hipFree(t->d_ingest_out);
    if (t->h_ingest_out) (void)hipHostFree(t->h_ingest_out);
    if (t->aux_stream) { (void)hipStreamSynchronize(t->aux_stream); (void)hipStreamDestroy(t->aux_stream); }
    if (t->ev_fork) (void)hipEventDestroy(t->ev_fork);
    if (t->ev_join) (void)hipEventDestroy(t->ev_join);
    if (t->ev_switch) (void)hipEventDestroy(t->ev_switch);
    if (t->own_stream) (void)hipStreamDestroy(t->own_stream);
    delete t;
    return PDOG_OK;
}

int pdog_get_info(const pdog_tracker *t, pdog_info *o)
{
    if (!t || !o) return fail(PDOG_E_ARG, "pdog_get_info: null");
    o->frame_h = t->fh; o->frame_w = t->fw;
    o->radius_h = t->r1; o->radius_w = t->r2;
    o->win_h = t->n1; o->win_w = t->n2;
    o->kernel_len = t->L;
    o->fill = t->fill;
    o->darker_target = t->darker;
    o->n_strips = t->nstrips;
    o->strip_w = t->var->tw();
    o->variant = t->var->id;
    o->sigma = t->sigma;
    o->target_width = t->tw;
    const int64_t th = t->n1 + t->L - 1, tw = t->n2 + t->L - 1;
    o->algorithmic_bytes_per_window = th * tw + 8;
    o->algorithmic_fma_per_window = 2LL * t->L * (th * t->n2 + (int64_t)t->n1 * t->n2);
    return PDOG_OK;
}

int pdog_kernel_for_batch(const pdog_tracker *t, int n, int *out_variant)
{
    if (!t || !out_variant || n < 0) return fail(PDOG_E_ARG, "pdog_kernel_for_batch: bad argument");
    *out_variant = path_for_batch(t, n);
    return PDOG_OK;
}

int pdog_set_fill(pdog_tracker *t, int fill)
{
    if (!t || fill < 0 || fill > 255) return fail(PDOG_E_ARG, "pdog_set_fill: bad argument");
    t->fill = fill;
    return PDOG_OK;
}

int pdog_set_stream(pdog_tracker *t, void *hip_stream)
{
    if (!t) return fail(PDOG_E_ARG, "pdog_set_stream: null tracker");
    hipStream_t ns = (hipStream_t)hip_stream;
    if (ns == t->stream) return PDOG_OK;
    // The tracker's scratch (strip partials, two-pass intermediate, DC levels, counters, chain state) is per tracker,
    // not per stream: whatever is still queued on the previous stream must finish before work on the new one may
    // touch it.  Stream-ordered, no host wait.
    HIP_TRY(hipSetDevice(t->device));
    HIP_TRY(hipEventRecord(t->ev_switch, t->stream));
    HIP_TRY(hipStreamWaitEvent(ns, t->ev_switch, 0));
    t->stream = ns;
    return PDOG_OK;
}

int pdog_get_stream(const pdog_tracker *t, void **out_hip_stream)
{
    if (!t || !out_hip_stream) return fail(PDOG_E_ARG, "pdog_get_stream: null pointer");
    *out_hip_stream = (void *)t->stream;
    return PDOG_OK;
}

int pdog_reserve(pdog_tracker *t, int max_windows)
{
    if (!t || max_windows <= 0) return fail(PDOG_E_ARG, "pdog_reserve: bad argument");
    HIP_TRY(hipSetDevice(t->device));
    HIP_TRY(hipStreamSynchronize(t->stream));
    return ensure_capacity(t, max_windows);
}

int pdog_set_variant(pdog_tracker *t, int variant)
{
    if (!t) return fail(PDOG_E_ARG, "pdog_set_variant: null tracker");
    if (variant >= 0 && !find_variant(variant)) return fail(PDOG_E_ARG, "pdog_set_variant: unknown variant id");
    HIP_TRY(hipSetDevice(t->device));
    return choose_variant(t, variant);
}

// Drain the tracker's stream and report what its kernels raised while they ran: every entry point that waits for the
// stream goes through here, so a fault surfaces at the first call that waits — not at a later, unrelated pdog_sync.
static int drain_and_check(pdog_tracker *t, const char *who)
{
    HIP_TRY(hipSetDevice(t->device)); // (group mode: the current device is whichever rank was touched last)
    HIP_TRY(hipStreamSynchronize(t->stream));
    if (const int32_t raised = __atomic_load_n(&t->h_pinned[5], __ATOMIC_ACQUIRE)) { // raised by a kernel
        __atomic_store_n(&t->h_pinned[5], 0, __ATOMIC_RELEASE);
        if (raised == 2) { // a wait between resident workgroups gave up (wait_counter): the positions of that work are not valid
            if (t->d_tiled_ctl) (void)hipMemset(t->d_tiled_ctl, 0, sizeof(int) * (4 * (size_t)t->tiled_ctl_cap + 1)); // counters, flags and the abort word
            return fail(PDOG_E_HIP, std::string(who) + ": a kernel gave up waiting for its other workgroups (device-side watchdog); the results of the work just finished are not valid");
        }
        // a device-resident guess was out of range
        return fail(PDOG_E_RANGE, std::string(who) + ": a guess of the work just finished lies outside the padded frame (reference: BoundsError, "
                                  "src/PawsomeTracker.jl:45-46); positions were computed with the fill value there");
    }
    return PDOG_OK;
}

int pdog_sync(pdog_tracker *t)
{
    if (!t) return fail(PDOG_E_ARG, "pdog_sync: null tracker");
    return drain_and_check(t, "pdog_sync");
}

int pdog_set_exact(pdog_tracker *t, int on)
{
    if (!t) return fail(PDOG_E_ARG, "pdog_set_exact: null tracker");
    if (on && refine_lds_bytes(t->n1, t->L, 1, 8) > kMaxLds - 8192)
        return fail(PDOG_E_ARG, "pdog_set_exact: window too tall for the refinement's LDS block");
    HIP_TRY(hipStreamSynchronize(t->stream));
    t->exact = on != 0;
    t->exact_all = on == 2;
    return PDOG_OK;
}

int pdog_set_tuning(pdog_tracker *t, const char *key, int value)
{
    if (!t || !key) return fail(PDOG_E_ARG, "pdog_set_tuning: null pointer");
    HIP_TRY(hipSetDevice(t->device));
    HIP_TRY(hipStreamSynchronize(t->stream));
    const std::string k(key);
    const bool on = value != 0;
    if (k == "host_copy") t->sw.host_copy = on;
    else if (k == "host_sync") t->sw.host_sync = on;
    else if (k == "twopass_4l") t->sw.twopass_4l = on;
    else if (k == "no_roll_map") t->sw.no_roll_map = on;
    else if (k == "no_fold") t->sw.no_fold = on;
    else if (k == "fold_always") t->sw.fold_always = on;
    else if (k == "fault_inject") t->sw.fault_inject = on;
    else if (k == "no_fused_c") {
        t->sw.no_fused_c = on;
        t->fused_resident = 0;
        setup_refine_geometry(t); // the tile layout, and with it the refinement's share of the kernel's LDS
        if (t->fused_ok)
            for (bool resp : {false, true})
                if (int rc = raise_lds_limit((const void *)fused_kernel_for(t, resp), fused_total_lds(t))) return rc;
        if (int rc = setup_tiled(t)) return rc;
    } else if (k == "no_tiled") {
        t->sw.no_tiled = on;
        if (int rc = setup_tiled(t)) return rc; // the tiled kernel's geometry is decided per tracker
    } else return fail(PDOG_E_ARG, "pdog_set_tuning: unknown key '" + k + "'");
    return PDOG_OK;
}

int pdog_get_exact_detail(pdog_tracker *t, uint64_t out[4])
{
    if (!t || !out) return fail(PDOG_E_ARG, "pdog_get_exact_detail: null pointer");
    if (int rc = drain_and_check(t, "pdog_get_exact_detail")) return rc;
    unsigned long long v[16];
    HIP_TRY(hipMemcpy(v, t->d_ref_stat, sizeof v, hipMemcpyDeviceToHost));
    for (int i = 0; i < 4; ++i) out[i] = (uint64_t)v[i];
#ifdef PDOG_ABLATIONS
    std::fprintf(stderr, "pdog refine phases (shader cycles, thread 0): setup %llu, tile %llu, row32 %llu, col32 %llu, row64 %llu, cand64 %llu, verdict %llu\n",
                 v[8], v[9], v[10], v[11], v[12], v[13], v[14]);
#endif
    return PDOG_OK;
}

int pdog_get_exact(pdog_tracker *t, int *out_on, double *out_threshold, uint64_t *out_refined)
{
    if (!t) return fail(PDOG_E_ARG, "pdog_get_exact: null tracker");
    if (out_on) *out_on = t->exact ? 1 : 0;
    if (out_threshold) { // 2δ of the tracker's batch kernel family (small batches may run another family with a tighter bound)
        const Variant &v = *t->var;
        *out_threshold = (double)exact_ctl(t, v.roll ? kFamRoll : v.twopass ? kFamTwoPass8 : v.fused ? kFamFused : kFamRing).T;
    }
    if (out_refined) {
        if (int rc = drain_and_check(t, "pdog_get_exact")) return rc;
        unsigned long long v = 0;
        HIP_TRY(hipMemcpy(&v, t->d_ref_stat, sizeof v, hipMemcpyDeviceToHost));
        *out_refined = (uint64_t)v;
    }
    return PDOG_OK;
}

int pdog_detect_batch(pdog_tracker *t, const uint8_t *d_frames, int64_t frame_stride, int64_t row_stride,
                      int n_frames, const int32_t *d_frame_index, const int32_t *d_guesses, int n,
                      int32_t *d_out_ij, float *d_out_resp)
{
    if (!t) return fail(PDOG_E_ARG, "pdog_detect_batch: null tracker");
    if (n == 0) return PDOG_OK;
    if (!d_frames || !d_guesses || !d_out_ij) return fail(PDOG_E_ARG, "pdog_detect_batch: null pointer");
    if (n < 0 || n_frames <= 0 || row_stride < t->fw || frame_stride < 0) return fail(PDOG_E_ARG, "pdog_detect_batch: bad size/stride");
    if (!d_frame_index && n > n_frames) return fail(PDOG_E_ARG, "pdog_detect_batch: more windows than frames and no frame index");
    if ((long long)n * t->nstrips > 0x7ffffff0LL) return fail(PDOG_E_ARG, "pdog_detect_batch: batch too large");
    HIP_TRY(hipSetDevice(t->device));
    if (n > t->cap_windows) {
        HIP_TRY(hipStreamSynchronize(t->stream));
        int rc = ensure_capacity(t, n);
        if (rc) return rc;
    }
    return launch_detect(t, d_frames, frame_stride, row_stride, d_frame_index, d_guesses, n, d_out_ij, d_out_resp);
}

int pdog_detect_host(pdog_tracker *t, const uint8_t *h_frame, int64_t row_stride, const int32_t guess[2],
                     int32_t out_ij[2], float *h_resp)
{
    if (!t || !h_frame || !guess || !out_ij) return fail(PDOG_E_ARG, "pdog_detect_host: null pointer");
    if (row_stride < t->fw) return fail(PDOG_E_ARG, "pdog_detect_host: row_stride < frame width");
    // The reference's PaddedView extends radii + l past the frame (:45-46) and the filter
    // reads radii + l÷2 around the guess: outside [-l÷2, sz + l÷2 + 1] it raises BoundsError.
    const int hw = t->L >> 1;
    if (guess[0] < -hw || guess[0] > t->fh + hw + 1 || guess[1] < -hw || guess[1] > t->fw + hw + 1)
        return fail(PDOG_E_RANGE, "pdog_detect_host: guess outside the padded frame (reference: BoundsError)");
    HIP_TRY(hipSetDevice(t->device));
    if (h_resp && !t->d_resp) HIP_TRY(hipMalloc(&t->d_resp, sizeof(float) * (size_t)t->n1 * t->n2));
    if (!t->sw.host_copy) {
        // Latency path: the tile is packed into pinned, device-mapped memory (fill materialised, as in
        // pdog_detect_batch_host) and the kernels read it in place over PCIe — no copy commands.  On the device the
        // tile is a frame of its own with the guess at its centre; the tile-local answer is mapped back and clamped
        // (:60-61) here.  Completion: the single-window kernels (fused, two-launch two-pass) publish a ticket right
        // after the answer (system-scope release into the host-coherent mailbox) and the host polls for it — the
        // answer is back before the kernel's end-of-grid bookkeeping, and the stream stays ordered for whatever is
        // queued next.  With a response copy, a pinned batch kernel that publishes no ticket, or a ticket that does
        // not show up in time (a failed launch), the stream is synchronised as usual.
        const int th = t->n1 + 2 * hw, tw = t->n2 + 2 * hw, pitch = round_up(tw, 16);
        if (!t->h_tile) {
            HIP_TRY(hipHostMalloc((void **)&t->h_tile, (size_t)th * pitch, hipHostMallocMapped));
            HIP_TRY(hipHostGetDevicePointer((void **)&t->d_tile_map, t->h_tile, 0));
        }
        uint8_t *d_tile = t->d_tile_map;
        int32_t *d_mail = t->d_mail_map;
        const bool trace = t->sw.host_trace; // diagnostic: where a call's wall time goes
        const auto t0 = std::chrono::steady_clock::now();
        pack_tile(t, h_frame, row_stride, guess[0], guess[1], t->h_tile, pitch);
        t->h_pinned[0] = t->r1 + hw + 1;   // the guess is the tile's centre
        t->h_pinned[1] = t->r2 + hw + 1;
        const auto t1 = std::chrono::steady_clock::now();
        if (t->cap_windows < 1) {
            if (int rc = ensure_capacity(t, 1)) return rc;
        }
        const int32_t ticket = t->ticket = t->ticket % 0x7fffffff + 1;   // 1 … 2^31 − 1, never the mailbox's initial 0
        bool armed = false;
        // the DC level from the tile just packed: the kernels' own 32×32 sample grid (dc_sample_sum / dc_from_sum), so the
        // fused and tiled kernels skip their sample loads and the reduction barrier (≈1.5 µs of a 10.8 µs frame)
        int dc_host;
        {
            int total = 0;
            for (int k = 0; k < 1024; ++k)
                total += t->h_tile[(size_t)(int)(((long long)(k >> 5) * th) >> 5) * pitch + (size_t)(int)(((long long)(k & 31) * tw) >> 5)];
            dc_host = (total + 512) >> 10;
            if (std::abs(dc_host - t->fill) <= 8) dc_host = t->fill;
        }
        int rc = launch_detect(t, d_tile, (int64_t)th * pitch, pitch, nullptr, d_mail, 1, d_mail + 2, h_resp ? t->d_resp : nullptr, th, tw,
                               d_mail + 4, ticket, &armed, t->sw.no_host_dc ? -1 : dc_host);
        if (rc) return rc;
        if (h_resp) HIP_TRY(hipMemcpyAsync(h_resp, t->d_resp, sizeof(float) * (size_t)t->n1 * t->n2, hipMemcpyDeviceToHost, t->stream));
        const auto t2 = std::chrono::steady_clock::now();
        bool done = false;
        if (armed && !h_resp && !t->sw.host_sync) {
            const auto deadline = t2 + std::chrono::microseconds(500);
            for (int spin = 0;; ++spin) {
                if (__atomic_load_n(&t->h_pinned[4], __ATOMIC_ACQUIRE) == ticket) { done = true; break; }
                if ((spin & 63) == 63 && std::chrono::steady_clock::now() > deadline) break;
                __builtin_ia32_pause();
            }
        }
        if (!done) HIP_TRY(hipStreamSynchronize(t->stream));
        if (trace) {
            const auto t3 = std::chrono::steady_clock::now();
            auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
            std::fprintf(stderr, "pdog functor: pack %.1f us, launch %.1f us, sync %.1f us\n", us(t0, t1), us(t1, t2), us(t2, t3));
        }
        out_ij[0] = std::min(std::max(guess[0] - t->r1 - hw + t->h_pinned[2] - 1, 1), t->fh);   // tile-local → frame, clamp (:60-61)
        out_ij[1] = std::min(std::max(guess[1] - t->r2 - hw + t->h_pinned[3] - 1, 1), t->fw);
        return PDOG_OK;
    }
    if (!t->d_frame) HIP_TRY(hipMalloc(&t->d_frame, (size_t)t->fh * t->fw));
    {
        // Only the window's padded tile is read by the kernels (anything else they touch feeds masked
        // lanes), so only that rectangle of the frame crosses PCIe: 109×109 B instead of 2 MB for the
        // default 45×45 window on a 1080p frame.
        const int r_lo = std::max(0, guess[0] - t->r1 - 1 - hw), r_hi = std::min(t->fh, guess[0] + t->r1 + hw);
        const int c_lo = std::max(0, guess[1] - t->r2 - 1 - hw), c_hi = std::min(t->fw, guess[1] + t->r2 + hw);
        if (r_hi > r_lo && c_hi > c_lo)
            HIP_TRY(hipMemcpy2DAsync(t->d_frame + (size_t)r_lo * t->fw + c_lo, t->fw, h_frame + (size_t)r_lo * row_stride + c_lo,
                                     row_stride, (size_t)(c_hi - c_lo), (size_t)(r_hi - r_lo), hipMemcpyHostToDevice, t->stream));
    }
    t->h_pinned[0] = guess[0];
    t->h_pinned[1] = guess[1];
    HIP_TRY(hipMemcpyAsync(t->d_small, t->h_pinned, sizeof(int32_t) * 2, hipMemcpyHostToDevice, t->stream));
    int rc = launch_detect(t, t->d_frame, (int64_t)t->fh * t->fw, t->fw, nullptr, t->d_small, 1, t->d_small + 2,
                           h_resp ? t->d_resp : nullptr);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(t->h_pinned + 2, t->d_small + 2, sizeof(int32_t) * 2, hipMemcpyDeviceToHost, t->stream));
    if (h_resp) HIP_TRY(hipMemcpyAsync(h_resp, t->d_resp, sizeof(float) * (size_t)t->n1 * t->n2, hipMemcpyDeviceToHost, t->stream));
    HIP_TRY(hipStreamSynchronize(t->stream));
    out_ij[0] = t->h_pinned[2];
    out_ij[1] = t->h_pinned[3];
    return PDOG_OK;
}

} // extern "C"

namespace {

// One window's padded tile, (n1+l-1) rows of `pitch` bytes: the frame rectangle the functor reads, with the
// PaddedView fill (:48) materialised wherever the rectangle leaves the frame.  Tile row a, column b is the
// padded frame at 1-based (g1 - r1 - l÷2 + a, g2 - r2 - l÷2 + b).
void pack_tile_geo(const uint8_t *frame, int fh, int fw, int64_t row_stride, int fill, int L, int r1, int r2, int g1, int g2,
                   uint8_t *dst, int64_t pitch, bool stream = false)
{
    const int hw = L >> 1, th = 2 * r1 + 1 + 2 * hw, tw = 2 * r2 + 1 + 2 * hw;
    const int i0 = g1 - r1 - hw - 1, j0 = g2 - r2 - hw - 1;        // 0-based frame coordinates of tile (0, 0)
    const int jl = std::min(tw, std::max(0, -j0));                 // columns left of the frame
    const int jr = std::max(jl, std::min(tw, fw - j0));            // first column right of the frame
    // stream: the tile goes to pinned staging that only the DMA engine reads next — assemble each row in a small
    // buffer and write it with non-temporal stores, so the copy engine finds the data in DRAM instead of having to
    // snoop dirty lines out of this core's cache
    const bool nt = stream && pitch % 16 == 0 && pitch <= 4096 && ((uintptr_t)dst & 15) == 0;
    alignas(16) uint8_t rowbuf[4096];
    for (int a = 0; a < th; ++a) {
        uint8_t *out = dst + (size_t)a * pitch;
        uint8_t *row = nt ? rowbuf : out;
        const int gi = i0 + a;
        if (gi < 0 || gi >= fh) {
            std::memset(row, fill, (size_t)pitch);
        } else {
            if (jl) std::memset(row, fill, (size_t)jl);
            if (jr > jl) std::memcpy(row + jl, frame + (size_t)gi * row_stride + (j0 + jl), (size_t)(jr - jl));
            if (pitch > jr) std::memset(row + jr, fill, (size_t)(pitch - jr));
        }
        if (nt)
            for (int64_t k = 0; k < pitch; k += 16)
                _mm_stream_si128((__m128i *)(out + k), _mm_load_si128((const __m128i *)(rowbuf + k)));
    }
    if (nt) _mm_sfence(); // the non-temporal stores are globally visible before the caller publishes the tile
}

void pack_tile(const pdog_tracker *t, const uint8_t *frame, int64_t row_stride, int g1, int g2, uint8_t *dst, int pitch)
{
    // cached stores: the functor's kernel reads this tile in place right away (non-temporal stores measured equal here)
    pack_tile_geo(frame, t->fh, t->fw, row_stride, t->fill, t->L, t->r1, t->r2, g1, g2, dst, pitch, false);
}

} // namespace

// The tile packer as a host-only entry (no GPU): what the host paths hand to the kernels, checkable on a CPU box.
extern "C" int pdog_window_tile(const uint8_t *h_frame, int frame_h, int frame_w, int64_t row_stride, int fill, double target_width,
                                int win_h, int win_w, const int32_t guess[2], uint8_t *h_out, int64_t out_pitch)
{
    if (!h_frame || !guess || !h_out) return fail(PDOG_E_ARG, "pdog_window_tile: null pointer");
    if (frame_h <= 0 || frame_w <= 0 || row_stride < frame_w || fill < 0 || fill > 255 || win_h <= 0 || win_w <= 0 || !(target_width > 0))
        return fail(PDOG_E_ARG, "pdog_window_tile: bad argument");
    const int L = kernel_len_of_sigma(sigma_of(target_width)), r1 = win_h / 2, r2 = win_w / 2;
    if (out_pitch < 2 * r2 + L) return fail(PDOG_E_ARG, "pdog_window_tile: out_pitch smaller than the tile width");
    pack_tile_geo(h_frame, frame_h, frame_w, row_stride, fill, L, r1, r2, guess[0], guess[1], h_out, out_pitch);
    return PDOG_OK;
}

namespace {

int ingest_threads(const pdog_tracker *t)
{
    if (t->sw.host_threads) return t->sw.host_threads;
    const unsigned hc = std::thread::hardware_concurrency();
    return (int)std::max(1u, std::min(16u, hc ? hc : 1u));
}

} // namespace

// Frame ingest for batches: the frames are in HOST memory, as `read!(vid, trckr.img.data)`
// (:166) leaves them.  Only each window's padded tile crosses PCIe (cfg3: 103 KB instead of the 2 MB
// frame): host threads pack the tiles of a chunk into pinned staging, one async copy per chunk moves them
// on a copy stream, and the kernels of chunk c run beside the copy of chunk c+1 and the packing of c+2.
// On the device every tile is a frame of its own with the guess at its centre, so the kernels see exactly
// the pixel values the padded frame would give them; the tile-local result is mapped back and clamped to
// the frame (:60-61) on the host.
extern "C" int pdog_detect_batch_host(pdog_tracker *t, const uint8_t *h_frames, int64_t frame_stride, int64_t row_stride,
                                      int n_frames, const int32_t *h_frame_index, const int32_t *h_guesses, int n,
                                      int32_t *h_out_ij)
{
    if (!t) return fail(PDOG_E_ARG, "pdog_detect_batch_host: null tracker");
    if (n == 0) return PDOG_OK;
    if (!h_frames || !h_guesses || !h_out_ij) return fail(PDOG_E_ARG, "pdog_detect_batch_host: null pointer");
    if (n < 0 || n_frames <= 0 || row_stride < t->fw || frame_stride < 0) return fail(PDOG_E_ARG, "pdog_detect_batch_host: bad size/stride");
    if (!h_frame_index && n > n_frames) return fail(PDOG_E_ARG, "pdog_detect_batch_host: more windows than frames and no frame index");
    const int hw = t->L >> 1;
    for (int b = 0; b < n; ++b) {
        const int g1 = h_guesses[2 * b], g2 = h_guesses[2 * b + 1];
        if (g1 < -hw || g1 > t->fh + hw + 1 || g2 < -hw || g2 > t->fw + hw + 1)
            return fail(PDOG_E_RANGE, "pdog_detect_batch_host: guess outside the padded frame (reference: BoundsError)");
        if (h_frame_index && (h_frame_index[b] < 0 || h_frame_index[b] >= n_frames))
            return fail(PDOG_E_ARG, "pdog_detect_batch_host: frame index out of range");
    }
    HIP_TRY(hipSetDevice(t->device));
    const int th = t->n1 + 2 * hw, tw = t->n2 + 2 * hw, pitch = round_up(tw, 16);
    const size_t tile_bytes = (size_t)th * pitch;
    // chunk: ≈32 MB of tiles, at least 64 windows (the batch kernels want ≥ 1000 strip-waves when they can get them)
    int chunk = (int)std::max<size_t>(64, ((size_t)32 << 20) / tile_bytes);
    if (t->sw.ingest_chunk) chunk = t->sw.ingest_chunk;
    chunk = std::min(chunk, n);
    constexpr int NS = pdog_tracker::kIngestSlots;
    if (!t->h2d_stream) {
        HIP_TRY(hipStreamCreateWithFlags(&t->h2d_stream, hipStreamNonBlocking));
        for (int k = 0; k < NS; ++k) {
            HIP_TRY(hipEventCreateWithFlags(&t->ev_h2d[k], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&t->ev_used[k], hipEventDisableTiming));
        }
    }
    if (t->ingest_slot_bytes < tile_bytes * chunk) {
        HIP_TRY(hipStreamSynchronize(t->stream));
        HIP_TRY(hipStreamSynchronize(t->h2d_stream));
        for (int k = 0; k < NS; ++k) {
            if (t->h_stage[k]) (void)hipHostFree(t->h_stage[k]);
            if (t->d_tiles[k]) (void)hipFree(t->d_tiles[k]);
            t->h_stage[k] = nullptr; t->d_tiles[k] = nullptr;
        }
        t->ingest_slot_bytes = 0;
        for (int k = 0; k < NS; ++k) {
            HIP_TRY(hipHostMalloc((void **)&t->h_stage[k], tile_bytes * chunk, hipHostMallocDefault));
            HIP_TRY(hipMalloc(&t->d_tiles[k], tile_bytes * chunk));
        }
        t->ingest_slot_bytes = tile_bytes * chunk;
    }
    if (t->ingest_guess_cap < chunk) {
        HIP_TRY(hipStreamSynchronize(t->stream));
        if (t->d_ingest_guess) (void)hipFree(t->d_ingest_guess);
        t->d_ingest_guess = nullptr; t->ingest_guess_cap = 0;
        HIP_TRY(hipMalloc(&t->d_ingest_guess, sizeof(int32_t) * 2 * (size_t)chunk));
        std::vector<int32_t> centre(2 * (size_t)chunk);
        for (int b = 0; b < chunk; ++b) { centre[2 * b] = t->r1 + hw + 1; centre[2 * b + 1] = t->r2 + hw + 1; }
        HIP_TRY(hipMemcpy(t->d_ingest_guess, centre.data(), sizeof(int32_t) * centre.size(), hipMemcpyHostToDevice));
        t->ingest_guess_cap = chunk;
    }
    if (t->ingest_cap < n) {
        HIP_TRY(hipStreamSynchronize(t->stream));
        if (t->d_ingest_out) (void)hipFree(t->d_ingest_out);
        if (t->h_ingest_out) (void)hipHostFree(t->h_ingest_out);
        t->d_ingest_out = nullptr; t->h_ingest_out = nullptr; t->ingest_cap = 0;
        HIP_TRY(hipMalloc(&t->d_ingest_out, sizeof(int32_t) * 2 * (size_t)n));
        HIP_TRY(hipHostMalloc((void **)&t->h_ingest_out, sizeof(int32_t) * 2 * (size_t)n, hipHostMallocDefault));
        t->ingest_cap = n;
    }
    if (chunk > t->cap_windows) {
        HIP_TRY(hipStreamSynchronize(t->stream));
        if (int rc = ensure_capacity(t, chunk)) return rc;
    }

    const int nchunks = (n + chunk - 1) / chunk;
    std::atomic<int> next{0}, submitted{0}, failed{0};
    std::vector<std::atomic<int>> packed(nchunks);
    for (auto &p : packed) p.store(0);
    auto worker = [&]() {
        (void)hipSetDevice(t->device);
        int waited_for = -1; // highest chunk whose slot this thread has seen released
        for (;;) {
            const int b = next.fetch_add(1);
            if (b >= n || failed.load()) return;
            const int c = b / chunk;
            if (c >= NS && waited_for < c) {
                // the slot was last used by chunk c - NS: wait until its copy has been enqueued, then done
                while (submitted.load(std::memory_order_acquire) <= c - NS) {
                    if (failed.load()) return;
                    std::this_thread::yield();
                }
                if (hipEventSynchronize(t->ev_h2d[c % NS]) != hipSuccess) { failed.store(1); return; }
                waited_for = c;
            }
            const int f = h_frame_index ? h_frame_index[b] : b;
            const bool nt_stores = !t->sw.ingest_no_nt;
            pack_tile_geo(h_frames + (int64_t)f * frame_stride, t->fh, t->fw, row_stride, t->fill, t->L, t->r1, t->r2,
                          h_guesses[2 * b], h_guesses[2 * b + 1], t->h_stage[c % NS] + (size_t)(b - c * chunk) * tile_bytes, pitch, nt_stores);
            packed[c].fetch_add(1, std::memory_order_release);
        }
    };
    const int nthreads = std::min(ingest_threads(t), n);
    std::vector<std::thread> pool;
    for (int k = 0; k < nthreads; ++k) pool.emplace_back(worker);
    int rc = PDOG_OK;
    const bool trace = t->sw.ingest_trace; // diagnostic: where the wall time of a call goes
    const auto t_begin = std::chrono::steady_clock::now();
    double wait_pack_ms = 0;
    for (int c = 0; c < nchunks && rc == PDOG_OK; ++c) {
        const int w0 = c * chunk, nw = std::min(chunk, n - w0), slot = c % NS;
        const auto tw0 = std::chrono::steady_clock::now();
        while (packed[c].load(std::memory_order_acquire) < nw && !failed.load()) std::this_thread::yield();
        wait_pack_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tw0).count();
        if (failed.load()) { rc = fail(PDOG_E_HIP, "pdog_detect_batch_host: event wait failed in a packing thread"); break; }
        hipError_t e = hipSuccess;
        if (c >= NS) e = hipStreamWaitEvent(t->h2d_stream, t->ev_used[slot], 0); // kernels of chunk c - NS are done with the device slot
        if (e == hipSuccess) e = hipMemcpyAsync(t->d_tiles[slot], t->h_stage[slot], tile_bytes * nw, hipMemcpyHostToDevice, t->h2d_stream);
        if (e == hipSuccess) e = hipEventRecord(t->ev_h2d[slot], t->h2d_stream);
        submitted.store(c + 1, std::memory_order_release);
        if (e == hipSuccess) e = hipStreamWaitEvent(t->stream, t->ev_h2d[slot], 0);
        if (e != hipSuccess) { rc = fail(PDOG_E_HIP, std::string("pdog_detect_batch_host: ") + hipGetErrorString(e)); break; }
        rc = launch_detect(t, t->d_tiles[slot], (int64_t)tile_bytes, pitch, nullptr, t->d_ingest_guess, nw,
                           t->d_ingest_out + 2 * (size_t)w0, nullptr, th, tw);
        if (rc == PDOG_OK && hipEventRecord(t->ev_used[slot], t->stream) != hipSuccess)
            rc = fail(PDOG_E_HIP, "pdog_detect_batch_host: hipEventRecord failed");
    }
    if (rc != PDOG_OK) { failed.store(1); submitted.store(nchunks + NS); }
    for (auto &th_ : pool) th_.join();
    if (rc != PDOG_OK) { (void)hipStreamSynchronize(t->h2d_stream); (void)hipStreamSynchronize(t->stream); return rc; }
    const auto t_submitted = std::chrono::steady_clock::now();
    HIP_TRY(hipMemcpyAsync(t->h_ingest_out, t->d_ingest_out, sizeof(int32_t) * 2 * (size_t)n, hipMemcpyDeviceToHost, t->stream));
    HIP_TRY(hipStreamSynchronize(t->stream));
    if (trace) {
        const auto t_end = std::chrono::steady_clock::now();
        std::fprintf(stderr, "pdog ingest: %d windows, %d chunks of %d, %d threads: submit loop %.2f ms (waiting for packers %.2f ms), drain %.2f ms\n",
                     n, nchunks, chunk, nthreads, std::chrono::duration<double, std::milli>(t_submitted - t_begin).count(), wait_pack_ms,
                     std::chrono::duration<double, std::milli>(t_end - t_submitted).count());
    }
    for (int b = 0; b < n; ++b) {
        // tile-local 1-based (p, q)  ->  padded-frame index  ->  clamp (:60-61)
        const int i = h_guesses[2 * b] - t->r1 - hw + t->h_ingest_out[2 * b] - 1;
        const int j = h_guesses[2 * b + 1] - t->r2 - hw + t->h_ingest_out[2 * b + 1] - 1;
        h_out_ij[2 * b] = std::min(std::max(i, 1), t->fh);
        h_out_ij[2 * b + 1] = std::min(std::max(j, 1), t->fw);
    }
    return PDOG_OK;
}

namespace {

// stream-ordered fallback: frame k's guess is frame k-1's (clamped) answer, read straight from the
// output array — stream order is the dependency, no host round trip per frame
int chain_by_launches(pdog_tracker *t, const uint8_t *d_frames, int64_t frame_stride, int64_t row_stride,
                      int n_frames, const int32_t *d_start, int32_t *d_out_ij)
{
    for (int k = 0; k < n_frames; ++k) {
        const int32_t *guess = k ? d_out_ij + 2 * (k - 1) : d_start;
        int rc = launch_detect(t, d_frames + (int64_t)k * frame_stride, frame_stride, row_stride, nullptr, guess, 1,
                               d_out_ij + 2 * k, nullptr);
        if (rc) return rc;
    }
    return PDOG_OK;
}

} // namespace

extern "C" int pdog_detect_chains(pdog_tracker *t, const uint8_t *d_frames, int64_t frame_stride, int64_t row_stride,
                                  int n_frames, int n_clips, const int32_t *d_start_guesses, int32_t *d_out_ij)
{
    if (!t || !d_frames || !d_start_guesses || !d_out_ij) return fail(PDOG_E_ARG, "pdog_detect_chains: null pointer");
    if (n_frames <= 0 || n_clips <= 0 || row_stride < t->fw || frame_stride < 0)
        return fail(PDOG_E_ARG, "pdog_detect_chains: bad size/stride");
    HIP_TRY(hipSetDevice(t->device));
    const Variant &v = *t->var;
    const int chain_strips = (t->n2 + ROLL_TW - 1) / ROLL_TW;
    // Enough clips to fill the GPU with one wave per strip → ONE persistent launch (a workgroup per clip walks
    // its frames).  Fewer clips are latency-bound by that single wave per strip; then each frame is a small
    // batch that the two-pass kernels spread over many workgroups (21 µs vs 48 µs per frame, one 45×45 window).
    // (round 3: the fused kernel's compile-time-l instances walk 256 … 4096 clips of 45×45 windows at 27–30 M frames/s, the persistent
    // roll chain 19–28 M; at 63×63 the chain wins from ≈1400 clips: 23.5 against 18.0 M frames/s at 2048)
    const bool fused_wins = t->fused_ok && !t->forced_variant && ((long long)t->n1 * t->n2 < 3000 || (long long)n_clips * chain_strips < 1400);
    const bool persistent = v.roll && v.chain && chain_strips <= 8 && !fused_wins &&
                            (t->forced_variant || !t->small_twopass || (long long)n_clips * chain_strips >= 1000);
    if (t->sw.tiled_force && n_clips == 1 && !t->forced_variant) {
        bool launched = false;
        if (int rc = launch_tiled(t, d_frames, frame_stride, row_stride, nullptr, d_start_guesses, 1, n_frames, d_out_ij, nullptr, t->fh, t->fw,
                                  nullptr, 0, false, &launched)) return rc;
        if (launched) return PDOG_OK;
    }
    if (v.fused || (!persistent && !t->forced_variant && t->fused_ok)) // one launch: a workgroup per clip loops over its frames
        return launch_fused(t, d_frames, frame_stride, row_stride, nullptr, d_start_guesses, n_clips, n_frames, d_out_ij, nullptr, t->fh, t->fw);
    if (persistent) {
        ChainGeo cg;
        LaunchGeo &g = cg.g;
        std::memset(&g, 0, sizeof g);
        g.frames = d_frames;
        g.frame_stride = frame_stride;
        g.row_stride = row_stride;
        g.fh = t->fh; g.fw = t->fw; g.r1 = t->r1; g.r2 = t->r2; g.n1 = t->n1; g.n2 = t->n2;
        g.L = t->L; g.fill = t->fill; g.nstrips = chain_strips; g.n = n_clips;
        g.nblocks = n_clips * chain_strips;
        g.nslots = chain_strips;
        cg.start = d_start_guesses;
        cg.out_ij = d_out_ij;
        cg.n_frames = n_frames;
        g.ex = exact_ctl(t, kFamRoll);
        cg.rp = t->exact ? t->d_rp : nullptr;
        cg.taps_col_plain = t->d_taps_col;
        // the strips' LDS doubles as the refinement's scratch: the widest block (with its pixel tile if possible) that
        // fits what the strips need anyway, so that exact mode does not cost the chain kernel occupancy
        const int NAc = t->n1 + t->L - 1;
        const size_t base = std::max((size_t)chain_strips * roll_lds_bytes(v.LT), refine_lds_bytes(t->n1, t->L, 1, 8));
        cg.ref_cbw = 1;
        cg.ref_rows = 8;
        for (int cbw = std::min(t->n2, t->ref_cbw); cbw >= 1; --cbw) { // widest block first; its tile fully resident if possible, else the tallest slice
            const size_t fixed_r = refine_lds_bytes(t->n1, t->L, cbw, 0);
            if (fixed_r + (size_t)8 * refine_tile_pitch(cbw, t->L) > base) continue;
            cg.ref_cbw = cbw;
            cg.ref_rows = (int)std::min<size_t>((size_t)NAc, (base - fixed_r) / (size_t)refine_tile_pitch(cbw, t->L));
            while (cg.ref_rows > 8 && refine_lds_bytes(t->n1, t->L, cbw, cg.ref_rows) > base) --cg.ref_rows;
            break;
        }
        const size_t lds = base;
        if (int rc = raise_lds_limit((const void *)v.chain, lds)) return rc;
        hipLaunchKernelGGL(v.chain, dim3(n_clips), dim3(64 * chain_strips), lds, t->stream, cg,
                           (const f2 *)t->d_taps_row, (const f2 *)t->d_taps_roll);
        HIP_TRY(hipGetLastError());
        return PDOG_OK;
    }
    if (t->cap_windows < n_clips) {
        HIP_TRY(hipStreamSynchronize(t->stream));
        int rc = ensure_capacity(t, n_clips);
        if (rc) return rc;
    }
    if (!t->forced_variant) { // the tiled kernel: one cooperative launch, every sub-window's workgroup of every clip resident (as many clips as that allows)
        bool launched = false;
        if (int rc = launch_tiled(t, d_frames, frame_stride, row_stride, nullptr, d_start_guesses, n_clips, n_frames, d_out_ij, nullptr, t->fh, t->fw,
                                  nullptr, 0, false, &launched)) return rc;
        if (launched) return PDOG_OK;
    }
    if (n_clips == 1) {
        // stream order is the dependency: frame k's guess is read straight from frame k-1's answer
        return chain_by_launches(t, d_frames, frame_stride, row_stride, n_frames, d_start_guesses, d_out_ij);
    }
    if (t->chain_tmp_cap < n_clips) {
        HIP_TRY(hipStreamSynchronize(t->stream));
        if (t->d_chain_tmp) (void)hipFree(t->d_chain_tmp);
        t->d_chain_tmp = nullptr; t->chain_tmp_cap = 0;
        HIP_TRY(hipMalloc(&t->d_chain_tmp, sizeof(int32_t) * 4 * (size_t)n_clips));
        t->chain_tmp_cap = n_clips;
    }
    int32_t *cur = t->d_chain_tmp, *step = t->d_chain_tmp + 2 * (size_t)n_clips;
    HIP_TRY(hipMemcpyAsync(cur, d_start_guesses, sizeof(int32_t) * 2 * (size_t)n_clips, hipMemcpyDeviceToDevice, t->stream));
    for (int k = 0; k < n_frames; ++k) {
        // step k: window c looks at clip c's frame k = frame (c*n_frames + k): a batch whose frame stride is one clip
        int rc = launch_detect(t, d_frames + (int64_t)k * frame_stride, frame_stride * n_frames, row_stride, nullptr, cur, n_clips,
                               step, nullptr);
        if (rc) return rc;
        hipLaunchKernelGGL(dog_chain_step_kernel, dim3((n_clips + 255) / 256), dim3(256), 0, t->stream, step, cur, d_out_ij,
                           n_clips, n_frames, k);
        HIP_TRY(hipGetLastError());
    }
    return PDOG_OK;
}

extern "C" int pdog_detect_chain(pdog_tracker *t, const uint8_t *d_frames, int64_t frame_stride, int64_t row_stride,
                                 int n_frames, const int32_t start_guess[2], int32_t *d_out_ij)
{
    if (!t || !d_frames || !start_guess || !d_out_ij) return fail(PDOG_E_ARG, "pdog_detect_chain: null pointer");
    if (n_frames <= 0 || row_stride < t->fw) return fail(PDOG_E_ARG, "pdog_detect_chain: bad size/stride");
    HIP_TRY(hipSetDevice(t->device));
    HIP_TRY(hipMemcpyAsync(t->d_small, start_guess, sizeof(int32_t) * 2, hipMemcpyHostToDevice, t->stream));
    return pdog_detect_chains(t, d_frames, frame_stride, row_stride, n_frames, 1, t->d_small, d_out_ij);
}


// ---- a chain whose positions can be consumed while it runs (the reference's Diagnose overlay,
// src/diagnose.jl:30-38, draws frame k as soon as ij[k] exists) ----
extern "C" int pdog_alloc_host(size_t bytes, void **out)
{
    if (!out || bytes == 0) return fail(PDOG_E_ARG, "pdog_alloc_host: bad argument");
    void *p = nullptr;
    HIP_TRY(hipHostMalloc(&p, bytes, hipHostMallocMapped | hipHostMallocCoherent | hipHostMallocPortable)); // portable: whichever device is current
    std::memset(p, 0, bytes);
    *out = p;
    return PDOG_OK;
}

extern "C" int pdog_free_host(void *p)
{
    if (p) HIP_TRY(hipHostFree(p));
    return PDOG_OK;
}

extern "C" int pdog_detect_chain_progress(pdog_tracker *t, const uint8_t *d_frames, int64_t frame_stride, int64_t row_stride,
                                          int n_frames, const int32_t start_guess[2], int32_t *h_out_ij, int32_t *h_progress)
{
    if (!t || !d_frames || !start_guess || !h_out_ij || !h_progress) return fail(PDOG_E_ARG, "pdog_detect_chain_progress: null pointer");
    if (n_frames <= 0 || row_stride < t->fw || frame_stride < 0) return fail(PDOG_E_ARG, "pdog_detect_chain_progress: bad size/stride");
    HIP_TRY(hipSetDevice(t->device));
    int32_t *d_out = nullptr, *d_prog = nullptr;
    if (hipHostGetDevicePointer((void **)&d_out, h_out_ij, 0) != hipSuccess || hipHostGetDevicePointer((void **)&d_prog, h_progress, 0) != hipSuccess)
        return fail(PDOG_E_ARG, "pdog_detect_chain_progress: h_out_ij / h_progress must come from pdog_alloc_host");
    __atomic_store_n(h_progress, 0, __ATOMIC_RELEASE);
    HIP_TRY(hipMemcpyAsync(t->d_small, start_guess, sizeof(int32_t) * 2, hipMemcpyHostToDevice, t->stream));
    if (t->var->fused || (!t->forced_variant && t->fused_ok)) // one launch: the kernel publishes k + 1 after every frame
        return launch_fused(t, d_frames, frame_stride, row_stride, nullptr, t->d_small, 1, n_frames, d_out, nullptr, t->fh, t->fw,
                            d_prog, 0, true);
    if (t->cap_windows < 1) {
        HIP_TRY(hipStreamSynchronize(t->stream));
        if (int rc = ensure_capacity(t, 1)) return rc;
    }
    if (!t->forced_variant) { // the tiled kernel: the combining workgroup publishes k + 1 after every frame
        bool launched = false;
        if (int rc = launch_tiled(t, d_frames, frame_stride, row_stride, nullptr, t->d_small, 1, n_frames, d_out, nullptr, t->fh, t->fw, d_prog, 0, true,
                                  &launched)) return rc;
        if (launched) return PDOG_OK;
    }
    for (int k = 0; k < n_frames; ++k) { // stream-ordered launches per frame; frame k's guess is read from the (host-mapped) answer k − 1
        bool armed = false;
        int rc = launch_detect(t, d_frames + (int64_t)k * frame_stride, frame_stride, row_stride, nullptr, k ? d_out + 2 * (k - 1) : t->d_small, 1,
                               d_out + 2 * k, nullptr, 0, 0, d_prog, k + 1, &armed);
        if (rc) return rc;
        if (!armed) {
            hipLaunchKernelGGL(dog_publish_kernel, dim3(1), dim3(64), 0, t->stream, d_prog, k + 1);
            HIP_TRY(hipGetLastError());
        }
    }
    return PDOG_OK;
}
