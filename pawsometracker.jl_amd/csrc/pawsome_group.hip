// pawsome_group.hip — several GPUs of one node behind one handle (include/pawsome_dog.h, pdog_group_*).
//
// The reference has no multi-device mode; the unit that shards is the functor applied to independent windows
// (/root/reference/src/PawsomeTracker.jl:55-62), and what comes back is the position list of :173.  One host
// process owns every device of the group (ncclCommInitAll): rank r keeps its shard of frames/guesses resident on
// its own device and runs the ordinary single-device path there (pdog_detect_batch through the public ABI — this
// file uses nothing else of pawsome_dog.hip); the results go to the root device with ONE ncclGather per batch,
// enqueued on each rank's tracker stream right behind its kernels (8 B per window: latency-bound over xGMI, no
// data-path collective).  Shards differ by at most one window, so the gather sends max-shard-sized blocks and a
// copy kernel on the root compacts them when n_total is not a multiple of the group size.
#include "../../include/pawsome_dog.h"
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>   // types and prototypes only: the library is opened when the first group is created (below)
#include <dlfcn.h>
#include <mutex>
#include <string>
#include <vector>

extern "C" __attribute__((visibility("hidden"))) void pdog_set_error_text(const char *msg); // pawsome_dog.hip: the thread's pdog_last_error() text

namespace {

int gfail(int code, const std::string &msg)
{
    pdog_set_error_text(msg.c_str());
    return code;
}
#define G_HIP(expr)                                                                                   \
    do {                                                                                              \
        hipError_t e__ = (expr);                                                                      \
        if (e__ != hipSuccess) return gfail(PDOG_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e__)); \
    } while (0)
// RCCL is opened lazily, by pdog_group_create: a C or Julia host that only uses the single-device ABI can load this
// library on a machine (or loader path) without librccl, and no second RCCL copy enters a process that never asks for a
// group.  dlopen by SONAME: a librccl the process already holds (PyTorch-ROCm's) is the one that is used.
struct Rccl {
    void *handle = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGather) Gather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string error;
};
Rccl &rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.handle) break;
        }
        if (!r.handle) { r.error = std::string("librccl not found (") + (dlerror() ? dlerror() : "dlopen failed") + ")"; return; }
        auto sym = [&](const char *n) { void *p = dlsym(r.handle, n); if (!p && r.error.empty()) r.error = std::string("librccl lacks ") + n; return p; };
        r.CommInitAll = (decltype(r.CommInitAll))sym("ncclCommInitAll");
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
        r.Gather = (decltype(r.Gather))sym("ncclGather");
        r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    });
    return r;
}
#define G_NCCL(expr)                                                                                  \
    do {                                                                                              \
        ncclResult_t r__ = (expr);                                                                    \
        if (r__ != ncclSuccess) return gfail(PDOG_E_HIP, std::string(#expr) + ": " + rccl().GetErrorString(r__)); \
    } while (0)

// Contiguous shards whose sizes differ by at most one: the first n_total % ndev ranks own one window more.
__host__ __device__ inline void shard_bounds(int n_total, int ndev, int rank, int &lo, int &hi)
{
    const int base = n_total / ndev, rem = n_total % ndev;
    lo = rank * base + (rank < rem ? rank : rem);
    hi = lo + base + (rank < rem ? 1 : 0);
}
// … and the inverse: which rank owns window w, and where in that rank's shard it sits.
__host__ __device__ inline void shard_owner(int n_total, int ndev, int w, int &rank, int &local)
{
    const int base = n_total / ndev, rem = n_total % ndev;
    const int cut = rem * (base + 1);
    rank = w < cut ? w / (base + 1) : rem + (w - cut) / (base > 0 ? base : 1);
    int lo, hi;
    shard_bounds(n_total, ndev, rank, lo, hi);
    local = w - lo;
}

// gathered[r][max_n][2] → out[lo_r + k][2], k < hi_r - lo_r (only needed when the shards are unequal)
__global__ void group_compact_kernel(const int32_t *__restrict__ gathered, int32_t *__restrict__ out, int n_total, int ndev, int max_n)
{
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_total) return;
    int r, k;
    shard_owner(n_total, ndev, w, r, k);
    out[2 * w] = gathered[2 * ((long long)r * max_n + k)];
    out[2 * w + 1] = gathered[2 * ((long long)r * max_n + k) + 1];
}

} // namespace

struct pdog_group {
    int ndev = 0;
    std::vector<int> dev;
    std::vector<pdog_tracker *> tr;
    std::vector<ncclComm_t> comm;
    std::vector<int32_t *> d_local;   // per rank: max_n x 2 results of its shard
    int32_t *d_gathered = nullptr;    // root: ndev x max_n x 2 (unequal shards only)
    int cap_local = 0, cap_gathered = 0;
};

extern "C" {

int pdog_group_shard(const pdog_group *g, int n_total, int rank, int *lo, int *hi)
{
    if (!g) return gfail(PDOG_E_ARG, "pdog_group_shard: null group");
    return pdog_shard_range(n_total, g->ndev, rank, lo, hi);
}

int pdog_shard_range(int n_total, int ndev, int rank, int *lo, int *hi)
{
    if (!lo || !hi || n_total < 0 || ndev <= 0 || rank < 0 || rank >= ndev) return gfail(PDOG_E_ARG, "pdog_shard_range: bad argument");
    shard_bounds(n_total, ndev, rank, *lo, *hi);
    return PDOG_OK;
}

int pdog_shard_owner(int n_total, int ndev, int window, int *rank, int *local_index)
{
    if (!rank || !local_index || ndev <= 0 || window < 0 || window >= n_total) return gfail(PDOG_E_ARG, "pdog_shard_owner: bad argument");
    shard_owner(n_total, ndev, window, *rank, *local_index);
    return PDOG_OK;
}

int pdog_group_destroy(pdog_group *g)
{
    if (!g) return PDOG_OK;
    for (int r = 0; r < (int)g->tr.size(); ++r)
        if (g->tr[r]) (void)pdog_sync(g->tr[r]);
    for (int r = 0; r < (int)g->comm.size(); ++r)
        if (g->comm[r]) (void)rccl().CommDestroy(g->comm[r]);
    for (int r = 0; r < (int)g->dev.size(); ++r) {
        (void)hipSetDevice(g->dev[r]);
        if (r < (int)g->d_local.size() && g->d_local[r]) (void)hipFree(g->d_local[r]);
        if (r == 0 && g->d_gathered) (void)hipFree(g->d_gathered);
    }
    for (int r = 0; r < (int)g->tr.size(); ++r)
        if (g->tr[r]) (void)pdog_destroy(g->tr[r]);
    delete g;
    return PDOG_OK;
}

int pdog_group_create(int ndev, const int *devices, int frame_h, int frame_w, double target_width, int win_h, int win_w,
                      int darker_target, int fill, pdog_group **out)
{
    if (!out) return gfail(PDOG_E_ARG, "pdog_group_create: out is null");
    *out = nullptr;
    if (ndev <= 0 || ndev > 64) return gfail(PDOG_E_ARG, "pdog_group_create: ndev must be 1 … 64");
    int have = 0;
    if (hipGetDeviceCount(&have) != hipSuccess || have <= 0)
        return gfail(PDOG_E_NODEV, "pdog_group_create: no HIP device (this library has no CPU path)");
    if (!rccl().error.empty() || !rccl().handle)
        return gfail(PDOG_E_NODEV, "pdog_group_create: RCCL is not available: " + rccl().error + " — the single-device entry points do not need it");
    pdog_group *g = new pdog_group();
    g->ndev = ndev;
    for (int r = 0; r < ndev; ++r) {
        const int d = devices ? devices[r] : r;
        if (d < 0 || d >= have) {
            delete g;
            return gfail(PDOG_E_NODEV, "pdog_group_create: device ordinal " + std::to_string(d) + " requested, " + std::to_string(have) + " visible");
        }
        for (int q = 0; q < r; ++q)
            if (g->dev[q] == d) { delete g; return gfail(PDOG_E_ARG, "pdog_group_create: a device appears twice (RCCL needs distinct devices)"); }
        g->dev.push_back(d);
    }
    g->tr.assign(ndev, nullptr);
    g->d_local.assign(ndev, nullptr);
    for (int r = 0; r < ndev; ++r) {
        int rc = pdog_create(g->dev[r], frame_h, frame_w, target_width, win_h, win_w, darker_target, fill, &g->tr[r]);
        if (rc) { pdog_group_destroy(g); return rc; } // pdog_last_error() already holds pdog_create's text
    }
    g->comm.assign(ndev, nullptr);
    ncclResult_t nr = rccl().CommInitAll(g->comm.data(), ndev, g->dev.data());
    if (nr != ncclSuccess) {
        g->comm.clear();
        pdog_group_destroy(g);
        return gfail(PDOG_E_HIP, std::string("pdog_group_create: ncclCommInitAll: ") + rccl().GetErrorString(nr));
    }
    *out = g;
    return PDOG_OK;
}

int pdog_group_size(const pdog_group *g) { return g ? g->ndev : 0; }

int pdog_group_tracker(pdog_group *g, int rank, pdog_tracker **out)
{
    if (!g || !out || rank < 0 || rank >= g->ndev) return gfail(PDOG_E_ARG, "pdog_group_tracker: bad argument");
    *out = g->tr[rank];
    return PDOG_OK;
}

int pdog_group_detect_batch(pdog_group *g, const uint8_t *const *d_frames, int64_t frame_stride, int64_t row_stride,
                            const int *n_frames, const int32_t *const *d_frame_index, const int32_t *const *d_guesses,
                            int n_total, int32_t *d_out_ij)
{
    if (!g) return gfail(PDOG_E_ARG, "pdog_group_detect_batch: null group");
    if (n_total == 0) return PDOG_OK;
    if (!d_frames || !n_frames || !d_guesses || !d_out_ij || n_total < 0) return gfail(PDOG_E_ARG, "pdog_group_detect_batch: bad argument");
    const int ndev = g->ndev;
    const int max_n = (n_total + ndev - 1) / ndev;
    const bool equal = n_total % ndev == 0;
    for (int r = 0; r < ndev; ++r) { // every shard's pointers are checked BEFORE anything is launched on any rank
        int lo, hi;
        shard_bounds(n_total, ndev, r, lo, hi);
        if (hi > lo && (!d_frames[r] || !d_guesses[r]))
            return gfail(PDOG_E_ARG, "pdog_group_detect_batch: null frames / guesses pointer for rank " + std::to_string(r) + " (shard of " + std::to_string(hi - lo) + " windows)");
    }
    // (re)size the per-rank result blocks and the root's gather buffer; every stream is drained first
    if (max_n > g->cap_local || (!equal && (long long)max_n * ndev > g->cap_gathered)) {
        for (int r = 0; r < ndev; ++r)
            if (int rc = pdog_sync(g->tr[r])) return rc;
        if (max_n > g->cap_local) {
            for (int r = 0; r < ndev; ++r) {
                G_HIP(hipSetDevice(g->dev[r]));
                if (g->d_local[r]) (void)hipFree(g->d_local[r]);
                g->d_local[r] = nullptr;
            }
            g->cap_local = 0;
            for (int r = 0; r < ndev; ++r) {
                G_HIP(hipSetDevice(g->dev[r]));
                G_HIP(hipMalloc(&g->d_local[r], sizeof(int32_t) * 2 * (size_t)max_n));
                G_HIP(hipMemset(g->d_local[r], 0, sizeof(int32_t) * 2 * (size_t)max_n));
            }
            g->cap_local = max_n;
        }
        if (!equal && (long long)max_n * ndev > g->cap_gathered) {
            G_HIP(hipSetDevice(g->dev[0]));
            if (g->d_gathered) (void)hipFree(g->d_gathered);
            g->d_gathered = nullptr;
            g->cap_gathered = 0;
            G_HIP(hipMalloc(&g->d_gathered, sizeof(int32_t) * 2 * (size_t)max_n * ndev));
            g->cap_gathered = max_n * ndev;
        }
    }
    // every rank: the ordinary single-device batch on its own shard
    std::vector<hipStream_t> st(ndev);
    for (int r = 0; r < ndev; ++r) {
        int lo, hi;
        pdog_group_shard(g, n_total, r, &lo, &hi);
        void *s = nullptr;
        if (int rc = pdog_get_stream(g->tr[r], &s)) return rc;
        st[r] = (hipStream_t)s;
        if (hi > lo) {
            int rc = pdog_detect_batch(g->tr[r], d_frames[r], frame_stride, row_stride, n_frames[r],
                                       d_frame_index ? d_frame_index[r] : nullptr, d_guesses[r], hi - lo, g->d_local[r], nullptr);
            if (rc) return rc;
        }
    }
    // one gather of the (row, col) pairs to the root, each rank's part enqueued behind its own kernels
    int32_t *recv = equal ? d_out_ij : g->d_gathered;
    G_NCCL(rccl().GroupStart());
    for (int r = 0; r < ndev; ++r) {
        ncclResult_t nr = rccl().Gather(g->d_local[r], recv, (size_t)2 * max_n, ncclInt32, 0, g->comm[r], st[r]);
        if (nr != ncclSuccess) {
            (void)rccl().GroupEnd();
            return gfail(PDOG_E_HIP, std::string("pdog_group_detect_batch: ncclGather: ") + rccl().GetErrorString(nr));
        }
    }
    G_NCCL(rccl().GroupEnd());
    if (!equal) {
        G_HIP(hipSetDevice(g->dev[0]));
        hipLaunchKernelGGL(group_compact_kernel, dim3((n_total + 255) / 256), dim3(256), 0, st[0], (const int32_t *)g->d_gathered, d_out_ij,
                           n_total, ndev, max_n);
        G_HIP(hipGetLastError());
    }
    return PDOG_OK;
}

int pdog_group_sync(pdog_group *g)
{
    if (!g) return gfail(PDOG_E_ARG, "pdog_group_sync: null group");
    // EVERY rank is drained (and its raised flags cleared) whatever the others report; the first error is returned
    int first = PDOG_OK;
    std::string text;
    for (int r = 0; r < g->ndev; ++r) {
        const int rc = pdog_sync(g->tr[r]);
        if (rc && !first) { first = rc; text = "rank " + std::to_string(r) + ": " + pdog_last_error(); }
    }
    return first ? gfail(first, text) : PDOG_OK;
}

// Test hook (tests/test_gpu_group.py): the copy kernel that compacts the gathered max-shard-sized blocks when the shards are
// unequal — a path no one-GPU box reaches through pdog_group_detect_batch (with one rank the shards are always equal).
// d_gathered: int32[ndev][max_n][2] with max_n = ⌈n_total / ndev⌉; d_out: int32[n_total][2]; both on the current device.
int pdog_group_test_compact(const int32_t *d_gathered, int n_total, int ndev, int32_t *d_out)
{
    if (!d_gathered || !d_out || n_total <= 0 || ndev <= 0) return gfail(PDOG_E_ARG, "pdog_group_test_compact: bad argument");
    const int max_n = (n_total + ndev - 1) / ndev;
    hipLaunchKernelGGL(group_compact_kernel, dim3((n_total + 255) / 256), dim3(256), 0, 0, d_gathered, d_out, n_total, ndev, max_n);
    G_HIP(hipGetLastError());
    G_HIP(hipStreamSynchronize(0));
    return PDOG_OK;
}

} // extern "C"
