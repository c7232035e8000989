/*
 * dog_oracle.c — CPU oracle for the PawsomeTracker DoG + argmax hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product
 * path: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may build, load or call it, and there only as the checker / CPU baseline.
 *
 * PARITY UNPINNED.  The reference is Julia (not installed here) and its
 * arithmetic lives in un-vendored third-party packages (ImageFiltering
 * 0.4-0.7, PaddedViews 0.4-0.5, StatsBase 0.24-0.34, FixedPointNumbers; no
 * Manifest.toml, so no pinned versions).  The reference's own test-suite
 * holds no numeric golden vector for this path
 * (test/test-basic-test.jl:139-148 asserts nothing numeric).  This file is a
 * restatement of the published behaviour of those packages, anchored on the
 * reference's call sites; it is cross-checked against an independent
 * NumPy/SciPy statement (oracle/dog_oracle_np.py) and against closed-form
 * known-answer cases (tests/test_oracle.py), not against the reference itself.
 *
 * Every function cites the reference file:line it follows
 * (paths relative to /root/reference).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* src/PawsomeTracker.jl:30  get_sigma(target_width) = target_width / 2sqrt(2log(2))
 * (Julia juxtaposition binds tighter than '/': tw / (2*sqrt(2*log(2)))). */
double pdo_sigma(double target_width)
{
    return target_width / (2.0 * sqrt(2.0 * log(2.0)));
}

/* src/PawsomeTracker.jl:64-68  guess_window_size: l = 4ceil(Int, sigma) + 1 */
int pdo_default_window(double target_width)
{
    return 4 * (int)ceil(pdo_sigma(target_width)) + 1;
}

/* src/PawsomeTracker.jl:43  Kernel.DoG(sigma) (ImageFiltering): the wide
 * Gaussian has sigma_m = sqrt(2)*sigma and fixes the length of both factors,
 * l = 4*ceil(sigma_m) + 1 (KernelFactors.gaussian default length). */
int pdo_kernel_len(double sigma)
{
    double sm = sigma * sqrt(2.0);
    return 4 * (int)ceil(sm) + 1;
}

/* ImageFiltering KernelFactors.gaussian(sigma, l): g[x] = exp(-x^2/(2 sigma^2)),
 * x = -w..w, then g ./ sum(g) (sum taken left to right). */
void pdo_gaussian_1d(double sigma, int l, double *g)
{
    int w = l >> 1;
    double s = 0.0;
    for (int x = -w; x <= w; ++x) {
        double v = exp(-((double)x * (double)x) / (2.0 * sigma * sigma));
        g[x + w] = v;
    }
    for (int i = 0; i < l; ++i) s += g[i];
    for (int i = 0; i < l; ++i) g[i] = g[i] / s;
}

/* src/PawsomeTracker.jl:41-43  kernel = direction * Kernel.DoG(sigma):
 * K = g_sigma (x) g_sigma - g_sigma_m (x) g_sigma_m, dense l x l, column-major
 * (K[i + l*j], i = first/row index), times -1 when darker_target. */
void pdo_dog_kernel(double sigma, int darker, int l, double *K)
{
    double *gp = (double *)malloc(sizeof(double) * (size_t)l);
    double *gm = (double *)malloc(sizeof(double) * (size_t)l);
    pdo_gaussian_1d(sigma, l, gp);
    pdo_gaussian_1d(sigma * sqrt(2.0), l, gm);
    double dir = darker ? -1.0 : 1.0;
    for (int j = 0; j < l; ++j)
        for (int i = 0; i < l; ++i)
            K[i + (size_t)l * j] = dir * (gp[i] * gp[j] - gm[i] * gm[j]);
    free(gp);
    free(gm);
}

/* src/PawsomeTracker.jl:47  fillvalue = mode(_img) (StatsBase.mode): scan the
 * h x w view in its iteration order (column-major: row index fastest), count
 * occurrences, and keep the value whose count FIRST exceeds the running
 * maximum count.  `img` is the raw row-major buffer (pixel (i,j) at
 * img[i*row_stride + j]), which is what the PermutedDimsArray wraps
 * (src/PawsomeTracker.jl:36). */
int pdo_mode_u8(const uint8_t *img, int h, int w, int64_t row_stride)
{
    int64_t cnt[256];
    memset(cnt, 0, sizeof cnt);
    int64_t mc = 0;
    int mv = img[0];
    for (int j = 0; j < w; ++j)
        for (int i = 0; i < h; ++i) {
            int v = img[(int64_t)i * row_stride + j];
            int64_t c = ++cnt[v];
            if (c > mc) { mc = c; mv = v; }
        }
    return mv;
}

/* PaddedViews.PaddedView(fill, img, pad_indices), src/PawsomeTracker.jl:48:
 * A[i,j] = in-bounds ? img[i,j] : fill.  (i,j) are 1-based here. */
static inline double padded_read(const uint8_t *img, int h, int w, int64_t stride,
                                 int fill, int i, int j)
{
    int v = (i >= 1 && i <= h && j >= 1 && j <= w) ? img[(int64_t)(i - 1) * stride + (j - 1)] : fill;
    /* FixedPointNumbers N0f8 -> Float64 is raw / 255 (a division). */
    return (double)v / 255.0;
}

/* The Tracker functor, src/PawsomeTracker.jl:55-62.
 *   :56  window = (g1-r1 : g1+r1, g2-r2 : g2+r2)
 *   :57  imfilter!(CPUThreads(FIR), buff, img, kernel, NoPad(), window):
 *        buff[I] = sum_J img[I+J] * K[J], a correlation accumulated from 0.0
 *        in kernel column-major order (first index fastest), Float64
 *   :58-59 findmax over the window view: first maximum in column-major order
 *   :60  window-local -> absolute index
 *   :61  clamp to [1, sz]
 * frame: raw row-major h x w u8, guess/out 1-based (row, col).
 * resp (optional): (2r1+1) x (2r2+1), column-major, the values findmax sees.
 * Threads split the window columns, like CPUThreads splits the output range. */
void pdo_detect_dense(const uint8_t *frame, int h, int w, int64_t stride, int fill,
                      const double *K, int l, int r1, int r2, int g1, int g2,
                      int *out_i, int *out_j, double *resp, int nthreads)
{
    const int hw = l >> 1;
    const int n1 = 2 * r1 + 1, n2 = 2 * r2 + 1;
    const int i0 = g1 - r1, j0 = g2 - r2; /* absolute index of window (1,1) */
    /* materialise the padded tile once (values are identical to lazy reads) */
    const int t1 = n1 + 2 * hw, t2 = n2 + 2 * hw;
    double *tile = (double *)malloc(sizeof(double) * (size_t)t1 * (size_t)t2);
    for (int b = 0; b < t2; ++b)
        for (int a = 0; a < t1; ++a)
            tile[a + (size_t)t1 * b] = padded_read(frame, h, w, stride, fill, i0 - hw + a, j0 - hw + b);
    double *out = resp ? resp : (double *)malloc(sizeof(double) * (size_t)n1 * (size_t)n2);
#ifdef _OPENMP
    if (nthreads < 1) nthreads = omp_get_max_threads();
#pragma omp parallel for schedule(static) num_threads(nthreads)
#endif
    for (int jj = 0; jj < n2; ++jj) {
        for (int ii = 0; ii < n1; ++ii) {
            double tmp = 0.0;
            for (int kj = 0; kj < l; ++kj) {
                const double *tcol = tile + (size_t)t1 * (jj + kj) + ii;
                const double *kcol = K + (size_t)l * kj;
                for (int ki = 0; ki < l; ++ki) tmp += tcol[ki] * kcol[ki];
            }
            out[ii + (size_t)n1 * jj] = tmp;
        }
    }
    /* findmax: first maximum in column-major order */
    double best = out[0];
    int bi = 0, bj = 0;
    for (int jj = 0; jj < n2; ++jj)
        for (int ii = 0; ii < n1; ++ii) {
            double v = out[ii + (size_t)n1 * jj];
            if (v > best) { best = v; bi = ii; bj = jj; }
        }
    int ai = i0 + bi, aj = j0 + bj;
    if (ai < 1) ai = 1;
    if (ai > h) ai = h;
    if (aj < 1) aj = 1;
    if (aj > w) aj = w;
    *out_i = ai;
    *out_j = aj;
    free(tile);
    if (!resp) free(out);
}

/* Float64 separable variant of the same response (NOT how the reference runs
 * it: ImageFiltering keeps the rank-2 DoG as one dense kernel).  Used only to
 * separate algorithmic from hardware speed-up in bench.py's report and as a
 * second statement in tests.  Row pass along j (contiguous), column pass
 * along i; D = sum_i gp[ki] Rp[i+ki] - sum_i gm[ki] Rm[i+ki]. */
void pdo_detect_separable(const uint8_t *frame, int h, int w, int64_t stride, int fill,
                          double sigma, int darker, int l, int r1, int r2, int g1, int g2,
                          int *out_i, int *out_j, double *resp, int nthreads)
{
    const int hw = l >> 1;
    const int n1 = 2 * r1 + 1, n2 = 2 * r2 + 1;
    const int i0 = g1 - r1, j0 = g2 - r2;
    const int t1 = n1 + 2 * hw;
    double *gp = (double *)malloc(sizeof(double) * (size_t)l);
    double *gm = (double *)malloc(sizeof(double) * (size_t)l);
    pdo_gaussian_1d(sigma, l, gp);
    pdo_gaussian_1d(sigma * sqrt(2.0), l, gm);
    double dir = darker ? -1.0 : 1.0;
    double *Rp = (double *)malloc(sizeof(double) * (size_t)t1 * (size_t)n2);
    double *Rm = (double *)malloc(sizeof(double) * (size_t)t1 * (size_t)n2);
    double *out = resp ? resp : (double *)malloc(sizeof(double) * (size_t)n1 * (size_t)n2);
#ifdef _OPENMP
    if (nthreads < 1) nthreads = omp_get_max_threads();
#pragma omp parallel num_threads(nthreads)
#endif
    {
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
        for (int a = 0; a < t1; ++a) {
            const int ai = i0 - hw + a;
            for (int jj = 0; jj < n2; ++jj) {
                double sp = 0.0, sm = 0.0;
                for (int k = 0; k < l; ++k) {
                    double v = padded_read(frame, h, w, stride, fill, ai, j0 + jj - hw + k);
                    sp += v * gp[k];
                    sm += v * gm[k];
                }
                Rp[a + (size_t)t1 * jj] = sp;
                Rm[a + (size_t)t1 * jj] = sm;
            }
        }
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
        for (int jj = 0; jj < n2; ++jj)
            for (int ii = 0; ii < n1; ++ii) {
                double sp = 0.0, sm = 0.0;
                const double *cp = Rp + (size_t)t1 * jj + ii;
                const double *cm = Rm + (size_t)t1 * jj + ii;
                for (int k = 0; k < l; ++k) { sp += cp[k] * gp[k]; sm += cm[k] * gm[k]; }
                out[ii + (size_t)n1 * jj] = dir * (sp - sm);
            }
    }
    double best = out[0];
    int bi = 0, bj = 0;
    for (int jj = 0; jj < n2; ++jj)
        for (int ii = 0; ii < n1; ++ii) {
            double v = out[ii + (size_t)n1 * jj];
            if (v > best) { best = v; bi = ii; bj = jj; }
        }
    int ai = i0 + bi, aj = j0 + bj;
    if (ai < 1) ai = 1;
    if (ai > h) ai = h;
    if (aj < 1) aj = 1;
    if (aj > w) aj = w;
    *out_i = ai;
    *out_j = aj;
    free(gp); free(gm); free(Rp); free(Rm);
    if (!resp) free(out);
}

/* Batched convenience: n independent applications of the functor
 * (src/PawsomeTracker.jl:55-62), frame b at frames + b*frame_stride. */
void pdo_detect_batch_dense(const uint8_t *frames, int64_t frame_stride, int n, int h, int w,
                            int64_t stride, int fill, const double *K, int l, int r1, int r2,
                            const int32_t *guesses, int32_t *out_ij, int nthreads)
{
    for (int b = 0; b < n; ++b) {
        int oi, oj;
        pdo_detect_dense(frames + (int64_t)b * frame_stride, h, w, stride, fill, K, l, r1, r2,
                         guesses[2 * b], guesses[2 * b + 1], &oi, &oj, NULL, nthreads);
        out_ij[2 * b] = oi;
        out_ij[2 * b + 1] = oj;
    }
}

/* The same n applications with the threads split ACROSS windows (one window per thread, each window
 * single-threaded): not how one reference `Tracker` call threads (CPUThreads splits the output range of a
 * single call, as pdo_detect_dense does), but how a host with many clips would use its cores
 * (README.md:214: concurrent `track` calls are supported) — the stronger CPU baseline for batches.
 * separable != 0 runs the separable Float64 statement instead of the dense one. */
void pdo_detect_batch_par(const uint8_t *frames, int64_t frame_stride, int n, int h, int w,
                          int64_t stride, int fill, const double *K, double sigma, int darker, int l,
                          int r1, int r2, const int32_t *guesses, int32_t *out_ij, int separable, int nthreads)
{
#ifdef _OPENMP
    if (nthreads < 1) nthreads = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
#endif
    for (int b = 0; b < n; ++b) {
        int oi, oj;
        const uint8_t *f = frames + (int64_t)b * frame_stride;
        if (separable)
            pdo_detect_separable(f, h, w, stride, fill, sigma, darker, l, r1, r2,
                                 guesses[2 * b], guesses[2 * b + 1], &oi, &oj, NULL, 1);
        else
            pdo_detect_dense(f, h, w, stride, fill, K, l, r1, r2,
                             guesses[2 * b], guesses[2 * b + 1], &oi, &oj, NULL, 1);
        out_ij[2 * b] = oi;
        out_ij[2 * b + 1] = oj;
    }
}

int pdo_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
