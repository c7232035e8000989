// Measures the shader clock the chip sustains under a dense packed-FMA load (all CUs busy) and the
// true cycles per instruction, using s_memtime (shader cycles) against s_memrealtime (100 MHz).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void k(unsigned long long *out, int iters, float seed)
{
    f2 a[8];
    for (int i = 0; i < 8; ++i) a[i] = f2{seed + i + threadIdx.x, seed - i};
    f2 t = {seed * 0.5f, seed * 0.25f};
    unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) a[i] = __builtin_elementwise_fma(a[i], t, a[i]);
                if (MODE == 1) a[i].x = __builtin_fmaf(a[i].x, t.x, a[i].y);
            }
    }
    unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float r = 0;
    for (int i = 0; i < 8; ++i) r += a[i].x + a[i].y;
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = c1 - c0; out[2 * blockIdx.x + 1] = (r1 - r0) + (r == 12345.f); }
}
template <int MODE>
void run(const char *name, int waves_per_simd, int iters)
{
    const int threads = 256, blocks = 256 * waves_per_simd;
    unsigned long long *d;
    hipMalloc(&d, 16 * blocks);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, 1000, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * blocks);
    hipMemcpy(h.data(), d, 16 * blocks, hipMemcpyDeviceToHost);
    std::vector<double> clk, cpi;
    for (int b = 0; b < blocks; ++b) {
        clk.push_back((double)h[2 * b] / ((double)h[2 * b + 1] / 100e6) / 1e9);
        cpi.push_back((double)h[2 * b] / ((double)iters * 64));
    }
    std::sort(clk.begin(), clk.end()); std::sort(cpi.begin(), cpi.end());
    const double wall_cyc = ms * 1e-3 * clk[blocks / 2] * 1e9 / ((double)iters * 64 * waves_per_simd);
    printf("%-10s waves/SIMD=%d: clock %.3f GHz (min %.3f max %.3f); in-kernel cycles/instr/wave %.2f (min %.2f max %.2f) -> per SIMD %.2f; wall %.2f ms -> %.2f cycles/instr per SIMD\n",
           name, waves_per_simd, clk[blocks / 2], clk[0], clk[blocks - 1], cpi[blocks / 2], cpi[0], cpi[blocks - 1], cpi[blocks / 2] / waves_per_simd, ms, wall_cyc);
    hipFree(d);
}
int main()
{
    for (int rep = 0; rep < 1; ++rep)
        for (int w : {1, 2, 3, 4, 8}) {
            run<0>("v_pk_fma", w, 100000);
            run<1>("v_fma", w, 100000);
        }
}
