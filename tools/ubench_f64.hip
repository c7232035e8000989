// Micro-benchmark: issue rate of the Float64 ops the refinement uses (v_fma_f64, v_add_f64, v_cvt_f64_i32,
// v_cvt_f64_f32, v_cvt_f32_ubyte0) on gfx950.  Build: hipcc --offload-arch=gfx950 -O3 -o ubench_f64 ubench_f64.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ void k(double *out, int iters, double seed)
{
    double a[8];
    int n[8];
    float f[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x; n[i] = (int)threadIdx.x + i; f[i] = (float)i + (float)seed; }
    const double s = seed * 0.5;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(a[i]) : "v"(s));
                if (MODE == 1) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(s));
                if (MODE == 2) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a[i]) : "v"(n[i]));
                if (MODE == 3) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[i]) : "v"(f[i]));
                if (MODE == 4) asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(f[i]) : "v"(n[i]));
                if (MODE == 5) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(s));
                if (MODE == 6) asm volatile("v_add_u32 %0, %0, %1" : "+v"(n[i]) : "v"(n[(i + 1) & 7]));
            }
        }
    }
    double r = 0;
    for (int i = 0; i < 8; ++i) r += a[i] + n[i] + f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE>
void run(const char *name, int waves_per_simd)
{
    const int threads = 256 * waves_per_simd > 1024 ? 1024 : 256 * waves_per_simd;
    const int blocks = 256 * ((256 * waves_per_simd) / threads);
    const int iters = 5000;
    double *d;
    hipMalloc(&d, sizeof(double) * blocks * threads);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, 100, 1.0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * 64 * waves_per_simd);
    printf("%-18s waves/SIMD=%d  %.3f ms  -> %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name, waves_per_simd, ms, cyc);
    hipFree(d);
}

int main()
{
    for (int w : {1, 2}) {
        run<0>("v_fma_f64", w);
        run<1>("v_add_f64", w);
        run<5>("v_mul_f64", w);
        run<2>("v_cvt_f64_i32", w);
        run<3>("v_cvt_f64_f32", w);
        run<4>("v_cvt_f32_ubyte0", w);
        run<6>("v_add_u32", w);
    }
    return 0;
}
