// Micro-benchmark: issue rate of v_fma_f32 vs v_pk_fma_f32 vs v_pk_add/v_add on gfx950,
// at 1/2/4 waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 -o ubench_valu ubench_valu.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void k(float *out, int iters, float seed)
{
    f2 a[8];
    for (int i = 0; i < 8; ++i) a[i] = f2{seed + i + threadIdx.x, seed - i};
    f2 t = {seed * 0.5f, seed * 0.25f};
    float s = seed * 0.125f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) { a[i].x = __builtin_fmaf(a[i].x, s, a[i].y); }                 // v_fma_f32
                if (MODE == 1) { a[i] = __builtin_elementwise_fma(a[i], t, a[i]); }           // v_pk_fma_f32
                if (MODE == 2) { a[i] = a[i] + t; }                                            // v_pk_add_f32
                if (MODE == 3) { a[i].x = a[i].x + s; }                                        // v_add_f32
                if (MODE == 4) { f2 b = {a[i].x, a[i].x}; a[i] = __builtin_elementwise_fma(b, t, a[i]); } // pk_fma bcast
                if (MODE == 5) { asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i].x) : "s"(s), "v"(a[(i + 1) & 7].y)); } // SGPR operand
                if (MODE == 6) { asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i].x) : "v"(s), "v"(a[(i + 1) & 7].y)); } // all VGPR
                if (MODE == 7) { asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "s"(t)); } // pk with SGPR pair
            }
        }
    }
    float r = 0;
    for (int i = 0; i < 8; ++i) r += a[i].x + a[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE>
void run(const char *name, int waves_per_simd)
{
    const int threads = 256 * waves_per_simd > 1024 ? 1024 : 256 * waves_per_simd;
    const int blocks_per_cu = (256 * waves_per_simd) / threads;
    const int blocks = 256 * blocks_per_cu;
    const int iters = 20000;
    float *d;
    hipMalloc(&d, sizeof(float) * blocks * threads);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, 100, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double insts_per_wave = (double)iters * 64;
    const double waves_per_simd_total = waves_per_simd;  // resident per SIMD
    // cycles per instruction per SIMD assuming 2.4 GHz
    const double cyc = ms * 1e-3 * 2.4e9 / (insts_per_wave * waves_per_simd_total);
    printf("%-14s waves/SIMD=%d  %.3f ms  -> %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name, waves_per_simd, ms, cyc);
    hipFree(d);
}

int main()
{
    for (int w : {1, 2, 4}) {
        run<0>("v_fma_f32", w);
        run<1>("v_pk_fma_f32", w);
        run<4>("v_pk_fma bcast", w);
        run<2>("v_pk_add_f32", w);
        run<3>("v_add_f32", w);
        run<5>("v_fmac v,s,v", w);
        run<6>("v_fmac v,v,v", w);
        run<7>("v_pk_fma v,s2", w);
    }
    return 0;
}
