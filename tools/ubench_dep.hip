// v_pk_fma_f32 throughput vs number of independent accumulator chains (dependency distance),
// with an SGPR-pair tap operand and an op_sel-broadcast VGPR operand like the column pass uses.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int NCH>
__global__ void k(float *out, int iters, float seed)
{
    f2 a[NCH];
    for (int i = 0; i < NCH; ++i) a[i] = f2{seed + i + threadIdx.x, seed - i};
    f2 r = {seed * 0.5f + threadIdx.x, seed * 0.25f};
    f2 t = {seed * 0.125f, seed * 0.0625f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 64 / NCH; ++rep)
#pragma unroll
            for (int i = 0; i < NCH; ++i)
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(a[i]) : "v"(r), "s"(t));
    }
    float s = 0;
    for (int i = 0; i < NCH; ++i) s += a[i].x + a[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NCH>
void run(int wps)
{
    const int threads = 256, blocks = 256 * wps, iters = 20000;
    float *d; hipMalloc(&d, 4 * blocks * threads);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NCH>, dim3(blocks), dim3(threads), 0, 0, d, 100, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NCH>, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("chains=%2d waves/SIMD=%d: %.2f cycles per pk_fma per SIMD (at 2.35 GHz)\n", NCH, wps, ms * 1e-3 * 2.35e9 / ((double)iters * 64 * wps));
    hipFree(d);
}
int main()
{
    for (int w : {1, 2, 3}) { run<1>(w); run<2>(w); run<3>(w); run<4>(w); run<8>(w); run<16>(w); }
}
