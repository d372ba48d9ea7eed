// cvt_rate.hip -- issue cost of v_cvt_f64_f32 next to v_fma_f64 and v_add_f64 on gfx950 (is the float -> double widening of a profile
// row, 22 per window in k_profile_fixed, a full-rate instruction?).   hipcc -O3 --offload-arch=gfx950 tools/cvt_rate.hip -o tools/cvt_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed)
{
    float f[8];
    double d[8];
    for (int i = 0; i < 8; ++i) { f[i] = seed + threadIdx.x + i; d[i] = f[i]; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(f[i]));
                if (OP == 1) asm volatile("v_fma_f64 %0, %1, %1, %0" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
                if (OP == 2) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
                if (OP == 3) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(d[i]));
                if (OP == 4) asm volatile("v_lshl_add_u32 %0, %1, 3, %0" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));
            }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += f[i] + (float)d[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int OP> static void run(const char *name, float *out)
{
    const int iters = 2000, grid = 256 * 8;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double insts = (double)grid * 4 /* waves */ * iters * 32;
    std::printf("%-16s %.3f ms  %.2f wave-instructions per ns chip-wide = %.2f cycles per instruction and SIMD at 2.1 GHz\n", name, ms,
                insts / (ms * 1e6), 1024.0 * 2.1 / (insts / (ms * 1e6)));
}
int main()
{
    float *out; hipMalloc(&out, 256 * 8 * 256 * 4);
    run<0>("v_cvt_f64_f32", out); run<1>("v_fma_f64", out); run<2>("v_add_f64", out); run<3>("v_cvt_f32_f64", out); run<4>("v_lshl_add_u32", out);
    return 0;
}
