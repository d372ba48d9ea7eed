// fp64_ilp.hip -- how much of the v_fma_f64 rate a given (waves per SIMD, independent chains per lane) sustains on one
// MI355X: the question behind k_profile_lib's 64 % (4 waves per SIMD, 5 accumulators per lane, SGPR multiplier).
//   hipcc -O3 --offload-arch=gfx950 tools/fp64_ilp.hip -o /tmp/fp64_ilp && /tmp/fp64_ilp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

// CHAINS accumulators per lane, each updated by 7 FMAs per "step" with 7 wave-uniform multipliers (SGPRs) and 7 per-lane
// values: the shape of one k_profile_lib step
template <int CHAINS, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_ilp(double *out, const double *__restrict__ coef, int iters)
{
    double acc[CHAINS], row[CHAINS][7];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) {
        acc[c] = 0.0;
#pragma unroll
        for (int k = 0; k < 7; ++k) row[c][k] = 1.0 + threadIdx.x * 1e-6 + c + k;
    }
    const __attribute__((address_space(4))) double *cf = (const __attribute__((address_space(4))) double *)coef;
    double C[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) C[k] = cf[k];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 12; ++u) {
#pragma unroll
            for (int k = 0; k < 7; ++k) {
#pragma unroll
                for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_fma(row[c][k], C[k], acc[c]);
            }
        }
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) s += acc[c];
    if (s == 12345.678) out[0] = s;
}

template <int CHAINS, int BLOCK>
static void run(int n_cu, int blocks_per_cu, double *coef)
{
    double *d;
    CHECK(hipMalloc(&d, 8));
    const int grid = n_cu * blocks_per_cu, iters = 2000;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_ilp<CHAINS, BLOCK>), dim3(grid), dim3(BLOCK), 0, 0, d, coef, iters);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double fma = (double)grid * BLOCK * iters * 12 * 7 * CHAINS;
    std::printf("chains %d, block %4d x %d per CU = %2d waves per SIMD: %.3f ms, %.1f TFLOP/s\n", CHAINS, BLOCK, blocks_per_cu,
                BLOCK / 64 * blocks_per_cu / 4, best, 2 * fma / (best * 1e-3) / 1e12);
    CHECK(hipFree(d));
}

int main()
{
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int n = p.multiProcessorCount;
    double h[7] = {0.5, 0.25, 0.125, 0.0625, 0.03125, 0.015625, 0.0078125}, *coef;
    CHECK(hipMalloc(&coef, sizeof(h)));
    CHECK(hipMemcpy(coef, h, sizeof(h), hipMemcpyHostToDevice));
    std::printf("%s, %d CUs\n", p.gcnArchName, n);
    run<5, 1024>(n, 1, coef);
    run<5, 256>(n, 4, coef);
    run<5, 256>(n, 8, coef);
    run<5, 256>(n, 2, coef);
    run<5, 256>(n, 1, coef);
    run<7, 1024>(n, 1, coef);
    run<8, 256>(n, 4, coef);
    run<8, 256>(n, 8, coef);
    run<3, 1024>(n, 1, coef);
    run<2, 1024>(n, 1, coef);
    run<1, 1024>(n, 1, coef);
    return 0;
}
