// fp64_mfma.hip -- v_mfma_f64_16x16x4_f64 on one MI355X: its rate alone, and whether it runs BESIDE v_fma_f64 (in the same
// wave; in different waves of a SIMD) or shares the rate with it.  4 waves per SIMD (1024-thread workgroups, one per CU).
//   hipcc -O3 --offload-arch=gfx950 tools/fp64_mfma.hip -o tools/fp64_mfma && tools/fp64_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
typedef double d4 __attribute__((ext_vector_type(4)));

// MODE 0: MFMA only (NACC independent accumulator tiles per wave); 1: v_fma_f64 only (8 chains); 2: both in every wave
// (one MFMA, then VPER vector FMAs); 3: even waves MFMA only, odd waves vector only
template <int MODE, int NACC, int VPER>
__global__ __launch_bounds__(1024) void k_mix(double *out, double seed, int iters)
{
    d4 acc[NACC];
    double va[8];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int i = 0; i < 8; ++i) va[i] = seed + i;
    const double a = seed + threadIdx.x * 1e-9, b = seed * 0.5;
    const bool mf = MODE == 0 || MODE == 2 || (MODE == 3 && ((threadIdx.x >> 6) & 1) == 0);
    const bool vf = MODE == 1 || MODE == 2 || (MODE == 3 && ((threadIdx.x >> 6) & 1) == 1);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (mf) {                                  // wave-uniform
#pragma unroll
                for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
            }
            if (vf) {
#pragma unroll
                for (int r = 0; r < (MODE == 2 ? VPER * NACC : 16 * NACC); ++r) va[r & 7] = __builtin_fma(va[r & 7], a, b);
            }
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += va[i];
    if (s == 12345.678) out[0] = s;
}

template <int MODE, int NACC, int VPER>
static void run(const char *what, int n_cu)
{
    double *d;
    CHECK(hipMalloc(&d, 8));
    const int iters = 2000;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_mix<MODE, NACC, VPER>), dim3(n_cu), dim3(1024), 0, 0, d, 1.000001, iters);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double waves = (double)n_cu * 16;
    double mfma = 0, vec = 0;                          // FMAs
    const double steps = (double)iters * 8;
    if (MODE == 0) mfma = waves * steps * NACC * 1024;
    if (MODE == 1) vec = waves * steps * 16 * NACC * 64;
    if (MODE == 2) { mfma = waves * steps * NACC * 1024; vec = waves * steps * VPER * NACC * 64; }
    if (MODE == 3) { mfma = waves / 2 * steps * NACC * 1024; vec = waves / 2 * steps * 16 * NACC * 64; }
    std::printf("%-58s %8.3f ms   matrix %6.1f + vector %6.1f = %6.1f TFLOP/s\n", what, best, 2 * mfma / (best * 1e-3) / 1e12,
                2 * vec / (best * 1e-3) / 1e12, 2 * (mfma + vec) / (best * 1e-3) / 1e12);
    CHECK(hipFree(d));
}

int main()
{
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int n = p.multiProcessorCount;
    run<0, 1, 0>("MFMA only, 1 accumulator tile per wave", n);
    run<0, 2, 0>("MFMA only, 2 accumulator tiles per wave", n);
    run<0, 4, 0>("MFMA only, 4 accumulator tiles per wave", n);
    run<1, 4, 0>("v_fma_f64 only", n);
    run<2, 4, 4>("same wave: 1 MFMA + 4 v_fma_f64 (0.25 of its FMAs)", n);
    run<2, 4, 8>("same wave: 1 MFMA + 8 v_fma_f64 (0.5)", n);
    run<2, 4, 16>("same wave: 1 MFMA + 16 v_fma_f64 (equal FMAs)", n);
    run<3, 4, 0>("even waves MFMA only, odd waves v_fma_f64 only", n);
    return 0;
}
