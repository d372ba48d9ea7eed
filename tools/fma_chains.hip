// fma_chains.hip -- how many INDEPENDENT v_fma_f64 chains a SIMD needs in flight before the fp64 pipe issues back to back.
// k_profile_fixed gives a wave five chains (one per window of the lane); profiles/r4/NOTES.md found the double-buffered form
// (two waves per SIMD) short of fp64 issue.  This measures the dependent-issue latency directly: CH chains per lane x
// W waves per SIMD, the time per FMA wave-instruction in cycles of s_memtime's clock rescaled by the 1-wave / 16-chain row.
//   hipcc -O3 --offload-arch=gfx950 tools/fma_chains.hip -o /tmp/fma_chains && /tmp/fma_chains
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

constexpr int INNER = 960;      // FMAs per lane and outer iteration: a multiple of every chain count below

template <int CH>
__global__ __launch_bounds__(256) void k_chains(double *out, double seed, int iters)
{
    double acc[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) acc[c] = seed + c + threadIdx.x;
    const double k1 = seed * 0.5, k2 = seed * 0.25;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < INNER / CH; ++u) {
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_fma(acc[c], k1, k2);
        }
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) s += acc[c];
    if (s == 12345.678) out[0] = s;
}

template <int CH>
static double run(int waves_per_simd, int n_cu, double *d)
{
    const int block = 256, grid = n_cu * waves_per_simd, iters = 300;      // a 256-thread block = one wave on each SIMD of a CU
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_chains<CH>, dim3(grid), dim3(block), 0, 0, d, 1.000001, iters);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    // wave-instructions per SIMD = waves_per_simd * iters * INNER; ns per wave-instruction on one SIMD:
    return (double)best * 1e6 / ((double)waves_per_simd * iters * INNER);
}

int main()
{
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int n_cu = p.multiProcessorCount;
    double *d;
    CHECK(hipMalloc(&d, 8));
    std::printf("%s, %d CUs: ns per v_fma_f64 wave-instruction and SIMD (lower = closer to back-to-back issue)\n", p.name, n_cu);
    std::printf("%-8s %10s %10s %10s %10s\n", "chains", "1 wave", "2 waves", "3 waves", "4 waves");
#define ROW(CH) std::printf("%-8d %10.3f %10.3f %10.3f %10.3f\n", CH, run<CH>(1, n_cu, d), run<CH>(2, n_cu, d), run<CH>(3, n_cu, d), run<CH>(4, n_cu, d))
    ROW(1);
    ROW(2);
    ROW(3);
    ROW(4);
    ROW(5);
    ROW(6);
    ROW(8);
    ROW(10);
    ROW(12);
    ROW(16);
#undef ROW
    CHECK(hipFree(d));
    return 0;
}
