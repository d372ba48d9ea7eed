// fp64_peak.hip -- what one MI355X sustains in v_fma_f64 (and v_add_f64 / v_cvt_f64_f32), measured:
// the denominator of the "fp64 issue floor" figures in DESIGN.md (SURVEY 8d asks for a measured one).
//   hipcc -O3 --offload-arch=gfx950 tools/fp64_peak.hip -o /tmp/fp64_peak && /tmp/fp64_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

constexpr int CHAINS = 8;       // independent accumulators per lane: hides the FMA latency inside one wave
constexpr int INNER = 512;

template <int OP>
__global__ __launch_bounds__(256) void k_issue(double *out, double seed, float fseed, int iters)
{
    double acc[CHAINS];
    float facc[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) {
        acc[c] = seed + c + threadIdx.x;
        facc[c] = fseed + c;
    }
    const double k1 = seed * 0.5, k2 = seed * 0.25;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < INNER / CHAINS; ++u) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) {
                if (OP == 0) acc[c] = __builtin_fma(acc[c], k1, k2);                 // v_fma_f64
                if (OP == 1) acc[c] = acc[c] + k2;                                    // v_add_f64
                if (OP == 2) asm volatile("v_cvt_f64_f32_e32 %0, %1" : "=v"(acc[c]) : "v"(facc[c]));   // v_cvt_f64_f32
            }
        }
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) s += acc[c] + facc[c];
    if (s == 12345.678) out[0] = s;      // never true: keeps the chains alive
}

template <int OP>
static void run(const char *name, int flop_per_op, int n_cu)
{
    double *d;
    CHECK(hipMalloc(&d, 8));
    const int grid = n_cu * 8, block = 256, iters = 400;     // 8 waves per SIMD
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_issue<OP>, dim3(grid), dim3(block), 0, 0, d, 1.000001, 1.5f, iters);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double ops = (double)grid * block * iters * INNER;
        const double per_s = ops / (ms * 1e-3);
        std::printf("%-16s rep %d: %.3f ms, %.2f T lane-ops/s", name, rep, ms, per_s / 1e12);
        if (flop_per_op) std::printf(" = %.1f TFLOP/s", per_s * flop_per_op / 1e12);
        // lane-ops per cycle per CU at an assumed clock is not derivable without the clock; print per-CU rate
        std::printf("  (%.1f G lane-ops/s per CU)\n", per_s / n_cu / 1e9);
    }
    CHECK(hipFree(d));
}

int main()
{
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    std::printf("%s (%s), %d CUs, clockRate %d kHz\n", p.name, p.gcnArchName, p.multiProcessorCount, p.clockRate);
    run<0>("v_fma_f64", 2, p.multiProcessorCount);
    run<1>("v_add_f64", 1, p.multiProcessorCount);
    run<2>("v_cvt_f64_f32", 0, p.multiProcessorCount);
    return 0;
}
