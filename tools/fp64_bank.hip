// fp64_bank.hip -- does v_fmac_f64 slow down when its two VGPR operands (accumulator pair and multiplicand pair) sit in
// the same register banks?  Explicit registers through inline asm; 4 waves per SIMD like k_profile_lib.
//   hipcc -O3 --offload-arch=gfx950 tools/fp64_bank.hip -o tools/fp64_bank && tools/fp64_bank
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

// 5 accumulators v[10:11] .. v[18:19]; multiplicands start at v[20 + SHIFT]: SHIFT 0 -> the pair of accumulator i and
// its multiplicand are 10 registers apart (banks differ by 2 mod 4: no overlap); SHIFT 2 -> 12 apart (same banks)
#define FMA5(a0, a1, a2, a3, a4) \
    "v_fmac_f64 v[10:11], s[20:21], v[" a0 "]\n v_fmac_f64 v[12:13], s[20:21], v[" a1 "]\n v_fmac_f64 v[14:15], s[20:21], v[" a2 "]\n" \
    "v_fmac_f64 v[16:17], s[20:21], v[" a3 "]\n v_fmac_f64 v[18:19], s[20:21], v[" a4 "]\n"

template <int MODE>
__global__ __launch_bounds__(1024) void k_bank(double *out, int iters)
{
    double r = 0;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0)        // accumulator banks (2,3),(0,1),(2,3),(0,1),(2,3); multiplicand in the OTHER banks
            asm volatile(FMA5("20:21", "22:23", "24:25", "26:27", "28:29") FMA5("20:21", "22:23", "24:25", "26:27", "28:29")
                         FMA5("20:21", "22:23", "24:25", "26:27", "28:29") FMA5("20:21", "22:23", "24:25", "26:27", "28:29")
                         FMA5("20:21", "22:23", "24:25", "26:27", "28:29") FMA5("20:21", "22:23", "24:25", "26:27", "28:29")
                         FMA5("20:21", "22:23", "24:25", "26:27", "28:29") FMA5("20:21", "22:23", "24:25", "26:27", "28:29")
                         ::: "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25",
                         "v26", "v27", "v28", "v29", "v30", "v31", "s20", "s21");
        else                  // multiplicand in the SAME banks as its accumulator
            asm volatile(FMA5("22:23", "24:25", "26:27", "28:29", "30:31") FMA5("22:23", "24:25", "26:27", "28:29", "30:31")
                         FMA5("22:23", "24:25", "26:27", "28:29", "30:31") FMA5("22:23", "24:25", "26:27", "28:29", "30:31")
                         FMA5("22:23", "24:25", "26:27", "28:29", "30:31") FMA5("22:23", "24:25", "26:27", "28:29", "30:31")
                         FMA5("22:23", "24:25", "26:27", "28:29", "30:31") FMA5("22:23", "24:25", "26:27", "28:29", "30:31")
                         ::: "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25",
                         "v26", "v27", "v28", "v29", "v30", "v31", "s20", "s21");
    }
    if (iters < 0) out[0] = r;
}

template <int MODE>
static void run(int n_cu)
{
    double *d;
    CHECK(hipMalloc(&d, 8));
    const int iters = 20000;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_bank<MODE>, dim3(n_cu), dim3(1024), 0, 0, d, iters);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double fma = (double)n_cu * 1024 * iters * 40;
    std::printf("%s: %.3f ms, %.1f TFLOP/s\n", MODE == 0 ? "operands in different banks" : "operands in the same banks     ", best,
                2 * fma / (best * 1e-3) / 1e12);
    CHECK(hipFree(d));
}

int main()
{
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    run<0>(p.multiProcessorCount);
    run<1>(p.multiProcessorCount);
    run<0>(p.multiProcessorCount);
    run<1>(p.multiProcessorCount);
    return 0;
}
