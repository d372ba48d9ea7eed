// hbm_streams.hip -- how fast can K sequential WRITE (or READ) streams go, depending on where they lie?  One sequential
// stream touches one DRAM bank per channel at a time; tools/hbm_ranks.hip showed two write streams into independent chunks at
// 1.47x the rate of one stream.  Here: chunks of physical memory (hipMemCreate), their pair matrix, then K = 1, 2, 4, 8
// streams of the same TOTAL size (a) side by side inside one chunk region, (b) into K chunks chosen to be mutually independent.
//   hipcc -O3 --offload-arch=gfx950 tools/hbm_streams.hip -o tools/hbm_streams && tools/hbm_streams [chunk MB = 1024] [chunks = 32]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct Streams { u32x4 *p[8]; };

// K streams of n16 vectors each; workgroup b serves stream b % K: its tiles of 256 x 4 vectors advance through the stream
template <bool WRITE>
__global__ __launch_bounds__(256) void k_streams(Streams s, int K, size_t n16)
{
    const int k = blockIdx.x % K;
    const size_t wg = blockIdx.x / K, nwg = gridDim.x / K;
    u32x4 *p = s.p[k];
    const u32x4 v = {1u, 2u, 3u, (uint32_t)threadIdx.x};
    uint32_t acc = 0;
    for (size_t i = wg * 256 + threadIdx.x; i < n16; i += nwg * 256) {
        if (WRITE) __builtin_nontemporal_store(v, p + i);
        else acc ^= __builtin_nontemporal_load(p + i).x;
    }
    if (!WRITE && acc == 0x12345679u) p[0] = v;
}

static hipEvent_t e0, e1;
template <bool WRITE>
static double run_ms(const Streams &s, int K, size_t bytes_each)
{
    double best = 1e30;
    for (int rep = 0; rep < 4; ++rep) {
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_streams<WRITE>, dim3(256 * 16 / K * K), dim3(256), 0, 0, s, K, bytes_each / 16);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) best = std::min(best, (double)ms / 3);
    }
    return best;
}

int main(int argc, char **argv)
{
    const size_t chunk = (size_t)(argc > 1 ? std::atoll(argv[1]) : 1024) << 20;
    const int n = argc > 2 ? std::atoi(argv[2]) : 32;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    void *base = nullptr;
    CHECK(hipMemAddressReserve(&base, chunk * n, 0, nullptr, 0));
    std::vector<hipMemGenericAllocationHandle_t> h(n);
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    for (int i = 0; i < n; ++i) {
        CHECK(hipMemCreate(&h[i], chunk, &prop, 0));
        CHECK(hipMemMap((char *)base + chunk * i, chunk, 0, h[i], 0));
    }
    CHECK(hipMemSetAccess(base, chunk * n, &acc, 1));
    auto at = [&](int i) { return (u32x4 *)((char *)base + chunk * i); };
    // pair matrix (two write streams over whole chunks)
    std::vector<double> L((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i)
        for (int j = i + 1; j < n; ++j) {
            Streams s = {};
            s.p[0] = at(i);
            s.p[1] = at(j);
            L[(size_t)i * n + j] = L[(size_t)j * n + i] = run_ms<true>(s, 2, chunk);
        }
    double lo = 1e30;
    for (int i = 0; i < n; ++i) for (int j = i + 1; j < n; ++j) lo = std::min(lo, L[(size_t)i * n + j]);
    // greedy set of mutually independent chunks
    std::vector<int> set = {0};
    while ((int)set.size() < 8) {
        int best = -1;
        double best_c = 1e300;
        for (int c = 0; c < n; ++c) {
            if (std::find(set.begin(), set.end(), c) != set.end()) continue;
            double worst = 0;
            for (int x : set) worst = std::max(worst, L[(size_t)c * n + x]);
            if (worst < best_c) { best_c = worst; best = c; }
        }
        set.push_back(best);
    }
    std::printf("%d chunks of %zu MB; lowest pair time %.4f ms; chosen set:", n, chunk >> 20, lo);
    for (int x : set) std::printf(" %d", x);
    std::printf("\n  worst pair inside the set (x lowest):");
    for (int k = 2; k <= 8; k *= 2) {
        double w = 0;
        for (int a = 0; a < k; ++a) for (int b = a + 1; b < k; ++b) w = std::max(w, L[(size_t)set[a] * n + set[b]]);
        std::printf("  K=%d %.2f", k, w / lo);
    }
    std::printf("\n");
    const size_t total = chunk;                       // every experiment moves `total` bytes
    for (int write = 1; write >= 0; --write)
        for (int K = 1; K <= 8; K *= 2) {
            Streams in_one = {}, spread = {};
            for (int k = 0; k < K; ++k) {
                in_one.p[k] = (u32x4 *)((char *)at(set[0]) + (total / K) * k);      // K parts of ONE chunk
                spread.p[k] = at(set[k]);                                           // the start of K independent chunks
            }
            const double a = write ? run_ms<true>(in_one, K, total / K) : run_ms<false>(in_one, K, total / K);
            const double b = write ? run_ms<true>(spread, K, total / K) : run_ms<false>(spread, K, total / K);
            std::printf("%s K=%d streams of %4zu MB:  inside one chunk %.4f ms = %.2f TB/s    in K independent chunks %.4f ms = %.2f TB/s\n",
                        write ? "write" : "read ", K, (total / K) >> 20, a, total / a * 1e-9, b, total / b * 1e-9);
        }
    return 0;
}
