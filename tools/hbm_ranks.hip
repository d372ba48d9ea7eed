// hbm_ranks.hip -- which parts of device memory disturb each other?  tools/placement_arena.py found that the headline scan runs
// 12 % faster when its output arrays lie in other parts of HBM than its input and than each other (arrays that share DRAM banks
// evict each other's open rows; physical addresses are not visible from user space, so the parts are found by TIMING).  This tool
// takes physical memory in chunks through the virtual memory API (hipMemCreate), and times two sequential WRITE streams into two
// chunks against one write stream of the same total size: the ratio is ~1.0 for chunks that do not disturb each other and
// 1.2 .. 1.9 for chunks that do.  It prints the ratio matrix of the first chunks and a class map built the way
// rnascan_amd/csrc/pfmscan_place.hip builds it.
//   hipcc -O3 --offload-arch=gfx950 tools/hbm_ranks.hip -o tools/hbm_ranks && tools/hbm_ranks [chunk MB = 256] [chunks = 128] [matrix = 24]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// n16 vectors to x and n16 vectors to y (both streams sequential, 16 B per lane, nontemporal); y == nullptr: 2 * n16 vectors to x
__global__ __launch_bounds__(256) void k_two_streams(u32x4 *__restrict__ x, u32x4 *__restrict__ y, size_t n16)
{
    const u32x4 v = {1u, 2u, 3u, (uint32_t)threadIdx.x};
    if (y) {
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
            __builtin_nontemporal_store(v, x + i);
            __builtin_nontemporal_store(v, y + i);
        }
    } else {
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < 2 * n16; i += (size_t)gridDim.x * 256) __builtin_nontemporal_store(v, x + i);
    }
}

static hipEvent_t e0, e1;
static double two_ms(void *x, void *y, size_t bytes_each)
{
    const size_t n16 = bytes_each / 16;
    double best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_two_streams, dim3(256 * 16), dim3(256), 0, 0, (u32x4 *)x, (u32x4 *)y, n16);
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
    const size_t chunk = (size_t)(argc > 1 ? std::atoll(argv[1]) : 256) << 20;
    const int n = argc > 2 ? std::atoi(argv[2]) : 128;
    const int nm = std::min(n, argc > 3 ? std::atoi(argv[3]) : 24);
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
    auto at = [&](int i) { return (char *)base + chunk * i; };
    const size_t each = chunk / 2;
    std::vector<double> single(n);
    for (int i = 0; i < n; ++i) single[i] = two_ms(at(i), nullptr, each);
    std::printf("%d chunks of %zu MB; two write streams of %zu MB each; ONE stream of the same total: %.4f .. %.4f ms\n", n, chunk >> 20, each >> 20,
                *std::min_element(single.begin(), single.end()), *std::max_element(single.begin(), single.end()));
    std::printf("ratio matrix of the first %d chunks (x 100):\n", nm);
    for (int i = 0; i < nm; ++i) {
        for (int j = 0; j < nm; ++j) {
            if (j == i) { std::printf("  . "); continue; }
            std::printf("%4.0f", 100.0 * two_ms(at(i), at(j), each) / (0.5 * (single[i] + single[j])));
        }
        std::printf("\n");
    }
    // classes as pfmscan_place.hip forms them: a chunk joins the class whose references it disturbs most when that mean ratio is
    // over 1.15; a chunk that disturbs no class founds a new one
    std::vector<std::vector<int>> refs;
    std::vector<int> cls(n, -1);
    for (int i = 0; i < n; ++i) {
        int best = -1;
        double best_r = 0;
        std::vector<double> rs;
        for (size_t k = 0; k < refs.size(); ++k) {
            double s = 0;
            for (int r : refs[k]) s += two_ms(at(r), at(i), each) / (0.5 * (single[r] + single[i]));
            s /= refs[k].size();
            rs.push_back(s);
            if (s > best_r) { best_r = s; best = (int)k; }
        }
        if (best < 0 || best_r < 1.15) {
            best = (int)refs.size();
            refs.push_back({});
        }
        cls[i] = best;
        if (refs[best].size() < 4) refs[best].push_back(i);
        std::printf("chunk %3d:", i);
        for (double r : rs) std::printf(" %.2f", r);
        std::printf(" -> %d\n", best);
    }
    std::printf("class map: ");
    for (int i = 0; i < n; ++i) std::printf("%c", cls[i] < 10 ? '0' + cls[i] : 'a' + cls[i] - 10);
    std::printf("\n%zu classes\n", refs.size());
    return 0;
}
