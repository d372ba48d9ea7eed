// pfmscan_place.hip -- device arrays of one scan placed in HBM so that they do not disturb each other.
//
// An all-scores scan is a handful of sequential streams that advance in lock-step: the code bytes and profile rows read, the
// float32 and fp64 scores written.  On MI355X their time depends on WHERE the arrays lie, by 12 % for the headline scan
// (profiles/r4/placement_*.txt: the same build, the same process, arrays at 2.19 ms or at 1.96 ms, reproducibly per
// allocation): two streams whose current addresses fall into the same DRAM banks close each other's open rows.  Physical
// addresses are not visible from user space, but the effect is measurable: two WRITE streams into two pieces of memory take
// 0.68 of the time of one stream of the same total when the pieces are independent and up to 1.3 of it when they share their
// banks, with three quantised steps in between (tools/hbm_ranks.hip prints the matrix; bank, bank group and stack level of the
// 12-high HBM3E stacks is our reading).
//
// So the arrays of one scan are allocated TOGETHER (pfmscan_place_alloc): physical memory is taken in chunks through the
// virtual memory API (hipMemCreate), about three times as many as needed, the pairwise disturbance of the empty chunks is
// measured, every array gets the chunks that disturb the chunks in use AT THE SAME TIME of the pass least (arrays of one scan
// are traversed proportionally, so chunk k of an array meets the chunks of the others that cover the same fraction of the
// pass), the chunks are mapped side by side into one address range per array, and the rest goes back.  Results never depend on
// any of this; PFMSCAN_PLACE_PLAIN (flag or environment) takes the chunks in the order the driver hands them out.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "pfmscan_ctx.hpp"
#include "pfmscan_device.hpp"

namespace pfmscan {

constexpr int PLACE_RETRY = 1 << 30;      // internal flag of pfmscan_place_alloc: this is the second attempt

struct PlaceSet {
    struct Arr {
        char *va = nullptr;
        size_t va_bytes = 0;
        std::vector<hipMemGenericAllocationHandle_t> handles;
    };
    std::vector<Arr> arrays;
    size_t chunk = 0;
    std::vector<int64_t> bytes;       // the request it was made for (a later request of the same sizes gets a retired set back)
    bool plain = false;
    std::string note;
};

// Address space pfmscan_place_alloc may reserve per context over its whole life (ranges are never reused, see below): an
// eighth of the 47-bit user space.  A measured allocation of the headline's size reserves ~60 GB (48 GB probe window + the
// arrays): ~270 of them.  Same-sized requests after a pfmscan_place_free cost nothing (the retired set is handed out again).
constexpr size_t PLACE_VA_BUDGET = (size_t)16 << 40;
constexpr size_t PLACE_RETIRED_MAX = 2;   // retired sets kept WITH their memory for reuse; older ones give their memory back

// n16 vectors to x and n16 vectors to y: two sequential nontemporal write streams (the probe of pfmscan_place_alloc)
__global__ __launch_bounds__(BLOCK) void k_place_probe(u32x4 *__restrict__ x, u32x4 *__restrict__ y, size_t n16)
{
    const u32x4 v = {0u, 0u, 0u, 0u};
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n16; i += (size_t)gridDim.x * BLOCK) {
        __builtin_nontemporal_store(v, x + i);
        __builtin_nontemporal_store(v, y + i);
    }
}

namespace {

struct Need {            // chunk k of array r covers the fraction [lo, hi) of the pass
    int r, k;
    double lo, hi;
    int chosen = -1;
};

void release_set(PlaceSet *s)
{
    for (auto &a : s->arrays) {
        if (a.va) {
            for (size_t k = 0; k < a.handles.size(); ++k) (void)hipMemUnmap(a.va + k * s->chunk, s->chunk);
            // the address range is NOT given back: see the note on address ranges at pfmscan_place_alloc
        }
        for (auto h : a.handles) (void)hipMemRelease(h);
    }
    delete s;
}

}  // namespace

void place_release_all(pfmscan_ctx *ctx)
{
    for (void *p : ctx->place_sets) release_set(static_cast<PlaceSet *>(p));
    ctx->place_sets.clear();
    for (void *p : ctx->place_retired) release_set(static_cast<PlaceSet *>(p));
    ctx->place_retired.clear();
}

}  // namespace pfmscan

using namespace pfmscan;

// ADDRESS RANGES ARE NEVER REUSED.  On ROCm 7.2 / MI355X a virtual address range that was unmapped, given back
// (hipMemAddressFree) and handed out again by hipMemAddressReserve for OTHER physical chunks read back other bytes than were
// written: profiles/r4/placement/ab_ranges_reused_corrupt.txt -- "inputs intact right after the copy: False" from the THIRD
// allocation of the process on (the first that can be given a range an earlier set had mapped and freed), 30-98 % of the
// scores wrong from there on, and gone (ab_ranges_kept.txt) when the ONLY change is that ranges stay reserved.  Not a missing
// synchronisation or access grant: the free path of that build already ran hipDeviceSynchronize() before hipMemUnmap, every new
// mapping gets its hipMemSetAccess, and a 4 s pause changed nothing.  What the record cannot tell is which layer keeps the old
// translation; it is treated as a driver limitation and avoided:
//   * every range this file maps stays reserved until the context is destroyed (address space, not memory);
//   * a freed set is RETIRED, not unmapped: the next request of the same sizes gets it back as it is (no new range, no
//     new measurement) -- a loop of allocate / scan / free does not grow;  the last PLACE_RETIRED_MAX retired sets keep
//     their memory (pfmscan_place_trim gives it back), older ones are unmapped and their memory released at once;
//   * the address space reserved per context is bounded (PLACE_VA_BUDGET): beyond it pfmscan_place_alloc fails with
//     PFMSCAN_E_OOM instead of eating the process's address space.
extern "C" {

int pfmscan_place_alloc(pfmscan_ctx *ctx, int n_arrays, const int64_t *bytes, void **ptrs, int flags)
{
    if (!ctx) return fail(nullptr, PFMSCAN_E_BADARG, "pfmscan_place_alloc: null context");
    if (n_arrays < 1 || n_arrays > 8 || !bytes || !ptrs) return fail(ctx, PFMSCAN_E_BADARG, "pfmscan_place_alloc: 1..8 arrays, sizes and a pointer array are needed");
    for (int r = 0; r < n_arrays; ++r)
        if (bytes[r] <= 0) return fail(ctx, PFMSCAN_E_BADARG, "pfmscan_place_alloc: array sizes must be positive");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    bool plain = (flags & PFMSCAN_PLACE_PLAIN) || std::getenv("PFMSCAN_PLACE_PLAIN");
    // a retired set of the same request: handed out again as it is
    for (size_t i = 0; i < ctx->place_retired.size(); ++i) {
        PlaceSet *s = static_cast<PlaceSet *>(ctx->place_retired[i]);
        if (s->plain == plain && (int)s->bytes.size() == n_arrays && std::equal(s->bytes.begin(), s->bytes.end(), bytes) && !s->arrays.empty() &&
            !s->arrays[0].handles.empty()) {
            for (int r = 0; r < n_arrays; ++r) ptrs[r] = s->arrays[r].va;
            ctx->place_retired.erase(ctx->place_retired.begin() + (long)i);
            ctx->place_sets.push_back(s);
            ctx->place_note = s->note + "; the set of an earlier pfmscan_place_free handed out again (same sizes: no new address range, no new measurement)";
            return PFMSCAN_OK;
        }
    }
    size_t chunk = (size_t)2048 << 20;
    int64_t largest = 0;
    for (int r = 0; r < n_arrays; ++r) largest = std::max(largest, bytes[r]);
    if (const char *v = std::getenv("PFMSCAN_PLACE_CHUNK_MB")) {
        chunk = (size_t)std::max(64, std::min(16384, std::atoi(v))) << 20;
    } else {
        // small sets: smaller chunks, and below 512 MB no measurement (such a scan is over before its streams can disturb each other)
        while (chunk > ((size_t)64 << 20) && (size_t)largest <= chunk / 2) chunk /= 2;
        if (chunk < ((size_t)512 << 20)) plain = true;
    }
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = ctx->device;
    size_t gran = 0;
    HIP_TRY(ctx, hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    if (gran == 0 || chunk % gran) return fail(ctx, PFMSCAN_E_HIP, "pfmscan_place_alloc: the chunk size is not a multiple of the allocation granularity");

    // what is needed: chunk k of array r and the fraction of the pass it covers
    std::vector<Need> need;
    std::vector<int> n_chunks(n_arrays);
    for (int r = 0; r < n_arrays; ++r) {
        n_chunks[r] = (int)(((size_t)bytes[r] + chunk - 1) / chunk);
        for (int k = 0; k < n_chunks[r]; ++k)
            need.push_back({r, k, (double)k * chunk / (double)bytes[r], std::min(1.0, (double)(k + 1) * chunk / (double)bytes[r])});
    }
    const int total = (int)need.size();
    size_t free_b = 0, total_b = 0;
    HIP_TRY(ctx, hipMemGetInfo(&free_b, &total_b));
    if ((size_t)total * chunk > free_b) return fail(ctx, PFMSCAN_E_OOM, "pfmscan_place_alloc: not enough free device memory for the arrays");
    int cand = plain ? total : std::max(3 * total, total + 16);
    if (const char *v = std::getenv("PFMSCAN_PLACE_CANDIDATES")) cand = std::max(total, std::atoi(v));
    cand = std::min(cand, 64);
    cand = std::max(total, std::min(cand, (int)((double)free_b * 0.85 / (double)chunk)));
    size_t budget = PLACE_VA_BUDGET;
    if (const char *v = std::getenv("PFMSCAN_PLACE_VA_BUDGET_GB")) budget = (size_t)std::max(0, std::atoi(v)) << 30;      // tests
    if (ctx->place_va_reserved + (size_t)(cand + total) * chunk > budget)
        return fail(ctx, PFMSCAN_E_OOM, "pfmscan_place_alloc: the address-space budget of this context is used up (" +
                                           std::to_string(ctx->place_va_reserved >> 30) + " GB reserved by earlier sets; ranges are never reused: include/pfmscan.h)");

    // candidates: chunks of physical memory, mapped side by side into a window for the measurement
    std::vector<hipMemGenericAllocationHandle_t> h;
    char *window = nullptr;
    auto drop_window = [&]() {
        if (window) {
            for (size_t i = 0; i < h.size(); ++i) (void)hipMemUnmap(window + i * chunk, chunk);
            // (the window's address range stays reserved, like every range this file ever mapped: see the note below)
            window = nullptr;
        }
    };
    auto drop_all = [&]() {
        drop_window();
        for (auto x : h) (void)hipMemRelease(x);
        h.clear();
    };
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    // The candidates come in three groups with memory held (and given back right away) between them.  On a box whose
    // memory is still in one piece the driver hands out consecutive physical memory, and tens of GB in a row share the upper
    // part of their bank address: 24 consecutive 2-GB chunks of a fresh box had NO pair at the independent level (lowest pair
    // time 0.85 ms instead of 0.66) and the headline ran 2.14 ms on the "best" of them.  Spacers of up to 32 GB put the
    // groups ~48 GB apart, where the 200-GB arena map (profiles/r4/placement/) changes class.
    std::vector<hipMemGenericAllocationHandle_t> spacers;
    const bool big = chunk >= ((size_t)1 << 30);      // spacers and the retry below are for GB-sized chunks: small ones show little contrast
    const int groups = (!plain && big && cand >= 3 * 4) ? 3 : 1;
    size_t spacer = 0;
    if (groups > 1) {
        const size_t spare = free_b > (size_t)cand * chunk ? free_b - (size_t)cand * chunk : 0;
        spacer = std::min<size_t>((size_t)32 << 30, spare / 4) / chunk * chunk;      // two spacers, half of what is spare at most
        if (std::getenv("PFMSCAN_PLACE_NO_SPACERS")) spacer = 0;
    }
    for (int i = 0; i < cand; ++i) {
        if (spacer && i > 0 && i % ((cand + groups - 1) / groups) == 0) {
            for (size_t got = 0; got < spacer;) {              // in pieces of 8 GB at most: one huge request fails on a fragmented box
                const size_t piece = std::min<size_t>(spacer - got, (size_t)8 << 30);
                hipMemGenericAllocationHandle_t sp;
                if (hipMemCreate(&sp, piece, &prop, 0) != hipSuccess) {
                    (void)hipGetLastError();
                    break;
                }
                spacers.push_back(sp);
                got += piece;
            }
        }
        hipMemGenericAllocationHandle_t x;
        if (hipMemCreate(&x, chunk, &prop, 0) != hipSuccess) {
            (void)hipGetLastError();
            break;                                   // fewer candidates: the memory is shared with others
        }
        h.push_back(x);
    }
    for (auto sp : spacers) (void)hipMemRelease(sp);
    if ((int)h.size() < total) {
        drop_all();
        return fail(ctx, PFMSCAN_E_OOM, "pfmscan_place_alloc: the driver did not give enough chunks of device memory");
    }
    const int n = (int)h.size();
    std::vector<int> order(n);                       // physical chunk of every need, in the order of `need`
    std::string note;
    double cost_chosen = 0.0, cost_plain = 0.0, level_lo = 0.0, level_hi = 0.0;
    bool tuned = false;
    if (!plain && n > total) {
        hipError_t e = hipMemAddressReserve((void **)&window, (size_t)cand * chunk, 0, nullptr, 0);
        if (e == hipSuccess) ctx->place_va_reserved += (size_t)cand * chunk;
        if (e == hipSuccess) {
            for (int i = 0; i < n && e == hipSuccess; ++i) e = hipMemMap(window + (size_t)i * chunk, chunk, 0, h[i], 0);
            if (e == hipSuccess) e = hipMemSetAccess(window, (size_t)n * chunk, &acc, 1);
        }
        if (e != hipSuccess) {
            (void)hipGetLastError();
            drop_window();
            note = std::string("measurement window failed (") + hipGetErrorString(e) + "), chunks taken in driver order";
        } else {
            // pairwise disturbance of the empty chunks: two write streams of `each` bytes, the faster of two launches
            const size_t each = chunk;          // whole chunks: a chunk may straddle two bank regions, its first part says nothing about its end
            std::vector<double> L((size_t)n * n, 0.0);
            hipEvent_t e0 = nullptr, e1 = nullptr;
            if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) {
                if (e0) (void)hipEventDestroy(e0);
                drop_all();
                return fail(ctx, PFMSCAN_E_HIP, "pfmscan_place_alloc: hipEventCreate failed");
            }
            const unsigned grid = (unsigned)(ctx->n_cu > 0 ? ctx->n_cu : 256) * 16;
            hipLaunchKernelGGL(k_place_probe, dim3(grid), dim3(BLOCK), 0, ctx->stream, (u32x4 *)window, (u32x4 *)(window + chunk), each / 16);   // warm
            for (int i = 0; i < n; ++i)
                for (int j = i + 1; j < n; ++j) {
                    double best = 1e30;
                    for (int rep = 0; rep < 2; ++rep) {
                        (void)hipEventRecord(e0, ctx->stream);
                        hipLaunchKernelGGL(k_place_probe, dim3(grid), dim3(BLOCK), 0, ctx->stream, (u32x4 *)(window + (size_t)i * chunk),
                                           (u32x4 *)(window + (size_t)j * chunk), each / 16);
                        (void)hipEventRecord(e1, ctx->stream);
                        (void)hipEventSynchronize(e1);
                        float ms = 0.f;
                        (void)hipEventElapsedTime(&ms, e0, e1);
                        best = std::min(best, (double)ms);
                    }
                    L[(size_t)i * n + j] = L[(size_t)j * n + i] = best;
                }
            (void)hipEventDestroy(e0);
            (void)hipEventDestroy(e1);
            e = hipGetLastError();
            if (e != hipSuccess) {
                drop_all();
                return fail_hip(ctx, e, "pfmscan_place_alloc: probe");
            }
            level_lo = 1e30;
            for (int i = 0; i < n; ++i)
                for (int j = i + 1; j < n; ++j) {
                    level_lo = std::min(level_lo, L[(size_t)i * n + j]);
                    level_hi = std::max(level_hi, L[(size_t)i * n + j]);
                }
            // No independent pair among the candidates (the fastest pair is usually at 0.49 of the slowest; once, on a box's first
            // process, it was at 0.63 and the scan ran 2.14 ms on the best choice): give everything back, hold 48 GB, take the
            // candidates from behind them -- once.
            if (!(flags & PLACE_RETRY) && ((big && level_lo > 0.56 * level_hi) || std::getenv("PFMSCAN_PLACE_FORCE_RETRY")) && free_b > (size_t)cand * chunk + ((size_t)96 << 30)) {
                drop_all();
                std::vector<hipMemGenericAllocationHandle_t> lead;
                for (int i = 0; i < 6; ++i) {
                    hipMemGenericAllocationHandle_t sp;
                    if (hipMemCreate(&sp, (size_t)8 << 30, &prop, 0) != hipSuccess) {
                        (void)hipGetLastError();
                        break;
                    }
                    lead.push_back(sp);
                }
                const int rc = pfmscan_place_alloc(ctx, n_arrays, bytes, ptrs, flags | PLACE_RETRY);
                for (auto sp : lead) (void)hipMemRelease(sp);
                if (rc == PFMSCAN_OK) ctx->place_note += "; second attempt (the first candidates had no independent pair)";
                return rc;
            }
            // weight of a pair of needs = overlap of their fractions of the pass x the smaller array (a bank changes hands as often
            // as the slower stream comes by)
            auto weight = [&](const Need &a, const Need &b) {
                if (a.r == b.r) return 0.0;
                const double ov = std::min(a.hi, b.hi) - std::max(a.lo, b.lo);
                return ov > 0 ? ov * (double)std::min(bytes[a.r], bytes[b.r]) : 0.0;
            };
            auto total_cost = [&](const std::vector<int> &pick) {
                double c = 0;
                for (int a = 0; a < total; ++a)
                    for (int b = a + 1; b < total; ++b) {
                        const double w = weight(need[a], need[b]);
                        if (w > 0) c += w * (L[(size_t)pick[a] * n + pick[b]] - level_lo);
                    }
                return c;
            };
            std::vector<int> plain_pick(total);
            for (int a = 0; a < total; ++a) plain_pick[a] = a;
            cost_plain = total_cost(plain_pick);
            // greedy in the order of the pass, the heavier arrays first; then swaps with unused chunks while they pay
            std::vector<int> idx(total);
            for (int a = 0; a < total; ++a) idx[a] = a;
            std::sort(idx.begin(), idx.end(), [&](int a, int b) {
                if (need[a].lo != need[b].lo) return need[a].lo < need[b].lo;
                return bytes[need[a].r] > bytes[need[b].r];
            });
            std::vector<int> pick(total, -1);
            std::vector<char> used(n, 0);
            for (int a : idx) {
                int best = -1;
                double best_c = 1e300;
                for (int c = 0; c < n; ++c) {
                    if (used[c]) continue;
                    double cc = 0;
                    for (int b = 0; b < total; ++b)
                        if (pick[b] >= 0) cc += weight(need[a], need[b]) * (L[(size_t)c * n + pick[b]] - level_lo);
                    if (cc < best_c) { best_c = cc; best = c; }
                }
                pick[a] = best;
                used[best] = 1;
            }
            for (int round = 0; round < 3; ++round) {
                bool moved = false;
                for (int a = 0; a < total; ++a) {
                    auto mine = [&](int c) {
                        double cc = 0;
                        for (int b = 0; b < total; ++b)
                            if (b != a) cc += weight(need[a], need[b]) * (L[(size_t)c * n + pick[b]] - level_lo);
                        return cc;
                    };
                    double cur = mine(pick[a]);
                    for (int c = 0; c < n; ++c)
                        if (!used[c]) {
                            const double cc = mine(c);
                            if (cc < cur - 1e-12) {
                                used[pick[a]] = 0;
                                pick[a] = c;
                                used[c] = 1;
                                cur = cc;
                                moved = true;
                            }
                        }
                }
                if (!moved) break;
            }
            cost_chosen = total_cost(pick);
            if (std::getenv("PFMSCAN_PLACE_DEBUG")) {
                std::fprintf(stderr, "place: %d candidates, levels x 100 of the lowest (%.4f ms):\n", n, level_lo);
                for (int i = 0; i < n; ++i) {
                    for (int j = 0; j < n; ++j) std::fprintf(stderr, "%4.0f", i == j ? 0.0 : 100.0 * L[(size_t)i * n + j] / level_lo);
                    std::fprintf(stderr, "\n");
                }
                for (int a = 0; a < total; ++a) {
                    std::fprintf(stderr, "  array %d chunk %d [%.2f, %.2f) -> candidate %2d; meets", need[a].r, need[a].k, need[a].lo, need[a].hi, pick[a]);
                    for (int b = 0; b < total; ++b)
                        if (weight(need[a], need[b]) > 0) std::fprintf(stderr, " %d.%d:%3.0f", need[b].r, need[b].k, 100.0 * L[(size_t)pick[a] * n + pick[b]] / level_lo);
                    std::fprintf(stderr, "\n");
                }
            }
            order = pick;
            order.resize(total);
            tuned = true;
            drop_window();
        }
    }
    if (!tuned)
        for (int a = 0; a < total; ++a) order[a] = a;

    // one address range per array, its chunks side by side
    PlaceSet *set = new (std::nothrow) PlaceSet();
    if (!set) {
        drop_all();
        return fail(ctx, PFMSCAN_E_OOM, "out of host memory");
    }
    set->chunk = chunk;
    set->bytes.assign(bytes, bytes + n_arrays);
    set->plain = (flags & PFMSCAN_PLACE_PLAIN) || std::getenv("PFMSCAN_PLACE_PLAIN");
    set->arrays.resize(n_arrays);
    std::vector<char> taken(n, 0);
    hipError_t e = hipSuccess;
    for (int r = 0; r < n_arrays && e == hipSuccess; ++r) {
        auto &arr = set->arrays[r];
        arr.va_bytes = (size_t)n_chunks[r] * chunk;
        e = hipMemAddressReserve((void **)&arr.va, arr.va_bytes, 0, nullptr, 0);
        if (e != hipSuccess) {
            arr.va = nullptr;
            break;
        }
        ctx->place_va_reserved += arr.va_bytes;
        for (int a = 0; a < total && e == hipSuccess; ++a)
            if (need[a].r == r) {
                e = hipMemMap(arr.va + (size_t)need[a].k * chunk, chunk, 0, h[order[a]], 0);
                if (e == hipSuccess) {
                    arr.handles.push_back(h[order[a]]);
                    taken[order[a]] = 1;
                }
            }
        if (e == hipSuccess) e = hipMemSetAccess(arr.va, arr.va_bytes, &acc, 1);
    }
    for (int i = 0; i < n; ++i)
        if (!taken[i]) (void)hipMemRelease(h[i]);
    h.clear();
    if (e != hipSuccess) {
        (void)hipGetLastError();
        release_set(set);
        return fail_hip(ctx, e, "pfmscan_place_alloc: mapping the arrays");
    }
    for (int r = 0; r < n_arrays; ++r) ptrs[r] = set->arrays[r].va;
    ctx->place_sets.push_back(set);
    char line[512];
    if (tuned)
        std::snprintf(line, sizeof(line),
                      "%d arrays in %d chunks of %zu MB; %d candidates in %d groups %zu GB apart, pair times %.4f .. %.4f ms; weighted disturbance of the "
                      "chosen chunks %.3g, of the first chunks in driver order %.3g",
                      n_arrays, total, chunk >> 20, n, groups, spacer >> 30, level_lo, level_hi, cost_chosen, cost_plain);
    else
        std::snprintf(line, sizeof(line), "%d arrays in %d chunks of %zu MB; NOT tuned: chunks in driver order%s", n_arrays, total, chunk >> 20,
                      plain ? "" : " (no spare chunks to choose from)");
    ctx->place_note = std::string(line) + (note.empty() ? "" : "; " + note);
    set->note = ctx->place_note;
    return PFMSCAN_OK;
}

int pfmscan_place_free(pfmscan_ctx *ctx, void *first_array)
{
    if (!ctx) return fail(nullptr, PFMSCAN_E_BADARG, "pfmscan_place_free: null context");
    for (size_t i = 0; i < ctx->place_sets.size(); ++i) {
        PlaceSet *s = static_cast<PlaceSet *>(ctx->place_sets[i]);
        if (!s->arrays.empty() && s->arrays[0].va == first_array) {
            HIP_TRY(ctx, hipSetDevice(ctx->device));
            HIP_TRY(ctx, hipDeviceSynchronize());
            ctx->place_sets.erase(ctx->place_sets.begin() + (long)i);
            // retired, not unmapped: the next request of the same sizes gets it back as it is.  Only the last few retired
            // sets keep their memory; an older one is unmapped and its memory released (its address ranges stay reserved).
            ctx->place_retired.push_back(s);
            while (ctx->place_retired.size() > PLACE_RETIRED_MAX) {
                release_set(static_cast<PlaceSet *>(ctx->place_retired.front()));
                ctx->place_retired.erase(ctx->place_retired.begin());
            }
            return PFMSCAN_OK;
        }
    }
    return fail(ctx, PFMSCAN_E_BADARG, "pfmscan_place_free: not the first array of a set of pfmscan_place_alloc");
}

int pfmscan_place_trim(pfmscan_ctx *ctx)
{
    if (!ctx) return fail(nullptr, PFMSCAN_E_BADARG, "pfmscan_place_trim: null context");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipDeviceSynchronize());
    for (void *p : ctx->place_retired) release_set(static_cast<PlaceSet *>(p));
    ctx->place_retired.clear();
    return PFMSCAN_OK;
}

const char *pfmscan_place_note(const pfmscan_ctx *ctx) { return ctx ? ctx->place_note.c_str() : ""; }

}  // extern "C"
