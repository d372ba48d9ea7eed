// pfmscan_sort.hip -- hits of the ctx-owned sharded buffers -> position order, on the device.
//
// The hits kernels append in arrival order to 32 shards (256 for library scans, which sort by (position, motif)).  The host API returns them sorted by position
// (rnascan's tables are in window order, rnascan.py:263-275).  Gathering the shards with one small copy each
// and sorting an index vector on the host cost 22 ms per motif for 760 k hits -- 100x the scan itself.  Here
// the shards are packed into one key/value run (key = stream position, value = source slot), sorted with
// rocPRIM's device radix sort over only the bits a position can have, and the scores are gathered in that
// order, so that the host receives three contiguous arrays.  (The sort is a library primitive on purpose;
// the scan kernels are in pfmscan_kernels.hip.)
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include "pfmscan_internal.hpp"

namespace pfmscan {

constexpr int SORT_BLOCK = 256;
constexpr int SORT_MAX_SHARDS = 256;

// element i of the packed run <- shard s, entry j (shards in order, min(count, cap) entries each)
__global__ __launch_bounds__(SORT_BLOCK) void k_pack_shards(const int64_t *__restrict__ hit_pos,
                                                            const unsigned long long *__restrict__ counts, int shards,
                                                            int64_t shard_cap, int64_t total,
                                                            int64_t *__restrict__ keys, int64_t *__restrict__ vals,
                                                            const int32_t *__restrict__ hit_motif, int motif_bits)
{
    __shared__ int64_t start[SORT_MAX_SHARDS + 1];
    for (int t = threadIdx.x; t < shards; t += SORT_BLOCK) {
        int64_t n = (int64_t)counts[t * HIT_COUNTER_STRIDE];
        start[t + 1] = n < shard_cap ? n : shard_cap;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        start[0] = 0;
        for (int s = 0; s < shards; ++s) start[s + 1] += start[s];
    }
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * SORT_BLOCK + threadIdx.x;
    if (i >= total || i >= start[shards]) return;
    int lo = 0, hi = shards;                   // largest s with start[s] <= i
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (start[mid] <= i) lo = mid; else hi = mid;
    }
    const int64_t src = (int64_t)lo * shard_cap + (i - start[lo]);
    // library scans sort by (position, motif): the motif index rides in the low bits of the key
    keys[i] = hit_motif ? ((hit_pos[src] << motif_bits) | (int64_t)hit_motif[src]) : hit_pos[src];
    vals[i] = src;
}

// composite keys of a sorted library run -> positions (in place) and motif indices
__global__ __launch_bounds__(SORT_BLOCK) void k_split_keys(int64_t *__restrict__ keys, int64_t total, int motif_bits,
                                                           int32_t *__restrict__ motif_out)
{
    const int64_t i = (int64_t)blockIdx.x * SORT_BLOCK + threadIdx.x;
    if (i >= total) return;
    const int64_t k = keys[i];
    motif_out[i] = (int32_t)(k & (((int64_t)1 << motif_bits) - 1));
    keys[i] = k >> motif_bits;
}

__global__ __launch_bounds__(SORT_BLOCK) void k_gather_scores(const int64_t *__restrict__ order, int64_t total,
                                                              const float *__restrict__ hit_seq,
                                                              const double *__restrict__ hit_struct,
                                                              float *__restrict__ seq_out, double *__restrict__ struct_out)
{
    const int64_t i = (int64_t)blockIdx.x * SORT_BLOCK + threadIdx.x;
    if (i >= total) return;
    const int64_t src = order[i];
    if (hit_seq) seq_out[i] = hit_seq[src];
    if (hit_struct) struct_out[i] = hit_struct[src];
}

hipError_t sort_temp_bytes(int64_t total, int key_bits, size_t *bytes)
{
    *bytes = 0;
    int64_t *k = nullptr;
    return rocprim::radix_sort_pairs(nullptr, *bytes, k, k, k, k, (size_t)total, 0u, (unsigned)key_bits, (hipStream_t)0);
}

hipError_t launch_gather_sorted(const GatherArgs &g, hipStream_t stream)
{
    if (g.total <= 0) return hipSuccess;
    if (g.shards > SORT_MAX_SHARDS) return hipErrorInvalidValue;
    const unsigned grid = (unsigned)((g.total + SORT_BLOCK - 1) / SORT_BLOCK);
    hipLaunchKernelGGL(k_pack_shards, dim3(grid), dim3(SORT_BLOCK), 0, stream, g.hit_pos, g.counts, g.shards, g.shard_cap,
                       g.total, g.keys_in, g.vals_in, g.hit_motif, g.motif_bits);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    size_t bytes = g.temp_bytes;
    e = rocprim::radix_sort_pairs(g.temp, bytes, g.keys_in, g.keys_out, g.vals_in, g.vals_out, (size_t)g.total, 0u,
                                  (unsigned)(g.key_bits + (g.hit_motif ? g.motif_bits : 0)), stream);
    if (e != hipSuccess) return e;
    if (g.hit_motif) {
        hipLaunchKernelGGL(k_split_keys, dim3(grid), dim3(SORT_BLOCK), 0, stream, g.keys_out, g.total, g.motif_bits, g.motif_out);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    if (g.hit_seq || g.hit_struct) {
        hipLaunchKernelGGL(k_gather_scores, dim3(grid), dim3(SORT_BLOCK), 0, stream, g.vals_out, g.total, g.hit_seq,
                           g.hit_struct, g.seq_out, g.struct_out);
        e = hipGetLastError();
    }
    return e;
}

}  // namespace pfmscan
