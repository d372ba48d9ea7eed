// pfmscan_device.hpp -- device-side helpers shared by the kernel translation units (pfmscan_kernels.hip,
// pfmscan_letters8.hip): workgroup size, vector types, the code-tile stager and the workgroup hit emitter.  Not installed.
#pragma once
#include <float.h>
#include <math.h>
#include <type_traits>
#include "pfmscan_internal.hpp"
#include "pfmscan_exact.hpp"

namespace pfmscan {

constexpr int BLOCK = 256;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));
// k_letters: a workgroup scores ITERS x 1024 windows (fewer for the widest PFM bucket, whose
// code registers would otherwise spill)
__host__ __device__ constexpr int let_iters(int ndw) { return ndw > 9 ? 2 : 4; }   // 8 rounds measured slower (occupancy 5)
__host__ __device__ constexpr int let_tile(int ndw) { return BLOCK * 4 * let_iters(ndw); }

// f(integral_constant<int, I>) for I = FIRST .. LAST - 1, expanded by the template machinery: the index is a constant
// expression BY CONSTRUCTION.  Loops whose counter indexes a register array (pk[q - 2k] of the credit prefilters) use this
// instead of `#pragma unroll`: when the unroller gives up on such a loop (it did for k_letters_cred8<16>: "loop not
// unrolled"), the array is promoted to a vector register tuple that is indexed at run time, the guarded update
// `if (0 <= u && u <= W) pk[u] += x` is if-converted into an UNCONDITIONAL s_set_gpr_idx_on / v_mov_b32 write with the
// index not clamped, and for u outside the array that write lands on whatever register lives at v[base + u]
// (profiles/r5/NOTES.md, tools/gpr_idx_oob.hip; rnascan_amd/build.py refuses a library that indexes VGPRs at run time).
template <int FIRST, int LAST, typename F> __device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (FIRST < LAST) {
        f(std::integral_constant<int, FIRST>{});
        static_for<FIRST + 1, LAST>(f);
    }
}

// numpy.nan_to_num defaults (rnascan.py:306): NaN -> 0, +-inf -> +-DBL_MAX.
__device__ __forceinline__ double nan_to_num(double d)
{
    double c = fmin(fmax(d, -DBL_MAX), DBL_MAX);
    return (d != d) ? 0.0 : c;
}

// 4 code bytes at stream position p (p % 4 == 0); positions >= n_pos read as SEP.
__device__ __forceinline__ uint32_t load_codes4(const uint8_t *__restrict__ codes, int64_t p, int64_t n_pos)
{
    if (p + 4 <= n_pos) return *reinterpret_cast<const uint32_t *>(codes + p);
    uint32_t w = 0x07070707u;
    if (p < n_pos) {
        for (int b = 0; b < 4; ++b)
            if (p + b < n_pos) w = (w & ~(0xFFu << (8 * b))) | ((uint32_t)codes[p + b] << (8 * b));
    }
    return w;
}

// A tile's codes (TILE window starts + CODE_HALO bytes of look-ahead) as 16-byte vectors: global -> registers
// (fetch_codes, issued early) -> LDS (park_codes).  Every thread then reads its own 8-byte-strided
// dwords from LDS.  Loading them straight from global -- each thread its 6..7 overlapping dwords,
// every cache line requested 7 times, a 64-bit bounds test per dword -- was HALF of k_letters' time.
constexpr int CODE_HALO = 80;                         // >= W - 1 + PFMSCAN_MAX_M + 1, multiple of 16
template <int TILE> struct CodeStage {
    static constexpr int NVEC = (TILE + CODE_HALO) / 16;
    static constexpr int PER = (NVEC + BLOCK - 1) / BLOCK;
    u32x4 r[PER];
    __device__ __forceinline__ void fetch(const uint8_t *__restrict__ codes, int64_t tile0, int64_t n_pos)
    {
        if (tile0 + TILE + CODE_HALO <= n_pos) {
            // the tile and its halo lie inside the stream (workgroup-uniform: tile0 comes from blockIdx): plain vector
            // loads off one base, no per-lane 64-bit bounds tests (they were ~20 VALU instructions per lane and tile of
            // the tile-walking hits kernels, whose whole prefilter is ~80 per lane and tile)
            const u32x4 *__restrict__ src = reinterpret_cast<const u32x4 *>(codes + tile0);
#pragma unroll
            for (int k = 0; k < PER; ++k) {
                const int i = threadIdx.x + k * BLOCK;
                u32x4 v = {0x07070707u, 0x07070707u, 0x07070707u, 0x07070707u};
                if ((k + 1) * BLOCK <= NVEC || i < NVEC) v = src[i];
                r[k] = v;
            }
            return;
        }
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = threadIdx.x + k * BLOCK;
            const int64_t p = tile0 + 16 * (int64_t)i;
            u32x4 v = {0x07070707u, 0x07070707u, 0x07070707u, 0x07070707u};
            if (i < NVEC) {
                if (p + 16 <= n_pos) {
                    v = *reinterpret_cast<const u32x4 *>(codes + p);
                } else if (p < n_pos) {
                    uint32_t t[4] = {0x07070707u, 0x07070707u, 0x07070707u, 0x07070707u};
                    for (int b = 0; b < 16; ++b)
                        if (p + b < n_pos) t[b >> 2] = (t[b >> 2] & ~(0xFFu << (8 * (b & 3)))) | ((uint32_t)codes[p + b] << (8 * (b & 3)));
                    v = u32x4{t[0], t[1], t[2], t[3]};
                }
            }
            r[k] = v;
        }
    }
    __device__ __forceinline__ void park(uint8_t *cbuf) const
    {
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = threadIdx.x + k * BLOCK;
            if (i < NVEC) *reinterpret_cast<u32x4 *>(cbuf + 16 * i) = r[k];
        }
    }
};

// Append the hits of one workgroup: every thread brings N windows.  Counts are
// scanned inside the wave (shuffles) and across the 4 waves (LDS), then ONE returning
// atomic per workgroup reserves the slots -- and none at all when the workgroup has no
// hit, the usual case at real thresholds.  (One atomic per wave-instruction saturated
// the counter word at percent-level hit rates: 9 ms on C2.)  Must be called by all 256
// threads of the workgroup.  Hits of a workgroup land in position order; workgroups land
// in arrival order (the host sorts).
template <int N, typename PosF, typename SeqF, typename StF>
__device__ __forceinline__ void emit_hits_block(const uint32_t passmask, PosF pos_of, SeqF seq_of, StF st_of, const ScanArgs &a)
{
    __shared__ unsigned long long hb_base;
    __shared__ int hb_wave[BLOCK / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cnt = __popc(passmask);
    int incl = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int y = __shfl_up(incl, d);
        if (lane >= d) incl += y;
    }
    if (lane == 63) hb_wave[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
#pragma unroll
        for (int w = 0; w < BLOCK / 64; ++w) {
            const int t = hb_wave[w];
            hb_wave[w] = run;
            run += t;
        }
        // sharded counters spread the returning atomics over several words (one word saturates
        // at ~88 atomics/us: 73k workgroups with hits cost 0.8 ms on a single counter)
        const int sh = blockIdx.x & (a.hit_shards - 1);
        hb_base = run ? atomicAdd(a.hit_count + sh * HIT_COUNTER_STRIDE, (unsigned long long)run) : 0ull;
    }
    __syncthreads();
    if (passmask) {
        unsigned long long slot = hb_base + (unsigned long long)(hb_wave[wave] + incl - cnt);
        const unsigned long long off = (unsigned long long)(blockIdx.x & (a.hit_shards - 1)) * (unsigned long long)a.capacity;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            if (passmask & (1u << i)) {
                if ((int64_t)slot < a.capacity) {                 // capacity is per shard
                    a.hit_pos[off + slot] = pos_of(i) + a.pos_offset;
                    if (a.hit_seq) a.hit_seq[off + slot] = seq_of(i);
                    if (a.hit_struct) a.hit_struct[off + slot] = st_of(i);
                }
                ++slot;
            }
        }
    }
    __syncthreads();                                   // hb_* may be reused by the next call
}

constexpr int WQ_CAP = 256;                           // hits a wave can park (k_letters_pre / _cred / _quad)

}  // namespace pfmscan
