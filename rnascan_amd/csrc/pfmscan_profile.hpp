// pfmscan_profile.hpp -- the pieces of k_profile (codes + averaged-structure profile, config 3) that its translation units
// share: the LDS tile layout, the LDS-DMA stager, the exact per-window fallback and the barrier-free output path.
// pfmscan_kernels.hip holds the width-generic kernel, pfmscan_profile_fixed.hip the ones unrolled for a fixed width.  Not installed.
#pragma once
#include <atomic>
#include "pfmscan_device.hpp"

namespace pfmscan {

__host__ __device__ constexpr int round_up(int x, int q) { return (x + q - 1) / q * q; }

template <int V, typename PROF_T>
struct ProfileLayout {
    static constexpr int TILE = V * BLOCK;
    // both regions are whole 1-KiB LDS-DMA pieces (one wave-instruction = 64 x 16 B)
    __host__ __device__ static constexpr int prof_bytes(int m) { return round_up((TILE + m) * 7 * (int)sizeof(PROF_T), 1024); }   // +1 row: the slide-in is unconditional
    __host__ __device__ static constexpr int code_bytes(int m) { return round_up(TILE + m, 1024); }
    __host__ __device__ static int buf_bytes(int m, bool has_seq) { return prof_bytes(m) + (has_seq ? code_bytes(m) : 0); }
    __host__ __device__ static int total(int m, bool has_seq, int nbuf)
    {
        return nbuf * buf_bytes(m, has_seq) + (has_seq ? m * 64 : 0);
    }
};

// exact per-window path (rows re-read from the staged LDS tile); only reached
// when the fast path left a non-finite sum behind
template <typename PROF_T>
__device__ __forceinline__ double struct_window_slow(const PROF_T *prof_lds, int local, const double *__restrict__ pssm, int m)
{
    double score = 0.0;
#pragma unroll 1
    for (int j = 0; j < m; ++j) {
        const PROF_T *r = prof_lds + (local + j) * 7;
        double d = (double)r[0] * pssm[j * 7];
#pragma unroll 1
        for (int k = 1; k < 7; ++k) d = fma((double)r[k], pssm[j * 7 + k], d);
        score += nan_to_num(d);
    }
    return score;
}

// Thresholded scans: the windows whose fast sum lies within a.struct_band of the threshold are scored again from the
// staged tile in the reference's rounded order, so that `score > thr` is decided on the same value (pfmscan_exact.hpp)
template <int V, typename PROF_T>
__device__ __forceinline__ void settle_near(const ScanArgs &a, const PROF_T *prof_lds, int la, double (&acc_st)[V])
{
    const double thr = a.thr_struct, band = a.struct_band;
    bool any = false;
#pragma unroll
    for (int v = 0; v < V; ++v) any = any || struct_near(acc_st[v], thr, band);
    if (!any) return;
    const double *pssm = a.struct_pssm;
    static_for<0, V>([&](auto vc) __attribute__((always_inline)) {      // (a rolled loop would index acc_st[] at run time)
        constexpr int v = decltype(vc)::value;
        if (struct_near(acc_st[v], thr, band))
            acc_st[v] = struct_window_rounded(prof_lds + (la + v) * 7, a.m, [&](int j, int k) { return pssm[j * 7 + k]; });
    });
}

// LDS-DMA issued through inline asm.  hipcc cannot tell which LDS buffer a
// `global_load_lds` writes, so with the builtin it drains vmcnt(0) before the
// next ds_read and the prefetch never overlaps the scoring loop.  Inline asm is
// invisible to its wait-count pass; the waits are placed by hand instead
// (dma_wait_all() before the barrier that publishes a buffer).  Untracked
// entries only make the compiler's own counted vmcnt waits more conservative.
// `lds_base` must be wave-uniform: the hardware writes LDS[m0 + 16 * lane].
__device__ __forceinline__ void dma_issue16(const void *gptr, uint32_t lds_base)
{
    // M0 is written and restored inside ONE statement: hipcc reserves M0 and does not see an asm as a definition of
    // it (an "m0" clobber only draws a warning), so a compiler-generated use of M0 scheduled across this statement
    // must find its own value again.
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gptr), "s"(lds_base)
                 : "memory");
}
__device__ __forceinline__ void dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ uint32_t lds_addr(const void *p)
{
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const unsigned char *)p;
}

// Stage one tile (profile rows [tile0, tile0+TILE+m-1) and their codes) into an
// LDS buffer.  Interior tiles: LDS-DMA (DMA) or 16-byte register staging;
// tiles that touch the end of the stream: dword loads with zero / SEP fill.
// Returns nothing; completion is observed by the caller's vmcnt(0) + barrier.
template <int V, bool HAS_SEQ, typename PROF_T, int DMA>   // DMA: 0 = through registers, 2 = LDS-DMA (inline asm)
__device__ __forceinline__ void stage_tile(const ScanArgs &a, int64_t tile0, unsigned char *buf, int m)
{
    using L = ProfileLayout<V, PROF_T>;
    const int tid = threadIdx.x;
    const int64_t n_pos = a.n_pos;
    const int prof_bytes = L::prof_bytes(m);
    const int code_bytes = L::code_bytes(m);
    const int64_t total_bytes = n_pos * 7 * (int64_t)sizeof(PROF_T);
    const int64_t g0 = tile0 * 7 * (int64_t)sizeof(PROF_T);         // multiple of 16: TILE*28 = 7168*V
    const unsigned char *gsrc = reinterpret_cast<const unsigned char *>(a.profile) + g0;
    const bool interior = (g0 + prof_bytes <= total_bytes) && (!HAS_SEQ || tile0 + code_bytes <= n_pos);
    if (interior) {
        if (DMA == 2) {
            // each wave-instruction moves 64 x 16 B into a lane-linear 1-KiB LDS piece
            const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
            const int npiece = prof_bytes >> 10;
            const uint32_t base = lds_addr(buf);
            // only the lanes whose 16 bytes hold rows / letters the tile reads take part in the LAST piece of each region: the
            // pieces are whole KiB of LDS, the tile needs (TILE + m) rows = 36 176 of 36 864 B and TILE + m of 2 048 code bytes at
            // w = 12 -- 1.4 KB per tile that the neighbouring tile (on another XCD) requests again (PFMSCAN_DMA_TAIL=0: whole pieces)
            const int pneed = a.dma_whole ? prof_bytes : (L::TILE + m) * 7 * (int)sizeof(PROF_T);
            const int cneed = a.dma_whole ? code_bytes : L::TILE + m;
            for (int pc = wave; pc < npiece; pc += BLOCK / 64)      // pc is wave-uniform: only a region's last piece pays the lane test
                if (pc + 1 < npiece || (pc << 10) + (lane << 4) < pneed)
                    dma_issue16(gsrc + ((size_t)pc << 10) + (lane << 4), base + ((uint32_t)pc << 10));
            if (HAS_SEQ) {
                const unsigned char *csrc = a.codes + tile0;
                const int ncp = code_bytes >> 10;
                for (int pc = wave; pc < ncp; pc += BLOCK / 64)
                    if (pc + 1 < ncp || (pc << 10) + (lane << 4) < cneed)
                        dma_issue16(csrc + ((size_t)pc << 10) + (lane << 4), base + (uint32_t)prof_bytes + ((uint32_t)pc << 10));
            }
        } else {
            // register staging, 4 x 16 B per thread in flight per round (measured faster
            // than issuing all ~9 loads first: 2.68 vs 2.99 ms on C3)
            const int nch = prof_bytes >> 4;
            // nontemporal: the stream is read once, keep it out of L2/MALL's way (-8 % on the
            // no-compute floor, -2 % end to end on C3)
#pragma unroll 4
            for (int c = tid; c < nch; c += BLOCK)
                reinterpret_cast<u32x4 *>(buf)[c] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(gsrc) + c);
            if (HAS_SEQ) {
                const int ncc = code_bytes >> 4;
                for (int c = tid; c < ncc; c += BLOCK)
                    reinterpret_cast<uint4 *>(buf + prof_bytes)[c] = reinterpret_cast<const uint4 *>(a.codes + tile0)[c];
            }
        }
    } else {
        const int ndw = prof_bytes >> 2;
        const int64_t valid_dw = (total_bytes - g0) >> 2;
        for (int c = tid; c < ndw; c += BLOCK)
            reinterpret_cast<uint32_t *>(buf)[c] = (c < valid_dw) ? reinterpret_cast<const uint32_t *>(gsrc)[c] : 0u;
        if (HAS_SEQ) {
            const int ncw = code_bytes >> 2;
            for (int c = tid; c < ncw; c += BLOCK)
                reinterpret_cast<uint32_t *>(buf + prof_bytes)[c] = load_codes4(a.codes, tile0 + 4 * (int64_t)c, n_pos);
        }
    }
}

// Output path without workgroup barriers: wave w stages its 64*V scores in the part of
// the tile buffer only IT reads -- rows [w*64V + m-1, (w+1)*64V): the first m-1 rows of its
// band are also the previous wave's halo and are left alone -- and copies them out itself
// as 16-byte stores.  LDS operations of one wave execute in order, so a wave-level fence
// is all the synchronisation needed.
template <int V, bool HAS_SEQ, typename PROF_T>
__device__ __forceinline__ void emit_tile_wave(const ScanArgs &a, int64_t tile0, int la, double (&acc_st)[V],
                                               double (&acc_sq)[V], unsigned char *tile_buf, const int m)
{
    constexpr int WN = 64 * V;                      // windows per wave
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // scalar: the band's addresses are one base + lane
    const int64_t n_pos = a.n_pos;
    const bool inside = tile0 + V * BLOCK + m <= n_pos;    // workgroup-uniform: false only for the last tile(s) of the stream
    if (!inside) {
        const double qnan = __longlong_as_double(0x7ff8000000000000ll);
#pragma unroll
        for (int v = 0; v < V; ++v)
            if (tile0 + la + v + m > n_pos) acc_st[v] = qnan;   // window runs past the stream end
    }
    const int start = ((wave * WN + m - 1) * 7 * (int)sizeof(PROF_T) + 15) & ~15;
    double *sto = reinterpret_cast<double *>(tile_buf + start);
    float *so = reinterpret_cast<float *>(tile_buf + start + WN * 8);
    const int lw = lane * V;                        // first window of this lane inside the wave's band
#pragma unroll
    for (int v = 0; v < V; ++v) {
        sto[lw + v] = acc_st[v];
        if (HAS_SEQ) so[lw + v] = (float)acc_sq[v];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int64_t w0 = tile0 + (int64_t)wave * WN;  // first window of the wave's band
    if (inside) {
        // every vector of the band lies inside the stream: one base per output, no per-lane 64-bit bounds tests
        if (HAS_SEQ && a.out_seq) {
            f32x4 *dst = reinterpret_cast<f32x4 *>(a.out_seq + w0);
#pragma unroll
            for (int c0 = 0; c0 < WN / 4; c0 += 64) {
                const int c = c0 + lane;
                if (c0 + 64 <= WN / 4 || c < WN / 4) __builtin_nontemporal_store(reinterpret_cast<const f32x4 *>(so)[c], dst + c);
            }
        }
        if (a.out_struct) {
            f64x2 *dst = reinterpret_cast<f64x2 *>(a.out_struct + w0);
#pragma unroll
            for (int c0 = 0; c0 < WN / 2; c0 += 64) {
                const int c = c0 + lane;
                if (c0 + 64 <= WN / 2 || c < WN / 2) __builtin_nontemporal_store(reinterpret_cast<const f64x2 *>(sto)[c], dst + c);
            }
        }
        return;
    }
    if (HAS_SEQ && a.out_seq) {
#pragma unroll
        for (int c0 = 0; c0 < WN / 4; c0 += 64) {
            const int c = c0 + lane;
            if (c < WN / 4) {
                const int64_t p = w0 + 4 * (int64_t)c;
                if (p + 4 <= n_pos) {
                    __builtin_nontemporal_store(reinterpret_cast<const f32x4 *>(so)[c], reinterpret_cast<f32x4 *>(a.out_seq + p));
                } else {
                    for (int e = 0; e < 4; ++e)
                        if (p + e < n_pos) a.out_seq[p + e] = so[4 * c + e];
                }
            }
        }
    }
    if (a.out_struct) {
#pragma unroll
        for (int c0 = 0; c0 < WN / 2; c0 += 64) {
            const int c = c0 + lane;
            if (c < WN / 2) {
                const int64_t p = w0 + 2 * (int64_t)c;
                if (p + 2 <= n_pos) {
                    __builtin_nontemporal_store(reinterpret_cast<const f64x2 *>(sto)[c], reinterpret_cast<f64x2 *>(a.out_struct + p));
                } else if (p < n_pos) {
                    a.out_struct[p] = sto[2 * c];
                }
            }
        }
    }
}

// outputs of one tile: hits, or LDS transpose + 16-byte coalesced stores.
// `stage` may alias the tile buffer (callers barrier before and after).
template <int V, bool HAS_SEQ, bool HITS>
__device__ __forceinline__ void emit_tile(const ScanArgs &a, int64_t tile0, int la, double (&acc_st)[V], double (&acc_sq)[V],
                                          unsigned char *stage)
{
    constexpr int TILE = V * BLOCK;
    const int tid = threadIdx.x;
    const int64_t n_pos = a.n_pos;
    if (tile0 + TILE + a.m > n_pos) {               // workgroup-uniform: only the last tile(s) of the stream
        const double qnan = __longlong_as_double(0x7ff8000000000000ll);
#pragma unroll
        for (int v = 0; v < V; ++v)
            if (tile0 + la + v + a.m > n_pos) acc_st[v] = qnan;   // window runs past the stream end
    }
    if (HITS) {
        uint32_t mask = 0;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            bool pass = (tile0 + la + v < n_pos) && (acc_st[v] > a.thr_struct);
            if (HAS_SEQ) pass = pass && ((double)(float)acc_sq[v] > a.thr_seq);
            if (pass) mask |= 1u << v;
        }
        emit_hits_block<V>(
            mask, [&](int i) { return tile0 + la + i; }, [&](int i) { return (float)acc_sq[i]; },
            [&](int i) { return acc_st[i]; }, a);
        return;
    }
    __syncthreads();                               // every wave is done with the tile
    float *so = reinterpret_cast<float *>(stage);
    double *sto = reinterpret_cast<double *>(stage + TILE * 4);
#pragma unroll
    for (int v = 0; v < V; ++v) {
        if (HAS_SEQ) so[la + v] = (float)acc_sq[v];
        sto[la + v] = acc_st[v];
    }
    __syncthreads();
    if (HAS_SEQ && a.out_seq) {
        for (int c = tid; c < TILE / 4; c += BLOCK) {
            const int64_t p = tile0 + 4 * (int64_t)c;
            if (p + 4 <= n_pos) {
                __builtin_nontemporal_store(reinterpret_cast<const f32x4 *>(so)[c], reinterpret_cast<f32x4 *>(a.out_seq + p));
            } else {
                for (int e = 0; e < 4; ++e)
                    if (p + e < n_pos) a.out_seq[p + e] = so[4 * c + e];
            }
        }
    }
    if (a.out_struct) {
        for (int c = tid; c < TILE / 2; c += BLOCK) {
            const int64_t p = tile0 + 2 * (int64_t)c;
            if (p + 2 <= n_pos) {
                __builtin_nontemporal_store(reinterpret_cast<const f64x2 *>(sto)[c], reinterpret_cast<f64x2 *>(a.out_struct + p));
            } else if (p < n_pos) {
                a.out_struct[p] = sto[2 * c];
            }
        }
    }
}

// allow the kernel all of the CU's 160 KB of LDS (static part included); per device, see allow_dynamic_lds
inline hipError_t allow_full_lds(const void *kern, std::atomic<uint64_t> &done)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (done.load(std::memory_order_acquire) & (1ull << (dev & 63))) return hipSuccess;     // the usual case: one load
    hipFuncAttributes fa;
    e = hipFuncGetAttributes(&fa, kern);
    if (e != hipSuccess) return e;
    return allow_dynamic_lds(kern, done, 160 * 1024 - (int)fa.sharedSizeBytes);
}

}  // namespace pfmscan
