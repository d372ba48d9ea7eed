// pfmscan_proflib.hip -- every motif of a STRUCTURE-ONLY PFM library in ONE pass over the averaged-structure profile
// (SURVEY 8f N1 for `-q library avgdir/`).  gfx950, wave64, 256-thread workgroups, fp64 VALU.
//
// Reference being replaced: scan_averaged_structure (rnascan/rnascan.py:293-315), which scores ONE motif per call
//     score(i) = sum_j nan_to_num(dot(profile[i+j, :], pssm[j, :]))          (rnascan.py:302-307, fp64)
//     hit <=> score > minscore                                                (rnascan.py:310, strict)
// and the multi-PFM file format the reference ships but never scans (rnascan/pfmutil.py:89-133).  Before this kernel a
// structure library ran one k_profile pass per motif, re-reading the 28-byte-per-position profile every time: 256
// motifs x 8.4 GB.  Here the profile is read ONCE; the work per window and motif is the 7 m fp64 FMAs themselves, so
// the pass is bound by the fp64 vector rate (v_fma_f64: 16 lanes per clock and SIMD), not by HBM.
//
//  * workgroup = 1024 threads = one tile of T = 256 V stream positions (V = 5) x FOUR motif groups: threads 256 g ..
//    256 g + 255 score the tile's windows for the motifs k = g (mod 4).  The tile's rows (+ m halo rows) are loaded with
//    16-byte coalesced loads, converted to fp64 ONCE and kept in LDS as [row][7] doubles (56 B per row, 72-75 KB per
//    tile): inside the motif loop a row costs 7 ds_read_b64 and no conversion (k_profile converts each row once per
//    thread and pass -- 24 v_cvt per window, which here would be paid per motif).  One tile serves 16 waves = 4 per
//    SIMD (with 256-thread workgroups the 72 KB tile allowed 2 per SIMD).  The per-lane LDS stride is 70 dwords: the
//    8-byte reads of the 32 lanes of a group fall on distinct bank pairs.
//  * thread = V consecutive windows, exactly k_profile's register sliding window: V rows live in registers, step j
//    multiplies them by PSSM row j and slides ONE new row in.  The PSSM row of (motif, j) is wave-uniform: it comes
//    through the constant address space (s_load) and is the SGPR operand of v_fma_f64, so the tables need no LDS and a
//    library of any size is one pass.  The row of step j + 1 (or row 0 of the group's next motif) is requested BEFORE the
//    35 FMAs of step j: scalar loads return out of order, so every wait for one is lgkmcnt(0) -- issued at its use, a
//    256-motif library (172 KB of PSSMs, far beyond the scalar cache) stalled every step for an L2 round trip.  The window sum is sequential in one lane, in k_profile's operation order: the
//    scores are bit-identical to the single-motif kernel's.
//  * per motif a wave-uniform branch picks the FINITE form (every PSSM cell finite: nan_to_num is the identity unless the
//    profile itself holds NaN/inf; a non-finite sum re-runs that window through the exact per-row path) or the generic
//    form (nan_to_num per row: v_max, v_min, v_cmp_o, 2 v_cndmask).
//  * hits (score > thr[motif]) go to a wave-private LDS queue (window, motif, score) that is flushed with ONE returning
//    atomic on one of 256 sharded counters when it cannot take the next 64 entries, and at the end of the tile.
#include <float.h>
#include <math.h>

#include "pfmscan_internal.hpp"
#include "pfmscan_exact.hpp"

namespace pfmscan {

constexpr int PL_LANES = 256;                         // threads that share the tile's windows (one motif group)
constexpr int PL_GROUPS = 4;                          // motif groups per workgroup: group g scores motifs k = g (mod 4)
constexpr int PL_BLOCK = PL_LANES * PL_GROUPS;
#ifndef PL_V_OVERRIDE
#define PL_V_OVERRIDE 5
#endif
constexpr int PL_V = PL_V_OVERRIDE;                   // windows per thread
constexpr int PL_NS = PL_V + 1;                       // row slots in registers: V in use + the one being filled for the next step
constexpr int PL_UNROLL = (PL_NS % 2) ? 2 * PL_NS : PL_NS;   // steps per loop iteration: slot and row buffer are compile-time
constexpr int PL_TILE = PL_LANES * PL_V;
#ifndef PL_WAIT_FIRST
#define PL_WAIT_FIRST 1
#endif
#ifndef PL_STAGGER
#define PL_STAGGER 0                                  // x 64 cycles
#endif
#ifndef PL_INTERLEAVE
#define PL_INTERLEAVE 0
#endif
constexpr int PL_QCAP = 128;                          // hits a wave can park (flushes before a 64-lane push could overflow)

typedef uint32_t pl_u32x4 __attribute__((ext_vector_type(4)));
typedef double pl_f64x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ double pl_nan_to_num(double d)
{
    double c = fmin(fmax(d, -DBL_MAX), DBL_MAX);
    return (d != d) ? 0.0 : c;
}

// doubles of one staged tile: T + m rows of 7 (the slide-in is unconditional: one row beyond the last window's), rounded up
// to whole 16-byte source vectors
__host__ __device__ constexpr int pl_tile_vals(int m) { return ((PL_TILE + m) * 7 + 3) / 4 * 4; }
__host__ __device__ constexpr int pl_tile_bytes(int m) { return pl_tile_vals(m) * 8; }
__host__ __device__ constexpr int pl_queue_bytes() { return (PL_BLOCK / 64) * PL_QCAP * (8 + 4); }

// exact per-window path from the fp64 LDS tile (rnascan.py:302-307 row by row); only after a non-finite fast sum
__device__ __forceinline__ double pl_window_slow(const double *tile, int local, const double *__restrict__ pssm, int m)
{
    double score = 0.0;
#pragma unroll 1
    for (int j = 0; j < m; ++j) {
        const double *r = tile + (local + j) * 7;
        double d = r[0] * pssm[j * 7];
#pragma unroll 1
        for (int k = 1; k < 7; ++k) d = fma(r[k], pssm[j * 7 + k], d);
        score += pl_nan_to_num(d);
    }
    return score;
}

// One motif over the thread's V windows.  P holds PSSM row 0 of this motif on entry and row 0 of the NEXT motif of the
// group (`pssm_next`) on exit: the row of the following step is always in flight while a step's FMAs run.
template <bool FINITE>
__device__ __forceinline__ void pl_score_motif(const double *tile, const double *__restrict__ pssm, const double *__restrict__ pssm_next,
                                               int m, int la, double (&acc)[PL_V], double (&P)[7])
{
    constexpr int V = PL_V, NS = PL_NS;
    // volatile: one ds_read_b64 per value.  Left to itself hipcc pairs neighbouring 8-byte reads into ds_read2_b64, which
    // moves its 16 bytes per lane at HALF the LDS rate (MI355X_MICROARCH.md, LDS table) -- and the row reads are ~50 % of
    // the LDS bandwidth already
    const volatile __attribute__((address_space(3))) double *vt = (const volatile __attribute__((address_space(3))) double *)tile;
    // NS = V + 1 row slots: step j uses the rows of positions la + j .. la + j + V - 1 and meanwhile fills the free slot
    // with position la + j + V for step j + 1 -- the LDS reads are a whole step ahead of their use, and no window's FMA
    // chain has to run first to free registers (with V slots hipcc issued the seven dependent FMAs of the oldest row back
    // to back before it could reload that slot).
    double rows[NS][7];
#pragma unroll
    for (int s = 0; s < V; ++s) {
#pragma unroll
        for (int k = 0; k < 7; ++k) rows[s][k] = vt[(la + s) * 7 + k];
        acc[s] = 0.0;
    }
#if defined(PL_ABLATE) && (PL_ABLATE & 2)
#pragma unroll
    for (int k = 0; k < 7; ++k) rows[V][k] = rows[0][k];
#endif
    // Two PSSM row buffers, P (even steps) and Q (odd steps): step j requests row j + 1 into the OTHER buffer before its own
    // FMAs, so the wait for that row is at the top of step j + 1, a whole step later, and no copy sits in between (a
    // copy at the end of the step put the wait there, 28 FMAs after the request).
    double Q[7];
    // one step: `uu` (compile-time after unrolling) picks the register slot and the row buffer, `j` is the PSSM row
    auto step = [&](const int uu, const int j) __attribute__((always_inline)) {
        const int u = uu % NS;                        // register slot that holds stream position la + j
        double (&C)[7] = (uu & 1) ? Q : P;            // row j
        double (&N)[7] = (uu & 1) ? P : Q;            // row j + 1, or row 0 of the group's next motif
#if PL_WAIT_FIRST
        // wait for the row requested a step ago BEFORE requesting the next one: scalar loads return out of order, so
        // the wait in front of this step's first FMA is lgkmcnt(0) -- left to the scheduler, the new requests were
        // hoisted above it in every other step and the wave sat out a full L2 round trip there
        __builtin_amdgcn_s_waitcnt(0xC07F);            // lgkmcnt(0)
        __builtin_amdgcn_sched_barrier(0);
#endif
        const double *nrow = (j + 1 < m) ? pssm + (j + 1) * 7 : pssm_next;
        const __attribute__((address_space(4))) double *prow = (const __attribute__((address_space(4))) double *)nrow;
#if defined(PL_ABLATE) && (PL_ABLATE & 1)                 // timing diagnostic builds only (tools/gpu_ab_c5s.sh): WRONG results
        (void)prow;
#pragma unroll
        for (int k = 0; k < 7; ++k) N[k] = C[k];
#else
#pragma unroll
        for (int k = 0; k < 7; ++k) N[k] = prow[k];
#endif
        // the free slot takes position la + j + V (always staged: the tile holds T + m rows)
#if defined(PL_ABLATE) && (PL_ABLATE & 2)
        // the slot keeps its stale row: no LDS read and no instruction in its place
#else
#pragma unroll
        for (int k = 0; k < 7; ++k) rows[(u + V) % NS][k] = vt[(la + j + V) * 7 + k];
#endif
        if (FINITE) {
            // same terms in the same order per window (bit-identical to k_profile's FINITE form)
#pragma unroll
            for (int k = 0; k < 7; ++k) {
#pragma unroll
                for (int v = 0; v < V; ++v) acc[v] = fma(rows[(u + v) % NS][k], C[k], acc[v]);
            }
        } else {
#pragma unroll
            for (int v = 0; v < V; ++v) {
                const int slot = (u + v) % NS;        // holds stream position la + v + j
                double d = rows[slot][0] * C[0];
#pragma unroll
                for (int k = 1; k < 7; ++k) d = fma(rows[slot][k], C[k], d);
                acc[v] += pl_nan_to_num(d);
            }
        }
    };
    // whole groups of PL_UNROLL steps run as ONE basic block (no test between the steps: a taken branch per 35 FMAs costs
    // the wave ~30 cycles of refetch); the last m mod PL_UNROLL steps are tested one by one.  PL_UNROLL is even and a
    // multiple of NS, so slot and row-buffer parity carry over from group to group.
    int j0 = 0;
#pragma unroll 1
    for (; j0 + PL_UNROLL <= m; j0 += PL_UNROLL) {
#pragma unroll
        for (int uu = 0; uu < PL_UNROLL; ++uu) step(uu, j0 + uu);
    }
#pragma unroll
    for (int uu = 0; uu < PL_UNROLL - 1; ++uu) {
        if (j0 + uu < m) step(uu, j0 + uu);           // wave-uniform
    }
    if (m & 1) {                                       // an odd number of steps leaves the next motif's row 0 in Q
#pragma unroll
        for (int k = 0; k < 7; ++k) P[k] = Q[k];
    }
    if (FINITE) {
#pragma unroll
        for (int v = 0; v < V; ++v)
            if (!(fabs(acc[v]) <= DBL_MAX)) acc[v] = pl_window_slow(tile, la + v, pssm, m);
    }
}

template <typename PROF_T>
__global__ __launch_bounds__(PL_BLOCK) void k_profile_lib(const ProfLibArgs a)
{
    constexpr int V = PL_V;
    constexpr int PER = 16 / (int)sizeof(PROF_T);      // profile values per 16-byte load
    typedef PROF_T vec_t __attribute__((ext_vector_type(PER)));
    extern __shared__ __align__(16) unsigned char smem[];
    const int m = a.m;
    double *tile = reinterpret_cast<double *>(smem);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int group = wave / (PL_LANES / 64);          // motif group of this wave
    double *q_sc = reinterpret_cast<double *>(smem + pl_tile_bytes(m)) + wave * PL_QCAP;
    uint32_t *q_wm = reinterpret_cast<uint32_t *>(smem + pl_tile_bytes(m) + (PL_BLOCK / 64) * PL_QCAP * 8) + wave * PL_QCAP;
    const int64_t n_pos = a.n_pos;
    const int64_t tile0 = (int64_t)blockIdx.x * PL_TILE;

    // ---- stage: values [tile0 * 7, (tile0 + T + m) * 7) of the flat profile -> fp64 in LDS, zeros past the stream end
    {
        const int nval = pl_tile_vals(m);
        const int64_t e0 = tile0 * 7, e_end = n_pos * 7;
        const PROF_T *__restrict__ src = reinterpret_cast<const PROF_T *>(a.profile) + e0;
        for (int i = threadIdx.x * PER; i < nval; i += PL_BLOCK * PER) {
            double d[PER];
            if (e0 + i + PER <= e_end) {
                const vec_t v = __builtin_nontemporal_load(reinterpret_cast<const vec_t *>(src + i));   // read once per pass
#pragma unroll
                for (int e = 0; e < PER; ++e) d[e] = (double)v[e];
            } else {
#pragma unroll
                for (int e = 0; e < PER; ++e) d[e] = (e0 + i + e < e_end) ? (double)src[i + e] : 0.0;
            }
#pragma unroll
            for (int e = 0; e < PER; e += 2) *reinterpret_cast<pl_f64x2 *>(tile + i + e) = pl_f64x2{d[e], d[e + 1]};
        }
    }
    if (threadIdx.x < PL_LANES / 64)                   // motif tickets, one per window range: the first PL_GROUPS motifs are fixed
        reinterpret_cast<uint32_t *>(smem + pl_tile_bytes(m) + pl_queue_bytes())[threadIdx.x] = PL_GROUPS;
    __syncthreads();                                   // the only workgroup barrier: the waves are independent from here on
#if PL_STAGGER
    // the four waves of a SIMD (one per motif group) run the same instruction stream and are issued round-robin: left in
    // phase they reach the branch and the wait at the end of every step TOGETHER and the FMA pipe idles through it.  A
    // quarter step of head start each keeps one wave's bubble under the other three's FMAs.
    if (group == 1) __builtin_amdgcn_s_sleep(PL_STAGGER);
    if (group == 2) __builtin_amdgcn_s_sleep(2 * PL_STAGGER);
    if (group == 3) __builtin_amdgcn_s_sleep(3 * PL_STAGGER);
#endif

    const int shard = blockIdx.x & (a.hit_shards - 1);
    unsigned long long *counter = a.hit_count + (size_t)shard * HIT_COUNTER_STRIDE;
    const unsigned long long shard_off = (unsigned long long)shard * (unsigned long long)a.shard_cap;
    int qn = 0;                                        // wave-uniform queue length
    auto flush = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(counter, (unsigned long long)qn);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)base), hi = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32));
        base = ((unsigned long long)hi << 32) | lo;
        for (int i = lane; i < qn; i += 64) {
            const unsigned long long slot = base + (unsigned long long)i;
            if ((int64_t)slot < a.shard_cap) {         // capacity is per shard
                const uint32_t wm = q_wm[i];
                a.hit_pos[shard_off + slot] = tile0 + (int64_t)(wm >> 16) + a.pos_offset;
                a.hit_motif[shard_off + slot] = a.motif_base + (int32_t)(wm & 0xFFFFu);
                a.hit_struct[shard_off + slot] = q_sc[i];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        qn = 0;
    };

    const int la = (threadIdx.x & (PL_LANES - 1)) * V;
    // windows this lane may report: they start inside the stream and their m rows end inside it (a window that runs over
    // the end has no score: NaN in the all-scores kernels)
    // (one wave-uniform count, compared at the end of each motif: five lane masks held across the motif loop cost ten SGPRs)
    const int64_t live_all = n_pos - m + 1 - tile0;
    const int live_n = live_all < 0 ? 0 : (live_all > PL_TILE ? PL_TILE : (int)live_all);
    const __attribute__((address_space(4))) double *thr = (const __attribute__((address_space(4))) double *)a.thr;
    const __attribute__((address_space(4))) int32_t *finite = (const __attribute__((address_space(4))) int32_t *)a.finite;

    // Which motif a wave scores next is decided at run time.  The four waves that share a window range (one per motif
    // group) sit on ONE SIMD, and its arbiter favours the oldest wave: with a fixed quarter of the library each, the
    // favoured wave was done at ~70 % of the workgroup's time and its slot stayed empty until the last one ended (a
    // 1024-thread workgroup frees its CU as a whole) -- 28 % of the wave slots idle in the counters.  The waves of a range
    // take motifs from a shared LDS ticket instead: they end within one motif of each other.  A wave knows its next motif
    // one motif ahead (the PSSM row prefetch needs it); the ticket for the one after is drawn between two motifs (one LDS
    // round trip per motif and wave, under the other waves' FMAs).
    uint32_t *ticket = reinterpret_cast<uint32_t *>(smem + pl_tile_bytes(m) + pl_queue_bytes()) + (wave & (PL_LANES / 64 - 1));
    auto draw = [&]() -> uint32_t {                    // lane 0's VGPR holds the ticket once the LDS atomic has returned
        uint32_t v = 0;
        if (lane == 0) v = atomicAdd(ticket, 1u);
        return v;
    };
    const int nm = a.n_motifs;
    int k = group;                                     // the first motif of every wave is fixed
    int kn = nm;
    double P[7];                                       // PSSM row in use (SGPRs); pl_score_motif leaves the next motif's row 0 in it
    int fin = 0;
    if (k < nm) {
        const __attribute__((address_space(4))) double *p0 = (const __attribute__((address_space(4))) double *)(a.pssm + (size_t)k * m * 7);
#pragma unroll
        for (int c = 0; c < 7; ++c) P[c] = p0[c];
        fin = finite[k];
        kn = (int)__builtin_amdgcn_readfirstlane(draw());
    }
#pragma unroll 1
    while (k < nm) {
        const double *pssm = a.pssm + (size_t)k * m * 7;
        const int kc = kn < nm ? kn : k;               // after the wave's last motif: any valid row
        const double *pssm_next = a.pssm + (size_t)kc * m * 7;
        // this motif's threshold and the next motif's form are requested HERE, a whole motif before their use
        const double t = thr[k];
        const int fin_next = finite[kc];
        double acc[V];
        if (fin)
            pl_score_motif<true>(tile, pssm, pssm_next, m, la, acc, P);
        else
            pl_score_motif<false>(tile, pssm, pssm_next, m, la, acc, P);
        fin = fin_next;
        // The compare every window pays is against t - band: everything that may be a hit -- and ONE branch per motif on the
        // OR over the thread's V windows (a branch per window cost the wave ~5 x 20 cycles per motif next to its 1700 of
        // FMAs).  Behind it (rare) a window within the band of the threshold is scored again in the reference's rounded
        // order, and that value decides and is reported (pfmscan_exact.hpp); the others pass on their fast score.
        const double t_lo = t - a.struct_band, t_hi = t + a.struct_band;
        bool maybe[V], any = false;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            maybe[v] = (la + v < live_n) && (acc[v] > t_lo);
            any = any || maybe[v];
        }
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(any) != 0, 0)) {          // wave-uniform
#pragma unroll
            for (int v = 0; v < V; ++v) {
                bool pass = maybe[v];
                if (__builtin_amdgcn_ballot_w64(pass) == 0) continue;                // wave-uniform
                if (pass && acc[v] <= t_hi) {
                    acc[v] = struct_window_rounded(tile + (la + v) * 7, m, [&](int j, int c) { return pssm[j * 7 + c]; });
                    pass = acc[v] > t;
                }
                const unsigned long long mk = __builtin_amdgcn_ballot_w64(pass);
                if (mk) {                              // wave-uniform
                    if (qn + 64 > PL_QCAP) flush();
                    if (pass) {
                        const int slot = qn + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
                        q_sc[slot] = acc[v];
                        q_wm[slot] = ((uint32_t)(la + v) << 16) | (uint32_t)k;
                    }
                    qn += __popcll(mk);
                }
            }
        }
        k = kn;
        if (kn < nm) kn = (int)__builtin_amdgcn_readfirstlane(draw());     // wave-uniform
    }
    if (qn > 0) flush();
}

int64_t profile_library_tile() { return PL_TILE; }

hipError_t launch_profile_library(const ProfLibArgs &a, hipStream_t stream)
{
    if (a.n_pos <= 0 || a.n_motifs <= 0) return hipSuccess;
    if (a.n_motifs > 65536 || a.m < 1 || a.m > PFMSCAN_MAX_M) return hipErrorInvalidValue;
    const unsigned grid = (unsigned)((a.n_pos + PL_TILE - 1) / PL_TILE);
    const size_t lds = (size_t)pl_tile_bytes(a.m) + pl_queue_bytes() + 16;       // + the motif tickets
    static std::atomic<uint64_t> done_f{0}, done_d{0};
    if (a.profile_dtype == PFMSCAN_PROFILE_F64) {
        hipError_t e = allow_dynamic_lds(reinterpret_cast<const void *>(k_profile_lib<double>), done_d, 160 * 1024);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_profile_lib<double>, dim3(grid), dim3(PL_BLOCK), lds, stream, a);
    } else {
        hipError_t e = allow_dynamic_lds(reinterpret_cast<const void *>(k_profile_lib<float>), done_f, 160 * 1024);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_profile_lib<float>, dim3(grid), dim3(PL_BLOCK), lds, stream, a);
    }
    return hipGetLastError();
}

}  // namespace pfmscan
