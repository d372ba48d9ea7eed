// pfmscan_library.hip -- all motifs of a PFM library in ONE pass over the stream (BASELINE
// config 5: 256 seq+struct PFM pairs x 100k x 3 kb).  gfx950, wave64, one 1024-thread workgroup per CU.
//
// The reference scans one PFM per run (rnascan.py:217-218, :262) although it ships a multi-PFM format
// (pfmutil.py:89-133).  A hit of motif k at window p needs seq_k(p) > thr_seq[k] (pssm.search, rnascan.py:263,
// strict) AND struct_k(p) > thr_struct[k] (rnascan.py:310), the inner join of combine() (rnascan.py:422-423).
//
// Per motif-window the letters side costs ceil(m/2) table look-ups, so a library scan is bound by the LDS look-up
// rate, not by HBM (the codes are read once for all motifs of a pass).  Design:
//
//  phase A (every window x every motif): "can this window be a hit?" from two-letter CREDIT tables: unsigned
//      fixed point, a GROUP of motifs interleaved per 16-byte entry ([pair row][group][16 entries][4 dwords] in LDS):
//      TWELVE 10-bit credits for PFMs up to width 16 (three per dword), EIGHT 16-bit ones beyond -> one ds_read_b128
//      per (window, pair row, group), 3 or 2 motifs per 32-bit add, two rows per v_add3_u32.  Integer adds are exact
//      and the host quantises every deficit DOWN (pfmscan_library_api.hip), so the test can only err towards keeping
//      a window.  The threshold is folded into pair row 0 so that "may be a hit" is the top bit of a credit sum: an
//      OR over the four accumulator registers and one mask test a whole group.  A lane owns ONE window; its pair
//      offsets are computed once per window and serve every group, and because the group count NG is a template
//      parameter every table offset is an immediate of the ds_read.  Windows covering a foreign letter or separator
//      are dropped here (their exact score is NaN, _pwm.c:61-66).
//  queue: flagged windows go to the wave's private LDS queue: no atomics, waves never synchronise with each other.  Twelve
//      motifs per entry: ONE push per 64-window chunk -- a lane's item is its position, its letters (two bits each) and the
//      bit mask of its flagged groups (a push per (chunk, group) made phase A VALU-bound: C5 12.9 -> 11.7 ms).  Eight per
//      entry: an item per (window, group) with the accumulators' flag bits.
//  phase B (dense, once 64 items wait): one item per lane.  The window's letters are read again (L2); the exact score
//      is the sequential fp64 sum of _pwm.c:34-68 from an LDS copy of the fp64 letter tables, cast to float32 and compared with the threshold --
//      only that decides.  A window that passes gets the exact structure score of rnascan.py:302-307
//      (per-row nan_to_num, fp64): rows straight from global memory (the neighbours of a wave's windows are in
//      L1/L2), the motif's PSSM from LDS.  Hits are compacted inside the wave; one returning atomic per batch on
//      one of 256 sharded counters.
//
// LDS per motif group at width m: (ceil(m/2)+1)*256 B (credits + the zero row) and, per motif, m*32 B (fp64 letters)
// + m*64 B (fp64 structure PSSM) + 16 B thresholds; a library larger than the 160 KB allow is scanned in several
// passes (256 pairs of width 12: passes of 96 / 96 / 64 motifs = 8 / 8 / 6 groups), each re-reading only the 1-byte codes.
#include <float.h>
#include <math.h>
#include <type_traits>

#include "pfmscan_device.hpp"

namespace pfmscan {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const unsigned char *lds_cptr;

__device__ __forceinline__ double lib_nan_to_num(double d)
{
    double c = fmin(fmax(d, -DBL_MAX), DBL_MAX);
    return (d != d) ? 0.0 : c;
}

__device__ __forceinline__ lds_cptr lds_ptr_of(const void *p) { return (lds_cptr)p; }

// 4 code bytes at the 4-byte-aligned stream position p; positions >= n_pos read as separators
__device__ __forceinline__ uint32_t lib_codes4(const uint8_t *__restrict__ codes, int64_t p, int64_t n_pos)
{
    if (p + 4 <= n_pos) return *reinterpret_cast<const uint32_t *>(codes + p);
    uint32_t w = 0x07070707u;
    for (int b = 0; b < 4; ++b)
        if (p + b < n_pos) w = (w & ~(0xFFu << (8 * b))) | ((uint32_t)codes[p + b] << (8 * b));
    return w;
}

// structure score of the window at stream position p for pass-local motif mo: rows from global memory as
// element-aligned 16-byte vectors -- a BLOCK is 7 of them = 4 float rows or 2 double rows (see struct_score_at in
// pfmscan_kernels.hip) -- PSSM cells from the transposed LDS copy pssm[((j*4 + c/2)*NMP + mo)*2 + (c&1)]: rows padded to
// 8 columns, two columns per 16 bytes, so a row is four ds_read_b128 (as 8-byte reads hipcc pairs them into
// ds_read2_b64: twice the LDS cycles per value).  Neighbouring motifs are 16 bytes apart: the distinct motifs of a wave
// spread over the banks; NMP is a compile-time constant, so the cell offsets are immediates.
// Two blocks (2 x 28 VGPRs, either precision) are in flight and the values are used from the vectors they arrived in:
// the double variant used to hold two 4-row blocks = 112 VGPRs + an unpacked copy and spilled 144 bytes per lane.
// FIN: every PSSM cell of the library is finite -> the 7 FMAs of a row go straight into the window sum (nan_to_num is the
// identity on finite row-dots; k_profile's FINITE variant does the same, same terms in the same order), and a sum that came out
// non-finite (NaN / inf in the PROFILE, or overflow) is recomputed by the per-row form.
template <typename PROF_T, int NMP, bool FIN>
__device__ __forceinline__ double lib_struct_score(const void *profile, int64_t p, int m, const double *pssm_lds, int mo)
{
    constexpr int RB = 16 / (int)sizeof(PROF_T);       // rows per block = elements per vector: 4 (float) or 2 (double)
    typedef PROF_T vec_t __attribute__((ext_vector_type(RB), aligned(sizeof(PROF_T))));
    typedef double f64x2 __attribute__((ext_vector_type(2)));
    const PROF_T *__restrict__ prof = reinterpret_cast<const PROF_T *>(profile) + p * 7;
    const f64x2 *P = reinterpret_cast<const f64x2 *>(pssm_lds) + mo;
    double score = 0.0;
    // rows [base, base + RB) of the window as 7 vectors; the last block of a width that is no multiple of RB starts
    // at m - RB and skips the rows it shares with the block before
    auto block_base = [&](int j0) { return j0 < m - RB ? j0 : m - RB; };
    auto fetch = [&](int j0, vec_t (&q)[7]) {
        const PROF_T *r = prof + block_base(j0) * 7;
#pragma unroll
        for (int k = 0; k < 7; ++k) q[k] = *reinterpret_cast<const vec_t *>(r + RB * k);
    };
    auto row_dot = [&](int j, double v0, double v1, double v2, double v3, double v4, double v5, double v6) {
        const f64x2 *Pj = P + (size_t)j * 4 * NMP;
        const f64x2 p01 = Pj[0], p23 = Pj[NMP], p45 = Pj[2 * NMP], p67 = Pj[3 * NMP];
        if (FIN) {
            score = fma(v0, p01.x, score);
            score = fma(v1, p01.y, score);
            score = fma(v2, p23.x, score);
            score = fma(v3, p23.y, score);
            score = fma(v4, p45.x, score);
            score = fma(v5, p45.y, score);
            score = fma(v6, p67.x, score);
        } else {
            double d = v0 * p01.x;
            d = fma(v1, p01.y, d);
            d = fma(v2, p23.x, d);
            d = fma(v3, p23.y, d);
            d = fma(v4, p45.x, d);
            d = fma(v5, p45.y, d);
            d = fma(v6, p67.x, d);
            score += lib_nan_to_num(d);
        }
    };
    // element e (0 .. 7 RB - 1) of a block = row e / 7, column e % 7; it sits in q[e / RB][e % RB]
    auto rows = [&](int j0, const vec_t (&q)[7]) {
        const int base = block_base(j0);
#pragma unroll
        for (int u = 0; u < RB; ++u) {
            const int j = base + u;
            if (j >= j0 && j < m) {
#define LIB_EL(c) ((double)q[(u * 7 + (c)) / RB][(u * 7 + (c)) % RB])
                row_dot(j, LIB_EL(0), LIB_EL(1), LIB_EL(2), LIB_EL(3), LIB_EL(4), LIB_EL(5), LIB_EL(6));
#undef LIB_EL
            }
        }
    };
    if (m < RB) {                                      // narrower than one block: element loads
        for (int j = 0; j < m; ++j)
            row_dot(j, (double)prof[j * 7], (double)prof[j * 7 + 1], (double)prof[j * 7 + 2], (double)prof[j * 7 + 3],
                    (double)prof[j * 7 + 4], (double)prof[j * 7 + 5], (double)prof[j * 7 + 6]);
        return score;
    }
    // two blocks in flight: the loads of block j0 + RB are issued before block j0 is scored, so a batch waits for the
    // memory once, not once per block (the 16 waves of a CU are not enough to hide three round trips per batch)
    vec_t qa[7], qb[7];
    fetch(0, qa);
    for (int j0 = 0; j0 < m; j0 += 2 * RB) {
        const bool more1 = j0 + RB < m, more2 = j0 + 2 * RB < m;
        if (more1) fetch(j0 + RB, qb);
        rows(j0, qa);
        if (more1) {
            if (more2) fetch(j0 + 2 * RB, qa);
            rows(j0 + RB, qb);
        }
    }
    return score;
}

// Two-FASTA libraries (PROF_T = uint8_t: the "profile" is the SECOND code stream, the structure strings of the same records, and
// the "PSSM" tables are [m][8] letter tables in the same LDS layout -- code c of row j at ((j * 4 + c / 2) * NMP + mo) * 2 + (c & 1),
// column 7 = NaN): the fp64 letter score of matrix.py:25-43 at window p, a sequential sum with no float32 cast; a foreign
// letter or separator makes it NaN, which never passes the strict `>` (rnascan.py:263).
template <int NMP, int NP>
__device__ __forceinline__ double lib_letters2_score(const uint8_t *__restrict__ codes2, int64_t p, int64_t n_pos, int m,
                                                     const double *tab_lds, int mo)
{
    constexpr int NRAW = NP / 2 + 1;
    const int64_t al = p & ~(int64_t)3;
    uint32_t raw[NRAW];
#pragma unroll
    for (int k = 0; k < NRAW; ++k) raw[k] = (k < (m + 3) / 4 + 1) ? lib_codes4(codes2, al + 4 * k, n_pos) : 0u;
    double st = 0.0;
#pragma unroll
    for (int k = 0; k < NP / 2; ++k) {
        const uint32_t w = __builtin_amdgcn_alignbyte(raw[k + 1], raw[k], (uint32_t)(p & 3));
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int j = 4 * k + b;
            if (j < m) {
                const uint32_t c = (w >> (8 * b)) & 7u;
                st += tab_lds[((size_t)(j * 4 + (int)(c >> 1)) * NMP + mo) * 2 + (c & 1u)];
            }
        }
    }
    return st;
}

// Credits of one motif group for the lane's window: K pair rows (compile-time), every look-up in flight before the
// first add (8 rows at a time for wide PFMs), two rows per v_add3_u32.  rowp[t] points at the lane's entry of pair
// row t in group 0; `off` (bytes, = group * 256) is an immediate when it is a constant.
template <int K, int NP>
__device__ __forceinline__ u32x4 lib_credits(const lds_cptr (&rowp)[NP], const int off)
{
    // look-ups in flight before the first add: 8 rows (all of them for PFMs up to width 16); the wider buckets, whose 16 or
    // 32 row addresses already fill the register budget of a 16-wave workgroup, take 4 at a time (8 spilled 20-36 bytes)
    constexpr int CH = NP > 8 ? 4 : 8;
    u32x4 acc = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int t0 = 0; t0 < K; t0 += CH) {
        u32x4 r[CH];
#pragma unroll
        for (int i = 0; i < CH; ++i)
            if (t0 + i < K) r[i] = *reinterpret_cast<const __attribute__((address_space(3))) u32x4 *>(rowp[t0 + i] + off);
#pragma unroll
        for (int i = 0; i < CH; i += 2) {
            if (t0 + i + 1 < K) {
                acc.x = acc.x + r[i].x + r[i + 1].x;
                acc.y = acc.y + r[i].y + r[i + 1].y;
                acc.z = acc.z + r[i].z + r[i + 1].z;
                acc.w = acc.w + r[i].w + r[i + 1].w;
            } else if (t0 + i < K) {
                acc.x += r[i].x;
                acc.y += r[i].y;
                acc.z += r[i].z;
                acc.w += r[i].w;
            }
        }
        if (NP > 8 && t0 + CH < K) __builtin_amdgcn_sched_barrier(0);      // or hipcc hoists every look-up to the front again
    }
    return acc;
}

// flagged lanes of one group -> the wave's queue.  mk = ballot of the flags (non-zero), qn = queue length.  An item is
// the window's position and the flag bits of the accumulators, decoded in phase B (the push itself is the rare path:
// ~1.3 items per group and chunk):
//   8 motifs per entry (16-bit credits): p0 = the four sign BYTES of x, y (motifs 0..3), p1 = those of z, w (motifs 4..7)
//     with the group g in bits 0-5 of p1;
//   12 motifs per entry (10-bit credits, fields at bits 0 / 10 / 20 of a dword): p0 = the group g in bits 24-29 (and, for
//     an item that went back to the queue, bit 31 + the flags still to do: bit (10 f + d) for motif 3 d + f);
//     p1 = the window's letters (widths up to 16: two bits each), so that the exact pass starts from LDS alone instead
//     of waiting for five scattered code loads.
template <int MPG>
__device__ __forceinline__ void lib_push(const u32x4 acc, const bool flag, const unsigned long long mk, const int qn, const int g,
                                         const uint32_t relpos, const uint32_t cw, uint32_t *q_pos, uint32_t *q_p0, uint32_t *q_p1)
{
    if (flag) {
        const int slot = qn + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
        q_pos[slot] = relpos;
        if constexpr (MPG == 12) {
            // nine groups out of ten push something at C5's candidate rate: the push is the common path, so it stores what
            // it has -- the group and the window's letters (two bits each) -- and phase B, 64 items at a time, looks the
            // item's credits up again to see WHICH motifs were flagged
            q_p0[slot] = (uint32_t)g << 24;
            q_p1[slot] = cw;
        } else {
            const uint32_t p0 = __builtin_amdgcn_perm(acc.y, acc.x, 0x07050301u);
            const uint32_t p1 = __builtin_amdgcn_perm(acc.w, acc.z, 0x07050301u);
            q_p0[slot] = p0;
            q_p1[slot] = (p1 & ~0x3Fu) | (uint32_t)g;          // v_bfi_b32
        }
    }
}

// Phase A, fast path: all NG groups of the lane's window, fully unrolled (immediate table offsets).  Pushes while
// the queue has room; returns the first group that did NOT fit (NG when all did) -- from there the slow path takes
// over after a drain.  A chunk brings one or two items per group at realistic thresholds, the queue takes >= 65.
template <int K, int NG, int NP, int GS = 256>      // GS: bytes of one motif group inside a table row (entries x 16)
__device__ __forceinline__ int lib_octets_fast(const lds_cptr (&rowp)[NP], int &qn, const uint32_t relpos, const uint32_t cw,
                                               uint32_t *q_pos, uint32_t *q_p0, uint32_t *q_p1, const int ng_real)
{
    int g_next = NG;
    int qs = __builtin_amdgcn_readfirstlane(qn);      // the queue length is wave-uniform: keep it (and the branches on it) scalar
    auto group = [&](const int g) {
        const u32x4 acc = lib_credits<K, NP>(rowp, g * GS);
        const bool flag = ((acc.x | acc.y | acc.z | acc.w) & (lib_mpg(NP) == 12 ? 0x20080200u : 0x80008000u)) != 0u;
        const unsigned long long mk = __builtin_amdgcn_ballot_w64(flag);
        if (mk && g_next == NG) {                     // wave-uniform
            const int n = __popcll(mk);
            // the bookkeeping first, in scalar registers, the (divergent) writes after it: behind the writes hipcc
            // carries both values in VGPRs (selects and moves per octet)
            const bool fits = qs + n <= LIB_QCAP;
            const int at = qs;
            g_next = fits ? NG : g;
            qs = fits ? qs + n : qs;
            asm volatile("" : "+s"(qs), "+s"(g_next));
            if (fits) lib_push<lib_mpg(NP)>(acc, flag, mk, at, g, relpos, cw, q_pos, q_p0, q_p1);
        }
    };
    if constexpr (GS == 128) {
        // k_library8: hipcc leaves the pragma loop below ROLLED for this kernel (-Wpass-failed; 16 row addresses advanced per
        // group instead of 16 immediates) -- expanded by template, the groups past ng_real skipped by a scalar branch each
        static_for<0, NG>([&](auto gc) {
            if ((int)gc < ng_real) group((int)gc);
        });
    } else {
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g >= ng_real) break;                  // wave-uniform: the groups from here on hold no motif
            group(g);
        }
    }
    qn = qs;
    return g_next;
}

// Phase A, slow path: groups g .. NG-1 one by one (runtime table offset), stopping as soon as 64 items wait.
// Needs qn < 64 on entry (a group brings at most 64 items, the queue holds LIB_QCAP >= 127).
template <int K, int NG, int NP, int GS = 256>
__device__ __forceinline__ int lib_octets_slow(int g, const lds_cptr (&rowp)[NP], int &qn, const uint32_t relpos, const uint32_t cw,
                                               uint32_t *q_pos, uint32_t *q_p0, uint32_t *q_p1, const int ng_real)
{
    int qs = __builtin_amdgcn_readfirstlane(qn);
    g = __builtin_amdgcn_readfirstlane(g);
    while (g < ng_real && qs < 64) {
        const u32x4 acc = lib_credits<K, NP>(rowp, g * GS);
        const bool flag = ((acc.x | acc.y | acc.z | acc.w) & (lib_mpg(NP) == 12 ? 0x20080200u : 0x80008000u)) != 0u;
        const unsigned long long mk = __builtin_amdgcn_ballot_w64(flag);
        if (mk) {
            lib_push<lib_mpg(NP)>(acc, flag, mk, qs, g, relpos, cw, q_pos, q_p0, q_p1);
            qs += __popcll(mk);
        }
        ++g;
    }
    qn = qs;
    return g;
}

// Phase A of the twelve-motifs-per-entry form (widths up to 16): the flags of ALL NG groups of the lane's window as one bit
// mask.  Nine groups out of ten have a flagged lane somewhere at realistic thresholds, so a push per (chunk, group) -- ballot,
// branch, slot arithmetic, three LDS stores: ~15 VALU instructions next to the group's 19.5 -- made phase A VALU-bound
// (+3.9 ms on C5).  Here a group adds one select and one OR; the chunk is pushed ONCE, a lane's item carries its group mask.
// `ng_real` (wave-uniform, <= NG): groups from there on hold no motif (the last pass of a library in the common table layout).
template <int K, int NG, int NP>
__device__ __forceinline__ uint32_t lib_groupmask(const lds_cptr (&rowp)[NP], const int ng_real)
{
    uint32_t gm = 0u;
    int ngr = __builtin_amdgcn_readfirstlane(ng_real);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        // (opaque: or hipcc turns the NG compares into NG lane masks computed before the chunk loop, spills them into VGPR lanes
        // and pays two v_readlane per group to get them back)
        asm volatile("" : "+s"(ngr));
        if (g < ngr) {                                  // scalar compare + branch; always taken for full passes
            const u32x4 acc = lib_credits<K, NP>(rowp, g * 256);
            const bool flag = ((acc.x | acc.y | acc.z | acc.w) & 0x20080200u) != 0u;
            gm |= flag ? (1u << g) : 0u;
        }
    }
    return gm;
}

template <int K, int NG, int NP>
__device__ __forceinline__ uint32_t lib_dispatch_mask(const int npair, const lds_cptr (&rowp)[NP], const int ng_real)
{
    if constexpr (K >= NP) {
        return lib_groupmask<NP, NG, NP>(rowp, ng_real);
    } else {
        if (npair == K) return lib_groupmask<K, NG, NP>(rowp, ng_real);
        return lib_dispatch_mask<K + 1, NG, NP>(npair, rowp, ng_real);
    }
}

// npair -> the K-row instantiation, over the pair counts of one width bucket (K = KLO .. NP)
template <bool FAST, int K, int NG, int NP>
__device__ __forceinline__ int lib_dispatch(const int npair, const int g, const lds_cptr (&rowp)[NP], int &qn,
                                            const uint32_t relpos, const uint32_t cw, uint32_t *q_pos, uint32_t *q_p0, uint32_t *q_p1,
                                            const int ng_real)
{
    if constexpr (K >= NP) {
        if (FAST) return lib_octets_fast<NP, NG, NP>(rowp, qn, relpos, cw, q_pos, q_p0, q_p1, ng_real);
        return lib_octets_slow<NP, NG, NP>(g, rowp, qn, relpos, cw, q_pos, q_p0, q_p1, ng_real);
    } else {
        if (npair == K) {
            if (FAST) return lib_octets_fast<K, NG, NP>(rowp, qn, relpos, cw, q_pos, q_p0, q_p1, ng_real);
            return lib_octets_slow<K, NG, NP>(g, rowp, qn, relpos, cw, q_pos, q_p0, q_p1, ng_real);
        }
        return lib_dispatch<FAST, K + 1, NG, NP>(npair, g, rowp, qn, relpos, cw, q_pos, q_p0, q_p1, ng_real);
    }
}

// NG = motif groups of the pass, NP = pair rows the code is unrolled for (8 / 16 / 32 for m <= 16 / 32 / 64)
template <int NG, int NP, typename PROF_T, bool HAS_STRUCT>
__global__ __launch_bounds__(lib_block(NP)) void k_library(const LibArgs a)
{
    constexpr int LIB_BLOCK = lib_block(NP);
    constexpr int LIB_WAVES = LIB_BLOCK / 64;
    constexpr int MPG = lib_mpg(NP);                // motifs per table entry: 12 (10-bit credits) or 8 (16-bit)
    constexpr int NMP = NG * MPG;                   // motifs of the pass (padding motifs never flag)
    constexpr int NRAW = NP / 2 + 1;                // aligned code dwords a lane loads
    extern __shared__ __align__(16) unsigned char smem[];
    const int m = a.m, npair = a.npair;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    // ---- LDS carve-up (same arithmetic as lib_lds_bytes) ----
    const int pair_bytes = npair * NG * 256;        // + one all-zero row of NG * 256 bytes: row 0 of the windows that cannot score
    uint32_t *ticket = reinterpret_cast<uint32_t *>(smem);          // the workgroup's chunk ticket (see the chunk loop): 16 bytes at a constant address
    uint32_t *pairs = reinterpret_cast<uint32_t *>(smem + 16);
    double *letters = reinterpret_cast<double *>(smem + 16 + pair_bytes + NG * 256);
    double *pssm = letters + (size_t)m * 4 * NMP;
    double *thr_s = pssm + (HAS_STRUCT ? (size_t)m * 8 * NMP : 0);
    double *thr_t = thr_s + NMP;
    uint32_t *qbase = reinterpret_cast<uint32_t *>(thr_t + NMP);
    uint32_t *q_pos = qbase + (size_t)wave * LIB_QCAP * 3;
    uint32_t *q_p0 = q_pos + LIB_QCAP;
    uint32_t *q_p1 = q_p0 + LIB_QCAP;

    // which pass this workgroup runs (teams: several passes of the library side by side in one launch, see LibArgs)
    int team = 0, team_b0 = 0, team_grid = (int)gridDim.x, ng_real = a.ng_real;
    if (a.n_teams > 1) {
        const int bx = (int)blockIdx.x;
        team = (bx >= a.team_first[1] ? 1 : 0) + (bx >= a.team_first[2] ? 1 : 0) + (bx >= a.team_first[3] ? 1 : 0);
        team_b0 = team == 0 ? 0 : (team == 1 ? a.team_first[1] : (team == 2 ? a.team_first[2] : a.team_first[3]));
        team_grid = (team == 0 ? a.team_first[1] : (team == 1 ? a.team_first[2] : (team == 2 ? a.team_first[3] : a.team_first[4]))) - team_b0;
        ng_real = team == 0 ? a.team_ng[0] : (team == 1 ? a.team_ng[1] : (team == 2 ? a.team_ng[2] : a.team_ng[3]));
    }
    const uint32_t *g_pairs = a.pairs + (size_t)team * a.stride_pairs;
    const double *g_letters = a.letters + (size_t)team * a.stride_letters;
    const double *g_pssm = HAS_STRUCT ? a.pssm + (size_t)team * a.stride_pssm : nullptr;
    const double *g_thr_seq = a.thr_seq + (size_t)team * a.stride_thr, *g_thr_struct = HAS_STRUCT ? a.thr_struct + (size_t)team * a.stride_thr : nullptr;
    const int motif_base = a.motif_base + team * NMP;
    const int bid = (int)blockIdx.x - team_b0;      // workgroup index inside the team

    for (int i = threadIdx.x; i < pair_bytes / 16; i += LIB_BLOCK)
        reinterpret_cast<u32x4 *>(pairs)[i] = reinterpret_cast<const u32x4 *>(g_pairs)[i];
    for (int i = threadIdx.x; i < NG * 16; i += LIB_BLOCK) reinterpret_cast<u32x4 *>(pairs)[pair_bytes / 16 + i] = u32x4{0u, 0u, 0u, 0u};
    for (int i = threadIdx.x; i < m * 4 * NMP; i += LIB_BLOCK) letters[i] = g_letters[i];
    if (HAS_STRUCT)
        for (int i = threadIdx.x; i < m * 8 * NMP; i += LIB_BLOCK) pssm[i] = g_pssm[i];
    for (int i = threadIdx.x; i < NMP; i += LIB_BLOCK) {
        thr_s[i] = g_thr_seq[i];
        thr_t[i] = HAS_STRUCT ? g_thr_struct[i] : -INFINITY;
    }
    if (threadIdx.x == 0) *ticket = LIB_WAVES;      // the first chunk of wave w is chunk w
    __syncthreads();                                // the only workgroup barrier: waves are independent from here on

    const int64_t n_pos = a.n_pos;
    const int shard = blockIdx.x & (a.hit_shards - 1);
    unsigned long long *counter = a.hit_count + (size_t)shard * HIT_COUNTER_STRIDE;
    const unsigned long long shard_off = (unsigned long long)shard * (unsigned long long)a.shard_cap;
    const lds_cptr pairs_lds = lds_ptr_of(pairs);

    // ---- phase B: the top cnt (<= 64) items [first, first + cnt) of this wave's queue, one per lane; items that
    // still have flagged motifs left are written back from `first` on, `requeued` of them ----
    int requeued = 0;
    auto dense = [&](int first, int cnt) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const bool have = lane < cnt;
        const int idx = first + (have ? lane : 0);
        uint32_t rel = q_pos[idx];
        uint32_t p0 = q_p0[idx], p1 = q_p1[idx];
        if (MPG == 12 && a.sort_batches) {
            // A/B (PFMSCAN_LIB_SORT=1): the batch in the order of the motif GROUP each item works on -- a counting sort by
            // ballots.  The 12 motifs of a group are 12 consecutive 16-byte PSSM cells = 12 distinct bank groups, so lanes
            // of one group read their cells without conflicts (unsorted: 96 motifs at random, ~2.7-way).
            const uint32_t key = have ? (uint32_t)__builtin_ctz((p0 & 0xFFFFu) | 0x10000u) : 31u;
            int rank = 0, base = 0;
#pragma unroll
            for (int v = 0; v < NG; ++v) {
                const unsigned long long mv = __builtin_amdgcn_ballot_w64(key == (uint32_t)v);
                const int below = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mv >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mv, 0u));
                rank = key == (uint32_t)v ? base + below : rank;
                base += __popcll(mv);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();               // every lane has read its item
            if (have) {
                q_pos[first + rank] = rel;
                q_p0[first + rank] = p0;
                q_p1[first + rank] = p1;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            rel = q_pos[idx];
            p0 = q_p0[idx];
            p1 = q_p1[idx];
        }
        // An item of the twelve-per-entry form: p0 = the mask of flagged groups (bits 0-15), p1 = the window's letters.  The
        // batch works on the LOWEST flagged group g: its credit sums are looked up once more (npair look-ups for 64 items at
        // once) to see which of its motifs were flagged -- bit 4 f + d of `bits` = motif 3 d + f.  An item that came back
        // with motifs of g still to do carries them in p0 bits 16-27 under bit 31.
        // The eight-per-entry form: p0 / p1 = the sign bytes of the accumulators (byte b, bit 0 = motif b, bit 1 = motif
        // 4 + b) with the group in bits 0-5 of p1.
        const uint32_t gmask = p0 & 0xFFFFu;
        const uint32_t g = MPG == 12 ? (uint32_t)__builtin_ctz(gmask | 0x10000u) : (p1 & 0x3Fu);
        uint32_t bits;
        if (MPG == 12) {
            u32x4 acc = {0u, 0u, 0u, 0u};
            const lds_cptr ent = pairs_lds + g * 256;
            for (int t = 0; t < npair; ++t) {
                const u32x4 r = *reinterpret_cast<const __attribute__((address_space(3))) u32x4 *>(ent + t * (NG * 256) + (((p1 >> (4 * t)) & 0xFu) << 4));
                acc.x += r.x;
                acc.y += r.y;
                acc.z += r.z;
                acc.w += r.w;
            }
            constexpr uint32_t M = 0x20080200u;
            const uint32_t spread = ((acc.x & M) >> 9) | ((acc.y & M) >> 8) | ((acc.z & M) >> 7) | ((acc.w & M) >> 6);   // bit 10 f + d
            const uint32_t fresh = (spread & 0xFu) | ((spread >> 6) & 0xF0u) | ((spread >> 12) & 0xF00u);              // bit 4 f + d
            bits = !have ? 0u : ((p0 & 0x80000000u) ? ((p0 >> 16) & 0xFFFu) : fresh);
        } else {
            bits = !have ? 0u : (((p0 >> 7) & 0x01010101u) | ((p1 >> 6) & 0x02020202u));
        }
        const int64_t p = a.pos_base + (int64_t)rel;
        const bool act = bits != 0;
        const int q = act ? __builtin_ctz(bits) : 0;
        bits &= bits - 1;
        // ONE motif per lane and batch: an item with more flagged motifs (or, twelve-per-entry, more flagged groups) goes
        // back to the queue with the rest (a second round for the one or two such lanes of a batch would cost as much as
        // a full batch)
        uint32_t back0 = 0u;                             // p0 of the item that goes back (0 = nothing left)
        if (MPG == 12) {
            const uint32_t rest = gmask & (gmask - 1u);  // the groups after g
            back0 = bits != 0 ? (0x80000000u | (bits << 16) | gmask) : rest;
            if (!have) back0 = 0u;
        }
        const bool again = MPG == 12 ? back0 != 0u : bits != 0;
        const unsigned long long more = __builtin_amdgcn_ballot_w64(again);
        if (more) {
            if (again) {
                const int slot = first + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(more >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)more, 0u));
                q_pos[slot] = rel;                  // slots first .. first + popc - 1 are being freed by this batch (all read above)
                if (MPG == 12) {
                    q_p0[slot] = back0;
                    q_p1[slot] = p1;
                } else {
                    q_p0[slot] = (bits & 0x01010101u) << 7;
                    q_p1[slot] = ((bits & 0x02020202u) << 6) | g;
                }
            }
            requeued = __popcll(more);
        } else {
            requeued = 0;
        }
        // the window's letters: from the item (12 motifs per entry), else read again (L2: the wave read them a few chunks
        // ago); the window has no foreign letter, or it would not be here
        uint32_t w[NP / 2];
        if (MPG != 12) {
            const int64_t al = p & ~(int64_t)3;
            uint32_t raw[NRAW];
#pragma unroll
            for (int k = 0; k < NRAW; ++k) raw[k] = (k < (m + 3) / 4 + 1) ? lib_codes4(a.codes, al + 4 * k, n_pos) : 0u;
#pragma unroll
            for (int k = 0; k < NP / 2; ++k) w[k] = __builtin_amdgcn_alignbyte(raw[k + 1], raw[k], (uint32_t)(p & 3));
        }
        const int mo = (int)g * MPG + (MPG == 12 ? 3 * (q & 3) + (q >> 2) : (q >> 3) + 4 * (q & 1));   // pass-local motif
        // exact sequence score: sequential fp64 sum (_pwm.c:36-64), float32 cast (:65)
        double sc = 0.0;
        const double *L = letters + mo;
#pragma unroll
        for (int j = 0; j < NP * 2; ++j) {
            if (j < m) {
                const uint32_t c = MPG == 12 ? (p1 >> (2 * j)) & 3u : (w[j >> 2] >> ((j & 3) * 8)) & 3u;
                sc += L[(j * 4) * NMP + c * NMP];
            }
        }
        const float f = (float)sc;
        bool ok = act && ((double)f > thr_s[mo]);
        double st = 0.0;
        if (HAS_STRUCT) {
            if (__builtin_amdgcn_ballot_w64(ok)) {
                if (ok) {
#ifdef LIB_DIAG_WRAP                                     // timing diagnostic only: the rows come from the first 2^20 positions (cache-resident): WRONG scores
                    const int64_t ps = p & 0xFFFFF;
#else
                    const int64_t ps = p;
#endif
                    if constexpr (std::is_same<PROF_T, uint8_t>::value) {
                        // two-FASTA library: the structure LETTERS of the same window (rnascan.py:416-434 joins the two tables)
                        st = lib_letters2_score<NMP, NP>(reinterpret_cast<const uint8_t *>(a.profile), ps, n_pos, m, pssm, mo);
                    } else {
                        bool exact = !a.struct_finite;                       // wave-uniform
                        if (a.struct_finite) {
                            st = lib_struct_score<PROF_T, NMP, true>(a.profile, ps, m, pssm, mo);
                            exact = !(fabs(st) <= DBL_MAX);                  // NaN / inf in the profile, or overflow: the per-row form decides
                        }
                        if (exact) st = lib_struct_score<PROF_T, NMP, false>(a.profile, ps, m, pssm, mo);
                        if (struct_near(st, thr_t[mo], a.struct_band))       // too close to call: the reference's rounded order decides
                            st = struct_window_rounded(reinterpret_cast<const PROF_T *>(a.profile) + ps * 7, m,
                                                       [&](int j, int k) { return pssm[((size_t)(j * 4 + (k >> 1)) * NMP + mo) * 2 + (k & 1)]; });
                    }
                    ok = st > thr_t[mo];
                }
            }
        }
        const unsigned long long hm = __builtin_amdgcn_ballot_w64(ok);
        if (hm) {
            const int nh = __popcll(hm);
            unsigned long long base = 0;
            if (lane == 0) base = atomicAdd(counter, (unsigned long long)nh);
            const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)base), hi = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32));
            base = ((unsigned long long)hi << 32) | lo;
            if (ok) {
                const unsigned long long slot = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(hm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hm, 0u));
                if ((int64_t)slot < a.shard_cap) {
                    a.hit_pos[shard_off + slot] = p + a.pos_offset;
                    a.hit_motif[shard_off + slot] = motif_base + mo;
                    a.hit_seq[shard_off + slot] = f;
                    if (HAS_STRUCT) a.hit_struct[shard_off + slot] = st;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };

    // ---- phase A: this workgroup's segments, this wave's 64-window chunks ----
    // The workgroup's chunks -- the 64-window chunks of its segments blockIdx.x, blockIdx.x + gridDim.x, ... in order -- are
    // handed out by an LDS ticket, not dealt round-robin: the SIMD arbiter favours the oldest wave, and with a fixed share
    // each the favoured waves of this persistent workgroup were done at ~85 % of the kernel's time, their slots empty for
    // the rest (SQ_WAVE_CYCLES against GRBM_GUI_ACTIVE in profiles/r3/bench_c5_pmc_summary_before_tickets.txt).  The ticket for the chunk
    // after this one is drawn before the chunk and read after it.
    int qn = 0;                                     // wave-uniform queue length (< 64 between chunks)
    constexpr int seg_shift = LIB_SEG_SHIFT - 6;    // chunks per segment (launch_library insists on seg_positions == 2^LIB_SEG_SHIFT)
    const int64_t my_segs = bid < a.n_seg ? (a.n_seg - bid + team_grid - 1) / team_grid : 0;
    const uint32_t n_units = (uint32_t)(my_segs << seg_shift);
    constexpr bool AHEAD = NP <= 8;                 // the wider buckets have no VGPR to hold the ticket across a chunk
    uint32_t drawn = 0;
    auto next_unit = [&]() -> uint32_t {
        if (!AHEAD && lane == 0) drawn = atomicAdd(ticket, 1u);
        return __builtin_amdgcn_readfirstlane(drawn);
    };
    for (uint32_t u = (uint32_t)wave; u < n_units; u = next_unit()) {
        if (AHEAD && lane == 0) drawn = atomicAdd(ticket, 1u);
        {
            // segment bid + (u >> seg_shift) team_grid, chunk u mod 2^seg_shift of it; relative to pos_base
            const int64_t rel0 = ((int64_t)((uint32_t)bid + (u >> seg_shift) * (uint32_t)team_grid) << (seg_shift + 6)) +
                                 (int64_t)((u & ((1u << seg_shift) - 1u)) << 6);
            if (rel0 >= a.span) continue;                    // wave-uniform (the last segment may run over the span)
            const int64_t p0 = a.pos_base + rel0;            // multiple of 64
            // the lane's letters: bytes [p0 + lane, p0 + lane + 2 NP) from NRAW aligned dwords
            const int64_t al = p0 + (lane & ~3);
            uint32_t raw[NRAW];
            if (p0 + 64 + 4 * NRAW <= n_pos) {
#pragma unroll
                for (int k = 0; k < NRAW; ++k) raw[k] = *reinterpret_cast<const uint32_t *>(a.codes + al + 4 * k);
            } else {
#pragma unroll
                for (int k = 0; k < NRAW; ++k) raw[k] = lib_codes4(a.codes, al + 4 * k, n_pos);
            }
            lds_cptr rowp[NP];
            uint32_t badbits = 0, cw = 0;               // cw: the window's letters, two bits each (12 motifs per entry: m <= 16)
#pragma unroll
            for (int k = 0; k < NP / 2; ++k) {
                const uint32_t w = __builtin_amdgcn_alignbyte(raw[k + 1], raw[k], (uint32_t)(lane & 3));
                // letters 4k .. 4k+3 of the window; only the first m count
                // (a width bucket starts above the previous one: m > NP for NP > 8, so its first NP / 4 dwords always count
                // in full -- constants instead of eight scalar masks kept live across the whole chunk loop)
                const int nb = m - 4 * k;
                const uint32_t vm = (NP > 8 && 4 * k + 4 <= NP) ? 0x04040404u
                                    : (nb >= 4 ? 0x04040404u : (nb > 0 ? (0x04040404u & ((1u << (8 * nb)) - 1u)) : 0u));
                badbits |= w & vm;
                const uint32_t x = w & 0x03030303u;
                const uint32_t y = x | (x >> 6);             // pair (b0,b1) in bits 0-3, pair (b2,b3) in bits 16-19
                if (MPG == 12) cw |= ((y & 0xFu) | ((y >> 12) & 0xF0u)) << (8 * k);
                rowp[2 * k] = pairs_lds + (2 * k) * (NG * 256) + ((y & 0xFu) << 4);
                rowp[2 * k + 1] = pairs_lds + (2 * k + 1) * (NG * 256) + (((y >> 16) & 0xFu) << 4);
            }
            // windows starting past the span belong to the next launch / do not exist; a window that cannot score (dead)
            // takes its row 0 from the zero row: without the folded threshold bit 15 of its sums stays clear
            const bool dead = (badbits != 0) || (rel0 + lane >= a.span);
            if (!__ballot(!dead)) continue;                  // nothing scorable in this chunk (wave-uniform)
            if (dead) rowp[0] = pairs_lds + pair_bytes;
            // The row addresses as opaque registers: left alone hipcc keeps only the lane part ((pair code) << 4) and re-adds
            // the LDS base of the tables at EVERY group's look-ups (four v_add_u32 per group of 23 VALU instructions in all);
            // as finished addresses the group offset g * 256 is the immediate of the ds_read_b128.
#pragma unroll
            for (int t = 0; t < NP; ++t) {
                uint32_t x = (uint32_t)(uintptr_t)rowp[t];
                asm volatile("" : "+v"(x));
                rowp[t] = (lds_cptr)(uintptr_t)x;
            }

            const uint32_t relpos = (uint32_t)(rel0 + lane);
            if constexpr (MPG == 12) {
                // all groups, then ONE push for the chunk: at most 64 items, and fewer than 64 were waiting -> always room
                const uint32_t gm = lib_dispatch_mask<1, NG, NP>(npair, rowp, ng_real);
                const unsigned long long mk = __builtin_amdgcn_ballot_w64(gm != 0u);
                if (mk) {                                    // wave-uniform
                    if (gm != 0u) {
                        const int slot = qn + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
                        q_pos[slot] = relpos;
                        q_p0[slot] = gm;
                        q_p1[slot] = cw;
                    }
                    qn += __popcll(mk);
                }
                while (qn >= 64) {                           // the top 64 items; the rest stays (LIFO)
                    dense(qn - 64, 64);
                    qn += requeued - 64;
                }
            } else {
                int g = lib_dispatch<true, (NP == 8 ? 1 : NP / 2 + 1), NG, NP>(npair, 0, rowp, qn, relpos, cw, q_pos, q_p0, q_p1, ng_real);
                for (;;) {
                    while (qn >= 64) {                       // the top 64 items; the rest stays (LIFO)
                        dense(qn - 64, 64);
                        qn += requeued - 64;
                    }
                    if (g >= ng_real) break;
                    g = lib_dispatch<false, (NP == 8 ? 1 : NP / 2 + 1), NG, NP>(npair, g, rowp, qn, relpos, cw, q_pos, q_p0, q_p1, ng_real);
                }
            }
        }
    }
    while (qn > 0) {                                    // the tail: at most 63 items + what they put back
        dense(0, qn);
        qn = requeued;
    }
}

// LDS bytes of one motif group of a pass (must match the carve-up in k_library): its slice of every pair row and of the
// zero row, and the exact tables + thresholds of its motifs
size_t lib_group_bytes(int m, int npair, bool has_struct, int np_bucket)
{
    return (size_t)(npair + 1) * 256 + (size_t)lib_mpg(np_bucket) * ((size_t)m * 32 + (has_struct ? (size_t)m * 64 : 0) + 16);
}

size_t lib_queue_bytes(int np_bucket) { return (size_t)(lib_block(np_bucket) / 64) * LIB_QCAP * 3 * 4 + 16; }     // + the chunk ticket

size_t lib_lds_bytes(int m, int npair, int ng, bool has_struct, int np_bucket)
{
    return lib_group_bytes(m, npair, has_struct, np_bucket) * (size_t)ng + lib_queue_bytes(np_bucket);
}

int lib_np_bucket(int m) { return m <= 16 ? 8 : (m <= 32 ? 16 : 32); }

// group counts a pass may have (each is a kernel instantiation), largest first
static const int LIB_NG_8[] = {16, 11, 8, 6, 4, 2};
static const int LIB_NG_16[] = {8, 4, 2};
static const int LIB_NG_32[] = {4, 2};

// the smallest supported group count >= want_groups that is <= max_groups; when none is large enough, the largest
// one within max_groups (the caller then needs more passes); 0 when even the smallest does not fit
int lib_pick_ng(int np_bucket, int want_groups, int max_groups)
{
    const int *set = np_bucket == 8 ? LIB_NG_8 : (np_bucket == 16 ? LIB_NG_16 : LIB_NG_32);
    const int n = np_bucket == 8 ? 6 : (np_bucket == 16 ? 3 : 2);
    int best = 0;
    for (int i = 0; i < n; ++i) {
        if (set[i] > max_groups) continue;
        if (best == 0 || set[i] >= want_groups) best = set[i];
    }
    return best;
}

template <int NG, int NP, typename PROF_T, bool HAS_STRUCT>
static hipError_t launch_library_inst(const LibArgs &a, unsigned grid, size_t lds, hipStream_t stream)
{
    auto kern = k_library<NG, NP, PROF_T, HAS_STRUCT>;
    static std::atomic<uint64_t> configured{0};     // per instantiation, one bit per device
    hipError_t e = allow_dynamic_lds(reinterpret_cast<const void *>(kern), configured, 160 * 1024);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(lib_block(NP)), lds, stream, a);
    return hipGetLastError();
}

template <int NG, int NP>
static hipError_t launch_library_ng(const LibArgs &a, unsigned grid, size_t lds, hipStream_t stream)
{
    if (!a.pssm) return launch_library_inst<NG, NP, float, false>(a, grid, lds, stream);
    if (a.profile_dtype == PROFILE_LETTERS2) return launch_library_inst<NG, NP, uint8_t, true>(a, grid, lds, stream);     // two-FASTA library
    if (a.profile_dtype == PFMSCAN_PROFILE_F64) return launch_library_inst<NG, NP, double, true>(a, grid, lds, stream);
    return launch_library_inst<NG, NP, float, true>(a, grid, lds, stream);
}

hipError_t launch_library(const LibArgs &a, int n_cu, hipStream_t stream)
{
    if (a.span <= 0 || a.nmp <= 0) return hipSuccess;
    const int np = lib_np_bucket(a.m);
    const size_t lds = lib_lds_bytes(a.m, a.npair, a.ng, a.pssm != nullptr, np);
    if (lds > 160 * 1024 || a.ng * lib_mpg(np) != a.nmp || a.seg_positions != ((int64_t)1 << LIB_SEG_SHIFT) ||
        ((a.n_seg + n_cu - 1) / n_cu) * (a.seg_positions >> 6) > 0x7FFFFFFF)
        return hipErrorInvalidValue;
    const unsigned grid = (unsigned)std::min<int64_t>(a.n_seg, n_cu);
#define LIB_CASE(NGV, NPV) \
    if (np == NPV && a.ng == NGV) return launch_library_ng<NGV, NPV>(a, grid, lds, stream)
    LIB_CASE(16, 8);
    LIB_CASE(11, 8);
    LIB_CASE(8, 8);
    LIB_CASE(6, 8);
    LIB_CASE(4, 8);
    LIB_CASE(2, 8);
    LIB_CASE(8, 16);
    LIB_CASE(4, 16);
    LIB_CASE(2, 16);
    LIB_CASE(4, 32);
    LIB_CASE(2, 32);
#undef LIB_CASE
    return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------
// k_library8 -- libraries of GENERIC-alphabet letter tables (up to 7 letters + the foreign code; the structure strings of
// `rnascan -q struct_library structs.fa`, SURVEY 8f N1 x N4) in one pass over an 8-code stream.
//
// Reference semantics: matrix.py:25-43 (_py_calculate: sequential sum of Python floats, fp64, NO float32 cast; an unknown
// letter makes the window NaN), rnascan.py:263 (hit <=> score > threshold, strict), pfmutil.py:89-133 (the multi-PFM format).
//
// k_library's machinery with ONE-letter table rows: row j of a motif group holds, for each of the 8 codes, the 16-bit credits
// of that letter at motif position j for the group's EIGHT motifs (16 bytes; 8 entries = 128 bytes = 32 banks once: lanes with
// the same letter read the same address, lanes with different letters different banks).  Rows are padded to a multiple of 4
// with rows of full credit (the host builds the credits for the padded count: pfmscan_library_api.hip), so the row count is
// one of 4 / 8 / 12 / 16 (/ 20 .. 32 in the wide bucket) and a compile-time constant of the unrolled look-ups.  The foreign
// code and separators (7), NaN and -inf cells carry NO credit: such a window cannot reach the flag bit.  Bits 3-7 of a code
// byte (bit 3 = "written in lower case") are ignored.  Phase B, 64 items at a time: the window's letters again (L2), the exact
// fp64 sum from an LDS copy of the tables (the layout of k_library's structure tables: [m * 4][NMP][2]), fp64 compare.
// NR = rows the code is unrolled for: 16 (m <= 16, 1024 threads) or 32 (m <= 32, 512 threads).
// ---------------------------------------------------------------------------
template <bool FAST, int NG, int NR>
__device__ __forceinline__ int lib8_dispatch(const int rows, const int g, const lds_cptr (&rowp)[NR], int &qn, const uint32_t relpos,
                                             uint32_t *q_pos, uint32_t *q_p0, uint32_t *q_p1, const int ng_real)
{
#define LIB8_ROWS(K)                                                                                              \
    if (rows == K) {                                                                                              \
        if (FAST) return lib_octets_fast<K, NG, NR, 128>(rowp, qn, relpos, 0u, q_pos, q_p0, q_p1, ng_real);       \
        return lib_octets_slow<K, NG, NR, 128>(g, rowp, qn, relpos, 0u, q_pos, q_p0, q_p1, ng_real);              \
    }
    if constexpr (NR == 16) {
        LIB8_ROWS(4) LIB8_ROWS(8) LIB8_ROWS(12)
        if (FAST) return lib_octets_fast<16, NG, NR, 128>(rowp, qn, relpos, 0u, q_pos, q_p0, q_p1, ng_real);
        return lib_octets_slow<16, NG, NR, 128>(g, rowp, qn, relpos, 0u, q_pos, q_p0, q_p1, ng_real);
    } else {
        LIB8_ROWS(20) LIB8_ROWS(24) LIB8_ROWS(28)
        if (FAST) return lib_octets_fast<32, NG, NR, 128>(rowp, qn, relpos, 0u, q_pos, q_p0, q_p1, ng_real);
        return lib_octets_slow<32, NG, NR, 128>(g, rowp, qn, relpos, 0u, q_pos, q_p0, q_p1, ng_real);
    }
#undef LIB8_ROWS
}

template <int NG, int NR>
__global__ __launch_bounds__(lib_block(NR)) void k_library8(const LibArgs a)
{
    constexpr int LIB_BLOCK = lib_block(NR);
    constexpr int LIB_WAVES = LIB_BLOCK / 64;
    constexpr int MPG = 8;
    constexpr int NMP = NG * MPG;
    constexpr int NRAW = NR / 4 + 1;                 // aligned code dwords a lane loads: NR letters at any lane & 3
    static_assert(lib_mpg(NR) == 8, "the (window, group) item format of the 16-bit credits");
    extern __shared__ __align__(16) unsigned char smem[];
    const int m = a.m, rows = a.npair;               // rows: m rounded up to a multiple of 4 (the padding rows are full credit)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    // ---- LDS carve-up (lib8_lds_bytes) ----
    const int cred_bytes = rows * NG * 128;          // + one all-zero row of NG * 128 bytes: row 0 of the windows that do not exist
    uint32_t *ticket = reinterpret_cast<uint32_t *>(smem);
    uint32_t *cred = reinterpret_cast<uint32_t *>(smem + 16);
    double *tab = reinterpret_cast<double *>(smem + 16 + cred_bytes + NG * 128);          // [m * 4][NMP][2]
    double *thr_t = tab + (size_t)m * 8 * NMP;
    uint32_t *qbase = reinterpret_cast<uint32_t *>(thr_t + NMP);
    uint32_t *q_pos = qbase + (size_t)wave * LIB_QCAP * 3;
    uint32_t *q_p0 = q_pos + LIB_QCAP;
    uint32_t *q_p1 = q_p0 + LIB_QCAP;

    for (int i = threadIdx.x; i < cred_bytes / 16; i += LIB_BLOCK)
        reinterpret_cast<u32x4 *>(cred)[i] = reinterpret_cast<const u32x4 *>(a.pairs)[i];
    for (int i = threadIdx.x; i < NG * 8; i += LIB_BLOCK) reinterpret_cast<u32x4 *>(cred)[cred_bytes / 16 + i] = u32x4{0u, 0u, 0u, 0u};
    for (int i = threadIdx.x; i < m * 8 * NMP; i += LIB_BLOCK) tab[i] = a.pssm[i];
    for (int i = threadIdx.x; i < NMP; i += LIB_BLOCK) thr_t[i] = a.thr_struct[i];
    if (threadIdx.x == 0) *ticket = LIB_WAVES;
    __syncthreads();                                // the only workgroup barrier

    const int64_t n_pos = a.n_pos;
    const int ng_real = a.ng_real;
    const int shard = blockIdx.x & (a.hit_shards - 1);
    unsigned long long *counter = a.hit_count + (size_t)shard * HIT_COUNTER_STRIDE;
    const unsigned long long shard_off = (unsigned long long)shard * (unsigned long long)a.shard_cap;
    const lds_cptr cred_lds = lds_ptr_of(cred);

    // ---- phase B: the top cnt (<= 64) items of this wave's queue, one per lane and ONE motif per lane; what is left of an
    // item goes back to the queue (k_library's eight-per-entry form: p0 / p1 = the sign bytes of the accumulators) ----
    int requeued = 0;
    auto dense = [&](int first, int cnt) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const bool have = lane < cnt;
        const int idx = first + (have ? lane : 0);
        const uint32_t rel = q_pos[idx], p0 = q_p0[idx], p1 = q_p1[idx];
        const uint32_t g = p1 & 0x3Fu;
        uint32_t bits = !have ? 0u : (((p0 >> 7) & 0x01010101u) | ((p1 >> 6) & 0x02020202u));
        const int64_t p = a.pos_base + (int64_t)rel;
        const bool act = bits != 0;
        const int q = act ? __builtin_ctz(bits) : 0;
        bits &= bits - 1;
        const bool again = bits != 0;
        const unsigned long long more = __builtin_amdgcn_ballot_w64(again);
        if (more) {
            if (again) {
                const int slot = first + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(more >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)more, 0u));
                q_pos[slot] = rel;                  // slots first .. first + popc - 1 are being freed by this batch (all read above)
                q_p0[slot] = (bits & 0x01010101u) << 7;
                q_p1[slot] = ((bits & 0x02020202u) << 6) | g;
            }
            requeued = __popcll(more);
        } else {
            requeued = 0;
        }
        const int mo = (int)g * MPG + (q >> 3) + 4 * (q & 1);           // pass-local motif
        // the window's letters again (L2: the wave read them a few chunks ago), then matrix.py:25-43: sequential fp64 sum
        const int64_t al = p & ~(int64_t)3;
        uint32_t raw[NRAW];
#pragma unroll
        for (int k = 0; k < NRAW; ++k) raw[k] = (k < (m + 3) / 4 + 1) ? lib_codes4(a.codes, al + 4 * k, n_pos) : 0u;
        double sc = 0.0;
#pragma unroll
        for (int k = 0; k < NR / 4; ++k) {
            const uint32_t w = __builtin_amdgcn_alignbyte(raw[k + 1], raw[k], (uint32_t)(p & 3));
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int j = 4 * k + b;
                if (j < m) {
                    const uint32_t c = (w >> (8 * b)) & 7u;
                    sc += tab[((size_t)(j * 4 + (int)(c >> 1)) * NMP + mo) * 2 + (c & 1u)];
                }
            }
        }
        const bool ok = act && (sc > thr_t[mo]);                          // fp64 compare: no float32 cast on this path
        const unsigned long long hm = __builtin_amdgcn_ballot_w64(ok);
        if (hm) {
            const int nh = __popcll(hm);
            unsigned long long base = 0;
            if (lane == 0) base = atomicAdd(counter, (unsigned long long)nh);
            const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)base), hi = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32));
            base = ((unsigned long long)hi << 32) | lo;
            if (ok) {
                const unsigned long long slot = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(hm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hm, 0u));
                if ((int64_t)slot < a.shard_cap) {
                    a.hit_pos[shard_off + slot] = p + a.pos_offset;
                    a.hit_motif[shard_off + slot] = a.motif_base + mo;
                    if (a.hit_seq) a.hit_seq[shard_off + slot] = (float)sc;
                    a.hit_struct[shard_off + slot] = sc;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };

    // ---- phase A: this workgroup's segments, this wave's 64-window chunks (LDS ticket, as in k_library) ----
    int qn = 0;
    constexpr int seg_shift = LIB_SEG_SHIFT - 6;
    const int bid = (int)blockIdx.x, grid = (int)gridDim.x;
    const int64_t my_segs = bid < a.n_seg ? (a.n_seg - bid + grid - 1) / grid : 0;
    const uint32_t n_units = (uint32_t)(my_segs << seg_shift);
    uint32_t drawn = 0;
    auto next_unit = [&]() -> uint32_t {
        if (lane == 0) drawn = atomicAdd(ticket, 1u);
        return __builtin_amdgcn_readfirstlane(drawn);
    };
    for (uint32_t u = (uint32_t)wave; u < n_units; u = next_unit()) {
        const int64_t rel0 = ((int64_t)((uint32_t)bid + (u >> seg_shift) * (uint32_t)grid) << (seg_shift + 6)) +
                             (int64_t)((u & ((1u << seg_shift) - 1u)) << 6);
        if (rel0 >= a.span) continue;                    // wave-uniform
        const int64_t p0 = a.pos_base + rel0;            // multiple of 64
        const int64_t al = p0 + (lane & ~3);
        uint32_t raw[NRAW];
        if (p0 + 64 + 4 * NRAW <= n_pos) {
#pragma unroll
            for (int k = 0; k < NRAW; ++k) raw[k] = *reinterpret_cast<const uint32_t *>(a.codes + al + 4 * k);
        } else {
#pragma unroll
            for (int k = 0; k < NRAW; ++k) raw[k] = lib_codes4(a.codes, al + 4 * k, n_pos);
        }
        lds_cptr rowp[NR];
#pragma unroll
        for (int k = 0; k < NR / 4; ++k) {
            const uint32_t w = __builtin_amdgcn_alignbyte(raw[k + 1], raw[k], (uint32_t)(lane & 3)) & 0x07070707u;
#pragma unroll
            for (int b = 0; b < 4; ++b)                  // (rows beyond `rows` are never looked up)
                rowp[4 * k + b] = cred_lds + (4 * k + b) * (NG * 128) + (((w >> (8 * b)) & 7u) << 4);
        }
        // windows starting past the span belong to the next launch / do not exist: their row 0 is the zero row, and without the
        // folded threshold the flag bit of their sums stays clear.  (Windows that run over the end of the stream hold code 7.)
        if (rel0 + lane >= a.span) rowp[0] = cred_lds + cred_bytes;
#pragma unroll
        for (int t = 0; t < NR; ++t) {                  // finished addresses: the group offset is the ds_read's immediate
            uint32_t x = (uint32_t)(uintptr_t)rowp[t];
            asm volatile("" : "+v"(x));
            rowp[t] = (lds_cptr)(uintptr_t)x;
        }
        const uint32_t relpos = (uint32_t)(rel0 + lane);
        int g = lib8_dispatch<true, NG, NR>(rows, 0, rowp, qn, relpos, q_pos, q_p0, q_p1, ng_real);
        for (;;) {
            while (qn >= 64) {                           // the top 64 items; the rest stays (LIFO)
                dense(qn - 64, 64);
                qn += requeued - 64;
            }
            if (g >= ng_real) break;
            g = lib8_dispatch<false, NG, NR>(rows, g, rowp, qn, relpos, q_pos, q_p0, q_p1, ng_real);
        }
    }
    while (qn > 0) {                                    // the tail
        dense(0, qn);
        qn = requeued;
    }
}

// LDS bytes of one motif group of a k_library8 pass: its slice of every (padded) credit row and of the zero row, the exact
// tables and thresholds of its 8 motifs
size_t lib8_group_bytes(int m) { return (size_t)(lib8_rows(m) + 1) * 128 + 8 * ((size_t)m * 64 + 8); }
size_t lib8_lds_bytes(int m, int ng) { return lib8_group_bytes(m) * (size_t)ng + lib_queue_bytes(m <= 16 ? 16 : 32); }

static const int LIB8_NG[] = {16, 8, 4, 2};
int lib8_pick_ng(int want_groups, int max_groups)
{
    int best = 0;
    for (int i = 0; i < 4; ++i) {
        if (LIB8_NG[i] > max_groups) continue;
        if (best == 0 || LIB8_NG[i] >= want_groups) best = LIB8_NG[i];
    }
    return best;
}

template <int NG, int NR>
static hipError_t launch_library8_inst(const LibArgs &a, unsigned grid, size_t lds, hipStream_t stream)
{
    auto kern = k_library8<NG, NR>;
    static std::atomic<uint64_t> configured{0};
    hipError_t e = allow_dynamic_lds(reinterpret_cast<const void *>(kern), configured, 160 * 1024);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(lib_block(NR)), lds, stream, a);
    return hipGetLastError();
}

hipError_t launch_library8(const LibArgs &a, int n_cu, hipStream_t stream)
{
    if (a.span <= 0 || a.nmp <= 0) return hipSuccess;
    const size_t lds = lib8_lds_bytes(a.m, a.ng);
    if (a.m < 1 || a.m > 32 || a.npair != lib8_rows(a.m) || lds > 160 * 1024 || a.ng * 8 != a.nmp ||
        a.seg_positions != ((int64_t)1 << LIB_SEG_SHIFT) || ((a.n_seg + n_cu - 1) / n_cu) * (a.seg_positions >> 6) > 0x7FFFFFFF)
        return hipErrorInvalidValue;
    const unsigned grid = (unsigned)std::min<int64_t>(a.n_seg, n_cu);
#define LIB8_CASE(NGV)                                                                       \
    if (a.ng == NGV) return a.m <= 16 ? launch_library8_inst<NGV, 16>(a, grid, lds, stream) \
                                      : launch_library8_inst<NGV, 32>(a, grid, lds, stream)
    LIB8_CASE(16);
    LIB8_CASE(8);
    LIB8_CASE(4);
    LIB8_CASE(2);
#undef LIB8_CASE
    return hipErrorInvalidValue;
}

}  // namespace pfmscan
