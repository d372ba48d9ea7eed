// pfmscan_letters8.hip -- thresholded hits of GENERIC-alphabet letter scans (up to 7 letters + the foreign code) with the
// score kept in fp64: the structure letter-string modes of rnascan (`-q pfm structs.fa`, and the structure side of
// `-p pfm -q pfm seqs.fa structs.fa`).
//
// Reference semantics restated here (upstream paths, v0.10.2):
//   matrix.py:25-43    _py_calculate: score = 0.0; score += self[letter][position] per motif position, Python floats
//                      (fp64, NO float32 cast); an unknown letter makes the window NaN
//   rnascan.py:263     pm.search(seq, threshold=minscore): hit <=> score > threshold (strict; NaN and -inf never pass)
//   rnascan.py:416-434 combine(): a combined hit needs a row in BOTH tables for (Sequence_ID, Start, End), i.e.
//                      seq > minscore AND struct > minscore
//
//  k_letters_cred8  hits of ONE letter table over an 8-code stream at a finite threshold, PFMs up to width 32.
//              The integer prefilter of k_letters_cred (pfmscan_kernels.hip) with ONE-letter rows: the credit table
//              has one entry per CODE holding that letter's credit for EVERY motif row, two rows per dword
//                  ctab[c] = { (row0 | row1 << 16), (row2 | row3 << 16), ... }                           NJ dwords
//              (8 entries of at most 64 bytes: lanes that hold the same letter read the same address, lanes with
//              different letters different banks -- conflict-free whatever the letters are).  A lane owns W = 16
//              consecutive windows and reads ctab once per position q of its band.  Row 2k of position q belongs to
//              window q - 2k, row 2k+1 to window q - 2k - 1, so with the packs P[u] = (window u | window u-1 << 16)
//                  P[u] += dword_k(q)   for q - 2k = u
//                  sum(w) = lo(P[w]) + hi(P[w + 1])           bit 15 set <=> the window may be a hit
//              (threshold folded into row 0, one-sided rounding: build_credits).  Foreign letters and separators
//              (code 7), NaN and -inf cells get NO credit: such a window scores NaN / -inf and can never pass the strict
//              `>`.  Survivors are queued per wave and get the exact fp64 score 64 at a time, one per lane; hits leave
//              through wave-private LDS queues with one returning global atomic per flush (as in k_letters_cred).
//
//  k_letters_at     fp64 letter score at a list of candidate windows of a SECOND code stream: the verify phase of the
//              two-FASTA combined scan (RNA letters pass over everything, the structure letters only at its hits).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "pfmscan_device.hpp"

namespace pfmscan {

struct Cred8Table {
    uint32_t d[8][16];                                 // [code][row pair k] = credit of row 2k | credit of row 2k+1 << 16
};

constexpr int Q8_CAP = 128;                            // hits a wave can park (12 bytes each)

#ifndef CRED8_MIN_WG
#define CRED8_MIN_WG                                   // A/B builds: -DCRED8_MIN_WG=",4" (tools/build_variant.sh)
#endif
template <int NJ>
__global__ __launch_bounds__(BLOCK CRED8_MIN_WG) void k_letters_cred8(const ScanArgs a, const Cred8Table ct)
{
    constexpr int W = 16;                              // windows per lane = one round per tile
    constexpr int LET_TILE = BLOCK * W;
    constexpr int NPOS = W + 2 * NJ - 1;               // positions a lane looks up: q = 0 .. W + 2 NJ - 2
    constexpr int NWD = (NPOS + 3) / 4;                // code dwords holding bytes 0 .. NPOS - 1
    // entry STRIDE in bytes: 8 / 16 / 32 for up to 8 row pairs (8 entries span at most 256 B = every bank once); wider entries
    // would be 64 bytes, 8 of them = the banks TWICE (codes c and c + 4 on the same banks: 2-way conflicts, measured +50 % at
    // w = 18 against w = 16) -- stride 80 puts the 16-byte pieces of the 8 codes on 8 different bank quads again
    constexpr int ESTR = NJ <= 2 ? 8 : (NJ <= 4 ? 16 : (NJ <= 8 ? 32 : 80));
    constexpr int PSH = NJ <= 2 ? 3 : (NJ <= 4 ? 4 : (NJ <= 8 ? 5 : 4));      // the codes are pre-shifted inside their bytes: byte = code * 8, * 16 or * 32 (7 * 32 still fits)
    constexpr int EDW = ESTR / 4;                      // dwords per entry
    constexpr int TROWS = 2 * NJ;                      // rows of the exact letter table (rows m .. are zeros)
    constexpr int NRAW = (2 * NJ + 3) / 4 + 1;         // aligned code dwords that hold a window's 2 NJ letters at any p & 3
    constexpr int NWAVE = BLOCK / 64;
    static_assert(NPOS - W < CODE_HALO - 4, "look-ahead exceeds the code halo");
    __shared__ __align__(16) double tbl[TROWS * 8];
    __shared__ __align__(16) uint32_t ctab[8 * EDW];
    __shared__ __align__(16) uint8_t cbuf[2][LET_TILE + CODE_HALO];
    __shared__ __align__(8) double q_sc[NWAVE][Q8_CAP];
    __shared__ uint32_t q_pos[NWAVE][Q8_CAP];          // positions in both queues are relative to the workgroup's first tile
    __shared__ uint32_t sv_pos[NWAVE][128];            // survivors of the prefilter waiting for their exact score
    __shared__ int q_n[NWAVE], snap[2][NWAVE];
    __shared__ unsigned long long s_base;
    const int m = a.m;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t n_pos = a.n_pos;
    const int ntile = a.tiles_per_block;
    const int64_t first = (int64_t)blockIdx.x * ntile * LET_TILE;
    if (first >= n_pos) return;                        // whole workgroup

    CodeStage<LET_TILE> cs;
    cs.fetch(a.codes, first, n_pos);
    // rows m .. are zeros: x + 0.0 == x for every x a sum that started at +0.0 can hold (never -0.0)
    for (int i = threadIdx.x; i < TROWS * 8; i += BLOCK) tbl[i] = i < m * 8 ? a.letter_table[i] : 0.0;
    // The credit table travels WITH the launch (a by-value argument: no device copy to keep in step with the threshold, no
    // synchronisation when it changes) and is read straight from the kernarg segment with per-lane vector loads: indexed
    // as `ct.d[..][..]` hipcc kept all 128 dwords in SGPRs and the widest instantiation spilled 57 of them to VGPR lanes.
    static_assert(sizeof(ScanArgs) % alignof(Cred8Table) == 0, "ct follows a in the kernarg segment");
    (void)ct;
    typedef const __attribute__((address_space(4))) unsigned char *karg_ptr;
    const __attribute__((address_space(4))) uint32_t *ktab =
        (const __attribute__((address_space(4))) uint32_t *)((karg_ptr)__builtin_amdgcn_kernarg_segment_ptr() + sizeof(ScanArgs));
    for (int i = threadIdx.x; i < 8 * EDW; i += BLOCK) ctab[i] = (i % EDW) < NJ ? ktab[(i / EDW) * 16 + (i % EDW)] : 0u;
    if (threadIdx.x < NWAVE) q_n[threadIdx.x] = 0;
    cs.park(cbuf[0]);
    if (ntile > 1 && first + LET_TILE < n_pos) cs.fetch(a.codes, first + LET_TILE, n_pos);
    __syncthreads();

    const char *cbytes = (const char *)ctab;
    const int shard = blockIdx.x & (a.hit_shards - 1);
    const unsigned long long shard_off = (unsigned long long)shard * (unsigned long long)a.capacity;
    unsigned long long *counter = a.hit_count + shard * HIT_COUNTER_STRIDE;
    uint32_t *my_pos = q_pos[wave];
    double *my_sc = q_sc[wave];

    auto store_hit = [&](unsigned long long slot, int64_t pos, double sc) {
        if ((int64_t)slot < a.capacity) {             // capacity is per shard
            a.hit_pos[shard_off + slot] = pos + a.pos_offset;
            if (a.hit_seq) a.hit_seq[shard_off + slot] = (float)sc;
            if (a.hit_struct) a.hit_struct[shard_off + slot] = sc;
        }
    };
    // this wave's queue -> global at base; all 64 lanes (LDS operations of one wave execute in order)
    auto drain = [&](unsigned long long base, int n) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int i = lane; i < n; i += 64) store_hit(base + i, first + (int64_t)my_pos[i], my_sc[i]);
        if (lane == 0) q_n[wave] = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    auto wave_flush = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int n = __builtin_amdgcn_readfirstlane(q_n[wave]);
        if (n == 0) return;
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(counter, (unsigned long long)n);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)base), hi = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32));
        drain(((unsigned long long)hi << 32) | lo, n);
    };

    int qn_ub = 0;                                     // wave-uniform upper bound of q_n[wave]
    int sv_n = 0;                                      // wave-uniform length of the survivor queue (< 64 between windows)
    uint32_t *my_sv = sv_pos[wave];
    // exact score of survivors [at, at + cnt) of this wave's queue, one per lane (matrix.py:25-43: sequential fp64 sum)
    auto exact_batch = [&](int at, int cnt) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (qn_ub + cnt > Q8_CAP) {                    // room for a hit per lane in the hit queue
            wave_flush();
            qn_ub = 0;
        }
        qn_ub += cnt;
        if (lane < cnt) {
            const int64_t p = first + (int64_t)my_sv[at + lane];
            const int64_t al = p & ~(int64_t)3;
            uint32_t raw[NRAW];
#pragma unroll
            for (int k = 0; k < NRAW; ++k) raw[k] = load_codes4(a.codes, al + 4 * k, n_pos);
            double sc = 0.0;
#pragma unroll
            for (int k = 0; k < NRAW - 1; ++k) {
                const uint32_t cw = __builtin_amdgcn_alignbyte(raw[k + 1], raw[k], (uint32_t)(p & 3));
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int j = 4 * k + b;
                    if (j < TROWS) sc += tbl[j * 8 + ((cw >> (8 * b)) & 7u)];           // rows m .. 2 NJ - 1 are zeros
                }
            }
            if (sc > a.thr_seq) {                      // fp64 compare: no float32 cast on this path
                const int slot = atomicAdd(&q_n[wave], 1);     // LDS
                my_pos[slot] = (uint32_t)(p - first);
                my_sc[slot] = sc;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    for (int tb = 0; tb < ntile; ++tb) {
        const int64_t tile0 = first + (int64_t)tb * LET_TILE;
        if (tile0 >= n_pos) break;                     // uniform; the previous tile flushed (it was the last)
        const uint8_t *cb = cbuf[tb & 1];
        const int off0 = threadIdx.x * W;
        // xs[d] byte k = (code at byte 4d + k) << PSH: the entry offset of position q is one v_bfe_u32 (and one more
        // for 80-byte entries; 32-byte entries used to pay a shift per position too: byte = code * 16, -1.7 of 13 VALU per window)
        uint32_t xs[NWD];
#pragma unroll
        for (int d = 0; d < NWD; ++d) xs[d] = (*reinterpret_cast<const uint32_t *>(cb + off0 + 4 * d) & 0x07070707u) << PSH;   // inside the halo
        uint32_t pk[W + 1];
#pragma unroll
        for (int i = 0; i < W + 1; ++i) pk[i] = 0u;
        // q, g and k are constant expressions (static_for, not `#pragma unroll`): every pk[] / dj[] index is a register name
        static_for<0, NPOS>([&](auto qc) __attribute__((always_inline)) {
            constexpr int q = decltype(qc)::value;
            // ONE v_bfe_u32 per position, written out: hipcc knows that only bits 5-7 of a byte can be set, narrows the
            // mask to 0xE0 and then no longer recognises the bit-field extract -- it issued a shift AND a mask per position,
            // 2 of the body's 9 VALU instructions per window
            uint32_t off;
            if constexpr ((q & 3) == 0) asm("v_bfe_u32 %0, %1, 0, 8" : "=v"(off) : "v"(xs[q >> 2]));
            else if constexpr ((q & 3) == 1) asm("v_bfe_u32 %0, %1, 8, 8" : "=v"(off) : "v"(xs[q >> 2]));
            else if constexpr ((q & 3) == 2) asm("v_bfe_u32 %0, %1, 16, 8" : "=v"(off) : "v"(xs[q >> 2]));
            else asm("v_bfe_u32 %0, %1, 24, 8" : "=v"(off) : "v"(xs[q >> 2]));
            if constexpr (ESTR == 80) off += off << 2;  // code * 16 * 5: one v_lshl_add_u32
            // position q feeds the windows u = q - 2k in [0, W], i.e. the row pairs k in [(q - W + 1) / 2, q / 2]: at most
            // W / 2 + 1 of the NJ dwords of its entry -- only the 16-byte groups that hold one of them are read
            constexpr int KMAX = NJ - 1;
            constexpr int klo = q > W ? (q - W + 1) / 2 : 0, khi = q / 2 < KMAX ? q / 2 : KMAX;
            uint32_t dj[16] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
            if constexpr (NJ <= 2) {
                const u32x2 e = *reinterpret_cast<const u32x2 *>(cbytes + off);
                dj[0] = e[0];
                dj[1] = e[1];
            } else {
                static_for<0, (NJ + 3) / 4>([&](auto gc) __attribute__((always_inline)) {
                    constexpr int g = decltype(gc)::value;
                    if constexpr (!(4 * g > khi || 4 * g + 3 < klo)) {
                        if constexpr (NJ - 4 * g >= 3) {
                            const u32x4 e = *reinterpret_cast<const u32x4 *>(cbytes + off + 16 * g);
                            dj[4 * g] = e[0];
                            dj[4 * g + 1] = e[1];
                            dj[4 * g + 2] = e[2];
                            dj[4 * g + 3] = e[3];
                        } else {                       // a last group of two row pairs (NJ = 6)
                            const u32x2 e = *reinterpret_cast<const u32x2 *>(cbytes + off + 16 * g);
                            dj[4 * g] = e[0];
                            dj[4 * g + 1] = e[1];
                        }
                    }
                });
            }
            static_for<0, NJ>([&](auto kc) __attribute__((always_inline)) {
                constexpr int k = decltype(kc)::value;
                constexpr int u = q - 2 * k;           // rows 2k (lo: window u) and 2k + 1 (hi: window u - 1)
                if constexpr (u >= 0 && u <= W) pk[u] += dj[k];
            });
        });
        // The 16 sums two at a time: with A = P[w+1] one v_alignbit (lo(P[w]) << 16 | hi(P[w+2])) + one v_pk_add_u16 give
        // (sum(w) << 16 | sum(w+1)); bit 15 of a sum (they stay below 2^16) is its flag.  `surv` collects the flags of the
        // pair k = 0..7 (windows 2k and 2k + 1) at bits 24 + k and 8 + k: two VALU instructions per window, hits or not.
        uint32_t surv = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t x = __builtin_amdgcn_alignbit(pk[2 * k], pk[2 * k + 2], 16);
            const u16x2 r = __builtin_bit_cast(u16x2, pk[2 * k + 1]) + __builtin_bit_cast(u16x2, x);
            surv = (surv >> 1) | (__builtin_bit_cast(uint32_t, r) & 0x80008000u);
        }
        // Survivors -> the wave's queue (positions only; windows past the end hold SEP codes: no credit).  ONE rolled
        // loop: every pass each lane that still has a survivor hands over its lowest one (an unrolled pass per window
        // inlined the exact score 17 times: 25 k instructions and 57 spilled SGPRs in the widest instantiation).
        while (__builtin_amdgcn_ballot_w64(surv != 0)) {
            const bool sv = surv != 0;
            const unsigned long long sb = __builtin_amdgcn_ballot_w64(sv);
            if (sv) {
                const int b = __builtin_ctz(surv);
                surv &= surv - 1;
                const int v = 2 * (b & 7) + (b < 16 ? 1 : 0);
                my_sv[sv_n + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(sb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)sb, 0u))] = (uint32_t)(tile0 - first) + (uint32_t)(off0 + v);
            }
            sv_n += __popcll(sb);
            if (sv_n >= 64) {                           // the top 64 get their exact score, the rest stays
                exact_batch(sv_n - 64, 64);
                sv_n -= 64;
            }
        }

        // tile boundary: publish the next tile's codes and this wave's queue length, ONE barrier
        const bool more = tb + 1 < ntile && tile0 + LET_TILE < n_pos;
        if (!more && sv_n > 0) {                       // last tile of the workgroup: the waiting survivors, then the final flush
            exact_batch(0, sv_n);
            sv_n = 0;
        }
        if (more) cs.park(cbuf[(tb + 1) & 1]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) snap[tb & 1][wave] = q_n[wave];
        __syncthreads();
        if (tb + 2 < ntile && tile0 + 2 * (int64_t)LET_TILE < n_pos) cs.fetch(a.codes, tile0 + 2 * (int64_t)LET_TILE, n_pos);
        int nq[NWAVE], total = 0, most = 0, before = 0;
#pragma unroll
        for (int k = 0; k < NWAVE; ++k) {
            nq[k] = snap[tb & 1][k];
            if (k < wave) before += nq[k];
            total += nq[k];
            most = most > nq[k] ? most : nq[k];
        }
        qn_ub = nq[wave];
        if (most >= Q8_CAP / 2 || (!more && total > 0)) {          // uniform: every thread read the same snapshot
            if (threadIdx.x == 0) s_base = atomicAdd(counter, (unsigned long long)total);
            __syncthreads();
            drain(s_base + (unsigned long long)before, nq[wave]);
            qn_ub = 0;
        }
    }
}

// Share of the windows that survives a credit table, for independent letters drawn uniformly from `ncol` codes: the
// exact distribution of the 16-bit credit sum (a convolution over the rows, only the reachable sums are visited).
static double credit_survival(const uint16_t *cr, int nrows, int nent, int ncol)
{
    std::vector<double> dist(65536, 0.0), next(65536, 0.0);
    std::vector<int> support{0}, nsupport;
    dist[0] = 1.0;
    for (int r = 0; r < nrows; ++r) {
        nsupport.clear();
        for (int v : support) {
            const double pv = dist[(size_t)v];
            for (int c = 0; c < ncol; ++c) {
                const int w2 = std::min(65535, v + (int)cr[r * nent + c]);
                if (next[(size_t)w2] == 0.0) nsupport.push_back(w2);
                next[(size_t)w2] += pv / ncol;
            }
            dist[(size_t)v] = 0.0;
        }
        dist.swap(next);                               // the old `dist` is all zeros again: it is the next `next`
        support.swap(nsupport);
    }
    double survive = 0.0;
    for (int v : support)
        if (v >= 32768) survive += dist[(size_t)v];
    return survive;
}

// The single-letter credit table of a motif at threshold thr -> cc (cached with the motif).  mode 1: use the prefilter;
// 2: dense threshold (more than 1/32 of the windows would survive); 3: no prefilter possible (+inf cells).
void build_cred8(const double *h_letters, int m, double thr, Cred8Cache *cc)
{
    cc->thr = thr;
    std::vector<double> rows((size_t)m * 8);
    int ncol = 0;
    for (int c = 0; c < 8; ++c) {
        bool used = false;
        for (int j = 0; j < m; ++j) {
            const double v = h_letters[j * 8 + c];
            // a NaN cell makes the window NaN, which never passes the strict `>` (rnascan.py:263): no credit, like -inf
            rows[(size_t)j * 8 + c] = std::isnan(v) ? -INFINITY : v;
            used = used || !std::isnan(v);
        }
        if (used) ncol = c + 1;
    }
    const double slack = build_credits(rows.data(), m, thr, cc->cr, 16, 8);
    cc->mode = std::isfinite(slack) ? 1 : 3;
    if (cc->mode == 1 && ncol > 0 && credit_survival(cc->cr, m, 8, ncol) > 1.0 / 32.0) cc->mode = 2;
}

// tiles one workgroup walks: ~24 workgroups per CU in the grid (see walk_tiles in pfmscan_kernels.hip)
static int walk_tiles8(int64_t ntiles, const Tuning &t)
{
    const int64_t per = (int64_t)t.n_cu * 18;
    return (int)std::min<int64_t>(32, std::max<int64_t>(1, (ntiles + per / 2) / per));
}

bool launch_letters_cred8(const ScanArgs &a, const Tuning &t, hipStream_t stream, hipError_t *err)
{
    if (!(a.hits && a.f64_hits && a.h_letters && a.cred8_cache && a.m <= 32 && t.credits && std::isfinite(a.thr_seq))) return false;
    Cred8Cache *cc = a.cred8_cache;
    Cred8Table ct;
    int mode;
    {
        // the motif keeps what the last threshold gave (credits + survivor prediction: a millisecond of host work); two host
        // threads scanning with the same motif take turns here, and each launch carries its own copy of the table
        static std::mutex cred8_mu;                    // (one lock for all motifs: pfmscan_motif objects are copied by value in places)
        std::lock_guard<std::mutex> lock(cred8_mu);
        if (!(cc->thr == a.thr_seq) || cc->mode == 0) build_cred8(a.h_letters, a.m, a.thr_seq, cc);
        mode = cc->mode;
        std::memset(&ct, 0, sizeof(ct));
        if (mode == 1)
            for (int c = 0; c < 8; ++c)
                for (int j = 0; j < a.m; ++j) ct.d[c][j >> 1] |= (uint32_t)cc->cr[j * 8 + c] << (16 * (j & 1));
    }
    if (mode != 1) return false;                       // -> the exact kernel (k_letters<..., double, HITS>)
    int nj = (a.m + 1) / 2;
    if (const char *v = std::getenv("PFMSCAN_CRED8_NJ")) nj = std::max(nj, std::atoi(v));   // tests: a wider bucket than the PFM needs
    constexpr int TILE = BLOCK * 16;
    ScanArgs b = a;
    const int64_t ntiles = (a.n_pos + TILE - 1) / TILE;
    b.tiles_per_block = walk_tiles8(ntiles, t);
    if (t.tiles_per_block > 0) b.tiles_per_block = t.tiles_per_block;
    const unsigned g = (unsigned)((ntiles + b.tiles_per_block - 1) / b.tiles_per_block);
    if (nj <= 2) hipLaunchKernelGGL((k_letters_cred8<2>), dim3(g), dim3(BLOCK), 0, stream, b, ct);
    else if (nj <= 4) hipLaunchKernelGGL((k_letters_cred8<4>), dim3(g), dim3(BLOCK), 0, stream, b, ct);
    else if (nj <= 6) hipLaunchKernelGGL((k_letters_cred8<6>), dim3(g), dim3(BLOCK), 0, stream, b, ct);
    else if (nj <= 8) hipLaunchKernelGGL((k_letters_cred8<8>), dim3(g), dim3(BLOCK), 0, stream, b, ct);
    else if (nj <= 10) hipLaunchKernelGGL((k_letters_cred8<10>), dim3(g), dim3(BLOCK), 0, stream, b, ct);     // the reference's example PFMs are 18 wide
    else if (nj <= 12) hipLaunchKernelGGL((k_letters_cred8<12>), dim3(g), dim3(BLOCK), 0, stream, b, ct);
    else hipLaunchKernelGGL((k_letters_cred8<16>), dim3(g), dim3(BLOCK), 0, stream, b, ct);
    *err = hipGetLastError();
    return true;
}

// ---------------------------------------------------------------------------
// k_letters_at -- verify phase of the two-FASTA combined scan.  The candidates are the hits of the sequence letters
// pass (positions + float32 scores, sharded like every ctx-owned hit buffer); one thread per candidate adds the m letter
// log-odds of the SECOND code stream at that position sequentially in fp64 (matrix.py:25-43) and keeps the window when
// the sum exceeds a.thr_struct (rnascan.py:263 on the structure side; combine() then joins the two tables,
// rnascan.py:422-423).  a.codes / a.letter_table are the second stream and its table here.
// ---------------------------------------------------------------------------
constexpr int LETTERS_AT_MAX_SHARDS = 64;

__global__ __launch_bounds__(BLOCK) void k_letters_at(const ScanArgs a, const int64_t *__restrict__ cand_pos,
                                                      const float *__restrict__ cand_seq,
                                                      const unsigned long long *__restrict__ cand_count,
                                                      const int64_t cand_shard_cap, const int cand_shards)
{
    __shared__ int chunk_end[LETTERS_AT_MAX_SHARDS];    // inclusive prefix of ceil(n_s / 256)
    __shared__ int64_t shard_n[LETTERS_AT_MAX_SHARDS];
    __shared__ __align__(16) double tbl[PFMSCAN_MAX_M * 8];
    const int m = a.m;
    const bool in_lds = m <= PFMSCAN_MAX_M;
    if (in_lds)
        for (int i = threadIdx.x; i < m * 8; i += BLOCK) tbl[i] = a.letter_table[i];
    if ((int)threadIdx.x < cand_shards) {
        int64_t n = (int64_t)cand_count[threadIdx.x * HIT_COUNTER_STRIDE];
        if (n > cand_shard_cap) n = cand_shard_cap;
        shard_n[threadIdx.x] = n;
        chunk_end[threadIdx.x] = (int)((n + BLOCK - 1) / BLOCK);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int s = 0; s < cand_shards; ++s) {
            run += chunk_end[s];
            chunk_end[s] = run;
        }
    }
    __syncthreads();
    const int chunks = chunk_end[cand_shards - 1];
    const double *tab = in_lds ? tbl : a.letter_table;
    const int64_t n_pos = a.n_pos;

    for (int c = blockIdx.x; c < chunks; c += gridDim.x) {          // workgroup-uniform
        int shard = 0;
        while (chunk_end[shard] <= c) ++shard;
        const int64_t i = (int64_t)(c - (shard ? chunk_end[shard - 1] : 0)) * BLOCK + threadIdx.x;
        uint32_t mask = 0;
        int64_t p = 0;
        float sq = 0.f;
        double score = 0.0;
        if (i < shard_n[shard]) {
            const int64_t at = (int64_t)shard * cand_shard_cap + i;
            p = cand_pos[at] - a.pos_offset;            // candidates carry stream positions; this buffer starts at pos_offset
            sq = cand_seq[at];
            for (int j = 0; j < m; ++j) {
                const uint32_t code = p + j < n_pos ? (uint32_t)a.codes[p + j] & 7u : (uint32_t)PFMSCAN_SEP;
                score += tab[j * 8 + code];
            }
            mask = score > a.thr_struct ? 1u : 0u;
        }
        emit_hits_block<1>(mask, [&](int) { return p; }, [&](int) { return sq; }, [&](int) { return score; }, a);
    }
}

hipError_t launch_letters_at(const ScanArgs &a, const int64_t *cand_pos, const float *cand_seq,
                             const unsigned long long *cand_count, int cand_shards, int64_t cand_shard_cap,
                             hipStream_t stream)
{
    if (cand_shard_cap <= 0 || cand_shards <= 0) return hipSuccess;
    if (cand_shards > LETTERS_AT_MAX_SHARDS || !a.codes || !a.letter_table) return hipErrorInvalidValue;
    const int64_t worst = (cand_shard_cap + BLOCK - 1) / BLOCK * cand_shards;
    const unsigned grid = (unsigned)std::min<int64_t>(worst, 2048);          // 8 workgroups per CU
    hipLaunchKernelGGL(k_letters_at, dim3(grid), dim3(BLOCK), 0, stream, a, cand_pos, cand_seq, cand_count, cand_shard_cap,
                       cand_shards);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// k_wide_letters -- letter tables WIDER than PFMSCAN_MAX_M (the reference's loops take any width: _pwm.c:34-68,
// matrix.py:25-43).  k_letters' inner loop with the table streamed through LDS in slabs of 64 rows: a workgroup stages
// the codes of its 2048 windows (+ m bytes of look-ahead) once, then for every slab loads 64 table rows (4 KB), each
// thread reads the 72 code bytes its 8 consecutive windows meet in that slab as 18 aligned dwords and adds the rows
// to its 8 running fp64 sums -- row after row, so the order of the additions, and with it every bit of the result, is
// the reference's.  One guard per row (wave-uniform) ends the last slab.  Scores (float32 of the sum or the fp64 sum
// itself) and hits (either compare) as in k_letters / k_wide.
// ---------------------------------------------------------------------------
constexpr int WIDE_SLAB = 64;
constexpr int WIDE_W = 8;
constexpr int WIDE_TILE = BLOCK * WIDE_W;
static inline int wide_code_bytes(int m) { return WIDE_TILE + ((m + 15) & ~15) + 64; }

template <typename OUT_T, bool HITS>
__global__ __launch_bounds__(BLOCK) void k_wide_letters(const ScanArgs a)
{
    constexpr int W = WIDE_W;
    extern __shared__ __align__(16) unsigned char wsmem[];
    double *tbl = reinterpret_cast<double *>(wsmem);                      // [64][8]
    uint8_t *cbuf = wsmem + WIDE_SLAB * 64;
    const int m = a.m;
    const int64_t n_pos = a.n_pos;
    const int64_t tile0 = (int64_t)blockIdx.x * WIDE_TILE;
    const int nvec = (WIDE_TILE + ((m + 15) & ~15) + 64) / 16;
    for (int i = threadIdx.x; i < nvec; i += BLOCK) {
        const int64_t p = tile0 + 16 * (int64_t)i;
        u32x4 v = {0x07070707u, 0x07070707u, 0x07070707u, 0x07070707u};
        if (p + 16 <= n_pos) {
            v = *reinterpret_cast<const u32x4 *>(a.codes + p);
        } else if (p < n_pos) {
            uint32_t t[4] = {0x07070707u, 0x07070707u, 0x07070707u, 0x07070707u};
            for (int b = 0; b < 16; ++b)
                if (p + b < n_pos) t[b >> 2] = (t[b >> 2] & ~(0xFFu << (8 * (b & 3)))) | ((uint32_t)a.codes[p + b] << (8 * (b & 3)));
            v = u32x4{t[0], t[1], t[2], t[3]};
        }
        *reinterpret_cast<u32x4 *>(cbuf + 16 * i) = v;
    }
    double acc[W];
#pragma unroll
    for (int v = 0; v < W; ++v) acc[v] = 0.0;
    for (int j0 = 0; j0 < m; j0 += WIDE_SLAB) {
        __syncthreads();                               // the codes are staged / the previous slab is consumed
        const int rows = m - j0 < WIDE_SLAB ? m - j0 : WIDE_SLAB;
        for (int i = threadIdx.x; i < rows * 8; i += BLOCK) tbl[i] = a.letter_table[(size_t)j0 * 8 + i];
        __syncthreads();
        uint32_t w[18];
#pragma unroll
        for (int d = 0; d < 18; ++d)
            w[d] = (*reinterpret_cast<const uint32_t *>(cbuf + threadIdx.x * W + j0 + 4 * d) & 0x07070707u) << 3;   // byte = code * sizeof(double)
#pragma unroll
        for (int j = 0; j < WIDE_SLAB; ++j) {
            if (j < rows) {
                const char *row = reinterpret_cast<const char *>(tbl) + j * 64;
#pragma unroll
                for (int v = 0; v < W; ++v) {
                    const int q = j + v;
                    const uint32_t b = (w[q >> 2] >> ((q & 3) * 8)) & 0xFFu;
                    acc[v] += *reinterpret_cast<const double *>(row + b);
                }
            }
        }
    }
    const int64_t p0 = tile0 + (int64_t)threadIdx.x * W;
    if (HITS) {
        uint32_t mask = 0;
#pragma unroll
        for (int v = 0; v < W; ++v) {
            const double cmp = (sizeof(OUT_T) == 4 && !a.f64_hits) ? (double)(float)acc[v] : acc[v];
            if (p0 + v < n_pos && cmp > a.thr_seq) mask |= 1u << v;
        }
        emit_hits_block<W>(mask, [&](int i) { return p0 + i; }, [&](int i) { return (float)acc[i]; }, [&](int i) { return acc[i]; }, a);
        return;
    }
    if (sizeof(OUT_T) == 4) {
        float *o = a.out_seq + p0;
#pragma unroll
        for (int h = 0; h < W / 4; ++h) {
            if (p0 + 4 * h + 4 <= n_pos) {
                const f32x4 r = {(float)acc[4 * h], (float)acc[4 * h + 1], (float)acc[4 * h + 2], (float)acc[4 * h + 3]};
                __builtin_nontemporal_store(r, reinterpret_cast<f32x4 *>(o + 4 * h));
            } else {
                for (int v = 0; v < 4; ++v)
                    if (p0 + 4 * h + v < n_pos) o[4 * h + v] = (float)acc[4 * h + v];
            }
        }
    } else {
        double *o = a.out_letters_f64 + p0;
#pragma unroll
        for (int h = 0; h < W / 2; ++h) {
            if (p0 + 2 * h + 2 <= n_pos) {
                const f64x2 r = {acc[2 * h], acc[2 * h + 1]};
                __builtin_nontemporal_store(r, reinterpret_cast<f64x2 *>(o + 2 * h));
            } else if (p0 + 2 * h < n_pos) {
                o[2 * h] = acc[2 * h];
            }
        }
    }
}

// letters-only scans of a PFM wider than PFMSCAN_MAX_M (launch_wide in pfmscan_kernels.hip keeps the plain kernel for
// scans with a structure part).  false: not this kernel's case.
bool launch_wide_letters(const ScanArgs &a, hipStream_t stream, hipError_t *err)
{
    if (a.struct_pssm || !a.letter_table || !a.codes || a.m <= PFMSCAN_MAX_M) return false;
    if (!a.hits && !a.out_seq && !a.out_letters_f64) return false;
    const unsigned grid = (unsigned)((a.n_pos + WIDE_TILE - 1) / WIDE_TILE);
    const int lds = WIDE_SLAB * 64 + wide_code_bytes(a.m);
    if (a.hits) {
        if (a.f64_hits) hipLaunchKernelGGL((k_wide_letters<double, true>), dim3(grid), dim3(BLOCK), lds, stream, a);
        else hipLaunchKernelGGL((k_wide_letters<float, true>), dim3(grid), dim3(BLOCK), lds, stream, a);
    } else if (a.out_letters_f64) {
        hipLaunchKernelGGL((k_wide_letters<double, false>), dim3(grid), dim3(BLOCK), lds, stream, a);
    } else {
        hipLaunchKernelGGL((k_wide_letters<float, false>), dim3(grid), dim3(BLOCK), lds, stream, a);
    }
    *err = hipGetLastError();
    return true;
}

}  // namespace pfmscan

// Diagnostics (host only): the single-letter credit table k_letters_cred8 would use for one motif at threshold thr
// (include/pfmscan.h).  tests/test_library_credits.py checks it exhaustively without a GPU.
extern "C" int pfmscan_debug_credit8_table(const double *letter_table, int m, double thr, uint16_t *credits, int *mode)
{
    if (!letter_table || !credits || m < 1 || m > 32 || std::isnan(thr)) return PFMSCAN_E_BADARG;
    pfmscan::Cred8Cache cc;
    pfmscan::build_cred8(letter_table, m, thr, &cc);
    std::memcpy(credits, cc.cr, sizeof(uint16_t) * (size_t)m * 8);
    if (mode) *mode = cc.mode;
    return PFMSCAN_OK;
}
