// pfmscan_kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the sliding-window
// PFM scanner.  wave64, 256-thread workgroups, fp64 accumulation, no MFMA (the
// contraction is width x alphabet -- far below one MFMA tile); the design goal
// is HBM streaming: every input byte is read once with 16-byte-per-lane
// coalesced accesses, every output byte written once with 16-byte-per-lane stores.
//
// Kernels (the measurements behind each choice: profiles/r3/NOTES.md, profiles/r4/NOTES.md):
//
//  k_letters   codes only (config 2 and the letter-string structure scan).  A workgroup
//              parks its tile's codes in LDS once (16-byte vector loads); a thread owns
//              8 (float32 / hits) or 4 (fp64 output) consecutive windows per round, looks
//              each letter up in an LDS copy of the [m][8] log-odds table (ds_read_b64;
//              all lanes of an instruction hit the SAME table row -> at most 8 distinct
//              addresses on 16 distinct banks, no conflict) and adds sequentially in fp64.
//              float32 scores leave through a wave-private LDS transpose as 16-byte
//              stores, 1 KiB contiguous per wave-instruction.  All code reads first, all
//              stores last (vmcnt is one in-order queue for loads and stores).
//
//  k_letters_pre  hits over a 4-letter alphabet (the CLI's default path at -m 6, and the
//              first pass of the combined hits scan): an fp32 score from a table of
//              two-letter sums decides which windows can reach the threshold; only those
//              get the exact fp64 score.  Hits go to per-wave LDS queues, one returning
//              global atomic per workgroup flush.
//
//  k_profile   codes + averaged-structure profile (config 3, the headline).
//              A workgroup stages one tile of T = 256*V positions (+ m halo rows)
//              of the [n_pos][7] profile into LDS with LDS-DMA (global_load_lds,
//              1 KiB lane-linear pieces), then each thread scores V consecutive
//              windows.  V is ODD so the per-lane LDS row stride (7*V dwords) is
//              odd and the row reads are bank-conflict free on the linear image.
//              The window sum over j is SEQUENTIAL in one lane (fp64), exactly like
//              the reference loops, so float32 sequence scores are bit-exact; lanes
//              cooperate on loading, never on the sum.  The thread keeps a sliding
//              window of V profile rows (fp64) in registers; step j multiplies all
//              V rows by PSSM row j (uniform, s_load'ed, SGPR operand of v_fmac_f64)
//              and slides in ONE new row, so a row is converted fp32->fp64 once per
//              thread.  Scores leave through a wave-private LDS transpose as
//              16-byte nontemporal stores.
//
//  k_struct_at structure score at a list of candidate windows (second pass of the
//              candidate-then-verify combined hits scan): a fixed grid strides over the
//              sharded candidate lists, 4 profile rows per 7 vector loads.
//
// Reference semantics restated here (upstream paths, v0.10.2):
//   _pwm.c:34-68       score = 0.0 (double); score += M[j][col]; (float)score;
//                      NaN when any covered letter is foreign (NaN table column)
//   matrix.py:25-43    same sum, fp64 result, NaN on unknown letter
//   rnascan.py:302-307 score += nan_to_num(dot(profile[i+j,:], pssm[j,:]))
//   rnascan.py:263,310 hit <=> score > threshold (strict; NaN/-inf never pass)
//   rnascan.py:422-423 combined hit <=> both tables hold (id, start, end)
#include <float.h>
#include <math.h>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <mutex>
#include "pfmscan_device.hpp"
#include "pfmscan_profile.hpp"

namespace pfmscan {


// ---------------------------------------------------------------------------
// k_letters: letter table only.  NDW = dwords of codes a thread may need:
// bytes 0 .. m+2 relative to its first window -> NDW = 5 / 9 / 17 for
// m <= 16 / 32 / 64.  OUT_T = float (_pwm.c) or double (matrix.py:25-43).
// ---------------------------------------------------------------------------
template <int NDW, typename OUT_T, bool HITS, int W>
__global__ __launch_bounds__(BLOCK) void k_letters(const ScanArgs a)
{
    // W consecutive windows per thread and round (4: one 16-byte store per round; 8: fewer code
    // loads and byte extractions per window, used where nothing is stored), 16 windows per thread
    constexpr int ROUNDS = (let_iters(NDW) * 4) / W;
    constexpr int LET_TILE = let_tile(NDW);
    constexpr int NW = NDW + (W - 4) / 4;              // code dwords per round: bytes 0 .. W + m - 2
    __shared__ __align__(16) double tbl[PFMSCAN_MAX_M * 8];
    __shared__ __align__(16) uint8_t cbuf[LET_TILE + CODE_HALO];
    const int m = a.m;
    const int64_t n_pos = a.n_pos;
    const int64_t tile0 = (int64_t)blockIdx.x * LET_TILE;
    CodeStage<LET_TILE> cs;
    if (!(a.ablate & 2)) cs.fetch(a.codes, tile0, n_pos);
    for (int i = threadIdx.x; i < m * 8; i += BLOCK) tbl[i] = a.letter_table[i];
    if (!(a.ablate & 2)) cs.park(cbuf);
    __syncthreads();

    OUT_T *__restrict__ out = reinterpret_cast<OUT_T *>(sizeof(OUT_T) == 4 ? (void *)a.out_seq : (void *)a.out_letters_f64);

    // Three phases per workgroup, so that no wave ever waits for its own stores: (1) every
    // code load of the workgroup's windows is issued up front, (2) all 16 windows per thread are
    // scored into registers, (3) the stores go out last and drain after the wave has retired.
    // (vmcnt counts loads and stores in one in-order queue on gfx950: interleaving
    // load -> score -> store per 1024 windows made every load wait behind the previous stores,
    // 0.47 ms on C2 = the SUM of the load-bound and the write-bound time instead of their max.)
    uint32_t wall[ROUNDS][NW];
#pragma unroll
    for (int it = 0; it < ROUNDS; ++it) {
        const int64_t p0 = tile0 + (int64_t)it * (BLOCK * W) + (int64_t)threadIdx.x * W;
#pragma unroll
        for (int d = 0; d < NW; ++d) {
            const uint32_t x = *reinterpret_cast<const uint32_t *>(cbuf + (p0 - tile0) + 4 * d);   // inside the halo
            wall[it][d] = (x & 0x07070707u) << 3;      // byte = code * sizeof(double)
        }
    }
    double res[ROUNDS][W];
#pragma unroll
    for (int it = 0; it < ROUNDS; ++it) {
        const uint32_t (&w)[NW] = wall[it];
        double acc[W];
#pragma unroll
        for (int v = 0; v < W; ++v) acc[v] = 0.0;
        // one guarded group per motif position: a guard, not a break (a constant trip count is what
        // lets hipcc unroll this), and no wider groups (2 or 4 positions per guard keep more look-ups
        // in flight but cost 2-3 waves of occupancy: 15-50 % slower at every width)
        if constexpr (NDW == 5) {
            // PFMs up to 16 wide: the table offset of every position the round touches is extracted ONCE (W + 15 values) instead
            // of once per (row, window) -- the same byte serves up to W windows at W different rows; the rows are immediate
            // offsets of the ds_read (C2 at w = 8 / 16: 0.278 -> 0.273 / 0.413 -> 0.404 ms, four interleaved pairs).  Wider
            // buckets would hold 39 / 71 such registers and lose the occupancy they live on.
            uint32_t adr[W + (NDW - 1) * 4 - 1];
#pragma unroll
            for (int q = 0; q < W + (NDW - 1) * 4 - 1; ++q) adr[q] = (w[q >> 2] >> ((q & 3) * 8)) & 0xFFu;
#pragma unroll
            for (int j = 0; j < (NDW - 1) * 4; ++j) {
                if (j < m) {
                    // (volatile: one ds_read_b64 per look-up; paired into ds_read2_b64 they run at half the LDS rate, see k_letters_fixed)
                    const __attribute__((address_space(3))) char *row = (const __attribute__((address_space(3))) char *)reinterpret_cast<const char *>(tbl) + j * 64;
#pragma unroll
                    for (int v = 0; v < W; ++v) acc[v] += *(const volatile __attribute__((address_space(3))) double *)(row + adr[j + v]);
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < (NDW - 1) * 4; ++j) {
                if (j < m) {
                    const __attribute__((address_space(3))) char *row = (const __attribute__((address_space(3))) char *)reinterpret_cast<const char *>(tbl) + j * 64;
#pragma unroll
                    for (int v = 0; v < W; ++v) {
                        const int q = j + v;    // byte index relative to p0, compile-time
                        const uint32_t b = (w[q >> 2] >> ((q & 3) * 8)) & 0xFFu;
                        acc[v] += *(const volatile __attribute__((address_space(3))) double *)(row + b);
                    }
                }
            }
        }
#pragma unroll
        for (int v = 0; v < W; ++v) res[it][v] = acc[v];
    }
    if (HITS) {
        uint32_t mask = 0;
        const int64_t pbase = tile0 + (int64_t)threadIdx.x * W;
#pragma unroll
        for (int it = 0; it < ROUNDS; ++it) {
#pragma unroll
            for (int v = 0; v < W; ++v) {
                const double cmp = sizeof(OUT_T) == 4 ? (double)(float)res[it][v] : res[it][v];
                if ((pbase + it * (BLOCK * W) + v < n_pos) && (cmp > a.thr_seq)) mask |= 1u << (W * it + v);
            }
        }
        if (a.ablate & 8) {                 // timing diagnostic: no hit emission at all
            if (mask == 0xdeadbeefu) a.hit_pos[0] = pbase;
            return;
        }
        emit_hits_block<ROUNDS * W>(
            mask, [&](int i) { return pbase + (int64_t)(i / W) * (BLOCK * W) + (i % W); },
            [&](int i) { return (float)res[i / W][i % W]; }, [&](int i) { return res[i / W][i % W]; }, a);
        return;
    }
    if (W == 8 && sizeof(OUT_T) == 4 && !(a.ablate & 4)) {
        // 8 windows per thread: transpose through a wave-private LDS strip so that every store
        // instruction still writes 1 KiB contiguous (LDS operations of one wave execute in order)
        __shared__ __align__(16) float strip[BLOCK / 64][64 * 8];
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        float *mine = strip[wave];
#pragma unroll
        for (int it = 0; it < ROUNDS; ++it) {
            f32x4 lo = {(float)res[it][0], (float)res[it][1], (float)res[it][2], (float)res[it][3]};
            f32x4 hi = {(float)res[it][4], (float)res[it][5], (float)res[it][6], (float)res[it][7]};
            reinterpret_cast<f32x4 *>(mine)[2 * lane] = lo;
            reinterpret_cast<f32x4 *>(mine)[2 * lane + 1] = hi;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int64_t w0 = tile0 + (int64_t)it * (BLOCK * 8) + (int64_t)wave * 512;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int c = lane + 64 * k;
                const int64_t p = w0 + 4 * (int64_t)c;
                float *o = reinterpret_cast<float *>(out) + p;
                if (p + 4 <= n_pos) {
                    __builtin_nontemporal_store(reinterpret_cast<const f32x4 *>(mine)[c], reinterpret_cast<f32x4 *>(o));
                } else {
                    for (int e = 0; e < 4; ++e)
                        if (p + e < n_pos) o[e] = mine[4 * c + e];
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        return;
    }
#pragma unroll
    for (int it = 0; it < ROUNDS; ++it) {
#pragma unroll
        for (int h = 0; h < W / 4; ++h) {
            const int64_t p0 = tile0 + (int64_t)it * (BLOCK * W) + (int64_t)threadIdx.x * W + 4 * h;
            const double *r4 = &res[it][4 * h];
            if (a.ablate & 4) {
                if (r4[0] + r4[1] + r4[2] + r4[3] == 1.2345e300) out[0] = (OUT_T)r4[0];
            } else if (sizeof(OUT_T) == 4) {
                float *o = reinterpret_cast<float *>(out) + p0;
                if (p0 + 4 <= n_pos) {
                    f32x4 r = {(float)r4[0], (float)r4[1], (float)r4[2], (float)r4[3]};
                    __builtin_nontemporal_store(r, reinterpret_cast<f32x4 *>(o));
                } else {
                    for (int v = 0; v < 4; ++v)
                        if (p0 + v < n_pos) o[v] = (float)r4[v];
                }
            } else {
                double *o = reinterpret_cast<double *>(out) + p0;
                if (p0 + 4 <= n_pos) {
                    f64x2 r0 = {r4[0], r4[1]}, r1 = {r4[2], r4[3]};
                    __builtin_nontemporal_store(r0, reinterpret_cast<f64x2 *>(o));
                    __builtin_nontemporal_store(r1, reinterpret_cast<f64x2 *>(o + 2));
                } else {
                    for (int v = 0; v < 4; ++v)
                        if (p0 + v < n_pos) o[v] = r4[v];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------
// k_letters_pre -- hits mode over a 4-letter alphabet with an fp32 prefilter.
// The exact score needs m dependent fp64 adds and m 8-byte LDS look-ups per window, and k_letters
// is bound by exactly those (it reads 1 byte per window).  A hit only has to be EXACT once we
// know it may be one: here every window first gets an approximate score from a table of
// two-letter sums (fp32, [ceil(m/2)][16], 64-byte rows -> conflict-free ds_read_b32): half the
// look-ups, half-width adds.  |approx - exact| <= pair_eps (bound computed on the host from the
// table), so "approx > thr - pair_eps" loses no hit; the survivors are re-scored with the
// sequential fp64 sum of _pwm.c:34-68 from the codes still in registers, and only that exact
// float32 score is compared with the threshold and reported.  Foreign letters are invisible to
// the prefilter (it looks at 2 bits per code) and rejected by the exact pass (NaN).
//
// Hits are rare and found in divergent code, so they are not scanned into place: a hit lane takes
// a slot of its wave's LDS queue with an LDS atomic.  A workgroup walks a.tiles_per_block tiles
// (codes double-buffered in LDS, ONE barrier per tile) and at a tile boundary flushes the four
// queues with ONE returning global atomic once one is half full, and at the end -- one atomic per
// 4096 windows serialised on the counter word (73k atomics = 0.45 ms on C2, more than the
// scoring).  A wave whose queue cannot take the survivors of the next 64 windows flushes alone
// (dense thresholds only).  Hits land in no particular order; the host sorts.
// ---------------------------------------------------------------------------

template <int NDW>
__global__ __launch_bounds__(BLOCK) void k_letters_pre(const ScanArgs a)
{
    constexpr int W = 8;
    constexpr int ROUNDS = (let_iters(NDW) * 4) / W;
    constexpr int LET_TILE = let_tile(NDW);
    constexpr int NW = NDW + 1;                        // code dwords per round: bytes 0 .. W + m - 1 (+1 for the pair)
    constexpr int NWAVE = BLOCK / 64;
    constexpr int MAXM = (NDW - 1) * 4;                // widest PFM of this instantiation (the launcher picks NDW from m)
    __shared__ __align__(16) double tbl[MAXM * 8];     // sized by the instantiation, not by PFMSCAN_MAX_M: m <= 16 fits 8 workgroups per CU
    __shared__ __align__(16) float ptab[(MAXM / 2) * 16];
    __shared__ __align__(16) uint8_t cbuf[2][LET_TILE + CODE_HALO];
    __shared__ uint32_t q_pos[NWAVE][WQ_CAP];          // relative to the workgroup's first tile (4 bytes: 22.7 instead of 26.8 KB of LDS)
    __shared__ float q_sc[NWAVE][WQ_CAP];
    __shared__ int q_n[NWAVE], snap[2][NWAVE];
    __shared__ unsigned long long s_base;
    const int m = a.m;
    const int npair = (m + 1) >> 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t n_pos = a.n_pos;
    const int ntile = a.tiles_per_block;
    const int64_t first = (int64_t)blockIdx.x * ntile * LET_TILE;
    if (first >= n_pos) return;                        // whole workgroup

    CodeStage<LET_TILE> cs;
    cs.fetch(a.codes, first, n_pos);
    // rows m .. roundup4(m)-1 are zeros: x + 0.0 == x for every x a sum that started at +0.0 can hold
    // (never -0.0), so the exact pass runs whole groups of 4 motif positions
    for (int i = threadIdx.x; i < ((m + 3) & ~3) * 8; i += BLOCK) tbl[i] = i < m * 8 ? a.letter_table[i] : 0.0;
    for (int i = threadIdx.x; i < npair * 16; i += BLOCK) ptab[i] = a.pair_table[i];
    if (threadIdx.x < NWAVE) q_n[threadIdx.x] = 0;
    cs.park(cbuf[0]);
    if (ntile > 1 && first + LET_TILE < n_pos) cs.fetch(a.codes, first + LET_TILE, n_pos);
    __syncthreads();

    const float thr_pre = a.thr_pre;                   // largest float <= thr_seq - pair_eps
    const char *pbytes = (const char *)ptab;
    const int shard = blockIdx.x & (a.hit_shards - 1);
    const unsigned long long shard_off = (unsigned long long)shard * (unsigned long long)a.capacity;
    unsigned long long *counter = a.hit_count + shard * HIT_COUNTER_STRIDE;
    uint32_t *my_pos = q_pos[wave];
    float *my_sc = q_sc[wave];

    auto store_hit = [&](unsigned long long slot, int64_t pos, float sc) {
        if ((int64_t)slot < a.capacity) {             // capacity is per shard
            a.hit_pos[shard_off + slot] = pos + a.pos_offset;
            if (a.hit_seq) a.hit_seq[shard_off + slot] = sc;
            if (a.hit_struct) a.hit_struct[shard_off + slot] = (double)sc;
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
    for (int tb = 0; tb < ntile; ++tb) {
        const int64_t tile0 = first + (int64_t)tb * LET_TILE;
        if (tile0 >= n_pos) break;                     // uniform; the previous tile flushed (it was the last)
        const uint8_t *cb = cbuf[tb & 1];

#pragma unroll
        for (int it = 0; it < ROUNDS; ++it) {
            const int off0 = it * (BLOCK * W) + threadIdx.x * W;
            uint32_t w[NW + 1];
#pragma unroll
            for (int d = 0; d < NW + 1; ++d) w[d] = *reinterpret_cast<const uint32_t *>(cb + off0 + 4 * d);   // inside the halo
            // z[d] byte k = 4 * (c[4d+k] | c[4d+k+1] << 2): byte offset in a pair-table row of the pair at byte 4d+k
            uint32_t z[NW];
#pragma unroll
            for (int d = 0; d < NW; ++d) {
                const uint32_t x0 = (w[d] & 0x03030303u) << 2, x1 = (w[d + 1] & 0x03030303u) << 2;
                z[d] = x0 | __builtin_amdgcn_alignbit(x1, x0, 6);
            }
            float acc[W];
#pragma unroll
            for (int v = 0; v < W; ++v) acc[v] = 0.f;
#pragma unroll
            for (int t = 0; t < (NDW - 1) * 2; ++t) {   // pairs: motif positions 2t, 2t+1
                if (t < npair)
#pragma unroll
                for (int v = 0; v < W; ++v) {
                    const int q = v + 2 * t;            // byte index of the pair's first letter, compile-time
                    const uint32_t off = (z[q >> 2] >> ((q & 3) * 8)) & 0xFFu;
                    acc[v] += *(const float *)(pbytes + off + t * 64);
                }
            }
            // Survivors: k_letters' exact score.  A wave enters a window's branch only when one of
            // its lanes survived there, so the cost stays bounded by k_letters' at any threshold.
            // (Windows past the end hold SEP codes: NaN below.)
#pragma unroll
            for (int v = 0; v < W; ++v) {
                const bool sv = acc[v] > thr_pre;
                const unsigned long long sb = __ballot(sv);
                if (sb) {                               // wave-uniform
                    const int ns = __popcll(sb);
                    if (qn_ub + ns > WQ_CAP) {
                        wave_flush();
                        qn_ub = 0;
                    }
                    qn_ub += ns;
                    if (sv) {
                        double sc = 0.0;
#pragma unroll
                        for (int j0 = 0; j0 < (NDW - 1) * 4; j0 += 4) {
                            if (j0 < m)
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                const int q = v + j0 + u;
                                const uint32_t c = (w[q >> 2] >> ((q & 3) * 8)) & 7u;
                                sc += tbl[(j0 + u) * 8 + c];
                            }
                        }
                        const float f = (float)sc;
                        if ((double)f > a.thr_seq) {
                            const int slot = atomicAdd(&q_n[wave], 1);     // LDS
                            my_pos[slot] = (uint32_t)(tile0 - first) + (uint32_t)(off0 + v);
                            my_sc[slot] = f;
                        }
                    }
                }
            }
        }

        // tile boundary: publish the next tile's codes and this wave's queue length, ONE barrier
        const bool more = tb + 1 < ntile && tile0 + LET_TILE < n_pos;
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
        if (most >= WQ_CAP / 2 || (!more && total > 0)) {          // uniform: every thread read the same snapshot
            if (threadIdx.x == 0) s_base = atomicAdd(counter, (unsigned long long)total);
            __syncthreads();
            drain(s_base + (unsigned long long)before, nq[wave]);
            qn_ub = 0;
        }
    }
}

// ---------------------------------------------------------------------------
// k_letters_cred -- k_letters_pre with the INTEGER prefilter of the library kernel, keyed by stream position.
// (PFMs up to width 32 over a 4-letter alphabet with a finite threshold)
//
// k_letters_pre spends its time on ceil(m/2) four-byte look-ups and as many fp32 adds per window, plus the index
// arithmetic of each.  Here the two-letter table is turned around: ONE entry per letter pair holds that pair's
// credit for EVERY pair row, two rows per dword (unsigned 16-bit fixed point, rounded so that a hit can never be
// dropped: pfmscan_library_api.hip, build_credits):
//     ctab[pair of the letters at positions q, q+1] = { (row0 | row1 << 16), (row2 | row3 << 16), ... }      NJ dwords
// A lane owns W = 16 consecutive windows and reads ctab ONCE per position of its band (W + 4 NJ - 2 reads of 4 NJ
// bytes, ~1.4 per window) instead of ceil(m/2) times per window.  Row 2j of position q belongs to window q - 4j,
// row 2j+1 to window q - 4j - 2, so with the packs  P[w] = (credits of window w+2 | credits of window w) << 16
//     P[w] = d0(w+2) + d1(w+6) + d2(w+10) + d3(w+14)          one v_add3_u32 (+ one add) per TWO windows' halves
//     sum(w) = (P[w] >> 16) + (P[w-2] & 0xffff)               bit 15 set <=> the window may be a hit
// (the threshold is folded into row 0).  Survivors are NOT re-scored in place (a wave pass per surviving window with one or
// two live lanes was 45 % of the kernel at -m 6): their positions go to a wave-private LDS queue that lives across tiles,
// and 64 at a time they get the exact score -- letters re-read from global memory (L2), sequential fp64 sum, float32
// cast, strict compare -- one survivor per lane.  Hits: k_letters_pre's LDS hit queues, one returning atomic per flush.
// ---------------------------------------------------------------------------
struct CredTable {
    uint32_t d[16][8];                                // [letter pair c0 | c1 << 2][row pair j]
};
template <int NJ> struct CredEntry { typedef u32x4 type; };          // 3 or 4 row pairs: 16-byte entries; 5..8: two of them
template <> struct CredEntry<2> { typedef u32x2 type; };
template <> struct CredEntry<1> { typedef uint32_t type; };

// PAIR: the two-FASTA combined scan in ONE launch (rnascan.py:119-123, :416-434).  A survivor whose exact float32 sequence
// score passes is scored right away on the SECOND code stream (a.codes2, a.letter_table2: the structure letters of the same
// positions, fp64 sum as matrix.py:25-43) and is a hit only when that exceeds a.thr_struct too -- no candidate list, no
// count read-back, no second launch.  The hit queue carries both scores (half as many slots: combined hits are rarer).
template <int NJ, bool PAIR>
__global__ __launch_bounds__(BLOCK) void k_letters_cred(const ScanArgs a, const CredTable ct)
{
    constexpr int W = 16;                              // windows per lane = one round per tile
    constexpr int LET_TILE = BLOCK * W;
    constexpr int NPOS = W + 4 * NJ - 2;               // positions a lane looks up: q = 0 .. W + 4 NJ - 3
    constexpr int NWD = (NPOS + 1 + 3) / 4;            // code dwords holding bytes 0 .. NPOS (the pair at q needs byte q + 1)
    // entry STRIDE in bytes: 4 / 8 / 16 for up to 4 row pairs (16 entries span at most 256 B = every bank once); 5..8 row pairs
    // need 32 bytes, 16 of them = the banks TWICE (entries e and e + 8 collide 2-way) -- stride 48 spreads the 16-byte pieces
    // of the 16 entries over 16 different bank quads again (12 e mod 64 is a permutation of the multiples of 4)
    constexpr int ESTR = NJ == 1 ? 4 : (NJ == 2 ? 8 : (NJ <= 4 ? 16 : 48));
    constexpr int TROWS = NJ <= 4 ? 16 : 32;           // rows of the exact letter table (rows m .. are zeros)
    constexpr int NWAVE = BLOCK / 64;
    typedef typename CredEntry<NJ>::type entry_t;
    __shared__ __align__(16) double tbl[TROWS * 8];
    __shared__ __align__(16) uint32_t ctab[16 * (ESTR / 4)];
    __shared__ __align__(16) uint8_t cbuf[2][LET_TILE + CODE_HALO];
    constexpr int QCAP = PAIR ? WQ_CAP / 2 : WQ_CAP;
    __shared__ __align__(16) double tbl2[PAIR ? TROWS * 8 : 1];      // the second stream's letter table (rows m .. are zeros)
    __shared__ __align__(8) double q_st[PAIR ? NWAVE : 1][PAIR ? QCAP : 1];
    __shared__ uint32_t q_pos[NWAVE][QCAP];            // positions in both queues are relative to the workgroup's first tile:
    __shared__ float q_sc[NWAVE][QCAP];
    __shared__ int q_n[NWAVE], snap[2][NWAVE];
    __shared__ unsigned long long s_base;
    __shared__ uint32_t sv_pos[NWAVE][128];            // 4-byte entries keep the workgroup under 20 KB of LDS = 8 per CU (26 KB: 6);
                                                       // sv_pos: survivors of the prefilter waiting for their exact score
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
    if (PAIR)
        for (int i = threadIdx.x; i < TROWS * 8; i += BLOCK) tbl2[i] = i < m * 8 ? a.letter_table2[i] : 0.0;
    for (int i = threadIdx.x; i < 16 * (ESTR / 4); i += BLOCK) ctab[i] = (i % (ESTR / 4)) < 8 ? ct.d[i / (ESTR / 4)][i % (ESTR / 4)] : 0u;
    if (threadIdx.x < NWAVE) q_n[threadIdx.x] = 0;
    cs.park(cbuf[0]);
    if (ntile > 1 && first + LET_TILE < n_pos) cs.fetch(a.codes, first + LET_TILE, n_pos);
    __syncthreads();

    const char *cbytes = (const char *)ctab;
    const int shard = blockIdx.x & (a.hit_shards - 1);
    const unsigned long long shard_off = (unsigned long long)shard * (unsigned long long)a.capacity;
    unsigned long long *counter = a.hit_count + shard * HIT_COUNTER_STRIDE;
    uint32_t *my_pos = q_pos[wave];
    float *my_sc = q_sc[wave];
    double *my_st = q_st[PAIR ? wave : 0];

    auto store_hit = [&](unsigned long long slot, int64_t pos, float sc, double st) {
        if ((int64_t)slot < a.capacity) {             // capacity is per shard
            a.hit_pos[shard_off + slot] = pos + a.pos_offset;
            if (a.hit_seq) a.hit_seq[shard_off + slot] = sc;
            if (a.hit_struct) a.hit_struct[shard_off + slot] = PAIR ? st : (double)sc;
        }
    };
    auto drain = [&](unsigned long long base, int n) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int i = lane; i < n; i += 64) store_hit(base + i, first + (int64_t)my_pos[i], my_sc[i], PAIR ? my_st[i] : 0.0);
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
    // exact score of survivors [at, at + cnt) of this wave's queue, one per lane (_pwm.c:34-68)
    auto exact_batch = [&](int at, int cnt) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (qn_ub + cnt > QCAP) {                      // room for a hit per lane in the hit queue
            wave_flush();
            qn_ub = 0;
        }
        qn_ub += cnt;
        if (lane < cnt) {
            const int64_t p = first + (int64_t)my_sv[at + lane];
            const int64_t al = p & ~(int64_t)3;
            uint32_t raw[NJ + 1];
#pragma unroll
            for (int k = 0; k < NJ + 1; ++k) raw[k] = load_codes4(a.codes, al + 4 * k, n_pos);
            double sc = 0.0;
#pragma unroll
            for (int k = 0; k < NJ; ++k) {
                const uint32_t cw = __builtin_amdgcn_alignbyte(raw[k + 1], raw[k], (uint32_t)(p & 3));
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int j = 4 * k + b;
                    if (j < 4 * NJ) sc += tbl[j * 8 + ((cw >> (8 * b)) & 7u)];      // rows m .. 4 NJ - 1 are zeros
                }
            }
            const float f = (float)sc;
            bool ok = (double)f > a.thr_seq;
            double st = 0.0;
            if (PAIR && ok) {                          // the structure letters of the same window: fp64 sum, fp64 compare
                uint32_t raw2[NJ + 1];
#pragma unroll
                for (int k = 0; k < NJ + 1; ++k) raw2[k] = load_codes4(a.codes2, al + 4 * k, n_pos);
#pragma unroll
                for (int k = 0; k < NJ; ++k) {
                    const uint32_t cw = __builtin_amdgcn_alignbyte(raw2[k + 1], raw2[k], (uint32_t)(p & 3));
#pragma unroll
                    for (int b = 0; b < 4; ++b) st += tbl2[(4 * k + b) * 8 + ((cw >> (8 * b)) & 7u)];      // rows m .. 4 NJ - 1 are zeros
                }
                ok = st > a.thr_struct;
            }
            if (ok) {
                const int slot = atomicAdd(&q_n[wave], 1);     // LDS
                my_pos[slot] = (uint32_t)(p - first);
                my_sc[slot] = f;
                if (PAIR) my_st[slot] = st;
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
        // xm[d] = the low two bits of the four codes at bytes 4d .. 4d+3.  The entry offset of the pair at byte q,
        // (c[q] | c[q+1] << 2) x ESTR, is ONE v_dot4_u32_u8 of that dword with the weights (1, 4) x ESTR placed on
        // bytes q & 3 and (q & 3) + 1 -- plus a v_alignbyte when the pair straddles two dwords.  (Shift / or / mask per
        // dword and a byte extraction + shift per position were a third of the kernel's VALU instructions.)
        uint32_t xm[NWD + 1];
#pragma unroll
        for (int d = 0; d < NWD + 1; ++d) xm[d] = *reinterpret_cast<const uint32_t *>(cb + off0 + 4 * d) & 0x03030303u;   // inside the halo
        constexpr uint32_t WT = (uint32_t)ESTR | (uint32_t)(4 * ESTR) << 8;       // weights of a pair's two letters (4 x 48 = 192 fits a byte)
        // one look-up per position; P[w + 2] in the text above is pk[w + 2] here (w = -2 .. W-1)
        uint32_t pk[W + 2];
#pragma unroll
        for (int i = 0; i < W + 2; ++i) pk[i] = 0u;
        // q and j are constant expressions (static_for, pfmscan_device.hpp): every pk[] / dj[] index is a register name
        static_for<0, NPOS>([&](auto qc) __attribute__((always_inline)) {
            constexpr int q = decltype(qc)::value;
            const uint32_t off = (q & 3) == 3 ? __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(xm[(q >> 2) + 1], xm[q >> 2], 3u), WT, 0u, false)
                                              : __builtin_amdgcn_udot4(xm[q >> 2], WT << (8 * (q & 3)), 0u, false);
            uint32_t dj[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
            if constexpr (NJ == 1) {
                dj[0] = *reinterpret_cast<const entry_t *>(cbytes + off);
            } else if constexpr (NJ <= 4) {
                const entry_t e = *reinterpret_cast<const entry_t *>(cbytes + off);
                static_for<0, NJ>([&](auto jc) __attribute__((always_inline)) { dj[decltype(jc)::value] = e[decltype(jc)::value]; });
            } else {                                   // 5..8 row pairs: a 32-byte entry
                const u32x4 e0 = *reinterpret_cast<const u32x4 *>(cbytes + off), e1 = *reinterpret_cast<const u32x4 *>(cbytes + off + 16);
                static_for<0, NJ>([&](auto jc) __attribute__((always_inline)) {
                    constexpr int j = decltype(jc)::value;
                    dj[j] = j < 4 ? e0[j & 3] : e1[j & 3];
                });
            }
            static_for<0, NJ>([&](auto jc) __attribute__((always_inline)) {
                constexpr int j = decltype(jc)::value;
                constexpr int wi = q - 2 - 4 * j;      // pack of the hi window w = q - 4j - 2
                if constexpr (wi >= -2 && wi <= W - 1) pk[wi + 2] += dj[j];
            });
        });
        // The 16 sums two at a time: with A = P[w], B = P[w-2], C = P[w+2] one v_alignbit + one v_pk_add_u16 give
        // (sum(w) << 16 | sum(w+2)); bit 15 of a sum (they stay below 2^16) is its flag.  `surv` collects the flags of the
        // pair k = 0..7 -- windows 4 (k >> 1) + (k & 1) and that + 2 -- at bits 24 + k and 8 + k: two VALU instructions
        // per window in all, hits or not.
        uint32_t surv = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int w = 4 * (k >> 1) + (k & 1);
            const uint32_t x = __builtin_amdgcn_alignbit(pk[w], pk[w + 4], 16);       // lo(P[w-2]) << 16 | hi(P[w+2])
            const u16x2 r = __builtin_bit_cast(u16x2, pk[w + 2]) + __builtin_bit_cast(u16x2, x);
            surv = (surv >> 1) | (__builtin_bit_cast(uint32_t, r) & 0x80008000u);
        }
        // Survivors -> the wave's queue (positions only; windows past the end hold SEP codes and score NaN later).  ONE
        // rolled loop, every pass each lane that still has a survivor hands over its lowest one: at -m 6 a sixth of the
        // (wave, window slot) pairs holds a survivor, and the unrolled pass per slot (test, ballot, branch, push) cost
        // 3.3 of the kernel's 12.3 VALU instructions per window; this form 1.1 + the two above
        while (__builtin_amdgcn_ballot_w64(surv != 0)) {
            const bool sv = surv != 0;
            const unsigned long long sb = __builtin_amdgcn_ballot_w64(sv);
            if (sv) {
                const int b = __builtin_ctz(surv);
                surv &= surv - 1;
                const int k = b & 7, v = 4 * (k >> 1) + (k & 1) + (b < 16 ? 2 : 0);
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
        if (most >= QCAP / 2 || (!more && total > 0)) {            // uniform: every thread read the same snapshot
            if (threadIdx.x == 0) s_base = atomicAdd(counter, (unsigned long long)total);
            __syncthreads();
            drain(s_base + (unsigned long long)before, nq[wave]);
            qn_ub = 0;
        }
    }
}

// ---------------------------------------------------------------------------
// k_letters_quad -- k_letters_cred with FOUR-letter credit tables (PFMs up to width 32, 4-letter alphabet, finite threshold).
//
// k_letters_cred is VALU-issue bound (14 instructions per window at w = 8: one table look-up per POSITION, but every
// look-up brings ceil(m/2)/2 dwords to add, and every position's index costs three bit operations).  Here a table
// entry covers four motif positions: entry[idx] for the 4-mer idx = c0 | c1 << 2 | c2 << 4 | c3 << 6 holds the credit of
// EVERY quad row t (motif positions 4t .. 4t+3), two rows per dword, ceil(m/4) rows in all (host: quad_sums +
// build_credits, same one-sided rounding, threshold folded into row 0, "may be a hit" = bit 15 of the sum):
//   * half the rows: w = 8 has TWO, one dword per entry -- sum(w) = lo(e[w]) + hi(e[w + 4]) is ONE v_add (SDWA), nothing
//     to accumulate; wider PFMs add one dword per 8 motif positions (PK[w] = sum_s dword_s(e[w + 8 s + 4]),
//     sum(w) = hi(PK[w]) + lo(PK[w - 4]));
//   * the index of the 4-mer at byte position q is ONE v_dot4_u32_u8 of the (pre-scaled) code bytes with the weights
//     1, 4, 16, 64 -- plus one v_alignbyte when q is not dword aligned -- and arrives already multiplied by the entry size;
//   * fewer rows also mean finer credits: V = 32767 / (rows - 1) levels, the prefilter keeps next to nothing but hits.
// The table is 256 entries of 4 / 8 / 16 bytes (1-4 KB of LDS, look-ups lane-random over all banks).  Survivors and hits:
// as in k_letters_cred (wave-private LDS queues, exact fp64 re-score 64 at a time, one returning atomic per flush).
// ---------------------------------------------------------------------------
template <int NQ> struct QuadEntry { typedef u32x4 type; };          // 5..8 quad rows: 16-byte entries
template <> struct QuadEntry<4> { typedef u32x2 type; };
template <> struct QuadEntry<3> { typedef u32x2 type; };
template <> struct QuadEntry<2> { typedef uint32_t type; };
template <> struct QuadEntry<1> { typedef uint32_t type; };

template <int NQ>
__global__ __launch_bounds__(BLOCK) void k_letters_quad(const ScanArgs a)
{
    constexpr int W = 16;                              // windows per lane = one round per tile
    constexpr int LET_TILE = BLOCK * W;
    constexpr int ND = (NQ + 1) / 2;                   // dwords of an entry that carry credits
    constexpr int ESH = NQ <= 2 ? 2 : (NQ <= 4 ? 3 : 4);   // log2 of the entry size in bytes
    constexpr int NPOS = W + 4 * (NQ - 1);             // positions a lane looks up: q = 0 .. W + 4 NQ - 5
    constexpr int NWD = (NPOS + 3 + 3) / 4;            // code dwords holding bytes 0 .. NPOS + 2 (the 4-mer at q ends at q + 3)
    constexpr int TROWS = NQ <= 4 ? 16 : 32;           // rows of the exact letter table (rows m .. are zeros)
    constexpr int NWAVE = BLOCK / 64;
    typedef typename QuadEntry<NQ>::type entry_t;
    __shared__ __align__(16) double tbl[TROWS * 8];
    __shared__ __align__(16) uint32_t qtab[256 << (ESH - 2)];
    __shared__ __align__(16) uint8_t cbuf[2][LET_TILE + CODE_HALO];
    __shared__ int64_t q_pos[NWAVE][WQ_CAP];
    __shared__ float q_sc[NWAVE][WQ_CAP];
    __shared__ int q_n[NWAVE], snap[2][NWAVE];
    __shared__ unsigned long long s_base;
    __shared__ int64_t sv_pos[NWAVE][128];             // survivors of the prefilter waiting for their exact score
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
    for (int i = threadIdx.x; i < (256 << (ESH - 2)); i += BLOCK) qtab[i] = a.d_quad[i];
    if (threadIdx.x < NWAVE) q_n[threadIdx.x] = 0;
    cs.park(cbuf[0]);
    if (ntile > 1 && first + LET_TILE < n_pos) cs.fetch(a.codes, first + LET_TILE, n_pos);
    __syncthreads();

    const char *qbytes = (const char *)qtab;
    const int shard = blockIdx.x & (a.hit_shards - 1);
    const unsigned long long shard_off = (unsigned long long)shard * (unsigned long long)a.capacity;
    unsigned long long *counter = a.hit_count + shard * HIT_COUNTER_STRIDE;
    int64_t *my_pos = q_pos[wave];
    float *my_sc = q_sc[wave];

    auto store_hit = [&](unsigned long long slot, int64_t pos, float sc) {
        if ((int64_t)slot < a.capacity) {             // capacity is per shard
            a.hit_pos[shard_off + slot] = pos + a.pos_offset;
            if (a.hit_seq) a.hit_seq[shard_off + slot] = sc;
            if (a.hit_struct) a.hit_struct[shard_off + slot] = (double)sc;
        }
    };
    auto drain = [&](unsigned long long base, int n) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int i = lane; i < n; i += 64) store_hit(base + i, my_pos[i], my_sc[i]);
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
    int64_t *my_sv = sv_pos[wave];
    // exact score of survivors [at, at + cnt) of this wave's queue, one per lane (_pwm.c:34-68)
    auto exact_batch = [&](int at, int cnt) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (qn_ub + cnt > WQ_CAP) {                    // room for a hit per lane in the hit queue
            wave_flush();
            qn_ub = 0;
        }
        qn_ub += cnt;
        if (lane < cnt) {
            const int64_t p = my_sv[at + lane];
            const int64_t al = p & ~(int64_t)3;
            uint32_t raw[NQ + 1];
#pragma unroll
            for (int k = 0; k < NQ + 1; ++k) raw[k] = load_codes4(a.codes, al + 4 * k, n_pos);
            double sc = 0.0;
#pragma unroll
            for (int k = 0; k < NQ; ++k) {
                const uint32_t cw = __builtin_amdgcn_alignbyte(raw[k + 1], raw[k], (uint32_t)(p & 3));
#pragma unroll
                for (int b = 0; b < 4; ++b) sc += tbl[(4 * k + b) * 8 + ((cw >> (8 * b)) & 7u)];      // rows m .. 4 NQ - 1 are zeros
            }
            const float f = (float)sc;
            if ((double)f > a.thr_seq) {
                const int slot = atomicAdd(&q_n[wave], 1);     // LDS
                my_pos[slot] = p;
                my_sc[slot] = f;
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
        // xs[d] byte k = (code at byte 4d + k, its low two bits) << ESH: a foreign letter or separator looks like one of
        // the four letters here and is rejected by the exact score (NaN), as in k_letters_cred
        uint32_t xs[NWD + 1];
#pragma unroll
        for (int d = 0; d < NWD + 1; ++d)
            xs[d] = (*reinterpret_cast<const uint32_t *>(cb + off0 + 4 * d) & 0x03030303u) << ESH;       // inside the halo
        // one look-up per position; PK[w] in the text above is pk[w + 4] here (w = -4 .. W-1)
        uint32_t pk[W + 4];
#pragma unroll
        for (int i = 0; i < W + 4; ++i) pk[i] = 0u;
        static_for<0, NPOS>([&](auto qc) __attribute__((always_inline)) {          // q, s2: constant expressions (see k_letters_cred)
            constexpr int q = decltype(qc)::value;
            const uint32_t by = (q & 3) ? __builtin_amdgcn_alignbyte(xs[(q >> 2) + 1], xs[q >> 2], (uint32_t)(q & 3)) : xs[q >> 2];
            const uint32_t off = __builtin_amdgcn_udot4(by, 0x40100401u, 0u, false);      // 4-mer index x entry size
            uint32_t dw[4] = {0u, 0u, 0u, 0u};
            if constexpr (ND == 1) {
                dw[0] = *reinterpret_cast<const entry_t *>(qbytes + off);
            } else {
                const entry_t e = *reinterpret_cast<const entry_t *>(qbytes + off);
                static_for<0, ND>([&](auto sc) __attribute__((always_inline)) { dw[decltype(sc)::value] = e[decltype(sc)::value]; });
            }
            static_for<0, ND>([&](auto sc) __attribute__((always_inline)) {
                constexpr int s2 = decltype(sc)::value;
                constexpr int wi = q - 8 * s2 - 4;     // rows 2 s2 (lo: window wi + 4) and 2 s2 + 1 (hi: window wi)
                if constexpr (wi >= -4 && wi <= W - 1) pk[wi + 4] += dw[s2];
            });
        });
        uint32_t sum[W];
        uint32_t any = 0;
#pragma unroll
        for (int v = 0; v < W; ++v) {
            sum[v] = (pk[v + 4] >> 16) + (pk[v] & 0xFFFFu);
            any |= sum[v];
        }
        // Survivors -> the wave's queue (positions only; windows past the end hold SEP codes and score NaN later)
        if (__builtin_amdgcn_ballot_w64((any & 0x8000u) != 0)) {
#pragma unroll
            for (int v = 0; v < W; ++v) {
                const bool sv = (sum[v] & 0x8000u) != 0;
                const unsigned long long sb = __builtin_amdgcn_ballot_w64(sv);
                if (sb) {                               // wave-uniform
                    if (sv) my_sv[sv_n + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(sb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)sb, 0u))] = tile0 + off0 + v;
                    sv_n += __popcll(sb);
                    if (sv_n >= 64) {                   // the top 64 get their exact score, the rest stays
                        exact_batch(sv_n - 64, 64);
                        sv_n -= 64;
                    }
                }
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
        if (most >= WQ_CAP / 2 || (!more && total > 0)) {          // uniform: every thread read the same snapshot
            if (threadIdx.x == 0) s_base = atomicAdd(counter, (unsigned long long)total);
            __syncthreads();
            drain(s_base + (unsigned long long)before, nq[wave]);
            qn_ub = 0;
        }
    }
}

// ---------------------------------------------------------------------------
// k_profile
// ---------------------------------------------------------------------------
// Score V consecutive windows per thread from a staged tile.  The sum over j is
// sequential inside one lane (fp64), like the reference loops.
template <int V, bool HAS_SEQ, typename PROF_T, bool FINITE>
__device__ __forceinline__ void compute_tile(const PROF_T *prof_lds, const unsigned char *code_lds, const char *tseq_lds,
                                             const double *__restrict__ pssm, int m, int la, double (&acc_st)[V],
                                             double (&acc_sq)[V])
{
    double rows[V][7];
    uint32_t sadr[V];                // LDS ADDRESS of this slot's letter in table row j0 (advanced per j0 round):
                                     // ds_read_b64 takes it as is, the row within the round is an immediate offset
    const uint32_t tbase = lds_addr(tseq_lds);
#pragma unroll
    for (int s = 0; s < V; ++s) {
        const PROF_T *r = prof_lds + (la + s) * 7;
#pragma unroll
        for (int k = 0; k < 7; ++k) rows[s][k] = (double)r[k];
        sadr[s] = HAS_SEQ ? tbase + (((uint32_t)code_lds[la + s] & 7u) << 3) : 0u;
        acc_st[s] = 0.0;
        acc_sq[s] = 0.0;
    }
#pragma unroll 1
    for (int j0 = 0; j0 < m; j0 += V) {
#pragma unroll
        for (int u = 0; u < V; ++u) {
            const int j = j0 + u;
            if (j < m) {                          // wave-uniform
                // constant address space: the table is read-only for the whole launch, so the
                // row is fetched with s_load (SGPR operands of v_fmac_f64) even in kernels that
                // also store to global memory
                const __attribute__((address_space(4))) double *prow =
                    (const __attribute__((address_space(4))) double *)(pssm + j * 7);
                double P[7];
#pragma unroll
                for (int k = 0; k < 7; ++k) P[k] = prow[k];
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    const int slot = (u + v) % V;  // holds stream position la + v + j
                    if (FINITE) {
                        // every row-dot is finite, nan_to_num is the identity: chain the
                        // 7 FMAs straight into the window sum (same terms, same order;
                        // differs from "round the dot, then add" by ~1e-16 relative)
                        double sacc = acc_st[v];
#pragma unroll
                        for (int k = 0; k < 7; ++k) sacc = fma(rows[slot][k], P[k], sacc);
                        acc_st[v] = sacc;
                    } else {
                        double d = rows[slot][0] * P[0];
#pragma unroll
                        for (int k = 1; k < 7; ++k) d = fma(rows[slot][k], P[k], d);
                        acc_st[v] += nan_to_num(d);
                    }
                    // table row j = j0 + u: the j0 part lives in sadr, u * 64 is an immediate offset
                    if (HAS_SEQ)
                        acc_sq[v] += *(const __attribute__((address_space(3))) double *)(uintptr_t)(sadr[slot] + u * 64);
                }
                // slot u is dead now: slide in position la + j + V (always staged: tile holds TILE + m rows)
                const PROF_T *r = prof_lds + (la + j + V) * 7;
#pragma unroll
                for (int k = 0; k < 7; ++k) rows[u][k] = (double)r[k];
                if (HAS_SEQ) sadr[u] = tbase + ((((uint32_t)code_lds[la + j + V]) & 7u) << 3) + (uint32_t)j0 * 64u;
            }
        }
        if (HAS_SEQ) {
#pragma unroll
            for (int s = 0; s < V; ++s) sadr[s] += V * 64;
        }
    }
    if (FINITE) {
        // A finite PSSM makes nan_to_num the identity unless the profile itself
        // holds NaN/inf (or the sum overflowed); any such event leaves a
        // non-finite sum behind, and only then is the exact path re-run.
#pragma unroll
        for (int v = 0; v < V; ++v)
            if (!(fabs(acc_st[v]) <= DBL_MAX)) acc_st[v] = struct_window_slow(prof_lds, la + v, pssm, m);
    }
}

// one tile per workgroup; overlap comes from several resident workgroups per CU
// launch bound = the residency the LDS tile allows (4 workgroups/CU at V=5, 3 at V=7)
template <int V, bool HAS_SEQ, typename PROF_T, bool FINITE, bool HITS, int DMA>
__global__ __launch_bounds__(BLOCK, (V <= 3 ? 5 : (V <= 5 ? 4 : 3))) void k_profile(const ScanArgs a)
{
    using L = ProfileLayout<V, PROF_T>;
    extern __shared__ __align__(16) unsigned char smem[];
    const int m = a.m;
    const int64_t tile0 = (int64_t)blockIdx.x * L::TILE;
    const int prof_bytes = L::prof_bytes(m);
    char *tseq_lds = reinterpret_cast<char *>(smem + L::buf_bytes(m, HAS_SEQ));
    // The waves of a NEW workgroup are the youngest on their SIMD and the arbiter serves the oldest first, so their ~60
    // staging instructions would queue behind the resident workgroups' FMAs; with raised priority the tile's loads are
    // in flight at once (C3: 2.209 -> 2.198 and 2.215 -> 2.195 ms in two interleaved A/B pairs, profiles/r4/NOTES.md;
    // the same for the stores of a finished wave changed nothing).  PFMSCAN_PRIO=0 turns it off.
    if (a.prio) __builtin_amdgcn_s_setprio(3);
    if (!(a.ablate & 2)) stage_tile<V, HAS_SEQ, PROF_T, DMA>(a, tile0, smem, m);
    if (HAS_SEQ)
        for (int i = threadIdx.x; i < m * 8; i += BLOCK) reinterpret_cast<double *>(tseq_lds)[i] = a.letter_table[i];
    if (a.prio) __builtin_amdgcn_s_setprio(0);
    if (DMA == 2) dma_wait_all();
    __syncthreads();
    const int la = threadIdx.x * V;
    double acc_st[V], acc_sq[V];
    if (a.ablate & 1) {            // timing diagnostic only (PFMSCAN_ABLATE): skip the scoring loop
#pragma unroll
        for (int v = 0; v < V; ++v) {
            acc_st[v] = (double)reinterpret_cast<const PROF_T *>(smem)[(la + v) * 7];
            acc_sq[v] = (double)smem[prof_bytes + la + v];
        }
    } else {
        compute_tile<V, HAS_SEQ, PROF_T, FINITE>(reinterpret_cast<const PROF_T *>(smem), smem + prof_bytes, tseq_lds,
                                                 a.struct_pssm, m, la, acc_st, acc_sq);
    }
    if (a.ablate & 4) {            // timing diagnostic only: skip the output path (keep the sums alive)
        double keep = 0.0;
#pragma unroll
        for (int v = 0; v < V; ++v) keep += acc_st[v] + acc_sq[v];
        if (keep == 1.2345e300) a.out_struct[0] = keep;
        return;
    }
    if (HITS) settle_near<V, PROF_T>(a, reinterpret_cast<const PROF_T *>(smem), la, acc_st);
    if (HITS || (a.ablate & 8))
        emit_tile<V, HAS_SEQ, HITS>(a, tile0, la, acc_st, acc_sq, smem);
    else
        emit_tile_wave<V, HAS_SEQ, PROF_T>(a, tile0, la, acc_st, acc_sq, smem, m);
}

// ---------------------------------------------------------------------------
// Exact structure score of the window at stream position p, rows straight from global memory
// (rnascan.py:302-307: score += nan_to_num(dot(profile[p+j], pssm[j]))).  Used where the windows are
// scattered (verification of letter-side candidates), so the cost is the number of vector-memory REQUESTS
// (every lane touches its own cache line), not bytes: 4 rows = 28 contiguous values are fetched as 7
// element-aligned 4-vectors (1.75 requests per row; 16 + 8 + 4 bytes per row took 3).  The last group is
// moved back to end with the window, rows already added are skipped.  The window's m rows must lie inside
// the stream (true for every window whose letters have no separator).
// ---------------------------------------------------------------------------
template <typename PROF_T>
__device__ __forceinline__ double struct_score_at(const void *profile, int64_t p, int m, const double *struct_pssm)
{
    typedef PROF_T v4_t __attribute__((ext_vector_type(4), aligned(sizeof(PROF_T))));
    const PROF_T *__restrict__ prof = reinterpret_cast<const PROF_T *>(profile) + p * 7;
    const __attribute__((address_space(4))) double *pssm = (const __attribute__((address_space(4))) double *)struct_pssm;
    double score = 0.0;
    for (int j0 = 0; j0 < m; j0 += 4) {
        PROF_T val[28];
        int base = j0;
        if (m >= 4) {
            base = j0 < m - 4 ? j0 : m - 4;
            const PROF_T *r = prof + base * 7;
#pragma unroll
            for (int k = 0; k < 7; ++k) {
                const v4_t q = *reinterpret_cast<const v4_t *>(r + 4 * k);
                val[4 * k] = q[0];
                val[4 * k + 1] = q[1];
                val[4 * k + 2] = q[2];
                val[4 * k + 3] = q[3];
            }
        } else {
#pragma unroll
            for (int e = 0; e < 28; ++e) val[e] = e < m * 7 ? prof[e] : (PROF_T)0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = base + u;
            if (j >= j0 && j < m) {
                double d = (double)val[u * 7] * pssm[j * 7];
#pragma unroll
                for (int c = 1; c < 7; ++c) d = fma((double)val[u * 7 + c], pssm[j * 7 + c], d);
                score += nan_to_num(d);
            }
        }
    }
    return score;
}

// The same score with the loads of ALL row groups issued before the first use (NG = ceil(m / 4) <= 4 groups, 7 vectors
// each): one memory round trip per candidate instead of NG dependent ones.  Same rows, same operation order, same bits.
template <typename PROF_T, int NG>
__device__ __forceinline__ double struct_score_at_wide(const void *profile, int64_t p, int m, const double *struct_pssm)
{
    typedef PROF_T v4_t __attribute__((ext_vector_type(4), aligned(sizeof(PROF_T))));
    const PROF_T *__restrict__ prof = reinterpret_cast<const PROF_T *>(profile) + p * 7;
    const __attribute__((address_space(4))) double *pssm = (const __attribute__((address_space(4))) double *)struct_pssm;
    v4_t q[NG][7];
    int base[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        base[g] = 4 * g < m - 4 ? 4 * g : m - 4;       // the last group is moved back to end with the window (m >= 4 here)
        const PROF_T *r = prof + base[g] * 7;
#pragma unroll
        for (int k = 0; k < 7; ++k) q[g][k] = *reinterpret_cast<const v4_t *>(r + 4 * k);
    }
    double score = 0.0;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = base[g] + u;
            if (j >= 4 * g && j < m) {                  // rows a previous group already added are skipped
                double d = (double)q[g][(u * 7) >> 2][(u * 7) & 3] * pssm[j * 7];
#pragma unroll
                for (int c = 1; c < 7; ++c) d = fma((double)q[g][(u * 7 + c) >> 2][(u * 7 + c) & 3], pssm[j * 7 + c], d);
                score += nan_to_num(d);
            }
        }
    }
    return score;
}

// ---------------------------------------------------------------------------
// k_struct_at -- verify phase of the candidate-then-verify combined scan.
// A combined hit needs seq > thr AND struct > thr (rnascan.py:422-433 joins two
// independently thresholded tables), and at real thresholds the letter side passes a
// tiny fraction of the windows.  So the combined scan can run the 1-byte-per-position
// letters kernel over everything and read the 28-byte-per-position profile only at its
// hits: one thread per candidate, m contiguous rows straight from global memory, the
// exact per-row nan_to_num path.  Same filter, same scores, ~29x fewer bytes.
// ---------------------------------------------------------------------------
// The candidate counts live on the device, so the grid cannot be sized to them: a fixed grid of
// workgroups strides over the 256-candidate chunks of all shards.  (A grid that covered every
// shard's CAPACITY spent 70 us launching 74k workgroups that read a count and left.)
constexpr int STRUCT_AT_MAX_SHARDS = 64;

template <typename PROF_T>
__global__ __launch_bounds__(BLOCK) void k_struct_at(const ScanArgs a, const int64_t *__restrict__ cand_pos,
                                                     const float *__restrict__ cand_seq,
                                                     const unsigned long long *__restrict__ cand_count,
                                                     const int64_t cand_shard_cap, const int cand_shards)
{
    __shared__ int chunk_end[STRUCT_AT_MAX_SHARDS];     // inclusive prefix of ceil(n_s / 256)
    __shared__ int64_t shard_n[STRUCT_AT_MAX_SHARDS];
    if ((int)threadIdx.x < cand_shards) {               // one count per lane: 32 dependent-latency loads in a row cost 60 us
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
    const int m = a.m;

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
            p = cand_pos[at];
            sq = cand_seq[at];
            // widths 5 .. 16: every row group in flight at once (wave-uniform choice; float rows: 84-112 VGPRs of loads)
            if (sizeof(PROF_T) == 4 && m > 8 && m <= 12) score = struct_score_at_wide<PROF_T, 3>(a.profile, p, m, a.struct_pssm);
            else if (sizeof(PROF_T) == 4 && m > 12 && m <= 16) score = struct_score_at_wide<PROF_T, 4>(a.profile, p, m, a.struct_pssm);
            else if (sizeof(PROF_T) == 4 && m > 4 && m <= 8) score = struct_score_at_wide<PROF_T, 2>(a.profile, p, m, a.struct_pssm);
            else score = struct_score_at<PROF_T>(a.profile, p, m, a.struct_pssm);
            if (struct_near(score, a.thr_struct, a.struct_band)) {       // too close to call: the reference's rounded order decides
                const double *pssm = a.struct_pssm;
                score = struct_window_rounded(reinterpret_cast<const PROF_T *>(a.profile) + p * 7, m, [&](int j, int k) { return pssm[j * 7 + k]; });
            }
            mask = score > a.thr_struct ? 1u : 0u;
        }
        emit_hits_block<1>(mask, [&](int) { return p; }, [&](int) { return sq; }, [&](int) { return score; }, a);
    }
}

hipError_t launch_struct_at(const ScanArgs &a, const int64_t *cand_pos, const float *cand_seq,
                            const unsigned long long *cand_count, int cand_shards, int64_t cand_shard_cap,
                            hipStream_t stream)
{
    if (cand_shard_cap <= 0 || cand_shards <= 0) return hipSuccess;
    if (cand_shards > STRUCT_AT_MAX_SHARDS) return hipErrorInvalidValue;
    const int64_t worst = (cand_shard_cap + BLOCK - 1) / BLOCK * cand_shards;
    const unsigned grid = (unsigned)std::min<int64_t>(worst, 2048);          // 8 workgroups per CU
    if (a.profile_dtype == PFMSCAN_PROFILE_F64)
        hipLaunchKernelGGL(k_struct_at<double>, dim3(grid), dim3(BLOCK), 0, stream, a, cand_pos, cand_seq, cand_count,
                           cand_shard_cap, cand_shards);
    else
        hipLaunchKernelGGL(k_struct_at<float>, dim3(grid), dim3(BLOCK), 0, stream, a, cand_pos, cand_seq, cand_count,
                           cand_shard_cap, cand_shards);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
// integer position-keyed prefilter (k_letters_cred) for a single 4-letter motif of width <= 32 with a finite threshold;
// Tiles one workgroup of a tile-walking hits kernel (k_letters_pre / _cred / _quad) takes.  A workgroup pays its table load,
// its final flush and its launch once, so more tiles per workgroup are cheaper -- until the grid is only a round or two of
// the 5-6 workgroups a CU holds (26 KB of LDS each): 32 tiles on C2 = 2290 workgroups = 1.5 rounds of 1536, and the second
// round ran on a half-empty chip (4.2 resident waves per SIMD on average in the counters).  ~24 workgroups per CU = four
// or five rounds: C2 (w = 8) -m 6 0.134 -> 0.128 ms, no hits 0.084 -> 0.079, k_letters_pre 0.211 -> 0.191, w = 4 -m 2
// 0.400 -> 0.355 (tools/gpu_sweep_tpb.sh).
static int walk_tiles(int64_t ntiles, const Tuning &t)
{
    // round 4: ~18 per CU.  Every workgroup ends with a serial tail -- the last survivors' exact scores (an L2 round trip), the
    // returning atomic, the drain -- so fewer, longer workgroups win once there ARE hits: C2 -m 6 0.1065 (12 tiles, 24 per CU)
    // -> 0.0956 (16 tiles, 18 per CU), no hits 0.0745 -> 0.0746, w = 18 0.1453 -> 0.1441; 20+ tiles lose again (no hits 0.078)
    const int64_t per = (int64_t)t.n_cu * 18;
    return (int)std::min<int64_t>(32, std::max<int64_t>(1, (ntiles + per / 2) / per));
}

// false when the motif has +inf / NaN two-letter sums (the fp32 prefilter handles those)
bool launch_letters_cred(const ScanArgs &a, const Tuning &t, hipStream_t stream, hipError_t *err)
{
    if (!(a.hits && !a.f64_hits && a.pair_table && a.h_pairsum && a.m <= 32 && t.credits && std::isfinite(a.thr_seq))) return false;
    constexpr int CRED_TILE = BLOCK * 16;              // k_letters_cred: 16 windows per lane
    const int npair = (a.m + 1) / 2, nj = (npair + 1) / 2;
    CredCache local, *cc = a.cred_cache ? a.cred_cache : &local;
    CredTable ct;
    {
    // the motif keeps what the last threshold gave; host threads that scan with one motif take turns here and every launch
    // carries its own copy of the table (as in launch_letters_cred8)
    static std::mutex cred_mu;
    std::lock_guard<std::mutex> lock(cred_mu);
    if (!(cc->thr == a.thr_seq) || cc->mode == 0) {
        cc->thr = a.thr_seq;
        const double slack = build_credits(a.h_pairsum, npair, a.thr_seq, cc->cr);
        cc->mode = std::isfinite(slack) ? 1 : 3;
        // Dense thresholds: the queued exact re-score re-reads the survivors' letters (L2) and runs one survivor per lane,
        // which loses to k_letters_pre's in-place re-score once more than a few per cent of the windows survive (w = 4 at
        // -m 2: 4.7 % survive, 0.45 against 0.40 ms).  There is no pilot pass on this path, so the survivor rate is
        // predicted from the credit table itself: the exact distribution of a window's credit sum for independent,
        // uniformly drawn letters.
        if (cc->mode == 1) {
            std::vector<double> dist(65536, 0.0), next(65536, 0.0);
            dist[0] = 1.0;
            int top = 0;
            for (int tr = 0; tr < npair; ++tr) {
                std::fill(next.begin(), next.end(), 0.0);
                int ntop = 0;
                for (int v = 0; v <= top; ++v) {
                    if (dist[(size_t)v] == 0.0) continue;
                    for (int i = 0; i < 16; ++i) {
                        const int w2 = std::min(65535, v + (int)cc->cr[tr * 16 + i]);
                        next[(size_t)w2] += dist[(size_t)v] * (1.0 / 16.0);
                        ntop = std::max(ntop, w2);
                    }
                }
                dist.swap(next);
                top = ntop;
            }
            double survive = 0.0;
            for (int v = 32768; v <= top; ++v) survive += dist[(size_t)v];
            if (survive > 1.0 / 32.0) cc->mode = 2;
        }
    }
    if (cc->mode != 1) return false;                       // -> k_letters_pre (launch_letters_ndw)
    const uint16_t *cr = cc->cr;
    std::memset(&ct, 0, sizeof(ct));
    for (int i = 0; i < 16; ++i)
        for (int tr = 0; tr < npair; ++tr) ct.d[i][tr >> 1] |= (uint32_t)cr[tr * 16 + i] << (16 * (tr & 1));
    }
    ScanArgs b = a;
    const int64_t ntiles = (a.n_pos + CRED_TILE - 1) / CRED_TILE;
    b.tiles_per_block = walk_tiles(ntiles, t);
    if (t.tiles_per_block > 0) b.tiles_per_block = t.tiles_per_block;
    const unsigned g = (unsigned)((ntiles + b.tiles_per_block - 1) / b.tiles_per_block);
#define CRED_CASE(N) case N: if (b.codes2) hipLaunchKernelGGL((k_letters_cred<N, true>), dim3(g), dim3(BLOCK), 0, stream, b, ct); \
                             else hipLaunchKernelGGL((k_letters_cred<N, false>), dim3(g), dim3(BLOCK), 0, stream, b, ct); break;
    switch (nj) {
    CRED_CASE(1) CRED_CASE(2) CRED_CASE(3) CRED_CASE(4) CRED_CASE(5) CRED_CASE(6) CRED_CASE(7)
    default: if (b.codes2) hipLaunchKernelGGL((k_letters_cred<8, true>), dim3(g), dim3(BLOCK), 0, stream, b, ct);
             else hipLaunchKernelGGL((k_letters_cred<8, false>), dim3(g), dim3(BLOCK), 0, stream, b, ct);
             break;
    }
#undef CRED_CASE
    *err = hipGetLastError();
    return true;
}

// four-letter credit tables (k_letters_quad): the table of the call's threshold is built on the host and kept with the motif in
// a ring of QuadCache::SLOTS (threshold, device table) slots.  A threshold the ring does not hold takes the oldest slot: its
// table goes through the slot's pinned host copy with hipMemcpyAsync on the CALLER's stream -- the `_dev` entry points stay
// asynchronous (no hipDeviceSynchronize, no blocking copy: ADVICE round 4 on the cred8 twin of this code).  Ordering:
// `ready` (recorded behind the copy) is what launches from other streams wait for; `used` (recorded behind every launch that
// reads the slot) is what the host waits for before it overwrites the slot -- with a device-wide wait only in the case that
// launches from several streams have read it, whose completion one event cannot vouch for.
static std::mutex quad_mu;

void quad_cache_release(QuadCache &qc)
{
    for (QuadSlot &s : qc.slot) {
        if (s.used) (void)hipEventSynchronize(s.used);
        if (s.d_tab) (void)hipFree(s.d_tab);
        if (s.h_tab) (void)hipHostFree(s.h_tab);
        if (s.ready) (void)hipEventDestroy(s.ready);
        if (s.used) (void)hipEventDestroy(s.used);
        s = QuadSlot();
    }
}

static bool launch_letters_quad(const ScanArgs &a, const Tuning &t, hipStream_t stream, hipError_t *err)
{
    if (!(a.hits && !a.f64_hits && a.pair_table && a.h_quadsum && a.quad_cache && a.m <= 32 && t.credits && t.quad && std::isfinite(a.thr_seq)))
        return false;
    constexpr int QUAD_TILE = BLOCK * 16;
    constexpr size_t TAB_BYTES = 256 * 16;
    const int nq = (a.m + 3) / 4;
    const int edw = nq <= 2 ? 1 : (nq <= 4 ? 2 : 4);       // dwords per table entry
    QuadCache &qc = *a.quad_cache;
    std::lock_guard<std::mutex> lock(quad_mu);
    if (qc.unusable) return false;
    QuadSlot *slot = nullptr;
    for (QuadSlot &s : qc.slot)
        if (s.d_tab && s.h_tab && s.ready && s.used && s.thr == a.thr_seq) slot = &s;
    if (!slot) {
        std::vector<uint16_t> cr((size_t)nq * 256);
        const double slack = build_credits(a.h_quadsum, nq, a.thr_seq, cr.data(), 16, 256);
        if (!std::isfinite(slack)) {                       // +inf / NaN four-letter sums: the fp32 prefilter handles those
            qc.unusable = true;                            // (a property of the motif, not of the threshold)
            return false;
        }
        slot = &qc.slot[qc.next];
        qc.next = (qc.next + 1) % QuadCache::SLOTS;
        if (!slot->d_tab || !slot->h_tab || !slot->ready || !slot->used) {     // first use (or an allocation failed half way)
            slot->thr = __builtin_nan("");
            if (!slot->d_tab && (*err = hipMalloc((void **)&slot->d_tab, TAB_BYTES)) != hipSuccess) return true;
            if (!slot->h_tab && (*err = hipHostMalloc((void **)&slot->h_tab, TAB_BYTES, hipHostMallocDefault)) != hipSuccess) return true;
            if (!slot->ready && (*err = hipEventCreateWithFlags(&slot->ready, hipEventDisableTiming)) != hipSuccess) return true;
            if (!slot->used && (*err = hipEventCreateWithFlags(&slot->used, hipEventDisableTiming)) != hipSuccess) return true;
        } else {
            // the launches that read the old table (and the copy that filled it) must be over before the pinned copy changes
            *err = slot->many_streams ? hipDeviceSynchronize() : hipEventSynchronize(slot->used);
            if (*err != hipSuccess) return true;
        }
        slot->thr = __builtin_nan("");
        std::memset(slot->h_tab, 0, TAB_BYTES);
        for (int i = 0; i < 256; ++i)
            for (int r = 0; r < nq; ++r) slot->h_tab[(size_t)i * edw + (r >> 1)] |= (uint32_t)cr[(size_t)r * 256 + i] << (16 * (r & 1));
        if ((*err = hipMemcpyAsync(slot->d_tab, slot->h_tab, (size_t)256 * edw * 4, hipMemcpyHostToDevice, stream)) != hipSuccess) return true;
        if ((*err = hipEventRecord(slot->ready, stream)) != hipSuccess) return true;
        if ((*err = hipEventRecord(slot->used, stream)) != hipSuccess) return true;     // (a reuse before any launch waits for the copy)
        slot->thr = a.thr_seq;
        slot->last_stream = stream;
        slot->many_streams = false;
    } else if (slot->last_stream != stream) {
        if ((*err = hipStreamWaitEvent(stream, slot->ready, 0)) != hipSuccess) return true;   // the copy ran on another stream
        slot->many_streams = true;
        slot->last_stream = stream;
    }
    ScanArgs b = a;
    b.d_quad = slot->d_tab;
    const int64_t ntiles = (a.n_pos + QUAD_TILE - 1) / QUAD_TILE;
    b.tiles_per_block = walk_tiles(ntiles, t);
    if (t.tiles_per_block > 0) b.tiles_per_block = t.tiles_per_block;
    const unsigned g = (unsigned)((ntiles + b.tiles_per_block - 1) / b.tiles_per_block);
    switch (nq) {
    case 1: hipLaunchKernelGGL((k_letters_quad<1>), dim3(g), dim3(BLOCK), 0, stream, b); break;
    case 2: hipLaunchKernelGGL((k_letters_quad<2>), dim3(g), dim3(BLOCK), 0, stream, b); break;
    case 3: hipLaunchKernelGGL((k_letters_quad<3>), dim3(g), dim3(BLOCK), 0, stream, b); break;
    case 4: hipLaunchKernelGGL((k_letters_quad<4>), dim3(g), dim3(BLOCK), 0, stream, b); break;
    case 5: hipLaunchKernelGGL((k_letters_quad<5>), dim3(g), dim3(BLOCK), 0, stream, b); break;
    case 6: hipLaunchKernelGGL((k_letters_quad<6>), dim3(g), dim3(BLOCK), 0, stream, b); break;
    case 7: hipLaunchKernelGGL((k_letters_quad<7>), dim3(g), dim3(BLOCK), 0, stream, b); break;
    default: hipLaunchKernelGGL((k_letters_quad<8>), dim3(g), dim3(BLOCK), 0, stream, b); break;
    }
    *err = hipGetLastError();
    if (*err == hipSuccess) *err = hipEventRecord(slot->used, stream);
    return true;
}

template <int NDW>
static hipError_t launch_letters_ndw(const ScanArgs &a, const Tuning &t, hipStream_t stream)
{
    constexpr int LET_TILE = let_tile(NDW);
    const unsigned grid = (unsigned)((a.n_pos + LET_TILE - 1) / LET_TILE);
    // 8 windows per thread (hits: 0.39 vs 0.49 ms on C2 w=12; scores, with the LDS transpose that keeps
    // the stores 1 KiB contiguous: 0.39 vs 0.42 ms on C2 w=8; without the transpose 0.56 ms)
    if (a.hits && a.f64_hits)            // generic alphabet, fp64 compare and score (matrix.py:25-43): the exact kernel
        hipLaunchKernelGGL((k_letters<NDW, double, true, 8>), dim3(grid), dim3(BLOCK), 0, stream, a);
    else if (a.hits && a.pair_table) {
        // >= 2048 workgroups when the stream allows, at most 32 tiles per workgroup
        ScanArgs b = a;
        const int64_t ntiles = (a.n_pos + LET_TILE - 1) / LET_TILE;
        b.tiles_per_block = walk_tiles(ntiles, t);
        if (t.tiles_per_block > 0) b.tiles_per_block = t.tiles_per_block;      // tests: the multi-tile walk on small streams
        const double lo = a.thr_seq - a.pair_eps;
        b.thr_pre = (float)lo;
        if ((double)b.thr_pre > lo) b.thr_pre = std::nextafterf(b.thr_pre, -INFINITY);
        const unsigned g = (unsigned)((ntiles + b.tiles_per_block - 1) / b.tiles_per_block);
        hipLaunchKernelGGL((k_letters_pre<NDW>), dim3(g), dim3(BLOCK), 0, stream, b);
    }
    else if (a.hits)
        hipLaunchKernelGGL((k_letters<NDW, float, true, 8>), dim3(grid), dim3(BLOCK), 0, stream, a);
    else if (a.out_letters_f64)
        hipLaunchKernelGGL((k_letters<NDW, double, false, 4>), dim3(grid), dim3(BLOCK), 0, stream, a);
    else
        hipLaunchKernelGGL((k_letters<NDW, float, false, 8>), dim3(grid), dim3(BLOCK), 0, stream, a);
    return hipGetLastError();
}

static hipError_t launch_letters(const ScanArgs &a, const Tuning &t, hipStream_t stream)
{
    hipError_t e = hipSuccess;
    if (launch_letters_quad(a, t, stream, &e)) return e;     // PFMSCAN_QUAD=1 only: measured slower (profiles/r3/NOTES.md, "tried and dropped")
    if (launch_letters_cred(a, t, stream, &e)) return e;
    if (launch_letters_cred8(a, t, stream, &e)) return e;    // fp64 hits of a generic alphabet at a finite threshold
    if (launch_letters_fixed(a, stream, &e)) return e;       // all float32 scores, widths 2 .. 32: the width is a compile-time constant
    if (a.m <= 16) return launch_letters_ndw<5>(a, t, stream);
    if (a.m <= 32) return launch_letters_ndw<9>(a, t, stream);
    return launch_letters_ndw<17>(a, t, stream);
}

template <int V, bool HAS_SEQ, typename PROF_T, bool FINITE, bool HITS, int DMA>
static hipError_t launch_profile_inst(const ScanArgs &a, hipStream_t stream)
{
    using L = ProfileLayout<V, PROF_T>;
    const unsigned grid = (unsigned)((a.n_pos + L::TILE - 1) / L::TILE);
    const int lds = L::total(a.m, HAS_SEQ, 1);
    auto kern = k_profile<V, HAS_SEQ, PROF_T, FINITE, HITS, DMA>;
    static std::atomic<uint64_t> configured{0};     // per instantiation, one bit per device
    hipError_t e = allow_full_lds(reinterpret_cast<const void *>(kern), configured);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(BLOCK), lds, stream, a);
    return hipGetLastError();
}

template <int V, bool HAS_SEQ, typename PROF_T, int DMA>
static hipError_t launch_profile_v(const ScanArgs &a, hipStream_t stream)
{
    if (a.hits) {
        if (a.struct_finite) return launch_profile_inst<V, HAS_SEQ, PROF_T, true, true, DMA>(a, stream);
        return launch_profile_inst<V, HAS_SEQ, PROF_T, false, true, DMA>(a, stream);
    }
    if (a.struct_finite) return launch_profile_inst<V, HAS_SEQ, PROF_T, true, false, DMA>(a, stream);
    return launch_profile_inst<V, HAS_SEQ, PROF_T, false, false, DMA>(a, stream);
}

template <bool HAS_SEQ, typename PROF_T>
static hipError_t launch_profile_t(const ScanArgs &a, const Tuning &t, hipStream_t stream)
{
    if (t.dma) {
        hipError_t e = hipSuccess;
        if (t.v == 5 && launch_profile_fixed(a, stream, &e)) return e;        // a width with an unrolled instantiation (all scores and the fused hits pass)
        if (t.v == 7) return launch_profile_v<7, HAS_SEQ, PROF_T, 2>(a, stream);
        return launch_profile_v<5, HAS_SEQ, PROF_T, 2>(a, stream);
    }
    if (t.v == 7) return launch_profile_v<7, HAS_SEQ, PROF_T, 0>(a, stream);
    return launch_profile_v<5, HAS_SEQ, PROF_T, 0>(a, stream);
}

// ---------------------------------------------------------------------------
// k_wide -- PFMs wider than PFMSCAN_MAX_M (the reference's loops take any width: _pwm.c:34-68, rnascan.py:302-307).
// The tuned kernels are unrolled and sized for widths up to 64; a wider motif -- none of the reference's examples is --
// takes this plain form: one thread per window, rolled loops over the motif rows in the reference's operation order
// (sequential fp64 sum of the letter table; per row d = r0 * s0, six FMAs, score += nan_to_num(d)), rows and codes
// straight from global memory (neighbouring threads read neighbouring bytes / rows; the tables through the caches).
// Same outputs and hit rule as k_letters / k_profile in every mode.  No roofline claim: O(m) dependent loads per window.
// ---------------------------------------------------------------------------
template <typename PROF_T>
__global__ __launch_bounds__(BLOCK) void k_wide(const ScanArgs a)
{
    const int m = a.m;
    const int64_t n_pos = a.n_pos;
    const int64_t p = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    const bool in = p < n_pos;
    const bool has_seq = a.letter_table != nullptr, has_st = a.struct_pssm != nullptr;
    double sq = 0.0, st = 0.0;
    if (in && has_seq) {
#pragma unroll 4
        for (int j = 0; j < m; ++j) {
            const uint32_t c = p + j < n_pos ? (uint32_t)a.codes[p + j] & 7u : (uint32_t)PFMSCAN_SEP;     // past the end: NaN column
            sq += a.letter_table[j * 8 + c];
        }
    }
    if (in && has_st) {
        if (p + m > n_pos) {
            st = __longlong_as_double(0x7ff8000000000000ll);       // the window runs past the stream end
        } else {
            const PROF_T *__restrict__ row = reinterpret_cast<const PROF_T *>(a.profile) + p * 7;
            const __attribute__((address_space(4))) double *pssm = (const __attribute__((address_space(4))) double *)a.struct_pssm;
#pragma unroll 2
            for (int j = 0; j < m; ++j) {
                double d = (double)row[j * 7] * pssm[j * 7];
#pragma unroll
                for (int k = 1; k < 7; ++k) d = fma((double)row[j * 7 + k], pssm[j * 7 + k], d);
                st += nan_to_num(d);
            }
        }
    }
    if (a.hits) {
        bool pass = in;
        if (in && has_st && struct_near(st, a.thr_struct, a.struct_band)) {      // (NaN is never near)
            const double *pssm = a.struct_pssm;
            st = struct_window_rounded(reinterpret_cast<const PROF_T *>(a.profile) + p * 7, m, [&](int j, int k) { return pssm[j * 7 + k]; });
        }
        if (has_st) pass = pass && (st > a.thr_struct);
        if (has_seq) pass = pass && ((a.f64_hits ? sq : (double)(float)sq) > a.thr_seq);
        // a letters-only scan reports its fp64 score in hit_struct (k_letters does), a scan with a structure part the structure score
        emit_hits_block<1>(pass ? 1u : 0u, [&](int) { return p; }, [&](int) { return (float)sq; }, [&](int) { return has_st ? st : sq; }, a);
        return;
    }
    if (!in) return;
    if (has_seq && a.out_seq) a.out_seq[p] = (float)sq;
    if (has_seq && a.out_letters_f64) a.out_letters_f64[p] = sq;
    if (has_st && a.out_struct) a.out_struct[p] = st;
}

static hipError_t launch_wide(const ScanArgs &a, hipStream_t stream)
{
    if ((a.letter_table && !a.codes) || (a.struct_pssm && !a.profile)) return hipErrorInvalidValue;
    hipError_t e = hipSuccess;
    if (!std::getenv("PFMSCAN_WIDE_PLAIN") && launch_wide_letters(a, stream, &e)) return e;      // letters only: the slab kernel
    const unsigned grid = (unsigned)((a.n_pos + BLOCK - 1) / BLOCK);
    if (a.struct_pssm && a.profile_dtype == PFMSCAN_PROFILE_F64)
        hipLaunchKernelGGL(k_wide<double>, dim3(grid), dim3(BLOCK), 0, stream, a);
    else
        hipLaunchKernelGGL(k_wide<float>, dim3(grid), dim3(BLOCK), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_scan(const ScanArgs &a, const Tuning &t, hipStream_t stream, const char **what)
{
    *what = "launch";
    if (a.n_pos <= 0) return hipSuccess;
    // PFMs wider than PFMSCAN_MAX_M: letters only -> the slab kernel (launch_wide); with a structure part k_profile itself
    // takes widths up to PROFILE_MAX_M (its loops are generic in m; the bound is the output staging of emit_tile_wave, which
    // lives in the part of a wave's band that the NEXT wave does not read: m - 1 + 64 V 12 / 28 rows < 64 V), beyond that
    // the plain one-thread-per-window kernel
    constexpr int PROFILE_MAX_M = 180;
    const bool profile_ok = a.struct_pssm && a.profile && !a.out_letters_f64 && a.m <= PROFILE_MAX_M && !std::getenv("PFMSCAN_WIDE_PLAIN");
    if (a.m > PFMSCAN_MAX_M && !profile_ok) return launch_wide(a, stream);
    if (!a.struct_pssm) return launch_letters(a, t, stream);
    if (a.out_letters_f64 || a.profile == nullptr) return hipErrorInvalidValue;
    const bool has_seq = a.letter_table != nullptr;
    if (a.profile_dtype == PFMSCAN_PROFILE_F64) {
        // fp64-stored profile (strict-parity storage): 56 B per position in LDS,
        // so the tile is kept at V = 5 (72 KB, two workgroups per CU).
        Tuning t5 = t;
        t5.v = 5;
        return has_seq ? launch_profile_t<true, double>(a, t5, stream) : launch_profile_t<false, double>(a, t5, stream);
    }
    return has_seq ? launch_profile_t<true, float>(a, t, stream) : launch_profile_t<false, float>(a, t, stream);
}

}  // namespace pfmscan
