// pfmscan_internal.hpp -- launch interface between the C ABI (pfmscan_api.hip)
// and the gfx950 kernels (pfmscan_kernels.hip).  Not installed.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include "../../include/pfmscan.h"

namespace pfmscan {

// what launch_letters_cred decided for a motif at the last threshold it saw (kept with the motif: building the credits and
// predicting the survivor rate is host work of a millisecond, the kernel takes a tenth of that)
struct CredCache {
    double thr = __builtin_nan("");
    int mode = 0;                // 1: k_letters_cred with cr; 2: dense threshold -> k_letters_pre; 3: no integer prefilter possible
    uint16_t cr[16 * 16];
};

// the same for launch_letters_cred8 (pfmscan_letters8.hip): single-letter credits of an 8-code alphabet, up to 32 rows
struct Cred8Cache {
    double thr = __builtin_nan("");
    int mode = 0;                // 1: k_letters_cred8 with cr; 2: dense threshold -> the exact kernel; 3: no integer prefilter possible
    uint16_t cr[32 * 8];
};

// k_letters_quad's credit tables (launch_letters_quad): a small ring of (threshold, device table) slots kept with the motif.
// A new threshold goes through the slot's PINNED host copy with hipMemcpyAsync on the caller's stream -- no device-wide
// synchronisation and no blocking copy inside the asynchronous `_dev` entry points; `ready` orders other streams behind
// the copy, `used` (recorded after the last launch that read the slot) is what a reuse of the slot waits for.
struct QuadSlot {
    double thr = __builtin_nan("");
    uint32_t *d_tab = nullptr;   // device: 256 entries of up to 16 bytes
    uint32_t *h_tab = nullptr;   // pinned host copy the asynchronous upload reads
    hipEvent_t ready = nullptr, used = nullptr;
    hipStream_t last_stream = nullptr;
    bool many_streams = false;   // launches from more than one stream have read this slot since it was filled
};
struct QuadCache {
    static constexpr int SLOTS = 4;
    QuadSlot slot[SLOTS];
    int next = 0;
    bool unusable = false;       // the motif's four-letter sums hold +inf / NaN: the fp32 prefilter handles it
};
void quad_cache_release(QuadCache &qc);      // pfmscan_kernels.hip: frees the slots' memory and events

struct ScanArgs {
    const uint8_t *codes;        // [n_pos] device, may be null when the motif has no letter table
    const void *profile;         // [n_pos][7] float or double, device, may be null
    int profile_dtype;           // PFMSCAN_PROFILE_*
    int64_t n_pos;
    const double *letter_table;  // [m][8] device or null
    const float *pair_table;     // [(m+1)/2][16] device or null: fp32 sums of two adjacent letters (4-letter alphabets),
                                 // index c0 | c1 << 2 -- the hits-mode prefilter of k_letters_pre
    int tiles_per_block;         // k_letters_pre: consecutive tiles one workgroup walks (set by the launcher)
    float thr_pre;               // k_letters_pre: largest float <= thr_seq - pair_eps (set by the launcher)
    double pair_eps;             // |fp32 pair-table score - exact score| <= pair_eps for every window
    const double *h_pairsum;     // HOST: [(m+1)/2][16] exact two-letter sums (4-letter alphabets): the launcher builds the
                                 // integer credit table of k_letters_cred from them for the call's threshold
    const double *h_quadsum;     // HOST: [(m+3)/4][256] exact four-letter sums (m <= 32): operand of k_letters_quad's credit table
    const uint32_t *d_quad;      // DEVICE: the table of the call's threshold (a slot of quad_cache; set by the launcher)
    QuadCache *quad_cache;       // HOST: owned by the motif
    CredCache *cred_cache;       // HOST: owned by the motif
    const uint8_t *codes2;       // two-FASTA combined scan fused into k_letters_cred: the second code stream (device) or null
    const double *letter_table2; // ... and its letter table [m][8] (device); hits then need seq > thr_seq AND letters2 > thr_struct
    const double *h_letters;     // HOST: the letter table [m][8] (operand of the single-letter credits of k_letters_cred8)
    Cred8Cache *cred8_cache;     // HOST: owned by the motif (the packed table itself travels with each launch)
    const double *struct_pssm;   // [m][7] device or null
    int m;
    int struct_finite;           // every struct_pssm cell finite -> fast path legal
    double struct_band;          // hits: |fast structure score - thr_struct| <= struct_band -> the window is scored again in the
                                 // reference's rounded order before the compare (pfmscan_exact.hpp)
    // all-scores outputs (position aligned), any may be null
    float *out_seq;
    double *out_struct;
    double *out_letters_f64;     // letter scan with fp64 output (matrix.py:25-43)
    // hits mode
    int hits;
    int f64_hits;                // letters-only hits of a generic alphabet (matrix.py:25-43): the fp64 sum itself is compared with
                                 // thr_seq (no float32 cast) and reported in hit_struct
    double thr_seq, thr_struct;
    int64_t capacity;
    int64_t *hit_pos;
    float *hit_seq;
    double *hit_struct;
    unsigned long long *hit_count;   // hit_shards counters, HIT_COUNTER_STRIDE words apart
    int hit_shards;                  // 1 (public _dev contract) or a power of two: workgroup b appends to shard
                                     // b & (hit_shards-1), i.e. to slots [shard*capacity, (shard+1)*capacity)
    int prio;                    // k_profile: staging instructions at raised wave priority (Tuning::prio)
    int dma_whole;               // k_profile: the last LDS-DMA piece of a region in full (Tuning::dma_whole; default: only the lanes with needed bytes)
    int ablate;                  // timing diagnostics (PFMSCAN_ABLATE): 1 no scoring, 2 no staging, 4 no output
    int64_t pos_offset;          // added to every reported hit position (chunked host pipeline: position of the chunk in the stream)
};

// Tuning knobs settable through the environment (read once per ctx), so that
// variants can be A/B-timed on the GPU box without a rebuild.
constexpr int HIT_COUNTER_STRIDE = 16;   // one 128-byte line per shard counter
constexpr int HIT_SHARDS = 32;           // shards of the ctx-owned hit / candidate buffers

struct Tuning {
    int v = 5;              // windows per thread in k_profile: 5 or 7 (odd: conflict-free LDS rows)
    int dma = 1;            // k_profile: stage the tile with LDS-DMA (global_load_lds, 2.34 ms on C3) instead of
                            // through registers (2.51 ms)
    int prio = 1;           // k_profile: s_setprio 3 while a new workgroup issues its tile's loads (PFMSCAN_PRIO=0: off)
    int dma_whole = 0;      // PFMSCAN_DMA_TAIL=0: 1
    int ablate = 0;         // see ScanArgs::ablate; results are WRONG when non-zero
    int prefilter = 1;      // hits over 4-letter alphabets: fp32 two-letter prefilter, exact fp64 re-score of survivors
    int tiles_per_block = 0; // k_letters_pre: 0 = pick from the stream length; > 0 forces it (PFMSCAN_TILES_PER_BLOCK, tests)
    int n_cu = 256;         // compute units of the ctx's device (set at ctx creation): sizes the grids of the tile-walking kernels
    int two_phase = 1;      // combined hits through the host/staged API: letters first, structure only at candidates
    int credits = 1;        // hits over 4-letter alphabets, m <= 32: integer position-keyed prefilter (k_letters_cred) instead of
                            // the fp32 one (k_letters_pre); PFMSCAN_CREDITS=0 for A/B runs and tests
    int quad = 0;           // PFMSCAN_QUAD=1: FOUR-letter credit tables (k_letters_quad) instead -- a third of the VALU instructions,
                            // but its 256-entry look-ups are lane-random over all LDS banks: measured 8-25 % SLOWER at every width
                            // (profiles/r3/ab_single_motif_hits_kernels.txt); kept for that A/B and covered by the parity tests
};

hipError_t launch_scan(const ScanArgs &a, const Tuning &t, hipStream_t stream, const char **what);
// pfmscan_letters8.hip: fp64 letter hits through the single-letter integer prefilter (false: not applicable, take the exact
// kernel); fp64 letter score of a SECOND code stream at the candidates of a letters pass (two-FASTA combined scan)
bool launch_letters_cred8(const ScanArgs &a, const Tuning &t, hipStream_t stream, hipError_t *err);
// pfmscan_kernels.hip: the integer-prefiltered hits kernel of 4-letter alphabets (false: not applicable -- dense threshold,
// +inf cells, width > 32); with a.codes2 / a.letter_table2 set it verifies the second stream in the same launch
bool launch_letters_cred(const ScanArgs &a, const Tuning &t, hipStream_t stream, hipError_t *err);
// letters-only scans of PFMs wider than PFMSCAN_MAX_M: codes in LDS, the table streamed through LDS in 64-row slabs
bool launch_wide_letters(const ScanArgs &a, hipStream_t stream, hipError_t *err);
bool launch_profile_fixed(const ScanArgs &a, hipStream_t stream, hipError_t *err);     // pfmscan_profile_fixed.hip
bool launch_letters_fixed(const ScanArgs &a, hipStream_t stream, hipError_t *err);     // pfmscan_letters_fixed.hip: all float32 scores, widths 2..32
hipError_t launch_letters_at(const ScanArgs &a, const int64_t *cand_pos, const float *cand_seq,
                             const unsigned long long *cand_count, int cand_shards, int64_t cand_shard_cap,
                             hipStream_t stream);

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a PER-DEVICE setting of a kernel: `done` (one per kernel
// instantiation) keeps one bit per device, so a second ctx on another device of the same process gets its attribute
// too, and two host threads may race here harmlessly (the worst case sets it twice).  Call under the ctx's hipSetDevice.
inline hipError_t allow_dynamic_lds(const void *kern, std::atomic<uint64_t> &done, int bytes)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const uint64_t bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
    e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return e;
    done.fetch_or(bit, std::memory_order_release);
    return hipSuccess;
}

// Credits of ONE motif at threshold thr (pfmscan_library_api.hip): pairsum [npair][16] exact two-letter sums ->
// out [npair][16] unsigned 16-bit credits with the threshold folded into row 0; "bit 15 of the sum clear" => the
// window cannot be a hit.  Returns the one-sided slack in score units (inf: no prefilter possible).  Host code.
double build_credits(const double *pairsum, int npair, double thr, uint16_t *out, int bits = 16, int nent = 16);
void pair_sums(const double *letter_table, int m, double *out);      // [m][8] -> [ceil(m/2)][16]
void quad_sums(const double *letter_table, int m, double *out);      // [m][8] -> [ceil(m/4)][256] four-letter sums

// Second phase of the candidate-then-verify combined scan: structure score of the windows
// listed in cand_pos[0 .. min(*cand_count, cand_cap)) (hits of a letters-only pass, whose
// float32 scores are cand_seq), kept when > a.thr_struct and appended to a.hit_*.
// Candidates are sharded: shard s holds min(cand_count[s * HIT_COUNTER_STRIDE], cand_shard_cap) entries at
// [s * cand_shard_cap, ...).
hipError_t launch_struct_at(const ScanArgs &a, const int64_t *cand_pos, const float *cand_seq,
                            const unsigned long long *cand_count, int cand_shards, int64_t cand_shard_cap,
                            hipStream_t stream);

// Sharded hit buffers -> one run sorted by position (pfmscan_sort.hip): keys_out holds the positions,
// seq_out / struct_out the scores in the same order.  All pointers are device memory.
struct GatherArgs {
    const int64_t *hit_pos;
    const float *hit_seq;                 // may be null
    const double *hit_struct;             // may be null
    const unsigned long long *counts;     // shards counters, HIT_COUNTER_STRIDE words apart
    int shards;
    int64_t shard_cap;
    int64_t total;                        // sum over shards of min(count, shard_cap)
    int key_bits;                         // bits a stream position can have
    int64_t *keys_in, *keys_out, *vals_in, *vals_out;   // [total] each
    void *temp;
    size_t temp_bytes;                    // >= sort_temp_bytes(total, key_bits)
    float *seq_out;                       // [total]
    double *struct_out;                   // [total]
    // library scans: motif index per hit, sorted with the hits; key = pos << motif_bits | motif
    const int32_t *hit_motif = nullptr;   // may be null
    int32_t *motif_out = nullptr;         // [total]
    int motif_bits = 0;
};

// ---- multi-PFM library scan (pfmscan_library.hip) ------------------------------------------------------
// one workgroup per CU, its waves independent and sharing the LDS tables: 16 waves (128 VGPRs each) for PFMs up to
// width 32, 8 waves (256 VGPRs) for the widest bucket, whose 32 pair addresses per lane would otherwise spill
__host__ __device__ constexpr int lib_block(int np_bucket) { return np_bucket > 16 ? 512 : 1024; }
constexpr int LIB_QCAP = 128;             // (window, motif group) items a wave can park (>= 64 + 63)
// motifs per 16-byte table entry and bits per credit: PFMs up to width 16 (at most 8 pair rows) pack THREE 10-bit credits
// per dword (12 motifs per look-up, V = 511 / (npair - 1) >= 73 levels per row), wider ones two 16-bit credits (8 motifs)
__host__ __device__ constexpr int lib_mpg(int np_bucket) { return np_bucket == 8 ? 12 : 8; }
__host__ __device__ constexpr int lib_credit_bits(int np_bucket) { return np_bucket == 8 ? 10 : 16; }
constexpr int LIB_SHARDS = 256;           // sharded hit buffers of a library scan: workgroup b appends to shard b & 255

constexpr int LIB_SEG_SHIFT = 14;           // windows per work segment of k_library: 16384 (the kernel turns chunk tickets into positions with shifts)
struct LibArgs {
    const uint8_t *codes;                 // [n_pos]
    const void *profile;                  // [n_pos][7] or null (sequence-only library)
    int profile_dtype;
    int64_t n_pos;                        // stream length (bounds of every read)
    int64_t pos_base, span;               // this launch scores the windows starting in [pos_base, pos_base + span), span < 2^32
    int64_t pos_offset;                   // added to every reported hit position (chunked host pipeline: where the buffer sits in the stream)
    int64_t seg_positions, n_seg;         // work split: segment s (seg_positions = 2^LIB_SEG_SHIFT windows) -> workgroup s mod grid
    // one pass = nmp = 8 * ng motifs, tables laid out for the kernel (pfmscan_library_api.hip builds them)
    const uint32_t *pairs;                // [npair][ng][16][8] u16 two-letter credits, threshold folded into pair row 0
    const double *letters;                // [m * 4][nmp] fp64, transposed
    const double *pssm;                   // [m * 4][nmp][2] fp64: row j, column pair c/2, motif, c&1 (column 7 = 0); null = no structure side
    const double *thr_seq, *thr_struct;   // [nmp]
    int struct_finite;                    // every cell of every structure PSSM of the library is finite: phase B chains the row FMAs
    double struct_band;                   // re-score band of the thresholded structure compare (max over the library's motifs, pfmscan_exact.hpp)
    int m, npair, nmp, ng, motif_base;
    int sort_batches;                     // A/B: phase B sorts each 64-item batch by motif group (PFMSCAN_LIB_SORT=1)
    int ng_real;                          // motif groups of the pass that hold motifs (<= ng, the layout of its tables): the rest is skipped
    // Teams: passes SIDE BY SIDE in one launch.  Workgroups [team_first[t], team_first[t + 1]) run pass t of n_teams
    // consecutive passes whose tables lie stride_* elements apart (same ng layout) and whose motifs start nmp apart;
    // team_first[t] = the grid for t >= n_teams.  n_teams <= 1: one pass, every workgroup (team_ng unused).
    int n_teams;
    int team_first[5];
    int team_ng[4];                       // ng_real per team
    int64_t stride_pairs, stride_letters, stride_pssm, stride_thr;      // in elements of the respective arrays
    // hits: LIB_SHARDS (or 1) regions of shard_cap slots, counters HIT_COUNTER_STRIDE words apart
    int64_t shard_cap;
    int hit_shards;
    int64_t *hit_pos;
    int32_t *hit_motif;
    float *hit_seq;
    double *hit_struct;
    unsigned long long *hit_count;
};
size_t lib_group_bytes(int m, int npair, bool has_struct, int np_bucket);   // LDS bytes per motif group of a pass
size_t lib_queue_bytes(int np_bucket);                          // LDS bytes of the wave queues
size_t lib_lds_bytes(int m, int npair, int ng, bool has_struct, int np_bucket);
int lib_np_bucket(int m);
int lib_pick_ng(int np_bucket, int want_groups, int max_groups);   // supported group count of a pass (0: none fits)
hipError_t launch_library(const LibArgs &a, int n_cu, hipStream_t stream);
// LibArgs::profile_dtype of a two-FASTA library: `profile` is the SECOND code stream and `pssm` holds [m][8] letter tables
constexpr int PROFILE_LETTERS2 = 100;
// generic-alphabet letter libraries (k_library8): `pairs` = single-letter credits [rows][ng][8 codes][8 motifs] u16, `npair` =
// lib8_rows(m) (the width rounded up to a multiple of 4), `pssm` = the exact tables [m * 4][nmp][2], thr_struct [nmp]; m <= 32
__host__ __device__ constexpr int lib8_rows(int m) { return (m + 3) / 4 * 4; }
size_t lib8_group_bytes(int m);
size_t lib8_lds_bytes(int m, int ng);
int lib8_pick_ng(int want_groups, int max_groups);
hipError_t launch_library8(const LibArgs &a, int n_cu, hipStream_t stream);

// ---- structure-only PFM library (pfmscan_proflib.hip): every motif in one pass over the profile ---------------------
struct ProfLibArgs {
    const void *profile;                  // [n_pos][7] float or double
    int profile_dtype;
    int64_t n_pos;
    const double *pssm;                   // [n_motifs][m][7] fp64, row-major as handed to pfmscan_library_create
    const double *thr;                    // [n_motifs] structure thresholds (hit <=> score > thr, rnascan.py:310)
    double struct_band;                   // re-score band of that compare (max over the library's motifs, pfmscan_exact.hpp)
    const int32_t *finite;                // [n_motifs] 1 = every cell of the motif's PSSM is finite
    int n_motifs, m, motif_base;
    int64_t pos_offset;                   // added to every reported hit position (chunked host pipeline)
    // hits: hit_shards regions of shard_cap slots, counters HIT_COUNTER_STRIDE words apart (workgroup b -> shard b & (shards-1))
    int64_t shard_cap;
    int hit_shards;
    int64_t *hit_pos;
    int32_t *hit_motif;
    double *hit_struct;
    unsigned long long *hit_count;
};
hipError_t launch_profile_library(const ProfLibArgs &a, hipStream_t stream);
int64_t profile_library_tile();           // stream positions per workgroup (workgroup b appends to shard b & (shards - 1))

hipError_t sort_temp_bytes(int64_t total, int key_bits, size_t *bytes);
hipError_t launch_gather_sorted(const GatherArgs &g, hipStream_t stream);

}  // namespace pfmscan
