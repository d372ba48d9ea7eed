/*
 * pfmscan.h -- C ABI of libpfmscan: MI355X (gfx950) sliding-window PFM scanner.
 *
 * This is the drop-in boundary for rnascan's per-position log-odds scoring
 * path.  Every entry point names the reference interface it replaces (paths
 * are relative to the upstream morrislab/rnascan checkout, v0.10.2).  Plain
 * pointers and sizes only; no C++ or torch types.  Loaded with ctypes
 * (rnascan_amd/_lib.py); INTEGRATION.md shows the stub a reference maintainer
 * would add.
 *
 * Conventions
 *   - every function returns a status (0 = PFMSCAN_OK, negative = error class);
 *     the message is read with pfmscan_last_error().  Foreign letters are never
 *     an error: they score NaN, as `ok = 0` does in _pwm.c:61-66.
 *   - the caller owns every buffer it passes; the library never frees caller
 *     memory and never returns memory the caller must free.  Device scratch is
 *     owned by the ctx, PSSM tables by the motif object.
 *   - a ctx is bound to one device and may be used by one host thread at a
 *     time; use one ctx per device / per thread.  No global mutable state: the
 *     host-only entry points (FASTA / profile text / TSV, no ctx) may be called
 *     from any number of threads, their message (pfmscan_last_error(NULL)) is
 *     thread-local.
 *   - positions are 0-based at this level; rnascan's 1-based inclusive
 *     Start/End (rnascan.py:264-271, :311) are made in the table layer.
 *
 * The packed record stream
 *   Records are concatenated into one stream of n_pos positions; every record
 *   is followed by exactly ONE separator position whose code is PFMSCAN_SEP.
 *     codes    uint8 [n_pos]      letter index 0..6, or PFMSCAN_SEP (7) for a
 *                                 separator or any letter outside the alphabet.
 *                                 Generic-alphabet scans (pfmscan_scan_letters_f64_*,
 *                                 pfmscan_hits_letters_f64_*, the second stream of
 *                                 pfmscan_hits_pair_*) read bits 0..2 only: the host may keep
 *                                 the letter's CASE in bit 3 (structure strings are reported as
 *                                 written, rnascan.py:186-197, :272)
 *     profile  float/double [n_pos][7] row-major averaged-structure profile
 *                                 (rnascan.py:296-297 after `del struct['PO']`),
 *                                 columns already paired with the PSSM's; the
 *                                 separator row may hold anything finite
 *   A window is named by the stream position of its first letter.  Windows that
 *   touch a separator (i.e. would cross a record end) or the end of the stream
 *   get a NaN sequence score, so they can never be hits; out arrays are
 *   position-aligned (length n_pos), record r's windows are
 *   [off_r, off_r + L_r - m + 1).  Base pointers must be 16-byte aligned.
 */
#ifndef PFMSCAN_H
#define PFMSCAN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PFMSCAN_ABI_VERSION 8
#define PFMSCAN_NCODE   8      /* columns of a letter table */
#define PFMSCAN_SEP     7      /* separator / foreign-letter code */
#define PFMSCAN_NSTRUCT 7      /* columns of a structure profile / structure PSSM */
#define PFMSCAN_MAX_M   64     /* widest PFM of the tuned kernels and of PFM libraries (pfmscan_library_create) */
#define PFMSCAN_MAX_WIDTH 4096 /* widest PFM accepted by pfmscan_motif_create / pfmscan_pwm_calculate: the reference's loops take
                                  any width (_pwm.c:34-68, rnascan.py:302-307).  Above PFMSCAN_MAX_M: letters-only scans run a
                                  slab-tiled kernel (the table through LDS 64 rows at a time), scans with a structure part the
                                  profile kernel up to 180 rows and a plain one-thread-per-window kernel beyond -- same results */

#define PFMSCAN_OK          0
#define PFMSCAN_E_BADARG   -1  /* NULL / negative / inconsistent argument   -> ValueError */
#define PFMSCAN_E_BADSHAPE -2  /* width out of range, misaligned pointer     -> ValueError (_pwm.c:96-113) */
#define PFMSCAN_E_OOM      -3  /* host or device allocation failed           -> MemoryError (_pwm.c:27-31) */
#define PFMSCAN_E_HIP      -4  /* HIP runtime error (message has the detail) -> RuntimeError */
#define PFMSCAN_E_CAPACITY -5  /* hit buffer too small; *n_hits = required   -> retry */

#define PFMSCAN_PROFILE_NONE 0
#define PFMSCAN_PROFILE_F32  1
#define PFMSCAN_PROFILE_F64  2

typedef struct pfmscan_ctx pfmscan_ctx;
typedef struct pfmscan_motif pfmscan_motif;

int pfmscan_abi_version(void);

/* Context bound to HIP device `device` (>= 0).  Fails with PFMSCAN_E_HIP when no
 * gfx950 device is visible -- there is no CPU fallback behind this ABI. */
int pfmscan_ctx_create(int device, pfmscan_ctx **out);
void pfmscan_ctx_destroy(pfmscan_ctx *ctx);
/* Last error text of `ctx`; ctx == NULL reads the calling thread's last
 * ctx-less error (a failed pfmscan_ctx_create).  Never NULL. */
const char *pfmscan_last_error(const pfmscan_ctx *ctx);
int pfmscan_device_info(const pfmscan_ctx *ctx, int *n_cu, int64_t *hbm_bytes,
                        char *name, int name_cap);
int pfmscan_synchronize(pfmscan_ctx *ctx);

/* ---- PSSM operands -------------------------------------------------------
 * Replaces the per-call list-of-lists -> ndarray conversion of
 * ExtendedPositionSpecificScoringMatrix._calculate (matrix.py:57-60) and the
 * `pm = pd.DataFrame(pm)` of scan_averaged_structure (rnascan.py:298-300):
 * the tables are uploaded once.
 *   letter_table  double [m][8] or NULL: log-odds per letter code; every column
 *                 that is not a letter of the alphabet (always column 7) must be
 *                 NaN.  RNA: columns A,C,G,U = sorted(alphabet.letters)
 *                 (matrix.py:57), i.e. the column order of _pwm.c:45-60.
 *   struct_pssm   double [m][7] or NULL: log-odds for the 7 profile columns,
 *                 paired with the profile's column order by the caller.
 * At least one must be given; both share the width m (rnascan.py:422-423 joins
 * on Start AND End, so unequal widths can never produce a combined hit). */
int pfmscan_motif_create(pfmscan_ctx *ctx, const double *letter_table,
                         const double *struct_pssm, int m, pfmscan_motif **out);
void pfmscan_motif_destroy(pfmscan_motif *motif);

/* ---- (iii) of SURVEY 8b: the reference's native entry point ----------------
 * Replaces `_pwm.calculate(sequence, matrix)` (_pwm.c:72-121, loop :34-68):
 * ASCII sequence of length s (A/a C/c G/g T/t/U/u; anything else poisons the
 * windows covering it), matrix double [m][4] in A,C,G,U column order ->
 * out float [s-m+1] = (float)(fp64 sequential sum).  s < m writes nothing.
 * Host buffers; runs on the ctx's device. */
int pfmscan_pwm_calculate(pfmscan_ctx *ctx, const char *sequence, int64_t s,
                          const double *matrix, int64_t m, float *out);

/* ---- all-scores over a packed stream, device-resident ----------------------
 * Replaces the window loops of `pssm.search` -> `calculate` (rnascan.py:263,
 * matrix.py:68-81, _pwm.c:34-68) and of scan_averaged_structure
 * (rnascan.py:302-307) for every record of the stream in one pass.
 *   d_codes / d_profile / d_out_* are DEVICE pointers (hipMalloc or torch).
 *   d_out_seq    float  [n_pos] or NULL: (float)(fp64 sequential sum), NaN for
 *                windows covering a foreign letter / separator / stream end
 *   d_out_struct double [n_pos] or NULL: sum_j nan_to_num(dot(profile[p+j], pssm[j]))
 *                in fp64 (rnascan.py:306); NaN for p + m > n_pos
 *   stream       hipStream_t, NULL = the ctx's own stream.  Asynchronous. */
int pfmscan_scan_dev(pfmscan_ctx *ctx, const pfmscan_motif *motif,
                     const uint8_t *d_codes, const void *d_profile,
                     int profile_dtype, int64_t n_pos, float *d_out_seq,
                     double *d_out_struct, void *stream);

/* Generic-alphabet letter scan with fp64 output: replaces
 * ExtendedPositionSpecificScoringMatrix._py_calculate (matrix.py:25-43), which
 * keeps Python floats (no float32 cast) and gives NaN on an unknown letter. */
int pfmscan_scan_letters_f64_dev(pfmscan_ctx *ctx, const pfmscan_motif *motif,
                                 const uint8_t *d_codes, int64_t n_pos,
                                 double *d_out, void *stream);

/* ---- thresholded hits over a packed stream, device-resident ----------------
 * Replaces `score > threshold` of pssm.search (rnascan.py:263; strict, NaN and
 * -inf never pass), `if score > minscore` (rnascan.py:310) and, when the motif
 * has both parts, the inner join of combine() (rnascan.py:422-423):
 *   hit  <=>  (no letter table  or seq(p)    > thr_seq)
 *         and (no struct pssm   or struct(p) > thr_struct)
 * The structure compare is taken on the value the reference's arithmetic gives (rnascan.py:306: every product and
 * every addition rounded, k ascending): a window whose fast score lies within 24 m 2^-53 1024 sum|pssm| of thr_struct
 * is scored again in that order before the compare, and that score is reported (this holds for profile entries up to
 * 1024 in magnitude; they are probabilities).  The same in every hits entry point of this header.
 * Hits are appended in no particular order:
 *   d_hit_pos int64 [capacity], d_hit_seq float [capacity] (or NULL),
 *   d_hit_struct double [capacity] (or NULL), d_hit_count: one uint64 the
 *   caller zeroes before the call; afterwards it holds the TOTAL number of
 *   hits, which may exceed capacity (only the first `capacity` are stored). */
int pfmscan_hits_dev(pfmscan_ctx *ctx, const pfmscan_motif *motif,
                     const uint8_t *d_codes, const void *d_profile,
                     int profile_dtype, int64_t n_pos, double thr_seq,
                     double thr_struct, int64_t capacity, int64_t *d_hit_pos,
                     float *d_hit_seq, double *d_hit_struct,
                     uint64_t *d_hit_count, void *stream);

/* Same contract and arguments as pfmscan_hits_dev, but SYNCHRONISES `stream` and may take
 * two passes: when the motif has both parts and thr_seq is finite, the letters kernel
 * (1 byte per position) runs over everything and the structure score is computed only at
 * its hits (k_struct_at), because a combined hit needs BOTH thresholds (rnascan.py:422-433)
 * and the letter side is selective at real thresholds.  Falls back to the fused pass when
 * more than 1/32 of the windows pass the letter threshold.  Same hits, same scores. */
int pfmscan_hits_adaptive_dev(pfmscan_ctx *ctx, const pfmscan_motif *motif,
                              const uint8_t *d_codes, const void *d_profile,
                              int profile_dtype, int64_t n_pos, double thr_seq,
                              double thr_struct, int64_t capacity, int64_t *d_hit_pos,
                              float *d_hit_seq, double *d_hit_struct,
                              uint64_t *d_hit_count, void *stream);

/* ---- host-buffer forms ------------------------------------------------------
 * Same contracts with HOST pointers: the ctx stages H2D into its own device
 * scratch, launches, copies back and synchronises. */
int pfmscan_scan_host(pfmscan_ctx *ctx, const pfmscan_motif *motif,
                      const uint8_t *codes, const void *profile,
                      int profile_dtype, int64_t n_pos, float *out_seq,
                      double *out_struct);
int pfmscan_scan_letters_f64_host(pfmscan_ctx *ctx, const pfmscan_motif *motif,
                                  const uint8_t *codes, int64_t n_pos, double *out);
/* Hits come back sorted by position.  PFMSCAN_E_CAPACITY: *n_hits holds a capacity
 * that suffices (>= the number of hits), nothing was written to the hit arrays. */
int pfmscan_hits_host(pfmscan_ctx *ctx, const pfmscan_motif *motif,
                      const uint8_t *codes, const void *profile,
                      int profile_dtype, int64_t n_pos, double thr_seq,
                      double thr_struct, int64_t capacity, int64_t *hit_pos,
                      float *hit_seq, double *hit_struct, int64_t *n_hits);

/* Hits of a host-resident stream of ANY length with bounded device scratch (a memory-mapped packed
 * profile store is handed over as is; replaces the per-record `pd.read_table` + scan of rnascan.py:296-310 and the
 * file fan-out of :351-366).  The stream is cut into chunks of chunk_positions (0 = 2^24) positions; the upload of
 * chunk k+1 (copy stream) runs beside the scan of chunk k (two alternating device buffers), every chunk is one fused
 * hits launch, hits come back sorted by stream position.  Same results and same capacity protocol as
 * pfmscan_hits_host; nothing stays staged afterwards. */
int pfmscan_hits_pipeline_host(pfmscan_ctx *ctx, const pfmscan_motif *motif,
                               const uint8_t *codes, const void *profile,
                               int profile_dtype, int64_t n_pos, int64_t chunk_positions,
                               double thr_seq, double thr_struct, int64_t capacity,
                               int64_t *hit_pos, float *hit_seq, double *hit_struct,
                               int64_t *n_hits);

/* ---- staged stream: upload once, scan with many motifs ------------------------
 * (multi-PFM libraries, SURVEY 8f N1: the reference reloads and rescans everything
 * per PFM file).  pfmscan_stage copies a packed stream into the ctx's device scratch
 * and keeps it there; the *_staged calls run one motif over it without any H2D.
 * codes / profile may be NULL when no motif that will be used needs them.  The host
 * forms above are exactly stage + *_staged. */
int pfmscan_stage(pfmscan_ctx *ctx, const uint8_t *codes, const void *profile,
                  int profile_dtype, int64_t n_pos);
/* positions of the stream staged in `ctx` (what pfmscan_scan_staged writes per output array), -1 when nothing is
 * staged.  pfmscan_scan_host / pfmscan_hits_host / pfmscan_pwm_calculate / the pipeline restage or unstage: size the
 * out arrays from THIS, not from what was passed to an earlier pfmscan_stage. */
int64_t pfmscan_staged_positions(const pfmscan_ctx *ctx);
int pfmscan_scan_staged(pfmscan_ctx *ctx, const pfmscan_motif *motif,
                        float *out_seq, double *out_struct);
int pfmscan_hits_staged(pfmscan_ctx *ctx, const pfmscan_motif *motif,
                        double thr_seq, double thr_struct, int64_t capacity,
                        int64_t *hit_pos, float *hit_seq, double *hit_struct,
                        int64_t *n_hits);

/* ---- thresholded hits of a generic-alphabet letter scan, score in fp64 -------------------------
 * (`rnascan -q pfm structs.fa`.)  Replaces `pm.search(seq, threshold=minscore)` (rnascan.py:263) over
 * ExtendedPositionSpecificScoringMatrix._py_calculate (matrix.py:25-43) for an alphabet that is not a nucleotide
 * alphabet: the score is the sequential fp64 sum (Python floats, NO float32 cast), an unknown letter makes the window
 * NaN, hit <=> score > thr (strict: NaN and -inf never pass, not even at thr = -inf).  Letters-only motif, up to 7
 * letters (codes 0..6).  Finite thresholds and widths up to 32 run an integer prefilter with one-sided rounding
 * (k_letters_cred8) and the exact sum for its survivors; otherwise every window gets the exact sum.
 *   d_hit_pos int64 [capacity], d_hit_score double [capacity]; d_hit_count: one uint64 the caller zeroes, afterwards
 *   the TOTAL number of hits (only the first `capacity` are stored).  Hits in no particular order; asynchronous. */
int pfmscan_hits_letters_f64_dev(pfmscan_ctx *ctx, const pfmscan_motif *motif,
                                 const uint8_t *d_codes, int64_t n_pos, double thr,
                                 int64_t capacity, int64_t *d_hit_pos, double *d_hit_score,
                                 uint64_t *d_hit_count, void *stream);
/* Staged (pfmscan_stage with codes) and host-buffer forms: hits sorted by position, capacity protocol of
 * pfmscan_hits_host. */
int pfmscan_hits_letters_f64_staged(pfmscan_ctx *ctx, const pfmscan_motif *motif, double thr,
                                    int64_t capacity, int64_t *hit_pos, double *hit_score,
                                    int64_t *n_hits);
int pfmscan_hits_letters_f64_host(pfmscan_ctx *ctx, const pfmscan_motif *motif,
                                  const uint8_t *codes, int64_t n_pos, double thr,
                                  int64_t capacity, int64_t *hit_pos, double *hit_score,
                                  int64_t *n_hits);

/* ---- two code streams: sequence letters AND structure letters in one call --------------------
 * (`rnascan -p pfm -q pfm seqs.fa structs.fa`, rnascan.py:119-123.)  Replaces the two scan_main passes and the
 * inner join of combine() (rnascan.py:416-434): codes = the RNA stream scored by motif_seq as in pfmscan_hits_dev
 * (float32 of the fp64 sum, _pwm.c:34-68), codes2 = the structure letters of the SAME records in the same layout
 * (same offsets, separators at the same positions) scored by motif_struct as in pfmscan_hits_letters_f64_dev.
 *   hit <=> seq(p) > thr_seq and struct(p) > thr_struct            (both strict)
 * Both motifs are letters-only and share the width m (the join is on Start AND End).  For a finite, selective thr_seq on a
 * PFM up to 32 wide this is ONE launch: the integer-prefiltered letters kernel scores the structure letters of its own
 * survivors (k_letters_cred<.., PAIR>).  Otherwise the sequence letters pass runs over everything and the structure
 * letters are read at its hits only (k_letters_at), which synchronises `stream` in between.
 * Hit arrays and count as in pfmscan_hits_dev (d_hit_struct = the fp64 structure score). */
int pfmscan_hits_pair_dev(pfmscan_ctx *ctx, const pfmscan_motif *motif_seq,
                          const pfmscan_motif *motif_struct, const uint8_t *d_codes,
                          const uint8_t *d_codes2, int64_t n_pos, double thr_seq,
                          double thr_struct, int64_t capacity, int64_t *d_hit_pos,
                          float *d_hit_seq, double *d_hit_struct, uint64_t *d_hit_count,
                          void *stream);
/* A second code stream beside the one staged by pfmscan_stage (same n_pos); forgotten by the next pfmscan_stage. */
int pfmscan_stage_codes2(pfmscan_ctx *ctx, const uint8_t *codes2, int64_t n_pos);
int pfmscan_hits_pair_staged(pfmscan_ctx *ctx, const pfmscan_motif *motif_seq,
                             const pfmscan_motif *motif_struct, double thr_seq, double thr_struct,
                             int64_t capacity, int64_t *hit_pos, float *hit_seq,
                             double *hit_struct, int64_t *n_hits);
int pfmscan_hits_pair_host(pfmscan_ctx *ctx, const pfmscan_motif *motif_seq,
                           const pfmscan_motif *motif_struct, const uint8_t *codes,
                           const uint8_t *codes2, int64_t n_pos, double thr_seq, double thr_struct,
                           int64_t capacity, int64_t *hit_pos, float *hit_seq, double *hit_struct,
                           int64_t *n_hits);

/* ---- multi-PFM libraries: every motif in ONE pass (BASELINE config 5) ----------
 * The reference loads one PFM file per run (load_motif, rnascan.py:217-218), scans only the first
 * motif of its dict (rnascan.py:262) and would re-read every sequence and profile per motif,
 * although it ships a multi-PFM format (pfmutil.py:89-133).  A library object holds n motifs of one
 * width m:
 *   letter_tables  double [n][m][8], each as in pfmscan_motif_create; the alphabet must be the 4 codes
 *                  0..3 (columns 4..7 NaN) -- otherwise PFMSCAN_E_BADSHAPE, scan such motifs one by one
 *   struct_pssms   double [n][m][7] or NULL: motif k's structure PSSM, paired with letter table k
 * Hit of motif k at window p (same filter as pfmscan_hits_dev, per motif):
 *   seq_k(p) > thr_seq[k]  and  (no structure PSSMs or struct_k(p) > thr_struct[k])
 * i.e. pssm.search's strict `>` (rnascan.py:263), `score > minscore` (:310) and combine()'s inner
 * join (:422-423) for the pair (sequence motif k, structure motif k).  Scores are the same numbers the
 * single-motif entry points give (float32 of the sequential fp64 sum; fp64 per-row nan_to_num sum).
 * thr_seq / thr_struct are HOST arrays of n doubles; thr_seq must be > -inf (at -inf every window is a hit:
 * use the all-scores entry points).
 *
 * STRUCTURE-ONLY libraries: letter_tables == NULL with struct_pssms given (`rnascan -q library avgdir/`; the reference
 * would call scan_averaged_structure, rnascan.py:293-315, once per motif and re-read every profile each time).  Every
 * motif is scored in ONE pass over the profile (k_profile_lib): the tile's rows are staged in LDS once, the motifs'
 * PSSM rows stream through the scalar cache, any library size is one pass.  Hit of motif k at window p:
 * struct_k(p) > thr_struct[k] (rnascan.py:310); thr_seq and the codes are ignored (may be NULL), hit_seq comes back NaN.
 * No codes means no separators: the caller drops windows that run over a record end, as for a structure-only motif. */
typedef struct pfmscan_library pfmscan_library;
int pfmscan_library_create(pfmscan_ctx *ctx, const double *letter_tables,
                           const double *struct_pssms, int n_motifs, int m,
                           pfmscan_library **out);
void pfmscan_library_destroy(pfmscan_library *lib);
/* passes the library needs (motifs that fit the 160 KB of LDS at once), the largest one-sided slack of the
 * integer prefilter at the last thresholds in score units (NaN before the first scan, inf when a motif with
 * +inf / NaN log-odds sums runs without prefilter); any pointer may be NULL */
int pfmscan_library_info(const pfmscan_library *lib, int *n_motifs, int *m, int *n_passes,
                         int *motifs_per_pass, double *max_eps);
/* Device-resident form, asynchronous on `stream`.  Hits in no particular order:
 *   d_hit_pos int64 [capacity], d_hit_motif int32 [capacity] (motif index 0..n-1),
 *   d_hit_seq float [capacity] or NULL, d_hit_struct double [capacity] or NULL,
 *   d_hit_count: one uint64 (need not be zeroed); afterwards the TOTAL number of hits.  A value
 *   above capacity means the hit arrays are incomplete: call again with at least that capacity. */
int pfmscan_library_hits_dev(pfmscan_ctx *ctx, pfmscan_library *lib,
                             const uint8_t *d_codes, const void *d_profile,
                             int profile_dtype, int64_t n_pos, const double *thr_seq,
                             const double *thr_struct, int64_t capacity,
                             int64_t *d_hit_pos, int32_t *d_hit_motif, float *d_hit_seq,
                             double *d_hit_struct, uint64_t *d_hit_count, void *stream);
/* Staged / host-buffer forms (see pfmscan_stage): hits come back sorted by (position, motif index).
 * PFMSCAN_E_CAPACITY: *n_hits holds a capacity that suffices, nothing was written. */
int pfmscan_library_hits_staged(pfmscan_ctx *ctx, pfmscan_library *lib,
                                const double *thr_seq, const double *thr_struct,
                                int64_t capacity, int64_t *hit_pos, int32_t *hit_motif,
                                float *hit_seq, double *hit_struct, int64_t *n_hits);
int pfmscan_library_hits_host(pfmscan_ctx *ctx, pfmscan_library *lib,
                              const uint8_t *codes, const void *profile, int profile_dtype,
                              int64_t n_pos, const double *thr_seq, const double *thr_struct,
                              int64_t capacity, int64_t *hit_pos, int32_t *hit_motif,
                              float *hit_seq, double *hit_struct, int64_t *n_hits);

/* pfmscan_hits_pipeline_host for libraries: a host-resident stream of ANY length (a memory-mapped packed profile store and
 * the codes of its records: the whole of config 5's input) with bounded device scratch; chunks of chunk_positions (0 = 2^24)
 * positions, the upload of chunk k+1 beside the scan of chunk k, hits sorted by (position, motif index).  Same results and
 * capacity protocol as pfmscan_library_hits_host; nothing stays staged afterwards. */
int pfmscan_library_hits_pipeline_host(pfmscan_ctx *ctx, pfmscan_library *lib,
                                       const uint8_t *codes, const void *profile, int profile_dtype,
                                       int64_t n_pos, int64_t chunk_positions, const double *thr_seq,
                                       const double *thr_struct, int64_t capacity, int64_t *hit_pos,
                                       int32_t *hit_motif, float *hit_seq, double *hit_struct, int64_t *n_hits);

/* ---- LETTER libraries: the structure side as letter strings (SURVEY 8f N1 x N4) -----------------------------------
 * The reference scans structure FASTA files (`-q pfm structs.fa`, and `-p pfm -q pfm seqs.fa structs.fa`;
 * rnascan.py:119-133) with _py_calculate (matrix.py:25-43: sequential sum of Python floats -- fp64, no float32 cast -- NaN
 * on an unknown letter), one PFM per run (rnascan.py:262), though it ships a multi-PFM format (pfmutil.py:89-133).
 *   struct_tables double [n][m][8]: letter tables of an alphabet of up to 7 letters (codes 0..6; column 7, the foreign
 *   code / separator, must be NaN; unused columns NaN).
 *   seq_tables == NULL: a STRUCTURE-LETTER library over ONE 8-code stream (`rnascan -q library structs.fa`), m <= 32.
 *     Hit of motif k at window p <=> score_k(p) > thr_struct[k] (fp64 compare, strict: NaN and -inf never pass); the
 *     score comes back in hit_struct, hit_seq is NaN / not written; thr_seq is ignored (may be NULL).  One pass per
 *     128 motifs at w = 12: single-letter integer credits for eight motifs per 16-byte table entry (k_library8), exact
 *     fp64 re-score of the survivors.
 *   seq_tables double [n][m][8] (4-letter alphabet): a TWO-FASTA library over two code streams of the same layout (the
 *     sequences and the structure strings of the same records), m <= 64.  Hit of pair k at window p <=>
 *     (double)(float)seq_k(p) > thr_seq[k] on the first stream AND struct_k(p) > thr_struct[k] on the second -- the inner
 *     join of combine() (rnascan.py:416-434).  k_library's prefilter on the sequence side; its survivors get the fp64
 *     letter score of the second stream.
 * Thresholds must be finite on the prefiltered side (thr_struct resp. thr_seq > -inf).  Scores are the numbers
 * pfmscan_hits_letters_f64_* / pfmscan_hits_pair_* give for each motif (pair) alone.  Scanned through
 * pfmscan_library_hits_staged (stage the stream with pfmscan_stage, the second one with pfmscan_stage_codes2) or the
 * two entry points below; d_codes2 / codes2 are NULL for a structure-letter library. */
int pfmscan_library_create_letters(pfmscan_ctx *ctx, const double *seq_tables,
                                   const double *struct_tables, int n_motifs, int m,
                                   pfmscan_library **out);
int pfmscan_library_hits_letters_dev(pfmscan_ctx *ctx, pfmscan_library *lib,
                                     const uint8_t *d_codes, const uint8_t *d_codes2, int64_t n_pos,
                                     const double *thr_seq, const double *thr_struct, int64_t capacity,
                                     int64_t *d_hit_pos, int32_t *d_hit_motif, float *d_hit_seq,
                                     double *d_hit_struct, uint64_t *d_hit_count, void *stream);
int pfmscan_library_hits_letters_host(pfmscan_ctx *ctx, pfmscan_library *lib,
                                      const uint8_t *codes, const uint8_t *codes2, int64_t n_pos,
                                      const double *thr_seq, const double *thr_struct, int64_t capacity,
                                      int64_t *hit_pos, int32_t *hit_motif, float *hit_seq,
                                      double *hit_struct, int64_t *n_hits);

/* Diagnostics (host only, no device needed): the unsigned two-letter credit table the prefilters use for ONE motif
 * (letter_table double [m][8], 4-letter alphabet) at threshold thr_seq: credits uint16 [ceil(m/2)][16], entry index
 * c0 | c1 << 2, with `bits` = 16 (k_letters_cred; k_library for PFMs wider than 16), 10 (k_library up to width 16:
 * three credits per dword, twelve motifs per 16-byte table entry) or 0 (what k_library uses at this width).  A window
 * whose credits sum modulo 2^bits has bit (bits - 1) clear cannot be a hit; *slack = how far below the threshold a kept
 * window's score may lie (score units). */
int pfmscan_debug_credit_table(const double *letter_table, int m, double thr_seq, int bits,
                               uint16_t *credits, double *slack);

/* The same for the FOUR-letter credit table of the single-motif hits kernel (k_letters_quad, PFMs up to width 32):
 * credits uint16 [ceil(m/4)][256], entry index c0 | c1 << 2 | c2 << 4 | c3 << 6 over the letters of motif positions
 * 4t .. 4t+3, 16-bit credits.  A window whose credits sum modulo 2^16 has bit 15 clear cannot be a hit. */
int pfmscan_debug_quad_table(const double *letter_table, int m, double thr_seq, uint16_t *credits, double *slack);
/* ... and the single-letter credits k_library8 uses for ONE motif of a structure-letter library (pfmscan_library_create_letters
 * without sequence tables): credits uint16 [4 ceil(m/4)][8] -- the width padded to a multiple of 4 with rows of full credit --
 * 16-bit sums, bit 15 clear = the window cannot be a hit (matrix.py:25-43 score, strict `>` of rnascan.py:263); m <= 32. */
int pfmscan_debug_library8_credits(const double *letter_table, int m, double thr, uint16_t *credits, double *slack);

/* The same for the SINGLE-letter credit table of the generic-alphabet hits kernel (k_letters_cred8, PFMs up to width 32,
 * letter_table double [m][8] with up to 7 letters): credits uint16 [m][8], entry index = the letter code; NaN and -inf
 * cells and the foreign code get no credit.  *mode = 1: the kernel uses the table; 2: more than 1/32 of the windows would
 * survive it, the exact kernel runs instead; 3: no prefilter possible (+inf cells).  A window whose credits sum modulo 2^16
 * has bit 15 clear cannot be a hit (fp64 score > thr, matrix.py:25-43). */
int pfmscan_debug_credit8_table(const double *letter_table, int m, double thr, uint16_t *credits, int *mode);

/* How the `_host` / `pfmscan_stage` / pipeline entry points move host memory to the device.
 *   PFMSCAN_UPLOAD_RUNTIME (default): hipMemcpyAsync from the caller's pages; the runtime pins them in place, which
 *     reaches the PCIe rate for ordinary (anonymous) memory.
 *   PFMSCAN_UPLOAD_STAGED: transfers of 256 MB and more are cut into 64-MiB pieces that a pool of host threads copies
 *     into pinned buffers while the DMA engine moves the previous piece.  For a source that is a FILE MAPPING (a packed
 *     profile store, rnascan/pfmutil.py:61-87 converted once) the page faults are then taken in parallel instead of
 *     inside the runtime's copy: 32 -> 56 GB/s on a page-cache-warm 4 GB store.
 * The mode stays set until changed. */
#define PFMSCAN_UPLOAD_RUNTIME 0
#define PFMSCAN_UPLOAD_STAGED  1
int pfmscan_set_upload_mode(pfmscan_ctx *ctx, int mode);
/* Tell the staged uploader that the host range [base, base + length) is a read-only MAPPING of `path` starting at byte
 * `file_offset` of that file (what numpy.memmap / mmap give the caller for a packed profile store).  A staged transfer
 * whose source lies inside the range then READS THE FILE (pread into the pinned buffers) instead of touching the mapping:
 * no page faults, no page-table entries to build and to tear down again -- unmapping an 8.4 GB store that was uploaded
 * through its mapping costs up to 0.2 s of munmap at exit on a box whose tmpfs / page cache has no huge pages.  The bytes
 * are the same either way.  length == 0 (path may be NULL) forgets the range: do that BEFORE unmapping it.  At most 8
 * ranges per context; a ninth replaces the oldest. */
int pfmscan_upload_source_file(pfmscan_ctx *ctx, const void *base, size_t length, const char *path, int64_t file_offset);
/* The same, for a caller that recorded fstat's st_dev / st_ino / st_size of the file WHEN IT MAPPED IT: the range is only
 * registered when `path` still names that file (a store re-packed by atomic rename in the meantime would otherwise be
 * read in the mapping's place); PFMSCAN_E_BADARG otherwise, and uploads keep reading the mapping. */
int pfmscan_upload_source_file_checked(pfmscan_ctx *ctx, const void *base, size_t length, const char *path,
                                       int64_t file_offset, int64_t st_dev, int64_t st_ino, int64_t st_size);

/* ---- host ingest and output (no device needed; no context: errors via pfmscan_last_error(NULL)) ---------------
 * The two pieces of host work that dwarf the kernel at scale, in native code.
 *
 * FASTA bytes -> packed code stream.  Replaces SeqIO.parse + preprocess_seq + the per-record encode
 * (rnascan/rnascan.py:170-174, :177-204): a record starts at a line whose first byte is '>', the header is that
 * line without '>' and without trailing \r \n, the letters are every later line stripped of ASCII whitespace at
 * both ends with embedded blanks removed; bytes before the first header are ignored.
 *
 * pfmscan_fasta_index: one pass over `buf` (cut into one piece per thread at line starts, stitched afterwards).
 *   capacity == 0 (arrays may be NULL): only counts the records and
 *   returns PFMSCAN_E_CAPACITY with *n_records set (PFMSCAN_OK when there are none).  Otherwise fills, per record,
 *   hdr_off / hdr_len (header bytes), seq_off / seq_end (byte range of its sequence lines), n_letters.
 * pfmscan_fasta_encode: records [lo, hi) -> codes[sum(n_letters + 1)]: lut256[byte] per letter, `separator`
 *   after each record (stream layout above); offsets[i] = stream position of record lo + i.  n_threads <= 0:
 *   as many as the host offers, at most 16. */
/* pfmscan_count_bytes: counts[256] = how often each byte value occurs in buf (parallel).  Over a packed code stream this
 * is compute_background's letter count (rnascan/rnascan.py:444-457: Seq.count per letter over every record) in one pass:
 * codes 0..6 are the alphabet's letters as written in upper case, 8..14 the lower-case ones, 7 separators / foreign. */
int pfmscan_count_bytes(const uint8_t *buf, int64_t n, int64_t *counts /* [256] */, int n_threads);
/* pfmscan_fasta_lone_cr: *found = 1 when buf holds a carriage return that is not followed by a line feed (old-Mac line
 * ends: the reference's universal-newline reader breaks lines there, pfmscan_fasta_index at \n only -- the caller
 * parses such a file the slow way).  One parallel pass at memory speed over the WHOLE buffer. */
int pfmscan_fasta_lone_cr(const uint8_t *buf, int64_t n, int *found, int n_threads);
int pfmscan_fasta_index(const uint8_t *buf, int64_t n, int64_t capacity, int64_t *hdr_off,
                        int64_t *hdr_len, int64_t *seq_off, int64_t *seq_end, int64_t *n_letters,
                        int64_t *n_records, int n_threads);
 /* pfmscan_fasta_ids: the id of every record = the first whitespace-separated word of its header
  *   (id_off, id_len: byte span in buf); *all_ascii = 0 when some header holds a byte >= 0x80 (the caller then
  *   decodes the headers itself). */
int pfmscan_fasta_ids(const uint8_t *buf, const int64_t *hdr_off, const int64_t *hdr_len,
                      int64_t n_records, int64_t *id_off, int64_t *id_len, int *all_ascii);
 /* pfmscan_gather_spans: the bytes of n_spans (offset, length) spans of buf, each followed by `separator`, back to back
  *   in out (all ids or headers of a file as ONE buffer the caller splits); *n_bytes = the size needed / written. */
int pfmscan_gather_spans(const uint8_t *buf, const int64_t *spans, int64_t n_spans, int separator,
                         uint8_t *out, int64_t capacity, int64_t *n_bytes);
int pfmscan_fasta_encode(const uint8_t *buf, const int64_t *seq_off, const int64_t *seq_end,
                         const int64_t *n_letters, int64_t lo, int64_t hi, const uint8_t *lut256,
                         int separator, uint8_t *codes, int64_t *offsets, int n_threads);

/* Averaged-structure profile text (written by rnascan/pfmutil.py:61-87: a header line, then one row per position:
 * <position> TAB <n_cols numbers>) -> float64 [n_rows][n_cols], the first column dropped, exactly as the reference reads
 * it: `pd.read_table` then `del struct['PO']` (rnascan/rnascan.py:296-297).  pandas' default converter is NOT correctly
 * rounded (at most 17 digits, leading zeros included, then one multiplication or division by a power of ten; about
 * half of all 17-digit reprs come out one ulp off) and the reference computes with those values, so that converter is
 * restated here bit for bit (tests/test_ingest.py compares millions of fields with pandas itself).  The header line is
 * skipped (the caller reads the column letters from it).  PFMSCAN_E_BADSHAPE: something only pandas should judge (blank
 * lines, ragged rows, nan / inf tokens, quotes): parse the file with pandas instead.  PFMSCAN_E_CAPACITY: *n_rows set. */
int pfmscan_profile_parse(const char *buf, int64_t n, int n_cols, int64_t capacity_rows,
                          double *out, int64_t *n_rows);

/* Hit columns -> the bytes `DataFrame.to_csv(sep='\t', index=False)` writes for them (rnascan/rnascan.py:555-567,
 * Match_ID :329-332), without building the table: one descriptor per column, rows formatted in parallel.
 *   CONST    data = bytes, width = their length (the same field in every row)
 *   I64      data = int64 [n_rows]
 *   F32/F64  data = float / double [n_rows]: shortest digits that round-trip in that precision, positional for
 *            1e-4 <= |x| < 1e16 else d.ddde+XX (numpy's str / Python's repr); NaN = empty field, inf = "inf"
 *   INDEXED  data = int64 index [n_rows], aux = int64 offsets [n_values + 1] into blob: row r holds
 *            blob[offsets[i] .. offsets[i + 1]) for i = index[r]  (record ids, descriptions, motif ids; the
 *            caller has applied csv quoting to the values)
 *   FIXED    data = bytes [n_rows][width], trailing NULs dropped (numpy 'S' arrays)
 *   WINDOW   data = int64 stream positions [n_rows], aux = the code stream, blob = 16 letters: the `width` letters
 *            blob[code & 15] of the window at each position (the Sequence column; codes 8..15 = the letters of codes
 *            0..7 as the input wrote them in lower case, see "codes" above)
 *   SPAN     data = int64 index [n_rows], aux = int64 [n_values][2] (offset, length) into blob: row r holds those
 *            bytes of blob (ids / headers straight from the mapped FASTA), csv-quoted here when they hold a tab, a
 *            double quote or a line break
 * first_match_id >= 0 appends a last column counting up from it.  Every row ends in '\n'; no header line.
 * The rows are written by up to PFMSCAN_TSV_MAX_PIECES threads, each into its own slice of `out` (no intermediate
 * buffer, no copy): afterwards piece k is out[pieces[2k] .. pieces[2k] + pieces[2k + 1]) and the table is the
 * pieces in order, k = 0 .. *n_pieces - 1.  `capacity` must cover the longest the rows can get (a bound per numeric
 * column, the real lengths of the INDEXED / SPAN values of the rows at hand); *need holds that size
 * (PFMSCAN_E_CAPACITY when capacity is smaller: nothing was written). */
#define PFMSCAN_TSV_CONST   0
#define PFMSCAN_TSV_I64     1
#define PFMSCAN_TSV_F32     2
#define PFMSCAN_TSV_F64     3
#define PFMSCAN_TSV_INDEXED 4
#define PFMSCAN_TSV_FIXED   5
#define PFMSCAN_TSV_WINDOW  6
#define PFMSCAN_TSV_SPAN    7
#define PFMSCAN_TSV_MAX_PIECES 16
typedef struct pfmscan_tsv_column {
    int32_t kind;
    int32_t reserved;
    const void *data;
    const void *aux;
    const void *blob;
    int64_t width;
} pfmscan_tsv_column;
int pfmscan_tsv_format(const pfmscan_tsv_column *cols, int n_cols, int64_t n_rows,
                       int64_t first_match_id, char *out, int64_t capacity, int64_t *need,
                       int64_t *pieces /* [2 * PFMSCAN_TSV_MAX_PIECES] */, int *n_pieces,
                       int n_threads);

/* Match_ID for rows that were formatted without it (rnascan/rnascan.py:329-332 numbers the rows 1..n AFTER every worker's
 * table has been concatenated; here the ranks of a multi-GPU run format their own rows, and rank 0 numbers them while it
 * relays them in rank order): copies `in` to `out` with "\t<id>" inserted in front of every line end that is not inside a
 * double-quoted field, ids counting up from first_id.  *in_quotes carries the quote state from one block of a stream to the
 * next (0 at the start).  capacity >= n + 21 x (line ends in the block) always suffices; PFMSCAN_E_CAPACITY otherwise. */
int pfmscan_tsv_number(const char *in, int64_t n, int64_t first_id, char *out, int64_t capacity,
                       int64_t *n_out, int64_t *n_rows, int *in_quotes);

/* `round(score, 3)` of rnascan.py:273 for Python floats (the structure letter scores of matrix.py:25-43 are Python
 * floats): out[i] = the double nearest to the decimal with `decimals` (0..15) digits after the point that is nearest
 * to in[i]'s exact value, ties to even -- float.__round__, which numpy.round is not.  NaN / inf pass through. */
int pfmscan_round_decimals(const double *in, int64_t n, int decimals, double *out, int n_threads);

/* ---- measurement helper ------------------------------------------------------
 * Average device time in milliseconds of `iters` back-to-back pfmscan_scan_dev
 * launches, bracketed by hipEvents on the launch stream (after `warmup`
 * untimed launches).  Used by bench.py for the roofline figure. */
int pfmscan_time_scan_dev(pfmscan_ctx *ctx, const pfmscan_motif *motif,
                          const uint8_t *d_codes, const void *d_profile,
                          int profile_dtype, int64_t n_pos, float *d_out_seq,
                          double *d_out_struct, void *stream, int warmup,
                          int iters, double *avg_ms);

/* ---- device arrays of one scan, placed together ------------------------------------
 * No counterpart in the reference (its arrays are numpy arrays in host memory: rnascan.py:296-301, matrix.py:59-62).
 * The arrays one all-scores scan walks in lock-step -- codes, profile rows, the two score arrays -- run up to 12 % faster
 * on MI355X when they lie in parts of HBM that do not share DRAM banks (rnascan_amd/csrc/pfmscan_place.hip).
 * pfmscan_place_alloc allocates n_arrays (1..8) device arrays of bytes[r] each TOGETHER: it takes chunks of physical memory
 * through the HIP virtual memory API, measures which chunks disturb each other, gives every array the chunks that disturb the
 * chunks the other arrays use at the same fraction of the pass least, and returns ptrs[r] (contiguous, 2 MB aligned, contents
 * undefined, usable like any device pointer with every *_dev entry point).  PFMSCAN_PLACE_PLAIN in `flags` (or in the
 * environment) skips the measurement.  pfmscan_place_note: one line about the last allocation (candidates, measured pair
 * times, weighted disturbance chosen / driver order).  Results of scans never depend on placement.
 *
 * COST IN ADDRESS SPACE, AND WHAT pfmscan_place_free DOES.  On ROCm 7.2 an address range that was unmapped, freed and handed
 * out again for other physical memory read back other bytes than were written (profiles/r4/placement/
 * ab_ranges_reused_corrupt.txt; rnascan_amd/csrc/pfmscan_place.hip has the analysis), so NO range this allocator reserves is
 * ever given back while the context lives:
 *   * a measured allocation reserves (candidates + needed chunks) x chunk size of address space -- ~60 GB for the headline
 *     scan's 12.3 GB (24 candidate chunks of 2 GB + the arrays), at most ~140 GB -- a plain one only the arrays' size;
 *   * pfmscan_place_free(ptrs[0]) synchronises the device and RETIRES the set: it stays mapped, memory included, and the
 *     next pfmscan_place_alloc of the same sizes and flags returns it as it is (no new range, no measurement: an
 *     allocate / scan / free loop costs nothing after its first round).  The two most recently retired sets keep their
 *     memory; an older one is unmapped and its memory released at once (its ranges stay reserved).  pfmscan_place_trim
 *     releases the memory of every retired set now.  pfmscan_ctx_destroy releases all memory;
 *   * a context may reserve 16 TB in all (an eighth of the 47-bit user address space: ~270 measured allocations of the
 *     headline's size with DIFFERENT sizes each time); beyond that pfmscan_place_alloc returns PFMSCAN_E_OOM and says why. */
#define PFMSCAN_PLACE_PLAIN 1
int pfmscan_place_alloc(pfmscan_ctx *ctx, int n_arrays, const int64_t *bytes, void **ptrs, int flags);
int pfmscan_place_free(pfmscan_ctx *ctx, void *first_array);
int pfmscan_place_trim(pfmscan_ctx *ctx);
const char *pfmscan_place_note(const pfmscan_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* PFMSCAN_H */
