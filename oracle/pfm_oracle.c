/*
 * pfm_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the rnascan per-position log-odds scoring path.
 * It is the checker that tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg compare the HIP path against.  Nothing under rnascan_amd/
 * may import, link or call it: the product path is the HIP library only.
 *
 * Every function cites the reference lines it restates (paths relative to the
 * upstream checkout, morrislab/rnascan v0.10.2).
 *
 * Parity pinning (see DESIGN.md "Oracle"):
 *   - oracle_pwm_calculate        pinned against the reference's own _pwm.c
 *                                 compiled where it lies (oracle/_ref) and the
 *                                 committed goldens in tests/golden/.
 *   - oracle_py_calculate,
 *     oracle_scan_averaged_structure
 *                                 pinned against goldens captured by running
 *                                 the reference's Python functions unmodified
 *                                 (tests/golden/make_golden.py).
 *   - Biopython's normalize/log_odds/search are NOT in the reference tree
 *     (setup.py:68 "biopython >= 1.66", un-vendored): parity unpinned for them.
 *
 * Build: see oracle/Makefile (gcc -O2 -fopenmp, no fast-math: NaN/inf rules matter).
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stddef.h>

/* ------------------------------------------------------------------------- */
/* A7  rnascan/BioAddons/motifs/_pwm.c:7-70 (loop :34-68)                     */
/*   n = s-m+1 windows; per window score=0.0 (double);                        */
/*   for j<m: switch(seq[i+j]) A/a->col0 C/c->col1 G/g->col2 T/t/U/u->col3,   */
/*   anything else -> ok=0 WITHOUT break; store ok ? (float)score : NaN.      */
/*   matrix is double[m][4], C-contiguous (PyArray_GETPTR2(array,j,c)).       */
/* ------------------------------------------------------------------------- */
void oracle_pwm_calculate(const char *sequence, int64_t s, const double *matrix,
                          int64_t m, float *out)
{
    int64_t n = s - m + 1;
    float fnan = 0.0f;
    fnan /= fnan;                       /* _pwm.c:18-19 */
    for (int64_t i = 0; i < n; i++) {
        double score = 0.0;
        int ok = 1;
        for (int64_t j = 0; j < m; j++) {
            char c = sequence[i + j];
            switch (c) {
            case 'A': case 'a': score += matrix[j * 4 + 0]; break;
            case 'C': case 'c': score += matrix[j * 4 + 1]; break;
            case 'G': case 'g': score += matrix[j * 4 + 2]; break;
            case 'T': case 't':
            case 'U': case 'u': score += matrix[j * 4 + 3]; break;
            default: ok = 0;
            }
        }
        out[i] = ok ? (float)score : fnan;  /* _pwm.c:65-66 */
    }
}

/* ------------------------------------------------------------------------- */
/* A8  rnascan/BioAddons/motifs/matrix.py:25-43  (_py_calculate)              */
/*   sequence.upper(); per window score=0.0; score += self[letter][position]; */
/*   KeyError -> score = NaN and BREAK.  Python floats = fp64, no f32 cast.   */
/*   `letters` gives the column order of table[m][nl].                        */
/* ------------------------------------------------------------------------- */
void oracle_py_calculate(const char *sequence, int64_t s, const char *letters,
                         int nl, const double *table, int64_t m, double *out)
{
    int64_t n = s - m + 1;
    for (int64_t i = 0; i < n; i++) {
        double score = 0.0;
        for (int64_t pos = 0; pos < m; pos++) {
            char c = sequence[i + pos];
            if (c >= 'a' && c <= 'z') c = (char)(c - 'a' + 'A');   /* .upper() */
            int col = -1;
            for (int k = 0; k < nl; k++)
                if (letters[k] == c) { col = k; break; }
            if (col < 0) { score = NAN; break; }                   /* KeyError */
            score += table[pos * nl + col];
        }
        out[i] = score;
    }
}

/* numpy.nan_to_num defaults: nan->0.0, +inf->DBL_MAX, -inf->-DBL_MAX */
static inline double nan_to_num(double x)
{
    if (x != x) return 0.0;
    if (x > DBL_MAX) return DBL_MAX;
    if (x < -DBL_MAX) return -DBL_MAX;
    return x;
}

/* ------------------------------------------------------------------------- */
/* A9  rnascan/rnascan.py:302-307  (scan_averaged_structure inner loops)      */
/*   for i in windows: score = 0                                              */
/*     for j<N: score += nan_to_num(dot(struct.iloc[i+j,:], pm.iloc[j,:]))    */
/*   profile is L x ncol fp64 row-major, pssm is N x ncol fp64 row-major,     */
/*   columns already paired (caller decides positional vs label-aligned).     */
/*   All L-N+1 scores are written; the caller applies `score > minscore`      */
/*   (rnascan.py:310, strict).                                                */
/*   np.dot on 7 terms is a plain sum of products; summation order inside     */
/*   BLAS ddot is unspecified at the 1e-16 level, we take k ascending.        */
/* ------------------------------------------------------------------------- */
void oracle_scan_averaged_structure(const double *profile, int64_t L, int ncol,
                                    const double *pssm, int64_t N, double *out)
{
    for (int64_t i = 0; i + N <= L; i++) {
        double score = 0.0;
        for (int64_t j = 0; j < N; j++) {
            const double *row = profile + (i + j) * ncol;
            const double *pj = pssm + j * ncol;
            double d = 0.0;
            for (int k = 0; k < ncol; k++) d += row[k] * pj[k];
            score += nan_to_num(d);
        }
        out[i] = score;
    }
}

/* ========================================================================= */
/* Stream forms: same arithmetic as A7/A9 on the packed record stream the     */
/* C-ABI (include/pfmscan.h) defines, so large seeded inputs can be checked.  */
/*   codes[n]    uint8, 0..7; each record is followed by one separator        */
/*               position holding PFMSCAN_SEP (7).                            */
/*   table[m][8] fp64 per-code log-odds, NaN in every column that is not a    */
/*               letter (so foreign letters and separators poison a window    */
/*               exactly like `ok=0` in _pwm.c:61-62).                        */
/*   profile     fp32 [n][7] row-major (fp64 variant below), exact in fp64.   */
/*   out[p]      score of the window starting at stream position p;           */
/*               p+m>n -> NaN.                                                */
/* ========================================================================= */
void oracle_stream_seq(const uint8_t *codes, int64_t n, const double *table,
                       int m, float *out)
{
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < n; p++) {
        if (p + m > n) { out[p] = NAN; continue; }
        double score = 0.0;
        for (int j = 0; j < m; j++) score += table[j * 8 + (codes[p + j] & 7)];
        out[p] = (float)score;
    }
}

/* generic-alphabet letter scan (A8) on the stream: fp64 out, no f32 cast */
void oracle_stream_letters_f64(const uint8_t *codes, int64_t n,
                               const double *table, int m, double *out)
{
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < n; p++) {
        if (p + m > n) { out[p] = NAN; continue; }
        double score = 0.0;
        for (int j = 0; j < m; j++) score += table[j * 8 + (codes[p + j] & 7)];
        out[p] = score;
    }
}

void oracle_stream_struct_f32(const float *profile, int64_t n,
                              const double *pssm, int m, double *out)
{
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < n; p++) {
        if (p + m > n) { out[p] = NAN; continue; }
        double score = 0.0;
        for (int j = 0; j < m; j++) {
            const float *row = profile + (p + j) * 7;
            const double *pj = pssm + j * 7;
            double d = 0.0;
            for (int k = 0; k < 7; k++) d += (double)row[k] * pj[k];
            score += nan_to_num(d);
        }
        out[p] = score;
    }
}

void oracle_stream_struct_f64(const double *profile, int64_t n,
                              const double *pssm, int m, double *out)
{
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < n; p++) {
        if (p + m > n) { out[p] = NAN; continue; }
        double score = 0.0;
        for (int j = 0; j < m; j++) {
            const double *row = profile + (p + j) * 7;
            const double *pj = pssm + j * 7;
            double d = 0.0;
            for (int k = 0; k < 7; k++) d += row[k] * pj[k];
            score += nan_to_num(d);
        }
        out[p] = score;
    }
}

/* combined pass = config 3 (one seq PFM + one struct PFM of the same width) */
void oracle_stream_seqstruct_f32(const uint8_t *codes, const float *profile,
                                 int64_t n, const double *table,
                                 const double *pssm, int m, float *out_seq,
                                 double *out_struct)
{
    oracle_stream_seq(codes, n, table, m, out_seq);
    oracle_stream_struct_f32(profile, n, pssm, m, out_struct);
}

/* hit filter of the combined scan: inner join of two independently           */
/* thresholded tables, rnascan.py:263 (`search`, strict >), :310, :422-423.   */
/* Returns the number of hits; writes at most cap of them in stream order.    */
int64_t oracle_stream_hits(const float *seq, const double *st, int64_t n,
                           double thr_seq, double thr_struct, int64_t cap,
                           int64_t *pos)
{
    int64_t k = 0;
    for (int64_t p = 0; p < n; p++) {
        int pass = 1;
        if (seq) pass = pass && ((double)seq[p] > thr_seq);
        if (st) pass = pass && (st[p] > thr_struct);
        if (pass) {
            if (k < cap) pos[k] = p;
            k++;
        }
    }
    return k;
}

/* torch sets the process-wide OpenMP thread count to the physical core count when it is imported; the cpu_baseline leg
 * of bench.py chooses its own count */
void oracle_set_num_threads(int n)
{
#ifdef _OPENMP
    extern void omp_set_num_threads(int);
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int oracle_num_threads(void)
{
#ifdef _OPENMP
    extern int omp_get_max_threads(void);
    return omp_get_max_threads();
#else
    return 1;
#endif
}
