// pfmscan_library_api.hip -- C ABI of the multi-PFM library scan (include/pfmscan.h, "library" section): table
// construction for k_library (pfmscan_library.hip), passes, hit packing / sorting.
//
// The prefilter of k_library may only DROP windows that cannot be hits.  For motif k with threshold thr and pair rows
// e_t[idx] (t < npair; exact fp64 sums of two letter log-odds), with hi_t = max finite e_t:
//     a window is a hit iff (double)(float)S64 > thr, S64 the sequential fp64 sum of its entries (_pwm.c:34-68);
//     |S64 - S| and the float32 rounding stay below delta = 2^-23 * sum_t max|e_t| + 1e-9 (S = the real-number sum),
//     so a hit has S > thr' = thr - delta, i.e. its DEFICIT sum_t (hi_t - e_t) < D = sum_t hi_t - thr'.
// Deficits are quantised DOWN to v_t = floor(min(hi_t - e_t, D) / q), q = D / V, and stored as credits w_t = V - v_t
// (unsigned, B = 16 or 10 bits).  hit => sum_t v_t <= sum_t (hi_t - e_t)/q < V  =>  sum_t w_t >= X = (npair - 1) V + 1;
// an entry with deficit >= D (-inf cells included) alone puts the sum at most at X - 1.  Pair row 0 also carries
// H - X with H = 2^(B-1), so "may be a hit" is bit B-1 of the B-bit sum.  V = (H - 1) / max(npair - 1, 1) keeps every sum
// below 2^B (two 16-bit or three 10-bit credits share a 32-bit word, a carry would corrupt the neighbour):
// H - X + npair V = H - 1 + V.  k_library uses B = 10 (twelve motifs per 16-byte table entry) for PFMs of up to 8 pair
// rows, where V >= 73 levels per row still leave the slack at a few per cent of D, and B = 16 (eight per entry) beyond.
// Integer adds are exact; the only slack is the rounding of the deficits (< npair * q in the score, one-sided).
// A motif with +inf / NaN two-letter sums (background 0 for a letter the PFM uses) gets no prefilter: its row 0
// is 32768 everywhere, every window goes to the exact pass.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "pfmscan_ctx.hpp"

using namespace pfmscan;

namespace {

struct LibPass {
    int motif_base = 0, n_real = 0, nmp = 0, ng = 0;      // ng: the group count its tables are laid out for (a kernel instantiation)
    int ng_real = 0;           // groups that hold motifs (<= ng): the kernel skips the rest
    size_t pairs_off = 0;      // uint16 elements into d_pairs
    size_t letters_off = 0;    // doubles into d_letters
    size_t pssm_off = 0;       // doubles into d_pssm
    size_t thr_off = 0;        // doubles into d_thr: [nmp] seq thresholds, then [nmp] structure thresholds
};

// Credits of ONE motif at threshold thr (see the header comment): pairsum [npair][16], out [npair][16].
// Returns the one-sided slack of the prefilter in score units (0 when no window can pass, inf without prefilter).
}  // namespace

double pfmscan::build_credits(const double *pairsum, int npair, double thr, uint16_t *out, int bits, int nent)
{
    // (written for rows of 16 two-letter sums; `nent` = 256 makes the same rows of FOUR-letter sums: k_letters_quad)
    std::fill(out, out + (size_t)npair * nent, (uint16_t)0);
    if (thr == INFINITY) return 0.0;                      // nothing exceeds +inf: all credits 0, the flag bit never set
    const int half = 1 << (bits - 1);                     // the flag bit of a credit sum: 32768 (16-bit) or 512 (10-bit credits)
    const int V = std::min(half - 1, (half - 1) / std::max(npair - 1, 1));
    const int X = (npair - 1) * V + 1;
    double sum_abs = 0.0, sum_hi = 0.0;
    std::vector<double> HI((size_t)npair, 0.0);
    bool special = false;
    for (int t = 0; t < npair; ++t) {
        double mx = 0.0, hi = -INFINITY;
        for (int i = 0; i < nent; ++i) {
            const double v = pairsum[t * nent + i];
            if (std::isfinite(v)) {
                mx = std::max(mx, std::fabs(v));
                hi = std::max(hi, v);
            } else if (!(v == -INFINITY)) {
                special = true;                           // +inf or NaN
            }
        }
        HI[t] = std::isfinite(hi) ? hi : 0.0;
        sum_abs += mx;
        sum_hi += HI[t];
    }
    if (special) {                                        // every window goes to the exact pass
        for (int i = 0; i < nent; ++i) out[i] = (uint16_t)half;
        return INFINITY;
    }
    for (int i = 0; i < nent; ++i) out[i] = (uint16_t)(half - X);
    const double delta = 0x1p-23 * sum_abs + 1e-9;
    double D = sum_hi - (thr - delta);
    D += 1e-12 * (std::fabs(D) + std::fabs(sum_hi) + std::fabs(thr)) + 1e-300;       // the fp64 evaluation of D itself
    if (!(D > 0.0)) return 0.0;                           // no window can reach the threshold: credits stay 0
    const double q = D / V;
    for (int t = 0; t < npair; ++t)
        for (int i = 0; i < nent; ++i) {
            const double e = pairsum[t * nent + i];
            int w = 0;
            if (e > -INFINITY) {
                double deficit = HI[t] - e;
                deficit -= 1e-12 * (std::fabs(HI[t]) + std::fabs(e));                 // rounded towards keeping the window
                const double v = std::floor(std::max(deficit, 0.0) / q * (1.0 - 0x1p-40));
                w = v >= (double)V ? 0 : V - (int)v;
            }
            out[t * nent + i] = (uint16_t)(out[t * nent + i] + w);
        }
    return q * npair;
}

// exact two-letter sums of one letter table [m][8] -> [npair][16], index c0 | c1 << 2 (an odd width's last pair
// ignores its second letter)
void pfmscan::pair_sums(const double *T, int m, double *out)
{
    const int npair = (m + 1) / 2;
    for (int t = 0; t < npair; ++t)
        for (int c0 = 0; c0 < 4; ++c0)
            for (int c1 = 0; c1 < 4; ++c1)
                out[t * 16 + (c0 | c1 << 2)] = T[(2 * t) * 8 + c0] + (2 * t + 1 < m ? T[(2 * t + 1) * 8 + c1] : 0.0);
}

// exact four-letter sums of one letter table [m][8] -> [ceil(m/4)][256], index c0 | c1 << 2 | c2 << 4 | c3 << 6 (positions
// beyond the width add nothing)
void pfmscan::quad_sums(const double *T, int m, double *out)
{
    const int nq = (m + 3) / 4;
    for (int t = 0; t < nq; ++t)
        for (int idx = 0; idx < 256; ++idx) {
            double v = 0.0;
            for (int i = 0; i < 4; ++i)
                if (4 * t + i < m) v += T[(4 * t + i) * 8 + ((idx >> (2 * i)) & 3)];
            out[t * 256 + idx] = v;
        }
}

struct pfmscan_library {
    pfmscan_ctx *ctx = nullptr;
    int n = 0, m = 0, npair = 0, np_bucket = 8;
    bool has_struct = false;
    bool has_letters = true;           // false: structure-only library (k_profile_lib, pfmscan_proflib.hip): one pass, no letter tables
    bool pair = false;                 // two-FASTA library: the structure side is [m][8] letter tables over a SECOND code stream (k_library<.., uint8_t>)
    bool letters8 = false;             // generic-alphabet letter library (k_library8): no sequence side, fp64 scores over an 8-code stream
    std::vector<double> rows8;         // letters8: [n][lib8_rows(m)][8] table rows as the credits see them (NaN -> -inf, padding rows 0)
    double *d_pssm_rows = nullptr;     // structure-only: [n][m][7] fp64 as handed in
    int32_t *d_finite = nullptr;       // structure-only: [n] 1 = every cell of the motif's PSSM is finite
    bool all_finite = false;           // seq + struct libraries: every cell of every structure PSSM is finite
    double struct_band = 0.0;          // pfmscan_exact.hpp: re-score band of the thresholded structure compare, max over the motifs
    std::vector<double> pairsum;       // [n][npair][16] exact two-letter sums, index c0 | c1 << 2
    std::vector<LibPass> passes;
    uint16_t *d_pairs = nullptr;
    double *d_letters = nullptr, *d_pssm = nullptr, *d_thr = nullptr;
    size_t pairs_elems = 0, thr_elems = 0;
    std::vector<uint16_t> h_pairs;     // staging of the thresholded credit tables
    std::vector<double> h_thr;
    std::vector<double> cur_seq, cur_struct, eps;   // thresholds the device tables were built for
    bool thr_valid = false;
};

static int lib_fail(pfmscan_ctx *ctx, int code, const std::string &msg) { return fail(ctx, code, msg); }

extern "C" {

// sequence (4-letter) libraries, alone or with a structure side: ncol = 7 -> averaged-structure PSSMs [n][m][7] (k_library over
// the profile), ncol = 8 -> structure LETTER tables [n][m][8] over a second code stream (two-FASTA library)
static int create_seq_library(pfmscan_ctx *ctx, const double *letter_tables, const double *struct_pssms, int ncol, int n_motifs, int m,
                              pfmscan_library **out);

int pfmscan_library_create(pfmscan_ctx *ctx, const double *letter_tables, const double *struct_pssms, int n_motifs, int m,
                           pfmscan_library **out)
{
    if (!ctx || !out || (!letter_tables && !struct_pssms)) return lib_fail(ctx, PFMSCAN_E_BADARG, "pfmscan_library_create: NULL argument");
    *out = nullptr;
    if (n_motifs < 1 || n_motifs > 65535) return lib_fail(ctx, PFMSCAN_E_BADSHAPE, "library size outside 1..65535");
    if (m < 1 || m > PFMSCAN_MAX_M)
        return lib_fail(ctx, PFMSCAN_E_BADSHAPE, "PFM width " + std::to_string(m) + " outside 1.." + std::to_string(PFMSCAN_MAX_M));
    if (!letter_tables) {
        // structure-only library: the PSSMs stay in global memory (k_profile_lib reads them through the scalar cache)
        HIP_TRY(ctx, hipSetDevice(ctx->device));
        pfmscan_library *lib = new (std::nothrow) pfmscan_library();
        if (!lib) return lib_fail(ctx, PFMSCAN_E_OOM, "out of host memory");
        lib->ctx = ctx;
        lib->n = n_motifs;
        lib->m = m;
        lib->has_struct = true;
        lib->has_letters = false;
        for (int k = 0; k < n_motifs; ++k) lib->struct_band = std::max(lib->struct_band, struct_band(struct_pssms + (size_t)k * m * 7, m));
        std::vector<int32_t> fin((size_t)n_motifs, 1);
        for (int k = 0; k < n_motifs; ++k)
            for (int i = 0; i < m * 7; ++i)
                if (!std::isfinite(struct_pssms[(size_t)k * m * 7 + i])) fin[(size_t)k] = 0;
        if (std::getenv("PFMSCAN_FORCE_GENERIC")) std::fill(fin.begin(), fin.end(), 0);
        const size_t cells = (size_t)n_motifs * m * 7;
        hipError_t e = hipMalloc((void **)&lib->d_pssm_rows, cells * 8);
        if (e == hipSuccess) e = hipMemcpy(lib->d_pssm_rows, struct_pssms, cells * 8, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMalloc((void **)&lib->d_finite, (size_t)n_motifs * 4);
        if (e == hipSuccess) e = hipMemcpy(lib->d_finite, fin.data(), (size_t)n_motifs * 4, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMalloc((void **)&lib->d_thr, (size_t)n_motifs * 8);
        lib->thr_elems = (size_t)n_motifs;
        if (e != hipSuccess) {
            pfmscan_library_destroy(lib);
            return fail_hip(ctx, e, "uploading the library tables");
        }
        *out = lib;
        return PFMSCAN_OK;
    }
    return create_seq_library(ctx, letter_tables, struct_pssms, 7, n_motifs, m, out);
}

}  // extern "C"

static int create_seq_library(pfmscan_ctx *ctx, const double *letter_tables, const double *struct_pssms, int ncol, int n_motifs, int m,
                              pfmscan_library **out)
{
    for (int64_t i = 0; i < (int64_t)n_motifs * m; ++i)
        for (int c = 4; c < 8; ++c)
            if (!std::isnan(letter_tables[i * 8 + c]))
                return lib_fail(ctx, PFMSCAN_E_BADSHAPE, "library scans need a 4-letter alphabet: columns 4..7 of every letter table must be NaN");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    pfmscan_library *lib = new (std::nothrow) pfmscan_library();
    if (!lib) return lib_fail(ctx, PFMSCAN_E_OOM, "out of host memory");
    lib->ctx = ctx;
    lib->n = n_motifs;
    lib->m = m;
    lib->npair = (m + 1) / 2;
    lib->np_bucket = lib_np_bucket(m);
    lib->has_struct = struct_pssms != nullptr;
    lib->pair = struct_pssms != nullptr && ncol == 8;
    if (struct_pssms && ncol == 7)
        for (int k = 0; k < n_motifs; ++k) lib->struct_band = std::max(lib->struct_band, struct_band(struct_pssms + (size_t)k * m * 7, m));
    if (struct_pssms && ncol == 7 && !std::getenv("PFMSCAN_FORCE_GENERIC")) {
        lib->all_finite = true;
        for (size_t i = 0; i < (size_t)n_motifs * m * 7 && lib->all_finite; ++i) lib->all_finite = std::isfinite(struct_pssms[i]);
    }
    const int npair = lib->npair;
    lib->pairsum.resize((size_t)n_motifs * npair * 16);
    for (int k = 0; k < n_motifs; ++k) pair_sums(letter_tables + (size_t)k * m * 8, m, lib->pairsum.data() + (size_t)k * npair * 16);
    // passes: as many motif groups as the 160 KB of LDS hold next to the wave queues (the group count of a pass is
    // a template parameter of the kernel, so it comes from a small supported set); full passes first, the rest last
    const int mpg = lib_mpg(lib->np_bucket);
    const size_t per_group = lib_group_bytes(m, npair, lib->has_struct, lib->np_bucket);
    const size_t fixed = lib_queue_bytes(lib->np_bucket);
    const int fit_groups = (int)((160 * 1024 - fixed - 64) / per_group);
    const int ng_max = lib_pick_ng(lib->np_bucket, 1 << 20, fit_groups);
    if (ng_max < 1) {
        delete lib;
        return lib_fail(ctx, PFMSCAN_E_BADSHAPE, "PFM too wide for the library kernel's LDS tables");
    }
    size_t pairs_elems = 0, letters_elems = 0, pssm_elems = 0, thr_elems = 0;
    for (int base = 0; base < n_motifs;) {
        LibPass ps;
        ps.motif_base = base;
        ps.n_real = std::min(n_motifs - base, ng_max * mpg);
        ps.ng_real = (ps.n_real + mpg - 1) / mpg;
        // seq + struct libraries of several passes: every pass in the layout of the full ones, so that
        // the passes can run side by side in ONE launch of one kernel instantiation (lib_run: teams); the groups without
        // motifs are skipped at run time
        const bool uniform = lib->has_struct && n_motifs > ng_max * mpg;
        ps.ng = uniform ? ng_max : lib_pick_ng(lib->np_bucket, ps.ng_real, ng_max);
        ps.nmp = ps.ng * mpg;
        ps.pairs_off = pairs_elems;
        ps.letters_off = letters_elems;
        ps.pssm_off = pssm_elems;
        ps.thr_off = thr_elems;
        pairs_elems += (size_t)ps.ng * npair * 16 * 8;
        letters_elems += (size_t)m * 4 * ps.nmp;
        pssm_elems += (size_t)m * 8 * ps.nmp;
        thr_elems += (size_t)2 * ps.nmp;
        base += ps.n_real;
        lib->passes.push_back(ps);
    }
    lib->pairs_elems = pairs_elems;
    lib->thr_elems = thr_elems;
    // transposed fp64 tables per pass: [m * 4][nmp] letters; [m * 4][nmp][2] structure PSSM (rows padded to 8 columns)
    std::vector<double> hl(letters_elems, 0.0), hp(lib->has_struct ? pssm_elems : 0, 0.0);
    for (const LibPass &ps : lib->passes)
        for (int l = 0; l < ps.n_real; ++l) {
            const int k = ps.motif_base + l;
            for (int j = 0; j < m; ++j) {
                for (int c = 0; c < 4; ++c)
                    hl[ps.letters_off + (size_t)(j * 4 + c) * ps.nmp + l] = letter_tables[((size_t)k * m + j) * 8 + c];
                if (lib->has_struct)
                    for (int c = 0; c < ncol; ++c)
                        hp[ps.pssm_off + (((size_t)(j * 4 + c / 2)) * ps.nmp + l) * 2 + (c & 1)] = struct_pssms[((size_t)k * m + j) * ncol + c];
            }
        }
    hipError_t e = hipMalloc((void **)&lib->d_pairs, pairs_elems * 2);
    if (e == hipSuccess) e = hipMalloc((void **)&lib->d_letters, letters_elems * 8);
    if (e == hipSuccess) e = hipMemcpy(lib->d_letters, hl.data(), letters_elems * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess && lib->has_struct) {
        e = hipMalloc((void **)&lib->d_pssm, pssm_elems * 8);
        if (e == hipSuccess) e = hipMemcpy(lib->d_pssm, hp.data(), pssm_elems * 8, hipMemcpyHostToDevice);
    }
    if (e == hipSuccess) e = hipMalloc((void **)&lib->d_thr, thr_elems * 8);
    if (e != hipSuccess) {
        pfmscan_library_destroy(lib);
        return fail_hip(ctx, e, "uploading the library tables");
    }
    *out = lib;
    return PFMSCAN_OK;
}

// generic-alphabet letter library (k_library8): [n][m][8] tables of up to 7 letters, fp64 scores, m <= 32
static int create_letters8_library(pfmscan_ctx *ctx, const double *tables, int n_motifs, int m, pfmscan_library **out)
{
    if (m > 32) return lib_fail(ctx, PFMSCAN_E_BADSHAPE, "generic-alphabet library scans take PFMs up to 32 wide (scan wider ones one by one)");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    pfmscan_library *lib = new (std::nothrow) pfmscan_library();
    if (!lib) return lib_fail(ctx, PFMSCAN_E_OOM, "out of host memory");
    lib->ctx = ctx;
    lib->n = n_motifs;
    lib->m = m;
    lib->npair = lib8_rows(m);
    lib->np_bucket = m <= 16 ? 16 : 32;
    lib->has_struct = true;             // its scores are reported (and thresholded) on the structure side: fp64
    lib->has_letters = false;
    lib->letters8 = true;
    const int rows = lib->npair;
    lib->rows8.assign((size_t)n_motifs * rows * 8, 0.0);
    for (int k = 0; k < n_motifs; ++k)
        for (int j = 0; j < m; ++j)
            for (int c = 0; c < 8; ++c) {
                const double v = tables[((size_t)k * m + j) * 8 + c];
                // a NaN cell makes the window NaN, which never passes the strict `>` (rnascan.py:263): no credit, like -inf
                lib->rows8[((size_t)k * rows + j) * 8 + c] = std::isnan(v) ? -INFINITY : v;
            }
    const size_t per_group = lib8_group_bytes(m);
    const int fit_groups = (int)((160 * 1024 - lib_queue_bytes(lib->np_bucket) - 64) / per_group);
    const int ng_max = lib8_pick_ng(1 << 20, fit_groups);
    if (ng_max < 1) {
        delete lib;
        return lib_fail(ctx, PFMSCAN_E_BADSHAPE, "PFM too wide for the library kernel's LDS tables");
    }
    size_t pairs_elems = 0, pssm_elems = 0, thr_elems = 0;
    for (int base = 0; base < n_motifs;) {
        LibPass ps;
        ps.motif_base = base;
        ps.n_real = std::min(n_motifs - base, ng_max * 8);
        ps.ng_real = (ps.n_real + 7) / 8;
        ps.ng = lib8_pick_ng(ps.ng_real, ng_max);
        ps.nmp = ps.ng * 8;
        ps.pairs_off = pairs_elems;
        ps.pssm_off = pssm_elems;
        ps.thr_off = thr_elems;
        pairs_elems += (size_t)ps.ng * rows * 8 * 8;          // [row][group][8 codes][8 motifs] u16
        pssm_elems += (size_t)m * 8 * ps.nmp;
        thr_elems += (size_t)2 * ps.nmp;
        base += ps.n_real;
        lib->passes.push_back(ps);
    }
    lib->pairs_elems = pairs_elems;
    lib->thr_elems = thr_elems;
    std::vector<double> hp(pssm_elems, 0.0);                  // [m * 4][nmp][2]: code c of row j at ((j * 4 + c / 2) * nmp + l) * 2 + (c & 1)
    for (const LibPass &ps : lib->passes)
        for (int l = 0; l < ps.n_real; ++l)
            for (int j = 0; j < m; ++j)
                for (int c = 0; c < 8; ++c)
                    hp[ps.pssm_off + (((size_t)(j * 4 + c / 2)) * ps.nmp + l) * 2 + (c & 1)] = tables[((size_t)(ps.motif_base + l) * m + j) * 8 + c];
    hipError_t e = hipMalloc((void **)&lib->d_pairs, pairs_elems * 2);
    if (e == hipSuccess) e = hipMalloc((void **)&lib->d_pssm, pssm_elems * 8);
    if (e == hipSuccess) e = hipMemcpy(lib->d_pssm, hp.data(), pssm_elems * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc((void **)&lib->d_thr, thr_elems * 8);
    if (e != hipSuccess) {
        pfmscan_library_destroy(lib);
        return fail_hip(ctx, e, "uploading the library tables");
    }
    *out = lib;
    return PFMSCAN_OK;
}

extern "C" {

int pfmscan_library_create_letters(pfmscan_ctx *ctx, const double *seq_tables, const double *struct_tables, int n_motifs, int m,
                                   pfmscan_library **out)
{
    if (!ctx || !out || !struct_tables) return lib_fail(ctx, PFMSCAN_E_BADARG, "pfmscan_library_create_letters: NULL argument");
    *out = nullptr;
    if (n_motifs < 1 || n_motifs > 65535) return lib_fail(ctx, PFMSCAN_E_BADSHAPE, "library size outside 1..65535");
    if (m < 1 || m > PFMSCAN_MAX_M)
        return lib_fail(ctx, PFMSCAN_E_BADSHAPE, "PFM width " + std::to_string(m) + " outside 1.." + std::to_string(PFMSCAN_MAX_M));
    for (int64_t i = 0; i < (int64_t)n_motifs * m; ++i)
        if (!std::isnan(struct_tables[i * 8 + 7]))
            return lib_fail(ctx, PFMSCAN_E_BADSHAPE, "column 7 of every letter table must be NaN (the foreign code)");
    if (seq_tables) return create_seq_library(ctx, seq_tables, struct_tables, 8, n_motifs, m, out);
    return create_letters8_library(ctx, struct_tables, n_motifs, m, out);
}

void pfmscan_library_destroy(pfmscan_library *lib)
{
    if (!lib) return;
    if (lib->ctx) (void)hipSetDevice(lib->ctx->device);
    if (lib->d_pairs) (void)hipFree(lib->d_pairs);
    if (lib->d_letters) (void)hipFree(lib->d_letters);
    if (lib->d_pssm) (void)hipFree(lib->d_pssm);
    if (lib->d_thr) (void)hipFree(lib->d_thr);
    if (lib->d_pssm_rows) (void)hipFree(lib->d_pssm_rows);
    if (lib->d_finite) (void)hipFree(lib->d_finite);
    delete lib;
}

int pfmscan_debug_credit_table(const double *letter_table, int m, double thr_seq, int bits, uint16_t *credits, double *slack)
{
    if (!letter_table || !credits || m < 1 || m > PFMSCAN_MAX_M || std::isnan(thr_seq)) return PFMSCAN_E_BADARG;
    if (bits == 0) bits = lib_credit_bits(lib_np_bucket(m));             // what k_library uses at this width
    if (bits != 10 && bits != 16) return PFMSCAN_E_BADARG;
    const int npair = (m + 1) / 2;
    std::vector<double> ps((size_t)npair * 16);
    pair_sums(letter_table, m, ps.data());
    const double s = build_credits(ps.data(), npair, thr_seq, credits, bits);
    if (slack) *slack = s;
    return PFMSCAN_OK;
}

int pfmscan_debug_library8_credits(const double *letter_table, int m, double thr, uint16_t *credits, double *slack)
{
    if (!letter_table || !credits || m < 1 || m > 32 || std::isnan(thr)) return PFMSCAN_E_BADARG;
    const int rows = lib8_rows(m);                        // the table of a letter library: the width padded to a multiple of 4 with full-credit rows
    std::vector<double> r8((size_t)rows * 8, 0.0);
    for (int j = 0; j < m; ++j)
        for (int c = 0; c < 8; ++c) r8[(size_t)j * 8 + c] = std::isnan(letter_table[j * 8 + c]) ? -INFINITY : letter_table[j * 8 + c];
    const double s = build_credits(r8.data(), rows, thr, credits, 16, 8);
    if (slack) *slack = s;
    return PFMSCAN_OK;
}

int pfmscan_debug_quad_table(const double *letter_table, int m, double thr_seq, uint16_t *credits, double *slack)
{
    if (!letter_table || !credits || m < 1 || m > 32 || std::isnan(thr_seq)) return PFMSCAN_E_BADARG;
    const int nq = (m + 3) / 4;
    std::vector<double> qs((size_t)nq * 256);
    quad_sums(letter_table, m, qs.data());
    const double s = build_credits(qs.data(), nq, thr_seq, credits, 16, 256);
    if (slack) *slack = s;
    return PFMSCAN_OK;
}

int pfmscan_library_info(const pfmscan_library *lib, int *n_motifs, int *m, int *n_passes, int *motifs_per_pass, double *max_eps)
{
    if (!lib) return PFMSCAN_E_BADARG;
    if (n_motifs) *n_motifs = lib->n;
    if (m) *m = lib->m;
    const bool tabled = lib->has_letters || lib->letters8;          // passes of LDS tables (a structure-only profile library is one pass whatever its size)
    if (n_passes) *n_passes = tabled ? (int)lib->passes.size() : 1;
    if (motifs_per_pass) *motifs_per_pass = !tabled ? lib->n : (lib->passes.empty() ? 0 : lib->passes[0].nmp);
    if (max_eps) {
        double mx = 0.0;
        for (double v : lib->eps) mx = std::max(mx, v);
        *max_eps = lib->thr_valid ? mx : NAN;
    }
    return PFMSCAN_OK;
}

}  // extern "C"

// (re)build the thresholded credit tables when the thresholds changed; uploads on `st`
static int lib_set_thresholds(pfmscan_ctx *ctx, pfmscan_library *lib, const double *thr_seq, const double *thr_struct, hipStream_t st)
{
    const int n = lib->n, npair = lib->npair;
    if (!lib->has_letters && !lib->letters8) {            // structure-only: the thresholds are the only per-call table
        for (int k = 0; k < n; ++k)
            if (std::isnan(thr_struct[k])) return lib_fail(ctx, PFMSCAN_E_BADARG, "NaN threshold");
        if (lib->thr_valid && std::equal(thr_struct, thr_struct + n, lib->cur_struct.begin())) return PFMSCAN_OK;
        HIP_TRY(ctx, hipStreamSynchronize(st));
        lib->h_thr.assign(thr_struct, thr_struct + n);
        HIP_TRY(ctx, hipMemcpyAsync(lib->d_thr, lib->h_thr.data(), (size_t)n * 8, hipMemcpyHostToDevice, st));
        lib->cur_struct.assign(thr_struct, thr_struct + n);
        lib->eps.assign((size_t)n, 0.0);
        lib->thr_valid = true;
        return PFMSCAN_OK;
    }
    if (lib->letters8) {
        // single-letter credits of every motif at its threshold: rows8 [rows][8] -> 16-bit credits, kernel layout
        // [row][group][8 codes][4 dwords] with motif 2 d + h of the group in half h of dword d
        const int rows = lib->npair;
        for (int k = 0; k < n; ++k) {
            if (std::isnan(thr_struct[k])) return lib_fail(ctx, PFMSCAN_E_BADARG, "NaN threshold");
            if (thr_struct[k] == -INFINITY)
                return lib_fail(ctx, PFMSCAN_E_BADARG, "library hits need a finite threshold (every window would be a hit; use the all-scores entry points)");
        }
        if (lib->thr_valid && std::equal(thr_struct, thr_struct + n, lib->cur_struct.begin())) return PFMSCAN_OK;
        HIP_TRY(ctx, hipStreamSynchronize(st));
        lib->thr_valid = false;
        lib->h_pairs.assign(lib->pairs_elems, (uint16_t)0);
        lib->h_thr.assign(lib->thr_elems, 0.0);
        lib->eps.assign((size_t)n, 0.0);
        std::vector<uint16_t> cr((size_t)rows * 8);
        uint32_t *words = reinterpret_cast<uint32_t *>(lib->h_pairs.data());
        for (const LibPass &ps : lib->passes)
            for (int l = 0; l < ps.nmp; ++l) {
                const int g = l / 8, slot = l % 8;
                if (l >= ps.n_real) {                     // padding motif: all credits 0, never flagged
                    lib->h_thr[ps.thr_off + l] = INFINITY;
                    lib->h_thr[ps.thr_off + ps.nmp + l] = INFINITY;
                    continue;
                }
                const int k = ps.motif_base + l;
                lib->h_thr[ps.thr_off + l] = thr_struct[k];
                lib->h_thr[ps.thr_off + ps.nmp + l] = thr_struct[k];
                lib->eps[k] = build_credits(lib->rows8.data() + (size_t)k * rows * 8, rows, thr_struct[k], cr.data(), 16, 8);
                for (int t = 0; t < rows; ++t)
                    for (int i = 0; i < 8; ++i) {
                        uint32_t *entry = words + ps.pairs_off / 2 + (((size_t)t * ps.ng + g) * 8 + i) * 4;
                        entry[slot / 2] |= (uint32_t)cr[t * 8 + i] << (16 * (slot % 2));
                    }
            }
        HIP_TRY(ctx, hipMemcpyAsync(lib->d_pairs, lib->h_pairs.data(), lib->pairs_elems * 2, hipMemcpyHostToDevice, st));
        HIP_TRY(ctx, hipMemcpyAsync(lib->d_thr, lib->h_thr.data(), lib->thr_elems * 8, hipMemcpyHostToDevice, st));
        lib->cur_struct.assign(thr_struct, thr_struct + n);
        lib->thr_valid = true;
        return PFMSCAN_OK;
    }
    for (int k = 0; k < n; ++k) {
        if (std::isnan(thr_seq[k]) || (lib->has_struct && std::isnan(thr_struct[k]))) return lib_fail(ctx, PFMSCAN_E_BADARG, "NaN threshold");
        if (thr_seq[k] == -INFINITY)
            return lib_fail(ctx, PFMSCAN_E_BADARG, "library hits need a finite sequence threshold (every window would be a hit; use the all-scores entry points)");
    }
    if (lib->thr_valid && std::equal(thr_seq, thr_seq + n, lib->cur_seq.begin()) &&
        (!lib->has_struct || std::equal(thr_struct, thr_struct + n, lib->cur_struct.begin())))
        return PFMSCAN_OK;
    HIP_TRY(ctx, hipStreamSynchronize(st));              // an earlier upload may still read the staging vectors
    lib->thr_valid = false;
    lib->h_pairs.assign(lib->pairs_elems, (uint16_t)0);
    lib->h_thr.assign(lib->thr_elems, 0.0);
    lib->eps.assign((size_t)n, 0.0);
    std::vector<uint16_t> cr((size_t)npair * 16);
    const int mpg = lib_mpg(lib->np_bucket), bits = lib_credit_bits(lib->np_bucket);
    uint32_t *words = reinterpret_cast<uint32_t *>(lib->h_pairs.data());       // an entry = 4 dwords (16-byte aligned offsets)
    for (const LibPass &ps : lib->passes) {
        for (int l = 0; l < ps.nmp; ++l) {
            const int g = l / mpg, slot = l % mpg;
            if (l >= ps.n_real) {                         // padding motif: all credits 0, never flagged
                lib->h_thr[ps.thr_off + l] = INFINITY;
                lib->h_thr[ps.thr_off + ps.nmp + l] = INFINITY;
                continue;
            }
            const int k = ps.motif_base + l;
            lib->h_thr[ps.thr_off + l] = thr_seq[k];
            lib->h_thr[ps.thr_off + ps.nmp + l] = lib->has_struct ? thr_struct[k] : -INFINITY;
            lib->eps[k] = build_credits(lib->pairsum.data() + (size_t)k * npair * 16, npair, thr_seq[k], cr.data(), bits);
            for (int t = 0; t < npair; ++t)               // kernel layout [pair row][group][entry][4 dwords]
                for (int i = 0; i < 16; ++i) {
                    uint32_t *entry = words + ps.pairs_off / 2 + (((size_t)t * ps.ng + g) * 16 + i) * 4;
                    if (mpg == 12)                        // motif 3 d + f of the group: bits 10 f .. 10 f + 9 of dword d
                        entry[slot / 3] |= (uint32_t)cr[t * 16 + i] << (10 * (slot % 3));
                    else                                  // motif 2 d + h: half h of dword d
                        entry[slot / 2] |= (uint32_t)cr[t * 16 + i] << (16 * (slot % 2));
                }
        }
    }
    HIP_TRY(ctx, hipMemcpyAsync(lib->d_pairs, lib->h_pairs.data(), lib->pairs_elems * 2, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(lib->d_thr, lib->h_thr.data(), lib->thr_elems * 8, hipMemcpyHostToDevice, st));
    lib->cur_seq.assign(thr_seq, thr_seq + n);
    if (lib->has_struct) lib->cur_struct.assign(thr_struct, thr_struct + n);
    lib->thr_valid = true;
    return PFMSCAN_OK;
}

constexpr int64_t LIB_SEG = (int64_t)1 << LIB_SEG_SHIFT;        // windows per work segment (segment s -> workgroup s mod grid, shard s mod 256)

static int64_t lib_work_unit(const pfmscan_library *lib);

struct LibSink {
    int64_t *pos;
    int32_t *motif;
    float *seq;
    double *st;
    unsigned long long *count;
    int shards;
    int64_t shard_cap;
};

// every pass of the library over [0, n_pos); asynchronous on `st`
static int lib_run(pfmscan_ctx *ctx, pfmscan_library *lib, const uint8_t *d_codes, const void *d_profile, int profile_dtype,
                   int64_t n_pos, const LibSink &sink, hipStream_t st, int64_t pos_offset = 0)
{
    if (lib->letters8) {
        // generic-alphabet letter library: its passes one after the other (each re-reads only the 1-byte codes)
        const int64_t max_span = (int64_t)1 << 31;
        for (int64_t base = 0; base < n_pos; base += max_span)
            for (const LibPass &ps : lib->passes) {
                LibArgs a;
                std::memset(&a, 0, sizeof(a));
                a.codes = d_codes;
                a.n_pos = n_pos;
                a.pos_base = base;
                a.span = std::min<int64_t>(max_span, n_pos - base);
                a.seg_positions = LIB_SEG;
                a.n_seg = (a.span + a.seg_positions - 1) / a.seg_positions;
                a.pairs = reinterpret_cast<const uint32_t *>(lib->d_pairs + ps.pairs_off);
                a.pssm = lib->d_pssm + ps.pssm_off;
                a.thr_seq = lib->d_thr + ps.thr_off;
                a.thr_struct = lib->d_thr + ps.thr_off + ps.nmp;
                a.m = lib->m;
                a.npair = lib->npair;
                a.nmp = ps.nmp;
                a.ng = ps.ng;
                a.ng_real = ps.ng_real;
                a.motif_base = ps.motif_base;
                a.pos_offset = pos_offset;
                a.shard_cap = sink.shard_cap;
                a.hit_shards = sink.shards;
                a.hit_pos = sink.pos;
                a.hit_motif = sink.motif;
                a.hit_seq = nullptr;
                a.hit_struct = sink.st;
                a.hit_count = sink.count;
                hipError_t e = launch_library8(a, ctx->n_cu, st);
                if (e != hipSuccess) return fail_hip(ctx, e, "launch k_library8");
            }
        return PFMSCAN_OK;
    }
    if (!lib->has_letters) {
        ProfLibArgs a;
        std::memset(&a, 0, sizeof(a));
        a.profile = d_profile;
        a.profile_dtype = profile_dtype;
        a.n_pos = n_pos;
        a.pssm = lib->d_pssm_rows;
        a.thr = lib->d_thr;
        a.struct_band = lib->struct_band;
        a.finite = lib->d_finite;
        a.n_motifs = lib->n;
        a.m = lib->m;
        a.motif_base = 0;
        a.pos_offset = pos_offset;
        a.shard_cap = sink.shard_cap;
        a.hit_shards = sink.shards;
        a.hit_pos = sink.pos;
        a.hit_motif = sink.motif;
        a.hit_struct = sink.st;
        a.hit_count = sink.count;
        hipError_t e = launch_profile_library(a, st);
        if (e != hipSuccess) return fail_hip(ctx, e, "launch k_profile_lib");
        return PFMSCAN_OK;
    }
    // Teams: up to four consecutive passes SIDE BY SIDE in one launch, each on a share of the workgroups, all walking the
    // stream at the same pace.  The structure verification reads ~every line of the profile through scattered loads; one
    // pass after the other that is the whole profile from HBM per pass (3 x 8.4 GB on C5), side by side the later readers
    // of a line find it in the memory-side cache (a diagnostic build whose rows come from a cache-resident region: C5
    // -13 %, float64 rows -37 %: the upper bound).  Streams were tried first and dropped: concurrent kernels need a free
    // hardware queue each and workgroup counts that are multiples of 8 per kernel (a grid is dealt round-robin to the 8
    // XCDs and one workgroup too many on an XCD waits for a whole persistent kernel) -- 29 ms instead of 10.
    const LibPass &first_pass = lib->passes[0];
    bool teams = lib->has_struct && !lib->pair && lib->passes.size() > 1 && n_pos >= (int64_t)8 * ctx->n_cu * LIB_SEG &&
                 ctx->n_cu >= 16 && !std::getenv("PFMSCAN_LIB_SEQUENTIAL");
    for (const LibPass &ps : lib->passes) teams = teams && ps.ng == first_pass.ng;      // the common layout (pfmscan_library_create)
    const int64_t max_span = (int64_t)1 << 31;
    for (int64_t base = 0; base < n_pos; base += max_span) {
        // a team launch needs the whole grid (a short last span of a very long stream runs its passes one by one)
        const int64_t span_segs = (std::min<int64_t>(max_span, n_pos - base) + LIB_SEG - 1) / LIB_SEG;
        const size_t per_launch = teams && span_segs >= ctx->n_cu ? 4 : 1;
        for (size_t p0 = 0; p0 < lib->passes.size(); p0 += per_launch) {
            const LibPass &ps = lib->passes[p0];
            const size_t nt = std::min(per_launch, lib->passes.size() - p0);
            LibArgs a;
            std::memset(&a, 0, sizeof(a));
            a.ng_real = ps.ng_real;
            {
                static const int sort_env = std::getenv("PFMSCAN_LIB_SORT") ? std::atoi(std::getenv("PFMSCAN_LIB_SORT")) : 0;
                a.sort_batches = sort_env;
            }
            if (nt > 1) {
                // shares of the grid ~ the cost of a pass: measured 0.385 ms per motif group + 0.66 ms on C5 with float32 rows
                // (flat around it: 0 .. 1.5 for the constant give the same time within the noise); with float64 rows the
                // part that does not scale with the groups is larger (88 / 88 / 80 workgroups beat 92 / 92 / 72 by 4 %)
                double w[4] = {0, 0, 0, 0}, wsum = 0;
                const double fixed = profile_dtype == PFMSCAN_PROFILE_F64 ? 3.0 : 0.66;
                for (size_t t = 0; t < nt; ++t) wsum += (w[t] = 0.385 * lib->passes[p0 + t].ng_real + fixed);
                a.n_teams = (int)nt;
                int placed = 0;
                for (size_t t = 0; t < nt; ++t) {
                    a.team_first[t] = placed;
                    a.team_ng[t] = lib->passes[p0 + t].ng_real;
                    const int room = ctx->n_cu - placed - (int)(nt - 1 - t);       // every later team keeps at least one workgroup
                    const int share = t + 1 == nt ? ctx->n_cu - placed : std::min(room, std::max(1, (int)std::lround(ctx->n_cu * w[t] / wsum)));
                    placed += share;
                }
                for (size_t t = nt; t < 5; ++t) a.team_first[t] = ctx->n_cu;
                a.stride_pairs = (int64_t)(lib->passes[p0 + 1].pairs_off - ps.pairs_off) / 2;        // uint16 elements -> dwords
                a.stride_letters = (int64_t)(lib->passes[p0 + 1].letters_off - ps.letters_off);
                a.stride_pssm = (int64_t)(lib->passes[p0 + 1].pssm_off - ps.pssm_off);
                a.stride_thr = (int64_t)(lib->passes[p0 + 1].thr_off - ps.thr_off);
            }
            a.codes = d_codes;
            a.profile = lib->has_struct ? d_profile : nullptr;       // (two-FASTA library: the second code stream)
            a.profile_dtype = lib->pair ? PROFILE_LETTERS2 : profile_dtype;
            a.n_pos = n_pos;
            a.pos_base = base;
            a.span = std::min<int64_t>(max_span, n_pos - base);
            a.seg_positions = LIB_SEG;
            a.n_seg = (a.span + a.seg_positions - 1) / a.seg_positions;
            a.pairs = reinterpret_cast<const uint32_t *>(lib->d_pairs + ps.pairs_off);
            a.letters = lib->d_letters + ps.letters_off;
            a.pssm = lib->has_struct ? lib->d_pssm + ps.pssm_off : nullptr;
            a.thr_seq = lib->d_thr + ps.thr_off;
            a.thr_struct = lib->d_thr + ps.thr_off + ps.nmp;
            a.struct_finite = lib->all_finite ? 1 : 0;
            a.struct_band = lib->struct_band;
            a.m = lib->m;
            a.npair = lib->npair;
            a.nmp = ps.nmp;
            a.ng = ps.ng;
            a.motif_base = ps.motif_base;
            a.pos_offset = pos_offset;
            a.shard_cap = sink.shard_cap;
            a.hit_shards = sink.shards;
            a.hit_pos = sink.pos;
            a.hit_motif = sink.motif;
            a.hit_seq = sink.seq;
            a.hit_struct = sink.st;
            a.hit_count = sink.count;
            hipError_t e = launch_library(a, ctx->n_cu, st);
            if (e != hipSuccess) return fail_hip(ctx, e, "launch k_library");
        }
    }
    return PFMSCAN_OK;
}

// positions one workgroup-visit covers: sizes the shards of the hit buffers
static int64_t lib_work_unit(const pfmscan_library *lib) { return (lib->has_letters || lib->letters8) ? LIB_SEG : profile_library_tile(); }

static int lib_check(pfmscan_ctx *ctx, const pfmscan_library *lib, const uint8_t *codes, const void *profile, int profile_dtype,
                     int64_t n_pos, const double *thr_seq, const double *thr_struct)
{
    if (!ctx || !lib) return lib_fail(ctx, PFMSCAN_E_BADARG, "NULL ctx or library");
    if (lib->ctx != ctx) return lib_fail(ctx, PFMSCAN_E_BADARG, "library belongs to another ctx");
    if (n_pos < 0) return lib_fail(ctx, PFMSCAN_E_BADARG, "negative n_pos");
    if ((lib->has_letters && !thr_seq) || (lib->has_struct && !thr_struct)) return lib_fail(ctx, PFMSCAN_E_BADARG, "threshold arrays are NULL");
    if (n_pos > 0 && lib->has_letters && !codes) return lib_fail(ctx, PFMSCAN_E_BADARG, "codes is NULL");
    if (lib->letters8) {
        if (n_pos > 0 && !codes) return lib_fail(ctx, PFMSCAN_E_BADARG, "codes is NULL");
        return PFMSCAN_OK;
    }
    if (lib->pair) {
        if (n_pos > 0 && !profile) return lib_fail(ctx, PFMSCAN_E_BADARG, "two-FASTA library: the second code stream is NULL");
        return PFMSCAN_OK;
    }
    if (lib->has_struct) {
        if (profile_dtype != PFMSCAN_PROFILE_F32 && profile_dtype != PFMSCAN_PROFILE_F64)
            return lib_fail(ctx, PFMSCAN_E_BADARG, "library has structure PSSMs: profile_dtype must be F32 or F64");
        if (n_pos > 0 && !profile) return lib_fail(ctx, PFMSCAN_E_BADARG, "library has structure PSSMs but profile is NULL");
    }
    return PFMSCAN_OK;
}

static int lib_scratch(pfmscan_ctx *ctx, int64_t capacity, int64_t n_pos, LibSink &sink, int64_t work_unit = LIB_SEG)
{
    // shard s = workgroup & 255 gets every 256th 16k-window segment: the shards in use fill evenly, each has room for
    // twice its share (short streams use few shards, small capacities let every shard take everything)
    const int64_t active = std::max<int64_t>(1, std::min<int64_t>(LIB_SHARDS, (n_pos + work_unit - 1) / work_unit));
    const int64_t shard_cap = std::max<int64_t>(std::min<int64_t>(capacity, capacity / active * 2 + 1024), 1);
    const size_t slots = (size_t)shard_cap * LIB_SHARDS;
    int rc;
    if ((rc = ensure(ctx, ctx->lib_pos, slots * 8))) return rc;
    if ((rc = ensure(ctx, ctx->lib_motif, slots * 4))) return rc;
    if ((rc = ensure(ctx, ctx->lib_seq, slots * 4))) return rc;
    if ((rc = ensure(ctx, ctx->lib_struct, slots * 8))) return rc;
    if ((rc = ensure(ctx, ctx->lib_count, (size_t)(LIB_SHARDS + 2) * HIT_COUNTER_STRIDE * 8 + (LIB_SHARDS + 1) * 8))) return rc;
    sink.pos = (int64_t *)ctx->lib_pos.p;
    sink.motif = (int32_t *)ctx->lib_motif.p;
    sink.seq = (float *)ctx->lib_seq.p;
    sink.st = (double *)ctx->lib_struct.p;
    sink.count = (unsigned long long *)ctx->lib_count.p;
    sink.shards = LIB_SHARDS;
    sink.shard_cap = shard_cap;
    return PFMSCAN_OK;
}

// ---- sharded hits -> the caller's contiguous device arrays (order unspecified) ----
namespace pfmscan {

constexpr int PACK_BLOCK = 256;

// starts[s] = exclusive prefix of min(count_s, shard_cap); *out_count = total hits (capacity + 1 at least when a shard
// overflowed although the total fits, so that the caller sees "incomplete" exactly when something was dropped)
__global__ __launch_bounds__(PACK_BLOCK) void k_lib_prefix(const unsigned long long *__restrict__ counts, int shards,
                                                           int64_t shard_cap, int64_t capacity, int64_t *__restrict__ starts,
                                                           unsigned long long *__restrict__ out_count)
{
    __shared__ int64_t held[LIB_SHARDS];
    __shared__ unsigned long long total_s;
    __shared__ int over_s;
    if (threadIdx.x == 0) {
        total_s = 0;
        over_s = 0;
    }
    __syncthreads();
    for (int s = threadIdx.x; s < shards; s += PACK_BLOCK) {
        const unsigned long long n = counts[(size_t)s * HIT_COUNTER_STRIDE];
        held[s] = (int64_t)n < shard_cap ? (int64_t)n : shard_cap;
        atomicAdd(&total_s, n);
        if ((int64_t)n > shard_cap) atomicOr(&over_s, 1);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int64_t run = 0;
        for (int s = 0; s < shards; ++s) {
            starts[s] = run;
            run += held[s];
        }
        starts[shards] = run;
        unsigned long long t = total_s;
        if (over_s && (int64_t)t <= capacity) t = (unsigned long long)capacity + 1;
        *out_count = t;
    }
}

__global__ __launch_bounds__(PACK_BLOCK) void k_lib_pack(const int64_t *__restrict__ starts, int shards, int64_t shard_cap,
                                                         int64_t capacity, const int64_t *__restrict__ s_pos,
                                                         const int32_t *__restrict__ s_motif, const float *__restrict__ s_seq,
                                                         const double *__restrict__ s_st, int64_t *__restrict__ pos,
                                                         int32_t *__restrict__ motif, float *__restrict__ seq, double *__restrict__ st)
{
    __shared__ int64_t start[LIB_SHARDS + 1];
    for (int s = threadIdx.x; s <= shards; s += PACK_BLOCK) start[s] = starts[s];
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * PACK_BLOCK + threadIdx.x;
    if (i >= start[shards] || i >= capacity) return;
    int lo = 0, hi = shards;                   // largest s with start[s] <= i
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (start[mid] <= i) lo = mid; else hi = mid;
    }
    const int64_t src = (int64_t)lo * shard_cap + (i - start[lo]);
    pos[i] = s_pos[src];
    if (motif) motif[i] = s_motif[src];
    if (seq) seq[i] = s_seq ? s_seq[src] : __builtin_nanf("");      // a library without a sequence side reports NaN there
    if (st && s_st) st[i] = s_st[src];
}

}  // namespace pfmscan

// the sharded hits of a library scan -> the caller's host arrays, sorted by (position, motif): capacity check, device
// sort (pfmscan_sort.hip), contiguous copies.  Synchronises ctx->stream.  Positions lie in [0, n_pos).
static int lib_finish_sorted(pfmscan_ctx *ctx, pfmscan_library *lib, int64_t n_pos, int64_t capacity, const LibSink &sink,
                             int64_t *hit_pos, int32_t *hit_motif, float *hit_seq, double *hit_struct, int64_t *n_hits)
{
    int rc;
    hipStream_t st = ctx->stream;
    const size_t counter_bytes = (size_t)LIB_SHARDS * HIT_COUNTER_STRIDE * 8;
    std::vector<unsigned long long> counters((size_t)LIB_SHARDS * HIT_COUNTER_STRIDE);
    HIP_TRY(ctx, hipMemcpyAsync(counters.data(), sink.count, counter_bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    uint64_t total = 0, worst = 0;
    for (int s = 0; s < LIB_SHARDS; ++s) {
        total += counters[(size_t)s * HIT_COUNTER_STRIDE];
        worst = std::max<uint64_t>(worst, counters[(size_t)s * HIT_COUNTER_STRIDE]);
    }
    *n_hits = (int64_t)total;
    if ((int64_t)total > capacity || (int64_t)worst > sink.shard_cap) {
        *n_hits = (int64_t)std::max<uint64_t>(total, worst * LIB_SHARDS);       // enough that every shard fits next time
        return lib_fail(ctx, PFMSCAN_E_CAPACITY, "hit buffer too small: " + std::to_string(total) + " hits, capacity " + std::to_string(capacity));
    }
    if (total == 0) return PFMSCAN_OK;
    // shards -> one run ordered by (position, motif) on the device (pfmscan_sort.hip)
    int key_bits = 1, motif_bits = 1;
    while (key_bits < 62 && ((int64_t)1 << key_bits) < n_pos) ++key_bits;
    while (motif_bits < 16 && (1 << motif_bits) < lib->n) ++motif_bits;
    if (key_bits + motif_bits > 63) return lib_fail(ctx, PFMSCAN_E_BADARG, "stream too long for the (position, motif) sort key");
    size_t temp_bytes = 0;
    HIP_TRY(ctx, sort_temp_bytes((int64_t)total, key_bits + motif_bits, &temp_bytes));
    if ((rc = ensure(ctx, ctx->sort_keys_in, total * 8))) return rc;
    if ((rc = ensure(ctx, ctx->sort_keys_out, total * 8))) return rc;
    if ((rc = ensure(ctx, ctx->sort_vals_in, total * 8))) return rc;
    if ((rc = ensure(ctx, ctx->sort_vals_out, total * 8))) return rc;
    if ((rc = ensure(ctx, ctx->sort_temp, std::max<size_t>(temp_bytes, 256)))) return rc;
    if ((rc = ensure(ctx, ctx->sort_seq, total * 4))) return rc;
    if ((rc = ensure(ctx, ctx->sort_struct, total * 8))) return rc;
    if ((rc = ensure(ctx, ctx->sort_motif, total * 4))) return rc;
    GatherArgs g;
    g.hit_pos = sink.pos;
    g.hit_seq = lib->has_letters ? sink.seq : nullptr;
    g.hit_struct = lib->has_struct ? sink.st : nullptr;
    g.counts = sink.count;
    g.shards = LIB_SHARDS;
    g.shard_cap = sink.shard_cap;
    g.total = (int64_t)total;
    g.key_bits = key_bits;
    g.keys_in = (int64_t *)ctx->sort_keys_in.p;
    g.keys_out = (int64_t *)ctx->sort_keys_out.p;
    g.vals_in = (int64_t *)ctx->sort_vals_in.p;
    g.vals_out = (int64_t *)ctx->sort_vals_out.p;
    g.temp = ctx->sort_temp.p;
    g.temp_bytes = ctx->sort_temp.cap;
    g.seq_out = (float *)ctx->sort_seq.p;
    g.struct_out = (double *)ctx->sort_struct.p;
    g.hit_motif = sink.motif;
    g.motif_out = (int32_t *)ctx->sort_motif.p;
    g.motif_bits = motif_bits;
    {
        hipError_t e = launch_gather_sorted(g, st);
        if (e != hipSuccess) return fail_hip(ctx, e, "gather + sort of the library hits");
    }
    HIP_TRY(ctx, hipMemcpyAsync(hit_pos, g.keys_out, total * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipMemcpyAsync(hit_motif, g.motif_out, total * 4, hipMemcpyDeviceToHost, st));
    if (hit_seq && lib->has_letters) HIP_TRY(ctx, hipMemcpyAsync(hit_seq, g.seq_out, total * 4, hipMemcpyDeviceToHost, st));
    if (hit_struct && lib->has_struct) HIP_TRY(ctx, hipMemcpyAsync(hit_struct, g.struct_out, total * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    if (hit_struct && !lib->has_struct) std::fill(hit_struct, hit_struct + total, (double)NAN);
    if (hit_seq && !lib->has_letters) std::fill(hit_seq, hit_seq + total, NAN);
    return PFMSCAN_OK;
}

extern "C" {

int pfmscan_library_hits_dev(pfmscan_ctx *ctx, pfmscan_library *lib, const uint8_t *d_codes, const void *d_profile,
                             int profile_dtype, int64_t n_pos, const double *thr_seq, const double *thr_struct, int64_t capacity,
                             int64_t *d_hit_pos, int32_t *d_hit_motif, float *d_hit_seq, double *d_hit_struct,
                             uint64_t *d_hit_count, void *stream)
{
    int rc = lib_check(ctx, lib, d_codes, d_profile, profile_dtype, n_pos, thr_seq, thr_struct);
    if (rc) return rc;
    if (capacity < 0 || !d_hit_count || (capacity > 0 && (!d_hit_pos || !d_hit_motif)))
        return lib_fail(ctx, PFMSCAN_E_BADARG, "pfmscan_library_hits_dev: bad hit buffers");
    if (misaligned(d_codes) || misaligned(d_profile)) return lib_fail(ctx, PFMSCAN_E_BADSHAPE, "stream base pointers must be 16-byte aligned");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    if ((rc = lib_set_thresholds(ctx, lib, thr_seq, thr_struct, st))) return rc;
    LibSink sink;
    if ((rc = lib_scratch(ctx, capacity, n_pos, sink, lib_work_unit(lib)))) return rc;
    const size_t counter_bytes = (size_t)LIB_SHARDS * HIT_COUNTER_STRIDE * 8;
    HIP_TRY(ctx, hipMemsetAsync(sink.count, 0, counter_bytes, st));
    if ((rc = lib_run(ctx, lib, d_codes, d_profile, profile_dtype, n_pos, sink, st))) return rc;
    int64_t *starts = reinterpret_cast<int64_t *>(reinterpret_cast<unsigned char *>(sink.count) + (size_t)(LIB_SHARDS + 2) * HIT_COUNTER_STRIDE * 8);
    hipLaunchKernelGGL(k_lib_prefix, dim3(1), dim3(PACK_BLOCK), 0, st, sink.count, LIB_SHARDS, sink.shard_cap, capacity, starts,
                       reinterpret_cast<unsigned long long *>(d_hit_count));
    HIP_TRY(ctx, hipGetLastError());
    if (capacity > 0) {
        const int64_t most = std::min<int64_t>(capacity, sink.shard_cap * LIB_SHARDS);
        hipLaunchKernelGGL(k_lib_pack, dim3((unsigned)((most + PACK_BLOCK - 1) / PACK_BLOCK)), dim3(PACK_BLOCK), 0, st, starts,
                           LIB_SHARDS, sink.shard_cap, capacity, sink.pos, sink.motif, lib->has_letters ? sink.seq : nullptr,
                           lib->has_struct ? sink.st : nullptr, d_hit_pos, d_hit_motif, d_hit_seq, d_hit_struct);
        HIP_TRY(ctx, hipGetLastError());
    }
    return PFMSCAN_OK;
}

int pfmscan_library_hits_staged(pfmscan_ctx *ctx, pfmscan_library *lib, const double *thr_seq, const double *thr_struct,
                                int64_t capacity, int64_t *hit_pos, int32_t *hit_motif, float *hit_seq, double *hit_struct,
                                int64_t *n_hits)
{
    if (!n_hits) return lib_fail(ctx, PFMSCAN_E_BADARG, "NULL argument");
    if (!ctx || !lib) return lib_fail(ctx, PFMSCAN_E_BADARG, "NULL ctx or library");
    if (ctx->staged_n < 0) return lib_fail(ctx, PFMSCAN_E_BADARG, "no stream staged (call pfmscan_stage first)");
    if ((lib->has_letters || lib->letters8) && !ctx->staged_codes && ctx->staged_n > 0) return lib_fail(ctx, PFMSCAN_E_BADARG, "library scans need staged codes");
    if (lib->pair && !ctx->staged_codes2 && ctx->staged_n > 0)
        return lib_fail(ctx, PFMSCAN_E_BADARG, "two-FASTA library: two code streams must be staged (pfmscan_stage + pfmscan_stage_codes2)");
    if (lib->has_struct && !lib->pair && !lib->letters8 && !ctx->staged_profile && ctx->staged_n > 0)
        return lib_fail(ctx, PFMSCAN_E_BADARG, "library has structure PSSMs but no profile is staged");
    const int64_t n_pos = ctx->staged_n;
    const void *second = lib->pair ? ctx->codes2.p : ctx->profile.p;     // the structure side's stream: profile rows, or the second code stream
    int rc = lib_check(ctx, lib, (const uint8_t *)ctx->codes.p, second, ctx->staged_dtype, n_pos, thr_seq, thr_struct);
    if (rc) return rc;
    if (capacity < 0) return lib_fail(ctx, PFMSCAN_E_BADARG, "negative size");
    *n_hits = 0;
    if (n_pos == 0) return PFMSCAN_OK;
    if (capacity > 0 && (!hit_pos || !hit_motif)) return lib_fail(ctx, PFMSCAN_E_BADARG, "hit_pos / hit_motif is NULL");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    if ((rc = lib_set_thresholds(ctx, lib, thr_seq, thr_struct, st))) return rc;
    LibSink sink;
    if ((rc = lib_scratch(ctx, capacity, n_pos, sink, lib_work_unit(lib)))) return rc;
    const size_t counter_bytes = (size_t)LIB_SHARDS * HIT_COUNTER_STRIDE * 8;
    HIP_TRY(ctx, hipMemsetAsync(sink.count, 0, counter_bytes, st));
    if ((rc = lib_run(ctx, lib, (const uint8_t *)ctx->codes.p, second, ctx->staged_dtype, n_pos, sink, st))) return rc;
    return lib_finish_sorted(ctx, lib, n_pos, capacity, sink, hit_pos, hit_motif, hit_seq, hit_struct, n_hits);
}

int pfmscan_library_hits_letters_dev(pfmscan_ctx *ctx, pfmscan_library *lib, const uint8_t *d_codes, const uint8_t *d_codes2,
                                     int64_t n_pos, const double *thr_seq, const double *thr_struct, int64_t capacity,
                                     int64_t *d_hit_pos, int32_t *d_hit_motif, float *d_hit_seq, double *d_hit_struct,
                                     uint64_t *d_hit_count, void *stream)
{
    if (!ctx || !lib) return lib_fail(ctx, PFMSCAN_E_BADARG, "NULL ctx or library");
    if (!lib->letters8 && !lib->pair) return lib_fail(ctx, PFMSCAN_E_BADARG, "not a letter library (pfmscan_library_create_letters)");
    return pfmscan_library_hits_dev(ctx, lib, d_codes, lib->pair ? (const void *)d_codes2 : nullptr, PFMSCAN_PROFILE_NONE, n_pos, thr_seq,
                                    thr_struct, capacity, d_hit_pos, d_hit_motif, d_hit_seq, d_hit_struct, d_hit_count, stream);
}

int pfmscan_library_hits_letters_host(pfmscan_ctx *ctx, pfmscan_library *lib, const uint8_t *codes, const uint8_t *codes2, int64_t n_pos,
                                      const double *thr_seq, const double *thr_struct, int64_t capacity, int64_t *hit_pos,
                                      int32_t *hit_motif, float *hit_seq, double *hit_struct, int64_t *n_hits)
{
    if (!ctx || !lib || !n_hits) return lib_fail(ctx, PFMSCAN_E_BADARG, "NULL argument");
    if (!lib->letters8 && !lib->pair) return lib_fail(ctx, PFMSCAN_E_BADARG, "not a letter library (pfmscan_library_create_letters)");
    if (n_pos < 0 || capacity < 0) return lib_fail(ctx, PFMSCAN_E_BADARG, "negative size");
    *n_hits = 0;
    if (n_pos == 0) return PFMSCAN_OK;
    if (!codes || (lib->pair && !codes2)) return lib_fail(ctx, PFMSCAN_E_BADARG, "codes is NULL");
    int rc = pfmscan_stage(ctx, codes, nullptr, PFMSCAN_PROFILE_NONE, n_pos);
    if (rc) return rc;
    if (lib->pair && (rc = pfmscan_stage_codes2(ctx, codes2, n_pos))) return rc;
    return pfmscan_library_hits_staged(ctx, lib, thr_seq, thr_struct, capacity, hit_pos, hit_motif, hit_seq, hit_struct, n_hits);
}

// The library twin of pfmscan_hits_pipeline_host (pfmscan_pipeline.hip): a HOST-resident stream of any length, chunk by
// chunk through two alternating device buffers, the upload of chunk k + 1 (copy stream) beside the scan of chunk k.  A chunk
// holds its positions plus the m - 1 after them, so every window starting inside it sees its letters / rows; windows starting
// in the overhang run past the buffer's end and are never reported (they belong to the next chunk).
int pfmscan_library_hits_pipeline_host(pfmscan_ctx *ctx, pfmscan_library *lib, const uint8_t *codes, const void *profile,
                                       int profile_dtype, int64_t n_pos, int64_t chunk_positions, const double *thr_seq,
                                       const double *thr_struct, int64_t capacity, int64_t *hit_pos, int32_t *hit_motif,
                                       float *hit_seq, double *hit_struct, int64_t *n_hits)
{
    if (!ctx || !lib || !n_hits) return lib_fail(ctx, PFMSCAN_E_BADARG, "NULL argument");
    if (n_pos < 0 || capacity < 0) return lib_fail(ctx, PFMSCAN_E_BADARG, "negative size");
    *n_hits = 0;
    if (n_pos == 0) return PFMSCAN_OK;
    if (lib->letters8 || lib->pair)
        return lib_fail(ctx, PFMSCAN_E_BADARG, "letter libraries scan staged code streams (1 byte per position): pfmscan_library_hits_letters_host");
    int rc = lib_check(ctx, lib, codes, profile, profile_dtype, n_pos, thr_seq, thr_struct);
    if (rc) return rc;
    if (capacity > 0 && (!hit_pos || !hit_motif)) return lib_fail(ctx, PFMSCAN_E_BADARG, "hit_pos / hit_motif is NULL");
    if (chunk_positions <= 0) chunk_positions = (int64_t)1 << 24;
    chunk_positions = std::max<int64_t>((chunk_positions + LIB_SEG - 1) / LIB_SEG * LIB_SEG, LIB_SEG);    // whole work segments, 16-byte aligned starts
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ctx->staged_n = -1;                                                      // nothing stays staged
    const int m = lib->m;
    const size_t row_bytes = lib->has_struct ? (size_t)7 * (profile_dtype == PFMSCAN_PROFILE_F32 ? 4 : 8) : 0;
    const int64_t buf_positions = std::min<int64_t>(n_pos, chunk_positions + m - 1);
    if (!ctx->copy_stream) HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    for (int i = 0; i < 2; ++i) {
        if (lib->has_letters && (rc = ensure(ctx, ctx->pipe_codes[i], (size_t)buf_positions))) return rc;
        if (lib->has_struct && (rc = ensure(ctx, ctx->pipe_profile[i], (size_t)buf_positions * row_bytes))) return rc;
        if (!ctx->pipe_copied[i]) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->pipe_copied[i], hipEventDisableTiming));
        if (!ctx->pipe_scanned[i]) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->pipe_scanned[i], hipEventDisableTiming));
    }
    hipStream_t st = ctx->stream;
    if ((rc = lib_set_thresholds(ctx, lib, thr_seq, thr_struct, st))) return rc;
    LibSink sink;
    if ((rc = lib_scratch(ctx, capacity, n_pos, sink, lib_work_unit(lib)))) return rc;
    const size_t counter_bytes = (size_t)LIB_SHARDS * HIT_COUNTER_STRIDE * 8;
    HIP_TRY(ctx, hipMemsetAsync(sink.count, 0, counter_bytes, st));          // cleared once: the chunks' hits accumulate

    const int64_t n_chunks = (n_pos + chunk_positions - 1) / chunk_positions;
    auto upload_chunk = [&](int64_t k) -> int {
        const int b = (int)(k & 1);
        const int64_t a0 = k * chunk_positions;
        const int64_t len = std::min<int64_t>(n_pos - a0, chunk_positions + m - 1);
        if (k >= 2) HIP_TRY(ctx, hipStreamWaitEvent(ctx->copy_stream, ctx->pipe_scanned[b], 0));    // the buffer's previous chunk is scanned
        if (lib->has_letters)
            if (int urc = pfmscan::upload(ctx, ctx->pipe_codes[b].p, codes + a0, (size_t)len, ctx->copy_stream)) return urc;
        if (lib->has_struct)
            if (int urc = pfmscan::upload(ctx, ctx->pipe_profile[b].p, reinterpret_cast<const unsigned char *>(profile) + (size_t)a0 * row_bytes,
                                          (size_t)len * row_bytes, ctx->copy_stream))
                return urc;
        HIP_TRY(ctx, hipEventRecord(ctx->pipe_copied[b], ctx->copy_stream));
        return PFMSCAN_OK;
    };
    if ((rc = upload_chunk(0))) return rc;
    for (int64_t k = 0; k < n_chunks; ++k) {
        const int b = (int)(k & 1);
        const int64_t a0 = k * chunk_positions;
        const int64_t len = std::min<int64_t>(n_pos - a0, chunk_positions + m - 1);
        HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->pipe_copied[b], 0));
        if ((rc = lib_run(ctx, lib, (const uint8_t *)ctx->pipe_codes[b].p, ctx->pipe_profile[b].p, profile_dtype, len, sink, st, a0))) return rc;
        HIP_TRY(ctx, hipEventRecord(ctx->pipe_scanned[b], st));
        if (k + 1 < n_chunks && (rc = upload_chunk(k + 1))) return rc;
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->copy_stream));
    return lib_finish_sorted(ctx, lib, n_pos, capacity, sink, hit_pos, hit_motif, hit_seq, hit_struct, n_hits);
}

int pfmscan_library_hits_host(pfmscan_ctx *ctx, pfmscan_library *lib, const uint8_t *codes, const void *profile,
                              int profile_dtype, int64_t n_pos, const double *thr_seq, const double *thr_struct, int64_t capacity,
                              int64_t *hit_pos, int32_t *hit_motif, float *hit_seq, double *hit_struct, int64_t *n_hits)
{
    if (!ctx || !lib || !n_hits) return lib_fail(ctx, PFMSCAN_E_BADARG, "NULL argument");
    if (n_pos < 0 || capacity < 0) return lib_fail(ctx, PFMSCAN_E_BADARG, "negative size");
    *n_hits = 0;
    if (n_pos == 0) return PFMSCAN_OK;
    if (lib->letters8 || lib->pair) return lib_fail(ctx, PFMSCAN_E_BADARG, "letter library: use pfmscan_library_hits_letters_host");
    if (lib->has_letters && !codes) return lib_fail(ctx, PFMSCAN_E_BADARG, "codes is NULL");
    if (lib->has_struct && !profile) return lib_fail(ctx, PFMSCAN_E_BADARG, "profile is NULL");
    int rc = pfmscan_stage(ctx, lib->has_letters ? codes : nullptr, lib->has_struct ? profile : nullptr, profile_dtype, n_pos);
    if (rc) return rc;
    return pfmscan_library_hits_staged(ctx, lib, thr_seq, thr_struct, capacity, hit_pos, hit_motif, hit_seq, hit_struct, n_hits);
}

}  // extern "C"
