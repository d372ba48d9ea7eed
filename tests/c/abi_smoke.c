/* Plain-C consumer of include/pfmscan.h: proves the boundary is a C ABI (no C++ or torch types)
 * and that the reference's native call, calculate(sequence, matrix), maps onto it one to one.
 * Built and run by tests/test_gpu_parity.py::test_c_program_through_the_abi on the GPU box. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "pfmscan.h"

int main(void)
{
    pfmscan_ctx *ctx = NULL;
    if (pfmscan_ctx_create(0, &ctx) != PFMSCAN_OK) {
        fprintf(stderr, "ctx: %s\n", pfmscan_last_error(NULL));
        return 2;
    }
    /* the loop of _pwm.c:34-68 on the host, as the expected answer */
    const char *seq = "ACGUNacgtTTGACCAGUUACGGA";
    const int64_t s = (int64_t)strlen(seq), m = 3;
    double M[3][4] = {{0.5, -1.25, 2.0, -0.75}, {1.5, 0.25, -2.0, 0.125}, {-0.5, 1.0, 0.75, -1.5}};
    float got[64], want[64];
    for (int64_t i = 0; i + m <= s; ++i) {
        double score = 0.0;
        int ok = 1;
        for (int64_t j = 0; j < m; ++j) {
            switch (seq[i + j]) {
            case 'A': case 'a': score += M[j][0]; break;
            case 'C': case 'c': score += M[j][1]; break;
            case 'G': case 'g': score += M[j][2]; break;
            case 'T': case 't': case 'U': case 'u': score += M[j][3]; break;
            default: ok = 0;
            }
        }
        want[i] = ok ? (float)score : NAN;
    }
    if (pfmscan_pwm_calculate(ctx, seq, s, &M[0][0], m, got) != PFMSCAN_OK) {
        fprintf(stderr, "calculate: %s\n", pfmscan_last_error(ctx));
        return 3;
    }
    int bad = 0;
    for (int64_t i = 0; i + m <= s; ++i) {
        if (isnan(want[i]) ? !isnan(got[i]) : (memcmp(&want[i], &got[i], sizeof(float)) != 0)) {
            fprintf(stderr, "window %lld: got %g want %g\n", (long long)i, got[i], want[i]);
            bad++;
        }
    }
    /* error convention: width out of range -> BADSHAPE with a message */
    if (pfmscan_pwm_calculate(ctx, seq, s, &M[0][0], PFMSCAN_MAX_WIDTH + 1, got) != PFMSCAN_E_BADSHAPE || !strlen(pfmscan_last_error(ctx))) bad++;
    /* the structure letter-string mode (matrix.py:25-43 + rnascan.py:263) from plain C: a 7-letter table over a packed
     * stream of two records, hits in fp64 with the strict `>`; the expected answer is the Python loop restated here */
    {
        const char *recs[2] = {"EEHHTTLLRRMMBBxEHTLE", "lehtEHTB"};
        const char *alphabet = "EHTBLRM";
        enum { W = 4 };
        double T[W][8];
        uint8_t codes[64] __attribute__((aligned(16)));
        double sc[64];
        int64_t n = 0;
        for (int j = 0; j < W; ++j)
            for (int c = 0; c < 8; ++c) T[j][c] = c < 7 ? (double)((j * 7 + c * 3) % 11) * 0.37 - 1.5 : NAN;
        for (int r = 0; r < 2; ++r) {
            for (const char *q = recs[r]; *q; ++q) {
                const char up = (char)(*q >= 'a' && *q <= 'z' ? *q - 32 : *q);
                const char *at = strchr(alphabet, up);
                codes[n++] = (uint8_t)(at ? (at - alphabet) | (up != *q ? 8 : 0) : PFMSCAN_SEP);   /* bit 3: written in lower case */
            }
            codes[n++] = PFMSCAN_SEP;
        }
        const double thr = 0.25;
        int64_t want_pos[64], n_want = 0;
        double want_sc[64];
        for (int64_t p = 0; p + W <= n; ++p) {
            double score = 0.0;
            for (int j = 0; j < W; ++j) score += T[j][codes[p + j] & 7];
            sc[p] = score;
            if (score > thr) { want_pos[n_want] = p; want_sc[n_want++] = score; }
        }
        pfmscan_motif *mo = NULL;
        int64_t hit_pos[64], n_hits = 0;
        double hit_sc[64];
        if (pfmscan_motif_create(ctx, &T[0][0], NULL, W, &mo) != PFMSCAN_OK ||
            pfmscan_hits_letters_f64_host(ctx, mo, codes, n, thr, 64, hit_pos, hit_sc, &n_hits) != PFMSCAN_OK) {
            fprintf(stderr, "hits_letters_f64: %s\n", pfmscan_last_error(ctx));
            bad++;
        } else {
            if (n_hits != n_want) bad++;
            for (int64_t i = 0; i < n_hits && i < n_want; ++i)
                if (hit_pos[i] != want_pos[i] || memcmp(&hit_sc[i], &want_sc[i], sizeof(double)) != 0) bad++;
            if (n_want < 3) bad++;                       /* the example must have hits to compare */
        }
        pfmscan_motif_destroy(mo);
        (void)sc;
    }
    pfmscan_ctx_destroy(ctx);
    printf(bad ? "FAIL\n" : "OK %lld windows\n", (long long)(s - m + 1));
    return bad ? 1 : 0;
}
