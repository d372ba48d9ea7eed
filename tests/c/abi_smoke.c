/* Plain-C consumer of include/pfmscan.h: proves the boundary is a C ABI (no C++ or torch types)
 * and that the reference's native call, calculate(sequence, matrix), maps onto it one to one.
 * Built and run by tests/test_gpu_parity.py::test_c_program_through_the_abi on the GPU box. */
#include <math.h>
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
    pfmscan_ctx_destroy(ctx);
    printf(bad ? "FAIL\n" : "OK %lld windows\n", (long long)(s - m + 1));
    return bad ? 1 : 0;
}
