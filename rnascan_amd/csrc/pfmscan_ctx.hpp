// pfmscan_ctx.hpp -- the context object behind the C ABI and the helpers every API translation unit
// shares (error reporting, device scratch).  Not installed.
#pragma once
#include <string>

#include "pfmscan_internal.hpp"

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
};

struct pfmscan_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;   // H2D of the next chunk while the previous one is scanned (pfmscan_pipeline.hip)
    std::string err;
    pfmscan::Tuning tune;
    int n_cu = 0;
    int64_t hbm = 0;
    char name[128] = {0};
    DevBuf codes, profile, out_seq, out_struct, hit_pos, hit_seq, hit_struct, count, table;
    DevBuf cand_pos, cand_seq, cand_count;      // candidates of the two-phase combined scan
    DevBuf sort_keys_in, sort_keys_out, sort_vals_in, sort_vals_out, sort_temp, sort_seq, sort_struct;   // pfmscan_sort.hip
    DevBuf hit_motif, sort_motif;               // library scans: motif index per hit
    DevBuf lib_pos, lib_motif, lib_seq, lib_struct, lib_count;   // library scans: sharded hits of the _dev form
    // staged stream (pfmscan_stage)
    int64_t staged_n = -1;
    int staged_dtype = PFMSCAN_PROFILE_NONE;
    bool staged_codes = false, staged_profile = false;
    // candidate-then-verify: the last full letters pass was selective -> skip the pilot next time
    bool two_phase_hot = false;
};

namespace pfmscan {

int fail(pfmscan_ctx *ctx, int code, const std::string &msg);
int fail_hip(pfmscan_ctx *ctx, hipError_t e, const char *what);
int ensure(pfmscan_ctx *ctx, DevBuf &b, size_t bytes);
void release(DevBuf &b);
inline bool misaligned(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) != 0; }

}  // namespace pfmscan

#define HIP_TRY(ctx, expr)                                                  \
    do {                                                                    \
        hipError_t e__ = (expr);                                            \
        if (e__ != hipSuccess) return pfmscan::fail_hip((ctx), e__, #expr); \
    } while (0)
