// pfmscan_ctx.hpp -- the context object behind the C ABI and the helpers every API translation unit
// shares (error reporting, device scratch).  Not installed.
#pragma once
#include <string>
#include <vector>

#include "pfmscan_internal.hpp"
#include "pfmscan_exact.hpp"

struct pfmscan_motif {
    pfmscan_ctx *ctx = nullptr;
    double *d_letters = nullptr;   // [m][8]
    float *d_pairs = nullptr;      // [(m+1)/2][16] two-letter fp32 sums (4-letter alphabets only), see k_letters_pre
    double pair_eps = 0.0;
    double h_pairsum[16 * 16];     // exact two-letter sums (m <= 32, 4-letter alphabets): operand of the integer prefilter
    bool has_pairsum = false;
    double *h_quadsum = nullptr;   // exact four-letter sums [ceil(m/4)][256] (same motifs): operand of k_letters_quad's credits
    mutable pfmscan::QuadCache quad_cache;   // k_letters_quad's credit tables by threshold (device + pinned host, allocated at first use)
    mutable pfmscan::CredCache cred_cache;
    double *h_letters = nullptr;   // host copy of the letter table [m][8] (m <= 32): operand of k_letters_cred8's credits
    mutable pfmscan::Cred8Cache cred8_cache;
    double *d_struct = nullptr;    // [m][7]
    int m = 0;
    int struct_finite = 0;
    double struct_band = 0.0;      // pfmscan_exact.hpp: half-width of the re-score band of thresholded structure compares
};

namespace pfmscan {
struct Uploader;   // pfmscan_upload.hip: pinned staging buffers + the host threads that fill them
}

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
    DevBuf codes2;                              // second code stream of the two-FASTA combined scan (pfmscan_stage_codes2)
    DevBuf codes, profile, out_seq, out_struct, hit_pos, hit_seq, hit_struct, count, table;
    DevBuf cand_pos, cand_seq, cand_count;      // candidates of the two-phase combined scan
    DevBuf sort_keys_in, sort_keys_out, sort_vals_in, sort_vals_out, sort_temp, sort_seq, sort_struct;   // pfmscan_sort.hip
    DevBuf hit_motif, sort_motif;               // library scans: motif index per hit
    DevBuf lib_pos, lib_motif, lib_seq, lib_struct, lib_count;   // library scans: sharded hits of the _dev form
    DevBuf pipe_codes[2], pipe_profile[2];      // chunked host pipeline: double-buffered chunk of the stream
    hipEvent_t pipe_copied[2] = {nullptr, nullptr}, pipe_scanned[2] = {nullptr, nullptr};
    // host ranges known to be read-only mappings of files (pfmscan_upload_source_file): the staged uploader preads them
    struct FileRange {
        const unsigned char *base = nullptr;
        size_t length = 0;
        int fd = -1;
        int64_t offset = 0;
    };
    static constexpr int N_FILE_RANGES = 8;
    FileRange file_ranges[N_FILE_RANGES];
    int file_range_next = 0;
    pfmscan::Uploader *up = nullptr;
    int upload_mode = PFMSCAN_UPLOAD_RUNTIME;
    // staged stream (pfmscan_stage)
    int64_t staged_n = -1;
    int staged_dtype = PFMSCAN_PROFILE_NONE;
    bool staged_codes = false, staged_profile = false, staged_codes2 = false;
    // candidate-then-verify: the last full letters pass was selective -> skip the pilot next time
    bool two_phase_hot = false;
    // pfmscan_place.hip: sets of arrays placed together (PlaceSet *), and a line about the last allocation
    std::vector<void *> place_sets;
    std::vector<void *> place_retired;        // freed sets kept mapped for the next request of the same sizes
    size_t place_va_reserved = 0;             // address space reserved so far (never given back while the context lives)
    std::string place_note;
};

namespace pfmscan {

int fail(pfmscan_ctx *ctx, int code, const std::string &msg);
int fail_hip(pfmscan_ctx *ctx, hipError_t e, const char *what);
int ensure(pfmscan_ctx *ctx, DevBuf &b, size_t bytes);
void release(DevBuf &b);
// pfmscan_api.hip: argument checks + ScanArgs of a scan of DEVICE buffers; launch on a stream; sharded hits -> sorted host arrays
int check_and_fill(pfmscan_ctx *ctx, const pfmscan_motif *mo, const uint8_t *d_codes, const void *d_profile, int profile_dtype,
                   int64_t n_pos, ScanArgs &a);
int do_launch(pfmscan_ctx *ctx, const ScanArgs &a, void *stream);
int finish_sorted_hits(pfmscan_ctx *ctx, bool has_seq, bool has_struct, int64_t n_pos, int64_t capacity, int64_t shard_cap,
                       int64_t *hit_pos, float *hit_seq, double *hit_struct, int64_t *n_hits);
// pfmscan_upload.hip: asynchronous host -> device copy on `st`; the source may be reused when it returns
int upload(pfmscan_ctx *ctx, void *d_dst, const void *h_src, size_t bytes, hipStream_t st);
void upload_release(pfmscan_ctx *ctx);
void place_release_all(pfmscan_ctx *ctx);   // pfmscan_place.hip
inline bool misaligned(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) != 0; }

}  // namespace pfmscan

#define HIP_TRY(ctx, expr)                                                  \
    do {                                                                    \
        hipError_t e__ = (expr);                                            \
        if (e__ != hipSuccess) return pfmscan::fail_hip((ctx), e__, #expr); \
    } while (0)
