// pfmscan_pipeline.hip -- thresholded hits of a HOST-resident packed stream of any length, chunk by chunk, with the
// upload of chunk k+1 running beside the scan of chunk k (the packed profile store is memory-mapped
// and handed to the device as is).
//
// The reference parses one structure.<id>.txt per record inside the scan loop (rnascan.py:296-297, :351); at config-3
// size the packed inputs are 8.7 GB against a 2 ms kernel, so the host path is the copy: pfmscan_stage + a scan
// needs the whole stream resident and runs copy and scan one after the other.  Here two chunk buffers alternate:
//
//    copy stream     H2D chunk 0 | H2D chunk 1 | H2D chunk 2 | ...
//    ctx stream                  | scan chunk 0| scan chunk 1| ...        (events order buffer reuse both ways)
//
// A chunk holds the positions [a, b) plus the m - 1 positions after b: windows that start in [a, b) see all their
// letters / rows, windows that start in the overhang run past the chunk's end and score NaN (they belong to the next
// chunk) -- the end-of-stream rule of the kernels does the bookkeeping.  Every chunk is ONE fused hits launch
// (k_profile / k_letters in hits mode; no count read-back in between), hits go to the ctx's sharded buffers with
// their stream position (ScanArgs::pos_offset) and are sorted once at the end.  Device scratch: two chunks,
// whatever the stream length.
#include <algorithm>
#include <cmath>
#include <string>

#include "pfmscan_ctx.hpp"

using namespace pfmscan;

extern "C" {

int pfmscan_hits_pipeline_host(pfmscan_ctx *ctx, const pfmscan_motif *mo, const uint8_t *codes, const void *profile,
                               int profile_dtype, int64_t n_pos, int64_t chunk_positions, double thr_seq, double thr_struct,
                               int64_t capacity, int64_t *hit_pos, float *hit_seq, double *hit_struct, int64_t *n_hits)
{
    if (!ctx || !mo || !n_hits) return fail(ctx, PFMSCAN_E_BADARG, "NULL argument");
    if (n_pos < 0 || capacity < 0) return fail(ctx, PFMSCAN_E_BADARG, "negative size");
    if (std::isnan(thr_seq) || std::isnan(thr_struct)) return fail(ctx, PFMSCAN_E_BADARG, "NaN threshold");
    *n_hits = 0;
    if (n_pos == 0) return PFMSCAN_OK;
    if (mo->d_letters && !codes) return fail(ctx, PFMSCAN_E_BADARG, "codes is NULL");
    if (mo->d_struct && !profile) return fail(ctx, PFMSCAN_E_BADARG, "profile is NULL");
    if (mo->d_struct && profile_dtype != PFMSCAN_PROFILE_F32 && profile_dtype != PFMSCAN_PROFILE_F64)
        return fail(ctx, PFMSCAN_E_BADARG, "profile_dtype must be F32 or F64");
    if (capacity > 0 && !hit_pos) return fail(ctx, PFMSCAN_E_BADARG, "hit_pos is NULL");
    if (chunk_positions <= 0) chunk_positions = (int64_t)1 << 24;            // 0.49 GB of float32 rows per buffer
    chunk_positions = std::max<int64_t>((chunk_positions + 1023) & ~(int64_t)1023, 4096);   // 16-byte aligned chunk starts
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ctx->staged_n = -1;                                                      // nothing stays staged
    const int m = mo->m;
    const size_t row_bytes = mo->d_struct ? (size_t)7 * (profile_dtype == PFMSCAN_PROFILE_F32 ? 4 : 8) : 0;
    const int64_t buf_positions = std::min<int64_t>(n_pos, chunk_positions + m - 1);
    int rc;
    if (!ctx->copy_stream) HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    for (int i = 0; i < 2; ++i) {
        if (mo->d_letters && (rc = ensure(ctx, ctx->pipe_codes[i], (size_t)buf_positions))) return rc;
        if (mo->d_struct && (rc = ensure(ctx, ctx->pipe_profile[i], (size_t)buf_positions * row_bytes))) return rc;
        if (!ctx->pipe_copied[i]) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->pipe_copied[i], hipEventDisableTiming));
        if (!ctx->pipe_scanned[i]) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->pipe_scanned[i], hipEventDisableTiming));
    }
    // the ctx's sharded hit buffers, as pfmscan_hits_staged sizes them; counters are cleared once, hits accumulate
    const int64_t shard_cap = std::max<int64_t>(std::min<int64_t>(capacity, capacity / HIT_SHARDS * 2 + 4096), 1);
    const size_t slots = (size_t)shard_cap * HIT_SHARDS;
    const size_t counter_bytes = (size_t)HIT_SHARDS * HIT_COUNTER_STRIDE * 8;
    if ((rc = ensure(ctx, ctx->hit_pos, slots * 8))) return rc;
    if ((rc = ensure(ctx, ctx->hit_seq, slots * 4))) return rc;
    if ((rc = ensure(ctx, ctx->hit_struct, slots * 8))) return rc;
    if ((rc = ensure(ctx, ctx->count, counter_bytes))) return rc;
    HIP_TRY(ctx, hipMemsetAsync(ctx->count.p, 0, counter_bytes, ctx->stream));

    const int64_t n_chunks = (n_pos + chunk_positions - 1) / chunk_positions;
    auto upload = [&](int64_t k) -> int {
        const int b = (int)(k & 1);
        const int64_t a0 = k * chunk_positions;
        const int64_t len = std::min<int64_t>(n_pos - a0, chunk_positions + m - 1);
        if (k >= 2) HIP_TRY(ctx, hipStreamWaitEvent(ctx->copy_stream, ctx->pipe_scanned[b], 0));    // the buffer's previous chunk is scanned
        if (mo->d_letters)
            if (int urc = pfmscan::upload(ctx, ctx->pipe_codes[b].p, codes + a0, (size_t)len, ctx->copy_stream)) return urc;
        if (mo->d_struct)
            if (int urc = pfmscan::upload(ctx, ctx->pipe_profile[b].p, reinterpret_cast<const unsigned char *>(profile) + (size_t)a0 * row_bytes,
                                          (size_t)len * row_bytes, ctx->copy_stream))
                return urc;
        HIP_TRY(ctx, hipEventRecord(ctx->pipe_copied[b], ctx->copy_stream));
        return PFMSCAN_OK;
    };
    if ((rc = upload(0))) return rc;
    for (int64_t k = 0; k < n_chunks; ++k) {
        const int b = (int)(k & 1);
        const int64_t a0 = k * chunk_positions;
        const int64_t len = std::min<int64_t>(n_pos - a0, chunk_positions + m - 1);
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->pipe_copied[b], 0));
        ScanArgs a;
        if ((rc = check_and_fill(ctx, mo, (const uint8_t *)ctx->pipe_codes[b].p, ctx->pipe_profile[b].p, profile_dtype, len, a))) return rc;
        a.hits = 1;
        a.thr_seq = thr_seq;
        a.thr_struct = thr_struct;
        a.capacity = shard_cap;
        a.hit_pos = (int64_t *)ctx->hit_pos.p;
        a.hit_seq = mo->d_letters ? (float *)ctx->hit_seq.p : nullptr;
        a.hit_struct = mo->d_struct ? (double *)ctx->hit_struct.p : nullptr;
        a.hit_count = (unsigned long long *)ctx->count.p;
        a.hit_shards = HIT_SHARDS;
        a.pos_offset = a0;
        if ((rc = do_launch(ctx, a, ctx->stream))) return rc;               // asynchronous: the next upload runs beside it
        HIP_TRY(ctx, hipEventRecord(ctx->pipe_scanned[b], ctx->stream));
        if (k + 1 < n_chunks && (rc = upload(k + 1))) return rc;
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->copy_stream));
    return finish_sorted_hits(ctx, mo->d_letters != nullptr, mo->d_struct != nullptr, n_pos, capacity, shard_cap, hit_pos, hit_seq, hit_struct, n_hits);
}

}  // extern "C"
