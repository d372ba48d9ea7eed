// pfmscan_api.hip -- the C ABI declared in include/pfmscan.h: contexts, PSSM
// operands, device-pointer launches, host-buffer staging, error reporting.
// There is no CPU fallback in this library: without a gfx950 device every
// entry point fails with PFMSCAN_E_HIP.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <numeric>
#include <string>
#include <vector>

#include "pfmscan_ctx.hpp"

using namespace pfmscan;

static thread_local std::string g_err;

namespace pfmscan {

int fail(pfmscan_ctx *ctx, int code, const std::string &msg)
{
    if (ctx) ctx->err = msg; else g_err = msg;
    return code;
}

int fail_hip(pfmscan_ctx *ctx, hipError_t e, const char *what)
{
    std::string msg = std::string(what) + ": " + hipGetErrorString(e);
    return fail(ctx, e == hipErrorOutOfMemory ? PFMSCAN_E_OOM : PFMSCAN_E_HIP, msg);
}

int ensure(pfmscan_ctx *ctx, DevBuf &b, size_t bytes)
{
    if (bytes <= b.cap) return PFMSCAN_OK;
    if (b.p) {
        HIP_TRY(ctx, hipFree(b.p));
        b.p = nullptr;
        b.cap = 0;
    }
    size_t want = (bytes + 255) & ~(size_t)255;
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) {
        b.p = nullptr;
        return fail(ctx, PFMSCAN_E_OOM, std::string("hipMalloc of ") + std::to_string(want) + " bytes failed: " + hipGetErrorString(e));
    }
    b.cap = want;
    return PFMSCAN_OK;
}

void release(DevBuf &b)
{
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
}

}  // namespace pfmscan

extern "C" {

int pfmscan_abi_version(void) { return PFMSCAN_ABI_VERSION; }

int pfmscan_ctx_create(int device, pfmscan_ctx **out)
{
    if (!out || device < 0) return fail(nullptr, PFMSCAN_E_BADARG, "pfmscan_ctx_create: bad argument");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(nullptr, PFMSCAN_E_HIP, std::string("no HIP device visible (") + hipGetErrorString(e) + "); libpfmscan has no CPU fallback");
    if (device >= n) return fail(nullptr, PFMSCAN_E_BADARG, "pfmscan_ctx_create: device index out of range");
    HIP_TRY(nullptr, hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(nullptr, hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, PFMSCAN_E_HIP, std::string("device is ") + prop.gcnArchName + ", this library carries gfx950 code only");
    pfmscan_ctx *ctx = new (std::nothrow) pfmscan_ctx();
    if (!ctx) return fail(nullptr, PFMSCAN_E_OOM, "out of host memory");
    ctx->device = device;
    ctx->n_cu = prop.multiProcessorCount;
    ctx->tune.n_cu = ctx->n_cu > 0 ? ctx->n_cu : 256;
    ctx->hbm = (int64_t)prop.totalGlobalMem;
    std::snprintf(ctx->name, sizeof(ctx->name), "%s (%s)", prop.name, prop.gcnArchName);
    e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete ctx;
        return fail_hip(nullptr, e, "hipStreamCreate");
    }
    if (const char *v = std::getenv("PFMSCAN_V")) {
        int x = std::atoi(v);
        if (x == 5 || x == 7) ctx->tune.v = x;
    }
    if (const char *v = std::getenv("PFMSCAN_DMA")) ctx->tune.dma = std::atoi(v) != 0;
    if (const char *v = std::getenv("PFMSCAN_ABLATE")) ctx->tune.ablate = std::atoi(v);
    if (const char *v = std::getenv("PFMSCAN_PRIO")) ctx->tune.prio = std::atoi(v) != 0;
    if (const char *v = std::getenv("PFMSCAN_DMA_TAIL")) ctx->tune.dma_whole = std::atoi(v) == 0;
    if (const char *v = std::getenv("PFMSCAN_TWO_PHASE")) ctx->tune.two_phase = std::atoi(v) != 0;
    if (const char *v = std::getenv("PFMSCAN_TILES_PER_BLOCK")) ctx->tune.tiles_per_block = std::max(0, std::min(1024, std::atoi(v)));
    if (const char *v = std::getenv("PFMSCAN_PREFILTER")) ctx->tune.prefilter = std::atoi(v) != 0;
    if (const char *v = std::getenv("PFMSCAN_CREDITS")) ctx->tune.credits = std::atoi(v) != 0;
    if (const char *v = std::getenv("PFMSCAN_QUAD")) ctx->tune.quad = std::atoi(v) != 0;
    *out = ctx;
    return PFMSCAN_OK;
}

void pfmscan_ctx_destroy(pfmscan_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipStreamDestroy(ctx->stream);
    }
    if (ctx->copy_stream) {
        (void)hipStreamSynchronize(ctx->copy_stream);
        (void)hipStreamDestroy(ctx->copy_stream);
    }
    for (DevBuf *b : {&ctx->codes, &ctx->profile, &ctx->out_seq, &ctx->out_struct, &ctx->hit_pos,
                      &ctx->hit_seq, &ctx->hit_struct, &ctx->count, &ctx->table, &ctx->cand_pos, &ctx->cand_seq,
                      &ctx->cand_count, &ctx->sort_keys_in, &ctx->sort_keys_out, &ctx->sort_vals_in, &ctx->sort_vals_out,
                      &ctx->sort_temp, &ctx->sort_seq, &ctx->sort_struct, &ctx->hit_motif, &ctx->sort_motif, &ctx->lib_pos,
                      &ctx->lib_motif, &ctx->lib_seq, &ctx->lib_struct, &ctx->lib_count, &ctx->pipe_codes[0], &ctx->pipe_codes[1],
                      &ctx->pipe_profile[0], &ctx->pipe_profile[1], &ctx->codes2})
        release(*b);
    upload_release(ctx);
    place_release_all(ctx);
    for (int i = 0; i < 2; ++i) {
        if (ctx->pipe_copied[i]) (void)hipEventDestroy(ctx->pipe_copied[i]);
        if (ctx->pipe_scanned[i]) (void)hipEventDestroy(ctx->pipe_scanned[i]);
    }
    delete ctx;
}

const char *pfmscan_last_error(const pfmscan_ctx *ctx) { return ctx ? ctx->err.c_str() : g_err.c_str(); }

int pfmscan_device_info(const pfmscan_ctx *ctx, int *n_cu, int64_t *hbm_bytes, char *name, int name_cap)
{
    if (!ctx) return PFMSCAN_E_BADARG;
    if (n_cu) *n_cu = ctx->n_cu;
    if (hbm_bytes) *hbm_bytes = ctx->hbm;
    if (name && name_cap > 0) std::snprintf(name, (size_t)name_cap, "%s", ctx->name);
    return PFMSCAN_OK;
}

int pfmscan_synchronize(pfmscan_ctx *ctx)
{
    if (!ctx) return PFMSCAN_E_BADARG;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PFMSCAN_OK;
}

int pfmscan_motif_create(pfmscan_ctx *ctx, const double *letter_table, const double *struct_pssm, int m,
                         pfmscan_motif **out)
{
    if (!ctx || !out) return fail(ctx, PFMSCAN_E_BADARG, "pfmscan_motif_create: NULL argument");
    *out = nullptr;
    if (!letter_table && !struct_pssm) return fail(ctx, PFMSCAN_E_BADARG, "pfmscan_motif_create: no table given");
    if (m < 1 || m > PFMSCAN_MAX_WIDTH)
        return fail(ctx, PFMSCAN_E_BADSHAPE, "PFM width " + std::to_string(m) + " outside 1.." + std::to_string(PFMSCAN_MAX_WIDTH));
    if (letter_table) {
        for (int j = 0; j < m; ++j)
            if (!std::isnan(letter_table[j * 8 + PFMSCAN_SEP]))
                return fail(ctx, PFMSCAN_E_BADSHAPE, "letter_table column 7 (separator / foreign letter) must be NaN");
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    pfmscan_motif *mo = new (std::nothrow) pfmscan_motif();
    if (!mo) return fail(ctx, PFMSCAN_E_OOM, "out of host memory");
    mo->ctx = ctx;
    mo->m = m;
    hipError_t e = hipSuccess;
    if (letter_table) {
        e = hipMalloc((void **)&mo->d_letters, sizeof(double) * m * 8);
        if (e == hipSuccess) e = hipMemcpy(mo->d_letters, letter_table, sizeof(double) * m * 8, hipMemcpyHostToDevice);
    }
    if (letter_table && m <= 32) {
        mo->h_letters = new (std::nothrow) double[(size_t)m * 8];
        if (mo->h_letters) std::memcpy(mo->h_letters, letter_table, sizeof(double) * m * 8);
    }
    if (e == hipSuccess && letter_table) {
        // hits-mode prefilter table: only when the alphabet is the 4 codes 0..3 (columns 4..7 all NaN)
        bool four = true;
        for (int j = 0; j < m && four; ++j)
            for (int c = 4; c < 8; ++c)
                if (!std::isnan(letter_table[j * 8 + c])) four = false;
        if (four) {
            const int npair = (m + 1) / 2;
            std::vector<float> pairs((size_t)npair * 16);
            double bound = 0.0;
            for (int t = 0; t < npair; ++t) {
                double mx = 0.0;
                for (int c0 = 0; c0 < 4; ++c0)
                    for (int c1 = 0; c1 < 4; ++c1) {
                        const double v = letter_table[(2 * t) * 8 + c0] + (2 * t + 1 < m ? letter_table[(2 * t + 1) * 8 + c1] : 0.0);
                        pairs[(size_t)t * 16 + (c0 | c1 << 2)] = (float)v;
                        if (std::isfinite(v)) mx = std::max(mx, std::fabs(v));
                        if (std::isfinite(v) && std::fabs(v) > 1e30) four = false;   // fp32 would overflow to inf
                    }
                bound += mx;
            }
            // rounding of the entries + of the fp32 adds + of the exact score's float cast <= ~4e-6 * bound at 32 pairs
            mo->pair_eps = bound * 0x1p-17 + 1e-30;
            if (four && m <= 32) {
                pair_sums(letter_table, m, mo->h_pairsum);
                mo->has_pairsum = true;
                mo->h_quadsum = new (std::nothrow) double[(size_t)((m + 3) / 4) * 256];
                if (mo->h_quadsum) quad_sums(letter_table, m, mo->h_quadsum);
            }
            if (four) e = hipMalloc((void **)&mo->d_pairs, sizeof(float) * pairs.size());
            if (four && e == hipSuccess) e = hipMemcpy(mo->d_pairs, pairs.data(), sizeof(float) * pairs.size(), hipMemcpyHostToDevice);
        }
    }
    if (e == hipSuccess && struct_pssm) {
        mo->struct_finite = 1;
        for (int i = 0; i < m * 7; ++i)
            if (!std::isfinite(struct_pssm[i])) mo->struct_finite = 0;
        mo->struct_band = struct_band(struct_pssm, m);
        e = hipMalloc((void **)&mo->d_struct, sizeof(double) * m * 7);
        if (e == hipSuccess) e = hipMemcpy(mo->d_struct, struct_pssm, sizeof(double) * m * 7, hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) {
        pfmscan_motif_destroy(mo);
        return fail_hip(ctx, e, "uploading PSSM tables");
    }
    if (std::getenv("PFMSCAN_FORCE_GENERIC")) mo->struct_finite = 0;
    *out = mo;
    return PFMSCAN_OK;
}

void pfmscan_motif_destroy(pfmscan_motif *mo)
{
    if (!mo) return;
    if (mo->ctx) (void)hipSetDevice(mo->ctx->device);
    if (mo->d_letters) (void)hipFree(mo->d_letters);
    if (mo->d_pairs) (void)hipFree(mo->d_pairs);
    if (mo->d_struct) (void)hipFree(mo->d_struct);
    quad_cache_release(mo->quad_cache);
    delete[] mo->h_quadsum;
    delete[] mo->h_letters;
    delete mo;
}

}  // extern "C"

// ---- shared argument checking + launch --------------------------------------
int pfmscan::check_and_fill(pfmscan_ctx *ctx, const pfmscan_motif *mo, const uint8_t *d_codes, const void *d_profile,
                          int profile_dtype, int64_t n_pos, ScanArgs &a)
{
    if (!ctx || !mo) return fail(ctx, PFMSCAN_E_BADARG, "NULL ctx or motif");
    if (mo->ctx != ctx) return fail(ctx, PFMSCAN_E_BADARG, "motif belongs to another ctx");
    if (n_pos < 0) return fail(ctx, PFMSCAN_E_BADARG, "negative n_pos");
    if (mo->d_letters && !d_codes && n_pos > 0) return fail(ctx, PFMSCAN_E_BADARG, "motif has a letter table but codes is NULL");
    if (mo->d_struct) {
        if (profile_dtype != PFMSCAN_PROFILE_F32 && profile_dtype != PFMSCAN_PROFILE_F64)
            return fail(ctx, PFMSCAN_E_BADARG, "motif has a structure PSSM: profile_dtype must be F32 or F64");
        if (!d_profile && n_pos > 0) return fail(ctx, PFMSCAN_E_BADARG, "motif has a structure PSSM but profile is NULL");
    }
    if (misaligned(d_codes) || misaligned(d_profile)) return fail(ctx, PFMSCAN_E_BADSHAPE, "stream base pointers must be 16-byte aligned");
    std::memset(&a, 0, sizeof(a));
    a.codes = d_codes;
    a.profile = mo->d_struct ? d_profile : nullptr;
    a.profile_dtype = profile_dtype;
    a.n_pos = n_pos;
    a.letter_table = mo->d_letters;
    a.pair_table = ctx->tune.prefilter ? mo->d_pairs : nullptr;
    a.pair_eps = mo->pair_eps;
    a.h_pairsum = mo->has_pairsum ? mo->h_pairsum : nullptr;
    a.h_quadsum = mo->h_quadsum;
    a.d_quad = nullptr;
    a.quad_cache = &mo->quad_cache;
    a.cred_cache = &mo->cred_cache;
    a.h_letters = mo->h_letters;
    a.cred8_cache = &mo->cred8_cache;
    a.struct_pssm = mo->d_struct;
    a.m = mo->m;
    a.struct_finite = mo->struct_finite;
    a.struct_band = mo->struct_band;
    a.ablate = ctx->tune.ablate;
    a.prio = ctx->tune.prio;
    a.dma_whole = ctx->tune.dma_whole;
    return PFMSCAN_OK;
}

int pfmscan::do_launch(pfmscan_ctx *ctx, const ScanArgs &a, void *stream)
{
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const char *what = "";
    hipError_t e = launch_scan(a, ctx->tune, stream ? (hipStream_t)stream : ctx->stream, &what);
    if (e != hipSuccess) return fail_hip(ctx, e, what);
    return PFMSCAN_OK;
}

// The ctx-owned sharded hit buffers (HIT_SHARDS regions of shard_cap slots, counters in ctx->count) -> the caller's host
// arrays, sorted by position: capacity check, device sort (pfmscan_sort.hip), three contiguous copies.  Synchronises
// ctx->stream.  Hit positions lie in [0, n_pos).
int pfmscan::finish_sorted_hits(pfmscan_ctx *ctx, bool has_seq, bool has_struct, int64_t n_pos, int64_t capacity, int64_t shard_cap,
                                int64_t *hit_pos, float *hit_seq, double *hit_struct, int64_t *n_hits)
{
    int rc;
    const size_t counter_bytes = (size_t)HIT_SHARDS * HIT_COUNTER_STRIDE * 8;
    std::vector<unsigned long long> counters((size_t)HIT_SHARDS * HIT_COUNTER_STRIDE);
    HIP_TRY(ctx, hipMemcpyAsync(counters.data(), ctx->count.p, counter_bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    uint64_t total = 0, worst = 0;
    for (int s = 0; s < HIT_SHARDS; ++s) {
        total += counters[(size_t)s * HIT_COUNTER_STRIDE];
        worst = std::max<uint64_t>(worst, counters[(size_t)s * HIT_COUNTER_STRIDE]);
    }
    *n_hits = (int64_t)total;
    if ((int64_t)total > capacity || (int64_t)worst > shard_cap) {
        // ask for enough that every shard fits next time
        *n_hits = (int64_t)std::max<uint64_t>(total, worst * HIT_SHARDS);
        return fail(ctx, PFMSCAN_E_CAPACITY, "hit buffer too small: " + std::to_string(total) + " hits, capacity " + std::to_string(capacity));
    }
    if (total == 0) return PFMSCAN_OK;
    // shards -> one run in position order, on the device (pfmscan_sort.hip); three contiguous copies come back
    int key_bits = 1;
    while (key_bits < 63 && ((int64_t)1 << key_bits) < n_pos) ++key_bits;
    size_t temp_bytes = 0;
    HIP_TRY(ctx, sort_temp_bytes((int64_t)total, key_bits, &temp_bytes));
    if ((rc = ensure(ctx, ctx->sort_keys_in, total * 8))) return rc;
    if ((rc = ensure(ctx, ctx->sort_keys_out, total * 8))) return rc;
    if ((rc = ensure(ctx, ctx->sort_vals_in, total * 8))) return rc;
    if ((rc = ensure(ctx, ctx->sort_vals_out, total * 8))) return rc;
    if ((rc = ensure(ctx, ctx->sort_temp, std::max<size_t>(temp_bytes, 256)))) return rc;
    if ((rc = ensure(ctx, ctx->sort_seq, total * 4))) return rc;
    if ((rc = ensure(ctx, ctx->sort_struct, total * 8))) return rc;
    GatherArgs g;
    g.hit_pos = (const int64_t *)ctx->hit_pos.p;
    g.hit_seq = has_seq ? (const float *)ctx->hit_seq.p : nullptr;
    g.hit_struct = has_struct ? (const double *)ctx->hit_struct.p : nullptr;
    g.counts = (const unsigned long long *)ctx->count.p;
    g.shards = HIT_SHARDS;
    g.shard_cap = shard_cap;
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
    {
        hipError_t e = launch_gather_sorted(g, ctx->stream);
        if (e != hipSuccess) return fail_hip(ctx, e, "gather + sort of the hits");
    }
    HIP_TRY(ctx, hipMemcpyAsync(hit_pos, g.keys_out, total * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (hit_seq && has_seq) HIP_TRY(ctx, hipMemcpyAsync(hit_seq, g.seq_out, total * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (hit_struct && has_struct) HIP_TRY(ctx, hipMemcpyAsync(hit_struct, g.struct_out, total * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (hit_seq && !has_seq) std::fill(hit_seq, hit_seq + total, NAN);
    if (hit_struct && !has_struct) std::fill(hit_struct, hit_struct + total, (double)NAN);
    return PFMSCAN_OK;
}

extern "C" {

int pfmscan_scan_dev(pfmscan_ctx *ctx, const pfmscan_motif *mo, const uint8_t *d_codes, const void *d_profile,
                     int profile_dtype, int64_t n_pos, float *d_out_seq, double *d_out_struct, void *stream)
{
    ScanArgs a;
    int rc = check_and_fill(ctx, mo, d_codes, d_profile, profile_dtype, n_pos, a);
    if (rc) return rc;
    if (!d_out_seq && !d_out_struct) return fail(ctx, PFMSCAN_E_BADARG, "no output array given");
    if (d_out_seq && !mo->d_letters) return fail(ctx, PFMSCAN_E_BADARG, "out_seq requested but the motif has no letter table");
    if (d_out_struct && !mo->d_struct) return fail(ctx, PFMSCAN_E_BADARG, "out_struct requested but the motif has no structure PSSM");
    if (misaligned(d_out_seq) || misaligned(d_out_struct)) return fail(ctx, PFMSCAN_E_BADSHAPE, "output pointers must be 16-byte aligned");
    a.out_seq = d_out_seq;
    a.out_struct = d_out_struct;
    if (mo->d_struct && !d_out_struct) {
        // sequence scores only from a combined motif: run the letters kernel alone
        a.struct_pssm = nullptr;
        a.profile = nullptr;
    }
    if (mo->d_struct && d_out_struct && !d_out_seq) a.letter_table = nullptr;   // structure scores only
    return do_launch(ctx, a, stream);
}

int pfmscan_scan_letters_f64_dev(pfmscan_ctx *ctx, const pfmscan_motif *mo, const uint8_t *d_codes, int64_t n_pos,
                                 double *d_out, void *stream)
{
    ScanArgs a;
    int rc = check_and_fill(ctx, mo, d_codes, nullptr, PFMSCAN_PROFILE_NONE, n_pos, a);
    if (rc) return rc;
    if (!mo->d_letters) return fail(ctx, PFMSCAN_E_BADARG, "motif has no letter table");
    if (!d_out) return fail(ctx, PFMSCAN_E_BADARG, "no output array given");
    if (misaligned(d_out)) return fail(ctx, PFMSCAN_E_BADSHAPE, "output pointers must be 16-byte aligned");
    a.struct_pssm = nullptr;
    a.profile = nullptr;
    a.out_letters_f64 = d_out;
    return do_launch(ctx, a, stream);
}

int pfmscan_hits_dev(pfmscan_ctx *ctx, const pfmscan_motif *mo, const uint8_t *d_codes, const void *d_profile,
                     int profile_dtype, int64_t n_pos, double thr_seq, double thr_struct, int64_t capacity,
                     int64_t *d_hit_pos, float *d_hit_seq, double *d_hit_struct, uint64_t *d_hit_count, void *stream)
{
    ScanArgs a;
    int rc = check_and_fill(ctx, mo, d_codes, d_profile, profile_dtype, n_pos, a);
    if (rc) return rc;
    if (capacity < 0 || !d_hit_count || (capacity > 0 && !d_hit_pos))
        return fail(ctx, PFMSCAN_E_BADARG, "pfmscan_hits_dev: bad hit buffers");
    if (std::isnan(thr_seq) || std::isnan(thr_struct)) return fail(ctx, PFMSCAN_E_BADARG, "NaN threshold");
    a.hits = 1;
    a.thr_seq = thr_seq;
    a.thr_struct = thr_struct;
    a.capacity = capacity;
    a.hit_pos = d_hit_pos;
    a.hit_seq = mo->d_letters ? d_hit_seq : nullptr;
    a.hit_struct = mo->d_struct ? d_hit_struct : nullptr;
    a.hit_count = reinterpret_cast<unsigned long long *>(d_hit_count);
    a.hit_shards = 1;
    return do_launch(ctx, a, stream);
}

// ---- combined hits, candidate-then-verify ---------------------------------------------
struct HitSink {                 // where hits go: `shards` regions of `shard_cap` slots, one counter per region
    int64_t *pos;
    float *seq;
    double *st;
    unsigned long long *count;   // shards counters, HIT_COUNTER_STRIDE words apart (zeroed by the caller)
    int shards;
    int64_t shard_cap;
};

static void fill_sink(ScanArgs &a, const pfmscan_motif *mo, const HitSink &k, double thr_seq, double thr_struct)
{
    a.hits = 1;
    a.thr_seq = thr_seq;
    a.thr_struct = thr_struct;
    a.capacity = k.shard_cap;
    a.hit_pos = k.pos;
    a.hit_seq = mo->d_letters ? k.seq : nullptr;
    a.hit_struct = mo->d_struct ? k.st : nullptr;
    a.hit_count = k.count;
    a.hit_shards = k.shards;
}

// One fused pass, or -- when the motif has both parts and the letter threshold is selective --
// letters pass + structure verification at its hits.  Synchronises `st` when it takes two passes.
static int hits_core(pfmscan_ctx *ctx, const pfmscan_motif *mo, const ScanArgs &base, double thr_seq, double thr_struct,
                     const HitSink &sink, hipStream_t st, bool allow_two_phase)
{
    const int64_t n_pos = base.n_pos;
    const bool two = allow_two_phase && ctx->tune.two_phase && mo->d_letters && mo->d_struct && !std::isinf(thr_seq) && n_pos > 0;
    ScanArgs fused = base;
    fill_sink(fused, mo, sink, thr_seq, thr_struct);
    if (!two) return do_launch(ctx, fused, st);
    // phase 1: letters only (1 B per position) -> candidates.  Measured on C3 (w = 12): letters pass 0.23 ms,
    // verify ~0.1 ms per 1 M candidates (0.3 % of the windows), fused pass 2.1 ms -> two passes pay while
    // <= ~1/16 of the windows pass the letter threshold; the candidate buffers are sized for 1/32.
    // A pilot over a prefix of the stream estimates that rate first.
    int rc;
    const int64_t cand_cap = std::max<int64_t>(n_pos / 32, 1024);
    const int64_t cand_shard_cap = std::min<int64_t>(cand_cap, cand_cap / HIT_SHARDS * 2 + 4096);
    const size_t cand_slots = (size_t)cand_shard_cap * HIT_SHARDS;
    const size_t counter_bytes = (size_t)HIT_SHARDS * HIT_COUNTER_STRIDE * 8;
    if ((rc = ensure(ctx, ctx->cand_pos, cand_slots * 8))) return rc;
    if ((rc = ensure(ctx, ctx->cand_seq, cand_slots * 4))) return rc;
    if ((rc = ensure(ctx, ctx->cand_count, counter_bytes))) return rc;
    ScanArgs a1 = base;
    a1.struct_pssm = nullptr;
    a1.profile = nullptr;
    HitSink cs = {(int64_t *)ctx->cand_pos.p, (float *)ctx->cand_seq.p, nullptr, (unsigned long long *)ctx->cand_count.p,
                  HIT_SHARDS, cand_shard_cap};
    pfmscan_motif letters_only = *mo;
    letters_only.d_struct = nullptr;
    fill_sink(a1, &letters_only, cs, thr_seq, -INFINITY);
    std::vector<unsigned long long> counters((size_t)HIT_SHARDS * HIT_COUNTER_STRIDE);
    auto read_counts = [&](uint64_t &total, uint64_t &worst) -> int {
        HIP_TRY(ctx, hipMemcpyAsync(counters.data(), ctx->cand_count.p, counter_bytes, hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipStreamSynchronize(st));
        total = worst = 0;
        for (int s = 0; s < HIT_SHARDS; ++s) {
            total += counters[(size_t)s * HIT_COUNTER_STRIDE];
            worst = std::max<uint64_t>(worst, counters[(size_t)s * HIT_COUNTER_STRIDE]);
        }
        return PFMSCAN_OK;
    };
    uint64_t n_cand = 0, worst = 0;
    const int64_t pilot_n = std::max<int64_t>((int64_t)1 << 22, n_pos / 64);
    if (pilot_n < n_pos && !ctx->two_phase_hot) {
        ScanArgs ap = a1;
        ap.n_pos = pilot_n;
        HIP_TRY(ctx, hipMemsetAsync(ctx->cand_count.p, 0, counter_bytes, st));
        if ((rc = do_launch(ctx, ap, st))) return rc;
        if ((rc = read_counts(n_cand, worst))) return rc;
        if ((int64_t)n_cand * 32 > pilot_n) {                                      // not selective: one fused pass
            ctx->two_phase_hot = false;
            return do_launch(ctx, fused, st);
        }
    }
    HIP_TRY(ctx, hipMemsetAsync(ctx->cand_count.p, 0, counter_bytes, st));
    if ((rc = do_launch(ctx, a1, st))) return rc;
    if ((rc = read_counts(n_cand, worst))) return rc;
    if ((int64_t)worst > cand_shard_cap) {                                          // pilot under-estimated
        ctx->two_phase_hot = false;
        return do_launch(ctx, fused, st);
    }
    // a library scan calls this once per motif with similar selectivity: skip the pilot while the letters
    // pass keeps coming back well under the candidate capacity
    ctx->two_phase_hot = (int64_t)n_cand * 4 < cand_cap;
    if (n_cand == 0) return PFMSCAN_OK;
    // phase 2: structure score at the candidates only
    hipError_t e = launch_struct_at(fused, (const int64_t *)ctx->cand_pos.p, (const float *)ctx->cand_seq.p,
                                    (const unsigned long long *)ctx->cand_count.p, HIT_SHARDS, cand_shard_cap, st);
    if (e != hipSuccess) return fail_hip(ctx, e, "launch k_struct_at");
    return PFMSCAN_OK;
}

int pfmscan_hits_adaptive_dev(pfmscan_ctx *ctx, const pfmscan_motif *mo, const uint8_t *d_codes, const void *d_profile,
                              int profile_dtype, int64_t n_pos, double thr_seq, double thr_struct, int64_t capacity,
                              int64_t *d_hit_pos, float *d_hit_seq, double *d_hit_struct, uint64_t *d_hit_count,
                              void *stream)
{
    ScanArgs a;
    int rc = check_and_fill(ctx, mo, d_codes, d_profile, profile_dtype, n_pos, a);
    if (rc) return rc;
    if (capacity < 0 || !d_hit_count || (capacity > 0 && !d_hit_pos)) return fail(ctx, PFMSCAN_E_BADARG, "bad hit buffers");
    if (std::isnan(thr_seq) || std::isnan(thr_struct)) return fail(ctx, PFMSCAN_E_BADARG, "NaN threshold");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HitSink sink = {d_hit_pos, d_hit_seq, d_hit_struct, reinterpret_cast<unsigned long long *>(d_hit_count), 1, capacity};
    return hits_core(ctx, mo, a, thr_seq, thr_struct, sink, stream ? (hipStream_t)stream : ctx->stream, true);
}

// ---- staged stream + host-buffer forms -------------------------------------------
int pfmscan_stage(pfmscan_ctx *ctx, const uint8_t *codes, const void *profile, int profile_dtype, int64_t n_pos)
{
    if (!ctx) return fail(ctx, PFMSCAN_E_BADARG, "NULL ctx");
    if (n_pos < 0) return fail(ctx, PFMSCAN_E_BADARG, "negative n_pos");
    ctx->staged_n = -1;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (codes && n_pos > 0) {
        int rc = ensure(ctx, ctx->codes, (size_t)n_pos);
        if (rc) return rc;
        if ((rc = upload(ctx, ctx->codes.p, codes, (size_t)n_pos, ctx->stream))) return rc;
    }
    if (profile && n_pos > 0) {
        if (profile_dtype != PFMSCAN_PROFILE_F32 && profile_dtype != PFMSCAN_PROFILE_F64)
            return fail(ctx, PFMSCAN_E_BADARG, "profile_dtype must be F32 or F64");
        size_t bytes = (size_t)n_pos * 7 * (profile_dtype == PFMSCAN_PROFILE_F32 ? 4 : 8);
        int rc = ensure(ctx, ctx->profile, bytes);
        if (rc) return rc;
        if ((rc = upload(ctx, ctx->profile.p, profile, bytes, ctx->stream))) return rc;
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));     // the caller may reuse its buffers
    ctx->staged_n = n_pos;
    ctx->staged_dtype = profile ? profile_dtype : PFMSCAN_PROFILE_NONE;
    ctx->staged_codes = codes != nullptr;
    ctx->staged_profile = profile != nullptr;
    ctx->staged_codes2 = false;
    return PFMSCAN_OK;
}

static int check_staged(pfmscan_ctx *ctx, const pfmscan_motif *mo)
{
    if (!ctx || !mo) return fail(ctx, PFMSCAN_E_BADARG, "NULL ctx or motif");
    if (ctx->staged_n < 0) return fail(ctx, PFMSCAN_E_BADARG, "no stream staged (call pfmscan_stage first)");
    if (mo->d_letters && !ctx->staged_codes && ctx->staged_n > 0) return fail(ctx, PFMSCAN_E_BADARG, "motif has a letter table but no codes are staged");
    if (mo->d_struct && !ctx->staged_profile && ctx->staged_n > 0) return fail(ctx, PFMSCAN_E_BADARG, "motif has a structure PSSM but no profile is staged");
    return PFMSCAN_OK;
}

int64_t pfmscan_staged_positions(const pfmscan_ctx *ctx) { return ctx ? ctx->staged_n : -1; }

int pfmscan_scan_staged(pfmscan_ctx *ctx, const pfmscan_motif *mo, float *out_seq, double *out_struct)
{
    int rc = check_staged(ctx, mo);
    if (rc) return rc;
    const int64_t n_pos = ctx->staged_n;
    if (n_pos == 0) return PFMSCAN_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (out_seq && (rc = ensure(ctx, ctx->out_seq, (size_t)n_pos * 4))) return rc;
    if (out_struct && (rc = ensure(ctx, ctx->out_struct, (size_t)n_pos * 8))) return rc;
    rc = pfmscan_scan_dev(ctx, mo, (const uint8_t *)ctx->codes.p, ctx->profile.p, ctx->staged_dtype, n_pos,
                          out_seq ? (float *)ctx->out_seq.p : nullptr, out_struct ? (double *)ctx->out_struct.p : nullptr,
                          ctx->stream);
    if (rc) return rc;
    if (out_seq) HIP_TRY(ctx, hipMemcpyAsync(out_seq, ctx->out_seq.p, (size_t)n_pos * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (out_struct) HIP_TRY(ctx, hipMemcpyAsync(out_struct, ctx->out_struct.p, (size_t)n_pos * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PFMSCAN_OK;
}

int pfmscan_scan_host(pfmscan_ctx *ctx, const pfmscan_motif *mo, const uint8_t *codes, const void *profile,
                      int profile_dtype, int64_t n_pos, float *out_seq, double *out_struct)
{
    if (!ctx || !mo) return fail(ctx, PFMSCAN_E_BADARG, "NULL ctx or motif");
    if (n_pos < 0) return fail(ctx, PFMSCAN_E_BADARG, "negative n_pos");
    if (n_pos == 0) return PFMSCAN_OK;
    if (mo->d_letters && !codes) return fail(ctx, PFMSCAN_E_BADARG, "codes is NULL");
    if (mo->d_struct && !profile) return fail(ctx, PFMSCAN_E_BADARG, "profile is NULL");
    int rc = pfmscan_stage(ctx, mo->d_letters ? codes : nullptr, mo->d_struct ? profile : nullptr, profile_dtype, n_pos);
    if (rc) return rc;
    return pfmscan_scan_staged(ctx, mo, out_seq, out_struct);
}

int pfmscan_scan_letters_f64_host(pfmscan_ctx *ctx, const pfmscan_motif *mo, const uint8_t *codes, int64_t n_pos, double *out)
{
    if (!ctx || !mo || !out) return fail(ctx, PFMSCAN_E_BADARG, "NULL argument");
    if (!mo->d_letters) return fail(ctx, PFMSCAN_E_BADARG, "motif has no letter table");
    if (n_pos < 0) return fail(ctx, PFMSCAN_E_BADARG, "negative n_pos");
    if (n_pos == 0) return PFMSCAN_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!codes) return fail(ctx, PFMSCAN_E_BADARG, "codes is NULL");
    ctx->staged_n = -1;                                   // the scratch is about to be overwritten
    int rc = ensure(ctx, ctx->codes, (size_t)n_pos);
    if (rc) return rc;
    if ((rc = upload(ctx, ctx->codes.p, codes, (size_t)n_pos, ctx->stream))) return rc;
    if ((rc = ensure(ctx, ctx->out_struct, (size_t)n_pos * 8))) return rc;
    rc = pfmscan_scan_letters_f64_dev(ctx, mo, (const uint8_t *)ctx->codes.p, n_pos, (double *)ctx->out_struct.p, ctx->stream);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(out, ctx->out_struct.p, (size_t)n_pos * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return PFMSCAN_OK;
}

int pfmscan_hits_staged(pfmscan_ctx *ctx, const pfmscan_motif *mo, double thr_seq, double thr_struct, int64_t capacity,
                        int64_t *hit_pos, float *hit_seq, double *hit_struct, int64_t *n_hits)
{
    if (!n_hits) return fail(ctx, PFMSCAN_E_BADARG, "NULL argument");
    int rc = check_staged(ctx, mo);
    if (rc) return rc;
    if (capacity < 0) return fail(ctx, PFMSCAN_E_BADARG, "negative size");
    if (std::isnan(thr_seq) || std::isnan(thr_struct)) return fail(ctx, PFMSCAN_E_BADARG, "NaN threshold");
    *n_hits = 0;
    const int64_t n_pos = ctx->staged_n;
    if (n_pos == 0) return PFMSCAN_OK;
    if (capacity > 0 && !hit_pos) return fail(ctx, PFMSCAN_E_BADARG, "hit_pos is NULL");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // ctx-owned, sharded hit buffers: shard s = workgroup & 31 gets every 32nd tile, so the shards fill
    // evenly; each has room for 2x its share
    // (small streams have few workgroups, i.e. few shards in use: there every shard can take everything)
    const int64_t shard_cap = std::max<int64_t>(std::min<int64_t>(capacity, capacity / HIT_SHARDS * 2 + 4096), 1);
    const size_t slots = (size_t)shard_cap * HIT_SHARDS;
    const size_t counter_bytes = (size_t)HIT_SHARDS * HIT_COUNTER_STRIDE * 8;
    if ((rc = ensure(ctx, ctx->hit_pos, slots * 8))) return rc;
    if ((rc = ensure(ctx, ctx->hit_seq, slots * 4))) return rc;
    if ((rc = ensure(ctx, ctx->hit_struct, slots * 8))) return rc;
    if ((rc = ensure(ctx, ctx->count, counter_bytes))) return rc;
    HIP_TRY(ctx, hipMemsetAsync(ctx->count.p, 0, counter_bytes, ctx->stream));
    ScanArgs a;
    if ((rc = check_and_fill(ctx, mo, (const uint8_t *)ctx->codes.p, ctx->profile.p, ctx->staged_dtype, n_pos, a))) return rc;
    HitSink sink = {(int64_t *)ctx->hit_pos.p, (float *)ctx->hit_seq.p, (double *)ctx->hit_struct.p,
                    (unsigned long long *)ctx->count.p, HIT_SHARDS, shard_cap};
    if ((rc = hits_core(ctx, mo, a, thr_seq, thr_struct, sink, ctx->stream, true))) return rc;
    return finish_sorted_hits(ctx, mo->d_letters != nullptr, mo->d_struct != nullptr, n_pos, capacity, shard_cap, hit_pos, hit_seq,
                              hit_struct, n_hits);
}

int pfmscan_hits_host(pfmscan_ctx *ctx, const pfmscan_motif *mo, const uint8_t *codes, const void *profile,
                      int profile_dtype, int64_t n_pos, double thr_seq, double thr_struct, int64_t capacity,
                      int64_t *hit_pos, float *hit_seq, double *hit_struct, int64_t *n_hits)
{
    if (!ctx || !mo || !n_hits) return fail(ctx, PFMSCAN_E_BADARG, "NULL argument");
    if (n_pos < 0 || capacity < 0) return fail(ctx, PFMSCAN_E_BADARG, "negative size");
    *n_hits = 0;
    if (n_pos == 0) return PFMSCAN_OK;
    if (mo->d_letters && !codes) return fail(ctx, PFMSCAN_E_BADARG, "codes is NULL");
    if (mo->d_struct && !profile) return fail(ctx, PFMSCAN_E_BADARG, "profile is NULL");
    int rc = pfmscan_stage(ctx, mo->d_letters ? codes : nullptr, mo->d_struct ? profile : nullptr, profile_dtype, n_pos);
    if (rc) return rc;
    return pfmscan_hits_staged(ctx, mo, thr_seq, thr_struct, capacity, hit_pos, hit_seq, hit_struct, n_hits);
}

// ---- generic-alphabet letter hits in fp64 (structure letter strings: SURVEY 8f N4) ---------------------------
static int fill_f64_hits(pfmscan_ctx *ctx, const pfmscan_motif *mo, const uint8_t *d_codes, int64_t n_pos, double thr, ScanArgs &a)
{
    int rc = check_and_fill(ctx, mo, d_codes, nullptr, PFMSCAN_PROFILE_NONE, n_pos, a);
    if (rc) return rc;
    if (!mo->d_letters) return fail(ctx, PFMSCAN_E_BADARG, "motif has no letter table");
    if (std::isnan(thr)) return fail(ctx, PFMSCAN_E_BADARG, "NaN threshold");
    a.struct_pssm = nullptr;
    a.profile = nullptr;
    a.hits = 1;
    a.f64_hits = 1;
    a.thr_seq = thr;
    a.thr_struct = -INFINITY;
    return PFMSCAN_OK;
}

int pfmscan_hits_letters_f64_dev(pfmscan_ctx *ctx, const pfmscan_motif *mo, const uint8_t *d_codes, int64_t n_pos, double thr,
                                 int64_t capacity, int64_t *d_hit_pos, double *d_hit_score, uint64_t *d_hit_count, void *stream)
{
    ScanArgs a;
    int rc = fill_f64_hits(ctx, mo, d_codes, n_pos, thr, a);
    if (rc) return rc;
    if (capacity < 0 || !d_hit_count || (capacity > 0 && !d_hit_pos)) return fail(ctx, PFMSCAN_E_BADARG, "pfmscan_hits_letters_f64_dev: bad hit buffers");
    a.capacity = capacity;
    a.hit_pos = d_hit_pos;
    a.hit_seq = nullptr;
    a.hit_struct = d_hit_score;
    a.hit_count = reinterpret_cast<unsigned long long *>(d_hit_count);
    a.hit_shards = 1;
    return do_launch(ctx, a, stream);
}

int pfmscan_hits_letters_f64_staged(pfmscan_ctx *ctx, const pfmscan_motif *mo, double thr, int64_t capacity, int64_t *hit_pos,
                                    double *hit_score, int64_t *n_hits)
{
    if (!n_hits) return fail(ctx, PFMSCAN_E_BADARG, "NULL argument");
    int rc = check_staged(ctx, mo);
    if (rc) return rc;
    if (!mo->d_letters) return fail(ctx, PFMSCAN_E_BADARG, "motif has no letter table");
    if (capacity < 0) return fail(ctx, PFMSCAN_E_BADARG, "negative size");
    *n_hits = 0;
    const int64_t n_pos = ctx->staged_n;
    if (n_pos == 0) return PFMSCAN_OK;
    if (capacity > 0 && !hit_pos) return fail(ctx, PFMSCAN_E_BADARG, "hit_pos is NULL");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int64_t shard_cap = std::max<int64_t>(std::min<int64_t>(capacity, capacity / HIT_SHARDS * 2 + 4096), 1);
    const size_t slots = (size_t)shard_cap * HIT_SHARDS;
    const size_t counter_bytes = (size_t)HIT_SHARDS * HIT_COUNTER_STRIDE * 8;
    if ((rc = ensure(ctx, ctx->hit_pos, slots * 8))) return rc;
    if ((rc = ensure(ctx, ctx->hit_struct, slots * 8))) return rc;
    if ((rc = ensure(ctx, ctx->count, counter_bytes))) return rc;
    HIP_TRY(ctx, hipMemsetAsync(ctx->count.p, 0, counter_bytes, ctx->stream));
    ScanArgs a;
    if ((rc = fill_f64_hits(ctx, mo, (const uint8_t *)ctx->codes.p, n_pos, thr, a))) return rc;
    a.capacity = shard_cap;
    a.hit_pos = (int64_t *)ctx->hit_pos.p;
    a.hit_seq = nullptr;
    a.hit_struct = (double *)ctx->hit_struct.p;
    a.hit_count = (unsigned long long *)ctx->count.p;
    a.hit_shards = HIT_SHARDS;
    if ((rc = do_launch(ctx, a, ctx->stream))) return rc;
    return finish_sorted_hits(ctx, false, true, n_pos, capacity, shard_cap, hit_pos, nullptr, hit_score, n_hits);
}

int pfmscan_hits_letters_f64_host(pfmscan_ctx *ctx, const pfmscan_motif *mo, const uint8_t *codes, int64_t n_pos, double thr,
                                  int64_t capacity, int64_t *hit_pos, double *hit_score, int64_t *n_hits)
{
    if (!ctx || !mo || !n_hits) return fail(ctx, PFMSCAN_E_BADARG, "NULL argument");
    if (n_pos < 0 || capacity < 0) return fail(ctx, PFMSCAN_E_BADARG, "negative size");
    *n_hits = 0;
    if (n_pos == 0) return PFMSCAN_OK;
    if (!codes) return fail(ctx, PFMSCAN_E_BADARG, "codes is NULL");
    int rc = pfmscan_stage(ctx, codes, nullptr, PFMSCAN_PROFILE_NONE, n_pos);
    if (rc) return rc;
    return pfmscan_hits_letters_f64_staged(ctx, mo, thr, capacity, hit_pos, hit_score, n_hits);
}

// ---- two code streams: sequence letters AND structure letters (two-FASTA RNASS mode) ---------------------------
// Phase 1: the sequence letters pass over everything (any of the letters hits kernels) -> candidates in the ctx's
// candidate buffers; phase 2: k_letters_at scores the structure letters at the candidates only.  The candidate buffers
// start at 1/32 of the windows; a denser threshold is retried once with the exact sizes the counters reported.
static int pair_core(pfmscan_ctx *ctx, const pfmscan_motif *mo_seq, const pfmscan_motif *mo_st, const uint8_t *d_codes,
                     const uint8_t *d_codes2, int64_t n_pos, double thr_seq, double thr_struct, const HitSink &sink, hipStream_t st)
{
    int rc;
    ScanArgs a1, a2;
    if ((rc = check_and_fill(ctx, mo_seq, d_codes, nullptr, PFMSCAN_PROFILE_NONE, n_pos, a1))) return rc;
    if ((rc = check_and_fill(ctx, mo_st, d_codes2, nullptr, PFMSCAN_PROFILE_NONE, n_pos, a2))) return rc;
    a1.struct_pssm = a2.struct_pssm = nullptr;
    a1.profile = a2.profile = nullptr;
    // the usual case -- a selective finite threshold on a PFM up to 32 wide -- is ONE launch: the integer-prefiltered letters
    // kernel verifies the second stream for its own survivors (k_letters_cred<.., PAIR>); PFMSCAN_PAIR_TWO_PHASE=1: A/B
    if (!std::getenv("PFMSCAN_PAIR_TWO_PHASE")) {
        ScanArgs f = a1;
        pfmscan_motif both1 = *mo_seq;
        both1.d_struct = mo_st->d_letters;
        fill_sink(f, &both1, sink, thr_seq, thr_struct);
        f.codes2 = d_codes2;
        f.letter_table2 = mo_st->d_letters;
        hipError_t e1 = hipSuccess;
        HIP_TRY(ctx, hipSetDevice(ctx->device));
        if (launch_letters_cred(f, ctx->tune, st, &e1)) {
            if (e1 != hipSuccess) return fail_hip(ctx, e1, "launch k_letters_cred (pair)");
            return PFMSCAN_OK;
        }
    }
    const size_t counter_bytes = (size_t)HIT_SHARDS * HIT_COUNTER_STRIDE * 8;
    if ((rc = ensure(ctx, ctx->cand_count, counter_bytes))) return rc;
    std::vector<unsigned long long> counters((size_t)HIT_SHARDS * HIT_COUNTER_STRIDE);
    pfmscan_motif letters_only = *mo_seq;
    letters_only.d_struct = nullptr;
    int64_t cand_shard_cap = std::max<int64_t>(n_pos / 32 / HIT_SHARDS * 2 + 4096, 1);
    uint64_t n_cand = 0, worst = 0;
    for (int attempt = 0; attempt < 2; ++attempt) {
        const size_t cand_slots = (size_t)cand_shard_cap * HIT_SHARDS;
        if ((rc = ensure(ctx, ctx->cand_pos, cand_slots * 8))) return rc;
        if ((rc = ensure(ctx, ctx->cand_seq, cand_slots * 4))) return rc;
        HitSink cs = {(int64_t *)ctx->cand_pos.p, (float *)ctx->cand_seq.p, nullptr, (unsigned long long *)ctx->cand_count.p,
                      HIT_SHARDS, cand_shard_cap};
        ScanArgs p1 = a1;
        fill_sink(p1, &letters_only, cs, thr_seq, -INFINITY);
        HIP_TRY(ctx, hipMemsetAsync(ctx->cand_count.p, 0, counter_bytes, st));
        if ((rc = do_launch(ctx, p1, st))) return rc;
        HIP_TRY(ctx, hipMemcpyAsync(counters.data(), ctx->cand_count.p, counter_bytes, hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipStreamSynchronize(st));
        n_cand = worst = 0;
        for (int s = 0; s < HIT_SHARDS; ++s) {
            n_cand += counters[(size_t)s * HIT_COUNTER_STRIDE];
            worst = std::max<uint64_t>(worst, counters[(size_t)s * HIT_COUNTER_STRIDE]);
        }
        if ((int64_t)worst <= cand_shard_cap) break;
        if (attempt == 1) return fail(ctx, PFMSCAN_E_HIP, "candidate counts changed between two identical passes");
        cand_shard_cap = (int64_t)worst;                    // the same pass again, every shard sized for what it reported
    }
    if (n_cand == 0) return PFMSCAN_OK;
    pfmscan_motif both = *mo_seq;                          // fill_sink: which hit arrays exist
    both.d_struct = mo_st->d_letters;
    fill_sink(a2, &both, sink, thr_seq, thr_struct);
    hipError_t e = launch_letters_at(a2, (const int64_t *)ctx->cand_pos.p, (const float *)ctx->cand_seq.p,
                                     (const unsigned long long *)ctx->cand_count.p, HIT_SHARDS, cand_shard_cap, st);
    if (e != hipSuccess) return fail_hip(ctx, e, "launch k_letters_at");
    return PFMSCAN_OK;
}

static int check_pair(pfmscan_ctx *ctx, const pfmscan_motif *mo_seq, const pfmscan_motif *mo_st, double thr_seq, double thr_struct)
{
    if (!ctx || !mo_seq || !mo_st) return fail(ctx, PFMSCAN_E_BADARG, "NULL ctx or motif");
    if (mo_seq->ctx != ctx || mo_st->ctx != ctx) return fail(ctx, PFMSCAN_E_BADARG, "motif belongs to another ctx");
    if (!mo_seq->d_letters || !mo_st->d_letters) return fail(ctx, PFMSCAN_E_BADARG, "both motifs need a letter table");
    if (mo_seq->m != mo_st->m)
        return fail(ctx, PFMSCAN_E_BADSHAPE, "sequence and structure PFMs must have the same width for a combined scan");
    if (std::isnan(thr_seq) || std::isnan(thr_struct)) return fail(ctx, PFMSCAN_E_BADARG, "NaN threshold");
    return PFMSCAN_OK;
}

int pfmscan_hits_pair_dev(pfmscan_ctx *ctx, const pfmscan_motif *mo_seq, const pfmscan_motif *mo_st, const uint8_t *d_codes,
                          const uint8_t *d_codes2, int64_t n_pos, double thr_seq, double thr_struct, int64_t capacity,
                          int64_t *d_hit_pos, float *d_hit_seq, double *d_hit_struct, uint64_t *d_hit_count, void *stream)
{
    int rc = check_pair(ctx, mo_seq, mo_st, thr_seq, thr_struct);
    if (rc) return rc;
    if (n_pos < 0) return fail(ctx, PFMSCAN_E_BADARG, "negative n_pos");
    if (capacity < 0 || !d_hit_count || (capacity > 0 && !d_hit_pos)) return fail(ctx, PFMSCAN_E_BADARG, "pfmscan_hits_pair_dev: bad hit buffers");
    if (n_pos == 0) return PFMSCAN_OK;
    if (!d_codes || !d_codes2) return fail(ctx, PFMSCAN_E_BADARG, "codes is NULL");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HitSink sink = {d_hit_pos, d_hit_seq, d_hit_struct, reinterpret_cast<unsigned long long *>(d_hit_count), 1, capacity};
    return pair_core(ctx, mo_seq, mo_st, d_codes, d_codes2, n_pos, thr_seq, thr_struct, sink, stream ? (hipStream_t)stream : ctx->stream);
}

int pfmscan_stage_codes2(pfmscan_ctx *ctx, const uint8_t *codes2, int64_t n_pos)
{
    if (!ctx) return fail(ctx, PFMSCAN_E_BADARG, "NULL ctx");
    if (ctx->staged_n < 0 || !ctx->staged_codes) return fail(ctx, PFMSCAN_E_BADARG, "no code stream staged (call pfmscan_stage first)");
    if (n_pos != ctx->staged_n) return fail(ctx, PFMSCAN_E_BADARG, "the second code stream must have the staged stream's length");
    ctx->staged_codes2 = false;
    if (n_pos > 0) {
        if (!codes2) return fail(ctx, PFMSCAN_E_BADARG, "codes2 is NULL");
        HIP_TRY(ctx, hipSetDevice(ctx->device));
        int rc = ensure(ctx, ctx->codes2, (size_t)n_pos);
        if (rc) return rc;
        if ((rc = upload(ctx, ctx->codes2.p, codes2, (size_t)n_pos, ctx->stream))) return rc;
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    ctx->staged_codes2 = true;
    return PFMSCAN_OK;
}

int pfmscan_hits_pair_staged(pfmscan_ctx *ctx, const pfmscan_motif *mo_seq, const pfmscan_motif *mo_st, double thr_seq,
                             double thr_struct, int64_t capacity, int64_t *hit_pos, float *hit_seq, double *hit_struct,
                             int64_t *n_hits)
{
    if (!n_hits) return fail(ctx, PFMSCAN_E_BADARG, "NULL argument");
    int rc = check_pair(ctx, mo_seq, mo_st, thr_seq, thr_struct);
    if (rc) return rc;
    if (ctx->staged_n < 0 || !ctx->staged_codes || !ctx->staged_codes2)
        return fail(ctx, PFMSCAN_E_BADARG, "two code streams must be staged (pfmscan_stage + pfmscan_stage_codes2)");
    if (capacity < 0) return fail(ctx, PFMSCAN_E_BADARG, "negative size");
    *n_hits = 0;
    const int64_t n_pos = ctx->staged_n;
    if (n_pos == 0) return PFMSCAN_OK;
    if (capacity > 0 && !hit_pos) return fail(ctx, PFMSCAN_E_BADARG, "hit_pos is NULL");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int64_t shard_cap = std::max<int64_t>(std::min<int64_t>(capacity, capacity / HIT_SHARDS * 2 + 4096), 1);
    const size_t slots = (size_t)shard_cap * HIT_SHARDS;
    const size_t counter_bytes = (size_t)HIT_SHARDS * HIT_COUNTER_STRIDE * 8;
    if ((rc = ensure(ctx, ctx->hit_pos, slots * 8))) return rc;
    if ((rc = ensure(ctx, ctx->hit_seq, slots * 4))) return rc;
    if ((rc = ensure(ctx, ctx->hit_struct, slots * 8))) return rc;
    if ((rc = ensure(ctx, ctx->count, counter_bytes))) return rc;
    HIP_TRY(ctx, hipMemsetAsync(ctx->count.p, 0, counter_bytes, ctx->stream));
    HitSink sink = {(int64_t *)ctx->hit_pos.p, (float *)ctx->hit_seq.p, (double *)ctx->hit_struct.p,
                    (unsigned long long *)ctx->count.p, HIT_SHARDS, shard_cap};
    if ((rc = pair_core(ctx, mo_seq, mo_st, (const uint8_t *)ctx->codes.p, (const uint8_t *)ctx->codes2.p, n_pos, thr_seq, thr_struct,
                        sink, ctx->stream)))
        return rc;
    return finish_sorted_hits(ctx, true, true, n_pos, capacity, shard_cap, hit_pos, hit_seq, hit_struct, n_hits);
}

int pfmscan_hits_pair_host(pfmscan_ctx *ctx, const pfmscan_motif *mo_seq, const pfmscan_motif *mo_st, const uint8_t *codes,
                           const uint8_t *codes2, int64_t n_pos, double thr_seq, double thr_struct, int64_t capacity,
                           int64_t *hit_pos, float *hit_seq, double *hit_struct, int64_t *n_hits)
{
    if (!ctx || !mo_seq || !mo_st || !n_hits) return fail(ctx, PFMSCAN_E_BADARG, "NULL argument");
    if (n_pos < 0 || capacity < 0) return fail(ctx, PFMSCAN_E_BADARG, "negative size");
    *n_hits = 0;
    if (n_pos == 0) return PFMSCAN_OK;
    if (!codes || !codes2) return fail(ctx, PFMSCAN_E_BADARG, "codes is NULL");
    int rc = pfmscan_stage(ctx, codes, nullptr, PFMSCAN_PROFILE_NONE, n_pos);
    if (rc) return rc;
    if ((rc = pfmscan_stage_codes2(ctx, codes2, n_pos))) return rc;
    return pfmscan_hits_pair_staged(ctx, mo_seq, mo_st, thr_seq, thr_struct, capacity, hit_pos, hit_seq, hit_struct, n_hits);
}

// ---- the reference's native entry point ------------------------------------------
int pfmscan_pwm_calculate(pfmscan_ctx *ctx, const char *sequence, int64_t s, const double *matrix, int64_t m, float *out)
{
    if (!ctx) return fail(ctx, PFMSCAN_E_BADARG, "NULL ctx");
    if (!sequence || !matrix || s < 0) return fail(ctx, PFMSCAN_E_BADARG, "pfmscan_pwm_calculate: NULL or negative argument");
    if (m < 1 || m > PFMSCAN_MAX_WIDTH)
        return fail(ctx, PFMSCAN_E_BADSHAPE, "position-weight matrix width " + std::to_string(m) + " outside 1.." + std::to_string(PFMSCAN_MAX_WIDTH));
    const int64_t n = s - m + 1;
    if (n <= 0) return PFMSCAN_OK;
    if (!out) return fail(ctx, PFMSCAN_E_BADARG, "out is NULL");
    // letter -> code exactly as the switch of _pwm.c:41-63
    uint8_t lut[256];
    std::memset(lut, PFMSCAN_SEP, sizeof(lut));
    lut[(unsigned char)'A'] = lut[(unsigned char)'a'] = 0;
    lut[(unsigned char)'C'] = lut[(unsigned char)'c'] = 1;
    lut[(unsigned char)'G'] = lut[(unsigned char)'g'] = 2;
    lut[(unsigned char)'T'] = lut[(unsigned char)'t'] = 3;
    lut[(unsigned char)'U'] = lut[(unsigned char)'u'] = 3;
    std::vector<uint8_t> codes;
    std::vector<double> table((size_t)m * 8);
    try {
        codes.resize((size_t)s);
    } catch (const std::bad_alloc &) {
        return fail(ctx, PFMSCAN_E_OOM, "failed to create output data");
    }
    for (int64_t i = 0; i < s; ++i) codes[(size_t)i] = lut[(unsigned char)sequence[i]];
    for (int64_t j = 0; j < m; ++j) {
        for (int c = 0; c < 4; ++c) table[(size_t)j * 8 + c] = matrix[j * 4 + c];
        for (int c = 4; c < 8; ++c) table[(size_t)j * 8 + c] = NAN;
    }
    pfmscan_motif *mo = nullptr;
    int rc = pfmscan_motif_create(ctx, table.data(), nullptr, (int)m, &mo);
    if (rc) return rc;
    std::vector<float> full;
    try {
        full.resize((size_t)s);
    } catch (const std::bad_alloc &) {
        pfmscan_motif_destroy(mo);
        return fail(ctx, PFMSCAN_E_OOM, "failed to create output data");
    }
    rc = pfmscan_scan_host(ctx, mo, codes.data(), nullptr, PFMSCAN_PROFILE_NONE, s, full.data(), nullptr);
    pfmscan_motif_destroy(mo);
    if (rc) return rc;
    std::memcpy(out, full.data(), (size_t)n * sizeof(float));
    return PFMSCAN_OK;
}

// ---- measurement helper --------------------------------------------------------------
int pfmscan_time_scan_dev(pfmscan_ctx *ctx, const pfmscan_motif *mo, const uint8_t *d_codes, const void *d_profile,
                          int profile_dtype, int64_t n_pos, float *d_out_seq, double *d_out_struct, void *stream,
                          int warmup, int iters, double *avg_ms)
{
    if (!ctx || !avg_ms || iters < 1 || warmup < 0) return fail(ctx, PFMSCAN_E_BADARG, "pfmscan_time_scan_dev: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    for (int i = 0; i < warmup; ++i) {
        int rc = pfmscan_scan_dev(ctx, mo, d_codes, d_profile, profile_dtype, n_pos, d_out_seq, d_out_struct, st);
        if (rc) return rc;
    }
    hipEvent_t e0, e1;
    HIP_TRY(ctx, hipEventCreate(&e0));
    HIP_TRY(ctx, hipEventCreate(&e1));
    HIP_TRY(ctx, hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i) {
        int rc = pfmscan_scan_dev(ctx, mo, d_codes, d_profile, profile_dtype, n_pos, d_out_seq, d_out_struct, st);
        if (rc) {
            (void)hipEventDestroy(e0);
            (void)hipEventDestroy(e1);
            return rc;
        }
    }
    HIP_TRY(ctx, hipEventRecord(e1, st));
    HIP_TRY(ctx, hipEventSynchronize(e1));
    float ms = 0.f;
    HIP_TRY(ctx, hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *avg_ms = (double)ms / iters;
    return PFMSCAN_OK;
}

}  // extern "C"
