// pfmscan_upload.hip -- host memory -> device at the PCIe rate, whatever the host memory is.
// hipMemcpy from pageable memory pins or stages the user's pages on ONE runtime thread: anonymous memory it has seen
// before moves at 56 GB/s, a first pass at 46 GB/s and a freshly mapped file (a packed profile store) at 32 GB/s
// (profiles/r2/mmap_upload_probe.txt) -- the page faults of the mapping sit inside the copy.  Here the source is cut
// into 64-MiB pieces; a small pool of host threads copies piece i into one of three pinned buffers (the faults are
// taken in parallel) while the DMA engine moves piece i-1 from another: the transfer is asynchronous on the caller's
// stream and the source may be reused as soon as upload() returns.  Anonymous memory gains nothing (the runtime's
// in-place pinning already reaches the PCIe rate), so the staged path is a MODE the caller selects for sources it knows
// to be file mappings (pfmscan_set_upload_mode; rnascan_amd/_lib.py does it for numpy memmaps).
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "pfmscan_ctx.hpp"

namespace pfmscan {

typedef long long v4di __attribute__((vector_size(32), aligned(32)));
typedef long long v4di_u __attribute__((vector_size(32), aligned(1)));

// d is 4-KiB aligned (a slice of a pinned buffer cut at page multiples); s is whatever the caller has
static void copy_slice(unsigned char *d, const unsigned char *s, size_t n, bool nt)
{
    if (!nt) {
        std::memcpy(d, s, n);
        return;
    }
    size_t i = 0;
    for (; i + 128 <= n; i += 128) {
        const v4di_u *sv = reinterpret_cast<const v4di_u *>(s + i);
        v4di *dv = reinterpret_cast<v4di *>(d + i);
        const v4di a = sv[0], b = sv[1], c = sv[2], e = sv[3];
        __builtin_nontemporal_store(a, dv + 0);
        __builtin_nontemporal_store(b, dv + 1);
        __builtin_nontemporal_store(c, dv + 2);
        __builtin_nontemporal_store(e, dv + 3);
    }
    if (i < n) std::memcpy(d + i, s + i, n - i);
#if defined(__x86_64__)
    __builtin_ia32_sfence();
#endif
}

// the same bytes from the file behind the mapping: no page of the mapping is touched.  false: fall back to the mapping
static bool read_slice(unsigned char *d, int fd, int64_t off, size_t n)
{
    while (n > 0) {
        const ssize_t got = pread(fd, d, n, (off_t)off);
        if (got < 0 && errno == EINTR) continue;
        if (got <= 0) return false;                    // error or a file shorter than its mapping
        d += got;
        off += got;
        n -= (size_t)got;
    }
    return true;
}

struct Uploader {
    static constexpr int NBUF = 3;
    size_t PIECE = (size_t)64 << 20;
    bool nt = true;                        // streaming stores into the pinned buffer: no read-for-ownership, nothing left in the caches
    void *pinned[NBUF] = {nullptr, nullptr, nullptr};
    hipEvent_t done[NBUF] = {nullptr, nullptr, nullptr};
    bool busy[NBUF] = {false, false, false};
    int next = 0;
    bool usable = false;
    // worker pool: every worker copies its slice of [src, src + bytes) whenever `gen` moves on
    std::vector<std::thread> workers;
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    const unsigned char *src = nullptr;
    unsigned char *dst = nullptr;
    size_t bytes = 0;
    int src_fd = -1;                       // >= 0: [src, src + bytes) is file src_fd from src_off on
    int64_t src_off = 0;
    uint64_t gen = 0;
    int pending = 0;
    bool quit = false;

    void work(int w, int n)
    {
        uint64_t seen = 0;
        for (;;) {
            const unsigned char *s;
            unsigned char *d;
            size_t nb;
            int fd;
            int64_t foff;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_work.wait(lk, [&] { return quit || gen != seen; });
                if (quit) return;
                seen = gen;
                s = src;
                d = dst;
                nb = bytes;
                fd = src_fd;
                foff = src_off;
            }
            const size_t a = (nb * (size_t)w / (size_t)n) & ~(size_t)4095, b = w + 1 == n ? nb : (nb * (size_t)(w + 1) / (size_t)n) & ~(size_t)4095;
            if (b > a && !(fd >= 0 && read_slice(d + a, fd, foff + (int64_t)a, b - a))) copy_slice(d + a, s + a, b - a, nt);
            {
                std::lock_guard<std::mutex> lk(mu);
                if (--pending == 0) cv_done.notify_one();
            }
        }
    }

    void copy(void *d, const void *s, size_t nb, int fd = -1, int64_t foff = 0)
    {
        std::unique_lock<std::mutex> lk(mu);
        src = static_cast<const unsigned char *>(s);
        dst = static_cast<unsigned char *>(d);
        bytes = nb;
        src_fd = fd;
        src_off = foff;
        pending = (int)workers.size();
        ++gen;
        cv_work.notify_all();
        cv_done.wait(lk, [&] { return pending == 0; });
    }

    bool init()
    {
        int n = (int)std::min<unsigned>(std::max(2u, std::thread::hardware_concurrency()), 16u);
        if (const char *v = std::getenv("PFMSCAN_UPLOAD_THREADS")) n = std::max(1, std::atoi(v));
        if (const char *v = std::getenv("PFMSCAN_UPLOAD_PIECE_MB")) PIECE = (size_t)std::max(1, std::atoi(v)) << 20;
        if (const char *v = std::getenv("PFMSCAN_UPLOAD_NT")) nt = std::atoi(v) != 0;
        for (int i = 0; i < NBUF; ++i) {
            if (hipHostMalloc(&pinned[i], PIECE, hipHostMallocDefault) != hipSuccess) return false;
            if (hipEventCreateWithFlags(&done[i], hipEventDisableTiming) != hipSuccess) return false;
        }
        for (int w = 0; w < n; ++w) workers.emplace_back([this, w, n] { work(w, n); });
        usable = true;
        return true;
    }

    ~Uploader()
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            quit = true;
        }
        cv_work.notify_all();
        for (auto &t : workers) t.join();
        for (int i = 0; i < NBUF; ++i) {
            if (done[i]) {
                (void)hipEventSynchronize(done[i]);
                (void)hipEventDestroy(done[i]);
            }
            if (pinned[i]) (void)hipHostFree(pinned[i]);
        }
    }
};

int upload(pfmscan_ctx *ctx, void *d_dst, const void *h_src, size_t bytes, hipStream_t st)
{
    if (bytes == 0) return PFMSCAN_OK;
    // PFMSCAN_UPLOAD=0 / 1 overrides the mode the caller set (A/B of the two paths): never / always staged
    const char *env = std::getenv("PFMSCAN_UPLOAD");
    const int forced = env ? (std::atoi(env) != 0 ? 1 : 0) : -1;
    const bool enabled = forced >= 0 ? forced == 1 : ctx->upload_mode == PFMSCAN_UPLOAD_STAGED;
    // the pinned buffers and the threads cost ~30 ms once per context: worth it from a few hundred MB on
    const size_t min_bytes = forced == 1 ? (size_t)32 << 20 : (size_t)256 << 20;
    if (enabled && bytes >= min_bytes && !ctx->up) {
        ctx->up = new Uploader();
        if (!ctx->up->init()) {                 // no pinned memory to be had: the plain copy below still works
            delete ctx->up;
            ctx->up = new Uploader();           // usable == false: do not try again
        }
    }
    Uploader *up = ctx->up;
    if (!enabled || bytes < min_bytes || !up || !up->usable) {
        HIP_TRY(ctx, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, st));
        return PFMSCAN_OK;
    }
    // a source inside a range registered as a file mapping is read from the file
    int fd = -1;
    int64_t foff = 0;
    const unsigned char *hs = static_cast<const unsigned char *>(h_src);
    if (!std::getenv("PFMSCAN_UPLOAD_NO_PREAD"))
        for (const pfmscan_ctx::FileRange &fr : ctx->file_ranges)
            if (fr.fd >= 0 && hs >= fr.base && hs + bytes <= fr.base + fr.length) {
                fd = fr.fd;
                foff = fr.offset + (int64_t)(hs - fr.base);
            }
    for (size_t off = 0; off < bytes; off += up->PIECE) {
        const size_t nb = std::min(up->PIECE, bytes - off);
        const int b = up->next;
        up->next = (b + 1) % Uploader::NBUF;
        if (up->busy[b]) HIP_TRY(ctx, hipEventSynchronize(up->done[b]));       // its previous piece has left the buffer
        up->copy(up->pinned[b], hs + off, nb, fd, foff + (int64_t)off);
        HIP_TRY(ctx, hipMemcpyAsync(static_cast<unsigned char *>(d_dst) + off, up->pinned[b], nb, hipMemcpyHostToDevice, st));
        HIP_TRY(ctx, hipEventRecord(up->done[b], st));
        up->busy[b] = true;
    }
    return PFMSCAN_OK;
}

}  // namespace pfmscan

extern "C" int pfmscan_set_upload_mode(pfmscan_ctx *ctx, int mode)
{
    if (!ctx || (mode != PFMSCAN_UPLOAD_RUNTIME && mode != PFMSCAN_UPLOAD_STAGED)) return pfmscan::fail(ctx, PFMSCAN_E_BADARG, "bad upload mode");
    ctx->upload_mode = mode;
    return PFMSCAN_OK;
}

extern "C" int pfmscan_upload_source_file_checked(pfmscan_ctx *ctx, const void *base, size_t length, const char *path,
                                                  int64_t file_offset, int64_t st_dev, int64_t st_ino, int64_t st_size);

extern "C" int pfmscan_upload_source_file(pfmscan_ctx *ctx, const void *base, size_t length, const char *path, int64_t file_offset)
{
    return pfmscan_upload_source_file_checked(ctx, base, length, path, file_offset, -1, -1, -1);
}

extern "C" int pfmscan_upload_source_file_checked(pfmscan_ctx *ctx, const void *base, size_t length, const char *path,
                                                  int64_t file_offset, int64_t st_dev, int64_t st_ino, int64_t st_size)
{
    if (!ctx || !base) return pfmscan::fail(ctx, PFMSCAN_E_BADARG, "pfmscan_upload_source_file: NULL argument");
    const unsigned char *b = static_cast<const unsigned char *>(base);
    for (pfmscan_ctx::FileRange &fr : ctx->file_ranges)       // a range with this base is replaced or forgotten
        if (fr.fd >= 0 && fr.base == b) {
            (void)close(fr.fd);
            fr = pfmscan_ctx::FileRange();
        }
    if (length == 0) return PFMSCAN_OK;
    if (!path || file_offset < 0) return pfmscan::fail(ctx, PFMSCAN_E_BADARG, "pfmscan_upload_source_file: no path / negative offset");
    const int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) return pfmscan::fail(ctx, PFMSCAN_E_BADARG, std::string("pfmscan_upload_source_file: cannot open ") + path);
    if (st_ino >= 0) {
        // the caller recorded which file it MAPPED: a path that now names another file (a store re-packed by atomic rename
        // since) must not be pread in the mapping's place -- the mapping itself stays a valid source
        struct stat sb;
        if (fstat(fd, &sb) != 0 || (int64_t)sb.st_dev != st_dev || (int64_t)sb.st_ino != st_ino || (int64_t)sb.st_size != st_size) {
            (void)close(fd);
            return pfmscan::fail(ctx, PFMSCAN_E_BADARG, std::string("pfmscan_upload_source_file: ") + path + " is not the file that was mapped any more");
        }
    }
    pfmscan_ctx::FileRange *slot = nullptr;
    for (pfmscan_ctx::FileRange &fr : ctx->file_ranges)
        if (fr.fd < 0 && !slot) slot = &fr;
    if (!slot) {                                              // full: the oldest goes
        slot = &ctx->file_ranges[ctx->file_range_next];
        ctx->file_range_next = (ctx->file_range_next + 1) % pfmscan_ctx::N_FILE_RANGES;
        (void)close(slot->fd);
    }
    slot->base = b;
    slot->length = length;
    slot->fd = fd;
    slot->offset = file_offset;
    return PFMSCAN_OK;
}

namespace pfmscan {

void upload_release(pfmscan_ctx *ctx)
{
    delete ctx->up;
    ctx->up = nullptr;
    for (pfmscan_ctx::FileRange &fr : ctx->file_ranges)
        if (fr.fd >= 0) {
            (void)close(fr.fd);
            fr = pfmscan_ctx::FileRange();
        }
}

}  // namespace pfmscan
