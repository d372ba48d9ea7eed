// pfmscan_ingest.hip -- the host work on either side of the scan, in native code (no device needed):
//   * FASTA bytes -> record index -> packed code stream        (replaces SeqIO.parse + preprocess_seq + the
//     per-record encode of the Python host: rnascan.py:170-174, :177-204)
//   * hit columns -> the bytes DataFrame.to_csv(sep='\t', index=False) writes (rnascan.py:555-567)
// At C3 size (100k records x 3 kb, 3x10^5 hits) the Python forms of these two cost seconds against a 0.1-2 ms
// kernel; these run at memory speed on the host cores the process may use.
#include <algorithm>
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "pfmscan_ctx.hpp"

using pfmscan::fail;

namespace {

inline bool is_space(uint8_t c) { return c == ' ' || (c >= 9 && c <= 13); }   // bytes.strip(): " \t\n\r\x0b\x0c"

// [a, b) of one line without its '\n', both ends stripped
inline void strip(const uint8_t *buf, int64_t &a, int64_t &b)
{
    while (a < b && is_space(buf[a])) ++a;
    while (b > a && is_space(buf[b - 1])) --b;
}

inline int64_t count_letters(const uint8_t *buf, int64_t a, int64_t b)
{
    strip(buf, a, b);
    int64_t spaces = 0;                                             // embedded blanks are rare: let memchr look for them
    const uint8_t *q = buf + a, *const end = buf + b;
    while (q < end && (q = static_cast<const uint8_t *>(std::memchr(q, ' ', (size_t)(end - q)))) != nullptr) {
        ++spaces;
        ++q;
    }
    return (b - a) - spaces;
}

int pick_threads(int n_threads, int64_t work_items)
{
    int t = n_threads > 0 ? n_threads : (int)std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 16u);
    return (int)std::max<int64_t>(1, std::min<int64_t>(t, work_items));
}

template <class F>
void parallel_ranges(int64_t n, int threads, F &&fn)
{
    if (threads <= 1) {
        fn(0, (int64_t)0, n);
        return;
    }
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; ++t) pool.emplace_back([&, t] { fn(t, n * t / threads, n * (t + 1) / threads); });
    for (auto &th : pool) th.join();
}

// ---- number formatting: the strings numpy's astype(str) / repr(float) give ----------------------------------
// shortest digits that round-trip in the value's own precision; positional for 1e-4 <= |x| < 1e16, else d.ddde+XX
template <class T>
char *put_float(char *p, T x)
{
    if (std::isnan(x)) return p;                                   // na_rep=''
    if (std::isinf(x)) {
        if (x < 0) *p++ = '-';
        std::memcpy(p, "inf", 3);
        return p + 3;
    }
    char tmp[40];
    auto r = std::to_chars(tmp, tmp + sizeof tmp, x, std::chars_format::scientific);
    const char *s = tmp, *end = r.ptr;
    if (*s == '-') *p++ = *s++;
    char digits[24];
    int nd = 0;
    while (s < end && *s != 'e') {
        if (*s != '.') digits[nd++] = *s;
        ++s;
    }
    ++s;                                                            // 'e'
    const bool eneg = *s == '-';
    ++s;
    int E = 0;
    while (s < end) E = E * 10 + (*s++ - '0');
    if (eneg) E = -E;
    const double ax = std::fabs((double)x);                         // numpy decides on the value, not on the printed digits:
    if ((ax >= 1e-4 && ax < 1e16) || ax == 0.0) {                                  // float32(1e-4) lies below 1e-4 and prints as 1e-04
        if (E >= 0) {
            for (int i = 0; i <= E; ++i) *p++ = i < nd ? digits[i] : '0';
            *p++ = '.';
            if (nd > E + 1)
                for (int i = E + 1; i < nd; ++i) *p++ = digits[i];
            else
                *p++ = '0';
        } else {
            *p++ = '0';
            *p++ = '.';
            for (int i = 0; i < -E - 1; ++i) *p++ = '0';
            for (int i = 0; i < nd; ++i) *p++ = digits[i];
        }
    } else {
        *p++ = digits[0];
        if (nd > 1) {
            *p++ = '.';
            for (int i = 1; i < nd; ++i) *p++ = digits[i];
        }
        *p++ = 'e';
        *p++ = E < 0 ? '-' : '+';
        const int a = E < 0 ? -E : E;
        if (a < 10) *p++ = '0';
        p = std::to_chars(p, p + 8, a).ptr;
    }
    return p;
}

inline char *put_int(char *p, int64_t v) { return std::to_chars(p, p + 24, v).ptr; }

}  // namespace

namespace {

// one record's sequence lines -> out[0 .. want] (want letters + the separator); nonzero when the bytes hold a different
// number of letters than the index says (never writes past out[want])
int encode_record(const uint8_t *__restrict buf, int64_t pos, const int64_t stop, const int64_t want, const uint8_t *__restrict lut,
                  const uint8_t separator, uint8_t *__restrict out)
{
    int64_t k = 0;
    int wrong = 0;
    while (pos < stop) {
        const uint8_t *nl = static_cast<const uint8_t *>(std::memchr(buf + pos, '\n', (size_t)(stop - pos)));
        int64_t e = nl ? nl - buf : stop;
        const int64_t next = nl ? e + 1 : stop;
        int64_t s = pos;
        strip(buf, s, e);
        if (k + (e - s) > want) {                                  // index and bytes disagree
            for (; s < e; ++s)
                if (buf[s] != ' ') {
                    if (k < want)
                        out[k++] = lut[buf[s]];
                    else
                        wrong = 1;
                }
        } else if (!std::memchr(buf + s, ' ', (size_t)(e - s))) {  // the usual line: no blanks
            const uint8_t *__restrict src = buf + s;
            uint8_t *__restrict dst = out + k;
            const int64_t n = e - s;
            for (int64_t i = 0; i < n; ++i) dst[i] = lut[src[i]];
            k += n;
        } else {
            for (; s < e; ++s)
                if (buf[s] != ' ') out[k++] = lut[buf[s]];
        }
        pos = next;
    }
    if (k != want) wrong = 1;
    for (; k < want; ++k) out[k] = separator;
    out[want] = separator;
    return wrong;
}

struct FastaRec {
    int64_t hdr_off, hdr_len, seq_off, seq_end, letters;
};

// records whose header line STARTS in [from, to) (both line starts); `lead` = letters of the lines before the first
// such header (they belong to a record that began in an earlier piece)
void index_piece(const uint8_t *buf, int64_t n, int64_t from, int64_t to, bool count_only, std::vector<FastaRec> &out,
                 int64_t &lead, int64_t &first_hdr, int64_t &count)
{
    int64_t pos = from, letters = 0;
    bool open = false;
    lead = 0;
    first_hdr = -1;
    count = 0;
    while (pos < to) {
        const uint8_t *nl = static_cast<const uint8_t *>(std::memchr(buf + pos, '\n', (size_t)(n - pos)));
        const int64_t e = nl ? nl - buf : n, next = nl ? e + 1 : n;
        if (buf[pos] == '>') {
            if (first_hdr < 0) first_hdr = pos;
            ++count;
            if (!count_only) {
                if (open) {
                    out.back().seq_end = pos;
                    out.back().letters = letters;
                } else {
                    lead = letters;
                }
                int64_t he = e;
                while (he > pos + 1 && (buf[he - 1] == '\r' || buf[he - 1] == '\n')) --he;    // rstrip("\r\n")
                out.push_back({pos + 1, he - (pos + 1), next, next, 0});
            }
            open = true;
            letters = 0;
        } else if (!count_only) {
            letters += count_letters(buf, pos, e);
        }
        pos = next;
    }
    if (!count_only) {
        if (open) {
            out.back().seq_end = to;
            out.back().letters = letters;
        } else {
            lead = letters;
        }
    }
}

}  // namespace

extern "C" {

int pfmscan_fasta_index(const uint8_t *buf, int64_t n, int64_t capacity, int64_t *hdr_off, int64_t *hdr_len,
                        int64_t *seq_off, int64_t *seq_end, int64_t *n_letters, int64_t *n_records, int n_threads)
{
    if ((!buf && n > 0) || n < 0 || capacity < 0 || !n_records) return fail(nullptr, PFMSCAN_E_BADARG, "fasta_index: bad argument");
    const bool fill = capacity > 0;
    if (fill && (!hdr_off || !hdr_len || !seq_off || !seq_end || !n_letters))
        return fail(nullptr, PFMSCAN_E_BADARG, "fasta_index: NULL output array");
    // pieces of the buffer that start at a line start, one per thread
    const int threads = n_threads > 0 ? (int)std::max<int64_t>(1, std::min<int64_t>(n_threads, n)) : pick_threads(0, n >> 20);
    std::vector<int64_t> cut((size_t)threads + 1, n);
    cut[0] = 0;
    for (int t = 1; t < threads; ++t) {
        const int64_t guess = std::max(cut[(size_t)t - 1], n * t / threads);
        const uint8_t *nl = guess < n ? static_cast<const uint8_t *>(std::memchr(buf + guess, '\n', (size_t)(n - guess))) : nullptr;
        cut[(size_t)t] = nl ? (nl - buf) + 1 : n;
    }
    std::vector<std::vector<FastaRec>> recs((size_t)threads);
    std::vector<int64_t> lead((size_t)threads, 0), first((size_t)threads, -1), count((size_t)threads, 0);
    parallel_ranges(threads, threads, [&](int, int64_t a, int64_t b) {
        for (int64_t t = a; t < b; ++t)
            index_piece(buf, n, cut[(size_t)t], cut[(size_t)t + 1], !fill, recs[(size_t)t], lead[(size_t)t], first[(size_t)t],
                        count[(size_t)t]);
    });
    int64_t total = 0;
    for (int64_t c : count) total += c;
    *n_records = total;
    if (total > capacity) return fill ? fail(nullptr, PFMSCAN_E_CAPACITY, "fasta_index: more records than capacity") : PFMSCAN_E_CAPACITY;
    // stitch: the lines a piece holds before its first header continue the last record of the pieces before it
    int64_t k = 0, last = -1;
    for (int t = 0; t < threads; ++t) {
        if (last >= 0) {
            n_letters[last] += lead[(size_t)t];
            seq_end[last] = first[(size_t)t] >= 0 ? first[(size_t)t] : cut[(size_t)t + 1];
        }
        for (const FastaRec &r : recs[(size_t)t]) {
            hdr_off[k] = r.hdr_off;
            hdr_len[k] = r.hdr_len;
            seq_off[k] = r.seq_off;
            seq_end[k] = r.seq_end;
            n_letters[k] = r.letters;
            last = k++;
        }
    }
    return PFMSCAN_OK;
}

int pfmscan_fasta_ids(const uint8_t *buf, const int64_t *hdr_off, const int64_t *hdr_len, int64_t n_records, int64_t *id_off,
                      int64_t *id_len, int *all_ascii)
{
    if (n_records < 0 || (n_records > 0 && (!buf || !hdr_off || !hdr_len || !id_off || !id_len)) || !all_ascii)
        return fail(nullptr, PFMSCAN_E_BADARG, "fasta_ids: bad argument");
    // str.split(None, 1)[0] on ASCII text: whitespace is \t \n \v \f \r, \x1c..\x1f and the blank
    auto ws = [](uint8_t c) { return c == ' ' || (c >= 9 && c <= 13) || (c >= 28 && c <= 31); };
    int ascii = 1;
    for (int64_t r = 0; r < n_records; ++r) {
        const uint8_t *h = buf + hdr_off[r];
        const int64_t n = hdr_len[r];
        int64_t a = 0;
        while (a < n && ws(h[a])) ++a;
        int64_t b = a;
        while (b < n && !ws(h[b])) ++b;
        id_off[r] = hdr_off[r] + a;
        id_len[r] = b - a;
        uint8_t any = 0;
        for (int64_t i = 0; i < n; ++i) any |= h[i];
        if (any & 0x80) ascii = 0;
    }
    *all_ascii = ascii;
    return PFMSCAN_OK;
}

int pfmscan_gather_spans(const uint8_t *buf, const int64_t *spans, int64_t n_spans, int separator, uint8_t *out, int64_t capacity,
                         int64_t *n_bytes)
{
    if (n_spans < 0 || (n_spans > 0 && (!buf || !spans)) || !n_bytes || capacity < 0 || (!out && capacity > 0))
        return fail(nullptr, PFMSCAN_E_BADARG, "gather_spans: bad argument");
    int64_t total = 0;
    for (int64_t i = 0; i < n_spans; ++i) total += spans[2 * i + 1] + 1;
    *n_bytes = total;
    if (total > capacity) return fail(nullptr, PFMSCAN_E_CAPACITY, "gather_spans: output buffer too small");
    uint8_t *p = out;
    for (int64_t i = 0; i < n_spans; ++i) {
        std::memcpy(p, buf + spans[2 * i], (size_t)spans[2 * i + 1]);
        p += spans[2 * i + 1];
        *p++ = (uint8_t)separator;
    }
    return PFMSCAN_OK;
}

int pfmscan_fasta_encode(const uint8_t *buf, const int64_t *seq_off, const int64_t *seq_end, const int64_t *n_letters,
                         int64_t lo, int64_t hi, const uint8_t *lut256, int separator, uint8_t *codes, int64_t *offsets,
                         int n_threads)
{
    if (!buf || !seq_off || !seq_end || !n_letters || !lut256 || !codes || !offsets || lo < 0 || hi < lo)
        return fail(nullptr, PFMSCAN_E_BADARG, "fasta_encode: bad argument");
    const int64_t nrec = hi - lo;
    int64_t acc = 0;
    for (int64_t i = 0; i < nrec; ++i) {
        offsets[i] = acc;
        acc += n_letters[lo + i] + 1;
    }
    std::vector<int> bad((size_t)pick_threads(n_threads, nrec), 0);
    parallel_ranges(nrec, (int)bad.size(), [&](int t, int64_t a, int64_t b) {
        int wrong = 0;
        for (int64_t i = a; i < b; ++i)
            wrong |= encode_record(buf, seq_off[lo + i], seq_end[lo + i], n_letters[lo + i], lut256, (uint8_t)separator, codes + offsets[i]);
        bad[(size_t)t] = wrong;
    });
    for (int v : bad)
        if (v) return fail(nullptr, PFMSCAN_E_BADARG, "fasta_encode: the index does not describe these bytes (file changed since it was indexed?)");
    return PFMSCAN_OK;
}

int pfmscan_tsv_format(const pfmscan_tsv_column *cols, int n_cols, int64_t n_rows, int64_t first_match_id, char *out,
                       int64_t capacity, int64_t *need, int64_t *pieces, int *n_pieces, int n_threads)
{
    if (!cols || n_cols <= 0 || n_rows < 0 || !need || !pieces || !n_pieces || capacity < 0 || (!out && capacity > 0))
        return fail(nullptr, PFMSCAN_E_BADARG, "tsv_format: bad argument");
    const int threads = pick_threads(std::min(n_threads > 0 ? n_threads : PFMSCAN_TSV_MAX_PIECES, PFMSCAN_TSV_MAX_PIECES),
                                     (n_rows + 16383) / 16384);                    // a thread is worth starting for ~16k rows
    // the most bytes a row can take: every thread then owns a slice of `out` it cannot overrun
    int64_t bound = n_cols + (first_match_id >= 0 ? 21 : 0);
    std::vector<int64_t> longest((size_t)n_cols, 0);
    for (int c = 0; c < n_cols; ++c) {
        const pfmscan_tsv_column &col = cols[c];
        if (col.kind < PFMSCAN_TSV_CONST || col.kind > PFMSCAN_TSV_SPAN || (!col.data && !(col.kind == PFMSCAN_TSV_CONST && col.width == 0)) ||
            col.width < 0 ||
            ((col.kind == PFMSCAN_TSV_INDEXED || col.kind == PFMSCAN_TSV_WINDOW || col.kind == PFMSCAN_TSV_SPAN) && (!col.aux || !col.blob)))
            return fail(nullptr, PFMSCAN_E_BADARG, "tsv_format: bad column descriptor");
        switch (col.kind) {
        case PFMSCAN_TSV_CONST: case PFMSCAN_TSV_FIXED: case PFMSCAN_TSV_WINDOW: bound += col.width; break;
        case PFMSCAN_TSV_I64: bound += 21; break;
        case PFMSCAN_TSV_F32: case PFMSCAN_TSV_F64: bound += 26; break;
        default: {                                   // INDEXED / SPAN: the longest value some row uses
            const int64_t *index = static_cast<const int64_t *>(col.data);
            const int64_t *aux = static_cast<const int64_t *>(col.aux);
            std::vector<int64_t> mx((size_t)threads, 0);
            parallel_ranges(n_rows, threads, [&](int t, int64_t a, int64_t b) {
                int64_t m = 0;
                if (col.kind == PFMSCAN_TSV_INDEXED)
                    for (int64_t r = a; r < b; ++r) m = std::max(m, aux[index[r] + 1] - aux[index[r]]);
                else
                    for (int64_t r = a; r < b; ++r) m = std::max(m, 2 * aux[2 * index[r] + 1] + 2);    // every byte a doubled quote
                mx[(size_t)t] = m;
            });
            for (int64_t m : mx) longest[(size_t)c] = std::max(longest[(size_t)c], m);
            bound += longest[(size_t)c];
        }
        }
    }
    *need = n_rows * bound;
    *n_pieces = 0;
    if (*need > capacity) return fail(nullptr, PFMSCAN_E_CAPACITY, "tsv_format: output buffer too small");
    std::vector<int64_t> used((size_t)threads, 0), from((size_t)threads, 0);
    parallel_ranges(n_rows, threads, [&](int t, int64_t a, int64_t b) {
        char *const base = out + a * bound;
        char *p = base;
        for (int64_t r = a; r < b; ++r) {
            for (int c = 0; c < n_cols; ++c) {
                const pfmscan_tsv_column &col = cols[c];
                if (c) *p++ = '\t';
                switch (col.kind) {
                case PFMSCAN_TSV_CONST:
                    std::memcpy(p, col.data, (size_t)col.width);
                    p += col.width;
                    break;
                case PFMSCAN_TSV_I64: p = put_int(p, static_cast<const int64_t *>(col.data)[r]); break;
                case PFMSCAN_TSV_F32: p = put_float(p, static_cast<const float *>(col.data)[r]); break;
                case PFMSCAN_TSV_F64: p = put_float(p, static_cast<const double *>(col.data)[r]); break;
                case PFMSCAN_TSV_INDEXED: {
                    const int64_t *off = static_cast<const int64_t *>(col.aux);
                    const int64_t v = static_cast<const int64_t *>(col.data)[r];
                    std::memcpy(p, static_cast<const char *>(col.blob) + off[v], (size_t)(off[v + 1] - off[v]));
                    p += off[v + 1] - off[v];
                    break;
                }
                case PFMSCAN_TSV_FIXED: {
                    const char *s = static_cast<const char *>(col.data) + r * col.width;
                    int64_t w = col.width;
                    while (w > 0 && s[w - 1] == 0) --w;              // numpy 'S' items are NUL padded
                    std::memcpy(p, s, (size_t)w);
                    p += w;
                    break;
                }
                case PFMSCAN_TSV_SPAN: {                             // bytes of the caller's buffer, csv.QUOTE_MINIMAL applied here
                    const int64_t v = static_cast<const int64_t *>(col.data)[r];
                    const int64_t *span = static_cast<const int64_t *>(col.aux) + 2 * v;
                    const char *src = static_cast<const char *>(col.blob) + span[0];
                    const int64_t len = span[1];
                    bool quote = false;
                    for (int64_t i = 0; i < len; ++i) quote |= src[i] == '\t' || src[i] == '"' || src[i] == '\n' || src[i] == '\r';
                    if (!quote) {
                        std::memcpy(p, src, (size_t)len);
                        p += len;
                    } else {
                        *p++ = '"';
                        for (int64_t i = 0; i < len; ++i) {
                            if (src[i] == '"') *p++ = '"';
                            *p++ = src[i];
                        }
                        *p++ = '"';
                    }
                    break;
                }
                default: {                                           // PFMSCAN_TSV_WINDOW
                    const uint8_t *codes = static_cast<const uint8_t *>(col.aux) + static_cast<const int64_t *>(col.data)[r];
                    const char *letters = static_cast<const char *>(col.blob);
                    for (int64_t j = 0; j < col.width; ++j) *p++ = letters[codes[j] & 7];
                    break;
                }
                }
            }
            if (first_match_id >= 0) {
                *p++ = '\t';
                p = put_int(p, first_match_id + r);
            }
            *p++ = '\n';
        }
        from[(size_t)t] = a * bound;
        used[(size_t)t] = p - base;
    });
    for (int t = 0; t < threads; ++t) {
        pieces[2 * t] = from[(size_t)t];
        pieces[2 * t + 1] = used[(size_t)t];
    }
    *n_pieces = threads;
    return PFMSCAN_OK;
}

}  // extern "C"
