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
    int64_t spaces = 0;
    for (int64_t i = a; i < b; ++i) spaces += buf[i] == ' ';
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

extern "C" {

int pfmscan_fasta_index(const uint8_t *buf, int64_t n, int64_t capacity, int64_t *hdr_off, int64_t *hdr_len,
                        int64_t *seq_off, int64_t *seq_end, int64_t *n_letters, int64_t *n_records)
{
    if ((!buf && n > 0) || n < 0 || capacity < 0 || !n_records) return fail(nullptr, PFMSCAN_E_BADARG, "fasta_index: bad argument");
    const bool fill = capacity > 0;
    if (fill && (!hdr_off || !hdr_len || !seq_off || !seq_end || !n_letters))
        return fail(nullptr, PFMSCAN_E_BADARG, "fasta_index: NULL output array");
    int64_t rec = -1, letters = 0, pos = 0;
    while (pos < n) {
        const uint8_t *nl = static_cast<const uint8_t *>(std::memchr(buf + pos, '\n', (size_t)(n - pos)));
        const int64_t e = nl ? nl - buf : n, next = nl ? e + 1 : n;
        if (buf[pos] == '>') {
            if (rec >= 0 && rec < capacity) {
                seq_end[rec] = pos;
                n_letters[rec] = letters;
            }
            ++rec;
            letters = 0;
            if (rec < capacity) {
                int64_t he = e;
                while (he > pos + 1 && (buf[he - 1] == '\r' || buf[he - 1] == '\n')) --he;    // rstrip("\r\n")
                hdr_off[rec] = pos + 1;
                hdr_len[rec] = he - (pos + 1);
                seq_off[rec] = next;
            }
        } else if (rec >= 0 && fill) {
            letters += count_letters(buf, pos, e);
        }
        pos = next;
    }
    if (rec >= 0 && rec < capacity) {
        seq_end[rec] = n;
        n_letters[rec] = letters;
    }
    *n_records = rec + 1;
    if (rec + 1 > capacity) return fill ? fail(nullptr, PFMSCAN_E_CAPACITY, "fasta_index: more records than capacity") : PFMSCAN_E_CAPACITY;
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
        for (int64_t i = a; i < b; ++i) {
            uint8_t *out = codes + offsets[i];
            int64_t k = 0, pos = seq_off[lo + i];
            const int64_t stop = seq_end[lo + i], want = n_letters[lo + i];
            while (pos < stop) {
                const uint8_t *nl = static_cast<const uint8_t *>(std::memchr(buf + pos, '\n', (size_t)(stop - pos)));
                int64_t e = nl ? nl - buf : stop;
                const int64_t next = nl ? e + 1 : stop;
                int64_t s = pos;
                strip(buf, s, e);
                if (k + (e - s) > want) {                              // index and bytes disagree: never write past the record
                    int64_t room = want - k;
                    for (; s < e && room > 0; ++s)
                        if (buf[s] != ' ') out[k++] = lut256[buf[s]], --room;
                    for (; s < e; ++s)
                        if (buf[s] != ' ') bad[(size_t)t] = 1;
                } else {
                    for (; s < e; ++s)
                        if (buf[s] != ' ') out[k++] = lut256[buf[s]];
                }
                pos = next;
            }
            if (k != want) bad[(size_t)t] = 1;
            for (; k < want; ++k) out[k] = (uint8_t)separator;
            out[want] = (uint8_t)separator;
        }
    });
    for (int v : bad)
        if (v) return fail(nullptr, PFMSCAN_E_BADARG, "fasta_encode: the index does not describe these bytes (file changed since it was indexed?)");
    return PFMSCAN_OK;
}

int pfmscan_tsv_format(const pfmscan_tsv_column *cols, int n_cols, int64_t n_rows, int64_t first_match_id, char *out,
                       int64_t capacity, int64_t *n_bytes, int n_threads)
{
    if (!cols || n_cols <= 0 || n_rows < 0 || !n_bytes || capacity < 0 || (!out && capacity > 0))
        return fail(nullptr, PFMSCAN_E_BADARG, "tsv_format: bad argument");
    int64_t fixed = 0;                       // bytes of a row that do not depend on the row
    for (int c = 0; c < n_cols; ++c) {
        const pfmscan_tsv_column &col = cols[c];
        if (col.kind < PFMSCAN_TSV_CONST || col.kind > PFMSCAN_TSV_SPAN || (!col.data && !(col.kind == PFMSCAN_TSV_CONST && col.width == 0)) ||
            col.width < 0 ||
            ((col.kind == PFMSCAN_TSV_INDEXED || col.kind == PFMSCAN_TSV_WINDOW || col.kind == PFMSCAN_TSV_SPAN) && (!col.aux || !col.blob)))
            return fail(nullptr, PFMSCAN_E_BADARG, "tsv_format: bad column descriptor");
        switch (col.kind) {
        case PFMSCAN_TSV_CONST: case PFMSCAN_TSV_FIXED: case PFMSCAN_TSV_WINDOW: fixed += col.width; break;
        case PFMSCAN_TSV_I64: fixed += 21; break;
        case PFMSCAN_TSV_F32: case PFMSCAN_TSV_F64: fixed += 26; break;
        default: break;
        }
    }
    fixed += n_cols + (first_match_id >= 0 ? 21 : 0);
    const int threads = pick_threads(n_threads, (n_rows + 4095) / 4096);
    std::vector<std::vector<char>> part((size_t)threads);
    parallel_ranges(n_rows, threads, [&](int t, int64_t a, int64_t b) {
        std::vector<char> &buf = part[(size_t)t];
        buf.resize((size_t)std::max<int64_t>(1 << 16, (b - a) * (fixed + 16)));
        size_t used = 0;
        for (int64_t r = a; r < b; ++r) {
            size_t need = (size_t)fixed;
            for (int c = 0; c < n_cols; ++c)
                if (cols[c].kind == PFMSCAN_TSV_INDEXED) {
                    const int64_t *off = static_cast<const int64_t *>(cols[c].aux);
                    const int64_t v = static_cast<const int64_t *>(cols[c].data)[r];
                    need += (size_t)(off[v + 1] - off[v]);
                } else if (cols[c].kind == PFMSCAN_TSV_SPAN) {
                    const int64_t v = static_cast<const int64_t *>(cols[c].data)[r];
                    need += 2 * (size_t)static_cast<const int64_t *>(cols[c].aux)[2 * v + 1] + 2;      // every byte a doubled quote
                }
            if (used + need > buf.size()) buf.resize(std::max(buf.size() * 2, used + need));
            char *p = buf.data() + used;
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
            used = (size_t)(p - buf.data());
        }
        buf.resize(used);
    });
    int64_t total = 0;
    for (auto &b : part) total += (int64_t)b.size();
    *n_bytes = total;
    if (total > capacity) return fail(nullptr, PFMSCAN_E_CAPACITY, "tsv_format: output buffer too small");
    std::vector<int64_t> at((size_t)threads, 0);
    for (int t = 1; t < threads; ++t) at[(size_t)t] = at[(size_t)t - 1] + (int64_t)part[(size_t)t - 1].size();
    parallel_ranges(threads, threads, [&](int, int64_t a, int64_t b) {
        for (int64_t t = a; t < b; ++t)
            if (!part[(size_t)t].empty()) std::memcpy(out + at[(size_t)t], part[(size_t)t].data(), part[(size_t)t].size());
    });
    return PFMSCAN_OK;
}

}  // extern "C"
