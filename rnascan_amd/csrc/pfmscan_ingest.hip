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
#include <cstdio>
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

int pfmscan_count_bytes(const uint8_t *buf, int64_t n, int64_t *counts, int n_threads)
{
    if ((!buf && n > 0) || n < 0 || !counts) return fail(nullptr, PFMSCAN_E_BADARG, "count_bytes: bad argument");
    const int threads = pick_threads(n_threads, n >> 20);
    std::vector<int64_t> part((size_t)threads * 256, 0);
    parallel_ranges(n, threads, [&](int t, int64_t a, int64_t b) {
        // four interleaved histograms: neighbouring equal bytes (homopolymers, separators) would otherwise serialise on one counter
        int64_t h[4][256] = {};
        int64_t i = a;
        for (; i + 4 <= b; i += 4) {
            ++h[0][buf[i]];
            ++h[1][buf[i + 1]];
            ++h[2][buf[i + 2]];
            ++h[3][buf[i + 3]];
        }
        for (; i < b; ++i) ++h[0][buf[i]];
        for (int v = 0; v < 256; ++v) part[(size_t)t * 256 + v] = h[0][v] + h[1][v] + h[2][v] + h[3][v];
    });
    for (int v = 0; v < 256; ++v) {
        int64_t sum = 0;
        for (int t = 0; t < threads; ++t) sum += part[(size_t)t * 256 + v];
        counts[v] = sum;
    }
    return PFMSCAN_OK;
}

int pfmscan_fasta_lone_cr(const uint8_t *buf, int64_t n, int *found, int n_threads)
{
    if ((!buf && n > 0) || n < 0 || !found) return fail(nullptr, PFMSCAN_E_BADARG, "fasta_lone_cr: bad argument");
    const int threads = pick_threads(n_threads, n >> 22);
    std::vector<int> hit((size_t)threads, 0);
    parallel_ranges(n, threads, [&](int t, int64_t a, int64_t b) {
        const uint8_t *q = buf + a, *const end = buf + b;
        while (q < end && (q = static_cast<const uint8_t *>(std::memchr(q, '\r', (size_t)(end - q)))) != nullptr) {
            if (q + 1 >= buf + n || q[1] != '\n') {          // (the byte after a piece's last one belongs to the next piece: still in buf)
                hit[(size_t)t] = 1;
                return;
            }
            ++q;
        }
    });
    *found = 0;
    for (int h : hit) *found |= h;
    return PFMSCAN_OK;
}

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

}  // extern "C"

namespace {

// pandas' default float converter of read_table's C engine ("high" precision: precise_xstrtod, pandas >= 1.2) restated:
// at most 17 digits INCLUDING leading zeros are accumulated in a double, the rest only move the exponent, then ONE
// multiplication or division by an exactly represented power of ten.  Not correctly rounded (about half of all 17-digit
// reprs come out one ulp off), which is exactly why it is restated: the reference reads its profiles with
// pd.read_table (rnascan.py:296) and computes with these values.  Returns false on a token it does not take
// (the caller then leaves the file to pandas).
const double POW10[] = {
    1e0,   1e1,   1e2,   1e3,   1e4,   1e5,   1e6,   1e7,   1e8,   1e9,   1e10,  1e11,  1e12,  1e13,  1e14,  1e15,  1e16,  1e17,  1e18,  1e19,
    1e20,  1e21,  1e22,  1e23,  1e24,  1e25,  1e26,  1e27,  1e28,  1e29,  1e30,  1e31,  1e32,  1e33,  1e34,  1e35,  1e36,  1e37,  1e38,  1e39,
    1e40,  1e41,  1e42,  1e43,  1e44,  1e45,  1e46,  1e47,  1e48,  1e49,  1e50,  1e51,  1e52,  1e53,  1e54,  1e55,  1e56,  1e57,  1e58,  1e59,
    1e60,  1e61,  1e62,  1e63,  1e64,  1e65,  1e66,  1e67,  1e68,  1e69,  1e70,  1e71,  1e72,  1e73,  1e74,  1e75,  1e76,  1e77,  1e78,  1e79,
    1e80,  1e81,  1e82,  1e83,  1e84,  1e85,  1e86,  1e87,  1e88,  1e89,  1e90,  1e91,  1e92,  1e93,  1e94,  1e95,  1e96,  1e97,  1e98,  1e99,
    1e100, 1e101, 1e102, 1e103, 1e104, 1e105, 1e106, 1e107, 1e108, 1e109, 1e110, 1e111, 1e112, 1e113, 1e114, 1e115, 1e116, 1e117, 1e118, 1e119,
    1e120, 1e121, 1e122, 1e123, 1e124, 1e125, 1e126, 1e127, 1e128, 1e129, 1e130, 1e131, 1e132, 1e133, 1e134, 1e135, 1e136, 1e137, 1e138, 1e139,
    1e140, 1e141, 1e142, 1e143, 1e144, 1e145, 1e146, 1e147, 1e148, 1e149, 1e150, 1e151, 1e152, 1e153, 1e154, 1e155, 1e156, 1e157, 1e158, 1e159,
    1e160, 1e161, 1e162, 1e163, 1e164, 1e165, 1e166, 1e167, 1e168, 1e169, 1e170, 1e171, 1e172, 1e173, 1e174, 1e175, 1e176, 1e177, 1e178, 1e179,
    1e180, 1e181, 1e182, 1e183, 1e184, 1e185, 1e186, 1e187, 1e188, 1e189, 1e190, 1e191, 1e192, 1e193, 1e194, 1e195, 1e196, 1e197, 1e198, 1e199,
    1e200, 1e201, 1e202, 1e203, 1e204, 1e205, 1e206, 1e207, 1e208, 1e209, 1e210, 1e211, 1e212, 1e213, 1e214, 1e215, 1e216, 1e217, 1e218, 1e219,
    1e220, 1e221, 1e222, 1e223, 1e224, 1e225, 1e226, 1e227, 1e228, 1e229, 1e230, 1e231, 1e232, 1e233, 1e234, 1e235, 1e236, 1e237, 1e238, 1e239,
    1e240, 1e241, 1e242, 1e243, 1e244, 1e245, 1e246, 1e247, 1e248, 1e249, 1e250, 1e251, 1e252, 1e253, 1e254, 1e255, 1e256, 1e257, 1e258, 1e259,
    1e260, 1e261, 1e262, 1e263, 1e264, 1e265, 1e266, 1e267, 1e268, 1e269, 1e270, 1e271, 1e272, 1e273, 1e274, 1e275, 1e276, 1e277, 1e278, 1e279,
    1e280, 1e281, 1e282, 1e283, 1e284, 1e285, 1e286, 1e287, 1e288, 1e289, 1e290, 1e291, 1e292, 1e293, 1e294, 1e295, 1e296, 1e297, 1e298, 1e299,
    1e300, 1e301, 1e302, 1e303, 1e304, 1e305, 1e306, 1e307, 1e308};

inline bool is_digit(char c) { return c >= '0' && c <= '9'; }

bool pandas_float(const char *p, const char *end, double *out)
{
    bool negative = false;
    if (p < end && (*p == '-' || *p == '+')) negative = *p++ == '-';
    double number = 0.0;
    int exponent = 0, num_digits = 0, num_decimals = 0;
    const int max_digits = 17;
    while (p < end && is_digit(*p)) {
        if (num_digits < max_digits) {
            number = number * 10. + (*p - '0');
            ++num_digits;
        } else {
            ++exponent;
        }
        ++p;
    }
    if (p < end && *p == '.') {
        ++p;
        while (num_digits < max_digits && p < end && is_digit(*p)) {
            number = number * 10. + (*p - '0');
            ++p;
            ++num_digits;
            ++num_decimals;
        }
        if (num_digits >= max_digits)
            while (p < end && is_digit(*p)) ++p;
        exponent -= num_decimals;
    }
    if (num_digits == 0) return false;
    if (negative) number = -number;
    if (p < end && (*p == 'e' || *p == 'E')) {
        ++p;
        bool eneg = false;
        if (p < end && (*p == '-' || *p == '+')) eneg = *p++ == '-';
        int nd = 0, n = 0;
        while (nd < max_digits && p < end && is_digit(*p)) {
            n = n * 10 + (*p - '0');
            ++nd;
            ++p;
        }
        if (nd == 0) return false;
        exponent += eneg ? -n : n;
    }
    if (p != end) return false;                       // anything else (nan, inf, blanks, thousands separators): pandas' business
    if (exponent > 308) return false;
    if (exponent > 0)
        number *= POW10[exponent];
    else if (exponent < -308) {
        if (exponent < -616)
            number = 0.;
        else {
            number /= POW10[-308 - exponent];
            number /= POW10[308];
        }
    } else
        number /= POW10[-exponent];
    if (std::isinf(number)) return false;
    *out = number;
    return true;
}

}  // namespace

extern "C" int pfmscan_profile_parse(const char *buf, int64_t n, int n_cols, int64_t capacity_rows, double *out, int64_t *n_rows)
{
    if ((!buf && n > 0) || n < 0 || n_cols < 1 || capacity_rows < 0 || !n_rows || (!out && capacity_rows > 0))
        return fail(nullptr, PFMSCAN_E_BADARG, "profile_parse: bad argument");
    // rows after the header line: <first column, dropped> TAB value x n_cols; LF or CRLF; nothing else
    int64_t pos = 0, rows = 0;
    const char *nl = n ? static_cast<const char *>(std::memchr(buf, '\n', (size_t)n)) : nullptr;
    pos = nl ? (nl - buf) + 1 : n;                      // the header is the caller's (it holds the column letters)
    while (pos < n) {
        nl = static_cast<const char *>(std::memchr(buf + pos, '\n', (size_t)(n - pos)));
        int64_t e = nl ? nl - buf : n;
        const int64_t next = nl ? e + 1 : n;
        if (e > pos && buf[e - 1] == '\r') --e;
        if (e == pos) return fail(nullptr, PFMSCAN_E_BADSHAPE, "profile_parse: empty line");     // pandas skips blank lines: leave it to pandas
        const char *p = buf + pos, *end = buf + e;
        const char *tab = static_cast<const char *>(std::memchr(p, '\t', (size_t)(end - p)));
        if (!tab) return fail(nullptr, PFMSCAN_E_BADSHAPE, "profile_parse: a row without fields");
        p = tab + 1;
        for (int c = 0; c < n_cols; ++c) {
            const char *q = c + 1 < n_cols ? static_cast<const char *>(std::memchr(p, '\t', (size_t)(end - p))) : end;
            if (!q) return fail(nullptr, PFMSCAN_E_BADSHAPE, "profile_parse: too few fields in a row");
            double v;
            if (!pandas_float(p, q, &v)) return fail(nullptr, PFMSCAN_E_BADSHAPE, "profile_parse: a field that is no plain number");
            if (rows < capacity_rows) out[rows * n_cols + c] = v;
            p = q + 1;
        }
        ++rows;
        pos = next;
    }
    *n_rows = rows;
    if (rows > capacity_rows) return fail(nullptr, PFMSCAN_E_CAPACITY, "profile_parse: more rows than capacity");
    return PFMSCAN_OK;
}

extern "C" {

// rows without their last column -> rows with it: "\t<id>" in front of every line end that is not inside a quoted field
int pfmscan_tsv_number(const char *in, int64_t n, int64_t first_id, char *out, int64_t capacity, int64_t *n_out, int64_t *n_rows,
                       int *in_quotes)
{
    if ((!in && n > 0) || (!out && capacity > 0) || !n_out || !n_rows || !in_quotes || n < 0 || capacity < 0 || first_id < 0)
        return fail(nullptr, PFMSCAN_E_BADARG, "tsv_number: bad argument");
    int q = *in_quotes ? 1 : 0;
    int64_t rows = 0;
    char *p = out, *const end = out + capacity;
    const char *s = in, *const stop = in + n;
    while (s < stop) {
        // everything up to the next quote or line end is copied as it is
        const char *e = s;
        while (e < stop && *e != '\n' && *e != '"') ++e;
        if (end - p < (e - s) + 24) return fail(nullptr, PFMSCAN_E_CAPACITY, "tsv_number: output buffer too small");
        std::memcpy(p, s, (size_t)(e - s));
        p += e - s;
        if (e == stop) break;
        if (*e == '"') {
            q ^= 1;                                                  // a doubled quote toggles twice
            *p++ = '"';
        } else if (q) {
            *p++ = '\n';                                             // a line break inside a quoted field
        } else {
            *p++ = '\t';
            p = put_int(p, first_id + rows);
            *p++ = '\n';
            ++rows;
        }
        s = e + 1;
    }
    *in_quotes = q;
    *n_out = p - out;
    *n_rows = rows;
    return PFMSCAN_OK;
}

int pfmscan_tsv_format(const pfmscan_tsv_column *cols, int n_cols, int64_t n_rows, int64_t first_match_id, char *out,
                       int64_t capacity, int64_t *need, int64_t *pieces, int *n_pieces, int n_threads)
{
    if (!cols || n_cols <= 0 || n_rows < 0 || !need || !pieces || !n_pieces || capacity < 0 || (!out && capacity > 0))
        return fail(nullptr, PFMSCAN_E_BADARG, "tsv_format: bad argument");
    const int threads = pick_threads(std::min(n_threads > 0 ? n_threads : PFMSCAN_TSV_MAX_PIECES, PFMSCAN_TSV_MAX_PIECES),
                                     (n_rows + 16383) / 16384);                    // a thread is worth starting for ~16k rows
    // Every thread owns a slice of `out` it cannot overrun: the most bytes ITS rows can take.  The fixed part of a row is a
    // bound per column kind; the values of INDEXED / SPAN columns are counted row by row (their real lengths, a doubled
    // quote for every byte of a span) -- sizing the buffer as rows x the longest value any row uses let ONE 1-kB FASTA header
    // in a 4 M-row chunk ask for 8 GB.
    int64_t fixed = n_cols + (first_match_id >= 0 ? 21 : 0);
    bool any_var = false;
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
        default: any_var = true;
        }
    }
    std::vector<int64_t> room((size_t)threads, 0);
    parallel_ranges(n_rows, threads, [&](int t, int64_t a, int64_t b) {
        int64_t var = 0;
        if (any_var)
            for (int c = 0; c < n_cols; ++c) {
                const pfmscan_tsv_column &col = cols[c];
                const int64_t *index = static_cast<const int64_t *>(col.data);
                const int64_t *aux = static_cast<const int64_t *>(col.aux);
                if (col.kind == PFMSCAN_TSV_INDEXED)
                    for (int64_t r = a; r < b; ++r) var += aux[index[r] + 1] - aux[index[r]];
                else if (col.kind == PFMSCAN_TSV_SPAN)
                    for (int64_t r = a; r < b; ++r) var += 2 * aux[2 * index[r] + 1] + 2;    // every byte a doubled quote
            }
        room[(size_t)t] = (b - a) * fixed + var;
    });
    std::vector<int64_t> start((size_t)threads + 1, 0);
    for (int t = 0; t < threads; ++t) start[(size_t)t + 1] = start[(size_t)t] + room[(size_t)t];
    *need = start[(size_t)threads];
    *n_pieces = 0;
    if (*need > capacity) return fail(nullptr, PFMSCAN_E_CAPACITY, "tsv_format: output buffer too small");
    std::vector<int64_t> used((size_t)threads, 0), from((size_t)threads, 0);
    parallel_ranges(n_rows, threads, [&](int t, int64_t a, int64_t b) {
        char *const base = out + start[(size_t)t];
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
                    for (int64_t j = 0; j < col.width; ++j) *p++ = letters[codes[j] & 15];
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
        from[(size_t)t] = start[(size_t)t];
        used[(size_t)t] = p - base;
    });
    for (int t = 0; t < threads; ++t) {
        pieces[2 * t] = from[(size_t)t];
        pieces[2 * t + 1] = used[(size_t)t];
    }
    *n_pieces = threads;
    return PFMSCAN_OK;
}

// round(x, decimals) of Python floats (rnascan.py:273 rounds every reported score to 3 places): the decimal string with
// `decimals` digits after the point that is NEAREST to the exact binary value (ties to even), read back as a double --
// float.__round__ does it through dtoa / strtod.  numpy.round (x * 10^d -> rint -> / 10^d) differs from it near ties.
// Fast path: y = x * 10^d is within half an ulp of the exact product, so when y is clear of a tie by more than that the
// integer r = rint(y) is the exact product's rounding too, and r / 10^d (ONE correctly rounded division of two exact
// integers) is the double nearest to the decimal.  Only values within a few ulps of a tie take the string route.
int pfmscan_round_decimals(const double *in, int64_t n, int decimals, double *out, int n_threads)
{
    if (n < 0 || (n > 0 && (!in || !out)) || decimals < 0 || decimals > 15) return fail(nullptr, PFMSCAN_E_BADARG, "pfmscan_round_decimals: bad argument");
    double scale = 1.0;
    for (int i = 0; i < decimals; ++i) scale *= 10.0;
    parallel_ranges(n, pick_threads(n_threads, n >> 16), [&](int, int64_t a, int64_t b) {
        char buf[400];
        for (int64_t i = a; i < b; ++i) {
            const double x = in[i];
            if (!std::isfinite(x)) {
                out[i] = x;
                continue;
            }
            const double y = x * scale;
            const double r = std::nearbyint(y);
            const double slack = 0.5 - std::fabs(y - r);
            if (std::fabs(y) < 0x1p51 && slack > std::fabs(y) * 0x1p-50) {
                out[i] = r / scale;
            } else if (std::fabs(y) >= 0x1p52) {
                out[i] = x;                                         // no fractional digits left to drop (decimals <= 15)
            } else {
                std::snprintf(buf, sizeof buf, "%.*f", decimals, x);
                out[i] = std::strtod(buf, nullptr);
            }
        }
    });
    return PFMSCAN_OK;
}

}  // extern "C"
