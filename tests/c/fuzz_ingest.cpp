// ASan/UBSan fuzz driver for the host-only ingest code (CPU build, tests/test_ingest.py::test_ingest_under_sanitizers):
// random FASTA-ish / profile-ish bytes in exact-size heap buffers through pfmscan_fasta_index / _ids / _encode,
// pfmscan_gather_spans, pfmscan_tsv_format, pfmscan_tsv_number and pfmscan_profile_parse.  usage: fuzz_ingest [iterations]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>
#include "pfmscan.h"
int main(int argc, char **argv) {
    std::mt19937_64 rng(1234);
    const int iterations = argc > 1 ? atoi(argv[1]) : 20000;
    const char alpha[] = "ACGUTacgutNn >\t\r\n\n\n>xX0123456789.eE+-";
    long checks = 0;
    for (int it = 0; it < iterations; ++it) {
        const size_t n = rng() % 400;
        std::string s(n, ' ');
        for (auto &c : s) c = alpha[rng() % (sizeof(alpha) - 1)];
        std::vector<uint8_t> exact(s.begin(), s.end());       // exact-size heap buffer: overreads trip ASan
        int64_t nrec = 0;
        const int threads = 1 + rng() % 5;
        std::vector<int64_t> ho(n + 1), hl(n + 1), so(n + 1), se(n + 1), nl(n + 1);
        int rc = pfmscan_fasta_index(exact.data(), (int64_t)n, (int64_t)n + 1, ho.data(), hl.data(), so.data(), se.data(), nl.data(), &nrec, threads);
        if (rc != 0) { printf("index rc %d\n", rc); return 1; }
        if (nrec) {
            std::vector<int64_t> io(nrec), il(nrec); int ascii = 0;
            pfmscan_fasta_ids(exact.data(), ho.data(), hl.data(), nrec, io.data(), il.data(), &ascii);
            int64_t total = 0; for (int64_t r = 0; r < nrec; ++r) total += nl[r] + 1;
            std::vector<uint8_t> codes(total), lut(256, 7); lut['A'] = 0; lut['C'] = 1; lut['G'] = 2; lut['U'] = 3;
            std::vector<int64_t> offs(nrec);
            rc = pfmscan_fasta_encode(exact.data(), so.data(), se.data(), nl.data(), 0, nrec, lut.data(), 7, codes.data(), offs.data(), threads);
            if (rc != 0) { printf("encode rc %d\n", rc); return 1; }
            std::vector<int64_t> spans(2 * nrec); for (int64_t r = 0; r < nrec; ++r) { spans[2 * r] = ho[r]; spans[2 * r + 1] = hl[r]; }
            int64_t nb = 0; std::vector<uint8_t> blob(n + nrec + 1);
            pfmscan_gather_spans(exact.data(), spans.data(), nrec, '\n', blob.data(), (int64_t)blob.size(), &nb);
            // rows: spans + windows + numbers
            std::vector<int64_t> idx(50), pos(50), iv(50); std::vector<float> f(50); std::vector<double> d(50);
            for (int i = 0; i < 50; ++i) { idx[i] = rng() % nrec; pos[i] = total > 4 ? rng() % (total - 4) : 0; iv[i] = (int64_t)rng(); f[i] = (float)((double)(int64_t)rng() * 1e-9); uint64_t b = rng(); memcpy(&d[i], &b, 8); }
            pfmscan_tsv_column cols[5] = {{PFMSCAN_TSV_SPAN, 0, idx.data(), spans.data(), exact.data(), 0},
                                          {PFMSCAN_TSV_I64, 0, iv.data(), nullptr, nullptr, 0},
                                          {PFMSCAN_TSV_F32, 0, f.data(), nullptr, nullptr, 0},
                                          {PFMSCAN_TSV_F64, 0, d.data(), nullptr, nullptr, 0},
                                          {PFMSCAN_TSV_WINDOW, 0, pos.data(), codes.data(), "ACGU????", total > 4 ? 4 : 0}};
            int64_t need = 0, pieces[32]; int np = 0;
            pfmscan_tsv_format(cols, 5, 50, 1, nullptr, 0, &need, pieces, &np, threads);
            std::vector<char> out(need);
            rc = pfmscan_tsv_format(cols, 5, 50, 1, out.data(), need, &need, pieces, &np, threads);
            if (rc != 0) { printf("tsv rc %d\n", rc); return 1; }
        }
        // Match_ID splice over the same bytes (quotes, tabs and line ends at random places), in two blocks cut at a
        // random point, into exact-size output buffers
        {
            const std::string t = s + "\"q\"\n";
            std::vector<char> in(t.begin(), t.end());
            const size_t cut = t.empty() ? 0 : rng() % (t.size() + 1);
            int state = 0; int64_t nid = 1 + (int64_t)(rng() % 1000000007ull);
            for (int part = 0; part < 2; ++part) {
                const char *src = in.data() + (part ? cut : 0);
                const int64_t len = part ? (int64_t)(t.size() - cut) : (int64_t)cut;
                int64_t lines = 0; for (int64_t i = 0; i < len; ++i) lines += src[i] == '\n';
                std::vector<char> exact_src(src, src + len);
                std::vector<char> outb((size_t)(len + 21 * (lines + 1) + 32));
                int64_t n_out = 0, n_rows = 0;
                rc = pfmscan_tsv_number(exact_src.data(), len, nid, outb.data(), (int64_t)outb.size(), &n_out, &n_rows, &state);
                if (rc != 0 || n_rows > lines || n_out < len) { printf("number rc %d\n", rc); return 1; }
                nid += n_rows;
            }
        }
        // profile-ish text
        std::vector<double> prof(7 * (n + 2)); int64_t rows = 0;
        std::vector<char> exact2(s.begin(), s.end());
        pfmscan_profile_parse(exact2.data(), (int64_t)n, 1 + rng() % 7, (int64_t)n + 2, prof.data(), &rows);
        ++checks;
    }
    printf("ok %ld\n", checks);
    return 0;
}
