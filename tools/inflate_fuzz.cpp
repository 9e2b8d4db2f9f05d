// Diagnostic / test helper (host only): the repository's own inflate (ParGzReader::MarkerInflate, faqcs_pargz.h) against zlib on
// generated deflate streams -- every level and strategy zlib has, stored blocks, fixed and dynamic codes -- and on damaged copies of them.
//   byte mode   : the bytes must equal zlib's; a damaged stream must either fail or give what zlib gives
//   symbol mode : started at the stream's beginning there are no markers and the symbols are the bytes; started at a LATER block
//                 boundary (found with zlib's Z_BLOCK) the symbols narrowed through the true 32 KB window must equal the text behind it
// Built by tests/test_cli_host.py with -fsanitize=address,undefined: a read or write outside the buffers fails the test.
//   g++ -O1 -g -std=c++17 -fsanitize=address,undefined -o inflate_fuzz tools/inflate_fuzz.cpp -lz && ./inflate_fuzz [seed] [cases]
#include "../faqcs_amd/csrc/faqcs_pargz.h"

#include <random>

using MI = ParGzReader::MarkerInflate;

static std::vector<uint8_t> make_text(std::mt19937_64 &g, size_t n)
{
    std::vector<uint8_t> t(n);
    const int kind = (int)(g() % 6);
    if (kind == 0) for (auto &x : t) x = (uint8_t)g();                                  // noise: stored blocks
    else if (kind == 1) for (auto &x : t) x = "ACGT"[g() & 3];                           // short matches everywhere
    else if (kind == 2) { uint8_t c = 'I'; for (auto &x : t) { if (g() % 40 == 0) c = (uint8_t)(33 + g() % 42); x = c; } } // runs: distance 1
    else if (kind == 3) { for (size_t i = 0; i < n; ++i) t[i] = (uint8_t)(i < 700 ? g() : (g() % 50 == 0 ? g() : t[i - 1 - g() % 700])); } // near copies
    else if (kind == 4) { for (size_t i = 0; i < n; ++i) t[i] = (uint8_t)(i < 40000 ? g() % 7 + 'a' : (g() % 900 == 0 ? g() : t[i - 30000 - g() % 2000])); } // far distances
    else { // FASTQ-like records
        size_t i = 0, id = 0;
        while (i < n) {
            char rec[700];
            std::string s, q;
            const size_t L = 30 + g() % 200;
            for (size_t k = 0; k < L; ++k) { s += "ACGTN"[g() % 41 == 0 ? 4 : g() & 3]; q += (char)(33 + (g() % 10 ? 40 : g() % 41)); }
            const int m = snprintf(rec, sizeof rec, "@SYN:%09zu/1\n%s\n+\n%s\n", id++, s.c_str(), q.c_str());
            for (int k = 0; k < m && i < n; ++k) t[i++] = (uint8_t)rec[k];
        }
    }
    return t;
}
static std::vector<uint8_t> deflate_raw(const std::vector<uint8_t> &t, int level, int strategy, std::mt19937_64 &g)
{
    z_stream z;
    memset(&z, 0, sizeof z);
    deflateInit2(&z, level, Z_DEFLATED, -15, 1 + (int)(g() % 9), strategy);
    std::vector<uint8_t> o(deflateBound(&z, (uLong)t.size()) + 4096 + t.size() / 8);
    z.next_out = o.data(); z.avail_out = (uInt)o.size();
    // in a few pieces, with full / sync flushes between them (empty stored blocks, byte alignment)
    size_t at = 0;
    while (at < t.size()) {
        const size_t n = std::min<size_t>(t.size() - at, 1 + g() % (t.size() / 2 + 1));
        z.next_in = const_cast<Bytef *>(t.data() + at); z.avail_in = (uInt)n;
        const int fl = (int)(g() % 4);
        deflate(&z, fl == 0 ? Z_SYNC_FLUSH : fl == 1 ? Z_FULL_FLUSH : fl == 2 ? Z_BLOCK : Z_NO_FLUSH);
        at += n;
    }
    z.next_in = nullptr; z.avail_in = 0;
    if (deflate(&z, Z_FINISH) != Z_STREAM_END) { fprintf(stderr, "deflate failed\n"); exit(2); }
    o.resize(o.size() - z.avail_out);
    deflateEnd(&z);
    return o;
}
// zlib's verdict on a raw stream: the bytes, or failure (want = how many bytes the caller expects at most)
static bool zlib_inflate(const std::vector<uint8_t> &d, size_t want, std::vector<uint8_t> &out)
{
    z_stream z;
    memset(&z, 0, sizeof z);
    inflateInit2(&z, -15);
    out.assign(want + 1, 0);
    z.next_in = const_cast<Bytef *>(d.data()); z.avail_in = (uInt)d.size();
    z.next_out = out.data(); z.avail_out = (uInt)out.size();
    const int rc = inflate(&z, Z_FINISH);
    out.resize(out.size() - z.avail_out);
    inflateEnd(&z);
    return rc == Z_STREAM_END;
}
template <class Sym>
static bool own_inflate(const uint8_t *d, size_t n, uint64_t start_bit, size_t limit, std::vector<Sym> &out)
{
    std::unique_ptr<MI> mi(new MI);
    mi->begin(d, n, start_bit);
    std::vector<Sym> buf(4096);
    Sym *o = buf.data();
    size_t pos = 0, cap = buf.size();
    auto grow = [&](size_t need) -> Sym * { if (need > limit + 400) return nullptr; buf.resize(std::max(need, buf.size() * 2)); cap = buf.size(); return buf.data(); };
    for (;;) {
        if (!mi->decode_block(o, pos, cap, grow)) return false;
        if (mi->final_block) break;
    }
    out.assign(buf.begin(), buf.begin() + (ptrdiff_t)pos);
    return true;
}

int main(int argc, char **argv)
{
    const uint64_t seed = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1;
    const int cases = argc > 2 ? atoi(argv[2]) : 300;
    std::mt19937_64 g(seed);
    size_t damaged_fail = 0, damaged_same = 0, mid_starts = 0, lenient = 0;
    for (int c = 0; c < cases; ++c) {
        const size_t n = c % 17 == 0 ? g() % 40 : 1 + g() % (c % 5 == 0 ? 400000 : 70000);
        const std::vector<uint8_t> text = make_text(g, n);
        static const int strategies[] = {Z_DEFAULT_STRATEGY, Z_FILTERED, Z_HUFFMAN_ONLY, Z_RLE, Z_FIXED};
        const int level = (int)(g() % 10), strategy = strategies[g() % 5];
        const std::vector<uint8_t> d = deflate_raw(text, level, strategy, g);
        // 1. byte mode, whole stream
        std::vector<uint8_t> b;
        if (!own_inflate<uint8_t>(d.data(), d.size(), 0, n, b) || b != text) { printf("case %d: byte mode differs (level %d strategy %d, %zu bytes)\n", c, level, strategy, n); return 1; }
        // 2. symbol mode from the beginning: no markers
        std::vector<uint16_t> s;
        if (!own_inflate<uint16_t>(d.data(), d.size(), 0, n, s) || s.size() != n) { printf("case %d: symbol mode fails\n", c); return 1; }
        for (size_t i = 0; i < n; ++i) if (s[i] != text[i]) { printf("case %d: symbol %zu differs\n", c, i); return 1; }
        // 3. symbol mode from a later block boundary (zlib tells where the blocks end)
        {
            z_stream z;
            memset(&z, 0, sizeof z);
            inflateInit2(&z, -15);
            std::vector<uint8_t> tmp(n + 1);
            z.next_in = const_cast<Bytef *>(d.data()); z.avail_in = (uInt)d.size();
            z.next_out = tmp.data(); z.avail_out = (uInt)tmp.size();
            std::vector<std::pair<uint64_t, size_t>> bounds; // (bit position, bytes in front)
            for (;;) {
                const int rc = inflate(&z, Z_BLOCK);
                if (rc != Z_OK) break;
                if ((z.data_type & 128) && !(z.data_type & 64)) bounds.emplace_back((uint64_t)(z.next_in - d.data()) * 8 - (uint64_t)(z.data_type & 63), (size_t)(z.next_out - tmp.data()));
            }
            inflateEnd(&z);
            if (!bounds.empty()) {
                const auto [bit, front] = bounds[g() % bounds.size()];
                std::vector<uint16_t> m;
                if (!own_inflate<uint16_t>(d.data(), d.size(), bit, n, m) || m.size() != n - front) { printf("case %d: start at bit %llu fails\n", c, (unsigned long long)bit); return 1; }
                uint8_t win[32768];
                memset(win, 0, sizeof win);
                const size_t have = std::min<size_t>(front, sizeof win);
                memcpy(win + sizeof win - have, text.data() + front - have, have);
                std::vector<uint8_t> nar(m.size() + 1);
                faqcs_narrow(m.data(), nar.data(), m.size(), win);
                if (memcmp(nar.data(), text.data() + front, m.size()) != 0) { printf("case %d: narrowed text differs behind byte %zu\n", c, front); return 1; }
                ++mid_starts;
            }
        }
        // 4. damaged copies: never a crash; a success must be zlib's success with zlib's bytes (a stream zlib refuses for an incomplete
        //    code that is never used may pass here: counted, the CRC behind every member is what catches damage)
        for (int k = 0; k < 6 && !d.empty(); ++k) {
            std::vector<uint8_t> x = d;
            const int how = (int)(g() % 3);
            if (how == 0) x[g() % x.size()] ^= (uint8_t)(1u << (g() % 8));
            else if (how == 1) x.resize(g() % x.size());
            else for (int r = 0; r < 4; ++r) x[g() % x.size()] = (uint8_t)g();
            std::vector<uint8_t> mine, theirs;
            const bool ok_mine = own_inflate<uint8_t>(x.data(), x.size(), 0, n + 70000, mine);
            const bool ok_z = zlib_inflate(x, n + 70000 + 400, theirs);
            if (!ok_mine) { ++damaged_fail; continue; }
            if (!ok_z) { ++lenient; continue; }
            if (mine != theirs) { printf("case %d: a damaged stream inflates to other bytes than zlib's\n", c); return 1; }
            ++damaged_same;
            std::vector<uint16_t> ms;
            (void)own_inflate<uint16_t>(x.data(), x.size(), 0, n + 70000, ms);
        }
    }
    printf("%d streams equal to zlib's; %zu starts inside a stream; damaged copies: %zu refused, %zu inflated as zlib inflates them, %zu accepted where zlib refuses\n",
           cases, mid_starts, damaged_fail, damaged_same, lenient);
    return 0;
}
