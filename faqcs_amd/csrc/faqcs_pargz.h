// faqcs_pargz.h -- parallel inflate of an ORDINARY gzip file: what real FaQCs users feed it (fastq.cpp:8-125 reads through gzread, one
// thread).  Host code only (threads, an own inflate, zlib for the start trials and the serial rest); used by faqcs_cli.cpp's Source,
// its BgzfReader and `faqcs_mi --pargz_cat`.  Not part of libfaqcs_mi.so.
//
// A deflate stream has no index, and a block can refer to the 32 KB in front of it.  The approach is rapidgzip's / pugz's:
//   1. the compressed file is cut at fixed byte offsets; the worker of a piece looks for the first bit position at or behind its offset
//      that can start a dynamic-Huffman block -- a complete precode, valid repeat codes, an end-of-block code, complete literal / distance
//      codes (what zlib's inflate_table accepts): about one bit position in a million passes, and a trial inflate decides;
//   2. it inflates from there, block by block, until it stands exactly where a later piece's worker started -- a start that no
//      predecessor arrives at was a false positive and is dropped (its predecessor goes on through it).  A piece nobody has claimed yet
//      is NOT run through: the block boundary becomes that piece's start and the worker ends there (the consumer is the slower side then);
//   3. the window in front of a piece is unknown, so the piece is inflated by this file's own decoder (MarkerInflate) into 16-bit SYMBOLS:
//      a byte, or 0x8000 | k = "the byte at offset k of the unknown window".  A copy that reaches in front of the piece writes such
//      markers, one that copies markers copies them on: ONE pass, out of band, any text.  (Round 5 made two zlib passes with dictionaries
//      that encode the offset in the bytes and needed ASCII: FAQCS_MI_PARGZ_TWO_PASS=1 keeps that path for A/B.)
//   4. a chain thread walks the pieces in file order and resolves only the LAST 32 KB of each -- the window of the next one --; the bulk of
//      a piece (in FASTQ 12 - 30 % of the symbols are markers: every header line copies the one before, through any chain of copies) is
//      narrowed to bytes by the workers in parallel once its window is known, 64 KB at a time together with its CRC-32 (PCLMULQDQ); the
//      consumer hands the pieces out in order; the CRCs (crc32_combine) must equal the member's trailer at the end, as must the length:
//      any slip of the speculation ends the input with an error (like a corrupt file under gzread), it never yields different bytes silently.
// The decoder (one shift of the bit buffer per code, the next entry loaded before a match is copied) also has a byte-output form for
// streams whose beginning is known: faqcs_cli.cpp's BgzfReader inflates BGZF members with it.
// What follows a member (concatenated members: `cat lane1.gz lane2.gz`, which gzread reads as one stream) starts the same machinery
// again at the next member's header when enough of the file is left; a short rest goes through zlib's gzip decoder in the consumer.
// When a member's last block has been seen, nobody claims a piece any more and the buffers of the pieces behind it are given back:
// the speculation does not run on through the members that follow (twice the text of such a file stayed resident).
#pragma once
#include <zlib.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

// ---- FAQCS_MI_TIMING=1: what each kind of thread did with its life (CPU seconds against seconds alive), printed by faqcs_mi at the end.
// A role whose threads were busy for all of their life is the stage the others wait for.
#include <map>
#include <time.h>
struct FaqcsThreadCpu {
    struct Acc { double cpu = 0, alive = 0, max_cpu = 0; unsigned n = 0; };
    static std::mutex &mu() { static std::mutex m; return m; }
    static std::map<std::string, Acc> &table() { static std::map<std::string, Acc> t; return t; }
    static bool on() { static const bool v = getenv("FAQCS_MI_TIMING") != nullptr; return v; }
    static double clock_s(clockid_t c) { timespec ts; clock_gettime(c, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }
    const char *role;
    double t0 = 0;
    explicit FaqcsThreadCpu(const char *r) : role(r) { if (on()) t0 = clock_s(CLOCK_MONOTONIC); }
    ~FaqcsThreadCpu()
    {
        if (!on()) return;
        const double cpu = clock_s(CLOCK_THREAD_CPUTIME_ID), alive = clock_s(CLOCK_MONOTONIC) - t0;
        std::lock_guard<std::mutex> l(mu());
        Acc &a = table()[role];
        a.cpu += cpu; a.alive += alive; a.max_cpu = std::max(a.max_cpu, cpu); ++a.n;
    }
    static void report(FILE *f)
    {
        if (!on()) return;
        std::lock_guard<std::mutex> l(mu());
        for (const auto &kv : table())
            fprintf(f, "[faqcs_mi] threads '%s': %u, %.3f CPU-s of %.3f s alive (the busiest %.3f CPU-s)\n", kv.first.c_str(), kv.second.n, kv.second.cpu, kv.second.alive, kv.second.max_cpu);
    }
};

// ---- CRC-32 (the gzip polynomial) by carry-less multiplication: zlib's table-driven crc32 does about 1 GB/s on a core, which made the
// CRC of a piece a sixth of the inflate's CPU time.  Folding of 64 bytes a step with PCLMULQDQ (Gopal et al., "Fast CRC Computation
// for Generic Polynomials Using PCLMULQDQ Instruction", Intel 2009; the constants are the paper's for the reflected polynomial
// 0xEDB88320).  Works on the raw (inverted) CRC state; the caller handles tails and machines without the instruction.
#include <immintrin.h>
__attribute__((target("pclmul,sse4.1"))) static inline uint32_t faqcs_crc32_fold(const uint8_t *buf, size_t len, uint32_t state) // len >= 64, a multiple of 16
{
    alignas(16) static const uint64_t k1k2[2] = {0x0154442bd4ull, 0x01c6e41596ull};
    alignas(16) static const uint64_t k3k4[2] = {0x01751997d0ull, 0x00ccaa009eull};
    alignas(16) static const uint64_t k5k0[2] = {0x0163cd6124ull, 0x0000000000ull};
    alignas(16) static const uint64_t poly[2] = {0x01db710641ull, 0x01f7011641ull};
    __m128i x1 = _mm_loadu_si128((const __m128i *)(buf + 0)), x2 = _mm_loadu_si128((const __m128i *)(buf + 16));
    __m128i x3 = _mm_loadu_si128((const __m128i *)(buf + 32)), x4 = _mm_loadu_si128((const __m128i *)(buf + 48));
    x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)state));
    __m128i x0 = _mm_load_si128((const __m128i *)k1k2);
    buf += 64; len -= 64;
    while (len >= 64) { // four independent 128-bit lanes, each folded over 512 bits
        const __m128i x5 = _mm_clmulepi64_si128(x1, x0, 0x00), x6 = _mm_clmulepi64_si128(x2, x0, 0x00);
        const __m128i x7 = _mm_clmulepi64_si128(x3, x0, 0x00), x8 = _mm_clmulepi64_si128(x4, x0, 0x00);
        x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x2 = _mm_clmulepi64_si128(x2, x0, 0x11);
        x3 = _mm_clmulepi64_si128(x3, x0, 0x11); x4 = _mm_clmulepi64_si128(x4, x0, 0x11);
        x1 = _mm_xor_si128(_mm_xor_si128(x1, x5), _mm_loadu_si128((const __m128i *)(buf + 0)));
        x2 = _mm_xor_si128(_mm_xor_si128(x2, x6), _mm_loadu_si128((const __m128i *)(buf + 16)));
        x3 = _mm_xor_si128(_mm_xor_si128(x3, x7), _mm_loadu_si128((const __m128i *)(buf + 32)));
        x4 = _mm_xor_si128(_mm_xor_si128(x4, x8), _mm_loadu_si128((const __m128i *)(buf + 48)));
        buf += 64; len -= 64;
    }
    x0 = _mm_load_si128((const __m128i *)k3k4); // the four lanes into one
    __m128i x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x3), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x4), x5);
    while (len >= 16) { // what is left in steps of 16
        x2 = _mm_loadu_si128((const __m128i *)buf);
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
        x1 = _mm_clmulepi64_si128(x1, x0, 0x11); x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
        buf += 16; len -= 16;
    }
    // 128 -> 64 bits
    x2 = _mm_clmulepi64_si128(x1, x0, 0x10);
    x3 = _mm_setr_epi32(~0, 0, ~0, 0);
    x1 = _mm_xor_si128(_mm_srli_si128(x1, 8), x2);
    x0 = _mm_loadl_epi64((const __m128i *)k5k0);
    x2 = _mm_srli_si128(x1, 4);
    x1 = _mm_and_si128(x1, x3);
    x1 = _mm_xor_si128(_mm_clmulepi64_si128(x1, x0, 0x00), x2);
    // Barrett reduction 64 -> 32 bits
    x0 = _mm_load_si128((const __m128i *)poly);
    x2 = _mm_and_si128(x1, x3);
    x2 = _mm_clmulepi64_si128(x2, x0, 0x10);
    x2 = _mm_and_si128(x2, x3);
    x2 = _mm_clmulepi64_si128(x2, x0, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    return (uint32_t)_mm_extract_epi32(x1, 1);
}
// crc32_z() with the same arguments and the same result
static inline uint32_t faqcs_crc32(uint32_t crc, const uint8_t *buf, size_t len)
{
    static const bool fold = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1") && !getenv("FAQCS_MI_NO_PCLMUL");
    if (fold && len >= 64) {
        const size_t head = len & ~(size_t)15;
        crc = ~faqcs_crc32_fold(buf, head, ~crc);
        buf += head; len -= head;
    }
    return len ? (uint32_t)crc32_z(crc, buf, len) : crc;
}
// 16-bit symbols -> text: a symbol below 256 is its byte, 0x8000 | k the byte at offset k of the 32 KB window `w`
// (true: there was a marker among them)
__attribute__((target("avx2"))) static inline bool faqcs_narrow_avx2(const uint16_t *sym, uint8_t *o, size_t n, const uint8_t *w)
{
    size_t k = 0;
    bool any = false;
    for (; k + 32 <= n; k += 32) {
        const __m256i a = _mm256_loadu_si256((const __m256i *)(sym + k)), b = _mm256_loadu_si256((const __m256i *)(sym + k + 16));
        if (_mm256_movemask_epi8(_mm256_or_si256(a, b)) & 0xAAAAAAAAu) { // a marker among the 32 (the top bit of a symbol)
            // without a branch per symbol: in the pieces of a FASTQ file 12 - 30 % of the symbols are markers, in no order a predictor could learn
            // (the window is read for a literal too -- at its own value, which is inside it -- and not used)
            for (size_t t = k; t < k + 32; ++t) { const uint32_t v = sym[t]; const uint8_t m = w[v & 0x7fffu]; o[t] = (v & 0x8000u) ? m : (uint8_t)v; }
            any = true;
            continue;
        }
        _mm256_storeu_si256((__m256i *)(o + k), _mm256_permute4x64_epi64(_mm256_packus_epi16(a, b), 0xD8));
    }
    for (; k < n; ++k) { const uint16_t v = sym[k]; any |= v >= 256; o[k] = v < 256 ? (uint8_t)v : w[v & 0x7fffu]; }
    return any;
}
static inline bool faqcs_narrow(const uint16_t *sym, uint8_t *o, size_t n, const uint8_t *w)
{
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2) return faqcs_narrow_avx2(sym, o, n, w);
    bool any = false;
    for (size_t k = 0; k < n; ++k) { const uint16_t v = sym[k]; any |= v >= 256; o[k] = v < 256 ? (uint8_t)v : w[v & 0x7fffu]; }
    return any;
}

struct ParGzReader {
    // output buffers: plain malloc'd memory, never zero-filled, recycled through a pool (a piece inflates to tens of megabytes: a fresh
    // std::vector per piece spent more time in page faults and memset than in inflate)
    struct Buf {
        uint8_t *p = nullptr; size_t cap = 0;
        uint8_t *data() { return p; }
        const uint8_t *data() const { return p; }
        size_t size() const { return cap; }
        uint8_t &operator[](size_t i) { return p[i]; }
        const uint8_t &operator[](size_t i) const { return p[i]; }
        void resize(size_t n) { if (n > cap) { p = (uint8_t *)realloc(p, n); cap = p ? n : 0; } }
    };
    std::vector<Buf> pool;
    Buf take_buf() { std::lock_guard<std::mutex> l(pm); if (pool.empty()) return Buf(); Buf b = pool.back(); pool.pop_back(); return b; }
    void give_buf(Buf &b) { if (!b.p) return; std::lock_guard<std::mutex> l(pm); pool.push_back(b); b = Buf(); }
    std::mutex pm;
    static constexpr size_t WIN = 32768;
    static constexpr uint64_t NO_START = ~0ull;
    struct Piece {
        std::atomic<uint64_t> start_bit{0};   // where its worker starts (NO_START: no block start found in its range); 0 = not known yet
        std::atomic<int> state{0};            // 0 free, 1 claimed (finding / inflating), 2 done, 3 failed
        uint64_t end_bit = 0;                 // the block boundary it stopped at (== start_bit of the piece that follows it in the chain)
        size_t next_piece = 0;                // index of that piece (or n_pieces: it ran to the end of the member)
        bool final_seen = false;              // the member's last block ended inside
        size_t trailer_at = 0;                // byte offset of the member's trailer (final_seen)
        bool known_window = false;            // piece 0: nothing lies in front of it (a marker in it is a distance too far back: invalid data)
        Buf out, mark;                        // the text; (pieces with an unknown window) the 16-bit symbols of MarkerInflate, narrowed into `out` by patch()
        size_t n_out = 0;
        std::vector<uint8_t> win;             // the true 32 KB in front of the piece (set by the chain thread)
        uint32_t crc = 0;
        std::atomic<int> crc_state{0};        // patch + CRC task: 0 not asked, 1 asked, 3 taken, 2 done
    };
    const uint8_t *base = nullptr;
    size_t size = 0, data_begin = 0, piece_bytes = 0, n_pieces = 0;
    int fd = -1;
    std::vector<Piece> pieces;
    std::vector<std::thread> workers;
    std::mutex m;
    std::condition_variable cv_work, cv_done;
    size_t next_claim = 0;       // next piece index a worker may claim
    size_t consumed = 0;         // pieces the consumer is done with (chain order, but indices only grow)
    size_t window_pieces = 0;    // how many pieces may be in flight behind `consumed`
    size_t armed_pieces = 0;     // (statistics) the pieces of the member armed last
    bool closing = false;
    bool failed = false;
    bool stop_claims = false;    // the member's last block has been seen by the chain: no piece behind it belongs to this member
    size_t rearm_min = 8u << 20; // a following member starts the parallel reader again when at least this much of the file is left
    int n_workers = 1;
    size_t piece_arg = 0;
    bool two_pass = [] { const char *e = getenv("FAQCS_MI_PARGZ_TWO_PASS"); return e && atoi(e) != 0; }(); // round 5's two zlib passes per piece (A/B; needs ASCII text)
    std::atomic<size_t> cancel_from{~(size_t)0}; // pieces from this index on start behind the member's trailer: their workers give up
    // FAQCS_PARGZ_TRACE=1: a time line of the pieces on stderr when the reader closes (diagnostic)
    struct Ev { double t; size_t piece; const char *what; size_t arg; };
    std::vector<Ev> trace_log;
    std::mutex trace_m;
    const bool tracing = getenv("FAQCS_PARGZ_TRACE") != nullptr;
    const std::chrono::steady_clock::time_point trace_t0 = std::chrono::steady_clock::now();
    void trace(size_t piece, const char *what, size_t arg = 0)
    {
        if (!tracing) return;
        const double t = std::chrono::duration<double>(std::chrono::steady_clock::now() - trace_t0).count();
        std::lock_guard<std::mutex> l(trace_m);
        trace_log.push_back(Ev{t, piece, what, arg});
    }
    // consumer state
    size_t cur = 0;              // piece to hand out next
    bool member_done = false, tail_init = false, tail_done = false, tail_mid = false;
    uint8_t last_win[WIN];       // the true last 32 KB handed out so far (index WIN - 1 = the most recent byte)
    size_t total_out = 0;
    std::vector<std::pair<uint32_t, size_t>> crcs; // (crc, length) of the pieces in chain order
    std::vector<size_t> crc_wait;                  // pieces whose CRC was asked for and not collected yet
    z_stream tz;
    std::vector<char> tail_out;
    size_t tail_from = 0;

    // ---- bits ----
    inline uint64_t peek(uint64_t bit) const
    { // 57 valid bits at `bit` (little-endian bit order, as deflate packs them); zeros past the end of the file
        const size_t by = (size_t)(bit >> 3);
        uint64_t v = 0;
        if (by + 8 <= size) memcpy(&v, base + by, 8);
        else for (size_t i = 0; by + i < size && i < 8; ++i) v |= (uint64_t)base[by + i] << (8 * i);
        return v >> (bit & 7);
    }
    // Can a non-final dynamic-Huffman block start at `bit`?  Mirrors zlib's checks (inflate.c TABLE .. CODELENS, inftrees.c).
    bool plausible_block(uint64_t bit) const
    {
        uint64_t v = peek(bit);
        if ((v & 7) != 4) return false; // BFINAL = 0, BTYPE = 2 (bits: 0, then 01 little-endian = value 2 -> 0b100)
        const unsigned hlit = (unsigned)((v >> 3) & 31) + 257, hdist = (unsigned)((v >> 8) & 31) + 1, hclen = (unsigned)((v >> 13) & 15) + 4;
        if (hlit > 286 || hdist > 30) return false;
        static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        uint8_t pl[19];
        memset(pl, 0, sizeof pl);
        uint64_t at = bit + 17 + 3 * (uint64_t)hclen;
        uint64_t pv = peek(bit + 17) & ((1ull << (3 * hclen)) - 1); // (<= 57 bits: the fields of three bits, 19 at most)
        // the precode must be complete (inftrees.c: an incomplete CODES set is an error): the Kraft sum, four fields per look-up --
        // all but one in 200 of the positions that come this far leave here
        static const std::vector<uint16_t> kraft4 = [] {
            std::vector<uint16_t> t(4096);
            for (unsigned x = 0; x < 4096; ++x) { unsigned k = 0; for (unsigned f = 0; f < 4; ++f) { const unsigned l = (x >> (3 * f)) & 7; if (l) k += 128u >> l; } t[x] = (uint16_t)k; }
            return t;
        }();
        const uint16_t *k4 = kraft4.data();
        if (k4[pv & 4095] + k4[(pv >> 12) & 4095] + k4[(pv >> 24) & 4095] + k4[(pv >> 36) & 4095] + k4[(pv >> 48) & 4095] != 128) return false;
        for (unsigned i = 0; i < hclen; ++i) { pl[order[i]] = (uint8_t)(pv & 7); pv >>= 3; }
        // canonical codes of the precode, a 7-bit look-up (code bits arrive LSB first: the table is indexed by the reversed code)
        uint8_t sym_of[128], len_of[128];
        {
            unsigned code = 0;
            for (unsigned l = 1; l <= 7; ++l) {
                for (unsigned s = 0; s < 19; ++s)
                    if (pl[s] == l) {
                        unsigned rev = 0;
                        for (unsigned b = 0; b < l; ++b) rev |= ((code >> b) & 1u) << (l - 1 - b);
                        for (unsigned fill = rev; fill < 128; fill += 1u << l) { sym_of[fill] = (uint8_t)s; len_of[fill] = (uint8_t)l; }
                        ++code;
                    }
                code <<= 1;
            }
        }
        uint8_t lens[286 + 30];
        unsigned n = 0, prev = 0;
        const unsigned total = hlit + hdist;
        while (n < total) {
            const uint64_t w = peek(at);
            const unsigned idx = (unsigned)(w & 127), s = sym_of[idx], l = len_of[idx];
            at += l;
            if (s < 16) { lens[n++] = (uint8_t)s; prev = s; continue; }
            unsigned rep, val;
            if (s == 16) { if (n == 0) return false; rep = 3 + (unsigned)((w >> l) & 3); at += 2; val = prev; }
            else if (s == 17) { rep = 3 + (unsigned)((w >> l) & 7); at += 3; val = 0; prev = 0; }
            else { rep = 11 + (unsigned)((w >> l) & 127); at += 7; val = 0; prev = 0; }
            if (n + rep > total) return false;
            while (rep--) lens[n++] = (uint8_t)val;
        }
        if ((at >> 3) >= size) return false;
        if (lens[256] == 0) return false; // no end-of-block code
        auto complete = [](const uint8_t *l, unsigned cnt, bool dist) {
            unsigned long sum = 0; unsigned mx = 0, used = 0;
            for (unsigned i = 0; i < cnt; ++i) if (l[i]) { sum += 32768ul >> l[i]; mx = std::max<unsigned>(mx, l[i]); ++used; }
            if (sum > 32768ul) return false;              // over-subscribed
            if (sum == 32768ul) return true;
            if (dist && used == 0) return true;           // no distance codes at all (a block of literals)
            return mx == 1;                               // incomplete: only the one-code case is accepted
        };
        return complete(lens, hlit, false) && complete(lens + hlit, hdist, true);
    }

    // the first bit position in [lo, hi) at which plausible_block() holds (NO_START: none).  Eight positions per 64-bit load are tested for
    // the three header bits and the two counts before the function is called: 8 of 9 positions leave here
    uint64_t find_plausible(uint64_t lo, uint64_t hi) const
    {
        uint64_t b = lo;
        while (b < hi) {
            const size_t by = (size_t)(b >> 3);
            if (by + 8 > size) { if (plausible_block(b)) return b; ++b; continue; }
            uint64_t v;
            memcpy(&v, base + by, 8);
            for (unsigned i = (unsigned)(b & 7); i < 8; ++i) {
                const uint64_t x = v >> i;
                if ((x & 7) != 4 || ((x >> 3) & 31) > 29 || ((x >> 8) & 31) > 29) continue;
                const uint64_t cand = (uint64_t)by * 8 + i;
                if (cand >= hi) return NO_START;
                if (plausible_block(cand)) return cand;
            }
            b = ((uint64_t)by + 1) * 8;
        }
        return NO_START;
    }

    // ---- a raw inflate that starts at a bit position with a given 32 KB dictionary ----
    struct Inflater {
        z_stream z;
        bool ok = false;
        Inflater() { memset(&z, 0, sizeof z); }
        ~Inflater() { if (ok) inflateEnd(&z); }
        bool begin(const uint8_t *base, size_t size, uint64_t bit, const uint8_t *dict)
        {
            if (ok) inflateEnd(&z);
            memset(&z, 0, sizeof z);
            ok = inflateInit2(&z, -15) == Z_OK;
            if (!ok) return false;
            const size_t by = (size_t)(bit >> 3);
            const int k = (int)(bit & 7);
            if (k) {
                if (inflatePrime(&z, 8 - k, base[by] >> k) != Z_OK) return false;
                z.next_in = const_cast<Bytef *>(base + by + 1);
                z.avail_in = (uInt)std::min<size_t>(size - by - 1, 1u << 30);
            } else {
                z.next_in = const_cast<Bytef *>(base + by);
                z.avail_in = (uInt)std::min<size_t>(size - by, 1u << 30);
            }
            return !dict || inflateSetDictionary(&z, dict, (uInt)WIN) == Z_OK;
        }
        void refill(const uint8_t *base, size_t size)
        {
            if (z.avail_in == 0) { const size_t at = (size_t)(z.next_in - base); z.avail_in = (uInt)std::min<size_t>(size - at, 1u << 30); }
        }
    };
    static const uint8_t *dict1() { static uint8_t d[WIN]; static bool init = [] { for (size_t k = 0; k < WIN; ++k) d[k] = (uint8_t)(k & 255); return true; }(); (void)init; return d; }
    static const uint8_t *dict2() { static uint8_t d[WIN]; static bool init = [] { for (size_t k = 0; k < WIN; ++k) d[k] = (uint8_t)(128 | (k >> 8)); return true; }(); (void)init; return d; }

    // ---- a raw inflate of 16-bit SYMBOLS that starts at a bit position without knowing the 32 KB in front of it (round 6); with Sym =
    //      uint8_t the same decoder inflates plain bytes (BGZF members in faqcs_cli.cpp: it is about three times as fast as zlib's) ----------
    // A symbol below 256 is a byte of the text; a symbol 0x8000 | k is "the byte at offset k of the unknown window" (k = 32 767: the
    // byte right in front of the piece).  A back-reference that reaches in front of the piece writes such markers, one that copies
    // them copies them on: ONE pass over the compressed bits where the two-dictionary scheme of round 5 made two zlib passes, and the
    // markers are out of band, so the text need not be ASCII.  Own Huffman decoder (zlib has no 16-bit output): an 11-bit first-level
    // table for literals / lengths, 8 bits for distances, second-level tables behind the longer codes; the bit buffer is refilled to
    // >= 56 bits, which a length code + extra + distance code + extra (<= 48 bits) never exhausts.
    struct MarkerInflate {
        enum { LIT_BITS = 11, DIST_BITS = 8 };
        // A table entry (the layout follows the decode loop's critical path: ONE shift of the bit buffer per look-up, the extra bits taken
        // from a saved copy beside it):
        //   bits 0-7   bits to consume at this look-up: the code's bits at this level + the extra bits of a length / distance
        //   bits 8-11  the code's bits at this level alone (the extra bits start behind them)
        //   bit 15 F_LIT a literal | bit 14 F_EXC not a symbol: bit 13 F_EOB end of block, bit 12 F_BAD no code; neither = a second-level table
        //   bits 16-31 the literal / the base of the length or distance / where the second-level table starts
        enum : uint32_t { F_LIT = 1u << 15, F_EXC = 1u << 14, F_EOB = 1u << 13, F_BAD = 1u << 12 };
        uint32_t lit[(1 << LIT_BITS) + 288 * 16], dist[(1 << DIST_BITS) + 30 * 128]; // (second-level tables of 2^(longest code - first level) entries: at most one per symbol)
        uint8_t lit_sub_bits = 0, dist_sub_bits = 0;
        const uint8_t *in = nullptr, *in_end = nullptr;
        uint64_t bitbuf = 0; unsigned bitcnt = 0;
        size_t in_over = 0; // zero bytes "read" past the end of the input (an error as soon as their bits are needed: checked at block ends)
        bool final_block = false, error = false;

        void begin(const uint8_t *base, size_t size, uint64_t bit)
        {
            in = base + (bit >> 3); in_end = base + size; bitbuf = 0; bitcnt = 0; in_over = 0; final_block = false; error = false;
            refill();
            const unsigned k = (unsigned)(bit & 7);
            bitbuf >>= k; bitcnt -= k;
        }
        inline void refill()
        {
            if (in + 8 <= in_end) {
                uint64_t v; memcpy(&v, in, 8);
                bitbuf |= v << bitcnt;
                in += (63 - bitcnt) >> 3;
                bitcnt |= 56;
            } else {
                while (bitcnt <= 56) {
                    if (in < in_end) bitbuf |= (uint64_t)*in++ << bitcnt; else ++in_over;
                    bitcnt += 8;
                }
            }
        }
        // bit position (in the file) of the next bit to be consumed
        uint64_t bit_pos(const uint8_t *base) const { return (uint64_t)(in - base) * 8 + (uint64_t)in_over * 8 - bitcnt; }
        inline uint32_t take(unsigned n) { const uint32_t v = (uint32_t)(bitbuf & ((1ull << n) - 1)); bitbuf >>= n; bitcnt -= n; return v; }

        static unsigned rev(unsigned code, unsigned len) { unsigned r = 0; for (unsigned b = 0; b < len; ++b) r |= ((code >> b) & 1u) << (len - 1 - b); return r; }
        // canonical Huffman code of lens[0 .. n) -> a two-level table; sym_entry(s) gives symbol s's payload and flags (bits 12-31) and the
        // number of its extra bits (bits 0-7); the code lengths are filled in here
        template <class F>
        static bool build(const uint8_t *lens, unsigned n, unsigned first_bits, uint32_t *tab, unsigned tab_cap, uint8_t &sub_bits_out, F &&sym_entry)
        {
            unsigned count[16] = {0}, maxlen = 0;
            for (unsigned i = 0; i < n; ++i) { ++count[lens[i]]; if (lens[i] > maxlen) maxlen = lens[i]; }
            count[0] = 0;
            // over-subscribed sets are an error; incomplete ones decode to F_BAD where no code is assigned
            { long left = 1; for (unsigned l = 1; l <= 15; ++l) { left = left * 2 - (long)count[l]; if (left < 0) return false; } }
            unsigned next[16]; { unsigned code = 0; for (unsigned l = 1; l <= 15; ++l) { code = (code + count[l - 1]) << 1; next[l] = code; } }
            const unsigned P = 1u << first_bits;
            const uint32_t bad = F_EXC | F_BAD | 1u;
            for (unsigned i = 0; i < P; ++i) tab[i] = bad;
            const unsigned sub_bits = maxlen > first_bits ? maxlen - first_bits : 0;
            sub_bits_out = (uint8_t)sub_bits;
            unsigned used = P;
            for (unsigned s = 0; s < n; ++s) {
                const unsigned l = lens[s];
                if (!l) continue;
                const unsigned r = rev(next[l]++, l);
                const uint32_t se = sym_entry(s);
                auto entry = [&](unsigned level_bits) { return (se & 0xfffff000u) | (level_bits << 8) | (level_bits + (se & 0xffu)); };
                if (l <= first_bits) {
                    const uint32_t e = entry(l);
                    for (unsigned i = r; i < P; i += 1u << l) tab[i] = e;
                } else {
                    const unsigned pre = r & (P - 1);
                    if ((tab[pre] & (F_LIT | F_EXC | F_EOB | F_BAD)) != F_EXC) { // (not a pointer yet:) a new second-level table behind this prefix
                        if (used + (1u << sub_bits) > tab_cap) return false;
                        tab[pre] = ((uint32_t)used << 16) | F_EXC | first_bits;
                        for (unsigned i = 0; i < (1u << sub_bits); ++i) tab[used + i] = bad;
                        used += 1u << sub_bits;
                    }
                    const unsigned start = tab[pre] >> 16, rest = l - first_bits;
                    const uint32_t e = entry(rest);
                    for (unsigned i = r >> first_bits; i < (1u << sub_bits); i += 1u << rest) tab[start + i] = e;
                }
            }
            return true;
        }
        bool build_tables(const uint8_t *ll, unsigned nl, const uint8_t *dl, unsigned nd)
        {
            static const uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
            static const uint8_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
            static const uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
            static const uint8_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
            if (!build(ll, nl, LIT_BITS, lit, sizeof lit / 4, lit_sub_bits, [&](unsigned s) -> uint32_t {
                    if (s < 256) return (s << 16) | F_LIT;
                    if (s == 256) return F_EXC | F_EOB;
                    if (s > 285) return F_EXC | F_BAD;
                    return ((uint32_t)lbase[s - 257] << 16) | lext[s - 257];
                })) return false;
            return build(dl, nd, DIST_BITS, dist, sizeof dist / 4, dist_sub_bits, [&](unsigned s) -> uint32_t {
                if (s > 29) return F_EXC | F_BAD;
                return ((uint32_t)dbase[s] << 16) | dext[s];
            });
        }
        // the header of the next block; false: invalid.  stored blocks are copied here (their length is returned through n_stored)
        int block_type = -1; uint32_t stored_left = 0;
        bool read_header()
        {
            refill();
            final_block = take(1) != 0;
            block_type = (int)take(2);
            if (block_type == 0) {
                take(bitcnt & 7); // to the byte boundary
                refill();
                const uint32_t len = take(16), nlen = take(16);
                if ((len ^ nlen) != 0xffffu) return false;
                stored_left = len;
                return true;
            }
            uint8_t ll[288 + 32], dl[32];
            if (block_type == 1) {
                for (unsigned i = 0; i < 144; ++i) ll[i] = 8;
                for (unsigned i = 144; i < 256; ++i) ll[i] = 9;
                for (unsigned i = 256; i < 280; ++i) ll[i] = 7;
                for (unsigned i = 280; i < 288; ++i) ll[i] = 8;
                for (unsigned i = 0; i < 30; ++i) dl[i] = 5;
                return build_tables(ll, 288, dl, 30);
            }
            if (block_type != 2) return false;
            const unsigned hlit = take(5) + 257, hdist = take(5) + 1, hclen = take(4) + 4;
            if (hlit > 286 || hdist > 30) return false;
            static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
            uint8_t pl[19]; memset(pl, 0, sizeof pl);
            refill();
            for (unsigned i = 0; i < hclen; ++i) { if (bitcnt < 3) refill(); pl[order[i]] = (uint8_t)take(3); }
            uint32_t pre[128 + 64]; uint8_t pre_sub = 0;
            if (!build(pl, 19, 7, pre, sizeof pre / 4, pre_sub, [&](unsigned s) -> uint32_t { return (s << 16) | F_LIT; })) return false;
            uint8_t lens[286 + 30 + 140];
            unsigned n = 0; const unsigned total = hlit + hdist;
            while (n < total) {
                refill();
                if (in_over > 8) return false;
                const uint32_t e = pre[bitbuf & 127];
                if (!(e & F_LIT)) return false;
                take(e & 63u);
                const unsigned sym = e >> 16;
                if (sym < 16) { lens[n++] = (uint8_t)sym; continue; }
                unsigned rep, val = 0;
                if (sym == 16) { if (!n) return false; rep = 3 + take(2); val = lens[n - 1]; }
                else if (sym == 17) rep = 3 + take(3);
                else rep = 11 + take(7);
                if (n + rep > total) return false;
                while (rep--) lens[n++] = (uint8_t)val;
            }
            if (lens[256] == 0) return false; // no end-of-block code
            return build_tables(lens, hlit, lens + hlit, hdist);
        }
        // One whole block into out[0 ..) from position pos on (out grows through `grow`, which returns the new base); positions in front
        // of 0 are the unknown window (markers).  Returns false on invalid data.
        // The loop carries the entry of the next code (looked up before the copy of a match, so that the load overlaps it) and keeps the
        // bit reader in locals; per symbol the bit buffer is shifted once, by the entry's low byte.
        // Sym = uint8_t: plain bytes for data whose start is the start of the stream (a BGZF member: nothing lies in front of position 0, a
        // distance that reaches there is invalid).  grow() may return nullptr: the block fails.
        template <class Sym, class Grow>
        bool decode_block(Sym *&out, size_t &pos_io, size_t &cap, Grow &&grow)
        {
            constexpr bool MARKERS = sizeof(Sym) == 2;
            if (!read_header()) return false;
            size_t pos = pos_io;
            if (block_type == 0) {
                // the bytes of a stored block: what is left in the bit buffer first (whole bytes), then straight from the input
                while (stored_left) {
                    if (pos + 1 > cap) { out = grow(pos + stored_left + 1); if (!out) return false; }
                    if (bitcnt >= 8) {
                        if (in_over * 8 + 8 > bitcnt) return false; // (the input ended: this byte is one of the zeros behind it)
                        out[pos++] = (Sym)take(8); --stored_left; continue;
                    }
                    bitbuf = 0; bitcnt = 0; // (byte aligned and drained: what the last refill loaded beyond bitcnt is read again from `in`)
                    if (in >= in_end) return false;
                    out[pos++] = *in++; --stored_left;
                }
                if (bitcnt < 8) { bitbuf = 0; bitcnt = 0; }
                pos_io = pos;
                return in_over * 8 <= bitcnt; // (the bits of bytes that are not there have not been consumed)
            }
            const uint32_t lmask = (1u << LIT_BITS) - 1, dmask = (1u << DIST_BITS) - 1;
            const uint32_t lsub = (1u << lit_sub_bits) - 1, dsub = (1u << dist_sub_bits) - 1;
            // the bit reader in locals (written back where the block ends)
            uint64_t bb = bitbuf; unsigned bc = bitcnt; const uint8_t *ip = in; const uint8_t *const ie = in_end; size_t over = in_over;
            Sym *o = out;
#define MI_REFILL()                                                                                                                       \
            do {                                                                                                                          \
                if (__builtin_expect(ip + 8 <= ie, 1)) { uint64_t v_; memcpy(&v_, ip, 8); bb |= v_ << bc; ip += (63 - bc) >> 3; bc |= 56; } \
                else {                                                                                                                    \
                    while (bc <= 56) { if (ip < ie) bb |= (uint64_t)*ip++ << bc; else ++over; bc += 8; }                                  \
                    if (over > 8) return false; /* (the input ends inside the block: the zeros behind it must not be decoded on and on) */ \
                }                                                                                                                         \
            } while (0)
#define MI_CONSUME(e_) do { bb >>= ((e_) & 63u); bc -= ((e_) & 63u); } while (0)
            MI_REFILL();
            uint32_t e = lit[bb & lmask];
            for (;;) {
                // here: >= 56 bits in the buffer (or the input has ended), e = the entry of the code in front
                if (__builtin_expect(pos + 320 > cap, 0)) { out = grow(pos + 320); o = out; if (!o) return false; }
                if (e & F_LIT) {
                    // up to three literals per refill (3 x 11 bits of first-level codes); a refill leaves the bits in front untouched, so the
                    // entry that ends the run is still the right one behind it
                    MI_CONSUME(e); o[pos++] = (Sym)(e >> 16);
                    e = lit[bb & lmask];
                    if (e & F_LIT) {
                        MI_CONSUME(e); o[pos++] = (Sym)(e >> 16);
                        e = lit[bb & lmask];
                        if (e & F_LIT) {
                            MI_CONSUME(e); o[pos++] = (Sym)(e >> 16);
                            MI_REFILL();
                            e = lit[bb & lmask];
                            continue;
                        }
                    }
                    MI_REFILL();
                }
                if (__builtin_expect(e & F_EXC, 0)) {
                    if (e & (F_EOB | F_BAD)) { if (e & F_BAD) return false; MI_CONSUME(e); break; }
                    MI_CONSUME(e); // (the first level's bits)
                    e = lit[(e >> 16) + (bb & lsub)];
                    if (e & F_LIT) { MI_CONSUME(e); o[pos++] = (Sym)(e >> 16); MI_REFILL(); e = lit[bb & lmask]; continue; }
                    if (e & F_EXC) { if (!(e & F_EOB)) return false; MI_CONSUME(e); break; }
                }
                // a length (<= 20 bits with its extra bits, second level included) and a distance (<= 28): <= 48 of the 56
                uint64_t saved = bb;
                MI_CONSUME(e);
                const unsigned len = (e >> 16) + (unsigned)(((uint32_t)saved & ((1u << (e & 63u)) - 1u)) >> ((e >> 8) & 15u));
                uint32_t d = dist[bb & dmask];
                if (__builtin_expect(d & F_EXC, 0)) {
                    if (d & F_BAD) return false;
                    MI_CONSUME(d);
                    d = dist[(d >> 16) + (bb & dsub)];
                    if (d & F_EXC) return false;
                }
                saved = bb;
                MI_CONSUME(d);
                const size_t dd = (size_t)(d >> 16) + (size_t)(((uint32_t)saved & ((1u << (d & 63u)) - 1u)) >> ((d >> 8) & 15u));
                MI_REFILL();
                e = lit[bb & lmask]; // (the next code's entry is on its way while the match is copied)
                Sym *dp = o + pos;
                constexpr size_t V = 16 / sizeof(Sym); // symbols in a 16-byte step
                if (__builtin_expect(dd > pos, 0)) { // the copy starts in front of position 0
                    if (!MARKERS || dd > pos + WIN) return false; // further back than there is anything (than the window in front of the piece)
                    const size_t before = dd - pos; // source positions -before .. -1: markers for that part
                    size_t i = 0;
                    const size_t nmark = before < len ? before : len;
                    for (; i < nmark; ++i) dp[i] = (Sym)(0x8000u | (uint32_t)(WIN - before + i));
                    for (; i < len; ++i) dp[i] = dp[i - dd];
                } else if (dd >= V) { // the common case: sixteen bytes a step, past the end of the match if need be (there is room for 320 symbols)
                    const Sym *sp = dp - dd;
                    memcpy(dp, sp, 16);
                    if (__builtin_expect(len > V, 0)) for (size_t k = V; k < len; k += V) memcpy(dp + k, sp + k, 16);
                } else if (dd == 1) { // a run of one symbol
                    const uint64_t v = (uint64_t)dp[-1] * (MARKERS ? 0x0001000100010001ull : 0x0101010101010101ull);
                    for (size_t k = 0; k < len; k += 8 / sizeof(Sym)) memcpy(dp + k, &v, 8);
                } else if (!MARKERS && dd >= 8) {
                    const Sym *sp = dp - dd;
                    for (size_t k = 0; k < len; k += 8) memcpy(dp + k, sp + k, 8);
                } else for (size_t i = 0; i < len; ++i) dp[i] = dp[i - dd];
                pos += len;
            }
#undef MI_REFILL
#undef MI_CONSUME
            bitbuf = bb; bitcnt = bc; in = ip; in_over = over;
            pos_io = pos;
            return in_over * 8 <= bitcnt; // (the bits of bytes that are not there have not been consumed)
        }
    };


    // ---- file ----
    static bool eligible(const std::string &path, size_t min_size = 8u << 20)
    {
        struct stat st;
        if (stat(path.c_str(), &st) != 0 || !S_ISREG(st.st_mode) || (size_t)st.st_size < min_size) return false;
        uint8_t h[4] = {0, 0, 0, 0};
        FILE *f = fopen(path.c_str(), "rb");
        if (!f) return false;
        const size_t n = fread(h, 1, 4, f);
        fclose(f);
        return n == 4 && h[0] == 31 && h[1] == 139 && h[2] == 8;
    }
    // offset of the deflate data behind a gzip member header at `o` (0: not a header / truncated)
    size_t header_end(size_t o) const
    {
        if (o + 10 > size || base[o] != 31 || base[o + 1] != 139 || base[o + 2] != 8) return 0;
        const unsigned flg = base[o + 3];
        size_t p = o + 10;
        if (flg & 4) { if (p + 2 > size) return 0; p += 2 + (base[p] | ((size_t)base[p + 1] << 8)); }
        if (flg & 8) { while (p < size && base[p]) ++p; ++p; }
        if (flg & 16) { while (p < size && base[p]) ++p; ++p; }
        if (flg & 2) p += 2;
        return p < size ? p : 0;
    }
    bool open(const std::string &path, int n_threads, size_t piece = 0)
    {
        fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) != 0) { ::close(fd); fd = -1; return false; }
        size = (size_t)st.st_size;
        void *mp = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (mp == MAP_FAILED) { ::close(fd); fd = -1; return false; }
        base = (const uint8_t *)mp;
        n_workers = std::max(1, n_threads);
        piece_arg = piece;
        if (!arm(0)) { close(); return false; }
        return true;
    }
    // the member whose header stands at byte `member_off` becomes the member in work: pieces, workers, chain thread; its first piece is
    // inflated here (its window is known: empty) and decides whether the member qualifies (ASCII).  false: nothing is running.
    bool arm(size_t member_off)
    {
        data_begin = header_end(member_off);
        if (!data_begin) return false;
        size_t piece_max = 2u << 20; // (4 MB until profiles/r6n/: 2.3 x that of text and 4.6 x of symbols per piece in flight)
        if (const char *e = getenv("FAQCS_MI_PARGZ_PIECE")) { const long long v = atoll(e); if (v >= (64 << 10) && v <= (64 << 20)) piece_max = (size_t)v; }
        piece_bytes = piece_arg ? piece_arg : std::max<size_t>(std::min<size_t>(512u << 10, piece_max), std::min<size_t>(piece_max, (size - data_begin) / (size_t)(8 * n_workers) + 1));
        n_pieces = (size - data_begin + piece_bytes - 1) / piece_bytes;
        armed_pieces = n_pieces;
        pieces = std::vector<Piece>(n_pieces);
        window_pieces = (size_t)(2 * n_workers + 2); // (3 n + 2 until the sweep of profiles/r6n/: the same speed from 12 ... 26 pieces in flight once a piece in front of an unclaimed one ends there; fewer = less memory to fault in)
        if (const char *e = getenv("FAQCS_MI_PARGZ_WINDOW")) { const long v = atol(e); if (v >= n_workers + 2 && v <= 1024) window_pieces = (size_t)v; } // (pieces in flight: memory against slack)
        closing = false; stop_claims = false; cancel_from.store(~(size_t)0);
        next_claim = 1; consumed = 0; cur = 0; hold = 0; release_lo = 0;
        member_done = false; total_out = 0;
        crcs.clear(); crc_index.clear(); crc_wait.clear();
        pieces[0].start_bit.store((uint64_t)data_begin * 8);
        pieces[0].known_window = true;
        pieces[0].state.store(1);
        for (int i = 0; i < n_workers; ++i) workers.emplace_back([this] { work(); }); // (they look for the starts piece 0 will stop at)
        chain_th = std::thread([this] { chain(); });
        inflate_piece(0);
        cv_done.notify_all();
        bool ok = pieces[0].state.load() == 2;
        if (two_pass) for (size_t i = 0; ok && i < pieces[0].n_out; ++i) if (pieces[0].out[i] >= 128) ok = false; // (two-pass scheme: not ASCII, the markers would be ambiguous)
        // text that compresses far better than FASTQ does (a first piece above 24 : 1) stays with the serial reader, whose memory does not depend on it
        if (ok && pieces[0].n_out / 24 > (size_t)((pieces[0].end_bit - pieces[0].start_bit.load()) >> 3) + 1) ok = false;
        if (!ok) disarm();
        return ok;
    }
    // workers and chain thread stopped, the pieces' buffers back in the pool
    void disarm()
    {
        { std::lock_guard<std::mutex> l(m); closing = true; }
        cv_work.notify_all(); cv_done.notify_all();
        for (auto &t : workers) if (t.joinable()) t.join();
        workers.clear();
        if (chain_th.joinable()) chain_th.join();
        for (auto &q : pieces) { give_buf(q.out); give_buf(q.mark); }
        pieces.clear();
        n_pieces = 0;
        { std::lock_guard<std::mutex> l(m); closing = false; }
    }
    void close()
    {
        disarm();
        if (tracing) for (const Ev &e : trace_log) fprintf(stderr, "[pargz %8.4f] piece %4zu %s %zu\n", e.t, e.piece, e.what, e.arg);
        if (tail_init && !tail_done) { inflateEnd(&tz); tail_done = true; }
        if (base) munmap(const_cast<uint8_t *>(base), size);
        base = nullptr;
        if (fd >= 0) ::close(fd);
        fd = -1;
        for (auto &b : pool) free(b.p);
        pool.clear();
    }
    uint64_t range_begin_bit(size_t i) const { return (uint64_t)(data_begin + i * piece_bytes) * 8; }

    // ---- workers ----
    void work()
    {
        FaqcsThreadCpu cpu_note("gzip inflate worker");
        for (;;) {
            size_t i = 0;
            bool crc_job = false;
            {
                std::unique_lock<std::mutex> l(m);
                for (;;) {
                    if (closing) return;
                    // a CRC somebody asked for goes first (the consumer waits for them at the end)
                    bool found = false;
                    if (!crc_wait.empty()) {
                        i = crc_wait.back(); crc_wait.pop_back();
                        pieces[i].crc_state.store(3); // taken
                        found = crc_job = true;
                    }
                    if (found) break;
                    if (next_claim < n_pieces && next_claim < consumed + window_pieces && !failed && !stop_claims) { i = next_claim++; pieces[i].state.store(1); break; }
                    cv_work.wait(l);
                }
            }
            if (crc_job) {
                Piece &p = pieces[i];
                trace(i, "narrowing and crc taken");
                bool invalid = false;
                if (!two_pass) { // the symbols become the text 64 KB at a time, and the CRC takes each stretch while it is in the cache
                    if (!p.out.p) p.out = take_buf();
                    p.out.resize(p.n_out + 1);
                    const uint16_t *sym = reinterpret_cast<const uint16_t *>(p.mark.data());
                    uint32_t c = 0;
                    bool any = false;
                    for (size_t lo = 0; lo < p.n_out; lo += 65536) {
                        const size_t n = std::min<size_t>(65536, p.n_out - lo);
                        any |= faqcs_narrow(sym + lo, p.out.data() + lo, n, p.win.data());
                        c = faqcs_crc32(c, p.out.data() + lo, n);
                    }
                    p.crc = c;
                    invalid = any && p.known_window;
                    give_buf(p.mark); // (still warm: the next piece to be inflated takes it)
                } else {
                    if (!p.known_window) patch(p, 0, p.n_out);
                    p.crc = faqcs_crc32(0, p.out.data(), p.n_out);
                }
                { std::lock_guard<std::mutex> l(m); if (invalid) p.state.store(3); p.crc_state.store(2); }
                trace(i, "patched, crc done");
                cv_done.notify_all();
                continue;
            }
            // find where this piece can start
            Piece &p = pieces[i];
            trace(i, "claimed");
            const uint64_t lo = range_begin_bit(i), hi = std::min<uint64_t>(range_begin_bit(i + 1), (uint64_t)size * 8);
            uint64_t s = p.start_bit.load(); // (not 0: the piece in front ended inside this range before anybody had claimed it -- that boundary is the start)
            if (s == 0) {
                s = NO_START;
                for (uint64_t b = lo; b < hi;) {
                    const uint64_t c = find_plausible(b, hi);
                    if (c == NO_START) break;
                    if (trial(c)) { s = c; break; }
                    b = c + 1;
                }
            }
            { std::lock_guard<std::mutex> l(m); if (p.start_bit.load() != 0) s = p.start_bit.load(); p.start_bit.store(s); if (s == NO_START) p.state.store(3); }
            cv_done.notify_all(); // (a predecessor may be waiting to learn where this piece starts)
            trace(i, "start found at bit", (size_t)(s - lo));
            if (s == NO_START) continue;
            inflate_piece(i);
            trace(i, "inflated, state", (size_t)p.state.load());
            cv_done.notify_all();
        }
    }
    // a candidate must inflate 64 KB (or to a block end) without an error
    bool trial(uint64_t bit) const
    {
        Inflater f;
        if (!f.begin(base, size, bit, dict1())) return false;
        std::vector<uint8_t> tmp(1u << 16);
        f.z.next_out = tmp.data(); f.z.avail_out = (uInt)tmp.size();
        const int rc = inflate(&f.z, Z_BLOCK);
        return rc == Z_OK || rc == Z_STREAM_END || (rc == Z_BUF_ERROR && f.z.avail_out == 0);
    }
    // Inflates piece i from its start to the first block boundary that is the start of a later piece (or to the end of the member).
    // `here` = the bit position of a block boundary: later pieces whose start lies at or before it are looked at; true: one starts exactly here
    bool stops_at(Piece &p, size_t &j, uint64_t here, bool &abort)
    {
        abort = false;
        while (j < n_pieces && range_begin_bit(j) <= here) {
            uint64_t s;
            { // wait for piece j's worker to publish its start (pieces past the claim window have none yet: go on through them)
                std::unique_lock<std::mutex> l(m);
                while (!closing && pieces[j].start_bit.load() == 0 && pieces[j].state.load() != 0) cv_done.wait(l);
                if (closing) { p.state.store(3); abort = true; return false; }
                if (pieces[j].state.load() == 0) {
                    // Nobody has claimed piece j yet (the pieces in flight are as many as may be: the consumer is the slower side).  This
                    // block boundary IS where the text of piece j's range begins: it becomes j's start -- whoever claims j need not look
                    // for one -- and this piece ends here.  (Round 6 at first ran on THROUGH such a piece: behind a slow consumer one
                    // worker after the other did, and the whole file was inflated by single threads into buffers of gigabytes --
                    // profiles/r6n/e2e_gz_knobs_before_the_fix.txt: 2.5 instead of 11 M reads/s whenever the window was a little smaller.)
                    if (pieces[j].start_bit.load() == 0 && here < range_begin_bit(j + 1)) { pieces[j].start_bit.store(here); return true; }
                    if (pieces[j].start_bit.load() == 0) { if (next_claim == j) ++next_claim; pieces[j].state.store(3); pieces[j].start_bit.store(NO_START); } // (no block starts inside it)
                }
                s = pieces[j].start_bit.load();
            }
            if (s == here) return true;
            if (s == NO_START || s < here) { ++j; continue; } // no start there, or one this chain never arrived at: dropped
            break;                                             // its start lies ahead
        }
        return false;
    }
    void inflate_piece(size_t i)
    {
        Piece &p = pieces[i];
        const uint64_t start = p.start_bit.load();
        auto fail = [&] { std::lock_guard<std::mutex> l(m); p.state.store(3); };
        p.n_out = 0; p.final_seen = false;
        size_t j = i + 1; // the first later piece whose start this one has not passed yet
        if (!two_pass) { // ONE pass that writes 16-bit symbols (round 6; the first piece too: this decoder is faster than zlib's)
            std::unique_ptr<MarkerInflate> mi(new MarkerInflate);
            mi->begin(base, size, start);
            if (!p.mark.p) p.mark = take_buf();
            p.mark.resize(std::max<size_t>(p.mark.size(), (piece_bytes * 4 + (1u << 20)) * 2));
            uint16_t *sym = reinterpret_cast<uint16_t *>(p.mark.data());
            size_t cap = p.mark.size() / 2, pos = 0;
            // (a piece that inflates to more than 128 times its share of the file is refused -- the input ends with an error there -- rather than
            // allowed to take gigabytes: deflate can reach 1 032 : 1, and as many pieces as there are workers grow at once; arm() keeps
            // files whose FIRST piece already inflates more than 24 : 1 away from this reader altogether.  FASTQ: 3 - 6 : 1)
            static const size_t sym_limit_env = [] { const char *e = getenv("FAQCS_MI_PARGZ_SYM_LIMIT"); return e ? (size_t)atoll(e) : (size_t)0; }(); // (tests)
            const size_t sym_limit = sym_limit_env ? sym_limit_env : std::max<size_t>(128 * piece_bytes, 64u << 20);
            auto grow = [&](size_t need) -> uint16_t * {
                if (need > sym_limit) return nullptr;
                size_t nc = cap + cap / 2; if (nc < need) nc = need;
                p.mark.resize(nc * 2); cap = p.mark.size() / 2;
                return reinterpret_cast<uint16_t *>(p.mark.data());
            };
            for (;;) {
                if (i >= cancel_from.load(std::memory_order_relaxed)) { fail(); return; } // (a start found inside a FOLLOWING member would run on to its end)
                if (!mi->decode_block(sym, pos, cap, grow)) { fail(); return; }
                p.n_out = pos;
                if (mi->final_block) {
                    const uint64_t end = mi->bit_pos(base);
                    p.final_seen = true; p.trailer_at = (size_t)((end + 7) >> 3); p.next_piece = n_pieces;
                    p.end_bit = (uint64_t)p.trailer_at * 8;
                    break;
                }
                const uint64_t here = mi->bit_pos(base);
                bool abort = false;
                if (stops_at(p, j, here, abort)) { p.end_bit = here; p.next_piece = j; break; }
                if (abort) return;
            }
            std::lock_guard<std::mutex> l(m);
            p.state.store(2);
            return;
        }
        Inflater f;
        if (!f.begin(base, size, start, p.known_window ? nullptr : dict1())) { fail(); return; }
        if (!p.out.p) p.out = take_buf();
        p.out.resize(std::max<size_t>(p.out.size(), piece_bytes * 4 + (1u << 20)));
        for (;;) {
            if (i >= cancel_from.load(std::memory_order_relaxed)) { fail(); return; } // (a start found inside a FOLLOWING member would run on to its end)
            if (p.n_out + (1u << 16) > p.out.size()) p.out.resize(p.out.size() + p.out.size() / 2);
            f.z.next_out = p.out.data() + p.n_out;
            f.z.avail_out = (uInt)std::min<size_t>(p.out.size() - p.n_out, 1u << 30);
            f.refill(base, size);
            const size_t before = f.z.avail_out;
            const int rc = inflate(&f.z, Z_BLOCK);
            p.n_out += before - f.z.avail_out;
            if (rc == Z_STREAM_END) {
                p.final_seen = true; p.trailer_at = (size_t)(f.z.next_in - base); p.next_piece = n_pieces;
                p.end_bit = (uint64_t)p.trailer_at * 8;
                break;
            }
            if (rc != Z_OK) { fail(); return; } // a data error, or Z_BUF_ERROR: there is always room for output, so the file ends inside the member
            if (!(f.z.data_type & 128)) continue; // not at a block boundary (output space ran out)
            const uint64_t here = (uint64_t)(f.z.next_in - base) * 8 - (uint64_t)(f.z.data_type & 63);
            bool abort = false;
            if (stops_at(p, j, here, abort)) { p.end_bit = here; p.next_piece = j; break; }
            if (abort) return;
        }
        if (!p.known_window) { // (FAQCS_MI_PARGZ_TWO_PASS=1: round 5's scheme) the same range again with the second dictionary: n_out bytes
            Inflater g;
            if (!g.begin(base, size, start, dict2())) { fail(); return; }
            if (!p.mark.p) p.mark = take_buf();
            p.mark.resize(p.n_out + 1);
            size_t got = 0;
            while (got < p.n_out) {
                g.z.next_out = p.mark.data() + got;
                g.z.avail_out = (uInt)std::min<size_t>(p.n_out - got, 1u << 30);
                g.refill(base, size);
                const size_t before = g.z.avail_out;
                const int rc = inflate(&g.z, Z_NO_FLUSH);
                got += before - g.z.avail_out;
                if (rc == Z_STREAM_END) break;
                if (rc != Z_OK && rc != Z_BUF_ERROR) { fail(); return; }
                if (rc == Z_BUF_ERROR && before == g.z.avail_out && g.z.avail_in == 0) { fail(); return; }
            }
            if (got != p.n_out) { fail(); return; }
        }
        std::lock_guard<std::mutex> l(m);
        p.state.store(2);
    }

    // (FAQCS_MI_PARGZ_TWO_PASS=1 only) out[k] of the bytes [lo, hi) that came from the window in front of the piece: the marker (mark[k] & 128) carries the window offset
    void patch(Piece &p, size_t lo, size_t hi)
    {
        const uint8_t *w = p.win.data();
        uint8_t *o = p.out.data();
        const uint8_t *mk = p.mark.data();
        size_t k = lo;
        for (; k + 8 <= hi; k += 8) {
            uint64_t w8;
            memcpy(&w8, mk + k, 8);
            if (w8 & 0x8080808080808080ull)
                for (size_t t = 0; t < 8; ++t) if (mk[k + t] & 128) o[k + t] = w[(size_t)o[k + t] | ((size_t)(mk[k + t] & 127) << 8)];
        }
        for (; k < hi; ++k) if (mk[k] & 128) o[k] = w[(size_t)o[k] | ((size_t)(mk[k] & 127) << 8)];
    }
    // The chain thread: pieces in file order; the window behind a piece = its last 32 KB resolved against the window in front of it
    // (a sequential step of 32 KB per piece); the piece's other bytes are left to a worker (patch + CRC task).
    void chain()
    {
        FaqcsThreadCpu cpu_note("gzip window chain");
        std::vector<uint8_t> W(WIN, 0), tail(WIN);
        size_t c = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> l(m);
                while (!closing && pieces[c].state.load() != 2 && pieces[c].state.load() != 3) cv_done.wait(l);
                if (closing || pieces[c].state.load() == 3) return; // (a failed piece on the chain: the consumer reports it when it gets there)
            }
            Piece &p = pieces[c];
            trace(c, "chain arrives, next piece", p.next_piece);
            if (!p.known_window || !two_pass) p.win = W; // (the first piece: zeros, never looked at)
            // the window behind the piece
            const size_t t = std::min(p.n_out, WIN), from = p.n_out - t;
            if (!two_pass) {
                const uint16_t *sym = reinterpret_cast<const uint16_t *>(p.mark.data());
                for (size_t k = 0; k < t; ++k) { const uint16_t v = sym[from + k]; tail[k] = v < 256 ? (uint8_t)v : W[v & 0x7fffu]; }
            } else
            for (size_t k = 0; k < t; ++k) {
                const uint8_t b = p.out[from + k];
                tail[k] = (!p.known_window && (p.mark[from + k] & 128)) ? W[(size_t)b | ((size_t)(p.mark[from + k] & 127) << 8)] : b;
            }
            if (t == WIN) W.swap(tail);
            else if (t) { memmove(W.data(), W.data() + t, WIN - t); memcpy(W.data() + WIN - t, tail.data(), t); }
            { std::lock_guard<std::mutex> l(m); p.crc_state.store(1); crc_wait.push_back(c); }
            cv_work.notify_all();
            if (p.final_seen) { // (what lies behind belongs to another member, or to nobody)
                std::lock_guard<std::mutex> l(m);
                stop_claims = true;
                cancel_from.store((p.trailer_at - data_begin) / piece_bytes + 1);
            }
            if (p.final_seen || p.next_piece >= n_pieces) return;
            c = p.next_piece;
        }
    }
    std::thread chain_th;

    // ---- consumer ----
    // the next run of inflated bytes in file order (valid until the next call); 0 = end of data, or `failed`
    size_t next(const char *&data)
    {
        if (failed) return 0;
        if (!member_done) {
            Piece *p;
            {
                std::unique_lock<std::mutex> l(m);
                // the piece the consumer held is free now; buffers of the pieces behind it go back (a handed-out piece once its CRC is
                // done, a dropped one once its worker has left it)
                if (cur > consumed) { consumed = cur; cv_work.notify_all(); }
                while (release_lo < consumed && release_lo < n_pieces) {
                    Piece &q = pieces[release_lo];
                    const int st = q.state.load(), cs = q.crc_state.load();
                    if (st == 1 || cs == 1 || cs == 3) break; // still in use
                    give_buf(q.out); give_buf(q.mark); std::vector<uint8_t>().swap(q.win);
                    ++release_lo;
                }
                for (;;) {
                    const int st = pieces[cur].state.load();
                    if (st == 3) { failed = true; return 0; } // a piece ON the chain failed: the stream is corrupt here
                    if (st == 2 && pieces[cur].crc_state.load() == 2) break; // inflated, patched, CRC taken
                    if (closing) return 0;
                    cv_done.wait(l);
                }
                p = &pieces[cur];
            }
            trace(cur, "handed out, bytes", p->n_out);
            total_out += p->n_out;
            crcs.emplace_back(0u, p->n_out);
            crc_index.push_back(cur);
            data = reinterpret_cast<const char *>(p->out.data());
            const size_t n = p->n_out;
            if (p->final_seen) { member_done = true; tail_from = p->trailer_at; if (!check_member()) { failed = true; return n ? n : 0; } hold = cur; cur = n_pieces; }
            else { hold = cur; cur = p->next_piece; if (cur >= n_pieces) { failed = true; return n; } } // (the chain ran out of file without a final block)
            if (n) return n;
            return next(data);
        }
        if (n_pieces) { // the member that just ended: its workers stop, its buffers go back (the bytes handed out last are not needed any more)
            disarm();
            // another member behind it, and enough of the file left for the speculation to pay: the same again from its header
            if (tail_from + 2 <= size && base[tail_from] == 31 && base[tail_from + 1] == 139 && size - tail_from >= rearm_min && arm(tail_from)) return next(data);
        }
        return next_tail(data);
    }
    std::vector<size_t> crc_index;
    size_t hold = 0, release_lo = 0;
    // the member's trailer against the pieces' CRCs and the length
    bool check_member()
    {
        { // every CRC asked for so far
            std::unique_lock<std::mutex> l(m);
            for (;;) {
                bool all = true;
                for (size_t k : crc_index) if (pieces[k].crc_state.load() != 2) { all = false; break; }
                if (all || closing) break;
                cv_done.wait(l);
            }
            if (closing) return false;
        }
        if (tail_from + 8 > size) return false;
        uint32_t crc = 0;
        bool first = true;
        for (size_t k = 0; k < crc_index.size(); ++k) {
            const Piece &p = pieces[crc_index[k]];
            if (first) { crc = p.crc; first = false; }
            else crc = (uint32_t)crc32_combine(crc, p.crc, (z_off_t)crcs[k].second);
        }
        uint32_t want_crc, want_len;
        memcpy(&want_crc, base + tail_from, 4); memcpy(&want_len, base + tail_from + 4, 4);
        tail_from += 8;
        return crc == want_crc && (uint32_t)total_out == want_len;
    }
    // what follows the first member: further gzip members through zlib's gzip decoder (gzread reads concatenated members); bytes that
    // do not start with the gzip magic are trailing garbage and end the data, as in zlib
    size_t next_tail(const char *&data)
    {
        if (tail_done) return 0;
        if (pending_rearm) { // (set at the end of a member read here: the next one is long enough for the parallel reader)
            pending_rearm = false;
            tail_from = rearm_at;
            if (arm(tail_from)) return next(data);
            rearm_min = ~(size_t)0; // it does not qualify (not ASCII): serially from here on
        }
        if (!tail_init) {
            if (tail_from >= size || size - tail_from < 2 || base[tail_from] != 31 || base[tail_from + 1] != 139) { tail_done = true; return 0; }
            memset(&tz, 0, sizeof tz);
            if (inflateInit2(&tz, 15 + 16) != Z_OK) { failed = true; tail_done = true; return 0; }
            tz.next_in = const_cast<Bytef *>(base + tail_from);
            tz.avail_in = (uInt)std::min<size_t>(size - tail_from, 1u << 30);
            tail_out.resize(4u << 20);
            tail_init = true; tail_mid = true;
        }
        for (;;) {
            if (tz.avail_in == 0) {
                const size_t at = (size_t)(tz.next_in - base);
                if (at >= size) { tail_done = true; inflateEnd(&tz); if (tail_mid) failed = true; return 0; }
                tz.avail_in = (uInt)std::min<size_t>(size - at, 1u << 30);
            }
            tz.next_out = reinterpret_cast<Bytef *>(tail_out.data());
            tz.avail_out = (uInt)tail_out.size();
            const int rc = inflate(&tz, Z_NO_FLUSH);
            const size_t got = tail_out.size() - tz.avail_out;
            if (rc == Z_STREAM_END) {
                tail_mid = false;
                const size_t at = (size_t)(tz.next_in - base);
                if (at + 2 <= size && base[at] == 31 && base[at + 1] == 139) {
                    if (size - at >= rearm_min && !pending_rearm) { pending_rearm = true; rearm_at = at; inflateEnd(&tz); tail_init = false; }
                    else { inflateReset(&tz); tz.next_in = const_cast<Bytef *>(base + at); tz.avail_in = (uInt)std::min<size_t>(size - at, 1u << 30); tail_mid = true; }
                }
                else { tail_done = true; inflateEnd(&tz); }
            } else if (rc != Z_OK && rc != Z_BUF_ERROR) { failed = true; tail_done = true; inflateEnd(&tz); return 0; }
            if (got && !pending_rearm) { data = tail_out.data(); return got; }
            if (pending_rearm) { // a long member follows the short one(s) read here
                if (got) { data = tail_out.data(); return got; } // (its bytes first; the next call finds pending_rearm set and tail_init cleared)
            }
            if (tail_done) return 0;
            if (pending_rearm) break;
        }
        return next_tail(data);
    }
    bool pending_rearm = false;
    size_t rearm_at = 0;
};
