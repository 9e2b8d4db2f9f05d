// skm_model.cpp -- host model of skm_extract / skm_combine's arithmetic (faqcs_amd/csrc/faqcs_kmer_skm_kernel.hip), built on the SAME
// header the kernels use (faqcs_skm.h).  Test infrastructure: compiled and run by tests/test_skm_model.py (no GPU).
//
// For random reads (N, lower case, repeats, windows, G -> N masks; 1 .. 700 bases; several k) it emulates a wave's extraction
// rounds lane by lane -- the exchange rows are arrays -- and checks that
//   * the k-mers of the items, expanded with SkmRoll, are exactly the canonical k-mers update_kmer() counts (trim.cpp:887-931:
//     every window of k valid bases inside the kept string), as a multiset;
//   * no item holds more than w k-mers;
//   * every occurrence of a canonical k-mer -- either strand, any read -- carries the same partition.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <string>
#include <vector>

#include "../faqcs_amd/csrc/faqcs_skm.h"

typedef unsigned long long u64;
struct Item { u64 x, y; };

static uint32_t umin_(uint32_t a, uint32_t b) { return a < b ? a : b; }
static int ffs32(uint32_t v) { return v ? __builtin_ctz(v) + 1 : 0; }

// one read through the extraction rounds of one (emulated) wave; valid_base[i]: what the kernel's classification says after G -> N
static void extract_read(const std::string &seq, const std::vector<uint8_t> &g2n_mask, int a, int n, const SkmGeom &g, uint32_t run,
                         std::vector<Item> &out, uint64_t &total)
{
    const int k = (int)g.k, w = (int)g.w;
    if (n < k) return;
    for (int pc = a;; pc += SKM_ADVANCE) {
        const bool last = !(pc + SKM_PIECE < a + n);
        const int limit = last ? 0x7fffffff : pc + SKM_ADVANCE;
        uint32_t codes[64], nb[64];
        const int st_end = a + n < pc + SKM_PIECE ? a + n : pc + SKM_PIECE;
        for (int lane = 0; lane < 64; ++lane) {
            const int p0 = pc + 4 * lane;
            uint32_t bw = 0;
            if (p0 < a + n)
                for (int j = 0; j < 4; ++j) bw |= (uint32_t)(uint8_t)(p0 + j < (int)seq.size() ? seq[p0 + j] : 'x') << (8 * j); // (bytes past the window: whatever follows)
            uint32_t valid;
            skm_classify4(bw, codes[lane], valid);
            for (int j = 0; j < 4; ++j)
                if (p0 + j < (int)seq.size() && g2n_mask[p0 + j]) valid &= ~(1u << j);
            nb[lane] = 0;
            for (int j = 0; j < 4; ++j) nb[lane] |= (p0 + j < st_end && !((valid >> j) & 1u)) ? 1u << j : 0u;
        }
        uint64_t bad[4] = {0, 0, 0, 0};
        bool has_bad = false;
        for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 4; ++j)
                if ((nb[lane] >> j) & 1u) { bad[j] |= 1ull << lane; has_bad = true; }
        int st_next = pc;
        while (st_next < st_end) {
            const int lo = st_next;
            int nbp = st_end;
            if (has_bad)
                for (int j = 0; j < 4; ++j) {
                    const int rel = lo - pc - j;
                    const int ln = rel <= 0 ? 0 : (rel + 3) >> 2;
                    if (ln < 64) {
                        const uint64_t m = bad[j] >> ln;
                        if (m) { const int cand = pc + 4 * (ln + __builtin_ctzll(m)) + j; nbp = cand < nbp ? cand : nbp; }
                    }
                }
            st_next = nbp + 1;
            if (nbp - lo < k) continue;
            const int sa = lo, sn = nbp - lo;
            // ---- stretch(sa, sn) ----
            uint32_t rowA[80], rowB[288];
            for (int i = 0; i < 80; ++i) rowA[i] = 0xfu;
            for (int i = 0; i < 288; ++i) rowB[i] = 0xfu;
            for (int lane = 0; lane < 64; ++lane) rowA[lane] = codes[lane];
            u64 c_lo[64], c_hi[64];
            uint32_t o[64][4], omin[64][4];
            for (int lane = 0; lane < 64; ++lane) {
                c_lo[lane] = 0; c_hi[lane] = 0;
                for (int i = 0; i < 8; ++i) c_lo[lane] |= (u64)rowA[lane + i] << (8 * i);
                for (int i = 8; i < SKM_LOOK; ++i) c_hi[lane] |= (u64)rowA[lane + i] << (8 * (i - 8));
                if (g.k == 31) skm_mmer_ords15(c_lo[lane], g, o[lane]);
                else for (int j = 0; j < 4; ++j) o[lane][j] = skm_mmer_ord_at(c_lo[lane], j, g);
                if (g.k == 31) { // both forms agree
                    for (int j = 0; j < 4; ++j)
                        if (o[lane][j] != skm_mmer_ord_at(c_lo[lane], j, g)) { fprintf(stderr, "mmer_ords15 != mmer_ord_at\n"); exit(1); }
                }
            }
            if (g.k == 31) {
                for (int lane = 0; lane < 64; ++lane) {
                    const uint32_t *ol = o[lane];
                    const uint32_t p2 = umin_(ol[0], ol[1]), p3 = umin_(p2, ol[2]), m4 = umin_(p3, ol[3]);
                    rowB[lane] = m4; rowB[72 + lane] = ol[0]; rowB[144 + lane] = p2; rowB[216 + lane] = p3;
                }
                for (int lane = 0; lane < 64; ++lane) {
                    const uint32_t *ol = o[lane];
                    const uint32_t p2 = umin_(ol[0], ol[1]), p3 = umin_(p2, ol[2]), m4 = umin_(p3, ol[3]);
                    const uint32_t s2 = umin_(ol[2], ol[3]), s1 = umin_(ol[1], s2);
                    const uint32_t mid = umin_(umin_(rowB[lane + 1], rowB[lane + 2]), rowB[lane + 3]);
                    omin[lane][0] = umin_(umin_(m4, mid), rowB[72 + lane + 4]);
                    omin[lane][1] = umin_(umin_(s1, mid), rowB[144 + lane + 4]);
                    omin[lane][2] = umin_(umin_(s2, mid), rowB[216 + lane + 4]);
                    omin[lane][3] = umin_(umin_(ol[3], mid), rowB[lane + 4]);
                }
            } else {
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 4; ++j) rowB[4 * lane + j] = o[lane][j];
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 4; ++j) {
                        uint32_t mn = o[lane][j];
                        for (int t = 1; t < w; ++t) mn = umin_(mn, rowB[4 * lane + j + t]);
                        omin[lane][j] = mn;
                    }
            }
            const int vlo = sa > pc ? sa : pc;
            int vhi = sa + sn - k + 1;
            vhi = vhi < limit ? vhi : limit;
            uint32_t vb[64], st[64], mask24[64];
            for (int lane = 0; lane < 64; ++lane) {
                const uint32_t prev = lane ? omin[lane - 1][3] : omin[0][3];
                vb[lane] = st[lane] = 0;
                for (int j = 0; j < 4; ++j) {
                    const int p = pc + 4 * lane + j;
                    const bool v = p >= vlo && p < vhi;
                    const bool s = v && (p == vlo || omin[lane][j] != (j ? omin[lane][j - 1] : prev));
                    vb[lane] |= v ? 1u << j : 0u;
                    st[lane] |= s ? 1u << j : 0u;
                }
                total += (uint64_t)__builtin_popcount(vb[lane]);
            }
            auto exchange = [&]() {
                for (int lane = 0; lane < 64; ++lane) rowA[lane] = st[lane] | (~vb[lane] & 0xfu);
                for (int lane = 0; lane < 64; ++lane) {
                    uint32_t m = rowA[lane];
                    for (int i = 1; i <= 5; ++i) m |= rowA[lane + i] << (4 * i);
                    mask24[lane] = m;
                }
            };
            exchange();
            bool any_long = false;
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 4; ++j)
                    any_long |= ((st[lane] >> j) & 1u) && ((mask24[lane] >> (j + 1)) & ((1u << w) - 1u)) == 0u;
            if (any_long) {
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 4; ++j) {
                        const int p = pc + 4 * lane + j;
                        if (((vb[lane] >> j) & 1u) && (uint32_t)(p - vlo) % (uint32_t)w == 0u) st[lane] |= 1u << j;
                    }
                exchange();
            }
            for (int lane = 0; lane < 64; ++lane) {
                uint32_t sbits = st[lane];
                while (sbits) {
                    const uint32_t j = (uint32_t)ffs32(sbits) - 1u;
                    sbits &= sbits - 1u;
                    const uint32_t len = (uint32_t)ffs32((mask24[lane] >> (j + 1u)) | (1u << w));
                    u64 w0, w1;
                    skm_pack(c_lo[lane], c_hi[lane], j, len, skm_part(omin[lane][j]), run, w0, w1);
                    out.push_back(Item{w0, w1});
                }
            }
        }
        if (last) break;
    }
}


// skm_extract16: a row of 16 lanes, 16 positions per lane, k = 31, windows of up to 256 bases; false: the read is left to extract_read
static bool extract_read16(const std::string &seq, const std::vector<uint8_t> &g2n_mask, int a, int n, const SkmGeom &g, uint32_t run,
                           std::vector<Item> &out, uint64_t &total)
{
    if (n < 31) return true;
    uint32_t c0[16], bad[16];
    for (int l = 0; l < 16; ++l) {
        uint32_t valid = 0;
        c0[l] = 0;
        const bool need = 16 * l < n;
        for (int d = 0; d < 4; ++d) {
            uint32_t bw = 0;
            if (need)
                for (int j = 0; j < 4; ++j) { const int p = a + 16 * l + 4 * d + j; bw |= (uint32_t)(uint8_t)(p < (int)seq.size() ? seq[p] : 'x') << (8 * j); }
            uint32_t cd, vd;
            skm_classify4(bw, cd, vd);
            for (int j = 0; j < 4; ++j) { const int p = a + 16 * l + 4 * d + j; if (p < (int)seq.size() && g2n_mask[p]) vd &= ~(1u << j); }
            c0[l] |= cd << (8 * d); valid |= vd << (4 * d);
        }
        int inw = n - 16 * l;
        inw = inw < 0 ? 0 : (inw > 16 ? 16 : inw);
        bad[l] = ((1u << inw) - 1u) & valid; // usable positions
    }
    auto at = [&](const uint32_t *v, int l) -> uint32_t { return l < 16 ? v[l] : 0u; }; // row_shl past the row's end: 0
    uint32_t omin[16][16], pre[16][16], o[16][16];
    for (int l = 0; l < 16; ++l) {
        const uint32_t c1 = at(c0, l + 1);
        const uint32_t r0 = skm_rev2_32((c1 << 4) | (c0[l] >> 28)), r1 = skm_rev2_32(c0[l] << 4);
        for (int j = 0; j < 16; ++j) {
            const u64 cc = ((u64)c1 << 32) | c0[l], rr = ((u64)r1 << 32) | r0;
            const uint32_t fwd = (uint32_t)(cc >> (2 * j)) & SKM_M30;
            const uint32_t rc = ((uint32_t)(rr >> (2 * (15 - j))) & SKM_M30) ^ 0x2AAAAAAAu;
            o[l][j] = skm_ord(fwd < rc ? fwd : rc, g);
            if (o[l][j] != skm_mmer_ord_at(cc, j, g)) { fprintf(stderr, "extract16: ord differs from mmer_ord_at\n"); exit(1); }
        }
        pre[l][0] = o[l][0];
        for (int j = 1; j < 16; ++j) pre[l][j] = umin_(pre[l][j - 1], o[l][j]);
    }
    bool deferred = false;
    uint32_t st[16], vb[16];
    u64 cont64[16];
    uint32_t cont[16];
    for (int l = 0; l < 16; ++l) {
        uint32_t suf = o[l][15];
        for (int j = 15; j >= 0; --j) { suf = umin_(suf, o[l][j]); omin[l][j] = umin_(suf, l + 1 < 16 ? pre[l + 1][j] : 0u); }
    }
    for (int l = 0; l < 16; ++l) {
        u64 z = ~((u64)bad[l] | ((u64)at(bad, l + 1) << 16) | ((u64)at(bad, l + 2) << 32));
        z |= z >> 1; z |= z >> 2; z |= z >> 4; z |= z >> 8;
        z |= z >> 15;
        vb[l] = 0xffffu & ~(uint32_t)z;
    }
    for (int l = 0; l < 16; ++l) {
        const uint32_t prev = l ? omin[l - 1][15] : 0u, vprev = l ? vb[l - 1] : 0u;
        st[l] = ~((vb[l] << 1) | (vprev >> 15));
        for (int j = 0; j < 16; ++j) st[l] |= (omin[l][j] != (j ? omin[l][j - 1] : prev)) ? 1u << j : 0u;
        st[l] &= vb[l];
        cont[l] = vb[l] & ~st[l];
    }
    for (int l = 0; l < 16; ++l) {
        cont64[l] = (u64)cont[l] | ((u64)at(cont, l + 1) << 16) | ((u64)at(cont, l + 2) << 32);
        u64 x = cont64[l];
        x &= x >> 1; x &= x >> 2; x &= x >> 4; x &= x >> 8;
        x &= cont64[l] >> 16;
        const bool too_long = ((uint32_t)(x >> 1) & st[l]) != 0u;
        if (too_long) deferred = true;
    }
    if (deferred) return false;
    for (int l = 0; l < 16; ++l) {
        total += (uint64_t)__builtin_popcount(vb[l]);
        const u64 brk64 = ~cont64[l];
        const uint32_t c1 = at(c0, l + 1), c2 = at(c0, l + 2), c3 = at(c0, l + 3);
        uint32_t sb = st[l];
        while (sb) {
            const uint32_t j = (uint32_t)ffs32(sb) - 1u;
            sb &= sb - 1u;
            const uint32_t len = (uint32_t)__builtin_ffsll((long long)(brk64 >> (j + 1u)));
            const uint32_t sh = 2u * j;
            auto align = [](uint32_t hi, uint32_t lo, uint32_t s_) -> uint32_t { return (uint32_t)((((u64)hi << 32) | lo) >> (s_ & 31u)); };
            const uint32_t d0 = align(c1, c0[l], sh), d1 = align(c2, c1, sh), d2 = align(c3, c2, sh) & SKM_M30;
            Item it;
            it.x = ((u64)d1 << 32) | d0;
            it.y = (u64)d2 | ((u64)(len - 1u) << SKM_NK_SHIFT) | ((u64)skm_part(omin[l][j]) << SKM_PART_SHIFT) | ((u64)run << SKM_RUN_SHIFT);
            out.push_back(it);
        }
    }
    return true;
}

static uint32_t code_of(char c)
{
    switch (c | 0x20) { case 'a': return 0; case 'c': return 1; case 't': return 2; case 'g': return 3; default: return 4; }
}
// the canonical k-mers update_kmer() counts over the string [a, a + n) (trim.cpp:887-931), in this repository's key encoding
static void reference_keys(const std::string &seq, const std::vector<uint8_t> &g2n_mask, int a, int n, const SkmGeom &g, std::vector<u64> &keys)
{
    const int k = (int)g.k;
    int word_len = 0;
    for (int i = a; i < a + n; ++i) {
        const uint32_t c = g2n_mask[i] ? 4u : code_of(seq[i]);
        if (c > 3) { word_len = 0; continue; }
        ++word_len;
        if (word_len >= k) {
            u64 fwd = 0, rc = 0;
            for (int t = 0; t < k; ++t) {
                const u64 ct = code_of(seq[i - k + 1 + t]);
                fwd |= ct << (2 * t);
                rc |= (ct ^ 2ull) << (2 * (k - 1 - t));
            }
            keys.push_back(fwd < rc ? fwd : rc);
        }
    }
}

static uint64_t rng_state = 88172645463325252ull;
static uint32_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (uint32_t)(rng_state >> 11); }

static std::string revcomp(const std::string &s)
{
    std::string r(s.rbegin(), s.rend());
    for (auto &c : r) {
        switch (c) { case 'A': c = 'T'; break; case 'T': c = 'A'; break; case 'C': c = 'G'; break; case 'G': c = 'C'; break;
                     case 'a': c = 't'; break; case 't': c = 'a'; break; case 'c': c = 'g'; break; case 'g': c = 'c'; break; default: break; }
    }
    return r;
}

int main(int argc, char **argv)
{
    const int n_reads = argc > 1 ? atoi(argv[1]) : 3000;
    if (argc > 2) rng_state ^= strtoull(argv[2], nullptr, 0) * 0x9E3779B97F4A7C15ull;
    const uint32_t ks[] = {31, 31, 31, 2, 3, 5, 11, 15, 16, 17, 20, 25, 30};
    uint64_t n_items = 0, n_keys = 0;
    for (uint32_t k : ks) {
        const uint64_t items0 = n_items, keys0 = n_keys;
        const SkmGeom g = skm_geom(k);
        // ord must be a bijection of [0, 4^m) for small m (checked exhaustively), and stays inside it for m = 15
        if (g.m <= 10) {
            std::vector<uint8_t> seen((size_t)g.mmask + 1, 0);
            for (uint32_t x = 0; x <= g.mmask; ++x) {
                const uint32_t o = skm_ord(x, g);
                if (o > g.mmask || seen[o]) { fprintf(stderr, "ord is not a bijection for m=%u\n", g.m); return 1; }
                seen[o] = 1;
            }
        }
        std::map<u64, uint32_t> part_of_key;
        std::string genome;
        for (int i = 0; i < 20000; ++i) genome.push_back("ACGT"[rnd() & 3]);
        for (int r = 0; r < n_reads; ++r) {
            std::string s;
            const int kind = rnd() % 10;
            int len = kind == 9 ? 1 + rnd() % 700 : (kind == 8 ? 1 + rnd() % 40 : 100 + rnd() % 157);
            if (kind < 5) { // a window of the genome, either strand, with a few substitutions: repeated k-mers across reads
                const int at = rnd() % (genome.size() - len);
                s = genome.substr(at, len);
                for (auto &c : s) if (rnd() % 200 == 0) c = "ACGT"[rnd() & 3];
                if (rnd() & 1) s = revcomp(s);
            } else if (kind == 5) { // low complexity: homopolymers and short tandem repeats (equal minimizers over long stretches)
                while ((int)s.size() < len) {
                    const int period = 1 + rnd() % 6, reps = 5 + rnd() % 60;
                    std::string unit;
                    for (int i = 0; i < period; ++i) unit.push_back("ACGT"[rnd() & 3]);
                    for (int i = 0; i < reps; ++i) s += unit;
                }
                s.resize(len);
            } else {
                for (int i = 0; i < len; ++i) s.push_back("ACGT"[rnd() & 3]);
            }
            for (auto &c : s) {
                const uint32_t x = rnd() % 400;
                if (x == 0) c = 'N';
                else if (x == 1) c = 'n';
                else if (x == 2) c = '.';
                else if (x < 8) c = (char)(c | 0x20);
            }
            if (kind == 7) for (int i = 0; i < 3 && len > 0; ++i) { const int at = rnd() % len; for (int t = at; t < len && t < at + (int)(rnd() % 5); ++t) s[t] = 'N'; }
            std::vector<uint8_t> mask(s.size() + 8, 0);
            if (rnd() % 4 == 0) for (size_t i = 0; i < s.size(); ++i) if (s[i] == 'G' && rnd() % 10 == 0) mask[i] = 1;
            int a = 0, n = len;
            if (rnd() % 3 == 0 && len > 2) { a = rnd() % (len / 2); n = 1 + rnd() % (len - a); }
            s += "ACGTNACGTACGTTTTT"; // (whatever follows the read in the arena)
            std::vector<Item> items;
            uint64_t total = 0;
            static uint64_t n16 = 0, n16_deferred = 0;
            if (k == 31 && n <= 256 && (r & 1)) { // the 16-positions-per-lane kernel's arithmetic, with its fall-back
                ++n16;
                if (!extract_read16(s, mask, a, n, g, r & 1023, items, total)) { ++n16_deferred; items.clear(); total = 0; extract_read(s, mask, a, n, g, r & 1023, items, total); }
                if (r == n_reads - 1 || r == n_reads - 2) printf("extract16: %llu reads, %llu left to the general kernel\n", (u64)n16, (u64)n16_deferred);
            } else
                extract_read(s, mask, a, n, g, r & 1023, items, total);
            std::vector<u64> got, want;
            for (const Item &it : items) {
                const uint32_t nk = skm_item_kmers(it.y), part = skm_item_part(it.y);
                if (nk > g.w || skm_item_run(it.y) != (uint32_t)(r & 1023)) { fprintf(stderr, "k=%u: item with %u k-mers (w = %u) / run field\n", k, nk, g.w); return 1; }
                SkmRoll roll = skm_roll_begin(it.x, it.y, g);
                for (uint32_t j = 0; j < nk; ++j) {
                    const u64 key = skm_roll_key(roll);
                    got.push_back(key);
                    auto f = part_of_key.find(key);
                    if (f == part_of_key.end()) part_of_key[key] = part;
                    else if (f->second != part) { fprintf(stderr, "k=%u read %d: key %llx in partitions %u and %u\n", k, r, key, f->second, part); return 1; }
                    skm_roll_next(roll, g);
                }
            }
            reference_keys(s, mask, a, n, g, want);
            if (total != want.size()) { fprintf(stderr, "k=%u read %d (len %d, window %d+%d): total %llu != %zu\n", k, r, len, a, n, (u64)total, want.size()); return 1; }
            std::sort(got.begin(), got.end());
            std::sort(want.begin(), want.end());
            if (got != want) { fprintf(stderr, "k=%u read %d (len %d, window %d+%d): %zu keys from %zu items, want %zu\n", k, r, len, a, n, got.size(), items.size(), want.size()); return 1; }
            n_items += items.size(); n_keys += want.size();
        }
        printf("k=%u (m=%u, w=%u): %llu k-mers in %llu items (%.2f per item)\n", k, g.m, g.w, (u64)(n_keys - keys0), (u64)(n_items - items0),
               (double)(n_keys - keys0) / (double)(n_items - items0 ? n_items - items0 : 1));
        // the partitions in use are spread: no partition holds more than a small share of the distinct keys of random data
        if (k == 31) {
            std::map<uint32_t, uint32_t> per;
            for (auto &kv : part_of_key) ++per[kv.second];
            uint32_t mx = 0;
            for (auto &kv : per) mx = std::max(mx, kv.second);
            if ((double)mx > 0.02 * (double)part_of_key.size() + 64) { fprintf(stderr, "k=31: one partition holds %u of %zu keys\n", mx, part_of_key.size()); return 1; }
        }
    }
    printf("ok: %llu k-mers in %llu items (%.2f per item)\n", (u64)n_keys, (u64)n_items, (double)n_keys / (double)n_items);
    return 0;
}
