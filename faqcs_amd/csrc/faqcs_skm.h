// faqcs_skm.h -- super-k-mers: the arithmetic of the k-mer pipeline (faqcs_kmer_skm_kernel.hip), written so that the SAME text
// compiles for the host: tests/skm_model.cpp runs these functions over emulated lanes and checks, without a GPU, that the items an
// extraction round produces expand to exactly the canonical k-mers update_kmer() (trim.cpp:887-931) counts, and that every
// occurrence of a canonical k-mer lands in the same partition.
//
// Why.  Through round 4 every k-mer occurrence travelled as an 8-byte item through two scatter passes: 61 bytes of HBM traffic per
// occurrence against 16 algorithmic.  Consecutive k-mers of a read overlap in k - 1 bases, and consecutive k-mers that share their
// MINIMIZER -- the smallest (in a pseudo-random order) canonical m-mer inside the k-mer -- can travel together if the partition
// of the key space is a function of the minimizer alone (KMC 2 / Gerbil): one 16-byte item carries up to w = k - m + 1 = 17
// consecutive 31-mers as 2 w + 2 (k - 1) bits of bases, about 1.8 bytes per occurrence.  The k-mers are expanded only inside
// the workgroup that counts a partition, in registers.
//
//   m        = min(k, 15)         (30-bit canonical m-mers; 4^15 / 2 of them spread the 65 536 partitions evenly)
//   ord(x)   = a bijection of [0, 4^m): the pseudo-random order; a bijection, so that two m-mers tie only when they are EQUAL --
//              a k-mer and its reverse complement see the same set of canonical m-mers, hence the same smallest ord, hence the
//              same partition, whichever occurrence of a repeated m-mer a scan happens to meet first
//   part(o)  = 19 bits mixed out of the smallest ord (the minimum itself is biased towards small values).  The top 8 bits are the level-1
//              bucket, the top 16 the partition of a group that is flushed into the table while the pass goes on, and the top 16 + F
//              (F = 0 .. 3, from the table size) the partition of a pass that is counted in one piece at its end (round 6)
//   run      = maximal stretch of consecutive valid k-mers with equal smallest ord, cut at w k-mers
//
// Item (16 bytes):  w0 = bases 0 .. 31 (2 bits each, base i at bits 2 i; A 0, C 1, T 2, G 3: complement = code ^ 2)
//                   w1 = bases 32 .. 46 in bits 0 .. 29 | (k-mers - 1) << 30 (5 bits) | partition << 35 (19 bits) | run or epoch << 54 (10 bits)
// k-mer j of an item = bases j .. j + k - 1.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define SKM_HD __host__ __device__ __forceinline__
#else
#define SKM_HD static inline
#endif

typedef unsigned long long skm_u64;

enum {
    SKM_M_MAX = 15,     // bases of a minimizer
    SKM_W_MAX = 17,     // k-mers of an item (k = 31)
    SKM_LOOK = 14,      // lanes a lane looks ahead in an extraction round (4 positions each: 56 >= 3 + 17 + 30 positions)
    SKM_PIECE = 256,    // positions of a round (64 lanes x 4)
    SKM_ADVANCE = 200,  // k-mer starts a piece of a longer read takes (the others need bases past the piece: next piece)
    SKM_NK_SHIFT = 30, SKM_PART_SHIFT = 35, SKM_PART_BITS = 19, SKM_RUN_SHIFT = 54, SKM_RUN_BITS = 10
};
#define SKM_M30 0x3fffffffu
#define SKM_RUN_MASK ((skm_u64)((1u << SKM_RUN_BITS) - 1u) << SKM_RUN_SHIFT)

struct SkmGeom { uint32_t k, m, w, mmask; skm_u64 kmask2; };
SKM_HD SkmGeom skm_geom(const uint32_t k)
{
    SkmGeom g;
    g.k = k; g.m = k < (uint32_t)SKM_M_MAX ? k : (uint32_t)SKM_M_MAX; g.w = k - g.m + 1u;
    g.mmask = (uint32_t)((1ull << (2u * g.m)) - 1ull);
    g.kmask2 = (1ull << (2u * k)) - 1ull;
    return g;
}

SKM_HD uint32_t skm_brev32(uint32_t v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __brev(v);
#else
    v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
    v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
    v = ((v >> 4) & 0x0f0f0f0fu) | ((v & 0x0f0f0f0fu) << 4);
    v = ((v >> 8) & 0x00ff00ffu) | ((v & 0x00ff00ffu) << 8);
    return (v >> 16) | (v << 16);
#endif
}
// the 16 two-bit groups of a dword in reverse order
SKM_HD uint32_t skm_rev2_32(uint32_t v)
{
    v = skm_brev32(v);
    return ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
}
// the 32 two-bit groups of a 64-bit word in reverse order
SKM_HD skm_u64 skm_rev2_64(const skm_u64 v) { return ((skm_u64)skm_rev2_32((uint32_t)v) << 32) | (skm_u64)skm_rev2_32((uint32_t)(v >> 32)); }

// Four bases (the bytes of dw, base j in byte j) -> their 2-bit codes (bits 2 j) and "is one of ACGTacgt" flags (bit j).
// code = (byte >> 1) & 3 maps A C T G (either case) to 0 1 2 3; a byte is a base iff it equals, lower-cased, the letter of its code.
SKM_HD void skm_classify4(const uint32_t dw, uint32_t &codes, uint32_t &valid)
{
    const uint32_t x = (dw >> 1) & 0x03030303u;
    codes = (x | (x >> 6) | (x >> 12) | (x >> 18)) & 0xffu;
    const uint32_t low = dw | 0x20202020u;
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t want = __builtin_amdgcn_perm(0u, 0x67746361u, x); // byte j = "actg"[code j]
#else
    uint32_t want = 0;
    for (int j = 0; j < 4; ++j) want |= ((0x67746361u >> (8u * ((x >> (8 * j)) & 3u))) & 0xffu) << (8 * j);
#endif
    const uint32_t d = low ^ want;                                             // a zero byte: a base
    const uint32_t z = ~(((d & 0x7f7f7f7fu) + 0x7f7f7f7fu) | d) & 0x80808080u; // bit 7 of every zero byte, exactly
    valid = ((z >> 7) | (z >> 14) | (z >> 21) | (z >> 28)) & 0xfu;
}

// the pseudo-random order of the canonical m-mers: a bijection of [0, 4^m) (xor-shifts and an odd multiplier modulo 4^m)
SKM_HD uint32_t skm_ord(uint32_t x, const SkmGeom &g)
{
    x ^= x >> g.m;
    x = (x * 0x9E3779B1u) & g.mmask;
    x ^= x >> (g.m - 1u);
    return x;
}
// partition of the key space a run belongs to: 19 bits mixed out of its smallest ord
SKM_HD uint32_t skm_part(const uint32_t ord_min) { return (uint32_t)(ord_min * 0x85EBCA6Bu) >> (32 - SKM_PART_BITS); }

// ord of the canonical m-mer that starts at base j (0 .. 3) of the window c (base i at bits 2 i; bases 0 .. 17 are looked at)
SKM_HD uint32_t skm_mmer_ord_at(const skm_u64 c, const int j, const SkmGeom &g)
{
    const uint32_t fwd = (uint32_t)(c >> (2 * j)) & g.mmask;
    const uint32_t rc = (skm_rev2_32(fwd) >> (32u - 2u * g.m)) ^ (0xAAAAAAAAu & g.mmask);
    return skm_ord(fwd < rc ? fwd : rc, g);
}
// the same for the four m-mers of a lane at once, m = 15: ONE reversal of the 18 bases they span
SKM_HD void skm_mmer_ords15(const skm_u64 c, const SkmGeom &g, uint32_t (&o)[4])
{
    const uint32_t lo = (uint32_t)c, hi = (uint32_t)(c >> 32);
    // bases 0 .. 17 moved to the top of a 64-bit word (base i -> group 14 + i), then all 32 groups reversed: base i -> group 17 - i
    const skm_u64 top = ((skm_u64)((lo >> 4) | (hi << 28)) << 32) | (skm_u64)(lo << 28);
    const skm_u64 rev = skm_rev2_64(top);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t fwd = (uint32_t)(c >> (2 * j)) & SKM_M30;
        const uint32_t rc = ((uint32_t)(rev >> (2 * (3 - j))) & SKM_M30) ^ 0x2AAAAAAAu; // bases j .. j + 14 reversed = groups 3 - j .. 17 - j
        o[j] = skm_ord(fwd < rc ? fwd : rc, g);
    }
}

SKM_HD skm_u64 skm_mix62(skm_u64 x) // (kmer_mix62 of faqcs_kmer.h: a bijection of [0, 2^62))
{
    x ^= x >> 31; x = (x * 0x9E3779B97F4A7C15ull) & ((1ull << 62) - 1ull);
    x ^= x >> 29;
    return x;
}

// ---- an item ----------------------------------------------------------------------------------------------------------------
// c_lo : c_hi = the bases at and after a lane's first position (c_hi: bases 32 .. 55), j = where the run starts (0 .. 3)
SKM_HD void skm_pack(const skm_u64 c_lo, const skm_u64 c_hi, const uint32_t j, const uint32_t n_kmers, const uint32_t part, const uint32_t run,
                     skm_u64 &w0, skm_u64 &w1)
{
    const uint32_t s = 2u * j;
    w0 = s ? (c_lo >> s) | (c_hi << (64u - s)) : c_lo;
    w1 = ((c_hi >> s) & (skm_u64)SKM_M30) | ((skm_u64)(n_kmers - 1u) << SKM_NK_SHIFT) | ((skm_u64)part << SKM_PART_SHIFT) | ((skm_u64)run << SKM_RUN_SHIFT);
}
SKM_HD uint32_t skm_item_kmers(const skm_u64 w1) { return ((uint32_t)(w1 >> SKM_NK_SHIFT) & 31u) + 1u; }
SKM_HD uint32_t skm_item_part(const skm_u64 w1) { return (uint32_t)(w1 >> SKM_PART_SHIFT) & ((1u << SKM_PART_BITS) - 1u); } // all 19 bits
SKM_HD uint32_t skm_item_bucket(const skm_u64 w1) { return (uint32_t)(w1 >> (SKM_PART_SHIFT + SKM_PART_BITS - 8)) & 0xffu; }    // level-1 bucket
SKM_HD uint32_t skm_item_p16(const skm_u64 w1) { return (uint32_t)(w1 >> (SKM_PART_SHIFT + SKM_PART_BITS - 16)) & 0xffffu; }    // partition of a table-mode group
SKM_HD skm_u64 skm_item_with_part(const skm_u64 w1, const uint32_t part) // (owner side of the exchange: the partition becomes the owner's local one)
{
    return (w1 & ~((skm_u64)((1u << SKM_PART_BITS) - 1u) << SKM_PART_SHIFT)) | ((skm_u64)part << SKM_PART_SHIFT);
}
SKM_HD uint32_t skm_item_run(const skm_u64 w1) { return (uint32_t)(w1 >> SKM_RUN_SHIFT); }

// the canonical keys of an item, one after the other: fwd >>= one base, rc <<= one base
struct SkmRoll {
    skm_u64 fwd, rc;
    uint32_t rest; // the bases behind the current k-mer, next one in bits 0 .. 1
};
SKM_HD SkmRoll skm_roll_begin(const skm_u64 w0, const skm_u64 w1, const SkmGeom &g)
{
    SkmRoll r;
    r.fwd = w0 & g.kmask2;
    r.rc = (skm_rev2_64(r.fwd) >> (64u - 2u * g.k)) ^ (0xAAAAAAAAAAAAAAAAull & g.kmask2);
    // bases k, k + 1, ...: at most w - 1 <= 16 of them are ever shifted in
    const skm_u64 hi = (w1 & (skm_u64)SKM_M30);
    r.rest = g.k < 32u ? (uint32_t)((w0 >> (2u * g.k)) | (hi << (64u - 2u * g.k))) : (uint32_t)hi;
    return r;
}
SKM_HD skm_u64 skm_roll_key(const SkmRoll &r) { return r.fwd < r.rc ? r.fwd : r.rc; }
SKM_HD void skm_roll_next(SkmRoll &r, const SkmGeom &g)
{
    const skm_u64 b = r.rest & 3u;
    r.rest >>= 2;
    r.fwd = (r.fwd >> 2) | (b << (2u * (g.k - 1u)));
    r.rc = ((r.rc << 2) | (b ^ 2u)) & g.kmask2;
}
